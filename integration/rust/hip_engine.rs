//! `src/hip_engine.rs` (new file in rocoder): the `extern "C"` surface of `librocoder_hip.so`
//! (`include/rocoder_hip.h`, ABI version 5) and the small safe wrappers the rest of the binding uses.
//!
//! Nothing here was compiled in the build container (no rustc there). What IS checked without rustc:
//! `tests/test_cabi_host.py::test_integration_md_rust_stub_matches_the_header` parses the `#[repr(C)]`
//! structs and the `extern "C"` block out of THIS FILE, lays the structs out by the repr(C) rules and holds
//! them to the table the C compiler printed (`tests/golden/abi_layout.json`), and holds every declared
//! function - name, argument count, every argument and return type - to the header's prototypes.
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

/// `rc_freq_kernel`: the C-ABI shape of the README's `apply(elapsed_ms, Vec<(f32,f32)>) -> Vec<(f32,f32)>`
/// (README.md:106-112, resolved and called at src/fft.rs:93-95). All N bins, interleaved (re, im), Unix-epoch
/// milliseconds; a non-zero return is the equivalent of a panic (src/fft.rs:100-106).
pub type RcFreqKernel = Option<
    unsafe extern "C" fn(time_ms: u64, in_reim: *const f32, out_reim: *mut f32, n_bins: usize,
                         user: *mut c_void) -> c_int>;

pub const RC_ABI_VERSION: c_int = 5;
pub const RC_OK: c_int = 0;
pub const RC_WOULD_BLOCK: c_int = 1;
pub const RC_DK_NONE: u32 = 0;
pub const RC_DK_GAIN: u32 = 1;
pub const RC_DK_BAND: u32 = 2;
pub const RC_DK_SHIFT: u32 = 3;

/// `rc_config`: the arguments of `Stretcher::new` (src/stretcher.rs:30-39) for all channels of a job.
#[repr(C)]
pub struct RcConfig {
    pub struct_size: u32, pub window_len: u32, pub factor: f32, pub amplitude: f32,
    pub pitch_multiple: i32, pub sample_rate: u32, pub channels: u32, pub buffer_secs: f32,
    pub seed: u64, pub device: i32, pub max_batch_hops: u32, pub window: *const f32,
    pub kernel: RcFreqKernel, pub kernel_user: *mut c_void, pub kernel_time_ms: u64,
    // since ABI version 2 (3 added rc_shard_plan / rc_multi_*, 4 the window view, rc_multi_set_staging and
    // rc_calib_valu, 5 rc_host_alloc / rc_host_free - no layout change since 2):
    pub kernel_threads: u32,      // 0/1 = the single DSP thread and its call order
    pub device_kernel: u32,       // RC_DK_NONE / GAIN / BAND / SHIFT: curated kernels that run on the GPU
    pub dk_gain: f32, pub dk_gain_outside: f32, pub dk_lo_bin: u32, pub dk_hi_bin: u32, pub dk_shift_bins: i32,
}
/// `rc_params`: the values `Stretcher::new` derives (src/stretcher.rs:40-56).
#[repr(C)]
pub struct RcParams {
    pub window_len: u32, pub half_window_len: u32, pub samples_needed_per_window: u64, pub sample_step_len: u32,
    pub hops_per_window: u32, pub window_out_len: u32, pub corrected_amp_factor: f32, pub pitch_shifted_factor: f32,
}
#[repr(C)] pub struct RcEngine { _private: [u8; 0] }
#[repr(C)] pub struct RcMulti { _private: [u8; 0] }
#[repr(C)]
pub struct RcShard { pub device_index: u32, pub ch_first: u32, pub ch_count: u32,
                     pub win_first: u64, pub win_count: u64 }                   // 32 bytes, win_first at 16

#[link(name = "rocoder_hip")]
extern "C" {
    pub fn rc_last_error() -> *const c_char;
    pub fn rc_abi_version() -> c_int;
    pub fn rc_kernel_id() -> *const c_char;
    pub fn rc_device_count() -> c_int;
    pub fn rc_derive_params(cfg: *const RcConfig, out: *mut RcParams) -> c_int;
    pub fn rc_offline_output_len(cfg: *const RcConfig, in_len: usize) -> usize;
    pub fn rc_phase_key(seed: u64, channel: u32, hop: u64) -> u64;
    pub fn rc_phase_hash(key: u64, counter: u32) -> u32;
    pub fn rc_phase_theta(key: u64, bin: u32, n_bins: u32) -> f32;
    pub fn rc_engine_create(cfg: *const RcConfig, out: *mut *mut RcEngine) -> c_int;
    pub fn rc_engine_destroy(e: *mut RcEngine);
    pub fn rc_engine_get_params(e: *const RcEngine, out: *mut RcParams) -> c_int;
    pub fn rc_engine_push_input(e: *mut RcEngine, channel: u32, samples: *const f32, n: usize) -> c_int;
    pub fn rc_engine_close_input(e: *mut RcEngine, channel: u32) -> c_int;
    pub fn rc_engine_next_window(e: *mut RcEngine, channel: u32, out: *mut f32, out_cap: usize,
                                 n_out: *mut usize) -> c_int;       // 0 ok, 1 would block
    pub fn rc_engine_next_window_view(e: *mut RcEngine, channel: u32, window: *mut *const f32,
                                      n_out: *mut usize) -> c_int;  // the same hand-out, no copy
    pub fn rc_engine_is_done(e: *const RcEngine, channel: u32) -> c_int;
    pub fn rc_engine_channel_bound(e: *const RcEngine) -> usize;
    pub fn rc_engine_stretch_host(e: *mut RcEngine, inp: *const *const f32, in_len: usize,
                                  out: *const *mut f32, out_cap: usize, out_len: *mut usize) -> c_int;
    pub fn rc_host_alloc(bytes: usize, out: *mut *mut c_void) -> c_int;   // page-locked rows: no staging copy
    pub fn rc_host_free(p: *mut c_void) -> c_int;
    pub fn rc_engine_stretch_device(e: *mut RcEngine, d_in: *const f32, in_stride: usize, in_len: usize,
                                    d_out: *mut f32, out_stride: usize, out_cap: usize, out_len: *mut usize,
                                    hip_stream: *mut c_void) -> c_int;
    pub fn rc_engine_stretch_device_range(e: *mut RcEngine, d_in: *const f32, in_stride: usize, in_len: usize,
                                          ch_first: u32, ch_count: u32, win_first: u64, win_count: u64,
                                          d_out: *mut f32, out_stride: usize, out_cap: usize,
                                          hip_stream: *mut c_void) -> c_int;
    pub fn rc_engine_synchronize(e: *mut RcEngine) -> c_int;
    pub fn rc_engine_last_kernel_stats(e: *mut RcEngine, kernel_ms: *mut f32, hops: *mut u64,
                                       launches: *mut u32) -> c_int;
    pub fn rc_engine_kernel_times(e: *mut RcEngine, ms: *mut f32, cap: usize, n_out: *mut usize) -> c_int;
    pub fn rc_engine_forward_fft(e: *mut RcEngine, samples: *const f32, out_reim: *mut f32) -> c_int;
    pub fn rc_engine_resynth(e: *mut RcEngine, channel: u32, hop: u64, samples: *const f32, out: *mut f32) -> c_int;
    // one process, several GPUs (`-o` mode; all channels are built in one process: src/main.rs:133-155)
    pub fn rc_shard_plan(channels: u32, total_windows: u64, n_devices: u32, out: *mut RcShard, cap: usize) -> usize;
    pub fn rc_multi_create(cfg: *const RcConfig, device_ids: *const i32, n_devices: u32,
                           out: *mut *mut RcMulti) -> c_int;
    pub fn rc_multi_destroy(m: *mut RcMulti);
    pub fn rc_multi_device_count(m: *const RcMulti) -> u32;
    pub fn rc_multi_set_staging(m: *mut RcMulti, force: c_int) -> c_int;
    pub fn rc_multi_stretch_host(m: *mut RcMulti, inp: *const *const f32, in_len: usize,
                                 out: *const *mut f32, out_cap: usize, out_len: *mut usize) -> c_int;
    pub fn rc_multi_stretch_device(m: *mut RcMulti, root: u32, d_in: *const f32, in_stride: usize, in_len: usize,
                                   d_out: *mut f32, out_stride: usize, out_cap: usize, out_len: *mut usize,
                                   hip_stream: *mut c_void) -> c_int;
    pub fn rc_calib_valu(device: c_int, hip_stream: *mut c_void, launches: u32, ms_per_launch: *mut f32,
                         ns_per_inst: *mut f32) -> c_int;
}

// ---------------------------------------------------------------------------------------------------------------
// safe-ish helpers shared by stretcher_hip.rs

/// The library's thread-local description of the last failure on this thread.
pub fn last_error() -> String {
    unsafe {
        let p = rc_last_error();
        if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
    }
}

/// Negative status codes are the cases in which the reference asserts, panics or never terminates
/// (src/stretcher.rs:40,69); the binding panics with the library's message, as `unwrap()` does there.
pub fn check(rc: c_int) -> c_int {
    if rc < 0 {
        panic!("rocoder_hip: status {}: {}", rc, last_error());
    }
    rc
}

/// Owner of one `rc_engine`. The C-ABI asks for one thread at a time per handle; `Stretcher` serialises its
/// calls through a `Mutex<EngineHandle>` (main builds the stretchers, the processor thread calls them).
pub struct EngineHandle(pub *mut RcEngine);
unsafe impl Send for EngineHandle {}
impl Drop for EngineHandle {
    fn drop(&mut self) {
        if !self.0.is_null() {
            unsafe { rc_engine_destroy(self.0) };
        }
    }
}

/// A config with the ABI guard filled in and everything optional at its default.
pub fn default_config() -> RcConfig {
    RcConfig {
        struct_size: std::mem::size_of::<RcConfig>() as u32, window_len: 16384, factor: 1.0, amplitude: 1.0,
        pitch_multiple: 1, sample_rate: 44100, channels: 1, buffer_secs: 1.0,
        seed: 0, device: 0, max_batch_hops: 0, window: std::ptr::null(),
        kernel: None, kernel_user: std::ptr::null_mut(), kernel_time_ms: 0,
        kernel_threads: 0, device_kernel: RC_DK_NONE,
        dk_gain: 1.0, dk_gain_outside: 1.0, dk_lo_bin: 0, dk_hi_bin: 0, dk_shift_bins: 0,
    }
}

/// Page-locked rows for the `-o` fast path (`rc_engine_stretch_host`): the DMA engines read / write them directly,
/// 13.5 Gsamples/s at BASELINE C2 against 10-12 through pageable `Vec<f32>` rows.
pub struct PinnedBuf { ptr: *mut f32, len: usize }
unsafe impl Send for PinnedBuf {}
impl PinnedBuf {
    pub fn new(len: usize) -> Self {
        let mut p: *mut c_void = std::ptr::null_mut();
        check(unsafe { rc_host_alloc(len * std::mem::size_of::<f32>(), &mut p) });
        PinnedBuf { ptr: p as *mut f32, len }
    }
    pub fn len(&self) -> usize { self.len }
    pub fn is_empty(&self) -> bool { self.len == 0 }
    pub fn as_ptr(&self) -> *const f32 { self.ptr }
    pub fn as_mut_ptr(&mut self) -> *mut f32 { self.ptr }
    pub fn as_slice(&self) -> &[f32] { unsafe { std::slice::from_raw_parts(self.ptr, self.len) } }
    pub fn as_mut_slice(&mut self) -> &mut [f32] { unsafe { std::slice::from_raw_parts_mut(self.ptr, self.len) } }
}
impl Drop for PinnedBuf {
    fn drop(&mut self) {
        unsafe { rc_host_free(self.ptr as *mut c_void) };
    }
}
