//! `src/stretcher.rs` over the gfx950 engine: the SAME public surface as the reference's `Stretcher`
//! (`new` with its eight arguments, `next_window`, `is_done`, `channel_bound`, the public `spec` field), so
//! `main` (src/main.rs:133-155) and `StretcherProcessor` (src/stretcher_processor.rs:26-89) compile unchanged.
//! A maintainer's steps: copy `hip_engine.rs`, this file (as `stretcher.rs`, or next to it under a cargo
//! feature), `kernel_trampoline.rs` and `build.rs` into the crate and add `pub mod hip_engine; pub mod
//! kernel_trampoline;` to src/lib.rs. `ReFFT`, `rustfft`, `rand` and `slice-deque` leave this path.
//!
//! One engine serves all channels of a job. The reference builds one `Stretcher` per channel, one after the
//! other, with identical arguments (src/main.rs:133-153); the first `new` of such a group creates the engine
//! for `spec.channels` channels and each later `new` with the same arguments takes the next channel index.
//! Not compiled in the build container (no rustc); `hip_engine.rs` is held to the C header by a CPU test.
use crate::audio::AudioSpec;
use crate::hip_engine::*;
use crate::kernel_trampoline::KernelStack;
use crossbeam_channel::Receiver;
use std::path::PathBuf;
use std::sync::{Arc, Mutex};
use std::time::Duration;

/// What makes two `Stretcher::new` calls part of one job.
#[derive(Clone, PartialEq)]
struct JobKey {
    window: Vec<u32>, // the window's bit patterns (f32 is not Eq)
    factor: u32,
    amplitude: u32,
    pitch_multiple: i8,
    sample_rate: u32,
    channels: u16,
    buffer_secs: u32,
    kernel_src: Option<PathBuf>,
}

struct Job {
    handle: Mutex<EngineHandle>,
    // the hot-swapped kernel libraries of this job (src/fft.rs:21,76-108); boxed: the engine keeps its address
    _kernels: Option<Box<KernelStack>>,
    window_len: usize,
}

struct Pending {
    key: JobKey,
    job: Arc<Job>,
    next_channel: u32,
}
static PENDING: Mutex<Option<Pending>> = Mutex::new(None);

/// concurrent vocoder for one channel of audio (src/stretcher.rs:11)
pub struct Stretcher {
    pub spec: AudioSpec,
    input: Receiver<Vec<f32>>,
    job: Arc<Job>,
    channel: u32,
}

impl Stretcher {
    #[allow(clippy::too_many_arguments)]
    pub fn new(
        spec: AudioSpec,
        input: Receiver<Vec<f32>>,
        factor: f32,
        amplitude: f32,
        pitch_multiple: i8,
        window: Vec<f32>,
        buffer_dur: Duration,
        frequency_kernel_src: Option<PathBuf>,
    ) -> Stretcher {
        assert!(pitch_multiple != 0); // src/stretcher.rs:40
        let key = JobKey {
            window: window.iter().map(|w| w.to_bits()).collect(),
            factor: factor.to_bits(),
            amplitude: amplitude.to_bits(),
            pitch_multiple,
            sample_rate: spec.sample_rate,
            channels: spec.channels,
            buffer_secs: buffer_dur.as_secs_f32().to_bits(),
            kernel_src: frequency_kernel_src.clone(),
        };
        let mut pending = PENDING.lock().unwrap();
        if let Some(p) = pending.as_mut() {
            if p.key == key && p.next_channel < spec.channels as u32 {
                let channel = p.next_channel;
                p.next_channel += 1;
                let job = p.job.clone();
                if p.next_channel == spec.channels as u32 {
                    *pending = None;
                }
                return Stretcher { spec, input, job, channel };
            }
        }
        // first channel of a job: Stretcher::new + ReFFT::new for all of its channels (src/stretcher.rs:40-59,
        // src/fft.rs:25-40). A window equal to windows::hanning(len) (src/main.rs:131) takes the engine's fast
        // kernels; the library compares the table itself.
        let mut kernels = frequency_kernel_src.map(|src| Box::new(KernelStack::new(src)));
        let mut cfg = default_config();
        cfg.window_len = window.len() as u32;
        cfg.factor = factor;
        cfg.amplitude = amplitude;
        cfg.pitch_multiple = pitch_multiple as i32;
        cfg.sample_rate = spec.sample_rate;
        cfg.channels = spec.channels as u32;
        cfg.buffer_secs = buffer_dur.as_secs_f32();
        cfg.seed = phase_seed();
        cfg.window = window.as_ptr(); // copied by rc_engine_create
        if let Some(k) = kernels.as_mut() {
            cfg.kernel = Some(crate::kernel_trampoline::dispatch);
            cfg.kernel_user = &mut **k as *mut KernelStack as *mut std::os::raw::c_void;
        }
        let mut raw: *mut RcEngine = std::ptr::null_mut();
        check(unsafe { rc_engine_create(&cfg, &mut raw) });
        let job = Arc::new(Job { handle: Mutex::new(EngineHandle(raw)), _kernels: kernels, window_len: window.len() });
        *pending = if spec.channels > 1 { Some(Pending { key, job: job.clone(), next_channel: 1 }) } else { None };
        Stretcher { spec, input, job, channel: 0 }
    }

    pub fn is_done(&self) -> bool {
        // src/stretcher.rs:78-80
        check(unsafe { rc_engine_is_done(self.job.handle.lock().unwrap().0, self.channel) }) == 1
    }

    pub fn channel_bound(&self) -> usize {
        // src/stretcher.rs:82-85
        unsafe { rc_engine_channel_bound(self.job.handle.lock().unwrap().0) }
    }

    pub fn next_window(&mut self) -> Vec<f32> {
        // src/stretcher.rs:87-121. The engine computes as many windows ahead as the input (and, on an open
        // channel, channel_bound()) allows in one launch and hands them out one per call.
        let mut out = vec![0.0f32; self.window_out_cap()];
        loop {
            let mut n = 0usize;
            let rc = check(unsafe {
                rc_engine_next_window(self.job.handle.lock().unwrap().0, self.channel, out.as_mut_ptr(), out.len(), &mut n)
            });
            if rc == RC_OK {
                out.truncate(n);
                return out;
            }
            // RC_WOULD_BLOCK: the reference blocks in recv here (src/stretcher.rs:125); the engine's lock is
            // not held while this thread waits
            match self.input.recv() {
                Ok(chunk) => {
                    check(unsafe {
                        rc_engine_push_input(self.job.handle.lock().unwrap().0, self.channel, chunk.as_ptr(), chunk.len())
                    });
                }
                Err(_) => {
                    // sender dropped: zero-pad the tail and finish (src/stretcher.rs:129-132)
                    check(unsafe { rc_engine_close_input(self.job.handle.lock().unwrap().0, self.channel) });
                }
            }
        }
    }

    /// `next_window` without the `Vec`: the window as a slice of the engine's pinned block, valid until the
    /// next hand-out of this channel (a sink that takes a slice: the WAV writer of src/main.rs:197-203).
    pub fn next_window_with<R>(&mut self, sink: impl FnOnce(&[f32]) -> R) -> R {
        loop {
            let (mut p, mut n) = (std::ptr::null::<f32>(), 0usize);
            let guard = self.job.handle.lock().unwrap();
            let rc = check(unsafe { rc_engine_next_window_view(guard.0, self.channel, &mut p, &mut n) });
            if rc == RC_OK {
                let r = sink(unsafe { std::slice::from_raw_parts(p, n) });
                drop(guard);
                return r;
            }
            drop(guard);
            match self.input.recv() {
                Ok(chunk) => {
                    check(unsafe {
                        rc_engine_push_input(self.job.handle.lock().unwrap().0, self.channel, chunk.as_ptr(), chunk.len())
                    });
                }
                Err(_) => {
                    check(unsafe { rc_engine_close_input(self.job.handle.lock().unwrap().0, self.channel) });
                }
            }
        }
    }

    fn window_out_cap(&self) -> usize {
        // pitch_multiple >= 1: window_len samples per call; negative multiples emit (S - 1) |p| samples
        // (src/resampler.rs:20-35): ask the engine
        let mut p = std::mem::MaybeUninit::<RcParams>::uninit();
        check(unsafe { rc_engine_get_params(self.job.handle.lock().unwrap().0, p.as_mut_ptr()) });
        let p = unsafe { p.assume_init() };
        (p.window_out_len as usize).max(self.job.window_len)
    }
}

/// The reference seeds its phases from the OS (`rand::thread_rng`, src/fft.rs:64): two runs never agree. The
/// engine's phase source is a pure function of (seed, channel, hop, bin); `ROCODER_SEED` pins it, otherwise the
/// clock does what `thread_rng` did.
fn phase_seed() -> u64 {
    if let Ok(s) = std::env::var("ROCODER_SEED") {
        if let Ok(v) = s.parse::<u64>() {
            return v;
        }
    }
    std::time::SystemTime::now()
        .duration_since(std::time::UNIX_EPOCH)
        .map(|d| d.as_nanos() as u64)
        .unwrap_or(0x5EED)
}

/// The `-o` fast path (src/main.rs:133-155 + `AudioBus::into_audio`, src/audio.rs:152-172, in one call): every
/// channel's whole output, upload / compute / download pipelined over window chunks. `channels` rows of equal
/// length in, `channels` rows out; rows from `PinnedBuf` cross PCIe without a staging copy.
pub fn stretch_offline(
    spec: AudioSpec,
    channels: &[&[f32]],
    factor: f32,
    amplitude: f32,
    pitch_multiple: i8,
    window: &[f32],
) -> Vec<PinnedBuf> {
    assert!(pitch_multiple != 0 && channels.len() == spec.channels as usize);
    let in_len = channels[0].len();
    assert!(channels.iter().all(|c| c.len() == in_len));
    let mut cfg = default_config();
    cfg.window_len = window.len() as u32;
    cfg.factor = factor;
    cfg.amplitude = amplitude;
    cfg.pitch_multiple = pitch_multiple as i32;
    cfg.sample_rate = spec.sample_rate;
    cfg.channels = spec.channels as u32;
    cfg.seed = phase_seed();
    cfg.window = window.as_ptr();
    let mut raw: *mut RcEngine = std::ptr::null_mut();
    check(unsafe { rc_engine_create(&cfg, &mut raw) });
    let eng = EngineHandle(raw);
    let out_len = unsafe { rc_offline_output_len(&cfg, in_len) };
    let mut outs: Vec<PinnedBuf> = (0..channels.len()).map(|_| PinnedBuf::new(out_len)).collect();
    let in_ptrs: Vec<*const f32> = channels.iter().map(|c| c.as_ptr()).collect();
    let out_ptrs: Vec<*mut f32> = outs.iter_mut().map(|o| o.as_mut_ptr()).collect();
    let mut n = 0usize;
    check(unsafe { rc_engine_stretch_host(eng.0, in_ptrs.as_ptr(), in_len, out_ptrs.as_ptr(), out_len, &mut n) });
    assert_eq!(n, out_len);
    outs
}
