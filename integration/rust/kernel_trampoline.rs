//! Frequency kernels (`--freq-kernel`, README.md:106-128) over the engine's C-ABI callback.
//!
//! The reference loads a Rust-ABI `apply(usize, Vec<(f32,f32)>) -> Vec<(f32,f32)>` with libloading
//! (src/fft.rs:93-94); the engine calls `rc_freq_kernel` (C ABI). User files stay as the README shows them:
//! `hotswapper::compile` (src/hotswapper.rs:52-86) hands rustc the user's source WITH `TRAMPOLINE_SRC`
//! appended (`with_trampoline`), so `apply` and the `extern "C"` wrapper around it are built by the same
//! rustc in the same crate - the only setting in which passing a `Vec` by value is sound.
//! `KernelStack` + `dispatch` are the host side: the reference's `kernels: Vec<Library>` stack, its
//! `try_recv` of new libraries and its pop-and-retry on a panic (src/fft.rs:21,76-108), called by the engine
//! once per hop and channel in the processor's call order.
//! Not compiled in the build container (no rustc).
use crate::hotswapper;
use crossbeam_channel::Receiver;
use libloading::{Library, Symbol};
use std::io::Write;
use std::os::raw::{c_int, c_void};
use std::path::{Path, PathBuf};

/// Appended to the user's kernel source before `rustc --crate-type dylib`.
pub const TRAMPOLINE_SRC: &str = r#"
#[no_mangle]
pub unsafe extern "C" fn rc_apply(time_ms: u64, inp: *const f32, out: *mut f32, n: usize,
                                  _user: *mut std::ffi::c_void) -> i32 {
    let v: Vec<(f32, f32)> = (0..n).map(|i| (*inp.add(2 * i), *inp.add(2 * i + 1))).collect();
    match std::panic::catch_unwind(|| apply(time_ms as usize, v)) {
        Ok(r) if r.len() == n => {
            for (i, (re, im)) in r.iter().enumerate() {
                *out.add(2 * i) = *re;
                *out.add(2 * i + 1) = *im;
            }
            0
        }
        _ => 1, // a panic, or a result of another length (which panics inside rustfft in the reference)
    }
}
"#;

/// `src` + the trampoline in a temporary file next to nothing the user owns; what `compile` passes to rustc.
pub fn with_trampoline(src: &Path) -> std::io::Result<PathBuf> {
    let mut text = std::fs::read_to_string(src)?;
    text.push_str(TRAMPOLINE_SRC);
    let mut f = tempfile::Builder::new().prefix("rocoder_kernel_").suffix(".rs").tempfile()?;
    f.write_all(text.as_bytes())?;
    let (_file, path) = f.keep().map_err(|e| e.error)?;
    Ok(path)
}

type RcApply = unsafe extern "C" fn(u64, *const f32, *mut f32, usize, *mut c_void) -> c_int;

/// The libraries of one job, newest last (src/fft.rs:21).
pub struct KernelStack {
    recv: Receiver<Library>,
    kernels: Vec<Library>,
}

impl KernelStack {
    pub fn new(src: PathBuf) -> KernelStack {
        // compiles once synchronously, then a thread polls the file every 100 ms (src/hotswapper.rs:14-33)
        KernelStack { recv: hotswapper::hotswap(src).unwrap(), kernels: vec![] }
    }

    fn apply(&mut self, time_ms: u64, inp: *const f32, out: *mut f32, n: usize) -> c_int {
        if let Ok(lib) = self.recv.try_recv() {
            log::info!("Got new kernel"); // src/fft.rs:78-81
            self.kernels.push(lib);
        }
        loop {
            let lib = match self.kernels.last() {
                Some(lib) => lib,
                None => return 1, // no kernel (left): the engine resynthesises the unmodified spectrum
            };
            let ok = unsafe {
                match lib.get::<RcApply>(b"rc_apply\0") {
                    Ok(f) => {
                        let f: Symbol<RcApply> = f; // re-resolved every hop, as src/fft.rs:93-94 does
                        f(time_ms, inp, out, n, std::ptr::null_mut()) == 0
                    }
                    Err(_) => false,
                }
            };
            if ok {
                return 0;
            }
            // the kernel panicked: drop it and retry with the previous one (src/fft.rs:100-106)
            log::warn!("kernel failed, falling back to the previous one");
            self.kernels.pop();
        }
    }
}

/// `rc_config.kernel`: `user` is the job's `KernelStack`. Non-zero = identity for this hop (src/fft.rs:100-106).
pub unsafe extern "C" fn dispatch(time_ms: u64, in_reim: *const f32, out_reim: *mut f32, n_bins: usize,
                                  user: *mut c_void) -> c_int {
    let stack = &mut *(user as *mut KernelStack);
    // a panic must not unwind into the C library
    match std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| stack.apply(time_ms, in_reim, out_reim, n_bins))) {
        Ok(rc) => rc,
        Err(_) => 1,
    }
}
