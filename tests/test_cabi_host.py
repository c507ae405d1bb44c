"""CPU tests (-m "not gpu") of the C-ABI library: it loads, exports every symbol the header
declares, its host-only helpers agree with the oracle, and compute fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from oracle import cbind as oc
from oracle import oracle_np as onp

rocoder_amd = pytest.importorskip("rocoder_amd")
from rocoder_amd import _lib  # noqa: E402
from rocoder_amd.stretcher import derive_params, make_config, offline_output_len  # noqa: E402


def _header_functions():
    src = open(os.path.join(ROOT, "include", "rocoder_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rc_[a-z0-9_]+)\s*\(", src)) - {"rc_freq_kernel"})


def test_library_loads_and_exports_every_declared_symbol():
    L = _lib.lib()
    names = _header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/rocoder_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == names, "ctypes table and header disagree"
    assert L.rc_abi_version() == 5


def test_product_library_reads_no_diagnostic_variable_and_hook_build_loads():
    """ROCODER_DIAG and the run-planner tuning variables change what an engine computes with; only the test-hook
    build (`make hooks`, -DRC_TEST_HOOKS=1) may read them - the product library must not even contain their names.
    The hook build exports the same C-ABI."""
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"ROCODER_DIAG", b"ROCODER_ROUNDS", b"ROCODER_MIN_RUN", b"ROCODER_B4_ROUNDS"):
        assert name not in blob, name
    assert b"hop3_kernel" not in blob  # the previous kernel generation lives in the hook build only
    hooks = open(_lib.HOOKS_PATH, "rb").read()
    assert b"ROCODER_DIAG" in hooks and b"hop3_kernel" in hooks
    with _lib.hooks_library() as H:
        assert _lib.lib() is H
        for n in _header_functions():
            assert hasattr(H, n), n
        assert H.rc_abi_version() == _lib.lib().rc_abi_version()
    assert _lib.lib() is not H


@pytest.mark.parametrize("N,f,p,a", [(16384, 1.0, 1, 1.0), (16384, 8.0, 1, 1.0), (16384, 8.0, 3, 1.0),
                                     (65536, 32.0, 1, 1.0), (1024, 2.0, 2, 0.7), (256, 0.5, 1, 1.0),
                                     (16384, 8.0, -2, 1.0), (16384, 40.0, 5, 2.0)])
def test_derive_params_matches_oracle(N, f, p, a):  # src/stretcher.rs:40-56
    got = derive_params(window_len=N, factor=f, pitch_multiple=p, amplitude=a)
    s = oc.Stretcher(factor=f, amplitude=a, pitch_multiple=p, window=np.ones(N, np.float32))
    assert got.sample_step_len == s.step
    assert got.samples_needed_per_window == s.samples_needed_per_window
    assert got.corrected_amp_factor == np.float32(s.amp)
    assert got.half_window_len == N // 2
    assert got.hops_per_window == (2 * p if p > 0 else -(-s.samples_needed_per_window // (N - N // 2)))


@pytest.mark.parametrize("bad", [dict(pitch_multiple=0), dict(factor=200.0, window_len=256),
                                 dict(pitch_multiple=-1),
                                 dict(window_len=1), dict(channels=0)])
def test_invalid_parameters_rejected(bad):
    with pytest.raises(_lib.RocoderError) as ei:
        derive_params(**bad)
    assert ei.value.code == _lib.RC_EINVAL


@pytest.mark.parametrize("L,N,f,p", [(26460000, 16384, 8.0, 1), (26460000, 16384, 8.0, 3),
                                     (2646000, 16384, 1.0, 1), (5292000, 65536, 32.0, 1),
                                     (0, 256, 1.0, 1), (100, 256, 4.0, 1), (256, 256, 1.0, 1),
                                     (3001, 256, 8.0, 3), (3000, 256, 1.5, -2), (20000, 1024, 4.0, -3)])
def test_offline_output_len_matches_oracle(L, N, f, p):
    assert offline_output_len(L, window_len=N, factor=f, pitch_multiple=p) == \
        oc.offline_output_len(L, N, f, p)


def test_baseline_config_lengths():  # BASELINE.md §3 derived columns
    assert offline_output_len(2646000, window_len=16384, factor=1.0) == 2637824
    assert offline_output_len(26460000, window_len=16384, factor=8.0) == 211566592
    assert offline_output_len(26460000, window_len=16384, factor=8.0, pitch_multiple=3) == 211763200
    assert offline_output_len(5292000, window_len=65536, factor=32.0) == 167313408


def test_phase_source_spec_matches_oracle(goldens):
    z, meta = goldens
    L = _lib.lib()
    key = L.rc_phase_key(0x5EED, 1, 7)
    assert key == meta["hop1024"]["key"] == oc.phase_key(0x5EED, 1, 7)
    for b, h in zip(z["phase/bins"], z["phase/hash"]):
        assert L.rc_phase_hash(key, int(b)) == int(h)
    rng = np.random.default_rng(0)
    for _ in range(50):
        seed, ch, hop, b = (int(rng.integers(0, 2**63)), int(rng.integers(0, 65536)),
                            int(rng.integers(0, 2**40)), int(rng.integers(0, 2**16)))
        k = L.rc_phase_key(seed, ch, hop)
        assert k == onp.phase_key(seed, ch, hop)
        assert L.rc_phase_hash(k, b) == int(onp.phase_hash(k, [b])[0])
        for n in (32, 16384, 65536):
            assert np.float32(L.rc_phase_theta(k, b % n, n)) == onp.phase_theta(k, [b % n], n)[0]


def test_compute_fails_loudly_without_gpu():
    L = _lib.lib()
    if L.rc_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.RocoderError) as ei:
        rocoder_amd.Engine(window_len=1024)
    assert ei.value.code == _lib.RC_ENODEVICE
    with pytest.raises(_lib.RocoderError):
        rocoder_amd.stretch(np.zeros((1, 4096), np.float32), window_len=1024)


def test_abi4_entry_points_fail_with_status_codes_without_gpu():
    """The entry points added with ABI 4 return status codes for bad arguments and, without a GPU, for the device:
    no crash, no silent success (rc_multi_create fails like rc_engine_create: there is no CPU fallback)."""
    L = _lib.lib()
    assert L.rc_multi_set_staging(None, 1) == _lib.RC_EINVAL
    ms, ns = C.c_float(0), C.c_float(0)
    assert L.rc_calib_valu(0, None, 0, C.byref(ms), C.byref(ns)) == _lib.RC_EINVAL  # zero launches
    assert L.rc_calib_valu(0, None, 1, None, None) == _lib.RC_EINVAL                # nowhere to write
    p, n = C.POINTER(C.c_float)(), C.c_size_t(0)
    assert L.rc_engine_next_window_view(None, 0, C.byref(p), C.byref(n)) == _lib.RC_EINVAL
    if L.rc_device_count() > 0:
        pytest.skip("a GPU is present: the rest is the no-device behaviour")
    assert L.rc_calib_valu(0, None, 1, C.byref(ms), C.byref(ns)) < 0
    with pytest.raises(_lib.RocoderError) as ei:
        rocoder_amd.MultiEngine([0, 0], window_len=1024, factor=2.0, channels=1)
    assert ei.value.code == _lib.RC_ENODEVICE


def test_unsupported_window_is_reported_not_faked():
    cfg, _k = make_config(window_len=1001)  # odd lengths (even ones that are not a power of two run as DFTs)
    h = C.c_void_p()
    rc = _lib.lib().rc_engine_create(C.byref(cfg), C.byref(h))
    assert rc == _lib.RC_EUNSUPPORTED and not h.value


def test_product_path_never_touches_the_oracle():
    """rocoder_amd/ must not import, link or execute anything under oracle/."""
    pk = os.path.join(ROOT, "rocoder_amd")
    for dp, _dn, fns in os.walk(pk):
        for fn in fns:
            if fn.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dp, fn), errors="replace").read()
                for pat in (r"^\s*(from|import)\s+oracle", r"#include\s*[\"<][^\n]*oracle",
                            r"librocoder_oracle", r"rco_[a-z_]+\s*\(", r"oracle[/.](cbind|oracle_np)"):
                    assert not re.search(pat, txt, flags=re.M), (os.path.join(dp, fn), pat)


def test_shard_plan_of_the_library_equals_the_python_plan():
    """rc_shard_plan (C++, the one-process multi-device layer) and rocoder_amd.distributed.shard_plan (one process
    per GPU) must cut a job identically: same shards, same order."""
    from rocoder_amd.distributed import shard_plan

    L = _lib.lib()
    for channels in (1, 2, 3, 8):
        for windows in (0, 1, 2, 7, 322, 5106, 25826):
            for n in (1, 2, 3, 4, 8):
                py = shard_plan(channels, windows, n)
                buf = (_lib.rc_shard * (3 * n))()
                cnt = L.rc_shard_plan(channels, windows, n, buf, 3 * n)
                got = [(s.device_index, s.ch_first, s.ch_count, s.win_first, s.win_count) for s in buf[:cnt]]
                want = [(s.rank, s.ch_first, s.ch_count, s.win_first, s.win_count) for s in py]
                assert got == want, (channels, windows, n, got, want)
                assert L.rc_shard_plan(channels, windows, n, None, 0) == cnt  # (count-only call)


def test_abi_struct_layout_matches_the_committed_table_and_the_ctypes_mirror(tmp_path):
    """The structs that cross the C-ABI, field by field: offsets and sizes printed by a C program compiled against
    the header (tools/abi_layout.c) == tests/golden/abi_layout.json (the table INTEGRATION.md shows beside the Rust
    #[repr(C)] structs) == the ctypes Structures of rocoder_amd/_lib.py. A field added, reordered or resized without
    updating a binding fails here, not in a maintainer's build."""
    import json
    import subprocess

    exe = tmp_path / "abi_layout"
    subprocess.run(["gcc", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", str(exe),
                    os.path.join(ROOT, "tools", "abi_layout.c")], check=True)
    now = json.loads(subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout)
    committed = json.load(open(os.path.join(ROOT, "tests", "golden", "abi_layout.json")))
    assert now == committed, "include/rocoder_hip.h changed layout: regenerate tests/golden/abi_layout.json + INTEGRATION.md"
    for name, st in (("rc_config", _lib.rc_config), ("rc_params", _lib.rc_params), ("rc_shard", _lib.rc_shard)):
        table = dict(committed[name])
        size, align = table.pop("sizeof")
        assert C.sizeof(st) == size and C.alignment(st) == align, name
        assert [f[0] for f in st._fields_] == list(table), name  # same fields, same order
        for fname, _t in st._fields_:
            d = getattr(st, fname)
            assert [d.offset, d.size] == table[fname], (name, fname)
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for fname, (off, size) in ((k, v) for k, v in committed["rc_config"].items() if k != "sizeof"):
        assert f"| `{fname}` | {off} | {size} |" in text, f"INTEGRATION.md's rc_config table lacks {fname} @ {off}"


# ------------------------------------------------------------------ INTEGRATION.md's Rust stub, checked without rustc
_RUST_SCALARS = {"u8": (1, 1), "i8": (1, 1), "u16": (2, 2), "i16": (2, 2), "u32": (4, 4), "i32": (4, 4), "f32": (4, 4),
                 "u64": (8, 8), "i64": (8, 8), "f64": (8, 8), "usize": (8, 8), "isize": (8, 8), "c_int": (4, 4),
                 "c_char": (1, 1), "RcFreqKernel": (8, 8)}  # Option<extern "C" fn> is pointer-sized (null = None)


RUST_DIR = os.path.join(ROOT, "integration", "rust")


def _rust_blocks():
    """The reference-side binding ships as files (integration/rust/*.rs; INTEGRATION.md names them): the FFI
    declarations are hip_engine.rs."""
    rust = open(os.path.join(RUST_DIR, "hip_engine.rs")).read()
    return re.sub(r"//[^\n]*", "", rust)


def _rust_struct_layout(rust, name):
    """repr(C): fields in declaration order, each at the next multiple of its alignment; size rounded up to the
    largest alignment (x86-64 / aarch64 LP64: pointers and usize are 8 bytes)."""
    m = re.search(r"#\[repr\(C\)\]\s*pub struct %s\s*\{(.*?)\}" % name, rust, flags=re.S)
    assert m, f"no #[repr(C)] struct {name} in integration/rust/hip_engine.rs"
    off, max_al, out = 0, 1, {}
    for fname, ty in re.findall(r"pub\s+(\w+)\s*:\s*([^,}]+?)\s*(?:,|$)", m.group(1).strip() + ","):
        ty = ty.strip()
        size, al = (8, 8) if ty.startswith("*") else _RUST_SCALARS[ty]
        off = (off + al - 1) // al * al
        out[fname] = [off, size]
        off += size
        max_al = max(max_al, al)
    out["sizeof"] = [(off + max_al - 1) // max_al * max_al, max_al]
    return out


def _c_type_to_rust(t):
    """`const float *const *` -> `*const *const f32` ... (pointers read right to left)."""
    t = " ".join(t.replace("*", " * ").split())
    base_map = {"float": "f32", "int": "c_int", "void": "c_void", "char": "c_char", "size_t": "usize",
                "uint64_t": "u64", "uint32_t": "u32", "int32_t": "i32", "uint16_t": "u16", "rc_config": "RcConfig",
                "rc_params": "RcParams", "rc_engine": "RcEngine", "rc_multi": "RcMulti", "rc_shard": "RcShard"}
    toks = t.split()
    # base type = the tokens before the first '*', minus const
    i = toks.index("*") if "*" in toks else len(toks)
    base_const = "const" in toks[:i]
    base = [x for x in toks[:i] if x != "const"]
    assert len(base) == 1, t
    rust = base_map[base[0]]
    # each '*' (with an optional following const, which qualifies that pointer level) wraps what is to its left
    pointee_const = base_const
    j = i
    while j < len(toks):
        assert toks[j] == "*", t
        rust = ("*const " if pointee_const else "*mut ") + rust
        pointee_const = j + 1 < len(toks) and toks[j + 1] == "const"
        j += 2 if pointee_const else 1
    return rust


def _header_prototypes():
    h = open(os.path.join(ROOT, "include", "rocoder_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    protos = {}
    for ret, name, args in re.findall(r"^([A-Za-z_][\w \*]*?)\b(rc_\w+)\s*\(([^;{]*?)\)\s*;", h, flags=re.M | re.S):
        params = []
        args = " ".join(args.split())
        if args != "void":
            for a in args.split(","):
                a = a.strip()
                ty = re.sub(r"\b\w+$", "", a).strip()  # drop the parameter name
                params.append(_c_type_to_rust(ty))
        r = ret.strip()
        protos[name] = (None if r == "void" else _c_type_to_rust(r), params)
    return protos


def test_integration_md_rust_stub_matches_the_header():
    """the reference-side binding (integration/rust/hip_engine.rs, the file
    INTEGRATION.md section 1 tells a maintainer to copy) cannot be compiled here (no rustc), but
    its text can be held to the C side: (a) the #[repr(C)] structs, laid out by the repr(C) rules, reproduce the table
    the C compiler printed (tests/golden/abi_layout.json); (b) the extern "C" block declares exactly the functions of
    include/rocoder_hip.h, with the header's argument and return types."""
    import json

    rust = _rust_blocks()
    committed = json.load(open(os.path.join(ROOT, "tests", "golden", "abi_layout.json")))
    for rname, cname in (("RcConfig", "rc_config"), ("RcParams", "rc_params"), ("RcShard", "rc_shard")):
        got = _rust_struct_layout(rust, rname)
        assert list(got) == [k for k in committed[cname] if k != "sizeof"] + ["sizeof"], (rname, list(got))
        assert got == {**{k: v for k, v in committed[cname].items() if k != "sizeof"}, "sizeof": committed[cname]["sizeof"]}, rname
    # the kernel callback type
    m = re.search(r"pub type RcFreqKernel = Option<\s*unsafe extern \"C\" fn\((.*?)\)\s*->\s*c_int>;", rust, flags=re.S)
    assert m
    cb = [a.split(":")[1].strip() for a in " ".join(m.group(1).split()).split(",")]
    assert cb == ["u64", "*const f32", "*mut f32", "usize", "*mut c_void"]
    # the extern block
    ext = re.search(r"extern \"C\" \{(.*?)\n\}", rust, flags=re.S)
    assert ext
    fns = {}
    for name, args, ret in re.findall(r"pub fn (\w+)\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", ext.group(1), flags=re.S):
        args = " ".join(args.split())
        params = [a.split(":", 1)[1].strip() for a in args.split(",")] if args else []
        fns[name] = (ret.strip() if ret else None, params)
    protos = _header_prototypes()
    assert sorted(fns) == sorted(protos), (sorted(set(protos) - set(fns)), sorted(set(fns) - set(protos)))
    assert sorted(protos) == _header_functions()
    for name in protos:
        assert fns[name] == protos[name], (name, fns[name], protos[name])


def test_rust_integration_files_are_consistent_with_each_other_and_with_integration_md():
    """What else can be held without rustc: INTEGRATION.md names exactly the files that exist; the ABI version and the
    constants hip_engine.rs declares are the header's; every rc_* function the other Rust files call is declared in
    hip_engine.rs's extern block with that many arguments; the trampoline's exported symbol has the rc_freq_kernel
    shape and is the symbol the dispatcher resolves; braces and parentheses balance in every file; and no file holds
    text of the reference's sources (new code only)."""
    files = sorted(f for f in os.listdir(RUST_DIR) if f.endswith(".rs"))
    assert files == ["build.rs", "hip_engine.rs", "kernel_trampoline.rs", "stretcher_hip.rs"]
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for f in files:
        assert f"integration/rust/{f}" in md, f
    assert "```rust" not in md, "Rust lives in integration/rust/, INTEGRATION.md refers to it"
    h = open(os.path.join(ROOT, "include", "rocoder_hip.h")).read()
    abi = int(re.search(r"#define RC_ABI_VERSION (\d+)", h).group(1))
    assert abi == _lib.lib().rc_abi_version()
    assert re.search(r"rc_abi_version\(\)`? (?:is|returns) %d\b" % abi, md), "INTEGRATION.md quotes another ABI version"
    eng = open(os.path.join(RUST_DIR, "hip_engine.rs")).read()
    assert re.search(r"pub const RC_ABI_VERSION: c_int = %d;" % abi, eng)
    for name in ("RC_OK", "RC_WOULD_BLOCK", "RC_DK_NONE", "RC_DK_GAIN", "RC_DK_BAND", "RC_DK_SHIFT"):
        c_val = int(re.search(r"#define %s \(?(-?\d+)\)?" % name, h).group(1))
        r_val = int(re.search(r"pub const %s: \w+ = (-?\d+);" % name, eng).group(1))
        assert c_val == r_val, name
    protos = _header_prototypes()
    for f in files:
        src = open(os.path.join(RUST_DIR, f)).read()
        code = re.sub(r"//[^\n]*", "", src)
        code_ns = re.sub(r'r#".*?"#', '""', code, flags=re.S)  # (raw strings hold Rust text of their own)
        for o, c in ("{}", "()", "[]"):
            assert code_ns.count(o) == code_ns.count(c), (f, o)
        if f == "hip_engine.rs":
            continue
        for name, args in re.findall(r"\b(rc_\w+)\(((?:[^()]|\([^()]*\))*)\)", code_ns):
            if name == "rc_apply":
                continue
            assert name in protos, (f, name)
            depth, n_args = 0, 1 if args.strip() else 0
            for ch in args:
                depth += ch in "([{"
                depth -= ch in ")]}"
                n_args += ch == "," and depth == 0
            assert n_args == len(protos[name][1]), (f, name, args)
        # new code only: none of the reference's own identifiers that the replacement makes obsolete
        for gone in ("SliceDeque", "rustfft", "FftPlanner", "thread_rng", "amp_correction_envelope"):
            assert gone not in code_ns, (f, gone)
    tr = open(os.path.join(RUST_DIR, "kernel_trampoline.rs")).read()
    m = re.search(r'pub unsafe extern "C" fn rc_apply\((.*?)\)\s*->\s*i32', tr, flags=re.S)
    assert m
    got = [a.split(":", 1)[1].strip() for a in " ".join(m.group(1).split()).split(",")]
    assert got == ["u64", "*const f32", "*mut f32", "usize", "*mut std::ffi::c_void"]
    assert 'b"rc_apply\\0"' in tr  # the dispatcher resolves the symbol the trampoline exports
    m = re.search(r'pub unsafe extern "C" fn dispatch\((.*?)\)\s*->\s*c_int', tr, flags=re.S)
    got = [a.split(":")[1].strip() for a in " ".join(m.group(1).split()).split(",")]
    assert got == ["u64", "*const f32", "*mut f32", "usize", "*mut c_void"]  # = RcFreqKernel


def test_kernel_id_is_a_hash_of_the_kernel_sources_and_current_profiles_match_it():
    """rc_kernel_id() = one hash per kernel family over its sources, the shared headers and the device flags, generated
    at build time (tools/kernel_id.py): the built library holds the ids of the sources in the tree, the test-hook build
    the same ones, and every counter summary profiles/CURRENT.json names as current is headed by the id the library
    holds for that family - a kernel changed after its counters were taken fails here (re-profile, or drop the entry:
    bench.py then quotes no counters for it)."""
    import json
    import re
    import subprocess
    import sys

    from conftest import ROOT

    L = _lib.lib()
    have = dict(re.findall(r"(\w+)=([0-9a-f]+)", L.rc_kernel_id().decode()))
    assert set(have) == {"hop4", "big4", "hopw", "generic", "spectrum"}, have
    csrc = os.path.join(ROOT, "rocoder_amd", "csrc")
    flags = subprocess.run(["make", "-s", "-C", csrc, "print-flags"], capture_output=True, text=True, check=True).stdout
    want = json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_id.py"), "--json", csrc] +
                                     flags.split(), capture_output=True, text=True, check=True).stdout)
    assert have == want, "librocoder_hip.so was built from other kernel sources than the tree holds: run build()"
    with _lib.hooks_library() as H:
        assert H.rc_kernel_id() == L.rc_kernel_id()
    cur = json.load(open(os.path.join(ROOT, "profiles", "CURRENT.json")))
    for fam, name in cur["pmc_summary"].items():
        txt = open(os.path.join(ROOT, "profiles", name)).read()
        m = re.search(r"^#\s*kernel_id:\s*(\S+)", txt, re.M)
        assert m and m.group(1) == f"{fam}={have[fam]}", (fam, name, m and m.group(1), have[fam])


def test_big5_index_model_closes():
    """tests/dev/proto_big5.py models every exchange of big5_kernel (rc_big5.hip): which thread holds which element
    after E1 ... E4 under the residue-class thread mapping, that the wave-local exchanges stay in their wave's region,
    that everything fits eight regions of 2111 slots, and the bank conflicts of every wave instruction (none). The
    kernel's index arithmetic was written from this model; the model stays green with it."""
    import subprocess
    import sys

    from conftest import ROOT

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dev", "proto_big5.py")], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "all layouts check out" in r.stdout
    worst = [int(ln.split(":")[-1]) for ln in r.stdout.splitlines() if "worst extra LDS cycles" in ln]
    assert len(worst) == 20 and max(worst) == 0, r.stdout  # 12 wave instructions of big5_kernel, 8 of big5s_kernel
