#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS figures of a built engine library (the code object's metadata note).
usage: kernel_meta.py lib.so [other.so]   - with two libraries: only the kernels whose figures differ"""
import re
import subprocess
import sys

RE = "/opt/rocm/lib/llvm/bin/"


def meta(path):
    """A library holds one offload bundle per translation unit, back to back in .hip_fatbin: cut them apart at the
    bundle magic and read every gfx950 code object's metadata note."""
    import os, tempfile
    d = tempfile.mkdtemp()
    subprocess.run([RE + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, d + "/fb"], check=True)
    blob = open(d + "/fb", "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)] + [len(blob)]
    res = {}
    for i in range(len(starts) - 1):
        fb, co = "%s/fb%d" % (d, i), "%s/co%d" % (d, i)
        open(fb, "wb").write(blob[starts[i]:starts[i + 1]])
        subprocess.run([RE + "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--input=" + fb, "--output=" + co], check=True, capture_output=True)
        out = subprocess.run([RE + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        import yaml
        for doc in re.findall(r"^\s*---\n(.*?)^\s*\.\.\.", out, re.S | re.M):
            for k in (yaml.safe_load(doc) or {}).get("amdhsa.kernels", []):
                res[k[".name"]] = {kk.lstrip("."): vv for kk, vv in k.items() if kk != ".args"}
    return res


def short(n):
    n = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    return n.replace("rc::(anonymous namespace)::", "").replace("(rc::HopParams)", "")


if __name__ == "__main__":
    a = meta(sys.argv[1])
    b = meta(sys.argv[2]) if len(sys.argv) > 2 else None
    keys = ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")
    for name in sorted(a):
        ra = tuple(a[name].get(k, "?") for k in keys)
        if b is None:
            print(short(name), *ra)
        else:
            rb = tuple(b.get(name, {}).get(k, "?") for k in keys)
            if ra != rb:
                print(short(name), ra, "->", rb)
