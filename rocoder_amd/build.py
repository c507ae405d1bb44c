"""In-tree build of the gfx950 engine library and of its test-hook twin (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")


def build(force: bool = False, verbose: bool = False) -> str:
    cmd = ["make", "-j6", "-C", CSRC, "all", "hooks"] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building librocoder_hip.so failed")
    return os.path.join(_HERE, "librocoder_hip.so")
