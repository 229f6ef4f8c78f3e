"""Build pyslice_amd/libmslice.so for gfx950 with hipcc (in-tree, no JIT cache).

    python -m pyslice_amd.build_native [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmslice.so")
# translation units, compiled in parallel (the 70 time-kernel instantiations are two thirds of the compile time)
SOURCES = ["mslice.hip", "slice_pass.hip", "slice_mixed_a.hip", "slice_mixed_b.hip", "slice_mixed_c.hip", "slice_mixed_d.hip", "slice_mixed_e.hip", "slice_mixed_f.hip", "tacaw_direct.hip", "tacaw_split.hip", "tacaw_split2.hip"]
def _deps():
    """every source and header of csrc/ plus the public header"""
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) + [os.path.join("..", "..", "include", "mslice.h")]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libmslice.so)")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    for d in _deps():
        p = os.path.join(CSRC, d)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC"]

    def compile_one(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        cmd = [_hipcc()] + flags + ["-c", "-o", obj, src]
        if verbose:
            print("[pyslice_amd] " + " ".join(cmd), flush=True)
        subprocess.run(cmd, cwd=CSRC, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print("[pyslice_amd] " + " ".join(cmd), flush=True)
    subprocess.run(cmd, cwd=CSRC, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
