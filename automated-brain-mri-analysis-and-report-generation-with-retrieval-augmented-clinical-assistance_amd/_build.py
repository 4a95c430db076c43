"""Builds the gfx950 shared library (C ABI, no torch dependency) in-tree with hipcc."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_DIR = PKG_DIR / "lib"
LIB_PATH = LIB_DIR / "libmi355_nnunet.so"
SOURCES = ["conv3d.hip", "conv3d_f16.hip", "tconv.hip", "elementwise.hip", "extras.hip", "unet.hip"]


def find_hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def needs_build() -> bool:
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    deps = list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + [PKG_DIR.parent / "include" / "mi355_nnunet.h"]
    return any(p.stat().st_mtime > t for p in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    """hipcc --offload-arch=gfx950 -shared: cross-compiles without a GPU."""
    if not force and not needs_build():
        return LIB_PATH
    LIB_DIR.mkdir(exist_ok=True)
    cmd = [find_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-result", "-Wno-pass-failed"]
    cmd += [str(CSRC / s) for s in SOURCES]
    tmp = LIB_PATH.with_suffix(".so.tmp")
    cmd += ["-o", str(tmp)]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
