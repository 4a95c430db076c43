"""Builds the gfx950 shared library (C ABI, no torch dependency) in-tree with hipcc.

Each ``csrc/*.hip`` is compiled to an object file in parallel (``hipcc -c``), then linked into
``<repo>/lib/libmi355_nnunet.so`` (a short in-tree path: the package directory's prescribed name is ~100 characters
long, and tools that list ``/proc/<pid>/maps`` truncate such lines); objects are rebuilt only when their source or a
header changed.  "Changed" is decided by content, not by time stamps: the library and every object carry a digest of
what they were built from (``*.digest`` beside them), so a snapshot that shuffles mtimes can neither force a rebuild nor -
the dangerous direction - let a stale library pass for a fresh one.  Concurrent builders (torch.distributed ranks, parallel pipeline processes) serialise on an flock and
link through a per-process temporary name, so nobody ever loads a half-written library.
"""
from __future__ import annotations

import fcntl
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from contextlib import contextmanager
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
# the repo keeps the library beside the package (it travels to the GPU box in-tree); an installed copy whose
# site-packages is read-only points MI355_LIB_DIR somewhere writable
LIB_DIR = Path(os.environ["MI355_LIB_DIR"]).resolve() if os.environ.get("MI355_LIB_DIR") else PKG_DIR.parent / "lib"
OBJ_DIR = LIB_DIR / "obj"
LIB_PATH = LIB_DIR / "libmi355_nnunet.so"
SOURCES = ["conv3d.hip", "conv3d_wino3.hip", "conv3d_f16.hip", "conv3d_f16_s2.hip", "conv_stem.hip", "tconv.hip", "elementwise.hip", "extras.hip", "resample.hip", "unet.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", "-Wno-pass-failed"]


def find_hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _headers():
    return list(CSRC.glob("*.h")) + [PKG_DIR.parent / "include" / "mi355_nnunet.h"]


def _digest(paths) -> str:
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for p in sorted(paths, key=lambda q: q.name):
        h.update(p.name.encode() + b"\0" + p.read_bytes() + b"\0")
    return h.hexdigest()


def _object_digest(src: Path) -> str:
    return _digest([src] + _headers())


def _library_digest() -> str:
    return _digest([CSRC / s for s in SOURCES] + _headers())


def _recorded(artefact: Path) -> str:
    f = artefact.with_name(artefact.name + ".digest")
    return f.read_text().strip() if f.exists() else ""


def _record(artefact: Path, digest: str):
    artefact.with_name(artefact.name + ".digest").write_text(digest + "\n")


def needs_build() -> bool:
    return not LIB_PATH.exists() or _recorded(LIB_PATH) != _library_digest()


@contextmanager
def _build_lock():
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    with open(LIB_DIR / ".build.lock", "w") as fh:
        fcntl.flock(fh, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(fh, fcntl.LOCK_UN)


def _listing(src: Path) -> Path:
    """the device assembly listing kept beside the object of `src` (what the ISA gate reads)"""
    return OBJ_DIR / (src.stem + ".gfx950.s")


def _compile(hipcc: str, src: Path, obj: Path, verbose: bool):
    """One source -> one object, in a private temporary directory (-save-temps=obj leaves the device listing there), then the ISA
    gate (_isa_gate.py: no scratch in the hand-counted kernels, no un-padded hazard inside inline asm) on that listing.  A listing
    that fails the gate fails the build: the object is not installed."""
    from . import _isa_gate
    tmpdir = OBJ_DIR / f".tmp.{src.stem}.{os.getpid()}"
    shutil.rmtree(tmpdir, ignore_errors=True)
    tmpdir.mkdir(parents=True)
    try:
        tmp = tmpdir / (src.stem + ".o")
        cmd = [hipcc, *FLAGS, "-save-temps=obj", "-c", str(src), "-o", str(tmp)]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}:\n" + res.stdout + res.stderr)
        lst = tmpdir / (src.stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")
        if not lst.exists():
            raise RuntimeError(f"hipcc left no device listing for {src.name} (-save-temps=obj): the ISA gate cannot run")
        findings = _isa_gate.check_asm_file(lst)
        os.replace(lst, _listing(src))   # (kept for inspection, also when the gate fails)
        if findings and os.environ.get("MI355_ISA_GATE", "1") != "0":
            raise RuntimeError(f"ISA gate failed on {src.name} ({len(findings)} findings; listing: {_listing(src)}):\n" + "\n".join(findings[:20]))
        os.replace(tmp, obj)
    finally:
        shutil.rmtree(tmpdir, ignore_errors=True)


def build(force: bool = False, verbose: bool = False) -> Path:
    """hipcc --offload-arch=gfx950: cross-compiles without a GPU."""
    if not force and not needs_build():
        return LIB_PATH
    with _build_lock():
        if not force and not needs_build():  # another process built it while we waited for the lock
            return LIB_PATH
        hipcc = find_hipcc()
        OBJ_DIR.mkdir(parents=True, exist_ok=True)
        lib_digest = _library_digest()  # (taken before compiling: an edit during the build leaves the digest stale, not wrong)
        jobs = []
        for s in SOURCES:
            src, obj = CSRC / s, OBJ_DIR / (s + ".o")
            d = _object_digest(src)
            if force or not obj.exists() or _recorded(obj) != d:
                jobs.append((src, obj, d))

        def compile_one(j):
            _compile(hipcc, j[0], j[1], verbose)
            _record(j[1], j[2])
        with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as pool:
            list(pool.map(compile_one, jobs))
        tmp = LIB_PATH.with_suffix(f".so.{os.getpid()}.tmp")
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[str(OBJ_DIR / (s + ".o")) for s in SOURCES], "-o", str(tmp)]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            tmp.unlink(missing_ok=True)
            raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
        os.replace(tmp, LIB_PATH)
        _record(LIB_PATH, lib_digest)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
