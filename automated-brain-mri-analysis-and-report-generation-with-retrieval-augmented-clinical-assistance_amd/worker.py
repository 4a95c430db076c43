"""Resident worker of the drop-in boundary.

``run_full_pipeline.py:162-182`` starts ``run_brats2021_inference_singlethread.py`` as a fresh process per case; on this
path that costs more than the GPU work (round 2: 2.2 s of a 4.5 s run were the interpreter, ``import torch``, the library
and the checkpoints).  A worker keeps all of it resident:

    python -m brats_amd.worker [--results_folder DIR] [--dtype f32|f16] [--folds 0 1 2 3 4] [--socket PATH]

loads both ensemble members once and serves requests on a Unix-domain socket (``$MI355_WORKER_SOCKET``, else
``$XDG_RUNTIME_DIR/worker.sock``, else ``/tmp/mi355_nnunet_<uid>/worker.sock`` in a 0700 directory).  The drop-in script is then a thin client: it sends
its argv, relays the worker's output and exits with its return code; when no worker answers it runs the path in-process as
before.  One request at a time (one GPU stream, one activation arena - INTEGRATION.md "Stream semantics"); requests from
concurrent pipelines (``api.py:322-327`` starts one thread per job) queue on the socket, which also serialises the GPU
between them (SURVEY.md 8e "Pitfall").

Protocol: one JSON object per line.  Request ``{"argv": [...], "script_dir": "..."}`` (paths absolute) or
``{"cmd": "ping" | "shutdown"}``; replies ``{"out": "text"}`` any number of times, then ``{"rc": int}``.
"""
from __future__ import annotations

import argparse
import contextlib
import io
import json
import os
import socket
import sys
import traceback


def default_socket_path() -> str:
    """$MI355_WORKER_SOCKET, else a socket in a per-user directory nobody else can write: $XDG_RUNTIME_DIR when set (0700 by
    definition), else /tmp/mi355_nnunet_<uid>/ created 0700 (a predictable name in world-writable /tmp itself could be
    pre-created by another local user, who would then receive the requests: ADVICE r3)."""
    if os.environ.get("MI355_WORKER_SOCKET"):
        return os.environ["MI355_WORKER_SOCKET"]
    base = os.environ.get("XDG_RUNTIME_DIR")
    if not (base and os.path.isdir(base) and os.stat(base).st_uid == os.getuid()):
        base = f"/tmp/mi355_nnunet_{os.getuid()}"
    return os.path.join(base, "worker.sock")


def _owned_by_me(path) -> bool:
    """The path is a socket owned by this user (lstat: a symlink planted by someone else does not count)."""
    import stat
    try:
        st = os.lstat(path)
    except OSError:
        return False
    return st.st_uid == os.getuid() and stat.S_ISSOCK(st.st_mode)


class _SocketWriter(io.TextIOBase):
    """stdout of a request: every write becomes one {"out": ...} line on the connection."""

    def __init__(self, fh):
        self.fh = fh

    def write(self, text):
        if text:
            try:
                self.fh.write(json.dumps({"out": text}) + "\n")
                self.fh.flush()
            except (BrokenPipeError, OSError):
                pass  # the client went away: finish the request, its products are files
        return len(text)

    def flush(self):
        pass


def _is_device_failure(e, mi355_error_type) -> bool:
    """A failed library call, or an error raised by torch's device runtime.  Classified by TYPE (a substring test on the
    message took '/data/ship01/case.nii.gz not found' for a HIP error and ended the serve loop): Mi355Error; torch's
    AcceleratorError / OutOfMemoryError / DeferredCudaCallError; a bare RuntimeError only when torch itself raised it from its
    CUDA/HIP layer, which it marks with the prefix 'CUDA error' / 'HIP error'."""
    if isinstance(e, mi355_error_type):
        return True
    try:
        import torch
        kinds = tuple(k for k in (getattr(torch, "AcceleratorError", None), getattr(torch.cuda, "OutOfMemoryError", None),
                                  getattr(torch.cuda, "DeferredCudaCallError", None)) if isinstance(k, type))
        if kinds and isinstance(e, kinds):
            return True
    except Exception:
        pass
    return type(e) is RuntimeError and str(e).startswith(("CUDA error", "HIP error"))


def _handle(conn, cache, state):
    from . import driver
    fh = conn.makefile("rw", encoding="utf-8", newline="\n")
    line = fh.readline()
    if not line:
        return
    req = json.loads(line)
    if req.get("cmd") == "ping":
        fh.write(json.dumps({"rc": 0, "pid": os.getpid(), "models": len(cache._models)}) + "\n")
        fh.flush()
        return
    if req.get("cmd") == "shutdown":
        state["run"] = False
        fh.write(json.dumps({"rc": 0}) + "\n")
        fh.flush()
        return
    rc = 1
    out = _SocketWriter(fh)
    try:
        with contextlib.redirect_stdout(out):
            try:
                rc = driver.main(list(req.get("argv", [])), script_dir=req.get("script_dir"), model_cache=cache)
            except SystemExit as e:  # argparse errors, the reference's sys.exit(1) on a missing model folder; sys.exit() = success
                rc = 0 if e.code is None else (e.code if isinstance(e.code, int) else 1)
            except Exception as e:
                print(traceback.format_exc())
                rc = 1
                from ._lib import Mi355Error
                if _is_device_failure(e, Mi355Error):
                    # a failed library call or a HIP error may be sticky (a faulted context fails every later launch): answer
                    # this request, then leave the serve loop so that a supervisor starts a fresh process and later clients
                    # fall back to in-process runs instead of collecting rc = 1 forever.  Never re-exec a GPU process.
                    state["run"] = False
                    state["failed"] = True
    finally:
        try:
            fh.write(json.dumps({"rc": int(rc or 0)}) + "\n")
            fh.flush()
        except (BrokenPipeError, OSError):
            pass


def serve(socket_path: str, preload=None, ready_fd=None):
    """Bind, optionally preload (list of (model_dir, folds, dtype)), then serve until a shutdown request."""
    from . import driver
    cache = driver.ModelCache()
    sock_dir = os.path.dirname(socket_path)
    if sock_dir and not os.path.isdir(sock_dir):
        os.makedirs(sock_dir, mode=0o700, exist_ok=True)
    if sock_dir:
        # the directory must be ours and closed to others, whether we made it or found it: another local user who pre-creates
        # /tmp/mi355_nnunet_<uid> could otherwise unlink or replace the socket
        st = os.lstat(sock_dir)
        import stat as _stat
        if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise SystemExit(f"{sock_dir} is not a directory owned by uid {os.getuid()} with mode 0700: refusing to bind a worker socket in it")
    if os.path.exists(socket_path):
        # a stale socket of a dead worker is replaced; a live one answers the ping and we refuse to start a second worker
        with contextlib.suppress(OSError), socket.socket(socket.AF_UNIX, socket.SOCK_STREAM) as probe:
            probe.settimeout(1.0)
            probe.connect(socket_path)
            raise SystemExit(f"a worker already listens on {socket_path}")
        os.unlink(socket_path)
    srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    srv.bind(socket_path)
    os.chmod(socket_path, 0o600)
    srv.listen(16)
    for md, folds, dtype in preload or []:
        cache.get(md, folds, dtype)
    print(f"[worker {os.getpid()}] listening on {socket_path} ({len(cache._models)} models resident)", flush=True)
    if ready_fd is not None:
        os.write(ready_fd, b"ready\n")
    state = {"run": True}
    try:
        while state["run"]:
            conn, _ = srv.accept()
            with conn:
                try:
                    _handle(conn, cache, state)
                except Exception:
                    traceback.print_exc()
    finally:
        srv.close()
        with contextlib.suppress(OSError):
            os.unlink(socket_path)
        cache.close()
    if state.get("failed"):
        raise SystemExit(70)  # EX_SOFTWARE: the device path failed; the supervisor restarts the worker


def request(argv, script_dir, socket_path=None, out=None, connect_timeout=2.0):
    """Client side (also used by the tests): returns the worker's rc, or None when no worker answered - the caller then
    runs the path in-process.  Stdlib only: the drop-in script calls this BEFORE importing torch."""
    path = socket_path or default_socket_path()
    out = out or sys.stdout
    if os.environ.get("MI355_NO_WORKER") == "1" or not os.path.exists(path) or not _owned_by_me(path):
        return None
    try:
        s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        s.settimeout(connect_timeout)
        s.connect(path)
        s.settimeout(None)
    except OSError:
        return None
    with s:
        fh = s.makefile("rw", encoding="utf-8", newline="\n")
        fh.write(json.dumps({"argv": list(argv), "script_dir": script_dir}) + "\n")
        fh.flush()
        for line in fh:
            msg = json.loads(line)
            if "out" in msg:
                out.write(msg["out"])
                out.flush()
            if "rc" in msg:
                return int(msg["rc"])
    return None  # the worker died mid-request: fall back


def absolutise(argv, cwd=None):
    """Paths in the drop-in's argv are relative to the CLIENT's working directory."""
    cwd = cwd or os.getcwd()
    out, path_next = [], False
    for a in argv:
        if path_next:
            a = os.path.abspath(os.path.join(cwd, a))
            path_next = False
        elif a in ("--input", "--output", "--results_folder"):
            path_next = True
        elif a.startswith(("--input=", "--output=", "--results_folder=")):
            k, v = a.split("=", 1)
            a = k + "=" + os.path.abspath(os.path.join(cwd, v))
        out.append(a)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description="resident worker of the MI355X drop-in predictor")
    ap.add_argument("--results_folder", default=None, help="preload both BraTS models from here (default: load on first request)")
    ap.add_argument("--folds", type=int, nargs="+", default=[0, 1, 2, 3, 4])
    ap.add_argument("--dtype", choices=("f32", "f16"), default="f32")
    ap.add_argument("--socket", default=None)
    args = ap.parse_args(argv)
    from . import driver
    preload = []
    if args.results_folder:
        base = os.path.join(args.results_folder, "3d_fullres", "Task500_BraTS2021")
        preload = [(os.path.join(base, m), tuple(args.folds), args.dtype) for m in (driver.MODEL1, driver.MODEL2)]
    serve(args.socket or default_socket_path(), preload)
    return 0


if __name__ == "__main__":
    sys.exit(main())
