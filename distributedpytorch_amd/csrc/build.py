"""In-tree build of the native libraries (no JIT cache: the ``.so`` files travel with the tree).

    python -m distributedpytorch_amd.csrc.build [--force] [--asm] [--check]

* ``libdpamd_hip.so``  — every ``*.hip`` file: device kernels + ``extern "C"`` launchers, compiled
  with ``hipcc --offload-arch=gfx950`` (cross-compiles without a GPU).
* ``libdpamd_host.so`` — ``host_runtime.cpp``: CPU-side C++ runtime (image warps for the CPU
  transform path, batch collation, bucket planning), compiled with ``g++ -O3 -fopenmp``.

Staleness is decided by CONTENT, not by mtime (a ``git checkout`` rewrites mtimes arbitrarily): every
object carries the SHA-256 of its source + the headers + the flags it was built from
(``build/<name>.o.hash``), and the hash of the whole source set is linked INTO the library
(``dp_build_hash()``), so ``ops._backend.lib()`` can tell at load time whether the ``.so`` it found
matches the sources beside it and rebuild (or refuse) when it does not.
"""
from __future__ import annotations

import fcntl
import glob
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from contextlib import contextmanager

HERE = os.path.dirname(os.path.abspath(__file__))
OBJ_DIR = os.path.join(HERE, "build")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffast-math",
             "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]
HOST_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fopenmp", "-Wall"]
HIP_LIB = os.path.join(HERE, "libdpamd_hip.so")
HOST_LIB = os.path.join(HERE, "libdpamd_host.so")


def _sha(paths, extra: str = "") -> str:
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def _headers():
    return sorted(glob.glob(os.path.join(HERE, "*.h")))


def hip_sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def source_hash() -> str:
    """Hash of everything ``libdpamd_hip.so`` is made from (sources, headers, flags)."""
    return _sha(hip_sources() + _headers(), " ".join(HIP_FLAGS))[:32]


def host_source_hash() -> str:
    return _sha([os.path.join(HERE, "host_runtime.cpp")] + _headers(), " ".join(HOST_FLAGS))[:32]


def _read(path: str) -> str:
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return ""


def _write(path: str, text: str) -> None:
    with open(path, "w") as f:
        f.write(text)


def _run(cmd, verbose):
    if verbose:
        print("+", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError(f"build failed: {' '.join(cmd)}")
    if verbose and r.stderr.strip():
        sys.stderr.write(r.stderr)


@contextmanager
def _build_lock():
    """One builder at a time (several ranks may find a stale library at once)."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(os.path.join(OBJ_DIR, ".lock"), "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def embedded_hash(lib_path: str, symbol: str = "dp_build_hash") -> str:
    """The source hash linked into a built library ('' when absent).  Read from the file's bytes, not through
    ``dlopen``: a stale library must not get mapped into the process that is about to replace it."""
    marker = (symbol.upper() + "=").encode()
    try:
        with open(lib_path, "rb") as f:
            blob = f.read()
    except OSError:
        return ""
    i = blob.find(marker)
    if i < 0:
        return ""
    j = i + len(marker)
    return blob[j:j + 32].decode("ascii", "replace")


def is_current() -> bool:
    return embedded_hash(HIP_LIB) == source_hash()


def build_all(force: bool = False, verbose: bool = False, save_asm: bool = False) -> bool:
    """Compile whatever is missing or stale; returns True when anything was rebuilt."""
    with _build_lock():
        return _build_all_locked(force, verbose, save_asm)


def _build_all_locked(force: bool, verbose: bool, save_asm: bool) -> bool:
    headers = _headers()
    hip_srcs = hip_sources()
    flags = " ".join(HIP_FLAGS)
    objs, jobs, stamps = [], [], []
    for src in hip_srcs:
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        objs.append(obj)
        want = _sha([src] + headers, flags)
        if force or not os.path.isfile(obj) or _read(obj + ".hash") != want:
            cmd = [HIPCC, *HIP_FLAGS, "-c", src, "-o", obj]
            if save_asm:
                cmd += ["-save-temps=obj"]
            jobs.append(cmd)
            stamps.append((obj + ".hash", want))
    workers = min(len(jobs), int(os.environ.get("MAX_JOBS", "8"))) or 1
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(lambda c: _run(c, verbose), jobs))
    for path, want in stamps:
        _write(path, want)
    rebuilt = bool(jobs)
    total = source_hash()
    if hip_srcs and (jobs or force or embedded_hash(HIP_LIB) != total):
        stamp_c = os.path.join(OBJ_DIR, "_build_stamp.cpp")
        _write(stamp_c, f'extern "C" const char* dp_build_hash() {{ return "DP_BUILD_HASH={total}" + 14; }}\n')
        stamp_o = os.path.join(OBJ_DIR, "_build_stamp.o")
        _run([shutil.which("g++") or "g++", "-O1", "-fPIC", "-c", stamp_c, "-o", stamp_o], verbose)
        tmp = HIP_LIB + f".{os.getpid()}.tmp"
        _run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", tmp, *objs, stamp_o], verbose)
        os.replace(tmp, HIP_LIB)
        rebuilt = True
    host_src = os.path.join(HERE, "host_runtime.cpp")
    if os.path.isfile(host_src):
        want = host_source_hash()
        if force or embedded_hash(HOST_LIB, "dp_host_build_hash") != want:
            stamp_c = os.path.join(OBJ_DIR, "_host_stamp.cpp")
            _write(stamp_c, f'extern "C" const char* dp_host_build_hash() {{ return "DP_HOST_BUILD_HASH={want}" + 19; }}\n')
            tmp = HOST_LIB + f".{os.getpid()}.tmp"
            _run([shutil.which("g++") or "g++", *HOST_FLAGS, "-shared", host_src, stamp_c, "-o", tmp], verbose)
            os.replace(tmp, HOST_LIB)
            rebuilt = True
    return rebuilt


def build_host_sanitized(verbose: bool = False) -> str:
    """host_runtime.cpp + its sanitizer driver as ONE executable under ASan + UBSan (host code only:
    GPU AddressSanitizer is not available on this platform).  Returns the path of the executable."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    exe = os.path.join(OBJ_DIR, "host_runtime_sanitize")
    srcs = [os.path.join(HERE, "host_runtime.cpp"), os.path.join(HERE, "host_runtime_sanitize.cpp")]
    want = _sha(srcs)
    if not os.path.isfile(exe) or _read(exe + ".hash") != want:
        cxx = shutil.which("g++") or "g++"
        _run([cxx, "-O1", "-g", "-std=c++17", "-fopenmp", "-fsanitize=address,undefined",
              "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", *srcs, "-o", exe], verbose)
        _write(exe + ".hash", want)
    return exe


if __name__ == "__main__":
    if "--sanitize" in sys.argv:
        print("built:", build_host_sanitized(verbose=True))
        sys.exit(0)
    if "--check" in sys.argv:
        ok = is_current()
        print("libdpamd_hip.so", "is current" if ok else "is STALE or missing", source_hash())
        sys.exit(0 if ok else 1)
    build_all(force="--force" in sys.argv, verbose=True, save_asm="--asm" in sys.argv)
    print("built:", [f for f in os.listdir(HERE) if f.endswith(".so")], "hash", source_hash())
