"""In-tree build of the native libraries (no JIT cache: the ``.so`` files travel with the tree).

    python -m distributedpytorch_amd.csrc.build [--force] [--asm]

* ``libdpamd_hip.so``  — every ``*.hip`` file: device kernels + ``extern "C"`` launchers, compiled
  with ``hipcc --offload-arch=gfx950`` (cross-compiles without a GPU).
* ``libdpamd_host.so`` — ``host_runtime.cpp``: CPU-side C++ runtime (image warps for the CPU
  transform path, bucket planning), compiled with ``g++ -O3 -fopenmp``.

Objects are cached by source mtime so a rebuild after editing one kernel takes seconds.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OBJ_DIR = os.path.join(HERE, "build")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffast-math",
             "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]
HOST_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fopenmp", "-Wall"]


def _newer(src: str, dst: str, deps=()) -> bool:
    if not os.path.isfile(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in (src, *deps))


def _run(cmd, verbose):
    if verbose:
        print("+", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError(f"build failed: {' '.join(cmd)}")
    if verbose and r.stderr.strip():
        sys.stderr.write(r.stderr)


def build_all(force: bool = False, verbose: bool = False, save_asm: bool = False) -> None:
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = sorted(glob.glob(os.path.join(HERE, "*.h")))
    hip_srcs = sorted(glob.glob(os.path.join(HERE, "*.hip")))
    objs, jobs = [], []
    for src in hip_srcs:
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _newer(src, obj, headers):
            cmd = [HIPCC, *HIP_FLAGS, "-c", src, "-o", obj]
            if save_asm:
                cmd += ["-save-temps=obj"]
            jobs.append(cmd)
    workers = min(len(jobs), int(os.environ.get("MAX_JOBS", "8"))) or 1
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(lambda c: _run(c, verbose), jobs))
    lib = os.path.join(HERE, "libdpamd_hip.so")
    if hip_srcs and (jobs or not os.path.isfile(lib)):
        _run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib, *objs], verbose)
    host_src = os.path.join(HERE, "host_runtime.cpp")
    host_lib = os.path.join(HERE, "libdpamd_host.so")
    if os.path.isfile(host_src) and (force or _newer(host_src, host_lib, headers)):
        cxx = shutil.which("g++") or "g++"
        _run([cxx, *HOST_FLAGS, "-shared", host_src, "-o", host_lib], verbose)


def build_host_sanitized(verbose: bool = False) -> str:
    """host_runtime.cpp + its sanitizer driver as ONE executable under ASan + UBSan (host code only:
    GPU AddressSanitizer is not available on this platform).  Returns the path of the executable."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    exe = os.path.join(OBJ_DIR, "host_runtime_sanitize")
    srcs = [os.path.join(HERE, "host_runtime.cpp"), os.path.join(HERE, "host_runtime_sanitize.cpp")]
    if any(_newer(s, exe) for s in srcs):
        cxx = shutil.which("g++") or "g++"
        _run([cxx, "-O1", "-g", "-std=c++17", "-fopenmp", "-fsanitize=address,undefined",
              "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", *srcs, "-o", exe], verbose)
    return exe


if __name__ == "__main__":
    if "--sanitize" in sys.argv:
        print("built:", build_host_sanitized(verbose=True))
        sys.exit(0)
    build_all(force="--force" in sys.argv, verbose=True, save_asm="--asm" in sys.argv)
    print("built:", [f for f in os.listdir(HERE) if f.endswith(".so")])
