"""Build librlx_hip.so (all HIP kernels + the C ABI of include/rlx.h) for gfx950, in-tree.

    python -m rlinf_amd.csrc.build [--force] [--save-temps]

hipcc cross-compiles without a GPU.  Objects are rebuilt only when a source or header is newer.
The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""

from __future__ import annotations

import argparse
import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
INCLUDE = os.path.join(ROOT, "include")
# dev: RLX_BUILD_TAG=nt builds librlx_hip_nt.so from objects in _build_nt (variant sweeps: pair with RLX_CXXFLAGS here and
# RLX_LIB_TAG at run time); unset = the product library
_TAG = os.environ.get("RLX_BUILD_TAG", "")
BUILD = os.path.join(HERE, "_build" + ("_" + _TAG if _TAG else ""))
LIB = os.path.join(os.path.dirname(HERE), "librlx_hip" + ("_" + _TAG if _TAG else "") + ".so")
ARCH = "gfx950"


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")
    return exe


def sources() -> list[str]:
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def _newest_header() -> float:
    hs = glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src: str, force: bool, save_temps: bool) -> str:
    obj = os.path.join(BUILD, os.path.basename(src) + ".o")
    stamp = max(os.path.getmtime(src), _newest_header())
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= stamp:
        return obj
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-Wall", "-Wno-unused-function", f"-I{INCLUDE}", f"-I{HERE}", "-c", src, "-o", obj]
    cmd[1:1] = os.environ.get("RLX_CXXFLAGS", "").split()  # dev: extra -D switches for kernel variant sweeps
    if save_temps:
        cmd.insert(1, "-save-temps=obj")
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=BUILD)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {os.path.basename(src)}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, save_temps: bool = False, verbose: bool = True) -> str:
    os.makedirs(BUILD, exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, save_temps), srcs))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[rlx build] linked {LIB} from {len(objs)} objects")
    elif verbose:
        print(f"[rlx build] {LIB} is up to date")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--save-temps", action="store_true")
    a = ap.parse_args()
    build(force=a.force, save_temps=a.save_temps)
