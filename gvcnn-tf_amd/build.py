"""Build libgvcnn_hip.so in-tree with hipcc for gfx950 (no GPU needed to compile).

    python gvcnn-tf_amd/build.py [--force]

The shared library is the product's only compute path; nothing falls back to CPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
# GV_PHASE_TIMES=1: the profiling build of tools/phase_times.py (per-workgroup phase timestamps inside conv_dma): its own
# object directory and library name — never the product library
PT = os.environ.get("GV_PHASE_TIMES") == "1"
LIB = os.path.join(HERE, "libgvcnn_hip_pt.so" if PT else "libgvcnn_hip.so")
BUILD = "build_pt" if PT else "build"
SOURCES = ["conv_igemm.hip", "conv_bf16s.hip", "conv_lp.hip", "conv_dma.hip", "conv_ws.hip", "conv_ws_x3.hip", "conv_chain.hip", "pool.hip", "lowp.hip", "grouping.hip", "plan.hip", "train.hip", "train_lp.hip", "wgrad_dma.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + (["-DGV_PHASE_TIMES"] if PT else [])


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # every header is a dependency of every object (a header-only edit must rebuild: lp_elem.h was once missing here)
    headers = [os.path.join(ROOT, "include", "gvcnn_hip.h")] + sorted(
        os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, BUILD), exist_ok=True)
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(HERE, BUILD, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
