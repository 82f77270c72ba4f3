"""In-tree build of libdecompdiff_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m decompdiff_amd.build [--force]

The shared object lands in decompdiff_amd/lib/ (git-ignored, but it travels to the GPU box
with the repo snapshot).  No torch headers are involved: the library exposes the plain C
ABI of include/decompdiff_hip.h.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libdecompdiff_hip.so")
SOURCES = ["dd_gemm.hip", "dd_graph.hip", "dd_attention2.hip", "dd_step.hip", "dd_scatter.hip", "dd_train.hip", "dd_api.hip"]
HEADERS = ["dd_common.hpp", "dd_kernels.hpp", "dd_gemm_tile.hpp", os.path.join("..", "..", "include", "decompdiff_hip.h"),
           os.path.join("..", "..", "include", "decompdiff_hip_debug.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, exact: bool = False, debug_options: bool = False) -> str:
    """``exact``: the -DDD_EXACT_MATH=1 variant (correctly rounded 1/sqrt and softmax division instead of v_rsq_f32 and
    one reciprocal per head) as lib/libdecompdiff_hip_exact.so -- a measurement aid for the parity study (EXPERIMENTS.md,
    profiles/parity_full_chain.json), selected at run time with DD_HIP_LIB; the default library is unaffected."""
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    # debug_options: the measurement build (-DDD_DEBUG_OPTIONS=1) with the alternative launch schedules / kernel variants behind
    # dd_debug_set_option, as lib/libdecompdiff_hip_dbg.so (tools/ab_*.py and the variant cross-check tests select it with
    # DD_HIP_LIB); the default library compiles none of them
    tag = "_exact" if exact else ("_dbg" if debug_options else "")
    lib = LIB.replace(".so", tag + ".so")
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace(".hip", tag + ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc] + FLAGS + (["-DDD_EXACT_MATH=1"] if exact else []) + (["-DDD_DEBUG_OPTIONS=1"] if debug_options else []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(lib, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
    return lib


def build_torch_ext(force: bool = False, verbose: bool = True) -> str:
    """The compiled torch extension csrc/torch_ext.cpp -> lib/decompdiff_torch_ext.so (host-only C++: TORCH_LIBRARY
    registration of the op-level C-ABI entry points; g++ against the installed torch headers, linked to the default
    libdecompdiff_hip.so through $ORIGIN).  Build the default library first."""
    import torch
    from torch.utils import cpp_extension as ce
    src = os.path.join(CSRC, "torch_ext.cpp")
    out = os.path.join(LIBDIR, "decompdiff_torch_ext.so")
    hdr = os.path.join(HERE, "..", "include", "decompdiff_hip.h")
    if not force and not _stale(out, [src, hdr, LIB]):
        return out
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-I", "/opt/rocm/include"]
    for p in ce.include_paths():
        cmd += ["-I", p]
    cmd += [src, "-o", out, "-L", tlib, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch", "-L", LIBDIR, "-ldecompdiff_hip",
            "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"g++ failed:\n{r.stdout}\n{r.stderr}")
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, exact="--exact" in sys.argv, debug_options="--debug-options" in sys.argv))
    if "--exact" not in sys.argv and "--debug-options" not in sys.argv:
        print(build_torch_ext(force="--force" in sys.argv))
