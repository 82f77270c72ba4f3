#!/usr/bin/env python
"""Time dd_gemm128 on the projection shapes of the shipped workload (profiling aid, needs a GPU)."""
import os as _os; _os.environ.setdefault("DD_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "decompdiff_amd", "lib", "libdecompdiff_hip_dbg.so"))  # measurement build: dd_debug_set_option

import ctypes, sys, torch
sys.path.insert(0, ".")
from decompdiff_amd import hip_lib
lib = hip_lib.load(); dev = torch.device("cuda:0")
st = torch.cuda.Stream()
def bench(rows, ncols, ln, reps=50):
    X = torch.randn(rows, 128, device=dev); W = torch.randn(ncols, 128, device=dev); b = torch.randn(ncols, device=dev)
    lnp = torch.randn(2, 128, device=dev); Y = torch.zeros(rows, ncols, device=dev)
    def call():
        rc = lib.dd_gemm128(hip_lib.ptr(X), rows, 0, 128, rows, hip_lib.ptr(W), hip_lib.ptr(b), hip_lib.ptr(lnp) if ln else None,
                            hip_lib.ptr(Y), rows, 0, ncols, ncols, 0, ctypes.c_void_p(st.cuda_stream))
        assert rc == 0
    with torch.cuda.stream(st):
        for _ in range(5): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps): call()
        e1.record(st)
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    fl = 2.0 * rows * ncols * 128
    by = 4.0 * (rows * 128 + rows * ncols + ncols * 128)
    print(f"rows {rows:6d} ncols {ncols:5d} ln {int(ln)}: {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s  {by / us / 1e3:7.1f} GB/s (min traffic)  tiles {((rows+63)//64)*((ncols+63)//64)}")
for ks in (1, 0):
    lib.dd_debug_set_option(1, ks); print("K-split", ks)
    for rows, ncols, ln in [(6960, 640, 0), (2640, 640, 0), (240, 1280, 0), (6960, 256, 0), (6960, 128, 1), (2640, 128, 1), (240, 128, 1),
                            (2640, 128, 0), (64, 64, 0), (64, 128, 1), (65536, 640, 0)]:
        bench(rows, ncols, ln)
