#!/usr/bin/env python
"""s_memtime phase stamps of the K-split projection GEMM tile (profiling aid, needs a GPU)."""
import ctypes, sys, torch, numpy as np
sys.path.insert(0, ".")
from decompdiff_amd import hip_lib
lib = hip_lib.load(); dev = torch.device("cuda:0"); st = torch.cuda.Stream()
names = ["fetch0+commit+sync", "MFMA half 0", "sync+commit1+sync", "MFMA half 1", "stores"]
for rows, ncols in [(6960, 640), (2640, 640), (64, 64), (2640, 128), (2640, 256), (6960, 256)]:
    X = torch.randn(rows, 128, device=dev); W = torch.randn(ncols, 128, device=dev); b = torch.randn(ncols, device=dev); Y = torch.zeros(rows, ncols, device=dev)
    nt = ((rows + 63) // 64) * ((ncols + 63) // 64)
    buf = torch.zeros(nt, 8, dtype=torch.int64, device=dev)
    def call():
        assert lib.dd_gemm128(hip_lib.ptr(X), rows, 0, 128, rows, hip_lib.ptr(W), hip_lib.ptr(b), None, hip_lib.ptr(Y), rows, 0, ncols, ncols, 0, ctypes.c_void_p(st.cuda_stream)) == 0
    for _ in range(3): call()
    torch.cuda.synchronize()
    lib.dd_debug_set_clock_buffer(ctypes.c_void_p(buf.data_ptr()), 100); call(); torch.cuda.synchronize(); lib.dd_debug_set_clock_buffer(None, -1)
    c = buf.cpu().numpy().astype(np.float64)
    d = np.diff(c[:, :6], axis=1)
    print(f"rows {rows} ncols {ncols}: {nt} tiles, kernel span {c[:,5].max()-c[:,0].min():.0f} ticks, first-start spread {np.ptp(c[:,0]):.0f}, median WG life {np.median(c[:,5]-c[:,0]):.0f}")
    print("   " + " | ".join(f"{n} {np.median(d[:, i]):.0f}" for i, n in enumerate(names)))
