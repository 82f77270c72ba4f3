#!/usr/bin/env python
"""HBM roofline of the op-level message-passing kernels: algorithmic bytes / HIP-event time.
Sizes are chosen so that k and v (hundreds of MB) exceed the 256 MB Infinity Cache.
usage: python tools/aggregate_bench.py [n_seg] [edges_per_seg]"""
import ctypes, sys, torch
sys.path.insert(0, ".")
from decompdiff_amd import hip_lib
lib = hip_lib.load(); dev = torch.device("cuda:0")
n_seg = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
E = n_seg * K
q = torch.randn(n_seg, 128, device=dev); k = torch.randn(E, 128, device=dev); v = torch.randn(E, 128, device=dev)
ew = torch.rand(E, device=dev); v16 = torch.randn(E, 16, device=dev); rel = torch.randn(E, 3, device=dev)
ptr = (torch.arange(n_seg + 1, device=dev, dtype=torch.int32) * K).contiguous()
out = torch.empty(n_seg, 128, device=dev); out3 = torch.empty(n_seg, 3, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream); P = hip_lib.ptr
def timed(fn, reps=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3
t = timed(lambda: lib.dd_attn_aggregate_node(P(q), 0, P(k), P(v), P(ew), P(ptr), n_seg, P(out), st))
bytes_node = 1032 * E + 1024 * n_seg                        # q, k, v, e_w, dst read once, out written once
print(f"aggregate_node: {n_seg} segments x {K} edges, {bytes_node/1e6:.0f} MB algorithmic, {t*1e6:.1f} us -> {bytes_node/t/1e12:.2f} TB/s "
      f"({bytes_node/t/8e12*100:.0f} % of 8 TB/s)")
t = timed(lambda: lib.dd_attn_aggregate_pos(P(q), P(k), P(v16), P(ew), P(rel), P(ptr), n_seg, P(out3), st))
bytes_pos = 596 * E + 524 * n_seg
print(f"aggregate_pos : {n_seg} segments x {K} edges, {bytes_pos/1e6:.0f} MB algorithmic, {t*1e6:.1f} us -> {bytes_pos/t/1e12:.2f} TB/s "
      f"({bytes_pos/t/8e12*100:.0f} % of 8 TB/s)")
