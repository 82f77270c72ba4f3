#!/usr/bin/env python
"""Occupancy timeline of ONE fused node launch (k_attn2_node) from per-workgroup shader clocks: which workgroups (persistent bond-layer,
node blocks NE / NB) run when, on which XCD / CU, and how much CU time the launch loses to ramp-up, whole rounds and its tail.
Needs the measurement variant: tools/build_variant.sh trace -DDD_NODE_TRACE=1, then
    DD_HIP_LIB=$PWD/decompdiff_amd/lib/libdecompdiff_hip_trace.so python tools/node_trace.py [B]"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, hip_lib, shipped_config, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
torch.manual_seed(0)
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(synth.make_pocket_small(0), B).items()}
lib = hip_lib.load()
m.sample_diffusion(num_steps=4, center_pos_mode="protein", seed=1, **b)            # (per-shape CU split measured, graph captured)
buf = torch.zeros(2048, 16, dtype=torch.int64, device=dev)
lib.dd_debug_set_clock_buffer(ctypes.c_void_p(buf.data_ptr()), 200)
m.sample_diffusion(num_steps=2, center_pos_mode="protein", seed=1, use_graph=False, **b)
torch.cuda.synchronize()
lib.dd_debug_set_clock_buffer(None, 200)
c = buf.cpu().numpy()[:, :4]; c = c[c[:, 0] != 0].copy()
np.save(f"gpurun_out/node_trace_raw_B{B}.npy", c)
kind = c[:, 2]; xcc = (c[:, 3] >> 32) & 15; hw = c[:, 3] & 0xffffffff
t0, t1 = c[:, 0].min(), c[:, 1].max()
span = float(t1 - t0)          # ticks of the 100 MHz real-time counter (10 ns)
cu = ((hw >> 8) & 15) | (((hw >> 13) & 7) << 4) | (((hw >> 12) & 1) << 7)                 # CU | SE | SH
print(f"B = {B}: {len(c)} workgroups, launch span {span:.0f} ticks of 10 ns ({span / 100:.1f} us); "
      f"{len(set(zip(xcc.tolist(), cu.tolist())))} distinct (XCD, CU) slots")
for k, name in ((2, "bond-layer (persistent)"), (0, "node blocks NE"), (4, "node half blocks NE"), (1, "node blocks NB")):
    s = c[kind == k]
    if len(s):
        life = (s[:, 1] - s[:, 0]).astype(float)
        print(f"  {name:24s}: {len(s):4d} workgroups, start {np.median(s[:, 0] - t0):8.0f} (median, max {float((s[:, 0] - t0).max()):.0f}), "
              f"life median {np.median(life):7.0f}, end median {np.median(s[:, 1] - t0):8.0f} min {float((s[:, 1] - t0).min()):.0f} max {float((s[:, 1] - t0).max()):.0f}; "
              f"CU-time {life.sum() / span:.1f} CUs x span")
busy = np.zeros(200)
edges = np.linspace(0, span, 201)
for s0, s1 in zip(c[:, 0] - t0, c[:, 1] - t0):
    lo, hi = np.searchsorted(edges, [s0, s1])
    for i in range(max(lo - 1, 0), min(hi, 200)):
        busy[i] += (min(s1, edges[i + 1]) - max(s0, edges[i])) / (edges[i + 1] - edges[i])
print("  resident workgroups over the launch (20 slices):", " ".join(f"{busy[i * 10:(i + 1) * 10].mean():.0f}" for i in range(20)))
print(f"  mean resident workgroups {busy.mean():.1f} (1 per CU possible: 256)")
bl = c[kind == 2]
if len(bl):
    nseg = B * 30 * 29
    print(f"  bond layer: {nseg / 8 / len(bl):.2f} trips of 8 segments per persistent workgroup -> {np.median(bl[:, 1] - bl[:, 0]) / 100 / (nseg / 8 / len(bl)):.2f} us per trip "
          f"(median life / trips); node blocks NE: {np.median((c[kind == 0][:, 1] - c[kind == 0][:, 0])) / 100:.2f} us each")
