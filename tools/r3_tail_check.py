#!/usr/bin/env python
"""Layer-tail queue schedule (option 8 = 5) against the graph-edge schedule (8 = 4) in one process: bit-equality of a short
chain (plain and with drift, dense and padded), then interleaved timing.  usage: python tools/r3_tail_check.py [B] [steps]"""
import os as _os; _os.environ.setdefault("DD_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "decompdiff_amd", "lib", "libdecompdiff_hip_dbg.so"))  # measurement build: dd_debug_set_option

import os, sys, time, statistics, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, hip_lib, shipped_config, synth
dev = torch.device("cuda:0")
cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
lib = hip_lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
DRIFT = [dict(type="armsca_prox", min_d=1.2, max_d=1.9), dict(type="clash", sigma=2, gamma=4)]
def batch(pocket, n):
    torch.manual_seed(0)
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(pocket, n).items()}
def sample(b, n, sched, drift=None, graph=True):
    assert lib.dd_debug_set_option(8, sched) == 0
    r = m.sample_diffusion(num_steps=n, center_pos_mode="protein", keep_traj=True, use_graph=graph, seed=11, energy_drift_opt=drift, **b)
    torch.cuda.synchronize()
    return r
def same(a, b):
    ok = torch.equal(a["pos"], b["pos"]) and torch.equal(a["v"], b["v"]) and torch.equal(a["bond"], b["bond"])
    ok = ok and all(torch.equal(x, y) for x, y in zip(a["pos_traj"], b["pos_traj"])) and all(torch.equal(x, y) for x, y in zip(a["bt_traj"], b["bt_traj"]))
    return ok, float((a["pos"] - b["pos"]).abs().max())
cases = [("small B=%d" % B, batch(synth.make_pocket_small(0), B), None), ("small B=2 drift", batch(synth.make_pocket_small(1), 2), DRIFT),
         ("347+37 B=3", batch(synth.make_pocket(7, 347, (9, 9), 19, num_full_protein=0), 3), None),
         ("large B=2 drift", batch(synth.make_pocket_large(6), 2), DRIFT)]
for name, b, drift in cases:
    r4 = sample(b, 6, 4, drift)
    r5 = sample(b, 6, 5, drift)
    r5e = sample(b, 6, 5, drift, graph=False)
    print(f"{name:18s} sched5 == sched4: {same(r4, r5)}   eager == graph: {same(r5, r5e)}   finite: {bool(torch.isfinite(r5['pos']).all())}", flush=True)
rb = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.ragged_demo_batch(5).items()}
print("ragged (padded)    sched5 == sched4:", same(sample(rb, 5, 4, DRIFT), sample(rb, 5, 5, DRIFT)), flush=True)
b = cases[0][1]
def run(sched, n):
    lib.dd_debug_set_option(8, sched)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.sample_diffusion(num_steps=n, center_pos_mode="protein", keep_traj=True, use_graph=True, seed=1, **b)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
for sc in (4, 5): run(sc, 20)
res = {4: [], 5: []}
for rnd in range(4):
    for sc in (4, 5): res[sc].append(run(sc, steps))
for sc in (4, 5):
    print(f"sched {sc}: median {statistics.median(res[sc]):.4f} ms/step  min {min(res[sc]):.4f}  all {[round(x, 4) for x in res[sc]]}")
b1 = batch(synth.make_pocket_small(0), 1)
b = b1
for sc in (4, 5): run(sc, 20)
print("B=1:", {sc: round(min(run(sc, steps) for _ in range(3)), 4) for sc in (4, 5)})
lib.dd_debug_set_option(8, 5)
