#!/usr/bin/env python
"""Wall time of successive short sample_diffusion calls exactly as bench.py issues them (no syncs inside the call).
usage: python tools/short_calls.py [steps] [reps]"""
import sys, time, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, shipped_config, synth
from decompdiff_amd import dist as ddist
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
torch.manual_seed(2021)
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(synth.make_pocket_small(0), 8, per_sample_std_scale=[1.0] * 8).items()}
t0 = time.perf_counter(); m.sample_diffusion(num_steps=5, center_pos_mode="protein", seed=1, **b); torch.cuda.synchronize()
print(f"warm-up call (5 steps, incl. per-shape measurement): {1e3*(time.perf_counter()-t0):.1f} ms")
for rep in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = m.sample_diffusion(num_steps=steps, center_pos_mode="protein", seed=2 + rep, **b)
    t1 = time.perf_counter(); cs = ddist.checksum(out); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"call {rep}: {steps} steps returned after {1e3*(t1-t0):.2f} ms, + checksum/sync {1e3*(t2-t1):.2f} ms")
