#!/usr/bin/env python
"""Two chain lengths in one process (round 6, EXPERIMENTS.md R6-11): a `first`-step call twice, then a `second`-step call twice, wall
time per step of each; under rocprofv3 --kernel-trace the timeline of a step of either phase (tools/timeline.py) shows what differs.
usage: python tools/two_lengths.py [first] [second] [B]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from decompdiff_amd import DecompScorePosNet3D, shipped_config, synth
first = int(sys.argv[1]) if len(sys.argv) > 1 else 500
second = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
torch.manual_seed(0)
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(synth.make_pocket_small(0), B).items()}
for n in (20, first, first, second, second, first):
    torch.cuda.synchronize(); t = time.perf_counter()
    m.sample_diffusion(num_steps=n, center_pos_mode="protein", keep_traj=True, use_graph=True, seed=1, **b)
    torch.cuda.synchronize()
    print(f"{n:5d} steps: {1e3 * (time.perf_counter() - t) / n:.4f} ms/step", flush=True)
