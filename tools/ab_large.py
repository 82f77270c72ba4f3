#!/usr/bin/env python
"""In-process A/B on the C-large workload (600 + 60 atoms, B=8): usage python tools/ab_large.py "13=33" ..."""
import os as _os; _os.environ.setdefault("DD_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "decompdiff_amd", "lib", "libdecompdiff_hip_dbg.so"))  # measurement build: dd_debug_set_option

import sys, time, statistics, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, hip_lib, shipped_config, synth
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
pocket = synth.make_pocket_large(0); torch.manual_seed(0)
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(pocket, 8).items()}
lib = hip_lib.load()
variants = [("baseline", [])] + [(a, [tuple(int(x) for x in kv.split("=")) for kv in a.split(",")]) for a in sys.argv[1:]]
DEFAULTS = {13: 64}
def run(settings, steps=40):
    for k, v in DEFAULTS.items(): lib.dd_debug_set_option(k, v)
    for k, v in settings: lib.dd_debug_set_option(k, v)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = m.sample_diffusion(num_steps=steps, center_pos_mode="protein", keep_traj=True, use_graph=True, **b)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / steps, float(r["pos"].double().sum())
for name, st in variants: run(st, 5)
res = {name: [] for name, _ in variants}
for rnd in range(3):
    for name, st in variants: res[name].append(run(st))
for name, _ in variants:
    print(f"{name:12s} median {statistics.median(x[0] for x in res[name]):.4f} ms/step   checksum {res[name][0][1]:.6f}")
for k, v in DEFAULTS.items(): lib.dd_debug_set_option(k, v)
