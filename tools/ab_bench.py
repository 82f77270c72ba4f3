#!/usr/bin/env python
"""Interleaved A/B timing of runtime variants inside ONE process (box-to-box variation is ~8%, so variants must be
compared within a run).  usage: python tools/ab_bench.py "0=1" "0=3" "1=0" ...   (key=value settings of
dd_debug_set_option; the baseline is always included)"""
import os as _os; _os.environ.setdefault("DD_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "decompdiff_amd", "lib", "libdecompdiff_hip_dbg.so"))  # measurement build: dd_debug_set_option

import sys, time, statistics, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, hip_lib, shipped_config, synth
dev = torch.device("cuda:0")
cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
pocket = synth.make_pocket_small(0); torch.manual_seed(0)
import os
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(pocket, int(os.environ.get("DD_B", "8"))).items()}
lib = hip_lib.load()
variants = [("baseline", [])] + [(a, [tuple(int(x) for x in kv.split("=")) for kv in a.split(",")]) for a in sys.argv[1:]]
DEFAULTS = {0: 1, 1: 1, 2: 8, 3: 1, 5: 4, 7: 1, 8: 4, 9: 1, 11: 0, 12: 1, 14: 0, 16: 1, 17: 1, 18: 1, 19: 1, 20: 1, 21: 1, 22: 1, 27: 0, 28: 1, 29: 0, 30: 0, 31: 0}
def run(settings, steps=200):
    for k, v in DEFAULTS.items(): lib.dd_debug_set_option(k, v)
    for k, v in settings: lib.dd_debug_set_option(k, v)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.sample_diffusion(num_steps=steps, center_pos_mode="protein", keep_traj=True, use_graph=True, seed=1, **b)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / steps
for name, st in variants: run(st, 20)
res = {name: [] for name, _ in variants}
for rnd in range(5):
    for name, st in variants: res[name].append(run(st))
for name, _ in variants:
    print(f"{name:20s} median {statistics.median(res[name]):.4f} ms/step   min {min(res[name]):.4f}   all {[round(x,3) for x in res[name]]}")
import hashlib
for name, st in variants:                       # bit-identity of the variants: hash of a seeded 10-step chain
    for k, v in DEFAULTS.items(): lib.dd_debug_set_option(k, v)
    for k, v in st: lib.dd_debug_set_option(k, v)
    o = m.sample_diffusion(num_steps=10, center_pos_mode="protein", keep_traj=True, use_graph=True, seed=5, **b)
    hsh = hashlib.sha256(o["pos"].cpu().numpy().tobytes() + o["v"].cpu().numpy().tobytes() + o["bond"].cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"{name:20s} sha256(pos|v|bond) after 10 steps: {hsh}")
for k, v in DEFAULTS.items(): lib.dd_debug_set_option(k, v)
