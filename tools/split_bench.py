#!/usr/bin/env python
"""A dense batch run as k concurrent sub-batches (one captured graph, stream and launching host thread each) against the
one-chain run.  usage: python tools/split_bench.py [steps] [fusion_mode for the split runs: 1 two streams per chain, 3 one]"""
import os as _os; _os.environ.setdefault("DD_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "decompdiff_amd", "lib", "libdecompdiff_hip_dbg.so"))  # measurement build: dd_debug_set_option

import os, sys, time, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, hip_lib, shipped_config, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
fmode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
torch.manual_seed(0)
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(synth.make_pocket_small(0), B).items()}
lib = hip_lib.load()
sync = torch.cuda.synchronize
def run_one(n):
    sync(); t0 = time.perf_counter(); r = m.sample_diffusion(num_steps=n, center_pos_mode="protein", seed=1, **b); sync()
    return time.perf_counter() - t0, r
def run_split(n, k):
    kw = dict(b)
    for key in ("protein_group_idx", "ligand_group_idx", "prior_group_idx"):
        kw.setdefault(key, None)
    kw.setdefault("full_protein_pos", None); kw.setdefault("full_batch_protein", None)
    sync(); t0 = time.perf_counter()
    r = m._sample_ragged(kw, None, n, "protein", None, None, 1, True, True, concurrent=True, split=k); sync()
    return time.perf_counter() - t0, r
run_one(5)
t, r0 = run_one(steps); t, r0 = run_one(steps)
print(f"one chain B={B}: {1e3*t/steps:.3f} ms/step", flush=True)
for k in (2, 4):
    lib.dd_debug_set_option(0, fmode)
    try:
        run_split(5, k)
        t, r = run_split(steps, k); t, r = run_split(steps, k)
        print(f"{k} concurrent chains of B={B//k} (fusion mode {fmode}): {1e3*t/steps:.3f} ms/step", flush=True)
    finally:
        lib.dd_debug_set_option(0, 1)
