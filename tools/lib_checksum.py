#!/usr/bin/env python
"""Checksums of a short seeded chain (C-small B=8 and a 37-atom-ligand batch) -- to confirm that two builds of the library
(DD_HIP_LIB=...) give bit-identical results.  usage: DD_HIP_LIB=path python tools/lib_checksum.py [steps]"""
import os as _os
if _os.environ.get("DD_OPTS"): _os.environ.setdefault("DD_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "decompdiff_amd", "lib", "libdecompdiff_hip_dbg.so"))  # measurement build: dd_debug_set_option

import os, sys, hashlib, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, hip_lib, shipped_config, synth
for kv in os.environ.get("DD_OPTS", "").split(","):      # dd_debug_set_option settings, e.g. DD_OPTS="24=0"
    if kv:
        k, v = kv.split("="); assert hip_lib.load().dd_debug_set_option(int(k), int(v)) == 0
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
for name, pocket, B in (("small", synth.make_pocket_small(0), 8), ("mid37", synth.make_pocket(5, 347, (12, 12), 13, num_full_protein=360), 4),
                        ("large", synth.make_pocket_large(0), 2)):
    torch.manual_seed(0)
    b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(pocket, B).items()}
    drift = ([dict(type="armsca_prox", min_d=1.2, max_d=1.9), dict(type="clash", sigma=2.0, gamma=4.0)]
             if os.environ.get("DD_DRIFT") == "1" else None)      # (configs/sampling_drift.yml values)
    r = m.sample_diffusion(num_steps=steps, center_pos_mode="protein", seed=7, energy_drift_opt=drift, **b)
    h = hashlib.sha256()
    for k in ("pos", "v", "bond"):
        h.update(r[k].cpu().numpy().tobytes())
    h.update(torch.stack(r["v0_traj"]).numpy().tobytes())
    print(name, h.hexdigest()[:16], flush=True)
