#!/usr/bin/env python
"""Node-launch time (HIP events, shipped fused launch) of the library named by DD_HIP_LIB -- used with the timing-only
-DDD_ABLATE=<mask> builds of dd_attention2.hip (tools/build_ablations.sh) to price the parts of the kernel."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from decompdiff_amd import DecompScorePosNet3D, hip_lib, shipped_config, synth
dev = torch.device("cuda:0"); cfg = shipped_config()
large = os.environ.get("DD_WORKLOAD") == "large"; B = int(os.environ.get("DD_B", "8"))
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
pocket = synth.make_pocket_large(0) if large else synth.make_pocket_small(0); torch.manual_seed(0)
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(pocket, B).items()}
lib = hip_lib.load()
m.sample_diffusion(num_steps=1, center_pos_mode="protein", seed=3, keep_traj=False, use_graph=False, **b)
s2, bufs = m._last
cats = (ctypes.c_float * len(hip_lib.PROF_CATS))(); rounds = []
for rnd in range(4):
    hip_lib.check(lib.dd_sampler_reset(ctypes.byref(s2), hip_lib.stream_ptr(dev)), "reset")
    hip_lib.check(lib.dd_profile_step(ctypes.byref(s2), 3, cats, hip_lib.stream_ptr(dev)), "profile")
    if rnd: rounds.append([float(c) for c in cats])
per = {k: min(r[i] for r in rounds) for i, k in enumerate(hip_lib.PROF_CATS)}
print(f"{os.path.basename(os.environ.get('DD_HIP_LIB', 'default')):40s} node launch {1e3 * (per['attn_BL'] / cfg.num_layers - per.get('event_pair', 0)):7.1f} us   "
      f"pos launch {1e3 * (per.get('attn_PE', 0) / cfg.num_layers - per.get('event_pair', 0)):6.1f} us")
