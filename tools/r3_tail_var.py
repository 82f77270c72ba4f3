#!/usr/bin/env python
"""Timing-only variants of the layer-tail queue (option 25) + serialized per-class times.  usage: python tools/r3_tail_var.py"""
import os as _os; _os.environ.setdefault("DD_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "decompdiff_amd", "lib", "libdecompdiff_hip_dbg.so"))  # measurement build: dd_debug_set_option

import os, sys, time, ctypes, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, hip_lib, shipped_config, synth
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
lib = hip_lib.load()
torch.manual_seed(0)
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(synth.make_pocket_small(0), int(os.environ.get("DD_B", "8"))).items()}
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    try:
        m.sample_diffusion(num_steps=n, center_pos_mode="protein", keep_traj=True, use_graph=True, seed=1, **b)
    except RuntimeError as e:
        print("   (", str(e)[:80], ")")
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
for sched, var in ((4, 0), (5, 0)):
    lib.dd_debug_set_option(8, sched); lib.dd_debug_set_option(25, var)
    run(20)
    print(f"sched {sched} variant {var}: {min(run(200) for _ in range(3)):.4f} ms/step", flush=True)
lib.dd_debug_set_option(25, 0)
for sched, var in ((4, 0), (5, 0), (5, 1), (5, 4)):
    lib.dd_debug_set_option(8, sched); lib.dd_debug_set_option(25, var)
    m.sample_diffusion(num_steps=1, center_pos_mode="protein", keep_traj=False, use_graph=False, seed=3, **b)
    s2, bufs2 = m._last
    cats = (ctypes.c_float * len(hip_lib.PROF_CATS))()
    best = None
    for rnd in range(3):
        hip_lib.check(lib.dd_sampler_reset(ctypes.byref(s2), hip_lib.stream_ptr(dev)), "reset")
        hip_lib.check(lib.dd_profile_step(ctypes.byref(s2), 5, cats, hip_lib.stream_ptr(dev)), "prof")
        cur = [float(c) for c in cats]
        best = cur if best is None else [min(a, c) for a, c in zip(best, cur)]
    print(f"sched {sched} variant {var} serialized per-class ms/step:", {k: round(v, 4) for k, v in zip(hip_lib.PROF_CATS, best)}, flush=True)
lib.dd_debug_set_option(25, 0)
