#!/usr/bin/env python
"""cProfile of short bench-style calls (configs[1]/[2] unit: 300 + 30 atoms, B = 8, optional drift): which host functions make
up the fixed ~1.7 ms a 20-step call costs on top of 20 x 1.215 ms.  usage: python tools/call_profile.py [steps] [drift 0|1]"""
import sys, time, cProfile, pstats, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, shipped_config, synth
from decompdiff_amd import dist as ddist
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
use_drift = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
u = ddist.plan_job(1, 1, batch=8, n_pockets=1, num_samples=8, drift=use_drift)[0][0]
DRIFT = [dict(type="armsca_prox", min_d=1.2, max_d=1.9), dict(type="clash", sigma=2, gamma=4)] if u.drift else None
pocket = synth.make_pocket(u.pocket_seed, u.num_protein, u.arm_atoms, u.scaffold_atoms, num_full_protein=3000 if u.drift else 0)
torch.manual_seed(u.init_seed)
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(pocket, u.n_samples, per_sample_std_scale=[1.0] * u.n_samples).items()}
def call(n, seed):
    out = m.sample_diffusion(num_steps=n, center_pos_mode="protein", energy_drift_opt=DRIFT, seed=seed, keep_traj=True, **b)
    return ddist.checksum(out)
call(5, 1)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); call(steps, 2 + i); torch.cuda.synchronize()
    print(f"call {i}: {1e3 * (time.perf_counter() - t0):.2f} ms")
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    call(steps, 10 + i); torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
