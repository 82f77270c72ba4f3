#!/usr/bin/env python
"""Compare dd_debug_set_option settings in alternating subprocesses (each timing the shipped workload, min of 3 x 200 steps).
usage: python tools/ab_opts.py "24=0" "24=2" "24=3,8=4" [rounds]"""
import os as _os; _os.environ.setdefault("DD_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "decompdiff_amd", "lib", "libdecompdiff_hip_dbg.so"))  # measurement build: dd_debug_set_option

import os, subprocess, sys, statistics
CHILD = r'''
import os, sys, time, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, hip_lib, shipped_config, synth
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
lib = hip_lib.load()
for kv in os.environ.get("DD_OPTS", "").split(","):
    if kv:
        k, v = kv.split("="); assert lib.dd_debug_set_option(int(k), int(v)) == 0
pocket = synth.make_pocket_small(0); torch.manual_seed(0)
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(pocket, int(os.environ.get("DD_B", "8"))).items()}
def run(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.sample_diffusion(num_steps=steps, center_pos_mode="protein", keep_traj=True, use_graph=True, **b)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / steps
run(20)
print(min(run(200) for _ in range(3)))
'''
args = sys.argv[1:]; rounds = int(args.pop()) if args and args[-1].isdigit() else 3
res = {o: [] for o in args}
for r in range(rounds):
    for o in args:
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, DD_OPTS=o), capture_output=True, text=True)
        if out.returncode != 0:
            print(o, "FAILED", out.stderr[-800:]); continue
        res[o].append(float(out.stdout.strip().splitlines()[-1]))
for o in args:
    if res[o]: print(f"{o:30s} median {statistics.median(res[o]):.4f} ms/step  all {[round(x, 4) for x in res[o]]}")
