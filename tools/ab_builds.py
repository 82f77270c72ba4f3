#!/usr/bin/env python
"""Compare two builds of the library on ONE box: alternating subprocesses, each timing the shipped workload.
usage: python tools/ab_builds.py libA.so libB.so [libC.so ...] [rounds]   (paths relative to the repo root)

DD_WORKLOAD=small|mid|large and DD_B select the batch.  Note: packed-weight slot layouts must be compatible with the Python package for both builds.
"""
import os, subprocess, sys, statistics
CHILD = r'''
import os, sys, time, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, shipped_config, synth
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
wl = os.environ.get("DD_WORKLOAD", "small")               # small (300 + 30), mid (347 + 37: 3-tile kernels), large (600 + 60: 4-tile)
pocket = {"small": lambda: synth.make_pocket_small(0), "mid": lambda: synth.make_pocket(5, 347, (12, 12), 13, num_full_protein=360),
          "large": lambda: synth.make_pocket_large(0)}[wl](); torch.manual_seed(0)
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(pocket, int(os.environ.get("DD_B", "8"))).items()}
def run(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.sample_diffusion(num_steps=steps, center_pos_mode="protein", keep_traj=True, use_graph=True, **b)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / steps
run(20)
n = 60 if wl == "large" else 200
print(min(run(n) for _ in range(3)))
'''
if __name__ == "__main__":
    args = sys.argv[1:]; rounds = int(args.pop()) if args and args[-1].isdigit() else 3; libs = args
    res = {l: [] for l in libs}
    for r in range(rounds):
        for l in libs:
            env = dict(os.environ, DD_HIP_LIB=os.path.abspath(l))
            out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
            if out.returncode != 0:
                print(l, "FAILED", out.stderr[-800:]); continue
            res[l].append(float(out.stdout.strip().splitlines()[-1]))
    for l in libs:
        if res[l]: print(f"{l:50s} median {statistics.median(res[l]):.4f} ms/step  all {[round(x, 4) for x in res[l]]}")
