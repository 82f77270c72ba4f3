#!/usr/bin/env python
"""Compare dd_debug_set_option variants in SEPARATE processes, alternating (each process measures its own per-shape CU
split of the node launch first -- ab_bench.py keeps the split of the first variant, which hides changes that speed up
only one part of that launch).  usage: python tools/ab_procs.py "" "22=0" "8=2,22=0" [rounds]   (DD_B = batch)"""
import os as _os; _os.environ.setdefault("DD_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "decompdiff_amd", "lib", "libdecompdiff_hip_dbg.so"))  # measurement build: dd_debug_set_option

import os, subprocess, sys, statistics
CHILD = r'''
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[0]))) if False else ".")
from decompdiff_amd import DecompScorePosNet3D, hip_lib, shipped_config, synth
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
lib = hip_lib.load()
for kv in filter(None, os.environ.get("DD_SET_OPTIONS", "").split(",")):
    k, v = kv.split("="); assert lib.dd_debug_set_option(int(k), int(v)) == 0
pocket = synth.make_pocket_large(0) if os.environ.get("DD_WORKLOAD") == "large" else synth.make_pocket_small(0); torch.manual_seed(0)
B = int(os.environ.get("DD_B", "8"))
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(pocket, B).items()}
def run(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.sample_diffusion(num_steps=steps, center_pos_mode="protein", keep_traj=True, use_graph=True, seed=1, **b)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / steps
run(20)
n = int(os.environ.get("DD_STEPS", "200"))
print(min(run(n) for _ in range(3)), lib.dd_debug_node_split(B, pocket.num_protein_atoms, pocket.num_ligand_atoms, 32))
'''
args = sys.argv[1:]; rounds = int(args.pop()) if args and args[-1].isdigit() else 3; variants = args or [""]
res = {v: [] for v in variants}; split = {}
for r in range(rounds):
    for v in variants:
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, DD_SET_OPTIONS=v), capture_output=True, text=True)
        if out.returncode != 0:
            print(repr(v), "FAILED", out.stderr[-800:]); continue
        t, sp = out.stdout.strip().splitlines()[-1].split()
        res[v].append(float(t)); split[v] = sp
for v in variants:
    if res[v]: print(f"{v or 'default':24s} median {statistics.median(res[v]):.4f} ms/step  all {[round(x, 4) for x in res[v]]}  node split {split[v]} CUs")
