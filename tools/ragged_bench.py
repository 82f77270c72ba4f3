#!/usr/bin/env python
"""Heterogeneous (ragged) batch timing: 16 samples whose ligands have different sizes (SURVEY.md 8f-1 — what the
reference's 'prior' / 'old' / 'stat' num_atoms modes collate), run (a) group after group, (b) all groups together
through dd_sample_steps_graph_multi, and (c) for scale, a dense batch of 16 equal samples.
usage: python tools/ragged_bench.py [steps] [n_distinct_sizes]"""
import os, sys, time, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, hip_lib, shipped_config, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_sizes = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev)
B = 16
sizes = [20 + (20 * i) // max(n_sizes - 1, 1) for i in range(n_sizes)]             # ligand sizes spread over [20, 40]
torch.manual_seed(0)
parts = []
for i in range(B):
    nl = sizes[i % n_sizes]
    arm = max(2, nl // 4)
    p = synth.make_pocket(seed=100 + i % n_sizes, num_protein=300, arm_atoms=(arm, arm), scaffold_atoms=nl - 2 * arm)
    parts.append(synth.build_sampling_batch(p, 1))
rag = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.concat_sampling_batches(parts).items()}
dense = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(synth.make_pocket_small(0), B).items()}


def run(batch, seq, mode="groups"):
    os.environ["DD_RAGGED_CONCURRENT"] = "0" if seq else "1"
    os.environ["DD_RAGGED_MODE"] = mode
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = m.sample_diffusion(num_steps=steps, center_pos_mode="protein", keep_traj=False, use_graph=True, seed=3, **batch)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best, r


t_seq, r_seq = run(rag, True)
t_con, r_con = run(rag, False)
t_pad, r_pad = run(rag, True, "padded")
t_den, _ = run(dense, False)
dense40 = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(synth.make_pocket(0, 300, (10, 10), 20), B).items()}
t_den40, _ = run(dense40, False)
same = all(torch.equal(r_seq[k], r_con[k]) for k in ("pos", "v", "bond"))
print(f"ligand sizes {sizes} x {B // n_sizes if B % n_sizes == 0 else '~' + str(B / n_sizes)}; {steps} steps (incl. graph capture / setup)")
print(f"ragged, group after group : {t_seq / steps * 1e3:8.3f} ms/step")
print(f"ragged, groups together   : {t_con / steps * 1e3:8.3f} ms/step   (x{t_seq / t_con:.2f}; results identical: {same})")
print(f"ragged, ONE padded sequence: {t_pad / steps * 1e3:8.3f} ms/step   ({t_den / t_pad:.2f}x the dense B=16 NL=30 rate, "
      f"{t_den40 / t_pad:.2f}x the dense B=16 NL=40 rate)")
print(f"dense B=16, NL=30         : {t_den / steps * 1e3:8.3f} ms/step")
print(f"dense B=16, NL=40         : {t_den40 / steps * 1e3:8.3f} ms/step")
for bb in (1, 2, 4):
    dd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(synth.make_pocket_small(0), bb).items()}
    print(f"dense B={bb}, NL=30          : {run(dd, False)[0] / steps * 1e3:8.3f} ms/step")
