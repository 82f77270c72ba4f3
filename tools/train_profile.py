#!/usr/bin/env python
"""Where a training step's time goes (loss + backward + Adam, B = 4, 300 + 30 atoms): torch.profiler table of the device
kernels and the host ops, launch count, GPU busy time vs wall time.  usage: python tools/train_profile.py [--batch 4]"""
import argparse, sys, time, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, shipped_config, synth
from torch.profiler import profile, ProfilerActivity
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=4)
args = ap.parse_args()
cfg = shipped_config(); torch.manual_seed(0)
bc = synth.build_sampling_batch(synth.make_pocket_small(0), args.batch)
dev = torch.device("cuda:0")
d = lambda t: t.to(dev) if torch.is_tensor(t) else t
kw = dict(protein_pos=d(bc["protein_pos"]), protein_v=d(bc["protein_v"]), batch_protein=d(bc["batch_protein"]),
          protein_group_idx=d(bc["protein_group_idx"]), ligand_pos=d(bc["init_ligand_pos"]), ligand_v=d(bc["init_ligand_v"]),
          ligand_v_aux=d(bc["ligand_v_aux"]), batch_ligand=d(bc["batch_ligand"]), ligand_group_idx=d(bc["ligand_group_idx"]),
          prior_centers=d(bc["prior_centers"]), prior_stds=d(bc["prior_stds"]), prior_num_atoms=d(bc["prior_num_atoms"]),
          batch_prior=d(bc["batch_prior"]), prior_group_idx=d(bc["prior_group_idx"]),
          ligand_decomp_batch=d(bc["ligand_decomp_batch"]), ligand_decomp_index=d(bc["ligand_decomp_index"]),
          ligand_fc_bond_index=d(bc["ligand_fc_bond_index"]), ligand_fc_bond_type=d(bc["init_ligand_fc_bond_type"]),
          batch_ligand_bond=d(bc["batch_ligand_bond"]))
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=5e-4)
def step():
    opt.zero_grad(set_to_none=True)
    r = m.get_diffusion_loss(**kw)
    loss = r["losses"]["pos"] + 100.0 * r["losses"]["v"] + 100.0 * r["losses"]["bond"]
    loss.backward(); opt.step(); return loss
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); print(f"step: {1e3 * (time.perf_counter() - t0) / 5:.1f} ms")
# phases by sync-bracketed timers
def timed(f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, 1e3 * (time.perf_counter() - t)
opt.zero_grad(set_to_none=True)
r, t_f = timed(lambda: m.get_diffusion_loss(**kw))
loss = r["losses"]["pos"] + 100.0 * r["losses"]["v"] + 100.0 * r["losses"]["bond"]
_, t_b = timed(lambda: loss.backward())
_, t_o = timed(lambda: opt.step())
print(f"forward+loss {t_f:.1f} ms, backward {t_b:.1f} ms, Adam {t_o:.1f} ms")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(2): step()
    torch.cuda.synchronize()
ka = prof.key_averages()
dev_total = sum(getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0)) for e in ka)
n_kern = sum(e.count for e in ka if getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0)) > 0 and e.self_cpu_time_total == 0)
print(f"device time per step {dev_total / 2 / 1e3:.1f} ms; device kernel launches per step ~{n_kern // 2}")
print(ka.table(sort_by="self_cuda_time_total", row_limit=30, max_name_column_width=70))
print(ka.table(sort_by="self_cpu_time_total", row_limit=15, max_name_column_width=70))
