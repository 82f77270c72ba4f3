#!/usr/bin/env python
"""Time of one training step (get_diffusion_loss + backward + Adam; SURVEY.md 8f-4, scripts/train_diffusion_decomp.py) at the
reference's batch size (configs/training.yml:66, batch_size 4) on C-small pockets (300 + 30 atoms), on the GPU through
decompdiff_amd.training (torch dense layers + the HIP graph ops with analytic backward).
The reference's own step on CPU cores: oracle/time_reference_training.py (needs /root/reference).
usage: python tools/train_step_time.py [--batch 4] [--steps 10]"""
import argparse, sys, time, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, shipped_config, synth
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=4); ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--ragged", action="store_true", help="four DIFFERENT complexes per batch (what the reference's loader yields): one dense sub-batch per size, eager")
args = ap.parse_args()
cfg = shipped_config()
torch.manual_seed(0)
bc = synth.build_sampling_batch(synth.make_pocket_small(0), args.batch)
if args.ragged:
    shapes = [(300, (8, 8), 14), (280, (7, 7), 12), (320, (9, 9), 16), (310, (8, 7), 13)][:max(2, args.batch)]
    bc = synth.concat_sampling_batches([synth.build_sampling_batch(synth.make_pocket(40 + i, n_p, arms, sca, num_full_protein=0), 1)
                                        for i, (n_p, arms, sca) in enumerate(shapes)])
def loss_kwargs(b, dev):
    d = lambda t: t.to(dev) if torch.is_tensor(t) else t
    return dict(protein_pos=d(b["protein_pos"]), protein_v=d(b["protein_v"]), batch_protein=d(b["batch_protein"]),
                protein_group_idx=d(b["protein_group_idx"]), ligand_pos=d(b["init_ligand_pos"]), ligand_v=d(b["init_ligand_v"]),
                ligand_v_aux=d(b["ligand_v_aux"]), batch_ligand=d(b["batch_ligand"]), ligand_group_idx=d(b["ligand_group_idx"]),
                prior_centers=d(b["prior_centers"]), prior_stds=d(b["prior_stds"]), prior_num_atoms=d(b["prior_num_atoms"]),
                batch_prior=d(b["batch_prior"]), prior_group_idx=d(b["prior_group_idx"]),
                ligand_decomp_batch=d(b["ligand_decomp_batch"]), ligand_decomp_index=d(b["ligand_decomp_index"]),
                ligand_fc_bond_index=d(b["ligand_fc_bond_index"]), ligand_fc_bond_type=d(b["init_ligand_fc_bond_type"]),
                batch_ligand_bond=d(b["batch_ligand_bond"]))
if torch.cuda.is_available():
    dev = torch.device("cuda:0")
    m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m.load_state_dict(sd); m = m.to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=5e-4)
    kw = loss_kwargs(bc, dev)
    def step():
        opt.zero_grad(set_to_none=True)
        r = m.get_diffusion_loss(**kw)
        loss = r["losses"]["pos"] + 100.0 * r["losses"]["v"] + 100.0 * r["losses"]["bond"]
        loss.backward(); opt.step(); return float(loss)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps): l = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
    print(f"GPU training step (B = {args.batch}, 300 + 30 atoms): {1e3 * dt:.1f} ms/step = {1 / dt:.2f} steps/s  (loss {l:.4f}, "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**20:.0f} MiB)")
    # the same iteration as one captured graph per batch shape (training.GraphedTrainStep; Adam with capturable=True)
    from decompdiff_amd import training
    torch.manual_seed(0)
    m2 = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m2.state_dict(); sd.update(synth.synthetic_state_dict(cfg, 0)); m2.load_state_dict(sd); m2 = m2.to(dev).train()
    opt2 = torch.optim.Adam(m2.parameters(), lr=5e-4, capturable=True, fused=True)
    gs = training.GraphedTrainStep(m2, opt2, loss_weights=(1.0, 100.0, 100.0))
    for _ in range(6): out = gs.step(**kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps): out = gs.step(**kw)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
    print(f"GPU training step, captured graph (B = {args.batch}): {1e3 * dt:.1f} ms/step = {1 / dt:.2f} steps/s  (loss {float(out['loss']):.4f}, "
          f"{gs.replays} replays, {gs.eager_steps} eager warm-up steps)")
    m.eval()
    with torch.no_grad():
        for _ in range(2): m.get_diffusion_loss(**kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(args.steps): m.get_diffusion_loss(**kw)
        torch.cuda.synchronize(); print(f"GPU validation objective (fused dd_forward, no_grad): {1e3 * (time.perf_counter() - t0) / args.steps:.2f} ms")
