"""ORACLE TOOLING (test infrastructure; never imported by the product; needs /root/reference, i.e. the build container).

Wall time of the REFERENCE's own training step on CPU cores -- get_diffusion_loss + backward + Adam
(models/decompdiff.py:419-550, scripts/train_diffusion_decomp.py) at configs/training.yml's batch size 4 on C-small pockets
(300 + 30 atoms) -- the context number for tools/train_step_time.py (the same step on the MI355X).

    python -m oracle.time_reference_training [--batch 4] [--steps 3] [--threads 8]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from decompdiff_amd import shipped_config, synth                # noqa: E402
from oracle import ref_shims                                    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    cfg = shipped_config()
    ref = ref_shims.load_reference_model(cfg.to_dict(), synth.synthetic_state_dict(cfg, seed=0))
    ref.train()
    torch.manual_seed(0)
    b = synth.build_sampling_batch(synth.make_pocket_small(0), args.batch)
    kw = dict(protein_pos=b["protein_pos"], protein_v=b["protein_v"], batch_protein=b["batch_protein"],
              protein_group_idx=b["protein_group_idx"], ligand_pos=b["init_ligand_pos"], ligand_v=b["init_ligand_v"],
              ligand_v_aux=b["ligand_v_aux"], batch_ligand=b["batch_ligand"], ligand_group_idx=b["ligand_group_idx"],
              prior_centers=b["prior_centers"], prior_stds=b["prior_stds"], prior_num_atoms=b["prior_num_atoms"],
              batch_prior=b["batch_prior"], prior_group_idx=b["prior_group_idx"],
              ligand_decomp_batch=b["ligand_decomp_batch"], ligand_decomp_index=b["ligand_decomp_index"],
              ligand_fc_bond_index=b["ligand_fc_bond_index"], ligand_fc_bond_type=b["init_ligand_fc_bond_type"],
              batch_ligand_bond=b["batch_ligand_bond"])
    opt = torch.optim.Adam(ref.parameters(), lr=5e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        r = ref.get_diffusion_loss(**kw)
        loss = r["losses"]["pos"] + 100.0 * r["losses"]["v"] + 100.0 * r["losses"]["bond"]
        loss.backward()
        opt.step()
        return float(loss)

    step()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        l = step()
    dt = (time.perf_counter() - t0) / args.steps
    print(f"reference training step on {args.threads} CPU threads (B = {args.batch}, 300 + 30 atoms): {dt:.2f} s/step  (loss {l:.4f})")


if __name__ == "__main__":
    main()
