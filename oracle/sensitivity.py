"""ORACLE TOOLING (test infrastructure; never imported by the product).

How sensitive is the 1000-step reverse chain to implementation-level rounding?  Replays the committed 1000-step fixtures
with the oracle (bit-exact restatement of the reference) while adding a tiny Gaussian perturbation to the ligand
coordinates after every step — the size of the per-step difference any independent fp32 implementation has (the HIP
forward differs from the reference by ~2e-6 on x0) — and reports how far the chain ends from the unperturbed fixture.
With the shipped drift guidance the dynamics amplify such differences ~20x more than without.

usage: python -m oracle.sensitivity [sigma]      (two 1000-step CPU chains, several minutes each)
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_utils as GU                                       # noqa: E402
from decompdiff_amd import synth                                # noqa: E402
from oracle import diffusion as OD                              # noqa: E402


def main():
    sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 2e-6
    torch.set_num_threads(os.cpu_count())
    for name, seedpocket in (("traj1000_plain", 3), ("traj1000_drift", 5)):
        g = GU.load(name)
        cfg, sd = GU.weights(int(g["weight_seed"]))
        b = GU.batch_from_npz(g)
        n_data = int(b["batch_ligand"].max()) + 1
        torch.manual_seed(int(g["seed"]))
        synth.build_sampling_batch(synth.make_pocket_small(seedpocket), n_data)
        noise = synth.draw_step_noise(1000, b["init_ligand_pos"].size(0), b["init_ligand_fc_bond_type"].size(0))
        gen = torch.Generator().manual_seed(1)

        def hook(step, t, pos, v, bond, preds):
            pos.add_(torch.randn(pos.shape, generator=gen) * sigma)

        r = OD.sample_diffusion(sd, cfg, num_steps=1000, energy_drift_opt=json.loads(str(g["drift"])), noise=noise,
                                step_hook=hook, **b)
        every = int(g["every"])
        tp = torch.stack(r["pos_traj"]).numpy()[every - 1::every]
        err = np.abs(tp.astype(np.float64) - g["traj_pos"]).reshape(len(tp), -1).max(1)
        nv = int((torch.stack(r["v_traj"]).numpy()[every - 1::every] != g["traj_v"]).sum())
        nb = int((torch.stack(r["bond_traj"]).numpy()[every - 1::every] != g["traj_bond"]).sum())
        print(f"{name}: per-step coordinate perturbation sigma={sigma:g} -> max |pos - fixture| at the checkpoints: "
              + " ".join(f"{e:.2g}" for e in err) + f"; type mismatches v={nv} bond={nb}", flush=True)


if __name__ == "__main__":
    main()
