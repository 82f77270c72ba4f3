"""Helpers shared by the CPU (oracle) and GPU (HIP) parity tests: golden fixture loading."""
import json
import os

import numpy as np
import torch

from decompdiff_amd import synth
from decompdiff_amd.config import shipped_config

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DRIFT = [dict(type="armsca_prox", min_d=1.2, max_d=1.9), dict(type="clash", sigma=2, gamma=4)]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def batch_from_npz(g, prefix="in_", device="cpu"):
    out = {}
    for k in g.files:
        if not k.startswith(prefix):
            continue
        a = g[k]
        t = torch.from_numpy(a.astype(np.float32) if k.endswith("protein_v") else a)
        out[k[len(prefix):]] = t.to(device)
    out.setdefault("ligand_atom_mask", None)
    return out


def weights(seed=0, cfg=None):
    cfg = cfg or shipped_config()
    return cfg, synth.synthetic_state_dict(cfg, seed=seed)


def noise_for(seed, pocket_builder, n_data, num_steps, std_scale=None):
    """Re-create (batch, noise) exactly as oracle/make_golden.py drew them."""
    torch.manual_seed(seed)
    batch = synth.build_sampling_batch(pocket_builder, n_data, per_sample_std_scale=std_scale)
    noise = synth.draw_step_noise(num_steps, batch["init_ligand_pos"].size(0), batch["init_ligand_fc_bond_type"].size(0))
    return batch, noise


def checksum(noise):
    return np.array([float(noise["u_v"].double().sum()), float(noise["u_b"].double().sum()),
                     float(noise["eps"].double().sum())])
