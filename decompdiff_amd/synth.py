"""Synthetic pockets, synthetic weights and the sampling-harness batch builder.

No CrossDocked data and no pretrained checkpoint exist in either container
(SURVEY.md §0, §8c), so every parity fixture, test and bench input is produced here
from seeded, version-stable formulas:

* :func:`make_pocket` — SURVEY.md §8d "Synthetic inputs": protein atoms uniform in a
  ball minus a cavity with a minimum spacing, PDB-like 3-decimal coordinates, 27+2
  protein features (utils/transforms.py:114-131,305-319), A arms + scaffold priors.
* :func:`synthetic_state_dict` — deterministic weights keyed by ``state_dict`` name
  (numpy ``default_rng([seed, crc32(key)])``), so the reference (in the survey
  container), the oracle and the HIP path all load bit-identical parameters without
  a weight blob being committed.
* :func:`build_sampling_batch` — what scripts/sample_diffusion_decomp.py:149-201
  (``ref_prior``) and :300-326 assemble before calling ``model.sample_diffusion``:
  same tensors, same torch-CPU RNG draw order, PyG collate increments of
  utils/data.py:439-444 applied by hand.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from .config import NUM_ATOM_CLASSES


# --------------------------------------------------------------------------------------
# pockets
# --------------------------------------------------------------------------------------
@dataclass
class Pocket:
    """One synthetic protein pocket + decomposed ligand prior (one `data` item)."""

    protein_pos: np.ndarray            # [NP,3] float32
    protein_atom_feature: np.ndarray   # [NP,29] float32 (27 + 2 arm indicator)
    full_protein_pos: np.ndarray       # [NF,3] float32 (for the clash drift)
    arm_num_atoms: List[int]           # atoms per arm
    scaffold_num_atoms: int
    prior_centers: np.ndarray          # [A+1,3] float32, arms then scaffold
    prior_stds: np.ndarray             # [A+1,3] float32
    seed: int = 0
    meta: Dict = field(default_factory=dict)

    @property
    def num_arms(self) -> int:
        return len(self.arm_num_atoms)

    @property
    def num_protein_atoms(self) -> int:
        return int(self.protein_pos.shape[0])

    @property
    def num_ligand_atoms(self) -> int:
        return int(sum(self.arm_num_atoms) + self.scaffold_num_atoms)


def _sample_shell(rng, n, r_in, r_out, min_dist, max_tries=200000, restart_after=None):
    """Rejection-sample n points in a spherical shell with a minimum spacing.  ``restart_after``: start over when that
    many batches of candidates in a row placed nothing (a few widely spaced points -- the prior centres -- can paint
    themselves into a corner; seeds that never got stuck draw exactly the same points as before)."""
    pts = np.zeros((0, 3), dtype=np.float64)
    tries = stuck = 0
    while pts.shape[0] < n:
        tries += 1
        stuck += 1
        if restart_after is not None and stuck > restart_after:
            pts = np.zeros((0, 3), dtype=np.float64)
            stuck = 0
        if tries > max_tries:
            raise RuntimeError("pocket generator: could not place atoms; enlarge the ball")
        cand = rng.uniform(-r_out, r_out, size=(64, 3))
        rr = np.linalg.norm(cand, axis=1)
        cand = cand[(rr <= r_out) & (rr >= r_in)]
        for c in cand:
            if pts.shape[0] == 0 or np.min(np.linalg.norm(pts - c, axis=1)) >= min_dist:
                pts = np.vstack([pts, c[None]])
                stuck = 0
                if pts.shape[0] == n:
                    break
    return pts


def make_pocket(seed: int = 0, num_protein: int = 300, arm_atoms=(8, 8), scaffold_atoms: int = 14,
                num_full_protein: int = 3000, prior_std: float = 1.2) -> Pocket:
    """SURVEY.md §8d pocket generator (C-small defaults; C-large = 600, (15,15), 30)."""
    rng = np.random.default_rng(seed)
    scale = (num_protein / 300.0) ** (1.0 / 3.0)
    r_out, r_in = 12.0 * scale, 4.0 * scale
    shift = rng.uniform(-20.0, 20.0, size=(1, 3))          # pockets are not at the origin
    ppos = _sample_shell(rng, num_protein, r_in, r_out, 1.2) + shift
    ppos = np.round(ppos, 3).astype(np.float32)

    elem = rng.choice(6, size=num_protein, p=[0.0, 0.63, 0.17, 0.18, 0.02, 0.0])
    res = rng.integers(0, 20, size=num_protein)
    bb = rng.random(num_protein) < 0.5
    arm_ind = rng.random(num_protein) < 0.4                 # "near an arm sub-pocket"
    feat = np.zeros((num_protein, 29), dtype=np.float32)
    feat[np.arange(num_protein), elem] = 1.0
    feat[np.arange(num_protein), 6 + res] = 1.0
    feat[:, 26] = bb
    feat[np.arange(num_protein), 27 + arm_ind.astype(np.int64)] = 1.0

    n_prior = len(arm_atoms) + 1
    centers = _sample_shell(rng, n_prior, 0.0, 0.75 * r_in, 0.85 * r_in, restart_after=200) + shift
    stds = np.full((n_prior, 3), prior_std, dtype=np.float32)

    n_extra = max(num_full_protein - num_protein, 0)
    extra = _sample_shell(rng, n_extra, r_out, 25.0 * scale, 1.0) + shift if n_extra else np.zeros((0, 3))
    full = np.concatenate([ppos.astype(np.float64), extra], 0)
    full = np.round(full, 3).astype(np.float32)

    return Pocket(protein_pos=ppos, protein_atom_feature=feat, full_protein_pos=full,
                  arm_num_atoms=[int(a) for a in arm_atoms], scaffold_num_atoms=int(scaffold_atoms),
                  prior_centers=centers.astype(np.float32), prior_stds=stds, seed=seed,
                  meta=dict(r_in=r_in, r_out=r_out))


def make_pocket_small(seed=0):
    """BASELINE config C-small: 300 protein + 30 ligand atoms (arms 8+8, scaffold 14)."""
    return make_pocket(seed, 300, (8, 8), 14)


def make_pocket_large(seed=0):
    """BASELINE config C-large: 600 protein + 60 ligand atoms (arms 15+15, scaffold 30)."""
    return make_pocket(seed, 600, (15, 15), 30, num_full_protein=4000)


def make_pocket_tiny(seed=0, num_protein=40, arm_atoms=(2, 2), scaffold_atoms=2):
    """Reduced-size pocket for fast oracle/golden tests (SURVEY.md §8c fixture 3)."""
    return make_pocket(seed, num_protein, arm_atoms, scaffold_atoms, num_full_protein=120)


# --------------------------------------------------------------------------------------
# harness: batch assembly (ref_prior) with the reference's RNG draw order
# --------------------------------------------------------------------------------------
def fc_bond_index(n_atoms: int) -> torch.Tensor:
    """Fully-connected directed bond list, dst-major (utils/transforms.py:331-337)."""
    dst = torch.repeat_interleave(torch.arange(n_atoms), n_atoms)
    src = torch.arange(n_atoms).repeat(n_atoms)
    keep = dst != src
    return torch.stack([src[keep], dst[keep]], 0)


def gumbel_argmax_uniform(n_rows: int, n_classes: int) -> torch.Tensor:
    """`log_sample_categorical(zeros)` (models/transitions.py:78-84) on the global CPU RNG."""
    u = torch.rand(n_rows, n_classes)
    g = -torch.log(-torch.log(u + 1e-30) + 1e-30)
    return g.argmax(-1)


def build_sampling_batch(pocket: Pocket, n_data: int, num_bond_classes: int = 5,
                         num_classes: int = NUM_ATOM_CLASSES, per_sample_std_scale=None,
                         device="cpu") -> Dict[str, Optional[torch.Tensor]]:
    """kwargs for ``model.sample_diffusion`` for ``n_data`` samples of one pocket.

    Mirrors the ``ref_prior`` branch of scripts/sample_diffusion_decomp.py:149-201 and
    the batch assembly at :300-326 (draws from torch's *global CPU* generator in the
    reference order: per sample {arm randn..., scaffold randn, bond-type Gumbel
    uniforms}, then one atom-type Gumbel draw for the whole batch).  PyG ``Batch``
    increments (utils/data.py:439-444) are applied explicitly.

    ``per_sample_std_scale`` (len n_data) reproduces the ``beta_prior`` flavour where
    every sample carries its own prior stds (sample_diffusion_decomp.py:203-295).
    """
    A = pocket.num_arms
    NL = pocket.num_ligand_atoms
    NP = pocket.num_protein_atoms
    centers = torch.from_numpy(pocket.prior_centers)
    base_stds = torch.from_numpy(pocket.prior_stds)

    init_pos, bond_types, stds_all = [], [], []
    decomp_index = []
    for a in range(A):
        decomp_index += [a] * pocket.arm_num_atoms[a]
    decomp_index += [-1] * pocket.scaffold_num_atoms
    decomp_index = torch.tensor(decomp_index, dtype=torch.long)
    decomp_mask = decomp_index.clone()
    decomp_mask[decomp_mask == -1] = A
    fc = fc_bond_index(NL)
    n_bond = fc.size(1)

    for s in range(n_data):
        stds = base_stds * (float(per_sample_std_scale[s]) if per_sample_std_scale is not None else 1.0)
        parts = []
        for a in range(A):
            parts.append(centers[a] + torch.randn(pocket.arm_num_atoms[a], 3) * stds[a].unsqueeze(0))
        parts.append(centers[-1] + torch.randn(pocket.scaffold_num_atoms, 3) * stds[-1].unsqueeze(0))
        bond_types.append(gumbel_argmax_uniform(n_bond, num_bond_classes))
        init_pos.append(torch.cat(parts, 0))
        stds_all.append(stds)

    batch_ligand = torch.repeat_interleave(torch.arange(n_data), NL)
    init_ligand_v = gumbel_argmax_uniform(n_data * NL, num_classes)

    aux = torch.nn.functional.one_hot((decomp_index >= 0).long(), 2).float()
    NF = pocket.full_protein_pos.shape[0]
    out = dict(
        protein_pos=torch.from_numpy(pocket.protein_pos).repeat(n_data, 1),
        protein_v=torch.from_numpy(pocket.protein_atom_feature).repeat(n_data, 1),
        batch_protein=torch.repeat_interleave(torch.arange(n_data), NP),
        protein_group_idx=torch.full((n_data * NP,), -1, dtype=torch.long),
        init_ligand_pos=torch.cat(init_pos, 0),
        init_ligand_v=init_ligand_v,
        ligand_v_aux=aux.repeat(n_data, 1),
        batch_ligand=batch_ligand,
        ligand_group_idx=torch.cat([decomp_mask + s * (A + 1) for s in range(n_data)]),
        ligand_atom_mask=None,
        prior_centers=centers.repeat(n_data, 1),
        prior_stds=torch.cat(stds_all, 0).float(),
        prior_num_atoms=torch.tensor((pocket.arm_num_atoms + [pocket.scaffold_num_atoms]) * n_data),
        batch_prior=torch.repeat_interleave(torch.arange(n_data), A + 1),
        prior_group_idx=torch.cat([torch.arange(A + 1) for _ in range(n_data)]),
        ligand_fc_bond_index=torch.cat([fc + s * NL for s in range(n_data)], 1),
        init_ligand_fc_bond_type=torch.cat(bond_types, 0),
        batch_ligand_bond=torch.repeat_interleave(torch.arange(n_data), n_bond),
        ligand_decomp_batch=torch.cat([decomp_mask + s * (A + 1) for s in range(n_data)]),
        ligand_decomp_index=decomp_index.repeat(n_data),
        full_protein_pos=torch.from_numpy(pocket.full_protein_pos).repeat(n_data, 1),
        full_batch_protein=torch.repeat_interleave(torch.arange(n_data), NF),
    )
    if device != "cpu":
        out = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in out.items()}
    return out


def concat_sampling_batches(batches) -> Dict[str, Optional[torch.Tensor]]:
    """Collate several ``build_sampling_batch`` results (different pockets / ligand sizes) into ONE flat batch, the way
    PyG's ``Batch.from_data_list`` does for the reference (utils/data.py:389-446: sample ids are renumbered, index
    fields are incremented by the running atom / prior-row counts).  The result is a *ragged* batch."""
    out: Dict[str, Optional[torch.Tensor]] = {}
    n_s = n_lig = n_prior = 0
    cat: Dict[str, list] = {}
    for b in batches:
        ns = int(b["batch_protein"].max()) + 1
        inc = {"batch_protein": n_s, "batch_ligand": n_s, "batch_prior": n_s, "batch_ligand_bond": n_s,
               "full_batch_protein": n_s, "ligand_fc_bond_index": n_lig, "ligand_decomp_batch": n_prior,
               "ligand_group_idx": n_prior}
        for k, v in b.items():
            if not torch.is_tensor(v):
                continue
            cat.setdefault(k, []).append(v + inc[k] if k in inc else v)
        n_s += ns
        n_lig += b["batch_ligand"].numel()
        n_prior += b["batch_prior"].numel()
    for k, vs in cat.items():
        out[k] = torch.cat(vs, 1 if k == "ligand_fc_bond_index" else 0)
    out["ligand_atom_mask"] = None
    return out


def ragged_demo_batch(seed: int) -> Dict[str, Optional[torch.Tensor]]:
    """A small ragged batch (two pocket / ligand sizes, groups interleaved: 1 + 2 + 1 samples) — the input of the
    ``traj10_ragged`` golden fixture (oracle/make_golden.py) and of the tests that replay it."""
    pa = make_pocket_tiny(11, num_protein=48, arm_atoms=(3, 2), scaffold_atoms=3)       # NL = 8
    pb = make_pocket_tiny(12, num_protein=40, arm_atoms=(2, 2), scaffold_atoms=2)       # NL = 6
    torch.manual_seed(seed)
    return concat_sampling_batches([build_sampling_batch(p, n) for p, n in ((pa, 1), (pb, 2), (pa, 1))])


def draw_step_noise(num_steps: int, n_ligand: int, n_bond: int, num_classes: int = NUM_ATOM_CLASSES,
                    num_bond_classes: int = 5):
    """Pre-draw the per-step noise exactly as the reference loop consumes it.

    Per step (models/decompdiff.py:620,633,680): ``rand_like([n_ligand,K])`` for atom
    types, ``rand_like([n_bond,Kb])`` for bond types, ``randn_like([n_ligand,3])`` for
    positions — from torch's global CPU generator, in that order.  Returns stacked
    tensors ``u_v [T,n_ligand,K]``, ``u_b [T,n_bond,Kb]``, ``eps [T,n_ligand,3]``.
    """
    u_v = torch.empty(num_steps, n_ligand, num_classes)
    u_b = torch.empty(num_steps, n_bond, num_bond_classes)
    eps = torch.empty(num_steps, n_ligand, 3)
    for s in range(num_steps):
        u_v[s] = torch.rand(n_ligand, num_classes)
        u_b[s] = torch.rand(n_bond, num_bond_classes)
        eps[s] = torch.randn(n_ligand, 3)
    return dict(u_v=u_v, u_b=u_b, eps=eps)


# --------------------------------------------------------------------------------------
# synthetic weights
# --------------------------------------------------------------------------------------
def learnable_param_shapes(config, protein_atom_feature_dim=29, ligand_atom_feature_dim=10,
                           num_classes=NUM_ATOM_CLASSES) -> "Dict[str, tuple]":
    """Names/shapes of the trainable tensors of the reference model for ``uni_o2_bond``.

    Follows the module tree of models/decompdiff.py:149-211 and
    models/encoders/uni_transformer_edge.py:16-347.  The order is
    irrelevant for ``load_state_dict``; tests pin the *set* against the key list captured
    from the reference (tests/golden/state_dict_spec.json).
    """
    H = config.hidden_dim
    nh = config.n_heads
    emb = H - 1 if config.node_indicator else H
    nb = getattr(config, "num_bond_classes", 1)
    G = 20  # GaussianSmearing(fix_offset=True) always has 20 centres (models/common.py:16-19)
    ef = G * config.edge_feat_dim + config.edge_feat_dim

    shapes: Dict[str, tuple] = {}

    def linear(name, out_d, in_d):
        shapes[f"{name}.weight"] = (out_d, in_d)
        shapes[f"{name}.bias"] = (out_d,)

    def mlp(name, in_d, out_d):
        linear(f"{name}.net.0", H, in_d)
        shapes[f"{name}.net.1.weight"] = (H,)
        shapes[f"{name}.net.1.bias"] = (H,)
        linear(f"{name}.net.3", out_d, H)

    linear("protein_atom_emb", emb, protein_atom_feature_dim)
    linear("ligand_atom_emb", emb, ligand_atom_feature_dim)
    linear("ligand_bond_emb", H, nb)
    mlp("refine_net.edge_pred_layer", config.num_r_gaussian, 1)
    for l in range(config.num_layers):
        p = f"refine_net.base_block.{l}"
        linear(f"{p}.lin_node", H, H)
        for f in ("hk_func", "hv_func"):
            mlp(f"{p}.node_layer_with_edge.{f}", 2 * H + ef, H)
        mlp(f"{p}.node_layer_with_edge.hq_func", H, H)
        for f in ("hk_func", "hv_func"):
            mlp(f"{p}.node_layer_with_bond.{f}", 3 * H, H)
        mlp(f"{p}.node_layer_with_bond.hq_func", H, H)
        kv_in = H + 2 * G + 13 + (2 * H if config.h_node_in_bond_net else 0)
        q_in = H + (H if config.h_node_in_bond_net else 0)
        for f in ("hk_func", "hv_func"):
            mlp(f"{p}.bond_layer.{f}", kv_in, H)
        mlp(f"{p}.bond_layer.hq_func", q_in, H)
        mlp(f"{p}.pos_layer_with_edge.xk_func", 2 * H + ef, H)
        mlp(f"{p}.pos_layer_with_edge.xv_func", 2 * H + ef, nh)
        mlp(f"{p}.pos_layer_with_edge.xq_func", H, H)
        mlp(f"{p}.pos_layer_with_bond.xk_func", 3 * H, H)
        mlp(f"{p}.pos_layer_with_bond.xv_func", 3 * H, nh)
        mlp(f"{p}.pos_layer_with_bond.xq_func", H, H)
    linear("v_inference.0", H, H)
    linear("v_inference.2", num_classes, H)
    if getattr(config, "bond_diffusion", False):
        linear("bond_inference.0", H, H)
        linear("bond_inference.2", nb, H)
    return shapes


def synthetic_tensor(key: str, shape, seed: int = 0) -> torch.Tensor:
    """Version-stable synthetic value for one learnable tensor."""
    rng = np.random.default_rng([int(seed), zlib.crc32(key.encode("utf-8"))])
    if len(shape) == 2:                       # Linear weight
        fan_in = shape[1]
        w = rng.standard_normal(shape) / np.sqrt(fan_in)
        if ".xv_func.net.3" in key:
            w *= 0.25                          # keep per-layer coordinate updates moderate
    elif key.endswith(".net.1.weight"):       # LayerNorm gamma
        w = 1.0 + 0.1 * rng.standard_normal(shape)
    else:                                      # biases / LayerNorm beta
        w = 0.1 * rng.standard_normal(shape)
    return torch.from_numpy(w.astype(np.float32))


def synthetic_state_dict(config, seed: int = 0, **dims) -> Dict[str, torch.Tensor]:
    """Deterministic weights for every learnable tensor (schedule tables are not included:
    they are computed by the model constructor)."""
    return {k: synthetic_tensor(k, s, seed) for k, s in learnable_param_shapes(config, **dims).items()}
