"""Sampling harness: the batch loop around ``model.sample_diffusion``.

Counterpart of ``sample_diffusion_ligand_decomp`` in
/root/reference/scripts/sample_diffusion_decomp.py:57-457 for the ``ref_prior`` / ``beta_prior(v2)`` modes
(equal atom counts per sample): per batch it draws the initial ligand coordinates around the arm / scaffold
prior centres, the initial bond and atom types (same torch-CPU RNG order as the reference, :163-194,306-312),
assembles the PyG-style flat batch (:300-326, increments of utils/data.py:439-444), calls
``model.sample_diffusion`` (:329-360) and splits the result per sample exactly like :361-410:

    pred_pos [NL,3] float64, pred_v [NL] int64, pred_pos_traj [T,NL,3] float64, pred_v_traj [T,NL],
    pred_v0_traj / pred_vt_traj [T,NL,8], pred_bond_index [2,Eb] (sample-local atom ids), pred_bond_type [Eb],
    pred_b_traj [T,Eb], pred_bt_traj [T,Eb,5], decomp_mask [NL] (arm id, -1 = scaffold)

RDKit reconstruction (:416-457) is a CPU chemistry step outside the hot path and is not done
here; the returned dicts carry everything it consumes.
"""
from __future__ import annotations

import time
from typing import Dict, List, Optional

import numpy as np
import torch

from . import synth
from .pocket_data import PocketData, build_batch


def unbatch_traj(traj: List[torch.Tensor], n_data: int, cum: np.ndarray, dtype=None, stacked=None) -> List[np.ndarray]:
    """reference: unbatch_v_traj (:46-53) — list over steps of [sum_n, ...] -> per sample [T, n_i, ...].  ``stacked``:
    the same trajectory as one [T, sum_n, ...] CPU tensor when the model provides it (no re-stacking; the per-sample
    arrays are then views of it)."""
    if len(traj) == 0:
        return [np.zeros((0,)) for _ in range(n_data)]
    stacked = stacked.numpy() if stacked is not None else torch.stack([t.cpu() for t in traj]).numpy()   # [T, sum_n, ...]
    if dtype is not None:
        stacked = stacked.astype(dtype)
    return [stacked[:, cum[k]:cum[k + 1]] for k in range(n_data)]


@torch.no_grad()
def sample_diffusion_ligand_decomp(model, pocket, num_samples: int, batch_size: int = 16,
                                   device="cuda:0", num_steps: Optional[int] = None, center_pos_mode: str = "protein",
                                   energy_drift_opt=None, per_sample_std_scale=None, noise_fn=None, seed: int = 0,
                                   use_graph: bool = True, prior_mode: str = "ref_prior", num_atoms_mode: str = "ref",
                                   arms_natoms_config=None, scaffold_natoms_config=None, natoms_sampler=None,
                                   atom_prior_probs=None, bond_prior_probs=None, pool_batches: int = 1) -> Dict[str, list]:
    """Sample ``num_samples`` ligands for one pocket in batches of ``batch_size``.

    ``pool_batches`` > 1 assembles that many consecutive batches exactly as the reference would (same random draws, same
    order) and hands them to the model as ONE collated batch: in the modes whose samples differ in size the model runs
    one dense group per distinct size, and pooling turns many groups of one or two samples into a few groups of several
    (4 pooled batches of 16 with 12 distinct sizes: 12 groups of ~5 instead of 4 x 12 groups of ~1).  Results come back
    in the reference's sample order; with device-generated noise the per-sample streams depend on the grouping, as the
    reference's depend on its batching.

    ``pocket`` is a :class:`synth.Pocket` (synthetic, ``ref_prior``-style batches with equal sizes) or a
    :class:`pocket_data.PocketData` (the reference's ``data`` fields; every ``prior_mode`` / ``num_atoms_mode`` of
    scripts/sample_diffusion_decomp.py:78-298, including the modes whose samples differ in size).

    ``noise_fn(batch_index, n_ligand_atoms, n_bonds, num_steps)`` may return pre-drawn reference-order noise for
    a batch (parity mode); otherwise the device Philox generator is used with ``seed + batch_index``.
    """
    out = {k: [] for k in ("pred_pos", "pred_v", "pred_pos_traj", "pred_v_traj", "pred_v0_traj", "pred_vt_traj",
                           "pred_bond_index", "pred_bond_type", "pred_b_traj", "pred_bt_traj", "decomp_mask")}
    time_list = []
    num_batch = int(np.ceil(num_samples / batch_size))
    pool_batches = max(1, int(pool_batches))
    steps = model.num_timesteps if num_steps is None else num_steps

    def build(i):
        """batch i as the reference assembles it: (kwargs, ligand sizes, injected noise or None)"""
        n_data = batch_size if i < num_batch - 1 else num_samples - batch_size * (num_batch - 1)
        if isinstance(pocket, PocketData):
            batch, n_atoms, _ = build_batch(pocket, n_data, prior_mode=prior_mode, num_atoms_mode=num_atoms_mode,
                                            num_classes=model.num_classes, num_bond_classes=model.num_bond_classes,
                                            arms_natoms_config=arms_natoms_config,
                                            scaffold_natoms_config=scaffold_natoms_config, natoms_sampler=natoms_sampler,
                                            atom_prior_probs=atom_prior_probs, bond_prior_probs=bond_prior_probs)
        else:
            scale = None
            if per_sample_std_scale is not None:
                scale = list(per_sample_std_scale[i * batch_size: i * batch_size + n_data])
            batch = synth.build_sampling_batch(pocket, n_data, num_bond_classes=model.num_bond_classes,
                                               num_classes=model.num_classes, per_sample_std_scale=scale)
            n_atoms = [pocket.num_ligand_atoms] * n_data
        noise = noise_fn(i, sum(n_atoms), sum(n * (n - 1) for n in n_atoms), steps) if noise_fn is not None else None
        return batch, n_atoms, noise

    for i in range(0, num_batch, pool_batches):
        parts = [build(j) for j in range(i, min(num_batch, i + pool_batches))]
        if len(parts) == 1:
            batch, n_atoms, noise = parts[0]
        else:                                                              # one collated batch (PyG increments re-applied)
            batch = synth.concat_sampling_batches([p[0] for p in parts])
            n_atoms = [n for p in parts for n in p[1]]
            noise = None if parts[0][2] is None else {k: torch.cat([p[2][k] for p in parts], 1) for k in parts[0][2]}
        n_data = len(n_atoms)
        n_bonds = [n * (n - 1) for n in n_atoms]
        dev_batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        extra = dict(noise=noise, seed=seed + i, use_graph=use_graph) if _accepts_extras(model) else {}
        t1 = time.time()
        r = model.sample_diffusion(num_steps=num_steps, center_pos_mode=center_pos_mode,
                                   energy_drift_opt=energy_drift_opt, **extra, **dev_batch)
        cum_atoms = np.cumsum([0] + n_atoms)                                   # (:367)
        cum_bonds = np.cumsum([0] + n_bonds)                                   # (:390-393)
        pos = r["pos"].cpu().numpy().astype(np.float64)
        v = r["v"].cpu().numpy()
        bond = r["bond"].cpu().numpy()
        bond_index = batch["ligand_fc_bond_index"].numpy()
        decomp = batch["ligand_decomp_index"].numpy()
        stk = r.get("_traj_stacked") or {}
        pos_traj = unbatch_traj(r["pos_traj"], n_data, cum_atoms, np.float64, stk.get("pos_traj"))
        v_traj = unbatch_traj(r["v_traj"], n_data, cum_atoms, None, stk.get("v_traj"))
        v0_traj = unbatch_traj(r["v0_traj"], n_data, cum_atoms, None, stk.get("v0_traj"))
        vt_traj = unbatch_traj(r["vt_traj"], n_data, cum_atoms, None, stk.get("vt_traj"))
        b_traj = unbatch_traj(r["bond_traj"], n_data, cum_bonds, None, stk.get("bond_traj"))
        bt_traj = unbatch_traj(r["bt_traj"], n_data, cum_bonds, None, stk.get("bt_traj"))
        for k in range(n_data):
            a0, a1, b0, b1 = cum_atoms[k], cum_atoms[k + 1], cum_bonds[k], cum_bonds[k + 1]
            out["pred_pos"].append(pos[a0:a1])
            out["pred_v"].append(v[a0:a1])
            out["pred_bond_index"].append(bond_index[:, b0:b1] - a0)          # sample-local atom ids (:395)
            out["pred_bond_type"].append(bond[b0:b1])
            out["decomp_mask"].append(decomp[a0:a1])
            out["pred_pos_traj"].append(pos_traj[k])
            out["pred_v_traj"].append(v_traj[k])
            out["pred_v0_traj"].append(v0_traj[k])
            out["pred_vt_traj"].append(vt_traj[k])
            out["pred_b_traj"].append(b_traj[k])
            out["pred_bt_traj"].append(bt_traj[k])
        time_list.append(time.time() - t1)
    out["time_list"] = time_list
    return out


def _accepts_extras(model) -> bool:
    """The HIP model takes ``noise / seed / use_graph``; a model with the bare reference signature does not."""
    import inspect
    try:
        return "use_graph" in inspect.signature(model.sample_diffusion).parameters
    except (TypeError, ValueError):
        return False


# index -> atomic number of the 8 atom classes, ligand_atom_mode 'basic' (utils/transforms.py:41-50,73-75)
ATOMIC_NUMBER_OF_CLASS = (1, 6, 7, 8, 9, 15, 16, 17)
# ligand_atom_mode 'add_aromatic': class -> (atomic number, aromatic), 13 classes (utils/transforms.py:52-66)
AROMATIC_CLASSES = ((1, False), (6, False), (6, True), (7, False), (7, True), (8, False), (8, True), (9, False), (15, False), (15, True),
                    (16, False), (16, True), (17, False))
# ligand_atom_mode 'full': class -> (atomic number, hybridisation, aromatic), 23 classes (utils/transforms.py:15-39)
FULL_CLASSES = ((1, "S", False), (6, "SP", False), (6, "SP2", False), (6, "SP2", True), (6, "SP3", False), (7, "SP", False),
                (7, "SP2", False), (7, "SP2", True), (7, "SP3", False), (8, "SP2", False), (8, "SP2", True), (8, "SP3", False),
                (9, "SP3", False), (15, "SP2", False), (15, "SP2", True), (15, "SP3", False), (15, "SP3D", False), (16, "SP2", False),
                (16, "SP2", True), (16, "SP3", False), (16, "SP3D", False), (16, "SP3D2", False), (17, "SP3", False))
NUM_CLASSES_OF_MODE = {"basic": 8, "add_aromatic": 13, "full": 23}


def atomic_numbers_from_index(pred_v, mode: str = "basic") -> List[int]:
    """trans.get_atomic_number_from_index(pred_v, mode) (utils/transforms.py:73-83)."""
    idx = [int(i) for i in np.asarray(pred_v).tolist()]
    if mode == "basic":
        return [ATOMIC_NUMBER_OF_CLASS[i] for i in idx]
    if mode == "add_aromatic":
        return [AROMATIC_CLASSES[i][0] for i in idx]
    if mode == "full":
        return [FULL_CLASSES[i][0] for i in idx]
    raise ValueError(mode)


def is_aromatic_from_index(pred_v, mode: str = "basic"):
    """trans.is_aromatic_from_index(pred_v, mode) (utils/transforms.py:86-95): None for 'basic'."""
    idx = [int(i) for i in np.asarray(pred_v).tolist()]
    if mode == "add_aromatic":
        return [AROMATIC_CLASSES[i][1] for i in idx]
    if mode == "full":
        return [FULL_CLASSES[i][2] for i in idx]
    if mode == "basic":
        return None
    raise ValueError(mode)


def bond_graph(pred_bond_index, pred_bond_type):
    """The undirected molecular graph the reference's reconstruction builds from the predicted fully connected bond list
    (utils/reconstruct.py:593-606: a directed entry (i, j) with i < j and type > 0 becomes one bond, type 1-4 = single,
    double, triple, aromatic): ``(bonds [(i, j, order)], n_fragments)`` without RDKit.  ``n_fragments == 1`` is the
    reference's "complete" criterion (``'.' not in smiles``, scripts/sample_diffusion_decomp.py:445)."""
    bi = np.asarray(pred_bond_index)
    bt = np.asarray(pred_bond_type)
    n = int(bi.max()) + 1 if bi.size else 0
    bonds = [(int(i), int(j), int(t)) for i, j, t in zip(bi[0], bi[1], bt) if i < j and t > 0]
    parent = list(range(n))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    for i, j, _ in bonds:
        parent[find(i)] = find(j)
    return bonds, len({find(a) for a in range(n)})


def to_result_records(out: Dict[str, list], ligand_filename: Optional[str] = None, reconstruct=None,
                      atom_enc_mode: str = "basic") -> List[dict]:
    """Per-sample records in the layout of the reference's ``result.pt`` (scripts/sample_diffusion_decomp.py:416-457,
    609-619): ``mol, smiles, pred_pos, pred_v, pred_pos_traj, pred_v_traj, decomp_mask, pred_bond_index (list),
    pred_bond_type`` (+ ``ligand_filename``), so that ``evaluate_mol_from_meta_full.py`` can consume them unchanged.

    Molecule reconstruction is RDKit/OpenBabel CPU chemistry outside the sampling hot path: pass the
    reference's ``recon.reconstruct_from_generated_with_bond`` (utils/reconstruct.py:579) -- or anything with its
    signature ``reconstruct(pred_pos, atomic_numbers, pred_bond_index, pred_bond_type) -> mol`` -- where those packages
    exist; a ``(mol, smiles)`` return is accepted too, otherwise ``smiles`` comes from ``Chem.MolToSmiles`` when RDKit is
    importable.  An exception from the callable is a failed reconstruction: ``mol`` None and ``smiles`` '' -- exactly
    what the reference stores (:440-443).  Without a callable the same placeholders are stored.  ``atom_enc_mode``: the
    ``ligand_atom_mode`` of the checkpoint (``basic`` / ``add_aromatic`` / ``full``: 8 / 13 / 23 atom classes, :422)."""
    records = []
    for i in range(len(out["pred_pos"])):
        bond_index = np.asarray(out["pred_bond_index"][i]).tolist()
        mol, smiles = None, ""
        if reconstruct is not None:
            try:
                r = reconstruct(out["pred_pos"][i], atomic_numbers_from_index(out["pred_v"][i], atom_enc_mode), bond_index,
                                out["pred_bond_type"][i])
                if isinstance(r, tuple):
                    mol, smiles = r
                else:
                    mol = r
                    try:
                        from rdkit import Chem                     # noqa: only where the chemistry stack exists
                        smiles = Chem.MolToSmiles(mol)
                    except ImportError:
                        smiles = ""
            except Exception:                                      # recon.MolReconsError in the reference
                mol, smiles = None, ""
        rec = {"mol": mol, "smiles": smiles, "pred_pos": out["pred_pos"][i], "pred_v": out["pred_v"][i],
               "pred_pos_traj": out["pred_pos_traj"][i], "pred_v_traj": out["pred_v_traj"][i],
               "decomp_mask": out["decomp_mask"][i], "pred_bond_index": bond_index,
               "pred_bond_type": out["pred_bond_type"][i]}
        if ligand_filename is not None:
            rec["ligand_filename"] = ligand_filename
        records.append(rec)
    return records


def save_result_pt(records: List[dict], path: str) -> None:
    """``torch.save(results, .../result.pt)`` (scripts/sample_diffusion_decomp.py:616-619): the list of per-sample dicts."""
    torch.save(records, path)


def load_result_pt(path: str) -> List[dict]:
    """What ``evaluate_mol_from_meta_full.py`` does first: ``torch.load(result.pt)`` (numpy arrays inside: full unpickling)."""
    return torch.load(path, weights_only=False)
