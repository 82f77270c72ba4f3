"""ORACLE TOOLING (survey container only — needs /root/reference; never shipped to or run on the GPU box).

Harness fixtures: runs the REFERENCE's own ``sample_diffusion_ligand_decomp``
(scripts/sample_diffusion_decomp.py:57-457, imported from where it lies) with the reference's own transforms
(utils/transforms.py) around a *recording* model, for every prior mode / atom-count mode, and stores

* the synthetic pocket ``data`` fields that went in,
* the keyword arguments the reference passed to ``model.sample_diffusion`` for every batch,
* the per-sample records it built from the (deterministic, fake) model output,

in ``tests/golden/harness_<case>.npz``.  ``torch_geometric`` is absent here, so ``Batch.from_data_list`` is the
documented-semantics shim in oracle/ref_shims.py driving the reference's ``ProteinLigandData.__inc__``.

usage:  python -m oracle.make_harness_golden
"""
from __future__ import annotations

import argparse
import importlib.util
import logging
import os
import pickle
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_shims                                    # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_utils import (GOLDEN as GOLDEN_DIR, NUM_CONFIG, LinearCountModel, RecordingModel, harness_cases,   # noqa: E402
                          make_pocket_fields)


def load_reference_script():
    ref_shims.install()
    spec = importlib.util.spec_from_file_location("ref_sample_script",
                                                  os.path.join(ref_shims.REFERENCE_ROOT, "scripts/sample_diffusion_decomp.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.logger = logging.getLogger("ref_harness")
    mod.args = argparse.Namespace(recon_with_bond=True)

    def no_recon(*a, **k):
        raise mod.recon.MolReconsError()
    mod.recon.reconstruct_from_generated_with_bond = no_recon       # RDKit/OpenBabel are not installed here
    return mod


def reference_data(fields):
    from utils.data import ProteinLigandData                    # noqa: the REFERENCE's class
    d = ProteinLigandData()
    for k in ("protein_pos", "protein_element", "protein_is_backbone", "protein_atom_to_aa_type", "pocket_atom_masks",
              "ligand_atom_mask", "ligand_pos"):
        d[k] = fields[k].clone()
    d.ligand_element = torch.full((fields["ligand_pos"].size(0),), 6, dtype=torch.long)
    d.num_arms, d.num_scaffold = int(fields["num_arms"]), int(fields["num_scaffold"])
    d.arms_prior = [(int(n), mu.clone(), cov.clone(), None, None) for n, mu, cov in fields["arms_prior"]]
    d.scaffold_prior = [(int(n), mu.clone(), cov.clone(), None, None) for n, mu, cov in fields["scaffold_prior"]]
    return d


def main():
    mod = load_reference_script()
    import utils.transforms as trans                            # noqa: the REFERENCE's transforms
    from torch_geometric.transforms import Compose              # (shim)
    arm_cfg = sca_cfg = NUM_CONFIG          # same layout as the reference's arm/scaffold_num_config.pkl, synthetic content
    for case in harness_cases():
        fields = make_pocket_fields(case["pocket_seed"], beta=case["prior_mode"] == "beta_prior",
                                    with_scaffold=case.get("with_scaffold", True), num_arms=case.get("num_arms", 2))
        data = reference_data(fields)
        data = Compose([trans.FeaturizeProteinAtom()])(data)
        indicator = trans.AddDecompIndicator(max_num_arms=10, global_prior_index=8, add_ord_feat=False)
        init_transform = Compose([trans.ComputeLigandAtomNoiseDist(version=case["prior_mode"]), indicator,
                                  trans.FeaturizeLigandBond(mode="fc", set_bond_type=False)])
        mod.full_protein_pos = fields["full_protein_pos"]
        model = RecordingModel()
        natoms_path = None
        if case["num_atoms_mode"] == "stat":
            tmp = tempfile.NamedTemporaryFile(suffix=".pkl", delete=False)
            pickle.dump({k: LinearCountModel(*v) for k, v in case["stat_models"].items()}, tmp)
            tmp.close()
            natoms_path = tmp.name
        torch.manual_seed(case["seed"])
        np.random.seed(case["seed"])
        results = mod.sample_diffusion_ligand_decomp(
            model, data, init_transform=init_transform, num_samples=case["num_samples"], batch_size=case["batch_size"],
            device="cpu", prior_mode=case["prior_mode"], num_steps=2, center_pos_mode="protein",
            num_atoms_mode=case["num_atoms_mode"],
            atom_prior_probs=np.array(case["atom_probs"]) if "atom_probs" in case else None,
            bond_prior_probs=np.array(case["bond_probs"]) if "bond_probs" in case else None,
            arms_natoms_config=arm_cfg, scaffold_natoms_config=sca_cfg, natoms_config=natoms_path,
            atom_enc_mode="add_aromatic", bond_fc_mode="fc", energy_drift_opt=None)
        if natoms_path:
            os.unlink(natoms_path)
        out = {"case": np.array(case["name"]), "n_batches": np.array(len(model.calls))}
        for bi, kw in enumerate(model.calls):
            for k, v in kw.items():
                if torch.is_tensor(v):
                    out[f"b{bi}_{k}"] = v.cpu().numpy()
        for si, r in enumerate(results):
            assert r["mol"] is None and r["smiles"] == ""
            for k in ("pred_pos", "pred_v", "pred_pos_traj", "pred_v_traj", "decomp_mask", "pred_bond_index",
                      "pred_bond_type"):
                out[f"s{si}_{k}"] = np.asarray(r[k])
        out["n_samples"] = np.array(len(results))
        sizes = [len(r["pred_v"]) for r in results]
        path = os.path.join(GOLDEN_DIR, f"harness_{case['name']}.npz")
        np.savez_compressed(path, **out)
        print(f"{case['name']:24s} batches={len(model.calls)} ligand sizes={sizes} -> {os.path.getsize(path)} B")


if __name__ == "__main__":
    main()
