"""Harness parity: our PyG-free batch assembly and unbatching against fixtures produced by the
REFERENCE's own ``sample_diffusion_ligand_decomp`` (scripts/sample_diffusion_decomp.py:57-457) run around the same
recording model — every prior mode and atom-count mode, including the ones whose samples differ in size.  Integer
tensors and random draws must match exactly, float tensors bit for bit (same torch CPU ops in the same order)."""
import numpy as np
import pytest
import torch

import golden_utils as GU
from decompdiff_amd import harness
from decompdiff_amd.pocket_data import NumAtomsSampler, PocketData

CASES = GU.harness_cases()


def _pocket(case):
    f = GU.make_pocket_fields(case["pocket_seed"], beta=case["prior_mode"] == "beta_prior",
                              with_scaffold=case.get("with_scaffold", True), num_arms=case.get("num_arms", 2))
    return PocketData(protein_pos=f["protein_pos"], protein_element=f["protein_element"],
                      protein_is_backbone=f["protein_is_backbone"], protein_atom_to_aa_type=f["protein_atom_to_aa_type"],
                      pocket_atom_masks=f["pocket_atom_masks"], num_arms=f["num_arms"], num_scaffold=f["num_scaffold"],
                      arms_prior=f["arms_prior"], scaffold_prior=f["scaffold_prior"], ligand_atom_mask=f["ligand_atom_mask"],
                      ligand_pos=f["ligand_pos"], full_protein_pos=f["full_protein_pos"])


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_harness_matches_reference(case):
    g = GU.load("harness_" + case["name"])
    model = GU.RecordingModel()
    sampler = None
    if case["num_atoms_mode"] == "stat":
        sampler = NumAtomsSampler({k: GU.LinearCountModel(*v) for k, v in case["stat_models"].items()})
    torch.manual_seed(case["seed"])
    np.random.seed(case["seed"])
    out = harness.sample_diffusion_ligand_decomp(
        model, _pocket(case), num_samples=case["num_samples"], batch_size=case["batch_size"], device="cpu", num_steps=2,
        center_pos_mode="protein", prior_mode=case["prior_mode"], num_atoms_mode=case["num_atoms_mode"],
        arms_natoms_config=GU.NUM_CONFIG, scaffold_natoms_config=GU.NUM_CONFIG, natoms_sampler=sampler,
        atom_prior_probs=np.array(case["atom_probs"]) if "atom_probs" in case else None,
        bond_prior_probs=np.array(case["bond_probs"]) if "bond_probs" in case else None)
    assert len(model.calls) == int(g["n_batches"])
    for bi, kw in enumerate(model.calls):
        ref_keys = {k[len(f"b{bi}_"):] for k in g.files if k.startswith(f"b{bi}_")}
        ours = {k for k, v in kw.items() if torch.is_tensor(v)}
        assert ref_keys == ours, (ref_keys ^ ours)
        for k in sorted(ref_keys):
            a, b = kw[k].numpy(), g[f"b{bi}_{k}"]
            assert a.dtype == b.dtype and a.shape == b.shape, (k, a.dtype, b.dtype, a.shape, b.shape)
            assert np.array_equal(a, b), k
        assert kw["num_steps"] == 2 and kw["center_pos_mode"] == "protein" and kw["energy_drift_opt"] is None
        assert kw["ligand_atom_mask"] is None
    recs = harness.to_result_records(out)
    assert len(recs) == int(g["n_samples"])
    for si, r in enumerate(recs):
        assert r["mol"] is None and r["smiles"] == ""
        for k in ("pred_pos", "pred_v", "pred_pos_traj", "pred_v_traj", "decomp_mask", "pred_bond_index", "pred_bond_type"):
            a, b = np.asarray(r[k]), g[f"s{si}_{k}"]
            assert a.shape == b.shape and a.dtype == b.dtype, (k, a.shape, b.shape, a.dtype, b.dtype)
            assert np.array_equal(a, b), k


def test_ragged_modes_really_are_ragged():
    sizes = {c["name"]: [g[f"s{i}_pred_v"].shape[0] for i in range(int(g["n_samples"]))]
             for c in CASES for g in [GU.load("harness_" + c["name"])]}
    assert len(set(sizes["beta_old"])) > 1 and len(set(sizes["subpocket_prior"])) > 1 and len(set(sizes["beta_stat"])) > 1
    assert len(set(sizes["ref_prior"])) == 1


@pytest.mark.parametrize("name", ["beta_old", "ref_prior"])
def test_pooled_batches_give_the_same_samples(name):
    """pool_batches > 1 collates several reference-order batches into one model call (so that the size groups of the
    ragged modes get larger); with a model that is a function of its inputs only, every per-sample result is unchanged."""
    case = [c for c in CASES if c["name"] == name][0]
    outs = []
    for pool in (1, 2):
        torch.manual_seed(case["seed"])
        np.random.seed(case["seed"])
        model = GU.RecordingModel()
        outs.append((harness.sample_diffusion_ligand_decomp(
            model, _pocket(case), num_samples=case["num_samples"], batch_size=case["batch_size"], device="cpu", num_steps=2,
            prior_mode=case["prior_mode"], num_atoms_mode=case["num_atoms_mode"], pool_batches=pool), len(model.calls)))
    (a, calls_a), (b, calls_b) = outs
    assert calls_a == 2 and calls_b == 1
    for k in ("pred_pos", "pred_v", "pred_pos_traj", "pred_v_traj", "pred_bond_index", "pred_bond_type", "pred_b_traj",
              "pred_bt_traj", "decomp_mask"):
        assert len(a[k]) == len(b[k]) == case["num_samples"]
        for x, y in zip(a[k], b[k]):
            assert np.array_equal(np.asarray(x), np.asarray(y)), k
