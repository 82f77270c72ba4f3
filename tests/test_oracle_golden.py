"""CPU: the oracle (oracle/*.py) replayed against fixtures generated from the REFERENCE itself
(oracle/make_golden.py).  Bit-exact on CPU: the oracle is a restatement, not an approximation."""
import json
import os

import numpy as np
import pytest
import torch

import golden_utils as GU
from decompdiff_amd import synth
from decompdiff_amd.config import shipped_config
from oracle import diffusion as OD
from oracle import model as OM
from oracle import ops


def _fwd(sd, cfg, b, trace=None):
    with torch.no_grad():
        return OM.forward(sd, cfg, b["protein_pos"], b["protein_v"], b["batch_protein"], b["init_ligand_pos"],
                          b["init_ligand_v"], b["ligand_v_aux"], b["batch_ligand"], b["ligand_fc_bond_index"],
                          b["init_ligand_fc_bond_type"], trace=trace)


def test_schedule_tables_match_reference():
    g = GU.load("schedules")
    cfg = shipped_config()
    pt = OD.position_tables(cfg)
    vt = OD.categorical_tables(cfg, 8)
    bt = OD.categorical_tables(cfg, 5)
    for k, v in pt.items():
        assert np.array_equal(g[k], v.numpy()), k
    for k, v in vt.items():
        assert np.array_equal(g["atom_type_trans__" + k], v.numpy()), k
    for k, v in bt.items():
        assert np.array_equal(g["bond_type_trans__" + k], v.numpy()), k
    # known-answer values, SURVEY.md Appendix A.1
    assert abs(float(pt["betas"][0]) - 5.04499894e-06) < 1e-12
    assert abs(float(pt["posterior_logvar"][0]) - (-12.8844023)) < 1e-5
    assert float(pt["posterior_logvar"][0]) == float(pt["posterior_logvar"][1])
    assert abs(float(vt["log_alphas_cumprod_v"][999]) - (-9.91987991)) < 1e-5


def test_forward_tiny_with_layer_intermediates():
    g = GU.load("forward_tiny")
    over = json.loads(str(g["cfg_json"]))
    cfg = shipped_config(**over)
    sd = synth.synthetic_state_dict(cfg, seed=int(g["weight_seed"]))
    b = GU.batch_from_npz(g)
    trace = []
    out = _fwd(sd, cfg, b, trace)
    for k in ("pred_ligand_pos", "pred_ligand_v", "pred_bond"):
        assert np.array_equal(g["out_" + k], out[k].numpy()), k
    for l in range(cfg.num_layers):
        assert np.array_equal(g[f"layer{l}_h"], trace[1 + l]["h"].numpy())
        assert np.array_equal(g[f"layer{l}_h_bond"], trace[1 + l]["h_bond"].numpy())
        assert np.array_equal(g[f"layer{l}_x"], trace[1 + l]["x"].numpy())


def test_forward_full_size():
    g = GU.load("forward_small")
    cfg, sd = GU.weights(int(g["weight_seed"]))
    out = _fwd(sd, cfg, GU.batch_from_npz(g))
    for k in ("pred_ligand_pos", "pred_ligand_v", "pred_bond"):
        assert np.array_equal(g["out_" + k], out[k].numpy()), k


def test_single_steps_all_times_plain_and_drift():
    g = GU.load("steps")
    cfg, sd = GU.weights(int(g["weight_seed"]))
    base = GU.batch_from_npz(g)
    for t_start in (999, 500, 1, 0):
        for tag, drift in (("plain", None), ("drift", GU.DRIFT)):
            p = f"t{t_start}_{tag}_"
            b = dict(base)
            for k in ("init_ligand_pos", "init_ligand_v", "init_ligand_fc_bond_type", "prior_stds"):
                b[k] = torch.from_numpy(g[p + "in_" + k])
            torch.manual_seed(int(g[p + "seed"]))
            synth.build_sampling_batch(synth.make_pocket_small(1), 2,
                                       per_sample_std_scale=[1.0, 0.8] if drift else None)   # advance the RNG
            noise = synth.draw_step_noise(1, b["init_ligand_pos"].size(0), b["init_ligand_fc_bond_type"].size(0))
            assert GU.same_checksum(GU.checksum(noise), g[p + "noise_checksum"])
            r = OD.sample_diffusion(sd, cfg, num_steps=1, energy_drift_opt=drift, noise=noise, t_start=t_start, **b)
            assert np.array_equal(g[p + "pos"], r["pos"].numpy()), p
            assert np.array_equal(g[p + "v"], r["v"].numpy()), p
            assert np.array_equal(g[p + "bond"], r["bond"].numpy()), p
            assert np.array_equal(g[p + "log_v_prob"], r["vt_traj"][0].numpy()), p
            assert np.array_equal(g[p + "log_b_prob"], r["bt_traj"][0].numpy()), p


def _replay_traj(name, num_steps, std_scale=None):
    g = GU.load(name)
    cfg, sd = GU.weights(int(g["weight_seed"]))
    b = GU.batch_from_npz(g)
    drift = json.loads(str(g["drift"]))
    n_data = int(b["batch_ligand"].max()) + 1
    torch.manual_seed(int(g["seed"]))
    synth.build_sampling_batch(_pocket_for(name), n_data, per_sample_std_scale=std_scale)
    nc = int(g["num_classes"]) if "num_classes" in g.files else 8
    if nc != 8:                                    # ligand_atom_mode add_aromatic / full: wider embedding + v head
        sd = synth.synthetic_state_dict(cfg, seed=int(g["weight_seed"]), ligand_atom_feature_dim=nc + 2, num_classes=nc)
        torch.manual_seed(int(g["seed"]))
        synth.build_sampling_batch(_pocket_for(name), n_data, per_sample_std_scale=std_scale, num_classes=nc)
    noise = synth.draw_step_noise(int(g["num_steps"]), b["init_ligand_pos"].size(0),
                                  b["init_ligand_fc_bond_type"].size(0), num_classes=nc)
    assert GU.same_checksum(GU.checksum(noise), g["noise_checksum"])
    noise = {k: v[:num_steps] for k, v in noise.items()}
    priors = {k: g[k] for k in ("prior_atom_types", "prior_bond_types") if k in g.files}
    if nc != 8:
        priors["num_classes"] = nc
    t_start = int(g["t_start"]) if "t_start" in g.files else cfg.num_diffusion_timesteps - 1
    r = OD.sample_diffusion(sd, cfg, num_steps=num_steps, energy_drift_opt=drift, noise=noise,
                            t_start=t_start, **priors, **b)
    return g, r


def _pocket_for(name):
    return {"traj20_plain": synth.make_pocket_small(2), "traj20_drift": synth.make_pocket_small(2),
            "traj1000_plain": synth.make_pocket_small(3), "traj12_priortypes": synth.make_pocket_small(4),
            "traj1000_drift": synth.make_pocket_small(5),
            "traj3_scale": synth.make_pocket(41, 80, (3, 3), 4, num_full_protein=200),
            "traj3_b16": synth.make_pocket(7, 347, (9, 9), 19, num_full_protein=0),
            "traj3_b8_plain": synth.make_pocket_small(8), "traj3_b8_drift": synth.make_pocket_small(8),
            "traj4_aromatic13": synth.make_pocket_small(9), "traj4_full23": synth.make_pocket_small(9),
            "traj3_large_drift": synth.make_pocket_large(6),
            "traj3_nl80": synth.make_pocket(13, 120, (27, 27), 26, num_full_protein=300)}[name]


def test_trajectory_20_steps_plain():
    g, r = _replay_traj("traj20_plain", 20)
    assert np.array_equal(g["out_pos"], r["pos"].numpy())
    assert np.array_equal(g["out_v"], r["v"].numpy())
    assert np.array_equal(g["out_bond"], r["bond"].numpy())
    assert np.array_equal(g["traj_pos"], torch.stack(r["pos_traj"]).numpy())


def test_trajectory_20_steps_drift():
    g, r = _replay_traj("traj20_drift", 20, std_scale=[1.0, 0.85])
    assert np.array_equal(g["out_pos"], r["pos"].numpy())
    assert np.array_equal(g["out_v"], r["v"].numpy())
    assert np.array_equal(g["out_bond"], r["bond"].numpy())


def test_trajectory_12_steps_prior_types():
    """Non-uniform class priors of the categorical transitions (DecompScorePosNet3D(prior_atom_types=, prior_bond_types=),
    transitions.py:118-120): fixture from a reference model built with them."""
    g, r = _replay_traj("traj12_priortypes", 12)
    assert np.array_equal(g["out_pos"], r["pos"].numpy())
    assert np.array_equal(g["out_v"], r["v"].numpy())
    assert np.array_equal(g["out_bond"], r["bond"].numpy())
    assert np.array_equal(g["traj_v"], torch.stack(r["v_traj"]).numpy().astype(np.int8))


def test_trajectory_drift_scale_option():
    """`scale: True` drift terms (decompdiff.py:656-657,667-668) at t = 600..598, fixture from the reference."""
    g, r = _replay_traj("traj3_scale", 3, std_scale=[1.0, 0.8])
    assert np.array_equal(g["out_pos"], r["pos"].numpy())
    assert np.array_equal(g["out_v"], r["v"].numpy()) and np.array_equal(g["out_bond"], r["bond"].numpy())


B8_STD = [1.0, 0.9, 0.8, 1.1, 1.0, 0.95, 1.05, 0.85]


@pytest.mark.parametrize("name,std_scale", [("traj3_b16", None), ("traj3_large_drift", [1.0, 0.9]), ("traj3_nl80", [1.0, 0.9]),
                                            ("traj3_b8_plain", None), ("traj3_b8_drift", B8_STD),
                                            ("traj4_aromatic13", [1.0, 0.9]), ("traj4_full23", [1.0, 0.9])])
def test_trajectory_bench_config_shapes_first_step(name, std_scale):
    """BASELINE configs[1] / configs[2] at the exact bench shape (C-small 300 + 30, B=8, plain / drift), configs[3] /
    configs[4] shapes (NP=347, NL=37, B=16; 600 + 60 atoms with drift), and the 13- / 23-class atom vocabularies of
    ligand_atom_mode add_aromatic / full (reference models of those widths): the first step of the
    reference's 3-step fixtures (the full 3 steps are replayed by the HIP path in tests/test_gpu_configs.py; the oracle
    reproduced all 3 bit for bit when the fixture was generated: `oracle_vs_reference_maxabs` = 0)."""
    g, r = _replay_traj(name, 1, std_scale=std_scale)
    assert float(g["oracle_vs_reference_maxabs"]) == 0.0
    assert np.array_equal(g["traj_pos"][0], r["pos_traj"][0].numpy())
    assert np.array_equal(g["traj_v"][0], r["v_traj"][0].numpy().astype(np.int8))
    assert np.array_equal(g["traj_bond"][0], r["bond_traj"][0].numpy().astype(np.int8))


@pytest.mark.parametrize("name", ["traj1000_plain", "traj1000_drift"])
def test_trajectory_1000_first_checkpoint(name):
    """The 1000-step goldens store every 50th state; replay the first 50 steps (the full chains are
    replayed on the GPU by tests/test_gpu_parity.py)."""
    if not os.path.exists(os.path.join(GU.GOLDEN, name + ".npz")):
        pytest.skip(f"{name}.npz not generated yet (python -m oracle.make_golden --only {name})")
    g, r = _replay_traj(name, 50)
    assert np.array_equal(g["traj_pos"][0], r["pos_traj"][49].numpy())
    assert np.array_equal(g["traj_v"][0], r["v_traj"][49].numpy().astype(np.int8))
    assert np.array_equal(g["traj_bond"][0], r["bond_traj"][49].numpy().astype(np.int8))


def test_injected_noise_equals_global_rng_draws():
    """rand(n,k)/randn(n,3) pre-draws are the same stream rand_like/randn_like consume in the loop."""
    cfg, sd = GU.weights(0)
    pocket = synth.make_pocket_tiny(7, num_protein=48, arm_atoms=(3, 2), scaffold_atoms=3)
    torch.manual_seed(5)
    b = synth.build_sampling_batch(pocket, 2)
    state = torch.get_rng_state()
    r1 = OD.sample_diffusion(sd, cfg, num_steps=3, **b)
    torch.set_rng_state(state)
    noise = synth.draw_step_noise(3, b["init_ligand_pos"].size(0), b["init_ligand_fc_bond_type"].size(0))
    r2 = OD.sample_diffusion(sd, cfg, num_steps=3, noise=noise, **b)
    assert torch.equal(r1["pos"], r2["pos"]) and torch.equal(r1["v"], r2["v"]) and torch.equal(r1["bond"], r2["bond"])


def test_bond_triplets_order_and_counts():
    fc = synth.fc_bond_index(5)
    i, j, idx_i, idx_j, idx_k, idx_kj, idx_ji = ops.bond_triplets(fc, 5)
    assert idx_ji.numel() == 5 * 4 * 3
    assert torch.all(idx_i != idx_k) and torch.all(idx_j != idx_k)
    # segments are contiguous, sorted, fixed length NL-2, k ascending inside a segment
    assert torch.equal(idx_ji, torch.arange(20).repeat_interleave(3))
    seg_k = idx_k.view(20, 3)
    assert torch.all(seg_k[:, 1:] > seg_k[:, :-1])
    # edge (k -> j) id in the dst-major layout: j*(NL-1) + k - (k > j)
    assert torch.equal(idx_kj, idx_j * 4 + idx_k - (idx_k > idx_j).long())


def test_knn_graph_semantics():
    torch.manual_seed(0)
    x = torch.randn(30, 3)
    batch = torch.cat([torch.zeros(18, dtype=torch.long), torch.ones(12, dtype=torch.long)])
    ei = ops.knn_graph(x, k=5, batch=batch)
    assert ei.shape == (2, 30 * 5)
    src, dst = ei
    assert torch.all(batch[src] == batch[dst]) and torch.all(src != dst)
    assert torch.equal(dst, torch.arange(30).repeat_interleave(5))
    d = (x[src] - x[dst]).norm(dim=-1).view(30, 5)
    assert torch.all(d[:, 1:] >= d[:, :-1])
    # fewer than k candidates -> all of them
    ei2 = ops.knn_graph(x[:4], k=5)
    assert ei2.shape == (2, 4 * 3)


def test_trajectory_ragged_batch():
    """Ragged batch (different protein / ligand sizes per sample): the oracle follows the reference's
    segmented ops, fixture generated by the reference itself (oracle/make_golden.py --only ragged)."""
    g = GU.load("traj10_ragged")
    batch = synth.ragged_demo_batch(int(g["seed"]))
    noise = synth.draw_step_noise(int(g["num_steps"]), batch["init_ligand_pos"].size(0), batch["init_ligand_fc_bond_type"].size(0))
    assert GU.same_checksum(GU.checksum(noise), g["noise_checksum"])
    stored = GU.batch_from_npz(g)
    for k, v in stored.items():
        if torch.is_tensor(v):
            assert torch.equal(v, batch[k].to(v.dtype)), k
    cfg, sd = GU.weights(0)
    r = OD.sample_diffusion(sd, cfg, num_steps=int(g["num_steps"]), energy_drift_opt=json.loads(str(g["drift"])), noise=noise, **batch)
    assert np.array_equal(g["out_pos"], r["pos"].numpy())
    assert np.array_equal(g["out_v"], r["v"].numpy()) and np.array_equal(g["out_bond"], r["bond"].numpy())
    assert np.array_equal(g["traj_pos"], torch.stack(r["pos_traj"]).numpy())


def test_arms_repul_energy_and_gradient_match_reference():
    """tests/golden/arms_repul.npz holds the REFERENCE's compute_batch_arms_repul_loss and its autograd
    gradient (oracle/make_golden.py gen_arms_repul); the oracle's restatement reproduces both bit for bit."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "arms_repul.npz"))
    cases = sorted({k.split("/")[0] for k in g.files if "/" in k})
    assert cases == ["c_small", "no_arms", "three_arms_gap", "two_arms"]
    n_active = 0
    for c in cases:
        pos, batch, dec = (torch.from_numpy(g[f"{c}/{k}"]) for k in ("pos", "batch_ligand", "decomp_index"))
        for mode in ("min", "all"):
            for max_d in (1.9, 3.0):
                key = f"{mode}_{max_d}"
                xt = pos.clone().requires_grad_(True)
                e, nv = OD.arms_repul_loss(xt, batch, dec, max_d, mode)
                assert nv == int(g[f"{c}/n_valid_{key}"])
                assert float(e.detach()) == float(g[f"{c}/energy_{key}"])
                grad = torch.autograd.grad(e, xt)[0] if (nv > 0 and e.requires_grad) else torch.zeros_like(pos)
                assert torch.equal(grad, torch.from_numpy(g[f"{c}/grad_{key}"])), (c, key)
                n_active += int(grad.abs().max() > 0)
    assert n_active >= 12                                     # the hinge is active in most cases (not a vacuous fixture)
