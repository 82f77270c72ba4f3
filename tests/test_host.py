"""CPU: host logic of the product package — checkpoint key layout, schedule tables, weight
packing + the restructured algebra (vs the oracle), the C-ABI library's exports, and the
'no CPU fallback' contract."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import dense_spec as DS
import golden_utils as GU
from decompdiff_amd import DecompScorePosNet3D, hip_lib, packing, shipped_config, synth
from oracle import model as OM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_layout_matches_reference_checkpoint(golden_dir):
    spec = json.load(open(os.path.join(golden_dir, "state_dict_spec.json")))
    m = DecompScorePosNet3D(shipped_config(), 29, 10, 8)
    sd = m.state_dict()
    assert set(sd) == set(spec) and len(sd) == 616
    for k, v in sd.items():
        assert list(v.shape) == spec[k]["shape"], k
    trainable = {n for n, p in m.named_parameters() if p.requires_grad}
    assert trainable == {k for k, v in spec.items() if v["trainable"]}
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 4975269
    # strict load of a reference-shaped checkpoint
    full = {k: torch.zeros(v["shape"]) for k, v in spec.items()}
    m.load_state_dict(full, strict=True)


def test_schedule_tables_equal_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "schedules.npz"))
    sd = DecompScorePosNet3D(shipped_config(), 29, 10, 8).state_dict()
    for k in g.files:
        assert np.array_equal(g[k], sd[k.replace("__", ".")].numpy()), k
    assert DecompScorePosNet3D(shipped_config(), 29, 10, 8).num_timesteps == 1000


def test_unsupported_configs_raise():
    for over in (dict(model_type="uni_o2"), dict(cutoff_mode="radius"), dict(hidden_dim=64), dict(time_emb_dim=8),
                 dict(add_prior_node=True), dict(bond_diffusion=False)):
        with pytest.raises(NotImplementedError):
            DecompScorePosNet3D(shipped_config(**over), 29, 10, 8)
    with pytest.raises(NotImplementedError):                  # the step kernels and buffers are 8 atom classes wide
        DecompScorePosNet3D(shipped_config(), 29, 10, 13)


def test_packing_slots_match_c_enum():
    hdr = open(os.path.join(ROOT, "include", "decompdiff_hip.h")).read()
    body = re.search(r"typedef enum dd_wslot \{(.*?)DD_NUM_LAYER_SLOTS", hdr, re.S).group(1)
    names = [n.strip()[3:] for n in body.replace("\n", " ").split(",") if n.strip()]
    assert names == packing.LAYER_SLOTS
    body = re.search(r"typedef enum dd_gslot \{(.*?)DD_NUM_GLOBAL_SLOTS", hdr, re.S).group(1)
    names = [n.strip()[5:] for n in body.replace("\n", " ").split(",") if n.strip()]
    assert names == packing.GLOBAL_SLOTS


def test_packed_arena_offsets_aligned_and_complete():
    cfg = shipped_config()
    arena, offs, named = packing.pack_model(synth.synthetic_state_dict(cfg, 0), cfg)
    assert len(offs) == cfg.num_layers * len(packing.LAYER_SLOTS) + len(packing.GLOBAL_SLOTS)
    assert all(int(o) % 64 == 0 for o in offs)
    for (key, t), o in zip(named.items(), offs.tolist()):
        assert torch.equal(arena[o:o + t.numel()], t.reshape(-1)), key
    # every learnable reference tensor ends up in the arena exactly once (bias of W2k is dropped: it cancels in the softmax)
    total = sum(t.numel() for t in named.values())
    assert 5.6e6 < total < 6.2e6


@pytest.mark.parametrize("NPn,arms,sca,B", [(48, (3, 2), 3, 2), (20, (1, 1), 1, 1)])
def test_restructured_algebra_matches_oracle(NPn, arms, sca, B):
    """packing.py + tests/dense_spec.py (the math the HIP kernels implement) == the oracle."""
    cfg, sd = GU.weights(0)
    pocket = synth.make_pocket(13, NPn, arms, sca, num_full_protein=NPn + 8)
    torch.manual_seed(4)
    b = synth.build_sampling_batch(pocket, B)
    NL = pocket.num_ligand_atoms
    with torch.no_grad():
        want = OM.forward(sd, cfg, b["protein_pos"], b["protein_v"], b["batch_protein"], b["init_ligand_pos"],
                          b["init_ligand_v"], b["ligand_v_aux"], b["batch_ligand"], b["ligand_fc_bond_index"],
                          b["init_ligand_fc_bond_type"])
        _, _, named = packing.pack_model(sd, cfg)
        x0, vl, bl, _ = DS.forward_dense(named, cfg, b["protein_pos"].view(B, NPn, 3), b["protein_v"].view(B, NPn, 29),
                                         b["init_ligand_pos"].view(B, NL, 3), b["init_ligand_v"].view(B, NL),
                                         b["ligand_v_aux"].view(B, NL, 2), b["init_ligand_fc_bond_type"].view(B, -1))
    assert float((x0.reshape(-1, 3) - want["pred_ligand_pos"]).abs().max()) < 2e-5
    assert float((vl.reshape(-1, 8) - want["pred_ligand_v"]).abs().max()) < 2e-5
    assert float((bl.reshape(-1, 5) - want["pred_bond"]).abs().max()) < 2e-5


def test_library_loads_and_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    lib = hip_lib.load()
    for header, symbols in (("decompdiff_hip.h", hip_lib.EXPORTED_SYMBOLS), ("decompdiff_hip_debug.h", hip_lib.DEBUG_SYMBOLS)):
        hdr = open(os.path.join(ROOT, "include", header)).read()
        hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)       # (comments mention entry points of the other header)
        declared = set(re.findall(r"\b(dd_[a-z0-9_]+)\s*\(", hdr))
        assert declared == set(symbols), (header, declared ^ set(symbols))
        for name in declared:
            if name in hip_lib.MEASUREMENT_ONLY_SYMBOLS:         # declared under #if DD_DEBUG_OPTIONS: not in the default library
                assert not hasattr(lib, name), name
            else:
                assert hasattr(lib, name), name
    dbg = ctypes.CDLL(os.path.join(ROOT, "decompdiff_amd", "lib", "libdecompdiff_hip_dbg.so"))
    assert all(hasattr(dbg, n) for n in hip_lib.MEASUREMENT_ONLY_SYMBOLS) and dbg.dd_build_flags() & 1
    # size_t results keep their width (a c_int restype would truncate them silently)
    assert lib.dd_workspace_floats.restype is ctypes.c_size_t and lib.dd_gemm128_tn_scratch_floats.restype is ctypes.c_size_t
    assert not any(n.startswith("dd_debug") or n.startswith("dd_profile") for n in hip_lib.EXPORTED_SYMBOLS)
    assert lib.dd_build_flags() == 0                         # the default library: no measurement variants compiled in
    assert lib.dd_abi_version() == hip_lib.ABI_VERSION == 9 and lib.dd_weights_form() == 1   # (ABI 9: the attention MLPs in kernel form)
    assert lib.dd_status_string(0) == b"ok" and b"workspace" in lib.dd_status_string(-3)
    # pure host helper: workspace size grows with the batch and is non-zero
    w1, w8 = lib.dd_workspace_floats(1, 300, 30, 32), lib.dd_workspace_floats(8, 300, 30, 32)
    assert 0 < w1 < w8 and lib.dd_workspace_floats(0, 300, 30, 32) == 0
    assert ctypes.sizeof(hip_lib.DDSampler) % 8 == 0


def test_no_cpu_fallback():
    """CPU tensors must raise; the product never routes through the oracle or torch-CPU math."""
    m = DecompScorePosNet3D(shipped_config(), 29, 10, 8)
    pocket = synth.make_pocket_tiny(1)
    b = synth.build_sampling_batch(pocket, 1)
    with pytest.raises(hip_lib.HipLibraryError):
        m.sample_diffusion(num_steps=1, center_pos_mode="protein", **b)
    src = "".join(open(os.path.join(ROOT, "decompdiff_amd", f)).read()
                  for f in os.listdir(os.path.join(ROOT, "decompdiff_amd")) if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src
    # the op-level wrappers refuse CPU tensors too
    from decompdiff_amd import functional as F2
    x = torch.randn(10, 3)
    with pytest.raises(hip_lib.HipLibraryError):
        F2.knn_graph(x, 4)
    with pytest.raises(hip_lib.HipLibraryError):
        F2.scatter_attention(torch.randn(2, 128), torch.randn(4, 128), torch.randn(4, 128), torch.tensor([0, 0, 1, 1]), 2)
    with pytest.raises(hip_lib.HipLibraryError):
        F2.scatter_softmax(torch.randn(4, 16), torch.tensor([0, 0, 1, 1]), dim=0)
    # ... and so do the dispatcher ops (registered for the HIP device type only)
    import decompdiff_amd.torch_ops  # noqa: F401
    for name in ("knn_graph", "scatter_attention", "scatter_attention_pos", "scatter_sum", "scatter_mean", "scatter_min",
                 "scatter_softmax"):
        assert hasattr(torch.ops.decompdiff_amd, name), name
    with pytest.raises(NotImplementedError):
        torch.ops.decompdiff_amd.scatter_sum(torch.randn(4, 2), torch.tensor([0, 0, 1, 1]), 0, 2)
    # null / bad arguments are status codes, not crashes (host-side checks: no GPU needed)
    lib = hip_lib.load()
    assert lib.dd_segment_reduce(None, None, 4, 16, 0, 0, None, None, None) != 0
    assert lib.dd_segment_softmax(None, None, 4, 0, None, None) != 0
    assert lib.dd_debug_philox(1, 0, 0, 1, None, None) != 0
    assert lib.dd_graph_launch(None, 1, None) != 0 and lib.dd_graph_destroy(None) != 0
    assert lib.dd_sample_steps_graph_multi(None, 0, 1, None) != 0
    assert lib.dd_reverse_step(None, None, None, None, None) != 0
    assert lib.dd_debug_node_split(8, 300, 30, 32) == -1          # nothing measured in this process


def test_harness_batch_builder_layout():
    pocket = synth.make_pocket_small(0)
    torch.manual_seed(0)
    b = synth.build_sampling_batch(pocket, 3)
    NL, NP, Eb = 30, 300, 870
    assert b["init_ligand_pos"].shape == (3 * NL, 3) and b["ligand_fc_bond_index"].shape == (2, 3 * Eb)
    assert int(b["ligand_fc_bond_index"].max()) == 3 * NL - 1
    assert torch.equal(b["ligand_decomp_batch"][NL:2 * NL], b["ligand_decomp_batch"][:NL] + 3)      # PyG __inc__ = num_arms+1
    assert torch.equal(b["ligand_decomp_index"][:NL], b["ligand_decomp_index"][NL:2 * NL])           # not incremented
    assert set(b["ligand_decomp_index"].tolist()) == {-1, 0, 1}
    assert b["prior_stds"].shape == (9, 3) and b["protein_v"].shape == (3 * NP, 29)
    assert int(b["init_ligand_v"].max()) < 8 and int(b["init_ligand_fc_bond_type"].max()) < 5
    # protein atoms keep a 1.2 A minimum spacing and PDB-like 3-decimal coordinates
    d = torch.cdist(torch.from_numpy(pocket.protein_pos), torch.from_numpy(pocket.protein_pos)) + 10 * torch.eye(NP)
    assert float(d.min()) > 1.19
    assert np.allclose(pocket.protein_pos, np.round(pocket.protein_pos, 3), atol=1e-6)


def test_result_records_layout():
    """harness.to_result_records: the reference's result.pt keys (sample_diffusion_decomp.py:444-456), no GPU needed."""
    from decompdiff_amd.harness import to_result_records
    NL, Eb, T = 4, 12, 3
    out = {"pred_pos": [np.zeros((NL, 3))] * 2, "pred_v": [np.zeros(NL, dtype=np.int64)] * 2,
           "pred_pos_traj": [np.zeros((T, NL, 3))] * 2, "pred_v_traj": [np.zeros((T, NL), dtype=np.int64)] * 2,
           "pred_bond_index": [np.zeros((2, Eb), dtype=np.int64)] * 2, "pred_bond_type": [np.zeros(Eb, dtype=np.int64)] * 2,
           "decomp_mask": [np.array([0, 0, 1, -1])] * 2}
    recs = to_result_records(out, ligand_filename="x/y.sdf", reconstruct=lambda p, v, bi, bt: ("MOL", "C"))
    assert len(recs) == 2
    assert set(recs[0]) == {"mol", "smiles", "pred_pos", "pred_v", "pred_pos_traj", "pred_v_traj", "decomp_mask",
                            "pred_bond_index", "pred_bond_type", "ligand_filename"}
    assert recs[0]["mol"] == "MOL" and recs[0]["smiles"] == "C" and isinstance(recs[0]["pred_bond_index"], list)
    assert to_result_records(out)[1]["mol"] is None and to_result_records(out)[1]["smiles"] == ""


def test_result_pt_round_trip_and_reconstruction_hand_off(tmp_path):
    """result.pt as the reference writes it (torch.save of the record list, sample_diffusion_decomp.py:
    616-619) read back as evaluate_mol_from_meta_full.py does; the reconstruction callable gets the reference's arguments
    (positions, ATOMIC NUMBERS, bond index list, bond types: sample_diffusion_decomp.py:421-430); a raising callable is a
    failed reconstruction (mol None, smiles ''); the bond graph / completeness criterion works without RDKit."""
    from decompdiff_amd import harness
    rng = np.random.default_rng(0)
    NL, T = 5, 2
    fc = synth.fc_bond_index(NL).numpy()
    btype = np.zeros(fc.shape[1], dtype=np.int64)
    for (i, j, t) in ((0, 1, 1), (1, 2, 2), (3, 4, 4)):                      # two fragments: {0,1,2} and {3,4}
        btype[(fc[0] == i) & (fc[1] == j)] = t
        btype[(fc[0] == j) & (fc[1] == i)] = t
    out = {"pred_pos": [rng.normal(size=(NL, 3))], "pred_v": [np.array([1, 2, 3, 1, 7])],
           "pred_pos_traj": [rng.normal(size=(T, NL, 3))], "pred_v_traj": [rng.integers(0, 8, (T, NL))],
           "pred_bond_index": [fc], "pred_bond_type": [btype], "decomp_mask": [np.array([0, 0, 1, -1, -1])]}
    seen = {}

    def recon(pos, atomic_nums, bond_index, bond_type):
        seen.update(pos=pos, nums=atomic_nums, bi=bond_index, bt=bond_type)
        return "MOL"
    recs = harness.to_result_records(out, ligand_filename="a/b.sdf", reconstruct=recon)
    assert seen["nums"] == [6, 7, 8, 6, 17] and isinstance(seen["bi"], list) and recs[0]["mol"] == "MOL"
    bonds, n_frag = harness.bond_graph(out["pred_bond_index"][0], out["pred_bond_type"][0])
    assert sorted(bonds) == [(0, 1, 1), (1, 2, 2), (3, 4, 4)] and n_frag == 2

    def failing(*a):
        raise RuntimeError("MolReconsError")
    assert harness.to_result_records(out, reconstruct=failing)[0]["mol"] is None
    path = str(tmp_path / "result.pt")
    harness.save_result_pt(recs, path)
    back = harness.load_result_pt(path)
    assert len(back) == 1 and set(back[0]) == set(recs[0])
    for k, v in recs[0].items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(back[0][k], v), k
        else:
            assert back[0][k] == v, k


def test_training_batch_layout_check():
    """training.check_batch_layout (run by get_diffusion_loss for both network paths): sizes of a ragged batch, and the
    layouts that would give wrong triplets are refused."""
    import pytest
    from decompdiff_amd import training
    b = synth.ragged_demo_batch(3)
    n_p, n_l = training.check_batch_layout(b["batch_protein"], b["batch_ligand"], b["ligand_fc_bond_index"], b["batch_ligand_bond"])
    assert n_p == [48, 40, 40, 48] and n_l == [8, 6, 6, 8]
    with pytest.raises(NotImplementedError, match="dst-major"):
        training.check_batch_layout(b["batch_protein"], b["batch_ligand"], b["ligand_fc_bond_index"].flip(0))
    with pytest.raises(NotImplementedError, match="sorted"):
        training.check_batch_layout(b["batch_protein"], b["batch_ligand"].flip(0), b["ligand_fc_bond_index"])
    with pytest.raises(NotImplementedError, match="batch_ligand_bond"):
        training.check_batch_layout(b["batch_protein"], b["batch_ligand"], b["ligand_fc_bond_index"], b["batch_ligand_bond"].flip(0))


def test_sensitivity_fixture_and_class_tables():
    """The oracle self-divergence fixture behind the bound of the free-running 1000-step drift chain (8 replays of the
    oracle with +-1 ulp nudges per step, oracle/make_sensitivity.py), and the class tables of the three ligand_atom_modes
    (utils/transforms.py:15-95; checked against the reference's functions when the fixture was added)."""
    g = GU.load("sens_traj1000_drift")
    e = g["pos_err"]
    assert e.shape == (8, 20) and str(g["fixture"]) == "traj1000_drift" and int(g["every"]) == 50
    assert np.array_equal(g["pos_err_median"], np.median(e, 0)) and (e[:, 0] < 1e-5).all()
    assert np.median(e, 0)[-1] > 1e-3 > np.median(e, 0)[5]          # chaotic tail: 1 ulp per step ends > 1e-3 from the reference
    assert int(g["v_mismatch"].sum()) == 0 and int(g["bond_mismatch"].sum(1).max()) > 0      # (one replay flips bond types)
    from decompdiff_amd import harness as H
    assert H.atomic_numbers_from_index(np.arange(8)) == [1, 6, 7, 8, 9, 15, 16, 17]
    assert H.atomic_numbers_from_index(np.arange(13), "add_aromatic") == [1, 6, 6, 7, 7, 8, 8, 9, 15, 15, 16, 16, 17]
    assert H.is_aromatic_from_index(np.arange(13), "add_aromatic") == [False, False, True, False, True, False, True, False, False, True,
                                                                        False, True, False]
    assert len(H.FULL_CLASSES) == 23 and H.atomic_numbers_from_index([0, 3, 12, 22], "full") == [1, 6, 9, 17]
    assert H.is_aromatic_from_index([3, 4], "full") == [True, False] and H.is_aromatic_from_index([1], "basic") is None


def test_compiled_torch_extension_registers_the_op_level_boundary():
    """csrc/torch_ext.cpp: TORCH_LIBRARY(decompdiff_hip) built by build(); every op has a HIP(=CUDA-key) kernel and no CPU one."""
    import __graft_entry__
    __graft_entry__.build()
    from decompdiff_amd import functional as Fn
    ops = Fn.torch_ext()
    assert ops is not None, "lib/decompdiff_torch_ext.so did not load"
    assert int(ops.abi_version()) == hip_lib.ABI_VERSION
    for name in ("knn", "segment_reduce", "segment_softmax", "attn_aggregate_node", "attn_aggregate_pos"):
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f"decompdiff_hip::{name}", "CUDA"), name
        assert not torch._C._dispatch_has_kernel_for_dispatch_key(f"decompdiff_hip::{name}", "CPU"), name
    with pytest.raises(NotImplementedError):
        ops.knn(torch.zeros(1, 4, 3), 2)                    # no CPU implementation: the dispatcher refuses


def test_sample_time_methods_match_reference():
    """training.sample_time against the REFERENCE's sample_time (tests/golden/sample_time.npz, oracle/make_golden.py
    gen_sample_time): 'symmetric'; 'importance' with the never-written history buffers (falls back to 'symmetric', which is
    what the reference always does); 'importance' with the buffers filled by the host (torch.multinomial branch)."""
    from decompdiff_amd import training
    g = np.load(os.path.join(ROOT, "tests", "golden", "sample_time.npz"))
    m = DecompScorePosNet3D(shipped_config(), 29, 10, 8)
    for tag, method, fill in (("symmetric", "symmetric", False), ("importance_empty", "importance", False),
                              ("importance_filled", "importance", True)):
        m.Lt_history.zero_(); m.Lt_count.zero_()
        if fill:
            m.Lt_history.copy_(torch.from_numpy(g["Lt_history_filled"])); m.Lt_count.fill_(11)
        for n in (4, 7):
            torch.manual_seed(100 + n)
            ts, pt = training.sample_time(m, n, "cpu", method)
            assert np.array_equal(ts.numpy(), g[f"{tag}/{n}/time_step"]), (tag, n)
            assert np.array_equal(pt.numpy(), g[f"{tag}/{n}/pt"]), (tag, n)
    assert not np.array_equal(g["importance_filled/7/time_step"], g["importance_empty/7/time_step"])
    with pytest.raises(ValueError):
        training.sample_time(m, 4, "cpu", "uniform")


def test_kernel_form_of_the_attention_mlps_is_exact_algebra():
    """packing.kernel_form_layer (mean-free first-Linear parts, sign / |gamma| of the LayerNorm moved into the neighbouring
    Linears, softmax scale inside W2k): for every key / value MLP of the five attention sub-layers the kernels' evaluation
    max(fma(P', rstd, beta'), 0) followed by the scaled second Linear equals relu(LayerNorm(P)) followed by the reference's --
    including negative and exactly-zero gammas."""
    cfg, sd = GU.weights(0)
    sd = {k: v.clone() for k, v in sd.items()}
    gen = torch.Generator().manual_seed(5)
    for k in sd:
        if k.endswith(".net.1.weight") and "base_block.0." in k:
            g = sd[k]
            g[torch.randperm(128, generator=gen)[:40]] *= -1.0          # negative gammas
            g[torch.randperm(128, generator=gen)[:5]] = 0.0             # and a few exact zeros
    can = packing.pack_layer({k: v.float() for k, v in sd.items()}, "refine_net.base_block.0")
    ker = packing.kernel_form_layer(can)
    assert list(ker.keys()) == packing.LAYER_SLOTS and all(ker[k].shape == can[k].shape for k in can)
    worst = 0.0
    for m in packing._KERNEL_FORM_MLPS:
        rows = 64
        xs = [torch.randn(rows, 128, generator=gen, dtype=torch.float64) for _ in m["rows"]]
        cs = [torch.randn(rows, can[t].reshape(-1, 128).shape[0], generator=gen, dtype=torch.float64) for t in m["tabs"]]

        def pre(slots):
            P = torch.zeros(rows, 128, dtype=torch.float64)
            for x, (wn, bn, r0) in zip(xs, m["rows"]):
                P = P + x @ slots[wn][r0:r0 + 128].double().t() + slots[bn][r0:r0 + 128].double()
            for c, t in zip(cs, m["tabs"]):
                P = P + c @ slots[t].reshape(-1, 128).double()
            return P
        P, Pk = pre(can), pre(ker)
        ln = can[m["ln"]].double()
        z = torch.relu(torch.nn.functional.layer_norm(P, (128,), ln[0], ln[1], 1e-5))
        # mean-free up to the signs: E[P'^2] is the variance of the reference's row
        assert float(((Pk * Pk).mean(-1) - P.var(-1, unbiased=False)).abs().max()) < 1e-5
        zk = torch.relu(Pk * torch.rsqrt((Pk * Pk).mean(-1, keepdim=True) + 1e-5) + ker[m["ln"]][1].double())
        for (w2n, axis, fac) in m["w2"]:
            W, Wk = can[w2n].double(), ker[w2n].double()
            y = fac * (z @ (W.t() if axis == 1 else W))
            yk = zk @ (Wk.t() if axis == 1 else Wk)
            err = float((y - yk).abs().max()) / max(1.0, float(y.abs().max()))
            worst = max(worst, err)
            assert err < 2e-6, (m["ln"], w2n, err)
    print(f"kernel form vs canonical: worst relative difference {worst:.2g}")
