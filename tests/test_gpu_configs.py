"""GPU (-m gpu): the BASELINE.json configurations the round-1 tests did not exercise, the segment-wise bound on the
1000-step chains, the statistics of the production (Philox) noise, and the multi-rank bench entry.
Everything goes through the C ABI; fixtures under tests/golden come from the reference itself (oracle/make_golden.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import golden_utils as GU
from decompdiff_amd import hip_lib, synth
from test_gpu_parity import POS_TOL, _sample_hip, dev, maxabs, model

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fixture_chain(name, pocket, n_data, std_scale=None):
    """(golden, batch, noise) of a gen_traj fixture: the batch is stored, the noise is re-drawn (checksum pinned)."""
    g = GU.load(name)
    b = GU.batch_from_npz(g)
    torch.manual_seed(int(g["seed"]))
    synth.build_sampling_batch(pocket, n_data, per_sample_std_scale=std_scale)
    noise = synth.draw_step_noise(int(g["num_steps"]), b["init_ligand_pos"].size(0), b["init_ligand_fc_bond_type"].size(0))
    assert GU.same_checksum(GU.checksum(noise), g["noise_checksum"])
    return g, b, noise


def _check_chain(name, r, g, steps):
    tp = torch.stack(r["pos_traj"]).numpy()
    per_step = np.abs(tp.astype(np.float64) - g["traj_pos"]).reshape(steps, -1).max(1)
    nv = int((torch.stack(r["v_traj"]).numpy() != g["traj_v"]).sum())
    nb = int((torch.stack(r["bond_traj"]).numpy() != g["traj_bond"]).sum())
    err = maxabs(r["pos"], g["out_pos"])
    print(f"{name}: per-step max pos err {' '.join(f'{e:.2g}' for e in per_step)}; final {err:.3g}; type mismatches v={nv} bond={nb}")
    assert err < POS_TOL and per_step.max() < POS_TOL
    assert nv == 0 and nb == 0
    assert np.array_equal(r["v"].cpu().numpy(), g["out_v"]) and np.array_equal(r["bond"].cpu().numpy(), g["out_bond"])


@pytest.mark.parametrize("name,std_scale", [("traj3_b8_plain", None), ("traj3_b8_drift", [1.0, 0.9, 0.8, 1.1, 1.0, 0.95, 1.05, 0.85])])
def test_config1_config2_exact_bench_shape_reference_golden(name, std_scale):
    """BASELINE configs[1] (and configs[2]: + armsca / clash drift) at the EXACT shape the metric is quoted on -- C-small
    300 + 30 atoms, batch of 8 -- 3 reverse steps against the reference's own output (oracle/make_golden.py --only b8)."""
    g, b, noise = _fixture_chain(name, synth.make_pocket_small(8), 8, std_scale)
    assert b["init_ligand_pos"].shape[0] == 8 * 30 and b["protein_pos"].shape[0] == 8 * 300
    r = _sample_hip(model(0), b, 3, json.loads(str(g["drift"])), noise)
    _check_chain(f"configs[1/2] exact shape ({name}: NP=300, NL=30, B=8)", r, g, 3)
    assert hip_lib.load().dd_debug_node_split(8, 300, 30, 32) >= 0          # the per-shape launch measurement ran


def _bench_shape_chain(name, drift_scale):
    """One of the two free-running bench-shape chains (300 + 30 atoms, B = 8, 1000 steps on the reference's injected noise) against the
    reference's own trajectory, judged against the LIKE-FOR-LIKE envelope of the same fixture:
    tests/golden/sens_<name>.npz = the oracle's code run on a third, independent fp32 implementation of every op (ATen's HIP kernels:
    rocBLAS GEMMs, ATen reductions; oracle/make_sensitivity.py --device cuda) -- once plain (seed -1) and ten times with every ligand
    coordinate moved by -1 / 0 / +1 ulp after every step -- with the per-sample distance to the reference at every checkpoint.
    Asserted: atom and bond types of all 8 samples exact at all 20 checkpoints (as in every one of those eleven chains); every sample at every
    checkpoint within max(1e-4, 2 x the largest distance any sample of any of the eleven chains has there or at a neighbouring checkpoint); the median sample within
    max(1e-4, 3 x the pooled median); and at step 600 at least as many samples within 1e-4 as the worst of the eleven chains, less one.
    Printed: where the HIP chain sits in that distribution (samples within 1e-4 at steps 600 / 1000 next to the replays' range)."""
    if not os.path.exists(os.path.join(GU.GOLDEN, name + ".npz")):
        pytest.skip(f"{name}.npz not generated (python -m oracle.make_golden --only b8long / b8long_drift)")
    g, b, noise = _fixture_chain(name, synth.make_pocket_small(8), 8, drift_scale)
    assert b["init_ligand_pos"].shape[0] == 8 * 30 and b["protein_pos"].shape[0] == 8 * 300 and int(g["num_steps"]) == 1000
    drift = json.loads(str(g["drift"]))
    r = _sample_hip(model(0), b, 1000, drift, noise)
    every = int(g["every"])
    tp = torch.stack(r["pos_traj"]).numpy()[every - 1::every]
    tv = torch.stack(r["v_traj"]).numpy()[every - 1::every]
    tb = torch.stack(r["bond_traj"]).numpy()[every - 1::every]
    n = len(tp)
    d = np.abs(tp.astype(np.float64) - g["traj_pos"]).reshape(n, 8, -1).max(2)          # [checkpoint, sample]
    mv = int((tv != g["traj_v"]).sum())
    mb = int((tb != g["traj_bond"]).sum())
    sens = GU.load("sens_" + name)
    Es = np.asarray(sens["pos_err_sample"], dtype=np.float64)                            # [chain, checkpoint, sample]
    assert int(sens["every"]) == every and Es.shape[1:] == (n, 8) and Es.shape[0] >= 4
    assert int(np.asarray(sens["v_mismatch"]).sum()) == 0 and int(np.asarray(sens["bond_mismatch"]).sum()) == 0
    pooled_max = Es.max((0, 2))
    # (a sample that leaves the band does so within a few steps: the envelope at a checkpoint also covers its two neighbours, so that the
    #  bound does not hinge on WHICH side of a checkpoint such a step falls)
    pooled_max = np.maximum(pooled_max, np.maximum(np.r_[pooled_max[1:], pooled_max[-1]], np.r_[pooled_max[0], pooled_max[:-1]]))
    pooled_med = np.median(Es.transpose(1, 0, 2).reshape(n, -1), 1)
    bound = np.maximum(POS_TOL, 2.0 * pooled_max)
    w600, w1000 = (Es[:, 11, :] < POS_TOL).sum(1), (Es[:, -1, :] < POS_TOL).sum(1)
    h600, h1000 = int((d[11] < POS_TOL).sum()), int((d[-1] < POS_TOL).sum())
    print(f"{name} (NP=300, NL=30, B=8), checkpoints every {every} steps")
    print("  max |pos - reference| over the batch:", " ".join(f"{e:.2g}" for e in d.max(1)))
    print("  envelope (2 x pooled max of 11 chains): ", " ".join(f"{e:.2g}" for e in bound))
    print("  per sample at step 1000:", " ".join(f"{e:.2g}" for e in d[-1]))
    print(f"  samples within 1e-4: step 600 {h600}/8 (the 11 envelope chains: {w600.min()}-{w600.max()}), step 1000 {h1000}/8 "
          f"({w1000.min()}-{w1000.max()}; the plain chain of that implementation {int(w1000[0])}); type mismatches: atoms {mv}, bonds {mb}")
    GU.note_parity(f"{name}: samples within 1e-4 at step 600 {h600}/8 (envelope chains {w600.min()}-{w600.max()}), at step 1000 {h1000}/8 "
                   f"(envelope chains {w1000.min()}-{w1000.max()})")
    GU.record_parity(f"{'configs[2]' if drift else 'configs[1]'} 300+30 B=8 1000 steps ({name}, reference)", GU.chain_parity_summary(
        d, every, POS_TOL, (mv, mb), bound, "types exact; every sample <= max(1e-4, 2 x the largest per-sample distance of the oracle's code on ATen's "
        f"HIP kernels at that checkpoint: 1 plain + 10 +-1-ulp chains of THIS fixture, sens_{name}.npz); median sample <= max(1e-4, 3 x pooled "
        "median); samples within 1e-4 at step 600 >= the envelope chains' minimum - 1"))
    assert mv == 0 and mb == 0
    assert np.array_equal(r["v"].cpu().numpy(), g["out_v"]) and np.array_equal(r["bond"].cpu().numpy(), g["out_bond"])
    worst = int(np.argmax(d.max(1) / bound))
    assert (d.max(1) <= bound).all(), f"checkpoint {worst}: {d.max(1)[worst]:.3g} > {bound[worst]:.3g}"
    assert (np.median(d, 1) <= np.maximum(POS_TOL, 3.0 * pooled_med)).all()
    assert h600 >= int(w600.min()) - 1
    assert (d[:8] < POS_TOL).all()                        # steps 50 ... 400: the flat tolerance of BASELINE.json, all samples


def test_config1_full_chain_at_the_bench_shape_reference_golden():
    """BASELINE configs[1] for the whole chain at the bench shape (oracle/make_golden.py --only b8long: the reference alone, 2 h 15 min
    of CPU; checkpoints every 50 steps), see _bench_shape_chain."""
    _bench_shape_chain("traj1000_b8_plain", None)


def test_config2_full_chain_at_the_bench_shape_reference_golden():
    """BASELINE configs[2] (armsca + clash drift), the same (--only b8long_drift).  The unscaled drift gradients make the chain chaotic
    from step ~500 on: in the envelope chains single samples leave 1e-4 at step 550 (1.7e-3) and reach 0.05-0.14 A by the end."""
    _bench_shape_chain("traj1000_b8_drift", [1.0, 0.9, 0.8, 1.1, 1.0, 0.95, 1.05, 0.85])


@pytest.mark.parametrize("name,nc", [("traj4_aromatic13", 13), ("traj4_full23", 23)])
def test_atom_vocabularies_of_the_other_ligand_atom_modes_reference_golden(name, nc):
    """ligand_atom_mode add_aromatic / full (utils/transforms.py:15-64,138-151; the sampling script passes
    num_classes = ligand_feature_dim, :538-540): 13 / 23 atom classes -- wider ligand embedding, v head, class posterior and
    Gumbel draw -- 4 reverse steps with drift against a reference model of that width, injected noise; and a chain on the
    device Philox streams (classes 8.. come from further counter blocks) keeps every class reachable and finite."""
    from decompdiff_amd import DecompScorePosNet3D, shipped_config
    g = GU.load(name)
    assert int(g["num_classes"]) == nc
    b = GU.batch_from_npz(g)
    torch.manual_seed(int(g["seed"]))
    synth.build_sampling_batch(synth.make_pocket_small(9), 2, per_sample_std_scale=[1.0, 0.9], num_classes=nc)
    noise = synth.draw_step_noise(4, b["init_ligand_pos"].size(0), b["init_ligand_fc_bond_type"].size(0), num_classes=nc)
    assert GU.same_checksum(GU.checksum(noise), g["noise_checksum"])
    cfg = shipped_config()
    m = DecompScorePosNet3D(cfg, 29, nc + 2, nc)
    sd = m.state_dict()
    sd.update(synth.synthetic_state_dict(cfg, 0, ligand_atom_feature_dim=nc + 2, num_classes=nc))
    m.load_state_dict(sd, strict=True)
    m = m.to(dev())
    assert int(b["init_ligand_v"].max()) >= 8                     # the fixture really uses the upper classes
    r = _sample_hip(m, b, 4, json.loads(str(g["drift"])), noise)
    _check_chain(f"{nc} atom classes ({name})", r, g, 4)
    assert r["vt_traj"][0].shape[-1] == nc and r["v0_traj"][0].shape[-1] == nc
    p = _sample_hip(m, b, 12, None, None, seed=5)
    v = torch.stack(p["v_traj"])
    assert torch.isfinite(p["pos"]).all() and int(v.min()) >= 0 and int(v.max()) < nc and int(v.max()) >= 8
    with pytest.raises(NotImplementedError):
        DecompScorePosNet3D(cfg, 29, 12, 10)                          # not one of the reference's three vocabularies


def test_config3_unit_batch16_reference_golden():
    """BASELINE configs[3]: one unit of the 100-pocket job -- a pocket in its size range (347 protein + 37 ligand atoms:
    the 3-tile kernel variants), batch of 16 -- 3 reverse steps against the reference's own output."""
    g, b, noise = _fixture_chain("traj3_b16", synth.make_pocket(7, 347, (9, 9), 19, num_full_protein=0), 16)
    r = _sample_hip(model(0), b, 3, None, noise)
    _check_chain("configs[3] unit (NP=347, NL=37, B=16)", r, g, 3)
    assert hip_lib.load().dd_debug_node_split(16, 347, 37, 32) >= 0         # the per-shape launch measurement ran


def test_config4_large_pocket_drift_reference_golden():
    """BASELINE configs[4] size (600 + 60 atoms: the 4-tile kernel variants) with armsca + clash drift, B=2, 3 reverse
    steps against the reference's own output."""
    g, b, noise = _fixture_chain("traj3_large_drift", synth.make_pocket_large(6), 2, [1.0, 0.9])
    r = _sample_hip(model(0), b, 3, json.loads(str(g["drift"])), noise)
    _check_chain("configs[4] size (NP=600, NL=60, B=2, drift)", r, g, 3)


def test_ligand_beyond_64_atoms_reference_golden():
    """A ligand of 80 atoms (num_atoms_mode ref_large / stat can exceed 64; the reference has no size limit,
    uni_transformer_edge.py:103-123,349-359): 120 + 80 atoms, B = 2, armsca + clash drift, 3 reverse steps against the
    reference's own output (oracle/make_golden.py --only nl80) -- the 8-tile variants of the segment kernels (79 bond members,
    78 triplet members) and the two-wave arm-scaffold drift kernel."""
    g, b, noise = _fixture_chain("traj3_nl80", synth.make_pocket(13, 120, (27, 27), 26, num_full_protein=300), 2, [1.0, 0.9])
    assert b["init_ligand_pos"].shape[0] == 2 * 80
    r = _sample_hip(model(0), b, 3, json.loads(str(g["drift"])), noise)
    _check_chain("NL = 80 (NP=120, B=2, drift)", r, g, 3)


def test_large_pocket_batch8_matches_batch2_rows():
    """C-large at the bench's batch size (B=8, where the persistent bond-layer split applies): rows of samples 0-1 equal the
    B=2 run bit for bit (sharding property at this size), everything finite."""
    g, b, noise = _fixture_chain("traj3_large_drift", synth.make_pocket_large(6), 2, [1.0, 0.9])
    drift = json.loads(str(g["drift"]))
    r2 = _sample_hip(model(0), b, 3, drift, noise)
    b8 = synth.concat_sampling_batches([b] * 4)
    n8 = {k: torch.cat([v] * 4, 1) for k, v in noise.items()}
    r8 = _sample_hip(model(0), b8, 3, drift, n8, _drift_norm_batch=2)      # (the armsca mean is over the reference batch of 2)
    assert torch.isfinite(r8["pos"]).all()
    nl2 = r2["pos"].shape[0]
    for k in range(4):
        assert torch.equal(r8["pos"][k * nl2:(k + 1) * nl2], r2["pos"]), f"copy {k}"
        assert torch.equal(r8["v"][k * nl2:(k + 1) * nl2], r2["v"])


def test_drift_scale_option_reference_golden():
    """`scale: True` of the drift terms (decompdiff.py:656-657,667-668) against the reference's own output, mid-chain
    (t = 600..598) where pos_score_coef is not tiny."""
    g, b, noise = _fixture_chain("traj3_scale", synth.make_pocket(41, 80, (3, 3), 4, num_full_protein=200), 2, [1.0, 0.8])
    drift = json.loads(str(g["drift"]))
    assert all(d["scale"] for d in drift)
    r = _sample_hip(model(0), b, 3, drift, noise, int(g["t_start"]))
    _check_chain("drift scale=True (t=600..598)", r, g, 3)
    unscaled = _sample_hip(model(0), b, 3, [dict(d, scale=False) for d in drift], noise, int(g["t_start"]))
    assert maxabs(unscaled["pos"], g["out_pos"]) > 1e-3                     # the option matters in this case


@pytest.mark.parametrize("name", ["traj1000_plain", "traj1000_drift"])
def test_chain_segments_from_reference_checkpoints(name):
    """SURVEY.md H2, the re-synchronised bound: the chain is restarted from EVERY 50-step checkpoint of the reference's
    1000-step run (positions, atom and bond types at steps 50, 100, ... are in the fixture) and run for the next 50 steps on
    the reference's noise; each 50-step segment must end within 1e-4 of the reference's next checkpoint with identical
    types.  This bounds the per-segment error of the HIP path without the chaotic amplification a free-running 1000-step
    chain adds on top (oracle/sensitivity.py), plain and with armsca + clash drift."""
    from test_gpu_parity import _traj_inputs
    g, b, noise = _traj_inputs(name)
    drift = json.loads(str(g["drift"]))
    every = int(g["every"])
    n_ck = g["traj_pos"].shape[0]
    m = model(0)
    errs, mv, mb = [], [], []
    for c in range(n_ck):
        start = c * every
        bb = dict(b)
        if c > 0:                                          # state after `start` steps: the reference's checkpoint c-1
            bb["init_ligand_pos"] = torch.from_numpy(g["traj_pos"][c - 1].astype(np.float32))
            bb["init_ligand_v"] = torch.from_numpy(g["traj_v"][c - 1].astype(np.int64))
            bb["init_ligand_fc_bond_type"] = torch.from_numpy(g["traj_bond"][c - 1].astype(np.int64))
        seg_noise = {k: v[start:start + every] for k, v in noise.items()}
        r = _sample_hip(m, bb, every, drift, seg_noise, start_step=start, keep_traj=False)
        errs.append(maxabs(r["pos"], g["traj_pos"][c]))
        mv.append(int((r["v"].cpu().numpy() != g["traj_v"][c]).sum()))
        mb.append(int((r["bond"].cpu().numpy() != g["traj_bond"][c]).sum()))
    print(f"{name}: 50-step segments restarted from the reference's checkpoints")
    print("  max |pos - golden| :", " ".join(f"{e:.2g}" for e in errs))
    print("  type mismatches    : atoms", sum(mv), "bonds", sum(mb))
    GU.note_parity(f"{name}: {n_ck} segments of {every} steps restarted from the reference's checkpoints: worst segment end "
                   f"{max(errs):.2g} (tolerance {POS_TOL:g}), type mismatches {sum(mv)}+{sum(mb)}")
    assert sum(mv) == 0 and sum(mb) == 0
    assert max(errs) < POS_TOL, f"segment error {max(errs):.3g}"


def test_philox_noise_statistics():
    """The production noise (device Philox4x32-10 + Box-Muller), drawn through the same device functions the step kernels
    use (dd_debug_philox): uniforms in [0,1) with mean 1/2 and variance 1/12, normals with mean 0, variance 1, 4th moment
    3, a Kolmogorov-Smirnov distance small for 10^6 draws, and no stream shared between rows, steps, classes or the
    atom / bond / coordinate streams."""
    from scipy import stats
    lib = hip_lib.load()

    def draw(seed, step, rows, kind):
        n = rows * {1: 8, 2: 5, 7: 1}[kind]
        out = torch.empty(n, device=dev())
        hip_lib.check(lib.dd_debug_philox(seed, step, rows, kind, hip_lib.ptr(out), hip_lib.stream_ptr()), "dd_debug_philox")
        torch.cuda.synchronize()
        return out.cpu().double().numpy()

    u = draw(2021, 3, 125000, 1)                               # 10^6 uniforms, atom-type stream
    ub = draw(2021, 3, 200000, 2)                              # 10^6 uniforms, bond-type stream
    z = draw(2021, 3, 1000000, 7)
    for name, a in (("u_v", u), ("u_b", ub)):
        ks = stats.kstest(a, "uniform").statistic
        print(f"{name}: mean {a.mean():.5f} var {a.var():.5f} min {a.min():.3g} max {a.max():.8f} KS {ks:.2g}")
        assert a.min() >= 0.0 and a.max() < 1.0
        assert abs(a.mean() - 0.5) < 1e-3 and abs(a.var() - 1.0 / 12.0) < 1e-3 and ks < 2.5e-3
    ks = stats.kstest(z, "norm").statistic
    m4 = float((z ** 4).mean())
    print(f"eps: mean {z.mean():.5f} var {z.var():.5f} 4th moment {m4:.4f} max |z| {np.abs(z).max():.2f} KS {ks:.2g}")
    assert abs(z.mean()) < 3e-3 and abs(z.var() - 1.0) < 5e-3 and abs(m4 - 3.0) < 0.05 and ks < 2.5e-3
    assert np.isfinite(z).all() and np.abs(z).max() < 6.0
    # classes of a row are distinct draws (8 classes = two Philox blocks), neighbouring rows / steps / seeds / streams too
    uu = u.reshape(-1, 8)
    assert abs(np.corrcoef(uu[:, 0], uu[:, 4])[0, 1]) < 5e-3 and abs(np.corrcoef(uu[:-1, 7], uu[1:, 0])[0, 1]) < 5e-3
    assert len(np.unique(u)) > 0.97 * len(u)                 # 24-bit mantissas: ~3 % birthday collisions at most
    for other in (draw(2021, 4, 125000, 1), draw(2022, 3, 125000, 1)):
        assert abs(np.corrcoef(u, other)[0, 1]) < 5e-3 and not np.array_equal(u, other)
    assert abs(np.corrcoef(u[:1000000:8][:100000], ub[:500000:5][:100000])[0, 1]) < 1e-2
    assert np.array_equal(u, draw(2021, 3, 125000, 1))         # counter-based: reproducible


def test_default_seed_is_fresh_per_call_and_follows_torch_manual_seed():
    """The reference's script passes no seed: successive calls must see independent noise, governed by torch.manual_seed
    (as the reference's torch.randn_like draws are)."""
    pocket = synth.make_pocket_small(5)
    torch.manual_seed(1)
    b = synth.build_sampling_batch(pocket, 2)
    m = model(0)
    torch.manual_seed(77)
    r1 = _sample_hip(m, b, 4, None, None)
    r2 = _sample_hip(m, b, 4, None, None)
    torch.manual_seed(77)
    r3 = _sample_hip(m, b, 4, None, None)
    assert not torch.equal(r1["pos"], r2["pos"])
    assert torch.equal(r1["pos"], r3["pos"]) and torch.equal(r1["bond"], r3["bond"])


def test_bench_two_ranks_equal_one_rank():
    """`python bench.py --gpus 2` (no torchrun environment) spawns two ranks itself; over gloo both may share this box's
    one GPU.  The per-unit checksums of the 2-rank job must equal those of the 1-rank job (units are defined without
    reference to the world size); the line reports 2 ranks on ONE distinct device (n_gpus counts devices, not ranks).
    Without --oversubscribe the same command is refused: more ranks than visible devices."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    common = ["--config", "4", "--num-samples", "16", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-rooflines"]

    def run(extra):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra + common, capture_output=True, text=True,
                           env=env, timeout=900)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)

    one = run(["--gpus", "1"])
    n_dev = torch.cuda.device_count()
    two = run(["--gpus", "2", "--backend", "gloo", "--oversubscribe"])
    assert one["n_gpus"] == 1 and one["ranks"] == 1 and two["ranks"] == 2 and len(two["per_rank"]) == 2
    assert two["n_gpus"] == two["distinct_devices"] == min(2, n_dev) and len(two["devices"]) == 2
    if n_dev == 1:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, capture_output=True, text=True,
                           env=env, timeout=900)
        assert p.returncode != 0 and "2 ranks but only 1 visible" in (p.stdout + p.stderr)
        assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert [r["units"] for r in two["per_rank"]] == [[0], [1]]
    strip = lambda rs: [(r["unit"], r["checksum"]) for r in rs]
    assert strip(one["per_unit"]) == strip(two["per_unit"])
    assert one["scaling"] == two["scaling"] == "strong"


def _bench_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR",
                                                             "MASTER_PORT", "DD_DIST_FORCE_GROUP")}


def _bench_line(argv, env, timeout=1500):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, env=env, timeout=timeout)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1]), p


def _keep_evidence(name, obj):
    """gpurun merges gpurun_out/ back: the lines these tests produce are copied to profiles/ by the evidence script."""
    try:
        d = os.path.join(ROOT, "gpurun_out", "tests")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name), "w") as fh:
            json.dump(obj, fh, indent=1, default=str)
    except OSError:
        pass


def test_rccl_single_rank_control_plane():
    """The RCCL branch of the control plane on real hardware: one rank with DD_DIST_FORCE_GROUP=1
    builds the gloo default group, the RCCL group on top (new_group("nccl")), passes the probe barrier, and the timed job
    then runs its two barriers and the all-reduce(MAX) of the wall time over RCCL.  On the one-GPU boxes the suite runs on
    this is the only positive RCCL test possible (two ranks on one device is invalid RCCL usage)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(_bench_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               DD_DIST_FORCE_GROUP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    argv = ["--gpus", "1", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-rooflines", "--strict-rccl"]
    probe = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, env=env, timeout=1500)
    if probe.returncode != 0 and "RCCL start-up failed" in (probe.stdout + probe.stderr):
        # the box cannot bring RCCL up at all (seen on none of the builder's boxes): an environment limit, not a result of this
        # code -- recorded, and the suite goes on (the driver runs it with -x)
        _keep_evidence("rccl_single_rank.json", {"rccl_unavailable": (probe.stdout + probe.stderr)[-1500:]})
        pytest.xfail("RCCL cannot start on this box: " + (probe.stdout + probe.stderr)[-300:])
    assert probe.returncode == 0, probe.stdout[-2000:] + probe.stderr[-4000:]
    line, p = json.loads([l for l in probe.stdout.splitlines() if l.startswith("{")][-1]), probe
    _keep_evidence("rccl_single_rank.json", {"line": line, "stderr_tail": p.stderr[-1500:]})
    assert line["config"]["control_plane"] == "nccl", line["config"]           # RCCL, not the gloo fall-back
    assert line["config"]["control_plane_note"] is None
    assert line["ranks"] == 1 and line["n_gpus"] == 1 and line["value"] > 0
    assert "cpu_affinity" in line["per_rank"][0]
    plain, _ = _bench_line(["--gpus", "1", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-rooflines"], _bench_env())
    assert plain["config"]["control_plane"] == "single process"
    strip = lambda rs: [(r["unit"], r["checksum"]) for r in rs]
    assert strip(plain["per_unit"]) == strip(line["per_unit"])                 # the same chain either way


def test_bench_eight_ranks_oversubscribed():
    """The shape of the driver's `--gpus 8` run on this box's one GPU: 8 ranks spawned by bench.py
    itself over gloo, 16 pockets of configs[3] assigned longest-first, gather of the per-unit records, imbalance, CPU
    affinity per rank.  The per-unit checksums equal those of the 1-rank job."""
    common = ["--config", "3", "--pockets", "16", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-rooflines"]
    eight, _ = _bench_line(["--gpus", "8", "--backend", "gloo", "--oversubscribe"] + common, _bench_env())
    one, _ = _bench_line(["--gpus", "1"] + common, _bench_env())
    _keep_evidence("bench_8ranks_oversubscribed.json", eight)
    from decompdiff_amd import dist as ddist
    units, _ = ddist.plan_job(3, 8, n_pockets=16)
    assert eight["ranks"] == 8 and len(eight["per_rank"]) == 8 and eight["n_gpus"] == eight["distinct_devices"] == min(8, torch.cuda.device_count())
    assert [r["units"] for r in eight["per_rank"]] == ddist.assign_lpt(units, 8)
    assert eight["config"]["control_plane"] == "gloo" and eight["config"]["imbalance_max_over_mean_busy"] >= 1.0
    assert all(isinstance(r["cpu_affinity"], dict) for r in eight["per_rank"])
    bound = [r["cpu_affinity"] for r in eight["per_rank"] if r["cpu_affinity"].get("bound")]
    if len(bound) == 8 and len({a["numa_cpus"] for a in bound}) == 1 and bound[0]["numa_cpus"] >= 8:
        assert len({a["cpus"] for a in bound}) == 8                            # ranks sharing a NUMA node: disjoint slices
    strip = lambda rs: [(r["unit"], r["checksum"]) for r in rs]
    assert strip(one["per_unit"]) == strip(eight["per_unit"]) and len(eight["per_unit"]) == 16
    assert one["scaling"] == eight["scaling"] == "strong"


def test_ligand_atom_mask_matches_reference_behaviour():
    """An all-True boolean ligand_atom_mask equals None (as in the reference); a mask with masked-out entries raises the
    RuntimeError the reference's own loop raises for it (decompdiff.py:321,611)."""
    pocket = synth.make_pocket_tiny(1)
    torch.manual_seed(0)
    b = synth.build_sampling_batch(pocket, 2)
    n = b["init_ligand_pos"].shape[0]
    m = model(0)
    r0 = _sample_hip(m, b, 2, None, None, seed=5)
    r1 = _sample_hip(m, dict(b, ligand_atom_mask=torch.ones(n, dtype=torch.bool)), 2, None, None, seed=5)
    assert torch.equal(r0["pos"], r1["pos"]) and torch.equal(r0["bond"], r1["bond"])
    for bad in (torch.tensor([True] * (n - 2) + [False] * 2), torch.tensor([1] * (n - 2) + [0] * 2)):
        with pytest.raises(RuntimeError):
            _sample_hip(m, dict(b, ligand_atom_mask=bad), 2, None, None, seed=5)


def test_chain_cache_reuses_buffers_and_graph_without_changing_results(monkeypatch):
    """Chains of one shape share device buffers and the captured step graph (start time and Philox key live in device
    memory): results must equal those of uncached runs, for another seed, another start step, other inputs, a drift
    option added later (re-capture) and a launch-structure option changed under the cache (options epoch)."""
    lib = hip_lib.load()
    m = model(0)
    p1, p2 = synth.make_pocket_small(5), synth.make_pocket_small(6)
    torch.manual_seed(1)
    b1 = synth.build_sampling_batch(p1, 2)
    b2 = synth.build_sampling_batch(p2, 2)
    runs = [(b1, None, 11, 0), (b1, None, 12, 0), (b2, None, 11, 0), (b2, GU.DRIFT, 13, 0), (b1, None, 11, 300), (b1, None, 11, 0)]

    def go(cache_on):
        monkeypatch.setenv("DD_CHAIN_CACHE", "1" if cache_on else "0")
        m._evict_chain_cache(0)
        outs = []
        for i, (b, drift, seed, start) in enumerate(runs):
            if i == 5:
                assert lib.dd_debug_set_fusion(3) == 0               # another launch structure (no second stream): the cached graph is stale
            outs.append(_sample_hip(m, b, 6 if i % 2 else 9, drift, None, seed=seed, start_step=start))
        assert lib.dd_debug_set_fusion(1) == 0
        return outs

    cached, fresh = go(True), go(False)
    for i, (c, f) in enumerate(zip(cached, fresh)):
        for k in ("pos", "v", "bond"):
            assert torch.equal(c[k], f[k]), (i, k)
        for k in ("pos_traj", "vt_traj", "bt_traj", "bond_traj"):
            assert torch.equal(torch.stack(c[k]), torch.stack(f[k])), (i, k)
    assert not torch.equal(cached[0]["pos"], cached[1]["pos"])       # another seed, another chain
    monkeypatch.setenv("DD_CHAIN_CACHE", "1")
    m._evict_chain_cache(0)
    _sample_hip(m, b1, 5, None, None, seed=1)
    _sample_hip(m, b1, 20, None, None, seed=2)                       # 5 and 20 steps share one entry (capacity 32)
    assert len(m._chain_cache) == 1
    _sample_hip(m, b1, 40, None, None, seed=2)                       # 40 steps: another capacity, a second entry ...
    assert len(m._chain_cache) == 2
    m._evict_chain_cache(0)
    m.traj_capacity_hint = 40                                        # ... unless the caller announced the length (bench.py does)
    try:
        short = _sample_hip(m, b1, 5, None, None, seed=1)
        _sample_hip(m, b1, 40, None, None, seed=2)
        assert len(m._chain_cache) == 1 and len(short["pos_traj"]) == 5
    finally:
        m.traj_capacity_hint = 0
        m._evict_chain_cache(0)


def _hetero_batch(sizes, n_protein, seed=0):
    torch.manual_seed(seed)
    parts = []
    for i, (nl, np_) in enumerate(zip(sizes, n_protein)):
        arm = max(2, nl // 4)
        p = synth.make_pocket(seed=300 + i, num_protein=np_, arm_atoms=(arm, arm), scaffold_atoms=nl - 2 * arm, num_full_protein=np_ + 150)
        parts.append(synth.build_sampling_batch(p, 1))
    return synth.concat_sampling_batches(parts)


def test_padded_heterogeneous_batch_equals_size_groups(monkeypatch):
    """North star "padded/masked fixed-size pocket+ligand graphs": a batch whose samples differ in pocket AND ligand size
    (ligands 9..37 atoms: one, two and three member tiles; pockets 120..260 atoms) run as ONE padded launch sequence must
    reproduce the per-size-group path (every group a dense batch, validated against the reference's ragged golden) on the
    same injected noise, with armsca + clash drift: coordinates to fp32 rounding, types exactly."""
    sizes = [9, 37, 20, 33, 17, 25]
    n_prot = [150, 260, 120, 200, 180, 131]
    b = _hetero_batch(sizes, n_prot, seed=4)
    steps = 4
    noise = synth.draw_step_noise(steps, b["init_ligand_pos"].size(0), b["init_ligand_fc_bond_type"].size(0))
    m = model(0)
    outs = {}
    for mode in ("padded", "groups"):
        monkeypatch.setenv("DD_RAGGED_MODE", mode)
        outs[mode] = _sample_hip(m, b, steps, GU.DRIFT, noise)
    p, g = outs["padded"], outs["groups"]
    err = maxabs(p["pos"], g["pos"])
    e_bt = maxabs(torch.stack(p["bt_traj"]), torch.stack(g["bt_traj"]))
    print(f"padded vs groups ({len(sizes)} distinct sizes): pos diff {err:.3g}, bond log-prob diff {e_bt:.3g}")
    assert p["pos"].shape == g["pos"].shape == (sum(sizes), 3)
    assert err < 5e-6 and e_bt < 5e-5
    assert torch.equal(p["v"], g["v"]) and torch.equal(p["bond"], g["bond"])
    assert torch.equal(torch.stack(p["v_traj"]), torch.stack(g["v_traj"]))
    # production noise: deterministic, finite, cached resources re-used for the next heterogeneous batch of these maxima
    monkeypatch.setenv("DD_RAGGED_MODE", "padded")
    r1 = _sample_hip(m, b, 6, GU.DRIFT, None, seed=5)
    b2 = _hetero_batch([37, 9, 25, 33, 20, 17], [260, 150, 131, 200, 120, 180], seed=4)     # same maxima, other per-sample counts
    _sample_hip(m, b2, 6, GU.DRIFT, None, seed=5)
    r3 = _sample_hip(m, b, 6, GU.DRIFT, None, seed=5)
    assert torch.isfinite(r1["pos"]).all() and torch.equal(r1["pos"], r3["pos"]) and torch.equal(r1["bond"], r3["bond"])


def test_same_shape_sub_batches_equal_the_whole_batch():
    """A dense batch cut into sub-batches of ONE shape (each a chain of its own with its own cached buffers -- the cache
    key carries a slot, otherwise two chains of a call would share device memory), run one after the other and as
    concurrent graphs on separate streams, reproduces the one-chain result bit for bit on the same injected noise
    (every sample's chain is independent of its batch; tools/split_bench.py times the same path)."""
    m = model(0)
    torch.manual_seed(3)
    b = synth.build_sampling_batch(synth.make_pocket_small(2), 4)
    steps = 3
    noise = synth.draw_step_noise(steps, b["init_ligand_pos"].size(0), b["init_ligand_fc_bond_type"].size(0))
    whole = _sample_hip(m, b, steps, None, noise)
    kw = {k: (v.to(dev()) if torch.is_tensor(v) else v) for k, v in b.items() if k != "ligand_atom_mask"}
    for concurrent in (False, True):
        for rep in range(2):                                   # second pass: every sub-batch finds its own cache entry
            part = m._sample_ragged(kw, None, steps, "protein", None, noise, 1, True, True, concurrent=concurrent, split=2)
            for k in ("pos", "v", "bond"):
                assert torch.equal(part[k].cpu(), whole[k].cpu()), (concurrent, rep, k)
            assert torch.equal(torch.stack(part["bt_traj"]), torch.stack(whole["bt_traj"]))
    # production noise: the two sub-batches of a call must not share buffers (identical halves would be the symptom)
    r = m._sample_ragged(kw, None, 5, "protein", None, None, 11, True, True, concurrent=True, split=2)
    r2 = m._sample_ragged(kw, None, 5, "protein", None, None, 11, True, True, concurrent=False, split=2)   # cached entries re-used
    nl = r["pos"].shape[0] // 4
    assert torch.isfinite(r["pos"]).all() and not torch.equal(r["pos"][:2 * nl], r["pos"][2 * nl:])
    assert torch.equal(r["pos"], r2["pos"]) and torch.equal(r["bond"], r2["bond"])


def test_bench_json_line_contract():
    """The driver's contract for bench.py: ONE JSON line from rank 0 with the metric of BASELINE.json, the whole-job value,
    and the roofline / cpu_baseline objects."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--cpu-steps", "1",
                        "--cpu-warmup", "1", "--cpu-threads", "16"], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] in base["metric"] and d["unit"] == "denoising steps/s"
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    assert "configs[1]" in d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.05 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 0
    g = d["roofline_gemm"]
    assert 0.0 < g["frac"] < 1.0
    o = d["roofline_op_level"]
    assert o["bound"] == "hbm" and o["unit"] == "GB/s" and 0.3 < o["frac"] < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "steps" in c["sample"]


def test_static_input_memo_is_invalidated_by_in_place_changes():
    """Validation / centring of the pocket-side inputs is remembered for calls that pass the same tensor objects
    (model._static_memo_get); an in-place change (version counter) or a new tensor must not see the old centroid."""
    m = model(0)
    torch.manual_seed(5)
    b = to_dev_local(synth.build_sampling_batch(synth.make_pocket(17, 60, (3, 3), 4, num_full_protein=0), 2))
    kw = dict(num_steps=4, center_pos_mode="protein", seed=77)
    r0 = m.sample_diffusion(**b, **kw)
    assert m.__dict__.get("_static_memo") is not None
    r1 = m.sample_diffusion(**b, **kw)                        # memo hit
    assert torch.equal(r0["pos"], r1["pos"]) and torch.equal(r0["v"], r1["v"])
    shift = torch.tensor([1.5, -2.0, 0.25], device=b["protein_pos"].device)
    b["protein_pos"].add_(shift)                              # in place: same object, new version
    b["init_ligand_pos"] = b["init_ligand_pos"] + shift
    r2 = m.sample_diffusion(**b, **kw)
    assert maxabs(r2["pos"], r0["pos"] + shift) < 1e-5        # translated frame: stale centroid would be off by |shift|
    b2 = dict(b)
    b2["protein_pos"] = b["protein_pos"] - shift              # a new tensor object
    b2["init_ligand_pos"] = b["init_ligand_pos"] - shift
    r3 = m.sample_diffusion(**b2, **kw)
    assert maxabs(r3["pos"], r0["pos"]) < 1e-5
    # a ragged batch never hits the memo of a dense one
    with pytest.raises((NotImplementedError, ValueError, RuntimeError, AssertionError)):
        bad = dict(b); bad["batch_ligand"] = b["batch_ligand"].clone(); bad["batch_ligand"][0] = 1
        m.sample_diffusion(**bad, **kw)


@pytest.mark.gpu
def test_pocket_side_uploads_are_skipped_only_for_unchanged_tensors():
    """A cached chain entry keeps the pocket-side device buffers (protein coordinates / embedded features, arm indicators,
    layer-0 protein rows) when the caller passes the very same tensors again (model._make_sampler: `pocket_in_place`); an
    in-place change of any of them, another centring mode or a new tensor object must be uploaded again."""
    m = model(0)
    torch.manual_seed(6)
    b = to_dev_local(synth.build_sampling_batch(synth.make_pocket(19, 60, (3, 3), 4, num_full_protein=0), 2))
    kw = dict(num_steps=4, seed=78)
    fresh = lambda bb, mode="protein": model(0).sample_diffusion(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in bb.items()},
                                                                 center_pos_mode=mode, **kw)
    r0 = m.sample_diffusion(**b, center_pos_mode="protein", **kw)
    ent = next(reversed(type(m)._chain_cache.values()))
    assert ent.get("uploaded") is not None
    r1 = m.sample_diffusion(**b, center_pos_mode="protein", **kw)            # uploads skipped
    assert torch.equal(r0["pos"], r1["pos"]) and torch.equal(r0["v"], r1["v"]) and torch.equal(r0["bond"], r1["bond"])
    # protein features changed in place (same object): the embedded rows must be rebuilt
    b["protein_v"][:, :6] = b["protein_v"][:, :6].roll(1, dims=1)
    r2 = m.sample_diffusion(**b, center_pos_mode="protein", **kw)
    w2 = fresh(b)
    assert torch.equal(r2["pos"], w2["pos"]) and torch.equal(r2["v"], w2["v"]) and not torch.equal(r2["pos"], r0["pos"])
    # another centring mode with the same tensors
    r3 = m.sample_diffusion(**b, center_pos_mode="none", **kw)
    w3 = fresh(b, "none")
    assert torch.equal(r3["pos"], w3["pos"]) and torch.equal(r3["v"], w3["v"])
    r4 = m.sample_diffusion(**b, center_pos_mode="protein", **kw)
    assert torch.equal(r4["pos"], r2["pos"])
    # arm / scaffold indicators swapped in place
    b["ligand_v_aux"].copy_(b["ligand_v_aux"].flip(1))
    r5 = m.sample_diffusion(**b, center_pos_mode="protein", **kw)
    w5 = fresh(b)
    assert torch.equal(r5["pos"], w5["pos"]) and torch.equal(r5["v"], w5["v"]) and not torch.equal(r5["pos"], r2["pos"])


_SPLIT_PROBE = r"""
import ctypes, json, sys, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, hip_lib, shipped_config, synth
from decompdiff_amd import dist as ddist
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, seed=0)); m.load_state_dict(sd); m = m.to(dev)
torch.manual_seed(3)
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(synth.make_pocket_small(0), 8).items()}
out = m.sample_diffusion(num_steps=3, center_pos_mode="protein", seed=5, **b)
lib = hip_lib.load(); buf = ctypes.create_string_buffer(1024)
hip_lib.check(lib.dd_debug_node_split_cache_path(buf, 1024))
print(json.dumps({"split": int(lib.dd_debug_node_split(8, 300, 30, 32)), "path": buf.value.decode(), "cs": ddist.checksum(out)}))
"""


def test_measured_node_split_is_kept_across_processes(tmp_path):
    """The CU split of the fused node launch is measured once per shape, device model and build, and kept in a file
    (dd_api.hip::node_split_cache_*): a second process -- another rank of the node, the next run of the script -- reads it
    instead of timing ~30 forward passes, and whatever split it reads the results are bit-identical."""
    env = dict(_bench_env(), DD_NODE_SPLIT_CACHE_DIR=str(tmp_path))
    run = lambda: json.loads(subprocess.run([sys.executable, "-c", _SPLIT_PROBE], cwd=ROOT, env=env, capture_output=True, text=True,
                                            timeout=600, check=True).stdout.strip().splitlines()[-1])
    first = run()
    assert first["path"].startswith(str(tmp_path)) and os.path.exists(first["path"])
    lines = open(first["path"]).read().split("\n")
    assert f"8 300 30 32 {first['split']}" in lines and first["split"] % 8 == 0
    other = 96 if first["split"] != 96 else 104
    with open(first["path"], "a") as fh:                       # (the last line of a shape wins)
        fh.write(f"8 300 30 32 {other}\n")
    second = run()
    assert second["split"] == other                            # read, not measured again
    assert second["cs"] == first["cs"]                         # the split only decides which CU computes a segment
    # (DD_NODE_SPLIT_CACHE=0 -> dd_debug_node_split_cache_path() == "": covered by the host-side getenv, not worth a third process)


_TWO_LENGTHS_PROBE = r"""
import json, sys, time, torch
sys.path.insert(0, ".")
from decompdiff_amd import DecompScorePosNet3D, shipped_config, synth
dev = torch.device("cuda:0"); cfg = shipped_config()
m = DecompScorePosNet3D(cfg, 29, 10, 8); sd = m.state_dict(); sd.update(synth.synthetic_state_dict(cfg, seed=0)); m.load_state_dict(sd); m = m.to(dev)
torch.manual_seed(3)
b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synth.build_sampling_batch(synth.make_pocket_small(0), 1).items()}
ms = {}
for n in (20, 100, 100, 100, 200, 200, 200):
    torch.cuda.synchronize(); t = time.perf_counter()
    m.sample_diffusion(num_steps=n, center_pos_mode="protein", seed=5, **b)
    torch.cuda.synchronize()
    ms.setdefault(n, []).append(1e3 * (time.perf_counter() - t) / n)
print(json.dumps({str(k): v for k, v in ms.items()}))
"""


def test_a_second_chain_length_runs_as_fast_as_the_first():
    """Round 6 (EXPERIMENTS.md R6-11): with the library's side stream at a priority of its own, the SECOND step graph of a process (here:
    the same pocket sampled with another num_steps, i.e. another cached chain entry) ran 42 % slower at B = 1 (14 % at B = 8) -- a fifth
    hardware queue in use.  Own process (stream set-up is per process); the later calls of each length are compared."""
    out = subprocess.run([sys.executable, "-c", _TWO_LENGTHS_PROBE], cwd=ROOT, env=_bench_env(), capture_output=True, text=True, timeout=600,
                         check=True).stdout.strip().splitlines()[-1]
    ms = json.loads(out)
    first, second = min(ms["100"][1:]), min(ms["200"][1:])
    print(f"\n  B = 1: {first:.4f} ms/step in 100-step calls, {second:.4f} in 200-step calls of the same process")
    assert second < 1.15 * first, ms                           # (the slow regime: 1.42 x)


def to_dev_local(batch):
    return {k: (v.to(dev()) if torch.is_tensor(v) else v) for k, v in batch.items()}


def test_layer0_tables_are_bit_identical_to_the_gemms(debug_options):
    """dd_sampler.l0_tables: the first layer's projection / query rows gathered from tables (16 atom combinations, 5 x 16
    bond combinations, static protein rows) instead of two GEMM launches -- same kernels built the tables, so forward and
    chain are bit-identical with the option off; rows that are not exactly one-hot in the arm flag switch them off."""
    if not debug_options:
        return
    lib = hip_lib.load()
    m = model(0)
    torch.manual_seed(9)
    pocket = synth.make_pocket_small(4)
    b = to_dev_local(synth.build_sampling_batch(pocket, 3))
    names = ["protein_pos", "protein_v", "batch_protein", "protein_group_idx", "init_ligand_pos", "init_ligand_v", "batch_ligand",
             "ligand_group_idx", "prior_centers", "prior_stds", "batch_prior", "prior_group_idx", "ligand_fc_bond_index",
             "init_ligand_fc_bond_type"]
    fwd = lambda bb: m(init_ligand_v_aux=bb["ligand_v_aux"], **{n: bb[n] for n in names})
    try:
        on = fwd(b)
        assert m._last[0].l0_tables                      # tables in use
        chain_on = m.sample_diffusion(num_steps=5, center_pos_mode="protein", energy_drift_opt=GU.DRIFT, seed=5, **b)
        assert lib.dd_debug_set_option(22, 0) == 0
        off = fwd(b)
        chain_off = m.sample_diffusion(num_steps=5, center_pos_mode="protein", energy_drift_opt=GU.DRIFT, seed=5, **b)
    finally:
        assert lib.dd_debug_set_option(22, 1) == 0
    for k in ("pred_ligand_pos", "pred_ligand_v", "pred_bond"):
        assert torch.equal(on[k], off[k]), k
    for k in ("pos", "v", "bond"):
        assert torch.equal(chain_on[k], chain_off[k]), k
    assert all(torch.equal(x, y) for x, y in zip(chain_on["pos_traj"], chain_off["pos_traj"]))
    soft = dict(b)
    soft["ligand_v_aux"] = b["ligand_v_aux"] * 0.75 + 0.125          # not an indicator any more: GEMM path
    r = fwd(soft)
    assert not m._last[0].l0_tables and torch.isfinite(r["pred_ligand_pos"]).all()
    assert not torch.equal(r["pred_ligand_v"], on["pred_ligand_v"])


@pytest.mark.parametrize("shape", ["small", "mid", "ragged", "large_drift"])
def test_projections_inside_the_coordinate_launch_are_bit_identical(debug_options, shape):
    """Round 5: the {P2, PL2} projections of the new h (and, in the last layer, the heads' first Linear) run in the leading /
    trailing workgroups of the coordinate launch (k_attn2_pos_g, in-launch hand-off through a tile counter) instead of a
    launch of their own.  Same tile code, same operands: forward and chains must be bit-identical with option 30 off (the
    two launches: what ships -- the variant is 2.6 % slower, EXPERIMENTS.md R5-2), graph replay and eager, dense / long-ligand /
    padded batches, with drift."""
    if not debug_options:
        return
    lib = hip_lib.load()
    m = model(0)
    torch.manual_seed(11)
    drift = None
    if shape == "small":
        b = to_dev_local(synth.build_sampling_batch(synth.make_pocket_small(3), 8))
    elif shape == "mid":                                   # 37 ligand atoms: the 3-tile kernels
        b = to_dev_local(synth.build_sampling_batch(synth.make_pocket(5, 347, (10, 9), 18, num_full_protein=0), 4))
    elif shape == "ragged":
        b = to_dev_local(_hetero_batch([12, 20, 9, 16], [70, 90, 60, 80]))
    else:
        b = to_dev_local(synth.build_sampling_batch(synth.make_pocket_large(2), 2))
        drift = GU.DRIFT
    run = lambda graph: m.sample_diffusion(num_steps=6, center_pos_mode="protein", energy_drift_opt=drift, seed=21, use_graph=graph, **b)
    try:
        assert lib.dd_debug_set_option(32, 0) == 0          # (the variant only exists beside a separate lin_node launch)
        assert lib.dd_debug_set_option(30, 1) == 0 and lib.dd_debug_schedule() & 1
        on_g, on_e = run(True), run(False)
        assert lib.dd_debug_set_option(30, 0) == 0
        off_g, off_e = run(True), run(False)
    finally:
        assert lib.dd_debug_set_option(30, 0) == 0
        assert lib.dd_debug_set_option(32, -1) == 0
    for a, c in ((on_g, off_g), (on_e, off_e), (on_g, on_e)):
        for k in ("pos", "v", "bond"):
            assert torch.equal(a[k], c[k]), (shape, k)
        for k in ("pos_traj", "v0_traj", "bt_traj"):
            assert all(torch.equal(x, y) for x, y in zip(a[k], c[k])), (shape, k)


def test_two_launch_head_is_bit_identical_to_the_four_launches(debug_options):
    """Head of a forward: k_head_graph (kNN by radix select + edge weights in one wave per centre, x_t read from the
    sampler's position buffers) beside k_head_rows (embeddings / context / counters + layer-0 rows) against the four
    separate launches (dd_debug_set_option(24, 0)) -- forward and chain bit-identical, dense and padded batches."""
    if not debug_options:
        return
    lib = hip_lib.load()
    m = model(0)
    torch.manual_seed(11)
    b = to_dev_local(synth.build_sampling_batch(synth.make_pocket_small(7), 3))
    hb = to_dev_local(_hetero_batch([9, 20, 33], [150, 120, 200], seed=6))
    run = lambda bb: m.sample_diffusion(num_steps=4, center_pos_mode="protein", energy_drift_opt=GU.DRIFT, seed=21, **bb)
    try:
        two, two_h = run(b), run(hb)
        assert lib.dd_debug_set_option(24, 0) == 0
        four, four_h = run(b), run(hb)
    finally:
        assert lib.dd_debug_set_option(24, 1) == 0
    for x, y in ((two, four), (two_h, four_h)):
        for k in ("pos", "v", "bond"):
            assert torch.equal(x[k], y[k]), k
        assert all(torch.equal(p, q) for p, q in zip(x["v0_traj"], y["v0_traj"]))


def test_tile_queue_schedule_is_bit_identical_to_the_graph_edge_schedule(debug_options):
    """Schedule 5 of the measurement build (EXPERIMENTS.md, round 3): the GEMMs between two node attentions as two ordered
    tile queues per layer (tickets, write-through tiles, device counters polled by the next assemble / node attention instead
    of graph edges).  Slower than the shipped schedule 4, kept as a measured alternative -- and it must not change a bit:
    dense, drift, long ligands, padded batch; graph replay and eager; no hand-off may time out (dd_queue_error)."""
    if not debug_options:
        return
    lib = hip_lib.load()
    m = model(0)
    torch.manual_seed(3)
    cases = [(synth.build_sampling_batch(synth.make_pocket_small(0), 4), None),
             (synth.build_sampling_batch(synth.make_pocket_small(1), 2), GU.DRIFT),
             (synth.build_sampling_batch(synth.make_pocket(7, 347, (9, 9), 19, num_full_protein=0), 2), None),
             (_hetero_batch([9, 20, 33], [150, 120, 200], seed=6), GU.DRIFT)]
    try:
        # (the tile queues keep lin_node as a GEMM job: the small batches of this test would otherwise take the round-6 form with
        #  lin_node inside the node launch under schedule 4 -- the same math in another association)
        assert lib.dd_debug_set_option(32, 0) == 0
        for b, drift in cases:
            outs = {}
            for sched, graph in ((4, True), (5, True), (5, False)):
                assert lib.dd_debug_set_option(8, sched) == 0
                outs[(sched, graph)] = _sample_hip(m, b, 5, drift, None, seed=9, use_graph=graph)
            for key in ((5, True), (5, False)):
                for k in ("pos", "v", "bond"):
                    assert torch.equal(outs[(4, True)][k], outs[key][k]), (key, k)
                assert all(torch.equal(x, y) for x, y in zip(outs[(4, True)]["bt_traj"], outs[key]["bt_traj"]))
    finally:
        assert lib.dd_debug_set_option(32, -1) == 0
        assert lib.dd_debug_set_option(8, 4) == 0
