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
    """Sums of the three noise tensors in double, summed by numpy (one thread, pairwise): the value does not depend on
    torch's thread count.  The fixtures hold torch's sums (make_golden.py, 8 threads): compare with `same_checksum`."""
    return np.array([float(np.sum(noise[k].numpy().astype(np.float64))) for k in ("u_v", "u_b", "eps")])


def same_checksum(a, b):
    """Pins a re-drawn noise stream to the fixture's.  Relative 1e-12: a parallel double sum of ~10^6 normals moves in its
    last bits with the number of threads (the uniform streams are multiples of 2^-24: their sums are exact), a different
    draw moves it in its first digits."""
    return bool(np.allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=1e-12, atol=0.0))


# ----------------------------------------------------------------------------------------------------------------
# harness fixtures (tests/golden/harness_*.npz): shared by oracle/make_harness_golden.py (which runs the reference's
# own harness around RecordingModel) and tests/test_harness_golden.py (which runs ours around the same model)
# ----------------------------------------------------------------------------------------------------------------
class RecordingModel:
    """Stands in for DecompScorePosNet3D inside the sampling harness: records the keyword arguments and returns a
    deterministic function of the initial state with the shapes of the real result (2 'steps')."""
    num_bond_classes = 5
    num_classes = 8
    bond_diffusion = True
    num_timesteps = 1000

    def __init__(self):
        self.calls = []

    def sample_diffusion(self, **kw):
        self.calls.append({k: (v.detach().clone().cpu() if torch.is_tensor(v) else v) for k, v in kw.items()})
        pos0, v0, b0 = kw["init_ligand_pos"].cpu(), kw["init_ligand_v"].cpu(), kw["init_ligand_fc_bond_type"].cpu()
        T = 2
        one_hot = lambda t, k: torch.nn.functional.one_hot(t, k).float()
        r = {"pos": pos0 * 0.5 + 1.0, "v": (v0 + 1) % 8, "bond": (b0 + 2) % 5,
             "pos_traj": [pos0 * (0.25 * (i + 1)) for i in range(T)], "v_traj": [(v0 + i) % 8 for i in range(T)],
             "v0_traj": [one_hot((v0 + i) % 8, 8) * 0.5 for i in range(T)],
             "vt_traj": [one_hot((v0 + 2 * i) % 8, 8) * 0.25 for i in range(T)],
             "bond_traj": [(b0 + i) % 5 for i in range(T)], "bt_traj": [one_hot((b0 + i) % 5, 5) * 0.75 for i in range(T)]}
        return r


class LinearCountModel:
    """A scikit-learn style regressor for the ``stat`` atom-count mode: predict(x) = w * x.sum(1) + b."""

    def __init__(self, w, b):
        self.w, self.b = float(w), float(b)

    def predict(self, x):
        x = np.asarray(x, dtype=np.float64)
        return self.w * x.reshape(len(x), -1).sum(1) + self.b


STAT_MODELS = {"arm_model": (0.002, 3.0), "armstd_model": (0.1, 0.5), "sca_model": (0.001, 4.0), "scastd_model": (0.08, 0.6)}
# layout of the reference's arm/scaffold_num_config.pkl (synthetic content): 9 bounds, 10 (counts, probabilities) bins
NUM_CONFIG = {"bounds": [float(b) for b in np.linspace(12.0, 20.0, 9)],
              "bins": [(tuple(range(2 + i, 6 + i)), (0.1, 0.4, 0.3, 0.2)) for i in range(10)]}


def harness_cases():
    base = dict(num_samples=5, batch_size=3, seed=7, pocket_seed=3)
    return [dict(base, name="ref_prior", prior_mode="ref_prior", num_atoms_mode="ref"),
            dict(base, name="beta_v2", prior_mode="beta_prior", num_atoms_mode="v2"),
            dict(base, name="beta_v2_noscaffold", prior_mode="beta_prior", num_atoms_mode="v2", with_scaffold=False),
            dict(base, name="beta_old", prior_mode="beta_prior", num_atoms_mode="old"),
            # (the reference's `stat` sampler only broadcasts for 3 arms: utils/prior.py:188 zips [1,A] distances with [A,3] stds)
            dict(base, name="beta_stat", prior_mode="beta_prior", num_atoms_mode="stat", stat_models=STAT_MODELS, num_arms=3),
            dict(base, name="subpocket_ref", prior_mode="subpocket", num_atoms_mode="ref"),
            dict(base, name="subpocket_ref_large", prior_mode="subpocket", num_atoms_mode="ref_large"),
            dict(base, name="subpocket_prior", prior_mode="subpocket", num_atoms_mode="prior"),
            # initial types drawn from prior probabilities (torch.multinomial) instead of the uniform Gumbel draw
            # (scripts/sample_diffusion_decomp.py:136-143,304-308: FeaturizeLigandAtom(prior_types=True))
            dict(base, name="ref_prior_typeprobs", prior_mode="ref_prior", num_atoms_mode="ref",
                 atom_probs=[0.3, 0.05, 0.2, 0.05, 0.2, 0.1, 0.05, 0.05], bond_probs=[0.6, 0.25, 0.1, 0.03, 0.02])]


def make_pocket_fields(seed, beta=False, with_scaffold=True, num_arms=2):
    """A small synthetic pocket `data` item (60 protein atoms, 2 arms [+ scaffold]) with the fields the reference's
    harness and transforms read."""
    rng = np.random.default_rng([seed, 991])
    NP, NF = 60, 90
    pos = rng.normal(size=(NP, 3)); pos = pos / np.linalg.norm(pos, axis=1, keepdims=True) * rng.uniform(4, 10, (NP, 1))
    pos = np.round(pos, 3).astype(np.float32)
    A = num_arms
    centers = np.array([[3.0, 0.5, 0.0], [-2.5, 1.0, 1.5], [0.5, 3.0, -2.0]][:A] + [[0.0, -1.0, -0.5]], dtype=np.float32)
    sizes = [3, 1, 2][:A] + [5 if with_scaffold else 0]
    mask = np.concatenate([np.full(n, a) for a, n in zip(list(range(A)) + [-1], sizes)]).astype(np.int64)
    lig = np.concatenate([centers[i] + 0.8 * rng.normal(size=(n, 3)) for i, n in enumerate(sizes)]).astype(np.float32)
    stds = [0.9, 0.7, 0.8][:A] + [1.3]
    t = torch.from_numpy
    arms_prior = [(sizes[a], t(centers[a]), torch.eye(3) * stds[a] ** 2) for a in range(A)]
    scaffold_prior = []
    if with_scaffold:
        cov = torch.tensor(stds[A] ** 2) if beta else torch.eye(3) * stds[A] ** 2
        scaffold_prior = [(sizes[A], t(centers[A]), cov)]
    masks = np.stack([np.linalg.norm(pos - centers[a], axis=1) < 7.0 for a in range(A)])
    full = np.round(rng.normal(size=(NF, 3)) * 8, 3).astype(np.float32)
    return dict(protein_pos=t(pos), protein_element=t(rng.choice([1, 6, 7, 8, 16, 34], NP, p=[.0, .63, .17, .18, .01, .01]).astype(np.int64)),
                protein_is_backbone=t(rng.random(NP) < 0.5), protein_atom_to_aa_type=t(rng.integers(0, 20, NP).astype(np.int64)),
                pocket_atom_masks=t(masks), ligand_atom_mask=t(mask), ligand_pos=t(lig), num_arms=A,
                num_scaffold=1 if with_scaffold else 0, arms_prior=arms_prior, scaffold_prior=scaffold_prior,
                full_protein_pos=t(full))


# ------------------------------------------------------------------------------------------------------------------
# Parity record of the full-chain GPU tests: written BEFORE their assertions, so a run that fails still leaves its numbers.
# gpurun merges gpurun_out/ back into the builder's tree; tools/gpu_round5_evidence.sh copies the file to
# profiles/parity_full_chain.json (committed: what `pytest -q` prints is not kept by the driver).
# ------------------------------------------------------------------------------------------------------------------
def chain_parity_summary(d, every, tol=1e-4, type_mismatches=(0, 0), bound=None, bound_name=None):
    """d [checkpoints, samples] = max |pos - reference| of each sample at each checkpoint."""
    d = np.asarray(d, dtype=np.float64)
    steps = [int(every * (i + 1)) for i in range(d.shape[0])]
    inside = (d < tol).all(1)
    first_out = next((steps[i] for i in range(len(steps)) if not inside[i]), None)
    rec = {"tolerance": tol, "checkpoint_steps": steps, "max_err_per_checkpoint": [float(f"{e:.4g}") for e in d.max(1)],
           "median_err_per_checkpoint": [float(f"{e:.4g}") for e in np.median(d, 1)],
           "per_sample_err_at_end": [float(f"{e:.4g}") for e in d[-1]],
           "samples_within_tol_per_checkpoint": [int(x) for x in (d < tol).sum(1)], "n_samples": int(d.shape[1]),
           "last_checkpoint_with_all_samples_within_tol": next((steps[i] for i in range(len(steps) - 1, -1, -1) if inside[:i + 1].all()), 0),
           "first_checkpoint_with_a_sample_outside_tol": first_out,
           "fraction_within_tol_at_end": float((d[-1] < tol).mean()),
           "type_mismatches": {"atoms": int(type_mismatches[0]), "bonds": int(type_mismatches[1])}}
    if bound is not None:
        rec["bound_used_per_checkpoint"] = [float(f"{e:.4g}") for e in np.broadcast_to(np.asarray(bound, dtype=np.float64), (d.shape[0],))]
        rec["bound"] = bound_name
    return rec


PARITY_LINES = []          # compact lines of this session's parity records: tests/conftest.py prints them in the terminal summary


def note_parity(line):
    """A measured parity number for the terminal summary (shown at -q too: the driver's log keeps only what pytest prints)."""
    PARITY_LINES.append(str(line))


def record_parity(name, rec):
    from decompdiff_amd import hip_lib
    try:
        e = rec["max_err_per_checkpoint"]
        mid = rec["checkpoint_steps"].index(600) if 600 in rec["checkpoint_steps"] else len(e) // 2
        note_parity(f"{name}: types {rec['type_mismatches']['atoms']}+{rec['type_mismatches']['bonds']} mismatches; max |pos - ref| "
                    f"{max(e[:mid + 1]):.2g} through step {rec['checkpoint_steps'][mid]}, {e[-1]:.2g} at step {rec['checkpoint_steps'][-1]}; "
                    f"samples within {rec['tolerance']:g} at the end: {rec['samples_within_tol_per_checkpoint'][-1]}/{rec['n_samples']}")
    except (KeyError, ValueError, IndexError):
        pass
    flags = int(hip_lib.load().dd_build_flags())
    build = "exact_math" if flags & hip_lib.BUILD_EXACT_MATH else ("measurement" if flags & hip_lib.BUILD_DEBUG_OPTIONS else "default")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "gpurun_out", "tests", "parity_full_chain.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        try:
            with open(path) as fh:
                allrec = json.load(fh)
        except (OSError, ValueError):
            allrec = {}
        allrec.setdefault(build, {})[name] = rec
        with open(path, "w") as fh:
            json.dump(allrec, fh, indent=1, sort_keys=True)
    except OSError:
        pass
