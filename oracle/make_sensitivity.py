"""ORACLE TOOLING (test infrastructure; never imported by the product).

Self-divergence of the ORACLE on the free-running 1000-step drift chain -- the yardstick for the one parity test whose
bound cannot be the flat 1e-4 of BASELINE.json (tests/test_gpu_parity.py::test_chain_1000_steps_golden, drift case).

The chain of tests/golden/traj1000_drift.npz (reference output) is replayed K times through the oracle (bit-exact
restatement of the reference: the unperturbed replay reproduces the fixture with max abs diff 0) and after every reverse
step each ligand coordinate is moved to a NEIGHBOURING fp32 value (-1 / 0 / +1 ulp, independent, probability 1/3 each):
the smallest difference two correct fp32 implementations of the same step can have.  Everything else -- weights, inputs,
the injected noise stream -- is identical.  Stored: for every run and every 50-step checkpoint the maximum coordinate
distance to the reference's trajectory and the number of atom / bond types that differ, plus the quantiles over the runs.

    python -m oracle.make_sensitivity [--runs 8] [--threads 4] [--name traj1000_drift]

writes tests/golden/sens_<name>.npz (K x 7 minutes of CPU).
"""
import argparse
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

POCKET_SEED = {"traj1000_plain": 3, "traj1000_drift": 5, "traj1000_b8_plain": 8, "traj1000_b8_drift": 8}
# the batches of 8 at the bench shape (configs[1] / configs[2], make_golden.py --only b8long / b8long_drift): per-sample prior scales
STD_SCALE = {"traj1000_b8_drift": [1.0, 0.9, 0.8, 1.1, 1.0, 0.95, 1.05, 0.85]}


def ulp_nudge(pos, gen):
    """In place: every element to its lower neighbour, itself or its upper neighbour in fp32 (1/3 each)."""
    if pos.device.type != "cpu":
        s = torch.randint(0, 3, pos.shape, generator=gen).to(pos.device)
        inf = torch.full_like(pos, float("inf"))
        pos.copy_(torch.where(s == 0, torch.nextafter(pos, -inf), torch.where(s == 2, torch.nextafter(pos, inf), pos)))
        return
    a = pos.numpy()
    s = torch.randint(0, 3, pos.shape, generator=gen).numpy()
    lo = np.nextafter(a, np.float32(-np.inf))
    hi = np.nextafter(a, np.float32(np.inf))
    a[...] = np.where(s == 0, lo, np.where(s == 2, hi, a))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=8)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--name", default="traj1000_drift")
    ap.add_argument("--first-seed", type=int, default=9000, help="seed of the first replay (parallel processes: distinct ranges)")
    ap.add_argument("--out", default=None, help="output file (default tests/golden/sens_<name>.npz)")
    ap.add_argument("--steps", type=int, default=0, help="(timing probe) stop after this many steps; nothing is written")
    ap.add_argument("--device", default="cpu", help="cpu (the pinned oracle) or cuda: the SAME oracle code on ATen's HIP kernels -- a third, "
                    "independent fp32 implementation of the step (rocBLAS GEMMs, ATen reductions), ~100 x faster; the un-nudged run is then "
                    "itself a chain that differs from the reference at the ulp level (stored as run -1: seed -1)")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    name = args.name
    g = GU.load(name)
    cfg, sd = GU.weights(int(g["weight_seed"]))
    b = GU.batch_from_npz(g)
    n_steps, every = int(g["num_steps"]), int(g["every"])
    n_data = int(b["batch_ligand"].max()) + 1
    torch.manual_seed(int(g["seed"]))
    synth.build_sampling_batch(synth.make_pocket_small(POCKET_SEED[name]), n_data,
                               per_sample_std_scale=STD_SCALE.get(name))              # advances the generator as make_golden did
    noise = synth.draw_step_noise(n_steps, b["init_ligand_pos"].size(0), b["init_ligand_fc_bond_type"].size(0))
    assert np.allclose(GU.checksum(noise), g["noise_checksum"])
    drift = json.loads(str(g["drift"]))
    dev = torch.device(args.device)
    if dev.type != "cpu":
        sd = {k: v.to(dev) for k, v in sd.items()}
        b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
        noise = {k: v.to(dev) for k, v in noise.items()}
    out_path = args.out or os.path.join(GU.GOLDEN, f"sens_{name}.npz")
    if args.steps:
        import time
        t0 = time.time()
        OD.sample_diffusion(sd, cfg, num_steps=args.steps, energy_drift_opt=drift, noise=noise, **b)
        print(f"{args.steps} steps: {(time.time() - t0) / args.steps:.2f} s / step at {args.threads} threads")
        return
    errs, errs_s, mvs, mbs = [], [], [], []
    seeds = list(range(args.first_seed, args.first_seed + args.runs))
    if dev.type != "cpu":
        seeds = [-1] + seeds                                  # the plain (un-nudged) chain of this implementation first
    for seed in seeds:
        gen = torch.Generator().manual_seed(max(seed, 0))

        def hook(step, t, pos, v, bond, preds, seed=seed):
            if seed >= 0:
                ulp_nudge(pos, gen)

        r = OD.sample_diffusion(sd, cfg, num_steps=n_steps, energy_drift_opt=drift, noise=noise, step_hook=hook, **b)
        r = {k: ([t.cpu() for t in v] if isinstance(v, list) else v.cpu()) for k, v in r.items()}
        run = seed
        tp = torch.stack(r["pos_traj"]).numpy()[every - 1::every]
        errs.append(np.abs(tp.astype(np.float64) - g["traj_pos"]).reshape(len(tp), -1).max(1))
        errs_s.append(np.abs(tp.astype(np.float64) - g["traj_pos"]).reshape(len(tp), n_data, -1).max(2))     # [checkpoint, sample]
        mvs.append((torch.stack(r["v_traj"]).numpy()[every - 1::every] != g["traj_v"]).reshape(len(tp), -1).sum(1))
        mbs.append((torch.stack(r["bond_traj"]).numpy()[every - 1::every] != g["traj_bond"]).reshape(len(tp), -1).sum(1))
        print(f"[sens {name}] run {run}: " + " ".join(f"{e:.2g}" for e in errs[-1]) + f"; type mismatches v={int(mvs[-1].sum())} "
              f"bond={int(mbs[-1].sum())}", flush=True)
        E = np.stack(errs)
        np.savez_compressed(out_path, fixture=np.array(name), perturbation=np.array("-1/0/+1 ulp per coordinate per step, p=1/3 each"),
                            seeds=np.array(seeds[:len(errs)]), device=np.array(str(dev)), every=np.array(every), num_steps=np.array(n_steps),
                            pos_err=E, pos_err_sample=np.stack(errs_s), v_mismatch=np.stack(mvs), bond_mismatch=np.stack(mbs),
                            pos_err_min=E.min(0), pos_err_median=np.median(E, 0), pos_err_max=E.max(0))


if __name__ == "__main__":
    main()
