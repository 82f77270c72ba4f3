#!/usr/bin/env python
"""Merge the per-seed outputs of tools/b8_sensitivity_on_box.sh into tests/golden/sens_traj1000_b8_{plain,drift}.npz."""
import glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in ("traj1000_b8_plain", "traj1000_b8_drift"):
    fs = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "sens", f"sens_{name}_*.npz")))
    if not fs:
        print(name, "no replays found"); continue
    ds = [np.load(f) for f in fs]
    E = np.concatenate([d["pos_err"] for d in ds]); Es = np.concatenate([d["pos_err_sample"] for d in ds])
    out = os.path.join(ROOT, "tests", "golden", f"sens_{name}.npz")
    np.savez_compressed(out, fixture=np.array(name), perturbation=ds[0]["perturbation"], seeds=np.concatenate([d["seeds"] for d in ds]),
                        every=ds[0]["every"], num_steps=ds[0]["num_steps"], pos_err=E, pos_err_sample=Es,
                        v_mismatch=np.concatenate([d["v_mismatch"] for d in ds]), bond_mismatch=np.concatenate([d["bond_mismatch"] for d in ds]),
                        pos_err_min=E.min(0), pos_err_median=np.median(E, 0), pos_err_max=E.max(0))
    print(name, len(fs), "replays ->", out)
    print("  per replay, max over the batch at step 1000:", " ".join(f"{e:.2g}" for e in E[:, -1]))
    print("  samples within 1e-4 at step 1000 per replay:", (Es[:, -1, :] < 1e-4).sum(1), " at step 600:", (Es[:, 11, :] < 1e-4).sum(1))
