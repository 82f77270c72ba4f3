#!/usr/bin/env python
"""Where does ONE free-running 1000-step chain sit in the distribution of chains that differ from it at the ulp level?

The bench-shape fixtures (tests/golden/traj1000_b8_plain.npz / traj1000_b8_drift.npz: the reference's own trajectories of 8
samples) are replayed through the HIP path N times; after every reverse step each ligand coordinate is moved to a neighbouring
fp32 value (-1 / 0 / +1 ulp, p = 1/3 each, seeded) -- the perturbation of oracle/make_sensitivity.py, here applied to the product
instead of the oracle (1 000 one-step calls per replay, ~5 s instead of 1.5 h).  Printed per replay: the samples within 1e-4 of the
reference at steps 600 / 1000 and the per-sample end distance; then the pooled quantiles.  DD_HIP_LIB selects the build
(DD_IGNORE_ABI=1 for the round-5 library), so two arithmetic variants can be compared as DISTRIBUTIONS instead of as one draw each.

usage: python tools/chain_replay_distribution.py [plain|drift] [replays=8] [--json out.json]"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_utils as GU                                      # noqa: E402
from decompdiff_amd import DecompScorePosNet3D, shipped_config, synth   # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "plain"
n_rep = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 8
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
name = f"traj1000_b8_{kind}"
scale = [1.0, 0.9, 0.8, 1.1, 1.0, 0.95, 1.05, 0.85] if kind == "drift" else None
g = GU.load(name)
b = GU.batch_from_npz(g)
torch.manual_seed(int(g["seed"]))
synth.build_sampling_batch(synth.make_pocket_small(8), 8, per_sample_std_scale=scale)
T = int(g["num_steps"]); every = int(g["every"])
noise = synth.draw_step_noise(T, b["init_ligand_pos"].size(0), b["init_ligand_fc_bond_type"].size(0))
assert GU.same_checksum(GU.checksum(noise), g["noise_checksum"])
drift = json.loads(str(g["drift"]))
dev = torch.device("cuda:0")
cfg, sd = GU.weights(int(g["weight_seed"]))
m = DecompScorePosNet3D(cfg, 29, 10, 8)
full = m.state_dict(); full.update(sd); m.load_state_dict(full); m = m.to(dev)
bd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
nd = {k: v.to(dev) for k, v in noise.items()}
ref = g["traj_pos"]                                            # [20, 8*30, 3] at steps 50, 100, ...


def replay(seed):
    """seed < 0: the plain chain in one call (the draw the parity tests see)."""
    if seed < 0:
        r = m.sample_diffusion(num_steps=T, center_pos_mode="protein", energy_drift_opt=drift, noise=nd, **bd)
        tp = torch.stack(r["pos_traj"]).numpy()[every - 1::every]
        tv = torch.stack(r["v_traj"]).numpy()[every - 1::every]
        return tp, int((tv != g["traj_v"]).sum())
    gen = torch.Generator().manual_seed(seed)
    cur = dict(bd)
    cks, tvs = [], []
    for k in range(T):
        r = m.sample_diffusion(num_steps=1, start_step=k, center_pos_mode="protein", energy_drift_opt=drift, use_graph=False,
                               noise={kk: vv[k:k + 1] for kk, vv in nd.items()}, **cur)
        pos = r["pos"].detach().cpu()
        if (k + 1) % every == 0:
            cks.append(pos.numpy().copy()); tvs.append(r["v"].cpu().numpy().copy())
        a = pos.numpy()
        s3 = torch.randint(0, 3, pos.shape, generator=gen).numpy()
        a[...] = np.where(s3 == 0, np.nextafter(a, np.float32(-np.inf)), np.where(s3 == 2, np.nextafter(a, np.float32(np.inf)), a))
        cur["init_ligand_pos"] = torch.from_numpy(a).to(dev)
        cur["init_ligand_v"] = r["v"].to(dev)
        cur["init_ligand_fc_bond_type"] = r["bond"].to(dev)
    return np.stack(cks), int((np.stack(tvs) != g["traj_v"]).sum())


rows = []
for seed in [-1] + list(range(7000, 7000 + n_rep)):
    tp, mv = replay(seed)
    d = np.abs(tp.astype(np.float64) - ref).reshape(len(tp), 8, -1).max(2)      # [checkpoint, sample]
    rows.append(dict(seed=seed, within_600=int((d[11] < 1e-4).sum()), within_1000=int((d[-1] < 1e-4).sum()),
                     end=[float(f"{e:.3g}") for e in d[-1]], max_600=float(d[:12].max()), v_mismatch=mv))
    tag = "un-nudged chain, one call" if seed < 0 else f"replay {seed}"
    print(f"{name} {tag}: within 1e-4 at step 600: {rows[-1]['within_600']}/8, at 1000: {rows[-1]['within_1000']}/8; end " +
          " ".join(f"{e:.2g}" for e in d[-1]) + f"; atom-type mismatches {mv}", flush=True)
E = np.array([r["end"] for r in rows[1:]])
w = np.array([r["within_1000"] for r in rows[1:]])
print(f"{name}: {n_rep} nudged replays: samples within 1e-4 at step 1000: mean {w.mean():.2f}/8 (min {w.min()}, max {w.max()}); "
      f"pooled end distance median {np.median(E):.2g}, 90 % {np.quantile(E, 0.9):.2g}, max {E.max():.2g}; the un-nudged chain: "
      f"{rows[0]['within_1000']}/8")
if out_json:
    json.dump(dict(fixture=name, lib=os.environ.get("DD_HIP_LIB", "default"), rows=rows), open(out_json, "w"))
