#!/usr/bin/env python
"""bench.py — denoising steps/sec of the reverse-diffusion sampling hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {1,2,3,4}]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

`--gpus N` without a torchrun environment re-executes itself under torch.distributed.run with N ranks (one process per
GPU, RCCL = backend "nccl" for init / two barriers / one all-reduce(MAX) of the wall time / one gather of per-unit
records; no data-path collective — every (pocket, sample) chain is independent).

`--config` indexes BASELINE.json `configs` (decompdiff_amd/dist.py::plan_job):
  1 (default)  one synthetic pocket (300 protein + 30 ligand atoms, ref_prior) per rank, batch of 8 samples — the
               configuration the metric is quoted on; weak scaling (per-GPU work fixed).
  2            the same with armsca_prox + clash drift guidance (configs/sampling_drift.yml).
  3            100 pockets (seeds 0..99, NP in [250,350], NL in [20,40]) x batch 16, pocket p -> rank p mod N; strong.
  4            one C-large pocket (600 + 60 atoms), 64 samples as contiguous shards of batches of 8; strong.
A "step" is one denoising step of one pocket batch: K steps of the 1000-step reverse chain per unit, device Philox
noise, all six trajectories recorded and streamed to the host inside the timed region (what the reference's
`sample_diffusion` returns).  Every unit is warmed up first (W steps: kernels, the one-off per-shape measurement of the
node-launch CU split, graph capture).  `value` = (units x K) / max-over-ranks seconds.

The JSON line also carries
  roofline          the dominant kernel of a step, the fused node-attention launch dd::v2::k_attn2_node (NE + NB + BL
                    sub-layers, 6 launches per step): mean launch duration measured live with HIP events on the launch
                    stream (dd_profile_step, launches serialised), against the fp32 MFMA peak with the FLOPs the matrix
                    cores really EXECUTE (exact v_mfma_f32_16x16x4_f32 count of this launch x 2048); the algorithmic
                    count of SURVEY.md 8d is kept under `algorithmic_tflops` and is not a utilisation;
  roofline_gemm     the projection / query / lin_node GEMM launches of a step (tiles, FLOPs, time, fraction);
  roofline_op_level the HBM-bandwidth regime: the op-level scatter_softmax + scatter_sum kernel of the
                    C ABI on q / k / v tables beyond the Infinity Cache, algorithmic bytes / HIP-event time;
  cpu_baseline      the oracle (CPU restatement of the reference, torch fp32) timed on this host on a bounded sample
                    of the same workload (N = 1 only).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32 vector == fp32 matrix peak
HBM_PEAK_GBS = 8000.0
MFMA_16x16x4_FLOP = 2048


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3, 4], help="index into BASELINE.json configs")
    ap.add_argument("--batch", type=int, default=None, help="samples per pocket batch (default: 8, config 3: 16)")
    ap.add_argument("--pockets", type=int, default=100, help="config 3: number of pockets")
    ap.add_argument("--num-samples", type=int, default=64, help="config 4: samples of the one pocket")
    ap.add_argument("--workload", default="small", choices=["small", "large"], help="configs 1/2: pocket size")
    ap.add_argument("--drift", action="store_true", help="armsca_prox + clash guidance (config 2 implies it)")
    ap.add_argument("--eager", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo lets several "
                                                    "ranks share one GPU for testing)")
    ap.add_argument("--oversubscribe", action="store_true", help="allow more ranks than visible devices (several ranks share a "
                                                                 "GPU; testing only: n_gpus then reports the DISTINCT devices)")
    ap.add_argument("--strict-rccl", action="store_true", help="an RCCL group that cannot start is an error (default: the control "
                                                               "plane -- two barriers, one all-reduce, one gather; the job exchanges no "
                                                               "data -- then runs over the gloo group and the JSON line says so in "
                                                               "config.control_plane / control_plane_note)")
    ap.add_argument("--allow-gloo-fallback", action="store_true", help="(the default since round 4; kept for old command lines)")
    ap.add_argument("--plan-only", action="store_true", help="print the job plan for --gpus N (units per rank, planned cost and imbalance, "
                    "device and CPU slice per rank) as one JSON line and exit; needs no GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rooflines", action="store_true")
    ap.add_argument("--no-steady", action="store_true", help="skip the long call behind the timed region (steady_ms_per_step, per_call_overhead_ms)")
    ap.add_argument("--cold", action="store_true", help="config 3: no warm-up -- every pocket timed from first touch (per-shape launch "
                    "measurement or split-cache lookup, graph capture, then the steps): cold_seconds_per_unit / per_shape_setup_ms")
    ap.add_argument("--cpu-steps", type=int, default=5, help="steps of the CPU baseline's timed sample (~5.5 s each at B=8)")
    ap.add_argument("--cpu-warmup", type=int, default=1)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (default 0: picked by a one-step probe of 16 / 64)")
    return ap.parse_args()


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: launch the N ranks ourselves, same arguments."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------------- FLOP counts
def node_launch_mfma_count(nbr, B, NP, NL, K, lin_in_node=False):
    """Exact number of v_mfma_f32_16x16x4_f32 wave-instructions of one fused node launch (dd_attention2.hip): per
    16-member tile the scores and the aggregation take 32 each; the first-Linear table contraction takes 40 per pass and
    source kind present in the tile (kNN modes: 20 Gaussians = 5 k-steps x 8 channel tiles -- the per-type constant row is added
    on the VALU since round 5, bit-identically; a tile mixing protein and ligand sources runs both tables) or 24 per pass
    (triplets: 12 merged angle codes); node_layer_with_bond has no table.  `nbr` [B,N,K] is the kNN graph of the step.
    `lin_in_node` (round 6): lin_node runs inside the NE / NB blocks as one 16 x 16 x 128 chain per wave -- 32 instructions per
    wave, 8 waves per block of 8 segments; half of the 16 rows repeat the 8 segments (counted separately as "lin")."""
    N = NP + NL
    tiles_e = (K + 15) // 16
    kinds = 0
    for t in range(tiles_e):
        m = nbr[:, :, 16 * t:min(K, 16 * t + 16)]
        has_p = (m < NP).any(-1)
        has_l = (m >= NP).any(-1)
        kinds += int(has_p.sum() + has_l.sum())
    ne = 2 * 40 * kinds + 64 * B * N * tiles_e
    nb = 64 * B * NL * ((NL - 1 + 15) // 16)
    bl = (2 * 24 + 64) * B * NL * (NL - 1) * ((NL - 2 + 15) // 16)
    # round 6: the node_layer_with_edge blocks (8 segments each) run their epilogue W2v . Z~ as one 16 x 16 x 128 chain per wave
    # (32 instructions x 8 waves per block; the two diagonal 8 x 8 blocks of the 16 x 16 result are the outputs)
    ne_blocks = B * ((NP + 7) // 8 + (NL + 7) // 8)
    ne_epi = ne_blocks * 8 * 32
    lin = 0
    if lin_in_node:
        lin = (ne_blocks + (B * NL + 7) // 8) * 8 * 32
    return ne + ne_epi + nb + bl + lin, {"NE": ne, "NE_epilogue": ne_epi, "NB": nb, "BL": bl, "lin": lin}


def algorithmic_flops_node_launch(B, NP, NL, K):
    """SURVEY.md 8d, factored count, of the three sub-layers one fused launch covers (per member two MLPs x (table
    contraction + 128x128 second Linear)): what the reference's layers would execute after the exact first-Linear
    factorisation.  The kernel does NOT execute this (DESIGN.md 3, items 2-3 fold the second Linears away)."""
    N = NP + NL
    e, eb, e3 = B * N * K, B * NL * (NL - 1), B * NL * (NL - 1) * (NL - 2)
    return 2.0 * (e * 2 * (21 * 128 + 128 * 128) + eb * 2 * (128 * 128) + e3 * 2 * (13 * 128 + 128 * 128))


def gemm_work_per_step(B, NP, NL, num_layers, layer0_tables=True, p2_in_pos=False, lin_in_node=False):
    """(useful FLOPs, 64x64 output tiles) of the dense GEMM launches of one step (dd_api.hip forward_impl).  With the
    layer-0 tables (dd_sampler.l0_tables) the first layer's projection and query launches do not run; with `p2_in_pos`
    (dd_debug_schedule() & 1) the projections of the new h and the heads' first Linear run inside the coordinate launch and are
    not GEMM launches any more (their time is then part of the coordinate launch's)."""
    N, Eb = NP + NL, NL * (NL - 1)
    first = [(B * N, 640), (B * NL, 1280), (B * Eb, 640),                     # projections of the old h / h_bond
             (B * Eb, 128), (B * N, 128), (B * NL, 128)]                      # query MLPs, second Linear
    rest = [(B * N, 128), (B * Eb, 256)]                                      # lin_node, bond projections (coordinates)
    if lin_in_node:
        rest = rest[1:]                                                       # (lin_node runs inside the node launch)
    heads = [(B * Eb, 128), (B * NL, 128)]
    if not p2_in_pos:
        rest += [(B * N, 256), (B * NL, 1024)]                                # projections of the new h
    else:
        heads = []
    jobs = (first + rest) * num_layers + heads
    if layer0_tables:
        jobs = jobs[len(first):]
    flops = tiles = 0
    for rows, cols in jobs:
        flops += 2.0 * rows * 128 * cols
        tiles += ((rows + 63) // 64) * ((cols + 63) // 64)
    return flops, tiles


def main():
    args = parse()
    if args.plan_only:
        from decompdiff_amd import dist as ddist
        print(json.dumps({"plan": ddist.describe_plan(args.config, args.gpus, batch=args.batch, n_pockets=args.pockets,
                                                      num_samples=args.num_samples, drift=args.drift),
                          "steps": args.steps, "warmup": args.warmup}), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_torchrun(args))

    import torch
    from decompdiff_amd import DecompScorePosNet3D, hip_lib, shipped_config, synth
    from decompdiff_amd import dist as ddist

    world, rank, local_rank = ddist.env_world()
    if world > 1:
        # torch CPU ops spin up every host core; with one process per GPU that starves the HIP runtime threads of the
        # other ranks (Trajectories) -- give each rank its share of the host
        torch.set_num_threads(max(1, min(16, (os.cpu_count() or 16) // (2 * world))))
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("bench.py needs a HIP device (the sampling hot path has no CPU implementation)")
    try:
        ddist.check_world_fits_devices(world, n_dev, args.oversubscribe)    # one rank per GPU unless asked otherwise
    except ValueError as e:
        raise SystemExit(f"bench.py: {e}")
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    if world > 1 or ddist.force_group():
        # this rank's host threads on the CPUs local to its GPU (ranks sharing a NUMA node take disjoint slices)
        ddist.bind_rank_to_local_cpus(dev_index, local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    dev = torch.device("cuda", dev_index)
    backend = args.backend or "nccl"
    # RCCL on ROCm; no-op for one process.  If RCCL cannot start the control plane stays on gloo, loudly (stderr + the JSON line's
    # control_plane / control_plane_note) -- a measured line with a visible note is worth more than no line; --strict-rccl or
    # DD_DIST_STRICT_RCCL=1 makes it an error instead.
    strict = args.strict_rccl or os.environ.get("DD_DIST_STRICT_RCCL") == "1"
    distributed = ddist.init_from_env(backend=backend, device_index=dev_index, allow_fallback=not strict)

    cfg = shipped_config()
    model = DecompScorePosNet3D(cfg, 29, 10, 8)
    sd = model.state_dict()
    sd.update(synth.synthetic_state_dict(cfg, seed=0))
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    # the job's chain length is known up front (the reference's script always samples num_steps): the warm-up calls size the cached
    # trajectory buffers for it, whatever --warmup / --steps are, instead of leaving a new chain entry + graph capture to the timed call
    model.traj_capacity_hint = min(int(args.steps), int(model.num_timesteps))

    units, scaling = ddist.plan_job(args.config, world, batch=args.batch, n_pockets=args.pockets,
                                    num_samples=args.num_samples, drift=args.drift)
    if args.workload == "large" and args.config in (1, 2):
        units = [ddist.Unit(u.uid, u.pocket_seed, 600, (15, 15), 30, u.n_samples, u.init_seed, u.noise_seed, u.drift) for u in units]
    DRIFT = [dict(type="armsca_prox", min_d=1.2, max_d=1.9), dict(type="clash", sigma=2, gamma=4)]
    cpu_batches = {}

    def prepare(u):
        """One unit's inputs as the reference's harness assembles them (ref_prior), resident on this rank's device."""
        n_full = (4000 if u.num_protein >= 600 else 3000) if u.drift else 0
        pocket = synth.make_pocket(u.pocket_seed, u.num_protein, u.arm_atoms, u.scaffold_atoms, num_full_protein=n_full)
        torch.manual_seed(u.init_seed)
        batch_cpu = synth.build_sampling_batch(pocket, u.n_samples, per_sample_std_scale=[1.0] * u.n_samples)
        cpu_batches[u.uid] = batch_cpu
        batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch_cpu.items()}
        return batch, (DRIFT if u.drift else None)

    def sample(state, n_steps, seed):
        batch, drift = state
        if os.environ.get("DD_BENCH_TRACE") == "1":        # wall-clock stamps of the call's phases on stderr (no extra syncs)
            model.__dict__["_trace"] = tr = [("call<", time.perf_counter())]
            out = model.sample_diffusion(num_steps=n_steps, center_pos_mode="protein", energy_drift_opt=drift, seed=seed,
                                         keep_traj=True, use_graph=not args.eager, **batch)
            tr.append(("call>", time.perf_counter()))
            print(f"[trace] {n_steps} steps: " + "  ".join(f"{k} {1e3 * (t - tr[0][1]):.2f}" for k, t in tr), file=sys.stderr, flush=True)
            model.__dict__["_trace"] = None
            return out
        return model.sample_diffusion(num_steps=n_steps, center_pos_mode="protein", energy_drift_opt=drift, seed=seed,
                                      keep_traj=True, use_graph=not args.eager, **batch)

    if args.cold:
        # configs[3] with the per-pocket costs INSIDE: a 100-pocket job meets ~100 distinct shapes, each sampled
        # once, so the per-shape set-up (CU-split lookup or measurement, graph capture, buffer allocation) is a per-pocket cost.
        # Pass 0 times every unit from first touch, pass 1 the same units again (everything cached): the difference is the set-up.
        if world != 1:
            raise SystemExit("bench.py --cold: single rank only")
        mine = ddist.units_of_rank(units, args.config, 0, 1)
        states = [prepare(u) for u in mine]
        secs = [[], []]
        for p_ in (0, 1):
            for u, st_ in zip(mine, states):
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                sample(st_, args.steps, u.noise_seed)
                torch.cuda.synchronize(dev)
                secs[p_].append(time.perf_counter() - t1)
        lib = hip_lib.load()
        import ctypes as _ct
        buf = _ct.create_string_buffer(1024)
        cache_path = buf.value.decode() if lib.dd_debug_node_split_cache_path(buf, 1024) == 0 else "?"
        cache_path = buf.value.decode()
        setup = [c - w_ for c, w_ in zip(*secs)]
        print(json.dumps({
            "metric": "denoising steps/sec (1000-step reverse) per pocket, every pocket from first touch", "unit": "denoising steps/s",
            "value": round(len(mine) * args.steps / sum(secs[0]), 3), "value_warm": round(len(mine) * args.steps / sum(secs[1]), 3),
            "n_gpus": 1, "steps": args.steps, "units": len(mine), "higher_is_better": True, "data": "synthetic", "dtype": "f32",
            "cold_seconds_per_unit": round(sum(secs[0]) / len(mine), 5), "warm_seconds_per_unit": round(sum(secs[1]) / len(mine), 5),
            "per_shape_setup_ms": round(1e3 * sum(setup) / len(mine), 3), "per_shape_setup_ms_max": round(1e3 * max(setup), 3),
            "setup_fraction_of_cold": round(sum(setup) / sum(secs[0]), 4),
            "node_split_cache_file": cache_path, "node_split_cache_env": os.environ.get("DD_NODE_SPLIT_CACHE", "1"),
            "config": {"workload": f"configs[3] --cold: {len(mine)} pockets (NP in [250,350], NL in [20,40]), batch={mine[0].n_samples}, "
                                   f"{args.steps} steps each, no warm-up; pass 0 from first touch, pass 1 cached"}}), flush=True)
        return
    job = ddist.run_job(units, args.config, rank, world, prepare, sample, args.steps, args.warmup, dev)
    out = job["last_out"]
    if out is not None:
        assert torch.isfinite(out["pos"]).all()
        assert len(out["pos_traj"]) == args.steps
    elapsed = job["elapsed"]

    if rank == 0:
        lib = hip_lib.load()
        steps_per_s = job["unit_steps"] / elapsed
        u0 = ddist.units_of_rank(units, args.config, 0, world)[0]
        B, NP, NL = u0.n_samples, u0.num_protein, sum(u0.arm_atoms) + u0.scaffold_atoms
        K = min(cfg.knn, NP + NL - 1)
        roofline = roofline_gemm = op_roofline = None
        # steady-state figure beside the per-call one (the timed region and `value` are untouched): one long call of the same
        # unit after the timed region -- ms per step once the fixed cost of a call (set-up, trajectory drain, checksum, two syncs) is
        # amortised -- and what that fixed cost is for the timed call
        steady = None
        if args.config in (1, 2) and not args.no_steady:
            st0 = prepare(u0)
            n_long = min(max(200, 10 * args.steps), int(model.num_timesteps))
            # (the long call twice: a process's FIRST call of a new length captures its graph and first-touches the pages of the result
            #  trajectories -- ~0.5 GB for 1000 steps at B = 16, 0.2 s -- which is a per-length set-up cost, not the steady state)
            sample(st0, n_long, u0.noise_seed + 101)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            sample(st0, n_long, u0.noise_seed + 102)
            torch.cuda.synchronize(dev)
            steady_ms = 1e3 * (time.perf_counter() - t1) / n_long
            over = 1e3 * elapsed / max(1, job["n_local_units"]) - args.steps * steady_ms
            steady = {"steady_ms_per_step": round(steady_ms, 4), "steady_steps": n_long,
                      # (only meaningful when the long call is much longer than the timed one, and not when the long call's larger
                      #  trajectory drains make its steps the slower ones: a negative difference is no overhead)
                      "per_call_overhead_ms": round(over, 3) if n_long >= 4 * args.steps and over >= 0 else None}
        if not args.no_rooflines:
            roofline, roofline_gemm = measure_step_rooflines(torch, model, hip_lib, lib, prepare(u0), cfg, B, NP, NL, K, dev,
                                                             args.config, args.workload)
            op_roofline = measure_op_level_roofline(torch, hip_lib, lib, dev)
        cpu = None
        if not args.no_cpu_baseline:                       # (rank 0 only, after the timed region and its closing barrier)
            cpu = cpu_baseline(torch, synth, cfg, cpu_batches[u0.uid], DRIFT if u0.drift else None, B, args.cpu_steps, args.cpu_warmup,
                               args.cpu_threads)
        wl = {1: "configs[1]: single pocket ref_prior",
              2: "configs[2]: single pocket + armsca_prox/clash drift guidance (batch assembled as ref_prior with unit std "
                 "scales; the beta_prior harness mode only changes the initial state and prior stds, not the per-step work)",
              3: f"configs[3]: {len(units)} pockets (NP in [250,350], NL in [20,40])", 4: "configs[4]: C-large pocket, "
              f"{args.num_samples} samples in shards"}[args.config]
        if args.config in (1, 2):
            wl += f", {NP} protein + {NL} ligand atoms, batch={B} per GPU"
        elif args.config == 3:
            wl += f", batch={B} each, pockets assigned longest-first to the least loaded of {world} rank(s)"
        else:
            wl += f" of {B} (600 protein + 60 ligand atoms), contiguous shards over {world} rank(s)"
        wl += ", trajectories recorded and streamed to the host"
        result = {
            "metric": "denoising steps/sec (1000-step reverse) per pocket", "value": round(steps_per_s, 3),
            "unit": "denoising steps/s", "n_gpus": job["distinct_devices"], "ranks": world,
            "devices": job["devices"], "distinct_devices": job["distinct_devices"], "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / max(1, job["n_local_units"] * args.steps), 4), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl, "baseline_config_index": args.config, "batch_per_unit": B, "units": len(units),
                       "sample_steps_per_s": round(steps_per_s * B, 2),
                       "parallelism": f"{world} rank(s), independent pocket batches (no data-path collective)",
                       "control_plane": ddist.control_backend() or "single process",
                       "control_plane_note": ddist.control_note(), "imbalance_max_over_mean_busy": job["imbalance"],
                       "launch": "eager" if args.eager else "hipGraph replay", "noise": "device Philox",
                       "warmup_calls": job["warmup_calls"],    # the W warm-up steps were issued as this many calls
                       "node_launch_split_cus": int(lib.dd_debug_node_split(B, NP, NL, K))},
            "steady_ms_per_step": steady["steady_ms_per_step"] if steady else None,
            "per_call_overhead_ms": steady["per_call_overhead_ms"] if steady else None,
            "steady_steps": steady["steady_steps"] if steady else None,
            "roofline": roofline, "roofline_gemm": roofline_gemm, "roofline_op_level": op_roofline, "cpu_baseline": cpu,
            "per_rank": job["per_rank"], "per_unit": job["per_unit"] if len(units) <= 16 else job["per_unit"][:16],
        }
        if len(units) > 16:
            import hashlib
            result["per_unit_digest"] = hashlib.sha256(json.dumps(job["per_unit"], sort_keys=True, default=str).encode()).hexdigest()[:16]
            result["per_unit_checksum_pos_sum"] = round(sum(r["checksum"]["pos"] for r in job["per_unit"]), 6)
        if world > 1 and ddist.control_backend() != "nccl" and backend == "nccl":
            result["warning"] = ("RCCL did not start: the control plane (two barriers, one all-reduce of the wall time) ran over gloo; "
                                 "the data path has no collective, so `value` is unaffected (--strict-rccl makes this an error)")
        if cpu:
            result["config"]["speedup_vs_cpu_baseline"] = round(steps_per_s / cpu["value"], 1)
        print(json.dumps(result), flush=True)
    if distributed:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def measure_step_rooflines(torch, model, hip_lib, lib, state, cfg, B, NP, NL, K, dev, config, workload):
    """HIP-event timing of every launch class of a step (dd_profile_step: launches serialised on one stream, shipped
    fused launch structure) and the executed-FLOP accounting of the node-attention and GEMM launches."""
    batch, drift = state
    model.sample_diffusion(num_steps=1, center_pos_mode="protein", energy_drift_opt=drift, seed=3, keep_traj=False,
                           use_graph=False, **batch)
    s2, bufs2 = model._last
    torch.cuda.synchronize(dev)
    view = hip_lib.DDWsView()
    hip_lib.check(lib.dd_workspace_view(ctypes.byref(s2), ctypes.byref(view)), "dd_workspace_view")
    ws = bufs2["workspace"]
    off = (view.nbr - ws.data_ptr()) // 4
    nbr = ws[off:off + B * (NP + NL) * K].view(torch.int32).view(B, NP + NL, K).cpu()
    lin_in_node = bool(view.lin_in_node)
    n_mfma, by_mode = node_launch_mfma_count(nbr, B, NP, NL, K, lin_in_node)
    hip_lib.check(lib.dd_sampler_reset(ctypes.byref(s2), hip_lib.stream_ptr(dev)), "dd_sampler_reset")
    # 4 rounds of 5 profiled steps (the first is a warm-up); per launch class the MEAN of the three round means (the minimum
    # of the rounds, kept as launch_ms_min, flattered the fraction by ~4 % against rocprofv3's average: VERDICT r3)
    cats = (ctypes.c_float * len(hip_lib.PROF_CATS))()
    rounds = []
    for rnd in range(4):
        hip_lib.check(lib.dd_sampler_reset(ctypes.byref(s2), hip_lib.stream_ptr(dev)), "dd_sampler_reset")
        hip_lib.check(lib.dd_profile_step(ctypes.byref(s2), 5, cats, hip_lib.stream_ptr(dev)), "dd_profile_step")
        if rnd > 0:
            rounds.append([float(cats[i]) for i in range(len(hip_lib.PROF_CATS))])
    per_cat = {k: sum(r[i] for r in rounds) / len(rounds) for i, k in enumerate(hip_lib.PROF_CATS)}
    per_cat_min = {k: min(r[i] for r in rounds) for i, k in enumerate(hip_lib.PROF_CATS)}
    L = cfg.num_layers
    # fused mode: the NE + NB + BL launch is recorded under attn_BL; the empty event pair recorded once per step
    # measures what the bracket itself adds to a launch (rocprofv3's kernel time has no such term)
    pair_ms = per_cat.get("event_pair", 0.0)
    launch_ms = per_cat["attn_BL"] / L - pair_ms
    executed = n_mfma * MFMA_16x16x4_FLOP
    achieved = executed / (launch_ms * 1e-3) / 1e12
    algorithmic = algorithmic_flops_node_launch(B, NP, NL, K) / (launch_ms * 1e-3) / 1e12
    # HBM traffic cannot be read by the process that is being timed (rocprofv3 --pmc needs its own passes): it comes from the
    # committed PMC passes in profiles/pmc_traffic.json, which name the kernel source they were taken on; an entry measured
    # on another dd_attention2.hip is refused (traffic = null) rather than reported as if it described this run
    traffic, traffic_note, traffic_source = None, "no PMC profile for this workload", None
    try:
        import hashlib
        with open(os.path.join(ROOT, "decompdiff_amd", "csrc", "dd_attention2.hip"), "rb") as fh:
            src_sha = hashlib.sha256(fh.read()).hexdigest()[:16]
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pm = json.load(fh)
        ent = pm.get("node_launch", {}).get(f"NP{NP}_NL{NL}_B{B}")
        if ent and ent.get("kernel_source_sha256_16") == src_sha:
            traffic = int((ent["fetch_kib_per_launch"] + ent["write_kib_per_launch"]) * 1024)
            traffic_note = ent["note"]
            traffic_source = f"committed PMC pass {ent.get('measured_at_commit', '?')} ({ent.get('source', 'profiles/')}); kernel source {src_sha}"
        elif ent:
            traffic_note = (f"stale: the committed PMC pass was taken on dd_attention2.hip {ent.get('kernel_source_sha256_16')}, this "
                            f"run's source is {src_sha} -- re-run the PMC passes (tools/gpu_round6_evidence.sh)")
    except (OSError, KeyError, ValueError):
        pass
    roofline = {
        "bound": "mfma", "kernel": "dd::v2::k_attn2_node (fused node_layer_with_edge + node_layer_with_bond + bond_layer launch, "
                                   f"{L} per step)",
        "achieved": round(achieved, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP32_PEAK_TFLOPS, 4),
        "traffic": traffic, "traffic_source": traffic_source, "traffic_note": traffic_note, "launch_ms": round(launch_ms, 4),
        "launch_ms_min": round(per_cat_min["attn_BL"] / L - per_cat_min.get("event_pair", 0.0), 4), "event_pair_ms": round(pair_ms, 4),
        "mfma_instructions_per_launch": n_mfma, "mfma_instructions_by_sublayer": by_mode,
        "algorithmic_tflops": round(algorithmic, 2),
        "note": "achieved = FLOPs EXECUTED on the matrix cores (exact count of v_mfma_f32_16x16x4_f32 wave-instructions of this "
                "launch from the step's kNN graph x 2048; cross-checked against SQ_INSTS_MFMA in "
                "profiles/ (round 6: 2 596 176 counted, 2 594 760 measured)) / mean duration of the shipped fused launch from HIP events on its stream; VALU work (LayerNorm, "
                "softmax, query fold, epilogue) is not counted.  algorithmic_tflops = SURVEY.md 8d factored FLOPs of the same "
                "three sub-layers / the same time: work the exact restructurings of DESIGN.md 3 remove, not a utilisation.  "
                "The kernel is fused: q / k / v never touch HBM, so it is priced against the fp32 MFMA peak, not HBM.  "
                "On gfx950 fp32 MFMA and fp32 VALU instructions do not overlap on a SIMD, within a wave or across waves "
                "(tools/bench_issue.hip, EXPERIMENTS.md R3-10): with this kernel's ~1:1 MFMA:VALU cycle mix the fraction is "
                "capped near 0.45.",
        "ms_per_step_by_launch_class": {k: round(v, 4) for k, v in per_cat.items()},
    }
    p2_in_pos = bool(lib.dd_debug_schedule() & 1) and NL <= 65
    g_flops, g_tiles = gemm_work_per_step(B, NP, NL, L, layer0_tables=bool(s2.l0_tables), p2_in_pos=p2_in_pos, lin_in_node=lin_in_node)
    g_ms = per_cat["gemm"]
    roofline_gemm = {"bound": "mfma", "kernel": "dd::k_gemm128_batch (projection / query / lin_node / head GEMMs, serialised)",
                     "flops_per_step": g_flops, "tiles_64x64_per_step": g_tiles, "ms_per_step": round(g_ms, 4),
                     "achieved": round(g_flops / (g_ms * 1e-3) / 1e12, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(g_flops / (g_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                     "note": "useful FLOPs (2 x rows x 128 x cols) of every GEMM launch of a step / their summed HIP-event "
                             "durations with the launches serialised (in the step graph half of them overlap on a second stream)"
                             + ("; the projections of the new h and the heads' first Linear run inside the coordinate launch "
                                "(k_attn2_pos_g) and are not counted here" if p2_in_pos else "")}
    return roofline, roofline_gemm


def measure_op_level_roofline(torch, hip_lib, lib, dev):
    """HBM roofline of the op-level message-passing kernel: the stand-alone scatter_softmax +
    scatter_sum op of the C ABI on k / v tables larger than the Infinity Cache, algorithmic bytes / event time."""
    try:
        n_seg, kk = 65536, 32
        E = n_seg * kk
        op_traffic = None                                    # HBM bytes per launch from the committed PMC passes
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                po = json.load(fh)["op_level"]
            if (po["n_seg"], po["edges_per_seg"]) == (n_seg, kk):
                op_traffic = int((po["fetch_kib_per_launch"] * po["fetch_correction"] + po["write_kib_per_launch"]) * 1024)
        except (OSError, KeyError, ValueError):
            pass
        tq, tk, tv = (torch.randn(n, 128, device=dev) for n in (n_seg, E, E))
        tw = torch.rand(E, device=dev)
        tp = (torch.arange(n_seg + 1, device=dev, dtype=torch.int32) * kk).contiguous()
        to = torch.empty(n_seg, 128, device=dev)
        cur = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        call = lambda: hip_lib.check(lib.dd_attn_aggregate_node(hip_lib.ptr(tq), 0, hip_lib.ptr(tk), hip_lib.ptr(tv), hip_lib.ptr(tw),
                                                                hip_lib.ptr(tp), n_seg, hip_lib.ptr(to), cur), "dd_attn_aggregate_node")
        for _ in range(3):
            call()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(10):
            call()
        ev1.record()
        torch.cuda.synchronize()
        sec = ev0.elapsed_time(ev1) / 10 * 1e-3
        nbytes = 1032 * E + 1024 * n_seg
        return {"bound": "hbm", "kernel": "dd_attn_aggregate_node (scatter_softmax + scatter_sum, q/k/v from HBM)",
                "achieved": round(nbytes / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(nbytes / sec / (HBM_PEAK_GBS * 1e9), 4), "traffic": op_traffic,
                "note": f"{n_seg} segments x {kk} edges, {nbytes / 1e6:.0f} MB algorithmic (1032 B/edge + 1024 B/segment, "
                        "SURVEY.md 8d), launch time from HIP events; not on the sampling path (the fused kernels keep "
                        "q/k/v on chip) -- the op-level boundary of the C ABI"}
    except Exception as exc:                                  # the headline numbers do not depend on this extra
        return {"error": str(exc)}


def cpu_baseline(torch, synth, cfg, batch_cpu, drift, B, n_cpu, n_warm, threads=0):
    """The oracle (the checker; CPU restatement of the reference) timed as the CPU baseline on this host: `n_cpu` steps
    after `n_warm` warm-up steps of the same pocket batch.  Small-op torch CPU code does not scale to 100+ threads, so
    the thread count is picked by a one-step probe of 16 / 64 / all threads first."""
    from oracle import diffusion as OD
    weights = synth.synthetic_state_dict(cfg, seed=0)
    n_cpu, n_warm = max(1, n_cpu), max(1, n_warm)
    # (a rank of a multi-GPU job is bound to its share of the CPUs local to its GPU: count those, not the whole host)
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    probe = {}
    for nthreads in ([min(threads, ncpu)] if threads > 0 else sorted({min(16, ncpu), min(64, ncpu)})):     # (all 128-256 threads of such a host are slower still: dropped from the probe)
        torch.set_num_threads(nthreads)
        torch.manual_seed(7)
        OD.sample_diffusion(weights, cfg, num_steps=1, energy_drift_opt=drift, keep_traj=False, **batch_cpu)
        t1 = time.perf_counter()
        OD.sample_diffusion(weights, cfg, num_steps=1, energy_drift_opt=drift, keep_traj=False, **batch_cpu)
        probe[nthreads] = time.perf_counter() - t1
    best = min(probe, key=probe.get)
    torch.set_num_threads(best)
    torch.manual_seed(7)
    OD.sample_diffusion(weights, cfg, num_steps=n_warm, energy_drift_opt=drift, keep_traj=False, **batch_cpu)
    t1 = time.perf_counter()
    OD.sample_diffusion(weights, cfg, num_steps=n_cpu, energy_drift_opt=drift, keep_traj=True, **batch_cpu)
    rate = n_cpu / (time.perf_counter() - t1)
    return {"value": round(rate, 4), "unit": "denoising steps/s", "cores": best, "kind": "port",
            "sample": f"{n_cpu} steps after {n_warm} warm-up steps of the same pocket batch (B={B}); oracle = CPU restatement of "
                      f"the reference (torch fp32); threads {'given (--cpu-threads)' if threads > 0 else 'chosen'} by a 1-step probe of {sorted(probe)} "
                      f"({', '.join(f'{k}: {v:.2f} s/step' for k, v in sorted(probe.items()))}); this process may use {ncpu} of the host's {os.cpu_count()} logical CPUs"}


if __name__ == "__main__":
    main()
