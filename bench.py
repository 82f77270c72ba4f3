#!/usr/bin/env python
"""bench.py — denoising steps/sec of the reverse-diffusion sampling hot path on MI355X.

    python bench.py --gpus 1 --steps 1000 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): one synthetic pocket (300 protein + 30 ligand atoms, ref_prior)
per rank, batch of 8 samples, K steps of the 1000-step reverse chain, no drift, device Philox noise,
all six trajectories recorded and streamed to the host inside the timed region (what the reference's
`sample_diffusion` returns); the warm-up call also pays the one-off per-shape measurement of the node-launch
CU split.  A "step" is one denoising step of the whole batch.  Multi-GPU: every
rank samples its own pocket (independent units, weak scaling); RCCL is used for init, the two
barriers and one all-reduce(MAX) of the wall time.  `value` = N * K / max-over-ranks seconds.

The JSON line also carries
  roofline     — the dominant kernel (bond-layer triplet attention), its mean launch duration measured
                 live with HIP events on the launch stream, against the fp32 peak (157.3 TFLOP/s vector =
                 matrix on CDNA4) with the algorithmic FLOP count of DESIGN.md §kernels;
  roofline_op_level — the HBM-bandwidth regime: the op-level scatter_softmax + scatter_sum
                 kernel of the C ABI on q / k / v tables beyond the Infinity Cache, algorithmic bytes / HIP-event time;
  cpu_baseline — the oracle (CPU restatement of the reference, torch fp32, all host threads) timed on a
                 bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from decompdiff_amd import DecompScorePosNet3D, hip_lib, shipped_config, synth  # noqa: E402
from decompdiff_amd import dist as ddist  # noqa: E402

FP32_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32 vector == matrix peak
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8, help="samples per pocket batch (BASELINE configs[1]: 8)")
    ap.add_argument("--workload", default="small", choices=["small", "large"])
    ap.add_argument("--drift", action="store_true", help="BASELINE configs[2]: armsca_prox + clash guidance")
    ap.add_argument("--eager", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2)
    return ap.parse_args()


def executed_flops_bond_layer(B, NL):
    """FLOPs the fused kernel really performs after the exact restructurings of DESIGN.md §3 (query-side folding of
    W2k, value projection after aggregation): per member angle contraction (2 x 13x128) + scores (16x128) +
    aggregation (16x128) MACs, per segment Q~ (128x128) + output projection (128x128) MACs."""
    eb = NL * (NL - 1)
    macs = eb * ((NL - 2) * (2 * 13 * 128 + 2 * 16 * 128) + 2 * 128 * 128)
    return 2.0 * B * macs


def algorithmic_flops_bond_layer(B, NL):
    """FLOPs of one bond_layer attention launch, factored count (SURVEY.md §8d / DESIGN.md):
    per triplet, two MLPs x (13-wide angle contraction + 128x128 second Linear), 2 FLOP per MAC."""
    e3 = NL * (NL - 1) * (NL - 2)
    return 2.0 * B * e3 * 2 * (13 * 128 + 128 * 128)


def main():
    args = parse()
    world, rank, local_rank = ddist.env_world()
    if world > 1:
        # torch CPU ops spin up every host core; with one process per GPU that starves the HIP runtime threads of the
        # other ranks (Trajectories) -- give each rank its share of the host
        torch.set_num_threads(max(1, min(16, (os.cpu_count() or 16) // (2 * world))))
    distributed = ddist.init_from_env(backend="nccl")      # RCCL on ROCm; no-op for a single process
    if not distributed:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if distributed else 0)

    cfg = shipped_config()
    model = DecompScorePosNet3D(cfg, 29, 10, 8)
    sd = model.state_dict()
    sd.update(synth.synthetic_state_dict(cfg, seed=0))
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)

    pocket = synth.make_pocket_small(seed=rank) if args.workload == "small" else synth.make_pocket_large(seed=rank)
    torch.manual_seed(2021 + rank)
    batch_cpu = synth.build_sampling_batch(pocket, args.batch, per_sample_std_scale=[1.0] * args.batch)
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch_cpu.items()}
    drift = [dict(type="armsca_prox", min_d=1.2, max_d=1.9), dict(type="clash", sigma=2, gamma=4)] if args.drift else None
    NP, NL = pocket.num_protein_atoms, pocket.num_ligand_atoms

    def run(n_steps, seed):
        return model.sample_diffusion(num_steps=n_steps, center_pos_mode="protein", energy_drift_opt=drift,
                                      seed=seed, keep_traj=True, use_graph=not args.eager, **batch)

    if args.warmup > 0:
        run(args.warmup, seed=1)
    ddist.barrier(dev)                                     # barrier + torch.cuda.synchronize on both sides
    t0 = time.perf_counter()
    out = run(args.steps, seed=2)
    ddist.barrier(dev)
    elapsed = ddist.max_over_ranks(time.perf_counter() - t0, dev)
    meta = ddist.gather_metadata({"rank": rank, "pocket_seed": rank, "checksum": ddist.checksum(out)})
    assert torch.isfinite(out["pos"]).all()
    assert len(out["pos_traj"]) == args.steps

    result = None
    if rank == 0:
        steps_per_s = world * args.steps / elapsed
        # ---- per-kernel-class timing with HIP events on the launch stream (live, this process)
        s, bufs = model._last
        cats = (ctypes.c_float * len(hip_lib.PROF_CATS))()
        # profile on a fresh short run so the step counter / trajectory indices stay in range
        model.sample_diffusion(num_steps=1, center_pos_mode="protein", energy_drift_opt=drift, seed=3, keep_traj=False,
                               use_graph=False, **batch)
        s2, bufs2 = model._last
        bufs2["step_counter"].zero_()
        n_prof = 10
        lib = hip_lib.load()
        lib.dd_debug_set_fusion(0)                # one launch per sub-layer so that each kernel class is timed alone
        try:
            hip_lib.check(lib.dd_profile_step(ctypes.byref(s2), n_prof, cats, hip_lib.stream_ptr(dev)), "dd_profile_step")
        finally:
            lib.dd_debug_set_fusion(1)
        per_cat = {k: float(cats[i]) for i, k in enumerate(hip_lib.PROF_CATS)}
        dom = max((k for k in per_cat if k.startswith("attn")), key=lambda k: per_cat[k])
        n_layers = cfg.num_layers
        launch_ms = per_cat["attn_BL"] / n_layers
        flops = algorithmic_flops_bond_layer(args.batch, NL)
        achieved = flops / (launch_ms * 1e-3) / 1e12
        executed = executed_flops_bond_layer(args.batch, NL) / (launch_ms * 1e-3) / 1e12
        # HBM bytes per launch of the same kernel: PMC counters cannot be collected from inside this process, so the
        # figure comes from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json) when the workload matches
        traffic, traffic_note = None, "no PMC profile for this workload"
        try:
            import json as _json
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")) as fh:
                pm = _json.load(fh)
            if pm["workload"] == {"name": args.workload, "batch": args.batch}:
                traffic = int((pm["fetch_kib_per_launch"] + pm["write_kib_per_launch"]) * 1024)
                traffic_note = f"FETCH_SIZE + WRITE_SIZE per launch, {pm['collected']} ({pm['source']}); {pm['note']}"
        except (OSError, KeyError, ValueError):
            pass
        roofline = {"bound": "mfma", "kernel": "dd::v2::k_attn2<M_BL> (bond_layer triplet attention)", "achieved": round(achieved, 3),
                    "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP32_PEAK_TFLOPS, 4),
                    "traffic": traffic, "traffic_note": traffic_note, "launch_ms": round(launch_ms, 4), "executed_tflops": round(executed, 3),
                    "executed_frac": round(executed / FP32_PEAK_TFLOPS, 4),
                    "note": "fp32 FLOP roofline (CDNA4 fp32 vector peak == fp32 MFMA peak); achieved = algorithmic FLOPs "
                            "(factored count of SURVEY.md 8d) / live HIP-event launch time of the bond-layer launch, measured "
                            "with one launch per sub-layer; executed_tflops = FLOPs really performed after the exact "
                            "restructurings (DESIGN.md 3,5); the kernel is fused, q/k/v never touch HBM",
                    "ms_per_step_by_kernel_class": {k: round(v, 4) for k, v in per_cat.items()},
                    "dominant_attention_class": dom}
        # ---- HBM roofline of the op-level message-passing kernel: the stand-alone scatter_softmax +
        #      scatter_sum op of the C ABI on k / v tables larger than the Infinity Cache, algorithmic bytes / event time
        op_roofline = None
        try:
            n_seg, kk = 65536, 32
            E = n_seg * kk
            op_traffic = None                                    # HBM bytes per launch from the committed PMC passes
            try:
                import json as _json
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")) as fh:
                    po = _json.load(fh)["op_level"]
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
            op_roofline = {"bound": "hbm", "kernel": "dd_attn_aggregate_node (scatter_softmax + scatter_sum, q/k/v from HBM)",
                           "achieved": round(nbytes / sec / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                           "frac": round(nbytes / sec / 8e12, 4), "traffic": op_traffic,
                           "note": f"{n_seg} segments x {kk} edges, {nbytes / 1e6:.0f} MB algorithmic (1032 B/edge + 1024 B/segment, "
                                   "SURVEY.md 8d), launch time from HIP events; not on the sampling path (the fused kernels keep "
                                   "q/k/v on chip) -- the op-level boundary of the C ABI"}
            del tq, tk, tv, tw, tp, to
        except Exception as exc:                                  # the headline numbers do not depend on this extra
            op_roofline = {"error": str(exc)}
        cpu = None
        if not args.no_cpu_baseline:
            from oracle import diffusion as OD          # the checker, timed as the CPU baseline only
            weights = synth.synthetic_state_dict(cfg, seed=0)
            n_cpu = max(1, args.cpu_steps)
            best = None
            for nthreads in sorted({min(16, os.cpu_count()), min(64, os.cpu_count()), torch.get_num_threads()}):
                torch.set_num_threads(nthreads)           # small-op torch CPU code does not scale to 128 threads
                torch.manual_seed(7)
                OD.sample_diffusion(weights, cfg, num_steps=1, energy_drift_opt=drift, keep_traj=False, **batch_cpu)
                t1 = time.perf_counter()
                OD.sample_diffusion(weights, cfg, num_steps=n_cpu, energy_drift_opt=drift, keep_traj=True, **batch_cpu)
                rate = n_cpu / (time.perf_counter() - t1)
                if best is None or rate > best[0]:
                    best = (rate, nthreads)
            cpu = {"value": round(best[0], 4), "unit": "denoising steps/s", "cores": best[1],
                   "kind": "port", "sample": f"{n_cpu} steps after 1 warm-up step, same pocket batch (B={args.batch}), "
                   f"oracle = CPU restatement of the reference (torch fp32), best of 16/64/all threads; "
                   f"host has {os.cpu_count()} logical CPUs"}
        result = {
            "metric": "denoising steps/sec (1000-step reverse) per pocket", "value": round(steps_per_s, 3),
            "unit": "denoising steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{'configs[1]' if args.workload == 'small' else 'C-large (configs[4] size)'}: single pocket ref_prior, {NP} protein + {NL} ligand atoms, batch={args.batch} "
                                   f"per GPU, {'drift guidance, ' if args.drift else ''}trajectories recorded and streamed to the host",
                       "batch_per_gpu": args.batch, "sample_steps_per_s": round(steps_per_s * args.batch, 2),
                       "parallelism": f"{world} independent pocket batches (no data-path collective)",
                       "launch": "eager" if args.eager else "hipGraph replay", "noise": "device Philox",
                       "node_launch_split_cus": int(lib.dd_debug_node_split(args.batch, NP, NL, min(cfg.knn, NP + NL - 1)))},
            "roofline": roofline, "roofline_op_level": op_roofline, "cpu_baseline": cpu, "per_rank": meta,
        }
        if cpu:
            result["config"]["speedup_vs_cpu_baseline"] = round(steps_per_s / cpu["value"], 1)
        print(json.dumps(result), flush=True)
    if distributed:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
