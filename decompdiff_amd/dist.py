"""Multi-GPU plumbing: one process per GPU, independent pocket batches per rank.

The reference has no distributed code at all: pockets are sampled one process
per pocket from a shell loop (`-i data_id`, scripts/sample_diffusion_decomp.py:469).  Every
(pocket, sample) chain is independent, so the MI355X mapping is static sharding with NO
data-path collective; RCCL (torch.distributed backend "nccl" on ROCm) is used only for
init / barrier / a final gather of per-rank metadata (timings, checksums).
"""
from __future__ import annotations

import datetime
import gc
import os
import sys
import time
from dataclasses import asdict, dataclass
from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init_from_env(backend: str = None, device_index: int = None) -> bool:
    """Initialise torch.distributed from torchrun's environment.  Returns True if world_size > 1."""
    world, rank, local_rank = env_world()
    if world <= 1:
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver only supports dmabuf IPC (RCCL init)
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank if device_index is None else device_index)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
        if backend == "nccl" and os.environ.get("DD_DIST_NO_FALLBACK") != "1":
            # The job exchanges control messages only (two barriers, one all-reduce of a scalar, one object gather), so a
            # box whose RCCL cannot start (IPC / topology trouble shows up in the first collective, symmetrically on every
            # rank) still measures the same thing over gloo: probe once, fall back loudly.
            try:
                dist.barrier()
                torch.cuda.synchronize()
            except Exception as e:                                   # noqa: BLE001 -- whatever RCCL raises
                print(f"[decompdiff_amd.dist] rank {rank}: RCCL start-up failed ({type(e).__name__}: {str(e)[:200]}); "
                      "control plane falls back to gloo", file=sys.stderr, flush=True)
                try:
                    dist.destroy_process_group()
                except Exception:                                    # noqa: BLE001
                    pass
                dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    return True


def control_backend() -> Optional[str]:
    """Backend the control plane ended up on ('nccl' = RCCL, 'gloo'), None for a single process."""
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else None


def shard_units(n_units: int, rank: int, world: int) -> List[int]:
    """Pocket p -> rank p mod world; every unit lands on exactly one rank."""
    return list(range(rank, n_units, world))


def shard_samples(n_samples: int, rank: int, world: int) -> range:
    """Contiguous shard of the samples of ONE pocket (cfg5-style: num_samples split over ranks)."""
    base, rem = divmod(n_samples, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def barrier(device=None):
    if device is not None and torch.cuda.is_available():
        torch.cuda.synchronize(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:      # (one rank: nothing to wait for)
        dist.barrier()
    if device is not None and torch.cuda.is_available():
        torch.cuda.synchronize(device)


def max_over_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    on_dev = device is not None and dist.get_backend() == "nccl"          # (gloo control plane: host tensor)
    t = torch.tensor([value], dtype=torch.float64, device=device if on_dev else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_metadata(local: Dict[str, Any]) -> List[Dict[str, Any]]:
    """All ranks' small result records (<= 1 KB each): timings, checksums of sampled coordinates/types."""
    if not (dist.is_available() and dist.is_initialized()):
        return [local]
    out: List[Any] = [None] * dist.get_world_size()
    dist.all_gather_object(out, local)
    return out


def checksum(result: Dict[str, torch.Tensor]) -> Dict[str, float]:
    """Order-independent fingerprint of one sampling result (for cross-rank / cross-run comparison)."""
    return {"pos": float(result["pos"].double().sum().item()), "v": int(result["v"].sum().item()),
            "bond": int(result["bond"].sum().item())}


# ------------------------------------------------------------------------------------------------------------------
# Sharded sampling jobs (BASELINE.json configs).  A job is a list of independent *units* -- one
# pocket batch each, the reference's unit of parallelism being one process per pocket
# (scripts/sample_diffusion_decomp.py:469, `-i data_id`).  Units are defined without reference to the world size, so a
# unit's result (and checksum) is the same whichever rank runs it; ranks only differ in WHICH units they take.
# ------------------------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class Unit:
    uid: int                # position in the job
    pocket_seed: int        # synthetic pocket generator seed (synth.make_pocket)
    num_protein: int
    arm_atoms: Tuple[int, ...]
    scaffold_atoms: int
    n_samples: int          # batch size of this unit
    init_seed: int          # torch.manual_seed before the harness draws the initial state (reference: seed_all)
    noise_seed: int         # key of the device Philox streams of the chain
    drift: bool = False


def cfg3_pocket_shape(p: int) -> Tuple[int, Tuple[int, int], int]:
    """Pocket p of the 100-pocket job (configs[3]): NP in [250, 350], NL in [20, 40], two arms + scaffold
    (SURVEY.md 8d 'cfg4'; CrossDocked itself is not available)."""
    rng = np.random.default_rng([int(p), 4242])
    n_p = int(rng.integers(250, 351))
    n_l = int(rng.integers(20, 41))
    arm = max(2, n_l // 4)
    return n_p, (arm, arm), n_l - 2 * arm


def plan_job(config: int, world: int, batch: Optional[int] = None, n_pockets: int = 100, num_samples: int = 64,
             drift: bool = False) -> Tuple[List[Unit], str]:
    """Units of BASELINE.json configs[config] and the scaling kind bench.py reports.

    1: one C-small pocket batch (B=8) per rank -- the headline workload, weak scaling (per-GPU work fixed);
    2: the same with armsca_prox + clash drift guidance;
    3: `n_pockets` pockets (seeds 0.., NP in [250,350], NL in [20,40]) x B=16 -- total work fixed, strong scaling;
    4: one C-large pocket (600 + 60 atoms), `num_samples` samples in shards of B=8 -- total work fixed, strong."""
    if config in (1, 2):
        b = batch or 8
        return [Unit(r, r, 300, (8, 8), 14, b, 2021 + r, 1_000_003 * r + 17, drift or config == 2) for r in range(world)], "weak"
    if config == 3:
        b = batch or 16
        units = []
        for p in range(n_pockets):
            n_p, arms, sca = cfg3_pocket_shape(p)
            units.append(Unit(p, p, n_p, arms, sca, b, 2021 + p, 1_000_003 * p + 17, drift))
        return units, "strong"
    if config == 4:
        b = batch or 8
        n_shards = (num_samples + b - 1) // b
        return [Unit(u, 0, 600, (15, 15), 30, min(b, num_samples - u * b), 2021 + u, 1_000_003 * u + 29, drift)
                for u in range(n_shards)], "strong"
    raise ValueError(f"config {config}: expected 1..4 (index into BASELINE.json configs)")


def units_of_rank(units: List[Unit], config: int, rank: int, world: int) -> List[Unit]:
    """configs 1-3: unit u -> rank u mod world (pockets are independent; round-robin balances the pocket sizes);
    config 4: contiguous shards of one pocket's samples (shard_samples)."""
    if config == 4:
        return [units[i] for i in shard_samples(len(units), rank, world)]
    return [units[i] for i in shard_units(len(units), rank, world)]


def run_job(units: List[Unit], config: int, rank: int, world: int, prepare: Callable[[Unit], Any],
            sample: Callable[[Any, int, int], Dict[str, torch.Tensor]], steps: int, warmup: int, device=None) -> Dict[str, Any]:
    """The timed multi-rank job of bench.py: every rank prepares its units (inputs resident on its device), warms each of
    them up (`warmup` steps: kernels, per-shape launch measurements, graph capture), then -- barrier + device sync on
    both sides -- advances each unit `steps` reverse steps; the wall time is the MAX over ranks.  No data-path
    collective: the only communication is the two barriers, one all-reduce(MAX) of the time and one gather of the
    per-unit records.  `prepare(unit)` -> opaque state; `sample(state, n_steps, seed)` -> result dict with pos / v / bond."""
    mine = units_of_rank(units, config, rank, world)
    states = [prepare(u) for u in mine]
    if warmup > 0:
        for u, st in zip(mine, states):
            checksum(sample(st, warmup, u.noise_seed + 1))      # (also loads the reduction kernels the timed region uses)
    gc.collect()
    gc.disable()                                   # no collector pause inside the timed region (re-enabled below)
    barrier(device)
    t0 = time.perf_counter()
    records = []
    for u, st in zip(mine, states):
        t1 = time.perf_counter()
        out = sample(st, steps, u.noise_seed)
        records.append({"unit": u.uid, "rank": rank, "pocket_seed": u.pocket_seed, "n_samples": u.n_samples,
                        "checksum": checksum(out), "seconds_enqueue": round(time.perf_counter() - t1, 6), "_out": out})
    barrier(device)
    local = time.perf_counter() - t0
    gc.enable()
    elapsed = max_over_ranks(local, device)
    last = records[-1].pop("_out") if records else None
    for r in records:
        r.pop("_out", None)
    gathered = gather_metadata({"rank": rank, "seconds": round(local, 6), "units": records})
    per_unit = sorted((r for g in gathered for r in g["units"]), key=lambda r: r["unit"])
    return {"elapsed": elapsed, "unit_steps": len(units) * steps, "per_rank": [{"rank": g["rank"], "seconds": g["seconds"],
            "units": [r["unit"] for r in g["units"]]} for g in gathered], "per_unit": per_unit, "last_out": last,
            "n_local_units": len(mine)}
