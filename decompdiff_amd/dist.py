"""Multi-GPU plumbing: one process per GPU, independent pocket batches per rank.

The reference has no distributed code at all: pockets are sampled one process
per pocket from a shell loop (`-i data_id`, scripts/sample_diffusion_decomp.py:469).  Every
(pocket, sample) chain is independent, so the MI355X mapping is static sharding with NO
data-path collective; RCCL (torch.distributed backend "nccl" on ROCm) is used only for
init / barrier / a final gather of per-rank metadata (timings, checksums).
"""
from __future__ import annotations

import os
from typing import Any, Dict, List

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init_from_env(backend: str = None) -> bool:
    """Initialise torch.distributed from torchrun's environment.  Returns True if world_size > 1."""
    world, rank, local_rank = env_world()
    if world <= 1:
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return True


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
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if device is not None and torch.cuda.is_available():
        torch.cuda.synchronize(device)


def max_over_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
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
