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


class ControlPlaneError(RuntimeError):
    """RCCL could not start and the gloo fall-back was not asked for."""


_CTL: Dict[str, Any] = {"group": None, "backend": None, "note": None, "forced": False, "affinity": None}   # the group the barriers / reductions use
PROBE_TIMEOUT_S = 90          # RCCL bring-up probe (new_group + one barrier): a rank that fails must not hold the others for long


def force_group() -> bool:
    """DD_DIST_FORCE_GROUP=1: build the process groups even for ONE rank, so that the RCCL branch (new_group("nccl"),
    barrier, all_reduce(MAX)) can be exercised on a single-GPU box (tests/test_gpu_configs.py)."""
    return os.environ.get("DD_DIST_FORCE_GROUP") == "1"


def init_from_env(backend: str = None, device_index: int = None, allow_fallback: Optional[bool] = None) -> bool:
    """Initialise the control plane from torchrun's environment.  Returns True if a process group is in use
    (world_size > 1, or one rank with DD_DIST_FORCE_GROUP=1).

    The default process group is ALWAYS gloo (host sockets: it starts wherever torchrun does).  With backend "nccl"
    (= RCCL; the default on a GPU box) a second group over RCCL is created on top and probed with one barrier; every
    rank then all-reduces (MIN) its success flag over gloo, so all ranks take the SAME decision -- a rank whose RCCL
    failed alone cannot leave the others waiting inside an RCCL collective.  If any rank failed:
      * allow_fallback False (THIS function's default; env DD_DIST_ALLOW_FALLBACK=1 turns it on): ControlPlaneError on
        every rank -- a library caller whose RCCL does not start gets an error, not a silent backend swap;
      * allow_fallback True (what bench.py passes unless --strict-rccl / DD_DIST_STRICT_RCCL=1 is given): the job runs
        its control messages over gloo (it exchanges no data), says so on stderr, and control_backend() /
        control_note() -- `config.control_plane` / `control_plane_note` of bench.py's JSON line, plus a top-level
        `warning` there -- carry it.
    A host that has ALREADY initialised torch.distributed (a torchrun application that embeds the sampler) keeps its
    default group: the control plane then runs on it, with device tensors if it is an RCCL group."""
    world, rank, local_rank = env_world()
    forced = force_group()
    if world <= 1 and not forced and not (dist.is_available() and dist.is_initialized()):
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver only supports dmabuf IPC (RCCL init)
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if allow_fallback is None:
        allow_fallback = os.environ.get("DD_DIST_ALLOW_FALLBACK") == "1"
    if dist.is_initialized():
        # the host's own default group: use it as it is (an RCCL default group takes device tensors, a gloo one host tensors)
        host_backend = str(dist.get_backend()).lower()
        _CTL.update(group=None, backend="nccl" if "nccl" in host_backend else "gloo", forced=forced,
                    note="process group initialised by the host application")
        return dist.get_world_size() > 1 or forced
    dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    _CTL.update(group=None, backend="gloo", note=None, forced=forced)
    if backend != "nccl":
        return True
    dev = local_rank if device_index is None else device_index
    ok, err, group = 1, "", None

    def all_ok(mine: int) -> bool:                                   # gloo: every rank sees the same verdict
        flag = torch.tensor([mine], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1
    try:
        torch.cuda.set_device(dev)
    except Exception as e:                                           # noqa: BLE001
        ok, err = 0, f"{type(e).__name__}: {str(e)[:200]}"
    try:
        group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=PROBE_TIMEOUT_S))   # (every rank must call this)
    except Exception as e:                                           # noqa: BLE001 -- whatever RCCL raises
        ok, err = 0, err or f"{type(e).__name__}: {str(e)[:200]}"
    started = all_ok(ok)
    if started:                                                      # only enter an RCCL collective if every rank can
        try:
            dist.barrier(group=group, device_ids=[dev])
            torch.cuda.synchronize(dev)
        except Exception as e:                                       # noqa: BLE001
            ok, err = 0, f"{type(e).__name__}: {str(e)[:200]}"
        started = all_ok(ok)
    if started:
        _CTL.update(group=group, backend="nccl", note=None)
        return True
    msg = f"RCCL start-up failed on rank {rank}: {err}" if not ok else "RCCL start-up failed on another rank"
    if not allow_fallback:
        try:
            dist.destroy_process_group()
        except Exception:                                            # noqa: BLE001
            pass
        raise ControlPlaneError(msg + " (strict mode: init_from_env(allow_fallback=False), bench.py --strict-rccl or "
                                      "DD_DIST_STRICT_RCCL=1; without it -- bench.py's default, or DD_DIST_ALLOW_FALLBACK=1 -- the "
                                      "control plane stays on gloo, loudly; the job exchanges no data)")
    print(f"[decompdiff_amd.dist] {msg}; control plane stays on gloo", file=sys.stderr, flush=True)
    _CTL.update(group=None, backend="gloo", note="RCCL start-up failed: control plane on gloo (fall-back allowed by the caller; bench.py --strict-rccl / DD_DIST_STRICT_RCCL=1 make this an error)")
    return True


def control_backend() -> Optional[str]:
    """Backend the control plane runs on ('nccl' = RCCL, 'gloo'), None for a single process."""
    return _CTL["backend"] if dist.is_available() and dist.is_initialized() else None


def control_note() -> Optional[str]:
    return _CTL["note"]


def device_identity(device=None) -> Dict[str, Any]:
    """What physically runs this rank: the HIP device's UUID / PCI bus id (distinct per GPU of a node)."""
    if device is None or not torch.cuda.is_available():
        return {"host": os.uname().nodename, "device": None, "uuid": f"cpu:{os.uname().nodename}:{os.getpid()}"}
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    prop = torch.cuda.get_device_properties(idx)
    uuid = str(getattr(prop, "uuid", "")) or None
    bus = None
    for attr in ("pci_bus_id", "pci_device_id", "pci_domain_id"):
        if hasattr(prop, attr):
            bus = (bus or "") + f"{attr[4:-3]}={getattr(prop, attr)} "
    ident = uuid or (bus.strip() if bus else None) or f"index{idx}"
    return {"host": os.uname().nodename, "device": idx, "name": prop.name, "uuid": f"{os.uname().nodename}:{ident}",
            "pci": bus.strip() if bus else None}


def check_world_fits_devices(world: int, n_devices: int, oversubscribe: bool = False) -> None:
    """One rank per GPU: more ranks than visible devices is refused unless the caller explicitly shares GPUs (tests).
    Raises ValueError (bench.py turns it into its exit message)."""
    if world > n_devices and not oversubscribe:
        raise ValueError(f"{world} ranks but only {n_devices} visible HIP device(s): refusing to put several ranks on "
                         "one GPU and report them as GPUs (pass --oversubscribe to share devices on purpose; the JSON line then "
                         "reports n_gpus = distinct devices)")


# ------------------------------------------------------------------------------------------------------------------
# CPU affinity: one process per GPU wants its host threads (the launching thread, the HIP runtime's helper threads, the
# pinned-memory trajectory drain) on the CPUs of the NUMA node its GPU hangs off.  The PCI function of the HIP device
# names its sysfs node, whose `local_cpulist` is that set; ranks whose GPUs share a node take disjoint slices of it.
# ------------------------------------------------------------------------------------------------------------------
def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    out: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            out.extend(range(int(lo), int(hi) + 1))
        else:
            out.append(int(part))
    return sorted(set(out))


def device_local_cpus(device_index: int, sysfs_root: str = "/sys/bus/pci/devices") -> Optional[List[int]]:
    """CPUs local to HIP device `device_index` (its PCI function's local_cpulist), None if it cannot be read."""
    try:
        prop = torch.cuda.get_device_properties(device_index)
        dom, bus, devn = int(getattr(prop, "pci_domain_id", 0)), int(prop.pci_bus_id), int(prop.pci_device_id)
        for fn in range(8):
            path = os.path.join(sysfs_root, f"{dom:04x}:{bus:02x}:{devn:02x}.{fn}", "local_cpulist")
            if os.path.exists(path):
                with open(path) as fh:
                    cpus = parse_cpulist(fh.read())
                return cpus or None
    except Exception:                                                # noqa: BLE001 -- affinity is an optimisation only
        return None
    return None


def slice_cpus(cpus: List[int], share: int, n_shares: int, allowed: Optional[List[int]] = None) -> List[int]:
    """The `share`-th of `n_shares` contiguous slices of `cpus` (restricted to `allowed`); never empty if cpus is not."""
    pool = [c for c in cpus if allowed is None or c in set(allowed)]
    if not pool:
        return []
    n_shares = max(1, n_shares)
    per = max(1, len(pool) // n_shares)
    lo = min(share, n_shares - 1) * per
    hi = len(pool) if share >= n_shares - 1 else lo + per
    return pool[lo:hi] if lo < len(pool) else pool[-per:]


def bind_rank_to_local_cpus(device_index: int, local_rank: int = 0, local_world: int = 1,
                            local_cpus_of: Optional[Callable[[int], Optional[List[int]]]] = None) -> Dict[str, Any]:
    """Pin this process to its share of the CPUs local to its GPU.  Ranks of this node whose devices report the SAME local
    set (one socket / NUMA node) split it evenly in device order; DD_DIST_NO_AFFINITY=1 turns the binding off.  Returns a
    record for bench.py's per_rank entry: {"bound": bool, "cpus": "lo-hi,..", "n_cpus": n, "numa_cpus": m, "why": ...}."""
    rec = _bind(device_index, local_rank, local_world, local_cpus_of)
    _CTL["affinity"] = rec
    return rec


def plan_affinity(device_index: int, local_rank: int = 0, local_world: int = 1,
                  local_cpus_of: Optional[Callable[[int], Optional[List[int]]]] = None) -> Dict[str, Any]:
    """What bind_rank_to_local_cpus WOULD do for this rank, without touching the process's affinity (bench.py --plan-only)."""
    return _bind(device_index, local_rank, local_world, local_cpus_of, dry_run=True)


def _bind(device_index, local_rank, local_world, local_cpus_of, dry_run=False) -> Dict[str, Any]:
    if os.environ.get("DD_DIST_NO_AFFINITY") == "1":
        return {"bound": False, "why": "DD_DIST_NO_AFFINITY=1"}
    if not hasattr(os, "sched_setaffinity"):
        return {"bound": False, "why": "no sched_setaffinity on this platform"}
    look = local_cpus_of or device_local_cpus
    mine = look(device_index)
    if not mine:
        return {"bound": False, "why": "local_cpulist of the device is not readable"}
    try:
        allowed = sorted(os.sched_getaffinity(0))
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else max(local_world, device_index + 1)
        # ranks run on devices (local_rank % n_dev); those whose device shares my CPU set, in rank order
        peers = [r for r in range(max(1, local_world)) if (look(r % max(1, n_dev)) or []) == mine]
        share = peers.index(local_rank) if local_rank in peers else 0
        cpus = slice_cpus(mine, share, len(peers) or 1, allowed)
        if not cpus:
            return {"bound": False, "why": "no allowed CPU in the device's local set", "numa_cpus": len(mine)}
        if not dry_run:
            os.sched_setaffinity(0, cpus)
        return {"bound": not dry_run, "planned": True, "cpus": format_cpulist(cpus), "n_cpus": len(cpus), "numa_cpus": len(mine),
                "share": f"{share + 1}/{len(peers) or 1}"}
    except Exception as e:                                           # noqa: BLE001
        return {"bound": False, "why": f"{type(e).__name__}: {str(e)[:120]}"}


def format_cpulist(cpus: List[int]) -> str:
    """[0, 1, 2, 3, 8] -> '0-3,8'."""
    out, i = [], 0
    cpus = sorted(cpus)
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)


def shard_units(n_units: int, rank: int, world: int) -> List[int]:
    """Pocket p -> rank p mod world; every unit lands on exactly one rank."""
    return list(range(rank, n_units, world))


def shard_samples(n_samples: int, rank: int, world: int) -> range:
    """Contiguous shard of the samples of ONE pocket (cfg5-style: num_samples split over ranks)."""
    base, rem = divmod(n_samples, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def _in_group() -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _CTL["forced"])


def barrier(device=None):
    if device is not None and torch.cuda.is_available():
        torch.cuda.synchronize(device)
    if _in_group():                                                   # (one unforced rank: nothing to wait for)
        if _CTL["backend"] == "nccl":
            dist.barrier(group=_CTL["group"], device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier(group=_CTL["group"])
    if device is not None and torch.cuda.is_available():
        torch.cuda.synchronize(device)


def max_over_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    on_dev = _CTL["backend"] == "nccl"                                # RCCL group: device tensor; gloo: host tensor
    if on_dev and device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([value], dtype=torch.float64, device=device if on_dev else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_CTL["group"])
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
    # (one device -> host copy for the three sums, not three synchronisations)
    sums = torch.stack([result["pos"].double().sum(), result["v"].sum().double(), result["bond"].sum().double()]).cpu()
    return {"pos": float(sums[0]), "v": int(sums[1]), "bond": int(sums[2])}


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


def unit_cost(u: Unit) -> float:
    """Relative cost of one reverse step of a unit (for load balancing only): per sample, node-attention blocks of 8
    centres at 1.0 and bond-layer batches of 8 segments at 0.1 + 0.35 per 16-member tile (the unit costs the node-launch
    split measurement starts from, dd_api.hip::autotune_node_split) -- the NL^3 term dominates: the 100 pockets of configs[3] spread 3.5x."""
    n_l = sum(u.arm_atoms) + u.scaffold_atoms
    n = u.num_protein + n_l
    tiles = (max(n_l - 2, 1) + 15) // 16
    return u.n_samples * (n / 8.0 + n_l / 8.0 + n_l * (n_l - 1) / 8.0 * (0.1 + 0.35 * tiles))


def assign_lpt(units: List[Unit], world: int) -> List[List[int]]:
    """Longest-processing-time-first: units in descending cost, each to the least loaded rank so far (ties: lowest rank).
    Deterministic and a function of (units, world) only, so every rank computes the same table without talking."""
    order = sorted(range(len(units)), key=lambda i: (-unit_cost(units[i]), i))
    load = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += unit_cost(units[i])
    return out


def units_of_rank(units: List[Unit], config: int, rank: int, world: int) -> List[Unit]:
    """configs 1-2: unit u -> rank u (one identical unit per rank); config 3: the pockets differ 3.5x in cost (NL^3 bond-layer term), so
    they are assigned longest-first to the least loaded rank (assign_lpt) instead of p mod N;
    config 4: contiguous shards of one pocket's samples (shard_samples)."""
    if config == 4:
        return [units[i] for i in shard_samples(len(units), rank, world)]
    if config == 3:
        return [units[i] for i in assign_lpt(units, world)[rank]]
    return [units[i] for i in shard_units(len(units), rank, world)]


def describe_plan(config: int, world: int, batch: Optional[int] = None, n_pockets: int = 100, num_samples: int = 64,
                  drift: bool = False, local_cpus_of: Optional[Callable[[int], Optional[List[int]]]] = None,
                  n_devices: Optional[int] = None) -> Dict[str, Any]:
    """The job `bench.py --gpus world --config config` would run, without running it (bench.py --plan-only; no GPU needed):
    the units, which rank takes which (LPT table for configs[3], contiguous shards for configs[4]), each rank's planned
    cost and the imbalance of the plan, the device and the CPU slice every rank would bind to.  A function of the arguments
    (and, for the affinity, of this host's PCI topology) only -- every rank computes the same table without talking."""
    units, scaling = plan_job(config, world, batch=batch, n_pockets=n_pockets, num_samples=num_samples, drift=drift)
    if n_devices is None:
        n_devices = torch.cuda.device_count() if torch.cuda.is_available() else 0
    ranks = []
    for r in range(world):
        mine = units_of_rank(units, config, r, world)
        rec = {"rank": r, "units": [u.uid for u in mine], "n_units": len(mine),
               "planned_cost": round(sum(unit_cost(u) for u in mine), 1),
               "device": (r % n_devices) if n_devices else None}
        if n_devices or local_cpus_of is not None:
            rec["cpu_affinity"] = plan_affinity(r % max(1, n_devices), r, world, local_cpus_of)
        else:
            rec["cpu_affinity"] = {"bound": False, "why": "no HIP device visible here: the PCI topology is read on the GPU host"}
        ranks.append(rec)
    costs = [x["planned_cost"] for x in ranks]
    mean = sum(costs) / max(1, len(costs))
    shapes = sorted({(u.num_protein, sum(u.arm_atoms) + u.scaffold_atoms, u.n_samples) for u in units})
    return {"config": config, "world": world, "scaling": scaling, "n_units": len(units), "ranks": ranks,
            "planned_imbalance_max_over_mean": round(max(costs) / mean, 4) if mean > 0 else None,
            "distinct_shapes_NP_NL_B": len(shapes), "shapes_sample": shapes[:6],
            "assignment": {3: "longest-processing-time first (assign_lpt)", 4: "contiguous sample shards (shard_samples)"}.get(
                config, "unit u -> rank u (one identical pocket batch per rank)"),
            "collectives": "none on the data path; control plane = 2 barriers + 1 all-reduce(MAX) of the wall time + 1 gather of records"}


def run_job(units: List[Unit], config: int, rank: int, world: int, prepare: Callable[[Unit], Any],
            sample: Callable[[Any, int, int], Dict[str, torch.Tensor]], steps: int, warmup: int, device=None) -> Dict[str, Any]:
    """The timed multi-rank job of bench.py: every rank prepares its units (inputs resident on its device), warms each of
    them up (`warmup` steps: kernels, per-shape launch measurements, graph capture), then -- barrier + device sync on
    both sides -- advances each unit `steps` reverse steps; the wall time is the MAX over ranks.  No data-path
    collective: the only communication is the two barriers, one all-reduce(MAX) of the time and one gather of the
    per-unit records.  `prepare(unit)` -> opaque state; `sample(state, n_steps, seed)` -> result dict with pos / v / bond."""
    mine = units_of_rank(units, config, rank, world)
    states = [prepare(u) for u in mine]
    # The collector runs BEFORE the warm-up, not between the warm-up and the timed region: a full collection of a torch process
    # takes tens of ms during which host and device idle, and the first call after it ran ~1.3 ms slower than a steady call
    # (profiles/round5_call_trace.txt: cold host caches in the call's set-up, +0.45 ms in its first 8 steps on the device).
    gc.collect()
    gc.disable()                                   # no collector pause inside the timed region (re-enabled below)
    try:
        if warmup > 0:
            # the W warm-up steps of a unit as up to DD_BENCH_WARMUP_CALLS separate calls (default 5; the step count stays W): a
            # call's host-side set-up and collection run ~0.7 ms slower the first few times they run in a process
            # (profiles/round5_call_trace_before.txt: 26.2 / 25.2 / 24.9 ms for three successive 20-step calls), and a sampling job
            # is many calls -- measured 787-796 steps/s on the 20-step line with one warm-up call, 811-813 with five
            # (profiles/round5_warmup_calls_ab.txt)
            n_calls = max(1, min(int(os.environ.get("DD_BENCH_WARMUP_CALLS", "5")), warmup))
            for u, st in zip(mine, states):
                left = warmup
                for c in range(n_calls):
                    n = left if c == n_calls - 1 else max(1, warmup // n_calls)
                    checksum(sample(st, n, u.noise_seed + 1 + c))   # (also loads the reduction kernels the timed region uses)
                    left -= n
        barrier(device)
        t0 = time.perf_counter()
        records = []
        trace = os.environ.get("DD_BENCH_TRACE") == "1"
        for u, st in zip(mine, states):
            t1 = time.perf_counter()
            out = sample(st, steps, u.noise_seed)
            t2 = time.perf_counter()
            cs = checksum(out)
            records.append({"unit": u.uid, "rank": rank, "pocket_seed": u.pocket_seed, "n_samples": u.n_samples,
                            "checksum": cs, "seconds_enqueue": round(time.perf_counter() - t1, 6), "_out": out})
            if trace:
                print(f"[trace] unit {u.uid}: since t0 {1e3 * (t1 - t0):.3f} ms, sample {1e3 * (t2 - t1):.3f} ms, checksum "
                      f"{1e3 * (time.perf_counter() - t2):.3f} ms", file=sys.stderr, flush=True)
        if device is not None and torch.cuda.is_available():
            torch.cuda.synchronize(device)                     # this rank's own work, without the wait for the others
        busy = time.perf_counter() - t0
        barrier(device)
        local = time.perf_counter() - t0
        if trace:
            print(f"[trace] busy {1e3 * busy:.3f} ms, local {1e3 * local:.3f} ms", file=sys.stderr, flush=True)
    finally:
        gc.enable()
    elapsed = max_over_ranks(local, device)
    last = records[-1].pop("_out") if records else None
    for r in records:
        r.pop("_out", None)
    gathered = gather_metadata({"rank": rank, "seconds": round(local, 6), "busy_seconds": round(busy, 6), "units": records,
                                "device": device_identity(device), "cost": round(sum(unit_cost(u) for u in mine), 1),
                                "affinity": _CTL["affinity"]})
    per_unit = sorted((r for g in gathered for r in g["units"]), key=lambda r: r["unit"])
    busy_all = [g["busy_seconds"] for g in gathered]
    devices = [g["device"]["uuid"] for g in gathered]
    return {"elapsed": elapsed, "unit_steps": len(units) * steps,
            "per_rank": [{"rank": g["rank"], "seconds": g["seconds"], "busy_seconds": g["busy_seconds"], "planned_cost": g["cost"],
                          "units": [r["unit"] for r in g["units"]], "device": g["device"], "cpu_affinity": g.get("affinity")}
                         for g in gathered],
            "imbalance": round(max(busy_all) / (sum(busy_all) / len(busy_all)), 4) if min(busy_all) > 0 else None,
            "devices": devices, "distinct_devices": len(set(devices)),
            "per_unit": per_unit, "last_out": last, "n_local_units": len(mine),
            "warmup_calls": max(1, min(int(os.environ.get("DD_BENCH_WARMUP_CALLS", "5")), warmup)) if warmup > 0 else 0}
