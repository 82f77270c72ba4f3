"""CPU, world_size 2 over gloo: the multi-GPU path of bench.py (sharding of independent pocket
batches, barrier, max-over-ranks timing, metadata gather) is correct by construction."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from decompdiff_amd import dist as ddist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_units, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    assert ddist.init_from_env(backend="gloo")
    mine = ddist.shard_units(n_units, rank, world)
    samples = list(ddist.shard_samples(13, rank, world))
    ddist.barrier()
    elapsed = 1.0 + rank                                   # pretend rank 1 is slower
    t_max = ddist.max_over_ranks(elapsed)
    fake = {"pos": torch.full((3, 3), float(rank)), "v": torch.tensor([rank]), "bond": torch.tensor([2 * rank])}
    meta = ddist.gather_metadata({"rank": rank, "units": mine, "samples": samples, "steps_per_s": 100.0 / elapsed,
                                  "checksum": ddist.checksum(fake)})
    ddist.barrier()
    if rank == 0:
        q.put((t_max, meta))
    dist.destroy_process_group()


def test_two_rank_sharding_and_reduction():
    world, n_units = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_units, q)) for r in range(world)]
    for p in procs:
        p.start()
    t_max, meta = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert t_max == 2.0                                     # MAX over ranks, as bench.py reports
    units = sorted(u for m in meta for u in m["units"])
    assert units == list(range(n_units))                    # every pocket exactly once
    samples = sorted(s for m in meta for s in m["samples"])
    assert samples == list(range(13))
    assert [m["rank"] for m in meta] == [0, 1]
    assert meta[1]["checksum"] == {"pos": 9.0, "v": 1, "bond": 2}
    # weak scaling aggregate = sum of per-rank steps over the max time
    agg = world * 100.0 / t_max
    assert agg == 100.0


def test_single_process_paths_are_noops():
    assert ddist.shard_units(5, 0, 1) == [0, 1, 2, 3, 4]
    assert list(ddist.shard_samples(5, 0, 1)) == [0, 1, 2, 3, 4]
    assert ddist.max_over_ranks(3.5) == 3.5
    assert ddist.gather_metadata({"a": 1}) == [{"a": 1}]


# ------------------------------------------------------------------------------------------------------------------
# The sharded job driver of bench.py (dist.plan_job / units_of_rank / run_job) with a stub model: the same code path the
# multi-GPU bench runs, on gloo.  Units are defined independently of the world size, so the union of the per-unit
# records of a 2-rank run must equal those of a 1-rank run of the same job.
# ------------------------------------------------------------------------------------------------------------------
def _stub_prepare(u):
    g = torch.Generator().manual_seed(u.init_seed)
    n_l = sum(u.arm_atoms) + u.scaffold_atoms
    return {"pos0": torch.randn(u.n_samples * n_l, 3, generator=g), "n_l": n_l, "uid": u.uid}


def _stub_sample(state, n_steps, seed):
    """Deterministic stand-in for model.sample_diffusion: a function of (initial state, steps, seed) only."""
    g = torch.Generator().manual_seed(seed)
    pos = state["pos0"].clone()
    for _ in range(n_steps):
        pos = 0.9 * pos + 0.1 * torch.randn(pos.shape, generator=g)
    n = pos.shape[0]
    return {"pos": pos, "v": torch.arange(n) % 8, "bond": (torch.arange(n * 3) + seed) % 5, "pos_traj": [pos] * n_steps}


def _job_worker(rank, world, port, config, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    assert ddist.init_from_env(backend="gloo")
    units, scaling = ddist.plan_job(config, world, n_pockets=7, num_samples=40)
    job = ddist.run_job(units, config, rank, world, _stub_prepare, _stub_sample, steps=3, warmup=1)
    if rank == 0:
        q.put((scaling, job["per_unit"], job["per_rank"], job["unit_steps"], job["elapsed"]))
    dist.barrier()
    dist.destroy_process_group()


def _run_job_world(world, config):
    if world == 1:
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            os.environ.pop(k, None)
        units, scaling = ddist.plan_job(config, 1, n_pockets=7, num_samples=40)
        job = ddist.run_job(units, config, 0, 1, _stub_prepare, _stub_sample, steps=3, warmup=1)
        return scaling, job["per_unit"], job["per_rank"], job["unit_steps"], job["elapsed"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_job_worker, args=(r, world, port, config, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_sharded_job_two_ranks_equals_one_rank():
    for config in (3, 4):
        s1, units1, ranks1, steps1, _ = _run_job_world(1, config)
        s2, units2, ranks2, steps2, t2 = _run_job_world(2, config)
        assert s1 == s2 == "strong" and steps1 == steps2 and t2 > 0
        strip = lambda rs: [{k: v for k, v in r.items() if k not in ("rank", "seconds_enqueue")} for r in rs]
        assert strip(units1) == strip(units2)                      # same units, same checksums, whoever ran them
        assert [u["unit"] for u in units2] == list(range(len(units2)))
        assert sorted(u for r in ranks2 for u in r["units"]) == list(range(len(units2)))
        if config == 3:
            units, _ = ddist.plan_job(3, 2, n_pockets=7)
            want = ddist.assign_lpt(units, 2)                                                  # longest-first, least loaded rank
            assert [r["units"] for r in ranks2] == want and sorted(want[0] + want[1]) == list(range(7))
            assert ranks2[0]["planned_cost"] > 0 and "busy_seconds" in ranks2[0] and ranks2[0]["device"]["uuid"] != ranks2[1]["device"]["uuid"]
        else:
            assert len(units2) == 5 and units2[-1]["n_samples"] == 8                           # 40 samples = 5 shards of 8
            assert ranks2[0]["units"] == [0, 1, 2] and ranks2[1]["units"] == [3, 4]            # contiguous sample shards


def test_plan_job_weak_configs_scale_with_world():
    for world in (1, 2, 8):
        units, scaling = ddist.plan_job(1, world)
        assert scaling == "weak" and len(units) == world and all(u.n_samples == 8 for u in units)
        assert [ddist.units_of_rank(units, 1, r, world)[0].uid for r in range(world)] == list(range(world))
    units, _ = ddist.plan_job(2, 1)
    assert units[0].drift
    units, scaling = ddist.plan_job(3, 8)
    assert len(units) == 100 and scaling == "strong" and all(u.n_samples == 16 for u in units)
    assert all(250 <= u.num_protein <= 350 and 20 <= sum(u.arm_atoms) + u.scaffold_atoms <= 40 for u in units)
    units, _ = ddist.plan_job(4, 8)
    assert len(units) == 8 and all(u.num_protein == 600 and sum(u.arm_atoms) + u.scaffold_atoms == 60 for u in units)


def test_lpt_assignment_balances_the_pocket_costs():
    """configs[3]: pocket costs spread 3.5x (the NL^3 bond-layer term); longest-first assignment keeps the planned per-rank load within a few
    per cent where p mod N does not have to, covers every pocket once, and is a pure function of (units, world)."""
    units, _ = ddist.plan_job(3, 8)
    cost = [ddist.unit_cost(u) for u in units]
    assert max(cost) / min(cost) > 3.0
    for world in (2, 4, 8):
        table = ddist.assign_lpt(units, world)
        assert sorted(i for r in table for i in r) == list(range(100))
        assert table == ddist.assign_lpt(list(units), world)
        load = [sum(cost[i] for i in r) for r in table]
        naive = [sum(cost[i] for i in range(r, 100, world)) for r in range(world)]
        assert max(load) / (sum(load) / world) < 1.03
        assert max(load) <= max(naive) + 1e-9
        assert [u.uid for u in ddist.units_of_rank(units, 3, 1, world)] == table[1]
    # the heaviest pocket goes first, to rank 0
    assert ddist.assign_lpt(units, 8)[0][0] == max(range(100), key=lambda i: (cost[i], -i))


def test_more_ranks_than_devices_is_refused():
    import pytest
    with pytest.raises(ValueError) as e:                    # (a library error; bench.py turns it into its exit message)
        ddist.check_world_fits_devices(8, 1)
    assert "8 ranks but only 1 visible" in str(e.value)
    ddist.check_world_fits_devices(8, 1, oversubscribe=True)
    ddist.check_world_fits_devices(8, 8)


def _fallback_worker(rank, world, port, allow, q):
    """backend 'nccl' on a box without a GPU: the RCCL group cannot start on any rank -> every rank takes the same branch."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    try:
        ddist.init_from_env(backend="nccl", device_index=0, allow_fallback=allow)
        ddist.barrier()
        t = ddist.max_over_ranks(1.0 + rank)
        q.put((rank, "ok", ddist.control_backend(), ddist.control_note(), t))
        dist.destroy_process_group()
    except ddist.ControlPlaneError as e:
        q.put((rank, "error", str(e), None, None))


def _run_fallback(allow):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fallback_worker, args=(r, 2, port, allow, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_rccl_failure_is_an_error_unless_the_fallback_is_requested():
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box where RCCL cannot start (no GPU)")
    res = _run_fallback(False)
    assert [r[1] for r in res] == ["error", "error"] and "RCCL start-up failed" in res[0][2]
    res = _run_fallback(True)
    assert [r[1] for r in res] == ["ok", "ok"] and all(r[2] == "gloo" and "fall-back" in r[3] and r[4] == 2.0 for r in res)


def _host_group_worker(rank, world, port, q):
    """A host application that has ALREADY initialised torch.distributed (torchrun + its own init_process_group): the
    control plane must run on the host's default group instead of leaving its bookkeeping empty."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    assert ddist.init_from_env(backend="nccl") is True        # must not try to start anything of its own
    assert ddist.control_backend() == "gloo" and "host application" in ddist.control_note()
    ddist.barrier()
    t = ddist.max_over_ranks(3.0 - rank)
    meta = ddist.gather_metadata({"rank": rank})
    q.put((rank, t, [m["rank"] for m in meta]))
    dist.destroy_process_group()


def test_host_initialised_process_group_is_used_as_it_is():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_host_group_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, 3.0, [0, 1]), (1, 3.0, [0, 1])]


def _forced_single_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      DD_DIST_FORCE_GROUP="1")
    assert ddist.init_from_env(backend="gloo") is True        # one rank, but the groups are built (DD_DIST_FORCE_GROUP)
    ddist.barrier()
    t = ddist.max_over_ranks(1.25)
    q.put((ddist.control_backend(), t, len(ddist.gather_metadata({"rank": 0}))))
    dist.destroy_process_group()


def test_forced_single_rank_group():
    """DD_DIST_FORCE_GROUP=1: the collective branch runs with one rank (the GPU suite uses the same switch to bring RCCL up
    on a one-GPU box: tests/test_gpu_configs.py::test_rccl_single_rank_control_plane)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_single_worker, args=(_free_port(), q))
    p.start()
    assert q.get(timeout=120) == ("gloo", 1.25, 1)
    p.join(timeout=60)
    assert p.exitcode == 0


def _eight_rank_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    assert ddist.init_from_env(backend="gloo")
    units, scaling = ddist.plan_job(3, world, n_pockets=16)

    def prepare(u):
        return u

    def sample(u, n_steps, seed):                             # a stand-in chain: the result depends on the unit only
        g = torch.Generator().manual_seed(u.pocket_seed * 7919 + n_steps)
        return {"pos": torch.rand(u.n_samples, 3, generator=g), "v": torch.tensor([u.uid]), "bond": torch.tensor([seed % 97])}
    job = ddist.run_job(units, 3, rank, world, prepare, sample, steps=3, warmup=1, device=None)
    if rank == 0:
        q.put({k: job[k] for k in ("unit_steps", "per_rank", "per_unit", "distinct_devices", "imbalance")})
    ddist.barrier()
    dist.destroy_process_group()


def test_eight_rank_job_over_gloo():
    """The shape of the driver's `--gpus 8` run (8 ranks, LPT table, gather, imbalance), on CPU ranks over gloo."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eight_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    job = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert job["unit_steps"] == 16 * 3
    assert [r["rank"] for r in job["per_rank"]] == list(range(8))
    units, _ = ddist.plan_job(3, world, n_pockets=16)
    table = ddist.assign_lpt(units, world)
    assert [r["units"] for r in job["per_rank"]] == table
    assert [r["unit"] for r in job["per_unit"]] == list(range(16))
    assert job["distinct_devices"] == 8                       # (CPU ranks identify themselves by pid)
    assert all("cpu_affinity" in r for r in job["per_rank"])


def test_cpu_affinity_helpers(monkeypatch):
    assert ddist.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert ddist.format_cpulist([0, 1, 2, 3, 8, 10, 11]) == "0-3,8,10-11"
    assert ddist.slice_cpus(list(range(64)), 1, 4) == list(range(16, 32))
    assert ddist.slice_cpus(list(range(10)), 3, 4) == [6, 7, 8, 9]          # the last share takes the remainder
    assert ddist.slice_cpus([0, 1], 5, 8) != []                             # never empty
    # two sockets x 4 GPUs: devices 0-3 local to CPUs 0-31, 4-7 to 32-63; rank r runs on device r
    bound = {}
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(64)))
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: bound.__setitem__("cpus", sorted(cpus)))
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    look = lambda d: list(range(0, 32)) if d < 4 else list(range(32, 64))
    rec = ddist.bind_rank_to_local_cpus(5, local_rank=5, local_world=8, local_cpus_of=look)
    assert rec["bound"] and rec["cpus"] == "40-47" and rec["share"] == "2/4" and bound["cpus"] == list(range(40, 48))
    rec = ddist.bind_rank_to_local_cpus(0, local_rank=0, local_world=8, local_cpus_of=look)
    assert rec["cpus"] == "0-7"
    monkeypatch.setenv("DD_DIST_NO_AFFINITY", "1")
    assert ddist.bind_rank_to_local_cpus(0, 0, 8, look) == {"bound": False, "why": "DD_DIST_NO_AFFINITY=1"}
    monkeypatch.delenv("DD_DIST_NO_AFFINITY")
    assert ddist.bind_rank_to_local_cpus(0, 0, 8, lambda d: None)["bound"] is False


def test_plan_only_describes_the_eight_rank_job_without_a_gpu():
    """bench.py --gpus 8 --plan-only: units per rank, planned cost / imbalance, device and CPU slice per rank -- a function of
    the arguments only (every rank computes the same table), no GPU needed."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--config", "3", "--plan-only"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    plan = json.loads(p.stdout.strip().splitlines()[-1])["plan"]
    assert plan["world"] == 8 and plan["n_units"] == 100 and plan["scaling"] == "strong"
    assert sorted(u for r in plan["ranks"] for u in r["units"]) == list(range(100))          # every pocket exactly once
    assert plan["planned_imbalance_max_over_mean"] < 1.03
    units, _ = ddist.plan_job(3, 8)
    assert [r["units"] for r in plan["ranks"]] == [[u.uid for u in ddist.units_of_rank(units, 3, r, 8)] for r in range(8)]
    # affinity plan with an injected topology: two NUMA nodes of 4 GPUs each, 128 CPUs per node -> 32 CPUs per rank, disjoint
    topo = lambda d: list(range(0, 128)) if d < 4 else list(range(128, 256))
    allowed = sorted(os.sched_getaffinity(0))
    plan = ddist.describe_plan(1, 8, local_cpus_of=topo, n_devices=8)
    if len(allowed) >= 256:
        cpus = [ddist.parse_cpulist(r["cpu_affinity"]["cpus"]) for r in plan["ranks"]]
        assert all(len(c) == 32 for c in cpus) and len({x for c in cpus for x in c}) == 256
    assert sorted(os.sched_getaffinity(0)) == allowed                                            # a plan binds nothing
    assert all(r["cpu_affinity"].get("bound") is False for r in plan["ranks"])
    cfg4 = ddist.describe_plan(4, 8, num_samples=64)
    assert [r["n_units"] for r in cfg4["ranks"]] == [1] * 8 and cfg4["planned_imbalance_max_over_mean"] == 1.0
