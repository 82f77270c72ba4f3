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
