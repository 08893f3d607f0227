"""CPU: the multi-GPU host logic (batch sharding, ragged slices, gather) with world_size 2 over
gloo.  The compute function is a stand-in (the HIP path needs a GPU); what is tested is that
sharded == unsharded for any split, which is what bench.py --gpus N relies on."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from voicesplit_amd.sharding import chunk_windows, run_sharded, shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 2, 5, 63, 64, 65, 512):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a0, b0), (a1, b1) in zip(spans, spans[1:]):
                assert b0 == a1
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)
    assert chunk_windows(3001) == 10 and chunk_windows(301) == 1 and chunk_windows(1) == 1


def _fake_mask(x, dvec):
    # per-utterance, purely elementwise (bit-reproducible for any batch split): a slice that
    # lands on the wrong utterance, or a leak across utterances, changes the result
    return x * dvec[:, 0][:, None, None] + dvec[:, 1][:, None, None]


def _worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, 7, 5, generator=g)
    d = torch.randn(B, 4, generator=g)
    full = run_sharded(_fake_mask, x, d, rank, world, gather=True)
    local = run_sharded(_fake_mask, x, d, rank, world, gather=False)
    ref = _fake_mask(x, d)
    lo, hi = shard_range(B, rank, world)
    ok = torch.equal(full, ref) and ((local is None and hi == lo) or torch.equal(local, ref[lo:hi]))
    # the bench's timing reduction: MAX over ranks
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and t.item() == world
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [1, 5, 8])
def test_world2_gloo_sharded_equals_unsharded(B):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]
