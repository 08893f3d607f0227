"""CPU: the multi-GPU host logic (batch sharding, ragged slices, gather) with world_size 2 over
gloo.  The compute function is a stand-in (the HIP path needs a GPU); what is tested is that
sharded == unsharded for any split, which is what bench.py --gpus N relies on."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from voicesplit_amd.sharding import GradientBucket, chunk_windows, run_sharded, shard_range, sync_buffers


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
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


# ---- data-parallel training exchange step ------------------------------------------------------

def _toy():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Tanh(), torch.nn.Linear(5, 3))


def _dp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8, 6, generator=g)
    y = torch.randn(8, 3, generator=g)
    m = _toy().eval()                      # frozen BN: the global-batch gradient is the mean of the shard gradients
    bucket = GradientBucket(m.parameters()).attach()
    lo, hi = shard_range(8, rank, world)
    loss = ((m(x[lo:hi]) - y[lo:hi]) ** 2).mean()
    loss.backward()
    bucket.all_reduce(world)
    flat = torch.cat([v.reshape(-1) for v in bucket.views])      # (the views start on 16-byte boundaries: flat itself holds padding between them)
    ref = _toy().eval()
    ((ref(x) - y) ** 2).mean().backward()
    ref_flat = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    ok = torch.allclose(flat, ref_flat, rtol=1e-5, atol=1e-7)
    ok = ok and all(p.grad.data_ptr() == v.data_ptr() and v.data_ptr() % 16 == 0 for p, v in zip(bucket.params, bucket.views))
    # identical optimizer step on every rank -> identical weights
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    opt.step()
    w = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    ws = [torch.empty_like(w) for _ in range(world)]
    dist.all_gather(ws, w)
    ok = ok and all(torch.equal(ws[0], t) for t in ws)
    # running statistics follow rank 0
    m[1].running_mean.fill_(float(rank + 1))
    sync_buffers(m)
    ok = ok and float(m[1].running_mean[0]) == 1.0
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_gradient_bucket_allreduce():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_single_rank_group_under_the_launcher_and_alone(tmp_path):
    """sharding.init_single_rank_group (bench.py's one-rank RCCL leg): as a plain process it builds a private tcp://
    store; started by torch.distributed.run it must take the launcher's store (env://) -- a private store there is a
    client without a server and waits for the collective timeout (found on the device: `torch.distributed.run
    --nproc-per-node 1 bench.py --gpus 1` hung).  gloo stands in for RCCL; both must finish within seconds."""
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "one_rank.py"
    script.write_text(
        "import sys, time\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import torch, torch.distributed as dist\n"
        "from voicesplit_amd.sharding import init_single_rank_group\n"
        "t0 = time.time()\n"
        "how = init_single_rank_group('gloo')\n"
        "x = torch.ones(4)\n"
        "dist.all_reduce(x)\n"
        "assert dist.get_world_size() == 1 and torch.equal(x, torch.ones(4))\n"
        "dist.destroy_process_group()\n"
        "print('HOW', how.split(':')[0], round(time.time() - t0, 1))\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    alone = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert alone.returncode == 0 and "HOW tcp" in alone.stdout, alone.stdout + alone.stderr
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    launched = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                               "--master-port", str(port), str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert launched.returncode == 0 and "HOW env" in launched.stdout, launched.stdout + launched.stderr


def test_bucket_reattaches_a_grad_the_sink_wrote_and_zeroes_one_nobody_wrote():
    """all_reduce(written=True) (the trainer's gradient sink wrote the views itself): a .grad set to None in between is
    re-attached with its contents intact; without `written` the old rule holds -- no gradient arrived, the slot is zeroed
    and .grad stays None so that the optimizer skips the parameter."""
    from voicesplit_amd.sharding import GradientBucket
    lin = torch.nn.Linear(5, 3)
    b = GradientBucket(lin.parameters()).attach()
    b.views[0].fill_(2.0)
    b.views[1].fill_(3.0)
    lin.weight.grad = None                      # zero_grad(set_to_none=True) by some outer loop
    b.all_reduce(1, written=True)
    assert lin.weight.grad is not None and lin.weight.grad.data_ptr() == b.views[0].data_ptr()
    assert float(b.views[0].sum()) == 2.0 * 15 and float(b.views[1].sum()) == 9.0
    lin.weight.grad = None
    b.all_reduce(1)
    assert lin.weight.grad is None and float(b.views[0].abs().sum()) == 0.0 and float(b.views[1].sum()) == 9.0


def test_written_views_win_over_a_stale_grad_and_frozen_parameters_stay_put():
    """a plain zero_grad(set_to_none=True) / backward loop over the same model leaves an autograd-made
    .grad that is NOT a bucket view; on the next sink step the library writes fresh gradients into the views and autograd
    gets nothing for those keys -- the stale tensor must not be copied over the fresh view.  And (low): a parameter frozen
    after the bucket was built is not re-attached (the optimizer keeps skipping it) and its slot is zeroed."""
    from voicesplit_amd.sharding import GradientBucket
    lin = torch.nn.Linear(5, 3)
    b = GradientBucket(lin.parameters()).attach()
    lin.weight.grad = torch.full_like(lin.weight, 7.0)      # what a plain backward() after set_to_none leaves behind
    b.views[0].fill_(2.0)                                   # the sink's fresh gradient
    b.views[1].fill_(3.0)
    b.all_reduce(1, written=True)
    assert lin.weight.grad.data_ptr() == b.views[0].data_ptr() and float(lin.weight.grad.sum()) == 2.0 * 15
    # without `written` the old contract holds: a foreign .grad was accumulated by autograd and is copied in
    lin.weight.grad = torch.full_like(lin.weight, 7.0)
    b.all_reduce(1)
    assert lin.weight.grad.data_ptr() == b.views[0].data_ptr() and float(b.views[0].sum()) == 7.0 * 15
    # frozen after construction
    lin.bias.requires_grad_(False)
    lin.bias.grad = None
    b.views[1].fill_(5.0)                                   # the library writes all gradients into the sink
    b.all_reduce(1, written=True)
    assert lin.bias.grad is None and float(b.views[1].abs().sum()) == 0.0
