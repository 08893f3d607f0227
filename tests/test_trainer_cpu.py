"""CPU: host logic of the data-parallel training loop (voicesplit_amd/trainer.py)
with a stand-in model and criterion -- index sharding, the step against a hand-written Adam step,
the reference's checkpoint format and partial initialisation, the on-disk dataset reader, and the
world-2 job over gloo (same weights as one process on the global batch, one loss value on every
rank, both ranks stopping together on a non-finite loss, rank 0 alone writing the checkpoint).
The HIP model itself is covered by the -m gpu tests."""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from voicesplit_amd import AttrDict
from voicesplit_amd.trainer import (EpochShard, LossExploded, SpecWavDataset, Trainer, load_wav, make_criterion,
                                    make_optimizer)


def _cfg(**train):
    c = AttrDict()
    tc = {"epochs": 2, "learning_rate": 1e-2, "optimizer": "adam", "batch_size": 2, "seed": 42,
          "summary_interval": 1, "checkpoint_interval": 3, "reinit_layers": None}
    tc.update(train)
    c.update({"model_name": "voicesplit", "train_config": AttrDict(tc),
              "loss": {"loss_name": "si_snr", "power": 0.3, "complex_loss_ratio": 0.113},
              "audio": {"backend": "voicefilter", "voicefilter": {"n_fft": 1200, "hop_length": 160, "win_length": 400,
                                                                "num_freq": 601, "sample_rate": 16000}}})
    return c


class _Standin(torch.nn.Module):
    """forward(mixed [B,T,F], emb [B,E]) -> mask [B,T,F] in (0,1); no BatchNorm, so the global-batch
    gradient is the mean of the shard gradients."""

    def __init__(self, F=5, E=3):
        super().__init__()
        torch.manual_seed(5)
        self.a = torch.nn.Linear(F + E, 7)
        self.b = torch.nn.Linear(7, F)

    def forward(self, mixed, emb):
        x = torch.cat([mixed, emb[:, None, :].expand(-1, mixed.shape[1], -1)], dim=2)
        return torch.sigmoid(self.b(torch.tanh(self.a(x))))


def _mse(mask, mixed, target, seq_len, phase):
    return ((mixed * mask - target) ** 2).mean()


def _batch(B, seed, T=4, F=5, E=3):
    g = torch.Generator().manual_seed(seed)
    mixed = torch.rand(B, T, F, generator=g)
    return (torch.randn(B, E, generator=g), mixed * torch.rand(B, T, F, generator=g), mixed,
            torch.full((B, 1), 10), None, torch.zeros(B, T, F))


def test_epoch_shard_is_a_partition_of_a_drop_last_shuffle():
    n, b, world = 23, 2, 4
    shards = [EpochShard(n, b, r, world, seed=9) for r in range(world)]
    assert all(s.steps_per_epoch() == 2 for s in shards)             # 23 // 8, the tail of 7 dropped
    for e in (0, 1):
        per_rank = [list(s.epoch(e)) for s in shards]
        assert all(len(x) == 2 and all(len(i) == b for i in x) for x in per_rank)
        seen = [i for x in per_rank for idx in x for i in idx]
        assert len(seen) == len(set(seen)) == 16 and set(seen) <= set(range(n))
        # k-th global batch = the k-th 8 items of ONE permutation shared by all ranks
        order = [i for k in range(2) for x in per_rank for i in x[k]]
        assert order == [i for k in range(2) for r in range(world) for i in list(shards[r].epoch(e))[k]]
    assert list(shards[0].epoch(0)) == list(EpochShard(n, b, 0, world, seed=9).epoch(0))
    assert list(shards[0].epoch(0)) != list(shards[0].epoch(1))
    assert list(EpochShard(6, 2, 1, 2, shuffle=False).epoch(0)) == [[2, 3]]
    assert list(EpochShard(3, 2, 0, 2).epoch(0)) == []               # fewer items than one global batch
    with pytest.raises(ValueError):
        EpochShard(4, 2, 2, 2)
    with pytest.raises(ValueError):
        EpochShard(4, 0)


def test_criterion_and_optimizer_selection_follow_train_py():
    c = _cfg()
    c.loss["loss_name"] = "l1"
    with pytest.raises(Exception, match="not suported"):
        make_criterion(c)
    c.train_config["optimizer"] = "sgd"
    with pytest.raises(Exception, match="optimizer supported"):
        make_optimizer(c, _Standin().parameters())
    assert isinstance(make_optimizer(_cfg(), _Standin().parameters()), torch.optim.Adam)
    assert callable(make_criterion(_cfg()))                           # 'si_snr' resolves without touching the GPU
    c2 = _cfg()
    c2.loss["loss_name"] = "power_law_compression"
    assert callable(make_criterion(c2))


def test_train_step_is_zero_grad_backward_adam_step():
    c = _cfg()
    tr = Trainer(_Standin(), c, criterion=_mse)
    ref = _Standin()
    opt = torch.optim.Adam(ref.parameters(), lr=c.train_config["learning_rate"])
    for s in range(3):
        batch = _batch(4, s)
        loss = tr.train_step(batch)
        opt.zero_grad()
        rl = _mse(ref(batch[2], batch[0]), batch[2], batch[1], None, None)
        rl.backward()
        opt.step()
        assert abs(loss - rl.item()) < 1e-7
    assert tr.step == 3
    for p, q in zip(tr.model.parameters(), ref.parameters()):
        assert torch.allclose(p, q, rtol=1e-6, atol=1e-8)
    # gradients still live in the flat bucket (no per-step re-allocation)
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(tr.bucket.params, tr.bucket.views))
    # bench.py's N > 1 self-diagnosis switch: harmless at world 1 / on the CPU (no collective, no events), and it resets its records
    tr.set_comm_timing(True)
    tr.train_step(_batch(4, 7))
    assert tr.time_comm and tr.bucket.time_events and tr.flag_ms == [] and tr.bucket.collective_ms() == []
    tr.set_comm_timing(False)
    assert not tr.time_comm and not tr.bucket.time_events


def test_loss_explosion_follows_train_py_order_and_fit_leaves_the_epoch():
    c = _cfg(epochs=2)
    calls = {"n": 0}

    def crit(mask, mixed, target, seq_len, phase):
        calls["n"] += 1
        return _mse(mask, mixed, target, seq_len, phase) * (float("nan") if calls["n"] == 2 else 1.0)

    tr = Trainer(_Standin(), c, criterion=crit)
    tr.train_step(_batch(2, 0))
    # train.py:109-117: optimizer.step(); step += 1; THEN loss.item() and the guard -- the step is counted
    with pytest.raises(LossExploded, match="Loss exploded to nan at step 2!"):
        tr.train_step(_batch(2, 1))
    assert tr.step == 2
    # fit(): the exploding step ends its epoch (train.py:115-117 `break`), the next epoch still runs
    calls["n"] = 0
    tr2 = Trainer(_Standin(), c, criterion=crit)
    logged = []
    tr2.fit(lambda e: (_batch(2, 10 * e + i) for i in range(3)), on_log=lambda s, l: logged.append(s))
    # as in the reference the NaN update has gone into the weights, so the next epoch explodes at once too
    assert tr2.step == 3 and logged == [1]


def test_checkpoint_format_resume_and_partial_init(tmp_path):
    c = _cfg()
    tr = Trainer(_Standin(), c, criterion=_mse)
    for s in range(2):
        tr.train_step(_batch(2, s))
    path = tr.save_checkpoint(str(tmp_path / "checkpoint_2.pt"))
    ck = torch.load(path, map_location="cpu")
    assert set(ck) == {"model", "optimizer", "step", "config_str"} and ck["step"] == 2      # train.py:127-132
    assert ck["config_str"] == str(c) and set(ck["model"]) == set(tr.model.state_dict())
    # resume: same weights, same Adam moments -> the next step is identical
    tr_b = Trainer(_Standin(), c, criterion=_mse)
    with torch.no_grad():
        for p in tr_b.model.parameters():
            p.add_(1.0)
    assert tr_b.load_checkpoint(path) == 2
    la, lb = tr.train_step(_batch(2, 7)), tr_b.train_step(_batch(2, 7))
    assert la == lb and all(torch.equal(p, q) for p, q in zip(tr.model.parameters(), tr_b.model.parameters()))
    # partial initialisation (utils/generic_utils.py:647-676): a layer of another size is skipped,
    # reinit_layers keeps fresh values, everything else is taken
    other = _Standin(F=5, E=3)
    other.b = torch.nn.Linear(7, 9)
    tr_c = Trainer(other, c, criterion=lambda m, x, t, s, p: m.mean())
    b_before = other.b.weight.detach().clone()
    tr_c.load_checkpoint(path)
    assert torch.equal(other.a.weight, ck["model"]["a.weight"]) and torch.equal(other.b.weight, b_before)
    tr_d = Trainer(_Standin(), c, criterion=_mse)
    a_before = tr_d.model.a.weight.detach().clone()
    tr_d.load_checkpoint(path, reinit_layers=["a."])
    assert torch.equal(tr_d.model.a.weight, a_before) and torch.equal(tr_d.model.b.weight, ck["model"]["b.weight"])


def test_spec_wav_dataset_reads_the_reference_layout(tmp_path):
    from scipy.io import wavfile
    c = _cfg()
    c.dataset = {"train_dir": str(tmp_path), "test_dir": str(tmp_path),
                 "format": {"emb": "*-emb.pt", "mixed": "*-mixed.pt", "target": "*-target.pt",
                            "target_wav": "*-target.wav", "mixed_wav": "*-mixed.wav"}}
    rng = np.random.default_rng(0)
    for i in range(3):
        stem = str(tmp_path / ("%06d" % i))
        torch.save(torch.randn(256), stem + "-emb.pt")
        torch.save(torch.rand(301, 601), stem + "-target.pt")
        wavfile.write(stem + "-mixed.wav", 16000, (rng.standard_normal(48000) * 0.1).astype(np.float32))
        wavfile.write(stem + "-target.wav", 16000, (rng.standard_normal(48000) * 3000).astype(np.int16))
    ds = SpecWavDataset(c)
    assert len(ds) == 3
    emb, target, mixed_wav, seq_len, target_wav = ds[1]
    assert emb.shape == (256,) and target.shape == (301, 601) and seq_len.tolist() == [48000]
    assert mixed_wav.dtype == torch.float32 and mixed_wav.shape == (48000,)
    assert target_wav.dtype == torch.float32 and target_wav.abs().max() < 1.0        # int16 -> [-1, 1)
    with pytest.raises(ValueError, match="sample rate"):
        load_wav(str(tmp_path / "000000-mixed.wav"), 8000)
    os.remove(str(tmp_path / "000002-emb.pt"))
    with pytest.raises(ValueError, match="not Match"):
        SpecWavDataset(c)
    c.dataset["train_dir"] = str(tmp_path / "nope")
    with pytest.raises(FileNotFoundError):
        SpecWavDataset(c)


def _write_dataset(tmp_path, n, seed=0, samples=4800, T=31, Fq=601):
    from scipy.io import wavfile
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    for i in range(n):
        stem = str(tmp_path / ("%06d" % i))
        torch.save(torch.randn(256, generator=g) if i != 3 else torch.tensor([0]), stem + "-emb.pt")      # item 3: "no embedding" (dataset.py:93-95)
        torch.save(torch.rand(T, Fq, generator=g), stem + "-target.pt")
        wavfile.write(stem + "-mixed.wav", 16000, (rng.standard_normal(samples) * 0.1).astype(np.float32))
        wavfile.write(stem + "-target.wav", 16000, (rng.standard_normal(samples) * 0.1).astype(np.float32))
    c = _cfg(num_workers=2)
    c.dataset = {"train_dir": str(tmp_path), "test_dir": str(tmp_path),
                 "format": {"emb": "*-emb.pt", "mixed": "*-mixed.pt", "target": "*-target.pt",
                            "target_wav": "*-target.wav", "mixed_wav": "*-mixed.wav"}}
    return c


def test_batch_feeder_worker_processes_yield_exactly_the_epoch_shards_batches(tmp_path):
    """BatchFeeder.host_batches: a DataLoader with worker processes whose batch sampler is the rank's EpochShard -- the same items
    in the same order as the synchronous ``[ds[i] for i in idx]`` loop it replaces, filtered items dropped, a batch with nothing left
    reported as None; the same for both ranks of a world-2 sharding."""
    from voicesplit_amd.trainer import BatchFeeder, host_collate
    c = _write_dataset(tmp_path, 13)
    ds = SpecWavDataset(c)
    for rank in (0, 1):
        shard = EpochShard(len(ds), 2, rank, 2, seed=4)
        feeder = BatchFeeder(ds, shard, "cpu")
        assert feeder.num_workers == 2
        for e in (0, 1):
            got = list(feeder.host_batches(e))
            want = [host_collate([ds[i] for i in idx]) for idx in shard.epoch(e)]
            assert len(got) == len(want) == 3
            for g_, w_, idx in zip(got, want, shard.epoch(e)):
                assert (g_ is None) == (w_ is None)
                if g_ is None:
                    continue
                assert len(g_[0]) == len([i for i in idx if i != 3])
                assert all(torch.equal(a, b) for a, b in zip(g_, w_))
    # a batch made of the filtered item alone
    one = BatchFeeder(ds, EpochShard(len(ds), 1, 0, 1, shuffle=False), "cpu", num_workers=0)
    hb = list(one.host_batches(0))
    assert hb[3] is None and sum(b is None for b in hb) == 1 and len(hb) == 13


def test_gradient_bucket_layout_spare_slots_in_front_and_an_early_segment():
    """flat = [extra | late parameters | early parameters], every view on a 16-byte boundary (zero padding between them): the views do
    not overlap, .grad is the view, the early segment is contiguous at the end and the spare slots sit in front of the small segment
    (sharding.GradientBucket)."""
    from voicesplit_amd.sharding import GradientBucket
    m = _Standin()
    params = list(m.parameters())                     # a.weight, a.bias, b.weight, b.bias
    early = [m.b.weight, m.b.bias]
    bk = GradientBucket(params, extra=2, early=early).attach()
    up = lambda n: (n + 3) // 4 * 4
    assert bk.has_early and bk.extra.data_ptr() == bk.flat.data_ptr() and bk.extra.numel() == 2
    off = {id(p): (v.data_ptr() - bk.flat.data_ptr()) // 4 for p, v in zip(bk.params, bk.views)}
    assert all(o % 4 == 0 for o in off.values())                                        # 16-byte boundaries
    assert off[id(m.a.weight)] == 4 and off[id(m.a.bias)] == up(4 + m.a.weight.numel())
    assert bk.split == up(off[id(m.a.bias)] + m.a.bias.numel())
    assert off[id(m.b.weight)] == bk.split and off[id(m.b.bias)] == up(bk.split + m.b.weight.numel())
    assert bk.flat.numel() == up(off[id(m.b.bias)] + m.b.bias.numel()) and bk.grads.data_ptr() == bk.flat.data_ptr() + 16
    assert all(p.grad is v and v.shape == p.shape for p, v in zip(bk.params, bk.views))
    for i, v in enumerate(bk.views):
        v.fill_(float(i + 1))
    order = {id(p): i for i, p in enumerate(params)}           # (list.index would compare tensors element-wise)
    want = torch.zeros(bk.flat.numel())
    for p in params:
        want[off[id(p)]:off[id(p)] + p.numel()] = float(order[id(p)] + 1)
    assert torch.equal(bk.flat, want)                          # the views cover their ranges once; spare slots and padding stay zero
    # no early segment: the old layout but for the spare slots in front
    bk2 = GradientBucket(params, extra=1)
    assert not bk2.has_early and bk2.split == bk2.flat.numel() and bk2.all_reduce_early(1) is None
    assert bk.all_frozen_free()
    m.a.bias.requires_grad_(False)
    assert not bk.all_frozen_free()


def test_fit_on_one_rank_is_the_plain_loop():
    """world 1: no collective, no look-ahead -- fit() feeds train_step batch by batch, an empty batch is skipped locally"""
    c = _cfg(epochs=1)
    tr = Trainer(_Standin(), c, criterion=_mse)
    ref = Trainer(_Standin(), c, criterion=_mse)
    seq = [_batch(2, 0), (None,) * 6, _batch(2, 1), _batch(2, 2)]
    logged = []
    tr.fit(lambda e: iter(seq), on_log=lambda s, l: logged.append((s, l)))
    want = [ref.train_step(b) for b in (seq[0], seq[2], seq[3])]
    assert tr.step == 3 and [l for _, l in logged] == want
    assert all(torch.equal(p, q) for p, q in zip(tr.model.parameters(), ref.model.parameters()))


# ---- world 2 over gloo ------------------------------------------------------------------------------

def _dp_worker(rank, world, port, tmp, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = _cfg()
    model = _Standin()
    if rank == 1:                       # a replica that starts elsewhere must be overwritten by rank 0's weights
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(0.5)
    tr = Trainer(model, c, rank, world, criterion=_mse)
    b, n = 2, 12
    shard = EpochShard(n, b, rank, world, seed=1)
    data = _batch(n, 123)
    pick = lambda idx: tuple(None if t is None else t[idx] for t in data)
    losses = [tr.train_step(pick(idx)) for idx in shard.epoch(0)]
    # one process on the global batches
    ref = Trainer(_Standin(), c, criterion=_mse)
    full = EpochShard(n, b * world, 0, 1, seed=1)
    ref_losses = [ref.train_step(pick(idx)) for idx in full.epoch(0)]
    ok = len(losses) == 3 and all(abs(a - r) < 1e-6 for a, r in zip(losses, ref_losses))
    ok = ok and all(torch.allclose(p, r, rtol=1e-5, atol=1e-7) for p, r in zip(tr.model.parameters(), ref.model.parameters()))
    # every rank saw the same loss values and holds the same weights
    w = torch.cat([p.detach().reshape(-1) for p in tr.model.parameters()] + [torch.tensor(losses)])
    ws = [torch.empty_like(w) for _ in range(world)]
    dist.all_gather(ws, w)
    ok = ok and torch.equal(ws[0], ws[1])
    # rank 0 alone writes the checkpoint
    path = os.path.join(tmp, "checkpoint_3.pt")
    out = tr.save_checkpoint(path)
    dist.barrier()
    ok = ok and ((rank == 0) == (out is not None)) and os.path.isfile(path)
    # one rank's slice filtered out entirely (utils/dataset.py:93-95): both ranks skip the step together
    from voicesplit_amd.trainer import EmptyBatch
    before = [p.detach().clone() for p in tr.model.parameters()]
    try:
        tr.train_step((None,) * 6 if rank == 1 else pick([0, 1]))
        ok = False
    except EmptyBatch:
        ok = ok and all(torch.equal(a, b) for a, b in zip(before, tr.model.parameters())) and tr.step == 3
    # a NaN on ONE rank stops both at the same step (counted, as train.py:111-117 does)
    tr.criterion = lambda m, x, t, s, p: _mse(m, x, t, s, p) * (float("nan") if rank == 1 else 1.0)
    try:
        tr.train_step(pick([0, 1]))
        ok = False
    except LossExploded:
        ok = ok and tr.step == 4
    with torch.no_grad():      # the NaN update went into the weights (as in the reference): restore for the rest
        for p_, b_ in zip(tr.model.parameters(), before):
            p_.copy_(b_)
    # ---- round 6: fit() carries the skip decision in the gradient bucket (no per-step flag collective) ---------------------------
    # five global batches; rank 1's slice of batch 2 was filtered out entirely: both ranks skip it together, and the job equals one
    # process that never saw that global batch.  Blocking flag reduces: one at the epoch's start + one at the skipped step.
    tr2 = Trainer(_Standin(), c, rank, world, criterion=_mse)
    shard5 = EpochShard(20, b, rank, world, seed=3)
    data5 = _batch(20, 321)
    pick5 = lambda idx: tuple(None if t is None else t[idx] for t in data5)
    mine = [pick5(idx) for idx in shard5.epoch(0)]
    if rank == 1:
        mine[2] = (None,) * 6
    tr2.set_comm_timing(True)
    seen = []
    tr2.fit(lambda e: iter(mine), epochs=1, on_log=lambda s, l: seen.append((s, l)))
    ref2 = Trainer(_Standin(), c, criterion=_mse)
    full5 = [pick5(idx) for idx in EpochShard(20, b * world, 0, 1, seed=3).epoch(0)]
    want5 = [ref2.train_step(full5[k]) for k in (0, 1, 3, 4)]
    ok = ok and tr2.step == 4 and len(tr2.flag_ms) == 2
    ok = ok and all(torch.allclose(p, r, rtol=1e-5, atol=1e-7) for p, r in zip(tr2.model.parameters(), ref2.model.parameters()))
    if rank == 0:
        ok = ok and len(seen) == 4 and all(abs(l - w_) < 1e-6 for (_, l), w_ in zip(seen, want5))
    # loss_lag = 1: the value returned is the step before's (the guard fires one step late), the weights are the same
    tr3 = Trainer(_Standin(), c, rank, world, criterion=_mse, loss_lag=1)
    lag = []
    tr3.fit(lambda e: iter(mine), epochs=1, on_log=lambda s, l: lag.append(l))
    ok = ok and all(torch.equal(p, r) for p, r in zip(tr3.model.parameters(), tr2.model.parameters()))
    if rank == 0:
        ok = ok and len(lag) == 4 and abs(lag[0] - want5[0]) < 1e-6 and all(abs(lag[k] - want5[k - 1]) < 1e-6 for k in (1, 2, 3))
    # the early segment: summed on its own + the rest == one all-reduce of the whole bucket
    from voicesplit_amd.sharding import GradientBucket
    m4 = _Standin()
    bk = GradientBucket(list(m4.parameters()), None, extra=2, early=[m4.b.weight, m4.b.bias]).attach()
    g4 = torch.Generator().manual_seed(50 + rank)
    bk.flat.copy_(torch.randn(bk.flat.numel(), generator=g4))
    mine4 = bk.flat.clone()
    both = [torch.empty_like(mine4) for _ in range(world)]
    dist.all_gather(both, mine4)
    work = bk.all_reduce_early(world)
    ok = ok and work is not None
    work.wait()
    ok = ok and torch.equal(bk.flat[:bk.split], mine4[:bk.split])           # the small segment is untouched so far
    bk.all_reduce(world, early_done=True)
    ok = ok and torch.allclose(bk.flat, (both[0] + both[1]) / world, rtol=0, atol=1e-6)
    # validation mean over ranks
    tr.criterion = _mse
    v = tr.validate([pick([2 * rank, 2 * rank + 1])])
    with torch.no_grad():
        m = tr.model.eval()
        want = sum(_mse(m(data[2][[2 * r, 2 * r + 1]], data[0][[2 * r, 2 * r + 1]]), data[2][[2 * r, 2 * r + 1]],
                        data[1][[2 * r, 2 * r + 1]], None, None).item() for r in range(world)) / world
    ok = ok and abs(v - want) < 1e-7 and not math.isnan(v)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_training_job_equals_one_process_on_the_global_batch(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_cli_refuses_to_run_without_a_gpu(tmp_path):
    """The training CLI has no CPU path: on a box without an MI355X it must say so, not fall back."""
    import json
    from voicesplit_amd import default_config, trainer
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    c = default_config()
    p = tmp_path / "config.json"
    p.write_text(json.dumps({k: (dict(v) if isinstance(v, dict) else v) for k, v in c.items()}))
    with pytest.raises(RuntimeError, match="needs a GPU"):
        trainer.main(["-c", str(p), "--synthetic-steps", "1"])
