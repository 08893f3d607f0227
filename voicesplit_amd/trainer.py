"""Data-parallel training loop around the mask path: what train.py:25-135 does
with the model, one process per GPU.

The reference trains on one GPU (run_train.sh:1) with ``DataLoader(shuffle=True, drop_last=True)``
(utils/dataset.py:60-68), Adam (train.py:33-35), one of two criteria (train.py:74-79) and the step

    mask = model(mixed, emb); output = mixed * mask; loss = criterion(...)      train.py:94-108
    optimizer.zero_grad(); loss.backward(); optimizer.step(); step += 1         train.py:109-112

This module keeps that loop and its checkpoint format (train.py:127-132: ``model``, ``optimizer``,
``step``, ``config_str`` -- loadable by the reference's test.py) and adds the N-GPU part:

* ``EpochShard``: every rank walks the same seeded permutation of the dataset and takes its slice
  of every global batch (world * batch_size items, incomplete global batches dropped as
  ``drop_last=True`` does), so all ranks run the same number of steps and no collective can hang;
* ``Trainer.train_step``: the step above with ONE collective -- the flat gradient bucket of
  ``sharding.GradientBucket`` (RCCL sum over xGMI, divided by the world size), which also carries
  the loss value in a spare slot so that logging and the loss-explosion guard (train.py:115-117) see
  the same number on every rank and stop together;
* BatchNorm statistics stay per replica as in the reference (no SyncBN); rank 0's running
  statistics are broadcast when a checkpoint is cut and rank 0 alone writes it.

Everything here is host logic over ``nn.Module`` / ``torch.distributed``: it runs under gloo with
a CPU stand-in model in tests/test_trainer_cpu.py; with ``VoiceSplit`` / ``VoiceFilter`` and the
criteria of ``losses.py`` every flop of the step is inside libvoicesplit_hip.so.
"""
import math
import os
import time
from glob import glob
from typing import Callable, Iterable, Iterator, List, Optional, Sequence

import torch

from .sharding import GradientBucket, sync_buffers


class LossExploded(RuntimeError):
    """train.py:115-117: ``loss > 1e8 or math.isnan(loss)`` ends the epoch."""


class EmptyBatch(RuntimeError):
    """Some rank's slice of this global batch was filtered out entirely (utils/dataset.py:93-95);
    raised on every rank at once, ``fit`` moves on to the next batch."""


# ---------------------------------------------------------------------------------------------
# which items a rank sees
# ---------------------------------------------------------------------------------------------
class EpochShard:
    """Index batches of one rank.  ``for idx in EpochShard(n, b, rank, world, seed).epoch(e)`` yields
    lists of ``b`` dataset indices; the union over ranks of the k-th lists is the k-th global batch
    of a ``shuffle=True, drop_last=True`` loader with batch size ``world * b``."""

    def __init__(self, n_items: int, batch_size: int, rank: int = 0, world: int = 1, seed: int = 0,
                 shuffle: bool = True):
        if world <= 0 or not (0 <= rank < world):
            raise ValueError(f"bad rank/world {rank}/{world}")
        if batch_size <= 0 or n_items < 0:
            raise ValueError("batch_size must be > 0 and n_items >= 0")
        self.n, self.b, self.rank, self.world, self.seed, self.shuffle = n_items, batch_size, rank, world, seed, shuffle

    def steps_per_epoch(self) -> int:
        return self.n // (self.b * self.world)

    def epoch(self, epoch: int) -> Iterator[List[int]]:
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed * 1000003 + epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        gb = self.b * self.world
        for k in range(self.steps_per_epoch()):
            lo = k * gb + self.rank * self.b
            yield order[lo:lo + self.b]


# ---------------------------------------------------------------------------------------------
# criteria of train.py:74-79
# ---------------------------------------------------------------------------------------------
def make_criterion(c) -> Callable:
    """``criterion(mask, mixed, target, seq_len, spec_phase) -> loss`` for ``c.loss['loss_name']``:
    'si_snr' (train.py:97-103 + SiSNR_With_Pit) or 'power_law_compression' (PowerLaw_Compressed_Loss,
    seq_len ignored as train.py:104-105 does).  Both are single calls into libvoicesplit_hip.so."""
    from . import losses
    name = c.loss["loss_name"]
    if name == "si_snr":
        audio_cfg = c.audio[c.audio["backend"]]
        return lambda mask, mixed, target, seq_len, phase: losses.sisnr_loss(mask, mixed, target, phase, seq_len, audio_cfg)
    if name == "power_law_compression":
        power, ratio = c.loss["power"], c.loss["complex_loss_ratio"]
        return lambda mask, mixed, target, seq_len, phase: losses.power_law_loss(mask, mixed, target, power, ratio)
    raise Exception(" The loss '" + name + "' is not suported")          # train.py:79


def make_optimizer(c, params):
    """train.py:33-37."""
    if c.train_config["optimizer"] == "adam":
        params = list(params)
        # same update rule and state_dict as the reference's torch.optim.Adam(lr=...); on the device the whole step is
        # one fused multi-tensor launch instead of 14 (0.4 -> 0.1 ms at the tail of every step)
        fused = len(params) > 0 and all(p.is_cuda for p in params)
        return torch.optim.Adam(params, lr=c.train_config["learning_rate"], fused=fused)
    raise Exception("The %s  not is a optimizer supported" % c.train_config["optimizer"])


# ---------------------------------------------------------------------------------------------
# the loop
# ---------------------------------------------------------------------------------------------
class _FlagOfStep:
    """A skip decision that is still on its way: slot of the host ring the step that carried the flag copied its extra slots to."""

    def __init__(self, slot: int):
        self.slot = slot


_END = object()


class Trainer:
    """One rank of the training job.  ``model``: VoiceSplit / VoiceFilter on this rank's GPU (any
    ``nn.Module`` with the same ``forward(mixed, emb) -> mask`` works: the CPU tests use one).

    The N > 1 step (round 6; SURVEY.md 8(e) asks for ONE exchange step and nothing else on the path):
      * no per-step flag collective: ``fit`` looks two batches ahead and lets "my batch k + 2 is empty" ride in the second spare
        slot of step k's gradient bucket; the verdict is read from pinned memory two steps later, when it is long complete
        (``train_step(batch)`` called directly keeps the blocking MIN-reduce: it cannot look ahead);
      * the rank-averaged loss is read through pinned memory + an event recorded right behind the all-reduce, not by an
        ``.item()`` behind ``optimizer.step()``; ``loss_lag = 1`` returns the value of the step BEFORE (the guard of train.py:115-117
        then fires one step late) and the host never waits for the device at all;
      * ``split_allreduce``: the gradients of the BiLSTM and the head (73 of 75.5 MB) are final ~5 ms into the backward pass
        (vs_grads.leaves_event): their all-reduce starts there, on a side stream beside the conv stack's backward, and the
        collective behind the backward moves the conv stack's 2.2 MB + the two spare slots only.
    ``force_collectives``: run exactly that code at world 1 over a one-rank group (tests, bench.py's N = 1 A/B of the path)."""

    def __init__(self, model: torch.nn.Module, c, rank: int = 0, world: int = 1, group=None,
                 criterion: Optional[Callable] = None, optimizer: Optional[torch.optim.Optimizer] = None,
                 force_collectives: bool = False, split_allreduce: Optional[bool] = None, loss_lag: int = 0):
        self.model, self.c, self.rank, self.world, self.group = model, c, rank, world, group
        self.criterion = criterion if criterion is not None else make_criterion(c)
        self.optimizer = optimizer if optimizer is not None else make_optimizer(c, model.parameters())
        self._force = bool(force_collectives)
        self._comm = world > 1 or self._force
        params = list(model.parameters())
        on_cuda = len(params) > 0 and params[0].is_cuda
        can_sink = hasattr(model, "set_gradient_sink") and on_cuda
        early = None
        if self._comm and can_sink and hasattr(model, "leaf_parameter_names") and (split_allreduce is None or split_allreduce):
            leaf = set(model.leaf_parameter_names())
            early = [p for n, p in model.named_parameters() if n in leaf and p.requires_grad]
        # two spare slots: [0] the loss value (every rank logs / guards the rank average), [1] "a rank has no batch two steps ahead"
        self.bucket = GradientBucket(params, group, extra=2, early=early).attach()
        # the HIP modules write their gradients straight into the bucket's .grad views (no `grad += new` per parameter,
        # no zeroing per step); any other module goes through autograd's accumulation into the zeroed bucket
        # (the sink is installed around train_step's own backward only: a plain zero_grad / backward / step loop over the
        # same model, before or after this trainer existed, gets its gradients through autograd as usual; parameters whose
        # requires_grad is switched on after construction are in neither the bucket nor the sink -- build a new Trainer)
        self._sink = can_sink and self.bucket.flat.is_cuda
        self._sink_map = None
        if self._sink:
            names = {id(p): n for n, p in model.named_parameters()}
            self._sink_map = {names[id(p)]: v for p, v in zip(self.bucket.params, self.bucket.views)}
        self.device = self.bucket.flat.device
        self.step = 0
        self.time_comm = False          # bench.py's N > 1 line: time the collectives of a step (set_comm_timing)
        self.flag_ms = []
        self.split_allreduce = bool(self._sink and self.bucket.has_early)
        self._comm_stream = self._leaves_event = None
        if self.split_allreduce:
            # high priority: its own hardware-queue pool.  With RCCL initialised the process has more normal-priority streams than the
            # device has hardware queues for them, and a stream that shares a queue with the caller's stream would see the leaves event --
            # and with it the early collective -- only behind the whole backward pass (the same effect cost the
            # library's side stream its overlap, 46.2 -> 49.5 ms per step, until it got a priority of its own)
            self._comm_stream = torch.cuda.Stream(self.device, priority=-1)
            self._leaves_event = torch.cuda.Event()
            self._leaves_event.record()                                     # the handle the library records is created by the first record
        # train.py:114 reads loss.item() behind optimizer.step(): a device-to-host read that drains the device -- 0.3 ms of idle
        # device per step until the next step's first launches arrive.  One rank on a GPU: the value is final as soon as the
        # criterion has run, so it is copied to pinned memory THEN (beside the backward pass) and read at the reference's point
        # without draining anything; the guard fires on the same value at the same place.  Several ranks read the rank-averaged
        # value, which exists only behind the gradient all-reduce: copied to pinned memory there, read through an event.
        cuda = self.device.type == "cuda"
        self.early_loss_read = not self._comm and cuda
        self._loss_host = torch.zeros(1, dtype=torch.float32).pin_memory() if self.early_loss_read else None
        self._loss_event = torch.cuda.Event() if self.early_loss_read else None
        self.loss_lag = int(loss_lag)
        if self.loss_lag not in (0, 1):
            raise ValueError("loss_lag must be 0 or 1")
        self._ring = self._ring_ev = None
        self._slot = 0                                                      # steps that copied their spare slots to the host ring
        self._last_value = None
        if self._comm:                                                      # (gloo on the CPU: plain tensors, nothing to wait for)
            self._ring = [torch.zeros(2, dtype=torch.float32).pin_memory() if cuda else torch.zeros(2) for _ in range(4)]
            self._ring_ev = [torch.cuda.Event() for _ in range(4)] if cuda else None
        if world > 1:           # every replica starts from rank 0's weights (train.py has one process)
            import torch.distributed as dist
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, src=0, group=group)

    def set_comm_timing(self, on: bool = True):
        """Self-diagnosis of the N > 1 step: HIP events around the gradient all-reduces (``bucket.collective_ms()``, ``early=True``
        for the BiLSTM + head segment) and the host wall time of the blocking EmptyBatch MIN-reduces (``flag_ms``: one per step when
        ``train_step`` is called without a decision, one per epoch under ``fit``)."""
        self.time_comm = bool(on)
        self.bucket.time_events = bool(on)
        self.flag_ms = []
        self.bucket._events = []
        self.bucket._early_events = []

    @staticmethod
    def _has(batch) -> bool:
        return batch is not None and batch is not _END and batch[0] is not None and len(batch[0]) > 0

    def _agree(self, haves: Sequence[bool]) -> List[bool]:
        """One blocking MIN-reduce: entry i is True when EVERY rank has its batch i."""
        if not self._comm:
            return [bool(h) for h in haves]
        import torch.distributed as dist
        flag = torch.tensor([1.0 if h else 0.0 for h in haves], device=self.device)
        t0 = time.perf_counter() if self.time_comm else 0.0
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        out = [v > 0 for v in flag.tolist()]
        if self.time_comm:                                                  # host wall time of the flag round trip (it ends in a
            self.flag_ms.append(1e3 * (time.perf_counter() - t0))           # device-to-host read: collective + drain)
        return out

    def _read_slot(self, slot: int, index: int) -> float:
        if self._ring is None:
            raise RuntimeError("no host ring on this trainer")
        if self._ring_ev is not None:
            self._ring_ev[slot % 4].synchronize()
        return float(self._ring[slot % 4][index])

    # -- checkpoints: train.py:38-60 (load), :125-133 (save) ---------------------------------------
    def load_checkpoint(self, path: str, reinit_layers: Optional[Sequence[str]] = None) -> int:
        ckpt = torch.load(path, map_location="cpu")
        try:
            if reinit_layers:
                raise RuntimeError
            self.model.load_state_dict(ckpt["model"])
        except RuntimeError:
            # partial initialisation, utils/generic_utils.py:647-676: same key, same numel; layers
            # named in reinit_layers keep their fresh values
            cur = self.model.state_dict()
            take = {k: v for k, v in ckpt["model"].items() if k in cur and v.numel() == cur[k].numel()}
            for name in reinit_layers or ():
                take = {k: v for k, v in take.items() if name not in k}
            cur.update({k: v.reshape(cur[k].shape) for k, v in take.items()})
            self.model.load_state_dict(cur)
        try:
            self.optimizer.load_state_dict(ckpt["optimizer"])
        except (ValueError, KeyError):
            pass                                  # train.py:57-58: optimizer state is optional
        self.step = int(ckpt["step"])
        # load_state_dict copies into the existing tensors, so .grad views into the bucket survive
        return self.step

    def save_checkpoint(self, path: str) -> Optional[str]:
        """All ranks call this (it contains a broadcast); rank 0 writes the file."""
        if self.world > 1:
            sync_buffers(self.model, 0, self.group)
        if self.rank != 0:
            return None
        torch.save({"model": self.model.state_dict(), "optimizer": self.optimizer.state_dict(),
                    "step": self.step, "config_str": str(self.c)}, path)
        return path

    # -- one step: train.py:86-117 ---------------------------------------------------------------------
    def train_step(self, batch, have: Optional[bool] = None, next_missing: float = 0.0) -> float:
        """batch = (emb, target, mixed, seq_len, target_wav, spec_phase) as train_collate_fn returns
        it (utils/dataset.py:84-114).  Returns the loss averaged over ranks; raises LossExploded on
        every rank at once.
        have: None = decide here whether every rank has a batch (one blocking MIN-reduce per step at N > 1); True / False = the
        decision the ranks already share (``fit`` carries it in the gradient bucket two steps ahead).  next_missing: this rank's
        1.0 / 0.0 for "I have no batch two steps from now", summed over ranks in the bucket's second spare slot."""
        # utils/dataset.py:93-95 drops items whose embedding is [0]; a rank whose whole slice was dropped
        # has nothing to run.  With several ranks that must be a collective decision (a rank that
        # raised or skipped alone would leave the others blocked in the gradient all-reduce): every
        # rank contributes "I have a batch", and the step is skipped everywhere unless all do.
        if have is None:
            have = self._has(batch)
            if self._comm:
                have = self._agree([have])[0]
        if not have:
            raise EmptyBatch("a rank has no items in this step (all filtered by the collate): step skipped on every rank")
        emb, target, mixed, seq_len, _target_wav, phase = batch
        dev = self.device
        emb, target, mixed, phase = (t.to(dev, non_blocking=True) for t in (emb, target, mixed, phase))
        if seq_len is not None:
            seq_len = seq_len.to(dev, non_blocking=True).reshape(-1)
        if not self.model.training:                                         # train.py:84 (once; the walk over the module tree
            self.model.train()                                              # is 0.1 ms of host time with an idle device)
        mask = self.model(mixed, emb)                                       # train.py:94
        loss = self.criterion(mask, mixed, target, seq_len, phase)          # :95-108
        early = self.early_loss_read and loss.is_cuda and loss.dtype == torch.float32
        if early:
            self._loss_host.copy_(loss.detach().reshape(1), non_blocking=True)
            self._loss_event.record()
        split = False
        if not self._sink:
            self.bucket.zero()                                              # optimizer.zero_grad()
            loss.backward()                                                 # :110
        else:
            # (a parameter frozen since the bucket was built gets its slot zeroed by reattach: not beside a running collective)
            split = self.split_allreduce and self.bucket.all_frozen_free()
            self.model.set_gradient_sink(self._sink_map)
            if split:
                self.model.set_leaves_event(self._leaves_event)
            try:
                loss.backward()                                             # :110, written straight into the bucket views
            finally:
                self.model.set_gradient_sink(None)
                if split:
                    self.model.set_leaves_event(None)
        self.bucket.extra[0] = loss.detach()
        if self._comm:
            self.bucket.extra[1] = float(next_missing)
        self.bucket.reattach(written=self._sink)
        work = None
        if split:
            # the BiLSTM + head segment: final at the library's event (~5 ms into the backward pass), summed on the side stream beside
            # the conv stack's backward; the caller's stream picks it up in front of the small collective below
            self._comm_stream.wait_event(self._leaves_event)
            with torch.cuda.stream(self._comm_stream):
                work = self.bucket.all_reduce_early(self.world, force=self._force)
            if work is not None:
                work.wait()
        self.bucket.all_reduce(self.world, force=self._force, early_done=work is not None, reattached=True)      # the one exchange step
        ring = self._ring is not None
        if ring:                                                            # the two spare slots to the host, behind the collective
            slot = self._slot % 4
            self._ring[slot].copy_(self.bucket.extra, non_blocking=True)
            if self._ring_ev is not None:
                self._ring_ev[slot].record()
            self._slot += 1
        self.optimizer.step()                                               # :111
        self.step += 1                                                      # :112
        if early:
            self._loss_event.synchronize()                                  # long since complete: the host does not wait for the backward
            value = float(self._loss_host[0])                               # :114
        elif ring:
            if self.loss_lag and self._slot >= 2:
                value = self._read_slot(self._slot - 2, 0)                  # the step before: its event completed a step ago
            else:
                value = self._read_slot(self._slot - 1, 0)                  # waits for this step's all-reduce, not for the optimizer
        else:
            value = float(self.bucket.extra[0].item())                      # :114 (the reference syncs here too)
        self._last_value = value
        # :115-117, in the reference's order: the update has been applied and counted when the guard
        # fires, so step numbering and checkpoint cadence after an explosion match train.py
        if value > 1e8 or math.isnan(value):
            # a persistent BiLSTM launch that lost its CUs mid-flight poisons its output with NaN: say so instead of
            # reporting a numerical explosion (the host is synchronised here anyway)
            status = getattr(self.model, "lstm_status", None)
            if callable(status) and status() == 1:          # None = not knowable any more: report the explosion as such
                raise RuntimeError("the persistent BiLSTM kernel gave up waiting for a peer workgroup (another process is using "
                                   "the GPU?): this step's results were NaN-poisoned; rerun, or select the per-step kernels "
                                   "with vs_set_lstm_kernel(1)")
            raise LossExploded("Loss exploded to %.02f at step %d!" % (value, self.step))
        return value

    @torch.no_grad()
    def validate(self, batches: Iterable) -> float:
        """Mean criterion value over this rank's validation batches in eval mode, averaged over ranks
        (the loss part of utils/generic_utils.py:476-530; its SDR column needs mir_eval and stays
        with the reference's test.py)."""
        self.model.eval()
        tot, cnt = 0.0, 0
        for emb, target, mixed, seq_len, _tw, phase in batches:
            if emb is None or len(emb) == 0:          # every item of this batch was filtered by the collate (utils/dataset.py:93-95):
                continue                              # nothing to score; the (sum, count) all-reduce below copes with unequal counts
            dev = self.device
            emb, target, mixed, phase = (t.to(dev) for t in (emb, target, mixed, phase))
            if seq_len is not None:
                seq_len = seq_len.to(dev).reshape(-1)
            tot += float(self.criterion(self.model(mixed, emb), mixed, target, seq_len, phase).item())
            cnt += 1
        acc = torch.tensor([tot, float(cnt)], dtype=torch.float64, device=self.device)
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(acc, group=self.group)
        self.model.train()
        return float(acc[0] / acc[1]) if acc[1] > 0 else float("nan")

    def _with_decisions(self, batches: Iterable):
        """Yields (batch, have, next_missing) for ``train_step``.  One rank: have = None (decided in the step, no collective).
        Several ranks: the epoch's first two decisions come from ONE blocking MIN-reduce; from then on every rank's "my batch k + 2 is
        empty" rides in step k's gradient bucket (``next_missing``) and the verdict for batch k + 2 is read from the host ring when its
        turn comes -- two steps after the collective that carried it, so the read waits for nothing.  A step that is skipped has no
        gradient collective to ride on: its flag gets a small blocking reduce of its own (empty batches are rare:
        utils/dataset.py:93-95).  Every rank must see the same number of batches (``EpochShard`` guarantees it)."""
        if not self._comm or self._ring is None:
            for batch in batches:
                yield batch, None, 0.0
            return
        it = iter(batches)
        ahead = []

        def pull():
            try:
                ahead.append(next(it))
            except StopIteration:
                ahead.append(_END)

        pull()
        pull()
        decided = list(self._agree([b is _END or self._has(b) for b in ahead]))
        while ahead[0] is not _END:
            batch, d = ahead.pop(0), decided.pop(0)
            pull()                                                          # ahead = [batch k + 1, batch k + 2]
            nxt = ahead[1]
            miss = 0.0 if (nxt is _END or self._has(nxt)) else 1.0
            have = d if isinstance(d, bool) else self._read_slot(d.slot, 1) < 0.5
            slot_before = self._slot
            if have:
                yield batch, True, miss                                     # the consumer runs train_step now
            if have and self._slot == slot_before + 1:
                decided.append(_FlagOfStep(slot_before))
            else:
                # skipped here, or the step raised before its collective (LossExploded leaves the epoch: the generator is dropped)
                if not have:
                    yield batch, False, 0.0                                 # train_step raises EmptyBatch on every rank
                decided.append(self._agree([miss == 0.0])[0])

    def fit(self, batches_for_epoch: Callable[[int], Iterable], epochs: Optional[int] = None, log_dir: Optional[str] = None,
            on_log: Optional[Callable[[int, float], None]] = None, validation_batches: Optional[Callable[[], Iterable]] = None):
        """train.py:81-135.  ``batches_for_epoch(e)`` yields this rank's batches of epoch e."""
        tc = self.c.train_config
        epochs = tc["epochs"] if epochs is None else epochs
        for e in range(epochs):
            if validation_batches is not None:                              # :82
                v = self.validate(validation_batches())
                if on_log and self.rank == 0:
                    on_log(-self.step, v)
            for batch, have, miss in self._with_decisions(batches_for_epoch(e)):
                try:
                    loss = self.train_step(batch, have=have, next_missing=miss)
                except EmptyBatch:
                    continue
                except LossExploded as err:                                 # :115-117: leave this epoch
                    if self.rank == 0:
                        print(err)
                    break
                if self.step % tc["summary_interval"] == 0 and on_log and self.rank == 0:     # :120-122
                    on_log(self.step, loss)
                if log_dir and self.step % tc["checkpoint_interval"] == 0:                     # :125-134
                    p = self.save_checkpoint(os.path.join(log_dir, "checkpoint_%d.pt" % self.step))
                    if p:
                        print("Saved checkpoint to: %s" % p)
                    if validation_batches is not None:                      # :134 validation after every checkpoint
                        v = self.validate(validation_batches())
                        if on_log and self.rank == 0:
                            on_log(-self.step, v)
        return self.step


# ---------------------------------------------------------------------------------------------
# data: the reference's on-disk training set, STFT moved to the GPU
# ---------------------------------------------------------------------------------------------
def load_wav(path: str, sample_rate: int) -> torch.Tensor:
    """float32 mono in [-1, 1] like ``librosa.load(path, sr=sample_rate)`` for files already at
    ``sample_rate`` (utils/audio_processor.py:516-518); there is no resampler here."""
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    if sr != sample_rate:
        raise ValueError(f"{path}: sample rate {sr} != {sample_rate} (resample the dataset first)")
    t = torch.from_numpy(data.copy())
    if t.dtype == torch.int16:
        t = t.float() / 32768.0
    elif t.dtype == torch.int32:
        t = t.float() / 2147483648.0
    elif t.dtype == torch.uint8:
        t = (t.float() - 128.0) / 128.0
    else:
        t = t.float()
    return t.mean(dim=1) if t.dim() == 2 else t


class SpecWavDataset:
    """The training items of utils/dataset.py:8-41 read from ``c.dataset['train_dir']`` with the
    globs of ``c.dataset['format']``: emb ``*-emb.pt``, target spectrogram ``*-target.pt``, and the
    two wavs.  ``__getitem__`` returns (emb, target_spec, mixed_wav, seq_len, target_wav): the mixed
    spectrogram and phase are NOT computed here per item by librosa but once per batch on the GPU
    (``collate``), which is the point of the front end of audio.py."""

    def __init__(self, c, train: bool = True):
        self.c = c
        self.dir = c.dataset["train_dir"] if train else c.dataset["test_dir"]
        if not os.path.isdir(self.dir):
            raise FileNotFoundError("Test or Train dataset dir is incorrect! Fix it in config.json: " + str(self.dir))
        fmt = c.dataset["format"]
        find = lambda g: sorted(glob(os.path.join(self.dir, g)))
        self.emb, self.target, self.target_wav, self.mixed_wav = (find(fmt[k]) for k in ("emb", "target", "target_wav", "mixed_wav"))
        if not (len(self.emb) == len(self.target) == len(self.mixed_wav) == len(self.target_wav)):
            raise ValueError(" The number of target and mixed Specs and Embs not Match! Check its")
        if not self.emb:
            raise ValueError(" Training files not found !")
        self.sr = int(c.audio[c.audio["backend"]]["sample_rate"])

    def __len__(self):
        return len(self.emb)

    def __getitem__(self, i):
        mixed_wav = load_wav(self.mixed_wav[i], self.sr)
        target_wav = load_wav(self.target_wav[i], self.sr)
        seq_len = torch.tensor([mixed_wav.shape[0]])
        return torch.load(self.emb[i]), torch.load(self.target[i]), mixed_wav, seq_len, target_wav

    def collate(self, items, device) -> tuple:
        """train_collate_fn (utils/dataset.py:84-114) + get_spec_from_audio for the whole batch on
        ``device``: returns (emb, target, mixed, seq_len, target_wav, mixed_phase)."""
        from . import audio
        items = [it for it in items if it[0].tolist() != [0]]               # :93-95
        if not items:
            return (None,) * 6              # Trainer.train_step turns this into a collective skip
        emb = torch.stack([it[0].float().reshape(-1) for it in items]).to(device)
        target = torch.stack([it[1].float() for it in items]).to(device)
        wav = torch.stack([it[2] for it in items]).to(device)
        seq_len = torch.stack([it[3] for it in items]).reshape(-1).to(device)
        target_wav = torch.stack([it[4] for it in items])
        mixed, phase = audio.wav_to_spec(wav, self.c.audio[self.c.audio["backend"]], want_phase=True)
        return emb, target, mixed, seq_len, target_wav, phase


def host_collate(items) -> tuple:
    """The CPU half of ``SpecWavDataset.collate`` (train_collate_fn, utils/dataset.py:84-114, without the per-item librosa STFT):
    drops items whose embedding is [0] (:93-95) and stacks the rest -> (emb [B,E], target [B,T,F], wav [B,S], seq_len [B],
    target_wav [B,S]), or None when nothing is left.  Runs in the DataLoader's worker processes."""
    items = [it for it in items if it[0].tolist() != [0]]
    if not items:
        return None
    return (torch.stack([it[0].float().reshape(-1) for it in items]), torch.stack([it[1].float() for it in items]),
            torch.stack([it[2] for it in items]), torch.stack([it[3] for it in items]).reshape(-1), torch.stack([it[4] for it in items]))


class BatchFeeder:
    """Feeds ``Trainer.fit`` from the on-disk training set so that the step never waits for its input (the
    reference: ``DataLoader(num_workers=14, pin_memory=True)``, utils/dataset.py:60-68, with librosa's STFT inside every worker).

      * items are read by ``num_workers`` worker processes (``c.train_config['num_workers']``) of a ``torch.utils.data.DataLoader``
        whose batch sampler is this rank's ``EpochShard`` -- the sharding is exactly the synchronous loop's -- into pinned memory;
      * batch k + 1 goes host -> device on a copy stream while step k runs (the stream first waits for what the caller's stream has
        been given so far: its buffers may be blocks that work still reads);
      * the GPU front end (``audio.wav_to_spec``: wav -> normalised dB spectrogram + phase for the whole batch) of batch k + 1 is
        enqueued on the caller's stream when the consumer comes back for it, i.e. behind step k's optimizer.
    ``feeder.epoch(e)`` yields what ``SpecWavDataset.collate`` returns: (emb, target, mixed, seq_len, target_wav, phase), or six Nones
    for a batch whose items were all filtered out.  On a CPU device (tests) the copies are plain and there are no streams."""

    def __init__(self, dataset: "SpecWavDataset", shard: EpochShard, device, num_workers: Optional[int] = None, prefetch_factor: int = 2):
        self.ds, self.shard, self.device = dataset, shard, torch.device(device)
        nw = dataset.c.train_config.get("num_workers", 0) if num_workers is None else num_workers
        self.num_workers = max(0, int(nw))
        self.prefetch_factor = prefetch_factor
        self.cuda = self.device.type == "cuda"
        self._copy = torch.cuda.Stream(self.device) if self.cuda else None
        self.acfg = dataset.c.audio[dataset.c.audio["backend"]]

    def host_batches(self, epoch: int, chain: int = 1):
        """The DataLoader of one epoch: host tuples (``host_collate``) in ``EpochShard`` order.  chain > 1: the index batches of
        epochs epoch .. epoch + chain - 1 behind one another in ONE loader pass (a benchmark over a small set: the worker processes
        start once, not every few steps)."""
        from torch.utils.data import DataLoader
        index_batches = [idx for e in range(epoch, epoch + chain) for idx in self.shard.epoch(e)]
        kw = dict(batch_sampler=index_batches, collate_fn=host_collate, num_workers=self.num_workers,
                  pin_memory=self.cuda)
        if self.num_workers > 0:
            kw.update(prefetch_factor=self.prefetch_factor, persistent_workers=False)
        return DataLoader(self.ds, **kw)

    def _start(self, host):
        """host tuple -> device buffers being filled on the copy stream (+ the event that says they are)"""
        if host is None:
            return None
        emb, target, wav, seq_len, target_wav = host
        if not self.cuda:
            return (emb.to(self.device), target.to(self.device), wav.to(self.device), seq_len.to(self.device), target_wav), None
        dst = [torch.empty(t.shape, dtype=t.dtype, device=self.device) for t in (emb, target, wav, seq_len)]
        self._copy.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._copy):
            for d, t in zip(dst, (emb, target, wav, seq_len)):
                d.copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        return (dst[0], dst[1], dst[2], dst[3], target_wav), (ev, host)       # the pinned source stays referenced until the copy is done

    def _finish(self, staged):
        """the front end of a staged batch on the caller's stream"""
        from . import audio
        if staged is None:
            return (None,) * 6
        (emb, target, wav, seq_len, target_wav), ev = staged
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev[0])
        mixed, phase = audio.wav_to_spec(wav, self.acfg, want_phase=True)
        return emb, target, mixed, seq_len, target_wav, phase

    def epoch(self, epoch: int, chain: int = 1):
        it = iter(self.host_batches(epoch, chain))
        host = next(it, _END)
        if host is _END:
            return
        cur = self._finish(self._start(host))
        while True:
            host = next(it, _END)                              # already waiting in the loader's queue in the steady state
            staged = self._start(host) if host is not _END else _END
            yield cur                                          # the consumer enqueues its step on this batch
            if staged is _END:
                return
            cur = self._finish(staged)                         # behind that step on the caller's stream


def synthetic_batches(steps: int, B: int, T: int, F: int, E: int, hop: int, device, seed: int = 0):
    """Batches of the reference's shape with random content (bench.py, smoke tests): one fixed batch
    per distinct seed, yielded ``steps`` times."""
    g = torch.Generator().manual_seed(seed)
    mixed = torch.rand(B, T, F, generator=g).to(device)
    target = (mixed.cpu() * torch.rand(B, T, F, generator=g)).to(device)
    phase = ((torch.rand(B, T, F, generator=g) - 0.5) * 6.2).to(device)
    emb = torch.randn(B, E, generator=g)
    emb = (emb / emb.norm(dim=1, keepdim=True)).to(device)
    seq_len = torch.full((B,), hop * (T - 1), dtype=torch.int32, device=device)
    for _ in range(steps):
        yield emb, target, mixed, seq_len, None, phase


def main(argv=None):
    """``python -m torch.distributed.run --nproc-per-node N -m voicesplit_amd.trainer -c config.json``
    (train.py's CLI: --config_path/-c, --checkpoint_path; plus --synthetic-steps for a dry run)."""
    import argparse
    import torch.distributed as dist
    from . import VoiceFilter, VoiceSplit, load_config
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config_path", required=True)
    ap.add_argument("--checkpoint_path", default=None)
    ap.add_argument("--synthetic-steps", type=int, default=0, help="train on random batches of the configured shape instead of c.dataset")
    ap.add_argument("--epochs", type=int, default=None)
    args = ap.parse_args(argv)
    c = load_config(args.config_path)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("voicesplit_amd.trainer needs a GPU: the mask path has no CPU implementation")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    torch.manual_seed(c.train_config["seed"])
    name = c.model_name
    if name == "voicefilter":
        model = VoiceFilter(c)
    elif name == "voicesplit":
        model = VoiceSplit(c)
    else:
        raise Exception(" The model '" + name + "' is not suported")       # train.py:31
    tr = Trainer(model.to(dev), c, rank, world)
    if args.checkpoint_path:
        tr.load_checkpoint(args.checkpoint_path, c.train_config.get("reinit_layers"))
    log_dir = os.path.join(c.train_config["logs_path"], c.model_name)      # train.py:154
    if rank == 0:
        os.makedirs(log_dir, exist_ok=True)
    acfg = c.audio[c.audio["backend"]]
    b = c.train_config["batch_size"]
    if args.synthetic_steps:
        T = 1 + (int(c.audio["audio_len"]) * acfg["sample_rate"]) // acfg["hop_length"]
        batches = lambda e: synthetic_batches(args.synthetic_steps, b, T, acfg["num_freq"], c.model["emb_dim"],
                                              acfg["hop_length"], dev, seed=1000 * e + rank)
    else:
        ds = SpecWavDataset(c, train=True)
        shard = EpochShard(len(ds), b, rank, world, c.train_config["seed"])
        feeder = BatchFeeder(ds, shard, dev)                  # worker processes + copy stream: the step never waits for its input
        batches = feeder.epoch
    log = lambda step, loss: print(("validation" if step <= 0 else "step %d" % step) + " loss %.5f" % loss, flush=True)
    tr.fit(batches, args.epochs, log_dir, log)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
