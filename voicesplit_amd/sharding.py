"""Batch sharding of the mask-prediction path across the GPUs of a node.

Utterances are independent in the forward pass (eval-mode BN; the reference has no SyncBN and no
multi-GPU code at all -- train.py:64-65, run_train.sh:1), so N GPUs run N contiguous slices of
the batch with replicated weights and no data-path collective.  The helpers here are host logic
only: slice arithmetic, and an optional all-gather for callers that want every rank to hold the
full mask (embarrassingly-parallel inference does not need it).
"""
import os
from typing import Callable, Optional, Tuple

import torch


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of `n_items` owned by `rank`; sizes differ by at most one and the
    first `n_items % world` ranks get the extra item (ragged batches are allowed, including
    ranks that own nothing when n_items < world)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    if n_items < 0:
        raise ValueError("n_items must be >= 0")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def chunk_windows(n_frames: int, window: int = 301) -> int:
    """BASELINE config 5: a long clip is cut into independent `window`-frame items (the last one
    zero padded); returns how many batch items a clip of n_frames becomes."""
    return max(1, -(-n_frames // window))


def run_sharded(fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], x: torch.Tensor,
                dvec: torch.Tensor, rank: int, world: int, group=None, gather: bool = False
                ) -> Optional[torch.Tensor]:
    """Apply `fn(x_slice, dvec_slice) -> mask_slice` to this rank's utterances.

    With gather=False returns the local slice (or None if the rank owns nothing).  With
    gather=True every rank returns the full [B, T, F] mask via one all_gather of padded slices
    (ranks may own different counts).
    """
    B = x.shape[0]
    lo, hi = shard_range(B, rank, world)
    local = fn(x[lo:hi].contiguous(), dvec[lo:hi].contiguous()) if hi > lo else None
    if not gather or world == 1:
        return local
    import torch.distributed as dist
    per = -(-B // world)                                   # padded slice length
    shape = None
    if local is not None:
        shape = torch.tensor(list(local.shape[1:]), dtype=torch.int64, device=x.device)
    else:
        shape = torch.zeros(2, dtype=torch.int64, device=x.device)
    dist.all_reduce(shape, op=dist.ReduceOp.MAX, group=group)
    tail = tuple(int(v) for v in shape.tolist())
    buf = torch.zeros((per,) + tail, dtype=x.dtype, device=x.device)
    if local is not None:
        buf[: hi - lo] = local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    out = []
    for r in range(world):
        a, b = shard_range(B, r, world)
        out.append(parts[r][: b - a])
    return torch.cat(out, dim=0)


# ---------------------------------------------------------------------------------------------
# data-parallel training: the one exchange step of the path (BASELINE config 4)
# ---------------------------------------------------------------------------------------------

class GradientBucket:
    """One flat fp32 bucket for every parameter gradient (18 876 001 floats = 75.5 MB at the
    default config) and a single sum-all-reduce over it per step -- RCCL over xGMI on the GPUs
    (backend "nccl"), gloo in the CPU tests.  One large collective instead of per-tensor ones:
    xGMI rings are per-link bound, so 44 small all-reduces would be latency dominated.

    Convention: gradients are averaged over ranks, so N ranks with local batch b and a
    mean-over-batch loss reproduce one process with batch N*b (utils/generic_utils.py:473 takes
    the mean over the batch).  BatchNorm statistics stay per replica exactly as in the reference
    (no SyncBN); ``sync_buffers`` broadcasts rank 0's running statistics when a checkpoint is cut.

    Layout of ``flat``: [extra slots | gradients of the "late" parameters | gradients of the ``early`` parameters].  ``early`` names
    the parameters whose gradients are final long before the backward pass ends (the BiLSTM and the head: 73 of the 75.5 MB, final
    ~5 ms into a 30 ms backward); ``all_reduce_early`` sums that segment on its own -- on a side stream beside the rest of the
    backward -- and ``all_reduce(early_done=True)`` then only moves the first segment, which carries the extra slots (loss value,
    the skip flag of the training loop) and the conv stack's 2.2 MB."""

    def __init__(self, params, group=None, extra: int = 0, early=None):
        """extra: spare floats at the front of the bucket (``self.extra``) that ride along in the same
        collective -- the trainer puts the loss there so that every rank logs the averaged value.
        early: the parameters of the early segment (any subset of params; None: no early segment)."""
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.numel = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        early_ids = {id(p) for p in (early or ())}
        late = [p for p in self.params if id(p) not in early_ids]
        first = [p for p in self.params if id(p) in early_ids]
        # every view starts on a 16-byte boundary (offsets rounded up to 4 elements; the padding is zeros that ride along in the collective):
        # the library's LSTM contraction writes dW_ih with 16-byte stores when it can and falls back to an older, slower kernel when it
        # cannot (gemm_bf16.hip: vec_ok) -- with two spare floats in front, every gradient used to sit 8 bytes off [r6, call 36: step 44.60 -> 44.35 ms]
        up = lambda n: (n + 3) // 4 * 4
        offs, off = {}, up(extra)
        for p in late:
            offs[id(p)] = off
            off = up(off + p.numel())
        self.split = off                                       # the early segment is flat[split:]
        for p in first:
            offs[id(p)] = off
            off = up(off + p.numel())
        self.flat = torch.zeros(off, dtype=p0.dtype, device=p0.device)
        self.extra = self.flat[:extra]
        self.grads = self.flat[up(extra):]                     # every gradient (and the alignment padding between them), without the extra slots
        self.has_early = len(first) > 0
        self.time_events = False                             # bench.py: record a HIP-event pair around every collective
        self._events = []
        self._early_events = []
        self.views = [self.flat[offs[id(p)]:offs[id(p)] + p.numel()].view_as(p) for p in self.params]

    def attach(self):
        """Make every ``p.grad`` a view into the bucket, so backward accumulates straight into it
        and no gather/scatter copy is needed around the all-reduce."""
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v
        return self

    def all_frozen_free(self) -> bool:
        """No parameter of the bucket has been frozen since it was built (a frozen one gets its slot zeroed by ``reattach``: a write
        the early collective must not race with)."""
        return all(p.requires_grad for p in self.params)

    def reattach(self, written: bool = False):
        """Bring ``.grad`` and the bucket views back together before a collective.
        written: the views were just written by the producer of the gradients itself (the library's
        gradient sink, trainer.py) and are authoritative: every ``.grad`` is re-attached to its view, whatever it held."""
        for p, v in zip(self.params, self.views):
            if written:
                # the views ARE the step's gradients.  Whatever .grad holds instead -- None after zero_grad(set_to_none=True), or a
                # tensor autograd created for a plain backward() over the same model before this step -- is stale: re-attach, never
                # copy it in.  A parameter frozen since the bucket was built (requires_grad_(False)) keeps its .grad untouched and its
                # slot zeroed (the library writes every gradient into the sink), so an optimizer that finds the attached view does nothing.
                if p.requires_grad:
                    if p.grad is not v:
                        p.grad = v
                else:
                    v.zero_()
            elif p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():        # someone re-created .grad and accumulated into it: copy in
                v.copy_(p.grad)
                p.grad = v

    def _timed(self, store):
        if self.time_events and self.flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            store.append(ev)
            return ev
        return None

    def all_reduce_early(self, world: int, force: bool = False):
        """Sum the early segment over ranks (no division: ``all_reduce`` divides the whole bucket).  Issued on the CURRENT stream --
        the caller makes that a side stream which waited for the event that says the segment is final (vs_grads.leaves_event) -- as
        an asynchronous collective; returns the work handle (``wait()`` makes the then-current stream wait for it; None when there is
        nothing to do)."""
        if not self.has_early or not (world > 1 or force):
            return None
        import torch.distributed as dist
        ev = self._timed(self._early_events)
        if ev:
            ev[0].record()
        work = dist.all_reduce(self.flat[self.split:], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if ev:
            work.wait()                                      # the current (side) stream waits for the collective: the pair brackets it
            ev[1].record()
        return work

    def all_reduce(self, world: int, force: bool = False, written: bool = False, early_done: bool = False, reattached: bool = False):
        """Sum over ranks, then divide by `world` (call after backward has finished: the
        collective must not be co-scheduled with the persistent LSTM kernels).  force: issue the
        collective even at world 1 (a one-rank RCCL all-reduce: used to exercise the device path).
        written: see ``reattach`` (skipped when the caller already did it: ``reattached``).  early_done: ``all_reduce_early`` has
        summed the early segment (and the current stream has waited for it): only flat[:split] is moved.  Returns
        the flat bucket; ``collective_ms()`` (when ``time_events`` is set) = HIP-event time of the collective itself."""
        import torch.distributed as dist
        if not reattached:
            self.reattach(written)
        if world > 1 or force:
            ev = self._timed(self._events)
            if ev:
                ev[0].record()
            part = self.flat[:self.split] if (early_done and self.has_early) else self.flat
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)
            if ev:
                ev[1].record()                               # the current stream waits for the collective: the pair brackets it
            if world > 1:
                self.flat.div_(world)
        return self.flat

    def collective_ms(self, early: bool = False):
        """HIP-event times (ms) of the all-reduces issued since the last call while ``time_events`` was on (synchronises);
        early: those of the early segment."""
        out = []
        store = self._early_events if early else self._events
        for e0, e1 in store:
            e1.synchronize()
            out.append(e0.elapsed_time(e1))
        del store[:]
        return out

    def zero(self):
        self.flat.zero_()


def sync_buffers(module: torch.nn.Module, src: int = 0, group=None):
    """Broadcast rank `src`'s BatchNorm running statistics / counters (DDP's default policy)."""
    import torch.distributed as dist
    for b in module.buffers():
        dist.broadcast(b, src=src, group=group)


def init_single_rank_group(backend: str, device_id=None, env=None):
    """A ONE-rank process group for this process (bench.py's RCCL leg at N = 1: the gradient bucket through a real
    ncclAllReduce although the step's exchange is a no-op).  Two ways to get there, and the wrong one waits for minutes:

    * started by ``torch.distributed.run`` (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment): under
      ``TORCHELASTIC_USE_AGENT_STORE`` torch creates EVERY TCPStore as a client of the launcher's store, so a private
      ``tcp://127.0.0.1:<free port>`` rendezvous waits for a server that does not exist until the timeout -- the group
      has to come from ``env://`` (the launcher's store, rank 0 of 1);
    * started as a plain process: a private ``tcp://`` store on a free local port (no environment needed).
    """
    import torch.distributed as dist
    env = os.environ if env is None else env
    kw = {"device_id": device_id} if device_id is not None else {}
    launched = all(k in env for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"))
    if launched:
        if int(env["WORLD_SIZE"]) != 1:
            raise RuntimeError("init_single_rank_group: this process belongs to a job of %s ranks" % env["WORLD_SIZE"])
        dist.init_process_group(backend, **kw)
        return "env://"
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    method = f"tcp://127.0.0.1:{port}"
    dist.init_process_group(backend, init_method=method, rank=0, world_size=1, **kw)
    return method
