"""Long-form inference (BASELINE.json configs[4]): a clip longer than the 301-frame training window
is cut into windows that become items of one batch, the mask-prediction path runs once, and the
per-window masks are stitched back.  The reference only ever runs whole utterances through the
model (test.py feeds the full spectrogram, utils/generic_utils.py:533-546); windowing is what makes
a 30 s clip a data-parallel job: windows are independent, so they shard across GPUs with no
collective (voicesplit_amd/sharding.py).

halo = 0 reproduces BASELINE's "independent 301-frame windows" exactly.  halo > 0 gives every
window `halo` extra frames of context on both sides (65 covers the conv stack's 131-frame
receptive field; the BiLSTM state is not carried across windows) and keeps only the centre frames.
"""
from typing import Callable, List, Tuple

import torch


def plan_windows(n_frames: int, window: int = 301, halo: int = 0) -> List[Tuple[int, int, int, int]]:
    """[(src_lo, src_hi, keep_lo, keep_hi)]: window w reads frames [src_lo, src_hi) of the clip
    (clipped to it; the rest of the `window` frames is zero padding) and contributes its local
    frames [keep_lo, keep_hi) to the output.  The kept spans tile [0, n_frames) exactly."""
    if n_frames <= 0:
        raise ValueError("n_frames must be positive")
    if window <= 2 * halo:
        raise ValueError("window must be larger than 2*halo")
    step = window - 2 * halo
    out = []
    start = 0
    while start < n_frames:
        src_lo = start - halo
        lo = max(src_lo, 0)
        hi = min(src_lo + window, n_frames)
        keep_lo = start - src_lo                      # == halo, also for the first window (left pad)
        keep_hi = min(keep_lo + step, hi - src_lo)
        out.append((lo, hi, keep_lo, keep_hi))
        start += step
    return out


def separate_long(model: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], spec: torch.Tensor,
                  dvec: torch.Tensor, window: int = 301, halo: int = 0, max_batch: int = 256) -> torch.Tensor:
    """mask [T_long, F] for one clip: spec [T_long, F] (same normalisation as training inputs),
    dvec [E].  `model` is the VoiceSplit/VoiceFilter module (or any (x[B,T,F], emb[B,E]) -> mask).
    Windows are processed `max_batch` at a time (BASELINE configs[4]: 256 windows per batch)."""
    if spec.dim() != 2 or dvec.dim() != 1:
        raise ValueError("spec must be [T, F] and dvec [E]")
    T_long, F = spec.shape
    plan = plan_windows(T_long, window, halo)
    batch = spec.new_zeros(len(plan), window, F)
    for w, (lo, hi, keep_lo, _) in enumerate(plan):
        dst = lo - (w * (window - 2 * halo) - halo)   # where frame `lo` lands inside the window
        batch[w, dst:dst + (hi - lo)] = spec[lo:hi]
    emb = dvec.unsqueeze(0)
    out = spec.new_empty(T_long, 0)
    pieces = []
    with torch.no_grad():
        for b0 in range(0, len(plan), max_batch):
            xb = batch[b0:b0 + max_batch].contiguous()
            mb = model(xb, emb.expand(xb.shape[0], -1).contiguous())
            pieces.append(mb)
    masks = torch.cat(pieces, dim=0)
    out = spec.new_empty(T_long, masks.shape[2])
    pos = 0
    for w, (_, _, keep_lo, keep_hi) in enumerate(plan):
        n = keep_hi - keep_lo
        out[pos:pos + n] = masks[w, keep_lo:keep_hi]
        pos += n
    assert pos == T_long
    return out
