"""Long-form inference (BASELINE.json configs[4]): a clip longer than the 301-frame training window
is cut into windows that become items of one batch, the mask-prediction path runs once, and the
per-window masks are stitched back.  The reference only ever runs whole utterances through the
model (test.py feeds the full spectrogram, utils/generic_utils.py:533-546); windowing is what makes
a 30 s clip a data-parallel job: windows are independent, so they shard across GPUs with no
collective (voicesplit_amd/sharding.py).

halo = 0 reproduces BASELINE's "independent 301-frame windows" exactly.  halo > 0 gives every
window `halo` extra frames of context on both sides (65 covers the conv stack's 131-frame
receptive field) and keeps only the centre frames; in ``separate_long`` the BiLSTM still restarts
in every window.

``separate_long_exact`` is the variant that equals a whole-clip pass (what the reference computes,
utils/generic_utils.py:495 feeds any T): only the conv stack -- 95 % of the work and all of the
activation memory -- is windowed (halo >= 65: every kept frame sees its full receptive field, clip
edges are zero-padded exactly as the whole-clip convolution pads them), the stitched feature
sequence then goes through the BiLSTM and the head ONCE at full length, so the recurrent state is
carried across every window boundary in both directions by construction.
"""
from typing import Callable, List, Tuple

import torch

CONV_RECEPTIVE_HALO = 65      # (131 - 1) / 2 frames: 1 + 6 + 4*(1 + 2 + 4 + 8 + 16) = 131 (models/voicesplit/model.py:17-48)


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



def separate_long_many(model: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], specs: torch.Tensor,
                       dvecs: torch.Tensor, window: int = 301, max_batch: int = 256) -> torch.Tensor:
    """BASELINE configs[4] as a batch job: specs [N, T_long, F] (N clips of equal length), dvecs [N, E] ->
    masks [N, T_long, F2].  Every clip is cut into independent `window`-frame items (the last one zero padded),
    the windows of ALL clips form one list that goes through the model `max_batch` at a time (256 windows per
    batch in the configuration), and the masks are cut back to T_long.  Same result as ``separate_long`` with
    halo = 0 clip by clip; it exists so that a batch never runs half empty at a clip boundary."""
    if specs.dim() != 3 or dvecs.dim() != 2 or specs.shape[0] != dvecs.shape[0]:
        raise ValueError("specs must be [N, T, F] and dvecs [N, E]")
    N, T_long, F = specs.shape
    nw = -(-T_long // window)
    pad = nw * window - T_long
    x = torch.nn.functional.pad(specs, (0, 0, 0, pad)) if pad else specs
    x = x.reshape(N * nw, window, F)
    emb = dvecs.unsqueeze(1).expand(N, nw, dvecs.shape[1]).reshape(N * nw, -1)
    out = None
    with torch.no_grad():
        for b0 in range(0, N * nw, max_batch):
            mb = model(x[b0:b0 + max_batch].contiguous(), emb[b0:b0 + max_batch].contiguous())
            if out is None:
                out = mb.new_empty(N * nw, window, mb.shape[2])
            out[b0:b0 + mb.shape[0]] = mb
    return out.reshape(N, nw * window, -1)[:, :T_long]


def plan_windows_exact(n_frames: int, window: int = 301, halo: int = CONV_RECEPTIVE_HALO) -> List[Tuple[int, int, int]]:
    """[(start, keep_lo, keep_hi)] for ``separate_long_exact``: every window lies INSIDE the clip (a frame
    outside the clip must stay a zero-padded activation in every conv layer, which a zero input frame
    inside a window is not: bias, BatchNorm shift and the activation make it non-zero after cnn1), the
    first starts at frame 0, the last ends at the last frame, interior windows advance by window - 2*halo.
    Window w covers frames [start, start + min(window, n_frames)) and contributes its local frames
    [keep_lo, keep_hi): at least `halo` frames away from every window edge that is not a clip edge."""
    if n_frames <= 0:
        raise ValueError("n_frames must be positive")
    if window <= 2 * halo:
        raise ValueError("window must be larger than 2*halo")
    if n_frames <= window:
        return [(0, 0, n_frames)]
    step = window - 2 * halo
    starts = [0]
    while starts[-1] + window < n_frames:
        starts.append(min(starts[-1] + step, n_frames - window))
    out, pos = [], 0
    for k, st in enumerate(starts):
        hi_abs = n_frames if k == len(starts) - 1 else st + window - halo
        out.append((st, pos - st, hi_abs - st))
        pos = hi_abs
    return out


def separate_long_exact(conv_stage: Callable[[torch.Tensor], torch.Tensor],
                        sequence_stage: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
                        spec: torch.Tensor, dvec: torch.Tensor, window: int = 301, halo: int = CONV_RECEPTIVE_HALO,
                        max_batch: int = 256) -> torch.Tensor:
    """mask [T_long, F2] equal to one whole-clip pass.  ``conv_stage(x[B,Tw,F]) -> feat[B,Tw,C]`` is the
    conv stack in eval mode (frame t of its output depends on input frames t-halo..t+halo only);
    ``sequence_stage(feat[1,T_long,C], dvec[1,E]) -> mask[1,T_long,F2]`` is everything behind it
    (d-vector concat, BiLSTM, head).  ``VoiceSplit.long_form_stages()`` returns the two for the HIP path."""
    if spec.dim() != 2 or dvec.dim() != 1:
        raise ValueError("spec must be [T, F] and dvec [E]")
    if halo < 0:
        raise ValueError("halo must be >= 0")
    T_long, F = spec.shape
    plan = plan_windows_exact(T_long, window, halo)
    wlen = min(window, T_long)
    batch = torch.stack([spec[st:st + wlen] for st, _k0, _k1 in plan])
    feats = []
    with torch.no_grad():
        for b0 in range(0, len(plan), max_batch):
            feats.append(conv_stage(batch[b0:b0 + max_batch].contiguous()))
        feat_w = torch.cat(feats, dim=0)
        full = feat_w.new_empty(1, T_long, feat_w.shape[2])
        pos = 0
        for w, (_st, keep_lo, keep_hi) in enumerate(plan):
            n = keep_hi - keep_lo
            full[0, pos:pos + n] = feat_w[w, keep_lo:keep_hi]
            pos += n
        assert pos == T_long
        return sequence_stage(full, dvec.unsqueeze(0).contiguous())[0]
