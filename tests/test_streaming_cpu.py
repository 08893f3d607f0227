"""CPU: the long-form windowing host logic (BASELINE configs[4]) with a stand-in model."""
import pytest
import torch

from voicesplit_amd.streaming import plan_windows, plan_windows_exact, separate_long, separate_long_exact


def test_plan_tiles_the_clip_exactly():
    for n in (1, 300, 301, 302, 3001, 1000):
        for halo in (0, 10, 65):
            plan = plan_windows(n, 301, halo)
            kept = sum(k1 - k0 for _, _, k0, k1 in plan)
            assert kept == n
            for lo, hi, k0, k1 in plan:
                assert 0 <= lo < hi <= n and hi - lo <= 301 and 0 <= k0 < k1 <= 301
    assert len(plan_windows(3001, 301, 0)) == 10          # 30 s clip -> 10 windows, last one padded
    with pytest.raises(ValueError):
        plan_windows(10, 100, 50)


def _local_model(x, emb):
    # frame-local stand-in: the stitched result must equal the model applied to the whole clip
    return torch.sigmoid(x * emb[:, :1, None] + emb[:, 1:2, None])


def _context_model(x, emb):
    # mixes +-3 frames (zero padded inside the window): only correct under stitching if halo >= 3
    k = torch.ones(1, 1, 7, 1) / 7
    return torch.nn.functional.conv2d(x[:, None], k, padding=(3, 0))[:, 0]


@pytest.mark.parametrize("T_long", [1, 300, 301, 777, 3001])
def test_stitching_matches_whole_clip(T_long):
    g = torch.Generator().manual_seed(0)
    spec = torch.rand(T_long, 13, generator=g)
    dvec = torch.randn(4, generator=g)
    whole = _local_model(spec[None], dvec[None])[0]
    for halo in (0, 65):
        got = separate_long(_local_model, spec, dvec, 301, halo, max_batch=4)
        assert torch.allclose(got, whole, atol=1e-6, rtol=0)
    ctx_whole = _context_model(spec[None], dvec[None])[0]
    got = separate_long(_context_model, spec, dvec, 301, halo=5)
    assert torch.allclose(got, ctx_whole, atol=1e-6)
    if T_long > 301:
        assert not torch.allclose(separate_long(_context_model, spec, dvec, 301, halo=0), ctx_whole, atol=1e-6)


@pytest.mark.parametrize("T_long", [1, 301, 302, 1000, 3001])
def test_exact_long_form_carries_sequence_state_across_windows(T_long):
    """Stand-ins with the structure of the path: a conv stage that mixes +-65 frames and a sequence stage
    with unbounded memory in BOTH directions (a prefix sum plus a suffix sum).  Windowed conv stage + one
    full-length sequence stage must equal the whole-clip computation; the restart-per-window variant must not."""
    g = torch.Generator().manual_seed(1)
    spec = torch.rand(T_long, 7, generator=g)
    dvec = torch.randn(3, generator=g)
    k = torch.rand(1, 1, 131, 1, generator=g) / 131

    def conv_stage(x):
        # two layers with a bias in front of the second: a zero INPUT frame is not a zero activation, so windows
        # must never reach past the clip (the whole-clip convolution zero-pads the activations of every layer)
        y = torch.nn.functional.conv2d(x[:, None], k[:, :, :65], padding=(32, 0)) + 0.3
        return torch.nn.functional.conv2d(torch.tanh(y), k[:, :, :67], padding=(33, 0))[:, 0]

    def sequence_stage(feat, emb):
        fwd = torch.cumsum(feat, dim=1)
        bwd = torch.flip(torch.cumsum(torch.flip(feat, dims=(1,)), dim=1), dims=(1,))
        return torch.tanh(0.01 * (fwd + 0.5 * bwd) * emb[:, :1, None])

    whole = sequence_stage(conv_stage(spec[None]), dvec[None])[0]
    got = separate_long_exact(conv_stage, sequence_stage, spec, dvec, 301, 65, max_batch=3)
    assert torch.allclose(got, whole, atol=1e-6, rtol=0)
    if T_long > 301:
        restart = separate_long(lambda x, e: sequence_stage(conv_stage(x), e), spec, dvec, 301, 65)
        assert not torch.allclose(restart, whole, atol=1e-4)
        short_halo = separate_long_exact(conv_stage, sequence_stage, spec, dvec, 301, 20)
        assert not torch.allclose(short_halo, whole, atol=1e-6)


def test_exact_plan_stays_inside_the_clip_and_keeps_frames_away_from_window_edges():
    for n in (1, 200, 301, 302, 431, 1000, 3001):
        for halo in (0, 20, 65):
            plan = plan_windows_exact(n, 301, halo)
            wlen = min(301, n)
            pos = 0
            for k, (st, k0, k1) in enumerate(plan):
                assert 0 <= st and st + wlen <= n                     # inside the clip
                assert st + k0 == pos and k1 > k0                     # kept spans tile the clip in order
                assert st == 0 or k0 >= halo                          # >= halo from a window edge that is not the clip start
                assert st + wlen == n or k1 <= wlen - halo            # ... or the clip end
                pos = st + k1
            assert pos == n


def test_separate_long_many_equals_clip_by_clip():
    """The batch form of BASELINE configs[4] (all windows of all clips in one list) == separate_long per clip."""
    import torch
    from voicesplit_amd import streaming
    torch.manual_seed(0)
    lin = torch.nn.Linear(7, 7)

    def model(x, emb):                       # any per-window function of (x, emb)
        return torch.sigmoid(lin(x) + emb[:, :7].unsqueeze(1) + x.mean(dim=1, keepdim=True))

    specs = torch.rand(5, 83, 7)
    dvecs = torch.randn(5, 9)
    many = streaming.separate_long_many(model, specs, dvecs, window=20, max_batch=6)
    for i in range(5):
        one = streaming.separate_long(model, specs[i], dvecs[i], window=20, halo=0, max_batch=4)
        assert torch.allclose(many[i], one, atol=1e-6)
    assert many.shape == (5, 83, 7)
