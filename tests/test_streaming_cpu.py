"""CPU: the long-form windowing host logic (BASELINE configs[4]) with a stand-in model."""
import pytest
import torch

from voicesplit_amd.streaming import plan_windows, separate_long


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
