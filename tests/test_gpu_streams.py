"""Calls that follow each other on different streams, RT_FLAG_STORE, and the bound on the launch
log of a long pipelined run."""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import _ffi, scenes

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def spot():
    if rt.device_count() < 1:
        pytest.fail("no HIP device visible")
    s = scenes.build("spot")
    s.upload(0)
    return s


def test_consecutive_calls_on_different_streams(spot):
    """The second call resets counters and queues the first call's last kernels still read: it must
    start behind them although it is queued on another stream."""
    w, h, spp = 320, 240, 4
    want = np.zeros((h, w, 3))
    for i in range(4):
        spot.render(want, rt.render_params(width=w, height=h, spp=spp, first_sample=i * spp, seed=11))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    canvases = [torch.zeros((h, w, 3), dtype=torch.float64, device="cuda") for _ in range(4)]
    torch.cuda.synchronize()
    for i in range(4):  # no host wait in between; each frame into a canvas of its own
        spot.render_device(canvases[i].data_ptr(), rt.render_params(width=w, height=h, spp=spp, first_sample=i * spp, seed=11),
                           streams[i & 1].cuda_stream)
    spot.wait()
    torch.cuda.synchronize()
    total = np.zeros((h, w, 3))
    for c in canvases:  # summed in call order, as the host canvas was
        total += c.cpu().numpy()
    assert total.tobytes() == want.tobytes()


def test_store_flag_overwrites_the_pixels_it_renders(spot):
    w, h = 200, 136
    p = dict(width=w, height=h, spp=3, seed=4)
    want = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    spot.render_device(want.data_ptr(), rt.render_params(**p), 0)
    got = torch.full((h, w, 3), 7.5, dtype=torch.float64, device="cuda")
    spot.render_device(got.data_ptr(), rt.render_params(flags=_ffi.RT_FLAG_STORE, **p), 0)
    spot.wait()
    assert torch.equal(got, want)
    # tile shard: pixels of other ranks keep what they held
    part = torch.full((h, w, 3), 7.5, dtype=torch.float64, device="cuda")
    spot.render_device(part.data_ptr(), rt.render_params(flags=_ffi.RT_FLAG_STORE, tile_size=64, tile_rank=1, tile_ranks=3, **p), 0)
    spot.wait()
    mine = (part != 7.5).any(dim=2)
    assert 0 < int(mine.sum()) < w * h
    assert torch.equal(part[mine], want[mine])
    sums = torch.full((h, w, 3), 3.0, dtype=torch.float32, device="cuda")
    ref = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    spot.render_sum_f32_device(ref.data_ptr(), rt.render_params(**p), 0)
    spot.render_sum_f32_device(sums.data_ptr(), rt.render_params(flags=_ffi.RT_FLAG_STORE, **p), 0)
    spot.wait()
    assert torch.equal(sums, ref)


def test_launch_log_of_a_long_run_stays_bounded(spot):
    """Twelve hundred pipelined frames queue more launches than the log keeps one by one: the oldest are
    folded per kind, totals are kept, and the canvas is what the frames one by one give."""
    w, h, frames = 48, 32, 1200
    canvas = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    flags = _ffi.RT_FLAG_PIPELINE | _ffi.RT_FLAG_TIMING
    for i in range(frames):
        spot.render_device(canvas.data_ptr(), rt.render_params(width=w, height=h, spp=1, first_sample=i, seed=2, flags=flags), stream)
    spot.wait()
    st = spot.last_stats()
    log = spot.last_launches()
    assert len(log) <= 4096 + 5
    assert any(l.folded for l in log)
    n_first = sum(max(1, l.folded) for l in log if l.kind == 0)
    n_resolve = sum(max(1, l.folded) for l in log if l.kind == 4)
    assert n_first == frames and n_resolve == frames
    assert sum(l.rays for l in log) == st.rays
    want = np.zeros((h, w, 3))
    rays = 0
    for i in range(frames):
        rays += spot.render(want, rt.render_params(width=w, height=h, spp=1, first_sample=i, seed=2))
    assert rays == st.rays
    assert canvas.cpu().numpy().tobytes() == want.tobytes()
