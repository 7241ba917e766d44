"""RT_FLAG_PIPELINE: a run of calls whose tails and canvas sums overlap the next call must leave
the same bits in the canvas, and the same totals in the counters, as the calls made one by one."""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import _ffi, scenes

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

W, H, SPP, ITS = 96, 72, 3, 5


def frame(i, flags=0, **kw):
    return rt.render_params(width=W, height=H, spp=SPP, first_sample=i * SPP, seed=3, flags=flags, **kw)


@pytest.fixture(scope="module")
def spot():
    if rt.device_count() < 1:
        pytest.fail("no HIP device visible")
    s = scenes.build("spot")
    s.upload(0)
    return s


@pytest.fixture(scope="module")
def one_by_one(spot):
    c = np.zeros((H, W, 3))
    rays = [spot.render(c, frame(i)) for i in range(ITS)]
    return c, rays


@pytest.mark.parametrize("env", [{}, {"RT_TAIL_THRESHOLD": "500"}, {"RT_WORK_BUFFER_MIB": "1"},
                                 {"RT_WORK_BUFFER_MIB": "1", "RT_TAIL_THRESHOLD": "200"}, {"RT_NO_OVERLAP": "1"}])
def test_pipelined_run_equals_calls_one_by_one(spot, one_by_one, monkeypatch, env):
    want, rays = one_by_one
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    canvas = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for i in range(ITS):
        spot.render_device(canvas.data_ptr(), frame(i, _ffi.RT_FLAG_PIPELINE | _ffi.RT_FLAG_TIMING), stream)
    spot.wait()
    assert canvas.cpu().numpy().tobytes() == want.tobytes()
    st = spot.last_stats()  # the run's totals
    assert st.rays == sum(rays)
    log = spot.last_launches()
    assert sum(1 for l in log if l.kind == 0) >= ITS and sum(1 for l in log if l.kind == 4) >= ITS
    assert sum(l.rays for l in log) == sum(rays)
    tm = spot.last_timing()
    assert tm.path_launches + tm.resolve_launches == len(log) and tm.ms_paths > 0.0


def test_run_ends_at_a_plain_call_and_at_a_stats_call(spot, one_by_one):
    want, rays = one_by_one
    canvas = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    spot.render_device(canvas.data_ptr(), frame(0, _ffi.RT_FLAG_PIPELINE), stream)
    spot.render_device(canvas.data_ptr(), frame(1, _ffi.RT_FLAG_PIPELINE), stream)
    assert spot.last_stats().rays == rays[0] + rays[1]  # waits, ends the run
    spot.render_device(canvas.data_ptr(), frame(2, _ffi.RT_FLAG_PIPELINE), stream)
    spot.render_device(canvas.data_ptr(), frame(3), stream)  # plain: ordered after the run, on the caller's stream
    torch.cuda.current_stream().synchronize()
    assert spot.last_stats().rays == rays[3]
    spot.render_device(canvas.data_ptr(), frame(4, _ffi.RT_FLAG_PIPELINE | _ffi.RT_FLAG_COUNT_WORK), stream)
    st = spot.last_stats()
    assert st.rays == rays[4] and st.paths == W * H * SPP and st.node_visits > st.rays
    assert canvas.cpu().numpy().tobytes() == want.tobytes()


def test_pipelined_f32_sums_and_host_entry_point(spot, one_by_one):
    want, rays = one_by_one
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    ref = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for i in range(ITS):
        spot.render_sum_f32_device(ref.data_ptr(), frame(i), stream)
        spot.render_sum_f32_device(acc.data_ptr(), frame(i, _ffi.RT_FLAG_PIPELINE), stream)
    spot.wait()
    assert torch.equal(acc, ref)
    c = np.zeros((H, W, 3))  # the blocking entry point accepts the flag and still returns a finished canvas
    assert [spot.render(c, frame(i, _ffi.RT_FLAG_PIPELINE)) for i in range(ITS)] == rays
    assert c.tobytes() == want.tobytes()


def test_stream_wait_orders_another_stream_behind_a_pipelined_frame(spot):
    """rt_scene_stream_wait: what a second stream reads after the wait is the finished frame, although
    the host never waited and the run goes on (the multi-GPU loop queues each frame's reduction so)."""
    big = dict(width=640, height=480, spp=4, seed=3)
    ref = torch.zeros((480, 640, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    spot.render_sum_f32_device(ref.data_ptr(), rt.render_params(**big), stream)
    spot.wait()
    other = torch.cuda.Stream()
    copies = []
    bufs = [torch.zeros((480, 640, 3), dtype=torch.float32, device="cuda") for _ in range(4)]  # alive until the run ends
    for buf in bufs:
        spot.render_sum_f32_device(buf.data_ptr(), rt.render_params(flags=_ffi.RT_FLAG_PIPELINE, **big), stream)
        with torch.cuda.stream(other):
            spot.stream_wait(other.cuda_stream)
            copies.append(buf.clone())  # runs on `other`, behind the frame's tail and sum
    other.synchronize()
    for c in copies:
        assert torch.equal(c, ref)
    spot.wait()
    assert spot.last_stats().rays > 0
    fresh = scenes.build("sphere_floor")
    fresh.stream_wait(0)  # nothing uploaded, nothing queued: returns at once


@pytest.mark.parametrize("flags", [0, _ffi.RT_FLAG_PIPELINE])
def test_frames_unlike_the_one_their_shape_was_learnt_on(monkeypatch, flags):
    """The launches of a frame are sized from the first frame of its shape (size, spp, depth cap,
    thresholds: not the camera). A later frame of that shape whose queues are several times longer,
    or much shorter, than the learnt ones must still render every path: the looping launches cover
    what the sized grids miss, and the tail takes whatever the learnt number of bounces left."""
    monkeypatch.setenv("RT_TAIL_THRESHOLD", "200")
    fovs = (150.0, 25.0, 60.0, 170.0, 8.0)  # mostly sky first, then mostly geometry, ...
    params = [rt.render_params(width=W, height=H, spp=SPP, first_sample=i * SPP, seed=9, fov_y_deg=f, flags=flags) for i, f in enumerate(fovs)]
    # reference: the host watches every frame's queue counts
    monkeypatch.setenv("RT_SYNC_ROUNDS", "1")
    ref = scenes.build("spot")
    ref.upload(0)
    want, rays = [], []
    for p in params:
        c = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
        ref.render_device(c.data_ptr(), p, torch.cuda.current_stream().cuda_stream)
        ref.wait()
        rays.append(ref.last_stats().rays)
        want.append(c.cpu().numpy().tobytes())
    assert max(rays) > 1.5 * min(rays)  # the frames do differ in work
    monkeypatch.delenv("RT_SYNC_ROUNDS")
    s = scenes.build("spot")
    s.upload(0)
    for p, w, r in zip(params, want, rays):
        c = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
        s.render_device(c.data_ptr(), p, torch.cuda.current_stream().cuda_stream)
        s.wait()
        assert s.last_stats().rays == r
        assert c.cpu().numpy().tobytes() == w
