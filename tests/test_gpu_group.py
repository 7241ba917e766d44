"""rt_group_*: one frame over several members driven from one process. The box the GPU tests run
on has one GPU, so the members here share device 0 (the library allows that for exactly this
purpose); what is checked is the sharding arithmetic, the ordering by events, and the combining kernel."""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import _ffi, parallel, scenes

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

W, H, SPP = 200, 136, 7  # 200 = 3 tiles of 64 + a ragged one; 136 = 2 + a ragged one


def params(**kw):
    kw.setdefault("width", W)
    kw.setdefault("height", H)
    kw.setdefault("spp", SPP)
    return rt.render_params(seed=5, **kw)


@pytest.fixture(scope="module")
def spot():
    if rt.device_count() < 1:
        pytest.fail("no HIP device visible")
    s = scenes.build("spot")
    s.upload(0)
    return s


@pytest.fixture(scope="module")
def single(spot):
    c = np.zeros((H, W, 3))
    rays = spot.render(c, params())
    return c, rays


def shard_sums(spot, shards):
    """What each member must leave behind: the f32 sum buffer of its shard, from the plain entry point."""
    out, rays = [], 0
    stream = torch.cuda.current_stream().cuda_stream
    for p in shards:
        buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        spot.render_sum_f32_device(buf.data_ptr(), p, stream)
        rays += spot.last_stats().rays
        out.append(buf.cpu().numpy().astype(np.float64))
    return out, rays


@pytest.mark.parametrize("n", [1, 2, 3, 8])
def test_sample_split_is_the_sum_of_the_members_shards(spot, single, n):
    want_single, rays_single = single
    g = rt.Group(spot, [0] * n)
    assert len(g) == n
    canvas = np.full((H, W, 3), 0.25)
    rays = g.render(canvas, params(), _ffi.RT_SPLIT_SAMPLES)
    sums, rays_shards = shard_sums(spot, [parallel.shard_samples(params(), r, n) for r in range(n)])
    acc = np.zeros((H, W, 3))
    for s in sums:  # member order, binary64
        acc += s
    want = np.full((H, W, 3), 0.25) + acc / SPP
    assert canvas.tobytes() == want.tobytes()
    assert rays == rays_shards == rays_single  # the sample streams do not depend on who renders them
    np.testing.assert_allclose(canvas - 0.25, want_single, rtol=1e-6, atol=1e-9)
    info = [g.member_info(i) for i in range(n)]
    assert [d for d, _, _ in info] == [0] * n and sum(r for _, r, _ in info) == rays
    assert all(ms > 0 for _, r, ms in info if r)


@pytest.mark.parametrize("n,tile", [(2, 0), (3, 64), (5, 8), (4, 128)])
def test_tile_split_equals_one_gpu(spot, single, n, tile):
    want_single, rays_single = single
    g = rt.Group(spot, [0] * n)
    canvas = np.zeros((H, W, 3))
    rays = g.render(canvas, params(tile_size=tile), _ffi.RT_SPLIT_TILES)
    assert rays == rays_single
    # every pixel comes from exactly one member, rounded to binary32 once
    sums, _ = shard_sums(spot, [params()])
    assert canvas.tobytes() == (sums[0] / SPP).tobytes()
    np.testing.assert_allclose(canvas, want_single, rtol=1e-6, atol=1e-9)
    busy = [g.member_info(i)[1] for i in range(n)]
    assert all(b > 0 for b in busy) or tile == 128  # 128-pixel tiles: 4 tiles over 4 members, all busy too
    assert sum(busy) == rays


def test_more_members_than_samples_and_a_device_canvas(spot):
    g = rt.Group(spot, [0, 0, 0, 0])
    p = params(spp=2)
    c = np.zeros((H, W, 3))
    want_rays = spot.render(c, p)
    dc = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    g.render_device(dc.data_ptr(), p, _ffi.RT_SPLIT_SAMPLES, torch.cuda.current_stream().cuda_stream)
    rays = g.wait()
    assert rays == want_rays
    assert sorted(g.member_info(i)[1] > 0 for i in range(4)) == [False, False, True, True]
    np.testing.assert_allclose(dc.cpu().numpy(), c, rtol=1e-6, atol=1e-9)
    # the group can be used again, with another frame size
    small = np.zeros((40, 56, 3))
    ref = np.zeros((40, 56, 3))
    spot.render(ref, params(width=56, height=40))
    g.render(small, params(width=56, height=40), _ffi.RT_SPLIT_TILES)
    np.testing.assert_allclose(small, ref, rtol=1e-6, atol=1e-9)


def test_group_outlives_its_scene_and_rejects_bad_frames(spot):
    s = scenes.build("sphere_floor")
    g = rt.Group(s, [0, 0])
    ref = np.zeros((H, W, 3))
    s.render(ref, params())
    del s
    c = np.zeros((H, W, 3))
    g.render(c, params())
    np.testing.assert_allclose(c, ref, rtol=1e-6, atol=1e-9)
    for bad, split in [(params(tile_rank=1, tile_ranks=2, tile_size=64), _ffi.RT_SPLIT_TILES), (params(tile_size=12), _ffi.RT_SPLIT_TILES),
                       (params(), 7), (params(max_depth=-1), _ffi.RT_SPLIT_SAMPLES)]:
        before = c.copy()
        with pytest.raises(_ffi.RtError):
            g.render(c, bad, split)
        assert c.tobytes() == before.tobytes()
    with pytest.raises(_ffi.RtError):
        rt.Group(spot, [0, rt.device_count()])


def test_a_member_the_first_device_cannot_reach_is_refused(monkeypatch):
    """The failure path of rt_group_create (no peer access), injected: this box has one GPU."""
    s = scenes.build("sphere_floor")
    monkeypatch.setenv("RT_GROUP_NO_PEER", "1")
    monkeypatch.setenv("RT_GROUP_LOG", "1")
    with pytest.raises(_ffi.RtError) as e:
        rt.Group(s, [0, 0])
    assert e.value.code == _ffi.RT_ERR_HIP and "cannot access" in str(e.value)
    monkeypatch.delenv("RT_GROUP_NO_PEER")
    assert len(rt.Group(s, [0, 0])) == 2


@pytest.mark.parametrize("split,n", [(_ffi.RT_SPLIT_SAMPLES, 3), (_ffi.RT_SPLIT_TILES, 4)])
def test_frames_queued_back_to_back_equal_frames_one_by_one(spot, split, n):
    """rt_group_render_device only queues: the members render frame k + 1 (into their other sum
    buffer) while the first device combines frame k, ordered by events alone. A run of frames
    queued without a wait in between must leave the bits that the same frames leave when each is
    waited for — and the canvas is ordered by the caller's stream: a clear queued on it before
    the first frame, and a copy queued on it after the last, need no synchronisation."""
    frames = 6
    ps = [params(first_sample=k * SPP, fov_y_deg=40.0 + 15.0 * k) for k in range(frames)]  # frames that differ in work
    one_by_one = rt.Group(spot, [0] * n)
    want = np.zeros((H, W, 3))
    rays_want = 0
    for p in ps:
        rays_want += one_by_one.render(want, p, split)
    g = rt.Group(spot, [0] * n)
    side = torch.cuda.Stream()
    dc = torch.full((H, W, 3), 7.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        dc.zero_()  # on the caller's stream: the first frame's combination must come after it
        for p in ps:
            g.render_device(dc.data_ptr(), p, split, side.cuda_stream)
        got = dc.clone()  # on the caller's stream, behind the last combination
    side.synchronize()
    rays = g.wait()
    assert rays == rays_want
    assert sum(g.member_info(i)[1] for i in range(n)) == rays
    # the host-canvas entry point adds frame by frame in binary64 on the host, the device one on the
    # device: the same additions in the same order
    assert got.cpu().numpy().tobytes() == want.tobytes()
    assert g.wait() == 0  # nothing queued since
    # a frame of another size afterwards (the sum buffers grow), then the first size again
    small = torch.zeros((40, 56, 3), dtype=torch.float64, device="cuda")
    ref = np.zeros((40, 56, 3))
    one_by_one.render(ref, params(width=56, height=40), split)
    with torch.cuda.stream(side):
        g.render_device(small.data_ptr(), params(width=56, height=40), split, side.cuda_stream)
        g.render_device(dc.data_ptr(), ps[0], split, side.cuda_stream)
    g.wait()
    assert small.cpu().numpy().tobytes() == ref.tobytes()
