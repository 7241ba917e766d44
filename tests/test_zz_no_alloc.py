"""rt_render_device_noalloc: the render call out of the library's static work memory must leave
the same bits in the canvas as rt_render_device, call after call (every call on a GPU shares the
one array, which keeps the last call's contents), must reject what does not fit, and must not
change the amount of free device memory."""
import ctypes as C

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import _ffi, scenes

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _scene(name):
    if rt.device_count() < 1:
        pytest.fail("no HIP device visible")
    s = scenes.build(name)
    s.upload(0)
    return s


def _emissive_scene():
    """An emissive floor under an emissive sphere and a glass sphere (emitted radiance rides in the queues)."""
    floor = rt.Opaque((0.8, 0.7, 0.6), 0.0, 1.0, 0.0, emission=(0.3, 0.2, 0.1))
    ball = rt.Opaque((0.5, 0.5, 0.9), 0.2, 0.8, 0.5, emission=(0.0, 0.1, 0.4))
    tv = np.array([[[30, -0.7, -30], [-30, -0.7, -30], [0, -0.7, 5]]], dtype=np.float64)
    tn = np.tile(np.array([0, 1.0, 0]), (1, 3, 1))
    s = rt.Scene.from_desc([_ffi.RT_OBJ_MESH, _ffi.RT_OBJ_SPHERE, _ffi.RT_OBJ_SPHERE], [floor, ball, rt.Dielectric(1.5)], tv, tn,
                           np.zeros((1, 3, 2)), [0], [0], [[0.5, 0, -3, 0.7], [-0.7, -0.2, -2.2, 0.5]], [1, 2])
    s.upload(0)
    return s


def _pair(scene, **kw):
    """(canvas of rt_render_device, canvas of rt_render_device_noalloc, free bytes before/after the latter)"""
    p = rt.render_params(**kw)
    h, w = p.height, p.width
    stream = torch.cuda.current_stream().cuda_stream
    want = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    scene.render_device(want.data_ptr(), p, stream)
    scene.wait()
    rays = scene.last_stats().rays
    got = torch.full((h, w, 3), 0.25, dtype=torch.float64, device="cuda")  # the call adds into the canvas
    want += 0.25
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    scene.render_device_noalloc(got.data_ptr(), p, stream)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert scene.last_stats().rays == rays
    return want.cpu().numpy(), got.cpu().numpy(), free0, free1


def test_two_calls_in_a_row_second_smaller():
    spot = _scene("spot")
    # the bigger call first: its queue records and counters are still in the array when the smaller runs
    want, got, f0, f1 = _pair(spot, width=160, height=120, spp=6, seed=5)
    assert got.tobytes() == want.tobytes()
    assert f1 == f0, "the call changed the amount of free device memory"
    want, got, f0, f1 = _pair(spot, width=72, height=40, spp=2, seed=9)
    assert got.tobytes() == want.tobytes()
    assert f1 == f0
    # and the first again, after the smaller one
    want, got, _, _ = _pair(spot, width=160, height=120, spp=6, seed=5)
    assert got.tobytes() == want.tobytes()


def test_several_sample_passes_emitters_and_tiles():
    # 640x480 at 24 spp is 7.4 M paths: more than the array holds at once, so the call runs in passes
    # with running per-pixel totals
    lens = _scene("lens")
    want, got, f0, f1 = _pair(lens, width=640, height=480, spp=24, seed=2)
    assert got.tobytes() == want.tobytes()
    assert f1 == f0
    # a tile shard of a frame (what a caller does with a frame that is too large for one call)
    want, got, _, _ = _pair(lens, width=200, height=136, spp=3, seed=2, tile_size=64, tile_rank=1, tile_ranks=3)
    assert got.tobytes() == want.tobytes()
    want, got, _, _ = _pair(_emissive_scene(), width=320, height=200, spp=5, seed=6)
    assert got.tobytes() == want.tobytes()
    # the single-kernel forms use the array too
    want, got, _, _ = _pair(lens, width=96, height=64, spp=2, seed=4, flags=_ffi.RT_FLAG_MEGAKERNEL)
    assert got.tobytes() == want.tobytes()


def test_rejects_what_does_not_fit_and_scenes_not_uploaded():
    spot = _scene("spot")
    canvas = torch.zeros((8, 8, 3), dtype=torch.float64, device="cuda")
    p = rt.render_params(width=3840, height=2160, spp=1)  # 8.3 M pixels at once: beyond the array
    with pytest.raises(_ffi.RtError) as e:
        spot.render_device_noalloc(canvas.data_ptr(), p, 0)
    assert e.value.code == _ffi.RT_ERR_INVALID and "static work memory" in str(e.value)
    fresh = scenes.build("sphere_floor")  # never uploaded: uploading would allocate
    with pytest.raises(_ffi.RtError) as e:
        fresh.render_device_noalloc(canvas.data_ptr(), rt.render_params(width=8, height=8, spp=1), 0)
    assert e.value.code == _ffi.RT_ERR_NO_DEVICE
    # the largest frame the header promises (1280x720) fits
    big = torch.zeros((720, 1280, 3), dtype=torch.float64, device="cuda")
    spot.render_device_noalloc(big.data_ptr(), rt.render_params(width=1280, height=720, spp=1), 0)
    torch.cuda.synchronize()
    assert float(big.sum()) > 0.0


def test_a_frame_captured_into_a_graph_replays_the_same_bits():
    """The allocation-free call under HIP graph capture (what it exists for): it makes no waiting
    call, whether the frame's shape has been seen before or not, so a frame can be captured on the
    caller's stream and replayed. Every replay adds the same frame to the canvas, bit for bit what
    rt_render_device adds."""
    spot = _scene("spot")
    side = torch.cuda.Stream()
    for shape, (w, h, spp) in {"a shape never rendered before": (136, 88, 3), "a shape rendered before": (160, 120, 4)}.items():
        p = rt.render_params(width=w, height=h, spp=spp, seed=11)
        want = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
        got = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
        with torch.cuda.stream(side):
            stream = side.cuda_stream
            # the scene's last call must have been on the capturing stream (a call on another stream
            # starts behind an event of the previous one, which a capture cannot wait for)
            warm = rt.render_params(width=32, height=24, spp=1, seed=1)
            spot.render_device(want.data_ptr(), warm, stream)
            if shape == "a shape rendered before":
                spot.render_device(want.data_ptr(), p, stream)
            spot.wait()
            want.zero_()
            for _ in range(2):
                spot.render_device(want.data_ptr(), p, stream)
            spot.wait()
            rays = spot.last_stats().rays
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                spot.render_device_noalloc(got.data_ptr(), p, torch.cuda.current_stream().cuda_stream)
            side.synchronize()
            assert float(got.abs().sum()) == 0.0, "capture must not run the frame"
            g.replay()
            g.replay()
            side.synchronize()
        torch.cuda.synchronize()
        assert got.cpu().numpy().tobytes() == want.cpu().numpy().tobytes(), shape
        assert spot.last_stats().rays == rays, shape
