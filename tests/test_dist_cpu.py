"""The N>1 path on CPU: world size 2 over gloo. Ranks shard a frame with raytracer_amd.parallel
(sample split and tile split), render their share with the oracle standing in for the GPU, and
combine through the same reduction helper bench.py uses; the result must equal the unsharded
frame."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_ffi as orc
        import raytracer_amd as rt
        from raytracer_amd import parallel

        o = orc.OracleScene.build("sphere_floor")
        w, h, spp = 64, 40, 6
        base = rt.render_params(width=w, height=h, spp=spp, seed=4)

        # sample split: fp32 sum buffers, one all-reduce
        p = parallel.shard_samples(base, rank, world)
        part = np.zeros((h, w, 3))
        o.render(part, p, threads=2)
        buf = torch.from_numpy((part * p.spp_total).astype(np.float32))  # sums, as the GPU path writes them
        parallel.reduce_frame_sum(buf, dist)
        frame_samples = buf.numpy().astype(np.float64) / spp

        # weak scaling: every rank renders spp samples of consecutive blocks
        pw = parallel.weak_samples(base, rank, world)
        partw = np.zeros((h, w, 3))
        o.render(partw, pw, threads=2)
        bufw = torch.from_numpy(partw.copy())
        parallel.reduce_frame_sum(bufw, dist)

        # tile split: disjoint pixels, sum == union
        pt = parallel.shard_tiles(base, rank, world, tile_size=16)
        partt = np.zeros((h, w, 3))
        rays_t, _ = o.render(partt, pt, threads=2)
        buft = torch.from_numpy(partt.copy())
        parallel.reduce_frame_sum(buft, dist, dst=0)
        rays = torch.tensor([rays_t], dtype=torch.int64)
        dist.all_reduce(rays)
        if rank == 0:
            q.put((frame_samples, bufw.numpy(), buft.numpy(), int(rays[0]), (p.first_sample, p.spp, p.spp_total)))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_equals_single():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    frame_samples, frame_weak, frame_tiles, rays_tiles, shard0 = got

    sys.path.insert(0, HERE)
    import oracle_ffi as orc
    import raytracer_amd as rt

    o = orc.OracleScene.build("sphere_floor")
    w, h, spp = 64, 40, 6
    whole = np.zeros((h, w, 3))
    rays, _ = o.render(whole, rt.render_params(width=w, height=h, spp=spp, seed=4))
    assert shard0 == (0, 3, 6)
    # fp32 accumulation buffers: within fp32 rounding of the f64 frame
    assert np.allclose(frame_samples, whole, rtol=3e-7, atol=1e-7)
    whole12 = np.zeros((h, w, 3))
    o.render(whole12, rt.render_params(width=w, height=h, spp=2 * spp, seed=4))
    assert np.allclose(frame_weak, whole12, rtol=1e-13, atol=1e-15)
    assert frame_tiles.tobytes() == whole.tobytes()
    assert rays_tiles == rays


def test_shard_helpers_cover_every_sample_and_tile():
    sys.path.insert(0, ROOT)
    import raytracer_amd as rt
    from raytracer_amd import parallel

    base = rt.render_params(width=100, height=60, spp=10, first_sample=20)
    for world in (1, 2, 3, 4, 8):
        seen = []
        for r in range(world):
            p = parallel.shard_samples(base, r, world)
            assert p.spp_total == 10
            seen += list(range(p.first_sample, p.first_sample + p.spp))
        assert seen == list(range(20, 30))
        pw = [parallel.weak_samples(base, r, world) for r in range(world)]
        assert [p.first_sample for p in pw] == [20 + 10 * r for r in range(world)]
        assert all(p.spp == 10 and p.spp_total == 10 * world for p in pw)
        pt = [parallel.shard_tiles(base, r, world) for r in range(world)]
        assert all(p.tile_size == 64 and p.tile_ranks == world and p.tile_rank == r for r, p in enumerate(pt))


def test_strong_sample_split_of_the_1024_spp_frame(oracle):
    """BASELINE.json configs[3]: 1024 samples per pixel divided over 2, 4 and 8 GPUs as 512, 256 and
    128 per GPU (bench.py --split samples_strong). The shards are exactly those blocks of each
    pixel's sample stream, and rendering the blocks separately and adding the sums gives the frame
    (here on the CPU restatement, at a size it finishes in seconds)."""
    sys.path.insert(0, ROOT)
    import numpy as np

    import raytracer_amd as rt
    from raytracer_amd import parallel

    frame = rt.render_params(width=3840, height=2160, spp=1024, first_sample=0)
    for world, each in ((2, 512), (4, 256), (8, 128)):
        shards = [parallel.shard_samples(frame, r, world) for r in range(world)]
        assert [(p.first_sample, p.spp, p.spp_total) for p in shards] == [(r * each, each, 1024) for r in range(world)]
    o = oracle.OracleScene.build("sphere_floor")
    small = rt.render_params(width=24, height=16, spp=16, first_sample=32, seed=9)
    whole = np.zeros((16, 24, 3))
    rays, _ = o.render(whole, small)
    parts = np.zeros((16, 24, 3))
    total = 0
    for r in range(8):
        p = parallel.shard_samples(small, r, 8)
        assert p.spp == 2 and p.first_sample == 32 + 2 * r
        n, _ = o.render(parts, p)  # adds (sum of the shard's samples) / spp_total
        total += n
    assert total == rays
    assert np.max(np.abs(parts - whole)) < 1e-12
