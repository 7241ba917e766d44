"""Sharding of one frame over the GPUs of a node: one process per GPU, torch.distributed
(backend "nccl" is RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The reference's only parallelism is rayon over pixels and over samples (src/main.rs:988, 994);
both are independent, the scene is read-only and small enough to replicate, so a frame shards
either by samples per pixel — every rank renders its own slice of each pixel's sample stream
into a private fp32 sum buffer, and ONE all-reduce(sum) of 3*W*H floats combines them — or by
interleaved image tiles, which needs no arithmetic collective at all (disjoint pixels).
"""
import copy


def shard_samples(params, rank, world, spp_total=None):
    """Sample split. `params.spp` is the whole frame's sample count unless spp_total is given;
    rank r renders samples [r*spp/world, (r+1)*spp/world) of every pixel's stream."""
    p = copy.copy(params)
    total = int(spp_total if spp_total is not None else params.spp)
    lo = (total * rank) // world
    hi = (total * (rank + 1)) // world
    p.first_sample = params.first_sample + lo
    p.spp = hi - lo
    p.spp_total = total
    return p


def weak_samples(params, rank, world):
    """Weak scaling: every rank renders params.spp samples, rank r taking the r-th block of the
    stream, so the reduced frame holds world * spp samples per pixel."""
    p = copy.copy(params)
    p.first_sample = params.first_sample + rank * params.spp
    p.spp_total = params.spp * world
    return p


def shard_tiles(params, rank, world, tile_size=64):
    """Tile split: square tiles of tile_size pixels, dealt round-robin over ranks (sky tiles are
    cheap, so interleaving balances the load)."""
    p = copy.copy(params)
    p.tile_size = tile_size
    p.tile_rank = rank
    p.tile_ranks = world
    return p


def reduce_frame_sum(buf, dist, dst=None, async_op=False):
    """The exchange step of the sample split: sum the ranks' fp32 [H][W][3] buffers in place.
    With dst=None every rank receives the sum (all-reduce); otherwise only rank dst does.
    async_op=True returns the collective's work handle (None when there is nothing to reduce):
    the reduction then runs behind whatever the caller queues next, and `handle.wait()` orders
    later work after it."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return None if async_op else buf
    if buf.is_cuda and dist.get_backend() == "gloo":
        # rehearsal on a box with fewer GPUs than ranks: gloo reduces host memory, so stage the
        # buffer through the host here (blocking) instead of leaving that to the backend
        host = buf.cpu()
        if dst is None:
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
        else:
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
        buf.copy_(host)
        return None if async_op else buf
    if dst is None:
        work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=async_op)
    else:
        work = dist.reduce(buf, dst=dst, op=dist.ReduceOp.SUM, async_op=async_op)
    return work if async_op else buf
