"""Batched, patch-sharded replacement for the reference's per-patch test loop
(``test()`` train.py:286-347, copy at inference.py:172-221).

The reference crops a scene with ``LFdivide``, runs the model on ONE patch at a time (``minibatch_for_test=1``,
``torch.cuda.empty_cache()`` per patch, D2H copy per patch into a CPU tensor) and stitches with
``LFintegrate``.  Patches are independent and ``LFintegrate`` only centre-crops, so here:

* everything stays on the device, patches go through the model in minibatches,
* with W ranks (one process per GPU) rank r owns a contiguous block of the patch list and crops its own SR patches to the
  tiles ``LFintegrate`` keeps; the only exchange is one gather (or all-gather) of those tiles -- a quarter of the SR bytes
  (RCCL on GPUs; gloo in the CPU tests) -- and the receiving rank places them (``LFintegrate`` = place o crop).

The tensor ops are injected (``ops``) so the host logic is testable without a GPU; the product passes
``HipOps`` (C-ABI kernels, no CPU fallback).
"""
import torch


def shard_range(n, rank, world):
    """Contiguous block [lo, hi) of n items owned by ``rank`` (sizes differ by at most one)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class HipOps:
    """LFdivide / LFintegrate (and its crop / place factors) on the MI355X through the C ABI."""

    @staticmethod
    def divide(lr, A, patch, stride):
        from lfsr_amd import capi
        return capi.lf_divide(lr.contiguous(), A, patch, stride)

    @staticmethod
    def integrate(sub, A, pz, stride, h, w):
        from lfsr_amd import capi
        return capi.lf_integrate(sub.contiguous(), A, pz, stride, h, w)

    @staticmethod
    def crop(sub, A, pz, stride):
        from lfsr_amd import capi
        return capi.lf_crop_tiles(sub, A, pz, stride)

    @staticmethod
    def place(tiles, out, A, numU, numV, first, stride):
        from lfsr_amd import capi
        return capi.lf_place_tiles(tiles, out, A, numU, numV, first, stride)


def sr_scene(net, lr_mosaic, A, scale, patch=32, stride=16, minibatch=32, ops=HipOps, group=None, data_info=None, dst=None):
    """Super-resolve one full light field.

    lr_mosaic: (A*h0, A*w0) SAI mosaic (what ``Lr_SAI_y.squeeze()`` is at train.py:295).
    Returns the (A, A, h0*scale, w0*scale) tensor ``LFintegrate`` returns at train.py:317.

    With W > 1 ranks each rank runs its contiguous block of the patch list and CROPS its own SR patches to what ``LFintegrate``
    keeps (the centre stride x stride of every view: a quarter of the SR bytes), so the one exchange of the path moves
    n * A^2 * (stride*scale)^2 elements in total instead of n * (A*patch*scale)^2 per rank: ``dst=r`` -- a gather to rank r, which alone
    places the tiles and returns the scene (the others return None; the reference's ``test()`` consumes the result in one process);
    ``dst=None`` -- an all-gather of the tiles, every rank returns the scene.
    """
    import torch.distributed as dist
    world, rank = 1, 0
    if group is not None or (dist.is_available() and dist.is_initialized()):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    h0, w0 = lr_mosaic.shape[0] // A, lr_mosaic.shape[1] // A
    sub = ops.divide(lr_mosaic, A, patch, stride)                        # (numU, numV, A*P, A*P)
    numU, numV = sub.shape[:2]
    n = numU * numV
    sub = sub.reshape(n, 1, A * patch, A * patch)
    lo, hi = shard_range(n, rank, world)
    pz = A * patch * scale
    outs = []
    with torch.no_grad():
        for i in range(lo, hi, minibatch):
            outs.append(net(sub[i:min(i + minibatch, hi)], data_info))
    mine = torch.cat(outs, 0) if outs else sub.new_zeros((0, 1, pz, pz))
    H, Wd, S2 = h0 * scale, w0 * scale, stride * scale
    if world == 1:
        return ops.integrate(mine.reshape(numU, numV, pz, pz), A, patch * scale, S2, H, Wd)
    # equal-size exchange: every shard's tiles padded to the largest shard (sizes differ by at most one patch)
    cap = (n + world - 1) // world
    tiles = ops.crop(mine, A, patch * scale, S2)                          # (hi - lo, A, A, S2, S2)
    buf = tiles.new_zeros((cap, A, A, S2, S2))
    buf[:tiles.shape[0]] = tiles
    # RCCL moves device buffers; the gloo rehearsal backend (CPU tests, one-GPU rehearsals of bench.py) gathers host buffers only
    host_xchg = buf.is_cuda and dist.get_backend(group) == "gloo"
    xb = buf.cpu() if host_xchg else buf
    if dst is None:
        gathered = [torch.empty_like(xb) for _ in range(world)]
        dist.all_gather(gathered, xb, group=group)
    else:
        gathered = [torch.empty_like(xb) for _ in range(world)] if rank == dst else None
        dist.gather(xb, gathered, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
        if rank != dst:
            return None
    if host_xchg:
        gathered = [g.to(buf.device) for g in gathered]
    out = buf.new_empty((A, A, H, Wd))
    for r in range(world):
        a, b = shard_range(n, r, world)
        if b > a:
            ops.place(gathered[r][:b - a], out, A, numU, numV, a, S2)
    return out
