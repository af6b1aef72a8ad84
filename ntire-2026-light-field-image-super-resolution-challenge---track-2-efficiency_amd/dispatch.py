"""Batched, patch-sharded replacement for the reference's per-patch test loop
(``test()`` train.py:286-347, copy at inference.py:172-221).

The reference crops a scene with ``LFdivide``, runs the model on ONE patch at a time (``minibatch_for_test=1``,
``torch.cuda.empty_cache()`` per patch, D2H copy per patch into a CPU tensor) and stitches with
``LFintegrate``.  Patches are independent and ``LFintegrate`` only centre-crops, so here:

* everything stays on the device, patches go through the model in minibatches,
* with W ranks (one process per GPU) rank r owns a contiguous block of the patch list; the only exchange is
  one all-gather of the SR patches (RCCL on GPUs; gloo in the CPU tests) before rank-local ``LFintegrate``.

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
    """LFdivide / LFintegrate on the MI355X through the C ABI."""

    @staticmethod
    def divide(lr, A, patch, stride):
        from lfsr_amd import capi
        return capi.lf_divide(lr.contiguous(), A, patch, stride)

    @staticmethod
    def integrate(sub, A, pz, stride, h, w):
        from lfsr_amd import capi
        return capi.lf_integrate(sub.contiguous(), A, pz, stride, h, w)


def sr_scene(net, lr_mosaic, A, scale, patch=32, stride=16, minibatch=32, ops=HipOps, group=None, data_info=None):
    """Super-resolve one full light field.

    lr_mosaic: (A*h0, A*w0) SAI mosaic (what ``Lr_SAI_y.squeeze()`` is at train.py:295).
    Returns the (A, A, h0*scale, w0*scale) tensor ``LFintegrate`` returns at train.py:317.
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
    if world > 1:
        # equal-size all-gather: pad every shard to the largest one (sizes differ by at most one patch)
        cap = (n + world - 1) // world
        buf = mine.new_zeros((cap, 1, pz, pz))
        buf[:mine.shape[0]] = mine
        gathered = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(gathered, buf, group=group)
        parts = []
        for r in range(world):
            a, b = shard_range(n, r, world)
            parts.append(gathered[r][:b - a])
        full = torch.cat(parts, 0)
    else:
        full = mine
    full = full.reshape(numU, numV, pz, pz)
    return ops.integrate(full, A, patch * scale, stride * scale, h0 * scale, w0 * scale)
