"""One data-parallel optimisation step (the hot loop of the reference's ``train()`` train.py:243-268, fp32) with the
one exchange the path has: a single all-reduce of the flat gradient bucket the HIP backward fills (RCCL over xGMI on
GPUs -- backend "nccl" on ROCm; gloo in the CPU tests).  The reference has no gradient exchange at all (single
process); everything else follows it: L1 loss, ``clip_grad_norm_(1.0)`` (train.py:266), AdamW step.
"""
import torch
import torch.distributed as dist


_CAPI_COMM = {}


def _capi_comm(group):
    """RCCL communicator created through the C ABI (lfsr_comm_init): rank 0's unique id travels over the existing process group."""
    key = id(group)
    if key not in _CAPI_COMM:
        from lfsr_amd import capi
        world, rank = dist.get_world_size(group), dist.get_rank(group)

        def exchange(raw):
            box = [raw if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            return box[0]
        _CAPI_COMM[key] = capi.RcclComm(world, rank, exchange)
    return _CAPI_COMM[key]


def allreduce_bucket(bucket, group=None):
    """Average a flat gradient bucket over the data-parallel group in ONE collective: torch.distributed's backend ("nccl" = RCCL on ROCm;
    gloo in the CPU tests) by default, or -- LFSR_ALLREDUCE=capi, GPU buckets only -- lfsr_allreduce of the C ABI on the current stream."""
    if not (dist.is_available() and dist.is_initialized()):
        return bucket
    world = dist.get_world_size(group)
    if world > 1:
        import os
        if os.environ.get("LFSR_ALLREDUCE", "") == "capi" and bucket.is_cuda:
            _capi_comm(group).allreduce_(bucket)
        else:
            dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
        bucket.div_(world)
    return bucket


def clip_bucket_(bucket, max_norm):
    """``torch.nn.utils.clip_grad_norm_`` (train.py:266) on the flat bucket every ``.grad`` is a view of: the total 2-norm over all parameters is the norm of the bucket, and the
    scale is torch's own ``max_norm / (total_norm + 1e-6)`` clamped to 1 -- three launches instead of one norm per parameter plus the foreach scaling (~25 launches at 137
    tensors); no host synchronisation."""
    total = torch.linalg.vector_norm(bucket)
    bucket.mul_((max_norm / (total + 1e-6)).clamp(max=1.0))
    return total


def bind_grads_to_bucket(net):
    """Make every parameter's .grad a view of net.grad_bucket (state_dict order) so the collective and the optimizer
    see the same memory."""
    for name, p in net.named_parameters():
        off, n = net._spans[name]
        p.grad = net.grad_bucket[off:off + n].view_as(p)


def train_step(net, criterion, optimizer, x, label, group=None, max_norm=1.0, data_info=None):
    optimizer.zero_grad(set_to_none=True)
    out = net(x, data_info)                     # train.py:257
    loss = criterion(out, label, data_info)     # train.py:258
    loss.backward()                             # train.py:264 -> HIP backward fills net.grad_bucket
    allreduce_bucket(net.grad_bucket, group)
    bind_grads_to_bucket(net)
    clip_bucket_(net.grad_bucket, max_norm)     # train.py:266 (clip_grad_norm_ over all parameters = the norm of the flat bucket they are views of)
    optimizer.step()
    return loss.detach(), out.detach()


def broadcast_parameters(net, group=None, src=0):
    """Identical replicas at start: rank ``src``'s weights to everyone."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        with torch.no_grad():
            for p in net.parameters():
                dist.broadcast(p, src=src, group=group)
    # a collective writes through the storage without bumping p._version, which is what the HIP runtimes key their packed copies on
    if hasattr(net, "invalidate_packed"):
        net.invalidate_packed()
