"""Data-parallel glue: batches shard over ranks (one process per GPU), the only
collective on the path is ONE all-reduce of the flat fp32 gradient per training
step (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).
SURVEY.md 8e.  Rollouts/evaluation need no collective."""
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def grad_parameters(model):
    """Parameters that take part in the bucket.  Parameters that never receive a
    gradient (decoder._context_proj for TSP/VRP, decoder._first_node for IRP) have
    .grad None on every rank and are skipped — Adam skips them too."""
    return [p for p in model.parameters() if p.grad is not None]


def flatten_grads(params):
    return torch.cat([p.grad.reshape(-1) for p in params])


def unflatten_grads(flat, params):
    off = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n


def allreduce_gradients(model):
    """mean over ranks of the flat gradient: one bucket, one collective (4.6 MB)."""
    if not is_distributed():
        return
    params = grad_parameters(model)
    if not params:
        return
    flat = flatten_grads(params)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    unflatten_grads(flat, params)


def gather_costs(cur, base):
    """The baseline-replacement decision must be identical on all ranks: every rank
    sees the costs of the whole batch (2*B floats, off the hot loop)."""
    if not is_distributed():
        return cur, base
    world = dist.get_world_size()
    both = torch.stack([cur, base]).contiguous()
    out = [torch.empty_like(both) for _ in range(world)]
    dist.all_gather(out, both)
    allc = torch.cat([o[0] for o in out])
    allb = torch.cat([o[1] for o in out])
    return allc, allb


def broadcast_model(model, src=0):
    """Replica consistency at start / after load_state_dict (params + BN buffers)."""
    if not is_distributed():
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src)


def average_buffers(model):
    """BatchNorm running statistics are buffers, not gradients: every rank accumulates its
    own shard's statistics.  They are averaged whenever the weights are copied into the
    baseline or checkpointed, so replicas stay identical."""
    if not is_distributed():
        return
    world = dist.get_world_size()
    for b in model.buffers():
        if b.dtype.is_floating_point:
            dist.all_reduce(b.data, op=dist.ReduceOp.SUM)
            b.data.div_(world)
        else:  # num_batches_tracked: identical on all ranks by construction; keep rank 0's
            dist.broadcast(b.data, src=0)
