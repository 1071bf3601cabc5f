"""Data-parallel glue: batches shard over ranks (one process per GPU), the only
collective on the path is ONE all-reduce of the flat fp32 gradient per training
step (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).
SURVEY.md 8e.  Rollouts/evaluation need no collective."""
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank():
    return dist.get_rank() if is_distributed() else 0


def world_size():
    return dist.get_world_size() if is_distributed() else 1


# bench.py sets this to a list to collect (start, end) device events around every gradient
# all-reduce (the only collective on the training path); None = no timing
ALLREDUCE_EVENTS = None


def global_means(*scalars):
    """Host floats of per-shard 0-d tensors averaged over ranks (equal shard sizes, so the
    mean of local means is the global mean): what rank 0 logs (graph_tsp_agent.py:191-206)."""
    t = torch.stack([torch.as_tensor(s, dtype=torch.float32).detach().reshape(())
                     .to(scalars[0].device) for s in scalars])
    if is_distributed():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t = t / dist.get_world_size()
    return tuple(t.tolist())


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


def _bucket_in_place(model):
    """The model's persistent flat gradient buffer if every .grad is its view of it (what the
    HIP backward leaves behind, agents/backward.py), else None."""
    hit = getattr(model, "_grad_bucket", None)
    if hit is None:
        return None
    flat, views, _ = hit
    by_param = {id(p): p for p in model.parameters()}
    for i, v in views.items():
        g = by_param[i].grad
        if g is None or g.data_ptr() != v.data_ptr() or g.numel() != v.numel():
            return None
    if any(p.grad is not None and id(p) not in views for p in model.parameters()):
        return None
    return flat


def allreduce_gradients(model):
    """mean over ranks of the flat gradient: one bucket, one collective (4.6 MB).  In place on
    the persistent bucket the backward pass wrote into (RCCL: ReduceOp.AVG, nothing else
    touches the gradient); a gradient that arrived some other way is packed and unpacked."""
    if not is_distributed():
        return
    flat = _bucket_in_place(model)
    params = None
    if flat is None:
        params = grad_parameters(model)
        if not params:
            return
        flat = flatten_grads(params)
    ev = None
    if ALLREDUCE_EVENTS is not None and flat.is_cuda:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if dist.get_backend() == "nccl":
        dist.all_reduce(flat, op=dist.ReduceOp.AVG)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(dist.get_world_size())
    if ev is not None:
        ev[1].record()
        ALLREDUCE_EVENTS.append(ev)
    if params is not None:
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
    """Replica consistency at start / after load_state_dict (params + BN buffers).  The
    in-place write bumps every parameter's version counter, and the folded decoder matrices
    derived from the old values are dropped explicitly as well."""
    if not is_distributed():
        return
    from . import runtime
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            buf = t.detach().clone()
            dist.broadcast(buf, src=src)
            t.copy_(buf)
    for m in model.modules():
        runtime.invalidate(m)


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
