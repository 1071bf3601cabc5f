"""Gradient of the sampled tour's log-probability (REINFORCE, graph_tsp_agent.py:178-186).

`loss.backward()` of the reference's training loop lands here: the autograd node below has
the model parameters as inputs and the HIP rollout's acc_log_prob as output; its backward
runs the hand-written HIP backward (vrp_decoder_backward over the recorded episode, then
vrp_encoder_backward over the tape the training rollout kept) and hands every parameter its
gradient.  No torch math is involved and nothing runs on the CPU.
"""
import torch

from . import runtime


class _TourLogProb(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, kind, res, *params):
        ctx.model, ctx.kind, ctx.res = model, kind, res
        ctx.param_ids = [id(p) for p in params]
        return res.acc_logp.clone()

    @staticmethod
    def backward(ctx, d_logp):
        model, kind, res = ctx.model, ctx.kind, ctx.res
        T = res.T
        loads = None if res.load_trace is None else res.load_trace[:T]
        # Fresh gradients (every .grad None, i.e. right after opt.zero_grad()): the kernels
        # write into the model's persistent flat bucket and each .grad becomes its view of it,
        # so the data-parallel all-reduce needs neither a cat nor a copy-back.  Otherwise
        # (somebody accumulates over several backward calls) autograd gets ordinary tensors.
        flat, views, _ = runtime.grad_bucket(model, kind)
        by_param = {id(p): p for p in model.parameters()}
        direct = (getattr(model, "grad_bucket_enabled", True)
                  and all(by_param[i].grad is None for i in views))
        out = views if direct else None
        dparams, dgrads, d_emb = runtime.decoder_backward(
            model.decoder, kind, res.emb, res.actions[:T], res.mask_trace[:T], loads,
            d_logp, T, out=out)
        eparams, egrads = runtime.encoder_backward(model.encoder, res.x3, res.depot_mask,
                                                   res.tape, d_emb, out=out)
        if direct:
            for i, v in views.items():
                by_param[i].grad = v
            return (None, None, None) + (None,) * len(ctx.param_ids)
        by_id = {id(p): g for p, g in zip(dparams + eparams, dgrads + egrads) if p is not None}
        return (None, None, None) + tuple(by_id.get(i) for i in ctx.param_ids)


def logp_with_grad(model, env, res):
    """acc_log_prob (B,) whose value is the HIP rollout's and whose gradient flows to the
    model parameters through the HIP backward.  `res` must come from a recording training
    rollout (runtime.rollout(..., train=True, record=True))."""
    if res.tape is None or res.mask_trace is None:
        raise RuntimeError("the rollout did not record what the backward pass needs "
                           "(runtime.rollout(..., train=True, record=True))")
    params = [p for p in model.parameters() if p.requires_grad]
    return _TourLogProb.apply(model, env.KIND, res, *params)
