"""The Monte-Carlo evaluation loop and its multi-GPU sharding.

Replaces the inline loop of reference experiments/utils.py:342-355 (`_evaluate_with_loader`): S stochastic forwards of
one batch reduced to the predictive mean of the softmax probabilities (classification branch, :355).  The build also
keeps sum p^2, so the per-class predictive variance is available (a superset of what the reference returns).

Multi-GPU: MC samples are independent given (parameters, input, seed).  Rank r of G evaluates the contiguous block of
global sample indices [r*S/G, (r+1)*S/G); the Philox subsequence is the GLOBAL sample index, so per-sample results do
not depend on G.  One sum all-reduce of the [2, B, C] fp32 partial moments (20 KB at B=256, C=10) over RCCL/xGMI.
"""
import os

import torch
import torch.distributed as dist

from . import _lib
from .layers import mc_context, timed


def shard_samples(samples, rank, world_size):
    """Contiguous block partition of range(samples): returns (begin, count) for `rank`."""
    base, rem = divmod(samples, world_size)
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)


def reduce_moments(probs, accumulate_into=None):
    """sum_s p and sum_s p^2 over the leading (sample) dim on the device, in sample order: returns [2, B, C]."""
    S = probs.shape[0]
    n = probs[0].numel()
    mom = accumulate_into if accumulate_into is not None else torch.empty((2,) + tuple(probs.shape[1:]), dtype=torch.float32, device=probs.device)
    with timed("reduce_moments"):
        _lib.check(_lib.lib().qbnn_reduce_moments(_lib.ptr(probs.contiguous()), S, n, int(accumulate_into is not None), _lib.ptr(mom),
                                                  _lib.current_stream()))
    return mom


def finalize_moments(moments, samples):
    """mean = sum/S ; var = (sum2 - S mean^2)/(S-1) (unbiased, as torch.var in experiments/utils.py:352)."""
    mean = moments[0] / samples
    if samples > 1:
        var = (moments[1] - samples * mean * mean).clamp_min(0) / (samples - 1)
    else:
        var = torch.zeros_like(mean)
    return mean, var


def all_reduce_moments(moments, group=None):
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or os.environ.get("QBNN_BENCH_FORCE_DIST", "0") == "1"):
        dist.all_reduce(moments, op=dist.ReduceOp.SUM, group=group)
    return moments


def mc_predict_regression(model, x, samples, seed, group=None):
    """Regression branch of the reference loop (experiments/utils.py:348-353):
    returns (mean_s mu_s, var_unbiased_s(mu_s) + mean_s var_s), each [B, 1]."""
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    begin, count = shard_samples(samples, rank, world)
    B = x.shape[0]
    moments = torch.zeros((2, B, 2), dtype=torch.float32, device=x.device)
    if count > 0:
        with mc_context(count, seed, begin):
            mu, var = model.forward_mc(x)
        moments = reduce_moments(torch.cat([mu, var], dim=-1).contiguous())      # [2, B, 2]: sums and sums of squares of (mu, var)
    all_reduce_moments(moments, group)
    mean_mu = moments[0, :, 0:1] / samples
    var_mu = (moments[1, :, 0:1] - samples * mean_mu * mean_mu).clamp_min(0) / max(samples - 1, 1)
    return mean_mu, var_mu + moments[0, :, 1:2] / samples


def mc_predict(model, x, samples, seed, return_var=False, chunk=None, group=None, return_probs=False):
    """Predictive mean (and variance) of `samples` stochastic forwards of `x`; equals the reference loop
    (experiments/utils.py:342-355) given identical per-sample weight noise."""
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    begin, count = shard_samples(samples, rank, world)
    chunk = count if chunk is None else chunk
    moments, all_probs = None, []
    done = 0
    while done < count:
        n = min(chunk, count - done)
        with mc_context(n, seed, begin + done):
            probs = model.forward_mc(x)
        moments = reduce_moments(probs, moments)
        if return_probs:
            all_probs.append(probs)
        done += n
    if moments is None:
        B = x.shape[0]
        moments = torch.zeros((2, B, model.output_size), dtype=torch.float32, device=x.device)
    all_reduce_moments(moments, group)
    mean, var = finalize_moments(moments, samples)
    out = (mean, var) if return_var else mean
    if return_probs:
        return out, (torch.cat(all_probs, 0) if all_probs else None)
    return out
