"""The Monte-Carlo evaluation loop and its multi-GPU sharding.

Replaces the inline loop of reference experiments/utils.py:342-355 (`_evaluate_with_loader`): S stochastic forwards of
one batch reduced to the predictive mean of the softmax probabilities (classification branch, :355).  The build also
keeps sum p^2, so the per-class predictive variance is available (a superset of what the reference returns).

Multi-GPU: MC samples are independent given (parameters, input, seed).  Rank r of G evaluates the contiguous block of
global sample indices [r*S/G, (r+1)*S/G); the Philox subsequence is the GLOBAL sample index, so per-sample results do
not depend on G.  One sum all-reduce of the [2, B, C] fp64 partial moments (40 KB at B=256, C=10) over RCCL/xGMI.
"""
import os

import torch
import torch.distributed as dist

from . import _lib
from .layers import mc_context, state_epoch, timed


def shard_samples(samples, rank, world_size):
    """Contiguous block partition of range(samples): returns (begin, count) for `rank`."""
    base, rem = divmod(samples, world_size)
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)


def reduce_moments(probs, accumulate_into=None, finalize_total=0, want_var=True):
    """sum_s p and sum_s p^2 over the leading (sample) dim on the device, in sample order, in fp64: returns [2, B, C].
    finalize_total > 0: the same launch also finalises (single rank, last chunk) -> returns (moments, mean, var)."""
    S = probs.shape[0]
    n = probs[0].numel()
    shape = tuple(probs.shape[1:])
    mom = accumulate_into if accumulate_into is not None else torch.empty((2,) + shape, dtype=torch.float64, device=probs.device)
    assert mom.dtype == torch.float64
    mean = var = None
    if finalize_total > 0:
        mean = torch.empty(shape, dtype=torch.float32, device=probs.device)
        var = torch.empty(shape, dtype=torch.float32, device=probs.device) if want_var else None
    with timed("reduce_moments"):
        _lib.check(_lib.lib().qbnn_reduce_moments(_lib.ptr(probs.contiguous()), S, n, int(accumulate_into is not None), _lib.ptr(mom),
                                                  int(finalize_total), _lib.ptr(mean), _lib.ptr(var), _lib.current_stream()))
    return (mom, mean, var) if finalize_total > 0 else mom


def finalize_moments(moments, samples, want_var=True):
    """mean = sum/S ; var = (sum2 - sum^2/S)/(S-1) (unbiased, as torch.var in experiments/utils.py:352), from fp64 sums.
    Device moments: one small kernel (qbnn_finalize_moments).  Host moments (the gloo tests' stand-in ranks): torch fp64."""
    if moments.device.type == "cuda" and moments.dtype == torch.float64:
        shape = tuple(moments.shape[1:])
        mean = torch.empty(shape, dtype=torch.float32, device=moments.device)
        var = torch.empty(shape, dtype=torch.float32, device=moments.device) if want_var else None
        with timed("finalize_moments"):
            _lib.check(_lib.lib().qbnn_finalize_moments(_lib.ptr(moments.contiguous()), mean.numel(), int(samples), _lib.ptr(mean), _lib.ptr(var),
                                                        _lib.current_stream()))
        return mean, var
    m64 = moments.to(torch.float64)
    mean = m64[0] / samples
    if samples > 1:
        var = (m64[1] - m64[0] * mean).clamp_min(0) / (samples - 1)
    else:
        var = torch.zeros_like(mean)
    return mean.to(torch.float32), var.to(torch.float32)


def _force_dist():
    return os.environ.get("QBNN_BENCH_FORCE_DIST", "0") == "1"


def _dist_active(group=None):
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or _force_dist())


def all_reduce_moments(moments, group=None):
    """The path's one collective: sum all-reduce of the [2, B, C] fp64 partial moments (40 KB at B=256, C=10) over RCCL / xGMI."""
    if _dist_active(group):
        dist.all_reduce(moments, op=dist.ReduceOp.SUM, group=group)
    return moments


def _rank_world(model, group):
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if world > 1 and getattr(model, "sequential_samples", False):
        # QAT evaluation with live observers (models_qat.py): sample s depends on samples < s through the EMA min/max of
        # every FakeQuantize, so a rank that starts at sample_begin > 0 would run with the wrong observer state
        raise RuntimeError("qat_eval models evaluate their MC samples in order (live observers) and cannot be sharded over ranks; "
                           "evaluate them on one rank")
    return rank, world


def mc_predict_regression(model, x, samples, seed, group=None):
    """Regression branch of the reference loop (experiments/utils.py:348-353):
    returns (mean_s mu_s, var_unbiased_s(mu_s) + mean_s var_s), each [B, 1]."""
    rank, world = _rank_world(model, group)
    begin, count = shard_samples(samples, rank, world)
    B = x.shape[0]
    moments = torch.zeros((2, B, 2), dtype=torch.float64, device=x.device)
    if count > 0:
        with mc_context(count, seed, begin):
            mu, var = model.forward_mc(x)
        moments = reduce_moments(torch.cat([mu, var], dim=-1).contiguous())      # [2, B, 2]: sums and sums of squares of (mu, var)
    all_reduce_moments(moments, group)
    mean, uvar = finalize_moments(moments, samples)                               # [B, 2] each: columns (mu, var)
    return mean[:, 0:1], uvar[:, 0:1] + mean[:, 1:2]


def mc_predict(model, x, samples, seed, return_var=False, chunk=None, group=None, return_probs=False):
    """Predictive mean (and variance) of `samples` stochastic forwards of `x`; equals the reference loop
    (experiments/utils.py:342-355) given identical per-sample weight noise."""
    rank, world = _rank_world(model, group)
    begin, count = shard_samples(samples, rank, world)
    chunk = count if chunk is None else chunk
    single = not _dist_active(group)                 # one rank: the last chunk's reduction launch also finalises
    moments, all_probs, mean, var = None, [], None, None
    done = 0
    while done < count:
        n = min(chunk, count - done)
        with mc_context(n, seed, begin + done):
            probs = model.forward_mc(x)
        if single and done + n == count:
            moments, mean, var = reduce_moments(probs, moments, finalize_total=samples, want_var=return_var)
        else:
            moments = reduce_moments(probs, moments)
        if return_probs:
            all_probs.append(probs)
        done += n
    if mean is None:
        if moments is None:
            B = x.shape[0]
            moments = torch.zeros((2, B, model.output_size), dtype=torch.float64, device=x.device)
        all_reduce_moments(moments, group)
        mean, var = finalize_moments(moments, samples, want_var=return_var)
    out = (mean, var) if return_var else mean
    if return_probs:
        return out, (torch.cat(all_probs, 0) if all_probs else None)
    return out


class GraphedPredictor:
    """`mc_predict` / `mc_predict_regression` captured ONCE per input shape into a HIP graph and replayed.

    A Monte-Carlo evaluation of a small network is 15-25 launches of a few microseconds each (LeNet, MLP; an ensemble: 13): the
    launches, not the kernels, set its time.  Kernel arguments are frozen at capture, so the Philox seed and the first global
    sample index are read from three words of device memory instead (`qbnn_set_device_noise_source`); a call copies the input
    into the graph's static buffer, writes the three words, and replays.  Same results as the eager path for the same
    (x, seed): the kernels and their order are identical.  One rank only evaluates the whole graph; with a process group the
    local moments come from the graph and the all-reduce + finalisation run after it.
    (The ResNet BBB step gains nothing from this -- its launch gaps are 0.02 ms of 3.7 -- but it works there too.)"""

    def __init__(self, model, samples, return_var=False, regression=False, group=None):
        self.model, self.samples, self.return_var, self.regression, self.group = model, int(samples), return_var, regression, group
        # A captured graph holds raw device pointers to the packed weights / qparams of the state it was captured with.  Every path
        # that may drop or replace such a buffer (a state dict loaded into any layer, a packed layout switched by an eager call with
        # `record=` / another fast-path flag) bumps layers.state_epoch(); an entry captured under an older epoch is captured again.
        self._graphs = {}
        self._is_ensemble = hasattr(model, "ensemble")

    def _local(self, x, begin, count):
        """The rank-local part: forward of `count` samples + their fp64 moments (and mean / var when this rank is all there is)."""
        with mc_context(count, 0, begin):
            out = self.model.forward_mc(x)
        if self.regression:
            mu, var = out
            return reduce_moments(torch.cat([mu, var], dim=-1).contiguous())
        if not _dist_active(self.group):
            return reduce_moments(out, None, finalize_total=self.samples, want_var=self.return_var)
        return reduce_moments(out)

    def _capture(self, x, begin, count):
        L = _lib.lib()
        static_x = x.clone()
        noise = torch.zeros(4, dtype=torch.int32, device=x.device)
        self._local(static_x, begin, count)                     # eager once: packs weights, builds every cache the launches read
        torch.cuda.synchronize()
        _lib.check(L.qbnn_set_device_noise_source(_lib.ptr(noise)))
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                res = self._local(static_x, begin, count)
        finally:
            _lib.check(L.qbnn_set_device_noise_source(None))
        return static_x, noise, graph, res, state_epoch()        # the epoch AFTER the warm-up call has settled layouts and caches

    def __call__(self, x, seed, sample_begin=0):
        import numpy as np
        if x.device.type != "cuda":
            raise RuntimeError("qbnn models run on an MI355X only (no CPU fallback)")
        rank, world = _rank_world(self.model, self.group)
        begin, count = shard_samples(self.samples, rank, world)
        if self._is_ensemble and sample_begin != 0:
            raise ValueError("GraphedPredictor: an ensemble's member indices are fixed at capture; sample_begin must be 0")
        if count == 0:          # a rank without samples (more ranks than samples): zero moments, as mc_predict does
            B = x.shape[0]
            width = 2 if self.regression else self.model.output_size
            moments = torch.zeros((2, B, width), dtype=torch.float64, device=x.device)
            all_reduce_moments(moments, self.group)
            if self.regression:
                mean, uvar = finalize_moments(moments, self.samples)
                return mean[:, 0:1], uvar[:, 0:1] + mean[:, 1:2]
            mean, var = finalize_moments(moments, self.samples, want_var=self.return_var)
            return (mean, var) if self.return_var else mean
        key = (tuple(x.shape), x.dtype, x.device.index, begin, count)
        ent = self._graphs.get(key)
        if ent is None or ent[4] != state_epoch():
            ent = self._graphs[key] = self._capture(x, begin, count)
        static_x, noise, graph, res, _epoch = ent
        static_x.copy_(x)
        # the three noise words go up from a FRESH pageable tensor with a blocking copy: the runtime stages pageable memory before
        # the call returns, so a later call can never overwrite words that an earlier replay has not read yet (one reused pinned
        # buffer with an asynchronous copy could, when replays queue up)
        words = np.array([seed & 0xffffffff, (seed >> 32) & 0xffffffff, (begin + sample_begin) & 0xffffffff, 0], np.uint32).view(np.int32)
        noise.copy_(torch.from_numpy(words), non_blocking=False)
        graph.replay()
        if self.regression:
            moments = res.clone()
            all_reduce_moments(moments, self.group)
            mean, uvar = finalize_moments(moments, self.samples)
            return mean[:, 0:1], uvar[:, 0:1] + mean[:, 1:2]
        if not _dist_active(self.group):
            _mom, mean, var = res
            return (mean.clone(), var.clone()) if self.return_var else mean.clone()
        moments = res.clone()
        all_reduce_moments(moments, self.group)
        mean, var = finalize_moments(moments, self.samples, want_var=self.return_var)
        return (mean, var) if self.return_var else mean
