"""fp32 Bayes-by-backprop layers / MLP behind the reference's API (BASELINE config 0).

Mirror of reference src/models/stochastic/bbb/linear.py (`Linear`, :8-50, eval branch) and models_bbb.py
(`LinearNetwork`, :32-90).  Parameters keep the reference names: `weight` (mu), `std` (rho, pre-softplus), `bias`.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .layers import _MC, mc_context, timed


class Linear(nn.Module):
    """reference bbb.linear.Linear(in_features, out_features, bias, sigma_prior=1.0, args=None).
    Eval-mode forward for S MC samples: sigma = softplus(rho); W_s = mu + eps_s * sigma; y_s = x_s @ W_s^T + b."""

    def __init__(self, in_features, out_features, bias, sigma_prior=1.0, args=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features).uniform_(-0.01, 0.01), requires_grad=False)
        self.std = nn.Parameter(torch.full((out_features, in_features), -3.0), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_features).uniform_(-0.01, 0.01), requires_grad=False) if bias else None
        self.std_prior = nn.Parameter(torch.ones((1,)) * sigma_prior, requires_grad=False)
        self.args = args
        self.layer_id = 0
        self._sigma = None

    def get_kl_divergence(self):
        """reference utils_bbb.kl_divergence (bbb/utils_bbb.py:3-5) against N(0, std_prior)."""
        sigma, mu, sp = F.softplus(self.std), self.weight, self.std_prior
        return 0.5 * (2 * torch.log(sp / sigma) - 1 + (sigma / sp).pow(2) + ((0 - mu) / sp).pow(2)).sum()

    def forward(self, x, act=0, eps=None):
        """x fp32 [S or 1, B, in_features] -> [S, B, out_features]."""
        if x.device.type != "cuda":
            raise RuntimeError("qbnn layers run on an MI355X only (no CPU fallback)")
        S = _MC.samples
        if self._sigma is None or self._sigma.device != x.device:
            self._sigma = F.softplus(self.std.detach().cpu().float()).to(x.device).contiguous()   # one-time, same op as the reference
        mu = self.weight.detach().to(x.device).contiguous()
        n = mu.numel()
        w = torch.empty((S, n), dtype=torch.float32, device=x.device)
        if eps is not None:
            eps = eps.to(device=x.device, dtype=torch.float32).contiguous()
        with timed("sample_weights_f32"):
            _lib.check(_lib.lib().qbnn_sample_weights_f32(_lib.ptr(mu), _lib.ptr(self._sigma), n, _MC.seed, self.layer_id,
                                                          _MC.sample_begin, S, _lib.ptr(eps), _lib.ptr(w), _lib.current_stream()))
        B = x.shape[1]
        y = torch.empty((S, B, self.out_features), dtype=torch.float32, device=x.device)
        xs = 0 if x.shape[0] == 1 else x[0].numel()
        b = None if self.bias is None else self.bias.detach().to(x.device).contiguous()
        with timed("linear_f32"):
            _lib.check(_lib.lib().qbnn_linear_f32_mc(_lib.ptr(x.contiguous()), xs, _lib.ptr(w), n, _lib.ptr(b), _lib.ptr(y), y[0].numel(),
                                                     B, self.in_features, self.out_features, act, S, _lib.current_stream()))
        return y


class LinearNetwork(nn.Module):
    """reference models_bbb.LinearNetwork (float, q=False): 3 x (Linear(100) + ReLU), heads `mu` and `log_var`;
    forward -> (mu, exp(log_var))."""

    def __init__(self, input_size, output_size, q, args):
        super().__init__()
        if q:
            raise NotImplementedError("the quantised MLP is not built yet")
        self.args = args
        self.input_size = 1
        for i in input_size:
            self.input_size *= int(i)
        self.output_size = int(output_size)
        sp = getattr(args, "sigma_prior", 1.0)
        widths = [100, 100, 100]
        self.layers = nn.ModuleList([])
        prev = self.input_size
        for wd in widths:
            self.layers.append(Linear(prev, wd, bias=True, sigma_prior=sp, args=args))
            self.layers.append(nn.ReLU())
            prev = wd
        self.mu = Linear(prev, 1, bias=True, sigma_prior=sp, args=args)
        self.log_var = Linear(prev, 1, bias=True, sigma_prior=sp, args=args)
        for i, m in enumerate(self.stochastic_layers()):
            m.layer_id = i
        self.q = q

    def stochastic_layers(self):
        return [self.layers[0], self.layers[2], self.layers[4], self.mu, self.log_var]

    def stochastic_layer_names(self):
        return ["layers.0", "layers.2", "layers.4", "mu", "log_var"]

    def load_reference_state(self, state):
        for n, m in zip(self.stochastic_layer_names(), self.stochastic_layers()):
            m.weight.data = torch.from_numpy(np.asarray(state[n + ".weight"], np.float32).copy())
            m.std.data = torch.from_numpy(np.asarray(state[n + ".std"], np.float32).copy())
            m.bias.data = torch.from_numpy(np.asarray(state[n + ".bias"], np.float32).copy())
            m._sigma = None
        return self

    def get_kl_divergence(self):
        return sum(m.get_kl_divergence() for m in self.stochastic_layers())

    def forward_mc(self, x):
        """All S samples -> (mu [S,B,1], var [S,B,1])."""
        h = x.to(torch.float32).reshape(1, x.shape[0], -1)
        for m in (self.layers[0], self.layers[2], self.layers[4]):
            h = m(h, act=1)
        return self.mu(h, act=0), self.log_var(h, act=2)

    def forward(self, x):
        with mc_context(1, _MC.seed, _MC.sample_begin, _MC.eps):
            mu, var = self.forward_mc(x)
        return mu[0], var[0]
