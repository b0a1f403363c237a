"""Flat-arena runtime for the optimizer side of the step.

All trainable parameters of the model are re-pointed into ONE contiguous fp32 device buffer
(`params`), with matching `grads`, Adam `exp_avg` and `exp_avg_sq` arenas.  Parameters that never
receive a gradient in the reference (ResNet `fc`, CCT `res_conv`) are placed at the
tail so the reduced / updated range is a single prefix.  Consequences:
  * gradient zeroing is one memset, the global-norm is one reduction, clip+Adam is one kernel;
  * the data-parallel all-reduce works on slices of one buffer (bucketed, RCCL over xGMI) instead of
    466 tensors (the reference flattens them every step, mono/core/utils/dist_utils.py:17-31).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ._lib import call

_NO_GRAD_SUFFIXES = (".fc.weight", ".fc.bias", ".res_conv.weight", ".res_conv.bias")


def _align(n, a=64):
    return (n + a - 1) // a * a


class FlatArena:
    def __init__(self, model: nn.Module, skip_prefixes=()):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        live, dead = [], []
        for n, p in named:
            is_dead = n.endswith(_NO_GRAD_SUFFIXES) or any(n.startswith(s) for s in skip_prefixes)
            (dead if is_dead else live).append((n, p))
        self.entries = []          # (name, param, offset, numel)
        off = 0
        for n, p in live + dead:
            self.entries.append((n, p, off, p.numel()))
            off += _align(p.numel())   # 256-B aligned slices: float4 kernels and RCCL-friendly
        self.total = off
        self.live_numel = 0
        for n, p, o, k in self.entries[:len(live)]:
            self.live_numel = o + _align(k)
        dev = named[0][1].device
        self.params = torch.zeros(self.total, device=dev, dtype=torch.float32)
        self.grads = torch.zeros(self.total, device=dev, dtype=torch.float32)
        for n, p, o, k in self.entries:
            self.params[o:o + k].copy_(p.data.reshape(-1))      # one-time setup (plumbing)
            p.data = self.params[o:o + k].view(p.shape)
            p.grad = self.grads[o:o + k].view(p.shape)
        self.exp_avg = None
        self.exp_avg_sq = None
        self.normsq = torch.zeros(1, device=dev, dtype=torch.float64)
        self.step_count = 0

    def zero_grad(self):
        # memset node on the current stream; keeps p.grad views alive (no reallocation)
        call("jp_fill", self.grads, self.live_numel, 0.0)

    def grad_norm_sq(self):
        call("jp_grad_sumsq", self.grads, self.normsq, self.live_numel, 0)
        return self.normsq

    def adam_step(self, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, max_norm=None, grad_scale=1.0):
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.params)
            self.exp_avg_sq = torch.zeros_like(self.params)
        self.step_count += 1
        normsq = None
        if max_norm is not None and max_norm > 0:
            normsq = self.grad_norm_sq()
        call("jp_adam_clip_step", self.params, self.grads, self.exp_avg, self.exp_avg_sq, self.live_numel, normsq,
             float(grad_scale), float(max_norm or 0.0), float(lr), float(betas[0]), float(betas[1]), float(eps),
             self.step_count)


class FlatAdam:
    """torch.optim.Adam-shaped facade over the arena (what build_optimizer returns, trainer.py:76-143)."""

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, skip_prefixes=()):
        if weight_decay:
            raise NotImplementedError("the north-star configs use weight_decay=0")
        self.arena = FlatArena(model, skip_prefixes)
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0)
        self.param_groups = [dict(params=[p for _, p, _, _ in self.arena.entries], **self.defaults)]
        self.max_norm = None          # set by DistOptimizerHook (grad_clip)
        self.grad_scale = 1.0         # 1/world_size folded into the Adam pass after an all-reduce(SUM)

    def zero_grad(self):
        self.arena.zero_grad()

    def step(self):
        g = self.param_groups[0]
        self.arena.adam_step(g["lr"], g["betas"], g["eps"], self.max_norm, self.grad_scale)

    def state_dict(self):
        a = self.arena
        return dict(step=a.step_count, exp_avg=a.exp_avg, exp_avg_sq=a.exp_avg_sq, param_groups=[
            {k: v for k, v in self.param_groups[0].items() if k != "params"}])

    def load_state_dict(self, sd):
        a = self.arena
        a.step_count = sd["step"]
        if sd["exp_avg"] is not None:
            a.exp_avg = sd["exp_avg"].to(a.params.device)
            a.exp_avg_sq = sd["exp_avg_sq"].to(a.params.device)
