"""Layout loss classes with the reference's names and call signatures:
`IoULoss(apply_nonlin)(x, y)` (mono/model/mono_baseline/dice_loss.py:293-331) and `BDLoss()(net_output, gt)`
(mono/model/mono_baseline/boundary_loss.py:150-192), as used by `compute_topview_loss` (net.py:554-585).

Both run the fused layout-loss kernels of libjperceiver_hip.so (`jp_layout_loss_fwd/bwd`, `jp_sdf`: softmax over
the two classes, IoU / CE / boundary terms in one pass; exact Euclidean distance transform + inner boundary on
the GPU instead of the reference's device -> host -> scipy/skimage -> device round trip) and are differentiable
through torch.autograd (one node each).  The train step itself calls the same kernels through its own tape
(ops_loss.layout_loss); these classes are the public, reference-shaped entry points.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .._lib import call


def _two_class(x, y):
    if x.dim() != 4 or x.shape[1] != 2:
        raise NotImplementedError("only 2-class (B,2,h,w) logits (num_class=2 in every north-star config)")
    if not x.is_cuda:
        raise RuntimeError("IoULoss/BDLoss run HIP kernels: inputs must be on the GPU")
    B, _, h, w = x.shape
    if y.dim() == 4 and y.shape[1] == 2:            # one-hot ground truth (dice_loss.py:53-55)
        y = y[:, 1]
    y = y.reshape(B, 1, h, w).to(device=x.device, dtype=torch.float32).contiguous()
    return x.contiguous().float(), y, B, h, w


class _LayoutLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, label, sdf, lw, cew, l2w, w0, w1, region=(1.0, 1.0, 1.0)):
        B, _, h, w = logits.shape
        sums = torch.empty(8 * B + 3, device=logits.device, dtype=torch.float64)
        out = torch.empty(1, device=logits.device, dtype=torch.float32)
        call("jp_layout_loss_fwd", logits, label, sdf, sums, out, B, h, w, float(w0), float(w1), float(lw), float(cew),
             float(l2w), *region)
        ctx.save_for_backward(logits, label, sums)
        ctx.sdf, ctx.k, ctx.region = sdf, (B, h, w, float(w0), float(w1), float(lw), float(cew), float(l2w)), region
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        logits, label, sums = ctx.saved_tensors
        B, h, w, w0, w1, lw, cew, l2w = ctx.k
        d = torch.empty_like(logits)
        call("jp_layout_loss_bwd", logits, label, ctx.sdf, sums, g.reshape(1).contiguous().float(), d, B, h, w, w0, w1, lw,
             cew, l2w, *ctx.region, 0)
        return d, None, None, None, None, None, None, None, None


def _is_softmax_dim1(fn) -> bool:
    """The reference always passes `lambda x: F.softmax(x, 1)` (net.py:563); probe the callable on a tiny tensor."""
    if fn is None:
        return False
    t = torch.tensor([[[[0.3]], [[-1.1]]]])
    try:
        return bool(torch.allclose(fn(t), torch.softmax(t, 1)))
    except Exception:
        return False


class _RegionLoss(nn.Module):
    """-mean_{b,c} (a*tp + smooth) / (a*tp + alpha*fp + beta*fn + smooth) on softmax probabilities."""
    REGION = (1.0, 1.0, 1.0)

    def __init__(self, apply_nonlin=None, batch_dice=False, do_bg=True, smooth=1., square=False):
        super().__init__()
        if batch_dice or not do_bg or smooth != 1.0 or square:
            raise NotImplementedError("only the reference's call pattern <Loss>(apply_nonlin=softmax) is built")
        if not _is_softmax_dim1(apply_nonlin):
            raise NotImplementedError("apply_nonlin must be softmax over dim 1 (fused into the kernel)")
        self.apply_nonlin, self.batch_dice, self.do_bg, self.smooth, self.square = apply_nonlin, False, True, 1.0, False

    def forward(self, x, y, loss_mask=None):
        if loss_mask is not None:
            raise NotImplementedError("loss_mask is never used by the reference's train step")
        x, y, B, h, w = _two_class(x, y)
        return _LayoutLossFn.apply(x, y, None, 1.0, 0.0, 0.0, 1.0, 1.0, self.REGION)


class IoULoss(_RegionLoss):
    """dice_loss.py:293-331."""
    REGION = (1.0, 1.0, 1.0)


class SoftDiceLoss(_RegionLoss):
    """dice_loss.py:255-290: (2 tp + s) / (2 tp + fp + fn + s)."""
    REGION = (2.0, 1.0, 1.0)


class TverskyLoss(_RegionLoss):
    """dice_loss.py:333-372: (tp + s) / (tp + 0.3 fp + 0.7 fn + s)."""
    REGION = (1.0, 0.3, 0.7)

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.alpha, self.beta = 0.3, 0.7


class FocalLoss(nn.Module):
    """focal_loss.py:7-92 as the reference builds it (net.py:568-570: `FocalLoss(apply_nonlin=softmax_helper)`):
    mean over all pixels of -alpha[t] * (1 - pt)^gamma * log(pt) with pt = sum_c clamp(onehot_c, s/(C-1), 1-s) * p_c + s,
    alpha = (alpha, 1 - alpha) for balance_index 0."""

    def __init__(self, apply_nonlin=None, alpha=0.25, gamma=2, balance_index=0, smooth=1e-5, size_average=True):
        super().__init__()
        if not _is_softmax_dim1(apply_nonlin):
            raise NotImplementedError("apply_nonlin must be softmax over dim 1 (fused into the kernel)")
        if not isinstance(alpha, float) or balance_index != 0 or smooth != 1e-5 or not size_average:
            raise NotImplementedError("only the reference's call pattern FocalLoss(apply_nonlin=softmax) (+ alpha / gamma) is built")
        self.apply_nonlin, self.alpha, self.gamma = apply_nonlin, float(alpha), float(gamma)
        self.balance_index, self.smooth, self.size_average = 0, 1e-5, True

    def forward(self, logit, target):
        x, y, B, h, w = _two_class(logit, target)
        return _LayoutLossFn.apply(x, y, None, 1.0, 0.0, 0.0, 1.0, 1.0, (-1.0, self.alpha, self.gamma))


def compute_sdf(label: torch.Tensor) -> torch.Tensor:
    """boundary_loss.py:121-147 for the foreground class on the GPU: label (B,1,h,w) or (B,h,w) {0,1} -> (B,h,w)."""
    B, h, w = label.shape[0], label.shape[-2], label.shape[-1]
    lab = label.reshape(B, 1, h, w).float().contiguous()
    sdf = torch.empty((B, h, w), device=lab.device, dtype=torch.float32)
    ws = torch.empty(2 * B * h * w + B, device=lab.device, dtype=torch.int32)
    call("jp_sdf", lab, sdf, ws, B, h, w)
    return sdf


class BDLoss(nn.Module):
    """mean(softmax(net_output)[:, 1:] * sdf(gt)[:, 1:]) (boundary_loss.py:150-192)."""

    def forward(self, net_output, gt):
        x, y, B, h, w = _two_class(net_output, gt)
        return _LayoutLossFn.apply(x, y, compute_sdf(y), 0.0, 0.0, 1.0, 1.0, 1.0)
