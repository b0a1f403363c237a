"""Tape ops for the CCT attention algebra and every loss of the JPerceiver train step.

Loss scalars live in ONE device vector (`LossVec.vals`); each loss op writes its slot and, in
backward, reads its upstream gradient from the matching slot of `LossVec.grads` *on the device*
(the kernels take a `gout` pointer), so neither forward nor backward ever synchronises with the host.
"""
from __future__ import annotations

import torch

from ._lib import call
from . import ops
from .ops import Var, _rec, _new, gemm


# ------------------------------------------------------------------------------------------- CCT pieces
def bmm_tn(k: Var, q: Var) -> Var:
    """energy[b] = k[b]^T @ q[b]  for k, q of shape (B, C, N): (B, N, N)
    (proj_key.permute(0,2,1) bmm proj_query, CrossViewTransformer.py:53-56)."""
    B, C, N = k.t.shape
    e = _new((B, N, N), k.t)
    gemm(k.t, q.t, e, N, N, C, N, N, N, C * N, C * N, N * N, B, tA=1)
    out = Var(e, k.rg or q.rg)

    def bwd():
        if out.g is None:
            return
        dE = out.g
        if k.rg:   # dk[c][i] = sum_j q[c][j] dE[i][j]
            g, acc = k.grad_buf()
            gemm(q.t, dE, g, C, N, N, N, N, N, C * N, N * N, C * N, B, tB=1, beta=float(acc))
        if q.rg:   # dq[c][j] = sum_i k[c][i] dE[i][j]
            g, acc = q.grad_buf()
            gemm(k.t, dE, g, C, N, N, N, N, N, C * N, N * N, C * N, B, beta=float(acc))
        out.g = None

    _rec(out.rg, bwd)
    return out


def colmax(e: Var, want_arg=True):
    """torch.max(e, dim=1) for e (B, R, N) -> values (B, N) Var, argmax (B, N) int64 tensor."""
    B, R, N = e.t.shape
    val = _new((B, N), e.t)
    arg = _new((B, N), e.t, torch.int64)
    call("jp_colmax", e.t, val, arg, B, R, N)
    out = Var(val, e.rg)

    def bwd():
        if out.g is None:
            return
        d = torch.empty_like(e.t)
        call("jp_colmax_bwd", out.g, arg, d, B, R, N)
        e.add_grad(d)
        out.g = None

    _rec(out.rg, bwd)
    return out, arg


def gather_cols(v: Var, arg: torch.Tensor) -> Var:
    """feature_selection(v, 2, arg): T[b,c,j] = v[b,c,arg[b,j]] (CrossViewTransformer.py:14-24,61)."""
    B, C, N = v.t.shape
    t = torch.empty_like(v.t)
    call("jp_gather_cols", v.t, arg, t, B, C, N)
    out = Var(t, v.rg)

    def bwd():
        if out.g is None:
            return
        d = torch.empty_like(v.t)
        call("jp_gather_cols_bwd", out.g, arg, d, B, C, N)
        v.add_grad(d)
        out.g = None

    _rec(out.rg, bwd)
    return out


def mul_bcast_c(a: Var, s: Var) -> Var:
    """a (B,C,H,W) * s (B,1,H,W)  (front_res * S, CrossViewTransformer.py:68)."""
    B, C, H, W = a.t.shape
    y = torch.empty_like(a.t)
    call("jp_mul_bcast_c", a.t, s.t, y, B, C, H * W)
    out = Var(y, a.rg or s.rg)

    def bwd():
        if out.g is None:
            return
        if a.rg:
            d = torch.empty_like(a.t)
            call("jp_mul_bcast_c", out.g, s.t, d, B, C, H * W)
            a.add_grad(d)
        if s.rg:
            d = torch.empty_like(s.t)
            call("jp_mul_bcast_c_bwd_s", out.g, a.t, d, B, C, H * W)
            s.add_grad(d)
        out.g = None

    _rec(out.rg, bwd)
    return out


def bcast_matmul(attn: Var, v: Var) -> Var:
    """(B,1,n,n) @ (B,C,n,n) -> (B,C,n,n): true n x n matrix product per channel (CrossViewTransformer.py:88)."""
    B, C, n, n2 = v.t.shape
    assert n == n2 and attn.t.shape[-1] == n and attn.t.shape[-2] == n, "CCT depth attention needs square maps"
    y = torch.empty_like(v.t)
    call("jp_bcast_matmul_fwd", attn.t, v.t, y, B, C, n)
    out = Var(y, attn.rg or v.rg)

    def bwd():
        if out.g is None:
            return
        da = torch.empty_like(attn.t) if attn.rg else None
        dv = torch.empty_like(v.t) if v.rg else None
        call("jp_bcast_matmul_bwd", attn.t, v.t, out.g, da, dv, B, C, n)
        if attn.rg:
            attn.add_grad(da)
        if v.rg:
            v.add_grad(dv)
        out.g = None

    _rec(out.rg, bwd)
    return out


def view(x: Var, shape) -> Var:
    """Reshape sharing storage; gradients share storage too (lazily allocated on first use)."""
    out = Var(x.t.view(shape), x.rg)

    def bwd():
        if out.g is None:
            return
        d = out.g.view(x.t.shape)
        if x.g is None:
            x.g = d
            x.gamax = out.gamax       # same values, same magnitude
        else:
            call("jp_axpby", x.g, d, x.g, d.numel(), 1.0, 1.0)
            x.gamax = None            # accumulated into: what a producer reported for the first addend no longer bounds it
        out.g = None

    _rec(out.rg, bwd)
    return out


# ------------------------------------------------------------------------------------------- loss vector
class LossVec:
    def __init__(self, names, device):
        self.names = list(names)
        self.index = {n: i for i, n in enumerate(self.names)}
        self.vals = torch.zeros(len(self.names), device=device, dtype=torch.float32)
        self.grads = torch.zeros(len(self.names), device=device, dtype=torch.float32)

    def val(self, name):
        i = self.index[name]
        return self.vals[i:i + 1]

    def grad(self, name):
        i = self.index[name]
        return self.grads[i:i + 1]


def combine(lv: LossVec, out_name, terms):
    """vals[out] = sum coef*vals[name] (layout_loss = topview + 0.001*transform + transform_topview,
    net.py:124-125); backward adds coef*grads[out] into the terms' gradient slots."""
    o = lv.val(out_name)
    first = True
    for name, c in terms:
        call("jp_axpby", lv.val(name), None if first else o, o, 1, float(c), 1.0)
        first = False

    def bwd():
        go = lv.grad(out_name)
        for name, c in terms:
            g = lv.grad(name)
            call("jp_axpby", go, g, g, 1, float(c), 1.0)

    _rec(True, bwd)


# ------------------------------------------------------------------------------------------- pose
class PoseP:
    """P = (K @ cam_T_cam)[:3] with a double-precision gradient accumulator shared by all scales."""
    __slots__ = ("T", "P", "dP")

    def __init__(self, T, P, dP):
        self.T, self.P, self.dP = T, P, dP


def pose(axisangle: Var, translation: Var, K: torch.Tensor, invert: bool) -> PoseP:
    """transformation_from_parameters (net.py:704-725) + Project's K@T (layers.py:74).
    axisangle / translation: (B,3) Vars."""
    B = axisangle.t.shape[0]
    T = _new((B, 4, 4), K)
    P = _new((B, 3, 4), K)
    call("jp_pose_fwd", axisangle.t, translation.t, K, T, P, B, int(invert))
    dP = torch.zeros((B, 12), device=K.device, dtype=torch.float64) if (axisangle.rg or translation.rg) else None
    pp = PoseP(T, P, dP)

    def bwd():
        ga, acc = axisangle.grad_buf()
        gt, acc2 = translation.grad_buf()
        assert acc == acc2
        call("jp_pose_bwd", dP, axisangle.t, translation.t, K, ga, gt, B, int(invert), acc)

    _rec(dP is not None, bwd)
    return pp


# ------------------------------------------------------------------------------------------- photometric
def ssim_l1(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """compute_reprojection_loss forward only (net.py:88-92) -> (B,1,H,W)."""
    B, _, H, W = pred.shape
    out = _new((B, 1, H, W), pred)
    call("jp_ssim_l1_fwd", pred, target, out, B, H, W)
    return out


def min_reprojection_loss(lv: LossVec, slot, disp: Var, poses, colors, target, invK, id_losses, noises,
                          H, W, min_depth, max_depth, n_scales):
    """One scale of net.py:145-175: CGT warp of every source frame, SSIM+L1, min over
    [identity(+noise)..., warped...], mean / n_scales.  Returns (pred list, min_index)."""
    B = disp.t.shape[0]
    hs, ws = disp.t.shape[2], disp.t.shape[3]
    preds, rls = [], []
    for pp, col in zip(poses, colors):
        pred = _new((B, 3, H, W), target)
        call("jp_cgt_warp_fwd", disp.t, hs, ws, invK, pp.P, col, pred, B, H, W, min_depth, max_depth)
        preds.append(pred)
        rls.append(ssim_l1(pred, target))
    cands = list(id_losses) + rls
    assert 1 <= len(cands) <= 4 and len(id_losses) <= 2
    c = cands + [None] * (4 - len(cands))
    nz = list(noises) + [None] * (2 - len(noises))
    idx = _new((B, H, W), target, torch.int64)
    acc = _new((1,), target, torch.float64)
    total = B * H * W
    call("jp_minreproj_fwd", c[0], c[1], c[2], c[3], nz[0], nz[1], idx, acc, total)
    gscale = 1.0 / (total * n_scales)
    call("jp_scalar_finalize", acc, lv.val(slot), 1, gscale)

    def bwd():
        gout = lv.grad(slot)
        dup = None
        for j, (pp, col, pred) in enumerate(zip(poses, colors, preds)):
            need_pose = pp.dP is not None
            if not (disp.rg or need_pose):
                continue
            dpred = torch.empty_like(pred)
            call("jp_ssim_l1_bwd", pred, target, idx, len(id_losses) + j, gout, gscale, dpred, B, H, W)
            first = dup is None
            if first:
                dup = _new((B, 1, H, W), target)
            dP = pp.dP if need_pose else torch.zeros((B, 12), device=target.device, dtype=torch.float64)
            call("jp_cgt_warp_bwd", dpred, disp.t, hs, ws, invK, pp.P, col, dup, dP, B, H, W, min_depth, max_depth,
                 0 if first else 1)
        if disp.rg and dup is not None:
            g, acc_ = disp.grad_buf()
            call("jp_bilinear_bwd", dup, g, B, hs, ws, H, W, acc_)

    _rec(True, bwd)
    return preds, idx


def scale_loss(lv: LossVec, slot, disp: Var, label: torch.Tensor, weight, min_depth, max_depth, crop=None):
    """get_scale_loss (net.py:193-211) * weight: masked abs-rel of the bilinearly resized depth."""
    B, _, hs, ws = disp.t.shape
    FH, FW = label.shape[2], label.shape[3]
    y0, y1, x0, x1 = crop if crop is not None else (0, FH, 0, FW)
    acc = _new((2,), disp.t, torch.float64)
    call("jp_scale_loss_fwd", disp.t, hs, ws, label, acc, B, FH, FW, min_depth, max_depth, y0, y1, x0, x1)
    call("jp_ratio_finalize", acc, lv.val(slot), float(weight))

    def bwd():
        if not disp.rg:
            return
        g, a = disp.grad_buf()
        call("jp_scale_loss_bwd", disp.t, hs, ws, label, acc, lv.grad(slot), float(weight), g, B, FH, FW, min_depth,
             max_depth, y0, y1, x0, x1, a)

    _rec(True, bwd)


def smooth_loss(lv: LossVec, slot, disp: Var, img_ds: torch.Tensor, weight):
    """disp mean-normalisation + get_smooth_loss (net.py:182-190,758-781) * weight."""
    B, _, h, w = disp.t.shape
    dsum = _new((B,), disp.t, torch.float64)
    call("jp_row_sum", disp.t, dsum, B, h * w)
    acc = _new((1,), disp.t, torch.float64)
    call("jp_smooth_fwd", disp.t, dsum, img_ds, acc, B, h, w)
    call("jp_scalar_finalize", acc, lv.val(slot), 1, float(weight))

    def bwd():
        if not disp.rg:
            return
        g = torch.empty_like(disp.t)
        gd = _new((B,), disp.t, torch.float64)
        dd, a = disp.grad_buf()
        call("jp_smooth_bwd", disp.t, dsum, img_ds, lv.grad(slot), float(weight), g, gd, dd, B, h, w, a)

    _rec(True, bwd)


def signed_distance(label: torch.Tensor) -> torch.Tensor:
    """compute_sdf for class 1 (boundary_loss.py:121-147) on the GPU.  label (B,1,h,w) {0,1} floats."""
    B, _, h, w = label.shape
    sdf = _new((B, h, w), label)
    ws = _new((2 * B * h * w + B,), label, torch.int32)
    call("jp_sdf", label, sdf, ws, B, h, w)
    return sdf


REGION = {"iou": (1.0, 1.0, 1.0), "dice": (2.0, 1.0, 1.0), "tversky": (1.0, 0.3, 0.7),   # (a, alpha, beta) of the overlap score
          "focal": (-1.0, 0.25, 2.0)}        # a < 0: FocalLoss (focal_loss.py), (alpha, gamma) = (0.25, 2), smooth 1e-5


def layout_loss(lv: LossVec, slot, logits: Var, label: torch.Tensor, sdf, w0, w1, lw, cew, l2w, region="iou"):
    """compute_topview_loss (net.py:554-585): lw*region + cew*CE(w0,w1) + l2w*BD in one fused pass; region =
    opt.loss_type: IoULoss / SoftDiceLoss / TverskyLoss (dice_loss.py:255-372) / FocalLoss (focal_loss.py:7-92)."""
    B, C, h, w = logits.t.shape
    assert C == 2
    ra, ral, rbe = REGION[region]
    sums = _new((8 * B + 3,), logits.t, torch.float64)
    call("jp_layout_loss_fwd", logits.t, label, sdf, sums, lv.val(slot), B, h, w, float(w0), float(w1), float(lw),
         float(cew), float(l2w), ra, ral, rbe)

    def bwd():
        if not logits.rg:
            return
        g, a = logits.grad_buf()
        call("jp_layout_loss_bwd", logits.t, label, sdf, sums, lv.grad(slot), g, B, h, w, float(w0), float(w1),
             float(lw), float(cew), float(l2w), ra, ral, rbe, a)

    _rec(True, bwd)


def l1_loss(lv: LossVec, slot, a: Var, b: Var):
    """nn.L1Loss()(a, b) (net.py:619-622)."""
    n = a.t.numel()
    acc = _new((1,), a.t, torch.float64)
    call("jp_l1_fwd", a.t, b.t, acc, n)
    call("jp_scalar_finalize", acc, lv.val(slot), 1, 1.0 / n)

    def bwd():
        da = torch.empty_like(a.t) if a.rg else None
        db = torch.empty_like(b.t) if b.rg else None
        if da is None and db is None:
            return
        call("jp_l1_bwd", a.t, b.t, lv.grad(slot), 1.0 / n, da, db, n)
        if a.rg:
            a.add_grad(da)
        if b.rg:
            b.add_grad(db)

    _rec(True, bwd)
