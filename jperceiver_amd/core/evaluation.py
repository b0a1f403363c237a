"""Evaluation metrics with the reference's names and semantics (mono/core/evaluation/pixel_error.py:27-118,
mono/core/evaluation/eval_hooks.py:147-199), computed by GPU reductions of libjperceiver_hip.so
(csrc/evalmetrics.hip) instead of numpy on host copies:

    compute_errors(gt, pred)            abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3 over two equally shaped tensors
    mean_IU(eval_segm, gt_segm)         per-class IoU list, classes = union of the labels present (pixel_error.py:82-118)
    mean_precision(eval_segm, gt_segm)  per-class precision list, classes = labels present in gt (pixel_error.py:59-79)
    eval_layout(logits, label)          the hook's per-sample layout block (eval_hooks.py:181-199): argmax + both lists
    eval_depth(disp, gt_depth)          the hook's per-sample depth block (eval_hooks.py:147-179): scaled disparity ->
                                        bilinear resize to the ground-truth size -> depth, range mask + Garg crop, median
                                        scaling (or the fixed x36 stereo scale), clamp, compute_errors -> dict incl. 'scale'
    disp_to_depth, AverageMeter         as in pixel_error.py

Segmentation inputs are 2-class maps ({0, 1}; num_class = 2 in every north-star config).  The list-length quirks of the
reference are kept: a class that occurs neither in the prediction nor in the label is simply absent from mean_IU's list.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .._lib import call

MIN_DEPTH = 1e-3
MAX_DEPTH = 80


class AverageMeter(object):
    """pixel_error.py:7-24."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def disp_to_depth(disp, min_depth=0.1, max_depth=100):
    """pixel_error.py:43-48 (elementwise; works on tensors and arrays)."""
    min_disp, max_disp = 1 / max_depth, 1 / min_depth
    scaled_disp = min_disp + (max_disp - min_disp) * disp
    return scaled_disp, 1 / scaled_disp


def _dev(t):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError("evaluation metrics run HIP kernels: pass CUDA tensors")
    return t.contiguous().float()


def _errors_from_sums(s):
    n = s[7]
    if n == 0:
        # no valid pixel (e.g. no LiDAR return inside the Garg crop): the reference's numpy path yields NaNs with a
        # RuntimeWarning (mean of an empty slice, pixel_error.py:27-40) and the evaluation carries on
        import warnings
        warnings.warn("depth metrics over an empty set of valid pixels: returning NaN", RuntimeWarning)
        return (float("nan"),) * 7
    return (s[5] / n, s[6] / n, math.sqrt(s[3] / n), math.sqrt(s[4] / n), s[0] / n, s[1] / n, s[2] / n)


def compute_errors(gt, pred):
    """pixel_error.py:27-40 on two equally shaped CUDA tensors (every element counts)."""
    gt, pred = _dev(gt).reshape(-1), _dev(pred).reshape(-1)
    assert gt.numel() == pred.numel() and gt.numel() > 0
    valid = torch.ones(gt.numel(), device=gt.device, dtype=torch.uint8)
    sums = torch.empty(8, device=gt.device, dtype=torch.float64)
    # fixed_scale 1.0 and a clamp range that never binds: the plain metric definitions
    call("jp_depth_errors", gt, pred, valid, gt.numel(), None, None, 1.0, 0.0, 3.0e38, sums)
    return _errors_from_sums(sums.cpu().tolist())


def _confusion(logits, label):
    logits, label = _dev(logits), _dev(label)
    B, C, h, w = logits.shape
    if C != 2:
        raise NotImplementedError("2-class layouts only (num_class = 2)")
    counts = torch.empty((B, 4), device=logits.device, dtype=torch.float64)
    call("jp_confusion2", logits, label.reshape(B, h * w), counts, B, h * w)
    return counts.cpu().numpy()          # [b][2*pred + true]


def _iu_from_counts(c):
    """mean_IU's list for one image from its confusion counts (pixel_error.py:82-118)."""
    n = {(p, t): c[2 * p + t] for p in (0, 1) for t in (0, 1)}
    out = []
    for k in (0, 1):
        n_eval, n_gt = n[(k, 0)] + n[(k, 1)], n[(0, k)] + n[(1, k)]
        if n_eval == 0 and n_gt == 0:
            continue                      # not in the union of classes
        if n_eval == 0 or n_gt == 0:
            out.append(0)
            continue
        out.append(n[(k, k)] / (n_gt + n_eval - n[(k, k)]))
    return out


def _prec_from_counts(c):
    """mean_precision's list for one image (pixel_error.py:59-79): classes present in gt; 0/0 -> 0."""
    n = {(p, t): c[2 * p + t] for p in (0, 1) for t in (0, 1)}
    out = []
    for k in (0, 1):
        if n[(0, k)] + n[(1, k)] == 0:
            continue
        n_eval = n[(k, 0)] + n[(k, 1)]
        out.append(0. if n_eval == 0 else n[(k, k)] / float(n_eval))
    return out


def _segm_counts(eval_segm, gt_segm):
    e, g = torch.as_tensor(eval_segm), torch.as_tensor(gt_segm)
    if e.shape != g.shape or e.dim() != 2:
        raise ValueError("DiffDim: Different dimensions of matrices!")
    e = e.cuda().float()
    logits = torch.stack([1.0 - e, e], 0).unsqueeze(0)          # argmax reproduces the given 0/1 prediction
    return _confusion(logits, g.cuda().float().reshape(1, 1, *g.shape))[0]


def mean_IU(eval_segm, gt_segm):
    return _iu_from_counts(_segm_counts(eval_segm, gt_segm))


def mean_precision(eval_segm, gt_segm):
    return _prec_from_counts(_segm_counts(eval_segm, gt_segm))


def eval_layout(logits, label):
    """eval_hooks.py:181-199 for a batch: per sample (mean_IU list, mean_precision list) of argmax(logits, 1) vs label."""
    c = _confusion(logits, label)
    return [(_iu_from_counts(ci), _prec_from_counts(ci)) for ci in c]


def eval_depth(disp, gt_depth, stereo_scale=False, min_depth=0.1, max_depth=100, mask_min=MIN_DEPTH, mask_max=MAX_DEPTH,
               stereo_factor=36.0):
    """eval_hooks.py:147-179 for one sample: disp (1,1,h,w) network output, gt_depth (H,W).  mask_min / mask_max /
    stereo_factor default to the hook's constants (1e-3, 80, 36); scripts/eval_depth_eigen.py uses (0.1, 80, 1)."""
    disp, gt = _dev(disp), _dev(gt_depth)
    assert disp.dim() == 4 and disp.shape[:2] == (1, 1) and gt.dim() == 2
    h, w = disp.shape[2:]
    H, W = gt.shape
    scaled = torch.empty_like(disp)
    # scaled_disp = min_disp + (max_disp - min_disp) * disp, then cv2.resize(INTER_LINEAR) == half-pixel bilinear
    call("jp_affine", disp, scaled, disp.numel(), 1.0 / min_depth - 1.0 / max_depth, 1.0 / max_depth)
    res = torch.empty((1, 1, H, W), device=disp.device, dtype=torch.float32)
    call("jp_bilinear_fwd", scaled, res, 1, h, w, H, W)
    crop = np.array([0.40810811 * H, 0.99189189 * H, 0.03594771 * W, 0.96405229 * W]).astype(np.int32)
    pred = torch.empty((H, W), device=disp.device, dtype=torch.float32)
    valid = torch.empty((H, W), device=disp.device, dtype=torch.uint8)
    call("jp_depth_eval_prepare", res, gt, pred, valid, H, W, int(crop[0]), int(crop[1]), int(crop[2]), int(crop[3]),
         float(mask_min), float(mask_max))
    med_g = torch.empty(2, device=disp.device, dtype=torch.float32)
    med_p = torch.empty(2, device=disp.device, dtype=torch.float32)
    call("jp_masked_median", gt, valid, H * W, med_g)
    call("jp_masked_median", pred, valid, H * W, med_p)
    sums = torch.empty(8, device=disp.device, dtype=torch.float64)
    call("jp_depth_errors", gt, pred, valid, H * W, med_g, med_p, float(stereo_factor) if stereo_scale else 0.0,
         float(mask_min), float(mask_max), sums)
    s = sums.cpu().tolist()
    mg, mp = med_g.cpu().tolist(), med_p.cpu().tolist()
    abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3 = _errors_from_sums(s)
    scale = mg[1] / mp[1] if (s[7] > 0 and mp[1] != 0) else float("nan")
    return dict(abs_rel=abs_rel, sq_rel=sq_rel, rmse=rmse, rmse_log=rmse_log, a1=a1, a2=a2, a3=a3, scale=scale,
                n_valid=int(s[7]))
