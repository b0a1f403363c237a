"""Inference entry points with the behaviour of the reference's demo scripts, on the HIP kernels:

    evaluate_depth(model, batches, gt_depths, stereo_scale=False)
        scripts/eval_depth_eigen.py:24-113: eval-mode forward -> ("disp", 0, 0) -> disp_to_depth(0.1, 100) -> resize to the
        ground truth -> 1/disp, range mask (0.1 .. 80 m) + Garg crop, per-image median scaling (or the fixed stereo
        factor 1), clamp, the seven depth metrics; returns (mean errors, median ratio, std of ratios / median).
    pose_between(pose_encoder, pose_decoder, img_a, img_b)
        the 4x4 transform `transformation_from_parameters(axisangle[:, 0], translation[:, 0])` of
        scripts/draw_odometry.py:69-71 for one pair (both images concatenated on channels, frame 0 first).
    chain_poses(transforms)
        scripts/draw_odometry.py:62-76: global_pose <- global_pose @ inv(T_k), rows 0..2 flattened -> (n+1, 12)
        (the KITTI odometry text format written by the script).
    odometry(pose_encoder, pose_decoder, frames)
        both of the above over a sequence of frames: (n, 3, H, W) -> (n, 12).
The networks must already hold a checkpoint (apis.load_checkpoint; the scripts copy `PoseEncoder.*` / `PoseDecoder.*`
out of the training checkpoint's state dict, which `pose_nets_from_checkpoint` restates)."""
from __future__ import annotations

import numpy as np
import torch

from .._lib import call
from ..core import evaluation as ev


def _eval(*mods):
    for m in mods:
        if m.training:
            raise RuntimeError("inference helpers expect eval-mode networks (call .eval(): BatchNorm must use running stats)")


@torch.no_grad()
def evaluate_depth(model, batches, gt_depths, stereo_scale=False, min_depth=0.1, max_depth=80.0):
    """batches: iterable of input dicts with ("color_aug"|"color", 0, 0) (1,3,H,W) CUDA tensors; gt_depths: sequence of
    (h, w) arrays / tensors in metres (0 where there is no LiDAR return)."""
    _eval(model)
    errors, ratios = [], []
    for inputs, gt in zip(batches, gt_depths):
        disp = model(inputs)[("disp", 0, 0)]
        gt = torch.as_tensor(np.asarray(gt) if not isinstance(gt, torch.Tensor) else gt, dtype=torch.float32).to(disp.device)
        for b in range(disp.shape[0]):
            r = ev.eval_depth(disp[b:b + 1], gt if gt.dim() == 2 else gt[b], stereo_scale=stereo_scale, min_depth=0.1,
                              max_depth=100, mask_min=min_depth, mask_max=max_depth, stereo_factor=1.0)
            errors.append([r[k] for k in ("abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3")])
            ratios.append(r["scale"])
    ratios = np.asarray(ratios)
    med = float(np.median(ratios))
    return dict(zip(("abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3"), np.asarray(errors).mean(0).tolist())), med, \
        float(np.std(ratios / med))


@torch.no_grad()
def pose_between(pose_encoder, pose_decoder, img_a, img_b):
    """(B,3,H,W) x 2 -> (B,4,4): the transform the pose head predicts for the pair [img_a | img_b] (no inversion)."""
    _eval(pose_encoder, pose_decoder)
    axisangle, translation = pose_decoder(pose_encoder(torch.cat([img_a, img_b], 1).contiguous()))
    aa, tr = axisangle[:, 0].reshape(-1, 3).contiguous(), translation[:, 0].reshape(-1, 3).contiguous()
    B = aa.shape[0]
    K = torch.eye(4, device=aa.device).repeat(B, 1, 1)                 # only T is used (P = K@T is a by-product)
    T = torch.empty((B, 4, 4), device=aa.device)
    P = torch.empty((B, 3, 4), device=aa.device)
    call("jp_pose_fwd", aa, tr, K, T, P, B, 0)
    return T


def chain_poses(transforms) -> np.ndarray:
    """(n,4,4) frame-to-frame transforms -> (n+1, 12) global poses, float64 on the host like the script."""
    T = np.asarray(transforms.detach().cpu().numpy() if isinstance(transforms, torch.Tensor) else transforms, dtype=np.float64)
    g = np.identity(4)
    out = [g[0:3, :].reshape(1, 12)]
    for k in range(T.shape[0]):
        g = g @ np.linalg.inv(T[k])
        out.append(g[0:3, :].reshape(1, 12))
    return np.concatenate(out, 0)


@torch.no_grad()
def odometry(pose_encoder, pose_decoder, frames) -> np.ndarray:
    """frames (n,3,H,W) CUDA tensor (already at the pose nets' resolution) -> (n, 12) chained poses."""
    Ts = [pose_between(pose_encoder, pose_decoder, frames[k:k + 1], frames[k + 1:k + 2])[0] for k in range(frames.shape[0] - 1)]
    return chain_poses(torch.stack(Ts)) if Ts else np.identity(4)[0:3].reshape(1, 12)


def pose_nets_from_checkpoint(checkpoint, pose_encoder, pose_decoder):
    """scripts/draw_odometry.py:52-56: copy `PoseEncoder.*` / `PoseDecoder.*` of a training checkpoint into the nets."""
    sd = checkpoint["state_dict"] if "state_dict" in checkpoint else checkpoint
    for prefix, net in (("PoseEncoder.", pose_encoder), ("PoseDecoder.", pose_decoder)):
        own = net.state_dict()
        missing = [n for n in own if prefix + n not in sd]
        if missing:
            raise KeyError(f"checkpoint lacks {prefix}{missing[0]} (+{len(missing) - 1} more)")
        net.load_state_dict({n: sd[prefix + n] for n in own}, strict=True)
    return pose_encoder, pose_decoder
