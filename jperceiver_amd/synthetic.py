"""Deterministic synthetic weights and 3-frame batches for the JPerceiver train step.

Everything here is a pure function of (seed, name, index) through an integer
hash, so the build container (golden-vector generation against the imported
reference), the CPU tests and the GPU box all regenerate bit-identical
weights and inputs without committing 215 MB of tensors.

Input keys / shapes follow the reference data pipeline
(mono/datasets/mono_dataset.py:84-125,161-171,417-431).
"""
from __future__ import annotations

import zlib
import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        x = x ^ (x >> np.uint64(31))
    return x


def name_key(name) -> int:
    return zlib.crc32(repr(name).encode()) & 0xFFFFFFFF


def hash_uniform(seed: int, name, shape) -> np.ndarray:
    """float32 uniform [0,1) of `shape`, a pure function of (seed, name, flat index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = np.uint64(((int(seed) & 0xFFFFFFFF) << 32) | name_key(name))
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _mix64(idx ^ _mix64(np.full(1, base, dtype=np.uint64)))
    u = (h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
    return u.reshape(shape)


def hash_normal(seed: int, name, shape) -> np.ndarray:
    """float32 standard normal (Box-Muller over two hash streams)."""
    u1 = hash_uniform(seed, (name, "bm1"), shape).astype(np.float64)
    u2 = hash_uniform(seed, (name, "bm2"), shape).astype(np.float64)
    r = np.sqrt(-2.0 * np.log(np.maximum(u1, 2.0 ** -25)))
    return (r * np.cos(2.0 * np.pi * u2)).astype(np.float32)


# --------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------

def _gain(name: str) -> float:
    """Per-layer std gain chosen so that the un-normalised DepthDecoder / CCT stay O(1)
    (sigmoid disparity heads un-saturated, attention energies moderate)."""
    if name.startswith("DepthDecoder."):
        if "_pointwise" in name:
            return 0.22
        if ".disp" in name:
            return 1.2
        return 0.75
    if "query_conv" in name or "key_conv" in name:
        return 0.45
    if name.startswith("CrossViewTransformer"):
        return 0.8
    return 1.13


def synth_tensor_for(name: str, shape, seed: int, bn_stats: bool = False) -> torch.Tensor:
    """Name-keyed deterministic initial value for one state-dict entry.  bn_stats: non-trivial BatchNorm running
    statistics (as after some training) instead of the (0, 1) initial ones -- used by the eval-mode fixtures."""
    shape = tuple(int(s) for s in shape)
    if name.endswith("num_batches_tracked"):
        return torch.zeros((), dtype=torch.long)
    if name.endswith("running_mean"):
        if bn_stats:
            return torch.from_numpy((0.3 * (hash_uniform(seed, name, shape) - 0.5)).astype(np.float32))
        return torch.zeros(shape)
    if name.endswith("running_var"):
        if bn_stats:
            return torch.from_numpy((0.6 + 0.8 * hash_uniform(seed, name, shape)).astype(np.float32))
        return torch.ones(shape)
    u = hash_uniform(seed, name, shape)
    if len(shape) >= 2:                       # conv / linear weight: Kaiming-uniform-like
        fan_in = int(np.prod(shape[1:]))
        a = np.sqrt(3.0) * _gain(name) / np.sqrt(fan_in)
        w = (u * 2.0 - 1.0) * np.float32(a)
        return torch.from_numpy(w.astype(np.float32))
    is_norm_weight = name.endswith(".weight")  # 1-D weight == BatchNorm gamma
    if is_norm_weight:
        return torch.from_numpy((1.0 + 0.2 * (u - 0.5)).astype(np.float32))
    return torch.from_numpy((0.1 * (u - 0.5)).astype(np.float32))  # biases / BN beta


def synth_state_dict(template: dict, seed: int = 0, bn_stats: bool = False) -> dict:
    """template: name -> tensor (only shape/dtype are used)."""
    out = {}
    for name, t in template.items():
        v = synth_tensor_for(name, t.shape, seed, bn_stats)
        out[name] = v.to(t.dtype) if t.dtype != torch.long else v
    return out


# --------------------------------------------------------------------------------------
# inputs
# --------------------------------------------------------------------------------------

def _scene(seed, tag, B, H, W, shift):
    """Smooth RGB scene in [0,1]: low-frequency sinusoids + a little hash noise.
    `shift` (pixels) moves the pattern so neighbouring frames look like camera motion."""
    ys = np.arange(H, dtype=np.float32)[None, None, :, None]
    xs = np.arange(W, dtype=np.float32)[None, None, None, :] + np.float32(shift)
    par = hash_uniform(seed, (tag, "par"), (B, 3, 4, 3))  # 4 waves x (fx, fy, phase)
    img = np.full((B, 3, H, W), 0.5, dtype=np.float32)
    amps = (0.18, 0.12, 0.08, 0.05)
    for k in range(4):
        fx = (par[:, :, k, 0] * 2 - 1)[:, :, None, None] * np.float32((k + 1) * 6.0 / W)
        fy = (par[:, :, k, 1] * 2 - 1)[:, :, None, None] * np.float32((k + 1) * 6.0 / H)
        ph = par[:, :, k, 2][:, :, None, None] * np.float32(2 * np.pi)
        img += np.float32(amps[k]) * np.sin(2 * np.pi * (fx * xs + fy * ys) + ph).astype(np.float32)
    return img


def make_batch(B: int, height: int, width: int, frame_ids=(0, -1, 1), occ: int | None = None,
               full_hw=(375, 1242), split: str = "odometry", seed: int = 1, rank: int = 0,
               step: int = 0) -> dict:
    """One synthetic per-GPU batch with the reference's input dict keys (CPU float32)."""
    occ = occ if occ is not None else height // 4
    sd = (seed * 1000003 + rank * 7919 + step * 104729) & 0x7FFFFFFF
    inp = {}
    FH, FW = full_hw
    for f in frame_ids:
        img = _scene(sd, "scene", B, height, width, shift=3.0 * f)
        img += 0.06 * (hash_uniform(sd, ("noise", f), (B, 3, height, width)) - 0.5)
        img = np.clip(img, 0.0, 1.0).astype(np.float32)
        inp[("color", f, 0)] = torch.from_numpy(img)
        inp[("color_aug", f, 0)] = torch.from_numpy(img.copy())
    # only the *shape* of the full-resolution frame is consumed (net.py:215-219,413)
    inp[("color", 0, -1)] = torch.from_numpy(
        np.clip(_scene(sd, "full", B, FH, FW, 0.0), 0, 1).astype(np.float32))
    K = torch.tensor([[0.58 * width, 0, 0.5 * width, 0],
                      [0, 1.92 * height, 0.5 * height, 0],
                      [0, 0, 1, 0],
                      [0, 0, 0, 1.0]], dtype=torch.float32)
    inp[("K", 0)] = K.repeat(B, 1, 1)
    inp[("inv_K", 0)] = torch.linalg.pinv(K).repeat(B, 1, 1)
    if split == "argo":
        ok = torch.tensor([[1400.0 * FW / 2464.0 * 2.6, 0, FW / 2.0, 0],
                           [0, 1400.0 * FW / 2464.0 * 2.6, FH / 2.0, 0],
                           [0, 0, 1, 0], [0, 0, 0, 1.0]])
        T = torch.eye(4)
        T[:3, :3] = torch.tensor([[0.0, -1, 0], [0, 0, -1], [1, 0, 0]])
        T[:3, 3] = torch.tensor([0.0, 1.5, -1.6])
    else:  # KITTI odometry-like calibration scaled to the full-res frame
        ok = torch.tensor([[718.856 * FW / 1242.0, 0, 607.19 * FW / 1242.0, 0],
                           [0, 718.856 * FH / 375.0, 185.2 * FH / 375.0, 0],
                           [0, 0, 1, 0], [0, 0, 0, 1.0]])
        T = torch.eye(4)
        T[:3, :3] = torch.tensor([[0.0, -1, 0], [0, 0, -1], [1, 0, 0]])
        T[:3, 3] = torch.tensor([0.0, -0.08, -0.27])
    inp[("odometry_K", 0, 0)] = ok.repeat(B, 1, 1)
    inp[("Tr_cam2_velo", 0, 0)] = T.repeat(B, 1, 1)
    # BEV labels: road = big convex-ish blob (~35 %), vehicle = small boxes (~3 %)
    yy, xx = np.meshgrid(np.arange(occ), np.arange(occ), indexing="ij")
    road = np.zeros((B, 1, occ, occ), np.float32)
    veh = np.zeros((B, 1, occ, occ), np.float32)
    pr = hash_uniform(sd, "bev", (B, 8))
    for b in range(B):
        cx = occ * (0.4 + 0.2 * pr[b, 0]); half = occ * (0.18 + 0.1 * pr[b, 1])
        top = occ * (0.15 + 0.2 * pr[b, 2])
        widen = 0.25 + 0.5 * pr[b, 3]
        m = (yy >= top) & (np.abs(xx - cx) <= half * (1 + widen * (yy - top) / occ))
        road[b, 0][m] = 1
        vy = int(occ * (0.45 + 0.3 * pr[b, 4])); vx = int(occ * (0.35 + 0.3 * pr[b, 5]))
        vh = max(2, int(occ * 0.05)); vw = max(2, int(occ * 0.03))
        veh[b, 0, vy:vy + vh, vx:vx + vw] = 1
        vy2 = int(occ * (0.2 + 0.2 * pr[b, 6])); vx2 = int(occ * (0.5 + 0.2 * pr[b, 7]))
        veh[b, 0, vy2:vy2 + vh, vx2:vx2 + vw] = 1
    inp[("bothS", 0, 0)] = torch.from_numpy(road)
    inp[("bothD", 0, 0)] = torch.from_numpy(veh)
    inp[("both_dynamic", 0, 0)] = torch.from_numpy(np.clip(road - veh, 0, 1))
    return inp


def make_dropout_masks(B, height, width, seed=1, rank=0, step=0):
    """Keep-masks (0/1 float) for DepthDecoder's Dropout(0.5) on l4 then l3
    (depth_decoder.py:52-53): shapes (B,512,H/32,W/32) and (B,256,H/16,W/16)."""
    sd = (seed * 1000003 + rank * 7919 + step * 104729) & 0x7FFFFFFF
    m4 = (hash_uniform(sd, "do4", (B, 512, height // 32, width // 32)) >= 0.5).astype(np.float32)
    m3 = (hash_uniform(sd, "do3", (B, 256, height // 16, width // 16)) >= 0.5).astype(np.float32)
    return torch.from_numpy(m4), torch.from_numpy(m3)


def make_automask_noise(B, height, width, n_scales, n_src, seed=1, rank=0, step=0):
    """Standard-normal noise tensors consumed by the automask (net.py:163), in the
    reference's call order: for scale: for source frame. Shape (B,1,H,W) each."""
    sd = (seed * 1000003 + rank * 7919 + step * 104729) & 0x7FFFFFFF
    return [[torch.from_numpy(hash_normal(sd, ("amn", s, j), (B, 1, height, width)))
             for j in range(n_src)] for s in range(n_scales)]
