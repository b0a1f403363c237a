#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference)
on CPU in the build container.  Never runs on the GPU box; no reference source is copied —
this file only stubs the third-party packages the reference imports but the image lacks
(cv2, imageio, pykitti, skimage, torchvision, torchgeometry) and feeds it
the deterministic synthetic weights/inputs of jperceiver_amd/synthetic.py.

Usage:  python tools/make_golden.py [case ...]     (default: all cases)
"""
import importlib.util
import os
import sys
import types
import zlib

import numpy as np
import scipy.ndimage as ndi
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jperceiver_amd import synthetic as syn  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


# ------------------------------------------------------------------ reference import harness
def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def _load(name, file):
    spec = importlib.util.spec_from_file_location(name, file)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def import_reference():
    np.bool = bool                                   # boundary_loss.py:137 (numpy>=1.24)
    torch.Tensor.cuda = lambda self, *a, **k: self   # hard .cuda() calls all over net.py/layers.py
    nn.Module.cuda = lambda self, *a, **k: self
    _stub("imageio"); _stub("pykitti")
    # cv2 is absent from the image: the two calls net.py makes (fillConvexPoly, cvtColor) are served by the numpy
    # restatement of OpenCV's published algorithm (oracle/cv2_restated.py; third party, parity unpinned)
    from oracle import cv2_restated
    sys.modules["cv2"] = cv2_restated

    def find_boundaries(img, mode="inner"):          # skimage rule, connectivity=1
        img = np.asarray(img).astype(np.uint8)
        fp = ndi.generate_binary_structure(2, 1)
        b = ndi.grey_dilation(img, footprint=fp) != ndi.grey_erosion(img, footprint=fp)
        return b & (img != 0)
    sk = _stub("skimage")
    sk.segmentation = _stub("skimage.segmentation", find_boundaries=find_boundaries)
    base = REF + "/mono/model/mono_baseline"
    _pkg("mono", REF + "/mono"); _pkg("mono.model", REF + "/mono/model"); _pkg("mono.model.mono_baseline", base)
    _load("mono.model.registry", REF + "/mono/model/registry.py")
    res = _load("mono.model.mono_baseline.resnet", base + "/resnet.py")
    tv = _stub("torchvision")
    tvm = _stub("torchvision.models", ResNet=res.ResNet, resnet18=lambda pretrained=False: res.resnet18(),
                resnet34=None, resnet50=None, resnet101=None, resnet152=None)
    tvm.resnet = types.SimpleNamespace(BasicBlock=res.BasicBlock, Bottleneck=res.Bottleneck, model_urls={})

    def rotate(x, angle):
        assert angle == 270
        return torch.rot90(x, k=3, dims=(-2, -1))
    tv.models = tvm
    tv.transforms = _stub("torchvision.transforms", functional=types.SimpleNamespace(rotate=rotate))

    def transform_points(T, pts):                    # torchgeometry 0.1.2 semantics
        ph = F.pad(pts, (0, 1), value=1.0)
        out = torch.matmul(T.unsqueeze(1), ph.unsqueeze(-1)).squeeze(-1)
        return out[..., :-1] / out[..., -1:]

    def npix(h, w):
        return torch.tensor([[2.0 / (w - 1), 0, -1], [0, 2.0 / (h - 1), -1], [0, 0, 1.0]]).unsqueeze(0)

    def warp_perspective(src, M, dsize):
        B, C, H, W = src.shape
        h, w = dsize
        sntd = torch.inverse(npix(h, w) @ M @ torch.inverse(npix(H, W)))
        ys, xs = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, w), indexing="ij")
        grid = torch.stack([xs, ys], -1).view(1, -1, 2).expand(B, -1, -1)
        return F.grid_sample(src, transform_points(sntd, grid).view(B, h, w, 2), mode="bilinear", padding_mode="zeros")
    _stub("torchgeometry"); _stub("torchgeometry.core")
    _stub("torchgeometry.core.imgwarp", warp_perspective=warp_perspective)
    _stub("torchgeometry.core.transformations", transform_points=transform_points)
    return _load("mono.model.mono_baseline.net", base + "/net.py")


class Opt(dict):
    __getattr__ = dict.__getitem__


class FixedDropout(nn.Module):
    """stands in for DepthDecoder.do (depth_decoder.py:13,52-53): x * keep * 1/(1-p)."""

    def __init__(self, masks):
        super().__init__()
        self.masks, self.i = masks, 0

    def forward(self, x):
        m = self.masks[self.i % len(self.masks)]
        self.i += 1
        return x * m * 2.0


# ------------------------------------------------------------------ helpers
def crc(t):
    return zlib.crc32(np.ascontiguousarray(t.detach().numpy()).tobytes()) & 0xFFFFFFFF


def pool_to(t, n=16):
    """adaptive-average to at most n x n (size-independent fingerprint of a map)."""
    t = t.detach().float()
    return F.adaptive_avg_pool2d(t, (min(n, t.shape[-2]), min(n, t.shape[-1]))).numpy()


def run_case(net, name, HW, B, FR, ty, split, full_hw, seed=1):
    occ = HW // 4
    opt = Opt(depth_num_layers=18, pose_num_layers=18, frame_ids=FR, imgs_per_gpu=B, height=HW, width=HW,
              scales=[0, 1, 2, 3], min_depth=0.1, max_depth=100.0, depth_pretrained_path=None,
              pose_pretrained_path=None, automask=True, disp_norm=True, smoothness_weight=1e-3,
              scale_weight=0.1, dynamic_weight=15., static_weight=5., occ_map_size=occ, num_class=2,
              loss_type="iou", loss_weight=20, loss_weightS=20, loss2_type="boundary", loss2_weight=20,
              loss2_weightS=20, type=ty, loss_sum=3, split=split)
    torch.manual_seed(0)
    model = net.Baseline(opt)
    sd = syn.synth_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(sd, strict=True)
    model.train()
    inp = syn.make_batch(B, HW, HW, FR, occ, full_hw, split, seed=seed)
    m4, m3 = syn.make_dropout_masks(B, HW, HW, seed=seed)
    model.DepthDecoder.do = FixedDropout([m4, m3])
    noise = syn.make_automask_noise(B, HW, HW, 4, len(FR) - 1, seed=seed)
    flat = [n.clone() for per_scale in noise for n in per_scale]
    it = iter(flat)
    real_randn = torch.randn
    torch.randn = lambda *a, **k: next(it)
    # capture the scale label the reference computes (it is an *input* of the parity-checked step)
    cap = {}
    orig = model.get_scale_label_both

    def grab(inputs, o):
        cap["scale_label"] = orig(inputs, o)
        return cap["scale_label"]
    model.get_scale_label_both = grab
    try:
        out, losses = model({k: v.clone() for k, v in inp.items()})
    finally:
        torch.randn = real_randn
    total = sum(v.mean() for v in losses.values())
    total.backward()

    g = {}
    for k, v in losses.items():
        g["loss/" + repr(k)] = np.float64(v.detach().double().item())
    g["loss/total"] = np.float64(total.detach().double().item())
    for f in FR[1:]:
        g[f"cam_T_cam/{f}"] = out[("cam_T_cam", 0, f)].detach().numpy()
    for s in range(4):
        d = out[("disp", 0, s)]
        g[f"disp{s}/pool"] = pool_to(d)
        g[f"disp{s}/mean"] = np.float64(d.double().mean().item())
        g[f"disp{s}/first"] = d.detach().numpy()[:, :, :8, :8].copy()
        g[f"min_index{s}/hist"] = np.bincount(out[("min_index", s)].reshape(-1).numpy(), minlength=4)
        for f in FR[1:]:
            g[f"color{f}_{s}/pool"] = pool_to(out[("color", f, s)])
    for k in ("topview", "transform_topview", "topviewB", "transform_topviewB"):
        g[k + "/pool"] = pool_to(out[k])
        g[k + "/first"] = out[k].detach().numpy()[:, :, :8, :8].copy()
    for k in ("features", "featuresB", "retransform_features", "retransform_featuresB",
              "transform_feature_road", "transform_feature_car", "cv_attn_road", "cm_attn_road",
              "cv_attn_car", "cm_attn_car", "origin_features"):
        g["feat/" + k] = out[k].detach().numpy()
    g["scale_label/pool"] = pool_to(cap["scale_label"], 32)
    g["scale_label/nnz"] = np.int64((cap["scale_label"] > 0).sum().item())
    # gradients: L2 norm per top-level module, per parameter, and a few probed entries
    mods = {}
    for n, p in model.named_parameters():
        top = n.split(".")[0]
        if p.grad is None:
            g["gradnone/" + n] = np.int64(1)
            continue
        gn = float(p.grad.double().pow(2).sum())
        mods[top] = mods.get(top, 0.0) + gn
        g["gradnorm/" + n] = np.float64(np.sqrt(gn))
        g["gradprobe/" + n] = p.grad.reshape(-1)[:4].detach().numpy().copy()
    for top, v in mods.items():
        g["gradnorm_module/" + top] = np.float64(np.sqrt(v))
    # BN buffer side effects (N4)
    for n, b in model.named_buffers():
        if n.endswith("num_batches_tracked"):
            g["nbt/" + n] = np.int64(b.item())
    for n in ("DepthEncoder.encoder.bn1.running_mean", "LayoutEncoder.resnet_encoder.encoder.bn1.running_mean",
              "LayoutDecoder.decoder.1.running_var", "LayoutDecoderB.decoder.1.running_var",
              "PoseEncoder.encoder.bn1.running_var"):
        g["buf/" + n] = dict(model.named_buffers())[n].detach().numpy().copy()
    meta = dict(HW=HW, B=B, FR=FR, type=ty, split=split, full_hw=list(full_hw), seed=seed, occ=occ)
    g["meta"] = np.array(repr(meta))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **g)
    print(name, "total loss", g["loss/total"], "keys", len(g))
    return model, inp, out, losses, cap


def subpath_case(net, name="subpath_320x1024_b2", H=320, W=1024, B=2, FR=(0, -1, 1), seed=4, full_hw=(514, 616)):
    """BASELINE.json's 1024(W) x 320(H): the reference's shape-agnostic sub-path run by the REFERENCE's own modules and its
    own `compute_losses` (net.py:94-192) -- DepthEncoder, DepthDecoder, predict_poses (:630-642), generate_images_pred
    (:690-702), compute_reprojection_loss (:84-92), automask + min (:159-175), get_scale_label_both (:400-476),
    get_scale_loss (:194-211), get_smooth_loss (:758-786).  The BEV-layout branch cannot run on a non-square input
    (CycledViewProjection / CrossViewTransformer flatten square maps), so `predict_layout*` is not
    called; `compute_losses` still reads their outputs, which are fed as constants and whose loss entries are discarded."""
    FR = list(FR)
    occ = 256
    opt = Opt(depth_num_layers=18, pose_num_layers=18, frame_ids=FR, imgs_per_gpu=B, height=H, width=W,
              scales=[0, 1, 2, 3], min_depth=0.1, max_depth=100.0, depth_pretrained_path=None,
              pose_pretrained_path=None, automask=True, disp_norm=True, smoothness_weight=1e-3,
              scale_weight=0.1, dynamic_weight=15., static_weight=5., occ_map_size=occ, num_class=2,
              loss_type="iou", loss_weight=20, loss_weightS=20, loss2_type="boundary", loss2_weight=20,
              loss2_weightS=20, type="Argo_both", loss_sum=3, split="argo")
    torch.manual_seed(0)
    model = net.Baseline(opt)
    model.load_state_dict(syn.synth_state_dict(model.state_dict(), seed=0), strict=True)
    model.train()
    inp = syn.make_batch(B, H, W, FR, occ, full_hw, "argo", seed=seed)
    m4, m3 = syn.make_dropout_masks(B, H, W, seed=seed)
    model.DepthDecoder.do = FixedDropout([m4, m3])
    noise = syn.make_automask_noise(B, H, W, 4, len(FR) - 1, seed=seed)
    it = iter([n.clone() for per_scale in noise for n in per_scale])
    real_randn = torch.randn
    torch.randn = lambda *a, **k: next(it)
    cap = {}
    orig = model.get_scale_label_both

    def grab(inputs, o):
        cap["scale_label"] = orig(inputs, o)
        return cap["scale_label"]
    model.get_scale_label_both = grab
    try:
        d = {k: v.clone() for k, v in inp.items()}
        out = model.DepthDecoder(model.DepthEncoder(d["color_aug", 0, 0]))
        const = torch.zeros(B, 2, occ, occ)
        feat = torch.zeros(B, 128, occ // 32, occ // 32)
        for k in ("topview", "transform_topview", "topviewB", "transform_topviewB"):
            out[k] = const
        for k in ("features", "retransform_features", "featuresB", "retransform_featuresB"):
            out[k] = feat
        out.update(model.predict_poses(d))
        losses = model.compute_losses(d, out)
    finally:
        torch.randn = real_randn
    losses = {k: v for k, v in losses.items() if isinstance(k, tuple)}        # the 12 depth-path terms
    total = sum(v.mean() for v in losses.values())
    total.backward()
    g = {}
    for k, v in losses.items():
        g["loss/" + repr(k)] = np.float64(v.detach().double().item())
    g["loss/total"] = np.float64(total.detach().double().item())
    for f in FR[1:]:
        g[f"cam_T_cam/{f}"] = out[("cam_T_cam", 0, f)].detach().numpy()
    for s in range(4):
        dsp = out[("disp", 0, s)]
        g[f"disp{s}/pool"] = pool_to(dsp)
        g[f"disp{s}/first"] = dsp.detach().numpy()[:, :, :8, :8].copy()
        g[f"disp{s}/last"] = dsp.detach().numpy()[:, :, -8:, -8:].copy()
        g[f"min_index{s}/hist"] = np.bincount(out[("min_index", s)].reshape(-1).numpy(), minlength=4)
        for f in FR[1:]:
            g[f"color{f}_{s}/pool"] = pool_to(out[("color", f, s)])
    g["scale_label/pool"] = pool_to(cap["scale_label"], 32)
    g["scale_label/nnz"] = np.int64((cap["scale_label"] > 0).sum().item())
    # the label itself (mostly zeros: compresses well).  It is an INPUT of the parity-checked step, and the abs-rel scale loss
    # divides by it: the few near-zero label values at the edge of the warped layout are decided by the evaluating host's
    # rounding (another x86 host's PyTorch moves scale_loss by ~1 %), so the GPU test feeds THIS label to both sides
    g["scale_label/full"] = cap["scale_label"].detach().numpy().astype(np.float32)
    mods = {}
    for n, p in model.named_parameters():
        top = n.split(".")[0]
        if p.grad is None:
            g["gradnone/" + n] = np.int64(1)
            continue
        gn = float(p.grad.double().pow(2).sum())
        mods[top] = mods.get(top, 0.0) + gn
        g["gradnorm/" + n] = np.float64(np.sqrt(gn))
        g["gradprobe/" + n] = p.grad.reshape(-1)[:4].detach().numpy().copy()
    for top, v in mods.items():
        g["gradnorm_module/" + top] = np.float64(np.sqrt(v))
    for n in ("DepthEncoder.encoder.bn1.running_mean", "DepthEncoder.encoder.layer4.1.bn2.running_var",
              "PoseEncoder.encoder.bn1.running_var"):
        g["buf/" + n] = dict(model.named_buffers())[n].detach().numpy().copy()
    meta = dict(H=H, W=W, B=B, FR=FR, type="Argo_both", split="argo", full_hw=list(full_hw), seed=seed, occ=occ)
    g["meta"] = np.array(repr(meta))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **g)
    print(name, "total loss", g["loss/total"], "keys", len(g), {k: float(v) for k, v in losses.items()})


def unit_vectors(net):
    """Small op-level known-answer vectors straight from the reference's layers/losses."""
    layers = sys.modules["mono.model.mono_baseline.layers"]
    bl = sys.modules["mono.model.mono_baseline.boundary_loss"]
    dl = sys.modules["mono.model.mono_baseline.dice_loss"]
    g = {}
    x = torch.from_numpy(syn.hash_uniform(7, "ssim_x", (2, 3, 16, 16)))
    y = torch.from_numpy(syn.hash_uniform(7, "ssim_y", (2, 3, 16, 16)))
    y = 0.7 * x + 0.3 * y
    g["ssim/out"] = layers.SSIM()(x, y).numpy()
    # rot_from_axisangle / transformation_from_parameters on 8 vectors incl. zero (net.py:704-756)
    vec = torch.from_numpy((syn.hash_uniform(7, "aa", (8, 1, 3)) - 0.5) * 0.2)
    vec[0] = 0
    tr = torch.from_numpy((syn.hash_uniform(7, "tr", (8, 1, 3)) - 0.5))
    B = net.Baseline
    dummy = types.SimpleNamespace()
    for nm in ("rot_from_axisangle", "get_translation_matrix", "transformation_from_parameters"):
        setattr(dummy, nm, types.MethodType(getattr(B, nm), dummy))
    g["pose/rot"] = dummy.rot_from_axisangle(vec).numpy()
    g["pose/M"] = dummy.transformation_from_parameters(vec, tr, invert=False).numpy()
    g["pose/Minv"] = dummy.transformation_from_parameters(vec, tr, invert=True).numpy()
    # SDF on hand-made masks: empty / blob / ring / single pixel / full (boundary_loss.py:121-147)
    n = 48
    masks = np.zeros((5, 2, n, n), np.float32)
    yy, xx = np.mgrid[:n, :n]
    masks[1, 1] = ((yy - 20) ** 2 + (xx - 25) ** 2 < 100)
    r2 = (yy - 24) ** 2 + (xx - 24) ** 2
    masks[2, 1] = (r2 < 300) & (r2 > 90)
    masks[3, 1, 10, 30] = 1
    masks[4, 1] = 1
    masks[:, 0] = 1 - masks[:, 1]
    g["sdf/out"] = bl.compute_sdf(masks, masks.shape)
    # IoU / BD / CE losses on random logits vs a blob label (dice_loss.py:293-331, boundary_loss.py:150-192)
    logits = torch.from_numpy((syn.hash_uniform(7, "logits", (5, 2, n, n)) - 0.5) * 4)
    gt = torch.from_numpy(masks[:, 1]).long()
    g["loss/iou"] = np.float64(dl.IoULoss(apply_nonlin=lambda t: F.softmax(t, 1))(logits, gt).item())
    sm = lambda t: F.softmax(t, 1)          # noqa: E731
    g["loss/dice"] = np.float64(dl.SoftDiceLoss(apply_nonlin=sm)(logits, gt).item())
    g["loss/tversky"] = np.float64(dl.TverskyLoss(apply_nonlin=sm)(logits, gt).item())
    fl = sys.modules["mono.model.mono_baseline.focal_loss"]
    g["loss/focal"] = np.float64(fl.FocalLoss(apply_nonlin=sm)(logits, gt).item())
    g["loss/bd"] = np.float64(bl.BDLoss()(logits, gt).item())
    g["loss/ce"] = np.float64(nn.CrossEntropyLoss(weight=torch.tensor([1.0, 5.0]))(logits, gt).item())
    # Backproject / Project / grid_sample (layers.py:41-82, net.py:690-702) on a tiny case
    Bn, H, W = 2, 12, 20
    K = torch.tensor([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * H, 0.5 * H, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]]).repeat(Bn, 1, 1)
    invK = torch.linalg.pinv(K)
    depth = 1.0 / (0.01 + 9.99 * torch.from_numpy(syn.hash_uniform(7, "d", (Bn, 1, H, W))))
    T = dummy.transformation_from_parameters(vec[1:3], tr[1:3] * 0.3, invert=False)
    cam = layers.Backproject(Bn, H, W)(depth, invK)
    pix = layers.Project(Bn, H, W)(cam, K, T)
    img = torch.from_numpy(syn.hash_uniform(7, "img", (Bn, 3, H, W)))
    g["warp/grid"] = pix.numpy()
    g["warp/out"] = F.grid_sample(img, pix, padding_mode="border").numpy()
    np.savez_compressed(os.path.join(OUT, "unit_vectors.npz"), **g)
    print("unit_vectors keys", len(g))


def eval_case(net, name="eval_256_b2", HW=256, B=2, seed=7):
    """model.eval() forward of the REFERENCE (BatchNorm running statistics, Dropout off, raw logits out of
    predict_layout) on synthetic weights with non-trivial running statistics -> disparities, layout logits, features."""
    occ = HW // 4
    opt = Opt(depth_num_layers=18, pose_num_layers=18, frame_ids=[0, -1, 1], imgs_per_gpu=B, height=HW, width=HW,
              scales=[0, 1, 2, 3], min_depth=0.1, max_depth=100.0, depth_pretrained_path=None, pose_pretrained_path=None,
              automask=True, disp_norm=True, smoothness_weight=1e-3, scale_weight=0.1, dynamic_weight=15., static_weight=5.,
              occ_map_size=occ, num_class=2, loss_type="iou", loss_weight=20, loss_weightS=20, loss2_type="boundary",
              loss2_weight=20, loss2_weightS=20, type="Argo_both", loss_sum=3, split="argo")
    torch.manual_seed(0)
    model = net.Baseline(opt)
    model.load_state_dict(syn.synth_state_dict(model.state_dict(), seed=0, bn_stats=True), strict=True)
    model.eval()
    inp = syn.make_batch(B, HW, HW, [0, -1, 1], occ, (257, 308), "argo", seed=seed)
    with torch.no_grad():
        out = model({k: v.clone() for k, v in inp.items()})
    g = {"meta": np.array(repr(dict(HW=HW, B=B, seed=seed, occ=occ)))}
    for s_ in range(4):
        d = out[("disp", 0, s_)]
        g[f"disp{s_}/pool"] = pool_to(d)
        g[f"disp{s_}/first"] = d.numpy()[:, :, :8, :8].copy()
    for k in ("topview", "transform_topview", "topviewB", "transform_topviewB"):
        g[k + "/pool"] = pool_to(out[k])
        g[k + "/first"] = out[k].numpy()[:, :, :8, :8].copy()
        g[k + "/argmax_frac"] = np.float64(out[k].argmax(1).float().mean().item())
    for k in ("features", "featuresB", "retransform_features", "origin_features"):
        g["feat/" + k] = out[k].numpy()
    nbt = dict(model.named_buffers())["DepthEncoder.encoder.bn1.num_batches_tracked"]
    g["nbt_unchanged"] = np.int64(nbt.item())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **g)
    print(name, "keys", len(g), "topview argmax frac", g["topview/argmax_frac"])


def eval_metric_vectors():
    """Known answers of the reference's metric functions (mono/core/evaluation/pixel_error.py, imported as is) and of the
    eval hook's per-sample depth block (eval_hooks.py:147-179, restated here line by line with cv2.resize ==
    half-pixel bilinear because the hook itself needs mmcv / cv2)."""
    pe = _load("mono_core_pixel_error", REF + "/mono/core/evaluation/pixel_error.py")
    g = {}
    rng = np.random.default_rng(5)
    n = 40
    cases = {"mixed": (rng.random((n, n)) > 0.6, rng.random((n, n)) > 0.7), "pred_empty": (np.zeros((n, n), bool), rng.random((n, n)) > 0.5),
             "gt_empty": (rng.random((n, n)) > 0.5, np.zeros((n, n), bool)), "both_empty": (np.zeros((n, n), bool), np.zeros((n, n), bool)),
             "both_full": (np.ones((n, n), bool), np.ones((n, n), bool))}
    for k, (p, t) in cases.items():
        g[f"seg/{k}/pred"], g[f"seg/{k}/true"] = p.astype(np.uint8), t.astype(np.uint8)
        g[f"seg/{k}/iu"] = np.asarray(pe.mean_IU(p.astype(np.int64), t.astype(np.int64)), np.float64)
        g[f"seg/{k}/prec"] = np.asarray(pe.mean_precision(p.astype(np.int64), t.astype(np.int64)), np.float64)
    gt = (rng.random((50, 70)) * 60 + 1).astype(np.float32)
    pr = (gt * (0.7 + 0.6 * rng.random((50, 70)))).astype(np.float32)
    g["err/gt"], g["err/pred"] = gt, pr
    g["err/out"] = np.asarray(pe.compute_errors(gt, pr), np.float64)
    # the hook's depth block on a synthetic disparity / sparse ground truth
    disp = torch.from_numpy(syn.hash_uniform(9, "evdisp", (1, 1, 48, 160)))
    H, W = 94, 311
    gtd = (rng.random((H, W)) * 90).astype(np.float32)
    gtd[rng.random((H, W)) > 0.35] = 0
    g["depth/disp"], g["depth/gt"] = disp.numpy(), gtd
    pred_disp, _ = pe.disp_to_depth(disp)
    pred_disp = F.interpolate(pred_disp, (H, W), mode="bilinear", align_corners=False)[0, 0].numpy()   # == cv2.resize INTER_LINEAR
    pred_depth = 1 / pred_disp
    mask = np.logical_and(gtd > 1e-3, gtd < 80)
    crop = np.array([0.40810811 * H, 0.99189189 * H, 0.03594771 * W, 0.96405229 * W]).astype(np.int32)
    cm = np.zeros(mask.shape)
    cm[crop[0]:crop[1], crop[2]:crop[3]] = 1
    mask = np.logical_and(mask, cm)
    pd, gd = pred_depth[mask], gtd[mask]
    ratio = np.median(gd) / np.median(pd)
    for tag, scale in (("median", ratio), ("stereo", 36.0)):
        q = pd * scale
        q[q < 1e-3] = 1e-3
        q[q > 80] = 80
        g[f"depth/{tag}/errors"] = np.asarray(pe.compute_errors(gd, q), np.float64)
    g["depth/ratio"], g["depth/n_valid"] = np.float64(ratio), np.int64(mask.sum())
    np.savez_compressed(os.path.join(OUT, "eval_metrics.npz"), **g)
    print("eval_metrics keys", len(g), "ratio", ratio, "n_valid", int(mask.sum()))


def sampler_vectors():
    """Index sequences emitted by the REFERENCE's own sampler classes (mono/datasets/loader/sampler.py, imported as is)."""
    sm = _load("mono_datasets_sampler", REF + "/mono/datasets/loader/sampler.py")

    class DS:
        def __init__(self, flag):
            self.flag = np.asarray(flag, dtype=np.int64)

        def __len__(self):
            return len(self.flag)
    g = {}
    cases = {"one_group_103": np.zeros(103, np.int64), "two_groups": np.array([0] * 37 + [1] * 22, np.int64)[np.random.default_rng(0).permutation(59)]}
    for name, flag in cases.items():
        g[f"{name}/flag"] = flag
        for world, spg in ((1, 3), (4, 3), (8, 1), (2, 8)):
            for epoch in (0, 5):
                for rank in range(world):
                    s_ = sm.DistributedGroupSampler(DS(flag), spg, world, rank)
                    s_.set_epoch(epoch)
                    g[f"{name}/dgs/w{world}_s{spg}_e{epoch}_r{rank}"] = np.asarray(list(iter(s_)), np.int64)
        for world in (1, 3):
            for shuffle in (True, False):
                for rank in range(world):
                    s_ = sm.DistributedSampler(DS(flag), world, rank, shuffle=shuffle)
                    s_.set_epoch(2)
                    g[f"{name}/ds/w{world}_sh{int(shuffle)}_r{rank}"] = np.asarray(list(iter(s_)), np.int64)
        np.random.seed(11)
        g[f"{name}/gs"] = np.asarray([int(v) for v in iter(sm.GroupSampler(DS(flag), 4))], np.int64)
    np.savez_compressed(os.path.join(OUT, "sampler.npz"), **g)
    print("sampler keys", len(g))


def preprocess_vectors():
    """Pillow's own output for the resizes of MonoDataset.preprocess (mono/datasets/mono_dataset.py:83-92,133-147:
    transforms.Resize(size, Image.ANTIALIAS) == Image.resize(..., LANCZOS) on 8-bit RGB) and the reference's
    process_topview / process_topview_both (:417-431), executed line by line with PIL on hash-generated images."""
    from PIL import Image
    g = {}
    for name, (H, W, OH, OW) in {"down": (94, 311, 64, 64), "up": (37, 50, 80, 120), "mixed": (60, 96, 90, 48), "same_w": (50, 64, 20, 64)}.items():
        img = (syn.hash_uniform(21, ("pp", name), (H, W, 3)) * 256).astype(np.uint8)
        g[f"resize/{name}/shape"] = np.array([H, W, OH, OW])
        g[f"resize/{name}/out"] = np.asarray(Image.fromarray(img).resize((OW, OH), Image.LANCZOS))
    # two-stage resize of the pipeline: raw -> full-res (375x1242 scaled down here) -> network size
    img = (syn.hash_uniform(21, ("pp", "chain"), (80, 200, 3)) * 256).astype(np.uint8)
    full = Image.fromarray(img).resize((124, 38), Image.LANCZOS)
    g["resize/chain/full"] = np.asarray(full)
    g["resize/chain/net"] = np.asarray(full.resize((64, 64), Image.LANCZOS))
    for name, (h, w, S) in {"sq": (128, 128, 32), "rect": (100, 150, 64), "up": (20, 24, 32)}.items():
        lab = ((syn.hash_uniform(22, ("tv", name), (h, w)) > 0.55) * 255).astype(np.uint8)
        lab3 = np.stack([lab, lab, lab], -1)
        for mode, arr in (("L", lab), ("RGB", lab3)):
            tv = Image.fromarray(arr).convert("1").resize((S, S), Image.NEAREST).convert("L")       # process_topview
            a = np.array(tv)
            o = np.zeros(a.shape)
            o[a == 255] = 1
            g[f"topview/{name}/{mode}"] = o.astype(np.uint8)
        tb = np.array(Image.fromarray(lab).resize((S, S), Image.NEAREST))                         # process_topview_both
        o = np.zeros(tb.shape)
        o[tb == 255] = 1
        g[f"topview_both/{name}"] = o.astype(np.uint8)
        g[f"topview/{name}/shape"] = np.array([h, w, S])
    np.savez_compressed(os.path.join(OUT, "preprocess.npz"), **g)
    print("preprocess keys", len(g))


def scale_label_cases(net):
    """get_scale_label_static / get_scale_label_dynamic (net.py:212-402) run by the REFERENCE on the synthetic
    calibration (third-party pieces stubbed as above) -> tests/golden/scale_labels.npz: the 0/1 support as packed
    bits, 32x32 pooled values, sums, and the polygon the reference handed to cv2.fillConvexPoly."""
    from oracle import cv2_restated
    g = {}
    cases = {"static_odom": ("static", "odometry", (375, 1242), 256, 2, 1),
             "dynamic_odom": ("dynamic", "odometry", (375, 1242), 256, 2, 1),
             "static_argo_small": ("static", "argo", (514, 616), 128, 2, 2),
             "static_argo_full": ("static", "argo", (2056, 2464), 256, 1, 3),
             "dynamic_argo_full": ("dynamic", "argo", (2056, 2464), 256, 1, 3)}
    for name, (ty, split, fhw, occ, B, seed) in cases.items():
        HW = occ * 4
        opt = Opt(occ_map_size=occ, split=split, type=ty)
        inp = syn.make_batch(B, HW, HW, [0], occ, fhw, split, seed=seed)
        dummy = types.SimpleNamespace()
        for nm in ("get_scale_label_static", "get_scale_label_dynamic", "SE3", "homography_from_calibration"):
            setattr(dummy, nm, types.MethodType(getattr(net.Baseline, nm), dummy))
        seen = {}
        real = cv2_restated.fillConvexPoly

        def spy(img, pts, color, lineType=8, shift=0):
            seen["pts"], seen["lineType"], seen["color"] = np.array(pts).reshape(-1, 2).copy(), lineType, tuple(color)
            return real(img, pts, color, lineType, shift)
        cv2_restated.fillConvexPoly = spy
        try:
            fn = dummy.get_scale_label_static if ty == "static" else dummy.get_scale_label_dynamic
            lab = fn({k: v.clone() for k, v in inp.items()}, opt).float()
        finally:
            cv2_restated.fillConvexPoly = real
        g[name + "/meta"] = np.array(repr(dict(type=ty, split=split, full_hw=list(fhw), occ=occ, B=B, seed=seed)))
        g[name + "/pts"] = seen["pts"].astype(np.int32)
        g[name + "/lineType"] = np.int64(seen["lineType"])
        g[name + "/n_nonfinite"] = np.int64((~torch.isfinite(lab)).sum().item())   # horizon pixels: x/0 in the warp
        g[name + "/support"] = np.packbits((lab > 0).numpy().reshape(-1))
        lab = torch.nan_to_num(lab, nan=0.0, posinf=0.0, neginf=0.0)
        g[name + "/pool"] = pool_to(lab, 32)
        g[name + "/sum"] = np.float64(lab.double().sum().item())
        g[name + "/nnz"] = np.int64((lab > 0).sum().item())
        g[name + "/rowsum"] = lab.double().sum(-1).numpy().astype(np.float32)
        print(name, "nnz", g[name + "/nnz"], "sum", g[name + "/sum"], "pts", seen["pts"].tolist())
    np.savez_compressed(os.path.join(OUT, "scale_labels.npz"), **g)


CASES = {
    # name: (HW, B, frames, type, split, full-res frame (shrunk), seed)
    "argo_both_256_b2": (256, 2, [0, -1, 1], "Argo_both", "argo", (257, 308), 1),
    "argo_both_512_b2": (512, 2, [0, -1, 1], "Argo_both", "argo", (514, 616), 2),
    "argo_both_1024_b1": (1024, 1, [0, -1], "Argo_both", "argo", (514, 616), 3),
}

if __name__ == "__main__":
    net = import_reference()
    want = sys.argv[1:] or (["unit", "scale_labels", "eval", "eval_metrics", "sampler", "preprocess", "subpath"] + list(CASES))
    for c in want:
        if c == "unit":
            unit_vectors(net)
        elif c == "eval":
            eval_case(net)
        elif c == "eval_metrics":
            eval_metric_vectors()
        elif c == "sampler":
            sampler_vectors()
        elif c == "preprocess":
            preprocess_vectors()
        elif c == "scale_labels":
            scale_label_cases(net)
        elif c == "subpath":
            subpath_case(net)
        else:
            run_case(net, c, *CASES[c][:6], seed=CASES[c][6])
