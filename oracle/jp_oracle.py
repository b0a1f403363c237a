"""ORACLE — test infrastructure only.  NOT part of the product path.

CPU restatement (PyTorch-CPU fp32 ops, functional style over a flat name->tensor
parameter dict) of the reference JPerceiver `Baseline` training step.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this file; `jperceiver_amd/` never does.

Pinning: this restatement is checked against golden vectors produced by the
*imported reference itself* in the build container (tools/make_golden.py ->
tests/golden/*.npz; test: tests/test_oracle_golden.py).  The scale-label
generation (net.py:212-476) depends on un-vendored torchgeometry / torchvision /
cv2 and is "parity unpinned" for those third-party pieces: the
`Argo_both` label path is restated from torchgeometry 0.1.2's documented
semantics and agrees with the harness stub, nothing more.

Every function cites the reference file:line it follows (paths relative to
/root/reference/mono/model/mono_baseline/ unless noted).
"""
from __future__ import annotations

import math
import numpy as np
import torch
import torch.nn.functional as F

try:  # scipy is only needed for the exact EDT of the boundary loss
    from scipy.ndimage import distance_transform_edt as _edt
except Exception:  # pragma: no cover
    _edt = None


# ----------------------------------------------------------------------------------
# parameter inventory (state-dict names == reference attribute paths, net.py:39-60)
# ----------------------------------------------------------------------------------

def _resnet18_shapes(prefix, in_ch):
    """resnet.py:86-121 (BasicBlock x [2,2,2,2]); includes the never-used fc."""
    s = {}
    s[prefix + "conv1.weight"] = (64, in_ch, 7, 7)
    _bn(s, prefix + "bn1", 64)
    inpl = 64
    for li, planes in enumerate([64, 128, 256, 512], start=1):
        for bi in range(2):
            p = f"{prefix}layer{li}.{bi}."
            stride = 2 if (li > 1 and bi == 0) else 1
            s[p + "conv1.weight"] = (planes, inpl, 3, 3)
            _bn(s, p + "bn1", planes)
            s[p + "conv2.weight"] = (planes, planes, 3, 3)
            _bn(s, p + "bn2", planes)
            if stride != 1 or inpl != planes:
                s[p + "downsample.0.weight"] = (planes, inpl, 1, 1)
                _bn(s, p + "downsample.1", planes)
            inpl = planes
    s[prefix + "fc.weight"] = (1000, 512)
    s[prefix + "fc.bias"] = (1000,)
    return s


def _bn(s, name, c):
    s[name + ".weight"] = (c,)
    s[name + ".bias"] = (c,)
    s[name + ".running_mean"] = (c,)
    s[name + ".running_var"] = (c,)
    s[name + ".num_batches_tracked"] = ()


def _conv(s, name, co, ci, k, bias=True):
    s[name + ".weight"] = (co, ci, k, k)
    if bias:
        s[name + ".bias"] = (co,)


def _bev_decoder_shapes(prefix):
    """layout_model.py:130-158: ModuleList order == OrderedDict insertion order."""
    s = {}
    dec = [16, 32, 64, 128, 256]
    idx = 0
    for i in range(4, -1, -1):
        cin = 128 if i == 4 else dec[i + 1]
        co = dec[i]
        _conv(s, f"{prefix}decoder.{idx}", co, cin, 3); idx += 1     # upconv i 0
        _bn(s, f"{prefix}decoder.{idx}", co); idx += 1                # norm i 0
        idx += 1                                                      # relu i 0
        _conv(s, f"{prefix}decoder.{idx}", co, co, 3); idx += 1       # upconv i 1
        _bn(s, f"{prefix}decoder.{idx}", co); idx += 1                # norm i 1
    _conv(s, f"{prefix}decoder.{idx}.conv", 2, 16, 3)                 # topview Conv3x3
    return s


def state_shapes(occ_map_size=256) -> dict:
    """name -> shape for the full Baseline state dict (766 tensors)."""
    s = {}
    s.update(_resnet18_shapes("DepthEncoder.encoder.", 3))
    d = "DepthDecoder."
    _conv(s, d + "reduce4.conv", 512, 512, 1, bias=False)
    _conv(s, d + "reduce3.conv", 256, 256, 1, bias=False)
    _conv(s, d + "reduce2.conv", 256, 128, 1, bias=False)
    _conv(s, d + "reduce1.conv", 256, 64, 1, bias=False)
    _conv(s, d + "iconv4.conv", 256, 512, 3)
    for k in (3, 2, 1):
        _conv(s, d + f"iconv{k}.conv", 256, 513, 3)
    for k in (4, 3, 2, 1):
        for j in range(1, 5):
            _conv(s, d + f"crp{k}.0.{j}_pointwise.conv", 256, 256, 1, bias=False)
    for k in (4, 3, 2, 1):
        _conv(s, d + f"merge{k}.conv", 256, 256, 3)
    for k in (4, 3, 2, 1):
        _conv(s, d + f"disp{k}.0.conv", 1, 256, 3)
    s.update(_resnet18_shapes("PoseEncoder.encoder.", 6))
    p = "PoseDecoder."
    _conv(s, p + "reduce", 256, 512, 1)
    _conv(s, p + "conv1", 256, 256, 3)
    _conv(s, p + "conv2", 256, 256, 3)
    _conv(s, p + "conv3", 6, 256, 1)
    s.update(_resnet18_shapes("LayoutEncoder.resnet_encoder.encoder.", 3))
    _conv(s, "LayoutEncoder.conv1.conv", 128, 512, 3)
    _conv(s, "LayoutEncoder.conv2.conv", 128, 128, 3)
    dim = occ_map_size // 32
    for sfx in ("", "B"):
        for mod in ("transform_module", "retransform_module"):
            for li in (0, 2):
                s[f"CycledViewProjection{sfx}.{mod}.fc_transform.{li}.weight"] = (dim * dim, dim * dim)
                s[f"CycledViewProjection{sfx}.{mod}.fc_transform.{li}.bias"] = (dim * dim,)
        c = f"CrossViewTransformer{sfx}."
        _conv(s, c + "query_conv", 16, 128, 1)
        _conv(s, c + "key_conv", 16, 128, 1)
        _conv(s, c + "value_conv", 128, 128, 1)
        _conv(s, c + "f_conv", 128, 256, 3)
        _conv(s, c + "res_conv", 16, 128, 1)
        _conv(s, c + "query_conv_depth", 16, 128, 1)
        _conv(s, c + "key_conv_depth", 16, 128, 1)
        _conv(s, c + "value_conv_depth", 128, 128, 1)
        _conv(s, c + "conv1.conv", 128, 512, 3)
        _conv(s, c + "conv2.conv", 128, 128, 3)
        s.update(_bev_decoder_shapes(f"LayoutDecoder{sfx}."))
        s.update(_bev_decoder_shapes(f"LayoutTransformDecoder{sfx}."))
    return s


def is_buffer(name):
    return name.endswith(("running_mean", "running_var", "num_batches_tracked"))


class Opt(dict):
    """options object with both attribute and item access (net.py:38,96)."""
    __getattr__ = dict.__getitem__


def default_opt(**kw):
    o = Opt(depth_num_layers=18, pose_num_layers=18, frame_ids=[0, -1, 1], imgs_per_gpu=1,
            height=1024, width=1024, scales=[0, 1, 2, 3], min_depth=0.1, max_depth=100.0,
            depth_pretrained_path=None, pose_pretrained_path=None, automask=True, disp_norm=True,
            smoothness_weight=1e-3, scale_weight=0.1, dynamic_weight=15.0, static_weight=5.0,
            occ_map_size=256, num_class=2, loss_type="iou", loss_weight=20, loss2_type="boundary",
            loss2_weight=20, type="static", loss_sum=3, split="odometry", name="Baseline")
    o.update(kw)
    return o


# ----------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------

class Ctx:
    """Holds parameters (P), buffers (Bf, updated in place like nn.BatchNorm2d) and mode."""

    def __init__(self, P, Bf, training=True):
        self.P, self.Bf, self.training = P, Bf, training


def bn(cx: Ctx, name, x):
    """nn.BatchNorm2d train/eval (resnet.py:21,24; layout_model.py:146,152)."""
    if cx.training:
        cx.Bf[name + ".num_batches_tracked"] += 1
    return F.batch_norm(x, cx.Bf[name + ".running_mean"], cx.Bf[name + ".running_var"],
                        cx.P[name + ".weight"], cx.P[name + ".bias"], cx.training, 0.1, 1e-5)


def conv(cx, name, x, stride=1, pad=0, refl=False):
    """nn.Conv2d, optionally behind ReflectionPad2d(1) (layers.py:156-167)."""
    if refl:
        x = F.pad(x, (1, 1, 1, 1), mode="reflect")
        pad = 0
    return F.conv2d(x, cx.P[name + ".weight"], cx.P.get(name + ".bias"), stride, pad)


def basic_block(cx, p, x, stride, has_ds):
    """resnet.py:29-45."""
    out = F.relu(bn(cx, p + "bn1", conv(cx, p + "conv1", x, stride, 1)))
    out = bn(cx, p + "bn2", conv(cx, p + "conv2", out, 1, 1))
    if has_ds:
        x = bn(cx, p + "downsample.1", conv(cx, p + "downsample.0", x, stride, 0))
    return F.relu(out + x)


def resnet18_features(cx, prefix, img):
    """depth_encoder.py:35-44 / pose_encoder.py:81-92 / ResnetEncoder.py:97-110."""
    x = (img - 0.45) / 0.225
    f0 = F.relu(bn(cx, prefix + "bn1", conv(cx, prefix + "conv1", x, 2, 3)))
    feats = [f0]
    x = F.max_pool2d(f0, 3, 2, 1)
    for li in range(1, 5):
        for bi in range(2):
            first = li > 1 and bi == 0
            x = basic_block(cx, f"{prefix}layer{li}.{bi}.", x, 2 if first else 1, first)
        feats.append(x)
    return feats


def crp(cx, p, x):
    """layers.py:184-199 (4 stages: maxpool5 -> 1x1 -> add)."""
    top = x
    for j in range(1, 5):
        top = F.max_pool2d(top, 5, 1, 2)
        top = conv(cx, f"{p}.0.{j}_pointwise.conv", top)
        x = top + x
    return x


def depth_decoder(cx, feats, drop_masks=None):
    """depth_decoder.py:45-137.  drop_masks = (keep4, keep3) replaces the train-mode
    Dropout(0.5) RNG; value = x * keep * 2."""
    l0, l1, l2, l3, l4 = feats
    if cx.training:
        if drop_masks is None:
            l4 = F.dropout(l4, 0.5, True)
            l3 = F.dropout(l3, 0.5, True)
        else:
            l4 = l4 * drop_masks[0] * 2.0
            l3 = l3 * drop_masks[1] * 2.0
    d = "DepthDecoder."
    out = {}
    x = conv(cx, d + "reduce4.conv", l4)
    x = F.leaky_relu(conv(cx, d + "iconv4.conv", x, refl=True))
    x = crp(cx, d + "crp4", x)
    x = F.leaky_relu(conv(cx, d + "merge4.conv", x, refl=True))
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    disp = torch.sigmoid(conv(cx, d + "disp4.0.conv", x, refl=True))
    out[("disp", 0, 3)] = disp
    for k, lk, sc in ((3, l3, 2), (2, l2, 1), (1, l1, 0)):
        r = conv(cx, d + f"reduce{k}.conv", lk)
        x = torch.cat((r, x, disp), 1)
        x = F.leaky_relu(conv(cx, d + f"iconv{k}.conv", x, refl=True))
        x = crp(cx, d + f"crp{k}", x)
        x = F.leaky_relu(conv(cx, d + f"merge{k}.conv", x, refl=True))
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        disp = torch.sigmoid(conv(cx, d + f"disp{k}.0.conv", x, refl=True))
        out[("disp", 0, sc)] = disp
    return out


def pose_decoder(cx, feats):
    """pose_decoder.py:16-26."""
    p = "PoseDecoder."
    o = F.relu(conv(cx, p + "reduce", feats[-1]))
    o = F.relu(conv(cx, p + "conv1", o, 1, 1))
    o = F.relu(conv(cx, p + "conv2", o, 1, 1))
    o = conv(cx, p + "conv3", o)
    o = 0.01 * o.mean(3).mean(2).view(-1, 1, 1, 6)
    return o[..., :3], o[..., 3:]


def rot_from_axisangle(vec):
    """net.py:727-756.  vec (B,1,3) -> (B,4,4)."""
    angle = torch.norm(vec, 2, 2, True)
    axis = vec / (angle + 1e-7)
    ca, sa = torch.cos(angle), torch.sin(angle)
    C = 1 - ca
    x, y, z = (axis[..., i].unsqueeze(1) for i in range(3))
    xs, ys, zs = x * sa, y * sa, z * sa
    xC, yC, zC = x * C, y * C, z * C
    xyC, yzC, zxC = x * yC, y * zC, z * xC
    B = vec.shape[0]
    rows = [torch.cat([x * xC + ca, xyC - zs, zxC + ys], 2),
            torch.cat([xyC + zs, y * yC + ca, yzC - xs], 2),
            torch.cat([zxC - ys, yzC + xs, z * zC + ca], 2)]
    R = torch.cat(rows, 1)                              # (B,3,3)
    rot = torch.zeros(B, 4, 4, dtype=vec.dtype)
    rot = rot.clone()
    rot[:, :3, :3] = R
    rot[:, 3, 3] = 1
    return rot


def transformation_from_parameters(axisangle, translation, invert=False):
    """net.py:704-725."""
    R = rot_from_axisangle(axisangle)
    t = translation.clone()
    if invert:
        R = R.transpose(1, 2)
        t = t * -1
    B = t.shape[0]
    T = torch.eye(4, dtype=t.dtype).repeat(B, 1, 1)
    T = T.clone()
    T[:, :3, 3] = t.reshape(B, 3)
    return torch.matmul(R, T) if invert else torch.matmul(T, R)


def predict_poses(cx, opt, inputs):
    """net.py:630-642."""
    out = {}
    pf = {f: F.interpolate(inputs[("color_aug", f, 0)], [192, 640], mode="bilinear",
                           align_corners=False) for f in opt.frame_ids}
    for f in opt.frame_ids[1:]:
        pair = [pf[f], pf[0]] if f < 0 else [pf[0], pf[f]]
        feats = resnet18_features(cx, "PoseEncoder.encoder.", torch.cat(pair, 1))
        aa, tr = pose_decoder(cx, feats)
        out[("axisangle", 0, f)] = aa
        out[("translation", 0, f)] = tr
        out[("cam_T_cam", 0, f)] = transformation_from_parameters(aa[:, 0], tr[:, 0], invert=(f < 0))
    return out


def layout_encoder(cx, img):
    """layout_model.py:86-113."""
    x = resnet18_features(cx, "LayoutEncoder.resnet_encoder.encoder.", img)[-1]
    x = F.max_pool2d(conv(cx, "LayoutEncoder.conv1.conv", x, refl=True), 2)
    x = F.max_pool2d(conv(cx, "LayoutEncoder.conv2.conv", x, refl=True), 2)
    return x


def cvp(cx, name, x):
    """CycledViewProjection.py:18-24,54-67."""
    def tm(mod, t):
        B, C, H, W = t.shape
        v = t.reshape(B, C, H * W)
        v = F.relu(F.linear(v, cx.P[f"{name}.{mod}.fc_transform.0.weight"], cx.P[f"{name}.{mod}.fc_transform.0.bias"]))
        v = F.relu(F.linear(v, cx.P[f"{name}.{mod}.fc_transform.2.weight"], cx.P[f"{name}.{mod}.fc_transform.2.bias"]))
        return v.reshape(B, C, H, W)
    t = tm("transform_module", x)
    return t, tm("retransform_module", t)


def _max_dim1(e, forced=None):
    """torch.max(e, dim=1); with `forced` indices the value is gathered instead (tie-tolerant parity:
    a hard arg-max is discrete, so tests replay the device's selection)."""
    if forced is None:
        return torch.max(e, dim=1)
    return torch.gather(e, 1, forced.unsqueeze(1)).squeeze(1), forced


def cct(cx, name, front_x, cross_x, front_x_hat, depth_feature, force=None):
    """CrossViewTransformer.py:45-92."""
    n = name + "."
    df = F.max_pool2d(conv(cx, n + "conv1.conv", depth_feature, refl=True), 2)
    df = F.max_pool2d(conv(cx, n + "conv2.conv", df, refl=True), 2)
    B, C, w, h = front_x.shape
    q = conv(cx, n + "query_conv", cross_x).view(B, -1, w * h)
    k = conv(cx, n + "key_conv", front_x).view(B, -1, w * h).permute(0, 2, 1)
    energy = torch.bmm(k, q)
    force = force or {}
    front_star, arg = _max_dim1(energy, force.get("cv"))
    v = conv(cx, n + "value_conv", front_x_hat).view(B, -1, w * h)
    T = torch.gather(v, 2, arg.view(B, 1, -1).expand(-1, v.shape[1], -1)).view(B, -1, w, h)
    S = front_star.view(B, 1, w, h)
    res = conv(cx, n + "f_conv", torch.cat((front_x, T), 1), 1, 1) * S
    out = front_x + res
    qd = conv(cx, n + "query_conv_depth", cross_x).view(B, -1, w * h)
    kd = conv(cx, n + "key_conv_depth", front_x).view(B, -1, w * h).permute(0, 2, 1)
    vd = conv(cx, n + "value_conv_depth", df).view(B, -1, w, h)
    attn = kd @ qd
    attn, arg_d = _max_dim1(attn, force.get("cm"))
    attn = attn.view(B, 1, w, h)
    out = out + attn @ vd          # (B,1,w,h)@(B,128,w,h): true matrix product, needs w==h
    return out, S, attn, arg, arg_d


def bev_decoder(cx, prefix, x):
    """layout_model.py:160-201 (training branch: raw logits)."""
    idx = 0
    for _ in range(5):
        x = F.relu(bn(cx, f"{prefix}decoder.{idx + 1}", conv(cx, f"{prefix}decoder.{idx}", x, 1, 1)))
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = bn(cx, f"{prefix}decoder.{idx + 4}", conv(cx, f"{prefix}decoder.{idx + 3}", x, 1, 1))
        idx += 5
    # raw logits in train AND eval: Baseline.predict_layout calls the decoder with its default is_training=True
    # (net.py:644-689), so the Softmax2d branch of layout_model.py:194-199 never runs inside the model
    return conv(cx, f"{prefix}decoder.{idx}.conv", x, refl=True)


def predict_layout(cx, inputs, depth_feature, sfx="", features=None, force=None):
    """net.py:644-689."""
    o = {}
    if features is None:
        features = layout_encoder(cx, inputs[("color_aug", 0, 0)])
    enc = features
    t, r = cvp(cx, "CycledViewProjection" + sfx, features)
    tag = "car" if sfx == "B" else "road"
    fc = None
    if force is not None and ("cv_argmax_" + tag) in force:
        fc = {"cv": force["cv_argmax_" + tag], "cm": force["cm_argmax_" + tag]}
    feats, S, attn, arg, arg_d = cct(cx, "CrossViewTransformer" + sfx, features, t, r, depth_feature[-1], fc)
    o["topview" + sfx] = bev_decoder(cx, f"LayoutDecoder{sfx}.", feats)
    o["transform_topview" + sfx] = bev_decoder(cx, f"LayoutTransformDecoder{sfx}.", t)
    o["features" + sfx] = feats
    o["retransform_features" + sfx] = r
    o["transform_feature_" + tag] = t
    o["cv_attn_" + tag] = S
    o["cm_attn_" + tag] = attn
    o["cv_argmax_" + tag] = arg
    o["cm_argmax_" + tag] = arg_d
    if sfx == "":
        o["origin_features"] = enc
    return o, enc


# ----------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------

def ssim(x, y):
    """layers.py:97-107."""
    x = F.pad(x, (1, 1, 1, 1), mode="reflect")
    y = F.pad(y, (1, 1, 1, 1), mode="reflect")
    mu_x, mu_y = F.avg_pool2d(x, 3, 1), F.avg_pool2d(y, 3, 1)
    sx = F.avg_pool2d(x * x, 3, 1) - mu_x ** 2
    sy = F.avg_pool2d(y * y, 3, 1) - mu_y ** 2
    sxy = F.avg_pool2d(x * y, 3, 1) - mu_x * mu_y
    n = (2 * mu_x * mu_y + 0.01 ** 2) * (2 * sxy + 0.03 ** 2)
    d = (mu_x ** 2 + mu_y ** 2 + 0.01 ** 2) * (sx + sy + 0.03 ** 2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


def reprojection_loss(pred, target):
    """net.py:84-92."""
    l1 = torch.sqrt((target - pred) ** 2 + 1e-3 ** 2).mean(1, True)
    return 0.85 * ssim(pred, target).mean(1, True) + 0.15 * l1


def disp_to_depth(disp, min_depth, max_depth):
    """layers.py:33-38."""
    min_disp, max_disp = 1 / max_depth, 1 / min_depth
    sd = min_disp + (max_disp - min_disp) * disp
    return sd, 1 / sd


def backproject(depth, inv_K):
    """layers.py:41-61."""
    B, _, H, W = depth.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(H * W)], 0).unsqueeze(0).repeat(B, 1, 1)
    cam = torch.matmul(inv_K[:, :3, :3], pix)
    cam = depth.view(B, 1, -1) * cam
    return torch.cat([cam, torch.ones(B, 1, H * W)], 1)


def project(points, K, T, H, W):
    """layers.py:73-82."""
    B = points.shape[0]
    P = torch.matmul(K, T)[:, :3, :]
    cam = torch.matmul(P, points)
    pix = cam[:, :2, :] / (cam[:, 2, :].unsqueeze(1) + 1e-7)
    pix = pix.view(B, 2, H, W).permute(0, 2, 3, 1)
    pix = torch.stack([pix[..., 0] / (W - 1), pix[..., 1] / (H - 1)], -1)
    return (pix - 0.5) * 2


def generate_images_pred(opt, inputs, outputs, scale):
    """net.py:690-702 (grid_sample: bilinear, border, align_corners default False, N7)."""
    H, W = opt.height, opt.width
    disp = F.interpolate(outputs[("disp", 0, scale)], [H, W], mode="bilinear", align_corners=False)
    _, depth = disp_to_depth(disp, opt.min_depth, opt.max_depth)
    for f in opt.frame_ids[1:]:
        T = outputs[("cam_T_cam", 0, f)]
        cam = backproject(depth, inputs[("inv_K", 0)])
        grid = project(cam, inputs[("K", 0)], T, H, W)
        outputs[("color", f, scale)] = F.grid_sample(inputs[("color", f, 0)], grid, mode="bilinear",
                                                     padding_mode="border", align_corners=False)
    return outputs


def gradient_xy(D):
    """net.py:783-786."""
    return D[:, :, :, 1:] - D[:, :, :, :-1], D[:, :, 1:] - D[:, :, :-1]


def smooth_loss(disp, img):
    """net.py:758-781."""
    b, _, h, w = disp.shape
    img = F.interpolate(img, (h, w), mode="area")
    ddx, ddy = gradient_xy(disp)
    idx, idy = gradient_xy(img)
    dxx, dxy = gradient_xy(ddx)
    dyx, dyy = gradient_xy(ddy)
    ixx, ixy = gradient_xy(idx)
    iyx, iyy = gradient_xy(idy)
    def term(d, i, a=0.5):
        return torch.mean(d.abs() * torch.exp(-a * i.abs().mean(1, True)))
    return term(ddx, idx) + term(ddy, idy) + term(dxx, ixx) + term(dxy, ixy) + term(dyx, iyx) + term(dyy, iyy)


def scale_loss(opt, depth_pred, scale_label):
    """net.py:193-211."""
    shape = scale_label.shape[2:4]
    dp = torch.clamp(F.interpolate(depth_pred, shape, mode="bilinear", align_corners=False), 1e-3, 80)
    mask = scale_label > 0
    if opt["type"] == "static_raw":
        crop = torch.zeros_like(mask)
        crop[:, :, 153:371, 44:1197] = 1
        mask = mask * crop
    gt = torch.masked_select(scale_label, mask)
    pr = torch.masked_select(dp, mask)
    return torch.mean(torch.abs(gt - pr) / gt)


def find_inner_boundary(mask: np.ndarray) -> np.ndarray:
    """skimage.segmentation.find_boundaries(mode='inner', connectivity=1) restated:
    foreground pixels with a 4-neighbour of a different label (image borders do not count).
    boundary_loss.py:142.  Third-party rule: parity unpinned."""
    m = mask.astype(bool)
    diff = np.zeros_like(m)
    diff[:-1, :] |= m[:-1, :] != m[1:, :]
    diff[1:, :] |= m[1:, :] != m[:-1, :]
    diff[:, :-1] |= m[:, :-1] != m[:, 1:]
    diff[:, 1:] |= m[:, 1:] != m[:, :-1]
    return diff & m


def compute_sdf(onehot: np.ndarray) -> np.ndarray:
    """boundary_loss.py:121-147: SDF = EDT(~m) - EDT(m), 0 on the inner boundary,
    all-zero for an empty mask; channel 0 left zero; float64."""
    out = np.zeros(onehot.shape, dtype=np.float64)
    for b in range(onehot.shape[0]):
        for c in range(1, onehot.shape[1]):
            pos = onehot[b, c].astype(bool)
            if pos.any():
                sdf = _edt(~pos) - _edt(pos)
                sdf[find_inner_boundary(pos)] = 0
                out[b, c] = sdf
    return out


def bd_loss(logits, gt):
    """boundary_loss.py:160-192.  gt (B,H,W) long."""
    p = F.softmax(logits, 1)
    oh = torch.zeros_like(p).scatter_(1, gt.unsqueeze(1), 1)
    phi = torch.from_numpy(compute_sdf(oh.detach().numpy()))       # float64, as in the reference
    return (p[:, 1:] * phi[:, 1:]).mean()


def iou_loss(logits, gt):
    """dice_loss.py:308-331 with get_tp_fp_fn :31-81 (smooth=1, per-(b,c), do_bg)."""
    p = F.softmax(logits, 1)
    oh = torch.zeros_like(p).scatter_(1, gt.unsqueeze(1), 1)
    tp = (p * oh).sum((2, 3))
    fp = (p * (1 - oh)).sum((2, 3))
    fn = ((1 - p) * oh).sum((2, 3))
    return -((tp + 1.0) / (tp + fp + fn + 1.0)).mean()


def region_loss(logits, gt, a=1.0, alpha=1.0, beta=1.0):
    """dice_loss.py's overlap losses in one formula, -mean_{b,c} (a*tp + 1) / (a*tp + alpha*fp + beta*fn + 1):
    IoULoss (1,1,1) :293-331, SoftDiceLoss (2,1,1) :255-290, TverskyLoss (1,0.3,0.7) :333-372."""
    p = F.softmax(logits, 1)
    oh = torch.zeros_like(p).scatter_(1, gt.unsqueeze(1), 1)
    tp = (p * oh).sum((2, 3))
    fp = (p * (1 - oh)).sum((2, 3))
    fn = ((1 - p) * oh).sum((2, 3))
    return -((a * tp + 1.0) / (a * tp + alpha * fp + beta * fn + 1.0)).mean()


def focal_loss(logits, gt, alpha=0.25, gamma=2.0, smooth=1e-5):
    """focal_loss.py:36-92 as built by net.py:568-570 (softmax non-linearity, float alpha with balance_index 0,
    size_average): pt = sum_c clamp(onehot_c, smooth/(C-1), 1-smooth) * p_c + smooth."""
    p = F.softmax(logits, 1)
    C = p.shape[1]
    pf = p.permute(0, 2, 3, 1).reshape(-1, C)
    t = gt.reshape(-1, 1)
    al = torch.full((C,), 1.0 - alpha)
    al[0] = alpha
    oh = torch.zeros_like(pf).scatter_(1, t, 1).clamp(smooth / (C - 1), 1.0 - smooth)
    pt = (oh * pf).sum(1) + smooth
    return (-al[t.squeeze(1)] * (1 - pt) ** gamma * pt.log()).mean()


_REGION = {"iou": (1.0, 1.0, 1.0), "dice": (2.0, 1.0, 1.0), "tversky": (1.0, 0.3, 0.7)}


def _loss_type(opt, logits, gt):
    """net.py:562-573: opt.loss_type picks the region / focal term."""
    ty = opt.get("loss_type", "iou")
    return focal_loss(logits, gt) if ty == "focal" else region_loss(logits, gt, *_REGION[ty])


def topview_loss(opt, logits, label, class_weight, wS=True):
    """net.py:554-617.  loss_weightS/loss2_weightS fall back to loss_weight/loss2_weight (N2)."""
    gt = label.long().squeeze(1)
    lw = opt.get("loss_weightS", opt["loss_weight"]) if wS else opt["loss_weight"]
    l2w = opt.get("loss2_weightS", opt["loss2_weight"]) if wS else opt["loss2_weight"]
    if opt["loss_sum"] == 1:
        out = _loss_type(opt, logits, gt) * lw
    elif opt["loss_sum"] == 2:
        out = _loss_type(opt, logits, gt) * lw + bd_loss(logits, gt) * l2w
    else:  # 3 (and the reference-undefined 0)
        ce = F.cross_entropy(logits, gt, weight=torch.tensor([1.0, float(class_weight)]))
        out = _loss_type(opt, logits, gt) * lw + ce + bd_loss(logits, gt) * l2w
    return out.mean()


# ----------------------------------------------------------------------------------
# scale label (Argo_both path; net.py:400-476) — third-party pieces restated
# ----------------------------------------------------------------------------------

def _normal_transform_pixel(h, w):
    return torch.tensor([[2.0 / (w - 1), 0, -1], [0, 2.0 / (h - 1), -1], [0, 0, 1.0]]).unsqueeze(0)


def warp_perspective(src, M, dsize):
    """torchgeometry 0.1.2 `warp_perspective` (unpinned third party):
    normalise by (size-1), invert, grid_sample(bilinear, zeros, align_corners=False here)."""
    B, C, H, W = src.shape
    h, w = dsize
    dst_norm = _normal_transform_pixel(h, w) @ M @ torch.inverse(_normal_transform_pixel(H, W))
    src_from_dst = torch.inverse(dst_norm)
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, w), indexing="ij")
    grid = torch.stack([xs, ys], -1).view(1, -1, 2).expand(B, -1, -1)
    gh = F.pad(grid, (0, 1), value=1.0)
    g = torch.matmul(src_from_dst.unsqueeze(1), gh.unsqueeze(-1)).squeeze(-1)
    g = (g[..., :-1] / g[..., -1:]).view(B, h, w, 2)
    return F.grid_sample(src, g, mode="bilinear", padding_mode="zeros", align_corners=False)


def scale_label_homography(opt, inputs):
    """inverse(shiftedground_H_img) of net.py:432-471 as a (B,3,3) tensor."""
    mapsize = opt.occ_map_size
    resolution = 40 / mapsize
    K = inputs[("odometry_K", 0, 0)][:, :3, :3]
    B = K.shape[0]
    Tcv = inputs[("Tr_cam2_velo", 0, 0)]
    hfg = 0.33 if opt.split == "argo" else 1.73
    ego_from_ground = torch.eye(4).repeat(B, 1, 1)
    ego_from_ground[:, :3, 3] = -torch.tensor([0.0, 0.0, hfg])     # inverse of (I, [0,0,h])
    cam_from_ground = torch.bmm(Tcv, ego_from_ground)
    r1, r2, t = cam_from_ground[:, :3, 0:1], cam_from_ground[:, :3, 1:2], cam_from_ground[:, :3, 3:4]
    img_H_ground = torch.bmm(K, torch.cat([r1, r2, t], 2))
    ground_H_img = torch.linalg.inv(img_H_ground)
    out_width = int(40 / resolution)
    shift = int(out_width // 2)
    sg = torch.tensor([[1 / resolution, 0, 0], [0, 1 / resolution, shift], [0, 0, 1.0]]).repeat(B, 1, 1)
    return torch.linalg.inv(torch.bmm(sg, ground_H_img))


def scale_label_both(opt, inputs):
    """net.py:400-476 (`type == "Argo_both"`)."""
    mapsize = opt.occ_map_size
    H, W = inputs[("color", 0, -1)].shape[2:4]
    lay = inputs[("both_dynamic", 0, 0)]
    B = lay.shape[0]
    off = 1.9 if opt.split == "argo" else 0.27
    z = torch.arange(mapsize, 0, step=-1).view(1, 1, mapsize, 1).repeat(B, 1, 1, mapsize) * (40 / mapsize) - off
    # torch.fliplr flips dim 1 (size 1) -> no-op; rotate(270) == rot90(k=3) on the last two dims
    lay = torch.rot90(lay, 3, (-2, -1))
    z = torch.rot90(z, 3, (-2, -1))
    M = scale_label_homography(opt, inputs)
    return warp_perspective(z, M, (H, W)) * warp_perspective(lay, M, (H, W))


def transform_points(T, pts):
    """torchgeometry 0.1.2 `transform_points` (unpinned third party): homogeneous matmul + perspective divide."""
    ph = F.pad(pts, (0, 1), value=1.0)
    out = torch.matmul(T.unsqueeze(1), ph.unsqueeze(-1)).squeeze(-1)
    return out[..., :-1] / out[..., -1:]


def _assumption_quad(opt, M):
    """net.py:235-248,292-299 (static) == :329-341,386-393 (dynamic): the 4 m x 2 m "assumption region" in front
    of the ego car, rotated into the BEV frame, projected with batch item 0's homography, rounded (half-to-even,
    torch.round) and re-ordered to the polygon order [0, 2, 3, 1] the reference hands to cv2.fillConvexPoly."""
    mapsize = opt.occ_map_size
    r1 = mapsize / 40
    pr = [(round(18 * r1), round(31 * r1)), (round(22 * r1), round(31 * r1)),
          (round(18 * r1), round(33 * r1)), (round(22 * r1), round(33 * r1))]
    rot = [[mapsize - pr[3][1] - 1, pr[0][0] - 1],
           [mapsize - pr[3][1] + (pr[2][1] - pr[1][1]) - 1, pr[0][0] - 1],
           [mapsize - pr[3][1] - 1, pr[1][0] - 1],
           [mapsize - pr[3][1] + (pr[2][1] - pr[1][1]) - 1, pr[1][0] - 1]]
    B = M.shape[0]
    pts = torch.tensor(np.asarray(rot), dtype=torch.float32).repeat(B, 1, 1)
    new = torch.round(transform_points(M, pts)).int()
    p = new[0]
    return np.array([[p[0][0], p[0][1]], [p[2][0], p[2][1]], [p[3][0], p[3][1]], [p[1][0], p[1][1]]]).astype(np.int32)


def _assumption_mask(opt, M, H, W):
    """net.py:300-305: fillConvexPoly(zeros(H,W,3) uint8, pts, (0,255,255), lineType=1) -> RGB2GRAY -> > 0."""
    from . import cv2_restated as cv2
    pts = _assumption_quad(opt, M).reshape((-1, 1, 2))
    img = cv2.fillConvexPoly(np.zeros((H, W, 3), np.uint8), pts, (0, 255, 255), 1)
    return cv2.cvtColor(img, cv2.COLOR_RGB2GRAY) > 0


def _distance_label(opt, B, offset):
    mapsize = opt.occ_map_size
    z = torch.arange(mapsize, 0, step=-1).view(1, 1, mapsize, 1).repeat(B, 1, 1, mapsize) * (40 / mapsize) - offset
    return torch.rot90(z, 3, (-2, -1))     # fliplr on dim 1 (size 1) is a no-op; rotate(270) == rot90(k=3)


def scale_label_static(opt, inputs, return_parts=False):
    """net.py:212-310 (`get_scale_label_static`): warped distance label x uint8(warped road layout) & filled
    assumption quad.  `.type_as(uint8)` truncates the bilinearly warped {0,1} layout, i.e. keeps the pixels whose
    interpolated value reaches 1.0 in the evaluating platform's fp32 arithmetic (rounding-sensitive, see
    tests/test_scale_label.py for the band that is treated as ambiguous)."""
    H, W = inputs[("color", 0, -1)].shape[2:4]
    lay = inputs[("bothS", 0, 0)]
    B = lay.shape[0]
    off = 1.9 if opt.split == "argo" else 0.27
    z = _distance_label(opt, B, off)
    lay = torch.rot90(lay, 3, (-2, -1))
    M = scale_label_homography(opt, inputs)
    zw = warp_perspective(z, M, (H, W))
    lw = warp_perspective(lay, M, (H, W))
    tri = torch.from_numpy(_assumption_mask(opt, M, H, W).astype(np.uint8) * 255).repeat(B, 1, 1).unsqueeze(1)
    a_and_b = torch.bitwise_and(lw.type_as(tri), tri)
    out = zw * a_and_b
    return (out, zw, lw, tri) if return_parts else out


def scale_label_dynamic(opt, inputs, return_parts=False):
    """net.py:311-399 (`get_scale_label_dynamic`): warped distance label x filled assumption quad.  Note the
    non-Argo distance label carries NO -0.27 offset here (net.py:328, commented out in the reference)."""
    H, W = inputs[("color", 0, -1)].shape[2:4]
    B = inputs[("bothS", 0, 0)].shape[0]
    off = 1.9 if opt.split == "argo" else 0.0
    z = _distance_label(opt, B, off)
    M = scale_label_homography(opt, inputs)
    zw = warp_perspective(z, M, (H, W))
    tri = torch.from_numpy(_assumption_mask(opt, M, H, W).astype(np.uint8)).repeat(B, 1, 1).unsqueeze(1)
    out = zw * tri
    return (out, zw, None, tri) if return_parts else out


def make_scale_label(opt, inputs):
    """net.py:139-144 dispatch on opt.type."""
    ty = opt["type"]
    if ty == "Argo_both":
        return scale_label_both(opt, inputs)
    if ty in ("dynamic", "Argo_dynamic"):
        return scale_label_dynamic(opt, inputs)
    return scale_label_static(opt, inputs)


# ----------------------------------------------------------------------------------
# the step
# ----------------------------------------------------------------------------------

def forward(P, Bf, opt, inputs, training=True, drop_masks=None, automask_noise=None, scale_label=None, force=None):
    """Baseline.forward + compute_losses (net.py:68-192), type-conditional layout losses
    per the root net.py:125-159.  Layout branch computed once; BN buffers of
    LayoutEncoder / LayoutDecoder / LayoutTransformDecoder get the reference's *double*
    momentum update (N4) by evaluating the branch twice under no_grad the second time.
    Returns (outputs, loss_dict)."""
    cx = Ctx(P, Bf, training)
    feats = resnet18_features(cx, "DepthEncoder.encoder.", inputs[("color_aug", 0, 0)])
    outputs = depth_decoder(cx, feats, drop_masks)
    if not opt.get("layout_branch", True):
        # NOT a reference option: the shape-agnostic sub-path only (depth + pose + CGT warp + photometric / smoothness /
        # scale losses, net.py:139-211,630-642,690-702,758-786) for non-square inputs such as BASELINE.json's 1024x320,
        # where the reference's CVP / CCT (square maps only) cannot run.  Pinned by tests/golden/subpath_320x1024_b2.npz.
        assert training
        outputs.update(predict_poses(cx, opt, inputs))
        return outputs, compute_losses(opt, inputs, outputs, automask_noise, scale_label, force)
    o, enc = predict_layout(cx, inputs, feats, "", force=force)
    outputs.update(o)
    if training:
        with torch.no_grad():      # net.py:74 duplicate call: only its BN-buffer side effect matters
            predict_layout(cx, inputs, [f.detach() for f in feats], "")
    outputs.update(predict_layout(cx, inputs, feats, "B", features=enc, force=force)[0])
    if not training:
        return outputs
    outputs.update(predict_poses(cx, opt, inputs))
    return outputs, compute_losses(opt, inputs, outputs, automask_noise, scale_label, force)


def compute_losses(opt, inputs, outputs, automask_noise=None, scale_label=None, force=None):
    """net.py:94-192 with root-net.py:125-159 type conditionals."""
    L = {}
    ty = opt["type"]
    do_S = ty in ("static_raw", "static", "Argo_static", "Argo_both", "static_eigen")
    do_B = ty in ("dynamic", "Argo_dynamic", "Argo_both")
    if not opt.get("layout_branch", True):
        do_S = do_B = False
    if scale_label is None:
        scale_label = make_scale_label(opt, inputs)
    if do_S:
        L["topview_loss"] = topview_loss(opt, outputs["topview"], inputs[("bothS", 0, 0)], opt.static_weight, True)
        L["transform_topview_loss"] = topview_loss(opt, outputs["transform_topview"], inputs[("bothS", 0, 0)], opt.static_weight, True)
        L["transform_loss"] = F.l1_loss(outputs["features"], outputs["retransform_features"])
        L["layout_loss"] = L["topview_loss"] + 0.001 * L["transform_loss"] + L["transform_topview_loss"]
    if do_B:
        L["topview_lossB"] = topview_loss(opt, outputs["topviewB"], inputs[("bothD", 0, 0)], opt.dynamic_weight, False)
        L["transform_topview_lossB"] = topview_loss(opt, outputs["transform_topviewB"], inputs[("bothD", 0, 0)], opt.dynamic_weight, False)
        L["transform_lossB"] = F.l1_loss(outputs["featuresB"], outputs["retransform_featuresB"])
        L["layout_lossB"] = L["topview_lossB"] + 0.001 * L["transform_lossB"] + L["transform_topview_lossB"]
    nS = len(opt.scales)
    target = inputs[("color", 0, 0)]
    for si, scale in enumerate(opt.scales):
        disp = outputs[("disp", 0, scale)]
        _, depth = disp_to_depth(disp, opt.min_depth, opt.max_depth)
        outputs[("depth", 0, scale)] = depth
        generate_images_pred(opt, inputs, outputs, scale)
        cands = []
        if opt.automask:
            for j, f in enumerate(opt.frame_ids[1:]):
                idl = reprojection_loss(inputs[("color", f, 0)], target)
                nz = automask_noise[si][j] if automask_noise is not None else torch.randn(idl.shape)
                cands.append(idl + nz * 1e-5)
        for f in opt.frame_ids[1:]:
            cands.append(reprojection_loss(outputs[("color", f, scale)], target))
        m, outputs[("min_index", scale)] = _max_dim1(-torch.cat(cands, 1), None if force is None else force.get(("min_index", scale)))
        m = -m
        L[("min_reconstruct_loss", scale)] = m.mean() / nS
        L[("scale_loss", scale)] = opt.scale_weight * scale_loss(opt, depth, scale_label) / (2 ** scale) / nS
        if opt.disp_norm:
            disp = disp / (disp.mean(2, True).mean(3, True) + 1e-7)
        L[("smooth_loss", scale)] = opt.smoothness_weight * smooth_loss(disp, target) / (2 ** scale) / nS
    return L


def total_loss(loss_dict):
    """trainer.py:35-46: sum of every entry (layout terms double-counted)."""
    return sum(v.mean() for v in loss_dict.values())


def make_params(shapes: dict, state: dict):
    """split a state dict into (params requiring grad, buffers)."""
    P, Bf = {}, {}
    for n in shapes:
        t = state[n].clone()
        if is_buffer(n):
            Bf[n] = t
        else:
            P[n] = t.requires_grad_(True)
    return P, Bf


def adam_step(P, state, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, max_norm=35.0):
    """dist_utils.py:58-60: clip_grad_norm_(35, L2) then torch.optim.Adam(lr=1e-4, wd=0)."""
    grads = [p.grad for p in P.values() if p.grad is not None]
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads))
    coef = min(1.0, max_norm / (total + 1e-6))
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    with torch.no_grad():
        for n, p in P.items():
            if p.grad is None:
                continue
            g = p.grad * coef
            m = state.setdefault(("m", n), torch.zeros_like(p))
            v = state.setdefault(("v", n), torch.zeros_like(p))
            m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
            v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
            bc1, bc2 = 1 - betas[0] ** t, 1 - betas[1] ** t
            p.addcdiv_(m, (v.sqrt() / math.sqrt(bc2)).add_(eps), value=-lr / bc1)
    return total
