"""Host-side mirror of the reference sub-modules of `Baseline` (mono/model/mono_baseline/*):
same class names, constructor arguments, attribute names and therefore state-dict keys, same
forward signatures — but every forward launches the HIP kernels of libjperceiver_hip.so through
`ops` (no torch.nn.functional compute anywhere).  torch.nn layers are used only as parameter
containers (shapes, default init, state-dict layout).

Each module exposes `_fwd(...)` on tape `Var`s (used by Baseline's train step) and the reference's
public `forward(...)` on plain tensors (inference / standalone use).
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn

from .. import ops, ops_loss
from ..ops import Var, param as P, ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SIGMOID, PAD_ZERO, PAD_REFLECT


_STEM_FUSE = os.environ.get("JP_STEM_FUSE", "1") != "0"      # bn1 -> relu -> maxpool of the ResNet stems in one pass each way
_BN_STATS_FUSE = os.environ.get("JP_BN_STATS_FUSE", "1") != "0"   # BatchNorm statistics out of the producing convolution's epilogue (round 6)


def _t(v):
    return v.t if isinstance(v, Var) else v


class BatchNorm2d(nn.BatchNorm2d):
    """Parameter/buffer container; num_batches_tracked is advanced on the host and folded into the
    buffer when a state dict is taken (no per-step scalar kernels)."""

    def __init__(self, c):
        super().__init__(c)
        self._pending = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        if self._pending:
            self.num_batches_tracked += self._pending
            self._pending = 0
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *a, **k):
        self._pending = 0
        super()._load_from_state_dict(*a, **k)


def bn_apply(bn: BatchNorm2d, x: Var, residual=None, relu=False, n_updates=1, groups=1) -> Var:
    if bn.training:
        bn._pending += n_updates * groups
        return ops.batchnorm_train(x, P(bn.weight), P(bn.bias), bn.running_mean, bn.running_var, residual, relu,
                                   bn.momentum, bn.eps, n_updates, groups)
    return ops.batchnorm_eval(x, bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, residual, relu, bn.eps)


def conv_apply(conv: nn.Conv2d, x, stride=None, pad=None, pad_mode=PAD_ZERO, act=ACT_NONE, srcs=None, bn_stats=False) -> Var:
    """`bn_stats`: the output goes straight into a train-mode BatchNorm (resnet.py:29-45): the convolution's epilogue leaves the
    statistics' partial sums for it where the kernel that runs the layer can (ops.conv2d)."""
    stride = conv.stride[0] if stride is None else stride
    pad = conv.padding[0] if pad is None else pad
    return ops.conv2d(x, P(conv.weight), P(conv.bias) if conv.bias is not None else None, stride, pad, pad_mode, act,
                      srcs, bn_stats=bn_stats)


# ------------------------------------------------------------------------------------------- layers.py
class Conv3x3(nn.Module):
    """ReflectionPad2d(1)/ZeroPad2d(1) + Conv2d(3) with bias (layers.py:156-167, layout_model.py:31-47)."""

    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        self.use_refl = use_refl
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), 3)

    def _fwd(self, x, act=ACT_NONE, srcs=None):
        return conv_apply(self.conv, x, 1, 1, PAD_REFLECT if self.use_refl else PAD_ZERO, act, srcs)

    def forward(self, x):
        return self._fwd(Var(x)).t


class Conv1x1(nn.Module):
    """layers.py:147-153."""

    def __init__(self, in_channels, out_channels, bias=False):
        super().__init__()
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), kernel_size=1, stride=1, bias=bias)

    def _fwd(self, x):
        return conv_apply(self.conv, x, 1, 0)

    def forward(self, x):
        return self._fwd(Var(x)).t


class CRPBlock(nn.Module):
    """Chained residual pooling (layers.py:184-199): x += conv1x1(maxpool5(top)) n_stages times."""

    def __init__(self, in_planes, out_planes, n_stages):
        super().__init__()
        for i in range(n_stages):
            setattr(self, "{}_{}".format(i + 1, "pointwise"), Conv1x1(in_planes if i == 0 else out_planes, out_planes, False))
        self.stride = 1
        self.n_stages = n_stages

    def _fwd(self, x):
        outer = ops.current_tape()
        train = outer is not None and x.rg
        # The output x + top_1 + ... + top_n is ONE pass over its n+1 terms (jp_sum_n, the reference's left-to-right
        # accumulation order) instead of n pairwise adds.
        # Training: the output gradient G reaches every `top_i` and the block input unchanged through the adds, so
        # the whole chain is ONE node of the outer tape.  Its backward replays a private tape of the pools and convs
        # with G folded into each max-pool backward kernel (g(top_{i-1}) = G + pool_bwd(conv_dgrad(g(top_i)))):
        # no per-stage gradient-accumulation pass.
        private, G = ops.Tape(), [None]
        tops, top = [], x
        with ops.recording(private if train else None):
            for i in range(self.n_stages):
                top = ops.maxpool(top, 5, 1, 2, bwd_addend=(lambda: G[0]) if train else None)
                top = getattr(self, "{}_{}".format(i + 1, "pointwise"))._fwd(top)
                tops.append(top)
        terms, acc, acc_am = [x.t] + [t.t for t in tops], None, None
        while len(terms) > 1:                      # up to 5 terms per launch (n_stages = 4 -> exactly one)
            chunk, terms = terms[:5], terms[5:]
            acc = torch.empty_like(x.t)
            acc_am = ops._out_slot(acc.device, len(terms) == 0)      # the last launch reports max |sum|: the merge convolution's operand scale
            ops.call("jp_sum_n", *(chunk + [None] * (5 - len(chunk))), acc, acc.numel(), acc_am)
            terms = [acc] + terms
        if not train:
            r = Var(acc if acc is not None else x.t, False)
            r.amax = acc_am if acc is not None else x.amax
            return r
        out = Var(acc, True)
        out.amax = acc_am
        last = top

        def bwd():
            if out.g is None:
                return
            G[0] = last.g = out.g
            last.gamax = out.gamax        # (the same tensor: what its producer reported about it still holds)
            private.backward()
            G[0] = out.g = None

        outer.record(bwd)
        return out

    def forward(self, x):
        return self._fwd(Var(x)).t


class SSIM(nn.Module):
    """layers.py:85-107: `SSIM()(x, y)` -> per-channel loss map clamp((1 - SSIM)/2, 0, 1) of the inputs' shape
    (ReflectionPad2d(1) + 3x3 average pooling).  Standalone forward = `jp_ssim_map`; the train step uses the fused
    SSIM+L1 kernels (`ops_loss.ssim_l1`, forward and backward) instead."""

    def forward(self, x, y):
        if x.shape != y.shape or x.dim() != 4:
            raise ValueError("SSIM expects two (B,C,H,W) tensors of equal shape")
        if x.requires_grad or y.requires_grad:
            raise NotImplementedError("standalone SSIM is forward-only; gradients come from the fused SSIM+L1 train-step kernels")
        B, C, H, W = x.shape
        x, y = x.contiguous().float(), y.contiguous().float()
        out = torch.empty_like(x)
        ops.call("jp_ssim_map", x, y, out, B * C, H, W)
        return out


class Backproject(nn.Module):
    """layers.py:41-61: `Backproject(B,H,W)(depth, inv_K)` -> homogeneous camera points (B,4,H*W).  The pixel grid the
    reference keeps as a buffer is generated in the kernel.  In the train step the geometry is fused into
    `jp_cgt_warp_*`; this is the module's public forward (`jp_backproject`)."""

    def __init__(self, batch_size, height, width):
        super().__init__()
        self.batch_size, self.height, self.width = batch_size, height, width

    def forward(self, depth, inv_K):
        B, H, W = self.batch_size, self.height, self.width
        if depth.numel() != B * H * W or tuple(inv_K.shape) != (B, 4, 4):
            raise ValueError("Backproject: depth must hold batch_size*height*width values and inv_K be (B,4,4)")
        if depth.requires_grad:
            raise NotImplementedError("standalone Backproject is forward-only; the train step differentiates jp_cgt_warp_*")
        out = torch.empty((B, 4, H * W), device=depth.device, dtype=torch.float32)
        ops.call("jp_backproject", depth.contiguous().float(), inv_K.contiguous().float(), out, B, H, W)
        return out


class Project(nn.Module):
    """layers.py:64-82: `Project(B,H,W)(points, K, T)` -> normalised sampling grid (B,H,W,2) for F.grid_sample
    (`jp_project`); fused into `jp_cgt_warp_*` in the train step."""

    def __init__(self, batch_size, height, width, eps=1e-7):
        super().__init__()
        self.batch_size, self.height, self.width, self.eps = batch_size, height, width, eps

    def forward(self, points, K, T):
        B, H, W = self.batch_size, self.height, self.width
        if tuple(points.shape) != (B, 4, H * W) or tuple(K.shape) != (B, 4, 4) or tuple(T.shape) != (B, 4, 4):
            raise ValueError("Project: points must be (B,4,H*W), K and T (B,4,4)")
        if points.requires_grad or T.requires_grad:
            raise NotImplementedError("standalone Project is forward-only; the train step differentiates jp_cgt_warp_*")
        out = torch.empty((B, H, W, 2), device=points.device, dtype=torch.float32)
        ops.call("jp_project", points.contiguous().float(), K.contiguous().float(), T.contiguous().float(), out, B, H, W,
                 float(self.eps))
        return out


# ------------------------------------------------------------------------------------------- resnet.py
class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def _fwd(self, x, n_updates=1, groups=1):
        tr = self.bn1.training and _BN_STATS_FUSE
        out = bn_apply(self.bn1, conv_apply(self.conv1, x, bn_stats=tr), relu=True, n_updates=n_updates, groups=groups)
        out = conv_apply(self.conv2, out, bn_stats=tr)
        res = x
        if self.downsample is not None:
            res = bn_apply(self.downsample[1], conv_apply(self.downsample[0], x), n_updates=n_updates, groups=groups)
        return bn_apply(self.bn2, out, residual=res, relu=True, n_updates=n_updates, groups=groups)


class ResNet(nn.Module):
    """ResNet-18 trunk with the reference's attribute names (resnet.py:86-136), incl. the unused fc."""

    def __init__(self, layers=(2, 2, 2, 2), in_ch=3, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(in_ch, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, planes, blocks, stride=1):
        ds = None
        if stride != 1 or self.inplanes != planes:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False), BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, ds)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(BasicBlock(planes, planes))
        return nn.Sequential(*layers)

    def features(self, img: Var, n_updates=1, ready_tag=None, groups=1, need_f0=True):
        """(x-0.45)/0.225 -> stem -> 4 stages; returns the 5-level pyramid (depth_encoder.py:35-44).
        `ready_tag`: report gradient completion in two steps (layer4, then the rest) to the data-parallel hook.
        `groups`: the batch stacks that many independent passes (BatchNorm statistics per group, ops.batchnorm_train).
        `need_f0=False` (the train step: neither DepthDecoder, PoseDecoder nor the layout Encoder reads level 0): the stem tail
        bn1 -> relu -> maxpool runs fused (ops.bn_relu_maxpool_train) and level 0 of the returned list is None."""
        if ready_tag:
            ops.grad_ready(ready_tag + ".lo")
        x = ops.affine(img, 1.0 / 0.225, -0.45 / 0.225)
        c0 = conv_apply(self.conv1, x)
        if not need_f0 and self.bn1.training and _STEM_FUSE and c0.t.shape[2] % 4 == 0 and c0.t.shape[3] % 4 == 0:
            self.bn1._pending += n_updates * groups
            feats = [None]
            x = ops.bn_relu_maxpool_train(c0, P(self.bn1.weight), P(self.bn1.bias), self.bn1.running_mean, self.bn1.running_var,
                                          self.bn1.momentum, self.bn1.eps, n_updates, groups)
        else:
            f0 = bn_apply(self.bn1, c0, relu=True, n_updates=n_updates, groups=groups)
            feats = [f0]
            x = ops.maxpool(f0, 3, 2, 1)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            if ready_tag and layer is self.layer4:
                ops.grad_ready(ready_tag + ".l4")
            for blk in layer:
                x = blk._fwd(x, n_updates, groups)
            feats.append(x)
        return feats


class DepthEncoder(nn.Module):
    """depth_encoder.py:7-44."""

    def __init__(self, num_layers=18, pretrained_path=None):
        super().__init__()
        if num_layers != 18:
            raise ValueError("only the ResNet-18 encoder of the north-star configs is built")
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        self.encoder = ResNet()
        if pretrained_path is not None:
            self.encoder.load_state_dict(torch.load(pretrained_path))

    def _fwd(self, img, n_updates=1, need_f0=False):
        return self.encoder.features(img, n_updates, ready_tag="DepthEncoder", need_f0=need_f0)

    def forward(self, input_image):
        return [f.t for f in self._fwd(Var(input_image), need_f0=True)]


class PoseEncoder(nn.Module):
    """pose_encoder.py:52-92 (6-channel stem)."""

    def __init__(self, num_layers=18, pretrained_path=None, num_input_images=2):
        super().__init__()
        if num_layers != 18:
            raise ValueError("only ResNet-18")
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        self.encoder = ResNet(in_ch=3 * num_input_images)
        if pretrained_path is not None:
            loaded = torch.load(pretrained_path)
            loaded["conv1.weight"] = torch.cat([loaded["conv1.weight"]] * num_input_images, 1) / num_input_images
            self.encoder.load_state_dict(loaded)

    def _fwd(self, img, n_updates=1, groups=1, need_f0=False):
        return self.encoder.features(img, n_updates, groups=groups, need_f0=need_f0)

    def forward(self, input_image):
        return [f.t for f in self._fwd(Var(input_image), need_f0=True)]


def find_imagenet_resnet18():
    """Where an offline copy of torchvision's ImageNet ResNet-18 weights may live (there is no network on the
    training hosts): $JPERCEIVER_RESNET18, then the torch hub cache torchvision / model_zoo would have filled."""
    import glob
    import os
    cand = [os.environ.get("JPERCEIVER_RESNET18", "")]
    hub = os.path.join(os.environ.get("TORCH_HOME", os.path.join(os.path.expanduser("~"), ".cache", "torch")), "hub", "checkpoints")
    cand += sorted(glob.glob(os.path.join(hub, "resnet18-*.pth")))
    for c in cand:
        if c and os.path.isfile(c):
            return c
    return None


class ResnetEncoder(nn.Module):
    """ResnetEncoder.py:71-110 (torchvision resnet18 there; identical architecture and keys).  `pretrained=True`
    loads the ImageNet ResNet-18 state dict (ResnetEncoder.py:60-66; conv1 replicated / averaged for multi-image
    input) from a local file — see find_imagenet_resnet18(); when none is available a RuntimeWarning says so
    loudly instead of silently training the layout encoder from scratch."""

    def __init__(self, num_layers=18, pretrained=False, num_input_images=1):
        super().__init__()
        if num_layers != 18:
            raise ValueError("{} is not a valid number of resnet layers (only ResNet-18 is built)".format(num_layers))
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        self.encoder = ResNet(in_ch=3 * num_input_images)
        self.pretrained_loaded = False
        if pretrained:
            path = find_imagenet_resnet18()
            if path is None:
                import warnings
                warnings.warn("ResnetEncoder(pretrained=True): no local ImageNet ResNet-18 weights found (set "
                              "$JPERCEIVER_RESNET18 or fill the torch hub cache); the layout encoder starts from "
                              "random initialisation, unlike the reference recipe", RuntimeWarning, stacklevel=2)
            else:
                loaded = torch.load(path, map_location="cpu")
                if num_input_images > 1:
                    loaded["conv1.weight"] = torch.cat([loaded["conv1.weight"]] * num_input_images, 1) / num_input_images
                self.encoder.load_state_dict(loaded)
                self.pretrained_loaded = True

    def _fwd(self, img, n_updates=1, need_f0=False):
        return self.encoder.features(img, n_updates, need_f0=need_f0)

    def forward(self, input_image):
        return [f.t for f in self._fwd(Var(input_image), need_f0=True)]


# ------------------------------------------------------------------------------------------- depth_decoder.py
class DepthDecoder(nn.Module):
    """depth_decoder.py:8-137.  upsample(x) and cat((reduce, x_up, disp)) are never materialised: the
    iconv/disp convolutions read their sources through the fused gather of jp_conv2d_*_src3."""

    def __init__(self, num_ch_enc):
        super().__init__()
        bott = 256
        self.do = nn.Dropout(p=0.5)
        self.reduce4 = Conv1x1(num_ch_enc[4], 512, bias=False)
        self.reduce3 = Conv1x1(num_ch_enc[3], bott, bias=False)
        self.reduce2 = Conv1x1(num_ch_enc[2], bott, bias=False)
        self.reduce1 = Conv1x1(num_ch_enc[1], bott, bias=False)
        self.iconv4 = Conv3x3(512, bott)
        self.iconv3 = Conv3x3(bott * 2 + 1, bott)
        self.iconv2 = Conv3x3(bott * 2 + 1, bott)
        self.iconv1 = Conv3x3(bott * 2 + 1, bott)
        for k in (4, 3, 2, 1):
            setattr(self, f"crp{k}", nn.Sequential(CRPBlock(bott, bott, 4)))
        for k in (4, 3, 2, 1):
            setattr(self, f"merge{k}", Conv3x3(bott, bott))
        for k in (4, 3, 2, 1):
            setattr(self, f"disp{k}", nn.Sequential(Conv3x3(bott, 1), nn.Sigmoid()))

    def _fwd(self, feats, drop_masks=None, frame_id=0):
        l0, l1, l2, l3, l4 = feats
        if self.training:
            if drop_masks is None:
                drop_masks = (ops.keep_mask(l4.t.shape, l4.t.device, 0.5), ops.keep_mask(l3.t.shape, l3.t.device, 0.5))
            l4 = ops.mul_mask(l4, drop_masks[0], 2.0)
            l3 = ops.mul_mask(l3, drop_masks[1], 2.0)
        out = {}
        x = self.reduce4._fwd(l4)
        x = self.iconv4._fwd(x, ACT_LEAKY)
        x = self.crp4[0]._fwd(x)
        x = self.merge4._fwd(x, ACT_LEAKY)                  # half-res; consumers read it upsampled
        disp = self.disp4[0]._fwd(None, ACT_SIGMOID, srcs=[(x, 1)])
        out[("disp", frame_id, 3)] = disp
        for k, lk, sc in ((3, l3, 2), (2, l2, 1), (1, l1, 0)):
            r = getattr(self, f"reduce{k}")._fwd(lk)
            x = getattr(self, f"iconv{k}")._fwd(None, ACT_LEAKY, srcs=[(r, 0), (x, 1), (disp, 0)])
            x = getattr(self, f"crp{k}")[0]._fwd(x)
            x = getattr(self, f"merge{k}")._fwd(x, ACT_LEAKY)
            disp = getattr(self, f"disp{k}")[0]._fwd(None, ACT_SIGMOID, srcs=[(x, 1)])
            out[("disp", frame_id, sc)] = disp
        return out

    def forward(self, input_features, frame_id=0):
        o = self._fwd([Var(f) for f in input_features], None, frame_id)
        self.outputs = {k: v.t for k, v in o.items()}
        return self.outputs


# ------------------------------------------------------------------------------------------- pose_decoder.py
class PoseDecoder(nn.Module):
    """pose_decoder.py:5-26."""

    def __init__(self, num_ch_enc, stride=1):
        super().__init__()
        self.reduce = nn.Conv2d(int(num_ch_enc[-1]), 256, 1)
        self.conv1 = nn.Conv2d(256, 256, 3, stride, 1)
        self.conv2 = nn.Conv2d(256, 256, 3, stride, 1)
        self.conv3 = nn.Conv2d(256, 6, 1)

    def _fwd(self, feats):
        o = conv_apply(self.reduce, feats[-1], act=ACT_RELU)
        o = conv_apply(self.conv1, o, act=ACT_RELU)
        o = conv_apply(self.conv2, o, act=ACT_RELU)
        o = conv_apply(self.conv3, o)
        return ops.spatial_mean(o, 0.01)                      # (B,6): [axisangle | translation]

    def forward(self, input_features):
        o = self._fwd([Var(f) for f in input_features]).t.view(-1, 1, 1, 6)
        return o[..., :3], o[..., 3:]


# ------------------------------------------------------------------------------------------- layout_model.py
class Encoder(nn.Module):
    """layout_model.py:56-113."""

    def __init__(self, num_layers=18, pretrained=True):
        super().__init__()
        self.resnet_encoder = ResnetEncoder(num_layers, pretrained)
        nc = self.resnet_encoder.num_ch_enc
        self.conv1 = Conv3x3(nc[-1], 128)
        self.conv2 = Conv3x3(128, 128)

    def _fwd(self, img, n_updates=1):
        x = self.resnet_encoder._fwd(img, n_updates)[-1]
        x = ops.maxpool(self.conv1._fwd(x), 2, 2, 0)
        return ops.maxpool(self.conv2._fwd(x), 2, 2, 0)

    def forward(self, x):
        return self._fwd(Var(x)).t


class Decoder(nn.Module):
    """layout_model.py:116-201: 5 x [conv3x3-BN-ReLU-up2x-conv3x3-BN] then reflect conv3x3 -> 2 logits.
    `decoder` is a ModuleList in the reference's OrderedDict order so the keys are decoder.{0..25}.*"""

    def __init__(self, num_ch_enc, num_class=2, type=""):
        super().__init__()
        self.num_output_channels = num_class
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        mods = []
        for i in range(4, -1, -1):
            cin = 128 if i == 4 else int(self.num_ch_dec[i + 1])
            co = int(self.num_ch_dec[i])
            mods += [nn.Conv2d(cin, co, 3, 1, 1), BatchNorm2d(co), nn.ReLU(True), nn.Conv2d(co, co, 3, 1, 1),
                     BatchNorm2d(co)]
        mods.append(Conv3x3(int(self.num_ch_dec[0]), num_class))
        self.decoder = nn.ModuleList(mods)

    def _fwd(self, x, n_updates=1):
        d = self.decoder
        for s in range(5):
            c0, b0, _, c1, b1 = d[5 * s], d[5 * s + 1], d[5 * s + 2], d[5 * s + 3], d[5 * s + 4]
            x = bn_apply(b0, conv_apply(c0, x), relu=True, n_updates=n_updates)
            x = conv_apply(c1, None, srcs=[(x, 1)])          # nearest 2x upsample fused into the conv gather
            x = bn_apply(b1, x, n_updates=n_updates)
        return d[25]._fwd(x)

    def forward(self, x, is_training=True):
        y = self._fwd(Var(x))
        if is_training:
            return y.t
        return ops.softmax2(y.t)


# ------------------------------------------------------------------------------------------- CVP / CCT
class TransformModule(nn.Module):
    """CycledViewProjection.py:27-67."""

    def __init__(self, dim=25):
        super().__init__()
        self.dim = dim
        self.mat_list = nn.ModuleList()
        self.fc_transform = nn.Sequential(nn.Linear(dim * dim, dim * dim), nn.ReLU(), nn.Linear(dim * dim, dim * dim),
                                          nn.ReLU())

    def _fwd(self, x: Var):
        B, C, H, W = x.t.shape
        v = ops_loss.view(x, (B, C, H * W))
        v = ops.linear_act(v, P(self.fc_transform[0].weight), P(self.fc_transform[0].bias), ACT_RELU)
        v = ops.linear_act(v, P(self.fc_transform[2].weight), P(self.fc_transform[2].bias), ACT_RELU)
        return ops_loss.view(v, (B, C, H, W))


class CycledViewProjection(nn.Module):
    """CycledViewProjection.py:11-24."""

    def __init__(self, in_dim):
        super().__init__()
        self.transform_module = TransformModule(dim=in_dim)
        self.retransform_module = TransformModule(dim=in_dim)

    def _fwd(self, x):
        t = self.transform_module._fwd(x)
        return t, self.retransform_module._fwd(t)

    def forward(self, x):
        t, r = self._fwd(Var(x))
        return t.t, r.t


class CrossViewTransformer(nn.Module):
    """CrossViewTransformer.py:27-92 (hard-argmax cross-view attention + depth cross-modal attention)."""

    def __init__(self, in_dim):
        super().__init__()
        self.query_conv = nn.Conv2d(in_dim, in_dim // 8, 1)
        self.key_conv = nn.Conv2d(in_dim, in_dim // 8, 1)
        self.value_conv = nn.Conv2d(in_dim, in_dim, 1)
        self.f_conv = nn.Conv2d(in_dim * 2, in_dim, 3, 1, 1, bias=True)
        self.res_conv = nn.Conv2d(in_dim, in_dim // 8, 1)          # never used by the reference either
        self.query_conv_depth = nn.Conv2d(in_dim, in_dim // 8, 1)
        self.key_conv_depth = nn.Conv2d(in_dim, in_dim // 8, 1)
        self.value_conv_depth = nn.Conv2d(in_dim, in_dim, 1)
        self.conv1 = Conv3x3(512, 128)
        self.conv2 = Conv3x3(128, 128)

    def _fwd(self, front_x, cross_x, front_x_hat, depth_feature):
        v = ops_loss.view
        df = ops.maxpool(self.conv1._fwd(depth_feature), 2, 2, 0)
        df = ops.maxpool(self.conv2._fwd(df), 2, 2, 0)
        B, C, w, h = front_x.t.shape
        n = w * h
        q = v(conv_apply(self.query_conv, cross_x), (B, C // 8, n))
        k = v(conv_apply(self.key_conv, front_x), (B, C // 8, n))
        energy = ops_loss.bmm_tn(k, q)
        front_star, arg = ops_loss.colmax(energy)
        val = v(conv_apply(self.value_conv, front_x_hat), (B, C, n))
        T = v(ops_loss.gather_cols(val, arg), (B, C, w, h))
        S = v(front_star, (B, 1, w, h))
        res = conv_apply(self.f_conv, None, srcs=[(front_x, 0), (T, 0)])
        out = ops.add(front_x, ops_loss.mul_bcast_c(res, S))
        qd = v(conv_apply(self.query_conv_depth, cross_x), (B, C // 8, n))
        kd = v(conv_apply(self.key_conv_depth, front_x), (B, C // 8, n))
        vd = conv_apply(self.value_conv_depth, df)
        attn, arg_d = ops_loss.colmax(ops_loss.bmm_tn(kd, qd))
        attn = v(attn, (B, 1, w, h))
        out = ops.add(out, ops_loss.bcast_matmul(attn, vd))
        return out, S, attn, arg, arg_d

    def forward(self, front_x, cross_x, front_x_hat, depth_feature):
        o, S, a, _, _ = self._fwd(Var(front_x), Var(cross_x), Var(front_x_hat), Var(depth_feature))
        return o.t, S.t, a.t
