"""Device-side `MonoDataset.preprocess` (mono/datasets/mono_dataset.py:126-171,417-431): the raw uint8 frames of a batch
are uploaded once (pinned, asynchronous — datasets/loader.py) and resized / converted / augmented by HIP kernels
(csrc/preprocess.hip) instead of PIL + torchvision on 24 host workers per GPU.

  resize          transforms.Resize((H, W), Image.ANTIALIAS): Pillow's 8-bit Lanczos resampler, bit-exact (the
                  fixed-point coefficient tables are built here exactly like libImaging/Resample.c builds them)
  to_tensor       HWC uint8 -> CHW float32 / 255
  ColorJitter     brightness / contrast / saturation (0.8, 1.2), hue (-0.1, 0.1), random order.  Decided PER ITEM
                  (`do_color_aug = random.random() > 0.5`, mono_dataset.py:202); the parameters are drawn PER FRAME:
                  the reference hands `preprocess` a `transforms.ColorJitter` object (its `get_params` line is commented
                  out, mono_dataset.py:338-339) and such an object re-draws order + factors on every call, i.e. for
                  every frame of the item (mono_dataset.py:153-156).  `jitter="per_item"` gives what the docstring there
                  intends (one draw shared by an item's frames).  Applied in torchvision's float-tensor arithmetic (the
                  reference applies it to the 8-bit PIL image: values differ by <= ~1/255 rounding)
  process_topview luma, binarise, NEAREST resize to H/4
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .._lib import call

PRECISION_BITS = 32 - 8 - 2


def _lanczos(x):
    x = np.asarray(x, dtype=np.float64)
    out = np.zeros_like(x)
    m = (x >= -3.0) & (x < 3.0)

    def sinc(v):
        r = np.ones_like(v)
        nz = v != 0.0
        pv = v[nz] * math.pi
        r[nz] = np.sin(pv) / pv
        return r
    out[m] = sinc(x[m]) * sinc(x[m] / 3.0)
    return out


def pil_resample_tables(in_size: int, out_size: int):
    """precompute_coeffs + normalize_coeffs_8bpc of Pillow's Resample.c for the Lanczos filter (support 3):
    -> bounds (out, 2) int32 {first tap, tap count}, kk (out, ksize) int32, ksize."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size      # box = (0, 0, in, in) held as C floats
    filterscale = max(scale, 1.0)
    support = 3.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = _lanczos((np.arange(xmax) + xmin - center + 0.5) * ss)
        ww = 0.0
        for v in w:                       # sequential sum, like the C loop
            ww += v
        if ww != 0.0:
            w = w / ww
        fx = np.where(w < 0, -0.5 + w * (1 << PRECISION_BITS), 0.5 + w * (1 << PRECISION_BITS))
        kk[xx, :xmax] = np.trunc(fx).astype(np.int32)          # (int) cast truncates toward zero
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


class ColorJitterParams:
    """torchvision.transforms.ColorJitter.get_params: a random order of the four ops and one factor each."""

    def __init__(self, brightness=(0.8, 1.2), contrast=(0.8, 1.2), saturation=(0.8, 1.2), hue=(-0.1, 0.1), generator=None):
        self.order = torch.randperm(4, generator=generator).tolist()
        u = lambda lo, hi: float(torch.empty(1).uniform_(lo, hi, generator=generator))
        self.factors = [u(*brightness), u(*contrast), u(*saturation), u(*hue)]     # indexed by op id 0..3


class DevicePreprocessor:
    """Preprocess one uploaded batch.  Tables are cached per (in, out) size; everything runs on the caller's stream."""

    def __init__(self, height, width, device):
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.h, self.w, self.dev = height, width, dev
        self._tab = {}

    def _tables(self, n_in, n_out):
        key = (n_in, n_out)
        if key not in self._tab:
            b, k, ks = pil_resample_tables(n_in, n_out)
            self._tab[key] = (torch.from_numpy(b).to(self.dev), torch.from_numpy(k).to(self.dev), ks)
        return self._tab[key]

    def resize_u8(self, frames: torch.Tensor, OH: int, OW: int, want_u8=False):
        """frames (N, H, W, C) uint8 on the device -> float (N, C, OH, OW) in [0, 1] (and the uint8 HWC image)."""
        N, H, W, C = frames.shape
        frames = frames.contiguous()
        if (H, W) == (OH, OW):
            out = torch.empty((N, C, OH, OW), device=self.dev, dtype=torch.float32)
            call("jp_u8_to_tensor", frames, out, N, H, W, C)
            return (out, frames) if want_u8 else out
        bh, kh, ksh = self._tables(W, OW)
        bv, kv, ksv = self._tables(H, OH)
        tmp = torch.empty((N, H, OW, C), device=self.dev, dtype=torch.uint8)
        if W != OW:
            call("jp_resample_h_u8", frames, tmp, bh, kh, N * H, W, OW, C, ksh)
        else:
            tmp = frames
        out = torch.empty((N, C, OH, OW), device=self.dev, dtype=torch.float32)
        o8 = torch.empty((N, OH, OW, C), device=self.dev, dtype=torch.uint8) if want_u8 else None
        if H != OH:
            call("jp_resample_v_u8", tmp, o8, out, bv, kv, N, H, OH, OW, C, ksv)
        else:
            call("jp_u8_to_tensor", tmp, out, N, OH, OW, C)
            o8 = tmp
        return (out, o8) if want_u8 else out

    def color_jitter_(self, x: torch.Tensor, params: ColorJitterParams):
        """in place on (N, 3, H, W) floats; the SAME parameters for every image of the call."""
        N, C, H, W = x.shape
        assert C == 3 and x.is_contiguous()
        sums = torch.empty(N, device=x.device, dtype=torch.float64)
        for op in params.order:
            call("jp_color_jitter_op", x, sums, N, H * W, int(op), float(params.factors[op]))
        return x

    def topview(self, label_u8: torch.Tensor, size: int, both=False):
        """process_topview / process_topview_both: (N, h, w[, C]) uint8 -> (N, 1, size, size) float {0, 1}."""
        if label_u8.dim() == 3:
            label_u8 = label_u8.unsqueeze(-1)
        N, h, w, C = label_u8.shape
        out = torch.empty((N, 1, size, size), device=self.dev, dtype=torch.float32)
        call("jp_topview_u8", label_u8.contiguous(), out, N, h, w, C, size, int(both))
        return out

    def __call__(self, raw: dict, frame_ids, full_hw, do_color_aug=None, generator=None, jitter="per_frame"):
        """raw: {("color", f, -1): (N, h, w, 3) uint8, ("bothS"|"bothD"|"both_dynamic", 0, 0): uint8 labels, calibration
        tensors ...} already on the device -> the input dict Baseline.forward expects.
        do_color_aug: None -> one coin per ITEM (mono_dataset.py:202); a bool -> every item; a sequence -> per item.
        jitter: "per_frame" (the reference's actual behaviour: every frame of an augmented item gets its own draw) or
        "per_item" (one draw shared by the item's frames).  Draw order (generator): the N coins, then per item, per frame."""
        if jitter not in ("per_frame", "per_item"):
            raise ValueError(f"jitter={jitter!r}")
        out = {}
        FH, FW = full_hw
        N = raw[("color", frame_ids[0], -1)].shape[0]
        if do_color_aug is None:
            coins = (torch.rand(N, generator=generator) > 0.5).tolist()
        elif isinstance(do_color_aug, (bool, int)):
            coins = [bool(do_color_aug)] * N
        else:
            coins = [bool(c) for c in do_color_aug]
            if len(coins) != N:
                raise ValueError(f"do_color_aug has {len(coins)} entries for {N} items")
        params = {}                      # (item, frame) -> ColorJitterParams
        for i, c in enumerate(coins):
            if not c:
                continue
            shared = ColorJitterParams(generator=generator) if jitter == "per_item" else None
            for f in frame_ids:
                params[(i, f)] = shared if shared is not None else ColorJitterParams(generator=generator)
        self.last_jitter = params        # for loggers / tests
        for f in frame_ids:
            full, full8 = self.resize_u8(raw[("color", f, -1)], FH, FW, want_u8=True)     # resize_full, then resize from it
            if f == 0:
                out[("color", 0, -1)] = full
            img = self.resize_u8(full8, self.h, self.w)
            out[("color", f, 0)] = img
            aug = img.clone() if any(coins) else img
            for i, c in enumerate(coins):
                if c:
                    self.color_jitter_(aug[i:i + 1], params[(i, f)])
            out[("color_aug", f, 0)] = aug
        for k, v in raw.items():
            if k[0] in ("bothS", "bothD"):
                out[k] = self.topview(v, self.h // 4)
            elif k[0] == "both_dynamic":
                out[k] = self.topview(v, self.h // 4, both=True)
            elif k[0] != "color":
                out[k] = v.float() if torch.is_tensor(v) and v.dtype != torch.float32 else v
        return out
