"""Checkpoint I/O in the layout the reference's runner writes and resumes from (mmcv 0.4.4 `save_checkpoint` /
`Runner.resume`, used by mono/apis/trainer.py:195-198 and the checkpoint hook of every config):

    {'meta': {..., 'epoch': e, 'iter': i}, 'state_dict': OrderedDict(name -> CPU tensor), 'optimizer': Adam state_dict}

`state_dict` keys are the reference's module paths (a leading 'module.' of a DDP wrapper is stripped on load, as mmcv
does); the optimizer entry is torch.optim.Adam's own format, which `FlatAdam` reads and writes — so checkpoints are
interchangeable with the reference in both directions (released weights load with strict=True: 766 tensors)."""
from __future__ import annotations

import os
import time
from collections import OrderedDict

import torch


def weights_to_cpu(state_dict):
    return OrderedDict((k, v.detach().cpu()) for k, v in state_dict.items())


def _unwrap(model):
    return getattr(model, "module", model)


def save_checkpoint(model, filename, optimizer=None, meta=None):
    meta = dict(meta or {})
    meta.setdefault("time", time.asctime())
    meta.setdefault("writer", "jperceiver_amd")
    ckpt = {"meta": meta, "state_dict": weights_to_cpu(_unwrap(model).state_dict())}
    if optimizer is not None:
        sd = optimizer.state_dict()
        sd["state"] = {k: {kk: (vv.detach().cpu() if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
                       for k, v in sd["state"].items()}
        ckpt["optimizer"] = sd
    d = os.path.dirname(filename)
    if d:
        os.makedirs(d, exist_ok=True)
    tmp = f"{filename}.tmp.{os.getpid()}"           # a reader (resume) never sees a half-written file
    torch.save(ckpt, tmp)
    os.replace(tmp, filename)
    return ckpt


def load_checkpoint(model, filename, map_location="cpu", strict=False):
    """mmcv.runner.load_checkpoint: accepts a bare state dict or {'state_dict': ...}; strips 'module.' prefixes."""
    ckpt = torch.load(filename, map_location=map_location, weights_only=False)
    if isinstance(ckpt, OrderedDict) or "state_dict" not in ckpt:
        sd = ckpt
    else:
        sd = ckpt["state_dict"]
    if list(sd.keys())[0].startswith("module."):
        sd = OrderedDict((k[7:], v) for k, v in sd.items())
    missing, unexpected = _unwrap(model).load_state_dict(sd, strict=strict)
    if (missing or unexpected) and not strict:
        import warnings
        warnings.warn(f"load_checkpoint: missing keys {list(missing)[:5]}..., unexpected keys {list(unexpected)[:5]}...")
    return ckpt
