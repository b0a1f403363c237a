"""Train API with the reference's surface (mono/apis/trainer.py:20-56,76-143): `change_input_variable`,
`batch_processor(model, data, train_mode)`, `build_optimizer(model, optimizer_cfg)`, plus a minimal
`Runner` that reproduces the mmcv hot loop order (forward -> DistOptimizerHook.after_train_iter)."""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.distributed as dist

from ..model.net import LossDict, scale_label_matrices
from ..runtime import FlatAdam


def change_input_variable(data, device="cuda", opt=None):
    """trainer.py:20-27: every entry -> float32 on the GPU (one pinned async copy each).  When `opt` is
    given, the tiny CGT scale-label homographies are derived here from the CPU calibration so the step
    itself never reads calibration back from the device."""
    if opt is not None and ("scale_H", 0, 0) not in data and ("odometry_K", 0, 0) in data \
            and not torch.as_tensor(data[("odometry_K", 0, 0)]).is_cuda:
        FH, FW = data[("color", 0, -1)].shape[2:4]
        Hm, quad = scale_label_matrices(opt, torch.as_tensor(data[("odometry_K", 0, 0)]).float(),
                                        torch.as_tensor(data[("Tr_cam2_velo", 0, 0)]).float(), FH, FW)
        data[("scale_H", 0, 0)], data[("scale_quad", 0, 0)] = Hm, quad
    for k, v in data.items():
        if k[0] == "bev_path" or "kp" in k:
            continue
        t = torch.as_tensor(v)
        if k == ("scale_quad", 0, 0):
            data[k] = t.to(device=device, dtype=torch.int32, non_blocking=True)
        elif t.dtype in (torch.int64, torch.int32) and k[0] in ("min_index",):
            data[k] = t.to(device, non_blocking=True)
        else:
            data[k] = t.float().to(device, non_blocking=True)
    return data


def batch_processor(model, data, train_mode):
    """trainer.py:30-56.  loss = sum of *every* loss_dict entry (layout terms are therefore counted twice,
    SURVEY.md N3); log_vars are fetched with ONE device->host copy instead of one .item() per term."""
    opt = getattr(getattr(model, "module", model), "opt", None)
    data = change_input_variable(data, opt=opt)
    model_out, losses = model(data)
    if isinstance(losses, LossDict):
        loss = losses.total()
        vals = losses._lv.vals.detach().cpu().tolist()          # the single sync of the step
        log_vars = OrderedDict((str(k), float(v)) for k, v in zip(losses._lv.names, vals))
        log_vars["loss"] = float(sum(vals))
    else:
        lv = OrderedDict((k, v.mean()) for k, v in losses.items())
        loss = sum(lv.values())
        log_vars = OrderedDict((str(k), v.item()) for k, v in lv.items())
        log_vars["loss"] = loss.item()
    return dict(loss=loss, log_vars=log_vars, num_samples=len(data[("color", 0, 0)]))


def build_optimizer(model, optimizer_cfg):
    """trainer.py:76-143 for the configs' `dict(type='Adam', lr=1e-4, weight_decay=0)`; returns the flat-arena
    Adam and attaches the arena to the model for allreduce_grads / DistOptimizerHook."""
    m = getattr(model, "module", model)
    cfg = dict(optimizer_cfg)
    if cfg.pop("type", "Adam") != "Adam":
        raise NotImplementedError("only Adam (all north-star configs)")
    if cfg.pop("paramwise_options", None) is not None:
        raise NotImplementedError("paramwise_options are not used by the north-star configs")
    skip = ()
    ty = getattr(m, "opt", {}).get("type", "Argo_both") if hasattr(m, "opt") else "Argo_both"
    if ty not in ("dynamic", "Argo_dynamic", "Argo_both"):
        skip = ("CycledViewProjectionB.", "CrossViewTransformerB.", "LayoutDecoderB.", "LayoutTransformDecoderB.")
    elif ty in ("dynamic", "Argo_dynamic"):
        skip = ("CycledViewProjection.", "CrossViewTransformer.", "LayoutDecoder.", "LayoutTransformDecoder.")
    if hasattr(m, "opt") and not m.opt.get("layout_branch", True):     # bench.py's secondary sub-path figure only
        skip = ("LayoutEncoder.", "CycledViewProjection", "CrossViewTransformer", "LayoutDecoder", "LayoutTransformDecoder")
    opt = FlatAdam(m, lr=cfg.get("lr", 1e-4), betas=tuple(cfg.get("betas", (0.9, 0.999))), eps=cfg.get("eps", 1e-8),
                   weight_decay=cfg.get("weight_decay", 0), skip_prefixes=skip)
    m._jp_arena = opt.arena
    return opt


class DataParallelShell(torch.nn.Module):
    """Stand-in for MMDistributedDataParallel (trainer.py:167): `.module`, forwards calls, broadcasts rank 0's
    parameters and buffers once at wrap time.  Gradient averaging is NOT done here (DistOptimizerHook does it
    once, on the flat arena)."""

    def __init__(self, module):
        super().__init__()
        self.module = module
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, 0)

    def forward(self, *a, **k):
        return self.module(*a, **k)


class Runner(object):
    """The slice of mmcv.Runner the hot loop needs: model, batch_processor, optimizer, outputs, one iteration."""

    def __init__(self, model, batch_processor, optimizer, optimizer_hook):
        self.model, self.batch_processor, self.optimizer, self.hook = model, batch_processor, optimizer, optimizer_hook
        self.outputs = None
        self.iter = 0

    def train_iter(self, data_batch):
        self.model.train()
        self.outputs = self.batch_processor(self.model, data_batch, train_mode=True)
        self.hook.after_train_iter(self)
        self.iter += 1
        return self.outputs
