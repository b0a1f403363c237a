"""Train API with the reference's surface (mono/apis/trainer.py:20-56,59-73,76-143,146-199): `change_input_variable`,
`batch_processor(model, data, train_mode)`, `build_optimizer(model, optimizer_cfg)`, `train_mono(model, dataset_train,
dataset_val, cfg, args, distributed, validate, logger)` -- what `train.py:89-96` calls -- plus a minimal `Runner` that
reproduces the mmcv hot loop order (forward -> DistOptimizerHook.after_train_iter)."""
from __future__ import annotations

import os
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


class LazyLogVars(OrderedDict):
    """`log_vars` whose floats travel with ONE asynchronous device->host copy issued right after the forward pass and
    are resolved on first read (mono/apis/trainer.py:44-53 calls `.item()` per term: a host sync in the middle of every
    step, during which the device idles until the backward pass is enqueued).  Reads like the reference's OrderedDict of
    Python floats; the step's kernels are all in flight by the time a logger looks at it."""
    _ring = {}

    def __init__(self, names, dev_vals):
        super().__init__()
        self._names = [str(n) for n in names]
        key = (dev_vals.device, dev_vals.numel(), dev_vals.dtype)
        ring = LazyLogVars._ring.setdefault(key, [None, None, 0])
        slot = ring[2] & 1
        ring[2] += 1
        prev = ring[slot]
        if prev is not None and prev[1]() is not None:
            prev[1]()._resolve()                              # the buffer's previous owner reads it before it is reused
        host = prev[0] if prev is not None else torch.empty(dev_vals.shape, dtype=dev_vals.dtype, pin_memory=True)
        host.copy_(dev_vals.detach(), non_blocking=True)
        self._host, self._ev, self._pending = host, torch.cuda.Event(), True
        self._ev.record(torch.cuda.current_stream(dev_vals.device))
        import weakref
        ring[slot] = (host, weakref.ref(self))

    def _resolve(self):
        if self._pending:
            self._pending = False
            self._ev.synchronize()
            vals = self._host.tolist()
            for k, v in zip(self._names, vals):
                OrderedDict.__setitem__(self, k, float(v))
            OrderedDict.__setitem__(self, "loss", float(sum(vals)))
        return self

    def __getitem__(self, k): return OrderedDict.__getitem__(self._resolve(), k)
    def __iter__(self): return OrderedDict.__iter__(self._resolve())
    def __len__(self): return OrderedDict.__len__(self._resolve())
    def __contains__(self, k): return OrderedDict.__contains__(self._resolve(), k)
    def __repr__(self): return OrderedDict.__repr__(self._resolve())
    def __eq__(self, o): return OrderedDict.__eq__(self._resolve(), o)
    def __reduce__(self): return (OrderedDict, (list(self.items()),))
    def get(self, k, d=None): return OrderedDict.get(self._resolve(), k, d)
    def items(self): return OrderedDict.items(self._resolve())
    def keys(self): return OrderedDict.keys(self._resolve())
    def values(self): return OrderedDict.values(self._resolve())
    def copy(self): return OrderedDict(self.items())


def batch_processor(model, data, train_mode):
    """trainer.py:30-56.  loss = sum of *every* loss_dict entry (layout terms are therefore counted twice,
    SURVEY.md N3); log_vars are fetched with ONE asynchronous device->host copy instead of one .item() per term."""
    opt = getattr(getattr(model, "module", model), "opt", None)
    data = change_input_variable(data, opt=opt)
    model_out, losses = model(data)
    if isinstance(losses, LossDict):
        loss = losses.total()
        if losses._lv.vals.is_cuda and os.environ.get("JP_LAZY_LOG", "1") != "0":
            log_vars = LazyLogVars(losses._lv.names, losses._lv.vals)     # no host sync inside the step
        else:
            vals = losses._lv.vals.detach().cpu().tolist()
            log_vars = OrderedDict((str(k), float(v)) for k, v in zip(losses._lv.names, vals))
            log_vars["loss"] = float(sum(vals))
    else:
        lv = OrderedDict((k, v.mean()) for k, v in losses.items())
        loss = sum(lv.values())
        log_vars = OrderedDict((str(k), v.item()) for k, v in lv.items())
        log_vars["loss"] = loss.item()
    return dict(loss=loss, log_vars=log_vars, num_samples=len(data[("color", 0, 0)]))


class CapturedStep:
    """One whole training iteration -- batch_processor's forward, the loss sum, DistOptimizerHook's zero_grad / backward /
    clip + Adam and the weight re-pack -- captured ONCE into a hipGraph and replayed per batch.

    Why: the step's launch sequence is static per input signature (own tape, no host sync inside the step, `ops.py`), and at
    one image per GPU (BASELINE.json configs[0] and [4]) its ~1 500 kernels are shorter than the ~16 us the host needs to
    issue each of them through Python + ctypes: the eager B = 1 step is host-bound at 25-31 ms.  Replay costs one launch.

    How: inputs are copied into static device buffers; everything that changes from step to step and used to be a kernel
    ARGUMENT lives in device memory the host refreshes before a replay with one small H2D copy each --
      * Adam's lr and bias corrections (`jp_adam_clip_step_dev`, runtime.FlatArena.dev_state),
      * the seed base of the step's Dropout / automask-noise draws (`jp_rng_*_dev`, ops.rng_capture): a replay draws exactly
        what the eager step would have drawn --
    and the host-side counters the eager step advances (Adam step, RNG counter, BatchNorm `num_batches_tracked`, the pack
    registry's weight epoch) are advanced per replay.  The first `WARMUP` iterations run eagerly on the same static buffers
    (they are real training steps: packs are recorded, scratch and optimizer state are allocated); the graph is captured at
    the next call and re-captured when the batch signature (keys, shapes, dtypes) changes.

    More than one rank (round 5; BASELINE.json's configs[4] is an 8-GPU, one-image-per-GPU config): the iteration is TWO graphs
    around the gradient exchange -- graph A = forward, loss sum, zero_grad, backward; then the bucketed SUM all-reduce of the
    arena's live prefix issued eagerly (work handles are host objects, not graph nodes); graph B = global norm, clip + Adam
    (1 / world folded in), weight re-pack.  What this gives up against the eager multi-rank step is the overlap of the exchange
    with the backward pass (208.7 MB: 0.35-2.4 ms on xGMI); what it removes is the host from ~1 500 launches per step."""
    WARMUP = 2

    def __init__(self, runner):
        from ..runtime import FlatAdam
        self.r = runner
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        if not isinstance(runner.optimizer, FlatAdam):
            raise RuntimeError("CapturedStep needs the flat-arena optimizer (apis.build_optimizer)")
        if runner.batch_processor is not batch_processor:
            raise RuntimeError("CapturedStep replays the stock batch_processor")
        self.m = getattr(runner.model, "module", runner.model)
        self.sig, self.static, self.graph, self.graph_b, self.eager_done = None, None, None, None, 0
        self.replays, self.recaptures = 0, 0
        self._table, self._h2d_done = None, None

    def _table_current(self, dev):
        """the pack registry still serves the job table (and, through its `live` list, the scratch tensors) the graph was
        captured with"""
        from .. import ops
        t = ops.PackRegistry.of(dev).table
        if t is None or self._table is None or t is not self._table:
            return False
        return all(e.param is not None and e.ptr == e.param.data_ptr() for e in t[3])

    # ---- inputs
    def _prepare(self, data):
        """device fp32 batch (change_input_variable) with the scale-label homographies present: they are host algebra on the
        calibration, which must not happen inside the captured region"""
        data = change_input_variable(dict(data), opt=self.m.opt)
        if ("scale_H", 0, 0) not in data and ("odometry_K", 0, 0) in data:
            FH, FW = data[("color", 0, -1)].shape[2:4]
            Hm, quad = scale_label_matrices(self.m.opt, data[("odometry_K", 0, 0)].float().cpu(),
                                            data[("Tr_cam2_velo", 0, 0)].float().cpu(), FH, FW)
            dev = data[("color", 0, 0)].device
            data[("scale_H", 0, 0)], data[("scale_quad", 0, 0)] = Hm.to(dev), quad.to(device=dev, dtype=torch.int32)
        return data

    @staticmethod
    def _signature(data):
        return tuple(sorted((repr(k), tuple(v.shape), str(v.dtype)) for k, v in data.items() if torch.is_tensor(v)))

    def _load(self, data):
        sig = self._signature(data)
        if sig != self.sig:            # new input signature: fresh static buffers, eager warm-up, new capture
            if self.graph is not None:
                # the old graph (and the tensors of its private pool) must be gone BEFORE the next capture starts: destroying a
                # hipGraph while a stream is capturing is an error ("operation not permitted when stream is capturing")
                import gc
                torch.cuda.synchronize()
                self.model_out = self.losses = self.loss = None
                self.graph = self.graph_b = None
                gc.collect()
            self.sig, self.graph, self.graph_b, self.eager_done = sig, None, None, 0
            self.static = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in data.items()}
            return
        for k, v in data.items():
            if torch.is_tensor(v) and v.data_ptr() != self.static[k].data_ptr():
                self.static[k].copy_(v, non_blocking=True)

    # ---- the iteration body (what train_iter does between its hooks)
    def _body(self):
        r = self.r
        model_out, losses = r.model(dict(self.static))
        loss = losses.total()
        r.outputs = dict(loss=loss, log_vars=None, num_samples=len(self.static[("color", 0, 0)]))
        r.hook.after_train_iter(r)
        return model_out, losses, loss

    # ---- more than one rank: the iteration in two halves around the exchange
    def _body_a(self):
        """forward, loss sum, zero_grad, backward -- everything in front of the exchange (no grad_ready hook: no collective here)"""
        r = self.r
        model_out, losses = r.model(dict(self.static))
        loss = losses.total()
        r.outputs = dict(loss=loss, log_vars=None, num_samples=len(self.static[("color", 0, 0)]))
        r.optimizer.zero_grad()
        loss.backward()
        return model_out, losses, loss

    def _exchange(self):
        from ..core.dist_utils import _Exchange
        ex = _Exchange(self.r.optimizer.arena, getattr(self.r.hook, "bucket_size_mb", -1))
        ex.events = getattr(self.r.hook, "exposed_events", None)
        ex.finish(with_norm=False)

    def _body_b(self):
        """global norm over the reduced prefix, clip + Adam with the 1 / world averaging folded in, weight re-pack epoch"""
        opt = self.r.optimizer
        gc = getattr(self.r.hook, "grad_clip", None)
        max_norm = gc.get("max_norm") if gc else None
        if max_norm:
            opt.arena.add_norm_partial(0, opt.arena.live_numel)
        opt.grad_scale, opt.max_norm = 1.0 / self.world, max_norm
        opt.step()

    def _finish(self, losses, loss, model_out):
        log_vars = LazyLogVars(losses._lv.names, losses._lv.vals)      # async D2H + event, OUTSIDE the captured region
        return dict(loss=loss, log_vars=log_vars, num_samples=len(self.static[("color", 0, 0)]), model_out=model_out)

    def _bn_modules(self):
        return [mod for mod in self.m.modules() if hasattr(mod, "_pending")]

    def _capture(self):
        from .. import ops
        opt, arena = self.r.optimizer, self.r.optimizer.arena
        dev = arena.params.device
        self.state_host = torch.empty(3, dtype=torch.float32).pin_memory()
        self.base_host = torch.empty(1, dtype=torch.int64).pin_memory()
        self.state_dev = torch.zeros(3, device=dev, dtype=torch.float32)
        self.base_dev = torch.zeros(1, device=dev, dtype=torch.int64)
        bns = self._bn_modules()
        saved = (arena.step_count, ops._EPOCH[0], [b._pending for b in bns])
        n_partials0 = getattr(arena, "_n_partials", 0)     # host-side counter of the norm partials: the capture must not move it
        import gc
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        arena.dev_state = self.state_dev
        gc.collect()                # (may drop dead models' pack entries -> the registry's job table is rebuilt next)
        ops.PackRegistry.of(dev).ensure_table()
        torch.cuda.synchronize(dev)
        gc_was = gc.isenabled()
        gc.disable()                # no collector run (it may destroy HIP objects) while the stream is capturing
        ops.amax_pool_reset()       # operand-scale slots: zeroed by fills that are part of THIS capture
        try:
            if self.world == 1:
                with ops.rng_capture(self.base_dev) as cap, torch.cuda.graph(self.graph):
                    self.model_out, self.losses, self.loss = self._body()
            else:
                with ops.rng_capture(self.base_dev) as cap, torch.cuda.graph(self.graph):
                    self.model_out, self.losses, self.loss = self._body_a()
                self.graph_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_b, pool=self.graph.pool()):
                    self._body_b()
        finally:
            ops.amax_pool_reset()
            arena.dev_state = None
            if gc_was:
                gc.enable()
        # the graph holds RAW pointers into the pack registry's job table and every pack entry's scratch: keep them alive here (the
        # registry drops its table whenever a conv of a new signature is recorded -- an eval / validation forward at another batch
        # or resolution, a second model on the device) and re-capture when the registry has moved on (step()).
        self._table = ops.PackRegistry.of(dev).table
        self.rng_calls = cap.calls
        self.bn_delta = [(b, b._pending - p0) for b, p0 in zip(bns, saved[2])]
        # the capture executed nothing: take the host-side counters back
        if hasattr(arena, "_n_partials"):
            # (graph B bakes the partials' layout of THIS capture in; an eager step must find the counter where it left it)
            arena._n_partials = n_partials0
        arena.step_count, ops._EPOCH[0] = saved[0], saved[1]
        for b, p0 in zip(bns, saved[2]):
            b._pending = p0

    def step(self, data_batch):
        from .. import ops
        from ..runtime import FlatArena
        r = self.r
        self._load(self._prepare(data_batch))
        if self.eager_done < self.WARMUP:                                # real steps, issued launch by launch
            self.eager_done += 1
            if self.world == 1:
                model_out, losses, loss = self._body()
            else:                                                        # the same three pieces the replays will run
                model_out, losses, loss = self._body_a()
                self._exchange()
                self._body_b()
            return self._finish(losses, loss, model_out)
        opt, arena = r.optimizer, r.optimizer.arena
        if self.graph is not None and not self._table_current(arena.params.device):
            # packs were recorded / dropped since the capture (a forward with a new conv signature in between): the captured
            # jp_pack_replay would refresh a stale job table.  The old table and scratch stay alive (self._table) until the
            # old graph is gone; then capture again against the registry's current table.
            import gc
            torch.cuda.synchronize()
            self.model_out = self.losses = self.loss = None
            self.graph = self.graph_b = None
            gc.collect()
            self.recaptures += 1
        if self.graph is None:
            self._capture()
        g = opt.param_groups[0]
        arena.step_count += 1
        if self._h2d_done is not None:
            self._h2d_done.synchronize()      # the previous replay's H2D copies have read the pinned words (no-op after a resolve)
        self.state_host.copy_(torch.from_numpy(FlatArena.step_state(g["lr"], g["betas"], arena.step_count)))
        self.base_host[0] = ops.rng_step_base()
        self.state_dev.copy_(self.state_host, non_blocking=True)
        self.base_dev.copy_(self.base_host, non_blocking=True)
        if self._h2d_done is None:
            self._h2d_done = torch.cuda.Event()
        self._h2d_done.record()
        # ORDER IS PART OF THE CONTRACT: graph B was captured into graph A's memory pool (its temporaries reuse A's), so it may only ever
        # be replayed after A and the exchange of the same iteration -- never alone, never twice
        self.graph.replay()
        if self.world > 1:
            self._exchange()
            self.graph_b.replay()
        ops.rng_advance(self.rng_calls)
        for b, d in self.bn_delta:
            b._pending += d
        ops.weights_changed()
        self.replays += 1
        return self._finish(self.losses, self.loss, self.model_out)


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


class StepLrUpdaterHook(object):
    """mmcv 0.4.4 `LrUpdaterHook` with policy='step' (the `lr_config` of every north-star config:
    config/cfg_kitti_baseline_kitti_odom_4gpus.py:84-91): lr = base * gamma ** (number of `step` epochs passed), with an
    optional linear warm-up over the first `warmup_iters` iterations starting at warmup_ratio * lr."""

    def __init__(self, step, gamma=0.1, warmup=None, warmup_iters=0, warmup_ratio=0.1, policy="step", **kw):
        if policy != "step":
            raise NotImplementedError("only policy='step' (all north-star configs)")
        if warmup not in (None, "constant", "linear", "exp"):
            raise ValueError(f'"{warmup}" is not a supported type for warming up')
        self.step, self.gamma = step, gamma
        self.warmup, self.warmup_iters, self.warmup_ratio = warmup, warmup_iters, warmup_ratio
        self.base_lr, self.regular_lr = None, None

    def get_lr(self, epoch, base_lr):
        if isinstance(self.step, int):
            return base_lr * (self.gamma ** (epoch // self.step))
        exp = len(self.step)
        for i, s in enumerate(self.step):
            if epoch < s:
                exp = i
                break
        return base_lr * self.gamma ** exp

    def get_warmup_lr(self, cur_iters):
        if self.warmup == "constant":
            return [lr * self.warmup_ratio for lr in self.regular_lr]
        if self.warmup == "linear":
            k = (1 - cur_iters / self.warmup_iters) * (1 - self.warmup_ratio)
            return [lr * (1 - k) for lr in self.regular_lr]
        k = self.warmup_ratio ** (1 - cur_iters / self.warmup_iters)
        return [lr * k for lr in self.regular_lr]

    @staticmethod
    def _set_lr(runner, lrs):
        for g, lr in zip(runner.optimizer.param_groups, lrs):
            g["lr"] = lr

    def before_run(self, runner):
        for g in runner.optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
        self.base_lr = [g["initial_lr"] for g in runner.optimizer.param_groups]

    def before_train_epoch(self, runner):
        if self.base_lr is None:
            self.before_run(runner)
        self.regular_lr = [self.get_lr(runner.epoch, b) for b in self.base_lr]
        self._set_lr(runner, self.regular_lr)

    def before_train_iter(self, runner):
        if self.warmup is None or runner.iter > self.warmup_iters:
            return
        if self.regular_lr is None:
            self.before_train_epoch(runner)
        if runner.iter == self.warmup_iters:
            self._set_lr(runner, self.regular_lr)
        else:
            self._set_lr(runner, self.get_warmup_lr(runner.iter))


class Runner(object):
    """The slice of mmcv.Runner the hot loop needs: model, batch_processor, optimizer, outputs, one iteration, the
    step-policy learning-rate hook, and checkpoints in mmcv's layout (save / load / resume)."""

    def __init__(self, model, batch_processor, optimizer, optimizer_hook, lr_config=None, work_dir=None,
                 checkpoint_config=None, step_graph=None):
        """`checkpoint_config` = the configs' `dict(interval=1)` (mmcv CheckpointHook): `train_epoch` then saves from its
        after-train-epoch point, i.e. BEFORE the epoch counter is incremented, exactly where mmcv's hook runs.
        `step_graph` (not a reference option): replay the whole iteration from ONE captured hipGraph (`CapturedStep`) -- for
        the B = 1 configs, whose ~1 500 launches per step are host-bound when issued one by one.  None = $JP_STEP_GRAPH
        (default off)."""
        self.model, self.batch_processor, self.optimizer, self.hook = model, batch_processor, optimizer, optimizer_hook
        if step_graph is None:
            step_graph = os.environ.get("JP_STEP_GRAPH", "0") not in ("0", "")
        self.step_graph = bool(step_graph)
        self.captured = None
        self.outputs = None
        self.iter = 0
        self.epoch = 0
        self.work_dir = work_dir
        self.checkpoint_config = dict(checkpoint_config) if checkpoint_config else None
        self.after_train_epoch_hooks = []          # callables(runner), run before the epoch counter moves (mmcv order)
        self.after_train_iter_hooks = []           # callables(runner), after the optimizer hook (log hooks)
        self.lr_hook = StepLrUpdaterHook(**lr_config) if lr_config else None
        if self.lr_hook is not None:
            self.lr_hook.before_run(self)

    def current_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def train_iter(self, data_batch):
        self.model.train()
        if self.lr_hook is not None:
            self.lr_hook.before_train_iter(self)
        if self.step_graph:
            if self.captured is None:
                self.captured = CapturedStep(self)
            self.outputs = self.captured.step(data_batch)
            lv = self.outputs["log_vars"]
            if isinstance(lv, LazyLogVars):
                lv._resolve()
            for h in self.after_train_iter_hooks:
                h(self)
            self.iter += 1
            return self.outputs
        self.outputs = self.batch_processor(self.model, data_batch, train_mode=True)
        self.hook.after_train_iter(self)
        lv = self.outputs.get("log_vars") if isinstance(self.outputs, dict) else None
        if isinstance(lv, LazyLogVars):
            lv._resolve()       # backward + optimizer are enqueued: waiting for the forward's scalars stalls nothing now
        for h in self.after_train_iter_hooks:
            h(self)
        self.iter += 1
        return self.outputs

    def run(self, data_loaders, workflow, max_epochs, **kwargs):
        """mmcv.Runner.run for the configs' `workflow = [('train', 1)]`: cycle the workflow's phases until `max_epochs`
        epochs have been trained (a resumed runner continues at its restored epoch)."""
        if len(data_loaders) != len(workflow):
            raise ValueError("one data loader per workflow phase")
        while self.epoch < max_epochs:
            for (mode, epochs), loader in zip(workflow, data_loaders):
                if mode != "train":
                    raise NotImplementedError("only ('train', n) workflow phases (all north-star configs)")
                for _ in range(epochs):
                    if self.epoch >= max_epochs:
                        return
                    self.train_epoch(loader)

    def train_epoch(self, data_loader):
        """mmcv.Runner.train: before_train_epoch hooks, one pass over the loader, after_train_epoch hooks (the checkpoint
        hook among them), THEN epoch += 1 -- so the file written after the first epoch is epoch_1.pth with meta.epoch = 1,
        and `resume()` continues with epoch 1 (second epoch), the step-LR schedule and the samplers' set_epoch in step with
        the reference."""
        if self.lr_hook is not None:
            self.lr_hook.before_train_epoch(self)
        if hasattr(getattr(data_loader, "sampler", None), "set_epoch"):
            data_loader.sampler.set_epoch(self.epoch)
        for batch in data_loader:
            self.train_iter(batch)
        for h in self.after_train_epoch_hooks:
            h(self)
        cc = self.checkpoint_config
        if cc is not None and (self.epoch + 1) % int(cc.get("interval", 1)) == 0:
            self.save_checkpoint(cc.get("out_dir"), save_optimizer=cc.get("save_optimizer", True))
        self.epoch += 1

    # ---- checkpoints (mmcv.Runner.save_checkpoint / load_checkpoint / resume)
    def save_checkpoint(self, out_dir=None, filename_tmpl="epoch_{}.pth", save_optimizer=True, meta=None):
        """mmcv.Runner.save_checkpoint: names the file and stamps meta.epoch with `self.epoch + 1`, because mmcv calls it
        from the after-train-epoch point of epoch `self.epoch`, before the counter moves.  Call it there too
        (`checkpoint_config=` / `after_train_epoch_hooks`); a manual call AFTER `train_epoch()` returned would label the file
        one epoch ahead, which `resume()` would then take at face value."""
        from .checkpoint import save_checkpoint
        meta = dict(meta or {}, epoch=self.epoch + 1, iter=self.iter)
        path = os.path.join(out_dir or self.work_dir or ".", filename_tmpl.format(self.epoch + 1))
        live = dist.is_available() and dist.is_initialized()
        if not live or dist.get_rank() == 0:    # mmcv's CheckpointHook.after_train_epoch is @master_only
            save_checkpoint(self.model, path, optimizer=self.optimizer if save_optimizer else None, meta=meta)
            # mmcv.Runner.save_checkpoint also points `latest.pth` at the new file (several reference configs resume from it:
            # cfg_kitti_baseline_kitti_odom_4pugsB12_lr1e-4_ce_eigen.py:61)
            latest = os.path.join(os.path.dirname(path) or ".", "latest.pth")
            tmp = latest + ".tmp"
            try:
                if os.path.lexists(tmp):
                    os.remove(tmp)
                os.symlink(os.path.basename(path), tmp)
                os.replace(tmp, latest)
            except OSError:                     # no symlinks on this file system: a copy
                import shutil
                shutil.copyfile(path, tmp)
                os.replace(tmp, latest)
        if live and dist.get_world_size() > 1:
            dist.barrier()                      # nobody resumes from / trains past a file that is still being written
        return path

    def load_checkpoint(self, filename, map_location="cpu", strict=False):
        from .checkpoint import load_checkpoint
        return load_checkpoint(self.model, filename, map_location, strict)

    def resume(self, checkpoint, resume_optimizer=True, map_location="cpu"):
        ckpt = self.load_checkpoint(checkpoint, map_location=map_location)
        self.epoch = ckpt["meta"]["epoch"]
        self.iter = ckpt["meta"]["iter"]
        if "optimizer" in ckpt and resume_optimizer:
            self.optimizer.load_state_dict(ckpt["optimizer"])
        return ckpt


def _cfg(cfg, name, default=None):
    """attribute / item / `.get` access over an mmcv `Config`, a dict or a namespace"""
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    v = getattr(cfg, name, default)
    return default if v is None else v


class _LogHook(object):
    """the slice of mmcv's TextLoggerHook `log_config=dict(interval=n, ...)` asks for: one line per `interval` iterations"""

    def __init__(self, logger, interval):
        self.logger, self.interval, self.n = logger, max(1, int(interval)), 0

    def __call__(self, runner):
        self.n += 1
        if self.logger is not None and self.n % self.interval == 0:
            lv = runner.outputs["log_vars"]
            self.logger.info("Epoch [%d] iter %d lr %.3e loss %.5f", runner.epoch + 1, runner.iter, runner.current_lr()[0],
                             lv["loss"])


_VAL_KEYS = ("abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3", "scale")


def _class1(lst):
    """eval_hooks.py:186-199: `np.array([0., 0.]) += mean_IU(...)` then `[1]` -- class 1's entry; a one-entry list (only one class
    in the union) broadcasts into both slots there."""
    return float(lst[1] if len(lst) == 2 else (lst[0] if len(lst) == 1 else 0.0))


def _validate_hook(dataset_val, cfg, logger):
    """DistEvalMonoHook (mono/core/evaluation/eval_hooks.py:120-262, 268-327) reduced to its numbers.  After every
    `validate_interval` epochs: the eval-mode forward over THIS rank's slice `range(rank, len(dataset), world_size)` (:128); per
    item the seven depth metrics + `scale` with median scaling (`cfg.data['stereo_scale']` -> x36; zeros for items without
    `gt_depth`, :203-224) and `iou_road / mAP_road` (`topview` vs `("bothS", 0, 0)`), `iou_vehicle / mAP_vehicle` (`topviewB` vs
    `("bothD", 0, 0)`) (:181-199, 225-228); the per-item dicts are gathered on rank 0 (the reference goes through pickle files in
    work_dir, here `all_gather_object`), averaged over the whole set (:296-326) and left in `runner.eval_result` on every rank."""
    from ..core import evaluation as ev
    interval = int(_cfg(cfg, "validate_interval", 1))
    data_cfg = _cfg(cfg, "data", {}) or {}
    stereo = bool(data_cfg.get("stereo_scale", False)) if isinstance(data_cfg, dict) else bool(getattr(data_cfg, "stereo_scale", False))

    def hook(runner):
        if (runner.epoch + 1) % interval or dataset_val is None:
            return
        live = dist.is_available() and dist.is_initialized()
        rank, world = (dist.get_rank(), dist.get_world_size()) if live else (0, 1)
        m = getattr(runner.model, "module", runner.model)
        was = m.training
        m.eval()
        mine = []
        with torch.no_grad():
            for idx in range(rank, len(dataset_val), world):
                item = dataset_val[idx]
                inputs = {k: torch.as_tensor(v).float().unsqueeze(0).cuda() for k, v in item.items()
                          if k != "gt_depth" and not isinstance(v, str)}
                out = m(inputs)
                res = dict.fromkeys(_VAL_KEYS, 0.0)
                if "gt_depth" in item:
                    gt = torch.as_tensor(item["gt_depth"]).float().cuda()
                    r = ev.eval_depth(out[("disp", 0, 0)], gt, stereo_scale=stereo)
                    res.update({k: r[k] for k in _VAL_KEYS})
                for name, key, tag in (("topview", ("bothS", 0, 0), "road"), ("topviewB", ("bothD", 0, 0), "vehicle")):
                    if name in out and key in inputs:
                        iu, prec = ev.eval_layout(out[name], inputs[key])[0]
                        res["iou_" + tag], res["mAP_" + tag] = _class1(iu), _class1(prec)
                mine.append((idx, res))
        if world > 1:
            parts = [None] * world
            dist.all_gather_object(parts, mine)
            mine = [p for part in parts for p in part]
        mine.sort(key=lambda p: p[0])
        if mine:
            keys = sorted({k for _, r in mine for k in r})
            avg = {k: float(sum(r.get(k, 0.0) for _, r in mine) / len(mine)) for k in keys}
            avg["scale mean"] = avg.pop("scale")
            runner.eval_result = (avg, len(mine))
            if logger is not None and rank == 0:
                logger.info("validation after epoch %d over %d items: %s", runner.epoch + 1, len(mine), avg)
        m.train(was)
    return hook


def _train(model, dataset_train, dataset_val, cfg, validate, logger, distributed):
    from ..core.dist_utils import DistOptimizerHook
    from ..datasets.loader import DeviceLoader, build_dataloader
    gpus = _cfg(cfg, "gpus", [0])
    loader = build_dataloader(dataset_train, _cfg(cfg, "imgs_per_gpu", 1), _cfg(cfg, "workers_per_gpu", 0),
                              num_gpus=1 if distributed else max(1, len(gpus)), dist=distributed)
    # `device` is not a reference option: the host-logic tests run this wiring on CPU with a stub step
    dev = torch.device(_cfg(cfg, "device", "cuda"))
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    model = model.to(dev)
    # MMDistributedDataParallel / MMDataParallel stand-in: `.module`, rank 0's weights everywhere; gradients are averaged
    # ONCE, on the flat arena, by the optimizer hook (one process per GPU also in the "non-distributed" launch)
    model = DataParallelShell(model)
    optimizer = build_optimizer(model, _cfg(cfg, "optimizer", dict(type="Adam", lr=1e-4, weight_decay=0)))
    oc = dict(_cfg(cfg, "optimizer_config", {}) or {})
    work_dir = _cfg(cfg, "work_dir", ".")
    runner = Runner(model, batch_processor, optimizer, DistOptimizerHook(**oc), lr_config=_cfg(cfg, "lr_config"),
                    work_dir=work_dir, checkpoint_config=_cfg(cfg, "checkpoint_config"))
    lc = _cfg(cfg, "log_config")
    if lc is not None and logger is not None:
        runner.after_train_iter_hooks.append(_LogHook(logger, dict(lc).get("interval", 50)))
    if validate:
        runner.after_train_epoch_hooks.append(_validate_hook(dataset_val, cfg, logger))
    if _cfg(cfg, "resume_from"):
        runner.resume(_cfg(cfg, "resume_from"))
    elif _cfg(cfg, "load_from"):
        runner.load_checkpoint(_cfg(cfg, "load_from"))
    # pinned double-buffered upload underneath the step (datasets/loader.py); `.sampler` stays visible for set_epoch
    runner.run([DeviceLoader(loader, dev) if dev.type == "cuda" else loader], _cfg(cfg, "workflow", [("train", 1)]), _cfg(cfg, "total_epochs", 1))
    return runner


def _dist_train(model, dataset_train, dataset_val, cfg, args=None, validate=False, logger=None):
    """trainer.py:146-199: DistributedGroupSampler loader, DDP wrap, optimizer, runner with lr / optimizer / checkpoint /
    log hooks and the sampler-seed hook (`Runner.train_epoch` calls `sampler.set_epoch`), resume_from | load_from, run."""
    return _train(model, dataset_train, dataset_val, cfg, validate, logger, True)


def _non_dist_train(model, dataset_train, dataset_val, cfg, validate=False, logger=None):
    """trainer.py:202-240: GroupSampler loader over `len(cfg.gpus)` x imgs_per_gpu; this build drives ONE device per
    process, so cfg.gpus longer than one entry is refused instead of silently training on a single GPU."""
    if len(_cfg(cfg, "gpus", [0])) > 1:
        raise NotImplementedError("single-process multi-GPU (MMDataParallel) is not built: launch one process per GPU "
                                  "(`--launcher pytorch`), the reference's own multi-GPU recipe")
    return _train(model, dataset_train, dataset_val, cfg, validate, logger, False)


def train_mono(model, dataset_train, dataset_val, cfg, args=None, distributed=False, validate=False, logger=None):
    """trainer.py:59-73, same signature: `train.py:89-96` runs unchanged with `from jperceiver_amd.apis import train_mono`.
    Returns the runner (the reference returns None)."""
    if logger is None:          # what train.py:78 passes: handler attached, INFO on rank 0, ERROR elsewhere
        from .env import get_root_logger
        logger = get_root_logger(_cfg(cfg, "log_level", "INFO"))
    if distributed:
        return _dist_train(model, dataset_train, dataset_val, cfg, args, validate=validate, logger=logger)
    return _non_dist_train(model, dataset_train, dataset_val, cfg, validate=validate, logger=logger)
