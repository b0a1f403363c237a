"""Host-side operator layer: every function here launches hand-written HIP kernels through the C ABI
(jperceiver_amd/_lib.py) and, when a `Tape` is active, records the matching backward launches.

Why an own tape instead of torch.autograd: the train step must run *only* our kernels (no ATen
gradient-accumulation adds, no autograd-engine thread hopping), gradients of parameters are written
straight into one flat arena (so the RCCL all-reduce and the fused clip+Adam see a single buffer),
and the recorded launch sequence is static per shape — the prerequisite for hipGraph capture.
PyTorch supplies device memory (caching allocator), streams and nn.Module/state_dict plumbing.
"""
from __future__ import annotations

import torch

from ._lib import call, lib as _jplib

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SIGMOID = 0, 1, 2, 3
PAD_ZERO, PAD_REFLECT = 0, 1


# ------------------------------------------------------------------------------------------- tape
class Var:
    """A device tensor plus its (lazily allocated) gradient buffer.  `p`: the nn.Parameter a weight Var was made from
    (ops.param) -- conv weights of parameters get persistent packed copies (PackRegistry), raw tensors do not."""
    __slots__ = ("t", "g", "rg", "p", "amax", "gamax", "bnst")

    def __init__(self, t: torch.Tensor, rg: bool = False, g: torch.Tensor | None = None, p=None):
        self.t, self.rg, self.g, self.p = t, rg, g, p
        # device scalars for the fp16 split kernels' operand scales: max|t| (or an upper bound: what the producer reported / a bound
        # handed through a max-pool or an activation; reduced on first request otherwise, _amax_of) and max|g| while g is exactly
        # what ONE producer wrote (dropped as soon as anything is accumulated into g)
        self.amax = None
        self.gamax = None
        # (scratch, parts): BatchNorm statistics partials the producing convolution's epilogue left (conv2d(..., bn_stats=True))
        self.bnst = None

    def grad_buf(self):
        """-> (buffer, accumulate flag) for kernels that can either write or add."""
        self.gamax = None
        if self.g is None:
            self.g = torch.empty_like(self.t)
            return self.g, 0
        return self.g, 1

    def add_grad(self, d: torch.Tensor, amax: torch.Tensor | None = None):
        """d must be a fresh tensor owned by the caller (amax: device scalar max|d| if its producer reported it)."""
        if self.g is None:
            self.g = d
            self.gamax = amax
        else:
            call("jp_axpby", self.g, d, self.g, d.numel(), 1.0, 1.0)
            self.gamax = None

    @property
    def shape(self):
        return self.t.shape


class Tape:
    def __init__(self):
        self.nodes = []

    def record(self, fn):
        self.nodes.append(fn)

    def backward(self):
        while self.nodes:
            self.nodes.pop()()


_TAPE: Tape | None = None


class recording:
    def __init__(self, tape: Tape):
        self.tape = tape

    def __enter__(self):
        global _TAPE
        self.prev, _TAPE = _TAPE, self.tape
        return self.tape

    def __exit__(self, *a):
        global _TAPE
        _TAPE = self.prev


def _rec(needs, fn):
    if _TAPE is not None and needs:
        _TAPE.record(fn)


# ---- gradient-ready markers.  `grad_ready(tag)` is recorded BEFORE a module's forward ops, so the tape replays it right
# AFTER that module's backward: every parameter gradient of arena segment `tag` is final at that point of the stream the
# tape is being replayed on.  The data-parallel optimizer hook installs a callback that starts the segment's all-reduce
# there, underneath the rest of the backward (core/dist_utils.py); without a callback the marker costs nothing.
_READY_HOOK = None


def set_grad_ready_hook(fn):
    global _READY_HOOK
    prev, _READY_HOOK = _READY_HOOK, fn
    return prev


def grad_ready(tag: str):
    def fire():
        if _READY_HOOK is not None:
            join_param_grad_streams()       # the segment's wgrad kernels run on the companion stream
            _READY_HOOK(tag)

    _rec(True, fire)


# ---- companion streams for parameter-gradient kernels (JP_WGRAD_STREAM=0 turns them off)
import os as _os
_WG_ON = _os.environ.get("JP_WGRAD_STREAM", "1") != "0"
_WG_STREAMS = {}
_WG_ACTIVE = [False]
_WG_MAIN_ONLY = _os.environ.get("JP_WGRAD_SIDE", "1") == "0"     # companion stream for the main stream's backward only
_WG_MAIN = [None]                                                  # cuda_stream handle the step's backward started on
_WG_DEFERRED = []


def _wgrad_stream():
    """The companion stream of the CURRENT stream while a tape is being replayed for a training step; None otherwise."""
    if not _WG_ON or not _WG_ACTIVE[0]:
        return None
    base = torch.cuda.current_stream()
    if _WG_MAIN_ONLY and _WG_MAIN[0] is not None and base.cuda_stream != _WG_MAIN[0]:
        return None
    key = (base.device, base.cuda_stream)
    st = _WG_STREAMS.get(key)
    if st is None:
        # (confining this stream or the side stream to a subset of the CUs -- hipExtStreamCreateWithCUMask, halves / quarters, disjoint
        # or not -- was measured in round 5: 82.4 -> 98-109 ms per step, profiles/r05_cumask_ab.log; stream priorities in round 4: no effect)
        st = _WG_STREAMS[key] = torch.cuda.Stream(device=base.device)
    return st


def join_param_grad_streams():
    """Make the current stream wait for the parameter-gradient kernels launched on its companion stream so far."""
    if not _WG_STREAMS:              # no companion stream was ever created (CPU-side tests of the tape / exchange logic)
        return
    base = torch.cuda.current_stream()
    st = _WG_STREAMS.get((base.device, base.cuda_stream))
    if st is None:
        return
    if _WG_MAIN[0] is not None and base.cuda_stream != _WG_MAIN[0] and torch.cuda.is_current_stream_capturing():
        # hipGraph capture (apis.trainer.CapturedStep): a stream forked from a forked stream may only be joined into the ORIGIN
        # stream -- hipStreamEndCapture dies on `side.wait_stream(companion of side)` (tools/debug/graph_streams_micro2.py:
        # "mainfirst" / "twice" crash, "flatjoin" is fine).  Nothing on the side stream reads parameter gradients, so the
        # companion is handed to the step's backward (model/net.py _StepFn), which joins it into the origin stream at its end.
        if st not in _WG_DEFERRED:
            _WG_DEFERRED.append(st)
        return
    base.wait_stream(st)


def join_deferred_param_grad_streams():
    """origin stream <- the side streams' companions whose join was deferred during a capture (see above)"""
    base = torch.cuda.current_stream()
    while _WG_DEFERRED:
        base.wait_stream(_WG_DEFERRED.pop())


def as_var(x) -> Var:
    return x if isinstance(x, Var) else Var(x.contiguous() if not x.is_contiguous() else x)


def param(p: torch.nn.Parameter) -> Var:
    """Parameter view: gradients accumulate into p.grad (a slice of the flat grad arena, zeroed by
    zero_grad) so a weight used several times per step (PoseEncoder runs twice) just adds up."""
    if p.grad is None and p.requires_grad:
        p.grad = torch.zeros_like(p.data)
    return Var(p.data, p.requires_grad, p.grad, p)


# ------------------------------------------------------------------------------------------- packed weights
# The MFMA convolutions read their weights from a packed copy (csrc/conv.hip "Weight packing").  For weights that are
# nn.Parameters the copy is PERSISTENT: the first call of a layer packs it and records the pack jobs the library issued;
# afterwards `PackRegistry.refresh_all()` (called by Baseline.forward) re-packs every layer of the model with ONE
# kernel launch whenever the weights changed (FlatAdam.step / load_state_dict / .to() bump the epoch, in-place edits of
# a parameter are caught through its version counter), and the conv entry points run with ws_state=1.
import weakref

import numpy as np

_JOB_DT = np.dtype([("w", "u8"), ("wp", "u8"), ("total", "i8"), ("begin", "i8"), ("mode", "i4"), ("p", "i4", (6,)),
                    ("pad", "i4")])
_EPOCH = [0]


def weights_changed():
    """Tell the pack cache that parameter memory was rewritten behind autograd's back (optimizer kernels, .data edits)."""
    _EPOCH[0] += 1


class _PackEntry:
    """One layer's packed-weight scratch.  The parameter is held WEAKLY: when its model is dropped the entry (scratch and pack
    jobs) goes with it instead of being replayed for ever."""
    __slots__ = ("ws", "jobs", "ptr", "epoch", "_param", "ver")

    @property
    def param(self):
        return self._param()


class PackRegistry:
    _by_device = {}

    def __init__(self, device):
        assert int(_jplib().fn["jp_pack_job_bytes"]()) == _JOB_DT.itemsize
        self.device = device
        self.entries = {}
        self.table = None            # (device jobs tensor, njobs, total elements)
        self.versions = None

    @classmethod
    def of(cls, device):
        r = cls._by_device.get(device)
        if r is None:
            r = cls._by_device[device] = PackRegistry(device)
        return r

    def scratch(self, w: Var, which: str, sig, nfloats: int):
        """-> (ws, ws_state, entry-or-None, record?) for one conv launch."""
        key = (id(w.p), which, sig)
        e = self.entries.get(key)
        ptr = w.t.data_ptr()
        if e is None or e.ptr != ptr or e.ws.numel() != nfloats or e.param is not w.p:
            e = _PackEntry()
            e.ws, e.jobs, e.ptr, e.epoch, e.ver = _new((nfloats,), w.t), None, ptr, -1, -1
            e._param = weakref.ref(w.p, lambda _r, key=key, reg=weakref.ref(self), drop=PackRegistry._drop: drop(reg, key, _r))
            self.entries[key] = e
            self.table = None
        return e

    @staticmethod
    def _drop(reg, key, ref):
        self = reg()
        if self is None:
            return
        e = self.entries.get(key)
        if e is not None and e._param is ref:        # (the id may already belong to a newer parameter's entry)
            del self.entries[key]
            self.table = None

    def ensure_table(self):
        """Build the device job table of every live pack if it is missing (a host-to-device copy: CapturedStep calls this
        BEFORE it starts capturing, refresh_all inside the capture then only launches).  -> False when there is nothing to replay."""
        if self.table is None:
            live = [e for e in self.entries.values()
                    if e.jobs is not None and e.param is not None and e.ptr == e.param.data_ptr()]
            jobs = np.concatenate([e.jobs for e in live]) if live else None
            if jobs is None or len(jobs) == 0:
                return False
            # the split packs (re-packed by their own LDS-staged kernel) go to the tail of the table: the generic kernel then walks
            # only the range in front of them (jp_pack_replay generic_elems)
            is_split = _jplib().fn["jp_pack_mode_is_split"]
            tail = np.array([bool(is_split(int(m))) for m in jobs["mode"]])
            jobs = np.concatenate([jobs[~tail], jobs[tail]])
            # jp_pack_replay walks the concatenated range in groups of 4 elements: every job begins on a multiple of 4
            padded = (jobs["total"] + 3) // 4 * 4
            jobs["begin"] = np.concatenate([[0], np.cumsum(padded)[:-1]])
            dev = torch.from_numpy(jobs.view(np.uint8).copy()).to(self.device)
            self.table = (dev, len(jobs), int(padded.sum()), live, int(padded[:int((~tail).sum())].sum()))
        return True

    def refresh_all(self):
        """One launch re-packs every registered layer if any weight changed since the last refresh."""
        if not self.entries:
            return
        vs = 0
        for e in list(self.entries.values()):
            p = e.param
            if p is not None:
                vs += p._version
        if vs != self.versions:
            if self.versions is not None:
                weights_changed()
            self.versions = vs
        ep = _EPOCH[0]
        stale = [e for e in self.entries.values() if e.epoch != ep and e.jobs is not None]
        if not stale:
            return
        if not self.ensure_table():
            return
        dev, n, total, live, generic = self.table
        call("jp_pack_replay", dev, n, total, generic)
        for e in live:
            p = e.param
            if p is not None:
                e.epoch, e.ver = ep, p._version


def _conv_call(name, w: Var, which: str, sig, nfloats: int, args_before_ws, args_after_ws):
    """Launch a conv entry point that takes (ws, ws_state): persistent pack for parameter weights, per-call scratch otherwise."""
    if nfloats == 0:
        call(name, *args_before_ws, None, 0, *args_after_ws)
        return
    if w.p is None:
        ws = _new((nfloats,), w.t)
        call(name, *args_before_ws, ws, 0, *args_after_ws)
        return
    reg = PackRegistry.of(w.t.device)
    e = reg.scratch(w, which, sig, nfloats)
    if e.jobs is None:            # first use: pack now and record what the library packed
        buf = np.zeros(16, dtype=_JOB_DT)
        L = _jplib()
        if L.fn["jp_pack_record_begin"](buf.ctypes.data, len(buf)) != 0:
            raise RuntimeError(L.last_error())
        try:
            call(name, *args_before_ws, e.ws, 0, *args_after_ws)
        finally:
            n = int(L.fn["jp_pack_record_end"]())
        if n > len(buf):
            raise RuntimeError(f"{name}: {n} weight packs in one call (record buffer too small)")
        e.jobs, e.epoch, e.ver = buf[:n].copy(), _EPOCH[0], w.p._version
        reg.table = None
    elif e.epoch != _EPOCH[0] or e.ver != w.p._version:
        # weights changed and nobody refreshed the registry (a sub-network used standalone -- inference.pose_between,
        # a module's own forward -- after an in-place load_state_dict / .copy_ on its parameters): re-pack this layer alone
        call(name, *args_before_ws, e.ws, 0, *args_after_ws)
        e.epoch, e.ver = _EPOCH[0], w.p._version
    else:
        call(name, *args_before_ws, e.ws, 1, *args_after_ws)


# ---- operand scales of the fp16 two-way split kernels (csrc/scale.hip): a tensor's largest magnitude is reduced ONCE and handed to
# every convolution call that reads the tensor (forward + weight gradient for an activation, dgrad + weight gradient for a gradient)
_SPLIT_SCHEME = []


def split_scheme() -> int:
    """2: the library's patch kernels use two fp16 splits per operand (three products), 3: three bf16 splits (six products)."""
    if not _SPLIT_SCHEME:
        _SPLIT_SCHEME.append(int(_jplib().fn["jp_split_scheme"]()))
    return _SPLIT_SCHEME[0]


def _amax_of(v) -> torch.Tensor | None:
    """Device scalar max|v|, reduced on first request and cached on a Var; None when the library does not use operand scales."""
    if split_scheme() != 2:
        return None
    if isinstance(v, Var):
        if v.amax is None:
            v.amax = _amax_of(v.t)
        return v.amax
    out = _amax_slot(v.device)
    call("jp_amax_into", v, v.numel(), out)
    if _AMAX_LOG is not None:
        import traceback
        fr = [f for f in traceback.extract_stack(limit=12) if "ops.py" not in f.filename]
        key = (tuple(v.shape), fr[-1].name + ":" + str(fr[-1].lineno) if fr else "?")
        _AMAX_LOG[key] = _AMAX_LOG.get(key, 0) + 1
    return out


import os as _os
_AMAX_LOG = {} if _os.environ.get("JP_AMAX_LOG") else None
if _AMAX_LOG is not None:
    import atexit

    def _dump():
        rows = sorted(_AMAX_LOG.items(), key=lambda kv: -kv[1] * int(np.prod(kv[0][0])))
        tot = sum(c * int(np.prod(k[0])) * 4 for k, c in rows)
        print(f"[amax log] {sum(c for _, c in rows)} reductions, {tot / 1e9:.2f} GB")
        for (shape, where), c in rows[:40]:
            print(f"[amax log] {c:4d} x {str(shape):28s} {c * int(np.prod(shape)) * 4 / 1e6:9.1f} MB  {where}")
    atexit.register(_dump)


_AMAX_POOL = {}


def amax_pool_reset():
    """Forget the zeroed slot pools (before and after a graph capture: a pool zeroed inside one capture must not serve another)."""
    _AMAX_POOL.clear()


def _amax_slot(dev) -> torch.Tensor:
    """One zeroed magnitude slot (jp_amax_slot_floats floats) of a pool that is zeroed 1024 slots at a time (one fill instead of a memset
    per reduction).  A pool belongs to the stream it was zeroed on -- and to the graph capture it was zeroed in: a captured step replays
    the fill with the reductions."""
    SF = _slot_floats()
    st = torch.cuda.current_stream(dev)
    key = (dev.index, st.cuda_stream, torch.cuda.is_current_stream_capturing())
    ent = _AMAX_POOL.get(key)
    if ent is None or ent[1] >= 1024:
        if len(_AMAX_POOL) > 64:
            _AMAX_POOL.clear()
        ent = _AMAX_POOL[key] = [torch.zeros(1024 * SF, device=dev, dtype=torch.float32), 0]
    ent[1] += 1
    return ent[0][(ent[1] - 1) * SF:ent[1] * SF]


_SLOT_FLOATS = []


def _slot_floats() -> int:
    if not _SLOT_FLOATS:
        _SLOT_FLOATS.append(int(_jplib().fn["jp_amax_slot_floats"]()))
    return _SLOT_FLOATS[0]


def _amax_ws(dev, *amaxes):
    """Caller scratch for the magnitudes a conv entry point has to reduce itself (`amax_ws`): None when every operand's
    magnitude is handed over (or the library has no operand scales), else jp_conv2d_amax_ws_floats floats."""
    if split_scheme() != 2 or all(a is not None for a in amaxes):
        return None
    SF, n = _slot_floats(), _ws_slots()
    st = torch.cuda.current_stream(dev)
    key = ("ws", dev.index, st.cuda_stream, torch.cuda.is_current_stream_capturing())
    ent = _AMAX_POOL.get(key)
    if ent is None:
        # one scratch per stream: the calls of a stream run in order, so the next call may overwrite what the last one reduced
        ent = _AMAX_POOL[key] = [torch.empty(n * SF, device=dev, dtype=torch.float32), 0]
    return ent[0]


_WS_SLOTS = []


def _ws_slots() -> int:
    if not _WS_SLOTS:
        _WS_SLOTS.append(int(_jplib().fn["jp_conv2d_amax_ws_floats"]()) // _slot_floats())
    return _WS_SLOTS[0]


def _out_slot(dev, on=True):
    """A zeroed slot for a producer's `amax_y` / `amax_dx` argument (None: the library has no operand scales / not wanted)."""
    return _amax_slot(dev) if (on and split_scheme() == 2) else None


import ctypes as _ct


def _ws_floats(Cin, Cout, KH, which):
    """Caller-owned packed-weight scratch of the conv fast path (jp_conv2d_ws_floats)."""
    return int(_jplib().fn["jp_conv2d_ws_floats"](Cin, Cout, KH, which))


def _pad32(c):
    return (c + 31) // 32 * 32


def _new(shape, like: torch.Tensor, dtype=None):
    return torch.empty(shape, device=like.device, dtype=dtype or like.dtype)


# ------------------------------------------------------------------------------------------- conv
def conv2d(x, w: Var, b: Var | None, stride=1, pad=0, pad_mode=PAD_ZERO, act=ACT_NONE, srcs=None, bn_stats=False) -> Var:
    """nn.Conv2d (+ReflectionPad2d, +bias, +activation epilogue).  `srcs` = [(Var, upsampled?)...] (<=3)
    feeds the conv with the channel-concat of the sources, half-resolution ones being read through a
    fused nearest-2x upsample (depth_decoder.py:68,76-77) — nothing is materialised."""
    if srcs is None:
        srcs = [(x, 0)]
    srcs = [(as_var(v), int(u)) for v, u in srcs]
    N = srcs[0][0].t.shape[0]
    H = srcs[0][0].t.shape[2] << srcs[0][1]
    W = srcs[0][0].t.shape[3] << srcs[0][1]
    for v, u in srcs:
        assert v.t.shape[2] << u == H and v.t.shape[3] << u == W and v.t.shape[0] == N
    Cout, Cin, KH, KW = w.t.shape
    assert KH == KW and sum(v.t.shape[1] for v, _ in srcs) == Cin
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - KH) // stride + 1
    y = _new((N, Cout, OH, OW), w.t)
    s3 = []
    for i in range(3):
        if i < len(srcs):
            s3 += [srcs[i][0].t, srcs[i][0].t.shape[1], srcs[i][1]]
        else:
            s3 += [None, 0, 0]
    bt = b.t if b is not None else None
    # scratch for the packed-weight (tap-major) fast path; None -> generic path (few reduction channels)
    nwf = _ws_floats(Cin, Cout, KH, 0)
    nsp = int(_jplib().fn["jp_conv2d_fwd_split_floats"](N, Cin, H, W, Cout, KH, stride, pad))
    ws_s = _new((nsp,), w.t) if nsp else None      # fixed-order split-K reduction of small-grid layers
    sig = (tuple(s3[1::3]), tuple(s3[2::3]), N, H, W, stride, pad, pad_mode)
    big = Cin >= 32 and Cout >= 32            # (the few-channel layers run direct kernels without operand scales)
    # operand magnitudes of the fp16 split kernels (explicit arguments of the entry points, include/jperceiver_hip.h): one slot per
    # source, reduced once per tensor (_amax_of) or reported by its producer; the few-channel layers leave it to the library
    x_am = [(_amax_of(srcs[i][0]) if (big and i < len(srcs)) else None) for i in range(3)]
    y_slot = _out_slot(y.device, big)
    y_done = _ct.c_int(0)
    # bn_stats: this convolution feeds a train-mode BatchNorm -- the 4-wave 3x3 patch kernels leave per-channel partial sums of y and
    # y^2 in `st_ws` (their epilogue), and BatchNorm folds those instead of reading y once more
    st_ws, st_parts = None, _ct.c_int(0)
    if bn_stats and big and KH == 3 and stride == 1 and bt is None and act == ACT_NONE and len(srcs) == 1:
        st_ws = _new((int(_jplib().fn["jp_conv2d_fwd_bn_stats_floats"](N, OH, OW, Cout)),), w.t)
    _conv_call("jp_conv2d_fwd_src3", w, "fwd", sig, nwf,
               (*s3, w.t, bt, y, N, H, W, Cout, KH, stride, pad, pad_mode, act),
               (ws_s, *x_am, y_slot, _ct.addressof(y_done),
                _amax_ws(y.device, *(x_am[:len(srcs)] if len(srcs) == 1 else (None,))),      # (several sources: their slots are folded into the scratch)
                st_ws, _ct.addressof(st_parts)))
    del ws_s
    out = Var(y, any(v.rg for v, _ in srcs) or w.rg)
    out.amax = y_slot if y_done.value else None   # max|y| from the kernel's epilogue when a patch kernel ran the layer: the next convolution's scale
    if st_parts.value > 0:
        out.bnst = (st_ws, st_parts.value)

    def bwd():
        if out.g is None:
            return
        dy = out.g
        dy_am = out.gamax             # what the producer of this gradient reported (BatchNorm backward), if nothing was added since
        bias_done = False
        if act != ACT_NONE:
            d2 = torch.empty_like(dy)
            ao = _out_slot(dy.device, big)
            if b is not None and b.rg and Cout <= 65535:
                # activation backward and bias gradient in one pass (the bias gradient is a sum over the tensor this pass writes)
                nbw = int(_jplib().fn["jp_act_bwd_bias_ws_floats"](N, Cout, OH * OW))
                call("jp_act_bwd_bias", dy, y, d2, b.g, N, Cout, OH * OW, act, ao, _new((nbw,), dy))    # (scratch: fixed-order fold)
                bias_done = True
            else:
                call("jp_act_bwd", dy, y, d2, dy.numel(), act, ao)
            dy, dy_am = d2, ao
        # (reduced here, on the tape's stream, before the wgrad stream forks off it, unless its producer reported it)
        if big and dy_am is None:
            dy_am = _amax_of(dy)
        elif not big:
            dy_am = None

        def param_grads():
            if b is not None and b.rg and not bias_done:
                ncw = int(_jplib().fn["jp_channel_sum_ws_floats"](N, Cout, OH * OW))
                call("jp_channel_sum", dy, b.g, N, Cout, OH * OW, 1, _new((ncw,), dy))
            if w.rg:
                wg_am = (*x_am, dy_am, _amax_ws(dy.device, dy_am, *x_am[:len(srcs)]))
                nms = 0
                if len(srcs) > 1:
                    nms = int(_jplib().fn["jp_conv2d_wgrad_src3_ws_floats"](s3[1], s3[2], s3[4], s3[5], s3[7], s3[8], N, H, W, Cout,
                                                                             KH, stride, pad, pad_mode))
                up_head = len(srcs) == 1 and srcs[0][1] and int(_jplib().fn["jp_conv2d_up_head_ok"](
                    s3[1], s3[2], 0, 0, Cout, KH, stride, pad, pad_mode, H, W))
                if up_head:     # disparity head on an upsampled source: upsample-aware direct kernel, nothing materialised
                    # 16 gathered dY sums per half-resolution pixel + the workgroups' partial sums (folded in a fixed order)
                    nup = int(_jplib().fn["jp_conv2d_wgrad_src3_ws_floats"](s3[1], s3[2], 0, 0, 0, 0, N, H, W, Cout, KH, stride, pad,
                                                                             pad_mode))
                    ws_w = _new((nup,), dy)
                    call("jp_conv2d_wgrad_src3", *s3, dy, w.g, N, H, W, Cout, KH, stride, pad, pad_mode, 1, ws_w, nup, *wg_am)
                    del ws_w
                elif nms:
                    # per-segment wgrad inside the library: full-resolution segments from their own tensors, the
                    # upsampled one in parity-class form -- no materialised concat
                    ws_w = _new((nms,), dy)
                    call("jp_conv2d_wgrad_src3", *s3, dy, w.g, N, H, W, Cout, KH, stride, pad, pad_mode, 1, ws_w, nms, *wg_am)
                    del ws_w
                elif (len(srcs) > 1 or srcs[0][1]) and Cin >= 32:
                    # materialise the virtual upsample+concat once: the single-source wgrad gather is ~2x faster
                    xc = _new((N, Cin, H, W), dy)
                    c0 = 0
                    for v, u in srcs:
                        C = v.t.shape[1]
                        if u:
                            call("jp_upsample2x_fwd", v.t, xc, N, C, H // 2, W // 2, Cin, c0)
                        else:
                            call("jp_copy_channels", v.t, xc, N, C, H * W, C, 0, Cin, c0, 0)
                        c0 += C
                    nws = int(_jplib().fn["jp_conv2d_wgrad_ws_floats"](N, Cin, H, W, Cout, KH, stride, pad))
                    ws_w = _new((nws,), dy) if nws else None
                    # (the concatenated copy's magnitude = the largest of its sources': the library folds the slots)
                    call("jp_conv2d_wgrad_src3", xc, Cin, 0, None, 0, 0, None, 0, 0, dy, w.g, N, H, W, Cout, KH, stride, pad,
                         pad_mode, 1, ws_w, nws, None, None, None, dy_am, _amax_ws(dy.device, None))
                    del xc, ws_w
                else:
                    single = len(srcs) == 1 and not srcs[0][1]
                    if single:
                        nws = int(_jplib().fn["jp_conv2d_wgrad_ws_floats"](N, Cin, H, W, Cout, KH, stride, pad))
                    else:       # few-channel layer on an upsampled source (BEV decoder 16 -> 16): partial sums of the direct kernel
                        nws = int(_jplib().fn["jp_conv2d_wgrad_src3_ws_floats"](s3[1], s3[2], s3[4], s3[5], s3[7], s3[8], N, H, W,
                                                                                 Cout, KH, stride, pad, pad_mode))
                    ws_w = _new((nws,), dy) if nws else None
                    call("jp_conv2d_wgrad_src3", *s3, dy, w.g, N, H, W, Cout, KH, stride, pad, pad_mode, 1, ws_w, nws, *wg_am)
                    del ws_w

        # Parameter gradients are off the critical path (nothing in the backward chain reads them): they run on a
        # companion stream of the tape's stream, so the wgrad kernels fill the CUs that the dgrad chain's kernel tails
        # (last, partially filled round of workgroups) leave idle.  Joined in grad_ready / at the end of the backward.
        wgs = _wgrad_stream()
        if wgs is None:
            param_grads()
        elif w.rg or (b is not None and b.rg):
            base = torch.cuda.current_stream(dy.device)
            wgs.wait_stream(base)
            with torch.cuda.stream(wgs):
                param_grads()
            dy.record_stream(wgs)
            for v, _ in srcs:
                v.t.record_stream(wgs)
            for a in (dy_am, *x_am):                # the operands' largest magnitudes are read on that stream too
                if a is not None:
                    a.record_stream(wgs)
        if any(v.rg for v, _ in srcs):
            dg_am = (dy_am, _amax_ws(dy.device, dy_am))
            nwd = _ws_floats(Cin, Cout, KH, 1) if Cout >= 16 else 0
            if len(srcs) == 1 and srcs[0][1] == 0:
                g, acc = srcs[0][0].grad_buf()
                nsd = int(_jplib().fn["jp_conv2d_dgrad_split_floats"](N, Cin, H, W, Cout, KH, stride, pad))
                ws_s = _new((nsd,), dy) if nsd else None
                dx_slot, dx_done = _out_slot(dy.device, big), _ct.c_int(0)
                _conv_call("jp_conv2d_dgrad", w, "dgrad", sig, nwd,
                           (dy, w.t, g, N, Cin, H, W, Cout, KH, stride, pad, pad_mode, acc),
                           (ws_s, dg_am[0], dx_slot, _ct.addressof(dx_done), dg_am[1]))
                del ws_s
                if dx_done.value:         # max |dx| out of the dgrad epilogue (+ border fold): the scale of the next convolution backward
                    srcs[0][0].gamax = dx_slot
            elif int(_jplib().fn["jp_conv2d_dgrad_src3_ok"](s3[1], s3[2], s3[4], s3[5], s3[7], s3[8], N, H, W, Cout, KH,
                                                           stride, pad, pad_mode)):
                # per-source dgrad inside the library: straight into each source's gradient buffer, the upsampled
                # source at its own (half) resolution
                ga = []
                for i in range(3):
                    if i < len(srcs) and srcs[i][0].rg:
                        g, acc = srcs[i][0].grad_buf()
                        ga += [g, s3[3 * i + 1], s3[3 * i + 2], acc]
                    else:
                        ga += [None, s3[3 * i + 1], s3[3 * i + 2], 0]
                rgs = tuple(ga[0::4][i] is not None for i in range(3))
                nsd = int(_jplib().fn["jp_conv2d_dgrad_src3_split_floats"](*[v for i in range(3) for v in (s3[3 * i + 1], s3[3 * i + 2])],
                                                                          N, H, W))
                ws_s = _new((nsd,), dy) if nsd else None
                dx_slot, dx_done = _out_slot(dy.device, big and rgs[0]), _ct.c_int(0)
                _conv_call("jp_conv2d_dgrad_src3", w, "dgrad3", sig + rgs, nwd,
                           (dy, w.t, *ga, N, H, W, Cout, KH, stride, pad, pad_mode),
                           (ws_s, dg_am[0], dx_slot, _ct.addressof(dx_done), dg_am[1]))
                del ws_s
                if dx_done.value:         # (the first source's gradient as STORED -- also when the call accumulated into it)
                    srcs[0][0].gamax = dx_slot
            else:   # gradient w.r.t. the virtual concat, then routed to the sources
                dcat = _new((N, Cin, H, W), dy)
                # (split-K scratch as in the single-source call: small-grid layers then fold their K slices in a fixed order
                # instead of meeting in fp32 atomics)
                nsd = int(_jplib().fn["jp_conv2d_dgrad_split_floats"](N, Cin, H, W, Cout, KH, stride, pad))
                ws_s = _new((nsd,), dy) if nsd else None
                _conv_call("jp_conv2d_dgrad", w, "dgrad", sig, nwd,
                           (dy, w.t, dcat, N, Cin, H, W, Cout, KH, stride, pad, pad_mode, 0), (ws_s, dg_am[0], None, None, dg_am[1]))
                del ws_s
                c0 = 0
                for v, u in srcs:
                    C = v.t.shape[1]
                    if v.rg:
                        g, acc = v.grad_buf()
                        if u:
                            call("jp_upsample2x_bwd", dcat, g, N, C, H // 2, W // 2, Cin, c0, acc)
                        else:
                            call("jp_copy_channels", dcat, g, N, C, H * W, Cin, c0, C, 0, acc)
                    c0 += C
        out.g = None

    _rec(out.rg, bwd)
    return out


# ------------------------------------------------------------------------------------------- batch norm
def batchnorm_train(x: Var, gamma: Var, beta: Var, running_mean, running_var, residual: Var | None = None,
                    relu=False, momentum=0.1, eps=1e-5, n_updates=1, groups=1) -> Var:
    """Train-mode BatchNorm2d (+ residual, ReLU).  `groups` > 1: the batch holds `groups` independent forward passes of the
    same network stacked along N (the two pose pairs): statistics, running-stat updates (in order) and the backward
    reductions are taken per group -- exactly what separate calls would do -- while the convolutions around it run once."""
    N, C, H, W = x.t.shape
    assert N % groups == 0
    Ng = N // groups
    y = torch.empty_like(x.t)
    mean = _new((groups, C), x.t)
    invstd = _new((groups, C), x.t)
    nbw = int(_jplib().fn["jp_bn_ws_doubles"](Ng, C, H * W))
    y_am = _out_slot(y.device, C >= 32)       # max|y| out of the apply kernel: the next convolution's operand scale
    # statistics partials from the producing convolution's epilogue (one group only: a partial never straddles two images, but the
    # kernels deal their pixel tiles out in an order of their own)
    st_ws, st_parts = x.bnst if (x.bnst is not None and groups == 1) else (None, 0)
    for g in range(groups):
        sl = slice(g * Ng, (g + 1) * Ng)
        ws = _new((nbw,), x.t, torch.float64)
        call("jp_bn_train_fwd", x.t[sl], gamma.t, beta.t, residual.t[sl] if residual is not None else None, y[sl], running_mean,
             running_var, mean[g], invstd[g], ws, Ng, C, H * W, momentum, eps, int(relu), n_updates, y_am, st_ws, st_parts)
    out = Var(y, x.rg or gamma.rg or (residual is not None and residual.rg))
    out.amax = y_am

    def bwd():
        if out.g is None:
            return
        dx = torch.empty_like(x.t)
        need_res = residual is not None and residual.rg
        dres = torch.empty_like(x.t) if need_res else None
        dx_am = _out_slot(dx.device, C >= 32)     # max|dx| out of the apply kernel: the scale of the convolution backward it feeds
        for g in range(groups):
            sl = slice(g * Ng, (g + 1) * Ng)
            ws2 = _new((nbw,), x.t, torch.float64)
            # residual-free ReLU layers: the kernel recomputes the mask from x (fmaf(x, sc, sh) > 0, bit-identical to the
            # forward's) instead of reading y
            call("jp_bn_train_bwd", out.g[sl], x.t[sl], y[sl] if (relu and residual is not None) else None, gamma.t, beta.t,
                 mean[g], invstd[g], dx[sl], dres[sl] if need_res else None, gamma.g, beta.g, ws2, Ng, C, H * W, int(relu), 1,
                 dx_am)
        if x.rg:
            x.add_grad(dx, dx_am)
        if need_res:
            residual.add_grad(dres)
        out.g = None

    _rec(out.rg, bwd)
    return out


def bn_relu_maxpool_train(x: Var, gamma: Var, beta: Var, running_mean, running_var, momentum=0.1, eps=1e-5, n_updates=1,
                          groups=1) -> Var:
    """ResNet stem tail, train mode: MaxPool2d(3, 2, 1)(relu(BatchNorm2d(x))) without the normalised map (resnet.py:92-94) --
    forward pools straight off the convolution output, backward gathers the pooled gradient through the argmax bytes inside
    the BatchNorm reduce / apply kernels (csrc/bn.hip).  `groups` as in batchnorm_train.  H and W must be multiples of 4."""
    N, C, H, W = x.t.shape
    assert N % groups == 0 and H % 4 == 0 and W % 4 == 0
    Ng = N // groups
    y = _new((N, C, H // 2, W // 2), x.t)
    idx = _new((N, C, H // 2, W // 2), x.t, torch.uint8)
    mean, invstd = _new((groups, C), x.t), _new((groups, C), x.t)
    nbw = int(_jplib().fn["jp_bn_ws_doubles"](Ng, C, H * W))
    y_am = _out_slot(y.device, C >= 32)       # max of the pooled map out of the kernel: the first residual block's operand scale
    for g in range(groups):
        sl = slice(g * Ng, (g + 1) * Ng)
        ws = _new((nbw,), x.t, torch.float64)
        call("jp_bn_relu_pool_fwd", x.t[sl], gamma.t, beta.t, y[sl], idx[sl], running_mean, running_var, mean[g], invstd[g], ws,
             Ng, C, H, W, momentum, eps, n_updates, y_am)
    out = Var(y, x.rg or gamma.rg)
    out.amax = y_am

    def bwd():
        if out.g is None:
            return
        dx = torch.empty_like(x.t)
        nb2 = int(_jplib().fn["jp_bn_relu_pool_bwd_ws_doubles"](Ng, C, H, W))
        for g in range(groups):
            sl = slice(g * Ng, (g + 1) * Ng)
            ws2 = _new((nb2,), x.t, torch.float64)
            call("jp_bn_relu_pool_bwd", out.g[sl], idx[sl], x.t[sl], gamma.t, beta.t, mean[g], invstd[g], dx[sl], gamma.g, beta.g,
                 ws2, Ng, C, H, W, 1)
        if x.rg:
            x.add_grad(dx)
        out.g = None

    _rec(out.rg, bwd)
    return out


def split_rows(v: Var, n: int):
    """(k*n, ...) -> k Vars of n rows each (views of v's storage); their gradients are gathered back into v's."""
    k = v.t.shape[0] // n
    parts = [Var(v.t[i * n:(i + 1) * n], v.rg) for i in range(k)]

    def bwd():
        if all(p.g is None for p in parts):
            return
        g, acc = v.grad_buf()
        for i, p in enumerate(parts):
            dst = g[i * n:(i + 1) * n]
            if p.g is None:
                if not acc:
                    dst.zero_()
            elif acc:
                dst.add_(p.g)
            else:
                dst.copy_(p.g)
            p.g = None

    _rec(v.rg, bwd)
    return parts


# ------------------------------------------------------------------------------------------- pooling etc.
def current_tape():
    return _TAPE


def maxpool(x: Var, k, s, p, bwd_addend=None) -> Var:
    """nn.MaxPool2d.  `bwd_addend` (callable -> tensor | None, evaluated at backward time) is added to the input
    gradient inside the backward kernel: a residual branch's gradient without an accumulation pass."""
    N, C, H, W = x.t.shape
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    y = _new((N, C, OH, OW), x.t)
    idx = _new((N, C, OH, OW), x.t, torch.uint8)
    call("jp_maxpool_fwd", x.t, y, idx, N * C, H, W, k, s, p)
    out = Var(y, x.rg)
    out.amax = x.amax             # |max over a window| <= max|x|: an upper bound is all an operand scale needs

    def bwd():
        if out.g is None:
            return
        dx = torch.empty_like(x.t)
        # CRP chains (5x5 stride 1): the gradient feeds a 1x1 convolution's dgrad / wgrad, the kernel reports its magnitude
        dx_am = _out_slot(dx.device, k == 5 and s == 1 and C >= 32)
        call("jp_maxpool_bwd", out.g, idx, dx, bwd_addend() if bwd_addend is not None else None, N * C, H, W, k, s, p, dx_am)
        x.add_grad(dx, dx_am)
        out.g = None

    _rec(out.rg, bwd)
    return out


def upsample2x(x: Var) -> Var:
    N, C, H, W = x.t.shape
    y = _new((N, C, 2 * H, 2 * W), x.t)
    call("jp_upsample2x_fwd", x.t, y, N, C, H, W, C, 0)
    out = Var(y, x.rg)

    def bwd():
        if out.g is None:
            return
        g, acc = x.grad_buf()
        call("jp_upsample2x_bwd", out.g, g, N, C, H, W, C, 0, acc)
        out.g = None

    _rec(out.rg, bwd)
    return out


def cat_channels(vs) -> Var:
    N, _, H, W = vs[0].t.shape
    Ct = sum(v.t.shape[1] for v in vs)
    y = _new((N, Ct, H, W), vs[0].t)
    c0 = 0
    for v in vs:
        C = v.t.shape[1]
        call("jp_copy_channels", v.t, y, N, C, H * W, C, 0, Ct, c0, 0)
        c0 += C
    out = Var(y, any(v.rg for v in vs))

    def bwd():
        if out.g is None:
            return
        c0 = 0
        for v in vs:
            C = v.t.shape[1]
            if v.rg:
                g, acc = v.grad_buf()
                call("jp_copy_channels", out.g, g, N, C, H * W, Ct, c0, C, 0, acc)
            c0 += C
        out.g = None

    _rec(out.rg, bwd)
    return out


def add(a: Var, b: Var) -> Var:
    y = torch.empty_like(a.t)
    call("jp_axpby", a.t, b.t, y, y.numel(), 1.0, 1.0)
    out = Var(y, a.rg or b.rg)

    def bwd():
        if out.g is None:
            return
        g = out.g
        owned = False
        for v in (a, b):
            if not v.rg:
                continue
            if v.g is None:
                v.g = _copy(g) if owned else g   # the first taker owns the buffer
                v.gamax = out.gamax              # (same values: the producer's magnitude still holds)
                owned = True
            else:
                call("jp_axpby", v.g, g, v.g, g.numel(), 1.0, 1.0)
                v.gamax = None                   # accumulated into: a magnitude reported for the first addend no longer bounds it
        out.g = None

    _rec(out.rg, bwd)
    return out


def _copy(t):
    o = torch.empty_like(t)
    call("jp_axpby", t, None, o, t.numel(), 1.0, 0.0)
    return o


def affine(x: Var, scale, shift) -> Var:
    """y = x*scale + shift on an input that needs no gradient (image normalisation (x-0.45)/0.225)."""
    y = torch.empty_like(x.t)
    call("jp_affine", x.t, y, y.numel(), scale, shift)
    assert not x.rg
    return Var(y)


def mul_mask(x: Var, mask: torch.Tensor, scale: float) -> Var:
    """Dropout with an explicit keep-mask: y = x * mask * scale (depth_decoder.py:52-53)."""
    y = torch.empty_like(x.t)
    call("jp_mul", x.t, mask, y, y.numel(), scale)
    out = Var(y, x.rg)

    def bwd():
        if out.g is None:
            return
        d = torch.empty_like(x.t)
        call("jp_mul", out.g, mask, d, d.numel(), scale)
        x.add_grad(d)
        out.g = None

    _rec(out.rg, bwd)
    return out


def act(x: Var, kind) -> Var:
    y = torch.empty_like(x.t)
    call("jp_act_fwd", x.t, y, y.numel(), kind)
    out = Var(y, x.rg)
    if kind in (ACT_RELU, ACT_LEAKY):
        out.amax = x.amax         # |act(x)| <= |x|: still an upper bound

    def bwd():
        if out.g is None:
            return
        d = torch.empty_like(x.t)
        call("jp_act_bwd", out.g, y, d, d.numel(), kind, None)
        x.add_grad(d)
        out.g = None

    _rec(out.rg, bwd)
    return out


def bilinear_resize(x: Var, OH, OW) -> Var:
    N, C, H, W = x.t.shape
    y = _new((N, C, OH, OW), x.t)
    call("jp_bilinear_fwd", x.t, y, N * C, H, W, OH, OW)
    out = Var(y, x.rg)

    def bwd():
        if out.g is None:
            return
        g, acc = x.grad_buf()
        call("jp_bilinear_bwd", out.g, g, N * C, H, W, OH, OW, acc)
        out.g = None

    _rec(out.rg, bwd)
    return out


def area_downsample(x: torch.Tensor, f: int) -> torch.Tensor:
    N, C, H, W = x.shape
    if f == 1:
        return x
    y = _new((N, C, H // f, W // f), x)
    call("jp_area_downsample", x, y, N * C, H, W, f)
    return y


# ------------------------------------------------------------------------------------------- small dense
def gemm(A, B, C, M, N, K, lda, ldb, ldc, sA=0, sB=0, sC=0, batch=1, tA=0, tB=0, alpha=1.0, beta=0.0):
    call("jp_gemm_strided_batched", A, B, C, M, N, K, lda, ldb, ldc, sA, sB, sC, batch, tA, tB, alpha, beta)


def linear_act(x: Var, w: Var, b: Var, act_kind=ACT_RELU) -> Var:
    """y = act(x @ w.T + b) on the last dim; x (..., K) treated as (M, K) rows."""
    K = x.t.shape[-1]
    M = x.t.numel() // K
    Nf = w.t.shape[0]
    y = _new(tuple(x.t.shape[:-1]) + (Nf,), x.t)
    gemm(x.t, w.t, y, M, Nf, K, K, K, Nf, tB=1)
    call("jp_bias_act_rows", y, b.t, M, Nf, act_kind)
    out = Var(y, x.rg or w.rg)

    def bwd():
        if out.g is None:
            return
        d = out.g
        if act_kind != ACT_NONE:
            d2 = torch.empty_like(d)
            call("jp_act_bwd", d, y, d2, d.numel(), act_kind, None)
            d = d2
        if b.rg:
            call("jp_colsum", d, b.g, M, Nf, 1)
        if w.rg:   # dW (Nf,K) += d^T (Nf,M) @ x (M,K)
            gemm(d, x.t, w.g, Nf, K, M, Nf, K, K, tA=1, beta=1.0)
        if x.rg:   # dx (M,K) = d (M,Nf) @ W (Nf,K)
            g, acc = x.grad_buf()
            gemm(d, w.t, g, M, K, Nf, Nf, K, K, beta=float(acc))
        out.g = None

    _rec(out.rg, bwd)
    return out


def spatial_mean(x: Var, scale: float) -> Var:
    N, C, H, W = x.t.shape
    y = _new((N, C), x.t)
    call("jp_spatial_mean", x.t, y, N * C, H * W, scale)
    out = Var(y, x.rg)

    def bwd():
        if out.g is None:
            return
        d = torch.empty_like(x.t)
        call("jp_spatial_mean_bwd", out.g, d, N * C, H * W, scale)
        x.add_grad(d)
        out.g = None

    _rec(out.rg, bwd)
    return out


def batchnorm_eval(x: Var, gamma, beta, running_mean, running_var, residual=None, relu=False, eps=1e-5) -> Var:
    N, C, H, W = x.t.shape
    y = torch.empty_like(x.t)
    call("jp_bn_eval_fwd", x.t, gamma, beta, running_mean, running_var, residual.t if residual is not None else None,
         y, N, C, H * W, eps, int(relu))
    return Var(y)


def softmax2(x: torch.Tensor) -> torch.Tensor:
    N, C, H, W = x.shape
    assert C == 2
    y = torch.empty_like(x)
    call("jp_softmax_c2", x, y, N, H * W)
    return y


# ------------------------------------------------------------------------------------------- RNG
_RNG_STATE = {"seed": 0x5EED, "ctr": 0, "dev": None}
_RNG_A, _RNG_B, _M64 = 0x9E3779B97F4A7C15, 0xD1B54A32D192ED03, 0xFFFFFFFFFFFFFFFF


def manual_seed(seed: int):
    _RNG_STATE["seed"], _RNG_STATE["ctr"] = int(seed), 0


def _next_seed():
    _RNG_STATE["ctr"] += 1
    return (_RNG_STATE["seed"] * _RNG_A + _RNG_STATE["ctr"] * _RNG_B) & _M64


def _i64(u: int) -> int:
    """uint64 -> the int64 with the same bits (the ABI passes the *_dev offsets / bases as `long`)"""
    u &= _M64
    return u - (1 << 64) if u >> 63 else u


def rng_step_base() -> int:
    """seed of call k (1-based) of the step that starts now = rng_step_base() + k * B (mod 2^64): what a captured step keeps
    in device memory (`rng_capture`), so that a replay draws exactly what the eager step would have drawn."""
    return _i64(_RNG_STATE["seed"] * _RNG_A + _RNG_STATE["ctr"] * _RNG_B)


class rng_capture:
    """While active (a training step is being captured into a hipGraph), keep_mask / randn launch the *_dev generators: the
    step's seed base is read from `base` (a 1-element int64 device tensor the host refreshes before every replay), the call's
    position in the step is a constant of the captured launch.  `.calls` = generator calls per step afterwards."""

    def __init__(self, base: torch.Tensor):
        self.base, self.calls = base, 0

    def __enter__(self):
        _RNG_STATE["dev"] = self
        return self

    def __exit__(self, *a):
        _RNG_STATE["dev"] = None


def rng_advance(calls: int):
    """account for the generator calls of one replayed step"""
    _RNG_STATE["ctr"] += int(calls)


def keep_mask(shape, device, p_drop=0.5) -> torch.Tensor:
    """Bernoulli(1-p) keep-mask from the counter-based device RNG (train-mode nn.Dropout, depth_decoder.py:13)."""
    m = torch.empty(tuple(shape), device=device, dtype=torch.float32)
    cap = _RNG_STATE["dev"]
    if cap is not None:
        cap.calls += 1
        call("jp_rng_keep_mask_dev", m, m.numel(), cap.base, _i64(cap.calls * _RNG_B), float(p_drop))
    else:
        call("jp_rng_keep_mask", m, m.numel(), _next_seed(), float(p_drop))
    return m


def randn(shape, device) -> torch.Tensor:
    """Standard-normal noise from the device RNG (automask tie-breaking noise, net.py:163)."""
    m = torch.empty(tuple(shape), device=device, dtype=torch.float32)
    cap = _RNG_STATE["dev"]
    if cap is not None:
        cap.calls += 1
        call("jp_rng_normal_dev", m, m.numel(), cap.base, _i64(cap.calls * _RNG_B))
    else:
        call("jp_rng_normal", m, m.numel(), _next_seed())
    return m
