"""Gradient exchange + optimizer hook with the reference's interface
(mono/core/utils/dist_utils.py:12-60): `allreduce_grads(model, coalesce, bucket_size_mb)` and
`DistOptimizerHook(grad_clip, coalesce, bucket_size_mb).after_train_iter(runner)`.

MI355X design.  Gradients already live in one flat arena whose live prefix is laid out in the order the
backward pass finishes them (runtime.SEGMENT_ORDER), so there is nothing to flatten or copy back and a
bucket is simply a slice:

  * OVERLAP.  The step's tape fires `ops.grad_ready(segment)` right after a segment's last backward kernel
    was enqueued.  The hook answers by launching that segment's SUM all-reduce (in <= bucket_size_mb pieces)
    with `async_op=True` from the stream the tape is replaying on: RCCL's own stream picks the bucket up as
    soon as those kernels finish and moves it over xGMI while the rest of the backward (the other branch,
    the encoders) is still computing.  This is what the reference gets from the DDP reducer
    (mono/apis/trainer.py:167); its second, redundant averaging pass is not reproduced.
  * TWO-PHASE CLIP + ADAM.  As each bucket lands, stage 1 of the deterministic global-norm reduction runs
    over it (`jp_grad_sumsq_partials`, overlapping the buckets still in flight); after the last one a single
    tiny kernel folds the partials (`jp_sum_doubles`) and the fused clip+Adam pass updates the whole prefix.
    The 1/world_size averaging rides along in that pass (grad_scale).  All ranks hold bit-identical reduced
    gradients and the norm reduction has a fixed order, so every rank derives the same clip coefficient and
    the replicas do not drift.
  * xGMI is point-to-point (7 links per GPU), so a few large messages beat many small ones: 5 segments of
    34-50 MB, split only above bucket_size_mb (default 64 MiB).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .. import ops
from ..runtime import FlatAdam


def _arena_of(model_or_opt):
    m = getattr(model_or_opt, "module", model_or_opt)
    a = getattr(m, "_jp_arena", None)
    if a is None:
        raise RuntimeError("no flat arena attached: build the optimizer with jperceiver_amd.apis.build_optimizer first")
    return a


def _world():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def _bucket_floats(bucket_size_mb):
    """bucket length in floats, a multiple of 64 (256 B): every bucket start stays aligned for the 16-byte norm kernel
    whatever (fractional) bucket_size_mb the config carries."""
    n = int((bucket_size_mb if bucket_size_mb and bucket_size_mb > 0 else 64) * 1024 * 1024 // 4)
    return max(64, (n + 63) // 64 * 64)


class _Exchange:
    """One step's worth of in-flight bucket all-reduces over the arena's live prefix."""

    def __init__(self, arena, bucket_size_mb=-1):
        self.arena, self.bucket = arena, _bucket_floats(bucket_size_mb)
        self.works = []                 # (work, offset, numel) in launch order
        self.done = set()
        self.events = None              # list to receive (before-waits, after-waits) event pairs (bench.py, N > 1)

    def launch(self, seg: str):
        """Start the all-reduce of one arena segment (idempotent).  Called from the tape (ops.grad_ready) with the
        replaying stream current: async_op=True makes the collective wait for exactly the kernels enqueued so far
        on that stream and run on the communicator's own stream from there."""
        if seg in self.done or seg not in self.arena.segments:
            return
        self.done.add(seg)
        off, n = self.arena.segments[seg]
        for o in range(off, off + n, self.bucket):
            k = min(self.bucket, off + n - o)
            w = dist.all_reduce(self.arena.grads[o:o + k], op=dist.ReduceOp.SUM, async_op=True)
            self.works.append((w, o, k))

    def finish(self, with_norm: bool):
        """Launch whatever the tape did not report, then wait bucket by bucket (in launch order) on the current stream
        and fold each landed bucket into the global-norm partials while later buckets are still in flight."""
        for seg in self.arena.segments:
            self.launch(seg)
        if self.events is not None and self.arena.grads.is_cuda:
            # everything the backward enqueued precedes e0; e1 follows the last bucket's arrival: e1 - e0 is the part of
            # the exchange the compute stream had to wait for (plus the interleaved norm partials, ~0.1 ms)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for w, o, k in self.works:
            w.wait()
            if with_norm:
                self.arena.add_norm_partial(o, k)
        if self.events is not None and self.arena.grads.is_cuda:
            e1.record()
            self.events.append((e0, e1))
        self.works = []


def allreduce_grads(model, coalesce=True, bucket_size_mb=-1, average_in_place=True):
    """All-reduce the gradient arena across ranks (no overlap: every bucket is launched here).
    bucket_size_mb <= 0 -> 64 MiB buckets."""
    world = _world()
    if world == 1:
        return
    arena = _arena_of(model)
    ex = _Exchange(arena, bucket_size_mb)
    ex.finish(with_norm=False)
    if average_in_place:
        from .._lib import call
        call("jp_axpby", arena.grads, None, arena.grads, arena.live_numel, 1.0 / world, 0.0)


class DistOptimizerHook(object):
    def __init__(self, grad_clip=None, coalesce=True, bucket_size_mb=-1, force_exchange=False):
        """`force_exchange` (not a reference option): run the bucketed all-reduce even in a single-rank process group, so
        that the RCCL path -- communicator stream, async work handles, stream waits -- can be exercised on one GPU
        (tests/test_distributed_gpu.py::test_rccl_single_rank_exchange)."""
        self.grad_clip = grad_clip
        self.coalesce = coalesce
        self.bucket_size_mb = bucket_size_mb
        self.force_exchange = force_exchange
        self.exposed_events = None      # set to [] to collect one (e0, e1) event pair per step (bench.py's allreduce_exposed_ms)

    def after_train_iter(self, runner):
        """zero_grad -> backward (+ overlapped all-reduce) -> global norm -> clip + Adam (dist_utils.py:54-60)."""
        opt = runner.optimizer
        opt.zero_grad()
        world = _world()
        if isinstance(opt, FlatAdam):
            max_norm = self.grad_clip.get("max_norm") if self.grad_clip else None
            ex = None
            if world > 1 or (self.force_exchange and dist.is_available() and dist.is_initialized()):
                ex = _Exchange(opt.arena, self.bucket_size_mb)
                ex.events = self.exposed_events
                prev = ops.set_grad_ready_hook(ex.launch)
                try:
                    runner.outputs["loss"].backward()
                finally:
                    ops.set_grad_ready_hook(prev)
                ex.finish(with_norm=bool(max_norm))
            else:
                runner.outputs["loss"].backward()
            # SUM all-reduce; the averaging rides along in the Adam pass (grad_scale)
            opt.grad_scale = 1.0 / world
            opt.max_norm = max_norm
            opt.step()
        else:  # foreign optimizer object: keep the reference's literal sequence
            runner.outputs["loss"].backward()
            allreduce_grads(runner.model, self.coalesce, self.bucket_size_mb)
            if self.grad_clip is not None:
                torch.nn.utils.clip_grad_norm_(runner.model.parameters(), **self.grad_clip)
            opt.step()
