#!/usr/bin/env python
"""Where the UN-TRACED overlapped step spends its time, per stream (round 6).

rocprofv3's kernel trace slows the host enough (64 -> 80 ms per step) that the traced step becomes host-bound in places where the
real one is not, so its gaps and its "critical chain" partly describe the tracer.  This tool times the real step with a handful of
HIP events instead: one after each network's forward (wrapping the modules' `_fwd`), one after each network's backward (the tape's
`grad_ready` markers, which are replayed right after a module's backward -- re-pointed here to record an event instead of starting an
all-reduce), on whichever stream the work runs, plus step start / end.  Output: per milestone the time since the step's start and the
stream it was recorded on, averaged over `--steps` steps -- i.e. which stream finishes last, and how long the main stream sits in its
joins.   usage (GPU box): python tools/stream_milestones.py [--steps 5] [--batch 8] [--hw 1024]"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                   # noqa: E402
from jperceiver_amd import ops                                                 # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--hw", type=int, default=1024)
args = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS[1]
B, HW, frames = args.batch, args.hw, cfg["frames"]
optd = bench.make_opt(B, HW, HW, frames, cfg["type"], cfg["split"], loss_sum=cfg["loss_sum"])
runner, batch = bench.build_runner(optd, dev, 1, 0, dict(B=B, height=HW, width=HW, frame_ids=frames, occ=HW // 4,
                                                         full_hw=cfg["full_hw"], split=cfg["split"], seed=1))
model = runner.model
marks = []          # (name, event, stream handle) of the current step
ON = [False]


def mark(name):
    if ON[0]:
        st = torch.cuda.current_stream(dev)
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(st)
        marks.append((name, ev, st.cuda_stream))


def wrap_fwd(mod, name):
    orig = mod._fwd

    def f(*a, **k):
        r = orig(*a, **k)
        mark("fwd " + name)
        return r
    mod._fwd = f


for n in ("DepthEncoder", "DepthDecoder", "PoseEncoder", "PoseDecoder", "LayoutEncoder"):
    if hasattr(model, n) and hasattr(getattr(model, n), "_fwd"):
        wrap_fwd(getattr(model, n), n)
orig_head = model._layout_head


def head(sfx, *a, **k):
    r = orig_head(sfx, *a, **k)
    mark("fwd head" + (sfx or "S"))
    return r


model._layout_head = head
ops.grad_ready = lambda tag: ops._rec(True, lambda: mark("bwd " + tag))      # no join, no all-reduce: just a timestamp

for _ in range(3):
    runner.train_iter(batch)
torch.cuda.synchronize()
acc = collections.OrderedDict()
streams = {}
tot = 0.0
for i in range(args.steps):
    marks.clear()
    ON[0] = True
    main = torch.cuda.current_stream(dev)
    mark("step start")
    runner.train_iter(batch)
    mark("step end (optimizer enqueued, main stream)")
    ON[0] = False
    torch.cuda.synchronize()
    t0 = marks[0][1]
    tot += t0.elapsed_time(marks[-1][1])
    for name, ev, st in marks[1:]:
        key = (name, "main" if st == main.cuda_stream else f"side {streams.setdefault(st, len(streams) + 1)}")
        acc.setdefault(key, []).append(t0.elapsed_time(ev))
print(f"# {args.steps} steps, B = {B}, {HW}^2: step {tot / args.steps:.2f} ms (events on the issuing streams; no tracer attached)")
print("| ms since step start | stream | milestone |\n|---|---|---|")
for (name, st), v in sorted(acc.items(), key=lambda kv: sum(kv[1]) / len(kv[1])):
    print(f"| {sum(v) / len(v):7.2f} | {st} | {name} ({len(v) // args.steps}x per step) |")
