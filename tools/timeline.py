#!/usr/bin/env python
"""GPU timeline of one training step from a rocprofv3 (rocpd sqlite) kernel trace: how busy the device is between two
optimizer launches, how much of the step has >= 2 kernels in flight (side stream / companion streams), the per-queue
sums, and the largest idle gaps with the kernels on either side.   usage: timeline.py trace.db [step_index_from_end]"""
import re
import sqlite3
import sys

db = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = next((q for q in ("queue_id", "stream_id", "queue") if q in cols), None)
rows = c.execute(f"select name, start, end, {qcol or '0'} from kernels order by start").fetchall()
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", n)[:90]
marks = [i for i, r in enumerate(rows) if r[0].startswith("adam_kernel") or "::adam_kernel" in r[0] or " adam_kernel" in r[0]]
if len(marks) < which + 1:
    print("columns:", cols, "\nnot enough optimizer launches:", len(marks))
    sys.exit(0)
lo, hi = marks[-which - 1] + 1, marks[-which] + 1
step = rows[lo:hi]
t0, t1 = rows[lo - 1][2], step[-1][2]
print(f"# step window {(t1 - t0) / 1e6:.3f} ms, {len(step)} dispatches, queue column: {qcol}")
ev = []
for n, s, e, q in step:
    ev.append((max(s, t0), 1))
    ev.append((e, -1))
ev.sort()
depth, last, hist = 0, t0, {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    last = t
    depth += d
tot = t1 - t0
for k in sorted(hist):
    print(f"  {k} kernels in flight: {hist[k] / 1e6:8.3f} ms  {100 * hist[k] / tot:5.1f} %")
per = {}
for n, s, e, q in step:
    per[q] = per.get(q, 0) + (e - s)
for q, v in sorted(per.items(), key=lambda kv: -kv[1]):
    print(f"  queue {q}: {v / 1e6:8.3f} ms of kernels")
# idle gaps (no kernel in flight)
gaps = []
cur_end, prev = t0, "(previous step)"
for n, s, e, q in sorted(step, key=lambda r: r[1]):
    if s > cur_end:
        gaps.append((s - cur_end, prev, n))
    if e > cur_end:
        cur_end, prev = e, n
gaps.sort(reverse=True)
print(f"# idle total {sum(g[0] for g in gaps) / 1e6:.3f} ms in {len(gaps)} gaps; > 20 us: "
      f"{sum(g[0] for g in gaps if g[0] > 20000) / 1e6:.3f} ms")
for g, a, b in gaps[:15]:
    print(f"  {g / 1e3:7.1f} us  after {short(a)}  before {short(b)}")

# ---- which kernels run ALONE (nothing else in flight): the step's un-overlapped, i.e. critical, time by kernel name
solo = {}
evs = sorted([(max(s_, t0), 1, i) for i, (n, s_, e, q) in enumerate(step)] + [(e, -1, i) for i, (n, s_, e, q) in enumerate(step)])
live, last = set(), t0
for t, d, i in evs:
    if len(live) == 1:
        n = short(step[next(iter(live))][0])
        solo[n] = solo.get(n, 0) + (t - last)
    last = t
    if d > 0:
        live.add(i)
    else:
        live.discard(i)
print(f"# time with exactly one kernel in flight, by kernel (total {sum(solo.values()) / 1e6:.3f} ms):")
for n, v in sorted(solo.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {v / 1e3:8.1f} us  {n}")
