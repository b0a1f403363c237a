#!/usr/bin/env python
"""Critical path of one training step from a rocprofv3 (rocpd sqlite) kernel trace of the OVERLAPPED step.

The trace has no dependency edges, so they are inferred: walking back from the step's last kernel, the predecessor of a kernel
on the critical chain is the kernel whose END is the latest one at or before its START (+ a tolerance for the dispatch latency) --
on its own queue if that one ended within the tolerance (queue order), otherwise on any queue (an event join).  The time between
the predecessor's end and the kernel's start is the chain's dispatch gap.  Output: the chain's length by kernel name (what the step
is waiting for), the gaps, how much of the chain runs with nothing else in flight, and per kernel name the SLACK of the launches
that are off the chain (time they could grow before they would reach it is not computable without edges: reported is simply their
total duration off the chain).
usage: critical_path.py trace.db [step_index_from_end] [tolerance_us]"""
import re
import sqlite3
import sys

db = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
tol = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 12e3
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = next((q for q in ("queue_id", "stream_id", "queue") if q in cols), None)
rows = c.execute(f"select name, start, end, {qcol or '0'} from kernels order by start").fetchall()
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", n)[:100]
marks = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
if len(marks) < which + 1:
    print("not enough optimizer launches:", len(marks))
    sys.exit(0)
lo, hi = marks[-which - 1] + 1, marks[-which] + 1
step = rows[lo:hi]
t0, t1 = rows[lo - 1][2], step[-1][2]
print(f"# step window {(t1 - t0) / 1e6:.3f} ms, {len(step)} dispatches")

# kernels in flight at any time (for "alone" accounting)
ev = sorted([(max(s, t0), 1) for _, s, e, _ in step] + [(e, -1) for _, s, e, _ in step])
times, depth = [], []
d = 0
for t, k in ev:
    d += k
    times.append(t)
    depth.append(d)
import bisect


def alone_time(s, e):
    """time inside [s, e) with exactly one kernel in flight"""
    i = bisect.bisect_right(times, s) - 1
    tot, cur = 0, s
    while cur < e:
        nxt = times[i + 1] if i + 1 < len(times) else e
        nxt = min(nxt, e)
        if i >= 0 and depth[i] == 1:
            tot += nxt - cur
        cur = nxt
        i += 1
    return tot


by_end = sorted(range(len(step)), key=lambda i: step[i][2])
ends = [step[i][2] for i in by_end]
last_on_queue = {}
prev_same_queue = {}
for i in sorted(range(len(step)), key=lambda i: step[i][1]):
    q = step[i][3]
    prev_same_queue[i] = last_on_queue.get(q)
    last_on_queue[q] = i

cur = max(range(len(step)), key=lambda i: step[i][2])
chain, gaps = [], []
seen = set()
while cur is not None and cur not in seen:
    seen.add(cur)
    chain.append(cur)
    n, s, e, q = step[cur]
    if s <= t0:
        break
    p = prev_same_queue.get(cur)
    if p is not None and step[p][2] <= s and s - step[p][2] <= tol:
        nxt = p                                         # queue order explains the start
    else:
        # latest end at or before s on any queue
        j = bisect.bisect_right(ends, s) - 1
        nxt = by_end[j] if j >= 0 else None
        if nxt is not None and p is not None and step[p][2] > step[nxt][2]:
            nxt = p
    if nxt is None:
        gaps.append((s - t0, "(step start)", n))
        break
    gaps.append((s - step[nxt][2], step[nxt][0], n))
    cur = nxt

tot_k = sum(step[i][2] - max(step[i][1], t0) for i in chain)
tot_g = sum(max(0, g[0]) for g in gaps)
alone = sum(alone_time(max(step[i][1], t0), step[i][2]) for i in chain)
print(f"# inferred critical chain: {len(chain)} kernels, {tot_k / 1e6:.3f} ms of kernels + {tot_g / 1e6:.3f} ms of gaps "
      f"(window {(t1 - t0) / 1e6:.3f}); {alone / 1e6:.3f} ms of the chain's kernels run with nothing else in flight")
per = {}
for i in chain:
    n = short(step[i][0])
    a = per.setdefault(n, [0, 0, 0])
    a[0] += step[i][2] - max(step[i][1], t0)
    a[1] += 1
    a[2] += alone_time(max(step[i][1], t0), step[i][2])
print("# chain time by kernel (ms, launches on the chain, ms alone on the device):")
for n, (v, k, al) in sorted(per.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"  {v / 1e6:8.3f}  {k:4d}  {al / 1e6:8.3f}  {n}")
# the chain in time order, merged into runs on one queue
print("# chain in time order (runs of consecutive chain kernels on one queue): t_start ms, length ms, queue, kernels, first .. last kernel")
runs = []
for i in reversed(chain):
    n, s_, e, q = step[i]
    if runs and runs[-1][2] == q:
        runs[-1][1] = e
        runs[-1][3] += 1
        runs[-1][5] = n
    else:
        runs.append([max(s_, t0), e, q, 1, n, n])
for s_, e, q, k, a, b in runs:
    if e - s_ > 150e3 or k > 3:
        print(f"  {(s_ - t0) / 1e6:8.3f} {(e - s_) / 1e6:8.3f}  q{q} {k:4d}  {short(a)[:48]} .. {short(b)[:48]}")
qs = {}
for i in chain:
    qs[step[i][3]] = qs.get(step[i][3], 0) + step[i][2] - max(step[i][1], t0)
print("# chain time by queue:", {k: round(v / 1e6, 3) for k, v in sorted(qs.items(), key=lambda kv: -kv[1])})
gaps.sort(key=lambda g: -g[0])
print(f"# largest gaps on the chain (us; > 20 us total {sum(g[0] for g in gaps if g[0] > 20000) / 1e6:.3f} ms):")
for g, a, b in gaps[:12]:
    print(f"  {g / 1e3:7.1f}  after {short(a)[:70]}  before {short(b)[:70]}")
# off-chain work by kernel
off = {}
for i in range(len(step)):
    if i in seen:
        continue
    n = short(step[i][0])
    off[n] = off.get(n, 0) + step[i][2] - step[i][1]
print(f"# off the chain: {sum(off.values()) / 1e6:.3f} ms of kernels; top:")
for n, v in sorted(off.items(), key=lambda kv: -kv[1])[:20]:
    print(f"  {v / 1e6:8.3f}  {n}")
