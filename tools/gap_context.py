#!/usr/bin/env python
"""Context of the largest device-idle gaps of one training step (rocprofv3 rocpd sqlite kernel trace): the kernels around
each gap with their queue and times.  usage: gap_context.py trace.db [n_gaps] [step_index_from_end]"""
import re
import sqlite3
import sys

db = sys.argv[1]
ngaps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
which = int(sys.argv[3]) if len(sys.argv) > 3 else 2
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = next((q for q in ("queue_id", "stream_id", "queue") if q in cols), "0")
rows = c.execute(f"select name, start, end, {qcol} from kernels order by start").fetchall()
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "")[:70]
marks = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
lo, hi = marks[-which - 1] + 1, marks[-which] + 1
step = rows[lo:hi]
t0 = rows[lo - 1][2]
# device-idle gaps: sweep
ev = sorted([(s, 1, i) for i, (n, s, e, q) in enumerate(step)] + [(e, -1, i) for i, (n, s, e, q) in enumerate(step)])
depth, gaps, last_end = 0, [], t0
for t, d, i in ev:
    if d == 1 and depth == 0 and t > last_end:
        gaps.append((t - last_end, last_end, t))
    depth += d
    if depth == 0:
        last_end = t
gaps.sort(reverse=True)
for g, a, b in gaps[:ngaps]:
    print(f"=== gap {g / 1e3:.1f} us at +{(a - t0) / 1e6:.3f} ms of the step")
    near = [(n, s, e, q) for n, s, e, q in step if e > a - 400e3 and s < b + 400e3]
    for n, s, e, q in near[-30:] if len(near) > 60 else near:
        side = "<" if e <= a else (">" if s >= b else "*")
        print(f"   {side} q{q}  +{(s - t0) / 1e6:8.3f} .. +{(e - t0) / 1e6:8.3f} ms  {short(n)}")
