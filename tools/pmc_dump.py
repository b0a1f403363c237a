#!/usr/bin/env python
"""Print per-kernel averages of every counter in a rocprofv3 --pmc rocpd database (GPU-box aid)."""
import collections
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for k, n, v in rows:
    agg[re.sub(r"\(anonymous namespace\)::", "", k)[:110]][n].append(v)
for k, d in agg.items():
    print(k)
    for n, vs in sorted(d.items()):
        print("   %-28s n=%-3d avg=%.4g" % (n, len(vs), sum(vs) / len(vs)))
