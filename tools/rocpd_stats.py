#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel stats table (the --stats view)."""
import re
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
                 "max(accum_vgpr_count), max(lds_size) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# rocprofv3 --kernel-trace --stats summary of {db}")
print(f"# total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds B |")
print("|---|---|---|---|---|---|---|---|---|---|")
for n, cnt, s, a, mn, mx, vg, ag, lds in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = n if len(n) < 110 else n[:107] + "..."
    print(f"| `{n}` | {cnt} | {s / 1e6:.3f} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * s / tot:.2f} | {vg} | {ag} | {lds} |")
