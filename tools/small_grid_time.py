#!/usr/bin/env python
"""GPU-box aid: how much kernel time sits in launches that cannot fill the chip (rocprofv3 rocpd database)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, duration, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z from kernels").fetchall()
tot = sum(r[1] for r in rows)
def wgs(r):
    return (r[2] // max(1, r[5])) * (r[3] // max(1, r[6])) * (r[4] // max(1, r[7]))
for lim in (64, 256, 512, 768):
    t = sum(r[1] for r in rows if wgs(r) < lim)
    n = sum(1 for r in rows if wgs(r) < lim)
    print("workgroups < %4d: %8.2f ms of %8.2f ms (%.1f%%), %d of %d launches" % (lim, t / 1e6, tot / 1e6, 100 * t / tot, n, len(rows)))
