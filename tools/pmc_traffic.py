#!/usr/bin/env python
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md §HBM /
§rocprofv3 PMC slots prescribe) of `JP_PMC_CALIB=1 python bench.py ...` into per-launch HBM traffic of the
dominant kernel.  The counters are calibrated on the known-byte streaming copy bench.py appends (axpby_kernel,
4 B/lane accesses like the igemm gather): gfx950's FETCH_SIZE under-reports wide coalesced reads, so
bytes = counter_KB * 1024 * (known copy bytes / counter_KB*1024 of the copy)."""
import json
import re
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, value, grid_size from counters_collection where counter_name=?", (counter,)).fetchall()
    out = {}
    for name, v, g in rows:
        out.setdefault(name, []).append((v, g))
    return out


def main(fetch_db, write_db, out_json):
    F, W = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    n = 1 << 28
    copy = [k for k in F if "axpby_kernel" in k][0]
    # the calibration launches are the three largest-grid axpby dispatches (1 GiB in, 1 GiB out each)
    cf = sorted(F[copy], key=lambda t: -t[1])[:3]
    cw = sorted(W[copy], key=lambda t: -t[1])[:3]
    known = 4.0 * n
    kf = known / (sum(v for v, _ in cf) / len(cf) * 1024.0)
    kw = known / (sum(v for v, _ in cw) / len(cw) * 1024.0)
    # the kernel bench.py reported as by-time dominant (bench_families.json -> roofline.kernel), matched on its
    # template-argument text; falls back to the dispatch-count-weighted largest igemm kernel
    want = None
    if len(sys.argv) > 4:
        want = json.load(open(sys.argv[4]))["roofline"]["kernel"].split(" \u2014 ")[0].split(" — ")[0]
    norm = lambda t: re.sub(r"^void", "", re.sub(r"\(anonymous namespace\)::", "", t).replace(" ", ""))
    # (default template arguments are printed by rocprofv3 but not by bench.py: match without the closing '>')
    dom = [k for k in F if want and norm(k).startswith(norm(want).rstrip('>'))]
    if not dom:
        ig = [k for k in F if "jp_igemm" in k or "jp_wgrad" in k]
        dom = [max(ig, key=lambda k: sum(v for v, _ in F[k]))]
    assert len(dom) == 1, dom
    fv = [v for v, _ in F[dom[0]]]
    wv = [v for v, _ in W[dom[0]]]
    res = {
        "kernel": re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", dom[0]).split("(")[0])[:160],
        "launches_profiled": len(fv),
        "fetch_KB_raw_avg": sum(fv) / len(fv), "write_KB_raw_avg": sum(wv) / len(wv),
        "calibration": {"copy_bytes_each_way": known, "fetch_factor": kf, "write_factor": kw,
                        "copy_fetch_KB_raw": [v for v, _ in cf], "copy_write_KB_raw": [v for v, _ in cw]},
    }
    res["traffic_bytes_per_launch"] = (res["fetch_KB_raw_avg"] * kf + res["write_KB_raw_avg"] * kw) * 1024.0
    json.dump(res, open(out_json, "w"), indent=1)
    print(json.dumps(res, indent=1))
    # per-kernel table of the whole profiled run (steps = argv[5], default 3 = 2 timed + 1 warm-up): calibrated FETCH + WRITE
    # bytes per step, to be read next to bench_families.json -> hbm_entry_points (algorithmic bytes of the same step)
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    short = lambda k: re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", k).split("(")[0])[:110]
    rows = []
    for k in F:
        if "axpby_kernel" in k and k == copy:
            fb = (sum(v for v, _ in F[k]) - sum(v for v, _ in cf)) * 1024.0 * kf
            wb = (sum(v for v, _ in W.get(k, [])) - sum(v for v, _ in cw)) * 1024.0 * kw
        else:
            fb = sum(v for v, _ in F[k]) * 1024.0 * kf
            wb = sum(v for v, _ in W.get(k, [])) * 1024.0 * kw
        rows.append((short(k), len(F[k]) / steps, fb / steps, wb / steps))
    rows.sort(key=lambda r: -(r[2] + r[3]))
    md = out_json.replace(".json", "_kernels.md")
    with open(md, "w") as f:
        f.write("# HBM traffic per kernel and step: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), calibrated on a 1 GiB copy\n")
        f.write(f"# fetch factor {kf:.4f}, write factor {kw:.4f}; {steps} profiled steps; total "
                f"{sum(r[2] + r[3] for r in rows) / 1e9:.2f} GB per step\n")
        f.write("| kernel | launches/step | fetch GB/step | write GB/step | total GB/step |\n|---|---|---|---|---|\n")
        for n, l, fb, wb in rows[:60]:
            f.write(f"| `{n}` | {l:.1f} | {fb / 1e9:.3f} | {wb / 1e9:.3f} | {(fb + wb) / 1e9:.3f} |\n")


if __name__ == "__main__":
    main(*sys.argv[1:4])
