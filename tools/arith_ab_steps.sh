#!/bin/bash
# the same optimizer steps under the default kernels and under the exact-fp32 MFMA kernels; prints per-step loss differences
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; O=$ROOT/gpurun_out/${1:-arith_ab}; mkdir -p $O; cd $ROOT
python tools/arith_ab_steps.py ${2:-256} ${3:-2} ${4:-8} 2>/dev/null | grep ARITH_AB | sed 's/^ARITH_AB //' > $O/split.json
JP_P9S=0 JP_W9S=0 JP_P9US=0 JP_P9SD=0 JP_P9S2=0 JP_P7S=0 python tools/arith_ab_steps.py ${2:-256} ${3:-2} ${4:-8} 2>/dev/null | grep ARITH_AB | sed 's/^ARITH_AB //' > $O/exact.json
python tools/arith_ab_steps.py ${2:-256} ${3:-2} ${4:-8} 2>/dev/null | grep ARITH_AB | sed 's/^ARITH_AB //' > $O/split_again.json
python - $O <<'PY'
import json, sys
O = sys.argv[1]
a, b, c = (json.load(open(f"{O}/{n}.json")) for n in ("split", "exact", "split_again"))
print(f"scheme {a['scheme']}: default kernels vs exact-fp32 MFMA kernels (and the default kernels run twice: the run-to-run spread of the atomically merged sums)")
for s, (x, y, z) in enumerate(zip(a["losses"], b["losses"], c["losses"])):
    k = "loss" if "loss" in x else sorted(x)[0]
    worst = max(abs(x[q] - y[q]) / max(abs(y[q]), 1e-6) for q in x)
    again = max(abs(x[q] - z[q]) / max(abs(x[q]), 1e-6) for q in x)
    print(f"  step {s}: {k} {x[k]:.6f} vs {y[k]:.6f}   worst relative difference over {len(x)} logged terms {worst:.2e}   (same kernels, second run: {again:.2e})")
pn = max(abs(a["params"][n] - b["params"][n]) / max(b["params"][n], 1e-12) for n in a["params"])
pa = max(abs(a["params"][n] - c["params"][n]) / max(a["params"][n], 1e-12) for n in a["params"])
print(f"  parameter norms after the last step: worst relative difference {pn:.2e} (second run of the same kernels: {pa:.2e})")
PY
