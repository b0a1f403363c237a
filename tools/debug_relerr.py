"""GPU-box debugging aid: element-wise gradient error of the HIP step vs the oracle (forced selections) per parameter."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.golden_util import load_case, run_oracle
from tests.test_step_parity_gpu import build_model, gpu_inputs
from jperceiver_amd.apis import build_optimizer
from oracle import jp_oracle as J

case = sys.argv[1] if len(sys.argv) > 1 else "argo_both_1024_b1"
pat = sys.argv[2] if len(sys.argv) > 2 else "DepthDecoder"
g, meta = load_case(case)
ora = run_oracle(meta, backward=False)
label = J.scale_label_both(ora["opt"], ora["inp"])
model, opt = build_model(meta)
optim = build_optimizer(model, dict(type="Adam", lr=1e-4, weight_decay=0))
optim.zero_grad()
out, losses = model(gpu_inputs(meta, label))
losses.total().backward()
torch.cuda.synchronize()
force = {("min_index", s): out[("min_index", s)].cpu() for s in range(4)}
for tag in ("road", "car"):
    force["cv_argmax_" + tag] = out["cv_argmax_" + tag].cpu()
    force["cm_argmax_" + tag] = out["cm_argmax_" + tag].cpu()
ora2 = run_oracle(meta, force=force)
rows = []
for n, p in model.named_parameters():
    r = ora2["P"][n].grad
    if r is None or pat not in n:
        continue
    rows.append((float((p.grad.detach().cpu() - r).norm() / (r.norm() + 1e-30)), n, float(r.norm())))
for e, n, rn in sorted(rows, reverse=True)[:int(os.environ.get("TOPN", "24"))]:
    print("%.4f  %-60s |g|=%.3e" % (e, n, rn))
