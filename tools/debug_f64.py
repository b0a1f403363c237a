"""GPU-box aid: is the HIP step as accurate as the fp32 CPU oracle?  Both are compared with the oracle run in
float64 (same forced discrete selections)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.golden_util import load_case, case_inputs, oracle_opt, run_oracle
from tests.test_step_parity_gpu import build_model, gpu_inputs
from jperceiver_amd.apis import build_optimizer
from jperceiver_amd import synthetic as syn
from oracle import jp_oracle as J

case = sys.argv[1] if len(sys.argv) > 1 else "argo_both_512_b2"
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
o32 = run_oracle(meta, force=force)
# float64 oracle
shapes = J.state_shapes(meta["occ"])
tmpl = {n: torch.empty(s, dtype=torch.long if n.endswith("num_batches_tracked") else torch.float32) for n, s in shapes.items()}
state = syn.synth_state_dict(tmpl, seed=0)
P, Bf = {}, {}
for n in shapes:
    t = state[n].clone()
    if J.is_buffer(n):
        Bf[n] = t.double() if t.dtype == torch.float32 else t
    else:
        P[n] = t.double().requires_grad_(True)
inp, masks, noise = case_inputs(meta)
inp64 = {k: v.double() for k, v in inp.items()}
torch.set_default_dtype(torch.float64)
o, L = J.forward(P, Bf, o32["opt"], inp64, True, tuple(m.double() for m in masks), [[n.double() for n in per] for per in noise],
                 label.double(), force)
J.total_loss(L).backward()
torch.set_default_dtype(torch.float32)
hp = dict(model.named_parameters())
worst = []
for n in P:
    if P[n].grad is None:
        continue
    r = P[n].grad
    rn = float(r.norm())
    eh = float((hp[n].grad.detach().cpu().double() - r).norm()) / (rn + 1e-30)
    ec = float((o32["P"][n].grad.double() - r).norm()) / (rn + 1e-30)
    worst.append((eh, ec, n))
worst.sort(reverse=True)
print("rel. gradient error vs float64 oracle:  HIP   |  fp32 CPU oracle")
for eh, ec, n in worst[:25]:
    print(f"  {eh:9.2e} | {ec:9.2e}  {n}")
pat = os.environ.get("PAT")
if pat:
    print("-- parameters matching", pat)
    for eh, ec, n in worst:
        if pat in n:
            print(f"  {eh:9.2e} | {ec:9.2e}  {n}")
import statistics
print("median HIP %.2e  median CPU32 %.2e" % (statistics.median(w[0] for w in worst), statistics.median(w[1] for w in worst)))
for k in L:
    print(k, float(losses[k]), float(L[k]), float(o32["L"][k]))
