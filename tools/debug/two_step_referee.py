"""Per top-level module: how far are the device's gradients from the float64 oracle at step 1 and at step 2 (and the fp32 CPU oracle's)?
    python tools/debug/two_step_referee.py [case=argo_both_512_b2] [seed offset of step 2 = 100]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import test_multi_step_parity_gpu as T
from jperceiver_amd import synthetic as syn
from jperceiver_amd.model import MONO
from jperceiver_amd.apis import build_optimizer
from oracle import jp_oracle as J
name = sys.argv[1] if len(sys.argv) > 1 else "argo_both_512_b2"
off = int(sys.argv[2]) if len(sys.argv) > 2 else 100
c = T.CASES[name]
opt = T._opt(c)
model = MONO.module_dict["Baseline"](opt)
state = syn.synth_state_dict(model.state_dict(), seed=0)
model.load_state_dict(state, strict=True)
model = model.cuda().train()
optim = build_optimizer(model, dict(type="Adam", lr=1e-4, weight_decay=0))
optim.max_norm, optim.grad_scale = 35.0, 1.0
P, Bf = J.make_params(J.state_shapes(c["HW"] // 4), state)
adam = {}
named = dict(model.named_parameters())
for step in (1, 2):
    inp, masks, noise = T._batch(c, c["seed"] + off * (step - 1))
    label = T._label(c, opt, inp)
    optim.zero_grad()
    out, losses = model({k: v.cuda() for k, v in T._device_batch(inp, masks, noise, label).items()})
    losses.total().backward()
    torch.cuda.synchronize()
    force = {("min_index", s): out[("min_index", s)].cpu() for s in range(4)}
    for tag in ("road", "car"):
        force["cv_argmax_" + tag] = out["cv_argmax_" + tag].cpu()
        force["cm_argmax_" + tag] = out["cm_argmax_" + tag].cpu()
    for p in P.values():
        p.grad = None
    P0 = {n: p.detach().clone() for n, p in P.items()}
    B0 = {n: b.clone() for n, b in Bf.items()}
    o2, L2 = J.forward(P, Bf, opt, inp, True, masks, noise, label, force)
    J.total_loss(L2).backward()
    g64 = T._f64_grads(c, opt, P0, B0, inp, masks, noise, label, force)
    per = collections.defaultdict(list)
    for n, p in named.items():
        if n not in g64 or P[n].grad is None:
            continue
        r64 = g64[n]
        eh = float((p.grad.detach().cpu().double() - r64).norm() / (r64.norm() + 1e-30))
        ec = float((P[n].grad.double() - r64).norm() / (r64.norm() + 1e-30))
        per[".".join(n.split(".")[:2])].append((eh, ec, n))
    print(f"== {name} step {step}: relative distance to the float64 oracle, device | fp32 CPU oracle (median, max over the group's tensors)")
    for gname, v in per.items():
        eh = sorted(x[0] for x in v); ec = sorted(x[1] for x in v)
        print(f"  {gname:42s} {len(v):3d}  hip {eh[len(eh)//2]:.2e} / {eh[-1]:.2e}   cpu32 {ec[len(ec)//2]:.2e} / {ec[-1]:.2e}")
    T._feed_device_grads(model, P)
    J.adam_step(P, adam, lr=1e-4, max_norm=35.0)
    optim.step()
    if os.environ.get("SYNC_PARAMS", "1") != "0":     # continue from the DEVICE's parameters (an ulp of difference moves the photometric gradients by percent)
        with torch.no_grad():
            for n, p in named.items():
                if n in P:
                    P[n].copy_(p.detach().cpu())
