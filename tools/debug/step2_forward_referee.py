"""Is the device's FORWARD further from float64 than the fp32 CPU oracle's at the second step?  (tools/debug/two_step_referee.py shows
the device's step-2 gradients 5-13 % from float64 on the pose networks / coarse decoder levels where the CPU's are 0.3-2 %; the
photometric terms' gradients are very sensitive to where the samples land, so a forward that deviates more would explain it.)
Prints, per step: relative distance to the float64 forward of the disparity maps, axis-angles, translations and loss terms --
device | fp32 CPU oracle.     python tools/debug/step2_forward_referee.py [case=argo_both_512_b2] [seed offset of step 2 = 100]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import test_multi_step_parity_gpu as T
from jperceiver_amd import synthetic as syn
from jperceiver_amd.model import MONO
from jperceiver_amd.apis import build_optimizer
from oracle import jp_oracle as J
name = sys.argv[1] if len(sys.argv) > 1 else "argo_both_512_b2"
off = int(sys.argv[2]) if len(sys.argv) > 2 else 100          # seed offset of step 2's batch (0: the same batch again)
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


def f64_forward(P32, Bf32, inp, masks, noise, label, force):
    P64 = {n: p.detach().double() for n, p in P32.items()}
    B64 = {n: (b.double() if b.dtype == torch.float32 else b.clone()) for n, b in Bf32.items()}
    inp64 = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in inp.items()}
    torch.set_default_dtype(torch.float64)
    try:
        with torch.no_grad():
            return J.forward(P64, B64, opt, inp64, True, tuple(m.double() for m in masks), [[z.double() for z in per] for per in noise],
                             label.double(), force)
    finally:
        torch.set_default_dtype(torch.float32)


def rel(a, r):
    a, r = a.detach().double().cpu().flatten(), r.detach().double().cpu().flatten()
    return float((a - r).norm() / (r.norm() + 1e-300))


for step in (1, 2):
    inp, masks, noise = T._batch(c, c["seed"] + off * (step - 1))
    label = T._label(c, opt, inp)
    optim.zero_grad()
    batch = {k: v.cuda() for k, v in T._device_batch(inp, masks, noise, label).items()}
    out, losses = model(batch)
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
    o64, L64 = f64_forward(P0, B0, inp, masks, noise, label, force)
    # the same float64 forward from the DEVICE's parameters (do the two trajectories' parameters differ enough to matter?)
    sd = {n: v.detach().cpu() for n, v in model.state_dict().items()}
    Pd = {n: sd[n].clone() for n in P0}
    o64d, _ = f64_forward(Pd, B0, inp, masks, noise, label, force)
    dif = sorted(((float((Pd[n] - P0[n]).abs().max()), float((Pd[n] - P0[n]).norm() / (P0[n].norm() + 1e-30)), n) for n in P0), reverse=True)
    print(f"== {name} step {step}: device parameters vs the oracle's: largest |diff| {dif[0][0]:.3e} ({dif[0][2]}), largest relative "
          f"tensor distance {max(d[1] for d in dif):.3e}; disp scale 0 of the float64 forward from the DEVICE's parameters: device "
          f"{rel(out[('disp', 0, 0)], o64d[('disp', 0, 0)]):.2e} from it, the oracle-parameter float64 forward {rel(o64[('disp', 0, 0)], o64d[('disp', 0, 0)]):.2e} from it")
    for d in dif[:6]:
        print(f"     |diff| {d[0]:.3e}  rel {d[1]:.3e}  {d[2]}")
    o32, L32 = J.forward(P, Bf, opt, inp, True, masks, noise, label, force)
    J.total_loss(L32).backward()
    print(f"== {name} step {step}: relative distance of the FORWARD to float64, device | fp32 CPU oracle")
    for s in range(4):
        k = ("disp", 0, s)
        print(f"  disp scale {s:<28d} hip {rel(out[k], o64[k]):.2e}   cpu32 {rel(o32[k], o64[k]):.2e}")
    for f in c["FR"][1:]:
        for key in ("axisangle", "translation", "cam_T_cam"):
            k = (key, 0, f)
            if k in o64 and k in out and k in o32:
                print(f"  {key + ' frame ' + str(f):<38s} hip {rel(out[k], o64[k]):.2e}   cpu32 {rel(o32[k], o64[k]):.2e}")
    for k in L64:
        a, b, r = float(losses[k]), float(L32[k]), float(L64[k])
        print(f"  loss {str(k):<33s} hip {abs(a - r) / (abs(r) + 1e-30):.2e}   cpu32 {abs(b - r) / (abs(r) + 1e-30):.2e}   (value {r:.6g})")
    T._feed_device_grads(model, P)
    J.adam_step(P, adam, lr=1e-4, max_norm=35.0)
    optim.step()
    del batch
