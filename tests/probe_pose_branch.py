#!/usr/bin/env python
"""Diagnostic (not collected by pytest): the pose branch's gradients of tests/test_multi_step_parity_gpu.py's cfg1 step 1, evaluated
several ways in one process -- as the test does, with a host sync between forward and backward, without the side stream, without the
weight-gradient companion streams -- next to the fp32 CPU oracle's.  Prints gradient norms per variant, plus the pose chain's
intermediates (dP, d axisangle, d translation) as the backward produced them.
usage (GPU box): python tests/probe_pose_branch.py [--oracle]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_multi_step_parity_gpu as T                              # noqa: E402
from jperceiver_amd import ops, ops_loss                                       # noqa: E402
from jperceiver_amd.model import net as netmod                                 # noqa: E402

c = T.CASES["cfg1_full_B8_1024"]
opt = T._opt(c)
model = T.MONO.module_dict["Baseline"](opt)
state = T.syn.synth_state_dict(model.state_dict(), seed=0)
model.load_state_dict(state, strict=True)
model = model.cuda().train()
optim = T.build_optimizer(model, dict(type="Adam", lr=1e-4, weight_decay=0))
inp, masks, noise = T._batch(c, c["seed"])
label = T._label(c, opt, inp)
named = dict(model.named_parameters())
WATCH = ["DepthEncoder.encoder.conv1.weight", "DepthDecoder.disp1.conv.weight", "PoseEncoder.encoder.conv1.weight",
         "PoseEncoder.encoder.layer4.1.conv2.weight", "PoseDecoder.reduce.weight", "PoseDecoder.conv3.weight", "PoseDecoder.conv3.bias"]
WATCH = [n for n in WATCH if n in named] or list(named)[:4]
snaps = []
orig_call = ops_loss.call


def spy(name, *a):
    if name == "jp_pose_fwd":
        r = orig_call(name, *a)
        snaps.append(("fwd:aa", a[0].detach().clone()))
        snaps.append(("fwd:tr", a[1].detach().clone()))
        snaps.append((f"fwd:K@{a[2].data_ptr():x}", a[2].detach().clone()))
        return r
    if name == "jp_pose_bwd":
        snaps.append(("dP-before", a[0].detach().clone()))
        snaps.append(("aa", a[1].detach().clone()))
        snaps.append(("tr", a[2].detach().clone()))
        snaps.append((f"K@{a[3].data_ptr():x}", a[3].detach().clone()))
    r = orig_call(name, *a)
    if name == "jp_pose_bwd":
        snaps.append(("dP", a[0].detach().clone()))
        snaps.append(("daa", a[4].detach().clone()))
        snaps.append(("dtr", a[5].detach().clone()))
    return r


ops_loss.call = spy


def run(tag, sync_mid=False, with_label=True):
    snaps.clear()
    optim.zero_grad()
    d = T._device_batch(inp, masks, noise, label)
    if not with_label:
        d.pop(("scale_label", 0, 0))
    out, losses = model({k: v.cuda() for k, v in d.items()})
    total = losses.total()
    if sync_mid:
        torch.cuda.synchronize()
    total.backward()
    torch.cuda.synchronize()
    g = {n: float(named[n].grad.double().norm()) for n in WATCH}
    s = " ".join(f"{k}={float(v.double().norm()):.6e}" for k, v in snaps)
    print(f"[{tag}] total {float(total):.6f}  " + "  ".join(f"{n.split('.')[0][:5]}..{'.'.join(n.split('.')[-2:])} {v:.4e}" for n, v in g.items()))
    print(f"[{tag}]   pose chain: {s}", flush=True)
    if os.environ.get("PROBE_SAVE"):
        os.makedirs(os.environ["PROBE_SAVE"], exist_ok=True)
        torch.save([(k, v.cpu()) for k, v in snaps], os.path.join(os.environ["PROBE_SAVE"], tag.replace(" ", "_").replace("#", "") + ".pt"))
    return {n: named[n].grad.detach().clone() for n in named}


ref = run("as-test #1")
run("as-test #2")
run("as-test #3")
if "--quick" in sys.argv:
    run("as-test #4")
    sys.exit(0)
run("sync between fwd and bwd", sync_mid=True)
run("label computed by the model (host sync mid-forward)", with_label=False)
netmod._POSE_STREAM = False
a = run("no side stream")
netmod._POSE_STREAM = True
ops._WG_ON = False
run("no weight-gradient companion streams")
ops._WG_ON = True
b = run("as-test #4")
worst = max(((float((a[n] - b[n]).norm() / (a[n].norm() + 1e-30)), n) for n in named if a[n] is not None), default=None)
print("largest relative difference, no-side-stream vs as-test #4 (BatchNorm buffers moved in between):", worst)
if "--oracle" in sys.argv:
    J = T.J
    shapes = J.state_shapes(c["HW"] // 4)
    P, Bf = J.make_params(shapes, state)
    optim.zero_grad()
    out, losses = model({k: v.cuda() for k, v in T._device_batch(inp, masks, noise, label).items()})
    losses.total().backward()
    torch.cuda.synchronize()
    force = {("min_index", s): out[("min_index", s)].cpu() for s in range(4)}
    for tag in ("road", "car"):
        force["cv_argmax_" + tag] = out["cv_argmax_" + tag].cpu()
        force["cm_argmax_" + tag] = out["cm_argmax_" + tag].cpu()
    # (the model's BatchNorm buffers have moved: only the norms' order of magnitude is comparable)
    o2, L2 = J.forward(P, Bf, opt, inp, True, masks, noise, label, force)
    J.total_loss(L2).backward()
    print("[fp32 CPU oracle, initial buffers] " + "  ".join(f"{n.split('.')[0][:5]}..{'.'.join(n.split('.')[-2:])} {float(P[n].grad.double().norm()):.4e}"
                                                            for n in WATCH if n in P and P[n].grad is not None))
