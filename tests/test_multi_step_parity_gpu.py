"""Parity BEYOND the first step: everything stateful on the step path -- persistent weight packs and their
one-launch replay, the weight-scale headers of the fp16 split packs, the operand-magnitude slots, BatchNorm counters, Adam's step
count and bias corrections, the captured hipGraph's baked-in pointers -- is right on first use by construction and can only be
wrong from the second step on (round 5's `jp_pack_replay` header bug was exactly that, and every step-parity test ran ONE step).

  * test_second_step_matches_oracle: step 1 (forward, backward, clip + Adam) then step 2 on a NEW batch, through `Baseline` +
    `FlatAdam` directly; step 2's losses, maps, poses and element-wise gradients against the oracle evaluated at the parameters /
    BatchNorm buffers the oracle reached after ITS step 1 (its Adam fed the device gradients, so both sides start step 2 from the
    same point to 1e-6), then the second optimizer update -- the bar of tests/test_step_parity_gpu.py, one step later.  Cases: 512^2
    B = 2 Argo_both and the benchmark's own step (cfg1_full_B8_1024).
  * test_trajectory_through_runner[eager|captured]: five iterations through `Runner.train_iter` (batch_processor +
    DistOptimizerHook; `captured`: two eager iterations, the capture, two replays of the hipGraph) on five different batches, the
    oracle walking alongside: every iteration's loss terms within 2e-3, every parameter's gradient norm within 5 %, parameters
    within 1e-6 of the oracle's Adam (fed the device gradients) after every iteration.
Reference: mono/apis/trainer.py:30-56, mono/core/utils/dist_utils.py:54-60 (the loop these iterations replace)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from jperceiver_amd import synthetic as syn                                    # noqa: E402
from jperceiver_amd.model import MONO                                          # noqa: E402
from jperceiver_amd.apis import batch_processor, build_optimizer, Runner      # noqa: E402
from jperceiver_amd.core import DistOptimizerHook                              # noqa: E402
from oracle import jp_oracle as J                                              # noqa: E402
from tests.golden_util import referee_bound                                   # noqa: E402

CASES = {
    "argo_both_512_b2": dict(HW=512, B=2, FR=[0, -1, 1], type="Argo_both", split="argo", loss_sum=3, full_hw=(129, 154), seed=31),
    "cfg1_full_B8_1024": dict(HW=1024, B=8, FR=[0, -1, 1], type="static", split="odometry", loss_sum=3, full_hw=(375, 1242), seed=1),
    "traj_256_b2": dict(HW=256, B=2, FR=[0, -1, 1], type="static", split="odometry", loss_sum=3, full_hw=(94, 311), seed=41),
}


def _opt(c):
    o = J.default_opt(frame_ids=c["FR"], imgs_per_gpu=c["B"], height=c["HW"], width=c["HW"], occ_map_size=c["HW"] // 4,
                      type=c["type"], split=c["split"], loss_sum=c["loss_sum"])
    if c["type"] == "Argo_both":
        o.update(loss_weightS=20, loss2_weightS=20)
    return o


def _batch(c, seed):
    HW, B, FR = c["HW"], c["B"], c["FR"]
    inp = syn.make_batch(B, HW, HW, FR, HW // 4, c["full_hw"], c["split"], seed=seed)
    masks = syn.make_dropout_masks(B, HW, HW, seed=seed)
    noise = syn.make_automask_noise(B, HW, HW, 4, len(FR) - 1, seed=seed)
    return inp, masks, noise


def _label(c, opt, inp):
    return torch.nan_to_num(J.make_scale_label(opt, inp), nan=0.0, posinf=0.0, neginf=0.0)


def _device_batch(inp, masks, noise, label):
    d = {k: v.clone() for k, v in inp.items()}
    d[("dropout_mask", 0)], d[("dropout_mask", 1)] = masks[0].clone(), masks[1].clone()
    for s, per in enumerate(noise):
        for j, nz in enumerate(per):
            d[("automask_noise", s, j)] = nz.clone()
    d[("scale_label", 0, 0)] = label.clone()
    return d


def _f64_grads(c, opt, P32, Bf32, inp, masks, noise, label, force):
    """float64 oracle from the GIVEN state (the referee of tests/golden_util.py, one step into the trajectory)"""
    P = {n: p.detach().double().requires_grad_(True) for n, p in P32.items()}
    Bf = {n: (b.double() if b.dtype == torch.float32 else b.clone()) for n, b in Bf32.items()}
    inp64 = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in inp.items()}
    torch.set_default_dtype(torch.float64)
    try:
        _, L = J.forward(P, Bf, opt, inp64, True, tuple(m.double() for m in masks), [[z.double() for z in per] for per in noise],
                         label.double(), force)
        J.total_loss(L).backward()
    finally:
        torch.set_default_dtype(torch.float32)
    return {n: p.grad for n, p in P.items() if p.grad is not None}


def _feed_device_grads(model, P):
    for n, p in model.named_parameters():
        if n in P:
            P[n].grad = None if P[n].grad is None else p.grad.detach().cpu().clone()


@pytest.mark.parametrize("name", ["argo_both_512_b2", "cfg1_full_B8_1024"])
def test_second_step_matches_oracle(name):
    c = CASES[name]
    opt = _opt(c)
    model = MONO.module_dict["Baseline"](opt)
    state = syn.synth_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(state, strict=True)
    model = model.cuda().train()
    optim = build_optimizer(model, dict(type="Adam", lr=1e-4, weight_decay=0))
    optim.max_norm, optim.grad_scale = 35.0, 1.0
    shapes = J.state_shapes(c["HW"] // 4)
    P, Bf = J.make_params(shapes, state)
    adam = {}
    named = dict(model.named_parameters())
    for step in (1, 2):
        inp, masks, noise = _batch(c, c["seed"] + 100 * (step - 1))          # step 2 sees another batch
        label = _label(c, opt, inp)
        optim.zero_grad()
        out, losses = model({k: v.cuda() for k, v in _device_batch(inp, masks, noise, label).items()})
        total = losses.total()
        total.backward()
        torch.cuda.synchronize()
        force = {("min_index", s): out[("min_index", s)].cpu() for s in range(4)}
        for tag in ("road", "car"):
            force["cv_argmax_" + tag] = out["cv_argmax_" + tag].cpu()
            force["cm_argmax_" + tag] = out["cm_argmax_" + tag].cpu()
        for p in P.values():
            p.grad = None
        P0 = {n: p.detach().clone() for n, p in P.items()}        # (the state this step starts from: float64 referee, step-2 checks)
        B0 = {n: b.clone() for n, b in Bf.items()}
        o2, L2 = J.forward(P, Bf, opt, inp, True, masks, noise, label, force)
        tot2 = J.total_loss(L2)
        tot2.backward()
        # ---- losses, poses, maps (both steps; step 2 is the point of the test)
        assert set(L2) == set(losses)
        for k in L2:
            a, b = float(losses[k]), float(L2[k])
            assert abs(a - b) <= 2e-3 * max(abs(b), 1e-4), (name, step, k, a, b)
        for f in c["FR"][1:]:
            np.testing.assert_allclose(out[("cam_T_cam", 0, f)].cpu().numpy(), o2[("cam_T_cam", 0, f)].detach().numpy(), atol=1e-4)
        for s in range(4):
            a, b = out[("disp", 0, s)].cpu(), o2[("disp", 0, s)].detach()
            assert float((a - b).abs().max() / b.abs().max()) < 1e-3, (name, step, "disp", s)
        for k in ("topview", "topviewB"):
            if k in o2 and o2[k] is not None and k in out:
                a, b = out[k].cpu(), o2[k].detach()
                assert float((a - b).abs().max() / b.abs().max()) < 1e-3, (name, step, k)
        # ---- gradients.  The photometric terms' gradients are ill-conditioned (sums over 10^6 signed per-pixel terms whose sample
        # positions move with every weight): a 1e-6 relative difference in the PARAMETERS moves them by several percent, and per term
        # the two fp32 evaluations sit 1-5 % from float64 (tools/debug/step2_terms.py).  Both sides therefore evaluate step 2 at the
        # device's parameters (synchronised below, after the Adam check), and step 2 is then as close as step 1 (measured medians
        # 0.1-0.5 %).  So:
        #  (a) total gradient: every parameter within 15 % of its norm around the fp32 oracle -- or, failing that, within 20 % of the
        #      float64 oracle -- and the median parameter within 1 % of the fp32 oracle (a pack, scale header or counter that did not
        #      follow the weights moves EVERY gradient by more than that);
        bad, errs = [], []
        for n, p in named.items():
            r = P[n].grad if n in P else None
            if r is None:
                assert float(p.grad.abs().max()) == 0.0, (step, n)
                continue
            rn = float(r.norm())
            err = float((p.grad.detach().cpu() - r).norm())
            if p.numel() > 1:
                errs.append(err / (rn + 1e-30))
            if err > 0.15 * rn + 2e-5 * abs(float(tot2)):
                bad.append((n, err, rn))
        if bad:
            # Referee as in the one-step tests: what misses the band around the fp32 oracle must be within 20 % of FLOAT64 (the
            # photometric terms cancel, so either fp32 evaluation may land a few percent out).  History: the "pose networks 88-100 %
            # off on some boxes" this branch once reported was NOT the oracle's noise but a real lifetime bug of the temporary batch
            # on the side stream (DESIGN section 1 "Streams", tests/test_batch_lifetime_gpu.py); since the fix the branch is not taken.
            g64 = _f64_grads(c, opt, P0, B0, inp, masks, noise, label, force)
            worse = []
            for n, err, rn in bad:
                r64 = g64[n]
                eh = float((named[n].grad.detach().cpu().double() - r64).norm() / (r64.norm() + 1e-30))
                ec = float((P[n].grad.double() - r64).norm() / (r64.norm() + 1e-30))
                print(f"referee {name} step {step} {n}: hip {eh:.4f} fp32-oracle {ec:.4f}")
                if eh > 0.2:
                    worse.append((n, eh, ec))
            assert not worse, f"{name} step {step}: gradients more than 20 % off the float64 oracle (name, hip, cpu32): {worse[:8]}"
        errs.sort()
        print(f"{name} step {step}: gradient distance to the fp32 oracle per parameter: median {errs[len(errs) // 2]:.3e}, max {errs[-1]:.3e}; "
              f"outside the band: {len(bad)}")
        assert errs[len(errs) // 2] <= 1e-2, (name, step, errs[len(errs) // 2])     # measured: 0.10-0.23 % (512^2), 0.45-0.50 % (B = 8, 1024^2), steps 1 and 2 alike
        if step == 2:
            #  (b) every term EXCEPT the photometric ones (scale, smoothness, layout / cycle losses: well-conditioned, and their
            #      backward runs through the same decoder / encoder / head kernels): element-wise, 2 % of each parameter's gradient
            #      norm -- the bar of the one-step tests, one step later
            wc = [k for k in L2 if not (isinstance(k, tuple) and k[0] == "min_reconstruct_loss")]
            lnames = list(losses._lv.names)
            sel = torch.zeros(len(lnames), device="cuda")
            for k in wc:
                sel[lnames.index(k)] = 1.0
            # (the two extra forward passes below must not advance the BatchNorm buffers a second and third time)
            bufs = {n: b_.detach().clone() for n, b_ in model.named_buffers()}
            pend = [(m, m._pending) for m in model.modules() if hasattr(m, "_pending")]
            optim.zero_grad()
            out_b, losses_b = model({k: v.cuda() for k, v in _device_batch(inp, masks, noise, label).items()})
            losses_b._node.backward(gradient=sel)
            torch.cuda.synchronize()
            Pb = {n: p.detach().clone().requires_grad_(True) for n, p in P0.items()}
            Bb = {n: b_.clone() for n, b_ in B0.items()}
            _, Lb = J.forward(Pb, Bb, opt, inp, True, masks, noise, label, force)
            sum(Lb[k].mean() for k in wc).backward()
            badb = []
            for n, p in named.items():
                r = Pb[n].grad if n in Pb else None
                if r is None or float(r.norm()) == 0.0:
                    continue
                rn = float(r.norm())
                err = float((p.grad.detach().cpu() - r).norm())
                tol = 8e-2 if p.numel() == 1 else 4e-2 if ("query_conv" in n or "key_conv" in n) else 2e-2
                if err > tol * rn + 2e-5 * abs(float(tot2)):
                    badb.append((n, err / rn))
            assert not badb, f"{name} step 2, non-photometric terms: gradients outside the 2 % band: {badb[:8]}"
            # restore the TOTAL gradient of step 2 for the optimizer check below
            optim.zero_grad()
            out_c, losses_c = model({k: v.cuda() for k, v in _device_batch(inp, masks, noise, label).items()})
            losses_c.total().backward()
            torch.cuda.synchronize()
            with torch.no_grad():
                for n, b_ in model.named_buffers():
                    b_.copy_(bufs[n])
            for m, v_ in pend:
                m._pending = v_
        # (step 1's element-wise gradient check with its referee is tests/test_step_parity_gpu.py / test_config_steps_gpu.py)
        # ---- clip + Adam, step `step`: the oracle's update fed the device gradients reproduces the arena update
        _feed_device_grads(model, P)
        norm_ref = J.adam_step(P, adam, lr=1e-4, max_norm=35.0)
        optim.step()
        torch.cuda.synchronize()
        assert abs(float(optim.arena.normsq.sqrt()) - norm_ref) <= 1e-5 * norm_ref, (name, step)
        worst = max(float((p.detach().cpu() - P[n].detach()).abs().max()) for n, p in named.items() if n in P)
        assert worst <= 1e-6 * step, f"{name}: parameters after step {step} differ from the oracle trajectory by {worst}"
        assert optim.arena.step_count == step == adam["t"]
        # Both trajectories continue from the DEVICE's parameters.  The two Adam updates agree to an ulp or two (asserted above), but the
        # photometric terms amplify a 1e-6 relative parameter difference into several percent of gradient: measured with
        # tools/debug/step2_forward_referee.py, the device's step-2 forward sits 2e-7 from the float64 forward AT ITS OWN parameters
        # and 2e-6 from the float64 forward at the oracle's, and the round's first "step-2 gradients 5-13 % from float64" were that
        # difference, not the kernels (profiles/r06_step2_forward_referee.log).
        with torch.no_grad():
            for n, p in named.items():
                if n in P:
                    P[n].copy_(p.detach().cpu())
        # BatchNorm buffers walk with the oracle's (running statistics after `step` momentum updates, counters)
        sd = model.state_dict()
        for n, b in Bf.items():
            if n.endswith("num_batches_tracked"):
                assert int(sd[n]) == int(b), (step, n)
            else:
                assert float((sd[n].cpu() - b).abs().max()) <= 1e-3 * max(1e-3, float(b.abs().max())), (step, n)


@pytest.mark.parametrize("mode", ["eager", "captured"])
def test_trajectory_through_runner(mode):
    c = CASES["traj_256_b2"]
    opt = _opt(c)
    model = MONO.module_dict["Baseline"](opt)
    state = syn.synth_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(state, strict=True)
    model = model.cuda().train()
    optim = build_optimizer(model, dict(type="Adam", lr=1e-4, weight_decay=0))
    runner = Runner(model, batch_processor, optim, DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2)),
                    step_graph=(mode == "captured"))
    shapes = J.state_shapes(c["HW"] // 4)
    P, Bf = J.make_params(shapes, state)
    adam = {}
    named = dict(model.named_parameters())
    STEPS = 5
    for it in range(STEPS):
        inp, masks, noise = _batch(c, c["seed"] + 7 * it)
        label = _label(c, opt, inp)
        out = runner.train_iter(_device_batch(inp, masks, noise, label))
        torch.cuda.synchronize()
        lv = dict(out["log_vars"])
        for p in P.values():
            p.grad = None
        _, L2 = J.forward(P, Bf, opt, inp, True, masks, noise, label)
        J.total_loss(L2).backward()
        for k in L2:
            a, b = lv[str(k)], float(L2[k])
            assert abs(a - b) <= 2e-3 * max(abs(b), 1e-4), (mode, it, k, a, b)
        # the gradients of THIS iteration are still in the arena (zero_grad opens the next iteration)
        badn = []
        tot = abs(float(J.total_loss(L2)))
        for n, p in named.items():
            r = P[n].grad if n in P else None
            if r is None:
                continue
            gn, rn = float(p.grad.double().norm()), float(r.double().norm())
            if abs(gn - rn) > (1e-1 if p.numel() == 1 else 5e-2) * rn + 2e-5 * tot:
                badn.append((n, gn, rn))
        assert not badn, f"{mode} iteration {it}: gradient norms off the oracle's: {badn[:6]}"
        _feed_device_grads(model, P)
        J.adam_step(P, adam, lr=1e-4, max_norm=35.0)
        worst = max(float((p.detach().cpu() - P[n].detach()).abs().max()) for n, p in named.items() if n in P)
        assert worst <= 1e-6 * (it + 1), f"{mode}: parameters after iteration {it} are {worst} off the oracle trajectory"
    assert optim.arena.step_count == STEPS
    if mode == "captured":
        assert runner.captured is not None and runner.captured.replays >= 2
