"""The captured training step (apis/trainer.py CapturedStep: one hipGraph per iteration for the B = 1 configs,
config/cfg_kitti_baseline_argo_both_boundary_ce_iou_1024_20_B1.py:4-6) against the eager step it was captured from.

Two runners start from identical weights and the same device-RNG seed and train the same 7 batches (three distinct ones,
cycled; the learning rate halves before step 5): one issues every launch (eager), the other runs 2 eager warm-up
iterations, captures, and replays.  Per-step losses, the parameters, Adam's moments and the BatchNorm counters must agree --
bit for bit when the eager step itself is reproducible run to run (checked first), at fp32 rounding otherwise (a few kernels
fold partial sums with atomics).  The Dropout masks and automask noise come from the device generator, so equality also
proves that a replay draws what the eager step would have drawn."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from jperceiver_amd import ops, synthetic as syn                                # noqa: E402
from jperceiver_amd.model import MONO                                          # noqa: E402
from jperceiver_amd.apis import batch_processor, build_optimizer, Runner       # noqa: E402
from jperceiver_amd.apis.trainer import CapturedStep                           # noqa: E402
from jperceiver_amd.core import DistOptimizerHook                              # noqa: E402
from oracle import jp_oracle as J                                              # noqa: E402

HW, B, FR, STEPS = 256, 1, [0, -1, 1], 7


def _run(ty, graph):
    opt = J.default_opt(frame_ids=FR, imgs_per_gpu=B, height=HW, width=HW, occ_map_size=HW // 4, type=ty,
                        split="argo" if ty.startswith("Argo") else "odometry", loss_weightS=20, loss2_weightS=20)
    model = MONO.module_dict["Baseline"](opt)
    model.load_state_dict(syn.synth_state_dict(model.state_dict(), seed=0))
    model = model.cuda().train()
    optim = build_optimizer(model, dict(type="Adam", lr=1e-4, weight_decay=0))
    runner = Runner(model, batch_processor, optim, DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2)), step_graph=graph)
    split = "argo" if ty.startswith("Argo") else "odometry"
    batches = [syn.make_batch(B, HW, HW, FR, HW // 4, (129, 154) if split == "argo" else (94, 311), split, seed=80 + i)
               for i in range(3)]
    ops.manual_seed(11)
    losses = []
    for i in range(STEPS):
        if i == 4:
            optim.param_groups[0]["lr"] = 5e-5
        out = runner.train_iter({k: v.clone() for k, v in batches[i % 3].items()})
        losses.append(dict(out["log_vars"]))
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    a = optim.arena
    return dict(losses=losses, sd=sd, m=a.exp_avg.cpu().clone(), v=a.exp_avg_sq.cpu().clone(), step=a.step_count,
                replays=runner.captured.replays if runner.captured else 0, ctr=ops._RNG_STATE["ctr"])


@pytest.mark.parametrize("ty", ["static", "Argo_both"])
def test_captured_step_equals_eager_step(ty):
    e1, e2, g = _run(ty, False), _run(ty, False), _run(ty, True)
    assert g["replays"] == STEPS - CapturedStep.WARMUP and g["step"] == e1["step"] == STEPS and g["ctr"] == e1["ctr"] > 0
    reproducible = all(torch.equal(e1["sd"][k], e2["sd"][k]) for k in e1["sd"])
    worst = 0.0
    for k in e1["sd"]:
        a, b = e1["sd"][k], g["sd"][k]
        if k.endswith("num_batches_tracked"):
            assert int(a) == int(b) > 0, k
            continue
        if reproducible:
            assert torch.equal(a, b), f"{k}: replayed step differs from the eager step"
        worst = max(worst, float((a.float() - b.float()).abs().max()))
    for i, (le, lg) in enumerate(zip(e1["losses"], g["losses"])):
        assert le.keys() == lg.keys()
        for k in le:
            tol = 0.0 if reproducible else 1e-4 * max(1.0, abs(le[k]))
            assert abs(le[k] - lg[k]) <= tol, (i, k, le[k], lg[k])
    if reproducible:
        assert torch.equal(e1["m"], g["m"]) and torch.equal(e1["v"], g["v"])
    else:   # lr-sized moves: a sign flip of a ~0 gradient moves a weight by ~2 lr per step
        assert worst <= 2.5e-4 * STEPS, worst
    print(f"{ty}: eager step reproducible run to run: {reproducible}; max |param difference| graph vs eager {worst:.3e}")


def test_captured_step_recaptures_on_a_new_signature_and_refuses_foreign_setups():
    opt = J.default_opt(frame_ids=[0, -1], imgs_per_gpu=1, height=HW, width=HW, occ_map_size=HW // 4, type="static", split="odometry")
    model = MONO.module_dict["Baseline"](opt)
    model.load_state_dict(syn.synth_state_dict(model.state_dict(), seed=0))
    model = model.cuda().train()
    optim = build_optimizer(model, dict(type="Adam", lr=1e-4, weight_decay=0))
    runner = Runner(model, batch_processor, optim, DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2)), step_graph=True)
    b1 = syn.make_batch(1, HW, HW, [0, -1], HW // 4, (94, 311), "odometry", seed=3)
    for _ in range(4):
        runner.train_iter(dict(b1))
    assert runner.captured.replays == 2
    b2 = syn.make_batch(1, HW, HW, [0, -1], HW // 4, (120, 400), "odometry", seed=4)        # another full-resolution frame size
    for _ in range(3):
        out = runner.train_iter(dict(b2))
    assert runner.captured.replays == 3 and runner.iter == 7
    assert all(v == v for v in out["log_vars"].values())
    with pytest.raises(RuntimeError):
        CapturedStep(Runner(model, lambda *a, **k: None, optim, None))
