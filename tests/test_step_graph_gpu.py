"""The captured training step (apis/trainer.py CapturedStep: one hipGraph per iteration for the B = 1 configs,
config/cfg_kitti_baseline_argo_both_boundary_ce_iou_1024_20_B1.py:4-6) against the eager step it was captured from.

An eager runner and a graph runner take the same 7 steps (three distinct batches, cycled; the learning rate halves before step
5); before every step the eager runner's state -- parameters, Adam moments, BatchNorm buffers, device-RNG position -- is copied
to the graph runner, so each replay is compared with the eager step FROM THE SAME STATE (Adam's early updates are ~lr * sign(g):
comparing whole trajectories would measure how fast rounding noise of the few atomically-folded sums is amplified, not the
capture).  The Dropout masks and automask noise come from the device generator, so agreement also proves that a replay draws
what the eager step would have drawn."""
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


def _runner(ty, graph):
    split = "argo" if ty.startswith("Argo") else "odometry"
    opt = J.default_opt(frame_ids=FR, imgs_per_gpu=B, height=HW, width=HW, occ_map_size=HW // 4, type=ty, split=split,
                        loss_weightS=20, loss2_weightS=20)
    model = MONO.module_dict["Baseline"](opt)
    model.load_state_dict(syn.synth_state_dict(model.state_dict(), seed=0))
    model = model.cuda().train()
    optim = build_optimizer(model, dict(type="Adam", lr=1e-4, weight_decay=0))
    return Runner(model, batch_processor, optim, DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2)), step_graph=graph), split


def _copy_state(src, dst):
    """dst <- src: parameters, Adam moments + step, BatchNorm buffers and counters (both runners then start a step identically)"""
    a, b = src.optimizer.arena, dst.optimizer.arena
    b.params.copy_(a.params)
    if a.exp_avg is not None:
        if b.exp_avg is None:
            b.exp_avg, b.exp_avg_sq = torch.zeros_like(b.params), torch.zeros_like(b.params)
        b.exp_avg.copy_(a.exp_avg)
        b.exp_avg_sq.copy_(a.exp_avg_sq)
    b.step_count = a.step_count
    ma, mb = src.model, dst.model
    for (na, ba), (nb, bb) in zip(ma.named_buffers(), mb.named_buffers()):
        assert na == nb
        bb.copy_(ba)
    for xa, xb in zip(ma.modules(), mb.modules()):
        if hasattr(xa, "_pending"):
            xb._pending = xa._pending
    ops.weights_changed()


@pytest.mark.parametrize("ty", ["static", "Argo_both"])
def test_captured_step_equals_eager_step(ty):
    """Step by step from IDENTICAL state (the eager runner's parameters, Adam moments, BatchNorm buffers and RNG position are
    copied over before every step): 2 eager warm-up iterations, the capture, then 5 replays, each against the eager step on the
    same batch.  Round 6: BIT FOR BIT -- every loss term, the gradient arena and the parameters after clip + Adam.  (Until round 5
    a few sums met in float atomics and the two issue orders differed in their last bits; tests/test_step_repro_gpu.py has the list
    of what was made order-independent.)  Adam's step counter / bias corrections, the halved lr from step 5 on, the BatchNorm
    counters and the device RNG position follow the eager run exactly."""
    E, split = _runner(ty, False)
    G, _ = _runner(ty, True)
    batches = [syn.make_batch(B, HW, HW, FR, HW // 4, (129, 154) if split == "argo" else (94, 311), split, seed=80 + i)
               for i in range(3)]
    ops.manual_seed(11)
    worst_p, worst_l, frac = 0.0, 0.0, 0.0
    for i in range(STEPS):
        lr = 1e-4 if i < 4 else 5e-5
        E.optimizer.param_groups[0]["lr"] = G.optimizer.param_groups[0]["lr"] = lr
        _copy_state(E, G)
        c0 = ops._RNG_STATE["ctr"]
        oe = E.train_iter({k: v.clone() for k, v in batches[i % 3].items()})
        c1 = ops._RNG_STATE["ctr"]
        ops._RNG_STATE["ctr"] = c0
        og = G.train_iter({k: v.clone() for k, v in batches[i % 3].items()})
        assert ops._RNG_STATE["ctr"] == c1 > c0, "the replay did not advance the device RNG like the eager step"
        torch.cuda.synchronize()
        le, lg = dict(oe["log_vars"]), dict(og["log_vars"])
        assert le.keys() == lg.keys()
        for k in le:
            d = abs(le[k] - lg[k]) / max(1.0, abs(le[k]))
            worst_l = max(worst_l, d)
            assert le[k] == lg[k], (i, k, le[k], lg[k])          # round 6: no run-dependent sum is left on the step path
        pe, pg = E.optimizer.arena.params, G.optimizer.arena.params
        n = E.optimizer.arena.live_numel
        assert torch.equal(E.optimizer.arena.grads[:n], G.optimizer.arena.grads[:n]), (i, "gradient arena")
        d = (pe[:n] - pg[:n]).abs()
        worst_p = max(worst_p, float(d.max()))
        frac = max(frac, float((d > 0).float().mean()))
        assert torch.equal(pe[:n], pg[:n]), (i, float(d.max()), float((d > 0).float().mean()))
        assert E.optimizer.arena.step_count == G.optimizer.arena.step_count == i + 1
        for (na, ba), (nb, bb) in zip(E.model.named_buffers(), G.model.named_buffers()):
            if ba.dtype.is_floating_point:
                assert float((ba - bb).abs().max()) <= 1e-5 * max(1.0, float(ba.abs().max())), (i, na)
    assert G.captured.replays == STEPS - CapturedStep.WARMUP
    sd_e, sd_g = E.model.state_dict(), G.model.state_dict()
    for k in sd_e:
        if k.endswith("num_batches_tracked"):
            assert int(sd_e[k]) == int(sd_g[k]) > 0, k
    print(f"{ty}: worst loss-term difference {worst_l:.2e} (rel), worst parameter difference {worst_p:.2e}, "
          f"largest share of elements that differ {frac:.2e}")


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


def test_captured_step_survives_an_eval_forward_with_new_conv_signatures():
    """the captured graph holds raw pointers into the pack registry's job table and every pack entry's
    scratch, and the registry drops its table whenever a conv of a new signature is recorded -- a validation / inference forward at
    another resolution between two replays.  CapturedStep keeps the captured table (and through it the scratch tensors) alive and
    re-captures when the registry has moved on; the step after the eval forward must equal the eager step from the same state."""
    E, split = _runner("static", False)
    G, _ = _runner("static", True)
    b = syn.make_batch(B, HW, HW, FR, HW // 4, (94, 311), split, seed=91)
    ops.manual_seed(3)
    for i in range(4):                                       # 2 warm-up iterations, the capture, 2 replays
        _copy_state(E, G)
        c0 = ops._RNG_STATE["ctr"]
        E.train_iter({k: v.clone() for k, v in b.items()})
        ops._RNG_STATE["ctr"] = c0
        G.train_iter({k: v.clone() for k, v in b.items()})
    assert G.captured.replays == 2 and G.captured.recaptures == 0
    # an eval-mode forward with another batch size: every conv gets a new (N, H, W) signature -> new pack entries, table dropped
    m = G.model
    m.eval()
    small = syn.make_batch(3, HW, HW, FR, HW // 4, (94, 311), split, seed=92)
    with torch.no_grad():
        out = m({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in small.items()})
    assert torch.isfinite(out[("disp", 0, 0)]).all()
    m.train()
    torch.cuda.synchronize()
    for i in range(2):
        _copy_state(E, G)
        c0 = ops._RNG_STATE["ctr"]
        oe = E.train_iter({k: v.clone() for k, v in b.items()})
        ops._RNG_STATE["ctr"] = c0
        og = G.train_iter({k: v.clone() for k, v in b.items()})
        torch.cuda.synchronize()
        le, lg = dict(oe["log_vars"]), dict(og["log_vars"])
        for k in le:
            assert abs(le[k] - lg[k]) <= 2e-5 * max(1.0, abs(le[k])), (i, k, le[k], lg[k])
        n = E.optimizer.arena.live_numel
        d = (E.optimizer.arena.params[:n] - G.optimizer.arena.params[:n]).abs()
        assert float(d.max()) <= 2.2e-4 and float((d > 1e-7).float().mean()) < 0.02
    assert G.captured.recaptures == 1, "the graph was replayed against a pack table the registry had dropped"


# ------------------------------------------------------------------------------------------- two ranks
def _two_rank_worker(rank, world, port, q):
    """One rank of a world-2 gloo job, both ranks on the one GPU: an eager runner (overlapped bucketed exchange through
    DistOptimizerHook) and a graph runner (CapturedStep: graph A | eager exchange | graph B) take the same steps from identical
    state on this rank's own batches; every collective is issued in the same order on both ranks."""
    import os
    import traceback
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        import hashlib
        import torch.distributed as dist
        from jperceiver_amd.apis import DataParallelShell
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        opt = J.default_opt(frame_ids=FR, imgs_per_gpu=B, height=HW, width=HW, occ_map_size=HW // 4, type="static", split="odometry")

        def runner(graph):
            model = MONO.module_dict["Baseline"](opt)
            model.load_state_dict(syn.synth_state_dict(model.state_dict(), seed=0))
            model = model.cuda().train()
            optim = build_optimizer(model, dict(type="Adam", lr=1e-4, weight_decay=0))
            return Runner(DataParallelShell(model), batch_processor, optim,
                          DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2), bucket_size_mb=16), step_graph=graph)
        E, G = runner(False), runner(True)
        batches = [syn.make_batch(B, HW, HW, FR, HW // 4, (94, 311), "odometry", seed=60 + 10 * rank + i) for i in range(2)]
        ops.manual_seed(5 + rank)
        worst_l, worst_p, frac = 0.0, 0.0, 0.0
        for i in range(5):
            _copy_state(E, G)
            c0 = ops._RNG_STATE["ctr"]
            oe = E.train_iter({k: v.clone() for k, v in batches[i % 2].items()})
            c1 = ops._RNG_STATE["ctr"]
            ops._RNG_STATE["ctr"] = c0
            og = G.train_iter({k: v.clone() for k, v in batches[i % 2].items()})
            assert ops._RNG_STATE["ctr"] == c1
            torch.cuda.synchronize()
            le, lg = dict(oe["log_vars"]), dict(og["log_vars"])
            for k in le:
                worst_l = max(worst_l, abs(le[k] - lg[k]) / max(1.0, abs(le[k])))
            n = E.optimizer.arena.live_numel
            d = (E.optimizer.arena.params[:n] - G.optimizer.arena.params[:n]).abs()
            worst_p, frac = max(worst_p, float(d.max())), max(frac, float((d > 1e-7).float().mean()))
            assert E.optimizer.arena.step_count == G.optimizer.arena.step_count == i + 1
        digest = hashlib.sha1(G.optimizer.arena.params.detach().cpu().numpy().tobytes()).hexdigest()
        q.put((rank, worst_l, worst_p, frac, G.captured.replays, digest, None))
        dist.destroy_process_group()
    except Exception:
        q.put((rank, 0, 0, 0, 0, "", traceback.format_exc()))


def test_captured_step_two_ranks_two_graphs_around_the_exchange():
    """CapturedStep with world_size 2 (gloo, both ranks on this GPU): 2 eager warm-up iterations (forward + backward | exchange |
    clip + Adam issued as the same three pieces), the two captures, then 3 replays -- each against the eager overlapped step from
    the same state; the replicas of the graph runner stay bit-identical to each other."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=1200) for _ in procs)
    for p in procs:
        p.join(60)
    for r in res:
        assert r[6] is None, f"rank {r[0]} failed:\n{r[6]}"
    for rank, worst_l, worst_p, frac, replays, digest, _ in res:
        assert worst_l <= 2e-5 and worst_p <= 2.2e-4 and frac < 0.02, (rank, worst_l, worst_p, frac)
        assert replays == 3
    assert res[0][5] == res[1][5], "the graph runner's replicas diverged"
