"""Weight packs refreshed by ONE `jp_pack_replay` launch (every step after the first) must equal the packs a layer builds on
first use.  Model A takes a training step (its packs are recorded, Adam rewrites the weights) and then runs a second step,
whose convolutions read REPLAYED packs (generic replay kernel + the LDS-staged split-pack replay kernel); model B is a
fresh model loaded with A's updated weights, whose first step packs every layer directly.  Same inputs: the forward pass is
deterministic, so losses and disparities must agree bit for bit; gradients to fp32 rounding (split-K atomics)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from jperceiver_amd import synthetic as syn                                    # noqa: E402
from jperceiver_amd.model import MONO                                          # noqa: E402
from jperceiver_amd.apis import build_optimizer                                # noqa: E402
from oracle import jp_oracle as J                                              # noqa: E402  (option dict only)


# (the 1024^2 case: the iconv dgrad kernel of the upsampled segment -- its own headered pack, PACK_SPLITUPD -- only runs on maps that large)
@pytest.mark.parametrize("ty,HW,B", [("static", 256, 2), ("Argo_both", 512, 1), ("static", 1024, 2)])
def test_replayed_packs_equal_first_use_packs(ty, HW, B):
    FR = [0, -1, 1]
    opt = J.default_opt(frame_ids=FR, imgs_per_gpu=B, height=HW, width=HW, occ_map_size=HW // 4, type=ty,
                        split="argo" if ty == "Argo_both" else "odometry", loss_weightS=20, loss2_weightS=20)
    inp = syn.make_batch(B, HW, HW, FR, HW // 4, (94, 311), opt.split, seed=51)
    masks = syn.make_dropout_masks(B, HW, HW, seed=51)
    noise = syn.make_automask_noise(B, HW, HW, 4, 2, seed=51)

    def feed():
        d = {k: v.cuda() for k, v in inp.items()}
        d[("dropout_mask", 0)], d[("dropout_mask", 1)] = masks[0].cuda(), masks[1].cuda()
        for s, per in enumerate(noise):
            for j, nz in enumerate(per):
                d[("automask_noise", s, j)] = nz.cuda()
        return d

    def step(model, optim, update):
        optim.zero_grad()
        out, losses = model(feed())
        losses.total().backward()
        torch.cuda.synchronize()
        res = dict(losses={k: float(v) for k, v in losses.items()}, disp=[out[("disp", 0, s)].clone() for s in range(4)],
                   top=out["topview"].clone(), grads=optim.arena.grads.clone(),
                   extra={str(k): v.detach().clone() for k, v in out.items()
                          if torch.is_tensor(v) and (k == "origin_features" or (isinstance(k, tuple) and k[0] in ("axisangle", "translation")))})
        if update:
            optim.max_norm, optim.grad_scale = 35.0, 1.0
            optim.step()
            torch.cuda.synchronize()
        return res

    A = MONO.module_dict["Baseline"](opt)
    A.load_state_dict(syn.synth_state_dict(A.state_dict(), seed=0))
    A = A.cuda().train()
    oa = build_optimizer(A, dict(type="Adam", lr=1e-3, weight_decay=0))      # a large step: every weight really changes
    step(A, oa, True)
    state = {k: v.detach().clone() for k, v in A.state_dict().items()}
    ra = step(A, oa, False)                     # reads packs refreshed by jp_pack_replay

    Bm = MONO.module_dict["Baseline"](opt)
    Bm.load_state_dict(state)
    Bm = Bm.cuda().train()
    ob = build_optimizer(Bm, dict(type="Adam", lr=1e-3, weight_decay=0))
    rb = step(Bm, ob, False)                    # packs built on first use

    assert ra["losses"].keys() == rb["losses"].keys()
    # where a difference starts (diagnostic for the assertions below)
    where = {k: float((ra["extra"][k] - rb["extra"][k]).abs().max()) for k in ra["extra"]}
    where.update({f"disp{s}": float((ra["disp"][s] - rb["disp"][s]).abs().max()) for s in range(4)})
    where["top"] = float((ra["top"] - rb["top"]).abs().max())
    where = {k: v for k, v in where.items() if v != 0.0}
    for k in ra["losses"]:
        assert ra["losses"][k] == rb["losses"][k], (k, ra["losses"][k], rb["losses"][k], where)
    for s in range(4):
        assert torch.equal(ra["disp"][s], rb["disp"][s]), f"disp scale {s} differs between replayed and first-use packs"
    assert torch.equal(ra["top"], rb["top"])
    ga, gb = ra["grads"], rb["grads"]
    assert float((ga - gb).norm() / gb.norm()) < 1e-5


def test_registry_forgets_dropped_models_and_replays_long_tables():
    """Pack entries die with their parameters (a registry that kept every model it ever saw alive replayed all of them each
    step), and a replay table longer than the split-replay kernel's 2048-job LDS prefix is still replayed completely."""
    import gc
    import numpy as np
    from jperceiver_amd import ops
    from jperceiver_amd.ops import Var, PackRegistry, param

    dev = torch.device("cuda", torch.cuda.current_device())
    reg = PackRegistry.of(dev)
    gc.collect()
    before = len(reg.entries)
    g = torch.Generator().manual_seed(3)
    x = Var(torch.randn(8, 128, 64, 64, generator=g).cuda())        # enough pixel tiles for the split-bf16 patch kernel

    def make(n):
        ps = [torch.nn.Parameter(torch.randn(128, 128, 3, 3, generator=g).cuda() * 0.03) for _ in range(n)]
        for p in ps:
            with ops.recording(ops.Tape()):
                ops.conv2d(x, param(p), None, 1, 1, 0, 0)
        return ps

    ps = make(3)
    assert len(reg.entries) == before + 3
    del ps
    gc.collect()
    assert len(reg.entries) == before

    # > 2048 split-pack jobs in one table: the last layer's pack must still follow its weights
    ps = make(2100)
    last = ps[-1]
    with ops.recording(ops.Tape()):
        y0 = ops.conv2d(x, param(last), None, 1, 1, 0, 0).t.clone()
    with torch.no_grad():
        last.mul_(2.0)                      # in-place edit: caught through the version counter
    reg.refresh_all()
    with ops.recording(ops.Tape()):
        y1 = ops.conv2d(x, param(last), None, 1, 1, 0, 0).t
    assert torch.allclose(y1, 2.0 * y0, rtol=1e-5, atol=1e-6), float((y1 - 2.0 * y0).abs().max())
    del ps, last
    gc.collect()
    assert len(reg.entries) == before
