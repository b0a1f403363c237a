"""The same step (bit-reproducible since round 6) with the BatchNorm statistics taken from the convolutions' epilogues and with the
pass over y: per parameter, how far apart the gradients are (expected: the last bits of fp32 sums in another order).
    python tools/debug/bnfuse_diff.py [HW=1024] [B=8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jperceiver_amd import ops, synthetic as syn
from jperceiver_amd.model import MONO, modules as mods
from jperceiver_amd.apis import batch_processor, build_optimizer, Runner
from jperceiver_amd.core import DistOptimizerHook
from oracle import jp_oracle as J
HW = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
FR = [0, -1, 1]
opt = J.default_opt(frame_ids=FR, imgs_per_gpu=B, height=HW, width=HW, occ_map_size=HW // 4, type="static", split="odometry")
batch = syn.make_batch(B, HW, HW, FR, HW // 4, (375, 1242), "odometry", seed=1)
res = {}
for fuse in (True, False):
    mods._BN_STATS_FUSE = fuse
    model = MONO.module_dict["Baseline"](opt)
    model.load_state_dict(syn.synth_state_dict(model.state_dict(), seed=0))
    model = model.cuda().train()
    optim = build_optimizer(model, dict(type="Adam", lr=1e-4, weight_decay=0))
    runner = Runner(model, batch_processor, optim, DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2)))
    ops.manual_seed(7)
    out = runner.train_iter({k: v.clone() for k, v in batch.items()})
    torch.cuda.synchronize()
    res[fuse] = (dict(out["log_vars"]), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    del model, optim, runner
la, ga = res[True]
lb, gb = res[False]
print("loss terms:", {k: (la[k], lb[k]) for k in la if abs(la[k] - lb[k]) > 1e-5 * max(1.0, abs(lb[k]))} or "equal to 1e-5")
rows = sorted(((float((ga[n] - gb[n]).norm() / (gb[n].norm() + 1e-30)), n) for n in ga), reverse=True)
for d, n in rows[:25]:
    print(f"  {d:.3e}  {n}")
