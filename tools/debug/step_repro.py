"""Is the training step bit-reproducible run to run?  Two (or N) steps from identical state on the same batch; per parameter, whether
the gradient arena differs between runs -- names the reductions that still merge partial sums in a run-dependent order (atomics).
    python tools/debug/step_repro.py [HW=256] [B=2] [runs=3] [type=static]"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from jperceiver_amd import ops, synthetic as syn
from jperceiver_amd.model import MONO
from jperceiver_amd.apis import batch_processor, build_optimizer, Runner
from jperceiver_amd.core import DistOptimizerHook
from oracle import jp_oracle as J
HW = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
RUNS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
TY = sys.argv[4] if len(sys.argv) > 4 else "static"
FR = [0, -1, 1]
split = "argo" if TY.startswith("Argo") else "odometry"
opt = J.default_opt(frame_ids=FR, imgs_per_gpu=B, height=HW, width=HW, occ_map_size=HW // 4, type=TY, split=split, loss_weightS=20, loss2_weightS=20)
batch = syn.make_batch(B, HW, HW, FR, HW // 4, (129, 154) if split == "argo" else (94, 311), split, seed=41)
grads, losses = [], []
for r in range(RUNS):
    model = MONO.module_dict["Baseline"](opt)
    model.load_state_dict(syn.synth_state_dict(model.state_dict(), seed=0))
    model = model.cuda().train()
    optim = build_optimizer(model, dict(type="Adam", lr=1e-4, weight_decay=0))
    runner = Runner(model, batch_processor, optim, DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2)))
    ops.manual_seed(7)
    out = runner.train_iter({k: v.clone() for k, v in batch.items()})
    torch.cuda.synchronize()
    grads.append({n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None})
    losses.append(dict(out["log_vars"]))
    del model, optim, runner
bad = {}
for r in range(1, RUNS):
    for n in grads[0]:
        if not torch.equal(grads[0][n], grads[r][n]):
            d = float((grads[0][n] - grads[r][n]).abs().max() / (grads[0][n].abs().max() + 1e-30))
            bad[n] = max(bad.get(n, 0.0), d)
lbad = [k for k in losses[0] if any(losses[0][k] != losses[r][k] for r in range(1, RUNS))]
print(f"HW={HW} B={B} type={TY}: {len(bad)} of {len(grads[0])} gradient tensors differ between {RUNS} runs; loss terms that differ: {lbad}")
import collections
groups = collections.Counter(".".join(n.split(".")[:2]) for n in bad)
for g, c in groups.most_common():
    worst = max(v for n, v in bad.items() if n.startswith(g))
    print(f"   {g:40s} {c:4d} tensors, worst relative difference {worst:.2e}")
if len(sys.argv) > 5:        # which tensors of one group are bit-equal / differ (localises where the run-dependence enters)
    grp = sys.argv[5]
    for n in grads[0]:
        if n.startswith(grp):
            print(f"   {'DIFF ' + format(bad[n], '.1e') if n in bad else 'equal        '}  {n}")
