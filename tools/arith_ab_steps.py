"""Does the arithmetic scheme change a training run?  The same K optimizer steps (same initial state, same batches, same seeds) under the
environment's kernel selection; prints one JSON line of per-step losses and the final parameters' digest-by-norm.  Run it once per
variant and compare (tools/arith_ab_steps.sh does):
    python tools/arith_ab_steps.py [HW=256] [B=2] [steps=8]            # default kernels (fp16 two-way splits in the default build)
    JP_P9S=0 JP_W9S=0 JP_P9US=0 JP_P9SD=0 JP_P9S2=0 JP_P7S=0 python tools/arith_ab_steps.py ...   # exact-fp32 MFMA kernels"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jperceiver_amd import ops, synthetic as syn
from jperceiver_amd.model import MONO
from jperceiver_amd.apis import batch_processor, build_optimizer, Runner
from jperceiver_amd.core import DistOptimizerHook
from oracle import jp_oracle as J

HW = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
FR = [0, -1, 1]
opt = J.default_opt(frame_ids=FR, imgs_per_gpu=B, height=HW, width=HW, occ_map_size=HW // 4, type="static", split="odometry",
                    loss_weightS=20, loss2_weightS=20)
model = MONO.module_dict["Baseline"](opt)
model.load_state_dict(syn.synth_state_dict(model.state_dict(), seed=0))
model = model.cuda().train()
optim = build_optimizer(model, dict(type="Adam", lr=1e-4, weight_decay=0))
runner = Runner(model, batch_processor, optim, DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2)))
ops.manual_seed(7)
hist = []
for s in range(K):
    batch = syn.make_batch(B, HW, HW, FR, HW // 4, (94, 311), "odometry", seed=100 + s)
    out = runner.train_iter(batch)
    torch.cuda.synchronize()
    hist.append({k: float(v) for k, v in out["log_vars"].items()})
norms = {n: float(p.detach().double().norm()) for n, p in model.named_parameters()}
print("ARITH_AB " + json.dumps(dict(scheme=ops.split_scheme(), env={k: v for k, v in os.environ.items() if k.startswith("JP_")},
                                    losses=hist, param_norm_total=float(sum(v * v for v in norms.values()) ** 0.5),
                                    params=norms)))
