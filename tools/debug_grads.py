"""GPU-box debugging aid: per-loss-term gradient comparison HIP vs oracle for selected parameters."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.golden_util import load_case, case_inputs, oracle_opt, run_oracle
from tests.test_step_parity_gpu import build_model, gpu_inputs
from jperceiver_amd.apis import build_optimizer
from oracle import jp_oracle as J

case = sys.argv[1] if len(sys.argv) > 1 else "argo_both_256_b2"
g, meta = load_case(case)
ora = run_oracle(meta, backward=False)
label = J.scale_label_both(ora["opt"], ora["inp"])
model, opt = build_model(meta)
optim = build_optimizer(model, dict(type="Adam", lr=1e-4, weight_decay=0))
params = dict(model.named_parameters())
watch = ["DepthDecoder.disp4.0.conv.weight", "DepthDecoder.crp4.0.1_pointwise.conv.weight", "DepthDecoder.disp1.0.conv.weight",
         "DepthDecoder.iconv2.conv.weight", "CrossViewTransformer.query_conv.weight", "CrossViewTransformer.key_conv.weight",
         "CrossViewTransformer.value_conv.weight", "PoseDecoder.conv3.weight", "DepthEncoder.encoder.conv1.weight",
         "LayoutEncoder.conv1.conv.weight", "CycledViewProjection.transform_module.fc_transform.0.weight"]
keys = list(ora["L"].keys())
for k in keys:
    optim.zero_grad()
    out, losses = model(gpu_inputs(meta, label))
    losses[k].backward()
    torch.cuda.synchronize()
    og = torch.autograd.grad(ora["L"][k], [ora["P"][n] for n in watch], retain_graph=True, allow_unused=True)
    row = []
    for n, r in zip(watch, og):
        h = params[n].grad.detach().cpu()
        if r is None:
            row.append(f"{n.split('.')[0][:6]}.{n.split('.')[-3][:8]}: none/{float(h.norm()):.1e}")
            continue
        e = float((h - r).norm() / (r.norm() + 1e-30))
        row.append(f"{n.split('.')[0][:6]}.{n.split('.')[-3][:8]}: {e:.1e}")
    print(repr(k), " | ".join(row), flush=True)
