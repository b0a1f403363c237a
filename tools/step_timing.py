"""GPU-box aid: time a few train steps at a given size and print a torch-event breakdown per C-ABI entry point."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_opt
from jperceiver_amd import synthetic as syn, _lib, ops, ops_loss
from jperceiver_amd.model import MONO, net as netmod
from jperceiver_amd.apis import batch_processor, build_optimizer, Runner, change_input_variable
from jperceiver_amd.core import DistOptimizerHook
import jperceiver_amd.runtime as rt, jperceiver_amd.model.modules as mods

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 512
ty = sys.argv[3] if len(sys.argv) > 3 else "static"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
model = MONO.module_dict["Baseline"](make_opt(B, HW, [0, -1, 1], ty, "argo" if ty == "Argo_both" else "odometry"))
model.load_state_dict(syn.synth_state_dict(model.state_dict(), seed=0))
model = model.cuda().train()
optim = build_optimizer(model, dict(type="Adam", lr=1e-4, weight_decay=0))
runner = Runner(model, batch_processor, optim, DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2)))
batch = change_input_variable(syn.make_batch(B, HW, HW, [0, -1, 1], HW // 4, (375, 1242), "odometry", seed=1), opt=model.opt)
torch.cuda.synchronize()
t0 = time.perf_counter(); runner.train_iter(batch); torch.cuda.synchronize(); print("warm-up step %.3f s" % (time.perf_counter() - t0), flush=True)
for i in range(steps):
    t0 = time.perf_counter(); o = runner.train_iter(batch); torch.cuda.synchronize()
    print("step %d: %.1f ms  loss %.4f" % (i, (time.perf_counter() - t0) * 1e3, o["log_vars"]["loss"]), flush=True)
# per-entry-point breakdown with events (serialises nothing: events are on the launch stream)
rec = []
orig = _lib.call
def timed(name, *a):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(name, *a); e1.record()
    key = name
    if name == "jp_conv2d_fwd_src3":
        key = "fwd   %4d->%4d k%d s%d @%4d fl=%.1f" % (a[1] + a[4] + a[7], a[15], a[16], a[17], a[13], 2e-9 * a[12] * (a[13] // a[17]) * (a[14] // a[17]) * a[15] * (a[1] + a[4] + a[7]) * a[16] ** 2)
    elif name == "jp_conv2d_dgrad":
        key = "dgrad %4d->%4d k%d s%d @%4d fl=%.1f" % (a[4], a[7], a[8], a[9], a[5], 2e-9 * a[3] * (a[5] // a[9]) * (a[6] // a[9]) * a[7] * a[4] * a[8] ** 2)
    elif name == "jp_conv2d_dgrad_src3":
        cin = a[3] + a[7] + a[11]
        key = "dgrad3 %4d->%4d k%d s%d @%4d fl=%.1f" % (cin, a[17], a[18], a[19], a[15], 2e-9 * a[14] * a[15] * a[16] * a[17] * cin * a[18] ** 2)
    elif name == "jp_conv2d_wgrad_src3":
        key = "wgrad %4d->%4d k%d s%d @%4d fl=%.1f" % (a[1] + a[4] + a[7], a[14], a[15], a[16], a[12], 2e-9 * a[11] * (a[12] // a[16]) * (a[13] // a[16]) * a[14] * (a[1] + a[4] + a[7]) * a[15] ** 2)
    elif name in ("jp_maxpool_fwd", "jp_maxpool_bwd"):
        o = 1 if name == "jp_maxpool_bwd" else 0
        key = "%s NC=%d %dx%d k%d s%d" % (name, a[3 + o], a[4 + o], a[5 + o], a[6 + o], a[7 + o])
    elif name == "jp_bn_train_fwd":
        key = "bn_fwd N=%d C=%d HW=%d res=%d MB=%.0f" % (a[10], a[11], a[12], a[3] is not None, 4e-6 * a[10] * a[11] * a[12])
    elif name == "jp_bn_train_bwd":
        key = "bn_bwd N=%d C=%d HW=%d res=%d MB=%.0f" % (a[11], a[12], a[13], a[7] is not None, 4e-6 * a[11] * a[12] * a[13])
    elif name == "jp_axpby":
        key = "axpby n=%.1fM b=%d" % (a[3] * 1e-6, a[1] is not None)
    elif name == "jp_upsample2x_fwd":
        key = "up2x_fwd N=%d C=%d %dx%d dstC=%d" % (a[2], a[3], a[4], a[5], a[6])
    rec.append((key, e0, e1))
for m in (ops, ops_loss, netmod, rt, mods):
    if hasattr(m, "call"): m.call = timed
t0 = time.perf_counter(); runner.train_iter(batch); torch.cuda.synchronize(); wall = time.perf_counter() - t0
agg = collections.defaultdict(lambda: [0, 0.0])
for n, e0, e1 in rec:
    agg[n][0] += 1; agg[n][1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print("instrumented step wall %.1f ms, sum of kernel events %.1f ms, %d launches" % (wall * 1e3, tot, len(rec)))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("TOPN", "25"))]:
    tf = ""
    if "fl=" in n:
        tf = "  %.1f TF" % (float(n.split("fl=")[1]) * c / t)
    print("  %-44s %4d x %9.2f ms  %5.1f%%%s" % (n, c, t, 100 * t / tot, tf))
print("max mem GB", torch.cuda.max_memory_allocated() / 2**30)
