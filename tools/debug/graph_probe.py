#!/usr/bin/env python
"""Which part of the training step survives hipGraph capture?  One probe per process (a failing capture can take the process
down):  python tools/debug/graph_probe.py {fwd|fwd_bwd|full} [hw]   (stream switches through JP_POSE_STREAM / JP_WGRAD_STREAM /
JP_LAYOUT_ENC_SIDE in the environment)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch                                                                   # noqa: E402
from jperceiver_amd import ops, synthetic as syn                               # noqa: E402
from jperceiver_amd.model import MONO                                          # noqa: E402
from jperceiver_amd.apis import batch_processor, build_optimizer, Runner, change_input_variable   # noqa: E402
from jperceiver_amd.core import DistOptimizerHook                              # noqa: E402
from oracle import jp_oracle as J                                              # noqa: E402


def main():
    mode = sys.argv[1]
    HW = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    FR = [0, -1, 1]
    opt = J.default_opt(frame_ids=FR, imgs_per_gpu=1, height=HW, width=HW, occ_map_size=HW // 4, type="static", split="odometry")
    model = MONO.module_dict["Baseline"](opt)
    model.load_state_dict(syn.synth_state_dict(model.state_dict(), seed=0))
    model = model.cuda().train()
    optim = build_optimizer(model, dict(type="Adam", lr=1e-4, weight_decay=0))
    runner = Runner(model, batch_processor, optim, DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2)))
    batch = change_input_variable(syn.make_batch(1, HW, HW, FR, HW // 4, (94, 311), "odometry", seed=3), opt=model.opt)
    for _ in range(2):
        runner.train_iter(batch)
    torch.cuda.synchronize()
    print(f"probe {mode}: eager warm-up done", flush=True)
    g = torch.cuda.CUDAGraph()
    state = torch.zeros(3, device="cuda")
    base = torch.zeros(1, device="cuda", dtype=torch.int64)
    optim.arena.dev_state = state
    with ops.rng_capture(base), torch.cuda.graph(g):
        out, losses = model(dict(batch))
        if mode != "fwd":
            loss = losses.total()
            if mode == "fwd_bwd":
                optim.zero_grad()
                loss.backward()
            else:
                runner.outputs = dict(loss=loss)
                runner.hook.after_train_iter(runner)
    print(f"probe {mode}: capture ended", flush=True)
    state.copy_(torch.tensor([1e-4, 0.1, 0.001]))
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print(f"probe {mode}: OK, loss vector sum {float(losses._lv.vals.sum()):.5f}", flush=True)


if __name__ == "__main__":
    main()
