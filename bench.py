#!/usr/bin/env python
"""bench.py — JPerceiver `Baseline` train-step throughput on N MI355X of one node.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" = one full training iteration of the reference hot loop on one synthetic batch that is already
resident in HBM: batch_processor (forward + all losses) -> DistOptimizerHook.after_train_iter
(zero_grad, backward, gradient all-reduce over RCCL, clip 35, Adam).  Workload at every N (weak scaling):
configs[1] of BASELINE.json, cfg_kitti_baseline_odometry_boundary_ce_iou_1024_20 — 1024x1024 (the size every
reference config uses), frames [0,-1,1], type static, loss_sum 3, occ 256, fp32, 8 images per GPU.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     — dominant kernel = the fp32-MFMA implicit-GEMM 3x3 reflection-pad convolution forward instance
                 (128x128 block tile); achieved =
                 algorithmic FLOPs (2*N*OH*OW*Cout*Cin*9 per launch) / HIP-event time of those launches in one
                 instrumented step; peak 157.3 TFLOP/s (dense fp32 MFMA, MI355X_MICROARCH.md).
  cpu_baseline — the oracle (PyTorch-CPU port of the reference step) timed on this host's cores at B=1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOAD = "cfg_kitti_baseline_odometry_boundary_ce_iou_1024_20"


def make_opt(B, HW, frames, ty="static", split="odometry"):
    return dict(name="Baseline", depth_num_layers=18, pose_num_layers=18, frame_ids=frames, imgs_per_gpu=B,
                height=HW, width=HW, scales=[0, 1, 2, 3], min_depth=0.1, max_depth=100.0,
                depth_pretrained_path=None, pose_pretrained_path=None, automask=True, disp_norm=True,
                smoothness_weight=1e-3, scale_weight=0.1, dynamic_weight=15.0, static_weight=5.0,
                occ_map_size=HW // 4, num_class=2, loss_type="iou", loss_weight=20, loss2_type="boundary",
                loss2_weight=20, type=ty, loss_sum=3, split=split)


def cpu_baseline(HW, frames, seconds_budget=30.0):
    """Oracle (port of the reference CPU path) on this host: B=1, same shapes, 1 warm-up + timed steps."""
    from oracle import jp_oracle as J
    from jperceiver_amd import synthetic as syn
    # threads = cores this process may actually run on (cgroup/affinity aware), capped: oneDNN oversubscription
    # on a 2-socket box is catastrophically slow
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(ncores, 32)))
    opt = J.default_opt(**{k: v for k, v in make_opt(1, HW, frames, "Argo_both", "argo").items() if k != "name"})
    opt.update(loss_weightS=20, loss2_weightS=20)
    shapes = J.state_shapes(opt.occ_map_size)
    tmpl = {n: torch.empty(s, dtype=torch.long if n.endswith("num_batches_tracked") else torch.float32)
            for n, s in shapes.items()}
    P, Bf = J.make_params(shapes, syn.synth_state_dict(tmpl, seed=0))
    inp = syn.make_batch(1, HW, HW, frames, HW // 4, (375, 1242), "argo", seed=11)
    label = J.scale_label_both(opt, inp)
    st = {}
    times = []
    t_begin = time.perf_counter()
    for it in range(8):
        for p in P.values():
            p.grad = None
        t0 = time.perf_counter()
        out, L = J.forward(P, Bf, opt, inp, True, None, None, label)
        J.total_loss(L).backward()
        J.adam_step(P, st)
        dt = time.perf_counter() - t0
        if it > 0:
            times.append(dt)
        if it >= 2 and time.perf_counter() - t_begin > seconds_budget:
            break
    ms = sum(times) / len(times)
    return dict(value=round(1.0 / ms, 4), unit="images/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{len(times)} timed steps (+1 warm-up) of the oracle train step, B=1, {HW}x{HW}, "
                       f"{len(frames)} frames, Argo_both losses, {ms:.2f} s/step")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU (BASELINE.json configs[1]: 8)")
    ap.add_argument("--hw", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > torch.cuda.device_count():
        # several ranks time-slicing one GPU (only the 1-GPU gloo smoke test of the N>1 code path does this): the pose
        # branch's side stream turns into cross-process queue ping-pong there, so keep the step single-stream
        os.environ["JP_POSE_STREAM"] = "0"
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local % torch.cuda.device_count())
        # "nccl" == RCCL over xGMI.  JP_DIST_BACKEND=gloo lets the N>1 code path be exercised on a 1-GPU box.
        dist.init_process_group(os.environ.get("JP_DIST_BACKEND", "nccl"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)

    from jperceiver_amd import synthetic as syn, _lib
    from jperceiver_amd.model import MONO
    from jperceiver_amd.apis import batch_processor, build_optimizer, Runner, DataParallelShell, change_input_variable
    from jperceiver_amd.core import DistOptimizerHook

    B, HW, frames = args.batch, args.hw, [0, -1, 1]
    optd = make_opt(B, HW, frames)
    model = MONO.module_dict["Baseline"](optd)
    model.load_state_dict(syn.synth_state_dict(model.state_dict(), seed=0))
    model = model.to(dev).train()
    optim = build_optimizer(model, dict(type="Adam", lr=1e-4, weight_decay=0))
    wrapped = DataParallelShell(model) if world > 1 else model
    hook = DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2))
    runner = Runner(wrapped, batch_processor, optim, hook)
    # per-rank shard of the synthetic "dataset"; resident in HBM before timing starts
    batch = syn.make_batch(B, HW, HW, frames, HW // 4, (375, 1242), "odometry", seed=1, rank=rank)
    batch = change_input_variable(batch, device=dev, opt=model.opt)
    torch.cuda.synchronize()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    for i in range(args.warmup):
        t_w = time.perf_counter()
        runner.train_iter(batch)
        torch.cuda.synchronize()
        log(f"warm-up step {i}: {(time.perf_counter() - t_w) * 1e3:.1f} ms")
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = runner.train_iter(batch)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms = dt / args.steps * 1e3
    value = B * world * args.steps / dt

    log(f"{args.steps} timed steps: {ms:.1f} ms/step, {value:.2f} images/s")
    if os.environ.get("JP_PMC_CALIB") and rank == 0:
        # known-byte streaming copy (4 B/lane loads and stores, the igemm gather's access width) so that the
        # FETCH_SIZE / WRITE_SIZE counters of a rocprofv3 --pmc pass can be calibrated (MI355X_MICROARCH.md §HBM)
        n = 1 << 28
        a, b = torch.empty(n, device=dev), torch.empty(n, device=dev)
        for _ in range(3):
            _lib.call("jp_axpby", a, None, b, n, 1.0, 0.0)
        torch.cuda.synchronize()
        del a, b
    roof, cpu = None, None
    if not args.no_roofline:
        # every rank runs the instrumented step (it contains the gradient all-reduce); rank 0 reports
        roof = measure_roofline(runner, batch, _lib)
        log(f"roofline: {roof}")
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("timing the CPU baseline (oracle) ...")
        cpu = cpu_baseline(HW, frames)
    if rank == 0:
        line = {
            "metric": "train images/sec (full train step, synthetic 3-frame batches)", "value": round(value, 3),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{WORKLOAD}: {HW}x{HW}, frames {frames}, {B} images/GPU, type static, loss_sum 3, "
                                   f"occ {HW // 4}, full-res frame 375x1242", "global_batch": B * world,
                       "parallelism": f"dp{world}", "loss": float(out["log_vars"]["loss"])},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def measure_roofline(runner, batch, _lib):
    """One extra instrumented step: HIP events (on the launch stream) around every conv-forward launch;
    the dominant kernel is the 3x3 instance with the 128x128 block tile (Cout > 64)."""
    rec = []
    orig = _lib.call

    def timed_call(name, *a):
        if name != "jp_conv2d_fwd_src3":
            return orig(name, *a)
        c0, c1, c2 = a[1], a[4], a[7]
        N, H, W, Cout, KH, stride, pad = a[12], a[13], a[14], a[15], a[16], a[17], a[18]   # jp_conv2d_fwd_src3 ABI order
        OH = (H + 2 * pad - KH) // stride + 1
        OW = (W + 2 * pad - KH) // stride + 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(name, *a)
        e1.record()
        Cin = c0 + c1 + c2
        alg_bytes = 4.0 * (N * Cin * H * W + N * Cout * OH * OW + Cout * Cin * KH * KH)   # read x once, write y once, read w
        fused_up = (c0 and a[2]) or (c1 and a[5]) or (c2 and a[8])   # iconv layers run the parity-class kernel instead
        row_tile = stride == 1 and W % 128 == 0       # runs jp_igemm_r3_kernel (the pixel tile is one image-row segment)
        rec.append((KH if (a[19] == 1 and not fused_up and row_tile) else -KH, Cout, N * OH * OW, 2.0 * N * OH * OW * Cout * Cin * KH * KH, e0, e1, alg_bytes))
    from jperceiver_amd import ops
    ops.call = timed_call
    try:
        runner.train_iter(batch)
        torch.cuda.synchronize()
    finally:
        ops.call = orig
    # dominant kernel = ONE instantiation: 3x3, reflection padding, 128x128 block tile (Cout > 64), row-tile kernel, direct epilogue
    # (>= 192 tiles, i.e. no small-grid split-K)
    dom = [(f, e0.elapsed_time(e1), ab) for KH, Cout, npix, f, e0, e1, ab in rec
           if KH == 3 and Cout > 64 and npix > 64 and ((Cout + 127) // 128) * ((npix + 127) // 128) >= 192]
    flops = sum(f for f, _, _ in dom)
    ms = sum(t for _, t, _ in dom)
    ach = flops / (ms * 1e-3) / 1e12
    # HBM/fabric bytes per launch of this kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE
    # in separate runs, FETCH_SIZE x2 per the gfx950 calibration; tools/pmc_traffic.py) — null if not profiled
    traffic = None
    try:
        import glob
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
        traffic = round(json.load(open(f))["traffic_bytes_per_launch"])
    except Exception:
        pass
    return {"bound": "mfma", "achieved": round(ach, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(ach / 157.3, 4),
            "traffic": traffic, "algorithmic_bytes_per_launch": round(sum(ab for _, _, ab in dom) / max(1, len(dom))), "kernel": "jp_igemm_r3_kernel<2,2,32,PackA,FwdBR3<true,false,128>,FwdEpi> (3x3 reflection-pad conv forward, Cout>64, row-tile variant; the event pair also spans the ~6 us weight-pack launch)",
            "launches": len(dom), "avg_launch_ms": round(ms / max(1, len(dom)), 4),
            "avg_launch_gflop": round(flops / max(1, len(dom)) / 1e9, 2)}


if __name__ == "__main__":
    main()
