import sys, time, os
sys.path.insert(0, "/root/repo")
import torch, bench
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS[1]
optd = bench.make_opt(8, 1024, 1024, cfg["frames"], cfg["type"], cfg["split"], loss_sum=cfg["loss_sum"])
runner, batch = bench.build_runner(optd, dev, 1, 0, dict(B=8, height=1024, width=1024, frame_ids=cfg["frames"], occ=256, full_hw=cfg["full_hw"], split=cfg["split"], seed=1))
for _ in range(3):
    runner.train_iter(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
hs = []
for _ in range(10):
    h0 = time.perf_counter()
    runner.train_iter(batch)
    hs.append(time.perf_counter() - h0)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"host enqueue total {t_host*100:.1f} ms/step, wall {dt*100:.1f} ms/step; per-iter host {[round(h*1e3,1) for h in hs]}")
