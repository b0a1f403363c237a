"""GPU-box aid: achieved HBM rate of the photometric kernels at the bench shape (8 x 3 x 1024 x 1024)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jperceiver_amd._lib import call

B, H, W = 8, 1024, 1024
dev = "cuda"
pred, tgt = torch.rand(B, 3, H, W, device=dev), torch.rand(B, 3, H, W, device=dev)
out = torch.empty(B, 1, H, W, device=dev)
dpred = torch.empty_like(pred)
mi = torch.randint(0, 4, (B, H, W), device=dev, dtype=torch.int64)
disp = torch.rand(B, 1, H, W, device=dev)
K = torch.tensor([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * H, 0.5 * H, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]], device=dev).repeat(B, 1, 1)
invK = torch.linalg.inv(K).contiguous()
T = torch.eye(4, device=dev).repeat(B, 1, 1); T[:, 0, 3] = 0.1
P = (K @ T)[:, :3].contiguous()
color = torch.rand(B, 3, H, W, device=dev)
warp = torch.empty_like(color)
ddisp = torch.empty(B, 1, H, W, device=dev)
dP = torch.zeros(B, 12, device=dev, dtype=torch.float64)


def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


px = B * H * W
for name, fn, byt in (
        ("ssim_l1_fwd", lambda: call("jp_ssim_l1_fwd", pred, tgt, out, B, H, W), px * 28),
        ("ssim_l1_bwd", lambda: call("jp_ssim_l1_bwd", pred, tgt, mi, 1, None, 1.0, dpred, B, H, W), px * (24 + 8 + 12)),
        ("cgt_warp_fwd", lambda: call("jp_cgt_warp_fwd", disp, H, W, invK, P, color, warp, B, H, W, 0.1, 100.0), px * 28),
        ("cgt_warp_bwd", lambda: call("jp_cgt_warp_bwd", dpred, disp, H, W, invK, P, color, ddisp, dP, B, H, W, 0.1, 100.0, 0), px * 32)):
    ms = t(fn)
    print(f"{name:14s} {ms * 1e3:7.1f} us  {byt / 1e6:6.0f} MB  {byt / ms / 1e9:5.2f} TB/s", flush=True)
# the same warp kernels on a SMOOTH disparity (a trained network's; the rows above use per-pixel random disparity = the
# random-initialised network of bench.py: neighbouring pixels then sample source rows several lines apart)
yy, xx = torch.meshgrid(torch.linspace(0, 1, H, device=dev), torch.linspace(0, 1, W, device=dev), indexing="ij")
disp_s = (0.2 + 0.6 * yy + 0.05 * torch.sin(6.28 * xx)).expand(B, 1, H, W).contiguous()
for name, fn, byt in (
        ("cgt_warp_fwd smooth", lambda: call("jp_cgt_warp_fwd", disp_s, H, W, invK, P, color, warp, B, H, W, 0.1, 100.0), px * 28),
        ("cgt_warp_bwd smooth", lambda: call("jp_cgt_warp_bwd", dpred, disp_s, H, W, invK, P, color, ddisp, dP, B, H, W, 0.1, 100.0, 0), px * 32)):
    ms = t(fn)
    print(f"{name:20s} {ms * 1e3:7.1f} us  {byt / 1e6:6.0f} MB  {byt / ms / 1e9:5.2f} TB/s", flush=True)
