import os, sys
sys.path.insert(0, "/root/repo")
import torch
from jperceiver_amd._lib import call
B, H, W = 8, 1024, 1024
dev = "cuda"
dpred = torch.rand(B, 3, H, W, device=dev)
K = torch.tensor([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * H, 0.5 * H, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]], device=dev).repeat(B, 1, 1)
invK = torch.linalg.inv(K).contiguous()
color = torch.rand(B, 3, H, W, device=dev)
ddisp = torch.zeros(B, 1, H, W, device=dev)
dP = torch.zeros(B, 12, device=dev, dtype=torch.float64)
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for hs, tx, rot, acc, smooth in [(1024, 0.1, 0.0, 0, False), (1024, 0.1, 0.0, 1, False), (512, 0.1, 0.0, 0, False), (1024, 0.5, 0.05, 0, False), (1024, 0.1, 0.0, 0, True), (128, 0.3, 0.02, 1, True)]:
    disp = torch.rand(B, 1, hs, hs, device=dev)
    if smooth:
        disp = torch.nn.functional.avg_pool2d(disp, 9, 1, 4).contiguous()
    T = torch.eye(4, device=dev).repeat(B, 1, 1); T[:, 0, 3] = tx
    c, s = torch.cos(torch.tensor(rot)), torch.sin(torch.tensor(rot))
    T[:, 0, 0] = c; T[:, 0, 2] = s; T[:, 2, 0] = -s; T[:, 2, 2] = c
    P = (K @ T)[:, :3].contiguous()
    us = t(lambda: call("jp_cgt_warp_bwd", dpred, disp, hs, hs, invK, P, color, ddisp, dP, B, H, W, 0.1, 100.0, acc))
    warp = torch.empty_like(color)
    uf = t(lambda: call("jp_cgt_warp_fwd", disp, hs, hs, invK, P, color, warp, B, H, W, 0.1, 100.0))
    print(f"hs={hs} tx={tx} rot={rot} acc={acc} smooth={smooth}: bwd {us:.1f} us  fwd {uf:.1f} us", flush=True)
