#!/usr/bin/env python
"""Which multi-stream patterns does hipGraph capture accept?  python tools/debug/graph_streams_micro.py {nested|event|both|autograd}"""
import sys
import torch

mode = sys.argv[1]
x = torch.zeros(1 << 20, device="cuda")
side, swg, mwg = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
ys = []


def work(n=3):
    for _ in range(n):
        ys.append(x * 2.0 + 1.0)


class Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t * 1.0

    @staticmethod
    def backward(ctx, gr):
        body()
        return gr


def body():
    cur = torch.cuda.current_stream()
    work()
    side.wait_stream(cur)
    ev = torch.cuda.Event()
    with torch.cuda.stream(side):
        work()
        if mode in ("nested", "both", "autograd"):
            swg.wait_stream(side)
            with torch.cuda.stream(swg):
                work()
        work()
        if mode in ("event", "both", "autograd"):
            ev.record(side)
        work()
        if mode in ("nested", "both", "autograd"):
            side.wait_stream(swg)
    mwg.wait_stream(cur)
    with torch.cuda.stream(mwg):
        work()
    if mode in ("event", "both", "autograd"):
        cur.wait_event(ev)
    work()
    cur.wait_stream(mwg)
    cur.wait_stream(side)
    work()


torch.cuda.synchronize()
with torch.cuda.graph(g):
    if mode == "autograd":
        p = torch.ones(4, device="cuda", requires_grad=True)
        Fn.apply(p).sum().backward()
    else:
        body()
print(f"micro {mode}: capture ended", flush=True)
g.replay()
torch.cuda.synchronize()
print(f"micro {mode}: OK", flush=True)
