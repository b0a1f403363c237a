#!/usr/bin/env python
"""hipGraph capture: nested fork variants.  python tools/debug/graph_streams_micro2.py {flatjoin|mainfirst|mainfirst_flat|twice}"""
import sys
import torch

mode = sys.argv[1]
x = torch.zeros(1 << 20, device="cuda")
side, swg = torch.cuda.Stream(), torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
ys = []


def work(n=3):
    for _ in range(n):
        ys.append(x * 2.0 + 1.0)


def body():
    cur = torch.cuda.current_stream()
    work()
    side.wait_stream(cur)
    if mode.startswith("mainfirst"):
        swg.wait_stream(cur)            # the companion enters the capture from the ORIGIN stream ...
    with torch.cuda.stream(side):
        work()
        swg.wait_stream(side)           # ... and then takes its dependency on the side stream
        with torch.cuda.stream(swg):
            work()
        work()
        if mode in ("mainfirst", "twice"):
            side.wait_stream(swg)       # join into the side stream
        if mode == "twice":             # a second fork / join round of the same pair
            work()
            swg.wait_stream(side)
            with torch.cuda.stream(swg):
                work()
            side.wait_stream(swg)
    if mode in ("flatjoin", "mainfirst_flat"):
        cur.wait_stream(swg)            # join straight into the origin stream
    cur.wait_stream(side)
    work()


torch.cuda.synchronize()
with torch.cuda.graph(g):
    body()
print(f"micro2 {mode}: capture ended", flush=True)
g.replay()
torch.cuda.synchronize()
print(f"micro2 {mode}: OK", flush=True)
