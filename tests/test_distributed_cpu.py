"""world_size-2 gloo tests (CPU) of the data-parallel path: the bucketed SUM all-reduce over the flat gradient
arena + 1/world averaging equals the mean of the ranks' gradients, parameters start identical after the
wrap-time broadcast, and each rank draws a different synthetic shard."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from jperceiver_amd import synthetic as syn


class _Arena:
    def __init__(self, n):
        self.grads = torch.zeros(n)
        self.live_numel = n - 7        # a dead tail must stay untouched


class _M(nn.Module):
    def __init__(self):
        super().__init__()
        self.w = nn.Parameter(torch.zeros(5))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from jperceiver_amd.apis import init_dist, DataParallelShell
    from jperceiver_amd.core.dist_utils import allreduce_grads
    init_dist("pytorch", backend="gloo")
    m = _M()
    with torch.no_grad():
        m.w.fill_(float(rank + 1))
    shell = DataParallelShell(m)                       # broadcast from rank 0
    n = 3 * 1024 * 1024 // 4 + 1234                    # spans several 1-MiB buckets
    m._jp_arena = _Arena(n)
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    m._jp_arena.grads.copy_(g)
    allreduce_grads(shell, bucket_size_mb=1, average_in_place=False)
    res = m._jp_arena.grads / world
    live = m._jp_arena.live_numel
    exp = torch.arange(n, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
    ok = torch.allclose(res[:live], exp[:live]) and torch.equal(m._jp_arena.grads[live:], g[live:])
    batch = syn.make_batch(1, 64, 64, (0, -1, 1), 16, (20, 30), "argo", seed=1, rank=rank)
    q.put((rank, bool(ok), float(m.w[0]), float(batch[("color", 0, 0)].sum())))
    dist.destroy_process_group()


def test_allreduce_broadcast_and_sharding_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
    assert all(r[1] for r in res), res
    assert res[0][2] == res[1][2] == 1.0            # rank 0's parameters everywhere
    assert res[0][3] != res[1][3]                   # disjoint synthetic shards
