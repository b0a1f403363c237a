"""Input pipeline: samplers, device-side preprocessing, pinned asynchronous batch feeding.

CPU: the sampler classes against index sequences the REFERENCE's own sampler classes emitted (tests/golden/sampler.npz),
the host-built resampling tables + a numpy two-pass against Pillow's output (tests/golden/preprocess.npz), the
ColorJitter oracle's basic identities.  GPU: the HIP kernels against the same Pillow / reference fixtures (bit-exact),
ColorJitter against the torchvision restatement, and the double-buffered DeviceLoader feeding real train steps."""
import os

import numpy as np
import pytest
import torch

from jperceiver_amd import synthetic as syn
from jperceiver_amd.datasets import (DistributedGroupSampler, DistributedSampler, GroupSampler, collate,
                                     pil_resample_tables, ColorJitterParams)
from tests.golden_util import GOLDEN


class _DS:
    def __init__(self, flag):
        self.flag = np.asarray(flag, dtype=np.int64)

    def __len__(self):
        return len(self.flag)


def test_samplers_emit_the_reference_sequences():
    g = np.load(os.path.join(GOLDEN, "sampler.npz"))
    n = 0
    for name in ("one_group_103", "two_groups"):
        ds = _DS(g[f"{name}/flag"])
        for k in g.files:
            if not k.startswith(name + "/") or k.endswith("/flag"):
                continue
            kind, spec = k.split("/")[1], k.split("/")[2] if k.count("/") > 1 else ""
            if kind == "dgs":
                w, s_, e, r = (int(p[1:]) for p in spec.split("_"))
                smp = DistributedGroupSampler(ds, s_, w, r)
                smp.set_epoch(e)
                seq = list(iter(smp))
                assert len(seq) == len(smp)
            elif kind == "ds":
                w, sh, r = spec.split("_")
                smp = DistributedSampler(ds, int(w[1:]), int(r[1:]), shuffle=bool(int(sh[2:])))
                smp.set_epoch(2)
                seq = list(iter(smp))
            else:
                np.random.seed(11)
                seq = [int(v) for v in iter(GroupSampler(ds, 4))]
            np.testing.assert_array_equal(np.asarray(seq, np.int64), g[k], err_msg=k)
            n += 1
    assert n == 78
    # every rank's block is made of whole per-GPU batches of ONE group, the ranks partition the (padded) epoch
    ds = _DS(g["two_groups/flag"])
    seen = []
    for r in range(4):
        smp = DistributedGroupSampler(ds, 3, 4, r)
        smp.set_epoch(1)
        seq = list(iter(smp))
        for i in range(0, len(seq), 3):
            assert len({int(ds.flag[j]) for j in seq[i:i + 3]}) == 1
        seen += seq
    assert set(seen) == set(range(len(ds)))


def _two_pass(img, OH, OW):
    H, W, C = img.shape
    bh, kh, _ = pil_resample_tables(W, OW)
    bv, kv, _ = pil_resample_tables(H, OH)
    tmp = img
    if W != OW:
        tmp = np.zeros((H, OW, C), np.uint8)
        for ox in range(OW):
            x0, n = bh[ox]
            ss = (1 << 21) + (img[:, x0:x0 + n, :].astype(np.int64) * kh[ox, :n][None, :, None]).sum(1)
            tmp[:, ox, :] = np.clip(ss >> 22, 0, 255)
    if H == OH:
        return tmp
    out = np.zeros((OH, OW, C), np.uint8)
    for oy in range(OH):
        y0, n = bv[oy]
        ss = (1 << 21) + (tmp[y0:y0 + n].astype(np.int64) * kv[oy, :n][:, None, None]).sum(0)
        out[oy] = np.clip(ss >> 22, 0, 255)
    return out


RESIZE = ["down", "up", "mixed", "same_w"]


def test_resample_tables_reproduce_pillow():
    g = np.load(os.path.join(GOLDEN, "preprocess.npz"))
    for name in RESIZE:
        H, W, OH, OW = g[f"resize/{name}/shape"]
        img = (syn.hash_uniform(21, ("pp", name), (H, W, 3)) * 256).astype(np.uint8)
        np.testing.assert_array_equal(_two_pass(img, OH, OW), g[f"resize/{name}/out"], err_msg=name)
    try:
        from PIL import Image
    except ImportError:
        return
    img = (syn.hash_uniform(3, "live", (41, 77, 3)) * 256).astype(np.uint8)
    np.testing.assert_array_equal(_two_pass(img, 64, 32), np.asarray(Image.fromarray(img).resize((32, 64), Image.LANCZOS)))


def test_color_jitter_oracle_identities_and_collate():
    from oracle import tv_restated as TV
    x = torch.rand(2, 3, 9, 11, generator=torch.Generator().manual_seed(0))
    for op in range(3):
        assert torch.allclose(TV.OPS[op](x, 1.0), x, atol=1e-6)
    assert torch.allclose(TV.adjust_hue(x, 0.0), x, atol=1e-5)
    assert torch.allclose(TV.adjust_hue(TV.adjust_hue(x, 0.3), -0.3), x, atol=1e-4)
    p = ColorJitterParams(generator=torch.Generator().manual_seed(4))
    assert sorted(p.order) == [0, 1, 2, 3] and all(0.8 <= f <= 1.2 for f in p.factors[:3]) and -0.1 <= p.factors[3] <= 0.1
    b = collate([{"a": torch.ones(2, 3), "k": np.zeros((4,), np.float32)} for _ in range(3)])
    assert b["a"].shape == (3, 2, 3) and b["k"].shape == (3, 4)


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_device_resize_is_bit_exact_pillow():
    from jperceiver_amd.datasets import DevicePreprocessor
    g = np.load(os.path.join(GOLDEN, "preprocess.npz"))
    pre = DevicePreprocessor(64, 64, "cuda")
    for name in RESIZE:
        H, W, OH, OW = (int(v) for v in g[f"resize/{name}/shape"])
        img = (syn.hash_uniform(21, ("pp", name), (H, W, 3)) * 256).astype(np.uint8)
        batch = torch.from_numpy(np.stack([img, img[::-1].copy()])).cuda()
        f, u8 = pre.resize_u8(batch, OH, OW, want_u8=True)
        np.testing.assert_array_equal(u8[0].cpu().numpy(), g[f"resize/{name}/out"], err_msg=name)
        ref = torch.from_numpy(g[f"resize/{name}/out"]).permute(2, 0, 1).float() / 255.0             # ToTensor
        assert torch.equal(f[0].cpu(), ref), name
        assert torch.equal(f[1].cpu(), u8[1].cpu().permute(2, 0, 1).float() / 255.0)
    img = (syn.hash_uniform(21, ("pp", "chain"), (80, 200, 3)) * 256).astype(np.uint8)
    full, full8 = pre.resize_u8(torch.from_numpy(img[None]).cuda(), 38, 124, want_u8=True)
    np.testing.assert_array_equal(full8[0].cpu().numpy(), g["resize/chain/full"])
    net, net8 = pre.resize_u8(full8, 64, 64, want_u8=True)
    np.testing.assert_array_equal(net8[0].cpu().numpy(), g["resize/chain/net"])


@pytest.mark.gpu
def test_device_topview_matches_reference_processing():
    from jperceiver_amd.datasets import DevicePreprocessor
    g = np.load(os.path.join(GOLDEN, "preprocess.npz"))
    pre = DevicePreprocessor(128, 128, "cuda")
    for name in ("sq", "rect", "up"):
        h, w, S = (int(v) for v in g[f"topview/{name}/shape"])
        lab = ((syn.hash_uniform(22, ("tv", name), (h, w)) > 0.55) * 255).astype(np.uint8)
        for mode, arr in (("L", lab[None]), ("RGB", np.stack([lab, lab, lab], -1)[None])):
            out = pre.topview(torch.from_numpy(arr).cuda(), S)
            assert out.shape == (1, 1, S, S)
            np.testing.assert_array_equal(out[0, 0].cpu().numpy().astype(np.uint8), g[f"topview/{name}/{mode}"], err_msg=name + mode)
        out = pre.topview(torch.from_numpy(lab[None]).cuda(), S, both=True)
        np.testing.assert_array_equal(out[0, 0].cpu().numpy().astype(np.uint8), g[f"topview_both/{name}"])


@pytest.mark.gpu
def test_device_color_jitter_matches_torchvision_restatement():
    from jperceiver_amd.datasets import DevicePreprocessor
    from oracle import tv_restated as TV
    pre = DevicePreprocessor(32, 48, "cuda")
    x = torch.rand(3, 3, 32, 48, generator=torch.Generator().manual_seed(7))
    x[0, :, :4] = 0.5                      # grey pixels: hue undefined (maxc == minc branch)
    x[1, :, :2] = 0.0
    for seed in range(6):
        p = ColorJitterParams(generator=torch.Generator().manual_seed(seed))
        got = pre.color_jitter_(x.clone().cuda(), p).cpu()
        ref = TV.color_jitter(x.clone(), p.order, p.factors)
        assert float((got - ref).abs().max()) < 2e-5, (seed, p.order)
    for op in range(4):                    # each op alone, strong factors
        p = ColorJitterParams()
        p.order, p.factors = [op], [1.7, 0.3, 1.9, -0.45]
        got = pre.color_jitter_(x.clone().cuda(), p).cpu()
        assert float((got - TV.OPS[op](x, p.factors[op])).abs().max()) < 2e-5, op


@pytest.mark.gpu
def test_device_loader_feeds_train_steps_in_order():
    from jperceiver_amd.datasets import DeviceLoader, DevicePreprocessor
    from jperceiver_amd.model import MONO
    from jperceiver_amd.apis import batch_processor, build_optimizer, Runner
    from jperceiver_amd.core import DistOptimizerHook
    from oracle import jp_oracle as J
    HW, B, FR = 256, 2, [0, -1, 1]
    FH, FW = 94, 311

    def raw_batch(i):
        d = syn.make_batch(B, HW, HW, FR, HW // 4, (FH, FW), "odometry", seed=40 + i)
        raw = {k: v for k, v in d.items() if k[0] not in ("color", "color_aug", "bothS", "bothD", "both_dynamic")}
        for f in FR:       # "camera" frames at native resolution, uint8 HWC
            raw[("color", f, -1)] = (torch.from_numpy(syn.hash_uniform(50 + i, ("raw", f), (B, 120, 400, 3))) * 256).to(torch.uint8)
        raw[("bothS", 0, 0)] = (d[("bothS", 0, 0)][:, 0] * 255).to(torch.uint8)
        raw[("bothD", 0, 0)] = (d[("bothD", 0, 0)][:, 0] * 255).to(torch.uint8)
        raw["tag"] = torch.full((1,), float(i))
        return raw

    pre = DevicePreprocessor(HW, HW, "cuda")
    gen = torch.Generator().manual_seed(0)
    loader = DeviceLoader((raw_batch(i) for i in range(4)), "cuda",
                          preprocess=lambda b: pre(b, FR, (FH, FW), generator=gen), depth=2)
    opt = J.default_opt(frame_ids=FR, imgs_per_gpu=B, height=HW, width=HW, occ_map_size=HW // 4, type="static", split="odometry")
    model = MONO.module_dict["Baseline"](opt)
    model.load_state_dict(syn.synth_state_dict(model.state_dict(), seed=0))
    model = model.cuda().train()
    runner = Runner(model, batch_processor, build_optimizer(model, dict(type="Adam", lr=1e-4, weight_decay=0)),
                    DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2)))
    tags, losses = [], []
    for batch in loader:
        assert batch[("color", 0, 0)].shape == (B, 3, HW, HW) and batch[("color", 0, 0)].is_cuda
        assert batch[("color", 0, -1)].shape == (B, 3, FH, FW) and batch[("bothS", 0, 0)].shape == (B, 1, HW // 4, HW // 4)
        assert float(batch[("color_aug", 1, 0)].min()) >= 0.0 and float(batch[("color_aug", 1, 0)].max()) <= 1.0
        tags.append(int(batch.pop("tag")[0]))
        losses.append(runner.train_iter(batch)["log_vars"]["loss"])
    assert tags == [0, 1, 2, 3] and all(np.isfinite(losses))


@pytest.mark.gpu
def test_color_jitter_is_decided_per_item_and_drawn_per_frame():
    """ADVICE r02 / mono_dataset.py:202,338-339,153-156: `do_color_aug` is one coin per ITEM; the `ColorJitter` object the
    reference passes re-draws order + factors per call, i.e. per FRAME.  Un-augmented items keep color_aug == color;
    every augmented (item, frame) equals the torchvision restatement applied with ITS parameters."""
    from jperceiver_amd.datasets import DevicePreprocessor
    from oracle import tv_restated as TV
    H, W, FR, N = 32, 48, [0, -1, 1], 4
    pre = DevicePreprocessor(H, W, "cuda")
    raw = {("color", f, -1): (torch.from_numpy(syn.hash_uniform(60, ("rawj", f), (N, 40, 60, 3))) * 256).to(torch.uint8).cuda()
           for f in FR}
    out = pre(raw, FR, (36, 54), do_color_aug=[True, False, True, False], generator=torch.Generator().manual_seed(2))
    P = pre.last_jitter
    assert set(P) == {(i, f) for i in (0, 2) for f in FR}
    assert len({(tuple(p.order), tuple(p.factors)) for p in P.values()}) == 6          # one draw per (item, frame)
    for f in FR:
        col, aug = out[("color", f, 0)].cpu(), out[("color_aug", f, 0)].cpu()
        for i in range(N):
            if (i, f) in P:
                ref = TV.color_jitter(col[i:i + 1].clone(), P[(i, f)].order, P[(i, f)].factors)
                assert float((aug[i:i + 1] - ref).abs().max()) < 2e-5, (i, f)
                assert float((aug[i] - col[i]).abs().max()) > 1e-3
            else:
                assert torch.equal(aug[i], col[i]), (i, f)
    # the coin itself: one per item from the generator; roughly half of many items are augmented
    gen = torch.Generator().manual_seed(5)
    raw1 = {k: v[:1].repeat(64, 1, 1, 1) for k, v in raw.items()}
    pre(raw1, FR, (36, 54), generator=gen)
    n_aug = len({i for i, _ in pre.last_jitter})
    assert 16 <= n_aug <= 48
    # jitter="per_item": the item's frames share one draw
    pre(raw, FR, (36, 54), do_color_aug=True, generator=torch.Generator().manual_seed(3), jitter="per_item")
    for i in range(N):
        assert len({id(pre.last_jitter[(i, f)]) for f in FR}) == 1
    assert len({id(p) for p in pre.last_jitter.values()}) == N
