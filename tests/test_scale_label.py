"""CGT scale-label generation (get_scale_label_static / _dynamic / _both, net.py:212-476).

CPU (`-m "not gpu"`): the oracle restatement (oracle/jp_oracle.py + oracle/cv2_restated.py) against
tests/golden/scale_labels.npz, which tools/make_golden.py produced by running the REFERENCE's own
get_scale_label_static / get_scale_label_dynamic (third-party cv2 / torchgeometry / torchvision stubbed).
GPU (`-m gpu`): the product path (`Baseline.get_scale_label` -> jp_warp_perspective, jp_fill_convex_poly,
jp_scale_label_assemble through the C ABI) against that oracle and the same fixture, incl. the Argoverse
frame size 2056x2464 and polygons that leave the image (cv::clipLine path).

Rounding band: `.type_as(uint8)` of the bilinearly warped {0,1} road layout keeps a pixel iff the four
interpolation weights sum to >= 1.0 in fp32 — for all-road neighbourhoods that is decided by the last ulp of
the evaluating platform (PyTorch-CPU and PyTorch-CUDA already disagree there).  Pixels whose oracle layout
value lies in [1 - 4e-6, 1) are therefore accepted either way (measured: ~1.5 % of the support);
everything else — polygon, support, distance values — is compared exactly / at 1e-4 relative.
"""
import ast

import numpy as np
import pytest
import torch

from jperceiver_amd import synthetic as syn
from oracle import cv2_restated as cv2
from oracle import jp_oracle as J
from tests.golden_util import GOLDEN

CASES = ["static_odom", "dynamic_odom", "static_argo_small", "static_argo_full", "dynamic_argo_full"]


def _golden():
    import os
    return np.load(os.path.join(GOLDEN, "scale_labels.npz"), allow_pickle=False)


def _case(g, name):
    meta = ast.literal_eval(str(g[name + "/meta"]))
    occ, B = meta["occ"], meta["B"]
    opt = J.default_opt(imgs_per_gpu=B, height=occ * 4, width=occ * 4, occ_map_size=occ, type=meta["type"],
                        split=meta["split"])
    inp = syn.make_batch(B, occ * 4, occ * 4, [0], occ, tuple(meta["full_hw"]), meta["split"], seed=meta["seed"])
    return meta, opt, inp


def _geo_inside(pts, H, W):
    yy, xx = np.mgrid[:H, :W]
    pos = np.zeros((H, W), int)
    neg = np.zeros((H, W), int)
    n = len(pts)
    for k in range(n):
        ax, ay = pts[k]
        bx, by = pts[(k + 1) % n]
        cr = (bx - ax) * (yy - ay) - (by - ay) * (xx - ax)
        pos += cr > 0
        neg += cr < 0
    return ~((pos > 0) & (neg > 0))


# ------------------------------------------------------------------------------------------- CPU: oracle
def test_fill_convex_poly_restatement_basics():
    img = cv2.fillConvexPoly(np.zeros((20, 30, 3), np.uint8), np.array([[3, 4], [3, 10], [12, 10], [12, 4]]).reshape(-1, 1, 2),
                             (0, 255, 255), 1)
    gray = cv2.cvtColor(img, cv2.COLOR_RGB2GRAY)
    assert gray.max() == 179 and int((gray > 0).sum()) == 7 * 10          # inclusive rectangle; 0.587*255+0.114*255
    assert (img[..., 0] == 0).all() and set(np.unique(img[..., 1])) == {0, 255}
    # a convex quad: OpenCV's fill + 4-connected outline is a (thin) superset of the closed polygon
    pts = np.array([[60, 10], [90, 40], [50, 70], [20, 35]], np.int32)
    m = cv2.cvtColor(cv2.fillConvexPoly(np.zeros((80, 110, 3), np.uint8), pts.reshape(-1, 1, 2), (0, 255, 255), 1), 7) > 0
    geo = _geo_inside(pts, 80, 110)
    assert not (geo & ~m).any() and 0 < int((m & ~geo).sum()) < 0.1 * geo.sum()
    # 4-connected line: consecutive pixels differ in exactly one coordinate by one, dx+dy+1 pixels
    line = cv2.line_points(100, 100, (5, 7), (40, 20), 4)
    assert len(line) == 35 + 13 + 1 and line[0] == (5, 7) and line[-1] == (40, 20)
    assert all(abs(a[0] - b[0]) + abs(a[1] - b[1]) == 1 for a, b in zip(line, line[1:]))
    # drawn left-to-right whatever the argument order
    assert cv2.line_points(100, 100, (40, 20), (5, 7), 4)[0] == (5, 7)
    # clipping: a polygon that leaves the image still fills up to the border
    big = np.array([[150, 90], [-20, 90], [10, 30], [120, 30]], np.int32)
    m2 = cv2.cvtColor(cv2.fillConvexPoly(np.zeros((60, 100, 3), np.uint8), big.reshape(-1, 1, 2), (0, 255, 255), 1), 7) > 0
    assert m2[59].all() and m2[30, 10:100].all() and not m2[29].any()


@pytest.mark.parametrize("name", CASES)
def test_oracle_scale_label_matches_reference_fixture(name):
    g = _golden()
    meta, opt, inp = _case(g, name)
    M = J.scale_label_homography(opt, inp)
    assert int(g[name + "/lineType"]) == 1
    np.testing.assert_array_equal(J._assumption_quad(opt, M), g[name + "/pts"])
    lab = J.make_scale_label(opt, inp)
    assert int((~torch.isfinite(lab)).sum()) == int(g[name + "/n_nonfinite"])
    sup = np.unpackbits(g[name + "/support"])[: lab.numel()].reshape(lab.shape).astype(bool)
    lab0 = torch.nan_to_num(lab, nan=0.0, posinf=0.0, neginf=0.0)
    if meta["type"] == "dynamic":        # no uint8 cast of an interpolated layout -> host-independent
        np.testing.assert_array_equal((lab > 0).numpy(), sup)
        assert float(lab0.double().sum()) == pytest.approx(float(g[name + "/sum"]), rel=1e-6)
        np.testing.assert_allclose(lab0.double().sum(-1).numpy(), g[name + "/rowsum"], rtol=1e-5, atol=1e-4)
    else:
        # static: bit-identical on the host that generated the fixture; another x86 host's PyTorch rounds the
        # interpolation weights of ~1 % of the support pixels the other way (module docstring)
        assert int(((lab > 0).numpy() ^ sup).sum()) <= 0.03 * int(g[name + "/nnz"])
        assert float(lab0.double().sum()) == pytest.approx(float(g[name + "/sum"]), rel=0.03)


# ------------------------------------------------------------------------------------------- GPU: product path
def _gpu_mask(pts, H, W):
    from jperceiver_amd._lib import call
    q = torch.as_tensor(np.asarray(pts, np.int32)).cuda().contiguous()
    mask = torch.full((H, W), 7, device="cuda", dtype=torch.uint8)
    call("jp_fill_convex_poly", q, len(pts), mask, H, W)
    return mask.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("pts,H,W", [
    ([[845, 365], [403, 365], [451, 323], [789, 323]], 375, 1242),          # KITTI odometry calibration
    ([[2741, 2291], [-62, 2291], [298, 1940], [2322, 1940]], 2056, 2464),   # Argoverse: leaves the frame on 3 sides
    ([[709, 593], [21, 593], [109, 490], [586, 490]], 514, 616),
    ([[60, 10], [90, 40], [50, 70], [20, 35]], 80, 110),                    # diamond: both scan edges slanted
    ([[3, 4], [3, 10], [12, 10], [12, 4]], 20, 30),
    ([[5, 5], [50, 6], [30, 5]], 40, 64),                                   # thin sliver triangle
    ([[10, -30], [200, 20], [120, 300], [-40, 100]], 128, 160),             # every vertex outside
    ([[300, 300], [320, 300], [310, 330]], 100, 100),                       # entirely outside
])
def test_fill_convex_poly_kernel_pixel_exact(pts, H, W):
    ref = cv2.cvtColor(cv2.fillConvexPoly(np.zeros((H, W, 3), np.uint8), np.asarray(pts, np.int32).reshape(-1, 1, 2),
                                          (0, 255, 255), 1), cv2.COLOR_RGB2GRAY) > 0
    got = _gpu_mask(pts, H, W)
    assert set(np.unique(got)) <= {0, 1}
    np.testing.assert_array_equal(got.astype(bool), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_get_scale_label_product_path(name):
    from jperceiver_amd.model import MONO
    from jperceiver_amd.apis import change_input_variable
    g = _golden()
    meta, opt, inp = _case(g, name)
    ref, zw, lw, tri = {"static": J.scale_label_static, "dynamic": J.scale_label_dynamic}[meta["type"]](opt, inp, True)
    model = MONO.module_dict["Baseline"](opt)        # weights are irrelevant here
    for via_host_matrices in (True, False):
        d = {k: v.clone() for k, v in inp.items()}
        if via_host_matrices:
            d = change_input_variable(d, opt=model.opt)     # the trainer's path: homography + quad from the CPU batch
        else:
            d = {k: v.cuda() for k, v in d.items()}         # calibration already on the device
        lab = model.get_scale_label(d).cpu()
        assert lab.shape == ref.shape
        finite = torch.isfinite(ref) & torch.isfinite(lab)
        # rounding band of the uint8 cast (module docstring): either outcome is accepted there
        amb = torch.zeros_like(finite)
        if lw is not None:
            amb = (lw >= 1 - 4e-6) & (lw < 1) & (tri > 0)
        chk = finite & ~amb
        sup_ref, sup_got = (ref > 0) & chk, (lab > 0) & chk
        n_bad = int((sup_ref ^ sup_got).sum())
        assert n_bad <= 2e-4 * max(1, int(sup_ref.sum())), f"{name}: {n_bad} support mismatches of {int(sup_ref.sum())}"
        both = sup_ref & sup_got
        err = ((lab - ref).abs() / ref.abs().clamp_min(1e-3))[both]
        assert float(err.max()) < 1e-4, f"{name}: distance values differ by {float(err.max())}"
        assert int(amb.sum()) <= 0.03 * max(1, int((ref > 0).sum()))
        # and against the reference-generated fixture.  Its support was computed on the build container's CPU, whose
        # rounding band sits at other pixels than this host's (observed: 1.1 % of the support differs between the
        # two x86 hosts for the SAME PyTorch code) -> only the band-sized tolerance is meaningful here; the strict
        # comparison is the one above, against the oracle evaluated on this host.
        sup_fix = torch.from_numpy(np.unpackbits(g[name + "/support"])[: lab.numel()].reshape(lab.shape).astype(bool))
        assert int(((sup_fix ^ (lab > 0)) & finite).sum()) <= 0.03 * int(g[name + "/nnz"])
        lab0 = torch.where(chk, torch.nan_to_num(lab, nan=0.0, posinf=0.0, neginf=0.0), torch.zeros_like(lab))
        ref0 = torch.where(chk, torch.nan_to_num(ref, nan=0.0, posinf=0.0, neginf=0.0), torch.zeros_like(ref))
        assert float(lab0.double().sum()) == pytest.approx(float(ref0.double().sum()), rel=2e-4)
