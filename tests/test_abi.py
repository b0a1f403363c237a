"""CPU-side checks of the drop-in boundary: the shared library loads without a GPU, exports every symbol
include/jperceiver_hip.h declares (and nothing undeclared), argument validation fails loudly, and the
host-side mirror keeps the reference's module names / state-dict keys."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

from jperceiver_amd import _lib


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    protos = _lib.parse_header()
    assert len(protos) >= 60
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (jp_\w+)", out))
    assert exported == set(protos), (exported ^ set(protos))
    assert L.fn["jp_abi_version"]() == 3


def test_bad_arguments_are_rejected_without_a_gpu():
    L = _lib.lib()
    rc = L.fn["jp_conv2d_fwd"](None, None, None, None, 1, 1, 4, 4, 1, 3, 1, 1, 0, 0, None, 0, None, None, None, None, None, None, None, None)
    assert rc == -1 and "null" in L.last_error()
    rc = L.fn["jp_conv2d_dgrad"](ctypes.c_void_p(8), ctypes.c_void_p(8), ctypes.c_void_p(8), 1, 1, 4, 4, 1, 5, 1, 1, 1, 0, None, 0, None,
                                 None, None, None, None, None)
    assert rc == -1
    if L.fn["jp_split_scheme"]() == 2:
        # ABI 3: an operand magnitude that is neither passed nor reducible (amax_ws == NULL) is a bad argument, not a hidden allocation
        done = ctypes.c_int(7)
        rc = L.fn["jp_conv2d_fwd"](ctypes.c_void_p(8), ctypes.c_void_p(8), None, ctypes.c_void_p(8), 1, 64, 32, 32, 64, 3, 1, 1, 0, 0, None, 0,
                                   None, None, None, ctypes.addressof(done), None, None, None, None)
        assert rc == -1 and "amax_ws" in L.last_error() and done.value == 0


def test_library_owns_no_device_memory_and_no_cross_call_state():
    """"the library never allocates or frees device memory ... stateless and re-entrant".  The sources may hold thread-local
    data only for the error string, the opt-in pack recorder and the opt-in profiler; no hipMalloc / hipFree anywhere; and the ABI has no
    entry point that parks an argument for a later call (the jp_amax_hint / jp_amax_out side channel of ABI version 2)."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jperceiver_amd", "csrc")
    allowed_tls = ("g_err", "g_pack_rec", "g_prof")
    for f in sorted(os.listdir(root)):
        if not f.endswith((".hip", ".h", ".cpp")):
            continue
        for ln, line in enumerate(open(os.path.join(root, f)), 1):
            code = line.split("//")[0]
            assert not re.search(r"\bhip(Malloc|Free|MallocAsync|FreeAsync|HostMalloc|MallocManaged)\b", code), f"{f}:{ln}: {line.strip()}"
            if "thread_local" in code:
                assert any(a in code for a in allowed_tls), f"{f}:{ln}: {line.strip()}"
    protos = _lib.parse_header()
    for gone in ("jp_amax_hint", "jp_amax_hint_clear", "jp_amax_out", "jp_amax_out_done"):
        assert gone not in protos
    for name in ("jp_conv2d_fwd", "jp_conv2d_fwd_src3", "jp_conv2d_dgrad", "jp_conv2d_dgrad_src3", "jp_conv2d_wgrad", "jp_conv2d_wgrad_src3"):
        args = [a for _, a in protos[name][1]]
        assert "amax_ws" in args and any(a.startswith("amax_") and a != "amax_ws" for a in args), name


def test_call_refuses_cpu_tensors():
    # the tensor check fires before any stream is touched, so this also holds on a GPU-less machine
    with pytest.raises(_lib.JPerceiverHipError):
        _lib.call("jp_fill", torch.zeros(4), 4, 0.0)


def test_model_surface_matches_reference_names():
    from jperceiver_amd.model import MONO
    from oracle import jp_oracle as J
    opt = J.default_opt(height=256, width=256, occ_map_size=64, imgs_per_gpu=2, type="Argo_both", split="argo")
    net = MONO.module_dict["Baseline"](opt)
    for name in ("DepthEncoder", "DepthDecoder", "PoseEncoder", "PoseDecoder", "LayoutEncoder", "CycledViewProjection",
                 "CrossViewTransformer", "LayoutDecoder", "LayoutTransformDecoder", "CycledViewProjectionB",
                 "CrossViewTransformerB", "LayoutDecoderB", "LayoutTransformDecoderB", "ssim", "backproject", "project_3d"):
        assert hasattr(net, name), name
    sd = net.state_dict()
    shapes = J.state_shapes(64)
    assert set(sd) == set(shapes) and len(sd) == 766
    assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in sd)
    assert sum(p.numel() for p in net.parameters()) == sum(int(torch.tensor(shapes[k]).prod()) for k in shapes if not J.is_buffer(k))
    with pytest.raises(RuntimeError):
        net({("color_aug", 0, 0): torch.zeros(1, 3, 256, 256)})      # product path has no CPU fallback


def test_no_oracle_import_in_product_path():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dp, _, files in os.walk(os.path.join(root, "jperceiver_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "/root/reference" not in src, f
