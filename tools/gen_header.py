#!/usr/bin/env python
"""Regenerate include/jperceiver_hip.h from the extern "C" definitions in jperceiver_amd/csrc.
The per-function notes (which reference interface each entry point replaces) live here."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = "mono/model/mono_baseline/"
GROUPS = [
    ("Convolution (fp32 MFMA implicit GEMM) — replaces nn.Conv2d / ReflectionPad2d+Conv2d: "
     + R + "resnet.py:6-13,91; " + R + "layers.py:147-167; " + R + "depth_decoder.py:15-39; " + R + "pose_decoder.py:9-12; "
     + R + "layout_model.py:31-47,138-153; " + R + "CrossViewTransformer.py:30-42.  *_src3: the input is the channel concat of up to 3 "
     "tensors, each optionally stored at half resolution (fused F.interpolate(scale_factor=2,'nearest') + torch.cat, "
     + R + "depth_decoder.py:68,76-77).  pad_mode 0=zero 1=reflect; act 0=none 1=relu 2=leaky_relu(0.01) 3=sigmoid.  ws / ws_state: caller scratch for the packed weights "
     "(jp_conv2d_ws_floats) and whether it already holds this layer's pack (1) or must be packed by this call (0); packs recorded with "
     "jp_pack_record_begin/end can be refreshed for the whole model by one jp_pack_replay launch per step (job table in device memory; every job's `begin` is the running sum of the totals rounded up to a multiple of 4, "
     "total_elems the rounded grand total: the replay kernel works on groups of 4 consecutive elements; generic_elems: where the jobs of the "
     "LDS-staged split-pack kernel -- jp_pack_mode_is_split(job.mode) -- begin when the caller put them at the tail of the table, 0 = not sorted).  "
     "bn_stats / bn_stats_parts (forward): optional scratch of jp_conv2d_fwd_bn_stats_floats floats for a convolution that feeds a train-mode "
     "BatchNorm (" + R + "resnet.py:29-45; no bias, no activation): the 4-wave 3x3 patch kernels leave per-channel partial sums of y and y^2 "
     "there and *bn_stats_parts (host int) = the partials per channel, to be handed to jp_bn_train_fwd (conv_stats / conv_parts) in place of "
     "its own statistics pass over y; 0 = the kernel that ran the layer does not (NULL: not wanted).  "
     "ARITHMETIC: fp32 in / out / accumulate.  The patch kernels (3x3, 1x1, 7x7 stem, iconv; JP_P9S / JP_W9S / JP_P9US / JP_P9SD / JP_P9S2 / JP_P7S, default on) "
     "form each fp32 product on the 16-bit matrix pipe.  Default build (JP_NS = 2, jp_split_scheme() == 2): 3 fp16 products a0 b0 + a0 b1 + a1 b0 "
     "of two-way fp16 splits of both operands, each operand TENSOR scaled by the power of two that puts its largest magnitude into "
     "[2^14, 2^15) (csrc/igemm_p9s.h:jp_split2h, csrc/scale.hip; magnitudes: the jp_amax* group below).  Operands are carried to 2^-23 "
     "relative for elements within 2^-17 of their tensor's largest, to 2^-40 OF THAT LARGEST below; the term left out is <= 2^-22 |a b|.  "
     "Measured against float64 on layer data the rms error is 0.62-1.00 x the exact-fp32 MFMA kernels' (profiles/r05_fp16x2_accuracy_vs_f64.md; "
     "tests/test_split_accuracy_gpu.py holds it to 1.25 x and every output to 2^-19 sum|a||b|; tools/split_study.py: the split error is 3-4 x "
     "below the fp32 accumulation's own rounding).  It is an error relative to the tensor's largest magnitude, not to each element: outputs "
     "that depend only on values > 2^29 below their tensor's largest lose relative precision.  MEASURED LIMITS on the GPU kernels "
     "(tests/test_split_accuracy_gpu.py::test_split_accuracy_where_one_scale_per_tensor_differs_from_fp32, profiles/r06_split_outliers.md; "
     "8 x 128 -> 128 3x3 @64^2 next to the exact-fp32 kernels): one activation 10^6 x the median costs the outputs that do NOT read it 1.6 x "
     "(forward) / 2.7 x (weight gradient) the rms error of the fp32 accumulation; a channel whose data all sits 2^20 below its tensor's "
     "largest has ITS weight gradients carried to 7e-6 relative rms (fp32: 3e-7); log-normal gradients (sigma 3 / 4), 10^4 outliers, a "
     "filter 2^12 above the rest: no further from float64 than the exact-fp32 kernels.  RANGE EDGES "
     "(test_split_range_edges_match_documented_behaviour): Inf / NaN inputs and finite ones with |x| >= 2^100 take no part in the scale and "
     "make exactly the outputs that read them NaN (an fp32 convolution yields +-Inf for an Inf input); everything else is bit-identical to "
     "the run without them.  Tensors of tiny values are lifted by their scale (2^-120 inputs: full accuracy).  The 7x7 stem and the stride-2 "
     "weight gradient (JP_P7S, W9S2) and a -DJP_NS=3 build use 6 bf16 products of exact 3-way bf16 splits (csrc/igemm_p9s.h:jp_split3: error "
     "<= the fp32 FMA chain's for 2^-109 <= |x| <= 3.3895e38; Inf -> NaN).  JP_P9S=0 JP_W9S=0 JP_P9US=0 JP_P9SD=0 JP_P9S2=0 selects the "
     "exact-fp32 MFMA kernels, which have none of these edges.",
     ["jp_conv2d_fwd", "jp_conv2d_fwd_src3", "jp_conv2d_dgrad", "jp_conv2d_dgrad_src3", "jp_conv2d_dgrad_src3_split_floats", "jp_conv2d_dgrad_src3_ok", "jp_conv2d_up_head_ok", "jp_conv2d_wgrad", "jp_conv2d_wgrad_src3", "jp_conv2d_ws_floats", "jp_conv2d_fwd_bn_stats_floats", "jp_conv2d_fwd_split_floats", "jp_conv2d_dgrad_split_floats", "jp_conv2d_wgrad_ws_floats", "jp_conv2d_wgrad_src3_ws_floats", "jp_channel_sum_ws_floats", "jp_channel_sum", "jp_pack_job_bytes", "jp_pack_record_begin", "jp_pack_record_end", "jp_pack_mode_is_split", "jp_pack_replay"]),
    ("Operand scales of the fp16 split kernels (csrc/scale.hip, csrc/igemm_p9s.h:jp_split2h) -- no counterpart in the reference: plumbing of "
     "the arithmetic above, and since ABI version 3 entirely in the entry points' own ARGUMENTS (no library-owned device memory, no state "
     "between calls).  A kernel that forms its fp32 products from two fp16 splits per operand reads the operand tensor's largest "
     "magnitude from device memory.  A magnitude lives in a SLOT of jp_amax_slot_floats() floats (512: 32 words one cache line apart -- "
     "producers spread their atomics over them, the kernels take the maximum; every `amax_*` / `out` argument is one).  "
     "INPUT magnitudes -- amax_x / amax_x0..2 (forward, weight gradient), amax_dy (dgrad, weight gradient) of the jp_conv2d_* entry points: "
     "a slot that holds max |operand| on `stream` (from jp_amax, or written by the producer of the tensor, below), or NULL; for every "
     "operand passed as NULL the call reduces the tensor itself (one extra read) into `amax_ws`, caller scratch of "
     "jp_conv2d_amax_ws_floats() floats that need not be initialised and may be reused by the next call on the same stream.  A call with "
     "a NULL operand magnitude AND amax_ws == NULL is a bad argument, and so is a forward call with more than one source and no amax_ws "
     "(its kernel reads ONE magnitude, the largest of the sources', folded into the scratch by a 64-lane launch); the -DJP_NS=3 build "
     "ignores all of them.  A magnitude that is too "
     "SMALL overflows fp16 (Inf / NaN outputs); one that is too large only costs precision (an upper bound is enough: max-pool and ReLU "
     "outputs may reuse their input's slot).  "
     "OUTPUT magnitudes -- a producer folds max |tensor it writes| into the slot with atomic maxima (the slot must hold 0, or an earlier "
     "maximum to extend): jp_bn_train_fwd / jp_bn_relu_pool_fwd amax_y, jp_sum_n amax_out, jp_bn_train_bwd / jp_act_bwd / jp_act_bwd_bias / "
     "jp_maxpool_bwd amax_dx (NULL: not wanted), and the convolutions -- jp_conv2d_fwd* amax_y, jp_conv2d_dgrad amax_dx, "
     "jp_conv2d_dgrad_src3 amax_dx0 (the first source's gradient): folded in by the kernel's epilogue (and, for the backward of "
     "reflection-padded layers, by the border fold) when a patch kernel runs the layer, in which case *amax_y_done / *amax_dx_done "
     "(host int, may be NULL) is set to 1 before the call returns; 0 = the kernel chosen for this shape does not report "
     "(use jp_amax on the tensor if the magnitude is needed).  "
     "jp_amax writes max |x[0..n)| (Inf / NaN / |x| >= 2^100 excluded) to *out (jp_amax_into: max(*out, that) -- *out pre-zeroed by the "
     "caller, no memset).  jp_split_scheme: 2 = this build's patch kernels use the fp16 two-way split (magnitudes are read), 3 = the "
     "bf16 three-way split (no operand scales).",
     ["jp_amax_slot_floats", "jp_conv2d_amax_ws_floats", "jp_amax", "jp_amax_into", "jp_split_scheme"]),
    ("Train-mode BatchNorm2d (+fused residual add / ReLU) — " + R + "resnet.py:21-24,41-45,92; " + R + "layout_model.py:146,152. "
     "ws = jp_bn_ws_doubles(N, C, HW) doubles of caller scratch.  n_updates = number of momentum updates of the running stats (2 for the layout "
     "branch the reference evaluates twice, " + R + "net.py:73-74).  jp_bn_relu_pool_*: the ResNet stem tail bn1 -> relu -> MaxPool2d(3, 2, 1) ("
     + R + "resnet.py:92-94) in one pass each way, for callers that do not read the normalised map itself (the decoders never do): "
     "pooled output + argmax byte in forward, the convolution-output gradient straight from the pooled gradient in backward.",
     ["jp_bn_ws_doubles", "jp_bn_train_fwd", "jp_bn_train_bwd", "jp_bn_eval_fwd", "jp_bn_relu_pool_fwd",
      "jp_bn_relu_pool_bwd_ws_doubles", "jp_bn_relu_pool_bwd"]),
    ("Pooling / resampling / elementwise — MaxPool2d " + R + "resnet.py:94, " + R + "layers.py:191, " + R + "layout_model.py:84; "
     "nearest upsample " + R + "layers.py:110; torch.cat; Dropout multiply " + R + "depth_decoder.py:52-53; F.interpolate bilinear "
     + R + "net.py:196,632,692 and area " + R + "net.py:762.",
     ["jp_maxpool_fwd", "jp_maxpool_bwd", "jp_upsample2x_fwd", "jp_upsample2x_bwd", "jp_copy_channels", "jp_axpby", "jp_sum_n", "jp_mul",
      "jp_affine", "jp_act_fwd", "jp_act_bwd", "jp_act_bwd_bias_ws_floats", "jp_act_bwd_bias", "jp_mul_bcast_c", "jp_mul_bcast_c_bwd_s", "jp_bilinear_fwd", "jp_bilinear_bwd",
      "jp_area_downsample", "jp_fill", "jp_warp_perspective", "jp_softmax_c2", "jp_disp_to_depth",
      "jp_scale_label_assemble", "jp_fill_convex_poly"]),
    ("Small dense algebra of the BEV branch — CVP MLP " + R + "CycledViewProjection.py:33-38,54-67; CCT attention "
     + R + "CrossViewTransformer.py:14-24,53-65,77-88; PoseDecoder spatial mean " + R + "pose_decoder.py:22-23.",
     ["jp_gemm_strided_batched", "jp_bias_act_rows", "jp_colsum", "jp_colmax", "jp_colmax_bwd", "jp_gather_cols",
      "jp_gather_cols_bwd", "jp_spatial_mean", "jp_spatial_mean_bwd", "jp_bcast_matmul_fwd", "jp_bcast_matmul_bwd",
      "jp_ratio_finalize"]),
    ("CGT view synthesis + photometric losses — pose " + R + "net.py:704-756; Backproject/Project/grid_sample "
     + R + "layers.py:41-82, " + R + "net.py:690-702; SSIM " + R + "layers.py:85-107; reprojection + automask min "
     + R + "net.py:84-92,159-175.",
     ["jp_pose_fwd", "jp_pose_bwd", "jp_cgt_warp_fwd", "jp_cgt_warp_bwd", "jp_ssim_l1_fwd", "jp_ssim_l1_bwd",
      "jp_minreproj_fwd", "jp_scalar_finalize", "jp_ssim_map", "jp_backproject", "jp_project"]),
    ("Other losses — smoothness " + R + "net.py:182-190,758-786; CGT scale loss " + R + "net.py:193-211; IoU "
     + R + "dice_loss.py:31-81,293-331 + weighted CE " + R + "net.py:561,583 + boundary loss " + R + "boundary_loss.py:121-192 "
     "(jp_sdf = exact EDT + inner boundary on the GPU, no host round trip); cycle L1 " + R + "net.py:619-622.",
     ["jp_row_sum", "jp_smooth_fwd", "jp_smooth_bwd", "jp_scale_loss_fwd", "jp_scale_loss_bwd", "jp_layout_loss_fwd",
      "jp_layout_loss_bwd", "jp_sdf", "jp_l1_fwd", "jp_l1_bwd"]),
    ("Optimizer over flat arenas + RNG — clip_grads(max_norm=35)+Adam mono/core/utils/dist_utils.py:58-60, "
     "config/cfg_kitti_baseline_odometry_boundary_ce_iou_1024_20.py:69-70; Dropout / randn " + R + "depth_decoder.py:13, "
     + R + "net.py:163.",
     ["jp_sumsq_blocks", "jp_grad_sumsq_partials", "jp_sum_doubles", "jp_adam_clip_step", "jp_adam_clip_step_dev", "jp_rng_keep_mask",
      "jp_rng_normal", "jp_rng_keep_mask_dev", "jp_rng_normal_dev"]),
    ("Evaluation metrics as GPU reductions — mean_IU / mean_precision mono/core/evaluation/pixel_error.py:59-118 (confusion counts "
     "of argmax(logits) vs label); depth compute_errors + median scaling + Garg crop mono/core/evaluation/pixel_error.py:27-40, "
     "mono/core/evaluation/eval_hooks.py:147-179.",
     ["jp_confusion2", "jp_depth_eval_prepare", "jp_masked_median", "jp_depth_errors"]),
    ("Device-side input pipeline — MonoDataset.preprocess mono/datasets/mono_dataset.py:126-171 (PIL ANTIALIAS resize, bit-exact "
     "Pillow fixed-point resampler; ToTensor; ColorJitter in torchvision tensor arithmetic) and process_topview :417-431, applied "
     "to raw uint8 frames after one pinned async upload.",
     ["jp_resample_h_u8", "jp_resample_v_u8", "jp_u8_to_tensor", "jp_color_jitter_op", "jp_topview_u8"]),
    ("Library plumbing.  jp_profile_*: opt-in per-kernel HIP-event timing of the implicit-GEMM launches on the streams they "
     "are launched on (bench.py's roofline leg; never active in the train step).",
     ["jp_abi_version", "jp_last_error_string", "jp_set_last_error", "jp_profile_begin", "jp_profile_count", "jp_profile_end",
      "jp_profile_get"]),
]


def collect():
    protos = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "jperceiver_amd/csrc/*.hip")) + [os.path.join(ROOT, "jperceiver_amd/csrc/capi.cpp")]):
        src = open(f).read()
        for m in re.finditer(r'extern "C" ((?:const )?\w+\*?) (jp_\w+)\(([^)]*)\)\s*\{', src):
            ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
            protos[name] = f"{ret} {name}({args});"
    return protos


def main():
    protos = collect()
    out = ["/* jperceiver_hip.h — C ABI of libjperceiver_hip.so (MI355X / gfx950 only).",
           " *",
           " * GENERATED by tools/gen_header.py from jperceiver_amd/csrc — do not edit by hand.",
           " *",
           " * Drop-in boundary for the JPerceiver `Baseline` training step (SURVEY.md §8b): the reference has no",
           " * native layer of its own (pure PyTorch), so these entry points are what a ctypes binding inside",
           " * mono/model/mono_baseline/* would call in place of the ATen/cuDNN ops cited per group.",
           " *",
           " * Conventions: fp32 NCHW contiguous tensors; raw device pointers; explicit int dims; every call is",
           " * enqueued on `stream` (a hipStream_t passed as void*), never synchronises and never allocates —",
           " * the caller owns all buffers including scratch; no entry point keeps state for a later call (the only",
           " * thread-local data are the error string and the opt-in pack recorder / profiler).  Return 0 on success, <0 for a bad argument, >0 =",
           " * hipError_t; jp_last_error_string() describes the last failure on the calling thread.",
           " * `gout` arguments are device pointers to the upstream scalar gradient (NULL = 1.0).",
           " */",
           "#ifndef JPERCEIVER_HIP_H", "#define JPERCEIVER_HIP_H", "#include <stdint.h>", "",
           "#ifdef __cplusplus", 'extern "C" {', "#endif", ""]
    seen = set()
    for doc, names in GROUPS:
        out.append("/* " + "\n * ".join(_wrap(doc)) + " */")
        for n in names:
            out.append(protos[n])
            seen.add(n)
        out.append("")
    missing = set(protos) - seen
    assert not missing, f"ungrouped entry points: {missing}"
    out += ["#ifdef __cplusplus", "}", "#endif", "#endif /* JPERCEIVER_HIP_H */", ""]
    with open(os.path.join(ROOT, "include", "jperceiver_hip.h"), "w") as f:
        f.write("\n".join(out))
    print("wrote", len(protos), "prototypes")


def _wrap(s, width=108):
    words, lines, cur = s.split(), [], ""
    for w in words:
        if len(cur) + len(w) + 1 > width:
            lines.append(cur)
            cur = w
        else:
            cur = (cur + " " + w).strip()
    lines.append(cur)
    return lines


if __name__ == "__main__":
    main()
