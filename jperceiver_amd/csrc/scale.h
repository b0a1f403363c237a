// Operand scales of the two-way fp16 split kernels (igemm_p9s.h, JP_NS == 2): a tensor is scaled by the power of two that puts its
// LARGEST MAGNITUDE into [2^14, 2^15) before it is split, so the kernels need that magnitude on the device.
#pragma once
#include "jp_common.h"
// Device pointer to max |x[0 .. n)|, valid for kernels launched on `st` after this call: the caller's hint for this tensor
// (jp_amax_hint: the host computed it once with jp_amax and several convolutions read the tensor) or a fresh reduction launched here.
const float* jp_amax_of(const float* x, long n, hipStream_t st);
// max over up to three tensors (the iconv kernels' channel segments): one slot holding the largest of the three
const float* jp_amax_of3(const float* x0, long n0, const float* x1, long n1, const float* x2, long n2, hipStream_t st);
