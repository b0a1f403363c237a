// Small-map split-bf16 patch launches (conv_p9sm.hip): plan + launch, called by conv.hip's dispatcher (not part of the ABI).
#pragma once
#include "jp_common.h"
#include "scale.h"

struct JpP9smPlan {
    int bmt, tr;          // M tile (64 | 128 rows) and its pixel-tile height (8 | 4 rows x 32 columns)
    int nst, sps, splits; // 16-channel (1x1: 32-channel) stages, stages per K slice, K slices (grid.z)
    long part_floats;     // caller scratch for the partial tiles when splits > 1
};
// rows = output rows of the GEMM (Cout forward, Cin dgrad), red = reduction channels, khw = 9 | 1
bool jp_p9sm_plan(int rows, int red, int N, int H, int W, int khw, JpP9smPlan* p);
void jp_p9sm_launch(const JpP9smPlan& p, const float* wp, const float* x, float* out, const float* bias, int act, int accumulate,
                    float* part, int rows, int red, int N, int H, int W, int khw, int reflect, int rev, const JpCall& st);
