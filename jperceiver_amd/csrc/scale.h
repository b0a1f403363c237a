// Operand scales of the two-way fp16 split kernels (igemm_p9s.h, JP_NS == 2): a tensor is scaled by the power of two that puts its
// LARGEST MAGNITUDE into [2^14, 2^15) before it is split, so the kernels need that magnitude on the device.
//
// Everything a conv entry point knows about magnitudes arrives through its own ARGUMENTS (include/jperceiver_hip.h: amax_x / amax_dy
// = slots the caller filled, amax_y = slot the kernel's epilogue folds max|y| into, amax_ws = caller scratch for the magnitudes
// the call has to reduce itself) and travels down the launchers with the stream, as a JpCall: no library-owned device memory, no state
// between calls.
#pragma once
#include "jp_common.h"

struct JpAmaxCtx {
    static constexpr int MAXH = 12;
    const float* t[MAXH];           // operand base pointers whose magnitude slot is known: given by the caller, or reduced by this call
    const float* a[MAXH];
    int nh = 0;
    float* ws = nullptr;            // caller scratch: JP_AMAX_WS_SLOTS slots (amax_ws)
    int ws_used = 0;
    unsigned* out = nullptr;        // the caller's amax_y: taken by the launcher whose kernel folds it into its epilogue
    bool out_taken = false;
    float* stats = nullptr;         // the caller's bn_stats scratch (forward): per-(pixel tile, wave) partial sums of y and y^2 per channel,
    int stats_parts = 0;            // written by the 4-wave 3x3 patch kernels' epilogue; > 0 once a launcher took it (= partials per channel)
    void know(const float* tensor, const float* slot) {
        if (tensor && slot && nh < MAXH) { t[nh] = tensor; a[nh] = slot; ++nh; }
    }
    const float* find(const float* tensor) const {
        for (int i = 0; i < nh; ++i)
            if (t[i] == tensor) return a[i];
        return nullptr;
    }
};
constexpr int JP_AMAX_WS_SLOTS = 4;     // distinct operands one call can have to reduce itself (wgrad_src3: three sources + dY)

// the stream of a call plus its magnitude arguments; converts to the plain stream wherever one is expected
struct JpCall {
    hipStream_t st;
    JpAmaxCtx* ax;
    JpCall(hipStream_t s, JpAmaxCtx* c = nullptr) : st(s), ax(c) {}
    operator hipStream_t() const { return st; }
};
// sets *flag = "the epilogue took amax_y" when the entry point returns, whichever return statement that is
struct JpAmaxDone {
    int* flag;
    const JpAmaxCtx* c;
    int* parts = nullptr;           // forward entry points: *parts = partial sums per channel the epilogue left in bn_stats (0: none)
    ~JpAmaxDone() {
        if (flag) *flag = c->out_taken ? 1 : 0;
        if (parts) *parts = c->stats_parts;
    }
};

// Device pointer to a slot holding max |x[0 .. n)|, valid for kernels launched on the call's stream after this: the slot the caller
// passed for this tensor, or a reduction launched here into the next slot of the caller's amax_ws (remembered for the rest of the
// call).  nullptr + jp_set_last_error when neither exists -- the entry points check their arguments before they get here.
const float* jp_amax_of(const float* x, long n, const JpCall& c);
// max over up to three tensors (the iconv kernels' channel segments): one slot holding the largest of the three
const float* jp_amax_of3(const float* x0, long n0, const float* x1, long n1, const float* x2, long n2, const JpCall& c);
// the caller's amax_y request, for a launcher whose kernel reports max |what it stores| (-> nullptr if none)
inline unsigned* jp_take_amax_out(const JpCall& c) {
    if (!c.ax || !c.ax->out) return nullptr;
    c.ax->out_taken = true;             // (several launches of one call may each write a part of y: all of them fold into the slot)
    return c.ax->out;
}
