// Error-string plumbing for the C-ABI (include/jperceiver_hip.h) and the opt-in per-kernel profiler.
#include <hip/hip_runtime.h>
#include <string.h>
#include <vector>
static thread_local char g_err[512] = "";
extern "C" void jp_set_last_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* jp_last_error_string(void) { return g_err; }
extern "C" int jp_abi_version(void) { return 3; }

// ---- per-kernel HIP-event timing of the implicit-GEMM launches (bench.py's roofline leg).
// Off by default: the launch helpers of conv.hip call jp_prof_before/after, which return at once unless a profile is
// open.  jp_profile_begin(n) creates 2n events (host objects, no device memory); while it is open every igemm
// dispatch is bracketed by two hipEventRecord calls ON THE STREAM IT IS LAUNCHED ON, so the timings are valid with
// the side stream running (torch.cuda.Event only sees torch's current stream).  Single-threaded by design.
namespace {
struct Rec { const char* tag; double flops; };
bool g_on = false;
int g_n = 0, g_cap = 0;
std::vector<hipEvent_t> g_ev;
std::vector<Rec> g_rec;
}  // namespace

void jp_prof_before(const char* tag, double flops, hipStream_t st) {
    if (!g_on || g_n >= g_cap) return;
    g_rec[g_n] = Rec{tag, flops};
    (void)hipEventRecord(g_ev[2 * g_n], st);
}
void jp_prof_after(hipStream_t st) {
    if (!g_on || g_n >= g_cap) return;
    (void)hipEventRecord(g_ev[2 * g_n + 1], st);
    ++g_n;
}

extern "C" int jp_profile_begin(int max_records) {
    if (max_records <= 0 || g_on) { jp_set_last_error("profile_begin: bad argument or already open"); return -1; }
    for (hipEvent_t e : g_ev) (void)hipEventDestroy(e);
    g_ev.assign(2 * (size_t)max_records, nullptr);
    g_rec.assign(max_records, Rec{nullptr, 0.0});
    for (auto& e : g_ev) {
        hipError_t rc = hipEventCreate(&e);
        if (rc != hipSuccess) { jp_set_last_error(hipGetErrorString(rc)); return (int)rc; }
    }
    g_cap = max_records;
    g_n = 0;
    g_on = true;
    return 0;
}
extern "C" int jp_profile_count(void) { return g_n; }
extern "C" int jp_profile_end(void) {
    g_on = false;
    return g_n;
}
// record i -> tag (the launch helper's template instantiation: tile shape, loaders, epilogue), executed FLOPs
// (2*M*N*K of the GEMM the launch computes) and elapsed milliseconds; call after the stream(s) were synchronised.
extern "C" int jp_profile_get(int i, char* tag, int tag_len, double* flops, float* ms) {
    if (i < 0 || i >= g_n || !tag || tag_len <= 0 || !flops || !ms) { jp_set_last_error("profile_get: bad args"); return -1; }
    strncpy(tag, g_rec[i].tag ? g_rec[i].tag : "", tag_len - 1);
    tag[tag_len - 1] = 0;
    *flops = g_rec[i].flops;
    hipError_t rc = hipEventElapsedTime(ms, g_ev[2 * i], g_ev[2 * i + 1]);
    if (rc != hipSuccess) { jp_set_last_error(hipGetErrorString(rc)); return (int)rc; }
    return 0;
}
