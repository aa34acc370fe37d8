// Optional per-launch timing of the MFMA kernels with HIP events recorded on the
// launch stream (bench.py's roofline leg). Off by default: zero overhead.
#include <vector>
#include <string>
#include <stdarg.h>
#include <stdlib.h>
#include "kernels.h"

namespace mpu {
namespace {
struct Rec { hipEvent_t a, b; int kind; double flops; };
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t take_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
}  // namespace

bool prof_on() { return g_on; }
// A scope times one "launch" of a family (a convolution with its split-K finish, the four launches of a tap-combined
// up-convolution, ...): launch_k (kernels.h) binds the start event to the scope's first dispatch and the stop event to
// every dispatch, so the pair spans first-kernel start .. last-kernel end. MPU_PROF_MARKERS=1: the round-1..3 form
// (hipEventRecord markers around the launches), kept for comparison.
namespace {
bool g_open = false, g_first = false;
int g_markers = -1;
bool markers() {
    if (g_markers < 0) g_markers = (int)env(ENV_PROF_MARKERS);
    return g_markers == 1;
}
void drop_open_scope() {                        // a scope that never saw a kernel (error return between begin and launch)
    if (!g_open) return;
    g_open = false;
    if (g_first && !g_recs.empty()) { g_pool.push_back(g_recs.back().a); g_pool.push_back(g_recs.back().b); g_recs.pop_back(); }
}
}  // namespace
void prof_begin(int kind, double flops, hipStream_t st) {
    drop_open_scope();
    Rec r; r.a = take_event(); r.b = take_event(); r.kind = kind; r.flops = flops;
    g_recs.push_back(r);
    if (markers()) { (void)hipEventRecord(r.a, st); return; }
    g_open = true; g_first = true;
}
void prof_end(hipStream_t st) {
    if (markers()) { if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().b, st); return; }
    drop_open_scope();
}
hipEvent_t prof_start_event() {
    if (!g_open || !g_first) return nullptr;
    g_first = false;
    return g_recs.back().a;
}
hipEvent_t prof_stop_event() { return g_open ? g_recs.back().b : nullptr; }

// ---- schedule log: which kernel schedule each conv / wgrad launch took (tests assert the intended dispatch) ----
namespace { bool g_sched_on = false; std::string g_sched; }
bool sched_log_on() { return g_sched_on; }
void sched_note(const char* fmt, ...) {
    if (!g_sched_on) return;
    char buf[256];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_sched += buf; g_sched += '\n';
}
// ---- dev aid: in-kernel s_memtime stamps (MPU_STAMPS=1) -------------------------------------------------------
namespace { unsigned long long* g_stamps = nullptr; int g_stamps_state = -1; }
unsigned long long* stamp_buffer() {
    if (g_stamps_state < 0) {
        g_stamps_state = (int)env(ENV_STAMPS);
        if (g_stamps_state) {
            if (hipMalloc((void**)&g_stamps, 64 * 8 * sizeof(unsigned long long)) != hipSuccess) { g_stamps = nullptr; g_stamps_state = 0; }
            else (void)hipMemset(g_stamps, 0, 64 * 8 * sizeof(unsigned long long));
        }
    }
    return g_stamps;
}
}  // namespace mpu

using namespace mpu;

extern "C" {

int mpu_debug_stamps_read(uint64_t* host_out, int32_t n) {
    MPU_REQUIRE(host_out && n >= 0 && n <= 64 * 8, "mpu_debug_stamps_read: bad argument");
    unsigned long long* b = stamp_buffer();
    if (!b) { for (int i = 0; i < n; ++i) host_out[i] = 0; return MPU_OK; }
    MPU_CHECK_HIP(hipDeviceSynchronize());
    MPU_CHECK_HIP(hipMemcpy(host_out, b, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    MPU_CHECK_HIP(hipMemset(b, 0, 64 * 8 * sizeof(unsigned long long)));
    return MPU_OK;
}

int mpu_profile_enable(int32_t on) {
    for (auto& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
    g_recs.clear();
    g_open = g_first = false;
    g_on = on != 0;
    return MPU_OK;
}

// Synchronises on the recorded events. kind: 0 = conv_igemm (forward + data gradient),
// 1 = wgrad_igemm. Any output pointer may be NULL.
int mpu_profile_summary(int32_t kind, double* total_ms, double* total_flops, int64_t* launches) {
    MPU_REQUIRE(kind >= 0 && kind < PROF_KINDS, "mpu_profile_summary: bad kind");
    double ms = 0, fl = 0; int64_t n = 0;
    for (auto& r : g_recs) {
        if (r.kind != kind) continue;
        MPU_CHECK_HIP(hipEventSynchronize(r.b));
        float t = 0.f;
        MPU_CHECK_HIP(hipEventElapsedTime(&t, r.a, r.b));
        ms += t; fl += r.flops; ++n;
    }
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    if (launches) *launches = n;
    return MPU_OK;
}

// Schedule log: one text line per conv / wgrad launch, "<family> <schedule> mode=.. B=.. H=.. W=.. Cin=.. Cout=.. ksplit=..".
int mpu_schedule_log_enable(int32_t on) {
    g_sched.clear();
    g_sched_on = on != 0;
    return MPU_OK;
}
// Copies the log (NUL-terminated, truncated to cap) and returns its full length in bytes.
int64_t mpu_schedule_log_read(char* buf, int64_t cap) {
    if (buf && cap > 0) {
        const size_t n = g_sched.size() < (size_t)(cap - 1) ? g_sched.size() : (size_t)(cap - 1);
        memcpy(buf, g_sched.data(), n); buf[n] = 0;
    }
    return (int64_t)g_sched.size();
}

}  // extern "C"
