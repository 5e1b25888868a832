// Optional per-launch timing of the hand-written kernels with HIP events recorded on the launch stream.
// Off by default (zero overhead: one relaxed load per launch).  bench.py switches it on for a separate,
// un-timed instrumented pass and reads per-kernel {launches, total ms, algorithmic bytes} back, which is
// what the "roofline" object of the bench line is computed from.  Not hipGraph-capturable while enabled.
#include "afan_common.h"
#include <mutex>
#include <string>
#include <vector>

namespace afan {
namespace prof {

struct Rec {
    const char* name;
    double bytes, flops;
    hipEvent_t a, b;
};

static std::mutex g_mu;
static std::vector<Rec> g_recs;
int g_enabled = 0;
thread_local unsigned g_launches = 0;

void begin(const char* name, double bytes, double flops, hipStream_t st, size_t* slot) {
    std::lock_guard<std::mutex> lk(g_mu);
    Rec r{name, bytes, flops, nullptr, nullptr};
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { *slot = (size_t)-1; return; }
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
    *slot = g_recs.size() - 1;
}

void end(size_t slot, hipStream_t st) {
    if (slot == (size_t)-1) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (slot < g_recs.size()) (void)hipEventRecord(g_recs[slot].b, st);
}

void cancel(size_t slot) {        // nothing was launched inside the scope: the record is dropped by collect (name == nullptr)
    if (slot == (size_t)-1) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (slot < g_recs.size()) g_recs[slot].name = nullptr;
}

}  // namespace prof
}  // namespace afan

extern "C" {

int afan_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(afan::prof::g_mu);
    afan::prof::g_enabled = on ? 1 : 0;
    return AFAN_OK;
}

// Synchronises the recorded events, aggregates per kernel name and clears the records.
// names_out: caller buffer of max_kernels * 64 chars; returns the number of distinct kernels written.
int afan_profile_collect(char* names_out, int64_t* launches, double* total_ms, double* total_bytes,
                         double* total_flops, int max_kernels) {
    using namespace afan::prof;
    std::lock_guard<std::mutex> lk(g_mu);
    std::vector<std::string> names;
    for (auto& r : g_recs) {
        float ms = 0.f;
        if (!r.name) {                               // cancelled: nothing was launched inside the scope (event b never recorded)
            (void)hipEventDestroy(r.a);
            (void)hipEventDestroy(r.b);
            continue;
        }
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) ms = 0.f;
        int k = -1;
        for (size_t i = 0; i < names.size(); ++i)
            if (names[i] == r.name) { k = (int)i; break; }
        if (k < 0 && (int)names.size() < max_kernels) {
            names.emplace_back(r.name);
            k = (int)names.size() - 1;
            launches[k] = 0; total_ms[k] = 0; total_bytes[k] = 0; total_flops[k] = 0;
            snprintf(names_out + 64 * k, 64, "%s", r.name);
        }
        if (k >= 0) { launches[k] += 1; total_ms[k] += ms; total_bytes[k] += r.bytes; total_flops[k] += r.flops; }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_recs.clear();
    return (int)names.size();
}

// Elapsed time of an EMPTY (event, event) bracket on `stream`, averaged over n pairs: what every instrumented launch's
// duration carries on top of the kernel itself (timestamp write + command-processor gap).
int afan_profile_event_overhead(int n, float* us_out, afan_stream_t stream) {
    if (!us_out) return AFAN_ENULL;
    if (n <= 0 || n > 4096) return AFAN_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    std::vector<hipEvent_t> ev(2 * (size_t)n);
    for (auto& e : ev)
        if (hipEventCreate(&e) != hipSuccess) return AFAN_ESHAPE;
    for (int i = 0; i < n; ++i) {
        (void)hipEventRecord(ev[2 * i], st);
        (void)hipEventRecord(ev[2 * i + 1], st);
    }
    hipError_t e = hipStreamSynchronize(st);
    double tot = 0;
    for (int i = 0; i < n && e == hipSuccess; ++i) {
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
        tot += ms;
    }
    for (auto& x : ev) (void)hipEventDestroy(x);
    if (e != hipSuccess) return (int)e;
    *us_out = (float)(tot / n * 1e3);
    return AFAN_OK;
}

}  // extern "C"
