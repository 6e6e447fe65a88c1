// Shared host-side helpers of libl3ac_hip.so (error plumbing, launch checks).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "l3ac_hip.h"

void l3ac_set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

#define L3AC_HIP_CHECK(expr)                                                                             \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            l3ac_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);   \
            return L3AC_EHIP;                                                                            \
        }                                                                                                \
    } while (0)

#define L3AC_LAUNCH_CHECK() L3AC_HIP_CHECK(hipGetLastError())

#define L3AC_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            l3ac_set_error(__VA_ARGS__);   \
            return L3AC_EINVAL;            \
        }                                  \
    } while (0)

#define L3AC_TRY(expr)             \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != L3AC_OK) return rc_; \
    } while (0)

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Launch-side state that is a property of the DEVICE (function attributes set with hipFuncSetAttribute, the CU count) is
// kept per device ordinal: the header allows one context per device, i.e. several devices in one process.  The flags are
// atomics (two threads may first-launch on two contexts at once; the configuration they guard is idempotent), and an ordinal
// outside the table is never folded onto another device's slot: it has no cached state and is configured on every launch.
constexpr int L3AC_MAX_DEVICES = 64;
static inline int l3ac_device_slot() {  // -1: no slot (hipGetDevice failed or ordinal >= L3AC_MAX_DEVICES)
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= L3AC_MAX_DEVICES) return -1;
    return d;
}
struct PerDeviceOnce {  // `if (once.first()) { ...configure...; once.done(); }` — per device of the calling thread
    std::atomic<bool> flag[L3AC_MAX_DEVICES] = {};
    bool first() const {
        const int s = l3ac_device_slot();
        return s < 0 || !flag[s].load(std::memory_order_acquire);
    }
    void done() {
        const int s = l3ac_device_slot();
        if (s >= 0) flag[s].store(true, std::memory_order_release);
    }
};
int l3ac_device_cu_count();  // multiprocessor count of the current device (cached per device; 256 if the query fails)
static inline int64_t round_up64(int64_t a, int64_t b) { return ceil_div64(a, b) * b; }

// ---------------------------------------------------------------------------------------------------------
// Per-launch profiler: when a thread has an active Profiler, every kernel launch made through L3AC_PROFILED is
// bracketed by HIP events recorded on the launch stream (so the durations are the kernels' own, measured where
// they run), tagged with the algorithmic FLOPs / bytes of that launch.  Off by default: zero overhead.
// ---------------------------------------------------------------------------------------------------------
#include <string>
#include <vector>

struct ProfRecord {
    char name[64];
    hipEvent_t start, stop;
    double flops, bytes;
};
struct Profiler {
    std::vector<ProfRecord> records;
    bool failed = false;
};
Profiler* l3ac_current_profiler();
void l3ac_set_current_profiler(Profiler* p);

struct ProfScope {
    Profiler* prof;
    hipStream_t stream;
    size_t slot = 0;
    ProfScope(hipStream_t s, const char* name, double flops, double bytes) : prof(l3ac_current_profiler()), stream(s) {
        if (!prof) return;
        ProfRecord r{};
        std::snprintf(r.name, sizeof(r.name), "%s", name);
        r.flops = flops;
        r.bytes = bytes;
        if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) {
            prof->failed = true;
            prof = nullptr;
            return;
        }
        (void)hipEventRecord(r.start, stream);
        slot = prof->records.size();
        prof->records.push_back(r);
    }
    ~ProfScope() {
        if (prof) (void)hipEventRecord(prof->records[slot].stop, stream);
    }
};
