// In-kernel s_memtime stamps of the diagnostic builds (never in the product library: every macro below is empty unless the build
// defines the kernel's switch — L3AC_EXTRA_HIPCC_FLAGS=-DL3AC_TS_STAMPS / -DL3AC_WIDE_STAMPS with L3AC_BUILD_TAG, read by
// tools/ts_stamps.py / tools/wide_stamps.py through the l3ac_debug_* entry points, which are not part of the ABI).  Stamp values go to
// a buffer of their own; no output depends on them.  A kernel file names its section (L3AC_DIAG_UNIT_*) before including this header:
// the buffer and its reader then exist in that translation unit only.
#pragma once

#include <hip/hip_runtime.h>

// ---- trans_stack_kernel: sums per phase of wave 0 of workgroup 0 (needs `tid` in scope) --------------------------------------------
#ifdef L3AC_DIAG_UNIT_TRANS_STACK
#ifdef L3AC_TS_STAMPS
__device__ long long g_ts_stamps[16];
#define TS_STAMP_INIT() long long ts_last = (long long)__builtin_amdgcn_s_memtime()
#define TS_STAMP(slot)                                                                        \
    do {                                                                                      \
        if (blockIdx.x == 0 && tid == 0) {                                                    \
            const long long now_ = (long long)__builtin_amdgcn_s_memtime();                   \
            g_ts_stamps[slot] += now_ - ts_last;                                              \
            ts_last = now_;                                                                   \
        }                                                                                     \
    } while (0)
#define TS_STAMP_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
extern "C" int l3ac_debug_ts_stamps(long long* out, int n, int reset) {
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ts_stamps), (size_t)n * sizeof(long long));
    if (reset) {
        long long zero[16] = {};
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ts_stamps), zero, sizeof(zero));
    }
    return rc;
}
#else
#define TS_STAMP_INIT() do { } while (0)
#define TS_STAMP(slot) do { } while (0)
#define TS_STAMP_DRAIN() do { } while (0)
#endif
#endif

// ---- conv_unit_wide_kernel: the phase boundaries of every pass of wave 0 (needs `lane`, `wave`, `pass_no` in scope) -----------------
#ifdef L3AC_DIAG_UNIT_CONV_UNIT_WIDE
#ifdef L3AC_WIDE_STAMPS
__device__ unsigned long long g_wide_stamps[512 * 16 * 8];
#define WIDE_STAMP(slot)                                                                                      \
    do {                                                                                                      \
        if (lane == 0 && wave == 0 && pass_no < 16)                                                           \
            g_wide_stamps[((size_t)blockIdx.x * 16 + pass_no) * 8 + (slot)] = __builtin_amdgcn_s_memtime();   \
    } while (0)
extern "C" int l3ac_debug_wide_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wide_stamps), (size_t)n * sizeof(unsigned long long));
}
#else
#define WIDE_STAMP(slot) do { } while (0)
#endif
#endif

// ---- gemm_split_kernel_w256: wave 0 of every workgroup: start | loop entry | loop exit | end (needs `lane`, `wave` in scope) ------------
#ifdef L3AC_DIAG_UNIT_GEMM_W256
#ifdef L3AC_W256_STAMPS
__device__ unsigned long long g_w256_stamps[4096 * 4];
#define W256_STAMP(slot)                                                                             \
    do {                                                                                             \
        if (lane == 0 && wave == 0 && blockIdx.x < 4096)                                             \
            g_w256_stamps[(size_t)blockIdx.x * 4 + (slot)] = __builtin_amdgcn_s_memtime();           \
    } while (0)
extern "C" int l3ac_debug_w256_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w256_stamps), (size_t)n * sizeof(unsigned long long));
}
#else
#define W256_STAMP(slot) do { } while (0)
#endif
#endif
