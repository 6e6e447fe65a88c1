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

// ---- legacy_unit_split_kernel: sums per phase (kept in registers, written once at the end) of wave 0 of workgroup 0 (slots 0..7) and of the
//      last workgroup (slots 8..15) --------------------------------------------------------------------------------------------------
#ifdef L3AC_DIAG_UNIT_LEGACY
#ifdef L3AC_LG_STAMPS
__device__ long long g_lg_stamps[20];  // [16], [17]: tiles taken by the two workgroups
__device__ long long g_lg_blocks[1024][2];  // per workgroup: lifetime, tiles (summed over launches)
#define LG_STAMP_INIT()                                          \
    long long lg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};              \
    long long lg_tiles = 0;                                      \
    long long lg_last = (long long)__builtin_amdgcn_s_memtime()
#define LG_STAMP(slot)                                                          \
    do {                                                                        \
        const long long now_ = (long long)__builtin_amdgcn_s_memtime();         \
        lg_acc[slot] += now_ - lg_last;                                         \
        lg_last = now_;                                                         \
    } while (0)
#define LG_STAMP_FLUSH()                                                                                 \
    do {                                                                                                 \
        if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {                      \
            for (int i_ = 0; i_ < 8; ++i_) g_lg_stamps[(blockIdx.x == 0 ? 0 : 8) + i_] += lg_acc[i_];   \
            g_lg_stamps[blockIdx.x == 0 ? 16 : 17] += lg_tiles;                                          \
        }                                                                                                \
        if (threadIdx.x == 0 && blockIdx.x < 1024) {                                                     \
            long long t_ = 0;                                                                            \
            for (int i_ = 0; i_ < 8; ++i_) t_ += lg_acc[i_];                                             \
            g_lg_blocks[blockIdx.x][0] += t_;                                                            \
            g_lg_blocks[blockIdx.x][1] += lg_tiles;                                                      \
        }                                                                                                \
    } while (0)
extern "C" int l3ac_debug_lg_blocks(long long* out, int reset) {
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lg_blocks), sizeof(long long) * 2048);
    if (reset) {
        static long long zero[2048] = {};
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_lg_blocks), zero, sizeof(zero));
    }
    return rc;
}
extern "C" int l3ac_debug_lg_stamps(long long* out, int n, int reset) {
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lg_stamps), (size_t)n * sizeof(long long));
    if (reset) {
        long long zero[20] = {};
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_lg_stamps), zero, sizeof(zero));
    }
    return rc;
}
#else
#define LG_STAMP_INIT() do { } while (0)
#define LG_STAMP(slot) do { } while (0)
#define LG_STAMP_FLUSH() do { } while (0)
#endif
#endif
