// Tiles by ticket (round 6).  The waves of these workgroups walk their tiles independently and a SIMD serves its OLDEST wave first: with
// equal static shares (tile = blockIdx WAVES + wave + k gridDim WAVES) the first waves of a workgroup finish early and the last ones run
// the end of the kernel on a thinly occupied CU (conv_unit_wide.hip, 'DYN': measured there with stamps; profiles/r06/tickets.md).  The
// workgroup keeps its tiles — ((round r) gridDim + blockIdx) WAVES + j — and a wave takes the next (r, j) by a ticket in LDS (an int the
// workgroup zeroes before its first barrier); n_tiles = nothing left.  Which wave computes a tile does not enter its arithmetic: the same bits.
#pragma once

#include <hip/hip_runtime.h>

template <int WAVES>
__device__ __forceinline__ int take_tile(int* ticket, const int lane, const int n_tiles) {
    for (;;) {
        int q = 0;
        if (lane == 0) q = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        q = __builtin_amdgcn_readfirstlane(q);
        const int base = ((q / WAVES) * (int)gridDim.x + (int)blockIdx.x) * WAVES;
        if (base >= n_tiles) return n_tiles;
        if (base + q % WAVES < n_tiles) return base + q % WAVES;  // (the last round's missing tiles are skipped)
    }
}
