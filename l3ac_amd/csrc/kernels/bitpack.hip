// Token wire format (the reference has none: indices stay int32 tensors, vq/fsq.py:68, and its
// 998.2 bps figure is frame_rate * log2(K), l3ac/__init__.py:38-41).
//
// Per clip, token t occupies bits [t * bits, (t + 1) * bits) of a little-endian bit stream (bit b of the stream is bit
// b % 8 of byte b / 8), bits = ceil(log2(codebook size)) = 17 at 1kbps (117 649 codes), 18 at 3kbps (250 047).
// Each clip's stream is padded with zero bits to a whole number of 32-bit words.
//
// pack: one thread per output word gathers the <= 3 tokens that overlap it.  unpack: one thread per token reads the
// <= 2 words it spans.  Integer/byte work, HBM-bound (4 B in, bits/8 B out per token), bit-exact by construction.
#include "../kernels.hpp"

namespace {

__global__ __launch_bounds__(256) void pack_kernel(const int32_t* __restrict__ idx, int n_tok, int bits, uint32_t* __restrict__ out,
                                                  int words_per_clip) {
    const int b = blockIdx.y;
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= words_per_clip) return;
    const int32_t* src = idx + (int64_t)b * n_tok;
    const int64_t bit0 = (int64_t)w * 32;
    int t = (int)(bit0 / bits);
    uint32_t word = 0;
    for (; t < n_tok && (int64_t)t * bits < bit0 + 32; ++t) {
        const int64_t off = (int64_t)t * bits - bit0;  // position of the token's bit 0 relative to this word
        const uint32_t v = (uint32_t)src[t] & ((bits == 32) ? 0xffffffffu : ((1u << bits) - 1u));
        if (off >= 0)
            word |= v << off;
        else
            word |= v >> (-off);
    }
    out[(int64_t)b * words_per_clip + w] = word;
}

__global__ __launch_bounds__(256) void unpack_kernel(const uint32_t* __restrict__ in, int words_per_clip, int bits,
                                                    int32_t* __restrict__ idx, int n_tok) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tok) return;
    const uint32_t* src = in + (int64_t)b * words_per_clip;
    const int64_t bit0 = (int64_t)t * bits;
    const int w = (int)(bit0 >> 5), sh = (int)(bit0 & 31);
    uint64_t two = src[w];
    if (sh + bits > 32) two |= (uint64_t)src[w + 1] << 32;
    const uint32_t mask = (bits == 32) ? 0xffffffffu : ((1u << bits) - 1u);
    idx[(int64_t)b * n_tok + t] = (int32_t)((uint32_t)(two >> sh) & mask);
}

}  // namespace

int launch_pack_indices(hipStream_t s, const int32_t* idx, int batch, int n_tok, int bits, uint32_t* out, int words_per_clip) {
    L3AC_REQUIRE(bits >= 1 && bits <= 32 && batch > 0 && batch <= 65535 && n_tok > 0, "pack: bad arguments");
    L3AC_REQUIRE((int64_t)words_per_clip * 32 >= (int64_t)n_tok * bits, "pack: output too small");
    ProfScope prof(s, "pack_kernel", 0.0, (double)batch * n_tok * (4.0 + bits / 8.0));
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)ceil_div64(words_per_clip, 256), (unsigned)batch), dim3(256), 0, s, idx, n_tok, bits,
                       out, words_per_clip);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}

int launch_unpack_indices(hipStream_t s, const uint32_t* in, int batch, int n_tok, int bits, int words_per_clip, int32_t* idx) {
    L3AC_REQUIRE(bits >= 1 && bits <= 32 && batch > 0 && batch <= 65535 && n_tok > 0, "unpack: bad arguments");
    L3AC_REQUIRE((int64_t)words_per_clip * 32 >= (int64_t)n_tok * bits, "unpack: input too small");
    ProfScope prof(s, "unpack_kernel", 0.0, (double)batch * n_tok * (4.0 + bits / 8.0));
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)ceil_div64(n_tok, 256), (unsigned)batch), dim3(256), 0, s, in, words_per_clip, bits,
                       idx, n_tok);
    L3AC_LAUNCH_CHECK();
    return L3AC_OK;
}
