// Launch API of the HIP kernels (one .hip file per kernel family under kernels/).
// All pointers are device pointers; activations are frame-major [batch][frame][channel].
#pragma once

#include "common.hpp"

// ------------------------------------------------------------------------------------------------
// fp32 MFMA GEMM:  c[m][n] = epilogue( sum_k A(m,k) * w[n][k] )
// ------------------------------------------------------------------------------------------------
enum GemmEpilogue : int {
    EPI_BIAS = 0,       // acc + bias[n]                               (bias may be null)
    EPI_BIAS_RES = 1,   // res[m][n] + (acc + bias[n])                 ConvUnit / LegacyUnit residual, attention/ff residual
    EPI_SNAKE = 2,      // snake(acc + bias[n], alpha[n])              LegacyUnit inner activation
    EPI_SNAKE_GRN = 3,  // s = snake(acc + bias); gamma*s + beta + s   ConvUnit pw_conv1 -> snake -> GRN (normaliser == 1)
    EPI_GEGLU = 4,      // value/gate column tiles interleaved: out[m][j] = v * gelu(g)   (FeedForward)
};

struct GemmArgs {
    // A operand.  taps == 1: plain rows, a[m * lda + k].  taps > 1: implicit 1-D convolution over the frames of
    // one clip: k = tap * cin + c reads a[(b*frames + t + (tap - taps/2) * dil) * lda + c], zero outside the clip.
    const float* a = nullptr;
    int64_t lda = 0;
    int taps = 1, dil = 1, cin = 0;
    int64_t frames = 0;
    // W operand [n][k] (row stride ldw), output c [m][ldc]
    const float* w = nullptr;
    int64_t ldw = 0;
    // optional pre-split bf16x3 image of w (gemm_split.hip); when set, enabled and the shape is eligible the product
    // runs on the bf16 matrix cores at fp32 accuracy, otherwise on the exact-fp32 MFMA kernel
    const unsigned char* w_img = nullptr;
    float* c = nullptr;
    int64_t ldc = 0;
    int64_t m = 0;
    int n = 0, k = 0;
    // epilogue
    int epi = EPI_BIAS;
    const float* bias = nullptr;
    const float* res = nullptr;
    int64_t ldres = 0;
    const float* alpha = nullptr;      // snake alpha[n]
    const float* inv_alpha = nullptr;  // 1 / (alpha[n] + 1e-8)
    const float* gamma = nullptr;      // GRN
    const float* beta = nullptr;
    int n_out = 0;                     // EPI_GEGLU: number of valid output columns (ff inner)
    // optional EnhanceBlock gate applied to the A operand while it is staged (tconv/__init__.py:35-44, same arithmetic as
    // the SRC_GATE row kernel): a'(m, k) = a + (gate_b[k] + gate_w[k][:] . instnorm(yi[m][:])) * a.  Plain A, k % 16 == 0.
    const float* gate_yi = nullptr;     // raw branch signals [m][4]
    const float* gate_stats = nullptr;  // [clip][8] = mean[4], 1/std[4]
    const float* gate_in_w = nullptr;   // InstanceNorm affine [4]
    const float* gate_in_b = nullptr;
    const float* gate_w = nullptr;      // merge conv [k][4]
    const float* gate_b = nullptr;      // [k]
    int64_t gate_frames = 0;            // rows per clip
};
int launch_gemm(hipStream_t s, const GemmArgs& g);

// bf16x3 split-operand GEMM (kernels/gemm_split.hip)
#define L3AC_SPLIT_TILE_BYTES 24576  // one k tile (32) of one column block (128): 3 planes x 128 rows x 64 B
bool gemm_split_default();                 // L3AC_GEMM_SPLIT (default 1): the route a new context starts on; the route of a
                                           // launch is its GemmArgs::w_img (null = exact fp32 MFMA kernel)
bool gemm_split_eligible(int n, int k);
bool gemm_split_conv_ok(const struct GemmArgs& g);  // taps > 1 on the split route: cin % 32 == 0, plain frame-major rows
int64_t gemm_split_image_bytes(int n, int k);
void gemm_split_image_host(const float* w, int64_t ldw, int n, int k, unsigned char* img);  // img: host buffer
int launch_gemm_split_image(hipStream_t s, const float* w, int64_t ldw, int n, int k, unsigned char* img);  // device
int launch_gemm_split(hipStream_t s, const GemmArgs& g);
bool gemm_split_w256_ok(const GemmArgs& g);
int launch_gemm_split_w256(hipStream_t s, const GemmArgs& g);  // gemm_split_w256.hip: a batch's rows, n % 256 == 0, an even number of whole k tiles

// ------------------------------------------------------------------------------------------------
// row kernels: one output row = one frame (all channels), optional per-row normalisation
// ------------------------------------------------------------------------------------------------
enum RowSource : int {
    SRC_PLAIN = 0,    // y = x
    SRC_DWCONV7 = 1,  // depth-wise conv k7 pad 3 along frames (ConvUnit.dw_conv)
    SRC_LERP = 2,     // linear upsample x scale, align_corners=False (nn.Upsample)
    SRC_GATE = 3,     // EnhanceBlock: x + (merge(instnorm(yi))) * x
};
enum RowNorm : int {
    NORM_NONE = 0,
    NORM_LN = 1,  // F.layer_norm over channels: (x - mu) * rsqrt(var + eps) * w + b
    NORM_CN = 2,  // channel_norm channels_first: (x - mu) / sqrt(var + eps) * w + b
};
struct RowArgs {
    const float* x = nullptr;
    float* y = nullptr;
    int64_t batch = 0, frames_in = 0, frames_out = 0;
    int c = 0;
    int src = SRC_PLAIN, norm = NORM_NONE;
    const float* dw_w = nullptr;  // [7][c]
    const float* dw_b = nullptr;  // [c]
    int scale = 1;                // SRC_LERP
    const float* yi = nullptr;    // SRC_GATE: raw branch signals [batch][frames][4]
    const float* stats = nullptr; // [batch][8] = mean[4], invstd[4]
    const float* in_w = nullptr;  // InstanceNorm affine [4]
    const float* in_b = nullptr;
    const float* gate_w = nullptr;  // merge conv [c][4]
    const float* gate_b = nullptr;  // [c]
    const float* nw = nullptr;      // norm affine [c]
    const float* nb = nullptr;
    float eps = 0.f;
};
int launch_rows(hipStream_t s, const RowArgs& r);

// elementwise
int launch_snake(hipStream_t s, const float* x, float* y, int64_t rows, int c, const float* alpha,
                 const float* inv_alpha, int mode = 0);  // mode: see snake_kernel
int launch_geglu(hipStream_t s, const float* h, int64_t ldh, float* y, int64_t ldy, int64_t rows, int inner);
int launch_grn_sumsq(hipStream_t s, const float* h, int64_t batch, int64_t per_clip, float* sumsq);
int launch_grn_apply(hipStream_t s, float* h, int64_t batch, int64_t frames, int c, const float* sumsq,
                     const float* gamma, const float* beta, float* min_track = nullptr);  // min_track: running minimum of sumsq

// FirstBlock (tconv/__init__.py:8-27): audio [batch][samples] -> y [batch][frames][d0], frames >= samples (zero tail)
struct FirstBlockW {
    const float* tw;   // trend convs [5][4][7]
    const float* tb;   // [5][4]
    const float* w1;   // conv_1 [80][20]
    const float* b1;   // [80]
    const float* w2;   // conv_2 [d0][81]
    const float* b2;   // [d0]
    int d0;
};
int launch_first_block(hipStream_t s, const FirstBlockW& w, const float* audio, int64_t audio_stride, int batch,
                       int samples, int frames, float* y);

// EnhanceBlock helpers (tconv/__init__.py:30-44)
struct EnhanceW {
    const float* tw;  // trend convs [4][7]
    const float* tb;  // [4]
};
int launch_enhance_branches(hipStream_t s, const EnhanceW& w, const float* x, int batch, int frames, int c, float* yi);
int launch_enhance_stats(hipStream_t s, const float* yi, int batch, int frames, float* stats);

// output head (modules.py:190-195 after the Snake1d): conv 24 -> 1 k7 pad 3, tanh
// pretanh (validation, l3ac_ctx_set_head_pretanh): store the conv result BEFORE the final tanh
int launch_head(hipStream_t s, const float* x, int batch, int frames, int c, const float* w /*[7][c]*/,
                const float* b, float* audio, bool pretanh = false);

// causal local attention, look-back one window (local_attention.LocalAttention); qkv [rows][3*heads*dh]
int launch_attention(hipStream_t s, const float* qkv, float* out, const float* bias_table /*[heads][2*window]*/,
                     int batch, int frames, int heads, int dh, int window);

// FSQ
struct FsqArgs {
    const float* x = nullptr;   // [n][feat] or null (latents is then the input)
    int64_t n = 0;
    int feat = 0, n_levels = 0;
    int levels[L3AC_MAX_LEVELS] = {0};
    const float* w_in = nullptr;
    const float* b_in = nullptr;
    const float* w_out = nullptr;
    const float* b_out = nullptr;
    const int32_t* idx_in = nullptr;  // decode path: indices are the input
    float* q_feature = nullptr;
    int32_t* indices = nullptr;
    float* level_indices = nullptr;
    float* latents = nullptr;
    bool act_in = false;              // x == idx_in == null and `latents` holds act = (tanh(lat) + 1) / 2 (vq/fsq.py:56)
    int* bad_count = nullptr;         // decode path: device counter of indices outside [0, codebook size) (they are clamped)
};
int launch_fsq(hipStream_t s, const FsqArgs& a);
// measurement aid: fsq_kernel's grid and access pattern (feat 128, 6 levels) with no arithmetic — its achievable HBM ceiling
int launch_fsq_copy_ceiling(hipStream_t s, const float* x, int64_t n, float* q, int32_t* idx, float* li, int blocks_per_cu = 0);
// token bit stream (kernels/bitpack.hip)
int launch_pack_indices(hipStream_t s, const int32_t* idx, int batch, int n_tok, int bits, uint32_t* out, int words_per_clip);
int launch_unpack_indices(hipStream_t s, const uint32_t* in, int batch, int n_tok, int bits, int words_per_clip, int32_t* idx);
// explicit-codebook L2 argmin (kernels/fsq.hip): scratch = vq_argmin_scratch_bytes(n, k) bytes, caller-provided
size_t vq_argmin_scratch_bytes(int64_t n, int k, int form = 0);
// form: 0 automatic, 1 the direct-form scan wherever the screened form would run (the reference the screened form is tested against)
int launch_vq_argmin(hipStream_t s, const float* queries, int64_t n, const float* codebook, int k, int dim, void* scratch,
                     int32_t* out_idx, int form = 0);
