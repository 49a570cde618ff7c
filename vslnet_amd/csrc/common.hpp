// Device-side building blocks shared by every VSLNet kernel (gfx950 / CDNA4 only, wave64).
//
//  * fp32-in / fp32-accumulate MFMA (v_mfma_f32_32x32x2_f32, v_mfma_f32_16x16x4_f32): exact fp32, the only
//    matrix path that can meet the 1e-4 logit gate (SURVEY 7 "hard parts").
//  * "packed" weight layout: a weight used as the B operand of Y = A * W^T is stored as [K/8][N][8] so that one
//    wave-wide float4 load per 8 k-values is a fully coalesced 1 KiB read straight into registers (no LDS staging,
//    no barrier) and feeds 4 MFMAs.  See pack_index().
//  * row tiles: every row-wise kernel owns TILE_M = 32 consecutive rows of the flattened (B*L, 128) activation;
//    the tile lives in LDS with a +4 float row pad (ds_read_b128 conflict-free: row stride == 4 banks mod 64).
//  * counter-based dropout: keep(site_seed, element) = fmix32 hash >= p * 2^32, reproducible in the backward.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vsl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int D = 128;          // model width (configs.dim); the kernels are specialised for it
constexpr int TILE_M = 32;      // rows per workgroup tile
constexpr int NTHREADS = 256;   // 4 waves: wave w owns output columns [32w, 32w+32)
constexpr int LDP = D + 4;      // padded LDS row stride for a 128-wide tile
constexpr float LN_EPS = 1e-6f;
constexpr float MASK_VALUE = -1e30f;
constexpr int DWK = 7;          // depthwise kernel size
constexpr int HALO = 3;
constexpr int HD = 16;          // attention head size the kernels are specialised for (dim 128 / 8 heads)
// LDS row stride of a (L, 16) head slice: padded to 20 floats, un-padded when the slices would not fit 160 KB
__host__ __device__ __forceinline__ int head_slice_stride(int Lp) { return Lp > 768 ? 16 : 20; }
constexpr int CATP = 4 * D + 4; // LDS row stride of the CQAttention concat tile
constexpr int MAX_LC = 32;      // max characters per word
constexpr int MAX_LQ = 64;      // max query words (LDS budget of k_cq_col_bwd)
constexpr int MAX_L = 1024;     // max clips per video (K/V head slices resident in LDS)

// ---------------------------------------------------------------------------------------------------------
// dropout
// ---------------------------------------------------------------------------------------------------------
struct Drop {
    uint32_t seed;     // per-site seed (host mixes step, site id)
    uint32_t thresh;   // p * 2^32 ; 0 => dropout disabled (eval / drop_rate 0)
    float scale;       // 1 / (1 - p)
};

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
// multiplier applied to element `idx` of the site: 0 or 1/(1-p)
__device__ __forceinline__ float drop_mul(const Drop& d, uint32_t idx) {
    if (d.thresh == 0u) return 1.0f;
    const uint32_t h = fmix32(idx * 0x9E3779B1u + d.seed);
    return h >= d.thresh ? d.scale : 0.0f;
}

// ---------------------------------------------------------------------------------------------------------
// wave reductions (64 lanes)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---------------------------------------------------------------------------------------------------------
// packed weights.  A logical matrix Bm[k][col] (k = contraction index, col = output column) is stored as
//   P[(kb * ncols + col) * 8 + e] = Bm[kb*8 + e][col],   kb = k / 8, e = k % 8   (K zero-padded to 8)
// For Y = A W^T (W is (N,K) row-major):  Bm[k][col] = W[col][k]      -> "forward pack"
// For dA = dY W:                         Bm[k][col] = W[k][col]      -> "transpose pack"
// ---------------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ size_t pack_index(int k, int col, int ncols) {
    return ((size_t)(k >> 3) * ncols + col) * 8 + (k & 7);
}
__host__ __device__ __forceinline__ size_t pack_size(int K, int ncols) { return (size_t)((K + 7) / 8) * ncols * 8; }

// ---------------------------------------------------------------------------------------------------------
// MFMA 32x32x2 fp32 row-tile GEMM:  acc[t] (32 rows x 32 cols) += As[32][K] * Bpack[:, col0 + t*cstep + 0..31]
//   As   : LDS, row-major, leading dimension lda floats (lda % 4 == 0, ideally lda == K + 4)
//   Bp   : packed global weights (see above), ncols columns in total
// A-operand lane map (32x32x2): lane l holds A[i = l & 31][k = l >> 5]; B: B[k = l >> 5][j = l & 31].
// One float4 of A and of B per lane covers 8 k-values -> 4 MFMAs (k order inside a block is permuted
// consistently for A and B, which only re-orders the fp32 summation).
// C/D lane map: acc[r] <-> row (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col l & 31.
// ---------------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void gemm32(const float* __restrict__ As, int lda, int K, const float* __restrict__ Bp,
                                       int ncols, int col0, int cstep, f32x16 (&acc)[NT]) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const float* arow = As + i * lda + 4 * h;
    const float4* bp = reinterpret_cast<const float4*>(Bp) + (size_t)(col0 + i) * 2 + h;
    const int nkb = K >> 3;
#pragma unroll 4
    for (int kb = 0; kb < nkb; ++kb) {
        const float4 a = *reinterpret_cast<const float4*>(arow + kb * 8);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 b = bp[((size_t)kb * ncols + t * cstep) * 2];
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[t], 0, 0, 0);
        }
    }
}
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

template <int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}

// ---------------------------------------------------------------------------------------------------------
// weight-gradient ("TN") tile GEMM: acc[t] (32 x 32) += sum_r G[r][n0 + 0..31] * A[r][k0 + 32 t + 0..31]
//   Gs, As : LDS row-major tiles of `rows` rows (rows % 2 == 0), leading dims ldg / lda.
// MFMA A operand = G^T: lane (i, h) supplies G[r + h][n0 + i]; B operand = A: lane (j, h) supplies A[r + h][k0 + j].
// ---------------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void gemm_tn(const float* __restrict__ Gs, int ldg, int n0, const float* __restrict__ As,
                                        int lda, int k0, int rows, f32x16 (&acc)[NT]) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const float* g = Gs + h * ldg + n0 + i;
    const float* a = As + h * lda + k0 + i;
#pragma unroll 4
    for (int r = 0; r < rows; r += 2) {
        const float gv = g[r * ldg];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float av = a[r * lda + 32 * t];
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv, av, acc[t], 0, 0, 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm helpers on LDS row tiles (128-wide rows).  One wave per row, lane owns columns lane and lane + 64.
// ---------------------------------------------------------------------------------------------------------
// normalises row in place: x <- (x - mu) * rstd * g + b ; returns (mu, rstd) to every lane
__device__ __forceinline__ void ln_row_inplace(float* row, const float* __restrict__ g, const float* __restrict__ b,
                                               float& mu, float& rstd) {
    const int lane = threadIdx.x & 63;
    const float x0 = row[lane], x1 = row[lane + 64];
    mu = wave_sum(x0 + x1) * (1.0f / D);
    const float c0 = x0 - mu, c1 = x1 - mu;
    const float var = wave_sum(c0 * c0 + c1 * c1) * (1.0f / D);
    rstd = rsqrtf(var + LN_EPS);
    row[lane] = c0 * rstd * g[lane] + b[lane];
    row[lane + 64] = c1 * rstd * g[lane + 64] + b[lane + 64];
}

// LayerNorm backward for one row held by a wave.
//   x0,x1 : input values (cols lane, lane+64); dy0,dy1 : grad wrt LN output; g : gamma
//   returns dx (two values) ; xhat and dy are returned through refs for the gamma/beta partial sums
__device__ __forceinline__ void ln_row_bwd(float x0, float x1, float dy0, float dy1, const float* __restrict__ g,
                                           float& dx0, float& dx1, float& xh0, float& xh1) {
    const int lane = threadIdx.x & 63;
    const float mu = wave_sum(x0 + x1) * (1.0f / D);
    const float c0 = x0 - mu, c1 = x1 - mu;
    const float var = wave_sum(c0 * c0 + c1 * c1) * (1.0f / D);
    const float rstd = rsqrtf(var + LN_EPS);
    xh0 = c0 * rstd; xh1 = c1 * rstd;
    const float g0 = dy0 * g[lane], g1 = dy1 * g[lane + 64];
    const float m1 = wave_sum(g0 + g1) * (1.0f / D);
    const float m2 = wave_sum(g0 * xh0 + g1 * xh1) * (1.0f / D);
    dx0 = rstd * (g0 - m1 - xh0 * m2);
    dx1 = rstd * (g1 - m1 - xh1 * m2);
}

// ---------------------------------------------------------------------------------------------------------
// tile I/O
// ---------------------------------------------------------------------------------------------------------
// loads `nrows` rows x 128 floats starting at global row `row0` into LDS (stride LDP); rows outside [0, R) -> 0
__device__ __forceinline__ void load_tile128(float* __restrict__ dst, const float* __restrict__ src, int row0, int nrows,
                                             int R) {
    for (int e = threadIdx.x; e < nrows * (D / 4); e += NTHREADS) {
        const int rr = e >> 5, c4 = e & 31;
        const int r = row0 + rr;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r >= 0 && r < R) v = *reinterpret_cast<const float4*>(src + (size_t)r * D + c4 * 4);
        *reinterpret_cast<float4*>(dst + rr * LDP + c4 * 4) = v;
    }
}

}  // namespace vsl
