// Device-side building blocks shared by every VSLNet kernel (gfx950 / CDNA4 only, wave64).
//
//  * fp32-in / fp32-accumulate MFMA (v_mfma_f32_32x32x2_f32, v_mfma_f32_16x16x4_f32): exact fp32, the only
//    matrix path that can meet the 1e-4 logit gate (SURVEY 7 "hard parts").
//  * "packed" weight layout: a weight used as the B operand of Y = A * W^T is stored as [K/8][N][8] so that one
//    wave-wide float4 load per 8 k-values is a fully coalesced 1 KiB read straight into registers (no LDS staging,
//    no barrier) and feeds 4 MFMAs.  See pack_index().
//  * row tiles: every row-wise kernel owns TILE_M = 32 consecutive rows of the flattened (B*L, 128) activation;
//    the tile lives in LDS with a +4 float row pad (ds_read_b128 conflict-free: row stride == 4 banks mod 64).
//  * counter-based dropout: keep(site seed + key, element) = drop_hash >= p * 2^32, reproducible in the backward.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>

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
constexpr int CATP = 4 * D + 4; // LDS row stride of the CQAttention concat tile
constexpr int MAX_LC = 40;      // max characters per word (LDS budget of k_embed_fwd: 8 words x char_dim x (MAX_LC + 4) floats)
constexpr int MAX_LQ = 128;     // max query words = the reference's own bound (queries are cut at max_pos_len = 128 words, data_gen.py:188).
                                // The CQAttention kernels keep a sample's whole query in LDS; above 96 words (no dataset: ActivityNet's longest
                                // query has 82, TACoS' 64 -- SURVEY 8d) they switch to leaner layouts (CQ_BIG_LQ below)
constexpr int CQ_BIG_LQ = 96;   // Lq > CQ_BIG_LQ: one wave per 32-word tile instead of four K-partial tiles, aliased / unstaged buffers
constexpr int MAX_L = 1024;     // max clips per video (tested limit; the attention kernels stream K/V in 256-row blocks)

// ---------------------------------------------------------------------------------------------------------
// dropout
// ---------------------------------------------------------------------------------------------------------
struct Drop {
    uint32_t seed;     // per-site seed (host mixes step, site id; a shard's sample offset is folded in additively)
    uint32_t thresh;   // p * 2^32 ; 0 => dropout disabled (eval / drop_rate 0)
    float scale;       // 1 / (1 - p)
    uint32_t key;      // second per-site word, xor-ed in BETWEEN the two multiply rounds: without it the masks of two sites (or steps) would
                       // be shifted windows of one 2^32-periodic sequence (element * odd + seed is a translation)
};

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
// round-to-nearest-even fp32 -> bfloat16 bits (finite inputs), and the exact widening back
__host__ __device__ __forceinline__ uint16_t f32_to_bf16(float v) {
    uint32_t b;
    memcpy(&b, &v, 4);
    b += 0x7FFFu + ((b >> 16) & 1u);
    return (uint16_t)(b >> 16);
}
// the keep decision's hash: murmur3's finaliser over (element * golden + seed) with the site key folded into the middle xor
// (v_xor3_b32: no extra instruction) -- WITHOUT the finaliser's last xor-shift (round 5): the decision is `hash >= p * 2^32`, the
// dropped step only touches the low 16 bits, i.e. it could change a decision for 2^-16 of the hashes; two instructions less at each of
// the ~130 M elements a training step hashes (forward + recomputation in the backward; ~17 % of the step's vector instructions are this hash)
__device__ __forceinline__ uint32_t drop_hash(uint32_t idx, uint32_t seed, uint32_t key) {
    uint32_t h = idx * 0x9E3779B1u + seed;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= (h >> 13) ^ key; h *= 0xC2B2AE35u;
    return h;
}
// multiplier applied to element `idx` of the site: 0 or 1/(1-p)
__device__ __forceinline__ float drop_mul(const Drop& d, uint32_t idx) {
    if (d.thresh == 0u) return 1.0f;
    return drop_hash(idx, d.seed, d.key) >= d.thresh ? d.scale : 0.0f;
}

// TWO decisions from one hash: elements 2 j and 2 j + 1 of a row share drop_hash(row * ceil(n / 2) + j); the odd one compares the hash rotated by
// 16 bits (its upper half decides for the even element, its lower half for the odd one; p stays exact to 2^-32).  Used by the attention kernels
// of L > 256 for the probability site -- L^2 decisions per head and sample, the hash was a third of their key loop's vector work
// (profiles/r05_notes.md section 7) -- and restated in tests/helpers.py (hip_dropout).
__device__ __forceinline__ uint32_t drop_hash_odd(uint32_t h) { return __builtin_amdgcn_alignbit(h, h, 16); }

// same multiplier WITHOUT the "dropout enabled?" test: for call sites that have already branched on d.thresh (block-uniform)
__device__ __forceinline__ float drop_keep_scale(const Drop& d, uint32_t idx) {
    return drop_hash(idx, d.seed, d.key) >= d.thresh ? d.scale : 0.0f;
}

// ---------------------------------------------------------------------------------------------------------
// XCD-aware block order.  Workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with a private L2.  The
// attention kernels read 64-byte head slices out of 512-byte rows, so the 8 heads of a sample should share an L2: with
// the default order they sit on 8 different XCDs and every row is fetched 8 times.  xcd_swizzle gives every XCD a
// contiguous chunk of the LOGICAL block order (x fastest, then y, then z): chunk = whole samples when the grid is a
// multiple of 8 workgroups, identity otherwise.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void xcd_swizzle(int& bx, int& by, int& bz) {
    const int nx = gridDim.x, ny = gridDim.y, nwg = nx * ny * gridDim.z;
    if (nwg & 7) { bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z; return; }
    const int bid = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const int l = (bid & 7) * (nwg >> 3) + (bid >> 3);
    bx = l % nx;
    const int r = l / nx;
    by = r % ny;
    bz = r / ny;
}

// ---------------------------------------------------------------------------------------------------------
// gate non-linearities of the LSTM kernels on the hardware exp / rcp instructions (about 1 ulp each).  fp32 MFMAs and the
// vector ALU share the SIMD, so the ~30-instruction libm expf / tanhf cost as much per step as the recurrent matmul itself.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoid_fast(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 2.0f * sigmoid_fast(2.0f * x) - 1.0f; }

// ---------------------------------------------------------------------------------------------------------
// Cross-lane exchanges of the butterfly reductions WITHOUT the LDS crossbar: __shfl_xor compiles to ds_bpermute_b32 (an LDS round
// trip of 100+ cycles per step, and every LayerNorm row is two 3-step reductions); DPP modifiers and gfx950's permlane swaps do the
// same exchanges in the vector ALU.  Each helper returns the partner's value of the named butterfly step, so `v op= xorN(v)` is
// bit-identical to the shuffle version:
//   xor 1, 2      quad_perm (exact)
//   xor 4         row_half_mirror: lane i <- 7 - i of its 8-lane group, = lane i ^ 4 once the quads are uniform (after the xor 1, 2 steps)
//                 -- or row_ror:4 once lanes i and i ^ 8 agree (descending butterflies, after the xor 8 step)
//   xor 8         row_ror:8 (exact)
//   xor 16, 32    v_permlane16_swap / v_permlane32_swap of the value with itself: one result holds the lane's own value, the other
//                 the partner's (which is which depends on the lane; + and max are commutative)
// ---------------------------------------------------------------------------------------------------------
#define VSL_DPP(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true))
__device__ __forceinline__ float lane_xor1(float v) { return VSL_DPP(v, 0xB1); }
__device__ __forceinline__ float lane_xor2(float v) { return VSL_DPP(v, 0x4E); }
__device__ __forceinline__ float lane_half_mirror(float v) { return VSL_DPP(v, 0x141); }
__device__ __forceinline__ float lane_ror4(float v) { return VSL_DPP(v, 0x124); }
__device__ __forceinline__ float lane_xor8(float v) { return VSL_DPP(v, 0x128); }
template <class OP> __device__ __forceinline__ float lane_pair16(float v, OP op) {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return op(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
template <class OP> __device__ __forceinline__ float lane_pair32(float v, OP op) {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return op(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// wave reductions (64 lanes), butterfly 32, 16, 8, 4, 2, 1 as before
__device__ __forceinline__ float wave_sum(float v) {
    auto add = [](float a, float b) { return a + b; };
    v = lane_pair32(v, add);
    v = lane_pair16(v, add);
    v += lane_xor8(v); v += lane_ror4(v); v += lane_xor2(v); v += lane_xor1(v);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    auto mx = [](float a, float b) { return fmaxf(a, b); };
    v = lane_pair32(v, mx);
    v = lane_pair16(v, mx);
    v = fmaxf(v, lane_xor8(v)); v = fmaxf(v, lane_ror4(v)); v = fmaxf(v, lane_xor2(v)); v = fmaxf(v, lane_xor1(v));
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
// Same GEMM with the packed-weight fragments software-pipelined through registers: KB k-blocks (8 k each) are
// fetched per stage, the next stage is in flight while the current one feeds the MFMAs.  The first stage can be
// issued at kernel entry (bfrag_load) so that its L2/HBM latency hides under the kernel's LDS prologue.
template <int NT, int KB>
struct BFrag { float4 b[NT][KB]; };

// k-block order: every workgroup contracts k in the same order, so a row's result does not depend on which tile (and, under
// data parallelism, which shard) it lands in.  (Round 1 rotated the order per workgroup to spread L2 traffic: no measured
// gain, and it made the rounding of a row depend on blockIdx.)
__device__ __forceinline__ int kb_rot(int) { return 0; }
template <int NT, int KB>
__device__ __forceinline__ void bfrag_load(BFrag<NT, KB>& f, const float* __restrict__ Bp, int ncols, int col0, int cstep,
                                           int kb0, int nkb) {
    const int lane = threadIdx.x & 63;
    const float4* bp = reinterpret_cast<const float4*>(Bp) + (size_t)(col0 + (lane & 31)) * 2 + (lane >> 5);
    if (kb0 + KB <= nkb) {                 // full stage: straight-line loads (all in flight together)
        const int rot = (nkb % KB == 0) ? kb_rot(nkb) : 0;
#pragma unroll
        for (int q = 0; q < KB; ++q) {
            int kb = kb0 + q + rot;
            kb = kb >= nkb ? kb - nkb : kb;
#pragma unroll
            for (int t = 0; t < NT; ++t) f.b[t][q] = bp[((size_t)kb * ncols + t * cstep) * 2];
        }
    } else {
#pragma unroll
        for (int q = 0; q < KB; ++q)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                f.b[t][q] = kb0 + q < nkb ? bp[((size_t)(kb0 + q) * ncols + t * cstep) * 2] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <int NT, int KB>
__device__ __forceinline__ void mma_stage(const float* __restrict__ arow, int kb0, int rot, int nkb, const BFrag<NT, KB>& cur,
                                          f32x16 (&acc)[NT]) {
    constexpr int AB = KB < 8 ? KB : 8;   // A fragments are read from LDS in batches of up to 8 blocks
#pragma unroll
    for (int q0 = 0; q0 < KB; q0 += AB) {
        float4 a[AB];
#pragma unroll
        for (int q = 0; q < AB; ++q) {
            int kb = kb0 + q0 + q + rot;
            kb = kb >= nkb ? kb - nkb : kb;
            a[q] = *reinterpret_cast<const float4*>(arow + kb * 8);
        }
#pragma unroll
        for (int q = 0; q < AB; ++q)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 b = cur.b[t][q0 + q];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, b.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, b.z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, b.w, acc[t], 0, 0, 0);
            }
    }
}

template <int NT, int KB>
__device__ __forceinline__ void gemm32p(const float* __restrict__ As, int lda, int K, const float* __restrict__ Bp,
                                        int ncols, int col0, int cstep, f32x16 (&acc)[NT], BFrag<NT, KB>& cur) {
    const int lane = threadIdx.x & 63;
    const float* arow = As + (lane & 31) * lda + 4 * (lane >> 5);
    const int nkb = K >> 3;
    const int nfull = nkb / KB;            // stages whose KB k-blocks all exist: straight-line code, no per-block guard
    const int rot = (nkb % KB == 0) ? kb_rot(nkb) : 0;
    // Two fragment sets used alternately (no register copies: fp32 MFMAs and v_mov share the SIMD), with scheduling
    // barriers so that the loads of the next stage stay in front of the MFMAs of the current one -- left alone the
    // compiler sinks every load to just before its first use and waits on it (load -> vmcnt(0) -> 4 MFMAs).
    BFrag<NT, KB> nxt;
    for (int kb0 = 0; kb0 + KB <= nkb; kb0 += 2 * KB) {
        const bool more_a = kb0 + KB < nkb;
        if (more_a) bfrag_load(nxt, Bp, ncols, col0, cstep, kb0 + KB, nkb);
        __builtin_amdgcn_sched_barrier(0);
        mma_stage<NT, KB>(arow, kb0, rot, nkb, cur, acc);
        __builtin_amdgcn_sched_barrier(0);
        if (kb0 + 2 * KB <= nkb) {
            if (kb0 + 2 * KB < nkb) bfrag_load(cur, Bp, ncols, col0, cstep, kb0 + 2 * KB, nkb);
            __builtin_amdgcn_sched_barrier(0);
            mma_stage<NT, KB>(arow, kb0 + KB, rot, nkb, nxt, acc);
            __builtin_amdgcn_sched_barrier(0);
        } else if (more_a) {                 // odd number of full stages and a ragged tail: its fragments sit in nxt
#pragma unroll
            for (int q = 0; q < KB; ++q)
#pragma unroll
                for (int t = 0; t < NT; ++t) cur.b[t][q] = nxt.b[t][q];
        }
    }
    for (int kb = nfull * KB; kb < nkb; ++kb) {         // ragged tail (K = 400): fragment q = kb - nfull*KB was zero-filled beyond nkb
        const float4 a = *reinterpret_cast<const float4*>(arow + kb * 8);
        const int q = kb - nfull * KB;
#pragma unroll
        for (int qq = 0; qq < KB; ++qq)
            if (qq == q) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float4 b = cur.b[t][qq];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[t], 0, 0, 0);
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// fp32-grade products on the BF16 matrix cores (round 3; tools/ubench/split_bf16.hip priced it).
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the fp32-input MFMA and overlaps with vector-ALU work.  Every fp32 operand x is split
// EXACTLY into three bfloat16 terms x = h + m + l (round-to-nearest at each level: |m| <= 2^-8 |x|, |l| <= 2^-16 |x|; l is exact because
// at most 8 significant bits remain) and the six products of weight >= 2^-16 -- hh, hm, mh, hl, lh, mm -- are accumulated in fp32 by the
// MFMA; the dropped ml, lm, ll are <= 2^-23 |a||b|, the class of ONE fp32 rounding of the product.  Measured against fp64 (R = 8192 rows):
// error / max sum|a||b| = 1.6e-8 against 2.2e-8 for the fp32 MFMA chain.
// Kernels that use these helpers live in translation units built with -fno-slp-vectorize (vslnet_amd/build.py): the SLP vectoriser turns
// the subtractions into v_pk_add_f32 + v_mov packing, which is slower beside MFMAs.
// ---------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
template <int I0, int I1, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I0 < I1) { f(std::integral_constant<int, I0>()); static_for<I0 + 1, I1>(f); }
}
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {       // {bf16(a) low half, bf16(b) high half}, RNE: v_cvt_pk_bf16_f32
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// the same instruction as an opaque asm: for call sites where the vector form above makes clang assemble the operand pair through
// scratch memory (the one-product instantiation of k_wgrad4: a masked row pair became a stack shuffle, 19 -> 47 us)
__device__ __forceinline__ uint32_t cvt_pk_bf16_asm(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (x0, x1) -> packed pairs of the three terms: 11 vector instructions
__device__ __forceinline__ void split3(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xFFFF0000u);
    m = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xFFFF0000u);
    l = cvt_pk_bf16(s0, s1);
}
__device__ __forceinline__ f32x16 mfma_bf16(u32x4_t a, u32x4_t b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16_bf16(u32x4_t a, u32x4_t b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// [rows][16] bf16 plane of a head slice (the attention kernels of L > 256): element offset of dims [8 half, 8 half + 8) of row `row`.  The two 16-byte
// halves of a row swap places every 8 rows, so the 16 lanes of a ds_read_b128 phase (16 consecutive rows, one half) cover all 64 banks once without padding
__device__ __forceinline__ int af_kp(int row, int half) { return row * 16 + ((half ^ ((row >> 3) & 1)) << 3); }
// host-side / k_pack: the three terms of one value (bf16 bit patterns)
__host__ __device__ __forceinline__ float bf16_to_f32(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
__host__ __device__ __forceinline__ void split3_scalar(float x, uint16_t& h, uint16_t& m, uint16_t& l) {
    h = f32_to_bf16(x);
    const float r = x - bf16_to_f32(h);
    m = f32_to_bf16(r);
    l = f32_to_bf16(r - bf16_to_f32(m));
}
// "split pack" of a weight used as the B operand of Y = A Bm: three bf16 planes (h, m, l), each [K16 / 16][ncols][16] -- lane (col, half)
// of v_mfma_f32_32x32x16_bf16 (or lane (col, k group) of 16x16x32) reads its 8 consecutive k as one 16-byte load, a wave reads 1 KiB.
__host__ __device__ __forceinline__ size_t pack3_plane(int K, int ncols) { return (size_t)((K + 15) / 16) * ncols * 16; }     // bf16 elements
__host__ __device__ __forceinline__ size_t pack3_index(int k, int col, int ncols) { return ((size_t)(k >> 4) * ncols + col) * 16 + (k & 15); }
__host__ __device__ __forceinline__ size_t pack3_floats(int K, int ncols) { return (3 * pack3_plane(K, ncols) + 1) / 2; }

// ---------------------------------------------------------------------------------------------------------
// Row-tile GEMM on the bf16 matrix cores from PRE-SPLIT operands:  acc[t] (32 x 32) += A[32][K] * Bm[:, col0 + t * cstep + 0..31]
//   A : three bf16 planes (terms h, m, l) of the 32-row tile in LDS, `ldb` elements per row (a multiple of 8 with ldb * 2 % 128 == 16:
//       conflict-free ds_read_b128), `ps` elements between planes.  The kernel splits every value ONCE where it stages the tile (a per-wave
//       split in the GEMM loop does not pay: profiles/r03_notes.md).
//   W3: split pack of Bm (PackJob type 6 / 7), K a multiple of 16.
// Lane (i, h) reads row i, k = 16 s + 8 h .. + 7 of each plane (one step ahead) and column col0 + i of each weight plane (GS3_NB steps ahead
// in a register ring); 6 MFMAs per step and tile, two alternating accumulators when there is one tile.
// ---------------------------------------------------------------------------------------------------------
struct Frag3 { u32x4_t t[3]; };
constexpr int GS3_NB = 4;
template <int NT> struct Ring3 { Frag3 b[GS3_NB][NT]; };
// weight fragments of K = 16 step s (past the end: a harmless re-read of the last step)
template <int NT>
__device__ __forceinline__ void ring3_load(Frag3 (&f)[NT], const uint16_t* __restrict__ W3, int K, int ncols, int col0, int cstep, int s) {
    const int lane = threadIdx.x & 63, ns = K >> 4;
    const size_t plane = pack3_plane(K, ncols);
    const uint16_t* p = W3 + ((size_t)min(s, ns - 1) * ncols + col0 + (lane & 31)) * 16 + 8 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 3; ++q) f[t].t[q] = *reinterpret_cast<const u32x4_t*>(p + q * plane + (size_t)t * cstep * 16);
}
// the first GS3_NB steps: a kernel calls this at its entry, so that the fragments' latency hides under the staging of the A tile
template <int NT>
__device__ __forceinline__ void ring3_prefetch(Ring3<NT>& r, const uint16_t* __restrict__ W3, int K, int ncols, int col0, int cstep) {
    static_for<0, GS3_NB>([&](auto uc) { ring3_load<NT>(r.b[decltype(uc)::value], W3, K, ncols, col0, cstep, decltype(uc)::value); });
}
template <int NT>
__device__ __forceinline__ void gemm32pl(const uint16_t* __restrict__ Ap, int ldb, int ps, int K, const uint16_t* __restrict__ W3, int ncols,
                                         int col0, int cstep, f32x16 (&acc)[NT], Ring3<NT>& ring) {
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const uint16_t* ar = Ap + i * ldb + 8 * h;
    const int ns = K >> 4;
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    Frag3 a[2];
    auto aread = [&](int s, Frag3& f) {
        const uint16_t* p = ar + 16 * min(s, ns - 1);
#pragma unroll
        for (int q = 0; q < 3; ++q) f.t[q] = *reinterpret_cast<const u32x4_t*>(p + q * ps);
    };
    aread(0, a[0]);
    for (int s0 = 0; s0 < ns; s0 += GS3_NB)
        static_for<0, GS3_NB>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            const int s = s0 + u;
            if (s < ns) {
                aread(s + 1, a[(u + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};       // (a term, b term): mm, hl, lh, hm, mh, hh
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if (NT == 1 && (p & 1)) acc2 = mfma_bf16(a[u & 1].t[TA[p]], ring.b[u][t].t[TB[p]], acc2);
                        else acc[t] = mfma_bf16(a[u & 1].t[TA[p]], ring.b[u][t].t[TB[p]], acc[t]);
                    }
                __builtin_amdgcn_sched_barrier(0);
                ring3_load<NT>(ring.b[u], W3, K, ncols, col0, cstep, s + GS3_NB);
            }
        });
    if (NT == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] += acc2[r];
    }
}
template <int NT>
__device__ __forceinline__ void gemm32pl(const uint16_t* __restrict__ Ap, int ldb, int ps, int K, const uint16_t* __restrict__ W3, int ncols,
                                         int col0, int cstep, f32x16 (&acc)[NT]) {
    Ring3<NT> ring;
    ring3_prefetch<NT>(ring, W3, K, ncols, col0, cstep);
    gemm32pl<NT>(Ap, ldb, ps, K, W3, ncols, col0, cstep, acc, ring);
}
// stores a float4 of a row tile (row r, columns c .. c + 3) into the three planes
__device__ __forceinline__ void split_store4(uint16_t* __restrict__ P0, int ldb, int ps, int r, int c, const float4& v) {
    uint32_t h0, m0, l0, h1, m1, l1;
    split3(v.x, v.y, h0, m0, l0);
    split3(v.z, v.w, h1, m1, l1);
    uint16_t* d = P0 + r * ldb + c;
    *reinterpret_cast<u32x2_t*>(d) = u32x2_t{h0, h1};
    *reinterpret_cast<u32x2_t*>(d + ps) = u32x2_t{m0, m1};
    *reinterpret_cast<u32x2_t*>(d + 2 * ps) = u32x2_t{l0, l1};
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

// Same product for a compile-time row count, software-pipelined by hand: the fragments of the next KS k-steps are
// requested from LDS before the MFMAs of the current KS are issued (register double buffer).  With one wave per SIMD
// nothing else hides the ds_read latency; left to itself the compiler waits on lgkmcnt(0) before every MFMA pair
// (measured 6.5k cycles per 32-row step against 4.1k of MFMA issue).
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
// The hook is the caller's traffic for LATER tiles (global loads, mask arithmetic, LDS stores).  It has no dependence on
// the MFMAs of its batch, and a wave stalled at the issue of an MFMA cannot run the VALU code behind it, so the hook is
// woven between the MFMAs explicitly: after every MFMA up to 2 LDS reads, VPM vector-ALU instructions, one global load and
// one LDS write are scheduled (sched_group_barrier).  Measured on k_wgrad: 5.9k -> 4.7k cycles per 32-row step.
template <int NT, int ROWS, int KS = 4, int VPM = 0, class Hook = NoHook>
__device__ __forceinline__ void gemm_tn_p(const float* __restrict__ Gs, int ldg, int n0, const float* __restrict__ As,
                                          int lda, int k0, f32x16 (&acc)[NT], Hook&& hook = Hook()) {
    static_assert(ROWS % (2 * KS) == 0, "ROWS must be a multiple of 2*KS");
    constexpr int NB = ROWS / (2 * KS);
    const int lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const float* g = Gs + h * ldg + n0 + i;
    const float* a = As + h * lda + k0 + i;
    float gq[2][KS], aq[2][KS][NT];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        gq[0][s] = g[2 * s * ldg];
#pragma unroll
        for (int t = 0; t < NT; ++t) aq[0][s][t] = a[2 * s * lda + 32 * t];
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        __builtin_amdgcn_sched_barrier(0);
        if (b + 1 < NB) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int r = 2 * ((b + 1) * KS + s);
                gq[(b + 1) & 1][s] = g[r * ldg];
#pragma unroll
                for (int t = 0; t < NT; ++t) aq[(b + 1) & 1][s][t] = a[r * lda + 32 * t];
            }
        }
        hook(b);
        if (VPM == 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gq[b & 1][s], aq[b & 1][s][t], acc[t], 0, 0, 0);
        }
        if (VPM > 0) {
#pragma unroll
            for (int m = 0; m < KS * NT; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // LDS reads of the next batch
                __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);    // hook arithmetic
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // hook global load
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // hook LDS store
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
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
// Tile-wide LayerNorm: 8 consecutive lanes own one row (lane sub = tid & 7 holds the float4 columns
// sub*4 + 32*j, j = 0..3), 32 rows per pass of the 256-thread workgroup -> two 3-step DPP reductions per row
// instead of a 6-step wave reduction per row executed serially by one wave.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float grp8_sum(float v) {
    v += lane_xor1(v); v += lane_xor2(v); v += lane_half_mirror(v);
    return v;
}
__device__ __forceinline__ float sum4(const float4& v) { return (v.x + v.y) + (v.z + v.w); }

// in place: tile[r][:] <- LN(tile[r][:]) * drop      for r < nrows (row stride ld floats).  `drow0` = global row of
// tile row 0 for the dropout element index (row * 128 + col).
__device__ __forceinline__ void ln_tile(float* tile, int nrows, int ld, const float* __restrict__ g,
                                        const float* __restrict__ b, const Drop& dp, int drow0) {
    const int sub = threadIdx.x & 7;
    float4 gv[4], bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        gv[j] = *reinterpret_cast<const float4*>(g + sub * 4 + 32 * j);
        bv[j] = *reinterpret_cast<const float4*>(b + sub * 4 + 32 * j);
    }
    for (int r = threadIdx.x >> 3; r < nrows; r += NTHREADS / 8) {
        float* row = tile + r * ld + sub * 4;
        float4 v[4];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = *reinterpret_cast<const float4*>(row + 32 * j); s += sum4(v[j]); }
        const float mu = grp8_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j].x -= mu; v[j].y -= mu; v[j].z -= mu; v[j].w -= mu;
            q += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
        }
        const float rstd = rsqrtf(grp8_sum(q) * (1.0f / D) + LN_EPS);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 o;
            o.x = v[j].x * rstd * gv[j].x + bv[j].x; o.y = v[j].y * rstd * gv[j].y + bv[j].y;
            o.z = v[j].z * rstd * gv[j].z + bv[j].z; o.w = v[j].w * rstd * gv[j].w + bv[j].w;
            if (dp.thresh) {
                const uint32_t base = (uint32_t)((drow0 + r) * D + sub * 4 + 32 * j);
                o.x *= drop_mul(dp, base); o.y *= drop_mul(dp, base + 1); o.z *= drop_mul(dp, base + 2); o.w *= drop_mul(dp, base + 3);
            }
            *reinterpret_cast<float4*>(row + 32 * j) = o;
        }
    }
}

// LayerNorm backward on a 32-row tile.
//   Ts : grad wrt the LN output (32 x 128, stride LDP) -- left in place (beta partial = its column sums)
//   Xs : raw LN input rows (32 x 128, stride LDP)       -- overwritten with dy * xhat (gamma partial = column sums)
//   out[r] = LN^T(Ts[r]) + resid[r] + extra[r]  for global rows r0 + rr < R (out nullable: the result then only goes to lds_out) ; partial slabs [blockIdx.x][128].
//   lds_out (nullable): the result tile is also left in LDS (stride LDP; rows >= R zero) for a fused follow-up GEMM.
// The residual rows (resid + resid2) a thread adds in ln_bwd_tile: requested at kernel entry so that their memory latency is
// not paid in the middle of the kernel.  Rows >= R are clamped for the load and zeroed.
struct LnResid { float4 v[4]; };
__device__ __forceinline__ void ln_resid_prefetch(LnResid& rs, const float* __restrict__ resid, const float* __restrict__ resid2,
                                                  int r0, int R) {
    const int sub = threadIdx.x & 7, r = r0 + (threadIdx.x >> 3);
    const size_t off = (size_t)min(r, R - 1) * D + sub * 4;
    const float keep = r < R ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (resid) a = *reinterpret_cast<const float4*>(resid + off + 32 * j);
        if (resid2) {
            const float4 e = *reinterpret_cast<const float4*>(resid2 + off + 32 * j);
            a.x += e.x; a.y += e.y; a.z += e.z; a.w += e.w;
        }
        rs.v[j] = make_float4(a.x * keep, a.y * keep, a.z * keep, a.w * keep);
    }
}
// active = false: a thread of a wider workgroup that only joins the barrier (tile bodies hosted by a 512-thread kernel: tile_bodies.hpp)
__device__ __forceinline__ void ln_bwd_tile(float* Ts, float* Xs, const LnResid& pre, const float* __restrict__ ln_g,
                                            float* __restrict__ out, float* __restrict__ p_lng, float* __restrict__ p_lnb,
                                            int r0, int R, float* lds_out = nullptr, bool active = true) {
    const int tid = threadIdx.x, sub = tid & 7, rr = tid >> 3;
    const int r = r0 + rr;
    if (active) {
        float* xr = Xs + rr * LDP + sub * 4;
        const float* tr = Ts + rr * LDP + sub * 4;
        float4 x[4], dy[4], rs[4];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x[j] = *reinterpret_cast<const float4*>(xr + 32 * j);
            dy[j] = *reinterpret_cast<const float4*>(tr + 32 * j);
            rs[j] = pre.v[j];
            s += sum4(x[j]);
        }
        const float mu = grp8_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x[j].x -= mu; x[j].y -= mu; x[j].z -= mu; x[j].w -= mu;
            q += x[j].x * x[j].x + x[j].y * x[j].y + x[j].z * x[j].z + x[j].w * x[j].w;
        }
        const float rstd = rsqrtf(grp8_sum(q) * (1.0f / D) + LN_EPS);
        float m1 = 0.f, m2 = 0.f;
        float4 gd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 gv = *reinterpret_cast<const float4*>(ln_g + sub * 4 + 32 * j);
            x[j].x *= rstd; x[j].y *= rstd; x[j].z *= rstd; x[j].w *= rstd;                       // xhat
            gd[j] = make_float4(dy[j].x * gv.x, dy[j].y * gv.y, dy[j].z * gv.z, dy[j].w * gv.w);  // dy * gamma
            m1 += sum4(gd[j]);
            m2 += gd[j].x * x[j].x + gd[j].y * x[j].y + gd[j].z * x[j].z + gd[j].w * x[j].w;
        }
        m1 = grp8_sum(m1) * (1.0f / D);
        m2 = grp8_sum(m2) * (1.0f / D);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 o;
            o.x = rstd * (gd[j].x - m1 - x[j].x * m2) + rs[j].x; o.y = rstd * (gd[j].y - m1 - x[j].y * m2) + rs[j].y;
            o.z = rstd * (gd[j].z - m1 - x[j].z * m2) + rs[j].z; o.w = rstd * (gd[j].w - m1 - x[j].w * m2) + rs[j].w;
            if (out && r < R) *reinterpret_cast<float4*>(out + (size_t)r * D + sub * 4 + 32 * j) = o;
            const bool ok = r < R;
            if (lds_out) *reinterpret_cast<float4*>(lds_out + rr * LDP + sub * 4 + 32 * j) = ok ? o : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(xr + 32 * j) = ok ? make_float4(dy[j].x * x[j].x, dy[j].y * x[j].y, dy[j].z * x[j].z, dy[j].w * x[j].w)
                                                         : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    if (active) {
        const int c = tid & 127;
        const float* src = tid < 128 ? Xs : Ts;
        float acc = 0.f;
#pragma unroll 8
        for (int i = 0; i < TILE_M; ++i) acc += src[i * LDP + c];
        if (tid < 128) p_lng[(size_t)blockIdx.x * D + c] = acc;
        else p_lnb[(size_t)blockIdx.x * D + c] = acc;
    }
}

// ---------------------------------------------------------------------------------------------------------
// tile I/O
// ---------------------------------------------------------------------------------------------------------
// loads `nrows` (<= 40) rows x 128 floats starting at global row `row0` into LDS (stride LDP); rows outside [0, R) -> 0.
// All of a thread's 16-byte loads are issued before the first LDS store, so the tile costs ONE memory latency.
__device__ __forceinline__ void load_tile128(float* __restrict__ dst, const float* __restrict__ src, int row0, int nrows,
                                             int R) {
    float4 v[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int e = threadIdx.x + q * NTHREADS;
        const int r = row0 + (e >> 5);
        v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < nrows * 32 && r >= 0 && r < R) v[q] = *reinterpret_cast<const float4*>(src + (size_t)r * D + (e & 31) * 4);
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int e = threadIdx.x + q * NTHREADS;
        if (e < nrows * 32) *reinterpret_cast<float4*>(dst + (e >> 5) * LDP + (e & 31) * 4) = v[q];
    }
}

// Reductions over the four lanes {i, i+16, i+32, i+48} of a wave (the k-groups of one MFMA column), result in all four: gfx950's
// v_permlane32_swap / v_permlane16_swap exchange half-waves / odd-even rows in the VALU -- no LDS round trip like ds_bpermute.
__device__ __forceinline__ float kgroup_max(float x) {
    const unsigned u = __float_as_uint(x);
    auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned v = __float_as_uint(fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1])));
    auto b = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float kgroup_sum(float x) {
    const unsigned u = __float_as_uint(x);
    auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned v = __float_as_uint(__uint_as_float(a[0]) + __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

}  // namespace vsl
