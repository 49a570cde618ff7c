// Row-tile GEMM kernels on the BF16 matrix cores at fp32 grade (3-way operand split, 6 products: common.hpp).  gfx950 only.
// Built with -fno-slp-vectorize (vslnet_amd/build.py).
#include "common.hpp"
#include "launch.hpp"
#include <type_traits>

namespace vsl {

// =========================================================================================================
// a2  VisualProjection (/root/reference/model/layers_t7.py:105-115):  Y = drop(X) W^T + b,  X (R, Dv) streamed from HBM once.
// 32-row tile per workgroup, K streamed in 128-wide chunks through LDS: three planes (the bf16 terms h, m, l) of [32][128], double buffered.
// WAVE ROLES (8 waves, two per SIMD -- waves w and w + 4 share a SIMD):
//   producers (waves 0-3): stream the tile's rows from HBM (three chunks ahead in registers), apply the dropout, split every value ONCE into
//                          its three terms and store them into the planes of the next chunk -- vector-ALU work, HBM loads only;
//   consumers (waves 4-7): wave = 32 output columns; per K = 16 step 3 A fragments from the planes (ds_read_b128 = the lane's 8 consecutive
//                          k), 3 B fragments from the split pack of the weight (L2, 16 bytes per lane, one chunk ahead) and 6 MFMAs on two
//                          alternating accumulators -- matrix work, L2 loads only.
// One barrier per chunk.  Why roles: bf16 MFMAs of one wave and vector work of ANOTHER wave overlap on a SIMD (tools/ubench/split_bf16.hip),
// and vmcnt retires in order -- in a wave that issues both, every wait for a weight fragment (L2) also waits for the row loads (HBM) issued
// before it, so row loads never got more time than the weight prefetch distance.  Single-role version (profiles/r03_notes.md): 21.7 us =
// 7.2 skeleton + 3.6 rows + 4.7 weights + 3.4 arithmetic + MFMAs, all additive, the same with two such waves per SIMD.  Round 2 (fp32-input
// MFMA): 41 us.
// =========================================================================================================
#ifdef VP3_STAMPS      // harness builds (tools/ubench/vproj_harness.hip): cycle stamps of workgroup 0, [wave][stamp]
constexpr int VP3_NST = 24;
__device__ long long g_vp_stamps[8][VP3_NST];
#define VPSTAMP(k) do { if (blockIdx.x == 0 && lane == 0 && (k) < VP3_NST) g_vp_stamps[w][k] = clock64(); } while (0)
#else
#define VPSTAMP(k) do { } while (0)
#endif
constexpr int VP3_KC = 128;
constexpr int VP3_LD = VP3_KC + 8;      // bf16 elements per LDS row (272 B: 16-byte aligned rows, conflict-free b128 reads)
constexpr int VP3_T = 512;
#ifndef VP3_NB
#define VP3_NB 4         // weight fragments: K = 16 steps ahead (a divisor of 8)
#endif
// FULL: R % 32 == 0, Dv % 128 == 0, no row mapping -> no guards anywhere (every BASELINE shape) ; DROPON: dropout enabled (block-uniform)
template <bool FULL, bool DROPON>
__global__ __launch_bounds__(VP3_T, 2) void k_vproj_fwd3(const float* __restrict__ X, const uint16_t* __restrict__ W3, const float* __restrict__ bias,
                                                         float* __restrict__ Y, int R, int Dv, Drop dp, int seg, int stride, int off) {
    constexpr int NKS = VP3_KC / 16;
    __shared__ __attribute__((aligned(16))) uint16_t As[2][3][TILE_M * VP3_LD];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r0 = blockIdx.x * TILE_M;
    // seg > 0 (GEMM use by the rnn head, no dropout): logical row r = physical row (r / seg) * stride + off + r % seg of X and Y
    auto phys = [&](int r) { return (!FULL && seg > 0) ? (r / seg) * stride + off + r % seg : r; };
    const int nchunk = (Dv + VP3_KC - 1) / VP3_KC;
    if (w < 4) {
        // ------------------------------------------------------------------ producers: thread = rows (tid >> 5) + 8 q, columns (tid & 31) * 4 ..
        struct Stage { float4 v[4]; };
        const float* xrow[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) xrow[q] = X + (size_t)phys(min(r0 + (tid >> 5) + 8 * q, R - 1)) * Dv + (tid & 31) * 4;
        auto gload = [&](int ch, Stage& st) {
            ch = min(ch, nchunk - 1);                     // past the end: a harmless re-read
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (FULL) st.v[q] = *reinterpret_cast<const float4*>(xrow[q] + ch * VP3_KC);
                else {
                    const int c = (tid & 31) * 4 + ch * VP3_KC;
                    st.v[q] = (r0 + (tid >> 5) + 8 * q < R && c < Dv) ? *reinterpret_cast<const float4*>(xrow[q] + ch * VP3_KC) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        };
        auto stage = [&](int ch, const Stage& st) {       // dropout + split + three 8-byte LDS stores per float4
            const int buf = ch & 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = (tid >> 5) + 8 * q, cl = (tid & 31) * 4;
                float4 v = st.v[q];
                if (DROPON) {
                    const uint32_t base = (uint32_t)((size_t)(r0 + rr) * Dv + cl + ch * VP3_KC);
                    v.x *= drop_keep_scale(dp, base); v.y *= drop_keep_scale(dp, base + 1);
                    v.z *= drop_keep_scale(dp, base + 2); v.w *= drop_keep_scale(dp, base + 3);
                }
                {
                    uint32_t h0, m0, l0, h1, m1, l1;
                    split3(v.x, v.y, h0, m0, l0);
                    split3(v.z, v.w, h1, m1, l1);
                    *reinterpret_cast<u32x2_t*>(&As[buf][0][rr * VP3_LD + cl]) = u32x2_t{h0, h1};
                    *reinterpret_cast<u32x2_t*>(&As[buf][1][rr * VP3_LD + cl]) = u32x2_t{m0, m1};
                    *reinterpret_cast<u32x2_t*>(&As[buf][2][rr * VP3_LD + cl]) = u32x2_t{l0, l1};
                }
            }
        };
        Stage st[4];                                      // rows of chunk c live in st[c & 3]
        VPSTAMP(0);
        gload(0, st[0]); gload(1, st[1]); gload(2, st[2]); gload(3, st[3]);
        stage(0, st[0]);
        gload(4, st[0]);
        VPSTAMP(1);
        __syncthreads();
        VPSTAMP(2);
        // iteration c: chunk c + 1 into buffer (c + 1) & 1 (the consumers left it at the previous barrier), rows of chunk c + 5 requested.
        // Whole groups of four run without conditions: a load whose use sits behind a condition is SUNK to the use by the compiler (past the
        // barrier -- the prefetch is gone).
        int c0 = 0;
        for (; c0 + 4 < nchunk; c0 += 4)
            static_for<0, 4>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
#ifndef VP3_NOSTAGE
                stage(c0 + u + 1, st[(u + 1) & 3]);
#endif
#ifndef VP3_NOGLOAD
                gload(c0 + u + 5, st[(u + 1) & 3]);
#endif
                VPSTAMP(3 + 2 * (c0 + u));
                __syncthreads();
                VPSTAMP(4 + 2 * (c0 + u));
            });
        static_for<0, 4>([&](auto uc) {                   // the last one to four chunks: nothing left to request
            constexpr int u = decltype(uc)::value;
            if (c0 + u < nchunk) {
                if (c0 + u + 1 < nchunk) stage(c0 + u + 1, st[(u + 1) & 3]);
                __syncthreads();
            }
        });
        __syncthreads();                                  // (the consumers' reduction barrier)
    } else {
        // ------------------------------------------------------------------ consumers: wave = 32 output columns
        const int cw = w - 4, i = lane & 31, h = lane >> 5;
        f32x16 acc[2];
        zero_acc(acc);
        const size_t plane = pack3_plane(nchunk * VP3_KC, D);     // the split pack is zero-padded to whole chunks (PackBuilder::fwd3 kpad)
        struct BF { u32x4_t t[3]; };
        const uint16_t* wp = W3 + ((size_t)(32 * cw + i)) * 16 + 8 * h;
        auto bload1 = [&](int ch, int j, BF& f) {         // weight fragments of step j of chunk ch (past the end: a harmless re-read of the last chunk)
            const uint16_t* p = wp + (size_t)(min(ch, nchunk - 1) * NKS + j) * D * 16;
#pragma unroll
            for (int t = 0; t < 3; ++t) f.t[t] = *reinterpret_cast<const u32x4_t*>(p + t * plane);
        };
        auto aread = [&](int buf, int j, BF& a) {
            const uint16_t* ap = &As[buf][0][i * VP3_LD + 8 * h + j * 16];
#pragma unroll
            for (int t = 0; t < 3; ++t) a.t[t] = *reinterpret_cast<const u32x4_t*>(ap + t * TILE_M * VP3_LD);
        };
        // a chunk: the A fragments of step j + 1 are requested before the MFMAs of step j (one wave per role and SIMD: nothing else hides the
        // LDS latency -- measured 3.4 k cycles per chunk instead of 1.5 k when every step waited for its own reads); the weight fragments run
        // VP3_NB steps ahead in a register ring (ring slot = step % VP3_NB)
        BF bq[VP3_NB];
        auto mma = [&](int buf, int ch) {
            BF a0, a1;
            aread(buf, 0, a0);
            static_for<0, NKS>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                BF& ac = (j & 1) ? a1 : a0;
                BF& an = (j & 1) ? a0 : a1;
                if (j + 1 < NKS) aread(buf, j + 1, an);
                __builtin_amdgcn_sched_barrier(0);
                BF& q = bq[j % VP3_NB];
                {
                    acc[0] = mfma_bf16(ac.t[1], q.t[1], acc[0]);          // two accumulators alternate ; small terms first
                    acc[1] = mfma_bf16(ac.t[0], q.t[2], acc[1]);
                    acc[0] = mfma_bf16(ac.t[2], q.t[0], acc[0]);
                    acc[1] = mfma_bf16(ac.t[0], q.t[1], acc[1]);
                    acc[0] = mfma_bf16(ac.t[1], q.t[0], acc[0]);
                    acc[1] = mfma_bf16(ac.t[0], q.t[0], acc[1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                bload1(ch + (j + VP3_NB) / NKS, (j + VP3_NB) % NKS, q);       // the slot's next occupant: step j + VP3_NB
            });
        };
        VPSTAMP(0);
        static_for<0, VP3_NB>([&](auto jc) { bload1(0, decltype(jc)::value, bq[decltype(jc)::value]); });
        VPSTAMP(1);
        __syncthreads();
        VPSTAMP(2);
        for (int c = 0; c < nchunk; ++c) {
            mma(c & 1, c);
            VPSTAMP(3 + 2 * c);
            __syncthreads();
            VPSTAMP(4 + 2 * c);
        }
        __syncthreads();                                  // keeps the barrier count of the two roles equal (producers: one per chunk + 2)
        const int col = 32 * cw + i;
        const float bv = bias[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gr = r0 + acc_row(r, lane);
            if (FULL || gr < R) Y[(size_t)phys(gr) * D + col] = (acc[0][r] + acc[1][r]) + bv;
        }
    }
}
void launch_vproj_fwd3(const float* X, const uint16_t* W3, const float* bias, float* Y, int R, int Dv, Drop dp, hipStream_t s, int seg, int stride, int off) {
    const dim3 grid((R + TILE_M - 1) / TILE_M), block(VP3_T);
    const bool full = R % TILE_M == 0 && Dv % VP3_KC == 0 && seg == 0;
    const int sel = (full ? 2 : 0) | (dp.thresh ? 1 : 0);
#define VP3_GO(F, DR) VSL_LAUNCH((k_vproj_fwd3<F, DR>), grid, block, 0, s, X, W3, bias, Y, R, Dv, dp, seg, stride, off)
    switch (sel) { case 3: VP3_GO(true, true); break; case 2: VP3_GO(true, false); break; case 1: VP3_GO(false, true); break; default: VP3_GO(false, false); }
#undef VP3_GO
}

// =========================================================================================================
// a8 backward, first half: dh1 = [dQ | dK | dV] [Wq; Wk; Wv] ; dx = dr + LN1^T(dh1 * m1)   (autograd of layers_t7.py:168-173)
// =========================================================================================================
constexpr int QKV3_LD = 3 * D + 8;      // bf16 elements per plane row (784 B: 16 mod 128)
__global__ __launch_bounds__(256) void k_qkv_bwd(const float* __restrict__ dQ, const float* __restrict__ dK,
                                                 const float* __restrict__ dV, const float* __restrict__ x,
                                                 const float* __restrict__ dr, const float* __restrict__ ln_g,
                                                 float* __restrict__ dx,
                                                 float* __restrict__ p_lng, float* __restrict__ p_lnb, int R, Drop d1, int dq_slabs,
                                                 size_t dq_slab, const uint16_t* __restrict__ WT3) {
    // the K = 384 product on the bf16 matrix cores at fp32 grade (gemm32pl): the staged [dQ | dK | dV] tile is split into three bf16 planes on
    // its way into LDS (75 KB)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* Ap = reinterpret_cast<uint16_t*>(smem);        // 3 planes of [32][QKV3_LD] = [dQ | dK | dV]
    float* Ts = smem + 3 * TILE_M * QKV3_LD / 2;             // [32][LDP]
    float* Xs = Ts + TILE_M * LDP;            // [32][LDP] raw LN1 input
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int r0 = blockIdx.x * TILE_M;
    {
        float4 a[4], bq[4], cv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + q * 256;
            const int r = r0 + (e >> 5), c = (e & 31) * 4;
            a[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            bq[q] = a[q]; cv[q] = a[q];
            if (r < R) {
                a[q] = *reinterpret_cast<const float4*>(dQ + (size_t)r * D + c);
                if (dq_slabs > 1) {     // L > 256: dQ arrives as one partial per key block (k_attn_bwd_long); the sum stays in slab 0 (weight gradient)
                    for (int sl = 1; sl < dq_slabs; ++sl) {
                        const float4 e = *reinterpret_cast<const float4*>(dQ + (size_t)sl * dq_slab + (size_t)r * D + c);
                        a[q].x += e.x; a[q].y += e.y; a[q].z += e.z; a[q].w += e.w;
                    }
                    *reinterpret_cast<float4*>(const_cast<float*>(dQ) + (size_t)r * D + c) = a[q];
                }
                bq[q] = *reinterpret_cast<const float4*>(dK + (size_t)r * D + c);
                cv[q] = *reinterpret_cast<const float4*>(dV + (size_t)r * D + c);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + q * 256;
            const int rr = e >> 5, c = (e & 31) * 4;
            split_store4(Ap, QKV3_LD, TILE_M * QKV3_LD, rr, c, a[q]);
            split_store4(Ap, QKV3_LD, TILE_M * QKV3_LD, rr, D + c, bq[q]);
            split_store4(Ap, QKV3_LD, TILE_M * QKV3_LD, rr, 2 * D + c, cv[q]);
        }
    }
    LnResid lres;
    ln_resid_prefetch(lres, dr, nullptr, r0, R);
    load_tile128(Xs, x, r0, TILE_M, R);                     // LN1 input rows: first used after the GEMM
    __syncthreads();
    f32x16 acc[1];
    zero_acc(acc);
    gemm32pl<1>(Ap, QKV3_LD, TILE_M * QKV3_LD, 3 * D, WT3, D, 32 * w, 0, acc);
    const int col = 32 * w + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, lane);
        Ts[row * LDP + col] = acc[0][r] * drop_mul(d1, (uint32_t)((r0 + row) * D + col));
    }
    __syncthreads();
    ln_bwd_tile(Ts, Xs, lres, ln_g, dx, p_lng, p_lnb, r0, R);
}
void launch_qkv_bwd(const float* dQ, const float* dK, const float* dV, const float* x, const float* dr,
                    const float* ln_g, const uint16_t* WT3, float* dx, float* p_lng, float* p_lnb, int R, Drop d1, hipStream_t s,
                    int dq_slabs) {
    const size_t shm = (size_t)(3 * TILE_M * QKV3_LD / 2 + 2 * TILE_M * LDP) * sizeof(float);
    static size_t lds_ok = 0;
    ensure_dynamic_lds((const void*)k_qkv_bwd, shm, lds_ok, "k_qkv_bwd");
    VSL_LAUNCH(k_qkv_bwd, dim3((R + TILE_M - 1) / TILE_M), dim3(256), shm, s, dQ, dK, dV, x, dr, ln_g, dx, p_lng,
                       p_lnb, R, d1, dq_slabs, (size_t)R * D, WT3);
}

// =========================================================================================================
// a5 Embedding.linear (layers_t7.py:81-87): Y = A W^T + b, A = [word | char] rows (R, K = word_dim + 100), and its data gradient
// dA = G W.  The query side is 40 row tiles at the headline shape -- pure latency -- and sits on the critical path of both directions
// (forward: ahead of the query encoder pass; backward: the tail of the step), so both kernels stage their WHOLE tile once, split it into
// bf16 planes on the way (one barrier) and run the product on the bf16 matrix cores.  fp32-input versions: k_linear_fwd (K streamed in four
// chunks with two barriers each, 16.6 us), kept for widths that are not multiples of 16.  Round 4: the rnn head's gate projections
// (128 -> 512, four column blocks per row tile) moved here from the fp32-input k_linear_bwd_data (15 us per launch whatever its size: 6.8 us of fp32 MFMA per wave).
// =========================================================================================================
// blockIdx.y = 128-column block of an operand with `ncols` columns (the rnn head's gate projection x W_ih^T: ncols = 512, no bias, output rows of
// leading dimension ldy); seg > 0: logical row r is physical row (r / seg) * stride + off + r % seg of A and Y -- the rows of one TIME CHUNK of a
// (B, T, .) tensor, which lets the recurrent kernels of the rnn head be pipelined chunk by chunk.
__global__ __launch_bounds__(256) void k_linear_fwd3(const float* __restrict__ A, const uint16_t* __restrict__ W3, const float* __restrict__ bias,
                                                     float* __restrict__ Y, int R, int K, int ldb, int ncols, int ldy, int seg, int stride, int off) {
    extern __shared__ __attribute__((aligned(16))) uint16_t pl[];     // three planes [32][ldb]
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int r0 = blockIdx.x * TILE_M, k4 = K >> 2, cb = blockIdx.y * D;
    auto phys = [&](int r) { return seg > 0 ? (r / seg) * stride + off + r % seg : r; };
    Ring3<1> ring;
    ring3_prefetch<1>(ring, W3, K, ncols, cb + 32 * w, 0);           // weight fragments first: in flight during the staging
    constexpr int MAXQ = 16;                                         // K <= 512: the whole tile in ONE round of loads (one memory latency)
    float4 v[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int e = tid + q * 256, rr = e / k4, c = (e - rr * k4) * 4;
        v[q] = (e < TILE_M * k4 && r0 + rr < R) ? *reinterpret_cast<const float4*>(A + (size_t)phys(r0 + rr) * K + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int e = tid + q * 256, rr = e / k4, c = (e - rr * k4) * 4;
        if (e < TILE_M * k4) split_store4(pl, ldb, TILE_M * ldb, rr, c, v[q]);
    }
    __syncthreads();
    f32x16 acc[1];
    zero_acc(acc);
    gemm32pl<1>(pl, ldb, TILE_M * ldb, K, W3, ncols, cb + 32 * w, 0, acc, ring);
    const int col = cb + 32 * w + (lane & 31);
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gr = r0 + acc_row(r, lane);
        if (gr < R) Y[(size_t)phys(gr) * ldy + col] = acc[0][r] + bv;
    }
}
static int plane_ld(int K) { int l = K + 8; while (((l / 8) & 1) == 0) l += 8; return l; }      // row stride / 16 bytes odd: conflict-free b128 reads
void launch_linear_fwd3(const float* A, const uint16_t* W3, const float* bias, float* Y, int R, int K, hipStream_t s, int ncols, int seg, int stride,
                        int off) {
    if (K > 512 || K % 16 || ncols % D) { fprintf(stderr, "[vslnet_hip] launch_linear_fwd3: K=%d ncols=%d unsupported\n", K, ncols); return; }
    const int ldb = plane_ld(K);
    const size_t shm = (size_t)3 * TILE_M * ldb * sizeof(uint16_t);
    static size_t ok = 0;
    ensure_dynamic_lds((const void*)k_linear_fwd3, shm, ok, "k_linear_fwd3");
    VSL_LAUNCH(k_linear_fwd3, dim3((R + TILE_M - 1) / TILE_M, ncols / D), dim3(256), shm, s, A, W3, bias, Y, R, K, ldb, ncols, ncols, seg, stride, off);
}
// dA (R, K) = G (R, 128) Bm, Bm = split transpose pack with ncols = Kc >= K columns (Kc a multiple of 128: whole column tiles)
__global__ __launch_bounds__(256) void k_linear_bwd_data3(const float* __restrict__ G, const uint16_t* __restrict__ WT3, float* __restrict__ dA,
                                                          int R, int K, int Kc) {
    __shared__ __attribute__((aligned(16))) uint16_t pl[3 * TILE_M * (D + 8)];
    constexpr int LD = D + 8;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int r0 = blockIdx.x * TILE_M;
    {
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + q * 256, rr = e >> 5;
            v[q] = r0 + rr < R ? *reinterpret_cast<const float4*>(G + (size_t)(r0 + rr) * D + (e & 31) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int e = tid + q * 256; split_store4(pl, LD, TILE_M * LD, e >> 5, (e & 31) * 4, v[q]); }
    }
    __syncthreads();
    for (int cb = 0; cb < Kc; cb += 512) {
        f32x16 acc[4];
        zero_acc(acc);
        gemm32pl<4>(pl, LD, TILE_M * LD, D, WT3, Kc, cb + 32 * w, D, acc);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int col = cb + 32 * w + t * D + (lane & 31);
            if (col < K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gr = r0 + acc_row(r, lane);
                    if (gr < R) dA[(size_t)gr * K + col] = acc[t][r];
                }
            }
        }
    }
}
void launch_linear_bwd_data3(const float* G, const uint16_t* WT3, float* dA, int R, int K, int Kc, hipStream_t s) {
    VSL_LAUNCH(k_linear_bwd_data3, dim3((R + TILE_M - 1) / TILE_M), dim3(256), 0, s, G, WT3, dA, R, K, Kc);
}

}  // namespace vsl
