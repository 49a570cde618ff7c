// Row-tile GEMM kernels on the BF16 matrix cores at fp32 grade (3-way operand split, 6 products: common.hpp).  gfx950 only.
// Built with -fno-slp-vectorize (vslnet_amd/build.py).
#include "common.hpp"
#include "launch.hpp"
#include <type_traits>

namespace vsl {

bool split_gemm_enabled() {
    static const bool off = getenv("VSL_F32_GEMM") && getenv("VSL_F32_GEMM")[0] == '1';
    return !off;
}

// =========================================================================================================
// a2  VisualProjection (/root/reference/model/layers_t7.py:105-115):  Y = drop(X) W^T + b,  X (R, Dv) streamed from HBM once.
// 32-row tile per workgroup, K streamed in 128-wide chunks.  A chunk's rows are dropped out and split into their three bf16 terms ONCE, on
// their way from global memory into LDS (three planes of [32][128] bf16, double buffered: 51 KB); every wave (= 32 output columns)
// reads its A fragments from the planes (ds_read_b128 = the lane's 8 consecutive k) and its B fragments from the split pack of the weight
// (global, 16 bytes per lane, fetched one chunk ahead in registers).  Per K = 16 step and wave: 3 + 3 fragment loads, 6 MFMAs on two
// alternating accumulators.  The round-2 kernel (k_vproj_fwd: fp32-input MFMA, 64 cycles per 32 x 32 x 2) took 41 us at the headline shape.
// =========================================================================================================
constexpr int VP3_KC = 128;
constexpr int VP3_LD = VP3_KC + 8;      // bf16 elements per LDS row (272 B: 16-byte aligned rows, conflict-free b128 reads)
// FULL: R % 32 == 0, Dv % 128 == 0, no row mapping -> no guards anywhere (every BASELINE shape) ; DROPON: dropout enabled (block-uniform)
template <bool FULL, bool DROPON>
__global__ __launch_bounds__(256, 1) void k_vproj_fwd3(const float* __restrict__ X, const uint16_t* __restrict__ W3, const float* __restrict__ bias,
                                                       float* __restrict__ Y, int R, int Dv, Drop dp, int seg, int stride, int off) {
    __shared__ __attribute__((aligned(16))) uint16_t As[2][3][TILE_M * VP3_LD];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int r0 = blockIdx.x * TILE_M;
    // seg > 0 (GEMM use by the rnn head, no dropout): logical row r = physical row (r / seg) * stride + off + r % seg of X and Y
    auto phys = [&](int r) { return (!FULL && seg > 0) ? (r / seg) * stride + off + r % seg : r; };
    f32x16 acc[2];
    zero_acc(acc);
    const int nchunk = (Dv + VP3_KC - 1) / VP3_KC;
    const size_t plane = pack3_plane(nchunk * VP3_KC, D);     // the split pack is zero-padded to whole chunks (PackBuilder::fwd3 kpad)
    struct Stage { float4 v[4]; };                           // a chunk's rows of this thread: row (tid >> 5) + 8 q, columns (tid & 31) * 4 ..
    const float* xrow[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) xrow[q] = X + (size_t)phys(min(r0 + (tid >> 5) + 8 * q, R - 1)) * Dv + (tid & 31) * 4;
    auto gload = [&](int ch, Stage& st) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (FULL) st.v[q] = *reinterpret_cast<const float4*>(xrow[q] + ch * VP3_KC);
            else {
                const int c = (tid & 31) * 4 + ch * VP3_KC;
                st.v[q] = (r0 + (tid >> 5) + 8 * q < R && c < Dv) ? *reinterpret_cast<const float4*>(xrow[q] + ch * VP3_KC) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    // piece pc (0..7) of a chunk's staging: half a float4 = one pair: dropout + split ; the odd piece stores the float4's three 8-byte words
    uint32_t keep_h, keep_m, keep_l;
    auto spiece = [&](int ch, int buf, int pc, const Stage& st) {
        const int q = pc >> 1, rr = (tid >> 5) + 8 * q, cl = (tid & 31) * 4;
        float x0 = (pc & 1) ? st.v[q].z : st.v[q].x, x1 = (pc & 1) ? st.v[q].w : st.v[q].y;
        if (DROPON) {
            const uint32_t base = (uint32_t)((size_t)(r0 + rr) * Dv + cl + ch * VP3_KC) + 2u * (pc & 1);
            x0 *= drop_keep_scale(dp, base); x1 *= drop_keep_scale(dp, base + 1);
        }
        uint32_t th, tm, tl;
        split3(x0, x1, th, tm, tl);
        if (pc & 1) {
            *reinterpret_cast<u32x2_t*>(&As[buf][0][rr * VP3_LD + cl]) = u32x2_t{keep_h, th};
            *reinterpret_cast<u32x2_t*>(&As[buf][1][rr * VP3_LD + cl]) = u32x2_t{keep_m, tm};
            *reinterpret_cast<u32x2_t*>(&As[buf][2][rr * VP3_LD + cl]) = u32x2_t{keep_l, tl};
        } else { keep_h = th; keep_m = tm; keep_l = tl; }
    };
    struct BQ { u32x4_t b[3][VP3_KC / 16]; };            // a chunk's weight fragments: [term][k step]
    const uint16_t* wp = W3 + ((size_t)(32 * w + i)) * 16 + 8 * h;
    auto bload = [&](int ch, BQ& q) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int kk = 0; kk < VP3_KC / 16; ++kk)
                q.b[t][kk] = *reinterpret_cast<const u32x4_t*>(wp + t * plane + (size_t)(ch * (VP3_KC / 16) + kk) * D * 16);
    };
    struct AF { u32x4_t t[3]; };
    auto aread = [&](int buf, int kk, AF& a) {
        const uint16_t* ap = &As[buf][0][i * VP3_LD + 8 * h + kk * 16];
#pragma unroll
        for (int t = 0; t < 3; ++t) a.t[t] = *reinterpret_cast<const u32x4_t*>(ap + t * TILE_M * VP3_LD);
    };
    auto mma6 = [&](const AF& a, const BQ& q, int kk) {   // two accumulators alternate ; small terms first
        acc[0] = mfma_bf16(a.t[1], q.b[1][kk], acc[0]);
        acc[1] = mfma_bf16(a.t[0], q.b[2][kk], acc[1]);
        acc[0] = mfma_bf16(a.t[2], q.b[0][kk], acc[0]);
        acc[1] = mfma_bf16(a.t[0], q.b[1][kk], acc[1]);
        acc[0] = mfma_bf16(a.t[1], q.b[0][kk], acc[0]);
        acc[1] = mfma_bf16(a.t[0], q.b[0][kk], acc[1]);
    };
    // One chunk: 8 k steps of 6 MFMAs from buffer bc / fragments qc.  With STAGE, piece kk of the NEXT chunk's staging (rows in `sn`) is woven
    // between the MFMAs of step kk (the compiler would issue the 48 MFMAs back to back and the vector work after them: an in-order wave cannot
    // overlap them unless they alternate in program order), the weight fragments of the next chunk and the rows of the one after are requested first.
    auto chunk = [&](auto stage_c, int ch, int bc, const BQ& qc, BQ& qn, const Stage& sn, Stage& sf) {
        constexpr bool STAGE = decltype(stage_c)::value;
        if (STAGE) { bload(ch + 1, qn); gload(min(ch + 2, nchunk - 1), sf); }      // (past the end: a harmless re-read)
        AF a0, a1;
        aread(bc, 0, a0);
        static_for<0, VP3_KC / 16>([&](auto kc) {
            constexpr int kk = decltype(kc)::value;
            AF& ac = (kk & 1) ? a1 : a0;
            AF& an = (kk & 1) ? a0 : a1;
            if (kk + 1 < VP3_KC / 16) aread(bc, kk + 1, an);
            mma6(ac, qc, kk);
            if (STAGE) spiece(ch + 1, bc ^ 1, kk, sn);
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  // one MFMA
                if (m < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // one fragment read of the next step
                if (STAGE) __builtin_amdgcn_sched_group_barrier(0x002, DROPON ? 6 : 2, 0);   // staging arithmetic
                if (STAGE && m == 5) __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);      // its LDS stores
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (STAGE) __syncthreads();
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    BQ q0, q1;
    Stage s0, s1;
    gload(0, s0);
    bload(0, q0);
    static_for<0, 8>([&](auto pc) { spiece(0, 0, decltype(pc)::value, s0); });
    gload(min(1, nchunk - 1), s1);
    __syncthreads();
    int ch = 0;
    for (; ch + 2 < nchunk; ch += 2) { chunk(T_(), ch, 0, q0, q1, s1, s0); chunk(T_(), ch + 1, 1, q1, q0, s0, s1); }
    if (nchunk - ch == 2) { chunk(T_(), ch, 0, q0, q1, s1, s0); chunk(F_(), ch + 1, 1, q1, q0, s0, s1); }
    else chunk(F_(), ch, 0, q0, q1, s1, s0);
    const int col = 32 * w + i;
    const float bv = bias[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gr = r0 + acc_row(r, lane);
        if (FULL || gr < R) Y[(size_t)phys(gr) * D + col] = (acc[0][r] + acc[1][r]) + bv;
    }
}
void launch_vproj_fwd3(const float* X, const uint16_t* W3, const float* bias, float* Y, int R, int Dv, Drop dp, hipStream_t s, int seg, int stride, int off) {
    const dim3 grid((R + TILE_M - 1) / TILE_M), block(256);
    const bool full = R % TILE_M == 0 && Dv % VP3_KC == 0 && seg == 0;
    if (full && dp.thresh) VSL_LAUNCH((k_vproj_fwd3<true, true>), grid, block, 0, s, X, W3, bias, Y, R, Dv, dp, seg, stride, off);
    else if (full) VSL_LAUNCH((k_vproj_fwd3<true, false>), grid, block, 0, s, X, W3, bias, Y, R, Dv, dp, seg, stride, off);
    else if (dp.thresh) VSL_LAUNCH((k_vproj_fwd3<false, true>), grid, block, 0, s, X, W3, bias, Y, R, Dv, dp, seg, stride, off);
    else VSL_LAUNCH((k_vproj_fwd3<false, false>), grid, block, 0, s, X, W3, bias, Y, R, Dv, dp, seg, stride, off);
}

}  // namespace vsl
