// Fused conv-block kernels of one FeatureEncoder application (gfx950).
//
// a6/a7 DepthwiseSeparableConvBlock (/root/reference/model/layers_t7.py:118-140) is four layers of
//     x <- x + drop(relu(pointwise(depthwise7(LN(x)))))
// Each layer reaches 3 rows up and down the sequence, so a row tile cannot run the block alone -- unless it recomputes
// its neighbours' rows: a workgroup owns 32 rows and carries a 12-row halo on each side that shrinks by 3 rows per
// layer (56 -> 50 -> 44 -> 38 -> 32 rows).  The dropout masks are counter-based hashes of the element index, so the
// recomputed rows are bit-identical to the owner's.  One launch replaces four (forward) / four + the fused GEMM stage
// (backward): 3 kernel boundaries, 3 tile loads from memory and 3 store drains less per encoder application, at the
// price of 1.5x (forward) / 1.75x (backward) of the block's matrix work -- the row-tile kernels ran their matrix
// pipes at 5 - 7 % (profiles/r01_j_pmc_mfma_busy.txt), so the price is paid out of idle cycles.
//
// Matrix work: v_mfma_f32_16x16x4_f32 (16-row granularity fits the shrinking row ranges: 4, 3, 3, 2 half-blocks).
// 8 waves; wave w owns output columns [16 w, 16 w + 16) of every row block, so a weight slice (128 x 16 = 8 KB = 32
// registers per lane) is fetched by exactly one wave, straight from the packed layout of common.hpp, one layer ahead.
#include "common.hpp"
#include "launch.hpp"
#include "tile_bodies.hpp"
#include <type_traits>

namespace vsl {

__device__ long long g_stamps_e[32];
__device__ int g_dbg_on_e = 0;
#ifdef VSL_STAMPS       // see kernels_bwd.hip: stamps are a separate build
#define ESTAMP(k) do { if (g_dbg_on_e && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_stamps_e[k] = clock64(); } while (0)
#else
#define ESTAMP(k) do { } while (0)
#endif
static int edbg_on() {
#ifndef VSL_STAMPS
    return 0;
#endif
    static int inited = 0, on = 0;
    if (!inited) { inited = 1; on = getenv("VSL_DEBUG_TIMING") != nullptr; if (on) { int one = 1; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_on_e), &one, sizeof one); } }
    return on;
}
static void edbg_report(const char* name, int nst, hipStream_t s, int& left) {
    if (left <= 0) return;
    long long h[32];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps_e), sizeof h);
    fprintf(stderr, "[%s cycles]", name);
    for (int i = 1; i < nst; ++i) fprintf(stderr, " %lld", h[i] - h[i - 1]);
    fprintf(stderr, " | total %lld\n", h[nst - 1] - h[0]);
    --left;
}

static void edbg_report2(const char* name, int i0, int i1, hipStream_t s, int& left) {
    if (left <= 0) return;
    long long h[32];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps_e), sizeof h);
    fprintf(stderr, "[%s cycles] (from stamp 1: %lld)", name, h[i0] - h[1]);
    for (int i = i0 + 1; i < i1; ++i) fprintf(stderr, " %lld", h[i] - h[i - 1]);
    fprintf(stderr, "\n");
}

constexpr int CB_T = 512;                       // threads per workgroup (8 waves, 2 per SIMD)
constexpr int CB_HALO = 4 * HALO;               // 12 rows each side

// ---------------------------------------------------------------------------------------------------------
// 16x16x4 fp32 MFMA GEMM on LDS row blocks.  Operand lane maps (lane = 16 g + i):
//   A: A[row i][k = g]      B: B[k = g][col i]      C/D reg r: D[row 4 g + r][col i]
// One float4 of A (row i, k = 16 kq + 4 g + 0..3) and one float4 of the packed weight (same k's, col i) feed 4 MFMAs:
// MFMA m of step kq contracts k = 16 kq + 4 g + m over g = 0..3.
// ---------------------------------------------------------------------------------------------------------
struct BF16 { float4 b[8]; };                   // a wave's 128 x 16 weight slice
__device__ __forceinline__ void bf16_load(BF16& f, const float* __restrict__ Bp, int ncols, int col0) {
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const float4* bp = reinterpret_cast<const float4*>(Bp) + ((size_t)(g >> 1) * ncols + col0 + j) * 2 + (g & 1);
#pragma unroll
    for (int kq = 0; kq < 8; ++kq) f.b[kq] = bp[(size_t)(2 * kq) * ncols * 2];
}
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// acc[nb][rb] (16 x 16) += As[16 rb + 0..15][0..127] * B_nb ; NB weight slices share every A fragment
template <int NRB, int NB>
__device__ __forceinline__ void gemm16(const float* __restrict__ As, int lda, const BF16 (&bf)[NB], f32x4 (&acc)[NB][NRB]) {
    const int lane = threadIdx.x & 63;
    const float* arow = As + (lane & 15) * lda + 4 * (lane >> 4);
#pragma unroll
    for (int kq = 0; kq < 8; ++kq) {
        float4 a[NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) a[rb] = *reinterpret_cast<const float4*>(arow + rb * 16 * lda + kq * 16);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc[nb][rb] = mfma16(a[rb].x, bf[nb].b[kq].x, acc[nb][rb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc[nb][rb] = mfma16(a[rb].y, bf[nb].b[kq].y, acc[nb][rb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc[nb][rb] = mfma16(a[rb].z, bf[nb].b[kq].z, acc[nb][rb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc[nb][rb] = mfma16(a[rb].w, bf[nb].b[kq].w, acc[nb][rb]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// The same GEMM on the BF16 matrix cores at fp32 grade (common.hpp: 3-way split, 6 products), v_mfma_f32_16x16x32_bf16.
//   A: three bf16 planes (terms h, m, l) of the activation tile in LDS, row stride CB_LDB elements; the kernel phase that produces the
//      tile splits every value ONCE while storing it.  Lane (i = lane & 15, g = lane >> 4) reads row i, k = 32 ks + 8 g .. + 7 as one
//      ds_read_b128 (288-byte rows: conflict-free for this lane map).
//   B: the wave's 128 x 16 slice of the weight's split pack (PackJob type 6 / 7), 3 x 4 x 16 bytes per lane = 48 registers.
//   C/D as for 16x16x4: reg r <-> row 4 g + r, column i.
// Per 16-row block and K = 32 step: 3 A reads, 6 MFMAs (16 passes of the matrix pipe against 32 x 4 for the fp32-input MFMA), and the bf16
// MFMAs overlap with the vector work of the SIMD's other wave.
// ---------------------------------------------------------------------------------------------------------
constexpr int CB_LDB = 144;
struct B3 { u32x4_t b[3][4]; };                 // [term][K = 32 step]
__device__ __forceinline__ void b3_load(B3& f, const uint16_t* __restrict__ W3, int K, int ncols, int col0) {
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const size_t plane = pack3_plane(K, ncols);
    const uint16_t* p = W3 + ((size_t)(g >> 1) * ncols + col0 + j) * 16 + 8 * (g & 1);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) f.b[t][ks] = *reinterpret_cast<const u32x4_t*>(p + t * plane + (size_t)(2 * ks) * ncols * 16);
}
// one 16-byte load of the slice: lets a caller spread a prefetch over a phase instead of issuing twelve loads at once
__device__ __forceinline__ void b3_load_one(B3& f, const uint16_t* __restrict__ W3, int K, int ncols, int col0, int t, int ks) {
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const uint16_t* p = W3 + ((size_t)(g >> 1) * ncols + col0 + j) * 16 + 8 * (g & 1);
    f.b[t][ks] = *reinterpret_cast<const u32x4_t*>(p + t * pack3_plane(K, ncols) + (size_t)(2 * ks) * ncols * 16);
}
// acc[rb] (16 x 16) += A[16 rb + 0..15][0..127] * B ; Ah = plane h at the block range's first row, `ps` = elements between planes
template <int NRB>
__device__ __forceinline__ void gemm16s(const uint16_t* __restrict__ Ah, int ps, const B3& bf, f32x4 (&acc)[1][NRB]) {
    const int lane = threadIdx.x & 63;
    const uint16_t* ar = Ah + (lane & 15) * CB_LDB + 8 * (lane >> 4);
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};       // (a term, b term): mm, hl, lh, hm, mh, hh -- small terms first
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        u32x4_t a[NRB][3];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int t = 0; t < 3; ++t) a[rb][t] = *reinterpret_cast<const u32x4_t*>(ar + t * ps + rb * 16 * CB_LDB + ks * 32);
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc[0][rb] = mfma16_bf16(a[rb][TA[p]], bf.b[TB[p]][ks], acc[0][rb]);      // consecutive MFMAs: different accumulators
    }
}
// Stores two adjacent channels' values of one row as a split pair.  Thread = channel c, values of rows i0 / i1 = i0 + 1 of its segment:
// even lanes take row i0 of channels (c, c + 1), odd lanes row i1 of channels (c - 1, c) -- one quad-permute DPP exchange per row pair.
__device__ __forceinline__ void split_store_pair(uint16_t* __restrict__ P0, int ps, int row0, int c, float v0, float v1, bool ok0, bool ok1) {
    const bool odd = c & 1;
    const float recv = lane_xor1(odd ? v0 : v1);
    const float x0 = odd ? recv : v0, x1 = odd ? v1 : recv;
    uint32_t th, tm, tl;
    split3(x0, x1, th, tm, tl);
    if (odd ? ok1 : ok0) {
        uint32_t* d = reinterpret_cast<uint32_t*>(P0 + (row0 + (odd ? 1 : 0)) * CB_LDB + (c & ~1));
        d[0] = th; d[ps / 2] = tm; d[ps] = tl;
    }
}

// LayerNorm of up to 64 rows by 512 threads: 8 lanes per row (lane sub owns float4 columns 4 sub + 32 j).  gamma / beta in
// LDS.  dst[r] = LN(src[r]) * dropout ; `drow0` = global row of row 0 (dropout element index = row * 128 + col).
// Rows outside [keep_lo, keep_hi) are written as zeros: rows of a neighbouring sample act as the conv's zero padding.
__device__ __forceinline__ void ln_rows512(const float* __restrict__ src, float* __restrict__ dst, int nrows,
                                           const float* __restrict__ g, const float* __restrict__ b, const Drop& dp, int drow0,
                                           int keep_lo = 0, int keep_hi = 1 << 30) {
    const int sub = threadIdx.x & 7, r = threadIdx.x >> 3;
    if (r >= nrows) return;
    const float* s = src + r * LDP + sub * 4;
    float* d = dst + r * LDP + sub * 4;
    if (r < keep_lo || r >= keep_hi) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(d + 32 * j) = z;
        return;
    }
    float4 v[4];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = *reinterpret_cast<const float4*>(s + 32 * j); sum += sum4(v[j]); }
    const float mu = grp8_sum(sum) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j].x -= mu; v[j].y -= mu; v[j].z -= mu; v[j].w -= mu;
        q += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
    }
    const float rstd = rsqrtf(grp8_sum(q) * (1.0f / D) + LN_EPS);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 gv = *reinterpret_cast<const float4*>(g + sub * 4 + 32 * j);
        const float4 bv = *reinterpret_cast<const float4*>(b + sub * 4 + 32 * j);
        float4 o;
        o.x = v[j].x * rstd * gv.x + bv.x; o.y = v[j].y * rstd * gv.y + bv.y;
        o.z = v[j].z * rstd * gv.z + bv.z; o.w = v[j].w * rstd * gv.w + bv.w;
        if (dp.thresh) {
            const uint32_t base = (uint32_t)((drow0 + r) * D + sub * 4 + 32 * j);
            o.x *= drop_keep_scale(dp, base); o.y *= drop_keep_scale(dp, base + 1);
            o.z *= drop_keep_scale(dp, base + 2); o.w *= drop_keep_scale(dp, base + 3);
        }
        *reinterpret_cast<float4*>(d + 32 * j) = o;
    }
}

// The same LayerNorm (all rows kept) whose output goes to the three bf16 planes of a split-GEMM A operand (row stride CB_LDB, `ps`
// elements between planes) and, optionally, straight to memory (`gout`: row 0 of the tile, rows < gn are stored).
__device__ __forceinline__ void ln_rows512_split(const float* __restrict__ src, uint16_t* __restrict__ P0, int ps, int nrows,
                                                 const float* __restrict__ g, const float* __restrict__ b, const Drop& dp, int drow0,
                                                 float* __restrict__ gout, int gn) {
    const int sub = threadIdx.x & 7, r = threadIdx.x >> 3;
    if (r >= nrows) return;
    const float* s = src + r * LDP + sub * 4;
    float4 v[4];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = *reinterpret_cast<const float4*>(s + 32 * j); sum += sum4(v[j]); }
    const float mu = grp8_sum(sum) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j].x -= mu; v[j].y -= mu; v[j].z -= mu; v[j].w -= mu;
        q += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
    }
    const float rstd = rsqrtf(grp8_sum(q) * (1.0f / D) + LN_EPS);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 gv = *reinterpret_cast<const float4*>(g + sub * 4 + 32 * j);
        const float4 bv = *reinterpret_cast<const float4*>(b + sub * 4 + 32 * j);
        float4 o;
        o.x = v[j].x * rstd * gv.x + bv.x; o.y = v[j].y * rstd * gv.y + bv.y;
        o.z = v[j].z * rstd * gv.z + bv.z; o.w = v[j].w * rstd * gv.w + bv.w;
        if (dp.thresh) {
            const uint32_t base = (uint32_t)((drow0 + r) * D + sub * 4 + 32 * j);
            o.x *= drop_keep_scale(dp, base); o.y *= drop_keep_scale(dp, base + 1);
            o.z *= drop_keep_scale(dp, base + 2); o.w *= drop_keep_scale(dp, base + 3);
        }
        if (gout && r < gn) *reinterpret_cast<float4*>(gout + (size_t)r * D + sub * 4 + 32 * j) = o;
        uint32_t h0, m0, l0, h1, m1, l1;
        split3(o.x, o.y, h0, m0, l0);
        split3(o.z, o.w, h1, m1, l1);
        uint16_t* d = P0 + r * CB_LDB + sub * 4 + 32 * j;
        *reinterpret_cast<u32x2_t*>(d) = u32x2_t{h0, h1};
        *reinterpret_cast<u32x2_t*>(d + ps) = u32x2_t{m0, m1};
        *reinterpret_cast<u32x2_t*>(d + 2 * ps) = u32x2_t{l0, l1};
    }
}

// =========================================================================================================
// forward: x0 = xin + pos ; 4 x [ v = LN(x) ; u = depthwise7(v) ; z = u Wp^T + b ; x += drop(relu(z)) ] ; then, row-local on
// the 32 owner rows, a8's first half (:168-173): h1 = drop(LN1(x)) ; [q | k | v] = h1 W^T + b.
// Saves (owner rows only) x0, every layer's output y, depthwise output u (A operand of the weight gradient) and ReLU
// bit-mask, h1, q, k, v.
// LDS: residual stream [56] + one [68]-row buffer that holds LN(x), then (in place, after the depthwise windows are in
// registers) the depthwise output = 72 KB with the small parameters, so two workgroups -- e.g. the video and the query
// pass, which run on different streams -- share a CU.
// =========================================================================================================
constexpr int CB_PS = 384;                      // per-layer small parameters in LDS: ln_g | ln_b | pw_b
// SH = rows a layer's valid range shrinks by on each side: 3 = row tiles with a recomputed halo (12 + 32 + 12 rows);
// 0 = SAMPLE tiles for sequences of at most 32 rows (the query pass: Lq = 20): one workgroup per sample, its rows at the top
// of a 32-row window, no halo and no recomputation -- rows >= L and the taps that leave the window are the conv's zero padding.
// FULL: R and L are multiples of the 32-row tile (every BASELINE shape): every tile is whole and lies inside one sample, so the boundary
// flags below are compile-time constants and the per-row store / tap predicates disappear (12 % of the kernel's instructions were
// v_cmp / v_cndmask / exec-mask branches).
// The five GEMMs run on the bf16 matrix cores at fp32 grade (gemm16s): the depthwise output / LN1 output is split into three bf16 planes
// while it is stored (LDS +18 KB for the row tiles), the weight slices come from the split packs.
template <int SH, bool FULL>
__global__ __launch_bounds__(CB_T, 2) void k_convblock_fwd(CbFwdArgs a) {
    // sample tiles: 3 zero rows above and below the window in the LN / depthwise buffer stand for the taps that leave it
    constexpr int HL = 4 * SH, NW = TILE_M + 2 * HL, VOFF = SH ? 0 : HALO, VUR = SH ? NW + 12 : NW + 2 * HALO;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                       // [56][LDP] residual stream
    float* VU = Xs + NW * LDP + VOFF * LDP; // (pointer to window row 0; sample tiles: rows -3 .. -1 and 32 .. 34 are the zero pad)           // [68][LDP] LN(x), then depthwise output = GEMM A operand (rows indexed by window row)
    // depthwise output = GEMM A operand: three bf16 planes [NW][CB_LDB] (window rows) in their own buffer, so no barrier between the window reads
    // and these writes
    uint16_t* Ub = reinterpret_cast<uint16_t*>(Xs + (NW + VUR) * LDP);
    constexpr int UPS = NW * CB_LDB;                     // elements between planes
    float* Ps = Xs + (NW + VUR) * LDP + 3 * UPS / 2;   // [4][CB_PS] per-layer small parameters | ln1_g | ln1_b | bq | bk | bv
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int R = a.R, L = a.L;
    const int r0 = SH ? blockIdx.x * TILE_M : blockIdx.x * L, rw0 = r0 - HL;      // global row of window row 0
    ESTAMP(0);
    // ---- window rows rw0 .. rw0 + 55 (+ positional rows, :202): every load first
    float4 xv[4], pv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + q * CB_T;
        const int r = rw0 + (e >> 5), c = (e & 31) * 4;
        xv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        pv[q] = xv[q];
        if (e < NW * 32 && (SH ? (r >= 0 && r < R) : (e >> 5) < L)) {
            xv[q] = *reinterpret_cast<const float4*>(a.xin + (size_t)r * D + c);
            pv[q] = *reinterpret_cast<const float4*>(a.pos + (size_t)(SH ? r % L : e >> 5) * D + c);
        }
    }
    // small parameters of all four layers and of the LN1 / QKV stage -> LDS ; depthwise taps of the thread's channel -> registers
    {
        float pl[4], pq[2];
#pragma unroll
        for (int l = 0; l < 4; ++l) pl[l] = tid < 128 ? a.ln_g[l][tid] : tid < 256 ? a.ln_b[l][tid - 128] : tid < 384 ? a.pw_b[l][tid - 256] : 0.f;
        pq[0] = tid < 128 ? a.qf.ln_g[tid] : tid < 256 ? a.qf.ln_b[tid - 128] : 0.f;
        pq[1] = tid < 128 ? a.qf.bq[tid] : tid < 256 ? a.qf.bk[tid - 128] : tid < 384 ? a.qf.bv[tid - 256] : 0.f;
#pragma unroll
        for (int l = 0; l < 4; ++l) if (tid < 384) Ps[l * CB_PS + tid] = pl[l];
        if (tid < 256) Ps[4 * CB_PS + tid] = pq[0];
        if (tid < 384) Ps[4 * CB_PS + 256 + tid] = pq[1];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + q * CB_T;
        const int wr = e >> 5, c = (e & 31) * 4;
        if (e < NW * 32) {
            const float4 v = make_float4(xv[q].x + pv[q].x, xv[q].y + pv[q].y, xv[q].z + pv[q].z, xv[q].w + pv[q].w);
            const int r = rw0 + wr;
            if (wr >= HL && wr < HL + TILE_M && (SH ? r < R : wr < L)) *reinterpret_cast<float4*>(a.x0_out + (size_t)r * D + c) = v;
            *reinterpret_cast<float4*>(&Xs[wr * LDP + c]) = v;
        }
    }
    if (!SH && tid < 2 * HALO * 32) {            // zero pad rows of the sample-tile buffer (never written again)
        const int pr = tid >> 5, c = (tid & 31) * 4;
        *reinterpret_cast<float4*>(&VU[(pr < HALO ? pr - HALO : NW + pr - HALO) * LDP + c]) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // first weight slice and depthwise taps: requested once the window registers are free (first used after LayerNorm 0)
    B3 b3A, b3B;
    b3_load(b3A, a.W3[0], D, D, 16 * w);
    float wkc[DWK];                                              // depthwise taps of this thread's channel, fetched a layer ahead
#pragma unroll
    for (int k = 0; k < DWK; ++k) wkc[k] = a.dw_w[0][(tid & 127) * DWK + k];
    // a window that lies inside one sample needs no boundary tests in the depthwise conv (block-uniform)
    const bool interior = SH && rw0 >= 0 && rw0 + NW <= R && (rw0 % L) + NW <= L;
    const bool full = FULL || (SH && r0 + TILE_M <= R);
    // Owner rows inside ONE sample (every tile when L % 32 == 0, and always in sample mode): the window rows of other samples
    // only ever act as that sample's zero padding (their own outputs feed no owner row), so their LayerNorm output is written
    // as zeros and the depthwise conv runs without per-tap tests.  [klo, khi) = window rows of the owner sample.
    const int s_own = r0 / L;
    const bool one_owner = FULL || !SH || (full && (r0 + TILE_M - 1) / L == s_own);
    const int klo = max(0, s_own * L - rw0), khi = min(NW, (s_own + 1) * L - rw0);
    const bool plain = FULL || interior || one_owner;
    auto row_ok = [&](int wr) { return FULL || (SH ? (full || rw0 + wr < R) : wr < L); };     // may window row wr (an owner row) be stored?
    __syncthreads();
    ESTAMP(1);
    const Drop nodrop{0u, 0u, 1.f};
    const int col = 16 * w + (lane & 15), g4 = 4 * (lane >> 4);

    auto layer = [&](auto LC, auto& cur, auto&& prefetch) {
        constexpr int l = decltype(LC)::value;
        constexpr int in0 = SH * l, nin = NW - 2 * SH * l;       // LayerNorm rows
        constexpr int o0 = in0 + SH, n = nin - 2 * SH;           // rows this layer produces
        constexpr int NRB = (n + 15) / 16, QS = (n + 3) / 4;     // 16-row blocks ; rows per depthwise segment
        const float* P = Ps + l * CB_PS;
        const Drop dp = a.dp[l];
        if (one_owner) ln_rows512(Xs + in0 * LDP, VU + in0 * LDP, nin, P, P + 128, nodrop, 0, klo - in0, khi - in0);
        else ln_rows512(Xs + in0 * LDP, VU + in0 * LDP, nin, P, P + 128, nodrop, 0);
        __syncthreads();
        if (l == 0) ESTAMP(8);
        // ---- depthwise conv k = 7 along the sequence: thread = (channel, quarter of the row range), window in registers
        {
            const int c = tid & 127, seg = tid >> 7;
            const int os = o0 + seg * QS;                        // first produced window row of the segment
            float win[QS + 2 * HALO], uo[QS];
#pragma unroll
            for (int i = 0; i < QS + 2 * HALO; ++i) win[i] = VU[(SH ? min(os - HALO + i, NW - 1) : os - HALO + i) * LDP + c];
            if (plain) {
#pragma unroll
                for (int i = 0; i < QS; ++i) {
                    float u = 0.f;
#pragma unroll
                    for (int k = 0; k < DWK; ++k) u += wkc[k] * win[i + k];
                    uo[i] = u;
                }
            } else {
                int t = (rw0 + os) % L;                          // position of the produced row inside its sample
                t = t < 0 ? t + L : t;
#pragma unroll
                for (int i = 0; i < QS; ++i) {
                    float u = 0.f;
#pragma unroll
                    for (int k = 0; k < DWK; ++k) u += ((unsigned)(t + k - HALO) < (unsigned)L) ? wkc[k] * win[i + k] : 0.f;
                    uo[i] = u;
                    t = t + 1 == L ? 0 : t + 1;
                }
            }
            float* ug = a.u[l] + (ptrdiff_t)(rw0 + os) * D + c;
#pragma unroll
            for (int i = 0; i < QS; ++i) {
                if (os + i < o0 + n) {                           // wave-uniform
                    const int wr = os + i;
                    if (wr >= HL && wr < HL + TILE_M && row_ok(wr)) ug[(ptrdiff_t)i * D] = uo[i];   // saved: A operand of the weight gradient
                }
            }
            {                                                    // GEMM operand: three bf16 planes, two channels of a row per dword
#pragma unroll
                for (int i = 0; i < QS; i += 2)
                    split_store_pair(Ub, UPS, os + i, c, uo[i], i + 1 < QS ? uo[i + 1] : 0.f, os + i < o0 + n, i + 1 < QS && os + i + 1 < o0 + n);
            }
        }
        __syncthreads();
        if (l == 0) ESTAMP(9);
        // ---- pointwise GEMM + bias + ReLU (+ dropout) + residual, in place on the residual stream
        f32x4 acc[1][NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) acc[0][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm16s<NRB>(Ub + o0 * CB_LDB, UPS, cur, acc);
        __builtin_amdgcn_sched_barrier(0);
        prefetch();                                              // weight slice of the next stage: most of a layer ahead of its use
        if (l < 3) {
#pragma unroll
            for (int k = 0; k < DWK; ++k) wkc[k] = a.dw_w[l < 3 ? l + 1 : 3][(tid & 127) * DWK + k];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (l == 0) ESTAMP(10);
        const float bv = P[256 + col];
        uint16_t* mk = reinterpret_cast<uint16_t*>(a.relu_mask[l]);
        const int mybit = 1 << (lane & 15);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const int t0 = o0 + 16 * rb;                         // folds after unrolling
            const bool owner_tile = t0 < HL + TILE_M && t0 + 16 > HL;
            int pos01 = 0, pos23 = 0;                            // this lane's decision bits of rows rr = 0, 1 | 2, 3 (16 bits each)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = 16 * rb + g4 + rr;               // relative to o0
                const int o = o0 + row;
                const float z = acc[0][rb][rr] + bv;
                float av = fmaxf(z, 0.f);
                if (dp.thresh) av *= drop_keep_scale(dp, (uint32_t)((rw0 + o) * D + col));
                if (rb < NRB - 1 || row < n) Xs[o * LDP + col] += av;
                const int bit = z > 0.f ? (mybit << (16 * (rr & 1))) : 0;
                if (rr < 2) pos01 |= bit; else pos23 |= bit;
            }
            // ReLU decisions of the owner rows (uint16 view of the (R, 4) words): OR over the 16 lanes of a row with four DPP steps
            // (quad swaps, half-row mirror, row mirror), two rows per register.  Not __ballot: four ballots kept in SGPR pairs
            // across the rr loop came out wrong for the waves 4..7 of a workgroup, timing dependent
            // (tests/test_hip_parity.py::test_saved_relu_decisions_match_the_saved_activations).
            if (owner_tile) {
                auto row_or = [](int v) {
                    v |= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
                    v |= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);       // quad_perm [2,3,0,1]
                    v |= __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);      // row_half_mirror
                    v |= __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);      // row_mirror
                    return v;
                };
                pos01 = row_or(pos01);
                pos23 = row_or(pos23);
                // lane j = 0..3 of the 16-lane row stores row rr = j
                const int j = lane & 15, o = t0 + g4 + (j & 3);
                const uint32_t wv = (uint32_t)((j & 2) ? pos23 : pos01) >> (16 * (j & 1));
                if (j < 4 && o >= HL && o < HL + TILE_M && row_ok(o)) mk[(size_t)(rw0 + o) * 8 + w] = (uint16_t)wv;
            }
        }
        if (l == 0) ESTAMP(11);
        __syncthreads();
        if (l == 0) ESTAMP(12);
        // owner rows of the layer output -> memory (16-byte stores)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + q * CB_T;
            const int rr = e >> 5, c = (e & 31) * 4;
            if (row_ok(HL + rr))
                *reinterpret_cast<float4*>(a.y[l] + (size_t)(r0 + rr) * D + c) = *reinterpret_cast<const float4*>(&Xs[(HL + rr) * LDP + c]);
        }
    };
    layer(std::integral_constant<int, 0>(), b3A, [&] { b3_load(b3B, a.W3[1], D, D, 16 * w); });
    ESTAMP(2);
    layer(std::integral_constant<int, 1>(), b3B, [&] { b3_load(b3A, a.W3[2], D, D, 16 * w); });
    ESTAMP(3);
    layer(std::integral_constant<int, 2>(), b3A, [&] { b3_load(b3B, a.W3[3], D, D, 16 * w); });
    ESTAMP(4);
    layer(std::integral_constant<int, 3>(), b3B, [&] { b3_load(b3A, a.Wqkv3, D, 3 * D, 16 * w); });
    ESTAMP(5);
    // ---- a8, first half (:168-173) on the owner rows: h1 = drop(LN1(y3)) ; [q | k | v] = h1 W^T + b  (wave w = head w)
    {
        const float* Pq = Ps + 4 * CB_PS;
        // LN1 output: three bf16 planes in the (now free) GEMM operand buffer, rows 0..31 ; h1 goes to memory straight from the registers
        const int gn = FULL ? TILE_M : (SH ? min(TILE_M, R - r0) : L);
        ln_rows512_split(Xs + HL * LDP, Ub, UPS, TILE_M, Pq, Pq + 128, a.qf.d1, r0, a.qf.h1 ? a.qf.h1 + (size_t)r0 * D : nullptr, gn);
        b3_load(b3B, a.Wqkv3, D, 3 * D, D + 16 * w);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        auto proj = [&](const B3& cur, float* __restrict__ outp, int t) {
            f32x4 acc[1][2];
            acc[0][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[0][1] = acc[0][0];
            gemm16s<2>(Ub, UPS, cur, acc);
            const float bv = Pq[256 + t * D + col];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int gr = r0 + 16 * rb + g4 + rr;
                    if (row_ok(HL + 16 * rb + g4 + rr)) outp[(size_t)gr * D + col] = acc[0][rb][rr] + bv;
                }
        };
        proj(b3A, a.qf.q, 0);
        b3_load(b3A, a.Wqkv3, D, 3 * D, 2 * D + 16 * w);
        __builtin_amdgcn_sched_barrier(0);
        proj(b3B, a.qf.k, 1);
        proj(b3A, a.qf.v, 2);
    }
    ESTAMP(6);
}
constexpr size_t cb_fwd_lds_split(int sh) {
    return (size_t)(((TILE_M + 8 * sh) + (sh ? TILE_M + 8 * sh + 12 : TILE_M + 2 * HALO)) * LDP + 3 * (TILE_M + 8 * sh) * CB_LDB / 2 + 4 * CB_PS + 640) * sizeof(float);
}

// =========================================================================================================
// k_convblock_fwd2: the forward conv block for WHOLE tiles (R and L multiples of 32, L > 32 -- every BASELINE shape), rebuilt around the
// vector-instruction budget (round 5: a layer of the kernel above is ~57 % vector-ALU issue, ~20 % matrix pipe, the rest stalls --
// tools/dbg/isa_count.py, profiles/r05_notes.md).  Same window, same saved tensors, same arithmetic per row; what changed:
//   * LayerNorm in packed fp32 (v_pk_add / v_pk_mul / v_pk_fma_f32: two columns per instruction); the LayerNorm thread of an owner row also
//     stores the row (= the previous layer's output y) to memory, so the separate store pass behind every layer's barrier is gone;
//   * depthwise conv per (channel PAIR, row segment = wave): 8-byte LDS reads, 7 packed FMAs per output pair, the pair goes straight into
//     split3 (no cross-lane exchange), u leaves as 8-byte stores (a wave writes whole 512-byte rows);
//   * pointwise GEMM one 16-row block at a time, two accumulators, and the epilogue of block rb - 1 (bias, ReLU, dropout hash, residual
//     update, ReLU bits) issued BETWEEN the MFMAs of block rb: the bf16 matrix pipe and the vector ALU overlap, also within a wave;
//   * the epilogue is branch-free: the residual stream has spare rows for the last block's overhang, the ReLU bit words go to an LDS
//     table and leave as one 16-byte store per owner row.
// =========================================================================================================
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk2(float s) { return f32x2{s, s}; }
constexpr int C2_XR = 68;                       // residual-stream rows: 56 window rows + the overhang of the last 16-row block (row 66 in layer 0)
constexpr int C2_VUR = 68;                      // LayerNorm-output rows (window rows + the taps a last, partly idle row segment reaches)
constexpr int C2_MBW = 8;                       // ReLU bit table: [68 rows][8 waves] uint16

// Global traffic of the kernel: raw buffer instructions, every one of them executed by every wave -- a row that is not to be stored (or
// loaded) gets an out-of-range offset (the hardware drops the store / returns zeros).  With loads or stores inside branches the compiler no
// longer knows how many are outstanding, every wait for a weight fragment becomes vmcnt(0), and the wave sits out the acknowledgement of the
// stores it issued a moment ago (a layer's GEMM phase opened with ~5 k cycles of that: profiles/r05_notes.md).
typedef __amdgpu_buffer_rsrc_t brsrc_t;
constexpr uint32_t BUF_OOB = 0x80000000u;
__device__ __forceinline__ brsrc_t buf_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ void buf_store4(brsrc_t r, uint32_t off, const float4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, r, off, 0, 0);
}
__device__ __forceinline__ float4 buf_load4(brsrc_t r, uint32_t off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}

template <bool DROP>
__global__ __launch_bounds__(CB_T, 2) void k_convblock_fwd2(CbFwdArgs a) {
    constexpr int HL = 12, NW = TILE_M + 2 * HL, NT = CB_T;
    constexpr int LPR = 8, NJ = 32 / LPR;                      // LayerNorm: lanes per row, float4 columns per lane (4 sub + 4 LPR j)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                                          // [68][LDP] residual stream (window row index)
    float* VU = Xs + C2_XR * LDP;                              // [68][LDP] LayerNorm output
    uint16_t* Ub = reinterpret_cast<uint16_t*>(VU + C2_VUR * LDP);   // depthwise output: three bf16 planes [56][CB_LDB]
    constexpr int UPS = NW * CB_LDB;
    float* Ps = VU + C2_VUR * LDP + 3 * UPS / 2;               // [4][CB_PS] | ln1_g | ln1_b | bq | bk | bv
    uint16_t* MB = reinterpret_cast<uint16_t*>(Ps + 4 * CB_PS + 640);
    float* TW = Ps + 4 * CB_PS + 640 + C2_XR * C2_MBW / 2;     // depthwise taps of the four layers, [layer][tap][channel]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = w;                                          // GEMM role: column block (16 output columns = attention head w)
    const int L = a.L;
    const int r0 = blockIdx.x * TILE_M, rw0 = r0 - HL;
    const uint32_t rbytes = (uint32_t)a.R * (D * 4);           // bytes of an (R, 128) fp32 tensor
    ESTAMP(0);
    constexpr int NQ = (NW * 32 + NT - 1) / NT;                // float4 items per thread of the window
    float4 xv[NQ], pv[NQ];
    const int s_own = r0 / L;
    const int klo = max(0, s_own * L - rw0), khi = min(NW, (s_own + 1) * L - rw0);      // window rows of the owner sample
    {
        const brsrc_t rx = buf_rsrc(a.xin, rbytes), rp = buf_rsrc(a.pos, (uint32_t)L * (D * 4));
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = tid + q * NT;
            const int wr = e >> 5, c = (e & 31) * 4;
            const bool in = e < NW * 32 && wr >= klo && wr < khi;       // rows of other samples only ever act as zero padding
            xv[q] = buf_load4(rx, in ? (uint32_t)(((rw0 + wr) * D + c) * 4) : BUF_OOB);
            pv[q] = buf_load4(rp, in ? (uint32_t)(((rw0 + wr - s_own * L) * D + c) * 4) : BUF_OOB);
        }
    }
    if (tid < 384) {
        float pl[4], pq[2];
#pragma unroll
        for (int l = 0; l < 4; ++l) pl[l] = tid < 128 ? a.ln_g[l][tid] : tid < 256 ? a.ln_b[l][tid - 128] : a.pw_b[l][tid - 256];
        pq[0] = tid < 128 ? a.qf.ln_g[tid] : tid < 256 ? a.qf.ln_b[tid - 128] : 0.f;
        pq[1] = tid < 128 ? a.qf.bq[tid] : tid < 256 ? a.qf.bk[tid - 128] : a.qf.bv[tid - 256];
#pragma unroll
        for (int l = 0; l < 4; ++l) Ps[l * CB_PS + tid] = pl[l];
        if (tid < 256) Ps[4 * CB_PS + tid] = pq[0];
        Ps[4 * CB_PS + 256 + tid] = pq[1];
    }
    {
        const brsrc_t r0o = buf_rsrc(a.x0_out, rbytes);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = tid + q * NT;
            const int wr = e >> 5, c = (e & 31) * 4;
            const float4 v = make_float4(xv[q].x + pv[q].x, xv[q].y + pv[q].y, xv[q].z + pv[q].z, xv[q].w + pv[q].w);
            buf_store4(r0o, (e < NW * 32 && wr >= HL && wr < HL + TILE_M) ? (uint32_t)(((rw0 + wr) * D + c) * 4) : BUF_OOB, v);
            if (e < NW * 32) *reinterpret_cast<float4*>(&Xs[wr * LDP + c]) = v;
        }
    }
    B3 b3A, b3B;                                               // the wave's 128 x 16 weight slice of the current / next stage
    b3_load(b3A, a.W3[0], D, D, 16 * cb);
    {   // depthwise taps of all four layers -> LDS, transposed to [tap][channel] (coalesced loads once per launch instead of 14 strided loads per layer and thread)
        float tw[4][2];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const brsrc_t rt = buf_rsrc(a.dw_w[l], D * DWK * 4);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) tw[l][hh] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rt, tid < 448 ? (uint32_t)((tid + 448 * hh) * 4) : BUF_OOB, 0, 0));
        }
        if (tid < 448) {
#pragma unroll
            for (int l = 0; l < 4; ++l)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) { const int e = tid + 448 * hh; TW[l * (D * DWK) + (e % DWK) * D + e / DWK] = tw[l][hh]; }
        }
    }
    f32x2 wk2[DWK];                                          // depthwise taps of the thread's channel pair (2 lane, 2 lane + 1)
    __syncthreads();
    ESTAMP(1);
    const int col = 16 * cb + (lane & 15), g4 = 4 * (lane >> 4);
    auto grp_sum = [](float v) { return grp8_sum(v); };       // sum over the LPR lanes of a row
    constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};       // (a term, b term): mm, hl, lh, hm, mh, hh -- small terms first

    // The next stage's weight slice (twelve 16-byte loads per lane) is requested ONE load at a time between pieces of this layer's work: a
    // wave that issues the twelve together sits at the full request queue (k_convblock_bwd; without the prefetch a layer is 1.6 k cycles shorter).
    auto layer = [&](auto LC, auto& cur, auto& nxt, const uint16_t* __restrict__ Wn, auto NCc) {
        constexpr int l = decltype(LC)::value;
        constexpr int ncn = decltype(NCc)::value;
        auto preB = [&](auto IC) {
            constexpr int I = decltype(IC)::value;
            __builtin_amdgcn_sched_barrier(0);
            b3_load_one(nxt, Wn, D, ncn, 16 * cb, I % 3, I / 3);
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int k = 0; k < DWK; ++k) wk2[k] = *reinterpret_cast<const f32x2*>(&TW[l * (D * DWK) + k * D + 2 * lane]);
        constexpr int in0 = 3 * l, nin = NW - 6 * l;             // LayerNorm rows
        constexpr int o0 = in0 + 3, n = nin - 6;                 // rows this layer produces
        constexpr int NRB = (n + 15) / 16, QS = (n + 7) / 8;     // 16-row blocks ; rows per depthwise segment (one per wave)
        const float* P = Ps + l * CB_PS;
        const Drop dp = a.dp[l];
        // ---- LayerNorm, LPR lanes per row (lane sub: columns 4 sub + 4 LPR j); owner rows of x_l = y[l - 1] leave for memory on the way
        {
            const int r = tid / LPR, sub = tid % LPR;
            const bool act = r < nin;
            const int wr = act ? in0 + r : in0;
            const float* s = Xs + wr * LDP + sub * 4;
            float4 v[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) v[j] = *reinterpret_cast<const float4*>(s + 4 * LPR * j);
            if (l > 0) {
                const brsrc_t ry = buf_rsrc(a.y[l > 0 ? l - 1 : 0], rbytes);
                const uint32_t yo = (act && wr >= HL && wr < HL + TILE_M) ? (uint32_t)(((rw0 + wr) * D + sub * 4) * 4) : BUF_OOB;
#pragma unroll
                for (int j = 0; j < NJ; ++j) buf_store4(ry, yo + 16 * LPR * j, v[j]);
            }
            preB(std::integral_constant<int, 0>());
            if (act) {
                float* d = VU + wr * LDP + sub * 4;
                if (wr < klo || wr >= khi) {
                    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) *reinterpret_cast<float4*>(d + 4 * LPR * j) = z;
                } else {
                    f32x2 x[2 * NJ];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) { x[2 * j] = f32x2{v[j].x, v[j].y}; x[2 * j + 1] = f32x2{v[j].z, v[j].w}; }
                    f32x2 s2 = x[0] + x[1];
#pragma unroll
                    for (int i = 2; i < 2 * NJ; i += 2) s2 += x[i] + x[i + 1];
                    const float mu = grp_sum(s2.x + s2.y) * (1.0f / D);
                    const f32x2 nmu = pk2(-mu);
#pragma unroll
                    for (int i = 0; i < 2 * NJ; ++i) x[i] += nmu;
                    f32x2 q2 = x[0] * x[0], q3 = x[1] * x[1];
#pragma unroll
                    for (int i = 2; i < 2 * NJ; i += 2) { q2 = pk_fma(x[i], x[i], q2); q3 = pk_fma(x[i + 1], x[i + 1], q3); }
                    q2 += q3;
                    const float rstd = rsqrtf(grp_sum(q2.x + q2.y) * (1.0f / D) + LN_EPS);
                    const f32x2 rs2 = pk2(rstd);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const float4 gv = *reinterpret_cast<const float4*>(P + sub * 4 + 4 * LPR * j);
                        const float4 bb = *reinterpret_cast<const float4*>(P + 128 + sub * 4 + 4 * LPR * j);
                        const f32x2 o0v = pk_fma(x[2 * j] * rs2, f32x2{gv.x, gv.y}, f32x2{bb.x, bb.y});
                        const f32x2 o1v = pk_fma(x[2 * j + 1] * rs2, f32x2{gv.z, gv.w}, f32x2{bb.z, bb.w});
                        *reinterpret_cast<float4*>(d + 4 * LPR * j) = make_float4(o0v.x, o0v.y, o1v.x, o1v.y);
                    }
                }
            }
            preB(std::integral_constant<int, 1>());
        }
        __syncthreads();
        if (l == 0) ESTAMP(8);
        // ---- depthwise conv k = 7: thread = (channel pair, row segment = wave); window of QS + 6 rows in registers.  A wave whose segment
        // lies beyond the layer's rows recomputes segment 0 and discards it (the same instruction stream in every wave)
        {
            const bool sact = w * QS < n;
            const int os = sact ? o0 + w * QS : o0;              // first produced window row of the segment (wave-uniform)
            f32x2 win[QS + 2 * HALO];
#pragma unroll
            for (int i = 0; i < QS + 2 * HALO; ++i) win[i] = *reinterpret_cast<const f32x2*>(&VU[(os - HALO + i) * LDP + 2 * lane]);
            uint32_t* ub = reinterpret_cast<uint32_t*>(Ub) + lane;
            const brsrc_t ru = buf_rsrc(a.u[l], rbytes);
#pragma unroll
            for (int i = 0; i < QS; ++i) {
                f32x2 u = wk2[0] * win[i];
#pragma unroll
                for (int k = 1; k < DWK; ++k) u = pk_fma(wk2[k], win[i + k], u);
                const int row = os + i;
                const bool rok = sact && row < o0 + n;           // wave-uniform
                // saved: A operand of the weight gradient (owner rows; a wave stores whole 512-byte rows)
                __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{__float_as_uint(u.x), __float_as_uint(u.y)}, ru,
                                                      (rok && row >= HL && row < HL + TILE_M) ? (uint32_t)(((rw0 + row) * D + 2 * lane) * 4) : BUF_OOB, 0, 0);
                if (rok) {
                    uint32_t th, tm, tl;
                    split3(u.x, u.y, th, tm, tl);
                    uint32_t* dpl = ub + row * (CB_LDB / 2);
                    dpl[0] = th; dpl[UPS / 2] = tm; dpl[UPS] = tl;
                }
                if (i == 0) preB(std::integral_constant<int, 2>()); else if (i == 1) preB(std::integral_constant<int, 3>());
                else if (i == 2) preB(std::integral_constant<int, 4>()); else if (i == 3) preB(std::integral_constant<int, 5>());
            }
        }
        __syncthreads();
        if (l == 0) ESTAMP(9);
        // ---- pointwise GEMM: this wave = 16 output columns x every row block, one block at a time, the epilogue of the previous block between
        // the MFMAs.  What the stamps and knock-out builds said on the way here (profiles/r05_notes.md): one accumulator chain per 16 x 16 tile
        // already runs the matrix pipe at its full rate (tools/ubench/mfma_rate.hip), the two waves of a SIMD take turns on it (the older one
        // first), and a request for A fragments per K step costs an LDS round trip per step -- a block's twelve fragments are requested one
        // block ahead instead; the residual values the epilogue updates are read ahead as well.
        const float bv = P[256 + col];
        const int row0 = o0 + g4;                                // window row of the lane's first element in block 0
        const uint32_t hb = (uint32_t)((rw0 + row0) * D + col) * 0x9E3779B1u + dp.seed;    // hash input of that element
        const int mybit = 1 << (lane & 15);
        float* xb = Xs + row0 * LDP + col;
        uint16_t* mb = MB + (row0 + (lane & 3)) * C2_MBW + cb;
        const uint16_t* ar = Ub + (o0 + (lane & 15)) * CB_LDB + 8 * (lane >> 4);
        Frag3 af[2][4];
        auto areadblk = [&](int i, Frag3 (&f)[4]) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int t = 0; t < 3; ++t) f[ks].t[t] = *reinterpret_cast<const u32x4_t*>(ar + t * UPS + i * 16 * CB_LDB + ks * 32);
        };
        f32x4 accp = {0.f, 0.f, 0.f, 0.f};
        float xold[4];
        int pos01 = 0, pos23 = 0;
        auto xread = [&](auto Ic) {                              // residual values of the lane's four rows of block i
            constexpr int i = decltype(Ic)::value;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) xold[rr] = xb[(16 * i + rr) * LDP];
        };
        // one quarter (row rr of the lane's four) of the epilogue of block i
        auto epi = [&](auto Ic, auto RRc) {
            constexpr int i = decltype(Ic)::value, rr = decltype(RRc)::value;
            const float z = accp[rr] + bv;
            float av = fmaxf(z, 0.f);
            if (DROP) {
                uint32_t h = hb + (uint32_t)((16 * i + rr) * D) * 0x9E3779B1u;
                h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= (h >> 13) ^ dp.key; h *= 0xC2B2AE35u;       // = drop_hash() with its first multiply-add hoisted (hb)
                av = h >= dp.thresh ? av * dp.scale : 0.f;
            }
            xb[(16 * i + rr) * LDP] = xold[rr] + av;
            const int bit = z > 0.f ? (mybit << (16 * (rr & 1))) : 0;
            if (rr == 0) pos01 = bit; else if (rr == 1) pos01 |= bit; else if (rr == 2) pos23 = bit; else pos23 |= bit;
            if (rr == 3) {
                auto row_or = [](int v) {
                    v |= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
                    v |= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);       // quad_perm [2,3,0,1]
                    v |= __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);      // row_half_mirror
                    v |= __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);      // row_mirror
                    return v;
                };
                const int p01 = row_or(pos01), p23 = row_or(pos23);
                // lane q of the 16-lane row writes row (q & 3): four lanes write the same value to the same address (no exec masking)
                const uint32_t wv = (uint32_t)((lane & 2) ? p23 : p01) >> (16 * (lane & 1));
                mb[16 * i * C2_MBW] = (uint16_t)wv;
            }
        };
        areadblk(0, af[0]);
        static_for<0, NRB>([&](auto Ic) {
            constexpr int i = decltype(Ic)::value;
            if constexpr (i > 0) xread(std::integral_constant<int, (i > 0 ? i - 1 : 0)>());
            if constexpr (i + 1 < NRB) areadblk(i + 1, af[(i + 1) & 1]);
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
            static_for<0, 4>([&](auto KSc) {
                constexpr int ks = decltype(KSc)::value;
                const Frag3& f = af[i & 1][ks];
                c0 = mfma16_bf16(f.t[1], cur.b[1][ks], c0);      // mm
                c1 = mfma16_bf16(f.t[0], cur.b[2][ks], c1);      // hl
                c0 = mfma16_bf16(f.t[2], cur.b[0][ks], c0);      // lh
                c1 = mfma16_bf16(f.t[0], cur.b[1][ks], c1);      // hm
                c0 = mfma16_bf16(f.t[1], cur.b[0][ks], c0);      // mh
                c1 = mfma16_bf16(f.t[0], cur.b[0][ks], c1);      // hh
                if constexpr (i > 0) epi(std::integral_constant<int, (i > 0 ? i - 1 : 0)>(), KSc);
                if constexpr (i < 2 && (ks & 1)) preB(std::integral_constant<int, 6 + 2 * (i < 2 ? i : 0) + (ks >> 1)>());      // loads 6 .. 9
            });
            accp = c0 + c1;
        });
        xread(std::integral_constant<int, NRB - 1>());
        preB(std::integral_constant<int, 10>());
        preB(std::integral_constant<int, 11>());
        static_for<0, 4>([&](auto RRc) { epi(std::integral_constant<int, NRB - 1>(), RRc); });
        if (l == 0) ESTAMP(10);
        __syncthreads();
        if (l == 0) ESTAMP(11);
        // ReLU decisions of the owner rows: one 16-byte store per row (the (R, 4) uint32 words)
        {
            const u32x4_t mv = *reinterpret_cast<const u32x4_t*>(MB + (HL + (tid & 31)) * C2_MBW);
            __builtin_amdgcn_raw_buffer_store_b128(mv, buf_rsrc(a.relu_mask[l], (uint32_t)a.R * 16), tid < TILE_M ? (uint32_t)((r0 + tid) * 16) : BUF_OOB, 0, 0);
        }
    };
    layer(std::integral_constant<int, 0>(), b3A, b3B, a.W3[1], std::integral_constant<int, D>());
    ESTAMP(2);
    layer(std::integral_constant<int, 1>(), b3B, b3A, a.W3[2], std::integral_constant<int, D>());
    ESTAMP(3);
    layer(std::integral_constant<int, 2>(), b3A, b3B, a.W3[3], std::integral_constant<int, D>());
    ESTAMP(4);
    layer(std::integral_constant<int, 3>(), b3B, b3A, a.Wqkv3, std::integral_constant<int, 3 * D>());
    ESTAMP(5);
    // ---- a8, first half (:168-173) on the owner rows: y3 -> memory ; h1 = drop(LN1(y3)) ; [q | k | v] = h1 W^T + b  (column block cb = head cb)
    {
        const float* Pq = Ps + 4 * CB_PS;
        const Drop d1 = a.qf.d1;
        // LayerNorm 1 in the row layout of the layers (LPR lanes per row, 32 rows): output to memory (h1) and, split, to the GEMM operand planes
        {
            const int r = tid / LPR, sub = tid % LPR;
            const bool act = r < TILE_M;
            const int wr = HL + (act ? r : 0);
            const float* s = Xs + wr * LDP + sub * 4;
            float4 v[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) v[j] = *reinterpret_cast<const float4*>(s + 4 * LPR * j);
            const uint32_t go = act ? (uint32_t)(((rw0 + wr) * D + sub * 4) * 4) : BUF_OOB;
            const brsrc_t ry = buf_rsrc(a.y[3], rbytes);
#pragma unroll
            for (int j = 0; j < NJ; ++j) buf_store4(ry, go + 16 * LPR * j, v[j]);
            f32x2 x[2 * NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) { x[2 * j] = f32x2{v[j].x, v[j].y}; x[2 * j + 1] = f32x2{v[j].z, v[j].w}; }
            f32x2 s2 = x[0] + x[1];
#pragma unroll
            for (int i = 2; i < 2 * NJ; i += 2) s2 += x[i] + x[i + 1];
            const float mu = grp_sum(s2.x + s2.y) * (1.0f / D);
            const f32x2 nmu = pk2(-mu);
#pragma unroll
            for (int i = 0; i < 2 * NJ; ++i) x[i] += nmu;
            f32x2 q2 = x[0] * x[0], q3 = x[1] * x[1];
#pragma unroll
            for (int i = 2; i < 2 * NJ; i += 2) { q2 = pk_fma(x[i], x[i], q2); q3 = pk_fma(x[i + 1], x[i + 1], q3); }
            q2 += q3;
            const float rstd = rsqrtf(grp_sum(q2.x + q2.y) * (1.0f / D) + LN_EPS);
            const f32x2 rs2 = pk2(rstd);
            const brsrc_t rhb = buf_rsrc(a.qf.h1, a.qf.h1 ? rbytes : 0u);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float4 gv = *reinterpret_cast<const float4*>(Pq + sub * 4 + 4 * LPR * j);
                const float4 bb = *reinterpret_cast<const float4*>(Pq + 128 + sub * 4 + 4 * LPR * j);
                f32x2 o0v = pk_fma(x[2 * j] * rs2, f32x2{gv.x, gv.y}, f32x2{bb.x, bb.y});
                f32x2 o1v = pk_fma(x[2 * j + 1] * rs2, f32x2{gv.z, gv.w}, f32x2{bb.z, bb.w});
                if (DROP) {
                    const uint32_t base = (uint32_t)((rw0 + wr) * D + sub * 4 + 4 * LPR * j);
                    o0v.x *= drop_keep_scale(d1, base); o0v.y *= drop_keep_scale(d1, base + 1);
                    o1v.x *= drop_keep_scale(d1, base + 2); o1v.y *= drop_keep_scale(d1, base + 3);
                }
                buf_store4(rhb, go + 16 * LPR * j, make_float4(o0v.x, o0v.y, o1v.x, o1v.y));
                if (act) {
                    uint32_t h0, m0, l0, h1, m1, l1;
                    split3(o0v.x, o0v.y, h0, m0, l0);
                    split3(o1v.x, o1v.y, h1, m1, l1);
                    uint16_t* d = Ub + (wr - HL) * CB_LDB + sub * 4 + 4 * LPR * j;
                    *reinterpret_cast<u32x2_t*>(d) = u32x2_t{h0, h1};
                    *reinterpret_cast<u32x2_t*>(d + UPS) = u32x2_t{m0, m1};
                    *reinterpret_cast<u32x2_t*>(d + 2 * UPS) = u32x2_t{l0, l1};
                }
            }
        }
        b3_load(b3B, a.Wqkv3, D, 3 * D, D + 16 * cb);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        // the 32 rows of h1: the A fragments of both row blocks and all four K steps, read once for the three projections
        Frag3 aq[2][4];
        {
            const uint16_t* ap = Ub + (lane & 15) * CB_LDB + 8 * (lane >> 4);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int t = 0; t < 3; ++t) aq[rb][ks].t[t] = *reinterpret_cast<const u32x4_t*>(ap + t * UPS + rb * 16 * CB_LDB + ks * 32);
        }
        auto proj = [&](const B3& cur, float* __restrict__ outp, int t) {
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) acc[rb] = mfma16_bf16(aq[rb][ks].t[TA[p]], cur.b[TB[p]][ks], acc[rb]);
            const float bvv = Pq[256 + t * D + col];
            const brsrc_t ro = buf_rsrc(outp, rbytes);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[rb][rr] + bvv), ro, (uint32_t)(((r0 + 16 * rb + g4 + rr) * D + col) * 4), 0, 0);
        };
        proj(b3A, a.qf.q, 0);
        __builtin_amdgcn_sched_barrier(0);
        b3_load(b3A, a.Wqkv3, D, 3 * D, 2 * D + 16 * cb);
        __builtin_amdgcn_sched_barrier(0);
        proj(b3B, a.qf.k, 1);
        proj(b3A, a.qf.v, 2);
    }
    ESTAMP(6);
}
constexpr size_t cb_fwd2_lds() {
    return (size_t)((C2_XR + C2_VUR) * LDP + 3 * (TILE_M + 24) * CB_LDB / 2 + 4 * CB_PS + 640 + C2_XR * C2_MBW / 2 + 4 * D * DWK) * sizeof(float);
}
template <bool DROP>
static void launch_cbf2_t(const CbFwdArgs& a, int grid, hipStream_t s) {
    static size_t ok = 0;
    const size_t lds = cb_fwd2_lds();
    ensure_dynamic_lds((const void*)k_convblock_fwd2<DROP>, lds, ok, "k_convblock_fwd2");
    VSL_LAUNCH((k_convblock_fwd2<DROP>), dim3(grid), dim3(CB_T), lds, s, a);
}
static void launch_cbf2(const CbFwdArgs& a, int grid, hipStream_t s) {
    if (a.dp[0].thresh) launch_cbf2_t<true>(a, grid, s); else launch_cbf2_t<false>(a, grid, s);
}
template <int SH, bool FULL>
static void launch_cbf(const CbFwdArgs& a, int grid, hipStream_t s) {
    static size_t ok = 0;
    const size_t lds = cb_fwd_lds_split(SH);
    ensure_dynamic_lds((const void*)k_convblock_fwd<SH, FULL>, lds, ok, "k_convblock_fwd");
    VSL_LAUNCH((k_convblock_fwd<SH, FULL>), dim3(grid), dim3(CB_T), lds, s, a);
}
void launch_convblock_fwd(const CbFwdArgs& a, hipStream_t s) {
    if (a.L <= TILE_M) {                    // sample tiles: one workgroup per sample
        launch_cbf<0, false>(a, a.R / a.L, s);
        return;
    }
    static const bool cb2 = !(getenv("VSL_CB2") && atoi(getenv("VSL_CB2")) == 0);
    if (a.R % TILE_M == 0 && a.L % TILE_M == 0) {              // whole tiles inside one sample each
        if (cb2) launch_cbf2(a, a.R / TILE_M, s);
        else launch_cbf<3, true>(a, a.R / TILE_M, s);
    } else launch_cbf<3, false>(a, (a.R + TILE_M - 1) / TILE_M, s);
    static int left = 6;
    if (edbg_on() && a.R > 4096) { int l2 = left, l3 = left; edbg_report("convblock_fwd: load | L0 | L1 | L2 | L3 | qkv", 7, s, left); edbg_report2("  L0: LN | dw | gemm | epilogue | barrier", 8, 13, s, l2); edbg_report2("  L0 gemm: (stamp 14 = group 0 done) group 1", 14, 16, s, l3); }
}

// =========================================================================================================
// backward of the conv block (autograd of layers_t7.py:131-140, four layers in one launch).  dy = grad wrt the block output on
// the 56-row window; per layer l = 3..0, on a row range that shrinks by 3 rows each side:
//   A  dz = dy * relu-bit * dropout           (saved to gz[l] on the owner rows: G operand of the pointwise weight gradient)
//   B  du = dz Wp                              (16x16x4 MFMA, 4 / 4 / 3 / 3 row blocks)
//   C  dv = depthwise^T(du) ; per-tile partials of the depthwise taps, LayerNorm gamma / beta over the OWNER rows
//   D  dy <- dy + LN^T(dv)                     (becomes the dy of the layer below; layer 0 writes dx0)
// The LayerNorm input x_l of a layer is re-normalised in LDS (xhat, rstd) for C and D.  Phase C works per (channel, row
// segment) with the 4 segments of a channel in 4 adjacent lanes, so the partial sums are combined with two quad shuffles.
// LDS 91 KB: fits beside a weight-gradient workgroup (66 KB) of the side stream.
// =========================================================================================================
// du = dz Wp on the bf16 matrix cores at fp32 grade (gemm16s): phase A stores dz as three bf16 planes, which share their LDS with dv
// (GU) -- dz is dead before phase C writes dv, but the next layer's phase A overwrites what phase D still reads: one more barrier per layer.
// TAIL (whole tiles only): the workgroup goes on with a row-tile kernel's body on its 32 rows of dx0 (tile_bodies.hpp): 1 = attention-output
// backward of the encoder pass below, 2 = CQConcatenate backward.  Waves 0-3 work, the others join the barriers.
// QKV (whole tiles only): the workgroup first does k_qkv_bwd's work on its whole 56-row window (CbBwdArgs::qk) -- dy is born in LDS.  The halo
// rows are recomputed (1.75 x the K = 384 product), but the product's price is its 288 KB weight stream per workgroup, which a 32-row and a 56-row
// tile pay alike; what goes away is a launch of the dependent chain with its own cold start, and dy's trip through memory.
template <int SH, bool FULL, int TAIL = 0, bool QKV = false>        // SH 3: row tiles with a recomputed halo ; 0: sample tiles for L <= 32 ; FULL: see k_convblock_fwd
__global__ __launch_bounds__(CB_T, 2) void k_convblock_bwd(CbBwdArgs a) {
    constexpr int HL = 4 * SH, NW = TILE_M + 2 * HL, VUR = SH ? NW + 12 : NW, XR = NW - 2 * SH;
    constexpr int ZPS = NW * CB_LDB;                     // elements between the dz planes
    constexpr int GUF = 3 * ZPS / 2 > VUR * LDP ? 3 * ZPS / 2 : VUR * LDP;      // floats of the dz / dv region
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* DY = smem;                       // [56][LDP] grad wrt the current layer's output (in place)
    float* GU = DY + NW * LDP;           // [68][LDP] dz (GEMM A operand), later dv
    uint16_t* Pz = reinterpret_cast<uint16_t*>(GU);      // dz as three bf16 planes [NW][CB_LDB]
    float* DU = GU + GUF;                // [68][LDP] du: its own buffer, so neither the GEMM's reads nor the conv windows need a barrier of their own
    float* Xh = DU + VUR * LDP;              // [50][LDP] x_l, normalised in place; row = window row - SH
    float* RS = Xh + XR * LDP;           // [64] rstd per window row
    float* VF = RS + 64;                    // [64] 1 = the window row belongs to the owner sample / is inside [0, R)
    float* GB = VF + 64;                    // [2][128] gamma of the current layer (row layout reads), double-buffered across layers
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int R = a.R, L = a.L;
    const int r0 = SH ? blockIdx.x * TILE_M : blockIdx.x * L, rw0 = r0 - HL;
    const bool interior = SH && rw0 >= 0 && rw0 + NW <= R && (rw0 % L) + NW <= L;
    const bool full = FULL || (SH && r0 + TILE_M <= R);
    const int s_own = r0 / L;
    const bool one_owner = FULL || !SH || (full && (r0 + TILE_M - 1) / L == s_own);
    const int klo = max(0, s_own * L - rw0), khi = min(NW, (s_own + 1) * L - rw0);
    const bool plain = FULL || interior || one_owner;
    auto row_ok = [&](int wr) { return FULL || (SH ? (full || rw0 + wr < R) : wr < L); };     // may window row wr (an owner row) be stored?
    auto row_in = [&](int wr) { return wr < NW && (SH ? (rw0 + wr >= 0 && rw0 + wr < R) : wr < L); };   // does it exist (in this sample)?
    // row layout of the loads and of phases A, LN, D: 8 lanes per window row, lane `sub` owns the columns 4 sub + 32 j .. + 3 (j = 0..3).
    // One thread keeps one row for the whole kernel, so what phase D of a layer writes (dy) and reads (dv, xhat) is exactly what phase
    // A of the next layer reads / overwrites: no barrier between them.
    const int r8 = tid >> 3, sub = tid & 7;
    // column layout of phase C
    const int cc = tid >> 2, seg = tid & 3;
    ESTAMP(0);
    float4 xv[4];
    uint32_t mw[4];
    // Global traffic as in k_convblock_fwd2: raw buffer instructions that EVERY wave executes, rows that do not exist / are not to be stored get
    // an out-of-range offset (zeros / dropped).  With loads or stores inside divergent branches the compiler cannot count what is outstanding and
    // every wait for a prefetched operand becomes vmcnt(0) -- i.e. a wait for the acknowledgement of the stores issued a moment ago.
    const uint32_t rbytes = (uint32_t)R * (D * 4);
    const bool rin = row_in(r8);
    const uint32_t rowoff = rin ? (uint32_t)(((rw0 + r8) * D + sub * 4) * 4) : BUF_OOB;       // this thread's row in an (R, 128) tensor
    auto fetch_part = [&](int l, int q) {   // one quarter of fetch_layer
        const brsrc_t rx = buf_rsrc(a.x[l], rbytes), rm = buf_rsrc(a.relu_mask[l], (uint32_t)R * 16);
        const uint32_t mo = rin ? (uint32_t)((rw0 + r8) * 16) : BUF_OOB;
        xv[q] = buf_load4(rx, rowoff + 128 * q);
        mw[q] = __builtin_amdgcn_raw_buffer_load_b32(rm, mo + 4 * q, 0, 0);
    };
    auto fetch_layer = [&](int l) {         // x_l rows and ReLU words of the window, for phase A of layer l
#pragma unroll
        for (int q = 0; q < 4; ++q) fetch_part(l, q);
    };
    B3 b3A, b3B;
    const int col = 16 * w + (lane & 15), g4 = 4 * (lane >> 4);
    if constexpr (QKV) {
        static_assert(((FULL && SH == 3) || SH == 0) && 3 * ZPS / 2 <= (VUR + XR) * LDP, "the q,k,v backward rides on whole tiles and on sample tiles");
        constexpr int NRQ = (NW + 15) / 16;                       // 16-row blocks of the window
        // dh1 = [dQ | dK | dV] [Wq; Wk; Wv] in three K = 128 chunks.  Plane buffers: Pz (the dz planes' place) and P2 (DU | Xh, unused before
        // layer 3) alternate, so a chunk is split and stored while the previous one is multiplied: one barrier per chunk.
        const QkvBwdFuse& qk = a.qk;
        uint16_t* P2 = reinterpret_cast<uint16_t*>(DU);
        const brsrc_t rq = buf_rsrc(qk.dq, rbytes), rk = buf_rsrc(qk.dk, rbytes), rv = buf_rsrc(qk.dv, rbytes);
        const size_t wchunk = (size_t)(D / 16) * D * 16;          // bf16 elements of 128 contraction rows in a split plane
        // two chunks of rows and weights are on their way at any time; a chunk's registers are reloaded as soon as its planes are stored / its
        // product is done (more in flight starves the GEMM of fragment registers: it then reads its A operand one fragment at a time)
        float4 g0[4], g1[4], xq[4], rs[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) g0[q] = buf_load4(rq, rowoff + 128 * q);
        b3_load(b3B, qk.WT3, 3 * D, D, 16 * w);
#pragma unroll
        for (int q = 0; q < 4; ++q) g1[q] = buf_load4(rk, rowoff + 128 * q);
        b3_load(b3A, qk.WT3 + wchunk, 3 * D, D, 16 * w);
        __builtin_amdgcn_sched_barrier(0);
        auto put = [&](uint16_t* Pb, const float4 (&g)[4]) {
            if (r8 < NW) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t h0, m0, l0, h1, m1, l1;
                    split3(g[q].x, g[q].y, h0, m0, l0);
                    split3(g[q].z, g[q].w, h1, m1, l1);
                    uint16_t* d = Pb + r8 * CB_LDB + sub * 4 + 32 * q;
                    *reinterpret_cast<u32x2_t*>(d) = u32x2_t{h0, h1};
                    *reinterpret_cast<u32x2_t*>(d + ZPS) = u32x2_t{m0, m1};
                    *reinterpret_cast<u32x2_t*>(d + 2 * ZPS) = u32x2_t{l0, l1};
                }
            }
        };
        put(Pz, g0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) g0[q] = buf_load4(rv, rowoff + 128 * q);
        ESTAMP(18);
        __syncthreads();
        ESTAMP(19);
        f32x4 acc[1][NRQ];
#pragma unroll
        for (int rb = 0; rb < NRQ; ++rb) acc[0][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm16s<NRQ>(Pz, ZPS, b3B, acc);
        __builtin_amdgcn_sched_barrier(0);
        b3_load(b3B, qk.WT3 + 2 * wchunk, 3 * D, D, 16 * w);
        __builtin_amdgcn_sched_barrier(0);
        put(P2, g1);
        ESTAMP(20);
        __syncthreads();
        ESTAMP(21);
        gemm16s<NRQ>(P2, ZPS, b3A, acc);
        __builtin_amdgcn_sched_barrier(0);
        b3_load(b3A, a.WT3[3], D, D, 16 * w);                     // layer 3's weight slice
        __builtin_amdgcn_sched_barrier(0);
        put(Pz, g0);
        {
            const brsrc_t rx = buf_rsrc(qk.x, rbytes), rdr = buf_rsrc(qk.dr, rbytes);
#pragma unroll
            for (int q = 0; q < 4; ++q) { xq[q] = buf_load4(rx, rowoff + 128 * q); rs[q] = buf_load4(rdr, rowoff + 128 * q); }
        }
        ESTAMP(22);
        __syncthreads();
        ESTAMP(23);
        gemm16s<NRQ>(Pz, ZPS, b3B, acc);
        __builtin_amdgcn_sched_barrier(0);
        fetch_layer(3);
        {   // Ts = dh1 * m1 -> DU (P2 is dead since the last barrier)
            const Drop d1 = qk.d1;
#pragma unroll
            for (int rb = 0; rb < NRQ; ++rb)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int row = 16 * rb + g4 + rr;
                    DU[row * LDP + col] = acc[0][rb][rr] * drop_mul(d1, (uint32_t)((rw0 + row) * D + col));
                }
        }
        ESTAMP(24);
        __syncthreads();
        ESTAMP(25);
        {   // LayerNorm-1 backward per window row (8 lanes per row): dy = dr + rstd (g - mean(g) - xhat mean(g xhat)), g = Ts * gamma
            const int rc = r8 < NW ? r8 : 0;
            const float* tr = DU + rc * LDP + sub * 4;
            float4 ts[4];
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { ts[j] = *reinterpret_cast<const float4*>(tr + 32 * j); sm += sum4(xq[j]); }
            const float mu = grp8_sum(sm) * (1.0f / D);
            float qv = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xq[j].x -= mu; xq[j].y -= mu; xq[j].z -= mu; xq[j].w -= mu;
                qv += xq[j].x * xq[j].x + xq[j].y * xq[j].y + xq[j].z * xq[j].z + xq[j].w * xq[j].w;
            }
            const float rstd = rsqrtf(grp8_sum(qv) * (1.0f / D) + LN_EPS);
            float m1 = 0.f, m2 = 0.f;
            float4 gd[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 gv = *reinterpret_cast<const float4*>(qk.ln_g + sub * 4 + 32 * j);
                xq[j].x *= rstd; xq[j].y *= rstd; xq[j].z *= rstd; xq[j].w *= rstd;                       // xhat
                gd[j] = make_float4(ts[j].x * gv.x, ts[j].y * gv.y, ts[j].z * gv.z, ts[j].w * gv.w);
                m1 += sum4(gd[j]);
                m2 += gd[j].x * xq[j].x + gd[j].y * xq[j].y + gd[j].z * xq[j].z + gd[j].w * xq[j].w;
            }
            m1 = grp8_sum(m1) * (1.0f / D);
            m2 = grp8_sum(m2) * (1.0f / D);
            const bool own = r8 >= HL && r8 < HL + TILE_M;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4 o;
                o.x = rstd * (gd[j].x - m1 - xq[j].x * m2) + rs[j].x; o.y = rstd * (gd[j].y - m1 - xq[j].y * m2) + rs[j].y;
                o.z = rstd * (gd[j].z - m1 - xq[j].z * m2) + rs[j].z; o.w = rstd * (gd[j].w - m1 - xq[j].w * m2) + rs[j].w;
                if (r8 < NW) *reinterpret_cast<float4*>(&DY[r8 * LDP + sub * 4 + 32 * j]) = o;
                // gamma's gradient: Ts * xhat of the OWNER rows, summed per column below (Xh rows 0..31 are free until layer 3)
                if (own) *reinterpret_cast<float4*>(&Xh[(r8 - HL) * LDP + sub * 4 + 32 * j]) =
                    make_float4(ts[j].x * xq[j].x, ts[j].y * xq[j].y, ts[j].z * xq[j].z, ts[j].w * xq[j].w);
            }
        }
        ESTAMP(26);
        __syncthreads();
        ESTAMP(27);
        {   // column sums over the 32 owner rows -> this tile's partial slabs of LN1's gamma / beta
            const int c = tid & 127;
            const float* src = tid < 128 ? Xh + c : DU + HL * LDP + c;
            float accs = 0.f;
#pragma unroll 8
            for (int i = 0; i < TILE_M; ++i) accs += src[i * LDP];
            const brsrc_t rgg = buf_rsrc(qk.p_lng, gridDim.x * (D * 4)), rgb = buf_rsrc(qk.p_lnb, gridDim.x * (D * 4));
            const uint32_t so = (uint32_t)((blockIdx.x * D + c) * 4);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(accs), rgg, tid < 128 ? so : BUF_OOB, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(accs), rgb, (tid >= 128 && tid < 256) ? so : BUF_OOB, 0, 0);
        }
    } else {
        float4 dv[4];
        const brsrc_t rdy = buf_rsrc(a.dy, rbytes);
#pragma unroll
        for (int q = 0; q < 4; ++q) dv[q] = buf_load4(rdy, rowoff + 128 * q);
        fetch_layer(3);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (r8 < NW) *reinterpret_cast<float4*>(&DY[r8 * LDP + sub * 4 + 32 * q]) = dv[q];
        b3_load(b3A, a.WT3[3], D, D, 16 * w);
    }
    float wk[DWK], gc, bc, gnext = 0.f;
#pragma unroll
    for (int k = 0; k < DWK; ++k) wk[k] = a.dw_w[3][cc * DWK + k];
    gc = a.ln_g[3][cc]; bc = a.ln_b[3][cc];
    gnext = a.ln_g[3][tid & (D - 1)];
    if (tid < 64) {
        VF[tid] = (one_owner ? (tid >= klo && tid < khi) : row_in(tid)) ? 1.f : 0.f;
    }
    ESTAMP(1);

    auto layer = [&](auto LC, auto& cur, auto& nxt) {
        constexpr int l = decltype(LC)::value;
        constexpr int ra = SH * (3 - l), rb_ = NW - ra;              // rows of dy / dz / du
        constexpr int n = rb_ - ra, NRB = (n + 15) / 16;
        constexpr int xlo = (ra + SH < HL - SH ? ra + SH : HL - SH);  // x rows kept: [xlo, NW - xlo)
        constexpr int nD = n - 2 * SH;                                // rows of phases C / D: [ra + SH, rb_ - SH)
        constexpr int NHL = SH * l, HQ = (NHL + 1) / 2;               // halo rows per side in phase C, per thread
        const Drop dp = a.dp[l];
        // Prefetch of the layer below as 16 units (12 weight-slice loads, 4 x (x row quarter + ReLU word)) that are issued ONE AT A TIME between
        // pieces of this layer's work: 128 KB per workgroup and layer arrive at ~26 bytes per cycle and CU (5 k cycles), and a wave that
        // issues its twenty loads together sits at the full request queue for that long (profiles/r05_notes.md: without the prefetch a layer
        // is 5 k cycles shorter wherever the block of loads is placed).
        constexpr bool SPREAD = FULL && l > 0;        // (the other instantiations issue the prefetch in one piece at the end of phase C)
        auto preB = [&](auto IC) {                    // weight-slice load IC of 12
            constexpr int I = decltype(IC)::value;
            if constexpr (SPREAD) {
                __builtin_amdgcn_sched_barrier(0);
                b3_load_one(nxt, a.WT3[l > 0 ? l - 1 : 0], D, D, 16 * w, I % 3, I / 3);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto preF = [&](auto IC) {                    // quarter IC of the x_(l-1) row + ReLU words (xv / mw are free once phase A is through)
            if constexpr (SPREAD) {
                __builtin_amdgcn_sched_barrier(0);
                fetch_part(l > 0 ? l - 1 : 0, decltype(IC)::value);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // ---- A: dz = dy * relu-bit * dropout -> GU (+ gz on the owner rows) ; x_l -> Xh ; gamma -> GB
        {
            const bool inA = r8 >= ra && r8 < rb_;
            const int wr = r8, wrc = inA ? r8 : ra;                    // (idle threads compute on row ra and discard)
            const brsrc_t rgz = buf_rsrc(a.gz[l], rbytes);
            const uint32_t gzo = (inA && wr >= HL && wr < HL + TILE_M && row_ok(wr)) ? rowoff : BUF_OOB;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c4 = sub * 4 + 32 * q;
                float4 v = *reinterpret_cast<const float4*>(&DY[wrc * LDP + c4]);
                const uint32_t bits = mw[q] >> (c4 & 31);
                float m[4] = {1.f, 1.f, 1.f, 1.f};
                if (dp.thresh) {
                    const uint32_t base = (uint32_t)((rw0 + wr) * D + c4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) m[i] = drop_keep_scale(dp, base + i);
                }
                v.x = (bits & 1u) ? v.x * m[0] : 0.f;
                v.y = (bits & 2u) ? v.y * m[1] : 0.f;
                v.z = (bits & 4u) ? v.z * m[2] : 0.f;
                v.w = (bits & 8u) ? v.w * m[3] : 0.f;
                if (inA) {
                    uint32_t h0, m0, l0, h1, m1, l1;
                    split3(v.x, v.y, h0, m0, l0);
                    split3(v.z, v.w, h1, m1, l1);
                    uint16_t* d = Pz + wr * CB_LDB + c4;
                    *reinterpret_cast<u32x2_t*>(d) = u32x2_t{h0, h1};
                    *reinterpret_cast<u32x2_t*>(d + ZPS) = u32x2_t{m0, m1};
                    *reinterpret_cast<u32x2_t*>(d + 2 * ZPS) = u32x2_t{l0, l1};
                }
                buf_store4(rgz, gzo + 128 * q, v);                     // owner rows: G operand of the pointwise weight gradient
                if (wr >= xlo && wr < NW - xlo) *reinterpret_cast<float4*>(&Xh[(wr - SH) * LDP + c4]) = xv[q];
                if (q == 0) preB(std::integral_constant<int, 0>()); else if (q == 1) preB(std::integral_constant<int, 1>());
                else if (q == 2) preB(std::integral_constant<int, 2>()); else preB(std::integral_constant<int, 3>());
            }
        }
        float* GBl = GB + (l & 1) * D;
        if (tid < D) GBl[tid] = gnext;
        if (l == 3) ESTAMP(8);
        __syncthreads();
        if (l == 3) ESTAMP(9);
        preB(std::integral_constant<int, 4>());
        // ---- x_l -> xhat in place, rstd per row (8 lanes per row)
        {
            if (r8 >= xlo && r8 < NW - xlo) {
                float* xr = Xh + (r8 - SH) * LDP + sub * 4;
                float4 v[4];
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = *reinterpret_cast<const float4*>(xr + 32 * j); sum += sum4(v[j]); }
                const float mu = grp8_sum(sum) * (1.0f / D);
                float qv = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j].x -= mu; v[j].y -= mu; v[j].z -= mu; v[j].w -= mu;
                    qv += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
                }
                const float rstd = rsqrtf(grp8_sum(qv) * (1.0f / D) + LN_EPS);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<float4*>(xr + 32 * j) = make_float4(v[j].x * rstd, v[j].y * rstd, v[j].z * rstd, v[j].w * rstd);
                if (sub == 0) RS[r8] = rstd;
            }
        }
        preF(std::integral_constant<int, 0>());
        if (l == 3) ESTAMP(10);
        preB(std::integral_constant<int, 5>());
        // ---- B: du = dz Wp
        f32x4 acc[1][NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) acc[0][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm16s<NRB>(Pz + ra * CB_LDB, ZPS, cur, acc);
        if (l == 3) ESTAMP(11);
        preF(std::integral_constant<int, 1>());
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = ra + 16 * rb + g4 + rr;
                DU[row * LDP + col] = acc[0][rb][rr] * VF[min(row, 63)];    // rows of another sample: zero padding of this one's conv
            }
        preB(std::integral_constant<int, 6>());
        if (l == 3) ESTAMP(12);
        __syncthreads();
        if (l == 3) ESTAMP(13);
        // ---- C: dv = depthwise^T(du) ; partial sums over the owner rows
        float dvo[8], dvh[HQ > 0 ? HQ : 1];
        float gw[DWK], slb = 0.f, slg = 0.f;
#pragma unroll
        for (int k = 0; k < DWK; ++k) gw[k] = 0.f;
        const int t0 = HL + 8 * seg;                                  // this thread's 8 owner rows
        const int hs = ((seg >> 1) ? HL + TILE_M : HL - NHL) + (seg & 1) * HQ;   // and its halo rows (l > 0)
        {
            float dwin[14], vv[14], xc[8];
#pragma unroll
            for (int j = 0; j < 14; ++j) {
                const int wr = t0 - 3 + j;                            // sample tiles: rows outside the window = zero padding
                const bool in = SH || (wr >= 0 && wr < NW);
                const int wc = SH ? wr : min(max(wr, 0), NW - 1);
                dwin[j] = in ? DU[wc * LDP + cc] : 0.f;
                const float xh = Xh[(wc - SH) * LDP + cc];
                vv[j] = in ? (xh * gc + bc) * VF[wc] : 0.f;
                if (j >= 3 && j < 11) xc[j - 3] = xh;
            }
            if (plain) {
                static_for<0, 8>([&](auto Ic) {
                    constexpr int i = decltype(Ic)::value;
                    float dv = 0.f;
#pragma unroll
                    for (int k = 0; k < DWK; ++k) { dv += wk[k] * dwin[i + 6 - k]; gw[k] += dwin[i + 3] * vv[i + k]; }
                    dvo[i] = dv;
                    // sample tiles: rows L .. 31 of the window lie outside the sample but receive dv from its last rows
                    slb += SH ? dv : dv * VF[t0 + i];
                    slg += dv * xc[i];
                    if constexpr (i == 1) preF(std::integral_constant<int, 2>());
                    else if constexpr (i == 3) preF(std::integral_constant<int, 3>());
                    else if constexpr (i == 0) preB(std::integral_constant<int, 7>());
                    else if constexpr (i == 2) preB(std::integral_constant<int, 8>());
                    else if constexpr (i < 7) preB(std::integral_constant<int, (i < 7 ? i + 5 : 0)>());       // i = 4, 5, 6 -> loads 9, 10, 11
                });
            } else {
                int p = (r0 + 8 * seg) % L;                             // position of the row inside its sample
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float dv = 0.f;
#pragma unroll
                    for (int k = 0; k < DWK; ++k) {
                        dv += ((unsigned)(p - k + HALO) < (unsigned)L) ? wk[k] * dwin[i + 6 - k] : 0.f;
                        gw[k] += ((unsigned)(p + k - HALO) < (unsigned)L) ? dwin[i + 3] * vv[i + k] : 0.f;
                    }
                    dvo[i] = dv;
                    slb += dv; slg += dv * xc[i];
                    p = p + 1 == L ? 0 : p + 1;
                }
            }
        }
        if (HQ > 0) {
            float hwin[HQ + 6];
#pragma unroll
            for (int j = 0; j < HQ + 6; ++j) hwin[j] = DU[min(max(hs - 3 + j, 0), VUR - 1) * LDP + cc];
            if (plain) {
#pragma unroll
                for (int i = 0; i < HQ; ++i) {
                    float dv = 0.f;
#pragma unroll
                    for (int k = 0; k < DWK; ++k) dv += wk[k] * hwin[i + 6 - k];
                    dvh[i] = dv;
                }
            } else {
                int p = (rw0 + hs) % L;
                p = p < 0 ? p + L : p;
#pragma unroll
                for (int i = 0; i < HQ; ++i) {
                    float dv = 0.f;
#pragma unroll
                    for (int k = 0; k < DWK; ++k) dv += ((unsigned)(p - k + HALO) < (unsigned)L) ? wk[k] * hwin[i + 6 - k] : 0.f;
                    dvh[i] = dv;
                    p = p + 1 == L ? 0 : p + 1;
                }
            }
        }
        // dv goes where dz was (dead since the barrier behind the GEMM)
#pragma unroll
        for (int i = 0; i < 8; ++i) GU[(t0 + i) * LDP + cc] = dvo[i];
        if (HQ > 0) {
#pragma unroll
            for (int i = 0; i < HQ; ++i)
                if ((seg & 1) * HQ + i < NHL) GU[(hs + i) * LDP + cc] = dvh[i];
        }
        // the 4 segments of a channel sit in 4 adjacent lanes
#pragma unroll
        for (int k = 0; k < DWK; ++k) { gw[k] += lane_xor1(gw[k]); gw[k] += lane_xor2(gw[k]); }
        slb += lane_xor1(slb); slb += lane_xor2(slb);
        slg += lane_xor1(slg); slg += lane_xor2(slg);
        {
            const uint32_t nsl = gridDim.x;
            const brsrc_t rpd = buf_rsrc(a.p_dw[l], nsl * (D * DWK * 4)), rpb = buf_rsrc(a.p_lnb[l], nsl * (D * 4)), rpg = buf_rsrc(a.p_lng[l], nsl * (D * 4));
            const uint32_t so = seg == 0 ? (uint32_t)((blockIdx.x * D + cc) * 4) : BUF_OOB;
            const uint32_t sd = seg == 0 ? (uint32_t)((blockIdx.x * D + cc) * (DWK * 4)) : BUF_OOB;
#pragma unroll
            for (int k = 0; k < DWK; ++k) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(gw[k]), rpd, sd + 4 * k, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(slb), rpb, so, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(slg), rpg, so, 0, 0);
        }
        // everything the layer below needs from memory: requested now, consumed after phase D
        if (l > 0) {
            constexpr int lm = l > 0 ? l - 1 : 0;
            if constexpr (!SPREAD) { fetch_layer(lm); b3_load(nxt, a.WT3[lm], D, D, 16 * w); }
#pragma unroll
            for (int k = 0; k < DWK; ++k) wk[k] = a.dw_w[lm][cc * DWK + k];
            gc = a.ln_g[lm][cc]; bc = a.ln_b[lm][cc];
            gnext = a.ln_g[lm][tid & (D - 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (l == 3) ESTAMP(14);
        __syncthreads();
        if (l == 3) ESTAMP(15);
        // ---- D: dy <- dy + LN^T(dv)   (8 lanes per row)
        {
            const bool inD = r8 >= ra + SH && r8 < ra + SH + nD;
            const brsrc_t rdx = buf_rsrc(a.dx0, rbytes);
            if (l == 0 || inD) {                                       // (layer 0: every thread, idle ones on a valid row, so that the store is unconditional)
                const int wr = inD ? r8 : ra + SH;
                const float* dvr = GU + wr * LDP + sub * 4;
                const float* xr = Xh + (wr - SH) * LDP + sub * 4;
                float* dyr = DY + wr * LDP + sub * 4;
                const float rstd = RS[wr];
                float4 gd[4], xh[4];
                float m1 = 0.f, m2 = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 dv = *reinterpret_cast<const float4*>(dvr + 32 * j);
                    const float4 gv = *reinterpret_cast<const float4*>(GBl + sub * 4 + 32 * j);
                    xh[j] = *reinterpret_cast<const float4*>(xr + 32 * j);
                    gd[j] = make_float4(dv.x * gv.x, dv.y * gv.y, dv.z * gv.z, dv.w * gv.w);
                    m1 += sum4(gd[j]);
                    m2 += gd[j].x * xh[j].x + gd[j].y * xh[j].y + gd[j].z * xh[j].z + gd[j].w * xh[j].w;
                }
                m1 = grp8_sum(m1) * (1.0f / D);
                m2 = grp8_sum(m2) * (1.0f / D);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 dy = *reinterpret_cast<const float4*>(dyr + 32 * j);
                    float4 o;
                    o.x = dy.x + rstd * (gd[j].x - m1 - xh[j].x * m2); o.y = dy.y + rstd * (gd[j].y - m1 - xh[j].y * m2);
                    o.z = dy.z + rstd * (gd[j].z - m1 - xh[j].z * m2); o.w = dy.w + rstd * (gd[j].w - m1 - xh[j].w * m2);
                    if ((l > 0 || TAIL) && inD) *reinterpret_cast<float4*>(dyr + 32 * j) = o;       // (TAIL: the owner rows of dx0 stay in LDS for the body that follows)
                    if (l == 0) buf_store4(rdx, (inD && row_ok(wr)) ? rowoff + 128 * j : BUF_OOB, o);
                }
            }
        }
        // no barrier: phase A of the next layer touches only what this thread itself read and wrote above -- except the dz planes of the
        // split path, which lie over the dv rows other threads are still reading
        if (l == 3) ESTAMP(16);
        if (l > 0) __syncthreads();
        if (l == 3) ESTAMP(17);
    };
    __syncthreads();
    layer(std::integral_constant<int, 3>(), b3A, b3B);
    ESTAMP(2);
    layer(std::integral_constant<int, 2>(), b3B, b3A);
    ESTAMP(3);
    layer(std::integral_constant<int, 1>(), b3A, b3B);
    ESTAMP(4);
    layer(std::integral_constant<int, 0>(), b3B, b3A);
    ESTAMP(5);
    if constexpr (TAIL != 0) {
        static_assert(TAIL == 3 ? SH == 0 : (FULL && SH == 3), "tails 1, 2 ride on whole tiles, tail 3 on sample tiles");
        __syncthreads();                                   // dx0's owner rows are in DY; every other LDS region is free
        const bool active = tid < 256;
        if constexpr (TAIL == 1) attn_out_bwd_tile(a.tail_ao, DY + HL * LDP, GU, DU, r0, R, active);
        else if constexpr (TAIL == 2) cqcat_bwd_tile(a.tail_cq, DY + HL * LDP, GU, DU, RS, r0, R, active);
        else linear_bwd_data_tile(DY, Pz, a.tail_lin_WT3, a.tail_lin_dA, r0, L, a.tail_lin_K, a.tail_lin_Kc);
    }
}
constexpr size_t cb_bwd_lds_split(int sh) {
    const int nw = TILE_M + 8 * sh, vur = sh ? nw + 12 : nw, zf = 3 * nw * CB_LDB / 2, guf = zf > vur * LDP ? zf : vur * LDP;
    return (size_t)((nw + vur + TILE_M + 6 * sh) * LDP + guf + 64 + 64 + 256) * sizeof(float);
}
template <int SH, bool FULL, int TAIL = 0, bool QKV = false>
static void launch_cbb(const CbBwdArgs& a, int grid, hipStream_t s) {
    static size_t ok = 0;
    const size_t lds = cb_bwd_lds_split(SH);
    ensure_dynamic_lds((const void*)k_convblock_bwd<SH, FULL, TAIL, QKV>, lds, ok, "k_convblock_bwd");
    VSL_LAUNCH((k_convblock_bwd<SH, FULL, TAIL, QKV>), dim3(grid), dim3(CB_T), lds, s, a);
}
bool convblock_bwd_hosts_tail(int R, int L) { return L > TILE_M && R % TILE_M == 0 && L % TILE_M == 0; }
bool convblock_bwd_hosts_linear(int R, int L) { return L <= TILE_M; }
bool convblock_bwd_hosts_qkv(int R, int L) { return L <= TILE_M || convblock_bwd_hosts_tail(R, L); }
void launch_convblock_bwd(const CbBwdArgs& a, hipStream_t s) {
    if (a.L <= TILE_M) {                    // sample tiles: one workgroup per sample (partial slabs per SAMPLE: convblock_slabs())
        if (a.qkv) { if (a.tail == 3) launch_cbb<0, false, 3, true>(a, a.R / a.L, s); else launch_cbb<0, false, 0, true>(a, a.R / a.L, s); }
        else if (a.tail == 3) launch_cbb<0, false, 3>(a, a.R / a.L, s);
        else launch_cbb<0, false>(a, a.R / a.L, s);
        return;
    }
    if (a.R % TILE_M == 0 && a.L % TILE_M == 0) {
        if (a.qkv) {
            if (a.tail == 1) launch_cbb<3, true, 1, true>(a, a.R / TILE_M, s);
            else if (a.tail == 2) launch_cbb<3, true, 2, true>(a, a.R / TILE_M, s);
            else launch_cbb<3, true, 0, true>(a, a.R / TILE_M, s);
        } else if (a.tail == 1) launch_cbb<3, true, 1>(a, a.R / TILE_M, s);
        else if (a.tail == 2) launch_cbb<3, true, 2>(a, a.R / TILE_M, s);
        else launch_cbb<3, true>(a, a.R / TILE_M, s);
    } else launch_cbb<3, false>(a, (a.R + TILE_M - 1) / TILE_M, s);
    static int left = 6;
    if (edbg_on() && a.R > 4096) { int l2 = left; edbg_report("convblock_bwd: load | L3 | L2 | L1 | L0", 6, s, left); edbg_report2("  L3: (A) sync | xhat | gemm | DU | sync | C | sync | D | sync", 8, 18, s, l2); if (a.qkv) { int l3 = l2 + 1; edbg_report2("  qkv: (from 0: load+split c0) | sync | gemm c0 + split c1 | sync | gemm c1 + split c2 | sync | gemm c2 + Ts | sync | LN bwd | sync | (to stamp 1: sums + rest)", 18, 28, s, l3); } }
}
// partial slabs the backward writes per parameter: one per workgroup
int convblock_slabs(int R, int L) { return L <= TILE_M ? R / L : (R + TILE_M - 1) / TILE_M; }

// =========================================================================================================
// a8, second and third part in ONE launch (layers_t7.py:174-190): attention core + output block.
//   workgroup = (32 queries of one sample) x all 8 heads: wave h = head h, two 16-query blocks, lane = one query column (S^T = K Q^T, so
//   the online softmax is lane-local, as in k_attn_fwd).  K / V fragments come straight from L2 in MFMA operand shape (a key row of a
//   head is 64 contiguous bytes), one key tile ahead -- no LDS staging, no barrier inside the key loop.  The eight 32 x 16 head outputs
//   meet in one LDS tile, and the same workgroup finishes the block on it: r = drop(att) + x -> LN2 -> dropout -> Wo GEMM -> dropout -> + r.
//   Saves att, LSE, r, h2 for the backward exactly like the two-kernel path that serves L > 256.
// =========================================================================================================
// HT (QB = 1 only): the workgroup goes on with both span heads on its tile (AttnBlockArgs::head_tail): threads 0-255 the start head, 256-511 the end
// head, whose features are the y tile this kernel has just produced; the other waves only join the barriers.
template <int QB, bool HT = false>     // 16-query blocks per wave: 2 = 8 waves (wave = head), 1 = 16 waves (wave = head x query block): twice the waves per SIMD
__global__ __launch_bounds__(1024 / QB, QB) void k_attn_block_fwd(AttnBlockArgs a) {
    constexpr int NT = 1024 / QB, NQ = 1024 / NT;      // threads ; float4 items per thread of a 32 x 128 tile
    __shared__ __attribute__((aligned(16))) float Rs[TILE_M * LDP];          // att, then r = drop(att) + x
    __shared__ __attribute__((aligned(16))) float Hs[TILE_M * LDP];          // drop(LN2(r)): GEMM A operand
    __shared__ float Pn[384];                                                 // ln2_g | ln2_b | bo
    extern __shared__ __attribute__((aligned(16))) float Mb[];               // key bias of the sample, padded to whole key tiles
    const int tid = threadIdx.x, h = (tid >> 6) & 7, lane = tid & 63;
    const int qoff = QB == 2 ? 0 : 16 * (tid >> 9);      // first query row of this wave inside the tile
    const int L = a.L, H = 8;
    int bxs, b, bzs;
    xcd_swizzle(bxs, b, bzs);                       // the query tiles of a sample on one XCD: its K / V rows sit in one L2
    const int q0 = bxs * TILE_M;
    const size_t rowbase = (size_t)b * L;
    const int qi = lane & 15, g = lane >> 4;
    ESTAMP(0);
    const int Lp = (L + 15) & ~15;
    for (int k = tid; k < Lp; k += NT) Mb[k] = k < L ? (1.0f - a.mask[rowbase + k]) * MASK_VALUE : MASK_VALUE;
    // per-wave (= per-head) uniform bases + 32-bit lane offsets, so a fragment load is one saddr + voffset instruction
    const int hw = __builtin_amdgcn_readfirstlane(h);
    const float* __restrict__ Qh = a.Q + rowbase * D + hw * HD;
    const float* __restrict__ Kh = a.K + rowbase * D + hw * HD;
    const float* __restrict__ Vh = a.V + rowbase * D + hw * HD;
    float4 qf[QB];
    float m[QB], l[QB];
    f32x4 o[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int q = q0 + qoff + 16 * qb + qi;
        qf[qb] = q < L ? *reinterpret_cast<const float4*>(Qh + (unsigned)(q * D + 4 * g)) : make_float4(0.f, 0.f, 0.f, 0.f);
        m[qb] = -3.0e38f; l[qb] = 0.f; o[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float scale = 0.25f;                    // 1 / sqrt(16), applied AFTER QK^T like the reference (:175)
    // operands of one key tile: K row (kt + qi), cols 4g..4g+3 ; V[kt + 4g + r][qi] ; key bias of keys kt + 4g + r (LDS)
    struct KT { float4 kf; float vv[4]; float4 mb; };
    const unsigned koff = (unsigned)(qi * D + 4 * g), voff = (unsigned)(4 * g * D + qi);
    auto load_kt = [&](int kt, KT& t) {
        if (kt + 16 <= L) {                                       // whole tile: no clamps, constant row strides fold into the instruction
            t.kf = *reinterpret_cast<const float4*>(Kh + (koff + (unsigned)(kt * D)));
#pragma unroll
            for (int r = 0; r < 4; ++r) t.vv[r] = Vh[voff + (unsigned)(kt * D) + (unsigned)(r * D)];
        } else {                                                  // ragged last tile: rows past L read row L - 1 (masked by the key bias)
            t.kf = *reinterpret_cast<const float4*>(Kh + (unsigned)(min(kt + qi, L - 1) * D + 4 * g));
#pragma unroll
            for (int r = 0; r < 4; ++r) t.vv[r] = Vh[(unsigned)(min(kt + 4 * g + r, L - 1) * D + qi)];
        }
    };
    KT cur, nxt;
    load_kt(0, cur);
    // not needed before the output stage: out_layer fragments, the residual rows x, LN2 / bias vectors
    BF16 bf[1];
    bf16_load(bf[0], a.Wpack, D, 16 * h);
    float4 xv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = tid + q * NT, rr = e >> 5, c = (e & 31) * 4;
        xv[q] = q0 + rr < L ? *reinterpret_cast<const float4*>(a.x + (rowbase + q0 + rr) * D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < 384) Pn[tid] = tid < 128 ? a.ln_g[tid] : tid < 256 ? a.ln_b[tid - 128] : a.bo[tid - 256];
    __syncthreads();
    cur.mb = *reinterpret_cast<const float4*>(&Mb[4 * g]);
    ESTAMP(1);
    for (int kt = 0; kt < Lp; kt += 16) {
        if (kt + 16 < Lp) { load_kt(kt + 16, nxt); nxt.mb = *reinterpret_cast<const float4*>(&Mb[kt + 16 + 4 * g]); }
        __builtin_amdgcn_sched_barrier(0);
        const float mbv[4] = {cur.mb.x, cur.mb.y, cur.mb.z, cur.mb.w};
        f32x4 s[QB];
        float p[QB][4], alpha[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            s[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
            s[qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.kf.x, qf[qb].x, s[qb], 0, 0, 0);
            s[qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.kf.y, qf[qb].y, s[qb], 0, 0, 0);
            s[qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.kf.z, qf[qb].z, s[qb], 0, 0, 0);
            s[qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.kf.w, qf[qb].w, s[qb], 0, 0, 0);
        }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float tmax = -3.0e38f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { p[qb][r] = s[qb][r] * scale + mbv[r]; tmax = fmaxf(tmax, p[qb][r]); }
            const float mn = fmaxf(m[qb], kgroup_max(tmax));
            alpha[qb] = __expf(m[qb] - mn);
            m[qb] = mn;
            l[qb] *= alpha[qb];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[qb][r] = __expf(p[qb][r] - mn);
                l[qb] += p[qb][r];
                o[qb][r] *= alpha[qb];
            }
        }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const int q = q0 + qoff + 16 * qb + qi;
            const uint32_t pbase = (uint32_t)(((size_t)(b + a.b_off) * H + h) * L + q) * (uint32_t)L + (uint32_t)(kt + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) p[qb][r] *= drop_mul(a.d2, pbase + r);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) o[qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.vv[r], p[qb][r], o[qb], 0, 0, 0);
        cur = nxt;
    }
    ESTAMP(2);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float lt = kgroup_sum(l[qb]);
        const int q = q0 + qoff + 16 * qb + qi;
        const float inv = 1.0f / lt;
        const float4 ov = make_float4(o[qb][0] * inv, o[qb][1] * inv, o[qb][2] * inv, o[qb][3] * inv);   // lane (qi, g): O[q][4g + reg]
        *reinterpret_cast<float4*>(&Rs[(qoff + 16 * qb + qi) * LDP + h * HD + 4 * g]) = ov;
        if (q < L) {
            *reinterpret_cast<float4*>(a.att + (rowbase + q) * D + h * HD + 4 * g) = ov;
            if (g == 0) a.lse[((size_t)b * H + h) * L + q] = m[qb] + __logf(lt);
        }
    }
    __syncthreads();
    ESTAMP(3);
    // ---- r = drop(att) + x (:183-184)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = tid + q * NT, rr = e >> 5, c = (e & 31) * 4;
        const int r = (int)rowbase + q0 + rr;
        float4 v = *reinterpret_cast<const float4*>(&Rs[rr * LDP + c]);
        if (q0 + rr < L) {
            const uint32_t base = (uint32_t)(r * D + c);
            v.x = v.x * drop_mul(a.d3, base) + xv[q].x; v.y = v.y * drop_mul(a.d3, base + 1) + xv[q].y;
            v.z = v.z * drop_mul(a.d3, base + 2) + xv[q].z; v.w = v.w * drop_mul(a.d3, base + 3) + xv[q].w;
            *reinterpret_cast<float4*>(a.r_out + (size_t)r * D + c) = v;
        } else v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&Rs[rr * LDP + c]) = v;
    }
    __syncthreads();
    ESTAMP(4);
    ln_rows512(Rs, Hs, TILE_M, Pn, Pn + 128, a.d4, (int)rowbase + q0);
    __syncthreads();
    ESTAMP(5);
    if (a.h2_out) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = tid + q * NT, rr = e >> 5, c = (e & 31) * 4;
            if (q0 + rr < L) *reinterpret_cast<float4*>(a.h2_out + (rowbase + q0 + rr) * D + c) = *reinterpret_cast<const float4*>(&Hs[rr * LDP + c]);
        }
    }
    f32x4 acc[1][QB];
#pragma unroll
    for (int rb = 0; rb < QB; ++rb) acc[0][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm16<QB, 1>(Hs + qoff * LDP, LDP, bf, acc);
    const int col = 16 * h + qi;
    const float bv = Pn[256 + col];
    float yv[QB][4];
#pragma unroll
    for (int rb = 0; rb < QB; ++rb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = qoff + 16 * rb + 4 * g + rr;
            yv[rb][rr] = 0.f;
            if (q0 + row < L) {
                const int r = (int)rowbase + q0 + row;
                yv[rb][rr] = (acc[0][rb][rr] + bv) * drop_mul(a.d5, (uint32_t)(r * D + col)) + Rs[row * LDP + col];
                a.y_out[(size_t)r * D + col] = yv[rb][rr];
            }
        }
    ESTAMP(6);
    if constexpr (HT) {
        static_assert(QB == 1, "the span heads ride on the 16-wave instantiation");
        __syncthreads();                                   // every wave is through with Hs (the GEMM's A operand)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Hs[(qoff + 4 * g + rr) * LDP + col] = yv[0][rr];      // the y tile, rows >= L zero
        __syncthreads();
        const int grp = tid >> 8;
        float* tl = Mb + Lp + (grp & 1) * (TILE_M * HEAD_LD + TILE_M * LDP);
        head_fwd_tile<8>(grp == 0 ? a.hs : a.he, a.head_x, a.head_vmask, tl, tl + TILE_M * HEAD_LD, grp == 1 ? Hs : nullptr,
                      rowbase + q0, min(TILE_M, L - q0), tid & 255, grp < 2);
    }
}
void launch_attn_block_fwd(const AttnBlockArgs& a, int B, hipStream_t s) {
    // L <= 128 only: 16 waves (wave = head x 16-query block).  Longer sequences take k_attn_fwd + k_attn_out_fwd (kernels_fwd.hip); the 8-wave
    // instantiation that served 128 < L <= 256 until round 6 is gone (-0.6 % on configs[2] / [3] without it: profiles/r06_notes.md section 9).
    const dim3 grid((a.L + TILE_M - 1) / TILE_M, B);
    const size_t shm = (size_t)((a.L + 15) & ~15) * sizeof(float);
    if (a.head_tail) {
        const size_t shm_t = shm + (size_t)2 * (TILE_M * HEAD_LD + TILE_M * LDP) * sizeof(float);
        static size_t ok = 0;
        ensure_dynamic_lds((const void*)k_attn_block_fwd<1, true>, shm_t, ok, "k_attn_block_fwd");
        VSL_LAUNCH((k_attn_block_fwd<1, true>), grid, dim3(1024), shm_t, s, a);
    } else VSL_LAUNCH(k_attn_block_fwd<1>, grid, dim3(1024), shm, s, a);
    static int left = 3;
    if (edbg_on() && B > 16) edbg_report("attn_block_fwd", 7, s, left);
}

}  // namespace vsl
