// The QUERY branch of VSLNet as sample-local kernels (gfx950): one workgroup (four waves, one per SIMD) per sample, the residual stream of the
// sample's 32-row window in registers, one fp32 tile (17 KB) + two or three bf16 operand-plane buffers (26 KB each) of LDS.
//
// Every kernel of the query branch is sample-local: Lq <= 32 words, and nothing on a word's path needs another sample -- Embedding.linear
// (/root/reference/model/layers_t7.py:83-88), FeatureEncoder at L = Lq (:193-205: positional rows, the four conv layers :131-140, the
// attention block :167-190), applied at /root/reference/model/VSLNet_t7.py:54,56.  As row-tile launches these are 3 (forward) / 3 (backward)
// kernels of 40 - 64 workgroups each plus two launch boundaries per direction; here ONE launch per direction.  (The first version kept to
// 17 KB of LDS and 228 registers so as to sit in the shadow of a video conv-block workgroup -- which turned out to own its CU's whole
// register file: last paragraph.  This version uses what a workgroup that has its CU to itself can use.)
//
// "T layout".  All products are computed TRANSPOSED on the matrix cores:  Y^T[n][m] = sum_k W[n][k] X[m][k]  with the WEIGHT as the MFMA's
// A operand (M dimension = 32 output channels of the wave) and the ACTIVATION as its B operand (N dimension = the sample's 32 rows):
//   * wave w owns output channels [32 w, 32 w + 32); lane (m = lane & 31, h = lane >> 5) owns row m;
//   * an accumulator register r of lane (m, h) is element [row m][channel 32 w + nl(r, h)], nl(r, h) = (r & 3) + 8 (r >> 2) + 4 h
//     (the C/D map of v_mfma_f32_32x32x*): 16 channels of its row, i.e. the residual stream x of a sample is ONE f32x16 per lane;
//   * the activation operand of the next product wants lane (m, h) to supply row m for a k that only has to match the weight operand's:
//     the window passes through the LDS tile once per product ([row][channel], 16-byte writes from the accumulator layout, 32-byte
//     reads of 8 consecutive k), and LayerNorm / the depthwise conv read the same tile by rows / by channel columns;
//   * per-head attention never leaves the wave: head = 16 of the wave's 32 channels, S^T = K Q^T and the P V product take their operands
//     straight from the accumulator registers (the lane's 16 channels / 16 keys ARE the contraction pairs (8 a + b, 8 a + 4 + b) of the two
//     half-waves), only V passes through the wave's own 32 columns of the tile.
// GEMMs run at fp32 grade on the bf16 matrix cores (common.hpp: 3-way split, six products) against the split packs the row-tile kernels
// read (PackJob type 6 / 7): the weight fragment of lane (n, h) is one 16-byte load per plane; the activation operand is split ONCE by the
// phase that produces it, into three bf16 planes in LDS (a per-wave split in the GEMM loop made a K = 128 product 7.3 k cycles for 1.5 k
// of matrix pipe: profiles/r06_notes.md), and the eight weight steps of the NEXT product stream into the fragment ring while the current one
// runs.  The 20 x 20 attention products use the fp32-input v_mfma_f32_32x32x2_f32.
// What round 6 measured about running beside the video chain: see profiles/r06_notes.md -- a conv-block workgroup of the video pass fills the
// register file of its CU (215 / 255 VGPRs x 2 waves per SIMD), so NOTHING co-resides with it, whatever its LDS footprint.
#include "common.hpp"
#include "launch.hpp"

namespace vsl {

__device__ long long g_stamps_q[40];
__device__ int g_dbg_on_q = 0;
#ifdef VSL_STAMPS       // phase stamps of workgroup 0: a separate build (vslnet_amd/build.py --stamps), see kernels_bwd.hip
#define QSTAMP(k) do { if (g_dbg_on_q && blockIdx.x == 0 && threadIdx.x == 0) g_stamps_q[k] = clock64(); } while (0)
#else
#define QSTAMP(k) do { } while (0)
#endif
static int qdbg_on() {
#ifndef VSL_STAMPS
    return 0;
#endif
    static int inited = 0, on = 0;
    if (!inited) { inited = 1; on = getenv("VSL_DEBUG_TIMING") != nullptr; if (on) { int one = 1; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_on_q), &one, sizeof one); } }
    return on;
}
static void qdbg_report(const char* name, int i0, int i1, hipStream_t s) {
    long long h[40];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps_q), sizeof h);
    fprintf(stderr, "[%s cycles]", name);
    for (int i = i0 + 1; i < i1; ++i) fprintf(stderr, " %lld", h[i] - h[i - 1]);
    fprintf(stderr, " | total %lld\n", h[i1 - 1] - h[i0]);
}

constexpr int QT = 256;                         // threads per workgroup: 4 waves, one per SIMD
constexpr int QROWS = 32;                       // rows of a sample window (L <= 32)

__device__ __forceinline__ int nl(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// accumulator layout -> LDS tile [row][channel] (all 32 rows: rows >= L carry finite don't-care values)
__device__ __forceinline__ void d2tile(const f32x16& x, float* __restrict__ T, int w, int m, int h) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
        *reinterpret_cast<float4*>(T + m * LDP + 32 * w + 8 * a + 4 * h) = make_float4(x[4 * a], x[4 * a + 1], x[4 * a + 2], x[4 * a + 3]);
}
// accumulator layout -> (rows, 128) row-major memory, rows < L ; g = row 0 of the sample
__device__ __forceinline__ void d2global(const f32x16& x, float* __restrict__ g, int w, int m, int h, int L) {
    if (m < L) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
            *reinterpret_cast<float4*>(g + (size_t)m * D + 32 * w + 8 * a + 4 * h) = make_float4(x[4 * a], x[4 * a + 1], x[4 * a + 2], x[4 * a + 3]);
    }
}
// (rows, 128) row-major memory -> accumulator layout (rows >= L: zero)
__device__ __forceinline__ void global2d(f32x16& x, const float* __restrict__ g, int w, int m, int h, int L) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float4 v = m < L ? *reinterpret_cast<const float4*>(g + (size_t)m * D + 32 * w + 8 * a + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
        x[4 * a] = v.x; x[4 * a + 1] = v.y; x[4 * a + 2] = v.z; x[4 * a + 3] = v.w;
    }
}
// a per-channel vector (bias, ...) in accumulator layout
__device__ __forceinline__ void vec2d(f32x16& x, const float* __restrict__ v, int w, int h) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float4 t = *reinterpret_cast<const float4*>(v + 32 * w + 8 * a + 4 * h);
        x[4 * a] = t.x; x[4 * a + 1] = t.y; x[4 * a + 2] = t.z; x[4 * a + 3] = t.w;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Operand planes: three bf16 planes (terms h, m, l) [32 rows][QLB], QPS elements between planes; 272-byte rows: conflict-free ds_read_b128.
// ---------------------------------------------------------------------------------------------------------
constexpr int QLB = D + 8;
constexpr int QPS = QROWS * QLB;
constexpr int QPLANES = 3 * QPS;                // bf16 elements of one operand buffer (26 112 B)
// T layout registers -> planes (lane (m, h): 4 consecutive channels per a)
__device__ __forceinline__ void d2planes(const f32x16& x, uint16_t* __restrict__ P, int w, int m, int h) {
#pragma unroll
    for (int a = 0; a < 4; ++a) split_store4(P, QLB, QPS, m, 32 * w + 8 * a + 4 * h, make_float4(x[4 * a], x[4 * a + 1], x[4 * a + 2], x[4 * a + 3]));
}
// COLUMN-phase producer (thread = channel c, two rows i0 / i0 + 1 of its segment): even lanes take row i0 of channels (c, c + 1), odd lanes
// row i0 + 1 of channels (c - 1, c) -- one quad-permute exchange per row pair
__device__ __forceinline__ void pair_split_store(uint16_t* __restrict__ P, int row0, int c, float v0, float v1) {
    const bool odd = c & 1;
    const float recv = lane_xor1(odd ? v0 : v1);
    uint32_t th, tm, tl;
    split3(odd ? recv : v0, odd ? v1 : recv, th, tm, tl);
    uint32_t* d = reinterpret_cast<uint32_t*>(P + (row0 + (odd ? 1 : 0)) * QLB + (c & ~1));
    d[0] = th; d[QPS / 2] = tm; d[QPS] = tl;
}

// ---------------------------------------------------------------------------------------------------------
// acc (32 channels x 32 rows, T layout) += sum over the ns <= 8 K = 16 steps held in the fragment ring:
//   weight operand : ring slot s = lane (n = lane & 31, h)'s fragment of step s (column col0 + n, k = 16 s + 8 h .. + 7 of each plane);
//   activation     : lane (m, h) reads planes[m][16 s + 8 h .. + 7] (three ds_read_b128, one step ahead).
// While step s runs, slot s is reloaded with step s of the NEXT product (`nx`), so a product's weights are in registers long before its
// operand planes are written.  Two alternating accumulators: consecutive MFMAs never wait for each other's result.
// ---------------------------------------------------------------------------------------------------------
struct AF3 { u32x4_t t[3]; };
struct WRing { Frag3 f[8]; };
// wl: this lane's address of step 0 ; ns = 0: nothing follows (wave-uniform) ; slim: last step that exists in the pack -- a product whose K is not a
// multiple of 128 still runs eight steps, the steps past the end re-read the last one against zero activations
struct WNext { const uint16_t* wl; size_t plane, sstep; int ns, slim; };
__device__ __forceinline__ WNext wnext(const uint16_t* W3, size_t plane, int ncols, int col0, int s0, int ns, int slim = 7) {
    const int lane = threadIdx.x & 63;
    return WNext{W3 + ((size_t)s0 * ncols + col0 + (lane & 31)) * 16 + 8 * (lane >> 5), plane, (size_t)ncols * 16, W3 ? ns : 0, slim};
}
__device__ __forceinline__ void wload(Frag3& f, const WNext& n, int s) {
    const uint16_t* p = n.wl + (size_t)min(s, n.slim) * n.sstep;
#pragma unroll
    for (int q = 0; q < 3; ++q) f.t[q] = *reinterpret_cast<const u32x4_t*>(p + q * n.plane);
}
__device__ __forceinline__ void wprefetch(WRing& r, const WNext& n) {
    static_for<0, 8>([&](auto sc) { constexpr int s = decltype(sc)::value; if (s < n.ns) wload(r.f[s], n, s); });
}
// NS = compile-time step count (8, or 1 for the tail chunk of a K that is not a multiple of 128)
template <int NS>
__device__ __forceinline__ void tgemm(const uint16_t* __restrict__ P, WRing& ring, f32x16& acc, const WNext& nx) {
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    const uint16_t* pr = P + m * QLB + 8 * h;
    // operand fragments two steps ahead (three buffers): the whole operand up front measured the same and cost 96 registers -- with them the
    // kernel no longer fits 256 VGPRs, i.e. no longer shares a CU with anything (profiles/r06_notes.md)
    AF3 b[3];
    auto bread = [&](int s, AF3& f) {
#pragma unroll
        for (int q = 0; q < 3; ++q) f.t[q] = *reinterpret_cast<const u32x4_t*>(pr + 16 * (s < NS ? s : NS - 1) + q * QPS);
    };
    bread(0, b[0]);
    if (NS > 1) bread(1, b[1]);
    // slots this product does not use: the next product's fragments can go there right away
    static_for<NS, 8>([&](auto sc) { constexpr int s = decltype(sc)::value; if (s < nx.ns) wload(ring.f[s], nx, s); });
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    static_for<0, NS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        constexpr int TW[6] = {1, 0, 2, 0, 1, 0}, TX[6] = {1, 2, 0, 1, 0, 0};      // (weight term, activation term): mm, hl, lh, hm, mh, hh
        if (s + 2 < NS) bread(s + 2, b[(s + 2) % 3]);
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            if (p & 1) acc2 = mfma_bf16(ring.f[s].t[TW[p]], b[s % 3].t[TX[p]], acc2);
            else acc = mfma_bf16(ring.f[s].t[TW[p]], b[s % 3].t[TX[p]], acc);
        }
        if (s < nx.ns) wload(ring.f[s], nx, s);
    });
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
}
// ---------------------------------------------------------------------------------------------------------
// ROW layout: 8 lanes per row (lane `sub` owns float4 columns 4 sub + 32 j), 32 rows per pass of the 256 threads.
// ---------------------------------------------------------------------------------------------------------
struct Row4 { float4 v[4]; };
__device__ __forceinline__ void row_vec(Row4& x, const float* __restrict__ v, int sub) {        // a per-channel vector (gamma, beta)
#pragma unroll
    for (int j = 0; j < 4; ++j) x.v[j] = *reinterpret_cast<const float4*>(v + sub * 4 + 32 * j);
}
__device__ __forceinline__ void row_load(Row4& x, const float* __restrict__ g, int rr, int sub, bool ok) {       // g = row 0 of the sample
#pragma unroll
    for (int j = 0; j < 4; ++j) x.v[j] = ok ? *reinterpret_cast<const float4*>(g + (size_t)rr * D + sub * 4 + 32 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void row_store(const Row4& x, float* __restrict__ g, int rr, int sub, bool ok) {
    if (ok) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(g + (size_t)rr * D + sub * 4 + 32 * j) = x.v[j];
    }
}
__device__ __forceinline__ void row_to_tile(const Row4& x, float* __restrict__ T, int rr, int sub) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(T + rr * LDP + sub * 4 + 32 * j) = x.v[j];
}
__device__ __forceinline__ void tile_to_row(Row4& x, const float* __restrict__ T, int rr, int sub) {
#pragma unroll
    for (int j = 0; j < 4; ++j) x.v[j] = *reinterpret_cast<const float4*>(T + rr * LDP + sub * 4 + 32 * j);
}
__device__ __forceinline__ void row_to_planes(const Row4& x, uint16_t* __restrict__ P, int rr, int sub) {
#pragma unroll
    for (int j = 0; j < 4; ++j) split_store4(P, QLB, QPS, rr, sub * 4 + 32 * j, x.v[j]);
}
// LayerNorm of the thread's row (+ dropout): x <- LN(x) * m for rows that exist, zeros otherwise (the conv's zero padding / a defined operand)
__device__ __forceinline__ void row_ln(Row4& x, const Row4& g, const Row4& b, const Drop& dp, int grow, int sub, bool ok) {
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += sum4(x.v[j]);
    const float mu = grp8_sum(sum) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        x.v[j].x -= mu; x.v[j].y -= mu; x.v[j].z -= mu; x.v[j].w -= mu;
        q += x.v[j].x * x.v[j].x + x.v[j].y * x.v[j].y + x.v[j].z * x.v[j].z + x.v[j].w * x.v[j].w;
    }
    const float rstd = rsqrtf(grp8_sum(q) * (1.0f / D) + LN_EPS);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float4 o;
        o.x = x.v[j].x * rstd * g.v[j].x + b.v[j].x; o.y = x.v[j].y * rstd * g.v[j].y + b.v[j].y;
        o.z = x.v[j].z * rstd * g.v[j].z + b.v[j].z; o.w = x.v[j].w * rstd * g.v[j].w + b.v[j].w;
        if (dp.thresh) {
            const uint32_t base = (uint32_t)(grow * D + sub * 4 + 32 * j);
            o.x *= drop_keep_scale(dp, base); o.y *= drop_keep_scale(dp, base + 1);
            o.z *= drop_keep_scale(dp, base + 2); o.w *= drop_keep_scale(dp, base + 3);
        }
        x.v[j] = ok ? o : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// =========================================================================================================
// k_query_fwd: Embedding.linear -> + positional rows -> 4 conv layers -> LN1 / q,k,v -> attention (8 heads, two per wave) -> output block.
// Saves exactly what the row-tile kernels save (qf, x0, y0..3, u0..3, ReLU bits, h1, q, k, v, LSE, att, r, h2, out): the backward and the
// weight-gradient launches do not care which forward produced them.
// =========================================================================================================
__global__ __launch_bounds__(QT) void k_query_fwd(QueryFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* T = smem;                            // [32][LDP] fp32 tile: residual stream by rows / by channel columns, V of the attention
    float* Mb = T + QROWS * LDP;                // key bias of the sample
    uint16_t* P0 = reinterpret_cast<uint16_t*>(Mb + QROWS);     // two operand-plane buffers, used alternately
    uint16_t* P1 = P0 + QPLANES;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, m = lane & 31, h = lane >> 5;
    const int rr = tid >> 3, sub = tid & 7;
    const int L = a.L, b = blockIdx.x, row0 = b * L;
    const size_t g0 = (size_t)row0 * D;
    const bool rok = rr < L;
    QSTAMP(0);
    if (tid < QROWS) Mb[tid] = tid < L ? (1.0f - a.mask[row0 + tid]) * MASK_VALUE : MASK_VALUE;
    const size_t plane_pw = pack3_plane(D, D), plane_qkv = pack3_plane(D, 3 * D);
    const Drop nodrop{0u, 0u, 1.f, 0u};
    WRing ring;

    // ---- Embedding.linear (:86-88): x = E W^T + b, E split into the operand planes in 128-column chunks (two buffers: one barrier per chunk)
    f32x16 X;
    {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int EW = a.EW, Kp = (EW + 15) & ~15, nch = (Kp + D - 1) / D;
        const size_t plane = pack3_plane(Kp, D);
        // every chunk runs eight K = 16 steps (columns past EW are zeros in the planes, steps past the pack's end re-read its last step)
        const int nst = Kp >> 4;
        auto chunk_w = [&](int ci) { return wnext(a.Wemb3, plane, D, 32 * w, 8 * ci, 8, nst - 1 - 8 * ci); };
        wprefetch(ring, chunk_w(0));
        f32x16 bv, pv;
        vec2d(bv, a.b_emb, w, h);
        global2d(pv, a.pos, w, m, h, L);        // positional rows 0 .. L - 1 (:202)
        auto erow = [&](Row4& e, int ci) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = D * ci + sub * 4 + 32 * j;
                e.v[j] = (rok && c < EW) ? *reinterpret_cast<const float4*>(a.E + (size_t)(row0 + rr) * EW + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        Row4 e, en;
        erow(e, 0);
#pragma unroll 1
        for (int ci = 0; ci < nch; ++ci) {
            erow(en, ci + 1);                   // next chunk's row: in flight during this chunk's product (past the end: zeros)
            uint16_t* P = (ci & 1) ? P1 : P0;
            row_to_planes(e, P, rr, sub);
            __syncthreads();
            const WNext nx = ci + 1 < nch ? chunk_w(ci + 1) : wnext(a.W3[0], plane_pw, D, 32 * w, 0, 8);
            tgemm<8>(P, ring, acc, nx);
            e = en;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += bv[r];
        if (a.qf) d2global(acc, a.qf + g0, w, m, h, L);
#pragma unroll
        for (int r = 0; r < 16; ++r) X[r] = acc[r] + pv[r];
    }
    QSTAMP(1);

    // ---- four conv layers (:133-139): x <- x + drop(relu(pointwise(depthwise7(LN(x)))))
    // (a real loop: the kernel executes every instruction ONCE per workgroup and one workgroup runs per CU, so unrolled it was 128 KB of
    // straight-line code against a 64 KB instruction cache -- instruction fetch, not the matrix pipe, set its time)
#pragma unroll 1
    for (int l = 0; l < 4; ++l) {
        // small operands of the layer: requested before the barriers they are needed behind
        Row4 gl, bl;
        row_vec(gl, a.ln_g[l], sub);
        row_vec(bl, a.ln_b[l], sub);
        const int c = tid & 127, os = 16 * (tid >> 7);
        float wk[DWK];
#pragma unroll
        for (int k = 0; k < DWK; ++k) wk[k] = a.dw_w[l][c * DWK + k];
        f32x16 bv;
        vec2d(bv, a.pw_b[l], w, h);
        d2tile(X, T, w, m, h);
        __syncthreads();
        if (l == 0) QSTAMP(10);
        {
            Row4 x;
            tile_to_row(x, T, rr, sub);
            row_store(x, (l == 0 ? a.x0 : a.y[max(l - 1, 0)]) + g0, rr, sub, rok);
            row_ln(x, gl, bl, nodrop, 0, sub, rok);
            row_to_tile(x, T, rr, sub);
        }
        __syncthreads();
        if (l == 0) QSTAMP(11);
        uint16_t* P = (l & 1) ? P1 : P0;
        {   // depthwise conv k = 7 along the sequence: thread = (channel, half of the rows); rows outside [0, L) are zeros in the tile
            float win[16 + 2 * HALO], uo[16];
#pragma unroll
            for (int i = 0; i < 16 + 2 * HALO; ++i) {
                const int r = os - HALO + i;
                win[i] = (r >= 0 && r < QROWS) ? T[r * LDP + c] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float u = 0.f;
#pragma unroll
                for (int k = 0; k < DWK; ++k) u += wk[k] * win[i + k];
                uo[i] = u;
            }
            if (l == 0) QSTAMP(12);
            float* ug = a.u[l] + g0 + c;
#pragma unroll
            for (int i = 0; i < 16; i += 2) pair_split_store(P, os + i, c, uo[i], uo[i + 1]);     // GEMM operand: three bf16 planes
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (os + i < L) ug[(size_t)(os + i) * D] = uo[i];       // saved: A operand of the weight gradient
        }
        __syncthreads();
        if (l == 0) QSTAMP(13);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        tgemm<8>(P, ring, acc, l < 3 ? wnext(a.W3[min(l + 1, 3)], plane_pw, D, 32 * w, 0, 8) : wnext(a.Wqkv3, plane_qkv, 3 * D, 32 * w, 0, 8));
        if (l == 0) QSTAMP(14);
        const Drop dp = a.dp[l];
        uint32_t bits[2] = {0u, 0u};            // ReLU decisions of this lane's channels inside the two 16-channel groups of the wave
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float z = acc[r] + bv[r];
            float av = fmaxf(z, 0.f);
            if (dp.thresh) av *= drop_keep_scale(dp, (uint32_t)((row0 + m) * D + 32 * w + nl(r, h)));
            X[r] += av;
            if (z > 0.f) bits[r >> 3] |= 1u << (nl(r, h) & 15);
        }
        {   // (R, 4) uint32 words seen as uint16: [row][16-channel group]; the two half-waves hold disjoint bits of both groups
            const unsigned u0 = bits[0], u1 = bits[1];
            auto r0 = __builtin_amdgcn_permlane32_swap(u0, u0, false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(u1, u1, false, false);
            const uint32_t g0b = r0[0] | r0[1], g1b = r1[0] | r1[1];
            uint16_t* mk = reinterpret_cast<uint16_t*>(a.relu_mask[l]);
            if (m < L) mk[(size_t)(row0 + m) * 8 + 2 * w + h] = (uint16_t)(h ? g1b : g0b);
        }
        if (l == 0) QSTAMP(15);
        QSTAMP(2 + l);
    }

    // ---- a8 first half (:168-173): h1 = drop(LN1(y3)) ; q, k, v = h1 W^T + b
    f32x16 Q, K, V;
    {
        Row4 gl, bl, x;
        row_vec(gl, a.ln1_g, sub);
        row_vec(bl, a.ln1_b, sub);
        f32x16 bq, bk, bvv;
        vec2d(bq, a.bq, w, h);
        vec2d(bk, a.bk, w, h);
        vec2d(bvv, a.bv, w, h);
        d2tile(X, T, w, m, h);
        __syncthreads();
        tile_to_row(x, T, rr, sub);
        row_store(x, a.y[3] + g0, rr, sub, rok);
        row_ln(x, gl, bl, a.d1, row0 + rr, sub, rok);
        if (a.h1) row_store(x, a.h1 + g0, rr, sub, rok);
        row_to_planes(x, P0, rr, sub);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) { Q[r] = 0.f; K[r] = 0.f; V[r] = 0.f; }
        tgemm<8>(P0, ring, Q, wnext(a.Wqkv3, plane_qkv, 3 * D, D + 32 * w, 0, 8));
        tgemm<8>(P0, ring, K, wnext(a.Wqkv3, plane_qkv, 3 * D, 2 * D + 32 * w, 0, 8));
        tgemm<8>(P0, ring, V, wnext(a.Wo3, plane_pw, D, 32 * w, 0, 8));
#pragma unroll
        for (int r = 0; r < 16; ++r) { Q[r] += bq[r]; K[r] += bk[r]; V[r] += bvv[r]; }
        d2global(Q, a.q + g0, w, m, h, L);
        d2global(K, a.k + g0, w, m, h, L);
        d2global(V, a.v + g0, w, m, h, L);
    }
    QSTAMP(6);
    d2tile(V, T, w, m, h);                      // V[key][channel]: each wave reads back its own 32 columns only (the tile's row-phase readers are behind the barrier above)
    __syncthreads();

    // ---- attention core (:174-182), heads 2 w and 2 w + 1: S^T = K Q^T (lane = query, registers = keys), softmax in the lane, O^T = V^T P^T
    f32x16 att;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
        for (int r = 8 * hh; r < 8 * hh + 8; ++r) S = __builtin_amdgcn_mfma_f32_32x32x2f32(K[r], Q[r], S, 0, 0, 0);
        float mx = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[r] = S[r] * 0.25f + Mb[nl(r, h)]; mx = fmaxf(mx, S[r]); }      // scaled AFTER QK^T (:175), keys masked (:176-178)
        mx = lane_pair32(mx, [](float p, float q) { return fmaxf(p, q); });
        float ls = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[r] = __expf(S[r] - mx); ls += S[r]; }
        ls = lane_pair32(ls, [](float p, float q) { return p + q; });
        const int head = 2 * w + hh;
        if (h == 0 && m < L) a.lse[((size_t)b * 8 + head) * L + m] = mx + __logf(ls);
        if (a.d2.thresh) {
            const uint32_t pbase = (uint32_t)(((size_t)(b + a.b_off) * 8 + head) * L + m) * (uint32_t)L;
#pragma unroll
            for (int r = 0; r < 16; ++r) S[r] *= drop_keep_scale(a.d2, pbase + nl(r, h));
        }
        f32x16 O;
#pragma unroll
        for (int r = 0; r < 16; ++r) O[r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) O = __builtin_amdgcn_mfma_f32_32x32x2f32(T[nl(r, h) * LDP + 32 * w + m], S[r], O, 0, 0, 0);
        const float inv = 1.0f / ls;
#pragma unroll
        for (int r = 8 * hh; r < 8 * hh + 8; ++r) att[r] = O[r] * inv;
    }
    d2global(att, a.att + g0, w, m, h, L);
    QSTAMP(7);
    // ---- output block (:183-190): r = drop(att) + x ; h2 = drop(LN2(r)) ; y = drop(h2 Wo^T + b) + r
    {
        Row4 gl, bl, x;
        row_vec(gl, a.ln2_g, sub);
        row_vec(bl, a.ln2_b, sub);
        f32x16 bo;
        vec2d(bo, a.bo, w, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) X[r] += att[r] * drop_mul(a.d3, (uint32_t)((row0 + m) * D + 32 * w + nl(r, h)));
        __syncthreads();                        // V's readers are done
        d2tile(X, T, w, m, h);
        __syncthreads();
        tile_to_row(x, T, rr, sub);
        row_store(x, a.r + g0, rr, sub, rok);
        row_ln(x, gl, bl, a.d4, row0 + rr, sub, rok);
        if (a.h2) row_store(x, a.h2 + g0, rr, sub, rok);
        row_to_planes(x, P1, rr, sub);
        __syncthreads();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        tgemm<8>(P1, ring, acc, wnext(a.Wo3, plane_pw, D, 32 * w, 0, 0));
#pragma unroll
        for (int r = 0; r < 16; ++r)
            X[r] += (acc[r] + bo[r]) * drop_mul(a.d5, (uint32_t)((row0 + m) * D + 32 * w + nl(r, h)));
    }
    d2global(X, a.out + g0, w, m, h, L);
    QSTAMP(8);
}

// =========================================================================================================
// k_query_bwd: the backward of k_query_fwd's encoder application + the Embedding linear's data gradient, one workgroup per sample.
//   dout (grad wrt the encoder output) -> output block backward (:183-190) -> attention backward (8 heads, two per wave, S / P / dP / dS
//   recomputed once per operand orientation) -> q,k,v backward + LN1^T -> conv layers 3..0 -> dx0 (= grad wrt Embedding.linear's output)
//   -> dE = dx0 W_emb.
// Row-tile launches replaced: attn_out_bwd + attn_bwd + convblock_bwd<0> (with its q,k,v prologue and linear tail).  It writes what
// they wrote: go, dq / dk / dv, gz[0..3] (G operands of the weight gradients), dx0, dE, and per-SAMPLE partial slabs of every
// LayerNorm gamma / beta and of the depthwise taps.
// Three thread layouts share the one LDS tile:  T layout (GEMM results / attention, see the file header);  ROW layout (8 lanes per row, lane
// `sub` owns float4 columns 4 sub + 32 j): LayerNorm statistics and their backward -- a thread keeps ITS row of dy / xhat in registers from
// phase to phase;  COLUMN layout (thread = channel c = tid >> 1, row half seg = tid & 1; the two halves of a channel in adjacent lanes):
// depthwise^T, the tap gradients and every per-channel sum over the sample's rows.
// =========================================================================================================
__device__ __forceinline__ void tile2d(f32x16& x, const float* __restrict__ T, int w, int m, int h) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float4 v = *reinterpret_cast<const float4*>(T + m * LDP + 32 * w + 8 * a + 4 * h);
        x[4 * a] = v.x; x[4 * a + 1] = v.y; x[4 * a + 2] = v.z; x[4 * a + 3] = v.w;
    }
}
// xhat / rstd of a row in ROW layout (rows that do not exist: zeros)
__device__ __forceinline__ void row_xhat(Row4& x, float& rstd, bool ok) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += sum4(x.v[j]);
    const float mu = grp8_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        x.v[j].x -= mu; x.v[j].y -= mu; x.v[j].z -= mu; x.v[j].w -= mu;
        q += x.v[j].x * x.v[j].x + x.v[j].y * x.v[j].y + x.v[j].z * x.v[j].z + x.v[j].w * x.v[j].w;
    }
    rstd = rsqrtf(grp8_sum(q) * (1.0f / D) + LN_EPS);
    const float k = ok ? rstd : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { x.v[j].x *= k; x.v[j].y *= k; x.v[j].z *= k; x.v[j].w *= k; }
}
// out = resid + rstd (gd - mean(gd) - xhat mean(gd xhat)), gd = dl * gamma  (ROW layout; rows that do not exist: zeros)
__device__ __forceinline__ void row_ln_bwd(Row4& out, const Row4& dl, const Row4& xh, float rstd, const float* __restrict__ g, const Row4& resid, int sub, bool ok) {
    Row4 gd;
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 gv = *reinterpret_cast<const float4*>(g + sub * 4 + 32 * j);
        gd.v[j] = make_float4(dl.v[j].x * gv.x, dl.v[j].y * gv.y, dl.v[j].z * gv.z, dl.v[j].w * gv.w);
        m1 += sum4(gd.v[j]);
        m2 += gd.v[j].x * xh.v[j].x + gd.v[j].y * xh.v[j].y + gd.v[j].z * xh.v[j].z + gd.v[j].w * xh.v[j].w;
    }
    m1 = grp8_sum(m1) * (1.0f / D);
    m2 = grp8_sum(m2) * (1.0f / D);
    const float k = ok ? rstd : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        out.v[j].x = resid.v[j].x + k * (gd.v[j].x - m1 - xh.v[j].x * m2); out.v[j].y = resid.v[j].y + k * (gd.v[j].y - m1 - xh.v[j].y * m2);
        out.v[j].z = resid.v[j].z + k * (gd.v[j].z - m1 - xh.v[j].z * m2); out.v[j].w = resid.v[j].w + k * (gd.v[j].w - m1 - xh.v[j].w * m2);
    }
}

__global__ __launch_bounds__(QT) void k_query_bwd(QueryBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* T = smem;                            // [32][LDP]
    float* Mb = T + QROWS * LDP;                // [32] key bias
    float* Ls = Mb + QROWS;                     // [8 heads][32] LSE per query
    float* Dqs = Ls + 8 * QROWS;                // [8 heads][32] D = dA . O per query
    uint16_t* P0 = reinterpret_cast<uint16_t*>(Dqs + 8 * QROWS);     // two operand-plane buffers, used alternately
    uint16_t* P1 = P0 + QPLANES;
    uint16_t* P2 = P1 + QPLANES;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, m = lane & 31, h = lane >> 5;
    const int rr = tid >> 3, sub = tid & 7;     // ROW layout
    const int cc = tid >> 1, seg = tid & 1;     // COLUMN layout: rows [16 seg, 16 seg + 16) of channel cc
    const int L = a.L, b = blockIdx.x, row0 = b * L;
    const size_t g0 = (size_t)row0 * D;
    const bool rok = rr < L, mok = m < L;
    QSTAMP(20);
    if (tid < QROWS) Mb[tid] = tid < L ? (1.0f - a.mask[row0 + tid]) * MASK_VALUE : MASK_VALUE;
    Ls[tid] = (tid & 31) < L ? a.lse[((size_t)b * 8 + (tid >> 5)) * L + (tid & 31)] : 0.f;
    const size_t plane_pw = pack3_plane(D, D), plane_t = pack3_plane(3 * D, D), plane_e = pack3_plane(D, a.EWc);
    WRing ring;
    wprefetch(ring, wnext(a.WoT3, plane_pw, D, 32 * w, 0, 8));

    // one LayerNorm backward whose incoming gradient is in T layout (`dln`, rows >= L zero) and whose input row is in `x` (ROW layout):
    // returns resid + LN^T(dln) in ROW layout, writes the per-sample gamma / beta slabs.  Leaves the tile with readers: barrier before the next write.
    auto ln_bwd_t = [&](Row4& x, const f32x16& dln, const float* __restrict__ g, const Row4& resid, float* __restrict__ p_g, float* __restrict__ p_b, Row4& out) {
        float rstd;
        row_xhat(x, rstd, rok);
        __syncthreads();                        // the tile is free
        row_to_tile(x, T, rr, sub);
        __syncthreads();
        float xc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) xc[i] = T[(16 * seg + i) * LDP + cc];
        __syncthreads();
        d2tile(dln, T, w, m, h);
        __syncthreads();
        float sb = 0.f, sg = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float d = T[(16 * seg + i) * LDP + cc]; sb += d; sg += d * xc[i]; }
        sb += lane_xor1(sb); sg += lane_xor1(sg);
        if (seg == 0) { p_g[(size_t)b * D + cc] = sg; p_b[(size_t)b * D + cc] = sb; }
        Row4 dl;
        tile_to_row(dl, T, rr, sub);
        row_ln_bwd(out, dl, x, rstd, g, resid, sub, rok);
    };

    // ---- output block backward (:183-190): go = dout * m5 ; dh2 = go Wo ; dr = dout + LN2^T(dh2 * m4)
    Row4 DY, DR;
    {
        Row4 xr, go;
        row_load(xr, a.r + g0, rr, sub, rok);
        row_load(DY, a.dout + g0, rr, sub, rok);         // (k_cq_bwd_d's work hosted here instead of a launch of its own: measured, 2.2 x slower inside this kernel)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t base = (uint32_t)((row0 + rr) * D + sub * 4 + 32 * j);
            go.v[j] = make_float4(DY.v[j].x * drop_mul(a.d5, base), DY.v[j].y * drop_mul(a.d5, base + 1), DY.v[j].z * drop_mul(a.d5, base + 2),
                                  DY.v[j].w * drop_mul(a.d5, base + 3));
        }
        row_store(go, a.go + g0, rr, sub, rok);         // G operand of the out_layer weight gradient
        row_to_planes(go, P0, rr, sub);
        __syncthreads();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        tgemm<8>(P0, ring, acc, wnext(a.WqkvT3, plane_t, D, 32 * w, 0, 8));
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = mok ? acc[r] * drop_mul(a.d4, (uint32_t)((row0 + m) * D + 32 * w + nl(r, h))) : 0.f;
        ln_bwd_t(xr, acc, a.ln2_g, DY, a.p_ln2g, a.p_ln2b, DR);
    }
    QSTAMP(21);

    // ---- attention backward (:174-182), heads 2 w and 2 w + 1
    f32x16 dQd, dKd, dVd;
    {
        f32x16 Q, K, V, dA, O;
        global2d(Q, a.q + g0, w, m, h, L);
        global2d(K, a.k + g0, w, m, h, L);
        global2d(V, a.v + g0, w, m, h, L);
        global2d(O, a.att + g0, w, m, h, L);
        __syncthreads();                        // ln_bwd_t's readers are done
        row_to_tile(DR, T, rr, sub);
        __syncthreads();
        tile2d(dA, T, w, m, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) dA[r] = mok ? dA[r] * drop_mul(a.d3, (uint32_t)((row0 + m) * D + 32 * w + nl(r, h))) : 0.f;      // r = drop3(att) + x (:183-184)
        float DqA[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            float s = 0.f;
#pragma unroll
            for (int r = 8 * hh; r < 8 * hh + 8; ++r) s += dA[r] * O[r];
            s = lane_pair32(s, [](float p, float q) { return p + q; });
            DqA[hh] = s;
            if (h == 0) Dqs[(2 * w + hh) * QROWS + m] = s;
        }
        // orientation A (lane = query, registers = keys): dS feeds dQ
        f32x16 dSA[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int head = 2 * w + hh;
            f32x16 S, dP;
#pragma unroll
            for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
#pragma unroll
            for (int r = 8 * hh; r < 8 * hh + 8; ++r) {
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(K[r], Q[r], S, 0, 0, 0);
                dP = __builtin_amdgcn_mfma_f32_32x32x2f32(V[r], dA[r], dP, 0, 0, 0);
            }
            const float lq = Ls[head * QROWS + m];
            const uint32_t pbase = (uint32_t)(((size_t)(b + a.b_off) * 8 + head) * L + m) * (uint32_t)L;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __expf(S[r] * 0.25f + Mb[nl(r, h)] - lq);
                const float m2 = drop_mul(a.d2, pbase + nl(r, h));
                dSA[hh][r] = p * (dP[r] * m2 - DqA[hh]) * 0.25f;
            }
        }
        __syncthreads();                        // dA's readers are done (and Dqs is written)
        d2tile(K, T, w, m, h);
        __syncthreads();
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(T[nl(r, h) * LDP + 32 * w + m], dSA[hh][r], acc, 0, 0, 0);
#pragma unroll
            for (int r = 8 * hh; r < 8 * hh + 8; ++r) dQd[r] = mok ? acc[r] : 0.f;
        }
        // orientation B (lane = key, registers = queries): dS feeds dK, P * m2 feeds dV
        f32x16 dSB[2], PdB[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int head = 2 * w + hh;
            f32x16 S, dP;
#pragma unroll
            for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
#pragma unroll
            for (int r = 8 * hh; r < 8 * hh + 8; ++r) {
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(Q[r], K[r], S, 0, 0, 0);
                dP = __builtin_amdgcn_mfma_f32_32x32x2f32(dA[r], V[r], dP, 0, 0, 0);
            }
            const float mbk = Mb[m];
            const uint32_t hb = (uint32_t)(((size_t)(b + a.b_off) * 8 + head) * L);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qq = nl(r, h);
                const float p = __expf(S[r] * 0.25f + mbk - Ls[head * QROWS + qq]);
                const float m2 = drop_mul(a.d2, (hb + (uint32_t)qq) * (uint32_t)L + (uint32_t)m);
                PdB[hh][r] = p * m2;
                dSB[hh][r] = p * (dP[r] * m2 - Dqs[head * QROWS + qq]) * 0.25f;
            }
        }
        __syncthreads();
        d2tile(Q, T, w, m, h);
        __syncthreads();
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(T[nl(r, h) * LDP + 32 * w + m], dSB[hh][r], acc, 0, 0, 0);
#pragma unroll
            for (int r = 8 * hh; r < 8 * hh + 8; ++r) dKd[r] = mok ? acc[r] : 0.f;
        }
        __syncthreads();
        d2tile(dA, T, w, m, h);
        __syncthreads();
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(T[nl(r, h) * LDP + 32 * w + m], PdB[hh][r], acc, 0, 0, 0);
#pragma unroll
            for (int r = 8 * hh; r < 8 * hh + 8; ++r) dVd[r] = mok ? acc[r] : 0.f;
        }
        d2global(dQd, a.dq + g0, w, m, h, L);   // G operands of the q,k,v weight gradient
        d2global(dKd, a.dk + g0, w, m, h, L);
        d2global(dVd, a.dv + g0, w, m, h, L);
    }
    QSTAMP(22);

    // ---- a8 first half backward (:168-173): dh1 = [dQ | dK | dV] [Wq; Wk; Wv] ; dy3 = dr + LN1^T(dh1 * m1)
    {
        Row4 x3;
        row_load(x3, a.y3 + g0, rr, sub, rok);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        d2planes(dQd, P0, w, m, h);             // three K = 128 chunks in three plane buffers: one barrier, one product body
        d2planes(dKd, P1, w, m, h);
        d2planes(dVd, P2, w, m, h);
        __syncthreads();
#pragma unroll 1
        for (int t = 0; t < 3; ++t)
            tgemm<8>(P0 + t * QPLANES, ring, acc, t < 2 ? wnext(a.WqkvT3, plane_t, D, 32 * w, 8 * (t + 1), 8) : wnext(a.WT3[3], plane_pw, D, 32 * w, 0, 8));
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = mok ? acc[r] * drop_mul(a.d1, (uint32_t)((row0 + m) * D + 32 * w + nl(r, h))) : 0.f;
        ln_bwd_t(x3, acc, a.ln1_g, DR, a.p_ln1g, a.p_ln1b, DY);
    }
    QSTAMP(23);

    // ---- conv layers 3 .. 0 (autograd of :133-139)
#pragma unroll 1
    for (int l = 3; l >= 0; --l) {
        Row4 xh;
        float rstd;
        row_load(xh, a.x[l] + g0, rr, sub, rok);
        const uint4 mw = rok ? *reinterpret_cast<const uint4*>(a.relu_mask[l] + (size_t)(row0 + rr) * 4) : make_uint4(0u, 0u, 0u, 0u);
        float wk[DWK];
#pragma unroll
        for (int k = 0; k < DWK; ++k) wk[k] = a.dw_w[l][cc * DWK + k];
        const float gc = a.ln_g[l][cc], bc = a.ln_b[l][cc];
        row_xhat(xh, rstd, rok);
        __syncthreads();                        // the tile is free
        row_to_tile(xh, T, rr, sub);
        __syncthreads();
        float xw[16 + 2 * HALO];                // xhat of this channel, rows 16 seg - 3 .. 16 seg + 18 (zero outside the sample)
#pragma unroll
        for (int i = 0; i < 16 + 2 * HALO; ++i) {
            const int r = 16 * seg - HALO + i;
            xw[i] = (r >= 0 && r < QROWS) ? T[r * LDP + cc] : 0.f;
        }
        {   // dz = dy * relu bit * dropout: G operand of the pointwise weight gradient, B operand of du = dz Wp
            const Drop dp = a.dp[l];
            const uint32_t mwv[4] = {mw.x, mw.y, mw.z, mw.w};
            Row4 dz;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t bits = mwv[j] >> (sub * 4);
                float mm[4] = {1.f, 1.f, 1.f, 1.f};
                if (dp.thresh) {
                    const uint32_t base = (uint32_t)((row0 + rr) * D + sub * 4 + 32 * j);
#pragma unroll
                    for (int i = 0; i < 4; ++i) mm[i] = drop_keep_scale(dp, base + i);
                }
                dz.v[j].x = (bits & 1u) ? DY.v[j].x * mm[0] : 0.f;
                dz.v[j].y = (bits & 2u) ? DY.v[j].y * mm[1] : 0.f;
                dz.v[j].z = (bits & 4u) ? DY.v[j].z * mm[2] : 0.f;
                dz.v[j].w = (bits & 8u) ? DY.v[j].w * mm[3] : 0.f;
            }
            row_store(dz, a.gz[l] + g0, rr, sub, rok);
            row_to_planes(dz, (l & 1) ? P1 : P0, rr, sub);
        }
        __syncthreads();
        f32x16 du;
#pragma unroll
        for (int r = 0; r < 16; ++r) du[r] = 0.f;
        tgemm<8>((l & 1) ? P1 : P0, ring, du,
              l > 0 ? wnext(a.WT3[max(l - 1, 0)], plane_pw, D, 32 * w, 0, 8) : wnext(a.WembT3, plane_e, a.EWc, 32 * w, 0, 32 * w < a.EW ? 8 : 0));
        if (!mok) {
#pragma unroll
            for (int r = 0; r < 16; ++r) du[r] = 0.f;       // rows outside the sample: the conv's zero padding
        }
        d2tile(du, T, w, m, h);                 // (the tile's last readers -- the xhat windows -- are behind the barrier above)
        __syncthreads();
        {   // dv = depthwise^T(du) ; tap / gamma / beta partial sums over the sample's rows
            float dw_[16 + 2 * HALO], dvo[16], gw[DWK], slb = 0.f, slg = 0.f;
#pragma unroll
            for (int i = 0; i < 16 + 2 * HALO; ++i) {
                const int r = 16 * seg - HALO + i;
                dw_[i] = (r >= 0 && r < QROWS) ? T[r * LDP + cc] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < DWK; ++k) gw[k] = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float dv = 0.f;
#pragma unroll
                for (int k = 0; k < DWK; ++k) {
                    dv += wk[k] * dw_[i + 2 * HALO - k];
                    const int rv = 16 * seg + i + k - HALO;                     // row of v this tap reads
                    const float vv = (rv >= 0 && rv < L) ? xw[i + k] * gc + bc : 0.f;
                    gw[k] += dw_[i + HALO] * vv;
                }
                dvo[i] = dv;
                slb += (16 * seg + i < L) ? dv : 0.f;
                slg += dv * xw[i + HALO];
            }
#pragma unroll
            for (int k = 0; k < DWK; ++k) gw[k] += lane_xor1(gw[k]);
            slb += lane_xor1(slb); slg += lane_xor1(slg);
            if (seg == 0) {
#pragma unroll
                for (int k = 0; k < DWK; ++k) a.p_dw[l][((size_t)b * D + cc) * DWK + k] = gw[k];
                a.p_lnb[l][(size_t)b * D + cc] = slb;
                a.p_lng[l][(size_t)b * D + cc] = slg;
            }
            __syncthreads();                    // every du window is in registers
#pragma unroll
            for (int i = 0; i < 16; ++i) T[(16 * seg + i) * LDP + cc] = dvo[i];
        }
        __syncthreads();
        {   // dy <- dy + LN^T(dv)
            Row4 dvr, out;
            tile_to_row(dvr, T, rr, sub);
            row_ln_bwd(out, dvr, xh, rstd, a.ln_g[l], DY, sub, rok);
            DY = out;
        }
        QSTAMP(24 + (3 - l));
    }
    row_store(DY, a.dx0 + g0, rr, sub, rok);            // grad wrt Embedding.linear's output (the positional table's partial slabs are these rows)

    // ---- Embedding.linear, data gradient (:81-87 backward): dE = dx0 W
    row_to_planes(DY, P1, rr, sub);             // (layer 0 read P0)
    __syncthreads();
    {
#pragma unroll 1
        for (int blk = w; 32 * blk < a.EW; blk += 4) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            tgemm<8>(P1, ring, acc, wnext(a.WembT3, plane_e, a.EWc, 32 * (blk + 4), 0, 32 * (blk + 4) < a.EW ? 8 : 0));
            if (mok) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = 32 * blk + 8 * q + 4 * h;
                    if (col < a.EW) *reinterpret_cast<float4*>(a.dE + (size_t)(row0 + m) * a.EW + col) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                }
            }
        }
    }
    QSTAMP(28);
}

bool query_fused_ok(int L, int H, int EW) { (void)EW; return L <= QROWS && H == 8; }
size_t query_fwd_lds() { return (size_t)(QROWS * LDP + QROWS) * sizeof(float) + 2 * QPLANES * sizeof(uint16_t); }
size_t query_bwd_lds() { return (size_t)(QROWS * LDP + QROWS + 16 * QROWS) * sizeof(float) + 3 * QPLANES * sizeof(uint16_t); }
void launch_query_bwd(const QueryBwdArgs& a, int B, hipStream_t s) {
    static size_t ok = 0;
    ensure_dynamic_lds((const void*)k_query_bwd, query_bwd_lds(), ok, "k_query_bwd");
    VSL_LAUNCH(k_query_bwd, dim3(B), dim3(QT), query_bwd_lds(), s, a);
    static int left = 2;
    if (qdbg_on() && B > 16 && left > 0) {
        --left;
        qdbg_report("query_bwd: out block | attention | qkv + LN1 | L3 | L2 | L1 | L0 | linear", 20, 29, s);
    }
}
void launch_query_fwd(const QueryFwdArgs& a, int B, hipStream_t s) {
    static size_t ok = 0;
    ensure_dynamic_lds((const void*)k_query_fwd, query_fwd_lds(), ok, "k_query_fwd");
    VSL_LAUNCH(k_query_fwd, dim3(B), dim3(QT), query_fwd_lds(), s, a);
    static int left = 2;
    if (qdbg_on() && B > 16 && left > 0) {
        --left;
        qdbg_report("query_fwd: linear | L0 | L1 | L2 | L3 | LN1+qkv | attention | out", 0, 9, s);
        qdbg_report("  query_fwd L0 from d2tile: LN | dw | dw write | gemm | epilogue", 10, 16, s);
    }
}

}  // namespace vsl
