// Weight gradients of every Conv1D on the path (gfx950):  dW[n][k] = sum_r G[r][n] A[r][k]  (+ bias = column sums of G), over 256-row chunks
// -> partial slabs that k_reduce adds.  The contraction runs over ROWS and both operands are row-major, so a 16-row step of G and of A is exactly
// one K slice of v_mfma_f32_32x32x16_bf16; every fp32 operand is split exactly into three bf16 terms and six products are accumulated in fp32
// (common.hpp).  k_wgrad4 serves every fp32 job, k_wgrad3 the bfloat16-feature job of the bf16 throughput mode.  (The fp32-input MFMA kernel of
// rounds 1-2 is gone; A/B baselines come from git revisions, tools/build_base.py.  A row-major-plane variant fed through ds_read_b64_tr_b16 with
// an in-kernel slab fold was built and measured in round 4 and not adopted: tools/ubench/wgrad5_kernel.inc, profiles/r04_notes.md.)
#include "common.hpp"
#include "launch.hpp"
#include <type_traits>

namespace vsl {

constexpr int WG2_T = 256;
// =====================================================================================================================
// k_wgrad3: the same product on the BF16 matrix cores at fp32 grade (round 3).
//
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the fp32-input MFMA and, unlike it, overlaps with vector-ALU work
// (tools/ubench/split_bf16.hip: an MFMA wave beside a VALU wave 64 | 85 -> 64 | 95 cycles; the fp32 MFMA was additive).  Every fp32
// operand x is split EXACTLY into three bfloat16 terms x = h + m + l (round-to-nearest at each level: |m| <= 2^-8 |x|, |l| <= 2^-16 |x|,
// l is exact because at most 8 significant bits remain) and the six products of weight >= 2^-16 -- hh, hm, mh, hl, lh, mm -- are
// accumulated in fp32 by the MFMA; the dropped ml, lm, ll are <= 2^-23 |g||a|: the class of ONE fp32 rounding of the product, which the
// fp32 chain commits as well.  Measured against an fp64 reference (same ubench, R = 8192): error / max sum|g||a| = 1.6e-8 against 2.2e-8
// for the fp32 MFMA chain -- the split path is no less accurate than the kernel it replaces.
//
// A step = 16 rows = one K slice of the MFMA: lane (i, h) holds rows 8 h .. 8 h + 7 of its 2 G columns and 2 A columns (the float2 loads
// of the LDS-free layout: lane = its two columns); two rows of a column pack into one operand dword.  Per step and wave: 16 pairs x 11
// vector instructions (v_cvt_pk_bf16_f32, shift / and, 2 subtractions per level) against 24 MFMAs (768 matrix cycles): the split of step
// s + 1 is woven between the MFMAs of step s by hand (sched_barrier per MFMA), the raw rows of step s + 3 are requested as soon as a row
// pair has been split.  No LDS, no barrier.  Built with -fno-slp-vectorize: packed fp32 adds (v_pk_add_f32 + v_mov packing) are slower.
// =====================================================================================================================
constexpr int WG3_STEP = 16;                       // rows per step
// Raw buffer descriptor over [p, p + bytes): loads past the end return 0 -- rows past R need neither a clamp nor a mask.  The range check
// covers the VGPR + immediate offset only (not the SGPR offset), so a step's descriptor is rebuilt from its own base (scalar ALU).
struct BufRange { uint32_t lo, hi, bytes; };           // wave-uniform
__device__ __forceinline__ BufRange make_range(const void* p, uint32_t bytes) {
    const uint64_t u = reinterpret_cast<uint64_t>(p);      // uniform by construction; readfirstlane makes that provable (no waterfall loop)
    return BufRange{(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)u), (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(u >> 32)),
                    (uint32_t)__builtin_amdgcn_readfirstlane(bytes)};
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_at(const BufRange& b, uint32_t off) {      // descriptor of [p + off, p + bytes), off < 2^31
    const uint32_t o = __builtin_amdgcn_readfirstlane(off);
    const uint32_t lo = b.lo + o, hi = b.hi + (lo < o ? 1u : 0u);
    const int n = max((int)b.bytes - (int)o, 0);
    const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(hi) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(lo);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(base), 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
}
template <bool DROP, bool ABF16>
__global__ __launch_bounds__(WG2_T, 1) void k_wgrad3(WgradBatch wb) {
    int ji = 0;
    while (ji + 1 < wb.n && (int)blockIdx.x >= wb.start[ji + 1]) ++ji;
    const WgradJob& j = wb.j[ji];
    const int K = j.K, R = j.R;
    const int crows = wgrad_rows(j);
    const int nkt = (K + 127) >> 7, nch = wgrad_chunks(R, crows);
    const int local = blockIdx.x - wb.start[ji];
    const int kt = local % nkt, ch = (local / nkt) % nch, gb = local / (nkt * nch);
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int nh = wv & 1, kh = wv >> 1;                    // this wave's 64 x 64 quadrant of the 128 x 128 block
    const int ldg = j.ldg ? j.ldg : D;
    const bool blocks = j.nA > 0;
    const int lda = blocks ? D : K;
    const int kloc = 64 * kh + 2 * i;                       // column inside the 128-wide k tile
    const int kglob = kt * 128 + kloc;                      // column of dW
    const bool kin = kglob < K;                             // K is even: both columns of the pair are in or out together
    const int rbeg = ch * crows, rend = min(R, rbeg + crows), nrows = rend - rbeg;
    const uint32_t dseed = j.dp.seed, dthr = j.dp.thresh, dkey = j.dp.key;
    const float dscale = j.dp.scale;
    constexpr int ASZ = ABF16 ? 2 : 4;                      // bytes per A element
    // Buffer descriptors over the chunk's rows [rbeg, rend) of the two operands: address = descriptor base + per-lane offset of the lane's row q
    // (8 + 8 loop-invariant registers) + the step's offset in an SGPR -- no address arithmetic in the loop; rows >= R read as zeros.
    // (the ranges end with the CHUNK: the run-ahead loads of the steps past it read zeros too, so the bias sums need no guard)
    const BufRange grg = make_range(j.G[gb] + (size_t)rbeg * ldg, (uint32_t)nrows * (uint32_t)ldg * 4u);
    const BufRange arg = make_range(reinterpret_cast<const char*>(blocks ? j.A[kt] : j.Afull) + (size_t)rbeg * lda * ASZ,
                                    (uint32_t)nrows * (uint32_t)lda * (uint32_t)ASZ);
    uint32_t goff[8], aoff[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        goff[q] = (uint32_t)((8 * h + q) * ldg + 64 * nh + 2 * i) * 4u;
        aoff[q] = (uint32_t)((8 * h + q) * lda + (blocks ? kloc : (kin ? kglob : 0))) * (uint32_t)ASZ;
    }
    const uint32_t gstep = (uint32_t)(WG3_STEP * ldg) * 4u, astep = (uint32_t)(WG3_STEP * lda * ASZ);

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bs[2] = {0.f, 0.f};
    struct Raw { float2 g[8]; float2 a[8]; };             // this lane's 8 rows of a step (ABF16: a[q].x carries the raw bf16 pair)
    struct Ops { u32x4_t g[3][2], a[3][2]; };             // [term h / m / l][column], 4 dwords = 8 bf16 = the lane's 8 rows
    constexpr int NM = 4 * (ABF16 ? 3 : 6), NPAIR = 16;   // bf16 features are their own (exact) h term: 3 products
    const int ns = ((nrows + 2 * WG3_STEP - 1) / (2 * WG3_STEP)) * 2;       // steps, even (a step past the rows multiplies zeros)

    auto run = [&](auto bias_c) {
        constexpr bool BIAS = decltype(bias_c)::value;
        auto ld_rows = [&](int s, int p, Raw& x) {        // row pair p (rows 2p, 2p+1 of the lane) of step s, both operands
            const __amdgpu_buffer_rsrc_t grs = rsrc_at(grg, (uint32_t)s * gstep), ars = rsrc_at(arg, (uint32_t)s * astep);
#pragma unroll
            for (int q = 2 * p; q < 2 * p + 2; ++q) {
                const u32x2_t gv = __builtin_amdgcn_raw_buffer_load_b64(grs, goff[q], 0, 0);
                x.g[q] = make_float2(__uint_as_float(gv[0]), __uint_as_float(gv[1]));
                if (ABF16) x.a[q].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ars, aoff[q], 0, 0));
                else {
                    const u32x2_t av = __builtin_amdgcn_raw_buffer_load_b64(ars, aoff[q], 0, 0);
                    x.a[q] = make_float2(__uint_as_float(av[0]), __uint_as_float(av[1]));
                }
            }
        };
        // pair pi of step s: pi = 4 * rowpair + column (columns 0, 1 = G ; 2, 3 = A)
        auto split_pair = [&](int s, int pi, const Raw& x, Ops& o) {
            const int jp = pi >> 2, c = pi & 3;
            uint32_t hh, mm, ll;
            if (c < 2) {
                const float g0 = c ? x.g[2 * jp].y : x.g[2 * jp].x, g1 = c ? x.g[2 * jp + 1].y : x.g[2 * jp + 1].x;
                if (BIAS) bs[c] += g0 + g1;
                split3(g0, g1, hh, mm, ll);
                o.g[0][c][jp] = hh; o.g[1][c][jp] = mm; o.g[2][c][jp] = ll;
            } else if (ABF16) {
                const int ca = c - 2;
                const uint32_t u0 = __float_as_uint(x.a[2 * jp].x), u1 = __float_as_uint(x.a[2 * jp + 1].x);
                uint32_t v = ca ? __builtin_amdgcn_perm(u1, u0, 0x07060302u) : __builtin_amdgcn_perm(u1, u0, 0x05040100u);   // column ca of rows 2jp, 2jp+1
                if (DROP) {       // exact zeroing of the bf16 inputs; the 1/(1-p) scale is applied to the fp32 sums at the end
                    const uint32_t base = (uint32_t)(rbeg + WG3_STEP * s + 8 * h + 2 * jp) * (uint32_t)K + (uint32_t)(kglob + ca);
                    if (drop_hash(base, dseed, dkey) < dthr) v &= 0xFFFF0000u;
                    if (drop_hash(base + (uint32_t)K, dseed, dkey) < dthr) v &= 0x0000FFFFu;
                }
                o.a[0][ca][jp] = v;
            } else {
                const int ca = c - 2;
                float a0 = ca ? x.a[2 * jp].y : x.a[2 * jp].x, a1 = ca ? x.a[2 * jp + 1].y : x.a[2 * jp + 1].x;
                if (DROP) {
                    const uint32_t base = (uint32_t)(rbeg + WG3_STEP * s + 8 * h + 2 * jp) * (uint32_t)K + (uint32_t)(kglob + ca);
                    a0 *= drop_hash(base, dseed, dkey) >= dthr ? dscale : 0.f;
                    a1 *= drop_hash(base + (uint32_t)K, dseed, dkey) >= dthr ? dscale : 0.f;
                }
                split3(a0, a1, hh, mm, ll);
                o.a[0][ca][jp] = hh; o.a[1][ca][jp] = mm; o.a[2][ca][jp] = ll;
            }
        };
        // MFMA m of a step: product type outer (small terms first), block inner -> consecutive MFMAs hit different accumulators
        auto mma1 = [&](int m, const Ops& o) {
            constexpr int TG6[6] = {1, 0, 2, 0, 1, 0}, TA6[6] = {1, 2, 0, 1, 0, 0};      // (g term, a term): mm, hl, lh, hm, mh, hh
            constexpr int TG3[3] = {2, 1, 0};                                             // bf16 A: l a, m a, h a
            const int t = m >> 2, a = (m >> 1) & 1, b = m & 1;
            if (ABF16) acc[a][b] = mfma_bf16(o.g[TG3[t]][a], o.a[0][b], acc[a][b]);
            else acc[a][b] = mfma_bf16(o.g[TG6[t]][a], o.a[TA6[t]][b], acc[a][b]);
        };
        Raw x0, x1;
        Ops o0, o1;
        auto step = [&](int s, const Ops& cur, Ops& nxt, Raw& xs) {   // MFMAs of step s ; split of step s + 1 (raw rows in xs) ; loads of step s + 3
            static_for<0, NM>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                mma1(m, cur);
                constexpr int p0 = (m * NPAIR + NM - 1) / NM, p1 = ((m + 1) * NPAIR + NM - 1) / NM;       // pairs pi with pi * NM / NPAIR == m
                static_for<p0, p1>([&](auto pc) {
                    constexpr int pi = decltype(pc)::value;
                    split_pair(s + 1, pi, xs, nxt);
                    if constexpr ((pi & 3) == 3) ld_rows(s + 3, pi >> 2, xs);      // the row pair is dead: reload it
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        static_for<0, 4>([&](auto pc) { ld_rows(0, decltype(pc)::value, x0); });
        static_for<0, 4>([&](auto pc) { ld_rows(1, decltype(pc)::value, x1); });
        static_for<0, NPAIR>([&](auto pc) { split_pair(0, decltype(pc)::value, x0, o0); });
        static_for<0, 4>([&](auto pc) { ld_rows(2, decltype(pc)::value, x0); });
        __builtin_amdgcn_sched_barrier(0);
        for (int s = 0; s < ns; s += 2) {
            step(s, o0, o1, x1);
            step(s + 1, o1, o0, x0);
        }
    };
    const bool want_bias = kt == 0 && kh == 0 && j.out_bias[gb] != nullptr;      // wave-uniform
    if (want_bias) run(std::true_type()); else run(std::false_type());
    // ---- partial slab: lane, register r of acc[a][b] = dW[n = 64 nh + 2 * acc_row(r) + a][k = kglob_of(lane & 31) + b]
    const int N = 128 * j.nG;
    float* out = j.out + ((size_t)ch * N + gb * 128 + 64 * nh) * K + kglob;
    const float osc = (ABF16 && DROP) ? dscale : 1.f;
    if (kin) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = 2 * acc_row(r, lane) + a;
                *reinterpret_cast<float2*>(out + (size_t)n * K) = make_float2(acc[a][0][r] * osc, acc[a][1][r] * osc);
            }
    }
    if (want_bias) {
        float2 b2 = make_float2(bs[0], bs[1]);
        b2.x = lane_pair32(b2.x, [](float a, float b) { return a + b; }); b2.y = lane_pair32(b2.y, [](float a, float b) { return a + b; });
        if (h == 0) *reinterpret_cast<float2*>(j.out_bias[gb] + (size_t)ch * D + 64 * nh + 2 * i) = b2;
    }
}

// =====================================================================================================================
// k_wgrad4: the split products of k_wgrad3 with every operand element split ONCE per workgroup (round 3, second half).
//
// k_wgrad3 is LDS-free: each of its four waves splits the G columns and the A columns of its own 64 x 64 quadrant, so inside a
// 128 x 128 block every element is split twice, and the VisualProjection job (K = 1024: eight k tiles) splits -- and, for A, hashes --
// G sixteen times: 50 M split pairs + hashes for 9.4 M distinct elements (26 us).  Here the 8 waves of a workgroup stage a 16-row step of
// both operands together: thread = one row pair x one column pair of G and of A (coalesced 8-byte loads, dropout, exact 3-way split),
// written as bf16 planes [operand][term][row half][column][8 rows] -- a lane's MFMA operand (its column, rows 8h .. 8h+7) is one
// ds_read_b128 and a wave reads 1 KiB without a bank conflict.  wave = 64 (n) x 32 (k) of the block: 9
// operand reads feed 12 MFMAs per step.  Two LDS buffers, ONE barrier per step (the reads of step s - 1 precede every wave's arrival at
// barrier s, the writes of step s + 1 follow it); raw rows travel three steps ahead in registers.  48 KB of LDS.
// =====================================================================================================================
constexpr int WG4_T = 512, WG4_NB = 4;
#ifdef WG4_STAMPS      // harness builds (tools/ubench/wgrad_harness.hip): 100 MHz wall-clock stamps of workgroup 0, thread 0
__device__ long long g_wg4_stamps[8];
#define WG4STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_wg4_stamps[k] = wall_clock64(); } while (0)
#else
#define WG4STAMP(k) do { } while (0)
#endif
// dword index of row pair rg (rows 2 rg, 2 rg + 1) of column c: the two 8-row halves of a column live in separate arrays, so the 16 lanes of a
// ds_read_b128 phase (consecutive columns, one half) cover all 64 banks once
__device__ __forceinline__ int wg4_idx(int buf, int o, int p, int c, int rg) { return (((((buf * 2 + o) * 3 + p) * 2 + (rg >> 2)) * 128 + c) << 2) + (rg & 3); }
template <bool DROP>
__global__ __launch_bounds__(WG4_T, 1) void k_wgrad4(WgradBatch wb) {
    __shared__ __attribute__((aligned(16))) uint32_t Ps[2 * 2 * 3 * 128 * 8];
    WG4STAMP(0);
    int ji = 0;
    while (ji + 1 < wb.n && (int)blockIdx.x >= wb.start[ji + 1]) ++ji;
    const WgradJob& j = wb.j[ji];
    const int K = j.K, R = j.R;
    const int crows = wgrad_rows(j);
    const int nkt = (K + 127) >> 7, nch = wgrad_chunks(R, crows);
    const int local = blockIdx.x - wb.start[ji];
    const int kt = local % nkt, ch = (local / nkt) % nch, gb = local / (nkt * nch);
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int ldg = j.ldg ? j.ldg : D;
    const bool blocks = j.nA > 0;
    const int lda = blocks ? D : K;
    const int rbeg = ch * crows, rend = min(R, rbeg + crows), nrows = rend - rbeg;
    const uint32_t dseed = j.dp.seed, dthr = j.dp.thresh, dkey = j.dp.key;
    const float dscale = j.dp.scale;
    WG4STAMP(1);
    // ---- staging role: row pair rg of the step (rows 2 rg, 2 rg + 1), column pair cp (columns 2 cp, 2 cp + 1) of both operands.
    //      lane = (rg, cp & 7): the 64 lanes of a wave write 32 distinct banks (all lanes on one row pair would hit 8)
    const int rg = lane >> 3, cp = 8 * wv + (lane & 7);
    const int kcol = kt * 128 + 2 * cp;                       // column of dW / of Afull
    const bool kin = kcol < K;                                // K is even
    // ---- MFMA role: wave = rows 64 nh .. + 63 (two 32-row blocks) x columns 32 kq .. + 31 of the 128 x 128 block
    const int nh = wv & 1, kq = wv >> 1, i = lane & 31, h = lane >> 5;
    const bool want_bias = kt == 0 && j.out_bias[gb] != nullptr;
    f32x16 acc[2];
    zero_acc(acc);
    float bs0 = 0.f, bs1 = 0.f;
    float2 rawg[WG4_NB][2], rawa[WG4_NB][2];
    // Loads are UNCONDITIONAL (a row past the chunk re-reads its last row, a column past K column 0) and the zeroing happens at staging time:
    // a load under an exec-masked branch costs the compiler its count of what is in flight, and its one wait per ring turn became vmcnt(0) --
    // for the four loads issued a few instructions earlier as well, a memory round trip every WG4_NB steps (round 4).
    const float* gbase = j.G[gb] + 2 * cp;
    const float* abase = blocks ? j.A[kt] + 2 * cp : j.Afull + (kin ? kcol : 0);
    auto ld = [&](int s, float2 (&g)[2], float2 (&a)[2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const size_t row = (size_t)min(rbeg + 16 * s + 2 * rg + q, rend - 1);
            g[q] = *reinterpret_cast<const float2*>(gbase + row * ldg);
            a[q] = *reinterpret_cast<const float2*>(abase + row * lda);
        }
    };
    // rows 2 rg, 2 rg + 1 of column c of operand o
    auto put = [&](int buf, int o, int c, float x0, float x1) {
        uint32_t hh, mm, ll;
        split3(x0, x1, hh, mm, ll);
        Ps[wg4_idx(buf, o, 0, c, rg)] = hh; Ps[wg4_idx(buf, o, 1, c, rg)] = mm; Ps[wg4_idx(buf, o, 2, c, rg)] = ll;
    };
    auto stage = [&](int s, int buf, const float2 (&gr)[2], const float2 (&ar)[2]) {
        float2 g[2], a[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bool ok = 16 * s + 2 * rg + q < nrows;
            g[q] = ok ? gr[q] : make_float2(0.f, 0.f);
            a[q] = (ok && kin) ? ar[q] : make_float2(0.f, 0.f);
        }
        if (want_bias) { bs0 += g[0].x + g[1].x; bs1 += g[0].y + g[1].y; }
        put(buf, 0, 2 * cp, g[0].x, g[1].x);
        put(buf, 0, 2 * cp + 1, g[0].y, g[1].y);
        float a00 = a[0].x, a01 = a[0].y, a10 = a[1].x, a11 = a[1].y;
        if (DROP) {
            const uint32_t base = (uint32_t)(rbeg + 16 * s + 2 * rg) * (uint32_t)K + (uint32_t)kcol;
            a00 *= drop_hash(base, dseed, dkey) >= dthr ? dscale : 0.f;
            a01 *= drop_hash(base + 1u, dseed, dkey) >= dthr ? dscale : 0.f;
            a10 *= drop_hash(base + (uint32_t)K, dseed, dkey) >= dthr ? dscale : 0.f;
            a11 *= drop_hash(base + (uint32_t)K + 1u, dseed, dkey) >= dthr ? dscale : 0.f;
        }
        put(buf, 1, 2 * cp, a00, a10);
        put(buf, 1, 2 * cp + 1, a01, a11);
    };
    struct Frag { u32x4_t g[2][3], a[3]; };
    auto frag_load = [&](int buf, Frag& f) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int a = 0; a < 2; ++a) f.g[a][p] = *reinterpret_cast<const u32x4_t*>(Ps + wg4_idx(buf, 0, p, 64 * nh + 32 * a + i, 4 * h));
            f.a[p] = *reinterpret_cast<const u32x4_t*>(Ps + wg4_idx(buf, 1, p, 32 * kq + i, 4 * h));
        }
    };
    auto mma = [&](const Frag& f) {
        constexpr int TG6[6] = {1, 0, 2, 0, 1, 0}, TA6[6] = {1, 2, 0, 1, 0, 0};      // (g term, a term): mm, hl, lh, hm, mh, hh (small terms first)
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int a = 0; a < 2; ++a) acc[a] = mfma_bf16(f.g[a][TG6[t]], f.a[TA6[t]], acc[a]);
    };
    const int ns = ((nrows + 16 * WG4_NB - 1) / (16 * WG4_NB)) * WG4_NB;      // steps, a multiple of the ring (a step past the rows multiplies zeros)
    static_for<0, WG4_NB - 1>([&](auto uc) { constexpr int u = decltype(uc)::value; ld(u, rawg[u], rawa[u]); });
    stage(0, 0, rawg[0], rawa[0]);
    WG4STAMP(2);
    for (int s0 = 0; s0 < ns; s0 += WG4_NB) {
        static_for<0, WG4_NB>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            const int s = s0 + u;
            ld(s + WG4_NB - 1, rawg[(u + WG4_NB - 1) % WG4_NB], rawa[(u + WG4_NB - 1) % WG4_NB]);       // the slot step s - 1 was staged from
            __syncthreads();
            // the operand reads of step s are issued BEFORE the staging of step s + 1 (the other buffer): their latency hides behind the
            // split arithmetic (21.3 -> 19.1 us).  Measured and dropped: MFMAs before the staging in half / all of the waves (19.3 / 19.3 us);
            // producer / consumer wave roles with 64 x 64 consumer tiles (12 reads per 24 MFMAs: 20.1 us); round 4: the staging woven between the
            // MFMAs in four parts (loop of workgroup 0: 11.8 -> 12.4 us, harness stamps: job lookup 0.7, first loads + stage 1.7, 16-step loop
            // 11.8, slab stores 0.6 us) -- the step is a mix of LDS bandwidth (96 KB per step, a quarter of it ds_write_b32 at half rate), the
            // barrier and the matrix pipe, none of them alone
            Frag f;
            frag_load(u & 1, f);
            __builtin_amdgcn_sched_barrier(0);
            stage(s + 1, (u + 1) & 1, rawg[(u + 1) % WG4_NB], rawa[(u + 1) % WG4_NB]);
            mma(f);
        });
    }
    WG4STAMP(3);
    // ---- partial slab: register r of acc[a] = dW[n = 64 nh + 32 a + acc_row(r)][k = kt * 128 + 32 kq + i]
    const int N = 128 * j.nG;
    const int kglob = kt * 128 + 32 * kq + i;
    float* out = j.out + ((size_t)ch * N + gb * 128 + 64 * nh) * K + kglob;
    if (kglob < K) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[(size_t)(32 * a + acc_row(r, lane)) * K] = acc[a][r];
    }
    WG4STAMP(4);
    if (want_bias) {            // column sums of G: the 8 row pairs of a column pair sit in 8 lanes (lane >> 3) of one wave
        __syncthreads();
        float* red = reinterpret_cast<float*>(Ps);
        red[rg * 128 + 2 * cp] = bs0; red[rg * 128 + 2 * cp + 1] = bs1;
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += red[q * 128 + tid];
            j.out_bias[gb][(size_t)ch * D + tid] = t;
        }
    }
}

// One launch per batch of jobs.  fp32 operands: k_wgrad4 (every element split once per workgroup through LDS); bfloat16 features (the bf16
// throughput mode's VisualProjection job): k_wgrad3, which widens them in registers.  The dropout hash (VisualProjection input) and the bf16
// operand are separate instantiations: a batch that mixes kinds is launched kind by kind, in order.
void launch_wgrad(const WgradBatch& wb0, hipStream_t s) {
    WgradBatch wb = wb0;
    int total = 0;
    for (int i = 0; i < wb.n; ++i) {
        wb.start[i] = total;
        total += wb.j[i].nG * ((wb.j[i].K + 127) / 128) * wgrad_chunks(wb.j[i].R, wgrad_rows(wb.j[i]));
    }
    wb.start[wb.n] = total;
    if (total == 0) return;
    auto kind = [](const WgradJob& j) { return (j.nA == 0 && j.drop_on_A && j.dp.thresh ? 1 : 0) | (j.nA == 0 && j.a_bf16 ? 2 : 0); };
    // A batch that mixes the two fp32 kinds is ONE launch of the dropout instantiation: the jobs without dropout ride in it with a neutral mask
    // (threshold 0, scale 1: every hash keeps, x 1.0 is exact) instead of a launch of their own -- one boundary less on the step's tail, where
    // the video pass' pointwise gradients and the VisualProjection gradient are the last two launches in front of the final reduction
    // (same box, four pairs: 0.8722 -> 0.8682 ms, profiles/r06_raw/q16_ab.txt; at the end of the round, with the batch in whole rounds: 0.8434 against 0.8712).
    {
        bool any1 = false, only01 = true;
        for (int i = 0; i < wb.n; ++i) { any1 = any1 || kind(wb.j[i]) == 1; only01 = only01 && kind(wb.j[i]) < 2; }
        if (any1 && only01) {
            total = 0;
            WgradBatch m;
            m.n = 0;
            for (int pass = 1; pass >= 0; --pass)                     // the dropout jobs (the long ones) first
                for (int i = 0; i < wb.n; ++i)
                    if (kind(wb.j[i]) == pass) {
                        WgradJob j = wb.j[i];
                        if (pass == 0) j.dp = Drop{0u, 0u, 1.0f, 0u};
                        m.start[m.n] = total;
                        total += j.nG * ((j.K + 127) / 128) * wgrad_chunks(j.R, wgrad_rows(j));
                        m.j[m.n++] = j;
                    }
            m.start[m.n] = total;
            VSL_LAUNCH((k_wgrad4<true>), dim3(total), dim3(WG4_T), 0, s, m);
            return;
        }
    }
    const int k0 = kind(wb.j[0]);
    bool mixed = false;
    for (int i = 1; i < wb.n; ++i) mixed = mixed || kind(wb.j[i]) != k0;
    if (mixed) {                             // split by kind, keeping the order
        for (int kd = 0; kd < 4; ++kd) {
            WgradBatch part;
            part.n = 0;
            for (int i = 0; i < wb.n; ++i) if (kind(wb.j[i]) == kd) part.j[part.n++] = wb.j[i];
            if (part.n) launch_wgrad(part, s);
        }
        return;
    }
    switch (k0) {
        case 0: VSL_LAUNCH((k_wgrad4<false>), dim3(total), dim3(WG4_T), 0, s, wb); break;
        case 1: VSL_LAUNCH((k_wgrad4<true>), dim3(total), dim3(WG4_T), 0, s, wb); break;
        case 2: VSL_LAUNCH((k_wgrad3<false, true>), dim3(total), dim3(WG2_T), 0, s, wb); break;       // bfloat16 features
        default: VSL_LAUNCH((k_wgrad3<true, true>), dim3(total), dim3(WG2_T), 0, s, wb); break;
    }
}

}  // namespace vsl
