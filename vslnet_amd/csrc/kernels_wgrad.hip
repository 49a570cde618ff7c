// Weight gradients of every Conv1D on the path (gfx950):  dW[n][k] = sum_r G[r][n] A[r][k]  (+ bias = column sums of G).
//
// The contraction runs over ROWS, and both operands are row-major with the contraction index as the slow one -- exactly
// the operand shape of v_mfma_f32_32x32x2_f32 (lane (i, h) supplies element i of row r + h).  So nothing is staged through
// LDS: a lane's float2 of a G row is TWO A-operands (output rows n = 2 i + 0, 1), its float2 of an A row is TWO B-operands
// (output columns k = 2 i + 0, 1), and one pair of rows feeds 4 MFMAs from 2 global loads -- no LDS traffic, no barrier,
// a handful of vector-ALU instructions per 512 matrix cycles (fp32 MFMAs and the vector ALU share the SIMD issue, so the
// LDS-staged round-1 kernel paid for every address, mask and ds_write: 74 % of the MFMA bound inside its loop).
// Workgroup = 4 waves (one per SIMD) = one 128 x 128 block of dW over one row chunk; wave = 64 (n) x 64 (k) quadrant, 64
// accumulator registers, two 8-byte loads per row pair feed 4 MFMAs.  Loads run WG2_PF row pairs ahead in a register ring.  Partial slab per (row chunk) as before; k_reduce sums them.
#include "common.hpp"
#include "launch.hpp"
#include <type_traits>

namespace vsl {

constexpr int WG2_T = 256;
constexpr int WG2_PF = 16;           // row pairs in flight per wave (16 x 16 B per lane)

template <bool DROP, bool ABF16>     // ABF16: Afull is bfloat16 (bf16 throughput mode: the video features), widened to fp32 on load
__global__ __launch_bounds__(WG2_T, 2) void k_wgrad2(WgradBatch wb) {
    int ji = 0;
    while (ji + 1 < wb.n && (int)blockIdx.x >= wb.start[ji + 1]) ++ji;
    const WgradJob& j = wb.j[ji];
    const int K = j.K, R = j.R;
    const int nkt = (K + 127) >> 7, nch = (R + WG_ROWS - 1) / WG_ROWS;
    const int local = blockIdx.x - wb.start[ji];
    const int kt = local % nkt, ch = (local / nkt) % nch, gb = local / (nkt * nch);
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int nh = wv & 1, kh = wv >> 1;                    // this wave's 64 x 64 quadrant of the 128 x 128 block
    const int ldg = j.ldg ? j.ldg : D;
    const float* Gp = j.G[gb] + 64 * nh + 2 * i;
    const bool blocks = j.nA > 0;
    const int lda = blocks ? D : K;
    const int kloc = 64 * kh + 2 * i;                       // column inside the 128-wide k tile
    const int kglob = kt * 128 + kloc;                      // column of dW
    const bool kin = kglob < K;                             // K is even: both columns of the float2 are in or out together
    const float* Ap = blocks ? j.A[kt] + kloc : j.Afull + (kin ? kglob : 0);
    const int rbeg = ch * WG_ROWS, rend = min(R, rbeg + WG_ROWS);
    const int np = (rend - rbeg + 1) >> 1;                  // row pairs
    const uint32_t dseed = j.dp.seed, dthr = j.dp.thresh, dkey = j.dp.key;
    const float dscale = j.dp.scale;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float2 bs = make_float2(0.f, 0.f);
    float2 gq[WG2_PF], aq[WG2_PF];
    const uint16_t* Ap16 = reinterpret_cast<const uint16_t*>(j.Afull) + (kin ? kglob : 0);
    auto ld = [&](int p, float2& g, float2& a) {            // rows past the chunk re-read its last row (masked when used)
        const size_t row = (size_t)min(rbeg + 2 * p + h, rend - 1);
        g = *reinterpret_cast<const float2*>(Gp + row * ldg);
        if (ABF16) {          // the raw pair travels through the ring in a.x; widened where it is used (a conversion here would wait for the load)
            a.x = __uint_as_float(*reinterpret_cast<const uint32_t*>(Ap16 + row * lda));
        } else {
            a = *reinterpret_cast<const float2*>(Ap + row * lda);
        }
    };
#pragma unroll
    for (int q = 0; q < WG2_PF; ++q) ld(q, gq[q], aq[q]);
    // Columns past K (ragged last k tile) need no mask: an output column depends on its own B-operand lane only, and those
    // lanes (clamped address, finite garbage) are not stored.  Rows past the chunk are masked in the one tail block.
    auto block = [&](int p0, auto masked_c) {
#pragma unroll
        for (int q = 0; q < WG2_PF; ++q) {
            float2 g = gq[q];
            float2 a = aq[q];
            if (ABF16) {      // two bf16 = the high halves of two floats
                const uint32_t u = __float_as_uint(a.x);
                a = make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xFFFF0000u));
            }
            const int row = rbeg + 2 * (p0 + q) + h;
            if (decltype(masked_c)::value) {
                const float mg = row < rend ? 1.f : 0.f;
                g.x *= mg; g.y *= mg;
            }
            if (DROP) {
                const uint32_t base = (uint32_t)row * (uint32_t)K + (uint32_t)kglob;
                a.x *= drop_hash(base, dseed, dkey) >= dthr ? dscale : 0.f;
                a.y *= drop_hash(base + 1u, dseed, dkey) >= dthr ? dscale : 0.f;
            }
            bs.x += g.x; bs.y += g.y;
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g.x, a.x, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g.y, a.x, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g.x, a.y, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g.y, a.y, acc[1][1], 0, 0, 0);
            // the slot is reloaded AFTER its MFMAs were issued: the old value is dead, so the load lands in the same
            // registers and the loop needs no copies (a copy of a just-loaded value is a vmcnt(0) at the loop end)
            ld(p0 + q + WG2_PF, gq[q], aq[q]);
            // keep every slot's reload next to its own MFMAs: left alone, the scheduler sinks all 2 x PF loads to the end of
            // the block and the next block opens with vmcnt(0) -- a full memory latency per block, nothing in flight
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int nfull = ((rend - rbeg) >> 1) / WG2_PF * WG2_PF;     // row pairs in blocks that lie completely inside the chunk
    for (int p0 = 0; p0 < nfull; p0 += WG2_PF) block(p0, std::false_type());
    if (nfull < np) block(nfull, std::true_type());
    // ---- partial slab: lane, register r of acc[a][b] = dW[n = 64 nh + 2 * acc_row(r) + a][k = kglob_of(lane & 31) + b]
    const int N = 128 * j.nG;
    float* out = j.out + ((size_t)ch * N + gb * 128 + 64 * nh) * K + kglob;
    if (kin) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = 2 * acc_row(r, lane) + a;
                *reinterpret_cast<float2*>(out + (size_t)n * K) = make_float2(acc[a][0][r], acc[a][1][r]);
            }
    }
    if (kt == 0 && kh == 0 && j.out_bias[gb]) {
        bs.x = lane_pair32(bs.x, [](float a, float b) { return a + b; }); bs.y = lane_pair32(bs.y, [](float a, float b) { return a + b; });
        if (h == 0) *reinterpret_cast<float2*>(j.out_bias[gb] + (size_t)ch * D + 64 * nh + 2 * i) = bs;
    }
}
void launch_wgrad2(const WgradBatch& wb0, hipStream_t s) {
    WgradBatch wb = wb0;
    int total = 0;
    for (int i = 0; i < wb.n; ++i) {
        wb.start[i] = total;
        total += wb.j[i].nG * ((wb.j[i].K + 127) / 128) * ((wb.j[i].R + WG_ROWS - 1) / WG_ROWS);
    }
    wb.start[wb.n] = total;
    if (total == 0) return;
    // the dropout hash (VisualProjection input) and the bf16 operand are separate instantiations: such jobs are launched on their own
    auto kind = [](const WgradJob& j) { return (j.nA == 0 && j.drop_on_A && j.dp.thresh ? 1 : 0) | (j.nA == 0 && j.a_bf16 ? 2 : 0); };
    const int k0 = kind(wb.j[0]);
    bool mixed = false;
    for (int i = 1; i < wb.n; ++i) mixed = mixed || kind(wb.j[i]) != k0;
    if (mixed) {                             // split by kind, keeping the order
        for (int kd = 0; kd < 4; ++kd) {
            WgradBatch part;
            part.n = 0;
            for (int i = 0; i < wb.n; ++i) if (kind(wb.j[i]) == kd) part.j[part.n++] = wb.j[i];
            if (part.n) launch_wgrad2(part, s);
        }
        return;
    }
    // dynamic LDS is requested only to bound how many of these workgroups share a CU (two of them serialise on its matrix pipes
    // while other CUs idle): VSL_WGRAD_LDS=<bytes>, default 84 KB = one per CU.  Measured at the headline shape, ms/step with
    // 0 / 66 / 84 / 96 KB: 1.119 / 1.115 / 1.104 / 1.107 (profiles/r02_notes.md)
    static const size_t pad = getenv("VSL_WGRAD_LDS") ? (size_t)atol(getenv("VSL_WGRAD_LDS")) : (size_t)84 * 1024;
    static size_t ok[4] = {0, 0, 0, 0};
    const void* fn[4] = {(const void*)k_wgrad2<false, false>, (const void*)k_wgrad2<true, false>, (const void*)k_wgrad2<false, true>,
                         (const void*)k_wgrad2<true, true>};
    ensure_dynamic_lds(fn[k0], pad, ok[k0], "k_wgrad2");
    switch (k0) {
        case 0: VSL_LAUNCH((k_wgrad2<false, false>), dim3(total), dim3(WG2_T), pad, s, wb); break;
        case 1: VSL_LAUNCH((k_wgrad2<true, false>), dim3(total), dim3(WG2_T), pad, s, wb); break;
        case 2: VSL_LAUNCH((k_wgrad2<false, true>), dim3(total), dim3(WG2_T), pad, s, wb); break;
        default: VSL_LAUNCH((k_wgrad2<true, true>), dim3(total), dim3(WG2_T), pad, s, wb); break;
    }
}

}  // namespace vsl
