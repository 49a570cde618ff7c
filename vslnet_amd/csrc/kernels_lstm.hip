// a15 DynamicRNN (/root/reference/model/layers_t7.py:302-313): the recurrence of nn.LSTM(128, 128) for the rnn predictor head, on
// FOUR-sample groups.  The recurrent product of a step is (samples x 128) x (128 x 512): with 16-row MFMA tiles a 16-sample group
// is the smallest unit, it owns ONE CU for the whole sequence (8.2 k matrix cycles per step: configs[0]'s B = 16 ran on 1 of 256
// CUs, 6.4 us per step).  v_mfma_f32_4x4x1_16B_f32 computes 16 independent 4 x 4 outer products per instruction at the same
// FLOP rate: 4 SAMPLES x 64 gate columns x 1 k, with the A operand (the 4 samples' h) broadcast from lanes 0-3 to all 16 blocks
// (cbsz = 4).  A 4-sample group then costs 2 k matrix cycles per step and B = 16 spreads over 4 CUs, B = 64 over 16.
//
//   workgroup = 4 samples, 8 waves.  Lane l of wave w: block b = l >> 2, j = l & 3.
//   forward : gate column of the lane = gate j of hidden unit u = 16 w + b  (B operand = row j * 128 + u of W_hh, 128 registers);
//             D register r = that gate's pre-activation for sample r.  A 256-float LDS scratch per wave turns [sample][unit][gate]
//             around so that lane (u, j) finishes SAMPLE j of unit u: all 512 (sample, unit) cells, one per lane.
//   backward: the same lane computes the cell's four gate gradients -> LDS (4 x 512, A operand) and memory; dh_{t-1} = dG_t W_hh:
//             wave w contracts gate rows 64 w .. 64 w + 63 into all 128 columns (B operand = W_hh[row][64 cg + 4 b + j], 128
//             registers), the 8 partial tiles are added through LDS by the cell owners.
// Same saved tensors, chunk / carry interface and launch signatures as the 16-sample kernels (VSL_LSTM4=0 selects those).
#include "common.hpp"
#include "launch.hpp"

namespace vsl {

constexpr int L4_HP = D + 4;            // LDS row stride of h / partial tiles
constexpr int L4_GP = 4 * D + 4;        // LDS row stride of the gate-gradient rows

// cbsz = 4 broadcasts the A operand of ONE 4-lane block to all 16 blocks, abid picks the block: the samples' h (or gate gradients) are
// read from LDS once, spread over the wave (block b holds its own k slice), instead of every block re-reading every k.
template <int AB> __device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, AB, 0); }
// forward: block AB holds h[sample j][8 AB .. 8 AB + 7]
template <int AB> __device__ __forceinline__ void l4_fwd_blk(const float (&hr)[8], const float (&wr)[D], f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3) {
    a0 = mfma4<AB>(hr[0], wr[8 * AB + 0], a0); a1 = mfma4<AB>(hr[1], wr[8 * AB + 1], a1);
    a2 = mfma4<AB>(hr[2], wr[8 * AB + 2], a2); a3 = mfma4<AB>(hr[3], wr[8 * AB + 3], a3);
    a0 = mfma4<AB>(hr[4], wr[8 * AB + 4], a0); a1 = mfma4<AB>(hr[5], wr[8 * AB + 5], a1);
    a2 = mfma4<AB>(hr[6], wr[8 * AB + 6], a2); a3 = mfma4<AB>(hr[7], wr[8 * AB + 7], a3);
}
// backward: block AB holds dG[sample j][64 w + 4 AB .. + 3]
template <int AB> __device__ __forceinline__ void l4_bwd_blk(const float4& av, const float (&wr)[2][64], f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3) {
    c0 = mfma4<AB>(av.x, wr[0][4 * AB], c0);     c1 = mfma4<AB>(av.x, wr[1][4 * AB], c1);
    c2 = mfma4<AB>(av.y, wr[0][4 * AB + 1], c2); c3 = mfma4<AB>(av.y, wr[1][4 * AB + 1], c3);
    c0 = mfma4<AB>(av.z, wr[0][4 * AB + 2], c0); c1 = mfma4<AB>(av.z, wr[1][4 * AB + 2], c1);
    c2 = mfma4<AB>(av.w, wr[0][4 * AB + 3], c2); c3 = mfma4<AB>(av.w, wr[1][4 * AB + 3], c3);
}
#define L4_ALL_BLOCKS(F, ...) F<0>(__VA_ARGS__); F<1>(__VA_ARGS__); F<2>(__VA_ARGS__); F<3>(__VA_ARGS__); F<4>(__VA_ARGS__); F<5>(__VA_ARGS__); \
    F<6>(__VA_ARGS__); F<7>(__VA_ARGS__); F<8>(__VA_ARGS__); F<9>(__VA_ARGS__); F<10>(__VA_ARGS__); F<11>(__VA_ARGS__); F<12>(__VA_ARGS__);   \
    F<13>(__VA_ARGS__); F<14>(__VA_ARGS__); F<15>(__VA_ARGS__)

__global__ __launch_bounds__(512, 2) void k_lstm4_fwd(const float* __restrict__ gi, const float* __restrict__ Whh,
                                                      const float* __restrict__ bih, const float* __restrict__ bhh,
                                                      const float* __restrict__ mask, float* __restrict__ gates,
                                                      float* __restrict__ cseq, float* __restrict__ hprev, float* __restrict__ out,
                                                      int B, int T, int t0, int t1) {
    __shared__ __attribute__((aligned(16))) float hs[2][4 * L4_HP];
    __shared__ __attribute__((aligned(16))) float zs[8][256];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b = lane >> 2, j = lane & 3;
    const int u = 16 * w + b;                             // hidden unit of this lane (as gate column: gate j ; as cell: sample j)
    const int b0 = blockIdx.x * 4;
    float wr[D];                                          // B operand: W_hh[j * 128 + u][0 .. 127]
    {
        const float4* p = reinterpret_cast<const float4*>(Whh + (size_t)(j * D + u) * D);
#pragma unroll
        for (int q = 0; q < D / 4; ++q) { const float4 v = p[q]; wr[4 * q] = v.x; wr[4 * q + 1] = v.y; wr[4 * q + 2] = v.z; wr[4 * q + 3] = v.w; }
    }
    float bsum[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bsum[g] = bih[g * D + u] + bhh[g * D + u];
    const bool ok = b0 + j < B;                           // cell (sample j, unit u)
    const int row = min(b0 + j, B - 1) * T;
    float cst = 0.f, Gc[4], Mk;
    auto gi_load = [&](int t) {
        const int tt = min(t, T - 1);
        const float* p = gi + (unsigned)((row + tt) * (4 * D) + u);
#pragma unroll
        for (int g = 0; g < 4; ++g) Gc[g] = p[g * D];
        Mk = mask[row + tt];
    };
    // a launch covers the steps [t0, t1): it resumes from what the previous chunk saved for the backward (hprev[t0], cseq[t0 - 1])
    hs[t0 & 1][j * L4_HP + u] = t0 > 0 ? hprev[(unsigned)((row + t0) * D + u)] : 0.f;
    if (t0 > 0) cst = cseq[(unsigned)((row + t0 - 1) * D + u)];
    gi_load(t0);
    __syncthreads();
    float* zw = zs[w];
    for (int t = t0; t < t1; ++t) {
        const int cur = t & 1;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        if (t > 0) {
            const float* hrow = &hs[cur][j * L4_HP + 8 * b];              // A operand: block b = h[samples 0-3][8 b .. 8 b + 7]
            const float4 h0 = *reinterpret_cast<const float4*>(hrow), h1 = *reinterpret_cast<const float4*>(hrow + 4);
            const float hr[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
            L4_ALL_BLOCKS(l4_fwd_blk, hr, wr, a0, a1, a2, a3);
        }
        // register r = pre-activation of gate j, unit u, sample r  ->  scratch [sample][unit][gate]  ->  lane reads its sample's 4 gates
#pragma unroll
        for (int r = 0; r < 4; ++r) zw[r * 64 + lane] = (a0[r] + a1[r]) + (a2[r] + a3[r]);
        const float4 z = *reinterpret_cast<const float4*>(&zw[(j * 16 + b) * 4]);
        const float ig = sigmoid_fast(z.x + Gc[0] + bsum[0]), fg = sigmoid_fast(z.y + Gc[1] + bsum[1]);
        const float gg = tanh_fast(z.z + Gc[2] + bsum[2]), og = sigmoid_fast(z.w + Gc[3] + bsum[3]);
        const float cn = fg * cst + ig * gg;
        const float hn = og * tanh_fast(cn);
        cst = cn;
        hs[cur ^ 1][j * L4_HP + u] = hn;
        if (ok) {
            const unsigned base = (unsigned)(row + t);
            float* gp = gates + base * (4 * D) + u;
            gp[0] = ig; gp[D] = fg; gp[2 * D] = gg; gp[3 * D] = og;
            cseq[base * D + u] = cn;
            out[base * D + u] = hn * Mk;
            if (t == 0) hprev[base * D + u] = 0.f;
            if (t + 1 < T) hprev[(base + 1) * D + u] = hn;
        }
        gi_load(t + 1);
        __syncthreads();
    }
}

__global__ __launch_bounds__(512, 2) void k_lstm4_bwd(const float* __restrict__ dout, const float* __restrict__ dout2,
                                                      const float* __restrict__ mask, const float* __restrict__ gates,
                                                      const float* __restrict__ cseq, const float* __restrict__ Whh,
                                                      float* __restrict__ dG, int B, int T, float* __restrict__ carry, int t0,
                                                      int t1) {
    __shared__ __attribute__((aligned(16))) float dGs[4 * L4_GP];         // gate gradients of the step: A operand
    __shared__ __attribute__((aligned(16))) float Pp[8][4 * L4_HP];       // per-wave partial tiles of dh_{t-1}
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b = lane >> 2, j = lane & 3;
    const int u = 16 * w + b;                             // cell (sample j, unit u)
    const int b0 = blockIdx.x * 4;
    float wr[2][64];                                      // B operand: W_hh[64 w + kk][64 cg + lane]
#pragma unroll
    for (int cg = 0; cg < 2; ++cg)
#pragma unroll
        for (int kk = 0; kk < 64; ++kk) wr[cg][kk] = Whh[(size_t)(64 * w + kk) * D + 64 * cg + lane];
    const bool ok = b0 + j < B;
    const int bb = min(b0 + j, B - 1);
    // a launch covers the steps [t0, t1) in reverse; a later time chunk hands dc_{t1} and dh_{t1 - 1} over through `carry`
    float dcn = 0.f, dhr = 0.f;
    if (t1 < T) { dcn = carry[((size_t)bb * 2 + 0) * D + u]; dhr = carry[((size_t)bb * 2 + 1) * D + u]; }
    for (int t = t1 - 1; t >= t0; --t) {
        const size_t base = (size_t)bb * T + t;
        float dh = dout[base * D + u];
        if (dout2) dh += dout2[base * D + u];
        dh *= mask[base];
        if (t < T - 1) dh += dhr;
        const float* gp = gates + base * (4 * D) + u;
        const float ig = gp[0], fg = gp[D], gg = gp[2 * D], og = gp[3 * D];
        const float ct = cseq[base * D + u], cp = t > 0 ? cseq[(base - 1) * D + u] : 0.f;
        const float tc = tanh_fast(ct);
        const float dc = dh * og * (1.f - tc * tc) + dcn;
        float dv[4] = {dc * gg * ig * (1.f - ig), dc * cp * fg * (1.f - fg), dc * ig * (1.f - gg * gg), dh * tc * og * (1.f - og)};
        dcn = dc * fg;
        if (!ok) { dv[0] = dv[1] = dv[2] = dv[3] = 0.f; }
#pragma unroll
        for (int g = 0; g < 4; ++g) dGs[j * L4_GP + g * D + u] = dv[g];
        if (ok) {
            float* op = dG + base * (4 * D) + u;
            op[0] = dv[0]; op[D] = dv[1]; op[2 * D] = dv[2]; op[3 * D] = dv[3];
        }
        if (t == 0) break;                               // dh_{-1} is not needed
        __syncthreads();
        // dh_{t-1}[i][n] = sum_k dG[i][k] W_hh[k][n]: this wave's 64 gate rows into all 128 columns (two 64-column groups)
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
        const float4 av = *reinterpret_cast<const float4*>(dGs + j * L4_GP + 64 * w + 4 * b);   // A operand: block b = rows 64 w + 4 b .. + 3
        L4_ALL_BLOCKS(l4_bwd_blk, av, wr, c0, c1, c2, c3);
#pragma unroll
        for (int r = 0; r < 4; ++r) {                     // register r = sample r ; lane = column inside the 64-column group
            Pp[w][r * L4_HP + lane] = c0[r] + c2[r];
            Pp[w][r * L4_HP + 64 + lane] = c1[r] + c3[r];
        }
        __syncthreads();
        dhr = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) dhr += Pp[q][j * L4_HP + u];          // fixed order: deterministic
    }
    if (t0 > 0 && ok) {
        carry[((size_t)bb * 2 + 0) * D + u] = dcn;
        carry[((size_t)bb * 2 + 1) * D + u] = dhr;
    }
}

// =========================================================================================================
// One-sample workgroups: the same recurrence on the VECTOR pipe.  A 4-sample group pays the full 128 x 512 product on one CU's matrix
// cores every step (2.1 k cycles -- three quarters of the 4 x 4 tiles' rows are all a small batch has to offer anyway); one sample per
// workgroup needs 128 packed FMAs per lane (v_pk_fma_f32, 512 cycles per SIMD per step) and spreads configs[0]'s B = 16 over 16 CUs.
//   lane (u = 16 w + b, j): gate column j * 128 + u, W_hh row in 128 registers; h is read from LDS as wave-uniform broadcasts.
//   the four gates of a unit sit in one quad: each lane activates ITS gate (tanh(x) = 2 sigmoid(2x) - 1, the same formula tanh_fast
//   uses), DPP quad broadcasts hand all four to every lane, the cell update is computed redundantly by the quad -- no LDS transposition.
//   backward: lane (u, j) contracts gate rows 128 j .. + 127 into column u and the quad adds its four slices with DPP: one LDS
//   exchange (the step's 512 gate gradients) and one barrier per step; everything a step loads is fetched a step ahead.
//   (Tried: h through DPP row_newbcast into v_fmac_f32_dpp -- 2 LDS reads per lane and step instead of 32 -- is no faster: 128 vs 118 us,
//   the DPP FMAs issue at half rate; profiles/r02_notes.md.)
// Used for B <= 256 (VSL_LSTM1=0 keeps the 4-sample kernels); saved tensors, chunk / carry interface identical.
// =========================================================================================================
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int I> __device__ __forceinline__ float quad_bcast(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), I * 0x55, 0xF, 0xF, false));
}

__global__ __launch_bounds__(512, 2) void k_lstm1_fwd(const float* __restrict__ gi, const float* __restrict__ Whh,
                                                      const float* __restrict__ bih, const float* __restrict__ bhh,
                                                      const float* __restrict__ mask, float* __restrict__ gates,
                                                      float* __restrict__ cseq, float* __restrict__ hprev, float* __restrict__ out,
                                                      int T, int t0, int t1) {
    __shared__ __attribute__((aligned(16))) float hs[2][D];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b = lane >> 2, j = lane & 3;
    const int u = 16 * w + b;
    const int row = blockIdx.x * T;
    f32x2 wr[D / 2];                                      // W_hh[j * 128 + u][0 .. 127]
    {
        const float4* p = reinterpret_cast<const float4*>(Whh + (size_t)(j * D + u) * D);
#pragma unroll
        for (int q = 0; q < D / 4; ++q) { const float4 v = p[q]; wr[2 * q] = f32x2{v.x, v.y}; wr[2 * q + 1] = f32x2{v.z, v.w}; }
    }
    const float bs = bih[j * D + u] + bhh[j * D + u];
    const float sc = j == 2 ? 2.0f : 1.0f;               // gate 2 is the tanh gate
    float cst = t0 > 0 ? cseq[(unsigned)((row + t0 - 1) * D + u)] : 0.f;
    if (j == 0) hs[t0 & 1][u] = t0 > 0 ? hprev[(unsigned)((row + t0) * D + u)] : 0.f;
    if (j == 3 && t0 == 0) hprev[(unsigned)(row * D + u)] = 0.f;
    // per-lane cursors, advanced by one time step per iteration: the loop holds no address arithmetic beyond the increments.
    // One store per step carries the lane's activated gate; a second one the quad's cell results (lane j = 0: c_t, 1: masked h_t, 2: h_t as
    // next step's hprev row).
    const float* gip = gi + (size_t)(row + t0) * (4 * D) + j * D + u;
    float* gtp = gates + (size_t)(row + t0) * (4 * D) + j * D + u;
    float* qp = (j == 0 ? cseq : j == 1 ? out : hprev + D) + (size_t)(row + t0) * D + u;
    const float* mkp = mask + row;
    float Gc = *gip, Mk = mkp[t0];
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
        const int cur = t & 1;
        f32x2 a0 = {0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        if (t > 0) {
            const float4* hp = reinterpret_cast<const float4*>(hs[cur]);
#pragma unroll
            for (int q = 0; q < D / 8; ++q) {
                const float4 h0 = hp[2 * q], h1 = hp[2 * q + 1];
                a0 = __builtin_elementwise_fma(f32x2{h0.x, h0.y}, wr[4 * q], a0);
                a1 = __builtin_elementwise_fma(f32x2{h0.z, h0.w}, wr[4 * q + 1], a1);
                a2 = __builtin_elementwise_fma(f32x2{h1.x, h1.y}, wr[4 * q + 2], a2);
                a3 = __builtin_elementwise_fma(f32x2{h1.z, h1.w}, wr[4 * q + 3], a3);
            }
        }
        const f32x2 as = (a0 + a1) + (a2 + a3);
        const float z = (as.x + as.y) + Gc + bs;
        const float sg = sigmoid_fast(z * sc);
        const float act = j == 2 ? 2.0f * sg - 1.0f : sg;
        const float ig = quad_bcast<0>(act), fg = quad_bcast<1>(act), gg = quad_bcast<2>(act), og = quad_bcast<3>(act);
        const float cn = fg * cst + ig * gg;
        const float hn = og * tanh_fast(cn);
        cst = cn;
        if (j == 0) hs[cur ^ 1][u] = hn;
        *gtp = act;
        gtp += 4 * D;
        const float qv = j == 0 ? cn : j == 1 ? hn * Mk : hn;
        if (j < 2 || (j == 2 && t + 1 < T)) *qp = qv;
        qp += D;
        if (t + 1 < T) {                                  // uniform
            gip += 4 * D;
            Gc = *gip;
            Mk = mkp[t + 1];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(512, 2) void k_lstm1_bwd(const float* __restrict__ dout, const float* __restrict__ dout2,
                                                      const float* __restrict__ mask, const float* __restrict__ gates,
                                                      const float* __restrict__ cseq, const float* __restrict__ Whh,
                                                      float* __restrict__ dG, int T, float* __restrict__ carry, int t0, int t1) {
    __shared__ __attribute__((aligned(16))) float dGs[2][4 * L4_HP];      // gate gradients of the step, one padded row per gate (double-buffered: one barrier per step)
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b = lane >> 2, j = lane & 3;
    const int u = 16 * w + b;                             // cell u ; this lane's gate j ; product: rows 128 j .. + 127 into column u
    const int bb = blockIdx.x;
    f32x2 wr[D / 2];                                      // W_hh[128 j + 2 i .. + 1][u]
#pragma unroll
    for (int i = 0; i < D / 2; ++i) wr[i] = f32x2{Whh[(size_t)(128 * j + 2 * i) * D + u], Whh[(size_t)(128 * j + 2 * i + 1) * D + u]};
    float dcn = 0.f, dhr = 0.f;
    if (t1 < T) { dcn = carry[((size_t)bb * 2 + 0) * D + u]; dhr = carry[((size_t)bb * 2 + 1) * D + u]; }
    // everything a step reads from memory is independent of the recurrence: fetched one step ahead
    float n_do, n_mk, n_act, n_cp;
    auto fetch = [&](int t) {
        const int tt = max(t, 0);
        const unsigned base = (unsigned)(bb * T + tt);
        n_do = dout[base * D + u];
        if (dout2) n_do += dout2[base * D + u];
        n_mk = mask[base];
        n_act = gates[base * (4 * D) + j * D + u];
        n_cp = tt > 0 ? cseq[(base - 1) * D + u] : 0.f;
    };
    float ct = cseq[(unsigned)((bb * T + t1 - 1) * D + u)];
    fetch(t1 - 1);
    for (int t = t1 - 1; t >= t0; --t) {
        const unsigned base = (unsigned)(bb * T + t);
        const int cur = t & 1;
        float dh = n_do * n_mk;
        const float act = n_act, cp = n_cp;
        fetch(t - 1);
        if (t < T - 1) dh += dhr;
        const float ig = quad_bcast<0>(act), fg = quad_bcast<1>(act), gg = quad_bcast<2>(act), og = quad_bcast<3>(act);
        const float tc = tanh_fast(ct);
        const float dc = dh * og * (1.f - tc * tc) + dcn;
        const float dv = j == 0 ? dc * gg * ig * (1.f - ig) : j == 1 ? dc * cp * fg * (1.f - fg) : j == 2 ? dc * ig * (1.f - gg * gg)
                                                                                                         : dh * tc * og * (1.f - og);
        dcn = dc * fg;
        ct = cp;
        dGs[cur][j * L4_HP + u] = dv;
        dG[base * (4 * D) + j * D + u] = dv;
        if (t == 0) break;                               // dh_{-1} is not needed
        __syncthreads();
        f32x2 a0 = {0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        const float4* gp = reinterpret_cast<const float4*>(dGs[cur] + L4_HP * j);
#pragma unroll
        for (int q = 0; q < D / 8; ++q) {
            const float4 g0 = gp[2 * q], g1 = gp[2 * q + 1];
            a0 = __builtin_elementwise_fma(f32x2{g0.x, g0.y}, wr[4 * q], a0);
            a1 = __builtin_elementwise_fma(f32x2{g0.z, g0.w}, wr[4 * q + 1], a1);
            a2 = __builtin_elementwise_fma(f32x2{g1.x, g1.y}, wr[4 * q + 2], a2);
            a3 = __builtin_elementwise_fma(f32x2{g1.z, g1.w}, wr[4 * q + 3], a3);
        }
        const f32x2 as = (a0 + a1) + (a2 + a3);
        float pr = as.x + as.y;                                            // rows 128 j .. + 127 ; the quad holds the four slices
        pr += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(pr), 0xB1, 0xF, 0xF, false));   // lanes (0,1) (2,3)
        pr += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(pr), 0x4E, 0xF, 0xF, false));   // pairs: same value in all four lanes
        dhr = pr;
    }
    if (t0 > 0 && j == 0) {
        carry[((size_t)bb * 2 + 0) * D + u] = dcn;
        carry[((size_t)bb * 2 + 1) * D + u] = dhr;
    }
}

static bool lstm_one_sample(int B) {
    static const bool on = !(getenv("VSL_LSTM1") && getenv("VSL_LSTM1")[0] == '0');
    return on && B <= 256;
}

void launch_lstm4_fwd(const float* gi, const float* Whh, const float* bih, const float* bhh, const float* mask, float* gates,
                      float* cseq, float* hprev, float* out, int B, int T, hipStream_t s, int t0, int t1) {
    if (lstm_one_sample(B)) {
        VSL_LAUNCH(k_lstm1_fwd, dim3(B), dim3(512), 0, s, gi, Whh, bih, bhh, mask, gates, cseq, hprev, out, T, t0, t1);
        return;
    }
    VSL_LAUNCH(k_lstm4_fwd, dim3((B + 3) / 4), dim3(512), 0, s, gi, Whh, bih, bhh, mask, gates, cseq, hprev, out, B, T, t0, t1);
}
void launch_lstm4_bwd(const float* dout, const float* dout2, const float* mask, const float* gates, const float* cseq,
                      const float* Whh, float* dG, int B, int T, hipStream_t s, float* carry, int t0, int t1) {
    if (lstm_one_sample(B)) {
        VSL_LAUNCH(k_lstm1_bwd, dim3(B), dim3(512), 0, s, dout, dout2, mask, gates, cseq, Whh, dG, T, carry, t0, t1);
        return;
    }
    VSL_LAUNCH(k_lstm4_bwd, dim3((B + 3) / 4), dim3(512), 0, s, dout, dout2, mask, gates, cseq, Whh, dG, B, T, carry, t0, t1);
}

}  // namespace vsl
