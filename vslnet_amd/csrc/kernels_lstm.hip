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
// These 4-sample kernels serve B > 256 (and VSL_LSTM1=0); smaller batches run one sample per workgroup on the vector pipe (second half of this
// file), up to 80 samples as ONE launch per direction for the whole rnn head (k_rnn_fwd / k_rnn_bwd at the end).
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
// workgroup needs 64 packed FMAs per lane (v_pk_fma_f32) and spreads configs[0]'s B = 16 over 16 CUs.
//
// A step is bound by VALU ISSUE: two waves per SIMD, every instruction of either costs the SIMD 4 cycles (16 for exp / rcp), and the
// barrier keeps all eight waves in lockstep -- 2 x 64 FMAs = 512 cycles that cannot shrink, plus 8 cycles for every other instruction
// of the step, plus the LDS exchange (write, barrier, read latency: ~300 cycles in which nothing issues).  tools/ubench/lstm_harness.hip
// takes the step apart (profiles/r04_notes.md section 8).  So the step is built to need few instructions besides the FMAs:
//   * a lane owns a QUARTER of the contraction for FOUR outputs (8 ds_read_b128 of the shared operand per step instead of 32 for one
//     output over the whole contraction), accumulated as packed {even k, odd k} pairs: both operands of the 64 v_pk_fma_f32 are natural
//     register pairs, one add per output at the end;
//   * register r of a lane holds output (own ^ r): the partial that lane (own ^ d) needs is every lane's register d, so the
//     transposing reductions are plain DPP adds, no selects;
//   * the weights come from an image k_pack lays out in register order (PackJob types 9 / 10): 32 coalesced 16-byte loads per lane
//     (1.7 us per launch instead of 6.4 for strided rows) straight into the register pairs;
//   * sigmoid / tanh through v_exp_f32 + v_rcp_f32 (1 ulp) instead of an IEEE division; the forward saves tanh(c_t) for the backward
//     in the store slot of the quad's fourth lane, the backward's per-step factors are computed while the LDS reads are in flight;
//   * memory: inputs are prefetched a block of L1_NB steps ahead, the mask once per block (lane j of a quad holds step j's), a
//     step's stores are issued during the next step's LDS reads.
//   forward : quad (u = 16 w + b) = the four gates of unit u, lane j contracts k in [32 j, 32 j + 32); after the reduction lane j holds
//             gate j and activates it (tanh(x) = 2 sigmoid(2x) - 1), quad broadcasts hand all four to every lane, the cell update is
//             computed redundantly by the quad -- no LDS transposition.
//   backward: a DPP row (16 lanes = 4 quads) owns the columns of its four units; lane m of the row contracts gate rows [32 m, 32 m + 32)
//             into them; row_mirror / row_half_mirror reduce across the quads (quad i keeps column i), two quad adds finish it: every
//             lane of unit u's quad holds dh_{t-1}[u].  One LDS exchange and one barrier per step.
//   h and the gate gradients sit in LDS as 32-float segments on a 36-float stride (the quarter / sixteenth a lane reads).
//   (Tried: h through DPP row_newbcast into v_fmac_f32_dpp -- no LDS reads at all -- is no faster: the DPP FMAs issue at half rate;
//   profiles/r02_notes.md.)
// Used for B <= 256 (VSL_LSTM1=0 keeps the 4-sample kernels, which read W_hh itself); saved tensors, chunk / carry interface identical,
// plus tseq = tanh(c_t).
// =========================================================================================================
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int L1_SEG = 36;              // LDS stride of a 32-float segment
constexpr int L1_NB = 4;                // steps per block of prefetched inputs
constexpr int L1_RING = 8;              // granules in flight per lane of a projection workgroup (granule_ring_arrived counts on 8)
template <int I> __device__ __forceinline__ float quad_bcast(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), I * 0x55, 0xF, 0xF, true));      // (bound_ctrl: no lane of these patterns is out of bounds, and the
}                                                                                                         //  destination needs no zero ahead of the move)
template <int CTRL> __device__ __forceinline__ float dpp_get(float x) {      // 0xB1: lane ^ 1, 0x4E: lane ^ 2, 0x1B: lane ^ 3, 0x140: row mirror, 0x141: half-row mirror
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
constexpr float L1_LOG2E = 1.4426950408889634f;
// Ahead of a step loop: everything the prologue loaded (the 128 weight registers above all) has arrived.  Without it the compiler's waits for
// those registers sit INSIDE the loop, the last of them as vmcnt(0): every step then also waits for the load or write-through store it issued
// a moment ago -- a full memory round trip per step.
__device__ __forceinline__ void lstm1_prologue_done() { __builtin_amdgcn_s_waitcnt(0x0F70); }      // vmcnt(0)
__device__ __forceinline__ float sigmoid_rcp(float x, float nk) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * nk)); }    // nk = -log2(e): sigmoid(x)
// float4 q of the lane's image = W_r[4 (q & 7) .. + 3] for output r = q >> 3: two {k, k + 1} pairs
__device__ __forceinline__ void lstm1_weights(const float* __restrict__ img, int w, int lane, f32x2 (&wr)[4][16]) {
    const float4* p = reinterpret_cast<const float4*>(img) + (size_t)(w * 32) * 64 + lane;
#pragma unroll
    for (int q = 0; q < 32; ++q) { const float4 v = p[q * 64]; wr[q >> 3][2 * (q & 7)] = f32x2{v.x, v.y}; wr[q >> 3][2 * (q & 7) + 1] = f32x2{v.z, v.w}; }
}
// o[r] = sum over the lane's 32 contraction indices of x_k W_r[k]: {even k, odd k} halves accumulated as packed pairs (both operands are
// natural register pairs: no broadcasts, no moves), one add per output at the end
__device__ __forceinline__ void lstm1_product(const float4 (&x)[8], const f32x2 (&wr)[4][16], float (&o)[4]) {
    f32x2 a[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = f32x2{0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = __builtin_elementwise_fma(f32x2{x[q].x, x[q].y}, wr[r][2 * q], a[r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = __builtin_elementwise_fma(f32x2{x[q].z, x[q].w}, wr[r][2 * q + 1], a[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = a[r].x + a[r].y;
}

// ---------------------------------------------------------------------------------------------------------
// In-launch hand-offs between the workgroups of the fused rnn head (k_rnn_fwd / k_rnn_bwd below): 8-byte {tag, value} granules, each
// written by ONE write-through (sc1) store and re-read with sc1 loads until its tag is the launch's epoch -- the data is the flag, no
// fence on either side (MI355X_MICROARCH.md, handoff-1to1: ~1 us per hop; per-XCD L2s are not coherent, plain stores would not do).
// A producer never waits for a consumer and is dispatched before it (lower block index), so the spins end whatever the residency; they
// are bounded all the same: a lost producer ends in NaNs, not in a hung GPU.
// ---------------------------------------------------------------------------------------------------------
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
__device__ __forceinline__ void granule_store(u64* g, unsigned epoch, float v) {
    __hip_atomic_store((gu64*)g, ((u64)epoch << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// two neighbouring granules as ONE 16-byte write-through store (each half carries its own tag: readers keep reading 8 bytes).  An 8-byte sc1
// store is one fabric write each -- 512 per step from a projection workgroup.  Through a buffer descriptor, not inline asm: the compiler
// has to count the store, or every later wait for a polled granule also waits for write-through stores a few steps old (~2.4 us each).
typedef __amdgpu_buffer_rsrc_t grsrc_t;      // (the descriptor type of the buffer builtins)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ grsrc_t granule_rsrc(u64* base, unsigned bytes) {      // base, bytes: wave-uniform
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(base), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ void granule_store2(grsrc_t rsrc, unsigned byte_off, unsigned epoch, float v0, float v1) {
    const u32x4 d = {__float_as_uint(v0), epoch, __float_as_uint(v1), epoch};
    __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, byte_off, 0, /* sc1 */ 16);
}
// index of gate j, unit u in a (512)-granule row: gates 2 p, 2 p + 1 of a unit side by side
__device__ __forceinline__ int granule_gate_index(int j, int u) { return ((j >> 1) * D + u) * 2 + (j & 1); }
__device__ __forceinline__ u64 granule_load(const u64* g) { return __hip_atomic_load((gu64*)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
constexpr unsigned L1_SPIN_LIMIT = 1u << 20;
// every lane of the wave holds N granules: re-read until all tags carry the epoch (wave-uniform loop).  The first check is straight-line
// code: the compiler waits for exactly these loads there (the loop's wait is vmcnt(0), i.e. for every younger load and store as well).
template <int N> __device__ __forceinline__ bool granule_ok(const u64 (&r)[N], unsigned epoch) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < N; ++k) ok &= (unsigned)(r[k] >> 32) == epoch;
    return __all(ok);
}
// The re-poll is ONE inline-asm statement (load + wait): invisible to the compiler's wait-count bookkeeping, so the state it merges behind the
// branch is the fast path's and the next first check still waits for exactly its own load -- with ordinary loads in the loop every later
// check of the unrolled step loop got vmcnt(0).
__device__ __forceinline__ u64 granule_load_sync(const u64* g) {
    u64 r;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(g) : "memory");
    return r;
}
// A projection workgroup's ring of polled granules is read with loads the compiler does not see, waited for by hand: behind the branches of
// the unrolled step loop its own bookkeeping ends in vmcnt(0) ahead of every check, i.e. a wait for the ring load and the write-through store
// issued one step ago -- a memory round trip per step, the projection slower than the recurrence it feeds.  Between a ring load and its check
// eight steps later the wave issues 7 ring loads and 8 granule stores, nothing else: vmcnt(15) (memory operations retire in order).
__device__ __forceinline__ u64 granule_load_ring(const u64* g) {
    u64 r;
    asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(r) : "v"(g) : "memory");
    return r;
}
__device__ __forceinline__ void granule_ring_arrived(u64& r) { asm volatile("s_waitcnt vmcnt(15)" : "+v"(r)::"memory"); }
template <int N, typename F> __device__ __forceinline__ void granule_wait(u64 (&r)[N], F&& ptr, unsigned epoch) {     // ptr(k): address of granule k
    if (__builtin_expect(granule_ok(r, epoch), 1)) return;
    for (unsigned spins = 0; spins < L1_SPIN_LIMIT; ++spins) {
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int k = 0; k < N; ++k) r[k] = granule_load_sync(ptr(k));
        if (granule_ok(r, epoch)) return;
    }
#pragma unroll
    for (int k = 0; k < N; ++k) r[k] = 0x7fc00000ull;      // a lost producer: NaNs
}

enum { L1_PLAIN = 0, L1_PUBLISH = 1, L1_GRANULES = 2 };    // inputs from memory ; the same + the step's result published ; the step's input from granules

// One sample's forward recurrence over the steps [t0, t1).  Wimg: PackJob type 9 image of W_hh -- lane (u, j), float4 q =
// W_hh[(j ^ (q >> 3)) * 128 + u][32 j + 4 (q & 7) + x] for x = 0 .. 3.
//   L1_PUBLISH : h_t * mask is also published to gout[(b T + t) 128 + u]
//   L1_GRANULES: the input projection comes from gin[(b T + t) 512 + granule_gate_index(gate, u)] instead of gi
template <int MODE>
__device__ __forceinline__ void lstm1_fwd_body(float (&hs)[2][4 * L1_SEG], int bidx, const float* __restrict__ gi, const u64* gin,
                                               const float* __restrict__ Wimg, const float* __restrict__ bih,
                                               const float* __restrict__ bhh, const float* __restrict__ mask,
                                               float* __restrict__ gates, float* __restrict__ cseq, float* __restrict__ tseq,
                                               float* __restrict__ hprev, float* __restrict__ out, u64* gout, unsigned epoch, int T,
                                               int t0, int t1) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b = lane >> 2, j = lane & 3;
    const int u = 16 * w + b;
    const int row = bidx * T;
    const int hoff = (u >> 5) * L1_SEG + (u & 31);
    f32x2 wr[4][16];
    lstm1_weights(Wimg, w, lane, wr);
    const float bs = bih[j * D + u] + bhh[j * D + u];
    const float nk = j == 2 ? -2.0f * L1_LOG2E : -L1_LOG2E;                  // gate 2 is the tanh gate: 2 sigmoid(2 z) - 1
    const float ma = j == 2 ? 2.0f : 1.0f, mb = j == 2 ? -1.0f : 0.0f;
    float cst = t0 > 0 ? cseq[(unsigned)((row + t0 - 1) * D + u)] : 0.f;
    if (j == 0) hs[t0 & 1][hoff] = t0 > 0 ? hprev[(unsigned)((row + t0) * D + u)] : 0.f;     // (h_{-1} = 0: step 0's product is 0)
    if (j == 3 && t0 == 0) hprev[(unsigned)(row * D + u)] = 0.f;
    // per-lane cursors.  One store per step carries the lane's activated gate; a second one the quad's cell results (lane j = 0: c_t,
    // 1: masked h_t, 2: h_t as next step's hprev row, 3: tanh(c_t)).  Step t's stores are issued during step t + 1's LDS reads; the last
    // step's after the loop -- the only place where h_t of t = T - 1 (no hprev row) has to be held back.
    float* gtp = gates + (size_t)(row + t0) * (4 * D) + j * D + u;
    float* qp = (j == 0 ? cseq : j == 1 ? out : j == 2 ? hprev + D : tseq) + (size_t)(row + t0) * D + u;
    float* qfirst = j == 2 ? tseq + (size_t)(row + t0) * D + u : qp;      // (placeholder target of lane 2: h_t's hprev row may not exist, t0 + 1 = T)
    const float* gib = MODE == L1_GRANULES ? nullptr : gi + (size_t)row * (4 * D) + j * D + u;
    const u64* ginb = MODE == L1_GRANULES ? gin + (size_t)row * (4 * D) + granule_gate_index(j, u) : nullptr;
    u64* goutp = MODE == L1_PUBLISH ? gout + (size_t)(row + t0) * D + u : nullptr;
    const float* mkb = mask + row;
    float Gc[L1_NB], Gn[L1_NB], Mq[L1_NB], Mn;        // input projection + bias of the block's steps ; their masks in lane 1 of the quad, 1 elsewhere (loaded: lane j holds step j's)
    const bool j12 = j == 1 || j == 2;
    u64 Rn[L1_NB];
#pragma unroll
    for (int s = 0; s < L1_NB; ++s) Gn[s] = 0.f;
    int rtb = t0;                                     // block the granules in Rn belong to
    auto load_blk = [&](int tb) {
#pragma unroll
        for (int s = 0; s < L1_NB; ++s) {
            const size_t o = (size_t)min(tb + s, T - 1) * (4 * D);
            if (MODE == L1_GRANULES) Rn[s] = granule_load(ginb + o);
            else Gn[s] = gib[o];
        }
        rtb = tb;
        Mn = mkb[min(tb + j, T - 1)];
    };
    // (nothing is computed on a loaded value before the block is rotated in, four steps after its loads were issued)
    auto rotate = [&] {
#pragma unroll
        for (int s = 0; s < L1_NB; ++s) asm volatile("" : "+v"(Gn[s]));      // (keeps the compiler from waiting for the loads any earlier)
        asm volatile("" : "+v"(Mn));
        if (MODE == L1_GRANULES) {
            granule_wait(Rn, [&](int k) { return ginb + (size_t)min(rtb + k, T - 1) * (4 * D); }, epoch);
#pragma unroll
            for (int s = 0; s < L1_NB; ++s) Gc[s] = __uint_as_float((unsigned)Rn[s]) + bs;
        } else {
#pragma unroll
            for (int s = 0; s < L1_NB; ++s) Gc[s] = Gn[s] + bs;
        }
        const float m0 = quad_bcast<0>(Mn), m1 = quad_bcast<1>(Mn), m2 = quad_bcast<2>(Mn), m3 = quad_bcast<3>(Mn);      // (every lane runs the exchanges: a DPP read of a masked-off lane returns 0)
        Mq[0] = j == 1 ? m0 : 1.f; Mq[1] = j == 1 ? m1 : 1.f; Mq[2] = j == 1 ? m2 : 1.f; Mq[3] = j == 1 ? m3 : 1.f;
    };
    load_blk(t0);
    rotate();
    float st_act = 0.f, st_q = 0.f;
    lstm1_prologue_done();
    __syncthreads();
    for (int tb = t0; tb < t1; tb += L1_NB) {
#pragma unroll
        for (int s = 0; s < L1_NB; ++s) {
            const int t = tb + s;
            if (t >= t1) break;                           // uniform
            const int cur = t & 1;
            float4 hv[8];
            {
                const float4* hp = reinterpret_cast<const float4*>(hs[cur] + L1_SEG * j);
#pragma unroll
                for (int q = 0; q < 8; ++q) hv[q] = hp[q];
            }
            // the previous step's stores.  The launch's first step has none: it writes placeholders into its own rows, which the next
            // step overwrites (same lanes, same addresses, in order) -- so that BOTH sides of the branch issue two stores: a memory
            // instruction on one side only leaves the compiler's wait counts behind the join at their worst case, vmcnt(0)
            if (s == 0 && tb == t0) {                     // (uniform)
                *gtp = st_act;
                *qfirst = st_q;
            } else {
                *gtp = st_act; gtp += 4 * D;
                *qp = st_q; qp += D;
            }
            if (s == 0) load_blk(tb + L1_NB);             // the next block's inputs
            float o[4];
            lstm1_product(hv, wr, o);
            // lane j's gate: its register 0 + register d of lane j ^ d
            const float z = ((o[0] + Gc[s]) + dpp_get<0xB1>(o[1])) + (dpp_get<0x4E>(o[2]) + dpp_get<0x1B>(o[3]));
            const float act = sigmoid_rcp(z, nk) * ma + mb;
            const float ig = quad_bcast<0>(act), fg = quad_bcast<1>(act), gg = quad_bcast<2>(act), og = quad_bcast<3>(act);
            const float cn = fg * cst + ig * gg;
            const float th = 2.0f * sigmoid_rcp(cn, -2.0f * L1_LOG2E) - 1.0f;
            const float hn = og * th;
            cst = cn;
            if (j == 0) hs[cur ^ 1][hoff] = hn;
            if (MODE == L1_PUBLISH) {
                if (j == 1) granule_store(goutp, epoch, hn * Mq[s]);
                goutp += D;
            }
            st_act = act;
            st_q = j12 ? hn * Mq[s] : (j == 0 ? cn : th);
            __syncthreads();
        }
        rotate();
    }
    if (t1 > t0) {
        *gtp = st_act;
        if (j != 2 || t1 < T) *qp = st_q;
    }
}

__global__ __launch_bounds__(512, 2) void k_lstm1_fwd(const float* __restrict__ gi, const float* __restrict__ Wimg,
                                                      const float* __restrict__ bih, const float* __restrict__ bhh,
                                                      const float* __restrict__ mask, float* __restrict__ gates,
                                                      float* __restrict__ cseq, float* __restrict__ tseq, float* __restrict__ hprev,
                                                      float* __restrict__ out, int T, int t0, int t1) {
    __shared__ __attribute__((aligned(16))) float hs[2][4 * L1_SEG];
    lstm1_fwd_body<L1_PLAIN>(hs, blockIdx.x, gi, nullptr, Wimg, bih, bhh, mask, gates, cseq, tseq, hprev, out, nullptr, 0u, T, t0, t1);
}

// One sample's backward recurrence over the steps [t0, t1) in reverse.  Wimg: PackJob type 10 image of W_hh -- lane m = lane & 15 of row
// r4 = lane >> 4, quad i = (lane >> 2) & 3, float4 q = W_hh[32 m + 4 (q & 7) + x][16 w + 4 r4 + (i ^ (q >> 3))] for x = 0 .. 3.
//   L1_PUBLISH : the step's gate gradients are also published to gout[(b T + t) 512 + granule_gate_index(gate, u)]
//   L1_GRANULES: the second incoming gradient comes from gin[(b T + t) 128 + u] instead of dout2
template <int MODE>
__device__ __forceinline__ void lstm1_bwd_body(float (&dGs)[2][16 * L1_SEG], int bb, const float* __restrict__ dout,
                                               const float* __restrict__ dout2, const u64* gin, const float* __restrict__ mask,
                                               const float* __restrict__ gates, const float* __restrict__ cseq,
                                               const float* __restrict__ tseq, const float* __restrict__ Wimg, float* __restrict__ dG,
                                               u64* gout, unsigned epoch, int T, float* __restrict__ carry, int t0, int t1) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b = lane >> 2, j = lane & 3;
    const int u = 16 * w + b;                             // cell u ; this lane's gate j
    const int m = lane & 15;                              // product: gate rows 32 m .. + 31 into the row's four columns ; quad i keeps column i (= u)
    f32x2 wr[4][16];
    lstm1_weights(Wimg, w, lane, wr);
    const int goff = (4 * j + (u >> 5)) * L1_SEG + (u & 31);
    const bool jodd = j & 1;
    float dcn = 0.f, dhr = 0.f;
    if (t1 < T) { dcn = carry[((size_t)bb * 2 + 0) * D + u]; dhr = carry[((size_t)bb * 2 + 1) * D + u]; }
    // Everything a step reads from memory is independent of the recurrence: fetched in blocks of L1_NB steps, one block ahead.  Per step and
    // lane: the incoming gradient(s), the lane's activated gate, and cx = c_{t-1} in lane 1 of the quad (the forget gate's factor),
    // tanh(c_t) in the others; the mask once per block (lane j: step tb - j).
    struct In { float d, act, cx; };
    struct Raw { float d, d2, act, cx; };
    const float* cxp = j == 1 ? cseq - D : tseq;           // (lane 1 at t = 0 reads row 0 instead and drops it: c_{-1} = 0)
    In xc[L1_NB];
    Raw xn[L1_NB];
#pragma unroll
    for (int s = 0; s < L1_NB; ++s) xn[s].d2 = 0.f;
    u64 Rn[L1_NB];
    int rtb = t1 - 1;                                 // block the granules in Rn belong to
    float Mk[L1_NB], Mn;
    auto load_blk = [&](int tb) {                         // steps tb, tb - 1, ...
#pragma unroll
        for (int s = 0; s < L1_NB; ++s) {
            const int tt = max(tb - s, 0);
            const unsigned base = (unsigned)(bb * T + tt);
            xn[s].d = dout[base * D + u];
            if (MODE == L1_GRANULES) Rn[s] = granule_load(gin + (size_t)base * D + u);
            else xn[s].d2 = dout2 ? dout2[base * D + u] : 0.f;
            xn[s].act = gates[base * (4 * D) + j * D + u];
            xn[s].cx = cxp[(base + (j == 1 && tt == 0 ? 1u : 0u)) * D + u];
        }
        rtb = tb;
        Mn = mask[(unsigned)(bb * T + max(tb - j, 0))];
    };
    // (nothing is computed on a loaded value before the block is rotated in, four steps after its loads were issued)
    auto rotate = [&](int tb) {
#pragma unroll
        for (int s = 0; s < L1_NB; ++s) asm volatile("" : "+v"(xn[s].d), "+v"(xn[s].d2), "+v"(xn[s].act), "+v"(xn[s].cx));      // (keeps the compiler from waiting for the loads any earlier)
        asm volatile("" : "+v"(Mn));
        if (MODE == L1_GRANULES) granule_wait(Rn, [&](int k) { return gin + (size_t)(bb * T + max(rtb - k, 0)) * D + u; }, epoch);
#pragma unroll
        for (int s = 0; s < L1_NB; ++s) {
            xc[s].d = xn[s].d + (MODE == L1_GRANULES ? __uint_as_float((unsigned)Rn[s]) : xn[s].d2);
            xc[s].act = xn[s].act;
            xc[s].cx = (j == 1 && tb - s <= 0) ? 0.f : xn[s].cx;
        }
        Mk[0] = quad_bcast<0>(Mn); Mk[1] = quad_bcast<1>(Mn); Mk[2] = quad_bcast<2>(Mn); Mk[3] = quad_bcast<3>(Mn);
    };
    // step 0 ends the recurrence (dh_{-1} is not needed): it runs behind the loop, on inputs of its own
    float z_d = 0.f, z_act = 0.f, z_cx = 0.f, z_mk = 0.f;
    if (t0 == 0) {                                        // (uniform)
        const unsigned base = (unsigned)(bb * T);
        z_d = dout[base * D + u];
        if (MODE != L1_GRANULES && dout2) z_d += dout2[base * D + u];
        z_act = gates[base * (4 * D) + j * D + u];
        z_cx = j == 1 ? 0.f : tseq[base * D + u];
        z_mk = mask[base];
    }
    load_blk(t1 - 1);
    rotate(t1 - 1);
    lstm1_prologue_done();
    float* dgp = dG + (size_t)(bb * T + t1 - 1) * (4 * D) + j * D + u;
    const grsrc_t grs = granule_rsrc(MODE == L1_PUBLISH ? gout + (size_t)bb * T * (4 * D) : nullptr, (unsigned)T * 4 * D * 8);
    unsigned goff8 = ((unsigned)(t1 - 1) * 4 * D + granule_gate_index(j & 2, u)) * 8;
    float st_dv = 0.f;
    // one step up to the gate gradients: dc = dh k1 + dc_next ; dv = (gate 3: dh, else dc) kdv ; dc_next = dc f
    auto gate_grads = [&](float d, float mk, float act, float cx) {
        const float tc = quad_bcast<0>(cx), fg = quad_bcast<1>(act), og = quad_bcast<3>(act);
        const float k1 = og * (1.f - tc * tc);
        // kdv: gate 0 (i): g i (1 - i) ; 1 (f): c_{t-1} f (1 - f) ; 2 (g): i (1 - g^2) ; 3 (o): tanh(c_t) o (1 - o)
        const float sw = dpp_get<0xC6>(act);                              // quad_perm [2,1,0,3]: lanes 0 and 2 swap
        const float pf = jodd ? cx : sw;
        const float kdv = pf * (__builtin_fmaf(-act, act, j == 2 ? 1.f : act));
        // ---- the recurrence
        const float dh = d * mk + dhr;                // (dhr = 0 going into t = T - 1)
        const float dc = __builtin_fmaf(dh, k1, dcn);
        const float dv = (j == 3 ? dh : dc) * kdv;
        dcn = dc * fg;
        if (MODE == L1_PUBLISH) {
            const float dvn = dpp_get<0xB1>(dv);
            if (!(j & 1)) granule_store2(grs, goff8, epoch, dv, dvn);
            goff8 -= 4 * D * 8;
        }
        return dv;
    };
    const int tl = max(t0, 1);
    for (int tb = t1 - 1; tb >= tl; tb -= L1_NB) {
#pragma unroll
        for (int s = 0; s < L1_NB; ++s) {
            const int t = tb - s;
            if (t < tl) break;                            // uniform
            const int cur = t & 1;
            const float dv = gate_grads(xc[s].d, Mk[s], xc[s].act, xc[s].cx);
            dGs[cur][goff] = dv;
            __syncthreads();
            float4 gv[8];
            {
                const float4* gp = reinterpret_cast<const float4*>(dGs[cur] + L1_SEG * m);
#pragma unroll
                for (int q = 0; q < 8; ++q) gv[q] = gp[q];
            }
            // the previous step's gate gradients (the launch's first step writes a placeholder into its own row instead: every step issues
            // the same memory instructions, see lstm1_fwd_body)
            *dgp = st_dv;
            dgp -= (s == 0 && tb == t1 - 1) ? 0 : 4 * D;
            st_dv = dv;
            if (s == 0) load_blk(tb - L1_NB);
            float o[4];
            lstm1_product(gv, wr, o);
            // 16 lanes hold sixteenth sums of four columns, register r = column i ^ r: row mirror pairs quad i with i ^ 3, the half-row
            // mirror with i ^ 1; the two passes reach every quad once, the quad adds the rest.
            const float r0 = o[0] + dpp_get<0x140>(o[3]), r1 = o[1] + dpp_get<0x140>(o[2]);
            float pr = r0 + dpp_get<0x141>(r1);
            pr += dpp_get<0xB1>(pr);
            pr += dpp_get<0x4E>(pr);                      // the same value in the quad's four lanes
            dhr = pr;
        }
        rotate(tb - L1_NB);
    }
    if (t0 == 0) {                                        // (uniform)
        if (MODE == L1_GRANULES) {
            u64 r[1] = {granule_load_sync(gin + (size_t)(bb * T) * D + u)};
            granule_wait(r, [&](int) { return gin + (size_t)(bb * T) * D + u; }, epoch);
            z_d += __uint_as_float((unsigned)r[0]);
        }
        const float dv = gate_grads(z_d, z_mk, z_act, z_cx);
        if (t1 - 1 >= tl) { *dgp = st_dv; dgp -= 4 * D; }
        st_dv = dv;
    }
    if (t1 > t0) *dgp = st_dv;
    if (t0 > 0 && j == 0) {
        carry[((size_t)bb * 2 + 0) * D + u] = dcn;
        carry[((size_t)bb * 2 + 1) * D + u] = dhr;
    }
}

__global__ __launch_bounds__(512, 2) void k_lstm1_bwd(const float* __restrict__ dout, const float* __restrict__ dout2,
                                                      const float* __restrict__ mask, const float* __restrict__ gates,
                                                      const float* __restrict__ cseq, const float* __restrict__ tseq,
                                                      const float* __restrict__ Wimg, float* __restrict__ dG, int T,
                                                      float* __restrict__ carry, int t0, int t1) {
    __shared__ __attribute__((aligned(16))) float dGs[2][16 * L1_SEG];    // gate gradients of the step (row k = 128 gate + unit), double-buffered: one barrier per step
    lstm1_bwd_body<L1_PLAIN>(dGs, blockIdx.x, dout, dout2, nullptr, mask, gates, cseq, tseq, Wimg, dG, nullptr, 0u, T, carry, t0, t1);
}

// =========================================================================================================
// The rnn head as ONE launch per direction (B <= RNN_FUSED_MAX_B).  The end LSTM consumes the start LSTM's output through a GEMM
// (x W_ih^T), step by step: as separate launches the two recurrences either run one after the other (2 x 80 us at T = 128) or are
// pipelined in time chunks over three streams, where every cross-stream hop costs 10 us and eats what the overlap gains
// (profiles/r04_notes.md section 8).  Here three workgroups per sample form a dataflow pipeline through granules:
//   forward : [start LSTM, publishes h_t * mask] -> [projection: W_ih h_t, 512 outputs per step] -> [end LSTM]
//   backward: [end LSTM backward, publishes its gate gradients] -> [projection: dG W_ih, 128 outputs] -> [start LSTM backward]
// A projection workgroup holds W_ih in the same register images as the recurrences hold W_hh (PackJob types 9 / 10 of W_ih) and has
// no recurrence of its own: it keeps up with its producer and adds ~2 us of lag.  Block order = pipeline order (see granule_wait).
// The fp32 FMAs replace the bf16x6 GEMM of the chunked path: the same fp32-grade product.
// =========================================================================================================
__device__ __forceinline__ void lstm1_proj_fwd(float (&hs)[2][4 * L1_SEG], int bidx, const u64* gin, const float* __restrict__ Wimg, u64* gout,
                                               unsigned epoch, int T) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b = lane >> 2, j = lane & 3;
    const int u = 16 * w + b;
    f32x2 wr[4][16];
    lstm1_weights(Wimg, w, lane, wr);
    // waves 0, 1: lane tid waits for unit tid's granule of the step and puts it into the LDS vector.  The granules of the next L1_RING steps
    // are always in flight: a granule another XCD has just written through is ~2.4 us away, and the projection has to run faster than the
    // recurrence it follows (0.4 against 0.57 us per step) to catch up whenever it lags -- one outstanding poll makes a step cost the whole
    // round trip, four 0.6 us.
    const int hoff = (tid >> 5) * L1_SEG + (tid & 31);
    const u64* rp = gin + (size_t)bidx * T * D + (tid & (D - 1));
    u64 ring[L1_RING];
    if (tid < D) {
#pragma unroll
        for (int k = 0; k < L1_RING; ++k) ring[k] = granule_load_ring(rp + (size_t)min(k, T - 1) * D);
    }
    const grsrc_t grs = granule_rsrc(gout + (size_t)bidx * T * (4 * D), (unsigned)T * 4 * D * 8);
    unsigned goff8 = granule_gate_index(j & 2, u) * 8;
    lstm1_prologue_done();
    for (int tb = 0; tb < T; tb += L1_RING) {
#pragma unroll
        for (int k = 0; k < L1_RING; ++k) {
            const int t = tb + k;
            if (t >= T) break;                            // uniform
            const int cur = t & 1;
            if (tid < D) {                                // (wave-uniform)
                granule_ring_arrived(ring[k]);
                u64 r[1] = {ring[k]};
                granule_wait(r, [&](int) { return rp + (size_t)t * D; }, epoch);
                hs[cur][hoff] = __uint_as_float((unsigned)r[0]);
                ring[k] = granule_load_ring(rp + (size_t)min(t + L1_RING, T - 1) * D);
            }
            __syncthreads();
            float4 hv[8];
            {
                const float4* hp = reinterpret_cast<const float4*>(hs[cur] + L1_SEG * j);
#pragma unroll
                for (int q = 0; q < 8; ++q) hv[q] = hp[q];
            }
            float o[4];
            lstm1_product(hv, wr, o);
            const float z = (o[0] + dpp_get<0xB1>(o[1])) + (dpp_get<0x4E>(o[2]) + dpp_get<0x1B>(o[3]));
            const float zn = dpp_get<0xB1>(z);
            if (!(j & 1)) granule_store2(grs, goff8, epoch, z, zn);
            goff8 += 4 * D * 8;
        }
    }
}

__device__ __forceinline__ void lstm1_proj_bwd(float (&dGs)[2][16 * L1_SEG], int bb, const u64* gin, const float* __restrict__ Wimg, u64* gout,
                                               unsigned epoch, int T) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b = lane >> 2, j = lane & 3;
    const int u = 16 * w + b;
    const int m = lane & 15;
    f32x2 wr[4][16];
    lstm1_weights(Wimg, w, lane, wr);
    const int goff = (4 * j + (u >> 5)) * L1_SEG + (u & 31);
    const u64* rp = gin + (size_t)bb * T * (4 * D) + granule_gate_index(j, u);
    u64 ring[L1_RING];                                       // the granules of the next L1_RING steps are always in flight (see lstm1_proj_fwd)
#pragma unroll
    for (int k = 0; k < L1_RING; ++k) ring[k] = granule_load_ring(rp + (size_t)max(T - 1 - k, 0) * (4 * D));
    u64* gp = gout + (size_t)(bb * T + T - 1) * D + u;
    lstm1_prologue_done();
    for (int tb = T - 1; tb >= 0; tb -= L1_RING) {
#pragma unroll
        for (int k = 0; k < L1_RING; ++k) {
            const int t = tb - k;
            if (t < 0) break;                             // uniform
            const int cur = t & 1;
            granule_ring_arrived(ring[k]);
            u64 r[1] = {ring[k]};
            granule_wait(r, [&](int) { return rp + (size_t)t * (4 * D); }, epoch);
            dGs[cur][goff] = __uint_as_float((unsigned)r[0]);
            ring[k] = granule_load_ring(rp + (size_t)max(t - L1_RING, 0) * (4 * D));
            __syncthreads();
            float4 gv[8];
            {
                const float4* gq = reinterpret_cast<const float4*>(dGs[cur] + L1_SEG * m);
#pragma unroll
                for (int q = 0; q < 8; ++q) gv[q] = gq[q];
            }
            float o[4];
            lstm1_product(gv, wr, o);
            const float r0 = o[0] + dpp_get<0x140>(o[3]), r1 = o[1] + dpp_get<0x140>(o[2]);
            float pr = r0 + dpp_get<0x141>(r1);
            pr += dpp_get<0xB1>(pr);
            pr += dpp_get<0x4E>(pr);
            if (j == 0) granule_store(gp, epoch, pr);
            gp -= D;
        }
    }
}

__global__ __launch_bounds__(512, 2) void k_rnn_fwd(RnnFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float hs[2][4 * L1_SEG];
    const int B = a.B;
    if ((int)blockIdx.x < B)
        lstm1_fwd_body<L1_PUBLISH>(hs, blockIdx.x, a.gi0, nullptr, a.Whh[0], a.bih[0], a.bhh[0], a.mask, a.gates[0], a.cseq[0], a.tseq[0], a.hprev[0],
                                   a.out[0], a.h_gran, a.epoch, a.T, 0, a.T);
    else if ((int)blockIdx.x < 2 * B)
        lstm1_proj_fwd(hs, blockIdx.x - B, a.h_gran, a.Wih1, a.gi_gran, a.epoch, a.T);
    else
        lstm1_fwd_body<L1_GRANULES>(hs, blockIdx.x - 2 * B, nullptr, a.gi_gran, a.Whh[1], a.bih[1], a.bhh[1], a.mask, a.gates[1], a.cseq[1], a.tseq[1],
                                    a.hprev[1], a.out[1], nullptr, a.epoch, a.T, 0, a.T);
}

__global__ __launch_bounds__(512, 2) void k_rnn_bwd(RnnBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float dGs[2][16 * L1_SEG];
    const int B = a.B;
    if ((int)blockIdx.x < B)
        lstm1_bwd_body<L1_PUBLISH>(dGs, blockIdx.x, a.dout[1], nullptr, nullptr, a.mask, a.gates[1], a.cseq[1], a.tseq[1], a.Whh[1], a.dG[1], a.dg_gran,
                                   a.epoch, a.T, nullptr, 0, a.T);
    else if ((int)blockIdx.x < 2 * B)
        lstm1_proj_bwd(dGs, blockIdx.x - B, a.dg_gran, a.Wih1, a.dx_gran, a.epoch, a.T);
    else
        lstm1_bwd_body<L1_GRANULES>(dGs, blockIdx.x - 2 * B, a.dout[0], nullptr, a.dx_gran, a.mask, a.gates[0], a.cseq[0], a.tseq[0], a.Whh[0], a.dG[0],
                                    nullptr, a.epoch, a.T, nullptr, 0, a.T);
}

static bool lstm_one_sample(int B) {
    static const bool on = !(getenv("VSL_LSTM1") && getenv("VSL_LSTM1")[0] == '0');
    return on && B <= 256;
}

bool rnn_fused_ok(int B) {            // VSL_RNN_FUSED=0 keeps the chunked launches (the path of 80 < B <= 256) at every batch size
    static const bool on = !(getenv("VSL_RNN_FUSED") && getenv("VSL_RNN_FUSED")[0] == '0');
    return on && lstm_one_sample(B) && B <= RNN_FUSED_MAX_B;
}
void launch_rnn_fwd(const RnnFwdArgs& a, hipStream_t s) { VSL_LAUNCH(k_rnn_fwd, dim3(3 * a.B), dim3(512), 0, s, a); }
void launch_rnn_bwd(const RnnBwdArgs& a, hipStream_t s) { VSL_LAUNCH(k_rnn_bwd, dim3(3 * a.B), dim3(512), 0, s, a); }
void launch_lstm_fwd(const float* gi, const float* Whh, const float* Wimg, const float* bih, const float* bhh, const float* mask, float* gates,
                      float* cseq, float* tseq, float* hprev, float* out, int B, int T, hipStream_t s, int t0, int t1) {
    if (t1 < 0) t1 = T;
    if (lstm_one_sample(B)) {
        VSL_LAUNCH(k_lstm1_fwd, dim3(B), dim3(512), 0, s, gi, Wimg, bih, bhh, mask, gates, cseq, tseq, hprev, out, T, t0, t1);
        return;
    }
    VSL_LAUNCH(k_lstm4_fwd, dim3((B + 3) / 4), dim3(512), 0, s, gi, Whh, bih, bhh, mask, gates, cseq, hprev, out, B, T, t0, t1);
}
void launch_lstm_bwd(const float* dout, const float* dout2, const float* mask, const float* gates, const float* cseq, const float* tseq,
                      const float* Whh, const float* Wimg, float* dG, int B, int T, hipStream_t s, float* carry, int t0, int t1) {
    if (t1 < 0) t1 = T;
    if (lstm_one_sample(B)) {
        VSL_LAUNCH(k_lstm1_bwd, dim3(B), dim3(512), 0, s, dout, dout2, mask, gates, cseq, tseq, Wimg, dG, T, carry, t0, t1);
        return;
    }
    VSL_LAUNCH(k_lstm4_bwd, dim3((B + 3) / 4), dim3(512), 0, s, dout, dout2, mask, gates, cseq, Whh, dG, B, T, carry, t0, t1);
}

}  // namespace vsl
