// Stand-alone timing harness for the one-sample LSTM forward recurrence (k_lstm1_fwd of vslnet_amd/csrc/kernels_lstm.hip): the product
// kernel beside its round-3 form (k_var: one gate column per lane over the whole contraction) and the intermediate candidates (k_opt,
// k_opt2) with parts of the step knocked out, to see what a step is made of (profiles/r04_notes.md section 8).  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 lstm_harness.hip -o lstm_harness.bin && ./lstm_harness.bin
#include "../../vslnet_amd/csrc/kernels_lstm.hip"
#include <vector>
#include <math.h>
namespace vsl { void vsl_launch_events(hipStream_t, hipEvent_t* a, hipEvent_t* b) { *a = nullptr; *b = nullptr; } }
using namespace vsl;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int vzero() { int z; asm volatile("v_mov_b32 %0, 0" : "=v"(z)); return z; }      // a zero the compiler takes for lane-dependent
enum { K_NOFMA = 1, K_NOLDS = 2, K_CHEAPACT = 4, K_NOSTORE = 8, K_NOLOAD = 16, K_NOBARRIER = 32, K_RCP = 64, K_NOXCHG = 128, K_STAMP = 256 };

__device__ __forceinline__ float sig_rcp(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

template <int KNOB>
__global__ __launch_bounds__(512, 2) void k_var(const float* __restrict__ gi, const float* __restrict__ Whh,
                                                const float* __restrict__ bih, const float* __restrict__ bhh,
                                                const float* __restrict__ mask, float* __restrict__ gates,
                                                float* __restrict__ cseq, float* __restrict__ hprev, float* __restrict__ out,
                                                int T, int t0, int t1, long long* clk) {
    __shared__ __attribute__((aligned(16))) float hs[2][4 * L1_SEG];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b = lane >> 2, j = lane & 3;
    const int u = 16 * w + b;
    const int row = blockIdx.x * T;
    const int hoff = (u >> 5) * L1_SEG + (u & 31);
    const long long c0 = clock64(), w0 = wall_clock64();
    f32x2 wr[4][16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4* p = reinterpret_cast<const float4*>(Whh + (size_t)(g * D + u) * D + 32 * j);
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float4 v = p[q]; wr[g][2 * q] = f32x2{v.x, v.y}; wr[g][2 * q + 1] = f32x2{v.z, v.w}; }
    }
    const float bs = bih[j * D + u] + bhh[j * D + u];
    const float sc = j == 2 ? 2.0f : 1.0f;
    const bool j1 = j & 1, j2 = j & 2;
    float cst = 0.f;
    if (j == 0) hs[t0 & 1][hoff] = 0.f;
    float* gtp = gates + (size_t)(row + t0) * (4 * D) + j * D + u;
    float* qp = (j == 0 ? cseq : j == 1 ? out : hprev + D) + (size_t)(row + t0) * D + u;
    const float* gib = gi + (size_t)row * (4 * D) + j * D + u;
    const float* mkv = mask + row + vzero();
    float Gc[L1_NB], Mk[L1_NB], Gn[L1_NB], Mn[L1_NB];
    auto load_blk = [&](float (&G)[L1_NB], float (&M)[L1_NB], int tb) {
#pragma unroll
        for (int s = 0; s < L1_NB; ++s) {
            const int tt = min(tb + s, T - 1);
            if (KNOB & K_NOLOAD) { G[s] = 0.01f * tt; M[s] = 1.f; }
            else { G[s] = gib[(size_t)tt * (4 * D)]; M[s] = mkv[tt]; }
        }
    };
    load_blk(Gc, Mk, t0);
    __syncthreads();
    const long long c1 = clock64();
    for (int tb = t0; tb < t1; tb += L1_NB) {
        load_blk(Gn, Mn, tb + L1_NB);
#pragma unroll
        for (int s = 0; s < L1_NB; ++s) {
            const int t = tb + s;
            if (t >= t1) break;
            const int cur = t & 1;
            f32x2 a[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) a[g][0] = a[g][1] = f32x2{0.f, 0.f};
            if (t > 0) {
                const float4* hp = reinterpret_cast<const float4*>(hs[cur] + L1_SEG * j);
                float4 hv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) hv[q] = (KNOB & K_NOLDS) ? float4{cst, cst + q, cst, cst} : hp[q];
                if (KNOB & K_NOFMA) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) a[q & 3][0] += f32x2{hv[q].x + hv[q].z, hv[q].y + hv[q].w};
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            a[g][0] = __builtin_elementwise_fma(f32x2{hv[q].x, hv[q].y}, wr[g][2 * q], a[g][0]);
                            a[g][1] = __builtin_elementwise_fma(f32x2{hv[q].z, hv[q].w}, wr[g][2 * q + 1], a[g][1]);
                        }
                }
            }
            float p[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) { const f32x2 v = a[g][0] + a[g][1]; p[g] = v.x + v.y; }
            float z;
            if (KNOB & K_NOXCHG) z = p[0] + p[1] + p[2] + p[3] + Gc[s] + bs;
            else {
                const float r0 = (j1 ? p[1] : p[0]) + dpp_get<0xB1>(j1 ? p[0] : p[1]);
                const float r1 = (j1 ? p[3] : p[2]) + dpp_get<0xB1>(j1 ? p[2] : p[3]);
                z = (j2 ? r1 : r0) + dpp_get<0x4E>(j2 ? r0 : r1) + Gc[s] + bs;
            }
            float act, hn, cn;
            if (KNOB & K_CHEAPACT) {
                act = z * 0.25f;
                const float ig = quad_bcast<0>(act), fg = quad_bcast<1>(act), gg = quad_bcast<2>(act), og = quad_bcast<3>(act);
                cn = fg * cst + ig * gg;
                hn = og * cn;
            } else if (KNOB & K_RCP) {
                const float sg = sig_rcp(z * sc);
                act = j == 2 ? 2.0f * sg - 1.0f : sg;
                const float ig = quad_bcast<0>(act), fg = quad_bcast<1>(act), gg = quad_bcast<2>(act), og = quad_bcast<3>(act);
                cn = fg * cst + ig * gg;
                hn = og * (2.0f * sig_rcp(2.0f * cn) - 1.0f);
            } else {
                const float sg = sigmoid_fast(z * sc);
                act = j == 2 ? 2.0f * sg - 1.0f : sg;
                const float ig = quad_bcast<0>(act), fg = quad_bcast<1>(act), gg = quad_bcast<2>(act), og = quad_bcast<3>(act);
                cn = fg * cst + ig * gg;
                hn = og * tanh_fast(cn);
            }
            cst = cn;
            if (j == 0) hs[cur ^ 1][hoff] = hn;
            if (!(KNOB & K_NOSTORE)) {
                *gtp = act;
                const float qv = j == 0 ? cn : j == 1 ? hn * Mk[s] : hn;
                if (j < 2 || (j == 2 && t + 1 < T)) *qp = qv;
            }
            gtp += 4 * D;
            qp += D;
            if (KNOB & K_NOBARRIER) __builtin_amdgcn_s_waitcnt(0xc07f); else __syncthreads();      // lgkmcnt(0) only
        }
#pragma unroll
        for (int s = 0; s < L1_NB; ++s) { Gc[s] = Gn[s]; Mk[s] = Mn[s]; }
    }
    if (KNOB & K_NOSTORE) { if (cst == 123.456f) *gtp = cst; }
    const long long c2 = clock64(), w2 = wall_clock64();
    if (clk && blockIdx.x == 0 && tid == 0) { clk[0] = c1 - c0; clk[1] = c2 - c1; clk[2] = w2 - w0; }
}


// ---- candidate: gate-pair packed accumulators with rotated gates (no horizontal adds, no selects), v_rcp activations, stores deferred
// behind the next step's LDS reads, pointer-increment prefetch
template <int R> __device__ __forceinline__ float quad_rot(float x) {       // value of lane (j + R) & 3 of the quad
    constexpr int c = ((0 + R) & 3) | (((1 + R) & 3) << 2) | (((2 + R) & 3) << 4) | (((3 + R) & 3) << 6);
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), c, 0xF, 0xF, false));
}
#define STAMP(i) do { if ((KNOB & K_STAMP) && t == 64) { __builtin_amdgcn_sched_barrier(0); stamp[i] = clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
template <int KNOB>
__global__ __launch_bounds__(512, 2) void k_opt(const float* __restrict__ gi, const float* __restrict__ Whh,
                                                const float* __restrict__ bih, const float* __restrict__ bhh,
                                                const float* __restrict__ mask, float* __restrict__ gates,
                                                float* __restrict__ cseq, float* __restrict__ hprev, float* __restrict__ out,
                                                int T, int t0, int t1, long long* clk) {
    __shared__ __attribute__((aligned(16))) float hs[2][4 * L1_SEG];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b = lane >> 2, j = lane & 3;
    const int u = 16 * w + b;
    const int row = blockIdx.x * T;
    const int hoff = (u >> 5) * L1_SEG + (u & 31);
    const long long c0 = clock64(), w0 = wall_clock64();
    long long stamp[7] = {0, 0, 0, 0, 0, 0, 0};
    // register r of a lane = gate (r + j) & 3: the partial a lane needs from lane (j + d) & 3 is that lane's register (4 - d) & 3
    f32x2 w01[32], w23[32];
    {
        const float* p0 = Whh + (size_t)(((0 + j) & 3) * D + u) * D + 32 * j;
        const float* p1 = Whh + (size_t)(((1 + j) & 3) * D + u) * D + 32 * j;
        const float* p2 = Whh + (size_t)(((2 + j) & 3) * D + u) * D + 32 * j;
        const float* p3 = Whh + (size_t)(((3 + j) & 3) * D + u) * D + 32 * j;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 v0 = reinterpret_cast<const float4*>(p0)[q], v1 = reinterpret_cast<const float4*>(p1)[q];
            const float4 v2 = reinterpret_cast<const float4*>(p2)[q], v3 = reinterpret_cast<const float4*>(p3)[q];
            w01[4 * q] = f32x2{v0.x, v1.x}; w01[4 * q + 1] = f32x2{v0.y, v1.y}; w01[4 * q + 2] = f32x2{v0.z, v1.z}; w01[4 * q + 3] = f32x2{v0.w, v1.w};
            w23[4 * q] = f32x2{v2.x, v3.x}; w23[4 * q + 1] = f32x2{v2.y, v3.y}; w23[4 * q + 2] = f32x2{v2.z, v3.z}; w23[4 * q + 3] = f32x2{v2.w, v3.w};
        }
    }
    const float bs = bih[j * D + u] + bhh[j * D + u];
    const float nk = j == 2 ? -2.0f * 1.4426950408889634f : -1.4426950408889634f;    // exp(-sc z) = exp2(nk z)
    const float ma = j == 2 ? 2.0f : 1.0f, mb = j == 2 ? -1.0f : 0.0f;
    float cst = t0 > 0 ? cseq[(unsigned)((row + t0 - 1) * D + u)] : 0.f;
    if (j == 0) hs[t0 & 1][hoff] = t0 > 0 ? hprev[(unsigned)((row + t0) * D + u)] : 0.f;
    if (j == 3 && t0 == 0) hprev[(unsigned)(row * D + u)] = 0.f;
    float* gtp = gates + (size_t)(row + t0) * (4 * D) + j * D + u;
    float* qp = (j == 0 ? cseq : j == 1 ? out : hprev + D) + (size_t)(row + t0) * D + u;
    const float* gib = gi + (size_t)row * (4 * D) + j * D + u;
    const float* mkv = mask + row + vzero();
    float Gc[L1_NB], Mk[L1_NB], Gn[L1_NB], Mn[L1_NB];
    auto load_blk = [&](float (&G)[L1_NB], float (&M)[L1_NB], int tb) {
#pragma unroll
        for (int s = 0; s < L1_NB; ++s) {
            const int tt = min(tb + s, T - 1);
            G[s] = gib[(size_t)tt * (4 * D)] + bs;
            M[s] = mkv[tt];
        }
    };
    load_blk(Gc, Mk, t0);
#pragma unroll
    for (int s = 0; s < L1_NB; ++s) { Gn[s] = Gc[s]; Mn[s] = Mk[s]; }
    float st_act = 0.f, st_q = 0.f;
    bool st_q_on = false;
    __syncthreads();
    const long long c1 = clock64();
    for (int tb = t0; tb < t1; tb += L1_NB) {
#pragma unroll
        for (int s = 0; s < L1_NB; ++s) {
            const int t = tb + s;
            if (t >= t1) break;
            const int cur = t & 1;
            STAMP(0);
            f32x2 a01[2], a23[2];
            a01[0] = a01[1] = a23[0] = a23[1] = f32x2{0.f, 0.f};
            float4 hv[8];
            if (t > 0) {
                const float4* hp = reinterpret_cast<const float4*>(hs[cur] + L1_SEG * j);
#pragma unroll
                for (int q = 0; q < 8; ++q) hv[q] = (KNOB & K_NOLDS) ? float4{cst, cst, cst, cst} : hp[q];
            }
            STAMP(1);
            // behind the LDS reads: the previous step's stores, the next block's loads
            if (t > t0 && !(KNOB & K_NOSTORE)) {
                *gtp = st_act; gtp += 4 * D;
                if (st_q_on) *qp = st_q;
                qp += D;
            }
            if (s == 0 && !(KNOB & K_NOLOAD)) {
#pragma unroll
                for (int x = 0; x < L1_NB; ++x) { Gc[x] = Gn[x]; Mk[x] = Mn[x]; }
                load_blk(Gn, Mn, tb + L1_NB);
            }
            STAMP(2);
            if (t > 0 && !(KNOB & K_NOFMA)) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    a01[0] = __builtin_elementwise_fma(f32x2{hv[q].x, hv[q].x}, w01[4 * q], a01[0]);
                    a23[0] = __builtin_elementwise_fma(f32x2{hv[q].x, hv[q].x}, w23[4 * q], a23[0]);
                    a01[1] = __builtin_elementwise_fma(f32x2{hv[q].y, hv[q].y}, w01[4 * q + 1], a01[1]);
                    a23[1] = __builtin_elementwise_fma(f32x2{hv[q].y, hv[q].y}, w23[4 * q + 1], a23[1]);
                    a01[0] = __builtin_elementwise_fma(f32x2{hv[q].z, hv[q].z}, w01[4 * q + 2], a01[0]);
                    a23[0] = __builtin_elementwise_fma(f32x2{hv[q].z, hv[q].z}, w23[4 * q + 2], a23[0]);
                    a01[1] = __builtin_elementwise_fma(f32x2{hv[q].w, hv[q].w}, w01[4 * q + 3], a01[1]);
                    a23[1] = __builtin_elementwise_fma(f32x2{hv[q].w, hv[q].w}, w23[4 * q + 3], a23[1]);
                }
            }
            if (KNOB & K_NOFMA) { a01[0] = f32x2{hv[0].x + hv[1].y + hv[2].z + hv[3].w, hv[4].x + hv[5].y}; a23[0] = f32x2{hv[6].x, hv[7].y}; }
            STAMP(3);
            const f32x2 s01 = a01[0] + a01[1], s23 = a23[0] + a23[1];
            // lane j's gate: its own register 0 + register 3 of lane j + 1, 2 of lane j + 2, 1 of lane j + 3
            const float z = ((s01.x + Gc[s]) + quad_rot<1>(s23.y)) + (quad_rot<2>(s23.x) + quad_rot<3>(s01.y));
            const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * nk));
            const float act = sg * ma + mb;
            STAMP(4);
            const float ig = quad_bcast<0>(act), fg = quad_bcast<1>(act), gg = quad_bcast<2>(act), og = quad_bcast<3>(act);
            const float cn = fg * cst + ig * gg;
            const float th = 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(cn * (-2.0f * 1.4426950408889634f))) - 1.0f;
            const float hn = og * th;
            cst = cn;
            STAMP(5);
            if (j == 0) hs[cur ^ 1][hoff] = hn;
            st_act = act;
            st_q = j == 0 ? cn : j == 1 ? hn * Mk[s] : hn;
            st_q_on = j < 2 || (j == 2 && t + 1 < T);
            if (KNOB & K_NOBARRIER) __builtin_amdgcn_s_waitcnt(0xc07f); else __syncthreads();
            STAMP(6);
        }
    }
    if (t1 > t0) { *gtp = st_act; if (st_q_on) *qp = st_q; }
    const long long c2 = clock64(), w2 = wall_clock64();
    if (clk && blockIdx.x == 0 && tid == 0) { clk[0] = c1 - c0; clk[1] = c2 - c1; clk[2] = w2 - w0; }
    if ((KNOB & K_STAMP) && clk && blockIdx.x == 0 && tid == 0) for (int i = 0; i < 7; ++i) clk[8 + i] = stamp[i];
}

#undef STAMP
#define STAMP(i) do { if ((KNOB & K_STAMP) && t == 64) { __builtin_amdgcn_sched_barrier(0); stamp[i] = clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
template <int KNOB>
__global__ __launch_bounds__(512, 2) void k_opt2(const float* __restrict__ gi, const float* __restrict__ Whh,
                                                const float* __restrict__ bih, const float* __restrict__ bhh,
                                                const float* __restrict__ mask, float* __restrict__ gates,
                                                float* __restrict__ cseq, float* __restrict__ hprev, float* __restrict__ out,
                                                int T, int t0, int t1, long long* clk) {
    __shared__ __attribute__((aligned(16))) float hs[2][4 * L1_SEG];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b = lane >> 2, j = lane & 3;
    const int u = 16 * w + b;
    const int row = blockIdx.x * T;
    const int hoff = (u >> 5) * L1_SEG + (u & 31);
    const long long c0 = clock64(), w0 = wall_clock64();
    long long stamp[7] = {0, 0, 0, 0, 0, 0, 0};
    // register r of a lane = gate (r + j) & 3: the partial a lane needs from lane (j + d) & 3 is that lane's register (4 - d) & 3
    f32x2 w01[32], w23[32];
    {
        const float* p0 = Whh + (size_t)(((0 + j) & 3) * D + u) * D + 32 * j;
        const float* p1 = Whh + (size_t)(((1 + j) & 3) * D + u) * D + 32 * j;
        const float* p2 = Whh + (size_t)(((2 + j) & 3) * D + u) * D + 32 * j;
        const float* p3 = Whh + (size_t)(((3 + j) & 3) * D + u) * D + 32 * j;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 v0 = reinterpret_cast<const float4*>(p0)[q], v1 = reinterpret_cast<const float4*>(p1)[q];
            const float4 v2 = reinterpret_cast<const float4*>(p2)[q], v3 = reinterpret_cast<const float4*>(p3)[q];
            w01[4 * q] = f32x2{v0.x, v1.x}; w01[4 * q + 1] = f32x2{v0.y, v1.y}; w01[4 * q + 2] = f32x2{v0.z, v1.z}; w01[4 * q + 3] = f32x2{v0.w, v1.w};
            w23[4 * q] = f32x2{v2.x, v3.x}; w23[4 * q + 1] = f32x2{v2.y, v3.y}; w23[4 * q + 2] = f32x2{v2.z, v3.z}; w23[4 * q + 3] = f32x2{v2.w, v3.w};
        }
    }
    const float bs = bih[j * D + u] + bhh[j * D + u];
    const float nk = j == 2 ? -2.0f * 1.4426950408889634f : -1.4426950408889634f;    // exp(-sc z) = exp2(nk z)
    const float ma = j == 2 ? 2.0f : 1.0f, mb = j == 2 ? -1.0f : 0.0f;
    float cst = t0 > 0 ? cseq[(unsigned)((row + t0 - 1) * D + u)] : 0.f;
    if (j == 0) hs[t0 & 1][hoff] = t0 > 0 ? hprev[(unsigned)((row + t0) * D + u)] : 0.f;
    if (j == 3 && t0 == 0) hprev[(unsigned)(row * D + u)] = 0.f;
    float* gtp = gates + (size_t)(row + t0) * (4 * D) + j * D + u;
    float* qp = (j == 0 ? cseq : j == 1 ? out : hprev + D) + (size_t)(row + t0) * D + u;     // (lane 3 duplicates lane 2's store)
    const float* gib = gi + (size_t)row * (4 * D) + j * D + u;
    const float* mkv = mask + row + vzero();
    float Gc[L1_NB], Mk[L1_NB], Gn[L1_NB], Mn[L1_NB];
    auto load_blk = [&](float (&G)[L1_NB], float (&M)[L1_NB], int tb) {
#pragma unroll
        for (int s = 0; s < L1_NB; ++s) {
            const int tt = min(tb + s, T - 1);
            G[s] = gib[(size_t)tt * (4 * D)] + bs;
            M[s] = mkv[tt];
        }
    };
    load_blk(Gc, Mk, t0);
#pragma unroll
    for (int s = 0; s < L1_NB; ++s) { Gn[s] = Gc[s]; Mn[s] = Mk[s]; }
    float st_act = 0.f, st_q = 0.f;
    bool st_q_on = false;
    __syncthreads();
    const long long c1 = clock64();
    for (int tb = t0; tb < t1; tb += L1_NB) {
#pragma unroll
        for (int s = 0; s < L1_NB; ++s) {
            const int t = tb + s;
            if (t >= t1) break;
            const int cur = t & 1;
            STAMP(0);
            f32x2 a01[2], a23[2];
            a01[0] = a01[1] = a23[0] = a23[1] = f32x2{0.f, 0.f};
            float4 hv[8];
            {
                const float4* hp = reinterpret_cast<const float4*>(hs[cur] + L1_SEG * j);
#pragma unroll
                for (int q = 0; q < 8; ++q) hv[q] = (KNOB & K_NOLDS) ? float4{cst, cst, cst, cst} : hp[q];
            }
            STAMP(1);
            // behind the LDS reads: the previous step's stores, the next block's loads
            if (!(KNOB & K_NOSTORE)) {
                *gtp = st_act;
                *qp = st_q;
                const int adv = t > t0 ? 1 : 0;
                gtp += adv * 4 * D; qp += adv * D;
            }
            if (s == 0 && !(KNOB & K_NOLOAD)) {
#pragma unroll
                for (int x = 0; x < L1_NB; ++x) { Gc[x] = Gn[x]; Mk[x] = Mn[x]; }
                load_blk(Gn, Mn, tb + L1_NB);
            }
            STAMP(2);
            if (!(KNOB & K_NOFMA)) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    a01[0] = __builtin_elementwise_fma(f32x2{hv[q].x, hv[q].x}, w01[4 * q], a01[0]);
                    a23[0] = __builtin_elementwise_fma(f32x2{hv[q].x, hv[q].x}, w23[4 * q], a23[0]);
                    a01[1] = __builtin_elementwise_fma(f32x2{hv[q].y, hv[q].y}, w01[4 * q + 1], a01[1]);
                    a23[1] = __builtin_elementwise_fma(f32x2{hv[q].y, hv[q].y}, w23[4 * q + 1], a23[1]);
                    a01[0] = __builtin_elementwise_fma(f32x2{hv[q].z, hv[q].z}, w01[4 * q + 2], a01[0]);
                    a23[0] = __builtin_elementwise_fma(f32x2{hv[q].z, hv[q].z}, w23[4 * q + 2], a23[0]);
                    a01[1] = __builtin_elementwise_fma(f32x2{hv[q].w, hv[q].w}, w01[4 * q + 3], a01[1]);
                    a23[1] = __builtin_elementwise_fma(f32x2{hv[q].w, hv[q].w}, w23[4 * q + 3], a23[1]);
                }
            }
            if (KNOB & K_RCP) {      // (knob reused: interleave the step's memory instructions with the FMAs)
#pragma unroll
                for (int x = 0; x < 10; ++x) { __builtin_amdgcn_sched_group_barrier(0x002, 6, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0); }
            }
            if (KNOB & K_NOFMA) { a01[0] = f32x2{hv[0].x + hv[1].y + hv[2].z + hv[3].w, hv[4].x + hv[5].y}; a23[0] = f32x2{hv[6].x, hv[7].y}; }
            STAMP(3);
            const f32x2 s01 = a01[0] + a01[1], s23 = a23[0] + a23[1];
            // lane j's gate: its own register 0 + register 3 of lane j + 1, 2 of lane j + 2, 1 of lane j + 3
            const float z = ((s01.x + Gc[s]) + quad_rot<1>(s23.y)) + (quad_rot<2>(s23.x) + quad_rot<3>(s01.y));
            const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * nk));
            const float act = sg * ma + mb;
            STAMP(4);
            const float ig = quad_bcast<0>(act), fg = quad_bcast<1>(act), gg = quad_bcast<2>(act), og = quad_bcast<3>(act);
            const float cn = fg * cst + ig * gg;
            const float th = 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(cn * (-2.0f * 1.4426950408889634f))) - 1.0f;
            const float hn = og * th;
            cst = cn;
            STAMP(5);
            if (j == 0) hs[cur ^ 1][hoff] = hn;
            st_act = act;
            st_q = j == 0 ? cn : j == 1 ? hn * Mk[s] : hn;
            st_q_on = j < 2 || (j == 2 && t + 1 < T);
            if (KNOB & K_NOBARRIER) __builtin_amdgcn_s_waitcnt(0xc07f); else __syncthreads();
            STAMP(6);
        }
    }
    if (t1 > t0) { gtp += (t1 - t0 > 1 ? 0 : 0); *gtp = st_act; if (st_q_on) *qp = st_q; }
    const long long c2 = clock64(), w2 = wall_clock64();
    if (clk && blockIdx.x == 0 && tid == 0) { clk[0] = c1 - c0; clk[1] = c2 - c1; clk[2] = w2 - w0; }
    if ((KNOB & K_STAMP) && clk && blockIdx.x == 0 && tid == 0) for (int i = 0; i < 7; ++i) clk[8 + i] = stamp[i];
}

typedef void (*kern_t)(const float*, const float*, const float*, const float*, const float*, float*, float*, float*, float*, int, int, int, long long*);
static void run_k(kern_t k, const char* name, float** d, int B, int T, int t0, int t1, long long* dclk);
template <int KNOB> static void run(const char* name, float** d, int B, int T, int t0, int t1, long long* dclk) { run_k(k_var<KNOB>, name, d, B, T, t0, t1, dclk); }
static void run_k(kern_t k_, const char* name, float** d, int B, int T, int t0, int t1, long long* dclk) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int reps = 20;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_, dim3(B), dim3(512), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], T, t0, t1, dclk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_, dim3(B), dim3(512), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], T, t0, t1, dclk);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long h[3]; CHECK(hipMemcpy(h, dclk, sizeof(h), hipMemcpyDeviceToHost));
    const int n = t1 - t0;
    printf("%-34s %7.2f us/launch  %6.3f us/step | prologue %6lld clk, loop %7lld clk = %5.0f clk/step, shader clock %.2f GHz\n", name, ms * 1e3 / reps, ms * 1e3 / reps / n,
           h[0], h[1], (double)h[1] / n, (double)(h[0] + h[1]) / (h[2] * 10.0) );
}

int main() {
    const int B = 16, T = 128;
    std::vector<float> hgi((size_t)B * T * 4 * D), hw((size_t)4 * D * D), hb(4 * D), hm((size_t)B * T, 1.f);
    srand(1);
    for (auto& v : hgi) v = (rand() / (float)RAND_MAX - 0.5f);
    for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.17f;
    for (auto& v : hb) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    float* d[9];
    size_t sz[9] = {hgi.size(), hw.size(), hb.size(), hb.size(), hm.size(), hgi.size(), (size_t)B * T * D, (size_t)B * T * D + 4 * D, (size_t)B * T * D};
    for (int i = 0; i < 9; ++i) { CHECK(hipMalloc(&d[i], sz[i] * 4)); CHECK(hipMemset(d[i], 0, sz[i] * 4)); }
    CHECK(hipMemcpy(d[0], hgi.data(), hgi.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d[1], hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d[2], hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d[3], hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d[4], hm.data(), hm.size() * 4, hipMemcpyHostToDevice));
    long long* dclk; CHECK(hipMalloc(&dclk, 256));
    // the register-order image k_pack builds (PackJob type 9) and the tanh(c_t) tensor of the product kernel
    std::vector<float> himg((size_t)4 * D * D);
    for (int e = 0; e < 4 * D * D; ++e) {
        const int x = e & 3, ln = (e >> 2) & 63, q = (e >> 8) & 31, wv = e >> 13, u = 16 * wv + (ln >> 2), jj = ln & 3;
        himg[e] = hw[(size_t)((jj ^ x) * D + u) * D + 32 * jj + q];
    }
    float *dimg, *dts;
    CHECK(hipMalloc(&dimg, himg.size() * 4)); CHECK(hipMemcpy(dimg, himg.data(), himg.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dts, (size_t)B * T * D * 4));
    // the product kernel, whole sequence and one chunk
    {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int pass = 0; pass < 2; ++pass) {
            const int t1 = pass ? 43 : T;
            for (int i = 0; i < 3; ++i) launch_lstm_fwd(d[0], d[1], dimg, d[2], d[3], d[4], d[5], d[6], dts, d[7], d[8], B, T, 0, 0, t1);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) launch_lstm_fwd(d[0], d[1], dimg, d[2], d[3], d[4], d[5], d[6], dts, d[7], d[8], B, T, 0, 0, t1);
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("product k_lstm1_fwd steps [0,%d): %.2f us/launch, %.3f us/step\n", t1, ms * 1e3 / 20, ms * 1e3 / 20 / t1);
        }
    }
    run<0>("round-3 form (k_var)", d, B, T, 0, T, dclk);
    run<0>("  .. steps [0, 43)", d, B, T, 0, 43, dclk);
    {   // candidate against the product kernel: every saved tensor
        std::vector<float> ref[4], got[4];
        const int idx[4] = {5, 6, 7, 8};
        launch_lstm_fwd(d[0], d[1], dimg, d[2], d[3], d[4], d[5], d[6], dts, d[7], d[8], B, T, 0, 0, T);
        CHECK(hipDeviceSynchronize());
        for (int i = 0; i < 4; ++i) { ref[i].resize(sz[idx[i]]); CHECK(hipMemcpy(ref[i].data(), d[idx[i]], sz[idx[i]] * 4, hipMemcpyDeviceToHost)); CHECK(hipMemset(d[idx[i]], 0xff, sz[idx[i]] * 4)); }
        const int cuts[4] = {0, 43, 86, T};
        for (int c = 0; c < 3; ++c) hipLaunchKernelGGL(k_opt<0>, dim3(B), dim3(512), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], T, cuts[c], cuts[c + 1], (long long*)nullptr);
        CHECK(hipDeviceSynchronize());
        const char* nm[4] = {"gates", "cseq", "hprev", "out"};
        for (int i = 0; i < 4; ++i) {
            got[i].resize(sz[idx[i]]); CHECK(hipMemcpy(got[i].data(), d[idx[i]], sz[idx[i]] * 4, hipMemcpyDeviceToHost));
            double md = 0; size_t n = (size_t)B * T * (i == 0 ? 4 * D : D); int bad = 0;
            for (size_t e = 0; e < n; ++e) { const double dd = fabs((double)got[i][e] - ref[i][e]); if (!(dd <= 1e30)) ++bad; else if (dd > md) md = dd; }
            printf("k_opt (3 chunks) vs product: %-6s max |diff| %.3e  non-finite %d\n", nm[i], md, bad);
        }
    }
    run_k(k_opt<0>, "k_opt", d, B, T, 0, T, dclk);
    run_k(k_opt<0>, "k_opt steps [0, 43)", d, B, T, 0, 43, dclk);
    run_k(k_opt2<0>, "k_opt2 (branch-free step)", d, B, T, 0, T, dclk);
    run_k(k_opt2<K_RCP>, "k_opt2 + interleaved VMEM", d, B, T, 0, T, dclk);
    run_k(k_opt2<K_NOSTORE | K_NOLOAD>, "k_opt2 no loads, no stores", d, B, T, 0, T, dclk);
    run_k(k_opt2<K_NOBARRIER>, "k_opt2 no barrier (wrong)", d, B, T, 0, T, dclk);
    run_k(k_opt<K_NOFMA>, "k_opt no FMAs", d, B, T, 0, T, dclk);
    run_k(k_opt<K_NOLDS>, "k_opt no LDS reads", d, B, T, 0, T, dclk);
    run_k(k_opt<K_NOSTORE>, "k_opt no stores", d, B, T, 0, T, dclk);
    run_k(k_opt<K_NOLOAD>, "k_opt no loads", d, B, T, 0, T, dclk);
    run_k(k_opt<K_NOLOAD | K_NOSTORE>, "k_opt no loads, no stores", d, B, T, 0, T, dclk);
    run_k(k_opt<K_NOBARRIER>, "k_opt no barrier (wrong)", d, B, T, 0, T, dclk);
    run_k(k_opt<K_NOLOAD | K_NOSTORE | K_NOFMA | K_NOLDS>, "k_opt chain + exchange only", d, B, T, 0, T, dclk);
    {
        run_k(k_opt<K_STAMP>, "k_opt with stamps", d, B, T, 0, T, dclk);
        long long h[16]; CHECK(hipMemcpy(h, dclk, sizeof(h), hipMemcpyDeviceToHost));
        printf("  step 64, wave 0: top->reads issued %lld, ->stores/loads issued %lld, ->FMAs issued %lld, ->gate activated %lld, ->h ready %lld, ->past barrier %lld clk\n",
               h[9] - h[8], h[10] - h[9], h[11] - h[10], h[12] - h[11], h[13] - h[12], h[14] - h[13]);
    }
    run<K_NOFMA>("no FMAs", d, B, T, 0, T, dclk);
    run<K_NOLDS>("no LDS reads", d, B, T, 0, T, dclk);
    run<K_NOLDS | K_NOFMA>("no LDS reads, no FMAs", d, B, T, 0, T, dclk);
    run<K_CHEAPACT>("cheap activations", d, B, T, 0, T, dclk);
    run<K_RCP>("v_rcp_f32 activations", d, B, T, 0, T, dclk);
    run<K_NOSTORE>("no stores", d, B, T, 0, T, dclk);
    run<K_NOLOAD>("no loads", d, B, T, 0, T, dclk);
    run<K_NOSTORE | K_NOLOAD>("no loads, no stores", d, B, T, 0, T, dclk);
    run<K_NOBARRIER>("no barrier (wrong results)", d, B, T, 0, T, dclk);
    run<K_NOXCHG>("no quad exchange (wrong)", d, B, T, 0, T, dclk);
    run<K_NOSTORE | K_NOLOAD | K_RCP>("no loads, no stores, v_rcp", d, B, T, 0, T, dclk);
    run<K_NOSTORE | K_NOLOAD | K_CHEAPACT | K_NOFMA | K_NOLDS>("barrier + LDS write only", d, B, T, 0, T, dclk);
    return 0;
}
