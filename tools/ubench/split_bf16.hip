// Round-3 micro-benchmark: is an fp32-grade GEMM on the bf16 matrix cores (3-way split, 6 products) worth building?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize split_bf16.hip -o split_bf16.bin && ./split_bf16.bin
// Every fp32 operand x is split EXACTLY into three bfloat16 terms x = h + m + l (round-to-nearest at each level: |m| <= 2^-8 |x|,
// |l| <= 2^-16 |x|) and the six products of weight >= 2^-16 (hh, hm, mh, hl, lh, mm) are issued on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation; the three dropped products are <= 2^-23 |a||b| -- the class of one fp32 rounding.
//  1. overlap: bf16 MFMA waves beside VALU waves on the same SIMD (waves 0-3 MFMA, 4-7 VALU) and both in ONE wave
//     (the fp32-input MFMAs of round 2 were ADDITIVE with vector work: valu_rates.hip);
//  2. cost of the split per element (v_cvt_pk_bf16_f32 + shift / and + subtract);
//  3. accuracy of dW = G^T A over R rows: fp32 MFMA chain, bf16x6, bf16x3 (hh, hm, mh) and bf16x1 against an fp64 host result;
//  4. throughput of the weight-gradient inner loop (R x 128 by R x 128 -> 128 x 128 per 256-row chunk, operands straight from global
//     memory in MFMA layout): fp32 32x32x2 (round-2 kernel) vs bf16x6 with 64 x 64 and 128 x 64 wave tiles.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {       // {bf16(a) in the low half, bf16(b) in the high half}, RNE
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));       // v_cvt_pk_bf16_f32
}
// x0, x1 -> packed (hi, mid, lo) pairs.  11 vector instructions per pair.
__device__ __forceinline__ void split3(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xFFFF0000u);
    m = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xFFFF0000u);
    l = cvt_pk_bf16(s0, s1);
}
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------------------
// 1. overlap.  mode 0: waves 0-3 MFMA ; 1: waves 4-7 VALU ; 2: both groups ; 3: every wave both (one MFMA + NV vector instructions)
// ---------------------------------------------------------------------------------------------------------------------------
#define N 4096
template <int NV>
__global__ __launch_bounds__(512) void k_overlap(long long* out, float* sink, int mode) {
    const int w = threadIdx.x >> 6;
    f32x16 acc0 = {0}, acc1 = {0};
    u32x4 a = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
    float fa = threadIdx.x * 1e-3f, fb = 1.f + fa, fc = fa + 1, fd = fb + 1;
    __syncthreads();
    long long t0 = clock64();
    const bool do_m = (mode == 0 || mode == 2) ? w < 4 : mode == 3;
    const bool do_v = (mode == 1 || mode == 2) ? w >= 4 : mode == 3;
    if (do_m && !do_v) {
        for (int i = 0; i < N; ++i) { acc0 = mfma_bf16(a, b, acc0); acc1 = mfma_bf16(b, a, acc1); }
    } else if (do_v && !do_m) {
        for (int i = 0; i < N; ++i) {
#pragma unroll
            for (int q = 0; q < 2 * NV / 4; ++q) { fa = fa * 1.0001f + fb; fb = fb * 0.9999f + fc; fc = fc * 1.0002f + fd; fd = fd * 0.9998f + fa; }
        }
    } else if (do_m && do_v) {
        for (int i = 0; i < N; ++i) {
            acc0 = mfma_bf16(a, b, acc0);
#pragma unroll
            for (int q = 0; q < NV / 4; ++q) { fa = fa * 1.0001f + fb; fb = fb * 0.9999f + fc; fc = fc * 1.0002f + fd; fd = fd * 0.9998f + fa; }
            acc1 = mfma_bf16(b, a, acc1);
#pragma unroll
            for (int q = 0; q < NV / 4; ++q) { fa = fa * 1.0001f + fb; fb = fb * 0.9999f + fc; fc = fc * 1.0002f + fd; fd = fd * 0.9998f + fa; }
        }
    }
    long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) out[w] = t1 - t0;
    sink[threadIdx.x] = acc0[0] + acc1[0] + fa + fb + fc + fd;
}

// 2. split cost: 8 pairs per iteration
__global__ void k_split(long long* out, uint32_t* sink) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = 1.0f + threadIdx.x * 1e-3f + i * 0.37f;
    uint32_t s = 0;
    long long t0 = clock64();
    for (int it = 0; it < N; ++it) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            uint32_t h, m, l;
            split3(x[2 * p], x[2 * p + 1], h, m, l);
            s += h ^ m ^ l;                                  // 3 more instructions per pair (v_xor3 + add): subtracted below
            x[2 * p] += 1e-3f; x[2 * p + 1] -= 1e-3f;        // 2 more
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------------------------------------
// 3 + 4. weight gradient dW[n][k] = sum_r G[r][n] A[r][k], G and A row-major R x 128.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int CH = 256;           // rows per chunk (workgroup)
// fp32 path = the round-2 kernel's loop (wave = 64 x 64 quadrant, float2 loads, 16 row pairs in flight)
__global__ __launch_bounds__(256, 2) void k_wg_f32(const float* __restrict__ G, const float* __restrict__ A, float* __restrict__ out, int R, int data_chunks) {
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int nh = wv & 1, kh = wv >> 1, ch = blockIdx.x;
    const float* Gp = G + 64 * nh + 2 * i;
    const float* Ap = A + 64 * kh + 2 * i;
    const int rbeg = (ch % data_chunks) * CH, rend = min(R, rbeg + CH);
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    constexpr int PF = 16;
    float2 gq[PF], aq[PF];
    auto ld = [&](int p, float2& g, float2& a) {
        const size_t row = (size_t)min(rbeg + 2 * p + h, rend - 1);
        g = *reinterpret_cast<const float2*>(Gp + row * 128);
        a = *reinterpret_cast<const float2*>(Ap + row * 128);
    };
#pragma unroll
    for (int q = 0; q < PF; ++q) ld(q, gq[q], aq[q]);
    for (int p0 = 0; p0 < (rend - rbeg) / 2; p0 += PF) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const float2 g = gq[q], a = aq[q];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g.x, a.x, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g.y, a.x, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g.x, a.y, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g.y, a.y, acc[1][1], 0, 0, 0);
            ld(p0 + q + PF, gq[q], aq[q]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float* o = out + ((size_t)ch * 128 + 64 * nh) * 128 + 64 * kh + 2 * i;
    for (int a = 0; a < 2; ++a)
        for (int r = 0; r < 16; ++r) {
            const int n = 2 * ((r & 3) + 8 * (r >> 2) + 4 * h) + a;
            *reinterpret_cast<float2*>(o + (size_t)n * 128) = make_float2(acc[a][0][r], acc[a][1][r]);
        }
}

template <int I0, int I1, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I0 < I1) { f(std::integral_constant<int, I0>()); static_for<I0 + 1, I1>(f); }
}
template <int W> struct Vec;
template <> struct Vec<2> { typedef float2 T; };
template <> struct Vec<4> { typedef float4 T; };
template <int W> __device__ __forceinline__ float vget(const typename Vec<W>::T& v, int c) { return reinterpret_cast<const float*>(&v)[c]; }

// bf16 split path.  Wave tile = (32 MB) x (32 NB) of the 128 x 128 block; NTW waves cover the block, the remaining factor RG = 4 / NTW
// splits the chunk's rows (row groups meet in LDS).  NP = products: 6 (fp32 grade), 3 (hh, hm, mh), 1 (hh).
// One step = 16 rows = one K slice of v_mfma_f32_32x32x16_bf16: lane (i, h) holds rows 8 h .. 8 h + 7 of its MB + NB columns.  The step's
// MFMAs (operands split in the previous step) are interleaved by hand with the split of the NEXT step's raw rows and with the loads of
// the step after: sched_barrier pins the order [MFMA, one pair's split (11 VALU)], ...; the loads are two steps ahead.
// DATA_ROWS: the benchmark reads chunk (blockIdx % (DATA_ROWS / 256)) so that the working set can be made cache resident.
template <int MB, int NB, int NP>
__global__ __launch_bounds__(256, 1) void k_wg_bf16(const float* __restrict__ G, const float* __restrict__ A, float* __restrict__ out, int R, int data_chunks) {
    constexpr int NTW = (128 / (32 * MB)) * (128 / (32 * NB));     // waves per 128 x 128 tile
    constexpr int RG = 4 / NTW;                                    // row groups
    constexpr int NPAIR = 4 * (MB + NB), NM = NP * MB * NB;
    extern __shared__ float red[];
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int tw = wv % NTW, rg = wv / NTW;
    const int nh = tw % (128 / (32 * MB)), kh = tw / (128 / (32 * MB)), ch = blockIdx.x;
    typedef typename Vec<MB>::T GV;
    typedef typename Vec<NB>::T AV;
    const int rows = CH / RG;
    const int rbeg = (ch % data_chunks) * CH + rg * rows, rend = min(R, rbeg + rows);
    const int ns = (rend - rbeg) / 16;                  // the benchmark uses whole steps only
    // wave-uniform base (SGPRs) + constant per-lane byte offset: the row of a load is an immediate offset
    const char* gbase = reinterpret_cast<const char*>(G + (size_t)rbeg * 128);
    const char* abase = reinterpret_cast<const char*>(A + (size_t)rbeg * 128);
    const uint32_t goff = (uint32_t)((8 * h) * 128 + 32 * MB * nh + MB * i) * 4u;
    const uint32_t aoff = (uint32_t)((8 * h) * 128 + 32 * NB * kh + NB * i) * 4u;
    f32x16 acc[MB][NB];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    struct Raw { float g[MB][8]; float a[NB][8]; };       // [column][row of the lane]
    struct Ops { u32x4 g[3][MB], a[3][NB]; };             // [term h / m / l][block], 4 dwords = 8 bf16 = the lane's 8 rows
    auto ld_pair = [&](int s, int p, Raw& x) {            // the two rows of pair p (pairs 0..3 of G: rows 2p, 2p+1 ; pairs 4..7: A)
        const int sc = min(s, ns - 1);                    // past the end: re-read the last step (never used)
        if (p < 4) {
#pragma unroll
            for (int q = 2 * p; q < 2 * p + 2; ++q) {
                const GV v = *reinterpret_cast<const GV*>(gbase + (size_t)sc * (16 * 512) + goff + q * 512);
#pragma unroll
                for (int c = 0; c < MB; ++c) x.g[c][q] = vget<MB>(v, c);
            }
        } else {
#pragma unroll
            for (int q = 2 * (p - 4); q < 2 * (p - 4) + 2; ++q) {
                const AV v = *reinterpret_cast<const AV*>(abase + (size_t)sc * (16 * 512) + aoff + q * 512);
#pragma unroll
                for (int c = 0; c < NB; ++c) x.a[c][q] = vget<NB>(v, c);
            }
        }
    };
    // split "pair" pi of the step: pi = rowpair * (MB + NB) + column: all columns of a row pair together, so that a raw row pair is dead
    // (and can be reloaded) as early as possible
    auto split_pair = [&](int pi, const Raw& x, Ops& o) {
        const int j = pi / (MB + NB), c = pi % (MB + NB);
        uint32_t hh, mm, ll;
        if (c < MB) { split3(x.g[c][2 * j], x.g[c][2 * j + 1], hh, mm, ll); o.g[0][c][j] = hh; o.g[1][c][j] = mm; o.g[2][c][j] = ll; }
        else { split3(x.a[c - MB][2 * j], x.a[c - MB][2 * j + 1], hh, mm, ll); o.a[0][c - MB][j] = hh; o.a[1][c - MB][j] = mm; o.a[2][c - MB][j] = ll; }
    };
    // MFMA m of a step: product type outer (small terms first), block inner -> consecutive MFMAs hit different accumulators
    auto mma1 = [&](int m, const Ops& o) {
        constexpr int TG[6] = {1, 0, 2, 0, 1, 0}, TA[6] = {1, 2, 0, 1, 0, 0};      // (g term, a term): mm, hl, lh, hm, mh, hh
        const int t = m / (MB * NB) + (6 - NP), ab = m % (MB * NB), a = ab / NB, b = ab % NB;
        acc[a][b] = mfma_bf16(o.g[TG[t]][a], o.a[TA[t]][b], acc[a][b]);
    };
    Raw x0, x1;
    Ops o0, o1;
    auto step = [&](int s, const Ops& cur, Ops& nxt, Raw& xs) {       // MFMAs of step s ; split of step s + 1 (raw in xs) ; loads of step s + 3 into xs
        static_for<0, NM>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            mma1(m, cur);
            constexpr int p0 = (m * NPAIR + NM - 1) / NM, p1 = ((m + 1) * NPAIR + NM - 1) / NM;      // pairs pi with pi * NM / NPAIR == m
            static_for<p0, p1>([&](auto pc) {
                constexpr int pi = decltype(pc)::value;
                split_pair(pi, xs, nxt);
                // after the last column of a row pair the pair's raw rows are dead: reload them for step s + 3
                if constexpr (pi % (MB + NB) == MB + NB - 1) { ld_pair(s + 3, pi / (MB + NB), xs); ld_pair(s + 3, 4 + pi / (MB + NB), xs); }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // prologue: raw rows of steps 0, 1 ; operands of step 0 ; raw rows of step 2 replace step 0's
    static_for<0, 8>([&](auto pc) { ld_pair(0, decltype(pc)::value, x0); ld_pair(1, decltype(pc)::value, x1); });
    static_for<0, NPAIR>([&](auto pc) { split_pair(decltype(pc)::value, x0, o0); });
    static_for<0, 8>([&](auto pc) { ld_pair(2, decltype(pc)::value, x0); });
    __builtin_amdgcn_sched_barrier(0);
    for (int s = 0; s < ns; s += 2) {
        step(s, o0, o1, x1);
        step(s + 1, o1, o0, x0);
    }
    // row groups meet in LDS (group 0 stores, the others add one after the other), group 0's layout = the output layout
    if (RG > 1) {
        for (int g = RG - 1; g >= 1; --g) {
            if (rg == g) {
#pragma unroll
                for (int a = 0; a < MB; ++a)
#pragma unroll
                    for (int b = 0; b < NB; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float* p = red + ((tw * MB * NB + a * NB + b) * 16 + r) * 64 + lane;
                            if (g == RG - 1) *p = acc[a][b][r]; else *p += acc[a][b][r];
                        }
            }
            __syncthreads();
        }
        if (rg == 0) {
#pragma unroll
            for (int a = 0; a < MB; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] += red[((tw * MB * NB + a * NB + b) * 16 + r) * 64 + lane];
        }
    }
    if (rg == 0) {
        float* o = out + ((size_t)ch * 128 + 32 * MB * nh) * 128 + 32 * NB * kh + NB * i;
#pragma unroll
        for (int a = 0; a < MB; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = MB * ((r & 3) + 8 * (r >> 2) + 4 * h) + a;
                AV v;
#pragma unroll
                for (int b = 0; b < NB; ++b) reinterpret_cast<float*>(&v)[b] = acc[a][b][r];
                *reinterpret_cast<AV*>(o + (size_t)n * 128) = v;
            }
    }
}

static double urand() { return (rand() + 0.5) / ((double)RAND_MAX + 1.0); }
static double nrand() { return sqrt(-2.0 * log(urand())) * cos(6.283185307179586 * urand()); }

int main() {
    long long *d, h[8]; float* fs; uint32_t* us;
    CHECK(hipFuncSetAttribute((const void*)k_wg_bf16<4, 2, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute((const void*)k_wg_bf16<4, 4, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipMalloc(&d, 64)); CHECK(hipMalloc(&fs, 4096)); CHECK(hipMalloc(&us, 4096));
    printf("== 1. overlap: v_mfma_f32_32x32x16_bf16 beside vector work (cycles per iteration; iteration = 2 MFMA and/or 2 NV fma)\n");
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_overlap<8>, dim3(1), dim3(512), 0, 0, d, fs, mode); CHECK(hipDeviceSynchronize()); }
        CHECK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost));
        printf("  NV=8  mode %d: wave0 %.1f wave4 %.1f\n", mode, (double)h[0] / N, (double)h[4] / N);
    }
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_overlap<4>, dim3(1), dim3(512), 0, 0, d, fs, 3); CHECK(hipDeviceSynchronize()); }
    CHECK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost)); printf("  NV=4  mode 3 (8 waves, each 2 x [MFMA + 4 fma]): wave0 %.1f\n", (double)h[0] / N);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_overlap<4>, dim3(1), dim3(256), 0, 0, d, fs, 3); CHECK(hipDeviceSynchronize()); }
    CHECK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost)); printf("  NV=4  mode 3 (4 waves, each 2 x [MFMA + 4 fma]): wave0 %.1f\n", (double)h[0] / N);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_overlap<8>, dim3(1), dim3(256), 0, 0, d, fs, 3); CHECK(hipDeviceSynchronize()); }
    CHECK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost)); printf("  NV=8  mode 3 (4 waves, each 2 x [MFMA + 8 fma]): wave0 %.1f\n", (double)h[0] / N);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_overlap<16>, dim3(1), dim3(256), 0, 0, d, fs, 3); CHECK(hipDeviceSynchronize()); }
    CHECK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost)); printf("  NV=16 mode 3 (4 waves, each 2 x [MFMA + 16 fma]): wave0 %.1f\n", (double)h[0] / N);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_overlap<16>, dim3(1), dim3(512), 0, 0, d, fs, 3); CHECK(hipDeviceSynchronize()); }
    CHECK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost)); printf("  NV=16 mode 3 (8 waves, each 2 x [MFMA + 16 fma]): wave0 %.1f\n", (double)h[0] / N);

    printf("== 2. 3-way split\n");
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, d, us); CHECK(hipDeviceSynchronize()); }
    CHECK(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost));
    printf("  %.1f cycles per pair of elements (11 split + 4 harness instructions), one wave alone\n", (double)h[0] / N / 8);

    printf("== 3. accuracy of dW = G^T A (128 x 128), rows R; err = max |dW - fp64| / max_nk sum_r |G||A|  (and / max |dW|)\n");
    for (int dist = 0; dist < 2; ++dist)
    for (int R : {256, 1024, 8192}) {
        std::vector<float> G((size_t)R * 128), A((size_t)R * 128);
        srand(1234 + R);
        for (auto& v : G) v = (float)(dist == 0 ? nrand() : nrand() * exp(4.0 * nrand()));      // dist 1: values spread over many binades
        for (auto& v : A) v = (float)(dist == 0 ? nrand() : nrand() * exp(4.0 * nrand()));
        const int nch = R / CH;
        float *dG, *dA, *dO;
        CHECK(hipMalloc(&dG, G.size() * 4)); CHECK(hipMalloc(&dA, A.size() * 4)); CHECK(hipMalloc(&dO, (size_t)nch * 128 * 128 * 4));
        CHECK(hipMemcpy(dG, G.data(), G.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
        std::vector<double> ref(128 * 128, 0.0), mag(128 * 128, 0.0);
        for (int r = 0; r < R; ++r)
            for (int n = 0; n < 128; ++n) {
                const double g = G[(size_t)r * 128 + n];
                for (int k = 0; k < 128; ++k) { const double p = g * A[(size_t)r * 128 + k]; ref[n * 128 + k] += p; mag[n * 128 + k] += fabs(p); }
            }
        double mmax = 0, rmax = 0;
        for (int e = 0; e < 128 * 128; ++e) { mmax = fmax(mmax, mag[e]); rmax = fmax(rmax, fabs(ref[e])); }
        std::vector<float> O((size_t)nch * 128 * 128);
        auto report = [&](const char* name) {
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost));
            double emax = 0, erel = 0;
            for (int e = 0; e < 128 * 128; ++e) {
                double s = 0;
                for (int c = 0; c < nch; ++c) s += O[(size_t)c * 128 * 128 + e];      // chunk partials added in fp64: isolates the in-chunk error
                emax = fmax(emax, fabs(s - ref[e]));
                erel = fmax(erel, fabs(s - ref[e]) / mag[e]);
            }
            printf("  dist %d R=%5d %-22s err/maxmag %.3e   max elementwise err/mag %.3e   err/max|dW| %.3e\n", dist, R, name, emax / mmax, erel, emax / rmax);
        };
        hipLaunchKernelGGL(k_wg_f32, dim3(nch), dim3(256), 0, 0, dG, dA, dO, R, nch); report("fp32 mfma 32x32x2");
        hipLaunchKernelGGL((k_wg_bf16<2, 2, 6>), dim3(nch), dim3(256), 0, 0, dG, dA, dO, R, nch); report("bf16x6 64x64");
        hipLaunchKernelGGL((k_wg_bf16<4, 2, 6>), dim3(nch), dim3(256), 65536, 0, dG, dA, dO, R, nch); report("bf16x6 128x64 (2 rg)");
        hipLaunchKernelGGL((k_wg_bf16<4, 4, 6>), dim3(nch), dim3(256), 65536, 0, dG, dA, dO, R, nch); report("bf16x6 128x128 (4 rg)");
        hipLaunchKernelGGL((k_wg_bf16<2, 2, 3>), dim3(nch), dim3(256), 0, 0, dG, dA, dO, R, nch); report("bf16x3 64x64");
        hipLaunchKernelGGL((k_wg_bf16<2, 2, 1>), dim3(nch), dim3(256), 0, 0, dG, dA, dO, R, nch); report("bf16x1 64x64");
        CHECK(hipFree(dG)); CHECK(hipFree(dA)); CHECK(hipFree(dO));
    }

    printf("== 4. throughput: 8 weight gradients of R = 8192 rows in one launch (256 workgroups), us per launch\n");
    for (int cached = 0; cached < 2; ++cached) {
        const int R = 8192 * 8, nch = R / CH, dch = cached ? 32 : nch;     // cached: every job reads the same 8192 rows (8 MB, L2 / MALL resident)
        printf("  -- operands %s\n", cached ? "cache resident (all jobs read the same 8192 rows)" : "streamed (67 MB per launch)");
        float *dG, *dA, *dO;
        CHECK(hipMalloc(&dG, (size_t)R * 128 * 4)); CHECK(hipMalloc(&dA, (size_t)R * 128 * 4)); CHECK(hipMalloc(&dO, (size_t)nch * 128 * 128 * 4));
        std::vector<float> G((size_t)R * 128);
        for (auto& v : G) v = (float)nrand();
        CHECK(hipMemcpy(dG, G.data(), G.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dA, G.data(), G.size() * 4, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        auto timeit = [&](const char* name, auto launch) {
            for (int i = 0; i < 3; ++i) launch();
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            for (int i = 0; i < 20; ++i) launch();
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1000.0 / 20;
            printf("  %-28s %7.2f us  (%.1f TFLOP/s fp32-equivalent)\n", name, us, 2.0 * R * 128 * 128 / us * 1e-6);
        };
        timeit("fp32 mfma (round 2)", [&] { hipLaunchKernelGGL(k_wg_f32, dim3(nch), dim3(256), 0, 0, dG, dA, dO, R, dch); });
        timeit("bf16x6 64x64 wave tiles", [&] { hipLaunchKernelGGL((k_wg_bf16<2, 2, 6>), dim3(nch), dim3(256), 0, 0, dG, dA, dO, R, dch); });
        timeit("bf16x6 128x64 wave tiles", [&] { hipLaunchKernelGGL((k_wg_bf16<4, 2, 6>), dim3(nch), dim3(256), 65536, 0, dG, dA, dO, R, dch); });
        timeit("bf16x6 128x128 wave tiles", [&] { hipLaunchKernelGGL((k_wg_bf16<4, 4, 6>), dim3(nch), dim3(256), 65536, 0, dG, dA, dO, R, dch); });
        timeit("bf16x3 64x64", [&] { hipLaunchKernelGGL((k_wg_bf16<2, 2, 3>), dim3(nch), dim3(256), 0, 0, dG, dA, dO, R, dch); });
        timeit("bf16x1 64x64", [&] { hipLaunchKernelGGL((k_wg_bf16<2, 2, 1>), dim3(nch), dim3(256), 0, 0, dG, dA, dO, R, dch); });
        CHECK(hipFree(dG)); CHECK(hipFree(dA)); CHECK(hipFree(dO));
    }
    return 0;
}
