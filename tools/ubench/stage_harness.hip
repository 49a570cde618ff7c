// Round-6 micro-benchmark (verdict r5 item 2): ONE conv-block GEMM stage -- rows x 128 times 128 x 128 as six bf16 split products, activations from
// pre-split LDS planes, weight slices streamed from L2 -- as a chain of NREP dependent stages (each stage's result is split back into the planes
// the next stage reads, like a layer of the conv block), timed per stage in shader cycles, with an fp64 check of the first stage.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/ubench/stage_harness.hip -o /tmp/stage_harness && /tmp/stage_harness
// Geometry = the "T layout" of kernels_query.hip (tgemm<8>: the WEIGHT is the MFMA's A operand, wave = 32 output channels, a 32-row window is the
// N dimension, the next stage's weight fragments stream into a register ring while the current stage multiplies):
//   variant 1: 4 waves = 4 channel groups x 1 row group  (32 rows: k_query_fwd's stage, one wave per SIMD)
//   variant 2: 8 waves = 4 channel groups x 2 row groups (64 rows >= the conv block's 56-row window, two waves per SIMD: one wave's operand reads
//              and weight loads under the other's MFMAs -- the 2 x 4 layout the verdict asks for; each wave reads only its row group's planes)
// and for each: the bare product chain (MFMA + operand reads + weight stream) and the chain with the split-store epilogue + barrier.
#include "../../vslnet_amd/csrc/kernels_query.hip"
#include <vector>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
namespace vsl { void vsl_launch_events(hipStream_t, hipEvent_t* a, hipEvent_t* b) { *a = nullptr; *b = nullptr; } }
using namespace vsl;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int NREP = 16;
// W3: [NREP] split packs of a 128 x 128 weight ; X0: (rows, 128) fp32 input ; Y: (rows, 128) result of stage 0 (check) ; cyc: [grid][2] stamps
template <int RG, bool EPI, bool WSTREAM>
__global__ __launch_bounds__(256 * RG) void k_stage(const uint16_t* __restrict__ W3, const float* __restrict__ X0, float* __restrict__ Y,
                                                   long long* __restrict__ cyc, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) uint16_t planes[];       // [RG][3 planes][32][QLB]
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, m = lane & 31, h = lane >> 5;
    const int w = wv & 3, rg = wv >> 2;                                      // channel group, row group
    uint16_t* P = planes + rg * QPLANES;
    const size_t plane = pack3_plane(D, D), wsz = 3 * plane;
    // stage the input: accumulator layout -> planes
    f32x16 x;
    global2d(x, X0 + (size_t)rg * 32 * D, w, m, h, 32);
    d2planes(x, P, w, m, h);
    WRing ring;
    wprefetch(ring, wnext(W3, plane, D, 32 * w, 0, 8));
    __syncthreads();
    const long long t0 = clock64();
    f32x16 acc;
    float keep = 0.f;
    for (int rep = 0; rep < NREP; ++rep) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const uint16_t* wn = W3 + (size_t)((rep + 1) % NREP) * wsz;
        tgemm<8>(P, ring, acc, wnext(WSTREAM ? wn : nullptr, plane, D, 32 * w, 0, 8));       // (no stream: every stage multiplies by the ring's stage-0 weights)
        if (rep == 0) d2global(acc, Y + (size_t)rg * 32 * D, w, m, h, 32);
        if (EPI) {
            __syncthreads();                                                 // every wave has read the planes
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f) * 0.125f;   // (a stand-in epilogue: ReLU + scale keeps the chain bounded)
            d2planes(acc, P, w, m, h);
            __syncthreads();
        } else keep += acc[rep & 15];
    }
    const long long t1 = clock64();
    if (tid == 0) { cyc[2 * blockIdx.x] = t0; cyc[2 * blockIdx.x + 1] = t1; }
    if (!EPI) sink[blockIdx.x * blockDim.x + tid] = keep;
}

static uint16_t bf16_rne(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf16_f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

template <int RG, bool EPI, bool WSTREAM = true>
static void run(const char* name, const uint16_t* dW, const float* dX, float* dY, long long* dC, float* dS, const std::vector<float>& Wf,
                const std::vector<float>& Xf, int grid) {
    const size_t shm = (size_t)RG * QPLANES * sizeof(uint16_t);
    CHECK(hipFuncSetAttribute((const void*)k_stage<RG, EPI, WSTREAM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k_stage<RG, EPI, WSTREAM>), dim3(grid), dim3(256 * RG), shm, 0, dW, dX, dY, dC, dS);
    CHECK(hipDeviceSynchronize());
    std::vector<long long> c(2 * grid);
    CHECK(hipMemcpy(c.data(), dC, c.size() * sizeof(long long), hipMemcpyDeviceToHost));
    double sum = 0, mx = 0, mn = 1e30;
    for (int b = 0; b < grid; ++b) { const double d = (double)(c[2 * b + 1] - c[2 * b]) / NREP; sum += d; mx = fmax(mx, d); mn = fmin(mn, d); }
    // fp64 check of stage 0 (workgroup `grid - 1` wrote last; every workgroup computes the same thing)
    std::vector<float> Y((size_t)32 * RG * D);
    CHECK(hipMemcpy(Y.data(), dY, Y.size() * sizeof(float), hipMemcpyDeviceToHost));
    double err = 0, ref_max = 0;
    for (int r = 0; r < 32 * RG; ++r)
        for (int o = 0; o < D; ++o) {
            double s = 0;
            for (int k = 0; k < D; ++k) s += (double)Wf[(size_t)k * D + o] * (double)Xf[(size_t)r * D + k];       // stage 0's weight [k][out]
            err = fmax(err, fabs(s - (double)Y[(size_t)r * D + o]));
            ref_max = fmax(ref_max, fabs(s));
        }
    // (clock64 = s_memtime: shader-clock cycles on this part -- the unit of every stamp table in profiles/)
    printf("%-58s %4d rows  cycles/stage mean %7.1f (min %7.1f max %7.1f)  stage-0 error %.2e of max |y| %.2f\n", name, 32 * RG, sum / grid, mn, mx, err, ref_max);
}

int main() {
    const int grid = 256;
    const size_t plane = pack3_plane(D, D), wsz = 3 * plane;
    std::vector<float> Wf((size_t)NREP * D * D), Xf((size_t)64 * D);
    srand(1);
    for (auto& v : Wf) v = (float)((rand() / (double)RAND_MAX - 0.5) * 0.35);
    for (auto& v : Xf) v = (float)((rand() / (double)RAND_MAX - 0.5) * 2.0);
    std::vector<uint16_t> W3((size_t)NREP * wsz);
    for (int rep = 0; rep < NREP; ++rep)
        for (int k = 0; k < D; ++k)
            for (int o = 0; o < D; ++o) {
                const float v = Wf[((size_t)rep * D + k) * D + o];
                const uint16_t hh = bf16_rne(v); const float r1 = v - bf16_f(hh);
                const uint16_t mm = bf16_rne(r1); const float r2 = r1 - bf16_f(mm);
                const uint16_t ll = bf16_rne(r2);
                const size_t i = (size_t)rep * wsz + pack3_index(k, o, D);
                W3[i] = hh; W3[i + plane] = mm; W3[i + 2 * plane] = ll;
            }
    uint16_t* dW; float *dX, *dY, *dS; long long* dC;
    CHECK(hipMalloc(&dW, W3.size() * 2)); CHECK(hipMalloc(&dX, Xf.size() * 4)); CHECK(hipMalloc(&dY, (size_t)64 * D * 4));
    CHECK(hipMalloc(&dC, 2 * grid * sizeof(long long))); CHECK(hipMalloc(&dS, (size_t)grid * 512 * 4));
    CHECK(hipMemcpy(dW, W3.data(), W3.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dX, Xf.data(), Xf.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> W0(Wf.begin(), Wf.begin() + (size_t)D * D);
    printf("one GEMM stage (rows x 128 . 128 x 128, bf16x6), chain of %d stages, %d workgroups (one per CU); matrix pipe alone: 48 MFMAs x 32 cycles = 1536 cycles per wave and stage\n", NREP, grid);
    run<1, false>("4 waves, product chain only", dW, dX, dY, dC, dS, W0, Xf, grid);
    run<1, true>("4 waves, + split-store epilogue and two barriers", dW, dX, dY, dC, dS, W0, Xf, grid);
    run<2, false>("8 waves (2 row groups x 4 channel groups), product chain", dW, dX, dY, dC, dS, W0, Xf, grid);
    run<2, true>("8 waves, + split-store epilogue and two barriers", dW, dX, dY, dC, dS, W0, Xf, grid);
    // what bounds the product chain: the same chains with NO weight stream (the ring keeps stage 0's fragments: matrix pipe + LDS operand reads only)
    run<1, false, false>("4 waves, product chain, weights resident in registers", dW, dX, dY, dC, dS, W0, Xf, grid);
    run<2, false, false>("8 waves, product chain, weights resident in registers", dW, dX, dY, dC, dS, W0, Xf, grid);
    run<2, true, false>("8 waves, + epilogue, weights resident in registers", dW, dX, dY, dC, dS, W0, Xf, grid);
    return 0;
}
