// Matrix-pipe rate of the two bf16 MFMA shapes, one workgroup, 1 or 2 (or 4) waves per SIMD, NCH independent accumulator chains:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define N 512
template <int NCH, bool BIG>
__global__ void k(long long* out, float* sink, int seed) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + seed + i); b[i] = (__bf16)(float)(seed - i); }
    f32x4 c4[NCH]; f32x16 c16[NCH];
    for (int j = 0; j < NCH; ++j) { for (int i = 0; i < 4; ++i) c4[j][i] = 0.f; for (int i = 0; i < 16; ++i) c16[j][i] = 0.f; }
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < N; ++it) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if (BIG) c16[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c16[j], 0, 0, 0);
            else c4[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4[j], 0, 0, 0);
        }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int j = 0; j < NCH; ++j) s += BIG ? c16[j][0] : c4[j][0];
    sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}
template <int NCH, bool BIG> void run(int threads, long long* d, float* s) {
    long long h[16];
    hipLaunchKernelGGL((k<NCH, BIG>), dim3(1), dim3(threads), 0, 0, d, s, 1); hipDeviceSynchronize();
    hipLaunchKernelGGL((k<NCH, BIG>), dim3(1), dim3(threads), 0, 0, d, s, 2); hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const int wps = threads / 256;
    long long mx = 0, mn = 1LL << 60;
    for (int w = 0; w < threads / 64; ++w) { mx = h[w] > mx ? h[w] : mx; mn = h[w] < mn ? h[w] : mn; }
    printf("%s  chains %d  waves/SIMD %d : fastest / slowest wave %.1f / %.1f cycles per MFMA -> %.1f per MFMA on the SIMD (slowest wave)\n", BIG ? "32x32x16" : "16x16x32",
           NCH, wps, (double)mn / (N * NCH), (double)mx / (N * NCH), (double)mx / (N * NCH) / wps);
}
int main() {
    long long* d; float* s; hipMalloc(&d, 256); hipMalloc(&s, 8192);
    for (int t : {256, 512, 1024}) {
        run<1, false>(t, d, s); run<2, false>(t, d, s); run<4, false>(t, d, s); run<8, false>(t, d, s);
        run<1, true>(t, d, s); run<2, true>(t, d, s); run<4, true>(t, d, s);
    }
    return 0;
}
