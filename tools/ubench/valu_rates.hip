// Micro-benchmarks that priced the design decisions of round 2 (run on an MI355X: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o /tmp/valu && /tmp/valu).
//  1. issue cost (cycles per wave-instruction) of the integer ops a counter-based dropout hash can be built from;
//  2. whether fp32 MFMAs of one wave overlap with VALU work of ANOTHER wave on the same SIMD (workgroup of 8 waves: waves 0-3 MFMA, 4-7 VALU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define N 4096
__device__ __forceinline__ uint32_t fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }
__global__ void k_rates(long long* out, uint32_t* sink, uint32_t seed) {
    uint32_t x = threadIdx.x + seed, y = x * 3 + 1, z = x ^ 0x1234567, u = y + 7;
    long long t0, t1;
    // mul_lo (4 independent chains)
    t0 = clock64();
    for (int i = 0; i < N; ++i) { x *= 0x85EBCA6Bu; y *= 0xC2B2AE35u; z *= 0x9E3779B1u; u *= 0x27D4EB2Fu; }
    t1 = clock64(); if (threadIdx.x == 0) out[0] = t1 - t0;
    t0 = clock64();
    for (int i = 0; i < N; ++i) { x ^= x >> 15; y ^= y >> 13; z ^= z >> 16; u ^= u >> 11; }
    t1 = clock64(); if (threadIdx.x == 0) out[1] = t1 - t0;
    t0 = clock64();
    for (int i = 0; i < N; ++i) { x = __umul24(x, 0x5EBCA6B) + y; y = __umul24(y, 0x2B2AE35) + z; z = __umul24(z, 0xE3779B1) + u; u = __umul24(u, 0x7D4EB2F) + x; }
    t1 = clock64(); if (threadIdx.x == 0) out[2] = t1 - t0;
    t0 = clock64();
    for (int i = 0; i < N; ++i) { x = fmix32(x + 0x9E3779B1u * i); y = fmix32(y + i); z = fmix32(z ^ i); u = fmix32(u - i); }
    t1 = clock64(); if (threadIdx.x == 0) out[3] = t1 - t0;
    t0 = clock64();
    for (int i = 0; i < N; ++i) { x = __builtin_amdgcn_alignbit(x, x, 19) + y; y = __builtin_amdgcn_alignbit(y, y, 7) ^ z; z = __builtin_amdgcn_alignbit(z, z, 13) + u; u = __builtin_amdgcn_alignbit(u, u, 25) ^ x; }
    t1 = clock64(); if (threadIdx.x == 0) out[4] = t1 - t0;
    float fa = x * 1e-9f, fb = y * 1e-9f, fc = z * 1e-9f, fd = u * 1e-9f;
    t0 = clock64();
    for (int i = 0; i < N; ++i) { fa = fa * 1.0001f + fb; fb = fb * 0.9999f + fc; fc = fc * 1.0002f + fd; fd = fd * 0.9998f + fa; }
    t1 = clock64(); if (threadIdx.x == 0) out[5] = t1 - t0;
    sink[threadIdx.x] = x + y + z + u + (uint32_t)(fa + fb + fc + fd);
}
// mode 0: waves 0-3 MFMA only ; 1: waves 4-7 VALU only ; 2: both ; 3: every wave does both interleaved
__global__ __launch_bounds__(512) void k_overlap(long long* out, float* sink, int mode, int mf16) {
    const int w = threadIdx.x >> 6;
    f32x16 acc = {0}; f32x4 a4 = {0, 0, 0, 0}, b4 = a4;
    float a = threadIdx.x * 1e-3f, b = 1.0f + a;
    float fa = a, fb = b, fc = a + 1, fd = b + 1;
    __syncthreads();
    long long t0 = clock64();
    const bool do_m = (mode == 0 || mode == 2) ? w < 4 : mode == 3;
    const bool do_v = (mode == 1 || mode == 2) ? w >= 4 : mode == 3;
    if (do_m && !do_v) {
        if (mf16) for (int i = 0; i < N; ++i) { a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a4, 0, 0, 0); b4 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, b4, 0, 0, 0); }
        else for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    } else if (do_v && !do_m) {
        for (int i = 0; i < N; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { fa = fa * 1.0001f + fb; fb = fb * 0.9999f + fc; fc = fc * 1.0002f + fd; fd = fd * 0.9998f + fa; }
        }
    } else if (do_m && do_v) {
        for (int i = 0; i < N; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) { fa = fa * 1.0001f + fb; fb = fb * 0.9999f + fc; fc = fc * 1.0002f + fd; fd = fd * 0.9998f + fa; }
        }
    }
    long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) out[w] = t1 - t0;
    sink[threadIdx.x] = acc[0] + a4[0] + b4[0] + fa + fb + fc + fd;
}
int main() {
    long long *d, h[8]; uint32_t* s; float* fs;
    hipMalloc(&d, 64); hipMalloc(&s, 4096); hipMalloc(&fs, 4096);
    hipLaunchKernelGGL(k_rates, dim3(1), dim3(64), 0, 0, d, s, 1u); hipDeviceSynchronize();
    hipLaunchKernelGGL(k_rates, dim3(1), dim3(64), 0, 0, d, s, 2u); hipDeviceSynchronize();
    hipMemcpy(h, d, 48, hipMemcpyDeviceToHost);
    const char* nm[6] = {"v_mul_lo_u32", "shift+xor pair", "mul_u24+add", "fmix32(+1 op)", "alignbit+add/xor", "v_fma_f32"};
    const int per[6] = {4, 8, 8, 4, 8, 4};
    for (int i = 0; i < 6; ++i) printf("%-18s %6.2f cycles per loop-iteration-of-%d-instr-groups -> %5.2f per unit\n", nm[i], (double)h[i] / N, per[i], (double)h[i] / N / per[i]);
    for (int mf = 0; mf < 2; ++mf)
        for (int mode = 0; mode < 4; ++mode) {
            if (mf && mode) continue;
            hipLaunchKernelGGL(k_overlap, dim3(1), dim3(512), 0, 0, d, fs, mode, mf); hipDeviceSynchronize();
            hipLaunchKernelGGL(k_overlap, dim3(1), dim3(512), 0, 0, d, fs, mode, mf); hipDeviceSynchronize();
            hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
            printf("overlap mode %d mfma16=%d: wave0 %.1f  wave4 %.1f cycles/iter (iter = 1 MFMA 32x32x2 [or 2 16x16x4] and/or 16 FMA)\n", mode, mf, (double)h[0] / N, (double)h[4] / N);
        }
    return 0;
}
