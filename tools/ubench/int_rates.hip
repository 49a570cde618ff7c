// Issue cost of the integer instructions a counter-based dropout hash is made of (one wave and four waves per SIMD; inline asm so that nothing folds):
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/int_rates.hip -o /tmp/int_rates && /tmp/int_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define N 2048
#define CHAIN8(OP) \
    for (int i = 0; i < N; ++i) { \
        asm volatile(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8" \
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(k)); }
__global__ void kern(long long* out, uint32_t* sink, uint32_t seed) {
    uint32_t x[8], k = 0x85EBCA6Bu ^ seed;
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 7 + j + seed;
    long long t[6];
    __syncthreads(); t[0] = clock64();
    CHAIN8("v_mul_lo_u32")
    __syncthreads(); t[1] = clock64();
    CHAIN8("v_mul_u32_u24")
    __syncthreads(); t[2] = clock64();
    CHAIN8("v_xor_b32")
    __syncthreads(); t[3] = clock64();
    CHAIN8("v_mul_hi_u32")
    __syncthreads(); t[4] = clock64();
    for (int i = 0; i < N; ++i) {
        asm volatile("v_mad_u32_u24 %0, %0, %8, %1\nv_mad_u32_u24 %1, %1, %8, %2\nv_mad_u32_u24 %2, %2, %8, %3\nv_mad_u32_u24 %3, %3, %8, %4\n"
                     "v_mad_u32_u24 %4, %4, %8, %5\nv_mad_u32_u24 %5, %5, %8, %6\nv_mad_u32_u24 %6, %6, %8, %7\nv_mad_u32_u24 %7, %7, %8, %0"
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(k));
    }
    __syncthreads(); t[5] = clock64();
    uint32_t s = 0; for (int j = 0; j < 8; ++j) s += x[j];
    sink[threadIdx.x] = s;
    if (threadIdx.x == 0) for (int j = 0; j < 5; ++j) out[j] = t[j + 1] - t[j];
}
int main() {
    long long* d; uint32_t* s; long long h[5];
    hipMalloc(&d, 64); hipMalloc(&s, 8192);
    const char* nm[5] = {"v_mul_lo_u32", "v_mul_u32_u24", "v_xor_b32", "v_mul_hi_u32", "v_mad_u32_u24"};
    for (int threads : {64, 256, 512, 1024}) {
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, d, s, 1u); hipDeviceSynchronize();
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, d, s, 2u); hipDeviceSynchronize();
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("%4d threads (%d wave(s) per SIMD):", threads, threads <= 256 ? 1 : threads / 256);
        for (int j = 0; j < 5; ++j) printf("  %s %.2f", nm[j], (double)h[j] / (8.0 * N));
        printf("   cycles per instruction per wave (whole-workgroup time)\n");
    }
    return 0;
}
