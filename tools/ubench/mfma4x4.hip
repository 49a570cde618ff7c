// Probe of v_mfma_f32_4x4x1_16B_f32 (lane layout, cbsz broadcast, issue rate) for the 4-sample LSTM kernels.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_layout(float* out, int cb) {
    const int l = threadIdx.x;
    const float a = 1.0f + l, b = 100.0f * (1 + l);          // A value of lane l, B value of lane l
    f32x4 c = {0, 0, 0, 0};
    if (cb) c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 0, 0);   // broadcast block 0's A to all 16 blocks
    else c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
__global__ void k_rate(long long* out, float* sink) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float a = threadIdx.x * 1e-3f, b = 1.f + a;
    long long t0 = clock64();
    for (int i = 0; i < 1024; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 4, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, c1, 4, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, a, c2, 4, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, b, c3, 4, 0, 0);
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
    float *d, h[256]; long long *t, ht; float* s;
    hipMalloc(&d, 1024); hipMalloc(&t, 8); hipMalloc(&s, 1024);
    for (int cb = 0; cb < 2; ++cb) {
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, d, cb); hipDeviceSynchronize();
        hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
        printf("cbsz=%d: lane 0: %.0f %.0f %.0f %.0f | lane 1: %.0f %.0f %.0f %.0f | lane 5: %.0f %.0f %.0f %.0f | lane 63: %.0f %.0f %.0f %.0f\n", cb ? 4 : 0,
               h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[20], h[21], h[22], h[23], h[252], h[253], h[254], h[255]);
    }
    hipLaunchKernelGGL(k_rate, dim3(1), dim3(64), 0, 0, t, s); hipDeviceSynchronize();
    hipLaunchKernelGGL(k_rate, dim3(1), dim3(64), 0, 0, t, s); hipDeviceSynchronize();
    hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost);
    printf("4096 x mfma_4x4x1 (4 chains): %.2f cycles each\n", (double)ht / 4096);
    return 0;
}
