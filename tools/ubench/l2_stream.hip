// What a CU gets out of the L2 when every workgroup streams the SAME weight block (the pattern of every row-tile kernel of the step: one tile per CU,
// the whole packed weight per workgroup):  hipcc --offload-arch=gfx950 -O3 tools/ubench/l2_stream.hip -o /tmp/l2_stream && /tmp/l2_stream
// Each of the 8 waves of a workgroup reads its own 1/8 of a `kb`-KB block with 16-byte-per-lane loads (1 KB per wave instruction), `inflight` loads
// issued back to back before the first is consumed; modes: every workgroup the same block / a private block per workgroup (L2-resident: 256 x kb KB
// would not be, so private blocks are 16 per XCD-sized groups) / same block but each workgroup starts at a different offset (rotated).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int INF>
__global__ __launch_bounds__(512) void k_stream(const u32x4* __restrict__ src, uint32_t* __restrict__ sink, long long* __restrict__ cyc, int kb, int mode, int reps) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int vec_per_block = kb * 1024 / 16;                 // 16-byte vectors in the block
    const int per_wave = vec_per_block / 8;                    // this wave's share
    const u32x4* base = src + (mode == 1 ? (size_t)(blockIdx.x % 16) * vec_per_block : 0) + (size_t)w * per_wave;
    const int rot = mode == 2 ? (blockIdx.x * 7) % (per_wave / 64) : 0;
    u32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        for (int i = 0; i < per_wave / 64; i += INF) {
            u32x4 v[INF];
#pragma unroll
            for (int j = 0; j < INF; ++j) { int k = i + j + rot; if (k >= per_wave / 64) k -= per_wave / 64; v[j] = __builtin_nontemporal_load(base + (size_t)k * 64 + lane); }
#pragma unroll
            for (int j = 0; j < INF; ++j) acc ^= v[j];
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    sink[blockIdx.x * 512 + tid] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    const size_t bytes = 64u << 20;
    u32x4* src; uint32_t* sink; long long* cyc;
    hipMalloc(&src, bytes); hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    hipMemset(src, 1, bytes);
    std::vector<long long> h(256);
    const char* mn[3] = {"same block", "16 private blocks", "same block, rotated start"};
    for (int kb : {96, 288}) for (int mode = 0; mode < 3; ++mode) for (int inf : {1, 4, 12}) for (int grid : {32, 256}) {
        const int reps = 4;
        for (int it = 0; it < 2; ++it) {
            if (inf == 1) hipLaunchKernelGGL(k_stream<1>, dim3(grid), dim3(512), 0, 0, src, sink, cyc, kb, mode, reps);
            else if (inf == 4) hipLaunchKernelGGL(k_stream<4>, dim3(grid), dim3(512), 0, 0, src, sink, cyc, kb, mode, reps);
            else hipLaunchKernelGGL(k_stream<12>, dim3(grid), dim3(512), 0, 0, src, sink, cyc, kb, mode, reps);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
        double s = 0; long long mx = 0; for (int i = 0; i < grid; ++i) { s += h[i]; if (h[i] > mx) mx = h[i]; }
        printf("%3d KB  %-26s  %2d in flight per wave  grid %3d : avg %7.0f max %7lld cycles per pass = %5.1f B/clk/CU\n", kb, mn[mode], inf, grid, s / grid / reps, mx / reps,
               kb * 1024.0 / (s / grid / reps));
    }
    return 0;
}
