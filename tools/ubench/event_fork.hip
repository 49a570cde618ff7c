// How much does a cross-stream ordering point cost the PRODUCER stream?
//   (a) chain of N kernels on s0, nothing else
//   (b) after every kernel: hipEventRecord(e, s0) + hipStreamWaitEvent(s1, e) + a small kernel on s1      (marker packet in s0)
//   (c) the same fork, but the event rides on the kernel's own dispatch packet: hipExtLaunchKernelGGL(..., stopEvent = e)
//   (d) like (b) plus s0 waits for the s1 kernel before its next launch (a join on the critical chain)
//   (e) like (d) with stop events on both kernels
//   (f) like (a) with a stop event on every dispatch and nobody waiting
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/event_fork.hip -o /tmp/event_fork ; run: /tmp/event_fork
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(float* p, int iters) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0000001f + 1e-7f;
    if (v == 123.f) p[threadIdx.x] = v;
}
int main() {
    float* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
    hipStream_t s0, s1; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    const int N = 200, IT = 1200;   // ~20 us kernels
    std::vector<hipEvent_t> ev(2 * N);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    auto run = [&](int mode) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) {
            if (mode == 2 || mode == 4 || mode == 5) hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s0, nullptr, ev[i], 0, d, IT);
            else hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s0, d, IT);
            if (mode == 1 || mode == 3) hipEventRecord(ev[i], s0);
            if (mode >= 1 && mode <= 4) {
                hipStreamWaitEvent(s1, ev[i], 0);
                if (mode == 4) hipExtLaunchKernelGGL(spin, dim3(16), dim3(256), 0, s1, nullptr, ev[N + i], 0, d, IT / 4);
                else hipLaunchKernelGGL(spin, dim3(16), dim3(256), 0, s1, d, IT / 4);
                if (mode == 3) hipEventRecord(ev[N + i], s1);
                if (mode >= 3) hipStreamWaitEvent(s0, ev[N + i], 0);
            }
        }
        hipStreamSynchronize(s0); hipStreamSynchronize(s1);
        auto t1 = std::chrono::steady_clock::now();
        return std::chrono::duration<double, std::micro>(t1 - t0).count() / N;
    };
    const char* names[] = {"(a) chain only", "(b) fork via hipEventRecord", "(c) fork via stopEvent on the dispatch", "(d) fork + join via records", "(e) fork + join via stop events", "(f) chain, a stop event on every dispatch, no waiter"};
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 6; ++m) printf("%-44s %7.2f us per link\n", names[m], run(m));
    return 0;
}
