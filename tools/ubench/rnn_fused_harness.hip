// Stand-alone timing of the fused rnn head's forward launch (k_rnn_fwd of vslnet_amd/csrc/kernels_lstm.hip): when each of the three roles of
// sample 0 starts and ends (100 MHz wall clock), against the start LSTM alone.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 rnn_fused_harness.hip -o rnn_fused_harness.bin && ./rnn_fused_harness.bin [B] [T]
#include "../../vslnet_amd/csrc/kernels_lstm.hip"
#include <vector>
#include <math.h>
namespace vsl { void vsl_launch_events(hipStream_t, hipEvent_t* a, hipEvent_t* b) { *a = nullptr; *b = nullptr; } }
using namespace vsl;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(512, 2) void k_rnn_fwd_stamped(RnnFwdArgs a, long long* st, int roles, int first = 0) {
    __shared__ __attribute__((aligned(16))) float hs[2][4 * L1_SEG];
    const int B = a.B;
    const int role = (int)blockIdx.x / B, b = (int)blockIdx.x % B;
    const long long w0 = wall_clock64();
    if (role < first || role >= roles) {}
    else if (role == 0)
        lstm1_fwd_body<L1_PUBLISH>(hs, b, a.gi0, nullptr, a.Whh[0], a.bih[0], a.bhh[0], a.mask, a.gates[0], a.cseq[0], a.tseq[0], a.hprev[0],
                                   a.out[0], a.h_gran, a.epoch, a.T, 0, a.T);
    else if (role == 1) lstm1_proj_fwd(hs, b, a.h_gran, a.Wih1, a.gi_gran, a.epoch, a.T);
    else
        lstm1_fwd_body<L1_GRANULES>(hs, b, nullptr, a.gi_gran, a.Whh[1], a.bih[1], a.bhh[1], a.mask, a.gates[1], a.cseq[1], a.tseq[1],
                                    a.hprev[1], a.out[1], nullptr, a.epoch, a.T, 0, a.T);
    const long long w1 = wall_clock64();
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = w0; st[2 * blockIdx.x + 1] = w1; }
}

__global__ __launch_bounds__(512, 2) void k_rnn_bwd_stamped(RnnBwdArgs a, long long* st, int roles, int first = 0) {
    __shared__ __attribute__((aligned(16))) float dGs[2][16 * L1_SEG];
    const int B = a.B;
    const int role = (int)blockIdx.x / B, b = (int)blockIdx.x % B;
    const long long w0 = wall_clock64();
    if (role < first || role >= roles) {}
    else if (role == 0)
        lstm1_bwd_body<L1_PUBLISH>(dGs, b, a.dout[1], nullptr, nullptr, a.mask, a.gates[1], a.cseq[1], a.tseq[1], a.Whh[1], a.dG[1], a.dg_gran, a.epoch, a.T, nullptr, 0, a.T);
    else if (role == 1) lstm1_proj_bwd(dGs, b, a.dg_gran, a.Wih1, a.dx_gran, a.epoch, a.T);
    else
        lstm1_bwd_body<L1_GRANULES>(dGs, b, a.dout[0], nullptr, a.dx_gran, a.mask, a.gates[0], a.cseq[0], a.tseq[0], a.Whh[0], a.dG[0], nullptr, a.epoch, a.T, nullptr, 0, a.T);
    const long long w1 = wall_clock64();
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = w0; st[2 * blockIdx.x + 1] = w1; }
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16, T = argc > 2 ? atoi(argv[2]) : 128;
    const size_t R = (size_t)B * T;
    std::vector<float> hgi(R * 4 * D), hw((size_t)4 * D * D), hb(4 * D), hm(R, 1.f);
    srand(1);
    for (auto& v : hgi) v = (rand() / (float)RAND_MAX - 0.5f);
    for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.17f;
    for (auto& v : hb) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    std::vector<float> himg((size_t)4 * D * D);
    for (int e = 0; e < 4 * D * D; ++e) {
        const int x = e & 3, ln = (e >> 2) & 63, q = (e >> 8) & 31, wv = e >> 13, u = 16 * wv + (ln >> 2), jj = ln & 3;
        himg[e] = hw[(size_t)((jj ^ x) * D + u) * D + 32 * jj + q];
    }
    auto dev = [&](size_t n, const float* src) { float* p; CHECK(hipMalloc(&p, n * 4)); CHECK(hipMemset(p, 0, n * 4)); if (src) CHECK(hipMemcpy(p, src, n * 4, hipMemcpyHostToDevice)); return p; };
    RnnFwdArgs a;
    memset(&a, 0, sizeof a);
    a.gi0 = dev(hgi.size(), hgi.data());
    float* img = dev(himg.size(), himg.data());
    float* bias = dev(hb.size(), hb.data());
    a.mask = dev(hm.size(), hm.data());
    for (int l = 0; l < 2; ++l) {
        a.Whh[l] = img; a.bih[l] = bias; a.bhh[l] = bias;
        a.gates[l] = dev(R * 4 * D, nullptr); a.cseq[l] = dev(R * D, nullptr); a.tseq[l] = dev(R * D, nullptr); a.hprev[l] = dev(R * D + D, nullptr); a.out[l] = dev(R * D, nullptr);
    }
    a.Wih1 = img;
    a.h_gran = (unsigned long long*)dev(R * 2 * D, nullptr); a.gi_gran = (unsigned long long*)dev(R * 8 * D, nullptr);
    a.B = B; a.T = T;
    long long* st; CHECK(hipMalloc(&st, 3 * B * 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    unsigned epoch = 0x7FE00000u;
    for (int roles = 1; roles <= 3; ++roles) {
        float best = 1e9f;
        std::vector<long long> h(6 * B);
        for (int rep = 0; rep < 6; ++rep) {
            a.epoch = ++epoch;
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_rnn_fwd_stamped, dim3(3 * B), dim3(512), 0, 0, a, st, roles);
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) { best = ms; CHECK(hipMemcpy(h.data(), st, 6 * B * 8, hipMemcpyDeviceToHost)); }
        }
        long long base = h[0];
        for (int i = 0; i < 3 * B; ++i) base = std::min(base, h[2 * i]);
        printf("B = %d, T = %d, roles 0 .. %d live: %.1f us per launch\n", B, T, roles - 1, best * 1e3);
        for (int role = 0; role < roles; ++role) {
            long long s0 = 1ll << 62, s1 = 0, e0_ = 1ll << 62, e1_ = 0;
            for (int b = 0; b < B; ++b) { const long long s = h[2 * (role * B + b)] - base, e = h[2 * (role * B + b) + 1] - base; s0 = std::min(s0, s); s1 = std::max(s1, s); e0_ = std::min(e0_, e); e1_ = std::max(e1_, e); }
            printf("   role %d: starts %.2f .. %.2f us, ends %.2f .. %.2f us\n", role, s0 * 0.01, s1 * 0.01, e0_ * 0.01, e1_ * 0.01);
        }
    }
    // each consumer alone on granules that are all there: its own pace
    for (int first = 1; first <= 2; ++first) {
        std::vector<long long> h(6 * B);
        hipLaunchKernelGGL(k_rnn_fwd_stamped, dim3(3 * B), dim3(512), 0, 0, a, st, first + 1, first);       // same epoch as the last full launch
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), st, 6 * B * 8, hipMemcpyDeviceToHost));
        long long lo = 1ll << 62, hi = 0;
        for (int b = 0; b < B; ++b) { lo = std::min(lo, h[2 * (first * B + b) + 1] - h[2 * (first * B + b)]); hi = std::max(hi, h[2 * (first * B + b) + 1] - h[2 * (first * B + b)]); }
        printf("role %d alone, every granule ready: %.2f .. %.2f us for %d steps\n", first, lo * 0.01, hi * 0.01, T);
    }
    {   // ---- backward: the same three roles in reverse (end LSTM, dG W_ih projection, start LSTM), on what the forward saved
        std::vector<float> himb((size_t)4 * D * D), hd(R * D);
        for (int e = 0; e < 4 * D * D; ++e) {
            const int x = e & 3, ln = (e >> 2) & 63, q = (e >> 8) & 31, wv = e >> 13, r = q >> 3, kk = 4 * (q & 7) + x;
            himb[e] = hw[(size_t)(32 * (ln & 15) + kk) * D + 16 * wv + 4 * (ln >> 4) + (((ln >> 2) & 3) ^ r)];
        }
        for (auto& v : hd) v = (rand() / (float)RAND_MAX - 0.5f) * 0.01f;
        RnnBwdArgs g;
        memset(&g, 0, sizeof g);
        float* imb = dev(himb.size(), himb.data());
        float* dd = dev(hd.size(), hd.data());
        g.mask = a.mask; g.Wih1 = imb;
        for (int l = 0; l < 2; ++l) { g.dout[l] = dd; g.gates[l] = a.gates[l]; g.cseq[l] = a.cseq[l]; g.tseq[l] = a.tseq[l]; g.Whh[l] = imb; g.dG[l] = dev(R * 4 * D, nullptr); }
        g.dg_gran = (unsigned long long*)dev(R * 8 * D, nullptr); g.dx_gran = (unsigned long long*)dev(R * 2 * D, nullptr);
        g.B = B; g.T = T;
        for (int roles = 1; roles <= 3; ++roles) {
            float best = 1e9f;
            std::vector<long long> h(6 * B);
            for (int rep = 0; rep < 6; ++rep) {
                g.epoch = ++epoch;
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_rnn_bwd_stamped, dim3(3 * B), dim3(512), 0, 0, g, st, roles, 0);
                CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) { best = ms; CHECK(hipMemcpy(h.data(), st, 6 * B * 8, hipMemcpyDeviceToHost)); }
            }
            long long base = h[0];
            for (int i = 0; i < 3 * B; ++i) base = std::min(base, h[2 * i]);
            printf("backward, roles 0 .. %d live: %.1f us per launch\n", roles - 1, best * 1e3);
            for (int role = 0; role < roles; ++role) {
                long long e0_ = 1ll << 62, e1_ = 0;
                for (int b = 0; b < B; ++b) { const long long e = h[2 * (role * B + b) + 1] - base; e0_ = std::min(e0_, e); e1_ = std::max(e1_, e); }
                printf("   role %d ends %.2f .. %.2f us\n", role, e0_ * 0.01, e1_ * 0.01);
            }
        }
        for (int first = 1; first <= 2; ++first) {
            std::vector<long long> h(6 * B);
            hipLaunchKernelGGL(k_rnn_bwd_stamped, dim3(3 * B), dim3(512), 0, 0, g, st, first + 1, first);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h.data(), st, 6 * B * 8, hipMemcpyDeviceToHost));
            long long lo = 1ll << 62, hi = 0;
            for (int b = 0; b < B; ++b) { lo = std::min(lo, h[2 * (first * B + b) + 1] - h[2 * (first * B + b)]); hi = std::max(hi, h[2 * (first * B + b) + 1] - h[2 * (first * B + b)]); }
            printf("backward role %d alone, every granule ready: %.2f .. %.2f us for %d steps\n", first, lo * 0.01, hi * 0.01, T);
        }
    }
    // product launch
    for (int rep = 0; rep < 3; ++rep) { a.epoch = ++epoch; launch_rnn_fwd(a, 0); }
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int rep = 0; rep < 10; ++rep) { a.epoch = ++epoch; launch_rnn_fwd(a, 0); }
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_rnn_fwd: %.1f us per launch\n", ms * 100.f);
    return 0;
}
