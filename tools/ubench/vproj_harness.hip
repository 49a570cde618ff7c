// Stand-alone harness for k_vproj_fwd3 (kernels_split.hip): timing, per-phase cycle stamps (-DVP3_STAMPS) and a numerical check against fp64.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DVP3_STAMPS vproj_harness.hip -o vproj_harness.bin && ./vproj_harness.bin
#include "../../vslnet_amd/csrc/kernels_split.hip"
#include <vector>
#include <math.h>
namespace vsl { void vsl_launch_events(hipStream_t, hipEvent_t* a, hipEvent_t* b) { *a = nullptr; *b = nullptr; } }
using namespace vsl;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static double urand() { return (rand() + 0.5) / ((double)RAND_MAX + 1.0); }
static double nrand() { return sqrt(-2.0 * log(urand())) * cos(6.283185307179586 * urand()); }
int main(int argc, char** argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 8192, Dv = argc > 2 ? atoi(argv[2]) : 1024;
    const int Kp = (Dv + 127) / 128 * 128;
    std::vector<float> X((size_t)R * Dv), W((size_t)D * Dv), b(D);
    for (auto& v : X) v = (float)nrand();
    for (auto& v : W) v = (float)(nrand() * 0.05);
    for (auto& v : b) v = (float)nrand();
    const size_t plane = pack3_plane(Kp, D);
    std::vector<uint16_t> W3(3 * plane, 0);
    for (int c = 0; c < D; ++c)
        for (int k = 0; k < Dv; ++k) {
            uint16_t h, m, l;
            split3_scalar(W[(size_t)c * Dv + k], h, m, l);
            const size_t o = pack3_index(k, c, D);
            W3[o] = h; W3[plane + o] = m; W3[2 * plane + o] = l;
        }
    float *dX, *db, *dY; uint16_t* dW;
    CHECK(hipMalloc(&dX, X.size() * 4)); CHECK(hipMalloc(&dW, W3.size() * 2)); CHECK(hipMalloc(&db, D * 4)); CHECK(hipMalloc(&dY, (size_t)R * D * 4));
    CHECK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dW, W3.data(), W3.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, b.data(), D * 4, hipMemcpyHostToDevice));
    for (int drop = 0; drop < 2; ++drop) {
        Drop dp{12345u, drop ? (uint32_t)(0.2 * 4294967296.0) : 0u, drop ? 1.25f : 1.f, 777u};
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) launch_vproj_fwd3(dX, dW, db, dY, R, Dv, dp, 0);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) launch_vproj_fwd3(dX, dW, db, dY, R, Dv, dp, 0);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("R=%d Dv=%d dropout %s: %.2f us per launch (back to back)\n", R, Dv, drop ? "on" : "off", ms * 1000 / 20);
        if (!drop) {
            std::vector<float> Y((size_t)R * D);
            CHECK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
            double emax = 0, ymax = 0;
            for (int r = 0; r < R; r += 97)
                for (int c = 0; c < D; ++c) {
                    double s = b[c];
                    for (int k = 0; k < Dv; ++k) s += (double)X[(size_t)r * Dv + k] * W[(size_t)c * Dv + k];
                    emax = fmax(emax, fabs(s - Y[(size_t)r * D + c])); ymax = fmax(ymax, fabs(s));
                }
            printf("  max |Y - fp64| = %.3e (max |Y| = %.2f)\n", emax, ymax);
        }
#ifdef VP3_STAMPS
        long long st[8][VP3_NST];
        CHECK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_vp_stamps), sizeof st));
        for (int w : {0, 4}) {
            printf("  wave %d stamps (cycles since the first):", w);
            for (int k = 1; k < VP3_NST && st[w][k]; ++k) printf(" %lld", st[w][k] - st[w][0]);
            printf("\n");
        }
#endif
    }
    return 0;
}
