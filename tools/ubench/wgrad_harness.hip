// Stand-alone harness for k_wgrad5 (wgrad5_kernel.inc, a round-4 experiment that was NOT adopted: profiles/r04_notes.md): the transposing LDS read's lane mapping, a numerical check of every
// operand-source combination (fp32 / bf16 planes, dropout, ragged rows, ragged K, three G blocks, the in-kernel fold of two partner jobs)
// against fp64, and timings against k_wgrad4 at the headline shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize wgrad_harness.hip -o wgrad_harness.bin && ./wgrad_harness.bin
#define WG4_STAMPS
#include "../../vslnet_amd/csrc/kernels_wgrad.hip"
#include "wgrad5_kernel.inc"
#include <vector>
#include <math.h>
#include <string.h>
namespace vsl { void vsl_launch_events(hipStream_t, hipEvent_t* a, hipEvent_t* b) { *a = nullptr; *b = nullptr; } }
using namespace vsl;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static double urand() { return (rand() + 0.5) / ((double)RAND_MAX + 1.0); }
static double nrand() { return sqrt(-2.0 * log(urand())) * cos(6.283185307179586 * urand()); }

__global__ void k_tr_probe(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 4];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (short)i;
    __syncthreads();
    const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + threadIdx.x * 4));
    for (int q = 0; q < 4; ++q) out[threadIdx.x * 4 + q] = v[q];
}
// LDS read rate of the operand pattern: 8 waves, each 18 transposing reads per "step" from the swizzled tile (mode 0), an unswizzled 256-byte-row
// tile (mode 1) and ds_read_b128 of the k_wgrad4 layout for comparison (mode 2: 9 reads of 16 bytes)
__global__ __launch_bounds__(512) void k_tr_rate(long long* out, int mode, uint32_t* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char Ls[6 * 4096];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 6 * 1024; i += 512) reinterpret_cast<uint32_t*>(Ls)[i] = i;
    __syncthreads();
    const int nh = wv & 1, kq = wv >> 1, h = lane >> 5, p16 = lane & 15, ihalf = (lane >> 4) & 1;
    auto fo = [&](int cb, bool swz) { return (8 * h + (p16 >> 2)) * 256 + (swz ? ((((cb >> 5) ^ (p16 >> 2)) & 3) << 6) : ((cb >> 5) << 6)) + ((16 * ihalf + 4 * (p16 & 3)) << 1); };
    const int g0 = fo(64 * nh, mode == 0), g1 = fo(64 * nh + 32, mode == 0), a0 = 3 * 4096 + fo(32 * kq, mode == 0);
    uint32_t acc = 0;
    const long long t0 = clock64();
    for (int it = 0; it < 256; ++it) {
        if (mode < 2) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const u32x4_t x = wg5_frag(Ls + t * 4096, g0), y = wg5_frag(Ls + t * 4096, g1), z = wg5_frag(Ls + t * 4096, a0);
                acc += x[0] ^ x[3] ^ y[1] ^ y[2] ^ z[0] ^ z[3];
            }
        } else {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const u32x4_t x = *reinterpret_cast<const u32x4_t*>(Ls + ((t * 64 + lane) * 16 + 1024 * wv) % (6 * 4096 - 16) / 16 * 16);
                acc += x[0] ^ x[3];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = clock64();
    if (lane == 0) out[wv] = t1 - t0;
    sink[tid] = acc;
}

struct Mat { std::vector<float> h; float* d = nullptr; uint16_t* d3 = nullptr; int R, C; };
static Mat make(int R, int C, double scale) {
    Mat m; m.R = R; m.C = C; m.h.resize((size_t)R * C);
    for (auto& v : m.h) v = (float)(nrand() * scale);
    CHECK(hipMalloc(&m.d, m.h.size() * 4));
    CHECK(hipMemcpy(m.d, m.h.data(), m.h.size() * 4, hipMemcpyHostToDevice));
    if (C == 128) {
        std::vector<uint16_t> p(3 * m.h.size());
        for (size_t i = 0; i < m.h.size(); ++i) { uint16_t a, b, c; split3_scalar(m.h[i], a, b, c); p[i] = a; p[m.h.size() + i] = b; p[2 * m.h.size() + i] = c; }
        CHECK(hipMalloc(&m.d3, p.size() * 2));
        CHECK(hipMemcpy(m.d3, p.data(), p.size() * 2, hipMemcpyHostToDevice));
    }
    return m;
}
static float host_keep(uint32_t idx, const Drop& d) {       // the device hash, restated (common.hpp drop_hash)
    uint32_t x = idx * 0x9E3779B1u + d.seed;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x ^= d.key; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x >= d.thresh ? d.scale : 0.f;
}
static void fill_starts(WgradBatch& wb) {
    int total = 0;
    for (int i = 0; i < wb.n; ++i) { wb.start[i] = total; total += wb.j[i].nG * ((wb.j[i].K + 127) / 128) * ((wb.j[i].R + WG_ROWS - 1) / WG_ROWS); }
    wb.start[wb.n] = total;
}
template <class F> static double time_us(F&& f, int iters = 20) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) f();
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.0 / iters;
}

int main() {
    // ---------------------------------------------------------------- 1. lane mapping of ds_read_b64_tr_b16
    {
        short* d; CHECK(hipMalloc(&d, 256 * 2));
        k_tr_probe<<<1, 64>>>(d);
        short hst[256]; CHECK(hipMemcpy(hst, d, 512, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) bad += hst[l * 4 + q] != (l & 15) + q * 16 + (l >> 4) * 64;
        printf("tr_b16 probe: lane l elem j = lds[(l&15) + 16 j + 64 (l>>4)] : %s (lane 5: %d %d %d %d)\n", bad ? "MISMATCH" : "ok", hst[20], hst[21], hst[22], hst[23]);
    }
    {
        long long* d; uint32_t* sink; CHECK(hipMalloc(&d, 64)); CHECK(hipMalloc(&sink, 2048));
        for (int mode = 0; mode < 3; ++mode) {
            k_tr_rate<<<1, 512>>>(d, mode, sink);
            long long t[8]; CHECK(hipMemcpy(t, d, 64, hipMemcpyDeviceToHost));
            printf("LDS operand reads, 8 waves, mode %d (%s): %.0f cycles per step and wave\n", mode, mode == 0 ? "tr, swizzled" : mode == 1 ? "tr, plain rows" : "b128 x 9", t[0] / 256.0);
        }
    }
    // ---------------------------------------------------------------- 2. numerics
    srand(1);
    double worst = 0;
    auto check = [&](const char* what, const WgradJob5& j, const std::vector<const Mat*>& G, const std::vector<const Mat*>& Ablk, const Mat* Afull, const std::vector<float>& got,
                     int ld, const std::vector<float>* gotb) {
        const int N = 128 * j.nG, K = j.K, R = j.R;
        double emax = 0, smax = 0, ebias = 0;
        for (int n = 0; n < N; n += 3)
            for (int k = 0; k < K; k += (K > 128 ? 5 : 1)) {
                double s = 0, sa = 0;
                for (int r = 0; r < R; ++r) {
                    const double g = G[n / 128]->h[(size_t)r * G[n / 128]->C + (n % 128) + (j.ldg ? (n / 128) * 0 : 0)];
                    double a = Afull ? Afull->h[(size_t)r * K + k] : Ablk[k / 128]->h[(size_t)r * 128 + k % 128];
                    if (Afull && j.drop_on_A) a *= host_keep((uint32_t)r * (uint32_t)K + (uint32_t)k, j.dp);
                    s += g * a; sa += fabs(g * a);
                }
                emax = fmax(emax, fabs(s - got[(size_t)n * ld + k])); smax = fmax(smax, sa);
            }
        if (gotb)
            for (int n = 0; n < N; ++n) {
                double s = 0;
                for (int r = 0; r < R; ++r) s += G[n / 128]->h[(size_t)r * 128 + n % 128];
                ebias = fmax(ebias, fabs(s - (*gotb)[n]));
            }
        printf("  %-58s err / max sum|g||a| = %.2e   bias err %.2e\n", what, emax / smax, ebias);
        worst = fmax(worst, emax / smax);
    };
    for (int variant = 0; variant < 6; ++variant) {
        // 0: fp32 / fp32   1: G planes   2: A planes   3: both planes   4: Afull K = 400 with dropout   5: three G blocks, K = 256 (two A blocks), G planes
        const int R = variant == 4 ? 1280 : 1000, rows = variant == 3 ? 512 : 0, CH = rows ? rows : WG_ROWS;
        const int nG = variant == 5 ? 3 : 1, K = variant == 4 ? 400 : variant == 5 ? 256 : 128, nA = variant == 4 ? 0 : K / 128;
        std::vector<Mat> Gm, Am; Mat Af;
        for (int g = 0; g < nG; ++g) Gm.push_back(make(R, 128, 0.3));
        for (int a = 0; a < nA; ++a) Am.push_back(make(R, 128, 1.0));
        if (!nA) Af = make(R, K, 1.0);
        // the same job twice (different data in the second): partner fold into one destination
        const int R2 = 300;
        std::vector<Mat> Gm2, Am2; Mat Af2;
        for (int g = 0; g < nG; ++g) Gm2.push_back(make(R2, 128, 0.3));
        for (int a = 0; a < nA; ++a) Am2.push_back(make(R2, 128, 1.0));
        if (!nA) Af2 = make(R2, K, 1.0);
        const int N = 128 * nG, nch = (R + CH - 1) / CH, nch2 = (R2 + CH - 1) / CH, nkt = (K + 127) / 128;
        float *slab, *slab2, *bsl, *bsl2, *dst, *dstb; unsigned* cnt;
        CHECK(hipMalloc(&slab, (size_t)nch * N * K * 4)); CHECK(hipMalloc(&slab2, (size_t)nch2 * N * K * 4));
        CHECK(hipMalloc(&bsl, (size_t)nch * N * 4)); CHECK(hipMalloc(&bsl2, (size_t)nch2 * N * 4));
        CHECK(hipMalloc(&dst, (size_t)N * K * 4)); CHECK(hipMalloc(&dstb, N * 4)); CHECK(hipMalloc(&cnt, 64 * 4));
        CHECK(hipMemset(cnt, 0, 256)); CHECK(hipMemset(dst, 0xff, (size_t)N * K * 4)); CHECK(hipMemset(dstb, 0xff, N * 4));
        WgradBatch5 wb; memset(&wb, 0, sizeof wb);
        const Drop dp{12345u, variant == 4 ? (uint32_t)(0.2 * 4294967296.0) : 0u, variant == 4 ? 1.25f : 1.f, 777u};
        for (int which = 0; which < 2; ++which) {
            WgradJob5 j; memset(&j, 0, sizeof j);
            std::vector<Mat>& GG = which ? Gm2 : Gm; std::vector<Mat>& AA = which ? Am2 : Am; Mat& AF = which ? Af2 : Af;
            j.nG = nG; j.nA = nA; j.K = K; j.R = which ? R2 : R; j.rows = rows; j.dp = dp; j.drop_on_A = variant == 4;
            for (int g = 0; g < nG; ++g) { j.G[g] = GG[g].d; if (variant == 1 || variant == 3 || variant == 5) j.G3[g] = GG[g].d3; }
            for (int a = 0; a < nA; ++a) { j.A[a] = AA[a].d; if (variant == 2 || variant == 3) j.A3[a] = AA[a].d3; }
            if (!nA) j.Afull = AF.d;
            j.out = which ? slab2 : slab;
            for (int g = 0; g < nG; ++g) {
                j.out_bias[g] = (which ? bsl2 : bsl) + (size_t)g * (which ? nch2 : nch) * 128;
                j.fold_dst[g] = dst + (size_t)g * 128 * K; j.fold_bias[g] = dstb + g * 128;
                j.fold_srcb1[g] = (which ? bsl : bsl2) + (size_t)g * (which ? nch : nch2) * 128;
            }
            j.fold_ld = K; j.fold_acc = 0; j.fold_cnt = cnt; j.fold_src1 = which ? slab : slab2; j.fold_n1 = which ? nch : nch2; j.fold_first = which == 0;
            wb.j[wb.n++] = j;
        }
        launch_wgrad5(wb, 0);
        CHECK(hipDeviceSynchronize());
        std::vector<float> got((size_t)N * K), gotb(N);
        CHECK(hipMemcpy(got.data(), dst, got.size() * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(gotb.data(), dstb, N * 4, hipMemcpyDeviceToHost));
        // reference over the concatenation of both jobs' rows
        std::vector<Mat> Gc(nG), Ac(nA); Mat Afc;
        auto cat = [&](const Mat& a, const Mat& b) { Mat m; m.R = a.R + b.R; m.C = a.C; m.h = a.h; m.h.insert(m.h.end(), b.h.begin(), b.h.end()); return m; };
        for (int g = 0; g < nG; ++g) Gc[g] = cat(Gm[g], Gm2[g]);
        for (int a = 0; a < nA; ++a) Ac[a] = cat(Am[a], Am2[a]);
        std::vector<const Mat*> Gp, Ap; for (auto& m : Gc) Gp.push_back(&m); for (auto& m : Ac) Ap.push_back(&m);
        WgradJob5 jr = wb.j[0]; jr.R = R + R2;
        if (!nA) {      // the dropout mask of the second job restarts at row 0: check the two jobs separately through their own slabs instead
            jr.R = R;
            std::vector<const Mat*> G1; for (auto& m : Gm) G1.push_back(&m);
            std::vector<float> s1((size_t)nch * N * K), acc1((size_t)N * K, 0.f);
            CHECK(hipMemcpy(s1.data(), slab, s1.size() * 4, hipMemcpyDeviceToHost));
            for (int c = 0; c < nch; ++c) for (size_t i = 0; i < acc1.size(); ++i) acc1[i] += s1[(size_t)c * N * K + i];
            check("Afull K=400 + dropout (first job's slabs)", jr, G1, {}, &Af, acc1, K, nullptr);
            // and the fold equals slab sums of both jobs, bit for bit
            std::vector<float> s2((size_t)nch2 * N * K);
            CHECK(hipMemcpy(s2.data(), slab2, s2.size() * 4, hipMemcpyDeviceToHost));
            size_t diff = 0;
            for (size_t i = 0; i < acc1.size(); ++i) { float t = 0.f; for (int c = 0; c < nch; ++c) t += s1[(size_t)c * N * K + i]; for (int c = 0; c < nch2; ++c) t += s2[(size_t)c * N * K + i]; diff += t != got[i]; }
            printf("  fold == ordered slab sum: %zu of %zu elements differ\n", diff, acc1.size());
        } else {
            const char* nm[6] = {"fp32 G, fp32 A (R=1000+300, fold of two jobs)", "G planes, fp32 A", "fp32 G, A planes", "both planes, 512-row chunks", "", "3 G blocks (planes), K=256"};
            check(nm[variant], jr, Gp, Ap, nullptr, got, K, &gotb);
        }
        // run twice more: counters must have been reset, results identical
        launch_wgrad5(wb, 0);
        CHECK(hipDeviceSynchronize());
        std::vector<float> got2((size_t)N * K);
        CHECK(hipMemcpy(got2.data(), dst, got2.size() * 4, hipMemcpyDeviceToHost));
        printf("  second launch bit-identical: %s\n", memcmp(got.data(), got2.data(), got.size() * 4) ? "NO" : "yes");
    }
    printf("worst relative error %.2e (k_wgrad4 class: 1.6e-8)\n", worst);
    // ---------------------------------------------------------------- 3. timings at the headline shape: 4 pointwise jobs of R = 8192
    {
        const int R = 8192;
        std::vector<Mat> Gm, Am;
        for (int q = 0; q < 4; ++q) { Gm.push_back(make(R, 128, 0.3)); Am.push_back(make(R, 128, 1.0)); }
        float *slab, *bsl, *dst, *dstb; unsigned* cnt;
        const int nchmax = R / 128;
        CHECK(hipMalloc(&slab, (size_t)4 * nchmax * 128 * 128 * 4)); CHECK(hipMalloc(&bsl, (size_t)4 * nchmax * 128 * 4));
        CHECK(hipMalloc(&dst, 4 * 128 * 128 * 4)); CHECK(hipMalloc(&dstb, 4 * 128 * 4)); CHECK(hipMalloc(&cnt, 256)); CHECK(hipMemset(cnt, 0, 256));
        auto batch = [&](int rows, bool gpl, bool apl, bool fold) {
            WgradBatch5 wb; memset(&wb, 0, sizeof wb);
            const int CH = rows ? rows : WG_ROWS, nch = (R + CH - 1) / CH;
            for (int q = 0; q < 4; ++q) {
                WgradJob5 j; memset(&j, 0, sizeof j);
                j.nG = 1; j.nA = 1; j.K = 128; j.R = R; j.rows = rows;
                j.G[0] = Gm[q].d; j.A[0] = Am[q].d; if (gpl) j.G3[0] = Gm[q].d3; if (apl) j.A3[0] = Am[q].d3;
                j.out = slab + (size_t)q * nch * 128 * 128; j.out_bias[0] = bsl + (size_t)q * nch * 128;
                if (fold) { j.fold_dst[0] = dst + q * 128 * 128; j.fold_bias[0] = dstb + q * 128; j.fold_ld = 128; j.fold_cnt = cnt + q; j.fold_first = 1; }
                wb.j[wb.n++] = j;
            }
            return wb;
        };
        {
            WgradBatch wb; memset(&wb, 0, sizeof wb);
            for (int q = 0; q < 4; ++q) {
                WgradJob j; memset(&j, 0, sizeof j);
                j.nG = 1; j.nA = 1; j.K = 128; j.R = R; j.G[0] = Gm[q].d; j.A[0] = Am[q].d;
                j.out = slab + (size_t)q * (R / WG_ROWS) * 128 * 128; j.out_bias[0] = bsl + (size_t)q * (R / WG_ROWS) * 128;
                wb.j[wb.n++] = j;
            }
            fill_starts(wb);
            const int total = wb.start[wb.n];
            printf("k_wgrad4, 4 jobs R=8192 (128 workgroups):              %.2f us\n", time_us([&] { hipLaunchKernelGGL((k_wgrad4<false, false>), dim3(total), dim3(WG4_T), 0, 0, wb); }));
            long long st[8];
            CHECK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_wg4_stamps), sizeof st));
            printf("  workgroup 0 (100 MHz wall clock ticks = 10 ns): job lookup %lld | address set-up + first loads + first stage %lld | 16-step loop %lld | slab stores issued %lld\n",
                   st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3]);
            {   // numerics of k_wgrad4 itself: job 0, slabs summed on the host, against fp64
                const int nch = R / WG_ROWS;
                std::vector<float> sl((size_t)nch * 128 * 128), bs((size_t)nch * 128);
                CHECK(hipMemcpy(sl.data(), slab, sl.size() * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(bs.data(), bsl, bs.size() * 4, hipMemcpyDeviceToHost));
                double emax = 0, smax = 0, eb = 0;
                for (int n = 0; n < 128; n += 5)
                    for (int k = 0; k < 128; k += 3) {
                        double sref = 0, sa = 0, got = 0;
                        for (int r = 0; r < R; ++r) { const double g = Gm[0].h[(size_t)r * 128 + n], a = Am[0].h[(size_t)r * 128 + k]; sref += g * a; sa += fabs(g * a); }
                        for (int c = 0; c < nch; ++c) got += sl[((size_t)c * 128 + n) * 128 + k];
                        emax = fmax(emax, fabs(got - sref)); smax = fmax(smax, sa);
                    }
                for (int n = 0; n < 128; ++n) { double sref = 0, got = 0; for (int r = 0; r < R; ++r) sref += Gm[0].h[(size_t)r * 128 + n]; for (int c = 0; c < nch; ++c) got += bs[(size_t)c * 128 + n]; eb = fmax(eb, fabs(got - sref)); }
                printf("  k_wgrad4 vs fp64 (R = 8192): err / max sum|g||a| = %.2e, bias err %.2e\n", emax / smax, eb);
            }
            // empty-kernel floor: same grid, same kernarg size
            printf("  (hipEvent pair around 20 back-to-back launches: includes ~2 us of launch gap each)\n");
        }
        for (int rows : {256, 512, 1024})
            for (int kind = 0; kind < 4; ++kind)
                for (int fold = 0; fold < 2; ++fold) {
                    WgradBatch5 wb = batch(rows, kind & 1, kind & 2, fold);
                    printf("k_wgrad5 rows %4d  G %-6s A %-6s fold %d:              %.2f us\n", rows, kind & 1 ? "planes" : "fp32", kind & 2 ? "planes" : "fp32", fold,
                           time_us([&] { launch_wgrad5(wb, 0); }));
                }
    }
    return 0;
}
