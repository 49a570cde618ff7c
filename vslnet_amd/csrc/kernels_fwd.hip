// Forward kernels of the VSLNet hot path (gfx950).  Every kernel cites the reference lines it reproduces
// (/root/reference/model/layers_t7.py unless noted).  Layout: activations are row-major (B*L, 128) fp32.
#include "common.hpp"
#include "launch.hpp"
#include <algorithm>
#include <type_traits>

namespace vsl {

// VSL_DEBUG_TIMING: block 0 / thread 0 of an instrumented kernel stamps the shader clock at its phase boundaries
__device__ long long g_stamps_f[32];
__device__ int g_dbg_on_f = 0;
#ifdef VSL_STAMPS       // see kernels_bwd.hip: stamps are a separate build
#define FSTAMP(k) do { if (g_dbg_on_f && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_stamps_f[k] = clock64(); } while (0)
#else
#define FSTAMP(k) do { } while (0)
#endif
static int fdbg_on() {
#ifndef VSL_STAMPS
    return 0;
#endif
    static int inited = 0, on = 0;
    if (!inited) { inited = 1; on = getenv("VSL_DEBUG_TIMING") != nullptr; if (on) { int one = 1; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_on_f), &one, sizeof one); } }
    return on;
}
static void fdbg_report(const char* name, int nst, hipStream_t s, int& left) {
    if (left <= 0) return;
    long long h[32];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps_f), sizeof h);
    fprintf(stderr, "[%s cycles]", name);
    for (int i = 1; i < nst; ++i) fprintf(stderr, " %lld", h[i] - h[i - 1]);
    fprintf(stderr, " | total %lld\n", h[nst - 1] - h[0]);
    --left;
}

// =========================================================================================================
// weight packing (one launch per forward; jobs table lives in the plan)
// =========================================================================================================
__global__ __launch_bounds__(256) void k_pack(const float* __restrict__ params, float* __restrict__ pack,
                                              const PackJob* __restrict__ jobs) {
    const PackJob j = jobs[blockIdx.y];
    if (j.transpose == 5) {         // bf16 forward pack (round to nearest even), K zero-padded to 16
        const int kn16 = (j.kn + 15) & ~15;
        uint16_t* p16 = reinterpret_cast<uint16_t*>(pack + j.dst);
        for (int e = blockIdx.x * 256 + threadIdx.x; e < kn16 * j.cn; e += gridDim.x * 256) {
            const int c = e / kn16, k = e - c * kn16;
            const float v = k < j.kn ? params[j.src + (size_t)c * j.ld + k] : 0.f;
            p16[((size_t)(k >> 4) * j.ncols + c) * 16 + (k & 15)] = f32_to_bf16(v);
        }
        return;
    }
    if (j.transpose == 6 || j.transpose == 7) {     // split pack: three bf16 planes of the operand's terms, K zero-padded to 16
        const int kn16 = j.kfill ? j.kfill : (j.kn + 15) & ~15;     // (a job that is not the last k segment of its operand has kn % 16 == 0)
        uint16_t* p16 = reinterpret_cast<uint16_t*>(pack + j.dst);
        const size_t plane = pack3_plane(j.ktot ? j.ktot : j.kn, j.ncols);
        for (int e = blockIdx.x * 256 + threadIdx.x; e < kn16 * j.cn; e += gridDim.x * 256) {
            int c, k;
            float v = 0.f;
            if (j.transpose == 6) { c = e / kn16; k = e - c * kn16; if (k < j.kn) v = params[j.src + (size_t)c * j.ld + k]; }       // Bm[k][c] = W[c][k]
            else { k = e / j.cn; c = e - k * j.cn; if (k < j.kn) v = params[j.src + (size_t)k * j.ld + c]; }                        // Bm[k][c] = W[k][c]
            uint16_t th, tm, tl;
            split3_scalar(v, th, tm, tl);
            const size_t o = pack3_index(j.k_off + k, j.col_off + c, j.ncols);
            p16[o] = th; p16[plane + o] = tm; p16[2 * plane + o] = tl;
        }
        return;
    }
    const int kn8 = j.transpose == 0 ? (j.kn + 7) & ~7 : j.kn;     // forward pack: zero the k rows that pad the last 8-block
    const int n = kn8 * j.cn;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        int k, c;
        float v;
        if (j.transpose == 3) {     // char-conv weight (nch, char_dim, 1, kw) -> image [ci][4 taps][100 channels]
            // kn = nch * char_dim * 4 (every tap slot written: zero beyond the kernel width) ; ld = kw ; ncols = char_dim ;
            // col_off = first channel of this conv
            const int kw = j.ld, cd = j.ncols;
            const int ch = e / (cd * 4), t = e - ch * cd * 4;
            const int ci = t >> 2, kk = t & 3;
            pack[j.dst + (ci * 4 + kk) * 100 + j.col_off + ch] = kk < kw ? params[j.src + (ch * cd + ci) * kw + kk] : 0.f;
            continue;
        }
        if (j.transpose == 8) {     // char-conv weight -> B operands of k_embed_bwd's dCe product in lane order: [channel tile][76 k-steps][64 lanes]
            // kn = channel tiles * 76 * 64 ; ld = kw ; ncols = char_dim ; col_off = first channel of this conv ; k_off = its channel count.
            // k-step q covers tap eb_tap(q), channels eb_oc0(q) + (lane >> 4); this job writes the slots of ITS channels (zero where the
            // tap lies beyond the kernel width or the input channel beyond char_dim), the four jobs together define the whole image
            const int kw = j.ld, cd = j.ncols;
            const int ct = e / (EB_IMG_Q * 64), rem = e - ct * EB_IMG_Q * 64, q = rem >> 6, ln = rem & 63;
            const int kk = q < 25 ? 0 : q < 48 ? 1 : q < 66 ? 2 : 3;
            const int oc = (q < 25 ? 4 * q : q < 48 ? 8 + 4 * (q - 25) : q < 66 ? 28 + 4 * (q - 48) : 60 + 4 * (q - 66)) + (ln >> 4);
            const int ocl = oc - j.col_off, ci = 16 * ct + (ln & 15);
            if (ocl >= 0 && ocl < j.k_off) pack[j.dst + e] = (ci < cd && kk < kw) ? params[j.src + (ocl * cd + ci) * kw + kk] : 0.f;
            continue;
        }
        if (j.transpose == 9 || j.transpose == 10) {     // a (512, 128) LSTM weight in the register order of the one-sample kernels: [wave][float4 q][lane][x]
            const int x = e & 3, ln = (e >> 2) & 63, q = (e >> 8) & 31, wv = e >> 13;
            const int r = q >> 3, kk = 4 * (q & 7) + x;                       // the lane's output r (named own ^ r), its contraction index kk of 32
            const int u = 16 * wv + (ln >> 2), jj = ln & 3;
            // forward: gate (jj ^ r) of unit u at k = 32 jj + kk ; backward: gate row 32 (ln & 15) + kk, column of unit 16 wv + 4 (ln >> 4) + (quad ^ r)
            const int src = j.transpose == 9 ? ((jj ^ r) * D + u) * D + 32 * jj + kk
                                             : (32 * (ln & 15) + kk) * D + 16 * wv + 4 * (ln >> 4) + (((ln >> 2) & 3) ^ r);
            pack[j.dst + e] = params[j.src + src];
            continue;
        }
        if (j.transpose == 4) {     // zero fill of kn * cn floats
            pack[j.dst + e] = 0.f;
            continue;
        }
        if (j.transpose) {          // Bm[k][c] = W[k][c]  (W row = contraction index): c is the fast source index
            k = e / j.cn; c = e - k * j.cn;
            v = params[j.src + (size_t)k * j.ld + c];
        } else {                    // Bm[k][c] = W[c][k]: k is the fast source index
            c = e / kn8; k = e - c * kn8;
            v = k < j.kn ? params[j.src + (size_t)c * j.ld + k] : 0.f;
        }
        pack[j.dst + pack_index(j.k_off + k, j.col_off + c, j.ncols)] = v;
    }
}
void launch_pack(const float* params, float* pack, const PackJob* jobs_dev, int njobs, hipStream_t s) {
    if (njobs == 0) return;
    VSL_LAUNCH(k_pack, dim3(64, njobs), dim3(256), 0, s, params, pack, jobs_dev);
}

// =========================================================================================================
// a2 in the bf16 THROUGHPUT mode (vsl_io.video_features_bf16): Y = (keep(X16) W16^T) / (1 - p) + b on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16: 16x the fp32 rate), X16 = bfloat16 features (half the HBM bytes), W16 = bf16-rounded weight
// (PackJob type 5), products exact in fp32, fp32 accumulation.  The dropout keeps / zeroes the bf16 inputs exactly and the
// scale 1 / (1 - p) multiplies the fp32 sum.  32 rows per workgroup, K streamed in 128-wide
// chunks through a double-buffered LDS tile (8 KB each).  Lane maps of the MFMA: A lane (i = l & 31, h = l >> 5) = 8
// consecutive k of row i starting at 8 h; B likewise for column i; C/D as for 32x32x2 (acc_row).
// =========================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int VP_KC = 128;
constexpr int VPB_LD = VP_KC + 8;       // bf16 elements per LDS row (272 B: 16-byte aligned rows, conflict-free b128 reads)
__global__ __launch_bounds__(256) void k_vproj_fwd_bf16(const uint16_t* __restrict__ X, const uint16_t* __restrict__ Wp,
                                                        const float* __restrict__ bias, float* __restrict__ Y, int R, int Dv, Drop dp) {
    __shared__ __attribute__((aligned(16))) uint16_t As[2][TILE_M * VPB_LD];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int r0 = blockIdx.x * TILE_M;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nchunk = (Dv + VP_KC - 1) / VP_KC;
    uint4 stage[2];
    auto gload = [&](int ch) {              // 32 rows x 128 k = 512 x 16 B: two per thread ; dropout = zeroing, exact in bf16
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + q * 256;
            const int rr = e >> 4, c = (e & 15) * 8 + ch * VP_KC;
            const int r = r0 + rr;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (r < R && c < Dv) {          // Dv is a multiple of 8 in this mode (checked by the launcher)
                v = *reinterpret_cast<const uint4*>(X + (size_t)r * Dv + c);
                if (dp.thresh) {
                    const uint32_t base = (uint32_t)((size_t)r * Dv + c);
                    uint32_t* u = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const uint32_t lo = drop_hash(base + 2 * p, dp.seed, dp.key) >= dp.thresh ? 0x0000FFFFu : 0u;
                        const uint32_t hi = drop_hash(base + 2 * p + 1, dp.seed, dp.key) >= dp.thresh ? 0xFFFF0000u : 0u;
                        u[p] &= lo | hi;
                    }
                }
            }
            stage[q] = v;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + q * 256;
            *reinterpret_cast<uint4*>(&As[buf][(e >> 4) * VPB_LD + (e & 15) * 8]) = stage[q];
        }
    };
    gload(0);
    sstore(0);
    __syncthreads();
    for (int ch = 0; ch < nchunk; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunk) gload(ch + 1);
        const int nk16 = (min(VP_KC, Dv - ch * VP_KC) + 15) >> 4;          // 16-wide k blocks in this chunk (zero padded in both operands)
        const uint16_t* wp = Wp + ((size_t)(ch * (VP_KC / 16)) * D + 32 * w + i) * 16 + 8 * h;
        uint4 bq[VP_KC / 16];
#pragma unroll
        for (int kk = 0; kk < VP_KC / 16; ++kk)
            bq[kk] = kk < nk16 ? *reinterpret_cast<const uint4*>(wp + (size_t)kk * D * 16) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int kk = 0; kk < VP_KC / 16; ++kk) {
            if (kk < nk16) {
                const uint4 av = *reinterpret_cast<const uint4*>(&As[buf][i * VPB_LD + kk * 16 + 8 * h]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bq[kk]), acc, 0, 0, 0);
            }
        }
        if (ch + 1 < nchunk) sstore(buf ^ 1);
        __syncthreads();
    }
    const int col = 32 * w + i;
    const float bv = bias[col], sc = dp.thresh ? dp.scale : 1.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gr = r0 + acc_row(r, lane);
        if (gr < R) Y[(size_t)gr * D + col] = acc[r] * sc + bv;
    }
}
void launch_vproj_fwd_bf16(const uint16_t* X, const uint16_t* Wpack16, const float* bias, float* Y, int R, int Dv, Drop dp, hipStream_t s) {
    VSL_LAUNCH(k_vproj_fwd_bf16, dim3((R + TILE_M - 1) / TILE_M), dim3(256), 0, s, X, Wpack16, bias, Y, R, Dv, dp);
}

// =========================================================================================================
// a3/a4  word + character embedding (:25-72) -> concatenated (Rq, 300 + 100) row, ready for the a5 linear.
//   One workgroup per EF_CHUNK = 8 query words.
//   word part : F.embedding over [pad; unk; glove] (:41) + dropout.
//   char CNN  : 4 x [Conv2d(50 -> c, (1,k)) + bias + ReLU -> max over char positions]; the arg-max position is saved
//               (int8) for the backward.  The 15000 conv weights are staged once per workgroup into LDS TRANSPOSED
//               ([ci*k + kk][channel], channel fastest: conflict-free for lanes that own consecutive channels); the
//               dropped-out char embeddings of the 8 words are staged transposed ([word][ci][position], zero padded) so
//               the sliding window of a channel is 3 vector LDS reads.  thread = (word, channel) items, 8 positions
//               at a time, no predicates in the inner loop.  Requires 4 <= Lc <= MAX_LC.
// =========================================================================================================
constexpr int EF_CHUNK = 8;
// LCM = compile-time bound of Lc (24 covers Charades / TACoS words; 40 the longest ActivityNet tokens): EF_PT = LCM + 4 padded
// positions per (word, ci) row (LCM + 3 taps, multiple of 4)
template <int LCM>
__global__ __launch_bounds__(512) void k_embed_fwd(const int64_t* __restrict__ word_ids, const int64_t* __restrict__ char_ids,
                                                   const float* __restrict__ pad_vec, const float* __restrict__ unk_vec,
                                                   const float* __restrict__ glove, const float* __restrict__ char_tab,
                                                   CharConvPtrs cc, const float* __restrict__ wimg, float* __restrict__ E,
                                                   int8_t* __restrict__ argpos, int Rq, int Lc, int word_dim, int char_dim,
                                                   int cb, Drop dw, Drop dc) {
    constexpr int EF_PT = LCM + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wt = smem;                                   // [cb][4 taps][100 channels], taps beyond a channel's width = 0
    float* CeT = Wt + cb * 400;                         // [EF_CHUNK][cb][EF_PT]
    __shared__ int cids[EF_CHUNK * LCM];
    const int tid = threadIdx.x, NT = blockDim.x;      // 512 threads: two waves per SIMD hide the LDS latency of the item loop
    const int EW = word_dim + 100;
    const int rbeg = blockIdx.x * EF_CHUNK, nw = min(EF_CHUNK, Rq - rbeg);
    FSTAMP(0);
    // ---- one block of `cb` input channels at a time (cb = char_dim up to 64: a single block; main_t7.py:24's char_dim 100 = two of 52):
    //      the LDS footprint is set by cb, not by char_dim, and the accumulators of a position tile run across the blocks
    auto stage_w = [&](int c0, int cbn) {
        // weights: straight vector copy of rows c0 .. c0 + cbn of the [ci][4 taps][100 channels] image k_pack built this step
        // (channel fastest: conflict-free for lanes that own consecutive channels; taps beyond a kernel width are 0)
        const float4* wsrc = reinterpret_cast<const float4*>(wimg + (size_t)c0 * 400);
        for (int e0 = 0; e0 < cbn * 100; e0 += 8 * NT) {
            float4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = e0 + tid + q * NT;
                v[q] = e < cbn * 100 ? wsrc[e] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int e = e0 + tid + q * NT; if (e < cbn * 100) reinterpret_cast<float4*>(Wt)[e] = v[q]; }
        }
    };
    auto stage_ce = [&](int c0, int cbn) {
        // the transposed dropped-out embeddings (zero beyond Lc and beyond the chunk)
        for (int pr = tid; pr < EF_CHUNK * cbn; pr += NT) {              // pair (word, ci): one padded row of EF_PT positions
            const int wi = pr / cbn, cl = pr - wi * cbn, ci = c0 + cl;
            float* row = CeT + (wi * cb + cl) * EF_PT;
            float v[LCM];
#pragma unroll
            for (int pp = 0; pp < LCM; ++pp)                            // all gathers of the row issued together
                v[pp] = (wi < nw && pp < Lc) ? char_tab[(size_t)cids[wi * LCM + pp] * char_dim + ci] : 0.f;
#pragma unroll
            for (int pp = 0; pp < LCM; ++pp)
                row[pp] = (pp < Lc) ? v[pp] * drop_mul(dc, (uint32_t)(((rbeg + wi) * Lc + pp) * char_dim + ci)) : 0.f;
#pragma unroll
            for (int pp = LCM; pp < EF_PT; ++pp) row[pp] = 0.f;
        }
    };
    if (cb >= char_dim) stage_w(0, char_dim);       // single block: the weight copy overlaps the id / word-vector phase
    // ---- char ids and word vectors (independent of everything below)
    for (int e = tid; e < EF_CHUNK * LCM; e += NT) {
        const int wi = e / LCM, pp = e - wi * LCM;
        cids[e] = (wi < nw && pp < Lc) ? (int)char_ids[(size_t)(rbeg + wi) * Lc + pp] : 0;
    }
    for (int e = tid; e < nw * word_dim; e += NT) {
        const int wi = e / word_dim, c = e - wi * word_dim;
        const int r = rbeg + wi;
        const int64_t wid = word_ids[r];
        const float* src = wid == 0 ? pad_vec : (wid == 1 ? unk_vec : glove + (size_t)(wid - 2) * word_dim);
        E[(size_t)r * EW + c] = src[c] * drop_mul(dw, (uint32_t)(r * word_dim + c));
    }
    __syncthreads();
    FSTAMP(1);
    // ---- char CNN on the matrix cores: out[p][oc] = sum_{kk, ci} Ce[p + kk][ci] W[oc][ci][kk] is a (positions x 4*char_dim)
    //      x (4*char_dim x 100) product per word (taps beyond a channel's kernel width are zero in the image).  wave = word,
    //      16 positions per MFMA tile, 7 tiles of 16 channels share every A operand; bias + ReLU + max / arg-max over the
    //      positions (:58, 69-70) happen in the accumulator registers and two shuffles.
    {
        const int w = tid >> 6, lane = tid & 63, jl = lane & 15, g4 = lane >> 4;
        const bool single = cb >= char_dim;
        if (single) { stage_ce(0, char_dim); __syncthreads(); }
        FSTAMP(2);
        const float* ce = CeT + w * cb * EF_PT;
        float best[7], bias[7];
        int bestp[7], kw[7];
#pragma unroll
        for (int nt = 0; nt < 7; ++nt) {
            const int oc = 16 * nt + jl;
            best[nt] = -1.f; bestp[nt] = 0;
            kw[nt] = oc < 10 ? 1 : oc < 30 ? 2 : oc < 60 ? 3 : 4;
            bias[nt] = oc < 10 ? cc.b[0][oc] : oc < 30 ? cc.b[1][oc - 10] : oc < 60 ? cc.b[2][oc - 30] : oc < 100 ? cc.b[3][oc - 60] : 0.f;
        }
        for (int mt = 0; 16 * mt < Lc; ++mt) {
            f32x4 acc[7];
#pragma unroll
            for (int nt = 0; nt < 7; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int c0 = 0; c0 < char_dim; c0 += cb) {
                const int cbn = min(cb, char_dim - c0);
                if (!single) { __syncthreads(); stage_w(c0, cbn); stage_ce(c0, cbn); __syncthreads(); }     // (block-uniform: every wave takes the barriers)
                if (w < nw) {
                    const int ncg = (cbn + 3) >> 2;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        // operands of step cg + 1 are requested before the MFMAs of step cg (one-deep register pipeline)
                        float av, bv[7], an, bn[7];
                        auto ld = [&](int cg, float& a_, float (&b_)[7]) {
                            const int ci = 4 * cg + g4;
                            const bool okc = ci < cbn;
                            a_ = okc ? ce[ci * EF_PT + 16 * mt + jl + kk] : 0.f;
                            const float* wr = Wt + (ci * 4 + kk) * 100 + jl;
#pragma unroll
                            for (int nt = 0; nt < 7; ++nt)
                                if ((kk < 2) || (kk == 2 && nt >= 1) || (kk == 3 && nt >= 3))
                                    b_[nt] = (okc && 16 * nt + jl < 100) ? wr[16 * nt] : 0.f;
                        };
                        ld(0, av, bv);
                        for (int cg = 0; cg < ncg; ++cg) {
                            if (cg + 1 < ncg) ld(cg + 1, an, bn);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int nt = 0; nt < 7; ++nt) {
                                // channel tiles whose widest kernel is narrower than this tap hold only zeros: skipped
                                if ((kk < 2) || (kk == 2 && nt >= 1) || (kk == 3 && nt >= 3))
                                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[nt], acc[nt], 0, 0, 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            av = an;
#pragma unroll
                            for (int nt = 0; nt < 7; ++nt) bv[nt] = bn[nt];
                        }
                    }
                }
            }
#pragma unroll
            for (int nt = 0; nt < 7; ++nt) {
                const int npos = Lc - kw[nt] + 1;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int pp = 16 * mt + 4 * g4 + rr;
                    const float v = fmaxf(acc[nt][rr] + bias[nt], 0.f);
                    if (pp < npos && v > best[nt]) { best[nt] = v; bestp[nt] = pp; }     // first maximum wins
                }
            }
        }
        if (w < nw) {
            const int r = rbeg + w;
#pragma unroll
            for (int nt = 0; nt < 7; ++nt) {
                float bv = best[nt];
                int bp = bestp[nt];
#pragma unroll
                for (int o = 16; o <= 32; o <<= 1) {               // combine the 4 position groups, lowest position on ties
                    const float ov = __shfl_xor(bv, o);
                    const int op = __shfl_xor(bp, o);
                    if (ov > bv || (ov == bv && op < bp)) { bv = ov; bp = op; }
                }
                const int oc = 16 * nt + jl;
                if (g4 == 0 && oc < 100) {
                    E[(size_t)r * EW + word_dim + oc] = bv;
                    argpos[(size_t)r * 100 + oc] = (int8_t)bp;
                }
            }
        }
    }
    FSTAMP(3);
}
void launch_embed_fwd(const int64_t* word_ids, const int64_t* char_ids, const float* pad_vec, const float* unk_vec,
                      const float* glove, const float* char_tab, CharConvPtrs cc, const float* wimg, float* E, int8_t* argpos,
                      int Rq, int Lc, int word_dim, int char_dim, Drop dw, Drop dc, hipStream_t s) {
    const int lcm = Lc <= 24 ? 24 : MAX_LC;
    // input-channel block: as many channels as fit the 160 KB of LDS beside the word rows (64 with the short-token instantiation, 52 with the
    // long one); wider character embeddings run in equal blocks (multiples of 4) with the accumulators carried across them
    const int cb_max = lcm == 24 ? 64 : 52;
    const int nblk = (char_dim + cb_max - 1) / cb_max;
    const int cb = nblk == 1 ? char_dim : (((char_dim + nblk - 1) / nblk) + 3) & ~3;
    const size_t shm = (size_t)(cb * 400 + EF_CHUNK * cb * (lcm + 4) + 32) * sizeof(float);   // + slack: invalid positions over-read
    static size_t ok24 = 0, ok40 = 0;
    const dim3 grid((Rq + EF_CHUNK - 1) / EF_CHUNK);
    if (lcm == 24) {
        ensure_dynamic_lds((const void*)k_embed_fwd<24>, shm, ok24, "k_embed_fwd<24>");
        VSL_LAUNCH(k_embed_fwd<24>, grid, dim3(512), shm, s, word_ids, char_ids, pad_vec, unk_vec, glove, char_tab, cc, wimg, E,
                           argpos, Rq, Lc, word_dim, char_dim, cb, dw, dc);
    } else {
        ensure_dynamic_lds((const void*)k_embed_fwd<MAX_LC>, shm, ok40, "k_embed_fwd<40>");
        VSL_LAUNCH(k_embed_fwd<MAX_LC>, grid, dim3(512), shm, s, word_ids, char_ids, pad_vec, unk_vec, glove, char_tab, cc, wimg,
                           E, argpos, Rq, Lc, word_dim, char_dim, cb, dw, dc);
    }
    static int left = 2;
    if (fdbg_on()) fdbg_report("embed_fwd: ids+words | weights+CeT | items", 4, s, left);
}

// =========================================================================================================
// generic row-tile linear  Y = A W^T + b  (Conv1D k=1, :12-22) for an (R, K) row-major A, K % 8 == 0.
// Used for a5 (Embedding.linear, K = 400).  A is staged through LDS in 128-wide chunks.
// =========================================================================================================
__global__ __launch_bounds__(256) void k_linear_fwd(const float* __restrict__ A, const float* __restrict__ Wpack,
                                                    const float* __restrict__ bias, float* __restrict__ Y, int R, int K) {
    __shared__ __attribute__((aligned(16))) float As[TILE_M * LDP];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int r0 = blockIdx.x * TILE_M;
    f32x16 acc[1];
    zero_acc(acc);
    for (int k0 = 0; k0 < K; k0 += 128) {
        const int kc = min(128, K - k0);
        for (int e = tid; e < TILE_M * 32; e += 256) {
            const int rr = e >> 5, c = (e & 31) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + rr < R && c < kc) v = *reinterpret_cast<const float4*>(A + (size_t)(r0 + rr) * K + k0 + c);
            *reinterpret_cast<float4*>(&As[rr * LDP + c]) = v;
        }
        const float* wp = Wpack + (size_t)(k0 / 8) * D * 8;
        BFrag<1, 16> bf;
        bfrag_load(bf, wp, D, 32 * w, 0, 0, kc >> 3);
        __syncthreads();
        gemm32p<1, 16>(As, LDP, kc, wp, D, 32 * w, 0, acc, bf);
        __syncthreads();
    }
    const int col = 32 * w + (lane & 31);
    const float bv = bias[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gr = r0 + acc_row(r, lane);
        if (gr < R) Y[(size_t)gr * D + col] = acc[0][r] + bv;
    }
}
void launch_linear_fwd(const float* A, const float* Wpack, const float* bias, float* Y, int R, int K, hipStream_t s) {
    {
        const size_t shm_sp = 0;
        VSL_LAUNCH(k_linear_fwd, dim3((R + TILE_M - 1) / TILE_M), dim3(256), shm_sp, s, A, Wpack, bias, Y, R, K);
    }
}

// =========================================================================================================
// a8 (attention core)  S = Q K^T / sqrt(hd) + (1 - mask[key]) * -1e30 ; P = softmax ; O = drop(P) V   (:174-182), sequences of L > 256
//   head size 16, fp32 MFMA 16x16x4.  Workgroup = (64 queries, head, sample); wave = 16 queries.
//   K/V head slices live in LDS; scores are computed transposed (S^T = K Q^T) so every lane owns one query column:
//   online softmax is lane-local (+2 shuffles across the 4 key groups) and P feeds the PV MFMA from registers.
//   Saves LSE = m + log(l) per (b, h, q) for the backward.
// Round 6: the key loop was issue-bound (765 cycles per 16-key tile and wave, 256 of them matrix work): it now takes 64 keys per turn -- four
// independent S tiles, ONE running-maximum update (2 shuffles, 1 rescale) per 64 keys instead of per 16 --, works in the log2 domain (x = S c +
// bias with c = log2(e) / 4: v_exp_f32 directly, LSE converted back at the end), draws two dropout decisions from every hash (drop_hash_odd) and
// feeds PV into two alternating accumulators.
// =========================================================================================================
#ifndef VSL_AF_KB
#define VSL_AF_KB 256
#endif
constexpr int AF_KB = VSL_AF_KB;    // keys staged per block: 45 KB of LDS whatever L is -> 3 workgroups per CU at L = 1024
constexpr int AF_NJ = AF_KB * 4 / 256;
constexpr float AF_LOG2E = 1.4426950408889634f, AF_LN2 = 0.6931471805599453f;
// PAIR: two dropout decisions per hash (L > 256) ; false: one hash per element, the masks of L <= 256 (the kernel then serves 128 < L <= 256 too)
template <bool PAIR>
__global__ __launch_bounds__(256) void k_attn_fwd(const float* __restrict__ Q, const float* __restrict__ K,
                                                  const float* __restrict__ V, const float* __restrict__ mask,
                                                  float* __restrict__ att, float* __restrict__ lse, int L, int H,
                                                  int b_off, Drop d2) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Lp = (L + 63) & ~63;                 // keys in turns of 64 (the pad keys carry the mask bias and zero K / V)
    const int KB = min(Lp, AF_KB);
    constexpr int kst = 20;
    float* Vs = smem;                 // [KB][20]  V head slice of the current key block
    float* Mb = Vs + KB * kst;        // [KB] additive key bias, log2 domain
    uint16_t* Kp = reinterpret_cast<uint16_t*>(Mb + KB);      // [3 terms][KB][16] bf16: the K head slice, split once per block (af_kp)
    const int KPL = KB * 16;          // elements per plane
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    int bxs, h, b;
    xcd_swizzle(bxs, h, b);                        // all heads of a sample on one XCD: its rows are fetched into one L2
    const size_t rowbase = (size_t)b * L;
    const int qi = lane & 15, g = lane >> 4;
    const int q = bxs * 64 + w * 16 + qi;
    const bool qok = q < L;
    // S^T = K Q^T on the bf16 matrix cores at fp32 grade: v_mfma_f32_16x16x32_bf16 contracts over 32 = TWO products of the 16 head dims, so
    // the six products of the 3-way split are three instructions (they overlap with the vector work of the SIMD's other waves; the fp32-input
    // MFMA does not: tools/ubench/split_bf16.hip):  [kh | km] x [qh | qh] + [kh | kl] x [qm | qh] + [kh | km] x [ql | qm].
    // lane (i, g): k group g < 2 = dims 8 g .. of the first term, g >= 2 = dims 8 (g - 2) .. of the second.
    u32x4_t bq1, bq2, bq3;
    {
        float4 qa = make_float4(0.f, 0.f, 0.f, 0.f), qb = qa;
        if (qok) {
            const float* qp = Q + (rowbase + q) * D + h * HD + 8 * (g & 1);
            qa = *reinterpret_cast<const float4*>(qp);
            qb = *reinterpret_cast<const float4*>(qp + 4);
        }
        uint32_t th[4], tm[4], tl[4];
        split3(qa.x, qa.y, th[0], tm[0], tl[0]); split3(qa.z, qa.w, th[1], tm[1], tl[1]);
        split3(qb.x, qb.y, th[2], tm[2], tl[2]); split3(qb.z, qb.w, th[3], tm[3], tl[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bq1[e] = th[e]; bq2[e] = g < 2 ? tm[e] : th[e]; bq3[e] = g < 2 ? tl[e] : tm[e]; }
    }
    // A operands of a key tile: planes (h | m) and (h | l) by k group
    const int ka1 = (g < 2 ? 0 : 1) * KPL + af_kp(qi, g & 1), ka2 = (g < 2 ? 0 : 2) * KPL + af_kp(qi, g & 1);
    const float c2 = 0.25f * AF_LOG2E;             // 1 / sqrt(16), applied AFTER QK^T like the reference (:175), times log2(e)
    float m = -3.0e38f, l = 0.f;
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    // dropout of the probabilities: decision of (row, key) = half (key & 1) of drop_hash(row * ceil(L / 2) + key / 2)
    const uint32_t prow = (uint32_t)(((size_t)(b + b_off) * H + h) * L + q) * (uint32_t)((L + 1) >> 1);
    const bool dropon = d2.thresh != 0u;
    // the rows of the NEXT key block travel in registers while the current one is worked on (8 float4 per thread): a block's global-memory
    // latency used to be exposed at every block boundary -- a third of the kernel's time with three workgroups per CU to hide it
    float4 pk[AF_NJ], pv[AF_NJ];
    float pm = 0.f;
    auto fetch = [&](int kb0) {
        const int nk = min(AF_KB, Lp - kb0);
#pragma unroll
        for (int j = 0; j < AF_NJ; ++j) {
            const int e = tid + 256 * j, key = kb0 + (e >> 2), c4 = (e & 3) * 4;
            pk[j] = make_float4(0.f, 0.f, 0.f, 0.f); pv[j] = pk[j];
            if (e < nk * 4 && key < L) {
                pk[j] = *reinterpret_cast<const float4*>(K + (rowbase + key) * D + h * HD + c4);
                pv[j] = *reinterpret_cast<const float4*>(V + (rowbase + key) * D + h * HD + c4);
            }
        }
        pm = kb0 + tid < L ? (1.0f - mask[rowbase + kb0 + tid]) * MASK_VALUE : MASK_VALUE;
    };
    fetch(0);
    for (int kb0 = 0; kb0 < Lp; kb0 += AF_KB) {
        const int nk = min(AF_KB, Lp - kb0);
        if (kb0) __syncthreads();                  // every wave is done with the previous block
#pragma unroll
        for (int j = 0; j < AF_NJ; ++j) {
            const int e = tid + 256 * j, kl = e >> 2, c4 = (e & 3) * 4;
            if (e < nk * 4) {
                uint32_t h0, m0, l0, h1, m1, l1;
                split3(pk[j].x, pk[j].y, h0, m0, l0);
                split3(pk[j].z, pk[j].w, h1, m1, l1);
                uint16_t* kd = Kp + af_kp(kl, c4 >> 3) + (c4 & 7);
                *reinterpret_cast<u32x2_t*>(kd) = u32x2_t{h0, h1};
                *reinterpret_cast<u32x2_t*>(kd + KPL) = u32x2_t{m0, m1};
                *reinterpret_cast<u32x2_t*>(kd + 2 * KPL) = u32x2_t{l0, l1};
                *reinterpret_cast<float4*>(&Vs[kl * kst + c4]) = pv[j];
            }
        }
        if (tid < nk) Mb[tid] = pm * AF_LOG2E;
        __syncthreads();
        if (kb0 + AF_KB < Lp) fetch(kb0 + AF_KB);
        for (int kt = 0; kt < nk; kt += 64) {
            // four S^T tiles: rows = keys kb0 + kt + 16 t + 4 g + reg, col = query qi
            f32x4 s[4];
            u32x4_t a1[4], a2[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a1[t] = *reinterpret_cast<const u32x4_t*>(Kp + ka1 + (kt + 16 * t) * 16);
                a2[t] = *reinterpret_cast<const u32x4_t*>(Kp + ka2 + (kt + 16 * t) * 16);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) s[t] = mfma16_bf16(a1[t], bq3, f32x4{0.f, 0.f, 0.f, 0.f});       // hl + mm (small terms first)
#pragma unroll
            for (int t = 0; t < 4; ++t) s[t] = mfma16_bf16(a2[t], bq2, s[t]);                            // hm + lh
#pragma unroll
            for (int t = 0; t < 4; ++t) s[t] = mfma16_bf16(a1[t], bq1, s[t]);                            // hh + mh
            float x[4][4];
            float tmax = -3.0e38f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float4 mb = *reinterpret_cast<const float4*>(&Mb[kt + 16 * t + 4 * g]);
                x[t][0] = fmaf(s[t][0], c2, mb.x); x[t][1] = fmaf(s[t][1], c2, mb.y);
                x[t][2] = fmaf(s[t][2], c2, mb.z); x[t][3] = fmaf(s[t][3], c2, mb.w);
                tmax = fmaxf(fmaxf(tmax, fmaxf(x[t][0], x[t][1])), fmaxf(x[t][2], x[t][3]));
            }
            tmax = lane_pair16(tmax, [](float a, float b) { return fmaxf(a, b); });
            tmax = lane_pair32(tmax, [](float a, float b) { return fmaxf(a, b); });
            const float mn = fmaxf(m, tmax);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            m = mn;
            l *= alpha;
#pragma unroll
            for (int r = 0; r < 4; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            // O^T += V^T P^T : A[i = dd][k = key] = V[key][dd], B[k = key][j = q] = P (in registers); fp32-input MFMA (P would have to be split per tile)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float pr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { pr[r] = __builtin_amdgcn_exp2f(x[t][r] - mn); l += pr[r]; }
                if (dropon && !PAIR) {
                    const uint32_t ei = (uint32_t)(((size_t)(b + b_off) * H + h) * L + q) * (uint32_t)L + (uint32_t)(kb0 + kt + 16 * t + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) pr[r] = drop_hash(ei + r, d2.seed, d2.key) >= d2.thresh ? pr[r] * d2.scale : 0.f;
                } else if (dropon) {
                    const uint32_t pi = prow + (uint32_t)((kb0 + kt + 16 * t + 4 * g) >> 1);
                    const uint32_t h0 = drop_hash(pi, d2.seed, d2.key), h1 = drop_hash(pi + 1u, d2.seed, d2.key);
                    pr[0] = h0 >= d2.thresh ? pr[0] * d2.scale : 0.f;
                    pr[1] = drop_hash_odd(h0) >= d2.thresh ? pr[1] * d2.scale : 0.f;
                    pr[2] = h1 >= d2.thresh ? pr[2] * d2.scale : 0.f;
                    pr[3] = drop_hash_odd(h1) >= d2.thresh ? pr[3] * d2.scale : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float vv = Vs[(kt + 16 * t + 4 * g + r) * kst + qi];
                    if (r & 1) o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, pr[r], o1, 0, 0, 0);
                    else o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, pr[r], o0, 0, 0, 0);
                }
            }
        }
    }
    l = lane_pair16(l, [](float a, float b) { return a + b; });
    l = lane_pair32(l, [](float a, float b) { return a + b; });
    if (qok) {
        const float inv = 1.0f / l;
        // lane (qi, g) holds O[q][dd = 4g + reg]
        *reinterpret_cast<float4*>(att + (rowbase + q) * D + h * HD + 4 * g) =
            make_float4((o0[0] + o1[0]) * inv, (o0[1] + o1[1]) * inv, (o0[2] + o1[2]) * inv, (o0[3] + o1[3]) * inv);
        if (g == 0) lse[((size_t)b * H + h) * L + q] = m * AF_LN2 + __logf(l);
    }
}
void launch_attn_fwd(const float* Q, const float* K, const float* V, const float* mask, float* att, float* lse, int B,
                     int L, int H, int b_off, Drop d2, hipStream_t s) {
    const int Lp = (L + 63) & ~63;
    const int KB = Lp < AF_KB ? Lp : AF_KB;
    const size_t shm = (size_t)(KB * 20 + KB) * sizeof(float) + (size_t)3 * KB * 16 * sizeof(uint16_t);
    static size_t lds_ok = 0, lds_ok_e = 0;
    if (L > 256) {
        ensure_dynamic_lds((const void*)k_attn_fwd<true>, shm, lds_ok, "k_attn_fwd");
        VSL_LAUNCH(k_attn_fwd<true>, dim3((L + 63) / 64, H, B), dim3(256), shm, s, Q, K, V, mask, att, lse, L, H, b_off, d2);
    } else {
        ensure_dynamic_lds((const void*)k_attn_fwd<false>, shm, lds_ok_e, "k_attn_fwd<false>");
        VSL_LAUNCH(k_attn_fwd<false>, dim3((L + 63) / 64, H, B), dim3(256), shm, s, Q, K, V, mask, att, lse, L, H, b_off, d2);
    }
}

// =========================================================================================================
// a8 (second half)  r = drop(att) + x ; y = drop( drop(LN2(r)) Wo^T + bo ) + r      (:183-190)
// =========================================================================================================
__global__ __launch_bounds__(256) void k_attn_out_fwd(const float* __restrict__ att, const float* __restrict__ x,
                                                      const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                      const float* __restrict__ Wpack, const float* __restrict__ bo,
                                                      float* __restrict__ r_out, float* __restrict__ h2_out,
                                                      float* __restrict__ y_out, int R, Drop d3, Drop d4, Drop d5) {
    __shared__ __attribute__((aligned(16))) float Rs[TILE_M * LDP];
    __shared__ __attribute__((aligned(16))) float Hs[TILE_M * LDP];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int r0 = blockIdx.x * TILE_M;
    BFrag<1, 16> bf;
    {
        float4 av[4], xv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + q * 256;
            const int r = r0 + (e >> 5), c = (e & 31) * 4;
            av[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            xv[q] = av[q];
            if (r < R) {
                av[q] = *reinterpret_cast<const float4*>(att + (size_t)r * D + c);
                xv[q] = *reinterpret_cast<const float4*>(x + (size_t)r * D + c);
            }
        }
        bfrag_load(bf, Wpack, D, 32 * w, 0, 0, D / 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + q * 256;
            const int rr = e >> 5, c = (e & 31) * 4;
            const int r = r0 + rr;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < R) {
                const uint32_t base = (uint32_t)(r * D + c);
                v.x = av[q].x * drop_mul(d3, base) + xv[q].x;
                v.y = av[q].y * drop_mul(d3, base + 1) + xv[q].y;
                v.z = av[q].z * drop_mul(d3, base + 2) + xv[q].z;
                v.w = av[q].w * drop_mul(d3, base + 3) + xv[q].w;
                *reinterpret_cast<float4*>(r_out + (size_t)r * D + c) = v;
            }
            *reinterpret_cast<float4*>(&Rs[rr * LDP + c]) = v;
            *reinterpret_cast<float4*>(&Hs[rr * LDP + c]) = v;
        }
    }
    __syncthreads();
    ln_tile(Hs, TILE_M, LDP, ln_g, ln_b, d4, r0);
    __syncthreads();
    if (h2_out)
        for (int e = tid; e < TILE_M * 32; e += 256) {
            const int rr = e >> 5, c = (e & 31) * 4;
            if (r0 + rr < R) *reinterpret_cast<float4*>(h2_out + (size_t)(r0 + rr) * D + c) = *reinterpret_cast<const float4*>(&Hs[rr * LDP + c]);
        }
    f32x16 acc[1];
    zero_acc(acc);
    gemm32p<1, 16>(Hs, LDP, D, Wpack, D, 32 * w, 0, acc, bf);
    const int col = 32 * w + (lane & 31);
    const float bvv = bo[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, lane);
        const int gr = r0 + row;
        if (gr < R)
            y_out[(size_t)gr * D + col] = (acc[0][r] + bvv) * drop_mul(d5, (uint32_t)(gr * D + col)) + Rs[row * LDP + col];
    }
}
void launch_attn_out_fwd(const float* att, const float* x, const float* ln_g, const float* ln_b, const float* Wpack,
                         const float* bo, float* r_out, float* h2_out, float* y_out, int R, Drop d3, Drop d4, Drop d5,
                         hipStream_t s) {
    {
        const size_t shm_sp = 0;
        VSL_LAUNCH(k_attn_out_fwd, dim3((R + TILE_M - 1) / TILE_M), dim3(256), shm_sp, s, att, x, ln_g, ln_b, Wpack, bo, r_out,
                       h2_out, y_out, R, d3, d4, d5);
    }
}

// =========================================================================================================
// a10  CQAttention (:208-243), three kernels.
//  (1) k_cq_score: trilinear score on dropped-out C, Q (:237-242) for a 32-row tile + row softmax over the
//      query words (:225).  Writes raw score S (B,T,Lq) and S_row (B,T,Lq).
//  (2) k_cq_col  : per sample: column softmax over clips (:226) -> S_col (B,T,Lq);  M = S_col^T C (Lq,128)
//      (re-association of (S_row S_col^T) C = S_row (S_col^T C): O(T Lq d) instead of O(T^2 d), fp32 rounding only);
//      plus a11's WeightedPool (:253-259) and the per-sample bias  pb = W2 pooled + b  of CQConcatenate (:268-274).
//  (3) k_cq_out  : c2q = S_row Q, q2c = S_row M, concat [C, c2q, C*c2q, C*q2c] (:231) -> Conv1D 4d->d (:232).
// =========================================================================================================
// a11's WeightedPool (:253-259) and the pooled half of CQConcatenate's Conv1D (:268-274) for one sample: alpha, pooled, pb = W2 pooled + b.  Query-side only;
// one 256-thread workgroup.  al: [128], pl: [128] floats of LDS.
struct CqPool { const float *pool_w, *Wcat, *bcat; float *alpha, *pooled, *pb; };
__device__ __forceinline__ void cq_pool_body(const float* __restrict__ Qf, const float* __restrict__ qmask, const CqPool& cp, int b, int Lq, float* al, float* pl) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const size_t qrow = (size_t)b * Lq;
    const float* pool_w = cp.pool_w; const float* Wcat = cp.Wcat; const float* bcat = cp.bcat;
    float* alpha = cp.alpha; float* pooled = cp.pooled; float* pb = cp.pb;
    // ---- WeightedPool: alpha = softmax_j(Q[j].w + mask) ; pooled = sum_j alpha_j Q[j]
    for (int jj = w; jj < Lq; jj += 4) {
        const float* row = Qf + (qrow + jj) * D;
        const float d = wave_sum(row[lane] * pool_w[lane] + row[lane + 64] * pool_w[lane + 64]);
        if (lane == 0) al[jj] = d + (1.f - qmask[qrow + jj]) * MASK_VALUE;
    }
    __syncthreads();
    if (w == 0) {                              // a lane owns words lane and lane + 64
        const float v0 = lane < Lq ? al[lane] : -3.0e38f, v1 = lane + 64 < Lq ? al[lane + 64] : -3.0e38f;
        const float mx = wave_max(fmaxf(v0, v1));
        const float e0 = lane < Lq ? __expf(v0 - mx) : 0.f, e1 = lane + 64 < Lq ? __expf(v1 - mx) : 0.f;
        const float inv = 1.0f / wave_sum(e0 + e1);
        if (lane < Lq) { al[lane] = e0 * inv; alpha[qrow + lane] = e0 * inv; }
        if (lane + 64 < Lq) { al[lane + 64] = e1 * inv; alpha[qrow + lane + 64] = e1 * inv; }
    }
    __syncthreads();
    if (tid < D) {
        float acc = 0.f;
        for (int j0 = 0; j0 < Lq; j0 += 8) {        // eight rows in flight; same summation order as one row at a time
            float qv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) qv[u] = Qf[(qrow + min(j0 + u, Lq - 1)) * D + tid];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (j0 + u < Lq) acc += al[j0 + u] * qv[u];
        }
        pl[tid] = acc;
        pooled[(size_t)b * D + tid] = acc;
    }
    __syncthreads();
    // ---- pb[o] = sum_c Wcat[o][128 + c] * pooled[c] + bcat[o]     (second half of the 2d -> d Conv1D); 8 lanes per row
    for (int o = tid >> 3; o < D; o += 32) {
        const int sub = tid & 7;
        const float* wr = Wcat + (size_t)o * 2 * D + D + sub * 4;
        float d = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 wv = *reinterpret_cast<const float4*>(wr + 32 * q);
            const float4 pv = *reinterpret_cast<const float4*>(pl + sub * 4 + 32 * q);
            d += wv.x * pv.x + wv.y * pv.y + wv.z * pv.z + wv.w * pv.w;
        }
        d = grp8_sum(d);
        if (sub == 0) pb[(size_t)b * D + o] = d + bcat[o];
    }
}

template <int NU>   // words per lane of the row softmax: 8 (Lq <= 64), 12 (<= 96) or 16 (<= MAX_LQ = 128)
__global__ __launch_bounds__(256) void k_cq_score(const float* __restrict__ C, const float* __restrict__ Qf,
                                                  const float* __restrict__ qmask, const float* __restrict__ w4C,
                                                  const float* __restrict__ w4Q, const float* __restrict__ w4mlu,
                                                  float* __restrict__ S, float* __restrict__ Srow, int T, int Lq, int b_off,
                                                  Drop dc, Drop dq, CqPool cp) {
    // cp.pool_w != nullptr: ONE extra workgroup per sample (blockIdx.x = number of tiles) runs the WeightedPool / pooled-bias path (the column kernel
    // is folded into k_cq_out then: launch_cq_out)
    if (cp.pool_w && blockIdx.x == gridDim.x - 1) {          // (block-uniform)
        extern __shared__ __attribute__((aligned(16))) float psm[];
        cq_pool_body(Qf, qmask, cp, blockIdx.y, Lq, psm, psm + 128);
        return;
    }
    // S[i][j] = Cd[i].w4C + Qd[j].w4Q + (Cd[i] * w4mlu).Qd[j]   (:233-243) on a 32-clip tile; Cd / Qd = dropped-out C / Q.
    // The trilinear term is a 32 x 32 MFMA tile per 32 query words with K = 128 split over the four waves (partial tiles
    // summed in wave order through LDS); the two rank-1 terms and the row softmax use 8 lanes per row.
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int NTJ = (Lq + 31) >> 5, PJ = 32 * NTJ + 1;
    float* Cs = smem;                         // [32][LDP]  Cd, then Cd * w4mlu
    float* Qs = Cs + TILE_M * LDP;            // [32 NTJ][LDP]  Qd (rows >= Lq zero)
    float* s0 = Qs + 32 * NTJ * LDP;          // [32]
    float* s1 = s0 + TILE_M;                  // [32 NTJ]
    float* Pp = s1 + 32 * NTJ;                // [4][32][PJ] per-wave partial tiles
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, hh = lane >> 5;
    const int b = blockIdx.y, t0 = blockIdx.x * TILE_M;
    const size_t crow = (size_t)b * T, qrow = (size_t)b * Lq;
    {   // one batch of 16-byte loads for both tiles, dropout applied on the way into LDS
        float4 cv[4], qv[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + q * 256, rr = e >> 5, c = (e & 31) * 4;
            const int t = min(t0 + rr, T - 1);
            cv[q] = *reinterpret_cast<const float4*>(C + (crow + t) * D + c);
        }
        auto load_q = [&](int q0) {               // query rows 8 q0 .. : 8 float4 per thread (64 words) per pass
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = tid + (q0 + q) * 256, j = e >> 5, c = (e & 31) * 4;
                qv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q0 + q < 4 * NTJ) qv[q] = *reinterpret_cast<const float4*>(Qf + (qrow + min(j, Lq - 1)) * D + c);
            }
        };
        auto store_q = [&](int q0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (q0 + q < 4 * NTJ) {
                    const int e = tid + (q0 + q) * 256, j = e >> 5, c = (e & 31) * 4;
                    float4 v = qv[q];
                    if (j < Lq) {
                        if (dq.thresh) {
                            const uint32_t base = (uint32_t)(((b + b_off) * Lq + j) * D + c);
                            v.x *= drop_keep_scale(dq, base); v.y *= drop_keep_scale(dq, base + 1);
                            v.z *= drop_keep_scale(dq, base + 2); v.w *= drop_keep_scale(dq, base + 3);
                        }
                    } else v = make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(&Qs[j * LDP + c]) = v;
                }
            }
        };
        load_q(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + q * 256, rr = e >> 5, c = (e & 31) * 4;
            const int t = t0 + rr;
            float4 v = cv[q];
            if (t < T) {
                if (dc.thresh) {
                    const uint32_t base = (uint32_t)(((b + b_off) * T + t) * D + c);
                    v.x *= drop_keep_scale(dc, base); v.y *= drop_keep_scale(dc, base + 1);
                    v.z *= drop_keep_scale(dc, base + 2); v.w *= drop_keep_scale(dc, base + 3);
                }
            } else v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&Cs[rr * LDP + c]) = v;
        }
        store_q(0);
        for (int q0 = 8; q0 < 4 * NTJ; q0 += 8) { load_q(q0); store_q(q0); }     // queries beyond 64 words (ActivityNet: up to 82)
    }
    __syncthreads();
    {   // s0[i] = Cd[i] . w4C ; s1[j] = Qd[j] . w4Q ; then Cd *= w4mlu in place.  8 lanes per row, 32 rows per pass.
        const int sub = tid & 7, rr = tid >> 3;
        float4 wc[4], wq[4], wm[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wc[k] = *reinterpret_cast<const float4*>(w4C + sub * 4 + 32 * k);
            wq[k] = *reinterpret_cast<const float4*>(w4Q + sub * 4 + 32 * k);
            wm[k] = *reinterpret_cast<const float4*>(w4mlu + sub * 4 + 32 * k);
        }
        float d = 0.f;
        float* crp = Cs + rr * LDP + sub * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float4 v = *reinterpret_cast<const float4*>(crp + 32 * k);
            d += v.x * wc[k].x + v.y * wc[k].y + v.z * wc[k].z + v.w * wc[k].w;
            v.x *= wm[k].x; v.y *= wm[k].y; v.z *= wm[k].z; v.w *= wm[k].w;
            *reinterpret_cast<float4*>(crp + 32 * k) = v;
        }
        d = grp8_sum(d);
        if (sub == 0) s0[rr] = d;
        for (int jb = 0; jb < 32 * NTJ; jb += 32) {
            const float* qrp = Qs + (jb + rr) * LDP + sub * 4;
            float e = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(qrp + 32 * k);
                e += v.x * wq[k].x + v.y * wq[k].y + v.z * wq[k].z + v.w * wq[k].w;
            }
            e = grp8_sum(e);
            if (sub == 0) s1[jb + rr] = e;
        }
    }
    __syncthreads();
    for (int nb = 0; nb < NTJ; nb += 2) {   // trilinear term: wave w contracts channels 32w .. 32w+31 ; two 32-word tiles per pass
        f32x16 acc[2];
        zero_acc(acc);
        const int i = lane & 31;
        const float* arow = Cs + i * LDP + 4 * hh;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            const int kb = 4 * w + kq;
            const float4 av = *reinterpret_cast<const float4*>(arow + kb * 8);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if (nb + nt < NTJ) {
                    const float4 bv = *reinterpret_cast<const float4*>(Qs + (32 * (nb + nt) + i) * LDP + kb * 8 + 4 * hh);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[nt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            if (nb + nt < NTJ) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Pp[(w * TILE_M + acc_row(r, lane)) * PJ + 32 * (nb + nt) + i] = acc[nt][r];
            }
    }
    __syncthreads();
    {   // raw score + row softmax over the query words (dim=2, :225) with the query mask; 8 lanes per clip
        const int sub = tid & 7, rr = tid >> 3;
        const int t = t0 + rr;
        float raw[NU], v[NU];
        float mx = -3.0e38f;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int j = sub + 8 * u;
            raw[u] = 0.f; v[u] = -3.0e38f;
            if (j < Lq) {
                raw[u] = Pp[rr * PJ + j] + Pp[(TILE_M + rr) * PJ + j] + Pp[(2 * TILE_M + rr) * PJ + j] + Pp[(3 * TILE_M + rr) * PJ + j] +
                         s0[rr] + s1[j];
                v[u] = raw[u] + (1.f - qmask[qrow + j]) * MASK_VALUE;
                mx = fmaxf(mx, v[u]);
            }
        }
        mx = fmaxf(mx, lane_xor1(mx)); mx = fmaxf(mx, lane_xor2(mx)); mx = fmaxf(mx, lane_half_mirror(mx));
        float sm = 0.f;
#pragma unroll
        for (int u = 0; u < NU; ++u) { v[u] = (sub + 8 * u < Lq) ? __expf(v[u] - mx) : 0.f; sm += v[u]; }
        const float inv = 1.0f / grp8_sum(sm);
        if (t < T) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int j = sub + 8 * u;
                if (j < Lq) { S[(crow + t) * Lq + j] = raw[u]; Srow[(crow + t) * Lq + j] = v[u] * inv; }
            }
        }
    }
}
void launch_cq_score(const float* C, const float* Qf, const float* qmask, const float* w4C, const float* w4Q,
                     const float* w4mlu, float* S, float* Srow, int B, int T, int Lq, int b_off, Drop dc, Drop dq,
                     hipStream_t s, const CqPoolArgs* pool) {
    CqPool cp{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (pool) cp = CqPool{pool->pool_w, pool->Wcat, pool->bcat, pool->alpha, pool->pooled, pool->pb};
    const int NTJ = (Lq + 31) / 32;
    const size_t shm = (size_t)((TILE_M + 32 * NTJ) * LDP + TILE_M + 32 * NTJ + 4 * TILE_M * (32 * NTJ + 1)) * sizeof(float);
    static size_t ok8 = 0, ok12 = 0, ok16 = 0;
    const dim3 grid((T + TILE_M - 1) / TILE_M + (pool ? 1 : 0), B);
    if (Lq <= 64) {
        ensure_dynamic_lds((const void*)k_cq_score<8>, shm, ok8, "k_cq_score<8>");
        VSL_LAUNCH(k_cq_score<8>, grid, dim3(256), shm, s, C, Qf, qmask, w4C, w4Q, w4mlu, S, Srow, T, Lq, b_off, dc, dq, cp);
    } else if (Lq <= 96) {
        ensure_dynamic_lds((const void*)k_cq_score<12>, shm, ok12, "k_cq_score<12>");
        VSL_LAUNCH(k_cq_score<12>, grid, dim3(256), shm, s, C, Qf, qmask, w4C, w4Q, w4mlu, S, Srow, T, Lq, b_off, dc, dq, cp);
    } else {
        ensure_dynamic_lds((const void*)k_cq_score<16>, shm, ok16, "k_cq_score<16>");
        VSL_LAUNCH(k_cq_score<16>, grid, dim3(256), shm, s, C, Qf, qmask, w4C, w4Q, w4mlu, S, Srow, T, Lq, b_off, dc, dq, cp);
    }
}

__global__ __launch_bounds__(256) void k_cq_col(const float* __restrict__ C, const float* __restrict__ Qf,
                                                const float* __restrict__ S, const float* __restrict__ cmask,
                                                const float* __restrict__ qmask, const float* __restrict__ pool_w,
                                                const float* __restrict__ Wcat, const float* __restrict__ bcat,
                                                float* __restrict__ Scol, float* __restrict__ Mpart, float* __restrict__ alpha,
                                                float* __restrict__ pooled, float* __restrict__ pb, int T, int Lq) {
    // Tile-parallel: workgroup = (32-clip tile, sample).  The column-softmax statistics need every clip of the sample, so
    // each workgroup recomputes them from the sample's whole score matrix (T x Lq floats, L2-resident, all 256 threads);
    // it then writes its S_col tile and the per-tile partial of M = S_col^T C (MFMA, gemm_tn); k_cq_out adds the partials.
    // One EXTRA workgroup per sample (blockIdx.x = number of tiles) runs the small WeightedPool / pooled-bias path beside the tiles
    // (it used to ride on tile 0 and made that workgroup the critical one).
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LQ1 = Lq + 1;
    float* Cs = smem;                         // [32][LDP]  C tile
    float* Ss = Cs + TILE_M * LDP;            // [32][LQ1]  S_col tile (+ slack for the 32-wide over-read of gemm_tn)
    constexpr int JS = 128;                   // word stride of the statistics arrays (Lq <= MAX_LQ <= 128)
    float* redm = Ss + TILE_M * LQ1 + 72;     // [8][JS] partial column maxima
    float* reds = redm + 8 * JS;              // [8][JS] partial column sums
    float* cmax = reds + 8 * JS;              // [JS]
    float* cinv = cmax + JS;                  // [JS]
    float* al = cinv + JS;                    // [JS]
    float* pl = al + JS;                      // [128]
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b = blockIdx.y, tl = blockIdx.x, t0 = tl * TILE_M, ntile = gridDim.x - 1;
    const size_t crow = (size_t)b * T, qrow = (size_t)b * Lq;
    if (tl < ntile) {                           // block-uniform
    load_tile128(Cs, C + crow * D, t0, TILE_M, T);
    // ---- column statistics over all T clips: thread = (word j, part); a part walks clips part, part + np, ... -- eight loads in
    //      flight at a time (one load per trip of a data-dependent loop costs a full memory latency per clip)
    const int JW = Lq <= 32 ? 32 : Lq <= 64 ? 64 : 128, np = 256 / JW;
    const int j = tid & (JW - 1), part = tid / JW;
    auto walk = [&](auto&& fold) {
        if (j >= Lq) return;
        for (int i0 = part; i0 < T; i0 += 8 * np) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u * np, T - 1);
                v[u] = S[(crow + i) * Lq + j] + (1.f - cmask[crow + i]) * MASK_VALUE;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + u * np < T) fold(v[u]);
        }
    };
    {
        float mx = -3.0e38f;
        walk([&](float v) { mx = fmaxf(mx, v); });
        redm[part * JS + j] = mx;
    }
    __syncthreads();
    float gm = -3.0e38f;
    for (int q = 0; q < np; ++q) gm = fmaxf(gm, redm[q * JS + j]);
    {
        float sm = 0.f;
        walk([&](float v) { sm += __expf(v - gm); });
        reds[part * JS + j] = sm;
    }
    __syncthreads();
    if (tid < JW) {
        float sm = 0.f;
        for (int q = 0; q < np; ++q) sm += reds[q * JS + j];
        cmax[j] = gm;
        cinv[j] = 1.0f / sm;
    }
    if (tid < 72) Ss[TILE_M * LQ1 + tid] = 0.f;          // finite values where gemm_tn over-reads
    __syncthreads();
    // ---- S_col tile (:226-227), written out and kept in LDS with a zero pad column
    for (int e = tid; e < TILE_M * LQ1; e += 256) {
        const int i = e / LQ1, jj = e - i * LQ1;
        float v = 0.f;
        if (jj < Lq && t0 + i < T) {
            const size_t o = (crow + t0 + i) * Lq + jj;
            v = __expf(S[o] + (1.f - cmask[crow + t0 + i]) * MASK_VALUE - cmax[jj]) * cinv[jj];
            Scol[o] = v;
        }
        Ss[e] = v;
    }
    __syncthreads();
    // ---- partial of M[j][c] = sum_t S_col[t][j] C[t][c] over this tile's clips
    {
        float* mp = Mpart + (size_t)(b * ntile + tl) * Lq * D;
        const int col = 32 * w + (lane & 31);
        const int NTJ = (Lq + 31) >> 5;
        for (int nt = 0; nt < NTJ; ++nt) {
            f32x16 acc[1];
            zero_acc(acc);
            gemm_tn_p<1, TILE_M>(Ss, LQ1, 32 * nt, Cs, LDP, 32 * w, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jj = 32 * nt + acc_row(r, lane);
                if (jj < Lq) mp[(size_t)jj * D + col] = acc[0][r];
            }
        }
    }
    return;
    }
    cq_pool_body(Qf, qmask, CqPool{pool_w, Wcat, bcat, alpha, pooled, pb}, b, Lq, al, pl);
}
bool cq_col_folds(int T, int Lq) { return T <= 4 * TILE_M && Lq <= 32; }      // (k_cq_out's fold: at most four tiles per sample, one 32-word tile of M)
void launch_cq_col(const float* C, const float* Qf, const float* S, const float* cmask, const float* qmask,
                   const float* pool_w, const float* Wcat, const float* bcat, float* Scol, float* Mpart, float* alpha,
                   float* pooled, float* pb, int B, int T, int Lq, hipStream_t s) {
    const size_t shm = (size_t)(TILE_M * LDP + TILE_M * (Lq + 1) + 72 + 2 * 8 * 128 + 3 * 128 + D) * sizeof(float);
    VSL_LAUNCH(k_cq_col, dim3((T + TILE_M - 1) / TILE_M + 1, B), dim3(256), shm, s, C, Qf, S, cmask, qmask, pool_w, Wcat, bcat, Scol,
                       Mpart, alpha, pooled, pb, T, Lq);
}

// CQConcatenate + HighLightLayer + gating (a11 / a12, VSLNet_t7.py:60) ride on the same kernel (row-local on the tile just produced)
struct CqCatFuse {
    const float *W1pack, *pb, *wh, *bh, *vmask;
    float *f2, *hscore, *gated;
    // the column kernel folded in (T <= 128, Lq <= 32: launch_cq_out): every workgroup computes its SAMPLE's column-softmax statistics and
    // M = S_col^T C over all (at most four) tiles itself instead of summing k_cq_col's per-tile partials -- 4 x 16 KB of L2 reads and four small
    // MFMA products per workgroup against a launch + a kernel boundary on the dependent chain.  S == nullptr: off.
    const float *S, *cmask;
    float* Scol;
};
__global__ __launch_bounds__(256) void k_cq_out(const float* __restrict__ C, const float* __restrict__ Qf,
                                                const float* __restrict__ Srow, const float* __restrict__ Mpart,
                                                float* __restrict__ M, const float* __restrict__ Wpack,
                                                const float* __restrict__ bias, float* __restrict__ cat_out,
                                                float* __restrict__ out, int T, int Lq, CqCatFuse cf) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LQ1 = Lq + 1;
    // Lq > CQ_BIG_LQ: M lives where the concat tile will be written (it is dead by then, behind one extra barrier): 101 instead of 167 KB
    const bool big = Lq > CQ_BIG_LQ;
    const int cat_floats = big ? max(TILE_M * CATP, Lq * LDP) : TILE_M * CATP;
    float* Cat = smem;                        // [32][CATP]
    float* Cs = Cat + cat_floats;             // [32][LDP]
    float* Ms = big ? Cat : Cs + TILE_M * LDP;   // [Lq][LDP]   M = S_col^T C, summed here from k_cq_col's per-tile partials
    float* Ss = Cs + TILE_M * LDP + (big ? 0 : Lq * LDP);   // [32][LQ1]   S_row tile, zero pad column
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, hh = lane >> 5;
    const int b = blockIdx.y, t0 = blockIdx.x * TILE_M, ntile = gridDim.x;
    const size_t crow = (size_t)b * T, qrow = (size_t)b * Lq;
    BFrag<1, 16> bf;
    load_tile128(Cs, C + crow * D, t0, TILE_M, T);
    if (cf.S) {                                // (block-uniform) k_cq_col's work for the whole sample, in the concat tile's LDS (dead until the concat)
        constexpr int JS = 128;
        float* Ct = Cat;                       // [32][LDP]   C tile of the pass
        float* Sc = Ct + TILE_M * LDP;         // [32][LQ1] + 72   S_col tile (+ slack for the 32-wide over-read of gemm_tn)
        float* redm = Sc + TILE_M * LQ1 + 72;  // [8][JS]
        float* reds = redm + 8 * JS;           // [8][JS]
        float* cmax = reds + 8 * JS;           // [JS]
        float* cinv = cmax + JS;               // [JS]
        const float* S = cf.S;
        const float* cmask = cf.cmask;
        const int JW = 32, np = 256 / JW;      // (Lq <= 32 here)
        const int j = tid & (JW - 1), part = tid / JW;
        auto walk = [&](auto&& fold) {
            if (j >= Lq) return;
            for (int i0 = part; i0 < T; i0 += 8 * np) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = min(i0 + u * np, T - 1);
                    v[u] = S[(crow + i) * Lq + j] + (1.f - cmask[crow + i]) * MASK_VALUE;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (i0 + u * np < T) fold(v[u]);
            }
        };
        {
            float mx = -3.0e38f;
            walk([&](float v) { mx = fmaxf(mx, v); });
            redm[part * JS + j] = mx;
        }
        __syncthreads();
        float gm = -3.0e38f;
        for (int q = 0; q < np; ++q) gm = fmaxf(gm, redm[q * JS + j]);
        {
            float sm = 0.f;
            walk([&](float v) { sm += __expf(v - gm); });
            reds[part * JS + j] = sm;
        }
        __syncthreads();
        if (tid < JW) {
            float sm = 0.f;
            for (int q = 0; q < np; ++q) sm += reds[q * JS + j];
            cmax[j] = gm;
            cinv[j] = 1.0f / sm;
        }
        if (tid < 72) Sc[TILE_M * LQ1 + tid] = 0.f;          // finite values where gemm_tn over-reads
        f32x16 macc[1];
        zero_acc(macc);
        for (int tl = 0; tl < ntile; ++tl) {                 // M[j][c] = sum over ALL clips of the sample, tile by tile (k_cq_col's order within a tile)
            const int tt0 = tl * TILE_M;
            __syncthreads();                                 // the previous pass' product is done with Ct / Sc ; cmax / cinv are written
            load_tile128(Ct, C + crow * D, tt0, TILE_M, T);
            for (int e = tid; e < TILE_M * LQ1; e += 256) {
                const int i = e / LQ1, jj = e - i * LQ1;
                float v = 0.f;
                if (jj < Lq && tt0 + i < T) {
                    const size_t o = (crow + tt0 + i) * Lq + jj;
                    v = __expf(S[o] + (1.f - cmask[crow + tt0 + i]) * MASK_VALUE - cmax[jj]) * cinv[jj];
                    if (tl == (int)blockIdx.x) cf.Scol[o] = v;       // this workgroup's own tile (saved for the backward)
                }
                Sc[e] = v;
            }
            __syncthreads();
            gemm_tn_p<1, TILE_M>(Sc, LQ1, 0, Ct, LDP, 32 * w, macc);
        }
        {
            const int colm = 32 * w + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jj = acc_row(r, lane);
                if (jj < Lq) {
                    Ms[jj * LDP + colm] = macc[0][r];
                    if (blockIdx.x == 0) M[(qrow + jj) * D + colm] = macc[0][r];     // saved for the backward
                }
            }
        }
        __syncthreads();                                     // Cat's region is free for the concat again ; Ms is complete
    } else
    for (int e = tid; e < Lq * (D / 4); e += 256) {
        const int j = e >> 5, c4 = (e & 31) * 4;
        float4 sm = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = 0; t < ntile; ++t) {
            const float4 v = *reinterpret_cast<const float4*>(Mpart + ((size_t)(b * ntile + t) * Lq + j) * D + c4);
            sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w;
        }
        *reinterpret_cast<float4*>(Ms + j * LDP + c4) = sm;
        if (blockIdx.x == 0) *reinterpret_cast<float4*>(M + (qrow + j) * D + c4) = sm;     // saved for the backward
    }
    for (int e = tid; e < TILE_M * LQ1; e += 256) {
        const int rr = e / LQ1, j = e - rr * LQ1;
        Ss[e] = (j < Lq && t0 + rr < T) ? Srow[(crow + t0 + rr) * Lq + j] : 0.f;
    }
    bfrag_load(bf, Wpack, D, 32 * w, 0, 0, 4 * D / 8);
    __syncthreads();
    // c2q = S_row Q, q2c = S_row M (:229-230) on the matrix cores (K = Lq, wave = 32 channels), then the concat tile (:231)
    const int col = 32 * w + (lane & 31);
    {
        f32x16 c2q[1], q2c[1];
        zero_acc(c2q);
        zero_acc(q2c);
        const float* sa = Ss + (lane & 31) * LQ1 + hh;
        for (int jc = 0; jc < Lq; jc += 16) {
            float sv[8], qv[8], mv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = jc + 2 * u + hh;
                const bool ok = j < Lq;
                sv[u] = ok ? sa[jc + 2 * u] : 0.f;
                qv[u] = ok ? Qf[(qrow + j) * D + col] : 0.f;
                mv[u] = ok ? Ms[j * LDP + col] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (jc + 2 * u < Lq) {
                    c2q[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[u], qv[u], c2q[0], 0, 0, 0);
                    q2c[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[u], mv[u], q2c[0], 0, 0, 0);
                }
            }
        }
        if (big) __syncthreads();                  // block-uniform: every wave is done with M before the concat tile overwrites it
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = acc_row(r, lane);
            const float cv = Cs[rr * LDP + col];
            float* o = Cat + rr * CATP;
            o[col] = cv;
            o[D + col] = c2q[0][r];
            o[2 * D + col] = cv * c2q[0][r];
            o[3 * D + col] = cv * q2c[0][r];
        }
    }
    __syncthreads();
    if (cat_out)                                   // saved: A operand (R, 512) of the cqa_linear weight gradient
        for (int e = tid; e < TILE_M * D; e += 256) {
            const int rr = e >> 7, c4 = (e & 127) * 4;
            if (t0 + rr < T)
                *reinterpret_cast<float4*>(cat_out + (crow + t0 + rr) * 4 * D + c4) = *reinterpret_cast<const float4*>(&Cat[rr * CATP + c4]);
        }
    f32x16 acc[1];
    zero_acc(acc);
    gemm32p<1, 16>(Cat, CATP, 4 * D, Wpack, D, 32 * w, 0, acc, bf);
    BFrag<1, 16> bf2;                              // weights of the fused CQConcatenate GEMM: in flight during the epilogue
    bfrag_load(bf2, cf.W1pack, D, 32 * w, 0, 0, D / 8);
    const float bv = bias[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, lane);
        const int t = t0 + row;
        const float v = acc[0][r] + bv;
        if (t < T) out[(crow + t) * D + col] = v;
        Cs[row * LDP + col] = t < T ? v : 0.f;     // the C tile is dead: A operand of the next GEMM
    }
    __syncthreads();
    // ---- f2 = f1 W1^T + (W2 pooled + b) ; h = sigmoid(mask_logits(f2 . w_h + b_h)) ; gated = f2 * h
    float* Fs = Cat;                               // [32][LDP] (the concat tile is dead after the barrier above)
    float* hs = Ss;                                // [32]
    f32x16 a2[1];
    zero_acc(a2);
    gemm32p<1, 16>(Cs, LDP, D, cf.W1pack, D, 32 * w, 0, a2, bf2);
    {
        const float pbv = cf.pb[(size_t)b * D + col];
#pragma unroll
        for (int r = 0; r < 16; ++r) Fs[acc_row(r, lane) * LDP + col] = a2[0][r] + pbv;
    }
    __syncthreads();
    {
        const int rr = tid >> 3, sub = tid & 7;
        const float* row = Fs + rr * LDP + sub * 4;
        float d = 0.f;
#pragma unroll
        for (int jq = 0; jq < 4; ++jq) {
            const float4 fv = *reinterpret_cast<const float4*>(row + 32 * jq);
            const float4 wv = *reinterpret_cast<const float4*>(cf.wh + sub * 4 + 32 * jq);
            d += fv.x * wv.x + fv.y * wv.y + fv.z * wv.z + fv.w * wv.w;
        }
        d = grp8_sum(d);
        if (sub == 0) {
            const int t = t0 + rr;
            float hv = 0.f;
            if (t < T) {
                const float lg = d + cf.bh[0] + (1.f - cf.vmask[crow + t]) * MASK_VALUE;      // mask_logits (:286)
                hv = 1.0f / (1.0f + __expf(-lg));
                cf.hscore[crow + t] = hv;
            }
            hs[rr] = hv;
        }
    }
    __syncthreads();
    for (int e = tid; e < TILE_M * 32; e += 256) {
        const int rr = e >> 5, c = (e & 31) * 4;
        const int t = t0 + rr;
        if (t < T) {
            const float4 v = *reinterpret_cast<const float4*>(&Fs[rr * LDP + c]);
            const float hv = hs[rr];
            *reinterpret_cast<float4*>(cf.f2 + (crow + t) * D + c) = v;
            *reinterpret_cast<float4*>(cf.gated + (crow + t) * D + c) = make_float4(v.x * hv, v.y * hv, v.z * hv, v.w * hv);
        }
    }
}
void launch_cq_out(const float* C, const float* Qf, const float* Srow, const float* Mpart, float* M, const float* Wpack,
                   const float* bias, float* cat_out, float* out, const float* W1pack, const float* pb, const float* wh,
                   const float* bh, const float* vmask, float* f2, float* hscore, float* gated, int B, int T, int Lq, hipStream_t s,
                   const float* S_fold, float* Scol_fold) {
    const size_t shm = (Lq > CQ_BIG_LQ ? (size_t)(std::max(TILE_M * CATP, Lq * LDP) + TILE_M * LDP + TILE_M * (Lq + 1) + 32)
                                       : (size_t)(TILE_M * CATP + TILE_M * LDP + Lq * LDP + TILE_M * (Lq + 1) + 32)) * sizeof(float);
    static size_t lds_ok = 0;
    ensure_dynamic_lds((const void*)k_cq_out, shm, lds_ok, "k_cq_out");
    VSL_LAUNCH(k_cq_out, dim3((T + TILE_M - 1) / TILE_M, B), dim3(256), shm, s, C, Qf, Srow, Mpart, M, Wpack, bias, cat_out, out,
                       T, Lq, CqCatFuse{W1pack, pb, wh, bh, vmask, f2, hscore, gated, S_fold, S_fold ? vmask : nullptr, Scol_fold});
}


// =========================================================================================================
// a14 heads (:328-337, 347-352): logits = mask_logits( Conv1D(d->1)( relu( Conv1D(2d->d)([LN(feat), x]) ) ) )
//   blockIdx.y selects start / end.  `hid` (relu output) is saved for the backward.
// =========================================================================================================
constexpr int HDP = 2 * D + 4;
__global__ __launch_bounds__(256) void k_head_fwd(HeadArgs a0, HeadArgs a1, const float* __restrict__ x,
                                                  const float* __restrict__ vmask, int R) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                     // [32][HDP] = [LN(feat) | x]
    float* Hd = As + TILE_M * HDP;        // [32][LDP] relu(z)
    const HeadArgs a = blockIdx.y == 0 ? a0 : a1;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int r0 = blockIdx.x * TILE_M;
    BFrag<1, 16> bf;
    {
        float4 fv[4], xv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + q * 256;
            const int r = r0 + (e >> 5), c = (e & 31) * 4;
            fv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            xv[q] = fv[q];
            if (r < R) {
                fv[q] = *reinterpret_cast<const float4*>(a.feat + (size_t)r * D + c);
                xv[q] = *reinterpret_cast<const float4*>(x + (size_t)r * D + c);
            }
        }
        bfrag_load(bf, a.W0pack, D, 32 * w, 0, 0, 2 * D / 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + q * 256;
            const int rr = e >> 5, c = (e & 31) * 4;
            *reinterpret_cast<float4*>(&As[rr * HDP + c]) = fv[q];
            *reinterpret_cast<float4*>(&As[rr * HDP + D + c]) = xv[q];
        }
    }
    __syncthreads();
    if (a.ln_g) {                          // transformer head: LayerNorm on the encoder features (:347-348); rnn: none
        ln_tile(As, TILE_M, HDP, a.ln_g, a.ln_b, Drop{0u, 0u, 1.f}, 0);
        __syncthreads();
    }
    if (a.lnfeat)
        for (int e = tid; e < TILE_M * 32; e += 256) {
            const int rr = e >> 5, c = (e & 31) * 4;
            if (r0 + rr < R) *reinterpret_cast<float4*>(a.lnfeat + (size_t)(r0 + rr) * D + c) = *reinterpret_cast<const float4*>(&As[rr * HDP + c]);
        }
    f32x16 acc[1];
    zero_acc(acc);
    gemm32p<1, 16>(As, HDP, 2 * D, a.W0pack, D, 32 * w, 0, acc, bf);
    const int col = 32 * w + (lane & 31);
    const float bv = a.b0[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, lane);
        const int gr = r0 + row;
        const float hv = fmaxf(acc[0][r] + bv, 0.f);
        Hd[row * LDP + col] = hv;
        if (gr < R) a.hid[(size_t)gr * D + col] = hv;
    }
    __syncthreads();
    {
        const int rr = tid >> 3, sub = tid & 7;
        const float* row = Hd + rr * LDP + sub * 4;
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 fv = *reinterpret_cast<const float4*>(row + 32 * j);
            const float4 wv = *reinterpret_cast<const float4*>(a.w1 + sub * 4 + 32 * j);
            d += fv.x * wv.x + fv.y * wv.y + fv.z * wv.z + fv.w * wv.w;
        }
        d = grp8_sum(d);
        const int gr = r0 + rr;
        if (sub == 0 && gr < R) a.logits[gr] = d + a.b1[0] + (1.f - vmask[gr]) * MASK_VALUE;
    }
}
void launch_head_fwd(const HeadArgs& a0, const HeadArgs& a1, const float* x, const float* vmask, int R, hipStream_t s) {
    const size_t shm = (size_t)(TILE_M * HDP + TILE_M * LDP) * sizeof(float);
    {
        const size_t shm_sp = shm;
        VSL_LAUNCH(k_head_fwd, dim3((R + TILE_M - 1) / TILE_M, 2), dim3(256), shm_sp, s, a0, a1, x, vmask, R);
    }
}

}  // namespace vsl
