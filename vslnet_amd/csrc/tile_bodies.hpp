// Bodies of row-tile kernels (one 32-row tile of the flattened (B*L, 128) activation per 256 threads) as device functions, so that a kernel whose
// workgroup owns the same tile can go on with them instead of ending: a tile-local fusion removes a kernel boundary of the dependent chain (the
// gap, the store drain, the cold start) and needs no other workgroup's data.  Hosts: k_head_bwd (-> attention-output backward of the second
// predictor pass), k_convblock_bwd (-> attention-output backward of the pass below / -> CQConcatenate backward).  A 512-thread host calls them
// with active = (tid < 256): the other waves only join the barriers.
#pragma once
#include "common.hpp"
#include "launch.hpp"

namespace vsl {

// one 32-row tile.  lds_dy != nullptr: the incoming gradient tile is already in LDS (stride LDP, rows >= R zero; dy2 is then ignored) -- the
// fused caller (k_head_bwd) hands over what it has just computed.  Gs / Xs: two [32][LDP] tiles, neither of them lds_dy.
// (dy2, when given, is added from memory in both modes)
__device__ __forceinline__ void attn_out_bwd_tile(const AttnOutBwdArgs& a, const float* lds_dy, float* Gs, float* Xs, int r0, int R, bool active = true) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    BFrag<1, 16> bf;
    LnResid lres;
    if (active) {
        float4 a1[4], a2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + q * 256;
            const int r = r0 + (e >> 5), c = (e & 31) * 4;
            a1[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            a2[q] = a1[q];
            if (lds_dy) a1[q] = *reinterpret_cast<const float4*>(lds_dy + (e >> 5) * LDP + c);
            else if (r < R) a1[q] = *reinterpret_cast<const float4*>(a.dy + (size_t)r * D + c);
            if (a.dy2 && r < R) a2[q] = *reinterpret_cast<const float4*>(a.dy2 + (size_t)r * D + c);
        }
        bfrag_load(bf, a.WTpack, D, 32 * w, 0, 0, D / 8);     // after the tile (in-order return), before the LN input tile
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + q * 256;
            const int rr = e >> 5, c = (e & 31) * 4;
            const int r = r0 + rr;
            float4 v = make_float4(a1[q].x + a2[q].x, a1[q].y + a2[q].y, a1[q].z + a2[q].z, a1[q].w + a2[q].w);
            if (r < R) {
                if (a.d5.thresh) {
                    const uint32_t base = (uint32_t)(r * D + c);
                    v.x *= drop_mul(a.d5, base); v.y *= drop_mul(a.d5, base + 1);
                    v.z *= drop_mul(a.d5, base + 2); v.w *= drop_mul(a.d5, base + 3);
                }
                *reinterpret_cast<float4*>(a.g_o + (size_t)r * D + c) = v;       // G operand of the out_layer weight gradient
            }
            *reinterpret_cast<float4*>(&Gs[rr * LDP + c]) = v;
        }
    }
    // residual path of LN2's backward: the same rows again (L2 hits, or the LDS tile), used last
    if (active) {
        if (lds_dy) {
            const int sub = tid & 7, rr = tid >> 3;
            LnResid more;
            ln_resid_prefetch(more, nullptr, a.dy2, r0, R);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = *reinterpret_cast<const float4*>(lds_dy + rr * LDP + sub * 4 + 32 * j);
                lres.v[j] = make_float4(v.x + more.v[j].x, v.y + more.v[j].y, v.z + more.v[j].z, v.w + more.v[j].w);
            }
        } else ln_resid_prefetch(lres, a.dy, a.dy2, r0, R);
        load_tile128(Xs, a.r_in, r0, TILE_M, R);              // only needed by the LayerNorm backward after the GEMM
    }
    __syncthreads();
    f32x16 acc[1];
    zero_acc(acc);
    if (active) gemm32p<1, 16>(Gs, LDP, D, a.WTpack, D, 32 * w, 0, acc, bf);
    __syncthreads();
    if (active) {
        const int col = 32 * w + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, lane);
            Gs[row * LDP + col] = acc[0][r] * drop_mul(a.d4, (uint32_t)((r0 + row) * D + col));
        }
    }
    __syncthreads();
    ln_bwd_tile(Gs, Xs, lres, a.ln_g, a.dr, a.p_lng, a.p_lnb, r0, R, nullptr, active);
}

// CrossEntropy seed of logit x at position t (mean over the batch folded into cs = w_loc * inv_batch)
__device__ __forceinline__ float loss_ce_seed(float x, float lse, int t, int label, float cs) { return cs * (expf(x - lse) - (t == label ? 1.f : 0.f)); }
// highlight (weighted BCE, VSLNet_t7.py / layers_t7.py:291-299) seed of score p with label y under mask m
__device__ __forceinline__ float loss_hl_seed(float p, float y, float m, float w_hl, float mask_sum) {
    const float wgt = y == 0.f ? 1.f : 2.f * y;
    // d BCE / dp = (p - y) / max(p (1 - p), 1e-12)   (torch's binary_cross_entropy_backward)
    return w_hl * wgt * m / (mask_sum + 1e-12f) * (p - y) / fmaxf(p * (1.f - p), 1e-12f);
}
// gating + HighLightLayer + CQConcatenate backward (VSLNet_t7.py:60, layers_t7.py:262-289 backward):  dgated = dg0 + dg1 + dg2 ;
// dlogit = (sum_c dgated * f2 + dh_loss) * h (1 - h) ; df2 = dgated * h + dlogit * wh ; df1 = df2 W1 ; partials of wh / bh.
// lds_dg0 != nullptr: dg0's tile is in LDS (stride LDP; rows >= R zero).  Gs / Fs: two [32][LDP] tiles (neither of them lds_dg0), dlg: 32 floats.
__device__ __forceinline__ void cqcat_bwd_tile(const CqcatBwdArgs& a, const float* lds_dg0, float* Gs, float* Fs, float* dlg, int r0, int R, bool active = true) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    if (active) {
        float4 g0[4], g1[4], g2[4], fv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + q * 256;
            const int r = r0 + (e >> 5), c = (e & 31) * 4;
            g0[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            g1[q] = g0[q]; g2[q] = g0[q]; fv[q] = g0[q];
            if (lds_dg0) g0[q] = *reinterpret_cast<const float4*>(lds_dg0 + (e >> 5) * LDP + c);
            if (r < R) {
                if (!lds_dg0) g0[q] = *reinterpret_cast<const float4*>(a.dg0 + (size_t)r * D + c);
                if (a.dg1) g1[q] = *reinterpret_cast<const float4*>(a.dg1 + (size_t)r * D + c);
                if (a.dg2) g2[q] = *reinterpret_cast<const float4*>(a.dg2 + (size_t)r * D + c);
                fv[q] = *reinterpret_cast<const float4*>(a.f2 + (size_t)r * D + c);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + q * 256;
            const int rr = e >> 5, c = (e & 31) * 4;
            *reinterpret_cast<float4*>(&Gs[rr * LDP + c]) = make_float4(g0[q].x + g1[q].x + g2[q].x, g0[q].y + g1[q].y + g2[q].y,
                                                                       g0[q].z + g1[q].z + g2[q].z, g0[q].w + g1[q].w + g2[q].w);
            *reinterpret_cast<float4*>(&Fs[rr * LDP + c]) = fv[q];
        }
    }
    __syncthreads();
    if (active) {
        const int rr = tid >> 3, sub = tid & 7;
        const int r = r0 + rr;
        float* grow = Gs + rr * LDP + sub * 4;
        const float* frow = Fs + rr * LDP + sub * 4;
        float4 g4[4];
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            g4[j] = *reinterpret_cast<const float4*>(grow + 32 * j);
            const float4 f4 = *reinterpret_cast<const float4*>(frow + 32 * j);
            d += g4[j].x * f4.x + g4[j].y * f4.y + g4[j].z * f4.z + g4[j].w * f4.w;
        }
        d = grp8_sum(d);
        float hv = 0.f, dl = 0.f;
        if (r < R) {
            hv = a.hscore[r];
            float dh = 0.f;                                                      // the highlight loss' seed: computed here (vsl_io.fused_loss) or read
            if (a.h_lab) dh = loss_hl_seed(hv, (float)a.h_lab[r], a.vmask[r], a.w_hl, a.mask_sum);
            else if (a.dh_loss) dh = a.dh_loss[r];
            dl = (d + dh) * hv * (1.f - hv);                                     // sigmoid backward; mask_logits is additive
        }
        // df2 = dgated * h + dlogit * wh
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 wv = *reinterpret_cast<const float4*>(a.wh + sub * 4 + 32 * j);
            *reinterpret_cast<float4*>(grow + 32 * j) = make_float4(g4[j].x * hv + dl * wv.x, g4[j].y * hv + dl * wv.y,
                                                                    g4[j].z * hv + dl * wv.z, g4[j].w * hv + dl * wv.w);
        }
        if (sub == 0) dlg[rr] = dl;
    }
    __syncthreads();
    if (active) {
        if (tid < 128) {
            float acc = 0.f;
            for (int rr = 0; rr < TILE_M; ++rr) acc += dlg[rr] * Fs[rr * LDP + tid];
            a.p_wh[(size_t)blockIdx.x * D + tid] = acc;
        } else if (tid == 128) {
            float acc = 0.f;
            for (int rr = 0; rr < TILE_M; ++rr) acc += dlg[rr];
            a.p_bh[blockIdx.x] = acc;
        }
        for (int e = tid; e < TILE_M * 32; e += 256) {
            const int rr = e >> 5, c = (e & 31) * 4;
            if (r0 + rr < R) *reinterpret_cast<float4*>(a.df2 + (size_t)(r0 + rr) * D + c) = *reinterpret_cast<const float4*>(&Gs[rr * LDP + c]);
        }
        f32x16 acc[1];
        zero_acc(acc);
        {
            BFrag<1, 16> bf;
            bfrag_load(bf, a.W1Tpack, D, 32 * w, 0, 0, D / 8);
            gemm32p<1, 16>(Gs, LDP, D, a.W1Tpack, D, 32 * w, 0, acc, bf);
        }
        const int col = 32 * w + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gr = r0 + acc_row(r, lane);
            if (gr < R) a.df1[(size_t)gr * D + col] = acc[0][r];
        }
    }
}

// One span head (a14, layers_t7.py:328-337, 347-352) on a 32-row tile: logits = mask_logits(Conv1D(d->1)(relu(Conv1D(2d->d)([LN(feat), x])))).
// Run by ONE 256-thread group of a wider workgroup (lt = thread index inside the group; `active` = false: only the barriers), so that two groups do
// the start and the end head side by side.  lds_feat != nullptr: the feature tile is in LDS (stride LDP) -- the end head's features are what the
// hosting attention-block kernel has just produced.  grow0 = global row of tile row 0, nvalid = rows of the tile that exist.
// As: [32][HEAD_LD] ; Hd: [32][LDP].
// KB = weight fragments in flight per wave (k blocks of 8): the 16-wave host has 128 registers per lane.
constexpr int HEAD_LD = 2 * D + 4;
template <int KB>
__device__ __forceinline__ void head_fwd_tile(const HeadArgs& a, const float* __restrict__ x, const float* __restrict__ vmask, float* As, float* Hd,
                                              const float* lds_feat, size_t grow0, int nvalid, int lt, bool active) {
    const int lw = lt >> 6, lane = lt & 63;
    BFrag<1, KB> bf;
    if (active) {
        float4 fv[4], xv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = lt + q * 256, rr = e >> 5, c = (e & 31) * 4;
            fv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            xv[q] = fv[q];
            if (rr < nvalid) {
                fv[q] = lds_feat ? *reinterpret_cast<const float4*>(lds_feat + rr * LDP + c) : *reinterpret_cast<const float4*>(a.feat + (grow0 + rr) * D + c);
                xv[q] = *reinterpret_cast<const float4*>(x + (grow0 + rr) * D + c);
            }
        }
        bfrag_load(bf, a.W0pack, D, 32 * lw, 0, 0, 2 * D / 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = lt + q * 256, rr = e >> 5, c = (e & 31) * 4;
            *reinterpret_cast<float4*>(&As[rr * HEAD_LD + c]) = fv[q];
            *reinterpret_cast<float4*>(&As[rr * HEAD_LD + D + c]) = xv[q];
        }
    }
    __syncthreads();
    if (active && a.ln_g) {                    // LayerNorm on the encoder features (:347-348), 8 lanes per row (as ln_tile)
        const int sub = lt & 7, r = lt >> 3;
        float* row = As + r * HEAD_LD + sub * 4;
        float4 v[4];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = *reinterpret_cast<const float4*>(row + 32 * j); s += sum4(v[j]); }
        const float mu = grp8_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j].x -= mu; v[j].y -= mu; v[j].z -= mu; v[j].w -= mu;
            q += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
        }
        const float rstd = rsqrtf(grp8_sum(q) * (1.0f / D) + LN_EPS);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 gv = *reinterpret_cast<const float4*>(a.ln_g + sub * 4 + 32 * j), bv = *reinterpret_cast<const float4*>(a.ln_b + sub * 4 + 32 * j);
            float4 o;
            o.x = v[j].x * rstd * gv.x + bv.x; o.y = v[j].y * rstd * gv.y + bv.y;
            o.z = v[j].z * rstd * gv.z + bv.z; o.w = v[j].w * rstd * gv.w + bv.w;
            *reinterpret_cast<float4*>(row + 32 * j) = o;
            if (a.lnfeat && r < nvalid) *reinterpret_cast<float4*>(a.lnfeat + (grow0 + r) * D + sub * 4 + 32 * j) = o;
        }
    }
    __syncthreads();
    if (active) {
        f32x16 acc[1];
        zero_acc(acc);
        gemm32p<1, KB>(As, HEAD_LD, 2 * D, a.W0pack, D, 32 * lw, 0, acc, bf);
        const int col = 32 * lw + (lane & 31);
        const float bv = a.b0[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, lane);
            const float hv = fmaxf(acc[0][r] + bv, 0.f);
            Hd[row * LDP + col] = hv;
            if (row < nvalid) a.hid[(grow0 + row) * D + col] = hv;
        }
    }
    __syncthreads();
    if (active) {
        const int rr = lt >> 3, sub = lt & 7;
        const float* row = Hd + rr * LDP + sub * 4;
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 fv = *reinterpret_cast<const float4*>(row + 32 * j);
            const float4 wv = *reinterpret_cast<const float4*>(a.w1 + sub * 4 + 32 * j);
            d += fv.x * wv.x + fv.y * wv.y + fv.z * wv.z + fv.w * wv.w;
        }
        d = grp8_sum(d);
        if (sub == 0 && rr < nvalid) a.logits[grow0 + rr] = d + a.b1[0] + (1.f - vmask[grow0 + rr]) * MASK_VALUE;
    }
}

// Embedding linear, data gradient (layers_t7.py:81-87 backward): dA (rows r0 .. r0 + nrows - 1, K columns) = G WT3, G = a 32-row tile in LDS (stride LDP;
// rows >= nrows are computed and dropped).  Called by ALL 8 waves of a 512-thread workgroup: the tile is split into bf16 planes `pl` (3 x [32][D + 8]),
// wave w owns the 32-column blocks 32 (w & 3) + 256 (w >> 2) + {0, 128} of every 512-column pass.
__device__ __forceinline__ void linear_bwd_data_tile(const float* lds_g, uint16_t* pl, const uint16_t* __restrict__ WT3, float* __restrict__ dA,
                                                     int r0, int nrows, int K, int Kc) {
    constexpr int LD = D + 8;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = tid + q * 512, rr = e >> 5, c = (e & 31) * 4;
        split_store4(pl, LD, TILE_M * LD, rr, c, *reinterpret_cast<const float4*>(lds_g + rr * LDP + c));
    }
    __syncthreads();
    for (int cb = 0; cb < Kc; cb += 512) {
        f32x16 acc[2];
        zero_acc(acc);
        const int col0 = cb + 32 * (w & 3) + 256 * (w >> 2);
        gemm32pl<2>(pl, LD, TILE_M * LD, D, WT3, Kc, col0, D, acc);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int col = col0 + t * D + (lane & 31);
            if (col < K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = acc_row(r, lane);
                    if (row < nrows) dA[(size_t)(r0 + row) * K + col] = acc[t][r];
                }
            }
        }
    }
}

}  // namespace vsl
