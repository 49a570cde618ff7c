// Loss / eval / backward kernels of the VSLNet hot path (gfx950).  The reference has no hand-written backward: it is
// whatever autograd derives from /root/reference/model/layers_t7.py; each kernel names the forward lines it
// differentiates.  Parameter gradients are produced as per-tile / per-row-chunk PARTIAL SLABS and summed by
// k_reduce into the flat gradient bucket (deterministic, no float atomics on global memory).
#include "common.hpp"
#include "launch.hpp"
#include "tile_bodies.hpp"
#include <type_traits>
#include <algorithm>

namespace vsl {

// VSL_DEBUG_TIMING: block 0 / thread 0 of an instrumented kernel stamps the shader clock at its phase boundaries
__device__ long long g_stamps[32];
// The stamps are compiled in only with -DVSL_STAMPS (vslnet_amd.build.build(stamps=True) -> libvslnet_hip_stamps.so): even a disabled
// stamp reads its enable flag from memory and waits for it (vmcnt(0)), which drains every prefetch in flight at that point.
#ifdef VSL_STAMPS
#define STAMP(k) do { if (g_dbg_on && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_stamps[k] = clock64(); } while (0)
#define STAMPS_ON(expr) (expr)
#else
#define STAMP(k) do { } while (0)
#define STAMPS_ON(expr) false
#endif
__device__ int g_dbg_on = 0;
static int dbg_budget(const char* name) {
    static int inited = 0, on = 0;
    if (!inited) { inited = 1; on = getenv("VSL_DEBUG_TIMING") != nullptr; if (on) { int one = 1; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_on), &one, sizeof one); } }
    (void)name;
    return on;
}
static void dbg_report(const char* name, int nst, hipStream_t s, int& left) {
    if (left <= 0) return;
    long long h[32];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof h);
    fprintf(stderr, "[%s cycles]", name);
    for (int i = 1; i < nst; ++i) fprintf(stderr, " %lld", h[i] - h[i - 1]);
    fprintf(stderr, " | total %lld\n", h[nst - 1] - h[0]);
    --left;
}

// =========================================================================================================
// a13 + a16 losses (layers_t7.py:291-299, 365-369) and their seeds d(total)/d(logits), d(total)/d(h_score)
//   total = w_loc * (CE(start) + CE(end)) + w_hl * highlight      (main_t7.py:105-107 uses 1, 5)
//   scratch layout (floats): [0,B) lse_s | [B,2B) lse_e | [2B,3B) ce | [3B,4B) hl numerator | [4B,5B) mask sum
// =========================================================================================================
__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    const int nw = blockDim.x >> 6;
    float r = red[0];
    for (int i = 1; i < nw; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}
__global__ __launch_bounds__(256) void k_loss_a(const float* __restrict__ sl, const float* __restrict__ el,
                                                const float* __restrict__ h, const int64_t* __restrict__ s_lab,
                                                const int64_t* __restrict__ e_lab, const int64_t* __restrict__ h_lab,
                                                const float* __restrict__ vmask, int B, int T, float* __restrict__ scratch) {
    __shared__ float red[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* s = sl + (size_t)b * T;
    const float* e = el + (size_t)b * T;
    float ms = -3.0e38f, me = -3.0e38f;
    for (int t = tid; t < T; t += 256) { ms = fmaxf(ms, s[t]); me = fmaxf(me, e[t]); }
    ms = block_reduce(ms, red, true);
    me = block_reduce(me, red, true);
    float ss = 0.f, se = 0.f, num = 0.f, den = 0.f;
    for (int t = tid; t < T; t += 256) {
        ss += expf(s[t] - ms);
        se += expf(e[t] - me);
        const float m = vmask[(size_t)b * T + t];
        const float y = (float)h_lab[(size_t)b * T + t];
        const float p = h[(size_t)b * T + t];
        const float wgt = y == 0.f ? 1.f : 2.f * y;                         // (:293)
        const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.f - p), -100.f);   // BCELoss log clamp
        num += -(y * lp + (1.f - y) * lq) * wgt * m;
        den += m;
    }
    ss = block_reduce(ss, red, false);
    se = block_reduce(se, red, false);
    num = block_reduce(num, red, false);
    den = block_reduce(den, red, false);
    if (tid == 0) {
        const float lses = ms + logf(ss), lsee = me + logf(se);
        scratch[b] = lses;
        scratch[B + b] = lsee;
        scratch[2 * B + b] = (lses - s[s_lab[b]]) + (lsee - e[e_lab[b]]);
        scratch[3 * B + b] = num;
        scratch[4 * B + b] = den;
    }
}
__global__ __launch_bounds__(256) void k_loss_b(int B, float inv_batch, float mask_sum_override, float w_loc, float w_hl,
                                                const float* __restrict__ scratch, float* __restrict__ losses) {
    __shared__ float red[8];
    float ce = 0.f, num = 0.f, den = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) { ce += scratch[2 * B + b]; num += scratch[3 * B + b]; den += scratch[4 * B + b]; }
    ce = block_reduce(ce, red, false);
    num = block_reduce(num, red, false);
    den = block_reduce(den, red, false);
    if (threadIdx.x == 0) {
        const float d = mask_sum_override > 0.f ? mask_sum_override : den;
        const float loc = ce * inv_batch;                                   // CrossEntropyLoss(mean) twice (:367-368)
        const float hl = num / (d + 1e-12f);                                // (:298)
        losses[0] = loc; losses[1] = hl; losses[2] = w_loc * loc + w_hl * hl; losses[3] = d;
    }
}
// gradient seeds + (block (0,0)) the loss values: every block re-reduces the B per-sample partials itself (a few hundred
// floats, L2-resident), which saves the single-block kernel between k_loss_a and this one on the critical path
__global__ __launch_bounds__(256) void k_loss_c(const float* __restrict__ sl, const float* __restrict__ el,
                                                const float* __restrict__ h, const int64_t* __restrict__ s_lab,
                                                const int64_t* __restrict__ e_lab, const int64_t* __restrict__ h_lab,
                                                const float* __restrict__ vmask, int B, int T, float inv_batch,
                                                float mask_sum_override, float w_loc, float w_hl,
                                                const float* __restrict__ scratch, float* __restrict__ losses,
                                                float* __restrict__ d_sl, float* __restrict__ d_el, float* __restrict__ d_h) {
    __shared__ float red[8];
    float ce = 0.f, num = 0.f, den = 0.f;
    for (int bb = threadIdx.x; bb < B; bb += 256) { ce += scratch[2 * B + bb]; num += scratch[3 * B + bb]; den += scratch[4 * B + bb]; }
    ce = block_reduce(ce, red, false);
    num = block_reduce(num, red, false);
    den = block_reduce(den, red, false);
    const float dsum = mask_sum_override > 0.f ? mask_sum_override : den;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        const float loc = ce * inv_batch;                                   // CrossEntropyLoss(mean) twice (:367-368)
        const float hl = num / (dsum + 1e-12f);                             // (:298)
        losses[0] = loc; losses[1] = hl; losses[2] = w_loc * loc + w_hl * hl; losses[3] = dsum;
    }
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const size_t i = (size_t)b * T + t;
    const float cs = w_loc * inv_batch;
    d_sl[i] = cs * (expf(sl[i] - scratch[b]) - (t == (int)s_lab[b] ? 1.f : 0.f));
    d_el[i] = cs * (expf(el[i] - scratch[B + b]) - (t == (int)e_lab[b] ? 1.f : 0.f));
    const float y = (float)h_lab[i], p = h[i], m = vmask[i];
    const float wgt = y == 0.f ? 1.f : 2.f * y;
    // d BCE / dp = (p - y) / max(p (1 - p), 1e-12)   (torch's binary_cross_entropy_backward)
    d_h[i] = w_hl * wgt * m / (dsum + 1e-12f) * (p - y) / fmaxf(p * (1.f - p), 1e-12f);
}
// log-sum-exp of one sample's T logits by a 256-thread workgroup (every thread gets the result).  k_loss_fused and k_head_bwd's inline seeds
// (vsl_io.fused_loss) both go through this function, so the seeds are the same bits whichever kernel computes them.
__device__ __forceinline__ float loss_sample_lse(const float* __restrict__ x, int T, float* red) {
    const int tid = threadIdx.x;
    float mx = -3.0e38f;
    for (int t = tid; t < T; t += 256) mx = fmaxf(mx, x[t]);
    mx = block_reduce(mx, red, true);
    float ss = 0.f;
    for (int t = tid; t < T; t += 256) ss += expf(x[t] - mx);
    ss = block_reduce(ss, red, false);
    return mx + logf(ss);
}
// One launch when the caller supplies the global mask sum (the data-parallel path and bench.py do): the sample's workgroup writes its
// gradient seeds itself -- they need nothing from other samples then -- and the LAST workgroup to arrive (agent-scope counter) adds the
// per-sample partials in index order, so the loss values do not depend on which one that is.  Saves k_loss_c on the critical chain.
__global__ __launch_bounds__(256) void k_loss_fused(const float* __restrict__ sl, const float* __restrict__ el,
                                                    const float* __restrict__ h, const int64_t* __restrict__ s_lab,
                                                    const int64_t* __restrict__ e_lab, const int64_t* __restrict__ h_lab,
                                                    const float* __restrict__ vmask, int B, int T, float inv_batch, float mask_sum,
                                                    float w_loc, float w_hl, float* __restrict__ scratch, float* __restrict__ losses,
                                                    float* __restrict__ d_sl, float* __restrict__ d_el, float* __restrict__ d_h,
                                                    unsigned* __restrict__ counter) {
    __shared__ float red[8];
    __shared__ unsigned last;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* s = sl + (size_t)b * T;
    const float* e = el + (size_t)b * T;
    float ms = -3.0e38f, me = -3.0e38f;
    for (int t = tid; t < T; t += 256) { ms = fmaxf(ms, s[t]); me = fmaxf(me, e[t]); }
    ms = block_reduce(ms, red, true);
    me = block_reduce(me, red, true);
    float ss = 0.f, se = 0.f, num = 0.f;
    for (int t = tid; t < T; t += 256) {
        ss += expf(s[t] - ms);
        se += expf(e[t] - me);
        const float m = vmask[(size_t)b * T + t];
        const float y = (float)h_lab[(size_t)b * T + t];
        const float p = h[(size_t)b * T + t];
        const float wgt = y == 0.f ? 1.f : 2.f * y;                         // (:293)
        const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.f - p), -100.f);   // BCELoss log clamp
        num += -(y * lp + (1.f - y) * lq) * wgt * m;
    }
    ss = block_reduce(ss, red, false);
    se = block_reduce(se, red, false);
    num = block_reduce(num, red, false);
    const float lses = ms + logf(ss), lsee = me + logf(se);
    const float cs = w_loc * inv_batch;
    const int sb = (int)s_lab[b], eb = (int)e_lab[b];
    for (int t = tid; t < T; t += 256) {
        const size_t i = (size_t)b * T + t;
        d_sl[i] = cs * (expf(s[t] - lses) - (t == sb ? 1.f : 0.f));
        d_el[i] = cs * (expf(e[t] - lsee) - (t == eb ? 1.f : 0.f));
        const float y = (float)h_lab[i], p = h[i], m = vmask[i];
        const float wgt = y == 0.f ? 1.f : 2.f * y;
        d_h[i] = w_hl * wgt * m / (mask_sum + 1e-12f) * (p - y) / fmaxf(p * (1.f - p), 1e-12f);
    }
    if (tid == 0) {
        // the two partials go out WRITE-THROUGH (sc1 stores) and are drained before the arrival is counted; the last arriver reads them with
        // sc1 loads.  No fence on either side: __threadfence() is an L2 write-back + invalidate on this part, ~3.5 us -- twice on the chain
        // of a kernel that has 4 us of work (MI355X_MICROARCH.md, visibility; the protocol of the rnn head's granules)
        __hip_atomic_store(scratch + 2 * B + b, (lses - s[sb]) + (lsee - e[eb]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(scratch + 3 * B + b, num, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(B - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    float ce = 0.f, nm = 0.f;
    for (int bb = tid; bb < B; bb += 256) {                                 // L2 reads: another CU wrote these
        ce += __hip_atomic_load(scratch + 2 * B + bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nm += __hip_atomic_load(scratch + 3 * B + bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ce = block_reduce(ce, red, false);
    nm = block_reduce(nm, red, false);
    if (tid == 0) {
        const float loc = ce * inv_batch, hl = nm / (mask_sum + 1e-12f);
        losses[0] = loc; losses[1] = hl; losses[2] = w_loc * loc + w_hl * hl; losses[3] = mask_sum;
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ready for the next call (stream order)
    }
}
void launch_loss(const float* sl, const float* el, const float* h, const int64_t* s_lab, const int64_t* e_lab,
                 const int64_t* h_lab, const float* vmask, int B, int T, float inv_batch, float mask_sum_override,
                 float w_loc, float w_hl, float* scratch, float* losses, float* d_sl, float* d_el, float* d_h,
                 hipStream_t s, unsigned* counter) {
    if (counter && d_sl && mask_sum_override > 0.f) {     // the caller supplies the global mask sum (data-parallel path): one launch
        VSL_LAUNCH(k_loss_fused, dim3(B), dim3(256), 0, s, sl, el, h, s_lab, e_lab, h_lab, vmask, B, T, inv_batch, mask_sum_override,
                   w_loc, w_hl, scratch, losses, d_sl, d_el, d_h, counter);
        return;
    }
    VSL_LAUNCH(k_loss_a, dim3(B), dim3(256), 0, s, sl, el, h, s_lab, e_lab, h_lab, vmask, B, T, scratch);
    if (d_sl)
        VSL_LAUNCH(k_loss_c, dim3((T + 255) / 256, B), dim3(256), 0, s, sl, el, h, s_lab, e_lab, h_lab, vmask, B, T,
                           inv_batch, mask_sum_override, w_loc, w_hl, scratch, losses, d_sl, d_el, d_h);
    else
        VSL_LAUNCH(k_loss_b, dim3(1), dim3(256), 0, s, B, inv_batch, mask_sum_override, w_loc, w_hl, scratch, losses);
}

// a17 extract_index (:355-363): argmax over the upper-triangular outer product of the two softmaxes, computed as
//   start = argmax_i ps[i] * max_{j>=i} pe[j],  end = argmax_j pe[j] * max_{i<=j} ps[i]   (no (B,T,T) tensor)
__global__ __launch_bounds__(256) void k_extract_index(const float* __restrict__ sl, const float* __restrict__ el,
                                                       int64_t* __restrict__ si, int64_t* __restrict__ ei, int T) {
    extern __shared__ float sm[];
    float* ps = sm;
    float* pe = sm + T;
    float* red = pe + T;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* s = sl + (size_t)b * T;
    const float* e = el + (size_t)b * T;
    float ms = -3.0e38f, me = -3.0e38f;
    for (int t = tid; t < T; t += 256) { ms = fmaxf(ms, s[t]); me = fmaxf(me, e[t]); }
    ms = block_reduce(ms, red, true);
    me = block_reduce(me, red, true);
    float ss = 0.f, se = 0.f;
    for (int t = tid; t < T; t += 256) {
        const float a = expf(s[t] - ms), c = expf(e[t] - me);
        ps[t] = a; pe[t] = c; ss += a; se += c;
    }
    ss = block_reduce(ss, red, false);
    se = block_reduce(se, red, false);
    for (int t = tid; t < T; t += 256) { ps[t] = ps[t] / ss; pe[t] = pe[t] / se; }
    __syncthreads();
    if (tid == 0) {          // start: scan from the right keeping the suffix max of pe
        float suf = 0.f, best = -1.f;
        int bi = 0;
        for (int i = T - 1; i >= 0; --i) {
            suf = fmaxf(suf, pe[i]);
            const float v = ps[i] * suf;
            if (v >= best) { best = v; bi = i; }       // >= while scanning right-to-left keeps the FIRST maximal index
        }
        si[b] = bi;
    } else if (tid == 64) {  // end: scan from the left keeping the prefix max of ps
        float pre = 0.f, best = -1.f;
        int bi = 0;
        for (int j = 0; j < T; ++j) {
            pre = fmaxf(pre, ps[j]);
            const float v = pe[j] * pre;
            if (v > best) { best = v; bi = j; }
        }
        ei[b] = bi;
    }
}
void launch_extract_index(const float* sl, const float* el, int64_t* si, int64_t* ei, int B, int T, hipStream_t s) {
    VSL_LAUNCH(k_extract_index, dim3(B), dim3(256), (size_t)(2 * T + 8) * sizeof(float), s, sl, el, si, ei, T);
}

// Weight gradients: kernels_wgrad.hip

// =========================================================================================================
// MHA block backward (a8, :167-190)
//  k_attn_out_bwd : do = dy * m5 ; dh2 = do Wo ; dr = dy + LN2^T(dh2 * m4)
//  k_attn_bwd_dq / k_attn_bwd_dkv : attention core (recompute P from Q, K and the saved LSE)
//  k_qkv_bwd      : dh1 = [dQ|dK|dV] [Wq;Wk;Wv] ; dx = dr + LN1^T(dh1 * m1)
// =========================================================================================================
__global__ __launch_bounds__(256) void k_attn_out_bwd(AttnOutBwdArgs a, int R) {
    __shared__ __attribute__((aligned(16))) float Gs[TILE_M * LDP];
    __shared__ __attribute__((aligned(16))) float Xs[TILE_M * LDP];
    attn_out_bwd_tile(a, nullptr, Gs, Xs, blockIdx.x * TILE_M, R);
}
void launch_attn_out_bwd(const float* dy, const float* dy2, const float* r, const float* ln_g, const float* WTpack, float* g_o,
                         float* dr, float* p_lng, float* p_lnb, int R, Drop d4, Drop d5, hipStream_t s) {
    {
        const size_t shm_sp = 0;
        const AttnOutBwdArgs a{dy, dy2, r, ln_g, WTpack, g_o, dr, p_lng, p_lnb, d4, d5};
        VSL_LAUNCH(k_attn_out_bwd, dim3((R + TILE_M - 1) / TILE_M), dim3(256), shm_sp, s, a, R);
    }
}

// =========================================================================================================
// heads backward (a14, :349-352): dlogit -> dz = dlogit * w1 * (hid > 0) -> [dLN(feat) | dx] = dz W0 ;
//   LN backward -> dfeat.  blockIdx.y selects start / end.
// =========================================================================================================
// `fuse` (transformer head): the END head's workgroups go on, on the tile of dfeat they have just produced, with the attention-output
// backward of the predictor encoder's second pass (k_attn_out_bwd's body) -- that gradient has no other consumer, so it never leaves the chip,
// and one kernel boundary of the dependent chain is gone (tile-local: no other workgroup's data is needed).
__global__ __launch_bounds__(256) void k_head_bwd(HeadBwdArgs a0, HeadBwdArgs a1, int R, int fuse, AttnOutBwdArgs ao) {
    __shared__ __attribute__((aligned(16))) float Gs[TILE_M * LDP];
    __shared__ __attribute__((aligned(16))) float Hs[TILE_M * LDP];
    __shared__ __attribute__((aligned(16))) float Fs[TILE_M * LDP];
    __shared__ float dl[TILE_M];
    __shared__ float red[8];
    const HeadBwdArgs a = blockIdx.y == 0 ? a0 : a1;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int r0 = blockIdx.x * TILE_M;
    BFrag<2, 8> bf;
    if (a.logits) {                        // (block-uniform) vsl_io.fused_loss: the tile's seeds from its samples' logits.  T >= 32: a tile touches at most two samples
        const int b0 = r0 / a.T, b1 = min(r0 + TILE_M - 1, R - 1) / a.T;
        const float lse0 = loss_sample_lse(a.logits + (size_t)b0 * a.T, a.T, red);
        float lse1 = lse0;
        if (b1 != b0) { __syncthreads(); lse1 = loss_sample_lse(a.logits + (size_t)b1 * a.T, a.T, red); }      // (block-uniform)
        if (tid < TILE_M) {
            const int r = r0 + tid, b = r / a.T;
            dl[tid] = r < R ? loss_ce_seed(a.logits[r], b == b0 ? lse0 : lse1, r - b * a.T, (int)a.label[b], a.cs) : 0.f;
        }
    } else
    if (tid < TILE_M) dl[tid] = r0 + tid < R ? a.dlogit[r0 + tid] : 0.f;
    load_tile128(Hs, a.hid, r0, TILE_M, R);
    load_tile128(Fs, a.feat, r0, TILE_M, R);
    bfrag_load(bf, a.W0Tpack, 2 * D, 32 * w, D, 0, D / 8);
    __syncthreads();
    for (int e = tid; e < TILE_M * 32; e += 256) {
        const int rr = e >> 5, c = (e & 31) * 4;
        const float4 hv = *reinterpret_cast<const float4*>(&Hs[rr * LDP + c]);
        const float4 wv = *reinterpret_cast<const float4*>(a.w1 + c);
        const float d = dl[rr];
        const float4 dz = make_float4(hv.x > 0.f ? d * wv.x : 0.f, hv.y > 0.f ? d * wv.y : 0.f, hv.z > 0.f ? d * wv.z : 0.f,
                                      hv.w > 0.f ? d * wv.w : 0.f);
        *reinterpret_cast<float4*>(&Gs[rr * LDP + c]) = dz;
        if (r0 + rr < R) *reinterpret_cast<float4*>(a.gz + (size_t)(r0 + rr) * D + c) = dz;
    }
    __syncthreads();
    {   // bias / w1 partials: threads 0..127 -> db0[c], 128..255 -> dw1[c]
        const int c = tid & 127;
        float acc = 0.f;
        if (tid < 128) { for (int rr = 0; rr < TILE_M; ++rr) acc += Gs[rr * LDP + c]; a.p_b0[(size_t)blockIdx.x * D + c] = acc; }
        else { for (int rr = 0; rr < TILE_M; ++rr) acc += dl[rr] * Hs[rr * LDP + c]; a.p_w1[(size_t)blockIdx.x * D + c] = acc; }
        if (tid == 0) { float sdl = 0.f; for (int rr = 0; rr < TILE_M; ++rr) sdl += dl[rr]; a.p_b1[blockIdx.x] = sdl; }
    }
    f32x16 acc[2];
    zero_acc(acc);
    gemm32p<2, 8>(Gs, LDP, D, a.W0Tpack, 2 * D, 32 * w, D, acc, bf);
    __syncthreads();                       // everyone is done reading Hs/Gs
    const int col = 32 * w + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, lane);
        Hs[row * LDP + col] = acc[0][r];                           // grad wrt LN(feat)
        if (r0 + row < R) a.dx[(size_t)(r0 + row) * D + col] = acc[1][r];
    }
    __syncthreads();
    if (a.ln_g) {
        LnResid nores;
#pragma unroll
        for (int j = 0; j < 4; ++j) nores.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool go_on = fuse && blockIdx.y == 1;
        ln_bwd_tile(Hs, Fs, nores, a.ln_g, go_on ? nullptr : a.dfeat, a.p_lng, a.p_lnb, r0, R, go_on ? Gs : nullptr);
        if (go_on) {                       // (block-uniform)
            __syncthreads();               // the column sums of ln_bwd_tile are done with Hs / Fs; Gs holds dfeat (rows >= R zero)
            attn_out_bwd_tile(ao, Gs, Hs, Fs, r0, R);
        }
    } else {
        for (int e = tid; e < TILE_M * D; e += 256) {
            const int rr = e >> 7, c = e & 127;
            if (r0 + rr < R) a.dfeat[(size_t)(r0 + rr) * D + c] = Hs[rr * LDP + c];
        }
    }
}
void launch_head_bwd(const HeadBwdArgs& a0, const HeadBwdArgs& a1, int R, hipStream_t s, const AttnOutBwdArgs* fuse) {
    {
        const size_t shm_sp = 0;
        AttnOutBwdArgs ao;
        memset(&ao, 0, sizeof ao);
        if (fuse) ao = *fuse;
        VSL_LAUNCH(k_head_bwd, dim3((R + TILE_M - 1) / TILE_M, 2), dim3(256), shm_sp, s, a0, a1, R, fuse ? 1 : 0, ao);
    }
}


// Fused attention backward for Lp <= 128: ONE workgroup (16 waves) per (sample, head) computes S, P, dP and dS once.
// Queries are processed in passes of 64 so that dS[64][keys] fits beside the head slices in < 80 KB of LDS: two workgroups
// share a CU (one stages while the other computes) and a weight-gradient workgroup of another stream still fits too.
//   phase 1: wave (ks = w & 7, qh = w >> 3) owns the 16 keys of strip ks and the query tiles qh, qh + 2 of the pass;
//            dK / dV of the strip accumulate in registers over all passes, dS goes to LDS as a [query][key] matrix;
//   phase 2: waves 0-3 each multiply the 16 rows of dS of one query tile with K  ->  dQ (4 independent MFMA chains).
// The two dK / dV partials (qh = 0, 1) are added in a fixed order through LDS, so the result does not depend on
// scheduling.  Against the two-kernel path this halves the exp / dropout-hash / S / dP work (fp32 MFMAs and the vector ALU
// share the SIMD, so that work is not hidden behind anything) and removes one kernel boundary.
// LMAX = 128: 8 key strips x 2 query halves (the two dK / dV partials are added through LDS); LMAX = 256: 16 key strips, a
// wave sees every query tile (no exchange), 152 KB of LDS, one workgroup per CU.
constexpr int AB_KST = 20, AB_QP = 64;
template <int LMAX> constexpr size_t ab_lds() { return (size_t)(4 * LMAX * AB_KST + AB_QP * (LMAX + 4) + 3 * LMAX) * sizeof(float); }
template <int LMAX>
__global__ __launch_bounds__(1024) void k_attn_bwd_fused(const float* __restrict__ Q, const float* __restrict__ K,
                                                         const float* __restrict__ V, const float* __restrict__ att,
                                                         const float* __restrict__ dr, const float* __restrict__ lse,
                                                         const float* __restrict__ mask, float* __restrict__ dQ,
                                                         float* __restrict__ dK, float* __restrict__ dV, int L, int H,
                                                         int b_off, Drop d2, Drop d3) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int AB_LMAX = LMAX, AB_DSP = LMAX + 4, NKS = LMAX / 16, NSPLIT = 16 / NKS;
    float* Ks = smem;                           // [LMAX][20]
    float* dSs = Ks + AB_LMAX * AB_KST;         // [64][132]   dS[query - qb][key] of the current pass
    float* Ls = dSs + AB_QP * AB_DSP;           // LSE per query
    float* Ds = Ls + AB_LMAX;                   // D = dA . O per query
    float* Mb = Ds + AB_LMAX;                   // additive key bias
    float* Qs = Mb + AB_LMAX;                   // [128][20]   Qs | Vs | As are dead after the last pass and become the
    float* Vs = Qs + AB_LMAX * AB_KST;          // [128][20]   exchange area of the dK / dV partials
    float* As = Vs + AB_LMAX * AB_KST;          // [128][20]   dA = dr * m3
    const int Lp = (L + 15) & ~15;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    int h, b, bzs;
    xcd_swizzle(h, b, bzs);                     // grid (H, B): all heads of a sample on one XCD (common.hpp)
    const size_t rowbase = (size_t)b * L;
    STAMP(0);
    for (int it = 0; it < LMAX / 128; ++it) {
        const int e = (tid & 511) + 512 * it, row = e >> 2, c4 = (e & 3) * 4;
        const size_t off = (rowbase + row) * D + h * HD + c4;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < 512) {
            float4 qv = z, kv = z;
            if (row < L) {
                qv = *reinterpret_cast<const float4*>(Q + off);
                kv = *reinterpret_cast<const float4*>(K + off);
            }
            *reinterpret_cast<float4*>(&Qs[row * AB_KST + c4]) = qv;
            *reinterpret_cast<float4*>(&Ks[row * AB_KST + c4]) = kv;
            if (it == 0 && tid < AB_LMAX) {
                Ls[tid] = tid < L ? lse[((size_t)b * H + h) * L + tid] : 0.f;
                Mb[tid] = tid < L ? (1.0f - mask[rowbase + tid]) * MASK_VALUE : MASK_VALUE;
            }
        } else {
            float4 vv = z, av = z, ov = z;
            if (row < L) {
                vv = *reinterpret_cast<const float4*>(V + off);
                av = *reinterpret_cast<const float4*>(dr + off);
                ov = *reinterpret_cast<const float4*>(att + off);
                if (d3.thresh) {                    // r = drop3(att) + x  (:183-184)
                    const uint32_t base = (uint32_t)off;
                    av.x *= drop_mul(d3, base); av.y *= drop_mul(d3, base + 1);
                    av.z *= drop_mul(d3, base + 2); av.w *= drop_mul(d3, base + 3);
                }
            }
            *reinterpret_cast<float4*>(&Vs[row * AB_KST + c4]) = vv;
            *reinterpret_cast<float4*>(&As[row * AB_KST + c4]) = av;
            float dsum = av.x * ov.x + av.y * ov.y + av.z * ov.z + av.w * ov.w;     // D_q = dA . O (= sum_k dP_k P_k)
            dsum += lane_xor1(dsum);
            dsum += lane_xor2(dsum);
            if ((e & 3) == 0) Ds[row] = dsum;
        }
    }
    STAMP(1);
    __syncthreads();
    STAMP(2);
    const int ki = lane & 15, g = lane >> 4;
    const int hi = w / NKS;                     // which of the NSPLIT dK / dV partials this wave produces
    const float scale = 0.25f;
    f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
    const int key = 16 * (w % NKS) + ki;
    const bool has_keys = 16 * (w % NKS) < Lp;
    float4 kf = make_float4(0.f, 0.f, 0.f, 0.f), vf = kf;
    float mb = MASK_VALUE;
    if (has_keys) {
        kf = *reinterpret_cast<const float4*>(&Ks[key * AB_KST + 4 * g]);
        vf = *reinterpret_cast<const float4*>(&Vs[key * AB_KST + 4 * g]);
        mb = Mb[key];
    }
    const uint32_t hb = (uint32_t)(((size_t)(b + b_off) * H + h) * L);
    for (int qb = 0; qb < Lp; qb += AB_QP) {
        const int qe = min(qb + AB_QP, Lp);
        if (has_keys) {
            for (int qt = qb + 16 * hi; qt < qe; qt += 16 * NSPLIT) {
                // S tile (rows = queries qt + 4g + reg, col = key) and dPd tile, same shape
                const float4 qa = *reinterpret_cast<const float4*>(&Qs[(qt + ki) * AB_KST + 4 * g]);
                const float4 aa = *reinterpret_cast<const float4*>(&As[(qt + ki) * AB_KST + 4 * g]);
                f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa.x, kf.x, sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(aa.x, vf.x, dp, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa.y, kf.y, sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(aa.y, vf.y, dp, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa.z, kf.z, sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(aa.z, vf.z, dp, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa.w, kf.w, sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(aa.w, vf.w, dp, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qq = qt + 4 * g + r;
                    const float p = __expf(sc[r] * scale + mb - Ls[qq]);
                    const float m2 = drop_mul(d2, (hb + (uint32_t)qq) * (uint32_t)L + (uint32_t)key);
                    const float pd = p * m2;
                    const float ds = p * (dp[r] * m2 - Ds[qq]) * scale;
                    dSs[(qq - qb) * AB_DSP + key] = ds;
                    // dV^T[dd][key] += dA[q][dd] * Pd[q][key] ; dK^T[dd][key] += Q[q][dd] * dS[q][key]
                    dv = __builtin_amdgcn_mfma_f32_16x16x4f32(As[qq * AB_KST + ki], pd, dv, 0, 0, 0);
                    dk = __builtin_amdgcn_mfma_f32_16x16x4f32(Qs[qq * AB_KST + ki], ds, dk, 0, 0, 0);
                }
            }
        }
        if (qb == 0) STAMP(3);
        __syncthreads();                        // dS of this pass complete
        if (qb == 0) STAMP(4);
        // phase 2: dQ^T[dd][q] += K[key][dd] * dS[q][key].  All 16 waves: wave = (query tile w & 3, key-tile residue w >> 2 mod 4); the four
        // partial tiles of a query tile meet in the (then dead) dS buffer and are added in residue order.  (One query tile per wave over
        // every key kept 12 of the 16 waves idle for a third of the pass.)
        const int qt2 = qb + 16 * (w & 3), kr = w >> 2;
        f32x4 dqs = {0.f, 0.f, 0.f, 0.f};
        if (qt2 < qe) {
            f32x4 dq[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) dq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* kp = Ks + g * AB_KST + ki;
            const float* sp = dSs + (16 * (w & 3) + ki) * AB_DSP + g;
            for (int k0 = 16 * kr; k0 < Lp; k0 += 64) {
                float ka[4], sb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { ka[i] = kp[(k0 + 4 * i) * AB_KST]; sb[i] = sp[k0 + 4 * i]; }
#pragma unroll
                for (int i = 0; i < 4; ++i) dq[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i], sb[i], dq[i], 0, 0, 0);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) dqs[c] = (dq[0][c] + dq[1][c]) + (dq[2][c] + dq[3][c]);
        }
        __syncthreads();                        // every wave has read its dS rows: the buffer becomes the exchange area
        float4* xq = reinterpret_cast<float4*>(dSs);          // [3 residues][4 query tiles][64 lanes]
        if (kr > 0) xq[((kr - 1) * 4 + (w & 3)) * 64 + lane] = make_float4(dqs[0], dqs[1], dqs[2], dqs[3]);
        __syncthreads();
        if (kr == 0 && qt2 < qe) {
            const float4 p1 = xq[(0 * 4 + w) * 64 + lane], p2 = xq[(1 * 4 + w) * 64 + lane], p3 = xq[(2 * 4 + w) * 64 + lane];
            const int q = qt2 + ki;
            if (q < L)
                *reinterpret_cast<float4*>(dQ + (rowbase + q) * D + h * HD + 4 * g) =
                    make_float4(((dqs[0] + p1.x) + p2.x) + p3.x, ((dqs[1] + p1.y) + p2.y) + p3.y,
                                ((dqs[2] + p1.z) + p2.z) + p3.z, ((dqs[3] + p1.w) + p2.w) + p3.w);
        }
        if (qb == 0) STAMP(5);
        __syncthreads();                        // dS buffer (and, after the last pass, Qs / Vs / As) free
        if (qb == 0) STAMP(6);
    }
    float* xkv = Qs;                            // [8 strips][64 lanes][8]
    if (NSPLIT == 2 && hi == 1) {
        *reinterpret_cast<float4*>(&xkv[((w & 7) * 64 + lane) * 8]) = make_float4(dk[0], dk[1], dk[2], dk[3]);
        *reinterpret_cast<float4*>(&xkv[((w & 7) * 64 + lane) * 8 + 4]) = make_float4(dv[0], dv[1], dv[2], dv[3]);
    }
    __syncthreads();
    if (hi == 0 && key < L) {
        float4 k1 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = k1;
        if (NSPLIT == 2) {
            k1 = *reinterpret_cast<const float4*>(&xkv[((w & 7) * 64 + lane) * 8]);
            v1 = *reinterpret_cast<const float4*>(&xkv[((w & 7) * 64 + lane) * 8 + 4]);
        }
        const size_t off = (rowbase + key) * D + h * HD + 4 * g;
        *reinterpret_cast<float4*>(dK + off) = make_float4(dk[0] + k1.x, dk[1] + k1.y, dk[2] + k1.z, dk[3] + k1.w);
        *reinterpret_cast<float4*>(dV + off) = make_float4(dv[0] + v1.x, dv[1] + v1.y, dv[2] + v1.z, dv[3] + v1.w);
    }
    STAMP(7);
}
// Single-pass attention backward for L > 256 (round 3; replaces k_attn_bwd_dq + k_attn_bwd_dkv, which computed S, P, dP and dS twice).
// Workgroup (16 waves) = (key block of 256 keys, head, sample).  The block's K / V head slices stay in LDS; the queries stream through in
// passes of 64 (Q, dA = dr * m3, LSE, D = dA . O staged per pass).  Per pass, as in k_attn_bwd_fused<256>:
//   phase 1: wave w owns the 16 keys of strip w and all four query tiles of the pass: S, P, dP, dS ONCE; dK / dV of the strip accumulate
//            in registers over every pass of the sequence; dS goes to LDS as a [query][key] matrix;
//   phase 2: dQ^T[dd][q] += K[key][dd] dS[q][key] over the block's keys (wave = query tile x key residue, partials added in residue order).
// dK / dV are complete; dQ is a partial over this key block and goes to slab `kb` of dQ (slab stride = the (R, 128) tensor): k_qkv_bwd adds
// the slabs while it stages its tile (and leaves the sum in slab 0 for the weight gradient).  122 KB of LDS, one workgroup per CU.
constexpr int ABL_KB = 256;
constexpr size_t abl_lds() { return (size_t)(2 * ABL_KB * AB_KST + ABL_KB + AB_QP * (ABL_KB + 4) + 2 * AB_QP * AB_KST + 2 * AB_QP) * sizeof(float) + (size_t)2 * 3 * AB_QP * 16 * sizeof(uint16_t); }
// PAIR: the probability site's masks are keyed by key PAIRS (the forward of L > 256, k_attn_fwd) ; false: one hash per element (the forward of
// L <= 256, k_attn_block_fwd) -- the kernel then serves 128 < L <= 256 as ONE key block (launch_attn_bwd)
template <bool PAIR>
__global__ __launch_bounds__(1024) void k_attn_bwd_long(const float* __restrict__ Q, const float* __restrict__ K,
                                                        const float* __restrict__ V, const float* __restrict__ att,
                                                        const float* __restrict__ dr, const float* __restrict__ lse,
                                                        const float* __restrict__ mask, float* __restrict__ dQ,
                                                        float* __restrict__ dK, float* __restrict__ dV, int L, int H,
                                                        int b_off, size_t slab, Drop d2, Drop d3) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int AB_DSP = ABL_KB + 4;
    float* Ks = smem;                           // [256][20]
    float* Vs = Ks + ABL_KB * AB_KST;           // [256][20]
    float* Mb = Vs + ABL_KB * AB_KST;           // additive key bias of the block
    float* dSs = Mb + ABL_KB;                   // [64][260] dS[query - qb][key - k0] of the current pass
    float* Qs = dSs + AB_QP * AB_DSP;           // [64][20] queries of the pass
    float* As = Qs + AB_QP * AB_KST;            // [64][20] dA = dr * m3
    float* Ls = As + AB_QP * AB_KST;            // [64] LSE
    float* Ds = Ls + AB_QP;                     // [64] D = dA . O
    // S = Q K^T and dP = dA V^T contract over the 16 head dims: on the bf16 matrix cores at fp32 grade like k_attn_fwd's K Q^T (three
    // v_mfma_f32_16x16x32_bf16 per product: the six split products, two per instruction) -- 96 pipe cycles that overlap with the vector work instead
    // of 256 of the fp32-input MFMA that do not.  Q and dA of the pass are split ONCE by the staging threads into three bf16 planes each.
    uint16_t* Qp = reinterpret_cast<uint16_t*>(Ds + AB_QP);     // [3 terms][64][16] (af_kp)
    uint16_t* Ap = Qp + 3 * AB_QP * 16;                         // [3 terms][64][16]
    constexpr int QPL = AB_QP * 16;
    const int Lp = (L + 15) & ~15;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    int kb, h, b;
    xcd_swizzle(kb, h, b);                      // grid (key blocks, H, B)
    const size_t rowbase = (size_t)b * L;
    const int k0 = kb * ABL_KB, nkeys = min(ABL_KB, Lp - k0);       // padded keys of this block (multiple of 16)
    {   // the block's K / V slices: 256 rows x 4 float4 each = 1024 items per tensor
        const int row = tid >> 2, c4 = (tid & 3) * 4;
        float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
        if (k0 + row < L) {
            const size_t off = (rowbase + k0 + row) * D + h * HD + c4;
            kv = *reinterpret_cast<const float4*>(K + off);
            vv = *reinterpret_cast<const float4*>(V + off);
        }
        *reinterpret_cast<float4*>(&Ks[row * AB_KST + c4]) = kv;
        *reinterpret_cast<float4*>(&Vs[row * AB_KST + c4]) = vv;
        if (tid < ABL_KB) Mb[tid] = k0 + tid < L ? (1.0f - mask[rowbase + k0 + tid]) * MASK_VALUE : MASK_VALUE;
    }
    const int ki = lane & 15, g = lane >> 4;
    const float scale = 0.25f;
    f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
    const int keyl = 16 * w + ki;               // key inside the block
    const bool has_keys = 16 * w < nkeys;
    const uint32_t hb = (uint32_t)(((size_t)(b + b_off) * H + h) * L);
    float4 kf = make_float4(0.f, 0.f, 0.f, 0.f), vf = kf;
    float mb = MASK_VALUE;
    // rows of the next pass: requested a pass ahead (threads 0-255: one float4 of Q, dA and att each)
    float4 nq, na, no;
    float nl = 0.f;
    auto fetch = [&](int qb) {
        const int row = qb + (tid >> 2), c4 = (tid & 3) * 4;
        nq = make_float4(0.f, 0.f, 0.f, 0.f); na = nq; no = nq; nl = 0.f;
        if (tid < 256 && row < L) {
            const size_t off = (rowbase + row) * D + h * HD + c4;
            nq = *reinterpret_cast<const float4*>(Q + off);
            na = *reinterpret_cast<const float4*>(dr + off);
            no = *reinterpret_cast<const float4*>(att + off);
            if ((tid & 3) == 0) nl = lse[((size_t)b * H + h) * L + row];
        }
    };
    auto stage = [&](int qb) {                  // (the previous pass' readers are behind a barrier)
        if (tid < 256) {
            const int rl = tid >> 2, c4 = (tid & 3) * 4;
            float4 av = na;
            if (d3.thresh && qb + rl < L) {     // r = drop3(att) + x  (:183-184)
                const uint32_t base = (uint32_t)((rowbase + qb + rl) * D + h * HD + c4);
                av.x *= drop_mul(d3, base); av.y *= drop_mul(d3, base + 1);
                av.z *= drop_mul(d3, base + 2); av.w *= drop_mul(d3, base + 3);
            }
            *reinterpret_cast<float4*>(&Qs[rl * AB_KST + c4]) = nq;
            *reinterpret_cast<float4*>(&As[rl * AB_KST + c4]) = av;
            {
                uint32_t h0, m0, l0, h1, m1, l1;
                const int po = af_kp(rl, c4 >> 3) + (c4 & 7);
                split3(nq.x, nq.y, h0, m0, l0); split3(nq.z, nq.w, h1, m1, l1);
                *reinterpret_cast<u32x2_t*>(Qp + po) = u32x2_t{h0, h1};
                *reinterpret_cast<u32x2_t*>(Qp + QPL + po) = u32x2_t{m0, m1};
                *reinterpret_cast<u32x2_t*>(Qp + 2 * QPL + po) = u32x2_t{l0, l1};
                split3(av.x, av.y, h0, m0, l0); split3(av.z, av.w, h1, m1, l1);
                *reinterpret_cast<u32x2_t*>(Ap + po) = u32x2_t{h0, h1};
                *reinterpret_cast<u32x2_t*>(Ap + QPL + po) = u32x2_t{m0, m1};
                *reinterpret_cast<u32x2_t*>(Ap + 2 * QPL + po) = u32x2_t{l0, l1};
            }
            float dsum = av.x * no.x + av.y * no.y + av.z * no.z + av.w * no.w;     // D_q = dA . O (= sum_k dP_k P_k)
            dsum += lane_xor1(dsum);
            dsum += lane_xor2(dsum);
            if ((tid & 3) == 0) { Ds[rl] = dsum; Ls[rl] = nl; }
        }
    };
    fetch(0);
    __syncthreads();
    // B operands of the wave's key tile (column = key ki; k group g < 2: dims 8 g .. of the first term of the pair, g >= 2: dims 8 (g - 2) .. of the
    // second):  [Q_h | Q_m] x [K_h | K_h] + [Q_h | Q_l] x [K_m | K_h] + [Q_h | Q_m] x [K_l | K_m], the same with (dA, V)
    u32x4_t bk1, bk2, bk3, bv1, bv2, bv3;
#pragma unroll
    for (int e = 0; e < 4; ++e) { bk1[e] = bk2[e] = bk3[e] = bv1[e] = bv2[e] = bv3[e] = 0u; }
    if (has_keys) {
        mb = Mb[keyl];
        auto split8 = [&](const float* row, u32x4_t& b1, u32x4_t& b2, u32x4_t& b3) {
            const float4 xa = *reinterpret_cast<const float4*>(row + 8 * (g & 1)), xb = *reinterpret_cast<const float4*>(row + 8 * (g & 1) + 4);
            uint32_t th[4], tm[4], tl[4];
            split3(xa.x, xa.y, th[0], tm[0], tl[0]); split3(xa.z, xa.w, th[1], tm[1], tl[1]);
            split3(xb.x, xb.y, th[2], tm[2], tl[2]); split3(xb.z, xb.w, th[3], tm[3], tl[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { b1[e] = th[e]; b2[e] = g < 2 ? tm[e] : th[e]; b3[e] = g < 2 ? tl[e] : tm[e]; }
        };
        split8(&Ks[keyl * AB_KST], bk1, bk2, bk3);
        split8(&Vs[keyl * AB_KST], bv1, bv2, bv3);
    }
    const int pa1 = (g < 2 ? 0 : 1) * QPL, pa2 = (g < 2 ? 0 : 2) * QPL;        // A operands: planes (h | m) and (h | l) by k group
    for (int qb = 0; qb < Lp; qb += AB_QP) {
        const int qe = min(qb + AB_QP, Lp);
        stage(qb);
        if (qb + AB_QP < Lp) fetch(qb + AB_QP);
        __syncthreads();
        if (has_keys) {
            for (int qt = qb; qt < qe; qt += 16) {
                const int ql0 = qt - qb;
                const int ao = af_kp(ql0 + ki, g & 1);
                const u32x4_t q1 = *reinterpret_cast<const u32x4_t*>(Qp + pa1 + ao), q2 = *reinterpret_cast<const u32x4_t*>(Qp + pa2 + ao);
                const u32x4_t a1 = *reinterpret_cast<const u32x4_t*>(Ap + pa1 + ao), a2 = *reinterpret_cast<const u32x4_t*>(Ap + pa2 + ao);
                f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                sc = mfma16_bf16(q1, bk3, sc);          // hl + mm (small terms first)
                dp = mfma16_bf16(a1, bv3, dp);
                sc = mfma16_bf16(q2, bk2, sc);          // hm + lh
                dp = mfma16_bf16(a2, bv2, dp);
                sc = mfma16_bf16(q1, bk1, sc);          // hh + mh
                dp = mfma16_bf16(a1, bv1, dp);
                // dropout of the probabilities as the forward drew it (k_attn_fwd): keys 2 j, 2 j + 1 of a query row share one hash.  The lane
                // and its neighbour (the other key of the pair, same four queries) hash two queries each and exchange the words.
                float m2v[4] = {1.f, 1.f, 1.f, 1.f};
                if (!PAIR) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) m2v[r] = drop_mul(d2, (hb + (uint32_t)(qb + ql0 + 4 * g + r)) * (uint32_t)L + (uint32_t)(k0 + keyl));
                } else
                if (d2.thresh != 0u) {
                    const uint32_t Lh = (uint32_t)((L + 1) >> 1), kp = (uint32_t)((k0 + keyl) >> 1);
                    const int odd = ki & 1;
                    uint32_t hm[2], hp[2];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        hm[jj] = drop_hash((hb + (uint32_t)(qb + ql0 + 4 * g + 2 * odd + jj)) * Lh + kp, d2.seed, d2.key);
                        hp[jj] = __float_as_uint(lane_xor1(__uint_as_float(hm[jj])));
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t hr = (r >> 1) == odd ? hm[r & 1] : hp[r & 1];
                        m2v[r] = (odd ? drop_hash_odd(hr) : hr) >= d2.thresh ? d2.scale : 0.f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ql = ql0 + 4 * g + r;
                    const float p = __expf(sc[r] * scale + mb - Ls[ql]);
                    const float m2 = m2v[r];
                    const float pd = p * m2;
                    const float ds = p * (dp[r] * m2 - Ds[ql]) * scale;
                    dSs[ql * AB_DSP + keyl] = ds;
                    dv = __builtin_amdgcn_mfma_f32_16x16x4f32(As[ql * AB_KST + ki], pd, dv, 0, 0, 0);
                    dk = __builtin_amdgcn_mfma_f32_16x16x4f32(Qs[ql * AB_KST + ki], ds, dk, 0, 0, 0);
                }
            }
        }
        __syncthreads();                        // dS of this pass complete
        const int qt2 = qb + 16 * (w & 3), kr = w >> 2;
        f32x4 dqs = {0.f, 0.f, 0.f, 0.f};
        if (qt2 < qe) {
            f32x4 dq[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) dq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* kp = Ks + g * AB_KST + ki;
            const float* sp = dSs + (16 * (w & 3) + ki) * AB_DSP + g;
            for (int kk = 16 * kr; kk < nkeys; kk += 64) {
                float ka[4], sb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { ka[i] = kp[(kk + 4 * i) * AB_KST]; sb[i] = sp[kk + 4 * i]; }
#pragma unroll
                for (int i = 0; i < 4; ++i) dq[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[i], sb[i], dq[i], 0, 0, 0);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) dqs[c] = (dq[0][c] + dq[1][c]) + (dq[2][c] + dq[3][c]);
        }
        __syncthreads();                        // every wave has read its dS rows: the buffer becomes the exchange area
        float4* xq = reinterpret_cast<float4*>(dSs);          // [3 residues][4 query tiles][64 lanes]
        if (kr > 0) xq[((kr - 1) * 4 + (w & 3)) * 64 + lane] = make_float4(dqs[0], dqs[1], dqs[2], dqs[3]);
        __syncthreads();
        if (kr == 0 && qt2 < qe) {
            const float4 p1 = xq[(0 * 4 + w) * 64 + lane], p2 = xq[(1 * 4 + w) * 64 + lane], p3 = xq[(2 * 4 + w) * 64 + lane];
            const int q = qt2 + ki;
            if (q < L)
                *reinterpret_cast<float4*>(dQ + (size_t)kb * slab + (rowbase + q) * D + h * HD + 4 * g) =
                    make_float4(((dqs[0] + p1.x) + p2.x) + p3.x, ((dqs[1] + p1.y) + p2.y) + p3.y,
                                ((dqs[2] + p1.z) + p2.z) + p3.z, ((dqs[3] + p1.w) + p2.w) + p3.w);
        }
        __syncthreads();                        // dS buffer, Qs / As / Ls / Ds free for the next pass
    }
    if (has_keys && k0 + keyl < L) {
        const size_t off = (rowbase + k0 + keyl) * D + h * HD + 4 * g;
        *reinterpret_cast<float4*>(dK + off) = make_float4(dk[0], dk[1], dk[2], dk[3]);
        *reinterpret_cast<float4*>(dV + off) = make_float4(dv[0], dv[1], dv[2], dv[3]);
    }
}
int attn_bwd_dq_slabs(int L) { const int Lp = (L + 15) & ~15; return Lp <= 256 ? 1 : (Lp + ABL_KB - 1) / ABL_KB; }
void launch_attn_bwd(const float* Q, const float* K, const float* V, const float* att, const float* dr, const float* lse,
                     const float* mask, float* dQ, float* dK, float* dV, int B, int L, int H, int b_off, Drop d2,
                     Drop d3, hipStream_t s) {
    const int Lp = (L + 15) & ~15;
    if (Lp <= 128) {
        static size_t ok128 = 0;
        ensure_dynamic_lds((const void*)k_attn_bwd_fused<128>, ab_lds<128>(), ok128, "k_attn_bwd_fused<128>");
        VSL_LAUNCH(k_attn_bwd_fused<128>, dim3(H, B), dim3(1024), ab_lds<128>(), s, Q, K, V, att, dr, lse, mask, dQ, dK, dV, L, H, b_off, d2, d3);
        static int left = 3;
        if (dbg_budget("attn_bwd") && L > 64) dbg_report("attn_bwd_fused: stage-issue | landed+sync | pass-0 phase1 | sync | phase2 | sync | pass 1 + final", 8, s, left);
        return;
    }
    if (Lp <= 256) {
        // 128 < L <= 256 as ONE 256-key block of the L > 256 kernel (S and dP on the bf16 pipe, Q / dA streamed in passes of 64) with the element
        // hashes of the L <= 256 forward.  (k_attn_bwd_fused<256>, 152 KB of LDS, served these lengths until round 6: +0.8 % on configs[2] / [3].)
        static size_t okb = 0;
        ensure_dynamic_lds((const void*)k_attn_bwd_long<false>, abl_lds(), okb, "k_attn_bwd_long<false>");
        VSL_LAUNCH(k_attn_bwd_long<false>, dim3(1, H, B), dim3(1024), abl_lds(), s, Q, K, V, att, dr, lse, mask, dQ, dK, dV,
                   L, H, b_off, (size_t)B * L * D, d2, d3);
        return;
    }
    // L > 256: one pass over S / dP per key block of 256; dQ arrives as attn_bwd_dq_slabs(L) partial slabs (k_qkv_bwd adds them)
    static size_t okl = 0;
    ensure_dynamic_lds((const void*)k_attn_bwd_long<true>, abl_lds(), okl, "k_attn_bwd_long");
    VSL_LAUNCH(k_attn_bwd_long<true>, dim3((Lp + ABL_KB - 1) / ABL_KB, H, B), dim3(1024), abl_lds(), s, Q, K, V, att, dr, lse, mask, dQ, dK, dV,
               L, H, b_off, (size_t)B * L * D, d2, d3);
}

// k_qkv_bwd lives in kernels_split.hip (its K = 384 product runs on the bf16 matrix cores: built without SLP vectorisation)

// =========================================================================================================
// a11 + a12 backward: gated = f2 * h ; h = sigmoid(mask_logits(f2 . wh + bh)) ; f2 = f1 W1^T + pb[b]
//   dgated = dg0 + dg1 + dg2 (two heads + predictor encoder input) ; dh_loss from the highlight loss.
// =========================================================================================================
__global__ __launch_bounds__(256) void k_cqcat_bwd(CqcatBwdArgs a, int R) {
    __shared__ __attribute__((aligned(16))) float Gs[TILE_M * LDP];     // dgated -> df2
    __shared__ __attribute__((aligned(16))) float Fs[TILE_M * LDP];     // f2
    __shared__ float dlg[TILE_M];
    cqcat_bwd_tile(a, nullptr, Gs, Fs, dlg, blockIdx.x * TILE_M, R);
}
void launch_cqcat_bwd(const float* dg0, const float* dg1, const float* dg2, const float* dh_loss, const float* f2,
                      const float* hscore, const float* wh, const float* W1Tpack, float* df2, float* df1, float* p_wh,
                      float* p_bh, int R, hipStream_t s, const HlSeed* hl) {
    {
        const size_t shm_sp = 0;
        CqcatBwdArgs a{dg0, dg1, dg2, dh_loss, f2, hscore, wh, W1Tpack, df2, df1, p_wh, p_bh, nullptr, nullptr, 0.f, 0.f};
        if (hl) { a.h_lab = hl->h_lab; a.vmask = hl->vmask; a.w_hl = hl->w_hl; a.mask_sum = hl->mask_sum; }
        VSL_LAUNCH(k_cqcat_bwd, dim3((R + TILE_M - 1) / TILE_M), dim3(256), shm_sp, s, a, R);
    }
}

// =========================================================================================================
// a10 / a11 backward (CQAttention :208-243, WeightedPool / CQConcatenate bias :246-274) as FOUR tile-parallel kernels.
// Grid = (32-clip tiles, samples) for A, B, C -- every reduction over the clips of a sample goes through small per-tile
// partials that the next kernel sums in its prologue -- and one workgroup per sample for the tiny finalise D.
//   A: dcat = df1 Wcqa ; split into dC(direct), dc2q, dq2c ; dS_row + row-softmax backward -> dSr ;
//      per-tile partials of dM = S_row^T dq2c and dQ(c2q) = S_row^T dc2q
//   B: dM = sum of partials ; dS_col = C dM^T (-> scratch) ; per-tile partial of the column-softmax dot
//   C: dS = dSr + S_col * (dS_col - dot) ; dC total ; per-tile partials of colsum(dS), dQ(trilinear), colsum(df2),
//      and the w4C / w4mlu parameter slabs
//   D: dQ total (+ WeightedPool / pooled-bias path), w4Q / pool / W2 / bias slabs
// =========================================================================================================
__global__ __launch_bounds__(256) void k_cq_bwd_a(CqBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = a.T, Lq = a.Lq, LQ1 = Lq + 1;
    const int NTJ = (Lq + 31) >> 5, PJ = 32 * NTJ + 1;      // 32-wide query-word tiles of the small MFMA products
    float* Dc = smem;                          // [32][CATP] grad wrt the concat tile
    float* Gs = Dc + TILE_M * CATP;            // [32][LDP]  df1 tile, later C tile
    float* Ss = Gs + TILE_M * LDP;             // [32][LQ1]  S_row, pad column zeroed (+ slack for the 32-wide over-read)
    float* Sd = Ss + TILE_M * LQ1 + 8;         // [32][LQ1]  dS_row
    float* Pp = Sd + TILE_M * LQ1 + 64;        // [4][32][PJ] per-wave partial sums of dS_row ; Lq > CQ_BIG_LQ: [32][PJ], one wave per word tile
    const bool big = Lq > CQ_BIG_LQ;            // NTJ = 4 then
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b = blockIdx.y, tl = blockIdx.x, t0 = tl * TILE_M, ntile = gridDim.x;
    const size_t crow = (size_t)b * T, qrow = (size_t)b * Lq;
    STAMP(0);
    load_tile128(Gs, a.df1 + crow * D, t0, TILE_M, T);
    for (int e = tid; e < TILE_M * LQ1; e += 256) {
        const int i = e / LQ1, j = e - i * LQ1;
        Ss[e] = (j < Lq && t0 + i < T) ? a.Srow[(crow + t0 + i) * Lq + j] : 0.f;
    }
    if (tid < 72) Sd[TILE_M * LQ1 - 8 + tid] = 0.f;          // finite values wherever the over-reads of Ss / Sd land
    BFrag<4, 4> bf;
    bfrag_load(bf, a.WcqaT, 4 * D, 32 * w, D, 0, D / 8);
    __syncthreads();
    STAMP(1);
    f32x16 acc[4];
    zero_acc(acc);
    gemm32p<4, 4>(Gs, LDP, D, a.WcqaT, 4 * D, 32 * w, D, acc, bf);
    const int col = 32 * w + (lane & 31), hh = lane >> 5;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) Dc[acc_row(r, lane) * CATP + t * D + col] = acc[t][r];
    STAMP(2);
    __syncthreads();
    load_tile128(Gs, a.C + crow * D, t0, TILE_M, T);      // Gs now holds the C tile
    // recompute c2q = S_row Q and q2c = S_row M (:229-230) on the matrix cores: K = Lq, wave = 32 output channels
    f32x16 c2q[1], q2c[1];
    zero_acc(c2q);
    zero_acc(q2c);
    {
        const float* sa = Ss + (lane & 31) * LQ1 + hh;
        for (int jc = 0; jc < Lq; jc += 16) {              // 8 k-steps per chunk: all loads first, then the MFMAs
            float sv[8], qv[8], mv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = jc + 2 * u + hh;
                const bool ok = j < Lq;
                sv[u] = ok ? sa[jc + 2 * u] : 0.f;
                qv[u] = ok ? a.Qf[(qrow + j) * D + col] : 0.f;
                mv[u] = ok ? a.M[(qrow + j) * D + col] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (jc + 2 * u < Lq) {
                    c2q[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[u], qv[u], c2q[0], 0, 0, 0);
                    q2c[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[u], mv[u], q2c[0], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();
    STAMP(3);
    // product-rule split of :231 in the accumulator layout (lane = channel, 16 rows)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = acc_row(r, lane);
        float* d = Dc + rr * CATP;
        const float cv = Gs[rr * LDP + col];
        const float d0 = d[col], d1 = d[D + col], d2 = d[2 * D + col], d3 = d[3 * D + col];
        d[D + col] = d1 + d2 * cv;             // dc2q, kept in LDS: dS dots and the dM / dQ partials below
        d[3 * D + col] = d3 * cv;              // dq2c
        if (t0 + rr < T) a.dC[(crow + t0 + rr) * D + col] = d0 + d2 * c2q[0][r] + d3 * q2c[0][r];
    }
    __syncthreads();
    STAMP(4);
    {   // dS_row[i][j] = dc2q[i] . Q[j] + dq2c[i] . M[j]: a 32 x 32 MFMA tile per 32 query words, K = 256 split over
        // the four waves (wave w: channels 32w .. 32w+31 of both halves), partial tiles summed in wave order through LDS
        const int i = lane & 31;
        if (big) {                                              // wave w = word tile w over the whole K = 256: no partial tiles
            f32x16 dsw[1];
            zero_acc(dsw);
            const int j = 32 * w + i;
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const float* src = part ? a.M : a.Qf;
                const float* arow = Dc + i * CATP + (part ? 3 * D : D) + 4 * hh;
                for (int kb = 0; kb < D / 8; kb += 4) {
                    float4 bq4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        bq4[q] = j < Lq ? *reinterpret_cast<const float4*>(src + (qrow + j) * D + (kb + q) * 8 + 4 * hh) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 av = *reinterpret_cast<const float4*>(arow + (kb + q) * 8);
                        dsw[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bq4[q].x, dsw[0], 0, 0, 0);
                        dsw[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bq4[q].y, dsw[0], 0, 0, 0);
                        dsw[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bq4[q].z, dsw[0], 0, 0, 0);
                        dsw[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bq4[q].w, dsw[0], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) Pp[acc_row(r, lane) * PJ + 32 * w + i] = dsw[0][r];
        } else
        for (int nb = 0; nb < NTJ; nb += 2) {                   // two 32-word tiles per pass (one pass up to 64 query words)
        f32x16 ds[2];
        zero_acc(ds);
        float4 bq[2][2][4];                                  // [part][nt][kq]: every B fragment is requested up front
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const float* src = part ? a.M : a.Qf;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) {
                    const int j = 32 * (nb + nt) + i;
                    bq[part][nt][kq] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (nb + nt < NTJ && j < Lq)
                        bq[part][nt][kq] = *reinterpret_cast<const float4*>(src + (qrow + j) * D + (4 * w + kq) * 8 + 4 * hh);
                }
        }
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const float* arow = Dc + i * CATP + (part ? 3 * D : D) + 4 * hh;
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                const float4 av = *reinterpret_cast<const float4*>(arow + (4 * w + kq) * 8);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if (nb + nt < NTJ) {
                        const float4 bv = bq[part][nt][kq];
                        ds[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, ds[nt], 0, 0, 0);
                        ds[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, ds[nt], 0, 0, 0);
                        ds[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, ds[nt], 0, 0, 0);
                        ds[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, ds[nt], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            if (nb + nt < NTJ) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Pp[(w * TILE_M + acc_row(r, lane)) * PJ + 32 * (nb + nt) + i] = ds[nt][r];
            }
        }
    }
    // per-tile partials on the matrix cores: dM_t[j][c] = sum_i Srow[i][j] dq2c[i][c] ; dQa_t[j][c] = sum_i Srow[i][j] dc2q[i][c]
    {
        float* p1 = a.P1 + ((size_t)(b * ntile + tl) * 2) * Lq * D;
        for (int nt = 0; nt < NTJ; ++nt) {
            f32x16 am[1], aq[1];
            zero_acc(am);
            zero_acc(aq);
            gemm_tn_p<1, TILE_M>(Ss, LQ1, 32 * nt, Dc + 3 * D, CATP, 32 * w, am);
            gemm_tn_p<1, TILE_M>(Ss, LQ1, 32 * nt, Dc + D, CATP, 32 * w, aq);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * nt + acc_row(r, lane);
                if (j < Lq) { p1[(size_t)j * D + col] = am[0][r]; p1[(size_t)(Lq + j) * D + col] = aq[0][r]; }
            }
        }
    }
    STAMP(5);
    __syncthreads();
    STAMP(6);
    if (tid < TILE_M) {                          // softmax backward over the query words (dim=2, :225), thread = clip
        const int t = t0 + tid;
        if (t < T) {
            const float* sr = Ss + tid * LQ1;
            float dot = 0.f;
            for (int j = 0; j < Lq; ++j) {
                const float g = big ? Pp[tid * PJ + j]
                                    : Pp[tid * PJ + j] + Pp[(TILE_M + tid) * PJ + j] + Pp[(2 * TILE_M + tid) * PJ + j] +
                                      Pp[(3 * TILE_M + tid) * PJ + j];
                Sd[tid * LQ1 + j] = g;
                dot += sr[j] * g;
            }
            for (int j = 0; j < Lq; ++j) a.dSr[(crow + t) * Lq + j] = sr[j] * (Sd[tid * LQ1 + j] - dot);
        }
    }
    STAMP(7);
}

// sums the per-tile partials of dM into LDS (dMs [Lq][LDP]); 8 loads in flight per thread
__device__ __forceinline__ void cq_sum_dM(float* dMs, const float* __restrict__ P1, int b, int ntile, int Lq) {
    for (int e = threadIdx.x; e < Lq * (D / 4); e += 256) {
        const int j = e >> 5, c4 = (e & 31) * 4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = 0; t < ntile; t += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                v[u] = t + u < ntile ? *reinterpret_cast<const float4*>(P1 + ((size_t)(b * ntile + t + u) * 2) * Lq * D + (size_t)j * D + c4)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        *reinterpret_cast<float4*>(dMs + j * LDP + c4) = s;
    }
}

__global__ __launch_bounds__(256) void k_cq_bwd_b(CqBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = a.T, Lq = a.Lq, LQ1 = Lq + 1;
    const int NTJ = (Lq + 31) >> 5, PJ = 32 * NTJ + 1;
    float* dMs = smem;                         // [32 NTJ][LDP]  dM (rows >= Lq zero)
    float* Cs = dMs + 32 * NTJ * LDP;          // [32][LDP]
    float* St = Cs + TILE_M * LDP;             // [32][LQ1] S_col tile
    float* Pp = St + TILE_M * LQ1;             // [4][32][PJ] per-wave partial tiles of dS_col ; Lq > CQ_BIG_LQ: [32][PJ], one wave per word tile
    const bool big = Lq > CQ_BIG_LQ;            // NTJ = 4 then
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, hh = lane >> 5;
    const int b = blockIdx.y, tl = blockIdx.x, t0 = tl * TILE_M, ntile = gridDim.x;
    const size_t crow = (size_t)b * T;
    load_tile128(Cs, a.C + crow * D, t0, TILE_M, T);
    for (int e = tid; e < TILE_M * Lq; e += 256) {
        const int i = e / Lq, j = e - i * Lq;
        St[i * LQ1 + j] = t0 + i < T ? a.Scol[(crow + t0) * Lq + e] : 0.f;
    }
    cq_sum_dM(dMs, a.P1, b, ntile, Lq);
    for (int e = tid; e < (32 * NTJ - Lq) * (D / 4); e += 256)         // zero rows Lq .. 32 NTJ - 1 (B operand of the padded tile)
        *reinterpret_cast<float4*>(dMs + (Lq + (e >> 5)) * LDP + (e & 31) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (big) {                              // wave w = word tile w over the whole K = 128
        f32x16 accw[1];
        zero_acc(accw);
        const int i = lane & 31;
        const float* arow = Cs + i * LDP + 4 * hh;
        const float* brow = dMs + (32 * w + i) * LDP + 4 * hh;
#pragma unroll
        for (int kb = 0; kb < D / 8; ++kb) {
            const float4 av = *reinterpret_cast<const float4*>(arow + kb * 8);
            const float4 bv = *reinterpret_cast<const float4*>(brow + kb * 8);
            accw[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, accw[0], 0, 0, 0);
            accw[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, accw[0], 0, 0, 0);
            accw[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, accw[0], 0, 0, 0);
            accw[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, accw[0], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Pp[acc_row(r, lane) * PJ + 32 * w + i] = accw[0][r];
    } else
    for (int nb = 0; nb < NTJ; nb += 2) {   // dS_col[i][j] = C[i] . dM[j]: 32 x 32 MFMA tile per 32 query words, K = 128 split over the waves
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
                    const float4 bv = *reinterpret_cast<const float4*>(dMs + (32 * (nb + nt) + i) * LDP + kb * 8 + 4 * hh);
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
    // sum the four partial tiles in wave order, write dS_col, and the per-tile partial of the column-softmax dot
    for (int e = tid; e < TILE_M * Lq; e += 256) {
        const int i = e / Lq, j = e - i * Lq;
        const float sv = big ? Pp[i * PJ + j]
                             : Pp[i * PJ + j] + Pp[(TILE_M + i) * PJ + j] + Pp[(2 * TILE_M + i) * PJ + j] + Pp[(3 * TILE_M + i) * PJ + j];
        Pp[i * PJ + j] = sv;                                                    // only this thread touches (i, j)
        if (t0 + i < T) a.dSs[(crow + t0) * Lq + e] = sv;
    }
    __syncthreads();
    if (tid < Lq) {
        float sacc = 0.f;
        for (int i = 0; i < TILE_M; ++i) sacc += Pp[i * PJ + tid] * St[i * LQ1 + tid];
        a.P2[(size_t)(b * ntile + tl) * Lq + tid] = sacc;
    }
}
__global__ __launch_bounds__(256) void k_cq_bwd_c(CqBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = a.T, Lq = a.Lq, LQ1 = Lq + 1;
    float* dMs = smem;                         // [Lq][LDP]  dM
    const bool big = Lq > CQ_BIG_LQ;            // then the dropped-out Q is not staged: its one use reads memory and re-applies the mask
    float* Qds = dMs + Lq * LDP;               // [Lq][LDP]  dropped-out Q (as used by the trilinear score)
    float* Cs = Qds + (big ? 0 : Lq * LDP);    // [32][LDP]  C tile, later df2 tile
    float* Cd = Cs + TILE_M * LDP;             // [32][LDP]  dropped C tile
    float* St = Cd + TILE_M * LDP;             // [32][LQ1]  S_col tile
    float* Sg = St + TILE_M * LQ1 + 8;         // [32][LQ1]  dS tile (+ slack for the chunked over-read)
    float* csum = Sg + TILE_M * LQ1 + 8;       // [Lq]
    float* rsum = csum + Lq;                   // [32]
    float* v128 = rsum + TILE_M;               // [4][128]
    const int tid = threadIdx.x;
    const int b = blockIdx.y, tl = blockIdx.x, t0 = tl * TILE_M, ntile = gridDim.x;
    const size_t crow = (size_t)b * T, qrow = (size_t)b * Lq;
    // ---- prologue: tile loads + the cross-tile sums
    load_tile128(Cs, a.C + crow * D, t0, TILE_M, T);
    cq_sum_dM(dMs, a.P1, b, ntile, Lq);
    if (!big)
    for (int e = tid; e < Lq * D; e += 256) {
        const int j = e >> 7, cc = e & 127;
        Qds[j * LDP + cc] = a.Qf[(qrow + j) * D + cc] * drop_mul(a.dq, (uint32_t)(((b + a.b_off) * Lq + j) * D + cc));
    }
    if (tid < Lq) {
        float s = 0.f;
        for (int t = 0; t < ntile; ++t) s += a.P2[(size_t)(b * ntile + t) * Lq + tid];
        csum[tid] = s;
    }
    __syncthreads();
    for (int e = tid; e < TILE_M * LQ1; e += 256) {
        const int i = e / LQ1, j = e - i * LQ1;
        float g = 0.f, st = 0.f;
        if (j < Lq && t0 + i < T) {
            const size_t o = (crow + t0 + i) * Lq + j;
            st = a.Scol[o];
            g = a.dSr[o] + st * (a.dSs[o] - csum[j]);
        }
        St[e] = st;
        Sg[e] = g;
    }
    for (int e = tid; e < TILE_M * D; e += 256) {
        const int rr = e >> 7, cc = e & 127;
        const int t = t0 + rr;
        Cd[rr * LDP + cc] = t < T ? Cs[rr * LDP + cc] * drop_mul(a.dc, (uint32_t)(((b + a.b_off) * T + t) * D + cc)) : 0.f;
    }
    __syncthreads();
    if (tid < TILE_M) { float s = 0.f; for (int j = 0; j < Lq; ++j) s += Sg[tid * LQ1 + j]; rsum[tid] = s; }
    if (tid >= 64 && tid < 64 + Lq) {          // per-tile partial of colsum(dS)
        const int j = tid - 64;
        float s = 0.f;
        for (int i = 0; i < TILE_M; ++i) s += Sg[i * LQ1 + j];
        a.P3[(size_t)(b * ntile + tl) * Lq + j] = s;
    }
    __syncthreads();
    float acc_w4C = 0.f, acc_mlu = 0.f;
    const int w = tid >> 6, lane = tid & 63, hh = lane >> 5, col = 32 * w + (lane & 31);
    {
        // tq = dS Qd and tm = S_col dM (K = Lq) on the matrix cores; wave = 32 channels, accumulator layout lane = channel
        f32x16 tq[1], tm[1];
        zero_acc(tq);
        zero_acc(tm);
        const float* sgp = Sg + (lane & 31) * LQ1 + hh;
        const float* stp = St + (lane & 31) * LQ1 + hh;
        for (int jc = 0; jc < Lq; jc += 16) {
            float gv[8], sv[8], qv[8], mv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = jc + 2 * u + hh;
                const bool ok = j < Lq;
                gv[u] = ok ? sgp[jc + 2 * u] : 0.f;
                sv[u] = ok ? stp[jc + 2 * u] : 0.f;
                qv[u] = !ok ? 0.f : !big ? Qds[j * LDP + col]
                                          : a.Qf[(qrow + j) * D + col] * drop_mul(a.dq, (uint32_t)(((b + a.b_off) * Lq + j) * D + col));
                mv[u] = ok ? dMs[j * LDP + col] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (jc + 2 * u < Lq) {
                    tq[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv[u], qv[u], tq[0], 0, 0, 0);   // sum_j dS[i][j] Qd[j][c]
                    tm[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[u], mv[u], tm[0], 0, 0, 0);   // sum_j Scol[i][j] dM[j][c]
                }
            }
        }
        const float wC = a.w4C[col], wM = a.w4mlu[col];
        float dcv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = t0 + acc_row(r, lane);
            dcv[r] = t < T ? a.dC[(crow + t) * D + col] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = acc_row(r, lane);
            const int t = t0 + rr;
            const float cdv = Cd[rr * LDP + col];
            const float dcd = rsum[rr] * wC + wM * tq[0][r];      // grad wrt dropped-out C
            acc_w4C += rsum[rr] * cdv;
            acc_mlu += tq[0][r] * cdv;
            if (t < T) a.dC[(crow + t) * D + col] = dcv[r] + tm[0][r] + dcd * drop_mul(a.dc, (uint32_t)(((b + a.b_off) * T + t) * D + col));
        }
        // per-tile partial of dQ(trilinear)[j][c] = mask_q * w4mlu[c] * sum_i dS[i][j] Cd[i][c]
        float* p4 = a.P4 + (size_t)(b * ntile + tl) * Lq * D;
        const int NTJ = (Lq + 31) >> 5;
        for (int nt = 0; nt < NTJ; ++nt) {
            f32x16 pq[1];
            zero_acc(pq);
            gemm_tn_p<1, TILE_M>(Sg, LQ1, 32 * nt, Cd, LDP, 32 * w, pq);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * nt + acc_row(r, lane);
                if (j < Lq) p4[(size_t)j * D + col] = pq[0][r] * wM * drop_mul(a.dq, (uint32_t)(((b + a.b_off) * Lq + j) * D + col));
            }
        }
    }
    v128[hh * D + col] = acc_w4C;
    v128[2 * D + hh * D + col] = acc_mlu;
    __syncthreads();                           // (also: everyone is done with Cs)
    load_tile128(Cs, a.df2 + crow * D, t0, TILE_M, T);
    if (tid < D) {
        a.p_w4C[(size_t)(b * ntile + tl) * D + tid] = v128[tid] + v128[D + tid];
        a.p_w4mlu[(size_t)(b * ntile + tl) * D + tid] = v128[2 * D + tid] + v128[3 * D + tid];
    }
    __syncthreads();
    if (tid < D) {                             // per-tile partial of dpb[o] = sum_t df2[b, t, o]
        float s = 0.f;
#pragma unroll 8
        for (int i = 0; i < TILE_M; ++i) s += Cs[i * LDP + tid];
        a.P5[(size_t)(b * ntile + tl) * D + tid] = s;
    }
}

__global__ __launch_bounds__(256) void k_cq_bwd_d(CqBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Lq = a.Lq, ntile = a.ntile;
    float* dQs = smem;                         // [Lq][LDP] dQ accumulator
    float* cs2 = dQs + Lq * LDP;               // [Lq] colsum(dS)
    float* sv = cs2 + Lq;                      // [Lq]
    float* v128 = sv + Lq;                     // [4][128]
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b = blockIdx.x;
    const size_t qrow = (size_t)b * Lq;
    // Every cross-tile / cross-word sum below issues its loads in batches before the (ordered) additions: the kernel is one workgroup per
    // sample with nothing else to hide a memory round trip per loop iteration (82 % of its wave cycles were parked; 26 -> 18 us).  Round 2
    // measured the same change as a net loss because the query chain then took CUs from the (then critical) video chain; since round 3 the
    // query chain IS the critical tail of the backward.
    if (tid < Lq) {
        float s = 0.f;
        for (int t0 = 0; t0 < ntile; t0 += 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = t0 + u < ntile ? a.P3[(size_t)(b * ntile + t0 + u) * Lq + tid] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) s += v[u];
        }
        cs2[tid] = s;
    }
    if (tid < D) {                             // dpb[o] = sum over tiles of colsum(df2)
        float s = 0.f;
        for (int t0 = 0; t0 < ntile; t0 += 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = t0 + u < ntile ? a.P5[(size_t)(b * ntile + t0 + u) * D + tid] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) s += v[u];
        }
        v128[tid] = s;
        a.p_bcat[(size_t)b * D + tid] = s;
    }
    __syncthreads();
    // dQ = sum_tiles (dQ(c2q) + dQ(trilinear)) + colsum(dS)[j] * w4Q * mask_q
    for (int e = tid; e < Lq * (D / 4); e += 256) {
        const int j = e >> 5, c4 = (e & 31) * 4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t0 = 0; t0 < ntile; t0 += 4) {
            float4 x[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = min(t0 + u, ntile - 1);
                x[u] = *reinterpret_cast<const float4*>(a.P1 + ((size_t)(b * ntile + t) * 2 + 1) * Lq * D + (size_t)j * D + c4);
                y[u] = *reinterpret_cast<const float4*>(a.P4 + (size_t)(b * ntile + t) * Lq * D + (size_t)j * D + c4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (t0 + u < ntile) { s.x += x[u].x + y[u].x; s.y += x[u].y + y[u].y; s.z += x[u].z + y[u].z; s.w += x[u].w + y[u].w; }
        }
        const float4 wq = *reinterpret_cast<const float4*>(a.w4Q + c4);
        const uint32_t base = (uint32_t)(((b + a.b_off) * Lq + j) * D + c4);
        s.x += cs2[j] * wq.x * drop_mul(a.dq, base); s.y += cs2[j] * wq.y * drop_mul(a.dq, base + 1);
        s.z += cs2[j] * wq.z * drop_mul(a.dq, base + 2); s.w += cs2[j] * wq.w * drop_mul(a.dq, base + 3);
        *reinterpret_cast<float4*>(dQs + j * LDP + c4) = s;
    }
    if (tid < D) {                             // dw4Q[c] = sum_j colsum(dS)[j] * Qd[j][c]
        float s = 0.f;
        for (int j0 = 0; j0 < Lq; j0 += 8) {
            float qv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) qv[u] = a.Qf[(qrow + min(j0 + u, Lq - 1)) * D + tid];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j0 + u < Lq) s += cs2[j0 + u] * qv[u] * drop_mul(a.dq, (uint32_t)(((b + a.b_off) * Lq + j0 + u) * D + tid));
        }
        a.p_w4Q[(size_t)b * D + tid] = s;
    }
    // ---- pooled-query path: pb = W2 pooled + bcat ; pooled = sum_j alpha_j Q[j] ; alpha = softmax(Q w + mask)
    for (int e = tid; e < D * D; e += 256)                // dW2[o][c] = dpb[o] * pooled[c]
        a.p_W2[(size_t)b * D * D + e] = v128[e >> 7] * a.pooled[(size_t)b * D + (e & 127)];
    {
        const int cc = tid & 127, hh = tid >> 7;
        float s8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) s8[q] = 0.f;
        for (int o = hh * 64; o < hh * 64 + 64; o += 8) { // dpooled[c] = sum_o Wcat[o][128 + c] * dpb[o]
#pragma unroll
            for (int q = 0; q < 8; ++q) s8[q] += a.Wcat[(size_t)(o + q) * 2 * D + D + cc] * v128[o + q];
        }
        v128[2 * D + hh * D + cc] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    }
    __syncthreads();
    if (tid < D) v128[D + tid] = v128[2 * D + tid] + v128[3 * D + tid];          // dpooled[c]
    __syncthreads();
    for (int j = w; j < Lq; j += 4) {                     // dalpha_j = dpooled . Q[j]
        const float* qr = a.Qf + (qrow + j) * D;
        const float d = wave_sum(qr[lane] * v128[D + lane] + qr[lane + 64] * v128[D + lane + 64]);
        if (lane == 0) sv[j] = d;
    }
    __syncthreads();
    if (w == 0) {
        const float a0 = lane < Lq ? a.alpha[qrow + lane] : 0.f, a1 = lane + 64 < Lq ? a.alpha[qrow + lane + 64] : 0.f;
        const float g0 = lane < Lq ? sv[lane] : 0.f, g1 = lane + 64 < Lq ? sv[lane + 64] : 0.f;
        const float dot = wave_sum(a0 * g0 + a1 * g1);
        if (lane < Lq) sv[lane] = a0 * (g0 - dot);        // dlogit_j
        if (lane + 64 < Lq) sv[lane + 64] = a1 * (g1 - dot);
    }
    __syncthreads();
    if (tid < D) {
        float s = 0.f;
        for (int j0 = 0; j0 < Lq; j0 += 8) {
            float qv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) qv[u] = a.Qf[(qrow + min(j0 + u, Lq - 1)) * D + tid];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j0 + u < Lq) s += sv[j0 + u] * qv[u];
        }
        a.p_pool[(size_t)b * D + tid] = s;
    }
    for (int e = tid; e < Lq * D; e += 256) {
        const int j = e >> 7, cc = e & 127;
        a.dQ[(qrow + j) * D + cc] = dQs[j * LDP + cc] + a.alpha[qrow + j] * v128[D + cc] + sv[j] * a.pool_w[cc];
    }
}

void launch_cq_bwd(const CqBwdArgs& a0, int B, hipStream_t s) {
    CqBwdArgs a = a0;
    const int Lq = a.Lq, ntile = (a.T + TILE_M - 1) / TILE_M;
    a.ntile = ntile;
    const int npart = Lq > CQ_BIG_LQ ? 1 : 4;         // partial tiles of the small MFMA products (see the kernels)
    const size_t shmA = (size_t)(TILE_M * CATP + TILE_M * LDP + 2 * TILE_M * (Lq + 1) + 72 + npart * TILE_M * (32 * ((Lq + 31) / 32) + 1)) * sizeof(float);
    const size_t shmB = (size_t)(32 * ((Lq + 31) / 32) * LDP + TILE_M * LDP + TILE_M * (Lq + 1) + npart * TILE_M * (32 * ((Lq + 31) / 32) + 1)) * sizeof(float);
    const size_t shmC = (size_t)((Lq > CQ_BIG_LQ ? 1 : 2) * Lq * LDP + 2 * TILE_M * LDP + 2 * TILE_M * (Lq + 1) + 16 + Lq + TILE_M + 4 * D) * sizeof(float);
    static size_t okA = 0, okB = 0, okC = 0;
    ensure_dynamic_lds((const void*)k_cq_bwd_a, shmA, okA, "k_cq_bwd_a");
    ensure_dynamic_lds((const void*)k_cq_bwd_b, shmB, okB, "k_cq_bwd_b");
    ensure_dynamic_lds((const void*)k_cq_bwd_c, shmC, okC, "k_cq_bwd_c");
    VSL_LAUNCH(k_cq_bwd_a, dim3(ntile, B), dim3(256), shmA, s, a);
    { static int left = 2; if (dbg_budget("cq_bwd_a")) dbg_report("cq_bwd_a: load | gemm512 | Dc store | C load + c2q/q2c | product rule | dS_row + partials | sync | softmax bwd", 8, s, left); }
    VSL_LAUNCH(k_cq_bwd_b, dim3(ntile, B), dim3(256), shmB, s, a);
    VSL_LAUNCH(k_cq_bwd_c, dim3(ntile, B), dim3(256), shmC, s, a);
}
// the per-sample tail (dQ, pooled-query path): nothing on the video side reads its outputs, so it runs on the query stream
void launch_cq_bwd_query(const CqBwdArgs& a0, int B, hipStream_t s) {
    CqBwdArgs a = a0;
    const int Lq = a.Lq;
    a.ntile = (a.T + TILE_M - 1) / TILE_M;
    const size_t shmD = (size_t)(Lq * LDP + 2 * Lq + 4 * D) * sizeof(float);
    static size_t okD = 0;
    ensure_dynamic_lds((const void*)k_cq_bwd_d, shmD, okD, "k_cq_bwd_d");
    VSL_LAUNCH(k_cq_bwd_d, dim3(B), dim3(256), shmD, s, a);
}

// =========================================================================================================
// a3 / a4 backward: unk_vec gradient, char-CNN weights / biases, char table (padding_idx = 0 gets none).
//   One workgroup of 8 waves per `chunk` query words (embed_bwd_chunk: as many as keep the grid inside one round of the 256 CUs);
//   the rows (word, position) of the chunk are packed densely.  Nothing here is sized by char_dim beyond an LDS row stride of 64
//   or 128 floats (char_dim <= 128; main_t7.py:24 prescribes 100 for ActivityNet).  After two shared staging phases the waves split
//   into two roles that never meet again:
//   waves 4-7 (i)   dW[oc][ci][kk] += g[oc] * Ce[pos[oc] + kk][ci]   thread = 25 fixed (oc, ci) pairs, 4 tap accumulators each: one
//         {pos, g} lookup per pair and word (one 8-byte LDS broadcast), shared by the taps.  Exact fp32 FMAs.
//   waves 0-3 (ii)  dCe[row][ci] = sum_k G[row][k] W[k][ci] over the k = (tap, channel) pairs of the four convs, where G has ONE
//         non-zero per (word, k): the arg-max position of the channel shifted by the tap.  The product runs on the matrix cores (wave
//         = 16 input channels, 76 MFMAs of 16x16x4 per 16-row tile) and G is never materialised: a lane builds its A operand from the
//         {pos, g} pair of (word, channel) -- g * (pos + tap == position) -- so there is no scatter, no clear and no barrier between
//         tiles.  k runs tap-major over aligned channel ranges (tap 0: channels 0..99, tap 1: 8..99, tap 2: 28..99, tap 3: 60..99; the
//         channels below a conv's first one carry zero weights), which makes (tap, channel) of an unrolled step a compile-time
//         constant plus lane >> 4.  The B operands (76 values per lane) come from an image k_pack lays out in lane order.
//         (iii) table[cid][ci] += dCe[row][ci] for the character of every row: ALSO a product on the matrix cores,
//         onehot[cid][row] x dCe[row][ci] (the one-hot operand is exact, the accumulation order is fixed: deterministic, no atomics,
//         no serial per-character loop, no LDS accumulator).  The wave that produced a channel tile of dCe consumes it: no barrier.
//   (Round 2 scattered G into LDS one word at a time -- two barriers, a clear and a 10-step serial table update per word: phases (ii)
//   + (iii) were 53 k of the kernel's 91 k cycles, and (i) ran before them on the same four waves.)
// =========================================================================================================
constexpr int EB_NP = 25;                     // (100 channels x 64 lanes) / 256 threads of the dW role
constexpr int EB_NQ = 76;                     // k-steps of 4 (tap, channel) pairs: 25 + 23 + 18 + 10
constexpr int EB_GS = 305;                    // LDS row stride of G (odd: conflict-free column reads)
__host__ __device__ constexpr int eb_tap(int q) { return q < 25 ? 0 : q < 48 ? 1 : q < 66 ? 2 : 3; }
__host__ __device__ constexpr int eb_oc0(int q) { return q < 25 ? 4 * q : q < 48 ? 8 + 4 * (q - 25) : q < 66 ? 28 + 4 * (q - 48) : 60 + 4 * (q - 66); }
__global__ __launch_bounds__(512) void k_embed_bwd(const float* __restrict__ dE, const int64_t* __restrict__ word_ids,
                                                   const int64_t* __restrict__ char_ids, const float* __restrict__ E,
                                                   const int8_t* __restrict__ argpos, const float* __restrict__ char_tab,
                                                   const float* __restrict__ wimg_b, float* __restrict__ p_cw, float* __restrict__ p_cb,
                                                   float* __restrict__ p_tab, float* __restrict__ p_unk, int Rq, int Lc,
                                                   int word_dim, int char_dim, int char_size, int cdp, int chunk, Drop dw, Drop dc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wtot = char_dim * 300;          // 10*1 + 20*2 + 30*3 + 40*4 = 300 taps per input channel
    const int ds = cdp + 16;                  // row stride of dce: the 4 k-rows of a B operand land in disjoint banks
    float* Ce = smem;                         // [chunk * Lc + 4][cdp] dropped char embeddings (+ zero rows)
    float* dce = Ce + (chunk * Lc + 4) * cdp;                     // [chunk * Lc][ds] char-embedding gradients
    int2* gp = reinterpret_cast<int2*>(dce + chunk * Lc * ds);    // [chunk][128] {arg-max position, gradient bits} (0 where inactive)
    int* cids = reinterpret_cast<int*>(gp + chunk * 128);         // [chunk * Lc + 4] character of every row
    float* Gm = reinterpret_cast<float*>(cids + ((chunk * Lc + 4 + 3) & ~3));   // [16 * ceil(rows / 16)][EB_GS] the tap matrix G (<= 300 non-zeros per word)
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, jl = lane & 15, g4 = lane >> 4;
    const int EW = word_dim + 100;
    const int rbeg = blockIdx.x * chunk, nw = min(chunk, Rq - rbeg);
    const int nrow = nw * Lc;
    const int nct = (char_dim + 15) >> 4;
    STAMP(0);
    // B operand of every MFMA of phase (ii): row-independent -> registers; the first channel tile of the wave is requested before
    // anything else so its latency hides behind the staging phases
    float wreg[EB_NQ];
    auto load_w = [&](int ct) {
        const float* src = wimg_b + (size_t)ct * EB_NQ * 64 + lane;
#pragma unroll
        for (int q = 0; q < EB_NQ; ++q) wreg[q] = src[q * 64];
    };
    if (w < 4 && w < nct) load_w(w);
    // ---- staging phase 1: G cleared, per-word metadata with all loads issued together
    {
        const int n4 = (((chunk * Lc + 15) >> 4) * 16 * EB_GS + 3) >> 2;
        for (int e = tid; e < n4; e += 512) reinterpret_cast<float4*>(Gm)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int e = tid; e < chunk * Lc + 4; e += 512) cids[e] = e < nrow ? (int)char_ids[(size_t)rbeg * Lc + e] : 0;
    for (int e = tid; e < chunk * 128; e += 512) {
        const int wi = e >> 7, oc = e & 127;
        float g = 0.f;
        int ps = 0;
        if (wi < nw && oc < 100) {
            const size_t o = (size_t)(rbeg + wi) * EW + word_dim + oc;
            g = E[o] > 0.f ? dE[o] : 0.f;                                  // relu + max: grad only to an active arg-max
            ps = argpos[(size_t)(rbeg + wi) * 100 + oc];
        }
        gp[e] = make_int2(ps, __float_as_int(g));
    }
    float uacc = 0.f;
    if (p_unk && tid < word_dim) {
        // unk_vec row of the [pad; unk; glove] table (:41): every row load of the chunk issued unconditionally, then masked
        float u[EMB_CHUNK_MAX];
#pragma unroll
        for (int wi = 0; wi < EMB_CHUNK_MAX; ++wi) u[wi] = wi < nw ? dE[(size_t)(rbeg + wi) * EW + tid] : 0.f;
#pragma unroll
        for (int wi = 0; wi < EMB_CHUNK_MAX; ++wi)
            if (wi < nw && word_ids[rbeg + wi] == 1) uacc += u[wi] * drop_mul(dw, (uint32_t)((rbeg + wi) * word_dim + tid));
    }
    __syncthreads();
    STAMP(1);
    // ---- staging phase 2: scatter G (one entry per (word, tap): row = word * Lc + pos[oc] + kk, column = the tap-major k of (kk, oc));
    //      gather the dropped-out char embeddings of every row of the chunk (+ 4 zero rows: taps past the last row)
    for (int e = tid; e < nw * 300; e += 512) {
        const int wi = e / 300, k = e - wi * 300;
        // taps in conv order: 10 x 1, 20 x 2, 30 x 3, 40 x 4
        int oc, kk;
        if (k < 10) { oc = k; kk = 0; }
        else if (k < 50) { oc = 10 + ((k - 10) >> 1); kk = (k - 10) & 1; }
        else if (k < 140) { oc = 30 + (k - 50) / 3; kk = (k - 50) - 3 * (oc - 30); }
        else { oc = 60 + ((k - 140) >> 2); kk = (k - 140) & 3; }
        const int2 pg = gp[wi * 128 + oc];
        const int col = kk == 0 ? oc : kk == 1 ? 100 + oc - 8 : kk == 2 ? 192 + oc - 28 : 264 + oc - 60;
        Gm[(wi * Lc + pg.x + kk) * EB_GS + col] = __int_as_float(pg.y);
    }
    {
        const int sh = cdp == 64 ? 6 : 7;
        for (int e = tid; e < (chunk * Lc + 4) * cdp; e += 512) {
            const int row = e >> sh, ci = e & (cdp - 1);
            float v = 0.f;
            if (row < nrow && ci < char_dim)
                v = char_tab[(size_t)cids[row] * char_dim + ci] * drop_mul(dc, (uint32_t)((rbeg * Lc + row) * char_dim + ci));
            Ce[e] = v;
        }
    }
    __syncthreads();
    STAMP(2);
    if (w >= 4) {
        // ---- (i) conv weight / bias gradients: pair q of this thread = (oc = ((tid >> 6) - 4) + 4 q, ci = cih + (tid & 63))
        const int s0 = 10 * char_dim, s1 = 20 * char_dim * 2, s2 = 30 * char_dim * 3;
        const int t = tid - 256;
        float bacc = 0.f;
        for (int cih = 0; cih < char_dim; cih += 64) {
            float wacc[EB_NP][4];
#pragma unroll
            for (int q = 0; q < EB_NP; ++q) { wacc[q][0] = 0.f; wacc[q][1] = 0.f; wacc[q][2] = 0.f; wacc[q][3] = 0.f; }
            const int ci = cih + lane, ocb = w - 4;
            for (int wi = 0; wi < nw; ++wi) {
                const int2* g = gp + wi * 128;
                const float* ce = Ce + wi * Lc * cdp + ci;
                if (cih == 0 && t < 100) bacc += __int_as_float(g[t].y);
#pragma unroll
                for (int q = 0; q < EB_NP; ++q) {
                    const int2 pg = g[ocb + 4 * q];
                    const float gv = __int_as_float(pg.y);
                    const float* row = ce + pg.x * cdp;       // taps beyond the kernel width read later rows / zero rows: unused
                    wacc[q][0] += gv * row[0];
                    wacc[q][1] += gv * row[cdp];
                    wacc[q][2] += gv * row[2 * cdp];
                    wacc[q][3] += gv * row[3 * cdp];
                }
            }
            if (ci < char_dim) {
                float* dst = p_cw + (size_t)blockIdx.x * wtot;
#pragma unroll
                for (int q = 0; q < EB_NP; ++q) {
                    const int oc = ocb + 4 * q;               // slab offsets are multiples of 4 floats: the vector stores are aligned
                    if (oc < 10) dst[oc * char_dim + ci] = wacc[q][0];
                    else if (oc < 30) *reinterpret_cast<float2*>(dst + s0 + ((oc - 10) * char_dim + ci) * 2) = make_float2(wacc[q][0], wacc[q][1]);
                    else if (oc < 60) {
                        float* d3 = dst + s0 + s1 + ((oc - 30) * char_dim + ci) * 3;
                        d3[0] = wacc[q][0]; d3[1] = wacc[q][1]; d3[2] = wacc[q][2];
                    } else
                        *reinterpret_cast<float4*>(dst + s0 + s1 + s2 + ((oc - 60) * char_dim + ci) * 4) =
                            make_float4(wacc[q][0], wacc[q][1], wacc[q][2], wacc[q][3]);
                }
            }
        }
        if (t < 100) p_cb[(size_t)blockIdx.x * 100 + t] = bacc;
    } else {
        // ---- (ii) + (iii)
        const int nrt = (nrow + 15) >> 4, nmc = (char_size + 15) >> 4;
        for (int ct = w; ct < nct; ct += 4) {
            const int ci = 16 * ct + jl;
            if (ct != w) load_w(ct);
            for (int rt = 0; rt < nrt; ++rt) {
                const float* grow = Gm + (16 * rt + jl) * EB_GS + g4;     // A operand: row = (word, position), k-slot = lane >> 4
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < EB_NQ; q += 2) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(grow[4 * q], wreg[q], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(grow[4 * q + 4], wreg[q + 1], a1, 0, 0, 0);
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int ro = 16 * rt + 4 * g4 + rr;
                    if (ro < nrow)
                        dce[ro * ds + ci] = ci < char_dim ? (a0[rr] + a1[rr]) * drop_mul(dc, (uint32_t)((rbeg * Lc + ro) * char_dim + ci)) : 0.f;
                }
            }
            STAMP(3);
            // (iii): the rows of this channel tile were written by this very wave (LDS is in order within a wave)
            for (int mc0 = 0; mc0 < nmc; mc0 += 4) {
                f32x4 acc[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int r0 = 0; r0 < nrow; r0 += 16) {
                    int cv[4];
                    float bv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int row = r0 + 4 * u + g4;
                        const bool ok = row < nrow;
                        cv[u] = ok ? cids[row] : -1;
                        bv[u] = ok ? dce[row * ds + ci] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            if (mc0 + m < nmc)
                                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[u] == 16 * (mc0 + m) + jl ? 1.f : 0.f, bv[u], acc[m], 0, 0, 0);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int co = 16 * (mc0 + m) + 4 * g4 + rr;
                        if (co < char_size && ci < char_dim)
                            p_tab[(size_t)blockIdx.x * char_size * char_dim + co * char_dim + ci] = co != 0 ? acc[m][rr] : 0.f;   // padding_idx = 0 (:51)
                    }
            }
        }
        STAMP(4);
    }
    if (p_unk && tid < word_dim) p_unk[(size_t)blockIdx.x * word_dim + tid] = uacc;
}
// words per workgroup: the smallest chunk whose grid fits one round of the 256 CUs, at least 4 (slab traffic), at most 8 and at most
// 80 (char_dim <= 64) / 48 rows (the LDS footprint: 145 / 113 KB at the bound)
int embed_bwd_chunk(int Rq, int Lc, int char_dim) {
    int c = (Rq + 255) / 256;
    c = c < 4 ? 4 : c > EMB_CHUNK_MAX ? EMB_CHUNK_MAX : c;
    const int rmax = (char_dim <= 64 ? 80 : 48) / Lc;
    return c > rmax ? (rmax < 1 ? 1 : rmax) : c;
}
void launch_embed_bwd(const float* dE, const int64_t* word_ids, const int64_t* char_ids, const float* E,
                      const int8_t* argpos, const float* char_tab, const float* wimg_b, float* p_cw, float* p_cb,
                      float* p_tab, float* p_unk, int Rq, int Lc, int word_dim, int char_dim, int char_size, Drop dw, Drop dc,
                      hipStream_t s) {
    const int cdp = char_dim <= 64 ? 64 : 128, chunk = embed_bwd_chunk(Rq, Lc, char_dim);
    const size_t shm = (size_t)((chunk * Lc + 4) * cdp + chunk * Lc * (cdp + 16) + chunk * 256 + ((chunk * Lc + 4 + 3) & ~3) +
                                ((chunk * Lc + 15) / 16) * 16 * EB_GS + 4) * sizeof(float);
    static size_t lds_ok = 0;
    ensure_dynamic_lds((const void*)k_embed_bwd, shm, lds_ok, "k_embed_bwd");
    VSL_LAUNCH(k_embed_bwd, dim3((Rq + chunk - 1) / chunk), dim3(512), shm, s, dE, word_ids, char_ids, E, argpos,
                       char_tab, wimg_b, p_cw, p_cb, p_tab, p_unk, Rq, Lc, word_dim, char_dim, char_size, cdp, chunk, dw, dc);
    static int left = 2;
    if (dbg_budget("embed_bwd")) dbg_report("embed_bwd: metadata | gather | dCe | table (waves 0-3)", 5, s, left);
}

// =========================================================================================================
// a3 backward of the TRAINABLE word table (WordEmbedding(word_vectors=None), layers_t7.py:36: nn.Embedding(word_size, word_dim,
// padding_idx=0)): gtab[wid] = sum over the occurrences of wid in the batch of dE[r][:word_dim] * dropout mask, rows that do not occur
// and row 0 are zero.  Dense gradient, as torch.optim.AdamW sees it.  One workgroup per occurrence r: it owns the table row only if no
// earlier row holds the same word, and then adds the occurrences in ascending r -- a fixed order: deterministic without atomics.  The
// table region is cleared by k_zero4 first (the reduction kernel does not cover it).
// =========================================================================================================
__global__ __launch_bounds__(256) void k_zero4(float4* __restrict__ dst, int64_t n4) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) dst[e] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ __launch_bounds__(128) void k_word_table_bwd(const float* __restrict__ dE, const int64_t* __restrict__ word_ids,
                                                        float* __restrict__ gtab, int Rq, int word_dim, Drop dw) {
    const int r = blockIdx.x, tid = threadIdx.x, EW = word_dim + 100;
    const int64_t wid = word_ids[r];
    if (wid == 0) return;                              // padding_idx = 0
    __shared__ int seen;
    if (tid == 0) seen = 0;
    __syncthreads();
    bool mine = false;
    for (int q = tid; q < r; q += 128) mine |= word_ids[q] == wid;
    if (mine) seen = 1;
    __syncthreads();
    if (seen) return;                                  // an earlier occurrence owns this table row
    float acc[4] = {0.f, 0.f, 0.f, 0.f};              // word_dim <= 512
    for (int q = r; q < Rq; ++q)
        if (word_ids[q] == wid) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = tid + 128 * j;
                if (c < word_dim) acc[j] += dE[(size_t)q * EW + c] * drop_mul(dw, (uint32_t)(q * word_dim + c));
            }
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = tid + 128 * j;
        if (c < word_dim) gtab[(size_t)wid * word_dim + c] = acc[j];
    }
}
void launch_word_table_bwd(const float* dE, const int64_t* word_ids, float* gtab, int Rq, int word_size, int word_dim, Drop dw, hipStream_t s) {
    const int64_t n4 = (int64_t)word_size * word_dim / 4;       // (the flat bucket keeps every parameter 4-float aligned; word_dim % 4 == 0)
    const int zb = (int)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024);
    VSL_LAUNCH(k_zero4, dim3(zb), dim3(256), 0, s, reinterpret_cast<float4*>(gtab), n4);
    VSL_LAUNCH(k_word_table_bwd, dim3(Rq), dim3(128), 0, s, dE, word_ids, gtab, Rq, word_dim, dw);
}

// =========================================================================================================
// final reduction of all partial slabs into the flat gradient bucket
// =========================================================================================================
// unit = 256 destination elements: 64 float4 lanes x 4 slab groups (group g sums slabs s = g, g + 4, ...), each
// thread keeps 8 independent 16-byte loads in flight; the 4 group sums are combined through LDS.
// sq[unit] = sum of squares of the unit's 256 results (also returned, valid in thread 0): every gradient of the bucket leaves this code, so the
// clip's global norm needs no pass of its own over the bucket (vsl_adamw.norm_from_backward; k_adamw adds the partials in unit order).
__device__ __forceinline__ float reduce_unit(const float* __restrict__ ws, float* __restrict__ grads, const ReduceSeg* __restrict__ segs,
                                             const int* __restrict__ blk2seg, float* __restrict__ sq, int unit, float4 (&part)[4][64],
                                             float4* keep = nullptr /* wave 0: this lane's four results (ragged units: one per 64-element pass) */) {
    const int si = blk2seg[2 * unit], off = blk2seg[2 * unit + 1];
    const ReduceSeg* __restrict__ sgp = segs + si;
    const int n = sgp->n, nsrc = sgp->nsrc, rl = sgp->rl, ds = sgp->ds, dst = sgp->dst;
    const int l4 = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int i = off + l4 * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sgp->vec) {
        if (i < n) {
            for (int q = 0; q < nsrc; ++q) {
                if (i >= sgp->vn[q]) continue;                       // this source covers only the first vn elements
                const float* p = ws + sgp->src[q] + i;
                const size_t ss = (size_t)sgp->ss[q];
                const int ns = sgp->nslabs[q];
                int s = grp;
                for (; s + 60 < ns; s += 64) {                       // 16 slabs of this group per iteration (the 256-slab units of the embedding backward)
                    float4 v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(s + 4 * u) * ss);
#pragma unroll
                    for (int u = 0; u < 16; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
                }
                for (; s + 28 < ns; s += 32) {                       // 8 slabs
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(s + 4 * u) * ss);
#pragma unroll
                    for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
                }
                for (; s < ns; s += 4) {
                    const float4 a = *reinterpret_cast<const float4*>(p + (size_t)s * ss);
                    acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
                }
            }
        }
        part[grp][l4] = acc;
        __syncthreads();
        if (grp == 0 && i < n) {
            const float4 b1 = part[1][l4], b2 = part[2][l4], b3 = part[3][l4];
            acc.x += (b1.x + b2.x) + b3.x; acc.y += (b1.y + b2.y) + b3.y; acc.z += (b1.z + b2.z) + b3.z; acc.w += (b1.w + b2.w) + b3.w;
            *reinterpret_cast<float4*>(grads + dst + (i / rl) * ds + (i % rl)) = acc;
        } else acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float q = 0.f;
        if (grp == 0) {                                  // (wave 0)
            if (keep) *keep = acc;
            q = wave_sum((acc.x * acc.x + acc.y * acc.y) + (acc.z * acc.z + acc.w * acc.w));
            if (l4 == 0) sq[unit] = q;
        }
        return q;
    } else {
        // ragged / unaligned segments (single biases, the 10- and 30-channel char-conv biases ...): same structure with
        // scalar loads -- 64 elements per pass, 4 slab groups, 8 loads in flight (a serial walk over up to 256 slabs is
        // 256 exposed memory latencies)
        float* sp = reinterpret_cast<float*>(&part[0][0]);
        float q2 = 0.f;
        float kp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int base = 0; base < 256; base += 64) {
            const int e = off + base + l4;
            float a = 0.f;
            if (e < n) {
                for (int q = 0; q < nsrc; ++q) {
                    if (e >= sgp->vn[q]) continue;
                    const float* p = ws + sgp->src[q] + e;
                    const size_t ss = (size_t)sgp->ss[q];
                    const int ns = sgp->nslabs[q];
                    int s = grp;
                    for (; s + 28 < ns; s += 32) {
                        float v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(s + 4 * u) * ss];
#pragma unroll
                        for (int u = 0; u < 8; ++u) a += v[u];
                    }
                    for (; s < ns; s += 4) a += p[(size_t)s * ss];
                }
            }
            __syncthreads();
            sp[grp * 64 + l4] = a;
            __syncthreads();
            if (grp == 0 && e < n) {
                const float r = (sp[l4] + sp[64 + l4]) + (sp[128 + l4] + sp[192 + l4]);
                grads[dst + (e / rl) * ds + (e % rl)] = r;
                q2 += r * r;
                kp[base >> 6] = r;
            }
        }
        if (grp == 0) {
            if (keep) *keep = make_float4(kp[0], kp[1], kp[2], kp[3]);
            q2 = wave_sum(q2);
            if (l4 == 0) sq[unit] = q2;
        }
        return q2;
    }
}
__global__ __launch_bounds__(256) void k_reduce(const float* __restrict__ ws, float* __restrict__ grads,
                                                const ReduceSeg* __restrict__ segs, const int* __restrict__ blk2seg,
                                                float* __restrict__ sq) {
    __shared__ float4 part[4][64];
    (void)reduce_unit(ws, grads, segs, blk2seg, sq, blockIdx.x, part);
}
void launch_reduce(const float* partial, float* grads, const ReduceSeg* segs_dev, const int* blk2seg_dev, int nblocks,
                   float* sq, hipStream_t s) {
    VSL_LAUNCH(k_reduce, dim3(nblocks), dim3(256), 0, s, partial, grads, segs_dev, blk2seg_dev, sq);
}

// =========================================================================================================
// optimizer step on the flat buckets (main_t7.py:111-112, VSLNet_t7.py:8-17): HBM-bound, 5 streams of n floats.
//   k_sqsum : OPT_BLOCKS partial sums of grads^2 (grid-stride, float4), fixed summation order -- skipped when the caller vouches that
//             `grads` is what vsl_backward left (vsl_adamw.norm_from_backward): k_reduce's per-block partials are used instead
//   k_adamw : every block re-reduces the partials (L2-resident, fixed order) -> clip factor, then updates its elements
// =========================================================================================================
__global__ __launch_bounds__(256) void k_sqsum(const float* __restrict__ g, int64_t n4, float* __restrict__ partials) {
    __shared__ float red[4];
    float s = 0.f;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = g4[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
struct AdamHp { float lr, b1, b2, eps, wd, bc1, bc2_sqrt; int hf_order; };
// clip + AdamW of one element (main_t7.py:111-112; eps 1e-6, decoupled decay)
__device__ __forceinline__ void adamw_elem(float& pe, float& me, float& ve, float ge, unsigned char dke, float coef, const AdamHp& hp) {
    const float lr = hp.lr, b1 = hp.b1, b2 = hp.b2, step_size = hp.lr / hp.bc1;
    ge *= coef;
    if (!hp.hf_order) pe *= 1.0f - lr * (dke ? hp.wd : 0.f);
    me = b1 * me + (1.0f - b1) * ge;
    ve = b2 * ve + (1.0f - b2) * ge * ge;
    if (!hp.hf_order) pe -= step_size * me / (sqrtf(ve) / hp.bc2_sqrt + hp.eps);
    else {                                   // transformers.AdamW: eps outside the bias correction, decay after the update
        pe -= step_size * hp.bc2_sqrt * me / (sqrtf(ve) + hp.eps);
        pe -= lr * (dke ? hp.wd : 0.f) * pe;
    }
}
__device__ __forceinline__ void adamw_vec4(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, const uint8_t* __restrict__ decay,
                                           int64_t i4, const float4 gv, float coef, const AdamHp& hp) {
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    float4 pv = p4[i4], mv = m4[i4], vv = v4[i4];
    const uchar4 dk = reinterpret_cast<const uchar4*>(decay)[i4];
    adamw_elem(pv.x, mv.x, vv.x, gv.x, dk.x, coef, hp); adamw_elem(pv.y, mv.y, vv.y, gv.y, dk.y, coef, hp);
    adamw_elem(pv.z, mv.z, vv.z, gv.z, dk.z, coef, hp); adamw_elem(pv.w, mv.w, vv.w, gv.w, dk.w, coef, hp);
    p4[i4] = pv; m4[i4] = mv; v4[i4] = vv;
}
// float4 elements [first, n4) with stride `stride`
__device__ __forceinline__ void adamw_span(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                           const uint8_t* __restrict__ decay, int64_t first, int64_t stride, int64_t n4, float coef, const AdamHp& hp) {
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (int64_t i = first; i < n4; i += stride) adamw_vec4(p, m, v, decay, i, g4[i], coef, hp);
}
__global__ __launch_bounds__(256) void k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, const uint8_t* __restrict__ decay, const float* __restrict__ partials,
                                               int np, int64_t n4, float clip, AdamHp hp, float* __restrict__ norm_out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int k = threadIdx.x; k < np; k += 256) s += partials[k];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = norm;
    const float coef = clip > 0.f ? fminf(1.0f, clip / (norm + 1e-6f)) : 1.0f;       // clip_grad_norm_
    adamw_span(p, g, m, v, decay, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256, n4, coef, hp);
}

// =========================================================================================================
// The step's tail as ONE launch (single process, vsl_io.fused_step): the final reduction, the clip's global norm and the AdamW update.
//   phase 1  the workgroup reduces its (at most TAIL_UNITS) units -- unit u = blockIdx.x + j gridDim.x, the code of k_reduce -- keeps their
//            results in LDS and adds their sums of squares;
//   hand-off thread 0 publishes that sum as an 8-byte {launch tag, value} granule (ONE agent-scope write-through store: the mechanism of the
//            fused rnn head, kernels_lstm.hip); wave 0 of every workgroup polls all gridDim.x granules until they carry this launch's tag
//            (~2.5 us across XCDs) while waves 1 - 3 add the partials of the EARLY reduction (an earlier launch);
//   phase 2  clip factor from the norm (fixed summation order: deterministic for a given grid), then AdamW on the elements THIS workgroup
//            reduced (from LDS) and on its share of the early reduction's units (their gradients are in `grads` since an earlier launch).
// Nothing but the granules crosses workgroups inside the launch, so there is no L2 write-back / invalidate in it (a first version that
// updated the bucket grid-stride behind release / acquire fences took 38 - 78 us for 256 - 1024 workgroups: profiles/r06_notes.md).
// All gridDim.x workgroups must be resident at once (they wait for each other): the launcher keeps the grid <= 4 per CU (16 of a CU's 32
// wave slots, 8 KB of LDS each) and nothing else is in flight on the caller's streams at this point of the step.
// =========================================================================================================
constexpr int TAIL_UNITS = 4;
struct FusedTail {
    float *p, *m, *v;
    const uint8_t* decay;
    const float* sq_early;      // the early reduction's per-unit sums of squares
    const int* blk2seg_early;   // ... and its unit table
    int n_early;
    unsigned long long* gran;   // [gridDim.x] {tag, value}
    unsigned tag;
    int nunits;
    float clip;
    AdamHp hp;
    float* norm_out;
};
// AdamW on the 256 elements of one reduction unit: lane l4 of the calling wave owns the unit's float4 l4 (ragged units: element l4 of each 64-element pass)
__device__ __forceinline__ void adamw_unit(const FusedTail& f, const ReduceSeg* __restrict__ segs, const int* __restrict__ blk2seg, int unit, int l4,
                                           const float4 g, float coef) {
    const int si = blk2seg[2 * unit], off = blk2seg[2 * unit + 1];
    const ReduceSeg* __restrict__ sgp = segs + si;
    const int n = sgp->n, rl = sgp->rl, ds = sgp->ds, dst = sgp->dst;
    if (sgp->vec) {
        const int i = off + l4 * 4;
        if (i < n) adamw_vec4(f.p, f.m, f.v, f.decay, (int64_t)((dst + (i / rl) * ds + (i % rl)) >> 2), g, coef, f.hp);
    } else {
        const float gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = off + 64 * k + l4;
            if (e < n) {
                const int d = dst + (e / rl) * ds + (e % rl);
                float pe = f.p[d], me = f.m[d], ve = f.v[d];
                adamw_elem(pe, me, ve, gs[k], f.decay[d], coef, f.hp);
                f.p[d] = pe; f.m[d] = me; f.v[d] = ve;
            }
        }
    }
}
// the gradients of one EARLY unit back from the bucket, in the layout adamw_unit takes
__device__ __forceinline__ float4 unit_grads(const float* __restrict__ grads, const ReduceSeg* __restrict__ segs, const int* __restrict__ blk2seg,
                                             int unit, int l4) {
    const int si = blk2seg[2 * unit], off = blk2seg[2 * unit + 1];
    const ReduceSeg* __restrict__ sgp = segs + si;
    const int n = sgp->n, rl = sgp->rl, ds = sgp->ds, dst = sgp->dst;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sgp->vec) {
        const int i = off + l4 * 4;
        if (i < n) g = *reinterpret_cast<const float4*>(grads + dst + (i / rl) * ds + (i % rl));
    } else {
        float gs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = off + 64 * k + l4;
            if (e < n) gs[k] = grads[dst + (e / rl) * ds + (e % rl)];
        }
        g = make_float4(gs[0], gs[1], gs[2], gs[3]);
    }
    return g;
}
__global__ __launch_bounds__(256) void k_reduce_adamw(const float* __restrict__ ws, float* __restrict__ grads,
                                                      const ReduceSeg* __restrict__ segs, const int* __restrict__ blk2seg,
                                                      float* __restrict__ sq, FusedTail f) {
    __shared__ float4 part[4][64];
    __shared__ float4 kept[TAIL_UNITS][64];
    __shared__ float red[4];
    const int l4 = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float mine = 0.f;                                    // (thread 0) sum of squares of this workgroup's units, in unit order
#pragma unroll
    for (int j = 0; j < TAIL_UNITS; ++j) {
        const int u = blockIdx.x + j * gridDim.x;        // (block-uniform)
        if (u < f.nunits) {
            float4 k4 = make_float4(0.f, 0.f, 0.f, 0.f);
            mine += reduce_unit(ws, grads, segs, blk2seg, sq, u, part, &k4);
            if (wv == 0) kept[j][l4] = k4;
            __syncthreads();                             // `part` is free again
        }
    }
    using gu64 = unsigned long long;
    if (threadIdx.x == 0)
        __hip_atomic_store(f.gran + blockIdx.x, ((gu64)f.tag << 32) | __float_as_uint(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float s = 0.f;
    if (threadIdx.x >= 64) {                             // waves 1 - 3: the early reduction's partials (an earlier launch: plain loads)
        for (int k = threadIdx.x - 64; k < f.n_early; k += 192) s += f.sq_early[k];
    } else {                                             // wave 0 polls (the other waves wait at the barrier below: no polling traffic of theirs)
        for (int k = threadIdx.x; k < (int)gridDim.x; k += 64) {
            gu64 g = __hip_atomic_load(f.gran + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while ((unsigned)(g >> 32) != f.tag) {
                __builtin_amdgcn_s_sleep(4);
                g = __hip_atomic_load(f.gran + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s += __uint_as_float((unsigned)g);
        }
    }
    s = wave_sum(s);
    if (l4 == 0) red[wv] = s;
    __syncthreads();
    const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    if (blockIdx.x == 0 && threadIdx.x == 0 && f.norm_out) *f.norm_out = norm;
    const float coef = f.clip > 0.f ? fminf(1.0f, f.clip / (norm + 1e-6f)) : 1.0f;       // clip_grad_norm_
    // phase 2: wave w takes the workgroup's unit w ...
    {
        const int u = blockIdx.x + wv * gridDim.x;
        if (wv < TAIL_UNITS && u < f.nunits) adamw_unit(f, segs, blk2seg, u, l4, kept[wv][l4], coef);
    }
    // ... and one early unit per wave and round (the early table sits in front of the late one: blk2seg_early)
    for (int e = 4 * blockIdx.x + wv; e < f.n_early; e += 4 * gridDim.x)
        adamw_unit(f, segs, f.blk2seg_early, e, l4, unit_grads(grads, segs, f.blk2seg_early, e, l4), coef);
}
int reduce_adamw_resident(int cus) {       // workgroups of k_reduce_adamw that may wait for each other: half of what the occupancy calculator admits, at most 4 per CU
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_reduce_adamw, 256, 0) != hipSuccess) return 0;
    return std::max(0, std::min(4, per_cu / 2)) * cus;
}
int reduce_adamw_grid(int nunits, int cap) {      // the grid launch_reduce_adamw takes for `nunits` late units with at most `cap` resident workgroups, 0 = does not fit
    if (nunits <= 0 || cap <= 0 || (int64_t)nunits > (int64_t)TAIL_UNITS * cap) return 0;
    return std::min(nunits, cap);
}
void launch_reduce_adamw(const float* partial, float* grads, const ReduceSeg* segs_dev, const int* blk2seg_dev, int nunits, float* sq,
                         const float* sq_early, const int* blk2seg_early, int n_early, int grid, unsigned long long* gran, unsigned tag,
                         float* params, float* m, float* v, const uint8_t* decay_mask, float lr, float b1, float b2, float eps, float wd,
                         float clip, float bc1, float bc2_sqrt, float* norm_out, int hf_order, hipStream_t s) {
    FusedTail f;
    f.p = params; f.m = m; f.v = v; f.decay = decay_mask; f.sq_early = sq_early; f.blk2seg_early = blk2seg_early; f.n_early = n_early;
    f.gran = gran; f.tag = tag; f.nunits = nunits; f.clip = clip; f.hp = AdamHp{lr, b1, b2, eps, wd, bc1, bc2_sqrt, hf_order};
    f.norm_out = norm_out;
    VSL_LAUNCH(k_reduce_adamw, dim3(grid), dim3(256), 0, s, partial, grads, segs_dev, blk2seg_dev, sq, f);
}
void launch_adamw(float* params, const float* grads, float* m, float* v, const uint8_t* decay_mask, float* partials, int64_t n,
                  float lr, float b1, float b2, float eps, float wd, float clip, float bc1, float bc2_sqrt, float* norm_out,
                  hipStream_t s, int hf_order, const float* sq_from_backward, int nsq) {
    const int64_t n4 = n / 4;                    // the bucket is a multiple of 4 floats (every tensor is 16-byte aligned)
    if (!sq_from_backward) VSL_LAUNCH(k_sqsum, dim3(OPT_BLOCKS), dim3(256), 0, s, grads, n4, partials);
    const int nb = (int)std::min<int64_t>(1024, (n4 + 255) / 256);
    VSL_LAUNCH(k_adamw, dim3(nb), dim3(256), 0, s, params, grads, m, v, decay_mask, sq_from_backward ? sq_from_backward : partials,
                       sq_from_backward ? nsq : OPT_BLOCKS, n4, clip, AdamHp{lr, b1, b2, eps, wd, bc1, bc2_sqrt, hf_order}, norm_out);
}

}  // namespace vsl
