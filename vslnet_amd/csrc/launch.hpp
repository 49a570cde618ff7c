// Host-side launcher prototypes + small POD argument structs shared by kernels_fwd.hip, kernels_bwd.hip, api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "common.hpp"

namespace vsl {

struct PackJob {          // one weight -> packed B-operand copy (see common.hpp pack_index)
    int src;              // float offset into the flat parameter buffer
    int dst;              // float offset into the pack buffer
    int kn, cn;           // extent of the contraction index / of the output-column index covered by this job
    int ld;               // leading dimension of the source matrix
    int transpose;        // 0: Bm[k][c] = W[c][k] (forward pack)   1: Bm[k][c] = W[k][c] (data-gradient pack)  2: copy  3: char-conv image  4: zero fill  8: char-conv B-operand image of k_embed_bwd
                          // 9 / 10: W_hh (512, 128) in the register order of k_lstm1_fwd / k_lstm1_bwd (kn = 65536 floats)
                          // 5: bf16 forward pack   6 / 7: SPLIT packs (three bf16 planes h, m, l; common.hpp pack3_index) of the forward / data-gradient operand
    int ncols;            // total columns of the packed operand
    int k_off, col_off;   // placement inside the packed operand
    int ktot;             // split packs: contraction extent of the whole operand incl. padding (plane size); 0 = kn
    int kfill;            // split packs: k rows this job writes (rows >= kn as zeros); 0 = kn rounded up to 16
};

struct CharConvPtrs { const float* w[4]; const float* b[4]; };
struct CharConvGrads { float* w[4]; float* b[4]; };

struct HeadArgs {
    const float* feat;    // (R,128) encoder / rnn features
    const float* ln_g;    // nullptr for the rnn predictor
    const float* ln_b;
    const float* W0pack;  // forward pack of (128, 256)
    const float* b0;
    const float* w1;      // (128)
    const float* b1;      // (1)
    float* hid;           // (R,128) relu output, saved
    float* lnfeat;        // (R,128) LN(feat), saved for the weight gradient
    float* logits;        // (R)
};

struct HeadBwdArgs {
    const float* dlogit;  // (R)
    const float* hid;     // (R,128)
    const float* feat;    // (R,128)
    const float* ln_g;
    const float* W0Tpack; // transpose pack, ncols = 256
    const float* w1;
    float* gz;            // (R,128) grad wrt the pre-relu activation (G operand of the weight gradient)
    float* dfeat;         // (R,128)
    float* dx;            // (R,128) grad wrt the gated features through this head
    float* p_b0;          // partial slabs [ntiles][128]
    float* p_w1;          // [ntiles][128]
    float* p_b1;          // [ntiles]
    float* p_lng;         // [ntiles][128]
    float* p_lnb;         // [ntiles][128]
    // vsl_io.fused_loss, whole tiles (T % 32 == 0): the tile's seeds dlogit[r] = cs * (softmax(logits of r's sample)[t] - [t == label]) are computed HERE
    // (the CrossEntropy gradient needs nothing but the sample's own T logits), so the loss launch is not on the dependent chain; logits == nullptr: read dlogit
    const float* logits;  // (B, T)
    const int64_t* label; // (B)
    float cs;             // w_loc * inv_batch
    int T;
};

// generic weight-gradient job: dW[n][k] = sum_r G[r][n] * A[r][k]  over row chunks -> partial slabs
struct WgradJob {
    const float* G[3];    // N = 128 * nG ; each (R,128)
    const float* A[4];    // column blocks of 128 (ldA == 128 each) when nA > 0
    const float* Afull;   // or one (R, K) matrix with leading dimension K (nA == 0)
    int nG, nA, K;
    int R;
    int ldg;              // row stride of the G blocks in floats (0 = 128): the LSTM gate gradients are 512 wide
    int drop_on_A;        // apply dropout to Afull on load (VisualProjection input)
    int a_bf16;           // Afull points to bfloat16 data (bf16 throughput mode: the features), widened on load
    Drop dp;
    float* out;           // partial slabs [nchunk][N][K]
    float* out_bias[3];   // partial slabs [nchunk][128] per G block (nullable)
    int rows;             // rows per chunk = workgroup of THIS job (0 = WG_ROWS; a multiple of 16): see wgrad_rows_one_round
};
#ifndef VSL_WG_ROWS
#define VSL_WG_ROWS 256
#endif
constexpr int WG_ROWS = VSL_WG_ROWS;     // rows per weight-gradient chunk = workgroup (one partial slab each)
__host__ __device__ inline int wgrad_rows(const WgradJob& j) { return j.rows > 0 ? j.rows : WG_ROWS; }
__host__ __device__ inline int wgrad_chunks(int R, int rows) { return (R + rows - 1) / rows; }
// A batch of `cols` 128 x 128 output blocks over R rows that is launched into an EMPTY chip (the step's last weight-gradient batch) takes
// ceil(cols * chunks / CUs) rounds of one-workgroup-per-CU chunks: 384 workgroups of 256 rows are two rounds, the second half empty.  Longer chunks
// that make it whole rounds cost rows / 256 each: -> the smallest multiple of 16 rows >= WG_ROWS with cols * chunks <= CUs * floor(rounds at WG_ROWS).
inline int wgrad_rows_whole_rounds(int R, int cols, int cus) {
    const int wgs = cols * wgrad_chunks(R, WG_ROWS);
    if (cus <= 0 || cols <= 0 || wgs <= cus || wgs % cus == 0) return WG_ROWS;
    const int target = cus * (wgs / cus);
    for (int rows = WG_ROWS + 16; rows <= 4 * WG_ROWS; rows += 16)
        if (cols * wgrad_chunks(R, rows) <= target) return rows;
    return WG_ROWS;
}
constexpr int MAX_WJOBS = 12;
struct WgradBatch { WgradJob j[MAX_WJOBS]; int n; int start[MAX_WJOBS + 1]; };   // start: first workgroup of each job (launch_wgrad fills it)

struct ReduceSeg {        // grads[dst + (i / rl) * ds + i % rl] = sum over sources q, slabs s of partial[src[q] + s * ss[q] + i],  i < n
    int dst, n;
    int rl, ds;           // destination row length / row stride (rl == n, ds == 0 for a contiguous destination)
    int nsrc;
    int src[4], nslabs[4], ss[4]; // float offsets into the WORKSPACE (partial arena or any saved buffer), slab count, slab stride
    int vec;                      // 1: n, every src, ss and vn are multiples of 4 floats -> 16-byte loads/stores
    int vn[4];                    // source q contributes to elements i < vn[q] only (shorter sequences of a shared table)
};

// Opt a kernel into more than the default 64 KiB of dynamic LDS.  Requests exactly what the launch needs (static LDS
// counts against the same 160 KiB), grows monotonically, and reports -- instead of silently poisoning
// hipGetLastError() -- when the runtime refuses.
// Every kernel of a multi-stream step is launched with a STOP EVENT on its own dispatch packet (hipExtLaunchKernelGGL): a cross-stream
// ordering point is then a bare hipStreamWaitEvent on the producer's last kernel.  hipEventRecord puts a marker packet into the producer
// stream instead, which costs it 4-5 us per fork and 7 us more per join (tools/ubench/event_fork.hip: chain 21.6 / fork by record 26.9 /
// fork by stop event 23.3 / stop event on every dispatch, no waiter 21.6 us per link).  api.hip hands out the events (none = plain launch).
// The same hook serves the built-in profiler (vsl_profile_select): a profiled launch gets a timing START and STOP event on its own packet, so
// the measured interval is the kernel's execution and the stream carries no extra marker packets.
void vsl_launch_events(hipStream_t s, hipEvent_t* start, hipEvent_t* stop);
#define VSL_LAUNCH(kernel, grid, block, shm, stream, ...)                                                                   \
    do {                                                                                                                   \
        hipEvent_t st__ = nullptr, sp__ = nullptr;                                                                         \
        vsl::vsl_launch_events(stream, &st__, &sp__);                                                                      \
        if (sp__) hipExtLaunchKernelGGL(kernel, grid, block, shm, stream, st__, sp__, 0, __VA_ARGS__);                     \
        else hipLaunchKernelGGL(kernel, grid, block, shm, stream, __VA_ARGS__);                                            \
    } while (0)

inline void ensure_dynamic_lds(const void* func, size_t bytes, size_t& granted, const char* name) {
    if (bytes <= granted || bytes <= 64 * 1024) return;
    const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        fprintf(stderr, "[vslnet_hip] hipFuncSetAttribute(%s, %zu B dynamic LDS) failed: %s\n", name, bytes, hipGetErrorString(e));
        (void)hipGetLastError();
    } else {
        granted = bytes;
    }
}

// ---------------------------------------------------------------- forward
void launch_pack(const float* params, float* pack, const PackJob* jobs_dev, int njobs, hipStream_t s);
// bf16 throughput mode: X bf16 (R, Dv), packed bf16 weight (PackJob type 5), fp32 accumulate / bias / output
void launch_vproj_fwd_bf16(const uint16_t* X, const uint16_t* Wpack16, const float* bias, float* Y, int R, int Dv, Drop dp, hipStream_t s);
// fp32-grade on the bf16 matrix cores (kernels_split.hip): W3 = split pack (PackJob type 6 / 7) of the (Dv, 128) operand
void launch_vproj_fwd3(const float* X, const uint16_t* W3, const float* bias, float* Y, int R, int Dv, Drop dp, hipStream_t s,
                       int seg = 0, int stride = 0, int off = 0);   // seg > 0: rows of one time chunk (see launch_linear_fwd3)
void launch_linear_fwd3(const float* A, const uint16_t* W3, const float* bias /* nullable */, float* Y, int R, int K, hipStream_t s, int ncols = 128,
                        int seg = 0, int stride = 0, int off = 0);      // ncols: columns of the operand and of Y's rows; seg > 0: rows of one time chunk of a (B, T, .) tensor
void launch_linear_bwd_data3(const float* G, const uint16_t* WT3, float* dA, int R, int K, int Kc, hipStream_t s);   // Kc: columns of the split pack
void launch_embed_fwd(const int64_t* word_ids, const int64_t* char_ids, const float* pad_vec, const float* unk_vec,
                      const float* glove, const float* char_tab, CharConvPtrs cc, const float* wimg, float* E, int8_t* argpos,
                      int Rq, int Lc, int word_dim, int char_dim, Drop dw, Drop dc, hipStream_t s);
void launch_linear_fwd(const float* A, const float* Wpack, const float* bias, float* Y, int R, int K, hipStream_t s);
// LN1 + fused QKV projection at the end of the conv-block kernel (row-local on the owner rows)
struct QkvFuse {
    const float *ln_g, *ln_b, *bq, *bk, *bv;
    float *h1, *q, *k, *v;
    Drop d1;
};
// fused conv block of one encoder application (kernels_enc.hip): 4 layers + LN1 / QKV in one launch, 12-row recomputed halo
struct CbFwdArgs {
    const uint16_t* W3[4];         // split packs (PackJob type 6) of the four pointwise weights
    const uint16_t* Wqkv3;         // split pack of the fused (128, 384) QKV operand
    const float *xin, *pos;
    float* x0_out;
    const float *ln_g[4], *ln_b[4], *dw_w[4], *pw_b[4];
    float *y[4], *u[4];
    uint32_t* relu_mask[4];
    Drop dp[4];
    QkvFuse qf;
    int R, L;
};
void launch_convblock_fwd(const CbFwdArgs& a, hipStream_t s);
// the attention-output backward of one encoder application (a8 :183-190 backward; kernels_bwd.hip attn_out_bwd_tile)
struct AttnOutBwdArgs {
    const float *dy, *dy2, *r_in, *ln_g, *WTpack;
    float *g_o, *dr, *p_lng, *p_lnb;
    Drop d4, d5;
};
// gating + HighLightLayer + CQConcatenate backward of one 32-row tile (kernels_bwd.hip k_cqcat_bwd / tile_bodies.hpp cqcat_bwd_tile)
struct CqcatBwdArgs {
    const float *dg0, *dg1, *dg2, *dh_loss, *f2, *hscore, *wh, *W1Tpack;
    float *df2, *df1, *p_wh, *p_bh;
    // vsl_io.fused_loss: the highlight loss' seed (elementwise: k_loss_fused's expression) computed in place of the dh_loss read; h_lab == nullptr: read dh_loss
    const int64_t* h_lab;
    const float* vmask;
    float w_hl, mask_sum;
};
// a8 backward, first half (k_qkv_bwd's work), hosted by the conv block's backward kernel on its 56-row window:
// dy = dr + LN1^T(([dQ | dK | dV] [Wq; Wk; Wv]) * m1) -- the conv block's incoming gradient never goes through memory
struct QkvBwdFuse {
    const float *dq, *dk, *dv;     // (R,128) each
    const float* x;                // (R,128) LN1 input = the conv block's output y3
    const float* dr;               // (R,128) gradient of the residual path (attention-output backward)
    const float* ln_g;             // LN1 gamma
    const uint16_t* WT3;           // split pack (type 7) of [Wq; Wk; Wv]: 384 contraction rows, 128 columns
    float *p_lng, *p_lnb;          // partial slabs [ntiles][128]
    Drop d1;
};
struct CbBwdArgs {
    const uint16_t* WT3[4];        // split packs (PackJob type 7) of the pointwise weights' data-gradient operand
    const float* dy;               // (R,128) grad wrt the block output (unused when qkv is set)
    const float* x[4];             // LayerNorm inputs of layers 0..3 (x0, y0, y1, y2)
    const uint32_t* relu_mask[4];
    const float *ln_g[4], *ln_b[4], *dw_w[4];
    Drop dp[4];
    float* gz[4];                  // out (R,128): dz per layer, G operand of the pointwise weight gradients
    float* dx0;                    // out (R,128): grad wrt the block input (x + pos)
    float *p_lng[4], *p_lnb[4], *p_dw[4];   // partial slabs [ntiles][128] / [ntiles][128 * 7]
    int R, L;
    // what the workgroup goes on with, on its own 32 rows of dx0 (1, 2: whole-tile instantiation only: convblock_bwd_hosts_tail):
    // 0 nothing ; 1 the attention-output backward of the encoder pass below (tail_ao; its dy = dx0) ; 2 the CQConcatenate backward (tail_cq; its dg0 = dx0)
    int tail;
    AttnOutBwdArgs tail_ao;
    CqcatBwdArgs tail_cq;
    // 3 (sample tiles only): the Embedding linear's data gradient dA (R, K) = dx0 WT3 -- the query pass' conv block goes on with k_linear_bwd_data3's work
    const uint16_t* tail_lin_WT3;  // split transpose pack, Kc >= K columns (a multiple of 512)
    float* tail_lin_dA;
    int tail_lin_K, tail_lin_Kc;
    int qkv;                       // 1: the workgroup first computes dy from qk (whole-tile instantiation only: convblock_bwd_hosts_tail)
    QkvBwdFuse qk;
};
void launch_convblock_bwd(const CbBwdArgs& a, hipStream_t s);
bool convblock_bwd_hosts_tail(int R, int L);      // does launch_convblock_bwd honour CbBwdArgs::tail for this shape?
bool convblock_bwd_hosts_linear(int R, int L);    // ... and tail 3 (the Embedding linear's data gradient)?
bool convblock_bwd_hosts_qkv(int R, int L);       // ... and CbBwdArgs::qkv?  (its LN1 partial slabs are then convblock_slabs(R, L) many)
int convblock_slabs(int R, int L);        // partial slabs per parameter of launch_convblock_bwd (= its grid)
void launch_attn_fwd(const float* Q, const float* K, const float* V, const float* mask, float* att, float* lse, int B,
                     int L, int H, int b_off, Drop d2, hipStream_t s);
void launch_attn_out_fwd(const float* att, const float* x, const float* ln_g, const float* ln_b, const float* Wpack,
                         const float* bo, float* r_out, float* h2_out, float* y_out, int R, Drop d3, Drop d4, Drop d5,
                         hipStream_t s);
// attention core + output block of one encoder application in one launch (kernels_enc.hip)
struct AttnBlockArgs {
    const float *Q, *K, *V, *mask, *x;            // q, k, v (R,128) ; key mask (B, L) ; residual input x = conv-block output
    const float *ln_g, *ln_b, *Wpack, *bo;         // LN2, out_layer (forward pack), its bias
    float *att, *lse, *r_out, *h2_out, *y_out;     // saved: att (R,128), LSE (B,H,L), r, h2 ; y = block output
    int L, b_off;
    Drop d2, d3, d4, d5;
    // head_tail (L <= 128 only: attn_block_fwd_hosts_heads): the workgroup goes on with both span heads on its 32 rows (tile_bodies.hpp
    // head_fwd_tile) -- the second predictor pass: hs.feat = the first pass' output (memory), he's features = this kernel's y tile
    int head_tail;
    HeadArgs hs, he;
    const float *head_x, *head_vmask;
};
void launch_attn_block_fwd(const AttnBlockArgs& a, int B, hipStream_t s);
inline bool attn_block_fwd_hosts_heads(int L) { return L <= 128; }
// the query branch as ONE sample-local launch (kernels_query.hip): Embedding.linear + the whole FeatureEncoder application at L = Lq <= 32,
// one workgroup per sample, 17 KB of LDS.  Saves the same tensors as linear_fwd + convblock_fwd + attn_block_fwd.
struct QueryFwdArgs {
    const float* E;                // (Rq, EW) concatenated word + char embedding rows
    const uint16_t* Wemb3;         // split pack (type 6) of the (128, EW) Embedding.linear weight, K padded to 16
    const float* b_emb;
    float* qf;                     // (Rq,128) Embedding.linear output (nullable)
    const float* pos;              // positional table
    float* x0;
    const uint16_t* W3[4];         // split packs of the pointwise weights
    const uint16_t* Wqkv3;         // split pack of the fused (128, 384) q,k,v operand
    const uint16_t* Wo3;           // split pack of the out_layer weight
    const float *ln_g[4], *ln_b[4], *dw_w[4], *pw_b[4];
    float *y[4], *u[4];
    uint32_t* relu_mask[4];
    Drop dp[4];
    const float *ln1_g, *ln1_b, *bq, *bk, *bv;
    float *h1, *q, *k, *v;
    Drop d1;
    const float* mask;             // (B, L) key mask
    const float *ln2_g, *ln2_b, *bo;
    float *att, *lse, *r, *h2, *out;
    Drop d2, d3, d4, d5;
    int EW, L, b_off;
};
bool query_fused_ok(int L, int H, int EW);         // does the sample-local path take this query length / embedding width?
void launch_query_fwd(const QueryFwdArgs& a, int B, hipStream_t s);
// the column kernel folded away (cq_col_folds): k_cq_score gets one extra workgroup per sample for the WeightedPool / pooled-bias path (`pool`),
// k_cq_out computes the sample's column softmax and M itself (S_fold / Scol_fold) and launch_cq_col is not called
struct CqPoolArgs { const float *pool_w, *Wcat, *bcat; float *alpha, *pooled, *pb; };
bool cq_col_folds(int T, int Lq);
void launch_cq_score(const float* C, const float* Qf, const float* qmask, const float* w4C, const float* w4Q,
                     const float* w4mlu, float* S, float* Srow, int B, int T, int Lq, int b_off, Drop dc, Drop dq,
                     hipStream_t s, const CqPoolArgs* pool = nullptr);
void launch_cq_col(const float* C, const float* Qf, const float* S, const float* cmask, const float* qmask,
                   const float* pool_w, const float* Wcat, const float* bcat, float* Scol, float* Mpart /*[B][ntile][Lq][128]*/,
                   float* alpha, float* pooled, float* pb, int B, int T, int Lq, hipStream_t s);
void launch_cq_out(const float* C, const float* Qf, const float* Srow, const float* Mpart, float* M, const float* Wpack,
                   const float* bias, float* cat_out, float* out, const float* W1pack, const float* pb, const float* wh,
                   const float* bh, const float* vmask, float* f2, float* hscore, float* gated, int B, int T, int Lq, hipStream_t s,
                   const float* S_fold = nullptr, float* Scol_fold = nullptr);
void launch_head_fwd(const HeadArgs& a0, const HeadArgs& a1, const float* x, const float* vmask, int R, hipStream_t s);

// ---------------------------------------------------------------- losses / eval
void launch_loss(const float* sl, const float* el, const float* h, const int64_t* s_lab, const int64_t* e_lab,
                 const int64_t* h_lab, const float* vmask, int B, int T, float inv_batch, float mask_sum_override,
                 float w_loc, float w_hl, float* scratch, float* losses /*[4]: loc, hl, total, mask_sum*/, float* d_sl,
                 float* d_el, float* d_h, hipStream_t s, unsigned* counter = nullptr);
void launch_extract_index(const float* sl, const float* el, int64_t* si, int64_t* ei, int B, int T, hipStream_t s);

// ---------------------------------------------------------------- backward
// fuse != nullptr: the end head's workgroups continue with that attention-output backward on their own tile of dfeat (which is then NOT stored)
void launch_head_bwd(const HeadBwdArgs& a0, const HeadBwdArgs& a1, int R, hipStream_t s, const AttnOutBwdArgs* fuse = nullptr);
// a15 DynamicRNN (layers_t7.py:302-313): recurrent part of nn.LSTM(128, 128); the input projection x W_ih^T is a plain GEMM
// (kernels_lstm.hip: one-sample workgroups for B <= 256, which read k_pack's register-order images of W_hh -- PackJob types 9 / 10 -- and
// save tanh(c_t) in tseq; 4-sample MFMA groups beyond, which read W_hh itself)
void launch_lstm_fwd(const float* gi, const float* Whh, const float* Wimg, const float* bih, const float* bhh, const float* mask, float* gates,
                     float* cseq, float* tseq, float* hprev, float* out, int B, int T, hipStream_t s, int t0 = 0, int t1 = -1);   // steps [t0, t1)
void launch_lstm_bwd(const float* dout, const float* dout2, const float* mask, const float* gates, const float* cseq, const float* tseq,
                     const float* Whh, const float* Wimg, float* dG, int B, int T, hipStream_t s, float* carry = nullptr, int t0 = 0, int t1 = -1);
                     // steps [t0, t1) in reverse; carry (B, 2, 128): dc / dh handed from one time chunk to the next
constexpr int LSTM_IMG_FLOATS = 4 * D * D;
// The rnn head as one launch per direction: three workgroups per sample (start LSTM, W_ih projection, end LSTM) handing steps over through
// {tag, value} granules.  B <= RNN_FUSED_MAX_B keeps all 3 B workgroups resident beside the side streams' kernels; beyond it the chunked
// launches above.  `epoch` is new for every launch (a NaN-patterned 21-bit counter); api.hip clears a plan's granule buffers once per
// workspace and once more whenever the counter has wrapped (rnn_granules_fresh), so a stale word never carries the live tag.
constexpr int RNN_FUSED_MAX_B = 80;
struct RnnFwdArgs {
    const float* gi0;                 // (R, 512) input projection of the start LSTM
    const float* Whh[2];              // PackJob type 9 images of W_hh
    const float* Wih1;                // PackJob type 9 image of the end LSTM's W_ih
    const float* bih[2]; const float* bhh[2];
    const float* mask;
    float* gates[2]; float* cseq[2]; float* tseq[2]; float* hprev[2]; float* out[2];
    unsigned long long* h_gran;       // (R, 128) start LSTM's h * mask
    unsigned long long* gi_gran;      // (R, 512) W_ih1 (h * mask)
    unsigned epoch;
    int B, T;
};
struct RnnBwdArgs {
    const float* dout[2];             // (R, 128) gradients wrt the LSTM outputs coming from the span heads
    const float* mask;
    const float* gates[2]; const float* cseq[2]; const float* tseq[2];
    const float* Whh[2];              // PackJob type 10 images of W_hh
    const float* Wih1;                // PackJob type 10 image of the end LSTM's W_ih
    float* dG[2];
    unsigned long long* dg_gran;      // (R, 512) end LSTM's gate gradients
    unsigned long long* dx_gran;      // (R, 128) dG W_ih1: the start LSTM's second incoming gradient
    unsigned epoch;
    int B, T;
};
bool rnn_fused_ok(int B);
void launch_rnn_fwd(const RnnFwdArgs& a, hipStream_t s);
void launch_rnn_bwd(const RnnBwdArgs& a, hipStream_t s);      // one image = W_hh in the register order of k_lstm1_fwd (type 9) / k_lstm1_bwd (type 10)
void launch_wgrad(const WgradBatch& wb, hipStream_t s);     // kernels_wgrad.hip
void launch_attn_out_bwd(const float* dy, const float* dy2, const float* r, const float* ln_g, const float* WTpack, float* g_o,
                         float* dr, float* p_lng, float* p_lnb, int R, Drop d4, Drop d5, hipStream_t s);
void launch_attn_bwd(const float* Q, const float* K, const float* V, const float* att, const float* dr, const float* lse,
                     const float* mask, float* dQ, float* dK, float* dV, int B, int L, int H, int b_off, Drop d2,
                     Drop d3, hipStream_t s);
int attn_bwd_dq_slabs(int L);             // L > 256: dQ is written as this many (R, 128) partial slabs (one per 256-key block); k_qkv_bwd adds them
void launch_qkv_bwd(const float* dQ, const float* dK, const float* dV, const float* x, const float* dr,
                    const float* ln_g, const uint16_t* WT3 /* split pack of the (384, 128) operand */, float* dx, float* p_lng, float* p_lnb, int R,
                    Drop d1, hipStream_t s, int dq_slabs = 1);
struct HlSeed { const int64_t* h_lab; const float* vmask; float w_hl, mask_sum; };     // (CqcatBwdArgs' inline highlight seed)
void launch_cqcat_bwd(const float* dg0, const float* dg1, const float* dg2, const float* dh_loss, const float* f2,
                      const float* hscore, const float* wh, const float* W1Tpack, float* df2, float* df1, float* p_wh,
                      float* p_bh, int R, hipStream_t s, const HlSeed* hl = nullptr);
struct CqBwdArgs {
    const float *df1, *df2, *C, *Qf, *Srow, *Scol, *M, *alpha, *pooled;      // saved forward tensors / incoming grads
    const float *WcqaT;                                                        // transpose pack of cqa_linear (ncols 512)
    const float *w4C, *w4Q, *w4mlu, *pool_w, *Wcat;
    float *dC;            // (B,T,128) out: total grad wrt the video encoder output
    float *dQ;            // (B,Lq,128) out: total grad wrt the query encoder output
    float *dSr, *dSs;     // (B,T,Lq) scratch: row-softmax backward, dS_col
    float *P1;            // [B][ntile][2][Lq][128] per-tile partials of dM and dQ(c2q)
    float *P2, *P3;       // [B][ntile][Lq] partials of the column-softmax dot / colsum(dS)
    float *P4;            // [B][ntile][Lq][128] partial of dQ(trilinear)
    float *P5;            // [B][ntile][128] partial of colsum(df2)
    float *p_w4C, *p_w4mlu;                              // parameter slabs [B * ntile][128]
    float *p_w4Q, *p_pool, *p_bcat;                      // parameter slabs [B][128]
    float *p_W2;                                         // [B][128][128] for Wcat[:, 128:]
    int T, Lq, b_off, ntile;
    Drop dc, dq;
};
void launch_cq_bwd(const CqBwdArgs& a, int B, hipStream_t s);          // kernels a, b, c: dC final, dQ partials
void launch_cq_bwd_query(const CqBwdArgs& a, int B, hipStream_t s);    // kernel d: dQ + pooled-query parameters
// the backward of that launch (autograd of the same lines) + the Embedding linear's data gradient; per-SAMPLE partial slabs
struct QueryBwdArgs {
    const float* dout;             // (Rq,128) grad wrt the encoder output
    const float *r, *ln2_g;        // output block
    const uint16_t* WoT3;          // split pack (type 7) of the out_layer weight
    float *go, *p_ln2g, *p_ln2b;
    Drop d2, d3, d4, d5;
    const float *q, *k, *v, *att, *lse, *mask;     // attention
    float *dq, *dk, *dv;
    const uint16_t* WqkvT3;        // split pack (type 7) of [Wq; Wk; Wv]: 384 contraction rows
    const float *y3, *ln1_g;
    float *p_ln1g, *p_ln1b;
    Drop d1;
    const uint16_t* WT3[4];        // conv layers
    const float* x[4];
    const uint32_t* relu_mask[4];
    const float *ln_g[4], *ln_b[4], *dw_w[4];
    Drop dp[4];
    float* gz[4];
    float *p_lng[4], *p_lnb[4], *p_dw[4];
    float* dx0;
    const uint16_t* WembT3;        // split pack (type 7) of the Embedding.linear weight, EWc >= EW columns
    float* dE;
    int EW, EWc, L, b_off;
};
void launch_query_bwd(const QueryBwdArgs& a, int B, hipStream_t s);
void launch_embed_bwd(const float* dE, const int64_t* word_ids, const int64_t* char_ids, const float* E,
                      const int8_t* argpos, const float* char_tab, const float* wimg_b /* type-8 pack */,
                      float* p_cw /*[nchunk][300 char_dim]*/,
                      float* p_cb /*[nchunk][100]*/, float* p_tab /*[nchunk][char_size*char_dim]*/,
                      float* p_unk /*[nchunk][word_dim]*/, int Rq, int Lc, int word_dim, int char_dim, int char_size, Drop dw,
                      Drop dc, hipStream_t s);
void launch_word_table_bwd(const float* dE, const int64_t* word_ids, float* gtab /*(word_size, word_dim) inside the gradient bucket*/, int Rq,
                           int word_size, int word_dim, Drop dw, hipStream_t s);
void launch_reduce(const float* partial, float* grads, const ReduceSeg* segs_dev, const int* blk2seg_dev, int nblocks,
                   float* sq /* [nblocks]: sum of squares of each block's results */, hipStream_t s);
// the final reduction + clip + AdamW as ONE launch (vsl_io.fused_step): `grid` = reduce_adamw_grid(nunits, cap) workgroups that wait for each other
// (cap = reduce_adamw_resident(CUs): half of what the device holds at once, at most 4 per CU); 0 = the units do not fit, take the two launches
int reduce_adamw_resident(int cus);
int reduce_adamw_grid(int nunits, int cap);
void launch_reduce_adamw(const float* partial, float* grads, const ReduceSeg* segs_dev, const int* blk2seg_dev, int nunits, float* sq,
                         const float* sq_early, const int* blk2seg_early, int n_early, int grid, unsigned long long* gran /*[grid]*/, unsigned tag,
                         float* params, float* m, float* v, const uint8_t* decay_mask, float lr, float b1, float b2, float eps, float wd,
                         float clip, float bc1, float bc2_sqrt, float* norm_out, int hf_order, hipStream_t s);
// fused optimizer (vsl_adamw_step): sum of squares partials, then clip + AdamW
constexpr int OPT_BLOCKS = 256;
void launch_adamw(float* params, const float* grads, float* m, float* v, const uint8_t* decay_mask, float* partials /*[OPT_BLOCKS + 1]*/,
                  int64_t n, float lr, float b1, float b2, float eps, float wd, float clip, float bc1, float bc2_sqrt,
                  float* norm_out, hipStream_t s, int hf_order = 0,
                  const float* sq_from_backward = nullptr /* k_reduce's partials instead of a k_sqsum pass */, int nsq = 0);
constexpr int EMB_CHUNK_MAX = 8;  // most query words per workgroup in the embedding backward
int embed_bwd_chunk(int Rq, int Lc, int char_dim);      // words per workgroup the launcher uses (the number of partial slabs follows from it)
constexpr int EB_IMG_Q = 76;      // k-steps of the embedding backward's B-operand image ([channel tile][76][64 lanes], PackJob type 8)
constexpr int CHARW_TOTAL = 15000;

}  // namespace vsl
