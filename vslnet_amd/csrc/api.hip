// C ABI of libvslnet_hip.so (declared in include/vslnet_hip.h): parameter layout, per-shape plan (workspace
// layout, weight-pack jobs, partial-slab table for the gradient reduction) and the forward / loss / backward
// launch sequences.  Host code only -- no device code lives here.
#include "../../include/vslnet_hip.h"
#include "launch.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

using namespace vsl;

static thread_local std::string g_err;
static int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}
#define HIP_OK(x)                                                                         \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) return fail("%s failed: %s", #x, hipGetErrorString(e_));    \
    } while (0)

namespace {

struct ParamInfo { std::string name; int64_t off, numel; int ndim; int64_t dims[4]; };

struct EncP { int pos, dw[4], pw[4], pwb[4], lng[4], lnb[4], qw, qb, kw, kb, vw, vb, ln1g, ln1b, ln2g, ln2b, ow, ob; };
struct EncPk { int pw_f[4], pw_t[4], qkv_f, qkv_t, o_f, o_t; int pw_f3[4], pw_t3[4], qkv_f3, qkv_t3, o_f3, o_t3; };   // ..3: split packs
struct ModelP {
    int unk, char_tab, ccw[4], ccb[4], emb_w, emb_b, va_w, va_b;
    EncP fe;
    int w4C, w4Q, w4mlu, cqa_w, cqa_b, pool_w, cat_w, cat_b, hl_w, hl_b;
    EncP pe;
    int sln_g, sln_b, eln_g, eln_b, s0w, s0b, s1w, s1b, e0w, e0b, e1w, e1b;
    int l_wih[2], l_whh[2], l_bih[2], l_bhh[2];     // rnn predictor: start / end DynamicRNN (layers_t7.py:302-313)
};
struct ModelPk { int va_f, va_f16, va_f3, l_t3[2], emb_f, emb_t, emb_f3, emb_t3, emb_t3_cols; EncPk fe, pe; int cqa_f, cqa_t, cat1_f, cat1_t, s0_f, s0_t, e0_f, e0_t, ccw_img, ccw_imgb;
                 int l_f3[2], l_t[2], l_hf[2], l_hb[2], l_if, l_ib, zero128; };

struct EncWs { int64_t x0, y[4], u[4], mask[4], h1, q, k, v, lse, att, r, h2, out; int R, L; };

struct LstmWs { int64_t gi, gates, cseq, tseq, hprev, out, dG, carry; };   // one DynamicRNN: x W_ih^T, activated gates, c_t, tanh(c_t), h_{t-1}, h * mask, gate grads, (B,2,128) dc/dh hand-over between time chunks

struct EncTmp { int64_t dr, dq, dk, dv, go, gz[4], ga; };   // backward temporaries of one encoder application

struct SlabRec { int dst, n, nslabs, ss, rl, ds, vn; int64_t src; };
// the reduction runs in two launches: `early` = parameters whose partials are complete before the final video/query fork
// (predictor, heads, CQ fusion), on a side stream; `late` = shared feature encoder, embedding stack, visual projection

struct Plan {
    int B, T, Lq, Lc;
    int64_t pack, vf, E, argpos, qf;
    EncWs ve, qe, p1, p2;
    LstmWs lstm[2];
    int64_t h_gran = -1, gi_gran = -1, dg_gran = -1, dx_gran = -1;     // granule buffers of the fused rnn head (8 bytes per value); -1: chunked launches
    int64_t gran_floats = 0;                // their total extent (contiguous in the workspace, from h_gran)
    // epoch generation and workspace these buffers were last cleared for (rnn_granules_fresh): the last four workspaces, so that a caller who
    // alternates two or three of them (double buffering) does not pay a memset of the granule range in every forward
    struct GranSeen { const void* ws = nullptr; long long gen = -1; } gran_seen[4];
    int gran_next = 0;
    int64_t S, Srow, Scol, M, alpha, pooled, pb, cat, f1, f2, gated, hid_s, hid_e, lnf_s, lnf_e;
    // backward temporaries
    int64_t loss_scratch, gz_s, gz_e, dfeat_s, dfeat_e, dxh_s, dxh_e, g_s1, g_gated;
    EncTmp tmp[4];                      // one set per encoder application: the applications' backward chains and their
                                        // weight-gradient launches overlap on different streams
    int64_t df2, df1, dC, dSr, dSs, dQtot, dvf, dqf, dE, cqP1, cqP2, cqP3, cqP4, cqP5;
    int64_t partial, partial_floats, total;
    std::vector<int64_t> part_offs;     // sequence of partial-arena allocations made by the backward
    ReduceSeg* segs_dev = nullptr;
    int* blk2seg_dev = nullptr;         // block table: [early blocks | late blocks]
    float* sq_dev = nullptr;            // [nblocks] sum of squares of each reduction block's results (the clip's global norm, vsl_adamw.norm_from_backward)
    bool sq_cover = false;              // the reduction blocks write every element of the gradient bucket exactly once
    int nblocks = 0, nblocks_early = 0;
    uint64_t last_use = 0;              // LRU stamp of the plan cache (get_plan)
};

}  // namespace

struct vsl_handle_s {
    vsl_config cfg;
    std::vector<ParamInfo> params;
    int64_t param_floats = 0;
    ModelP P;
    ModelPk K;
    int64_t pack_floats = 0;
    std::vector<PackJob> jobs;
    PackJob* jobs_dev = nullptr;
    int jobs_char = 0, jobs_first = 0, jobs_query = 0;  // jobs_dev = [char: the char-conv image embed_fwd reads | first: the video branch's opening kernels | query: the Embedding linear | everything else]
    unsigned* loss_counter = nullptr;    // arrival counter of k_loss_fused (zero between calls)
    uint8_t* decay_dev = nullptr;        // per-element weight-decay flag of the flat bucket (vsl_adamw_step)
    float* opt_scratch = nullptr;        // OPT_BLOCKS partial sums of grads^2
    unsigned long long* tail_gran = nullptr;   // vsl_io.fused_step: one {tag, sum of squares} granule per workgroup of the fused tail launch
    int cus = 0;                         // CUs of the device (vsl_create)
    int tail_cap = 0;                    // ... their number = the largest grid that launch may take (4 workgroups per CU)
    unsigned tail_tag = 0;               // ... the tag of the last such launch
    const float* sq_src = nullptr;       // the last vsl_backward's per-block sums of squares (Plan::sq_dev), their count, the bucket they describe
    int sq_n = 0;
    const float* sq_grads = nullptr;
    std::map<std::tuple<int, int, int, int>, Plan*> plans;
    uint64_t plan_clock = 0;
    // side streams for the independent chains (query branch, weight gradients) + fork/join events
    hipStream_t side[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> sync_pool;
    size_t sync_used = 0;
    // stop events riding on the kernel dispatches of the current call (launch.hpp VSL_LAUNCH); last one per stream
    std::vector<hipEvent_t> stop_pool;
    size_t stop_used = 0;
    bool stop_events = false;
    hipStream_t ev_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_last[4] = {nullptr, nullptr, nullptr, nullptr};
    int ev_n = 0;
    bool ev_dirty[4] = {false, false, false, false};     // a wait was enqueued on the stream AFTER its last kernel: that kernel's stop event no
                                                         // longer stands for "everything enqueued so far" (the shortcut is not transitive)
    hipEvent_t last_event(hipStream_t s) const {
        for (int i = 0; i < ev_n; ++i) if (ev_stream[i] == s) return ev_dirty[i] ? nullptr : ev_last[i];
        return nullptr;
    }
    void mark_waiting(hipStream_t s) { for (int i = 0; i < ev_n; ++i) if (ev_stream[i] == s) ev_dirty[i] = true; }
    bool multi_stream = true;
    // optional per-kernel timing with HIP events on the launch stream (vsl_profile_*), used by bench.py's roofline line
    bool prof_on = false;
    std::string prof_sel;
    std::vector<hipEvent_t> prof_pool;
    size_t prof_used = 0;
    struct ProfRec { const char* name; size_t e0, e1; hipStream_t s; double host_us; int deps[6]; int nd; };
    std::vector<ProfRec> prof_recs;
    const char* prof_name = nullptr;     // name of the LAUNCH being enqueued when it is profiled (vsl_launch_events)
    // dependency bookkeeping of the critical-path ledger (vsl_profile_launch): last profiled launch per stream, and the cross-stream
    // ordering points (Ctx::order) enqueued since -- they become `deps` of the consumer stream's next launch
    std::vector<std::pair<hipStream_t, int>> prof_last, prof_pending;
    std::chrono::steady_clock::time_point prof_t0;
    void prof_wait(hipStream_t from, hipStream_t to) {
        if (!prof_on || from == to) return;
        for (auto& kv : prof_last) if (kv.first == from) { prof_pending.push_back({to, kv.second}); return; }
    }
};

namespace {

// ------------------------------------------------------------------------------------------------ parameters
struct ParamBuilder {
    vsl_handle_s* h;
    int add(const std::string& name, std::initializer_list<int64_t> dims) {
        ParamInfo p;
        p.name = name;
        p.ndim = (int)dims.size();
        p.numel = 1;
        int i = 0;
        for (auto d : dims) { p.dims[i++] = d; p.numel *= d; }
        p.off = h->param_floats;
        h->param_floats += (p.numel + 3) & ~int64_t(3);          // keep every tensor 16-byte aligned
        h->params.push_back(p);
        return (int)p.off;
    }
};

void build_encoder_params(ParamBuilder& pb, const std::string& pre, EncP& e, int max_pos) {
    const int64_t d = D;
    e.pos = pb.add(pre + "pos_embedding.position_embeddings.weight", {max_pos, d});
    for (int i = 0; i < 4; ++i) {
        const std::string b = pre + "conv_block.depthwise_separable_conv." + std::to_string(i);
        e.dw[i] = pb.add(b + ".0.weight", {d, 1, 7});
        e.pw[i] = pb.add(b + ".1.weight", {d, d, 1});
        e.pwb[i] = pb.add(b + ".1.bias", {d});
    }
    for (int i = 0; i < 4; ++i) {
        const std::string b = pre + "conv_block.layer_norms." + std::to_string(i);
        e.lng[i] = pb.add(b + ".weight", {d});
        e.lnb[i] = pb.add(b + ".bias", {d});
    }
    const std::string a = pre + "attention_block.";
    e.qw = pb.add(a + "query.conv1d.weight", {d, d, 1});
    e.qb = pb.add(a + "query.conv1d.bias", {d});
    e.kw = pb.add(a + "key.conv1d.weight", {d, d, 1});
    e.kb = pb.add(a + "key.conv1d.bias", {d});
    e.vw = pb.add(a + "value.conv1d.weight", {d, d, 1});
    e.vb = pb.add(a + "value.conv1d.bias", {d});
    e.ln1g = pb.add(a + "layer_norm1.weight", {d});
    e.ln1b = pb.add(a + "layer_norm1.bias", {d});
    e.ln2g = pb.add(a + "layer_norm2.weight", {d});
    e.ln2b = pb.add(a + "layer_norm2.bias", {d});
    e.ow = pb.add(a + "out_layer.conv1d.weight", {d, d, 1});
    e.ob = pb.add(a + "out_layer.conv1d.bias", {d});
}

void build_params(vsl_handle_s* h) {
    // order == the reference's state_dict order restricted to trainable tensors
    const vsl_config& c = h->cfg;
    ParamBuilder pb{h};
    ModelP& P = h->P;
    const int64_t d = D;
    if (c.word_table) P.unk = pb.add("embedding_net.word_emb.word_emb.weight", {c.word_size, c.word_dim});      // rows 0 / 1 stand where pad_vec / unk_vec do
    else P.unk = pb.add("embedding_net.word_emb.unk_vec", {1, c.word_dim});
    P.char_tab = pb.add("embedding_net.char_emb.char_emb.weight", {c.char_size, c.char_dim});
    const int ch[4] = {10, 20, 30, 40};
    for (int i = 0; i < 4; ++i) {
        const std::string b = "embedding_net.char_emb.char_convs." + std::to_string(i) + ".0";
        P.ccw[i] = pb.add(b + ".weight", {ch[i], c.char_dim, 1, i + 1});
        P.ccb[i] = pb.add(b + ".bias", {ch[i]});
    }
    P.emb_w = pb.add("embedding_net.linear.conv1d.weight", {d, c.word_dim + 100, 1});
    P.emb_b = pb.add("embedding_net.linear.conv1d.bias", {d});
    P.va_w = pb.add("video_affine.linear.conv1d.weight", {d, c.video_feature_dim, 1});
    P.va_b = pb.add("video_affine.linear.conv1d.bias", {d});
    build_encoder_params(pb, "feature_encoder.", P.fe, c.max_pos_len);
    P.w4C = pb.add("cq_attention.w4C", {d, 1});
    P.w4Q = pb.add("cq_attention.w4Q", {d, 1});
    P.w4mlu = pb.add("cq_attention.w4mlu", {1, 1, d});
    P.cqa_w = pb.add("cq_attention.cqa_linear.conv1d.weight", {d, 4 * d, 1});
    P.cqa_b = pb.add("cq_attention.cqa_linear.conv1d.bias", {d});
    P.pool_w = pb.add("cq_concat.weighted_pool.weight", {d, 1});
    P.cat_w = pb.add("cq_concat.conv1d.conv1d.weight", {d, 2 * d, 1});
    P.cat_b = pb.add("cq_concat.conv1d.conv1d.bias", {d});
    P.hl_w = pb.add("highlight_layer.conv1d.conv1d.weight", {1, d, 1});
    P.hl_b = pb.add("highlight_layer.conv1d.conv1d.bias", {1});
    if (c.predictor == 0) {          // rnn head (layers_t7.py:319-321): two nn.LSTM(d, d), gate order i,f,g,o
        const char* nm[2] = {"predictor.start_encoder.lstm.", "predictor.end_encoder.lstm."};
        for (int l = 0; l < 2; ++l) {
            const std::string a = nm[l];
            P.l_wih[l] = pb.add(a + "weight_ih_l0", {4 * d, d});
            P.l_whh[l] = pb.add(a + "weight_hh_l0", {4 * d, d});
            P.l_bih[l] = pb.add(a + "bias_ih_l0", {4 * d});
            P.l_bhh[l] = pb.add(a + "bias_hh_l0", {4 * d});
        }
        P.sln_g = P.sln_b = P.eln_g = P.eln_b = -1;
    } else {
        build_encoder_params(pb, "predictor.encoder.", P.pe, c.max_pos_len);
        P.sln_g = pb.add("predictor.start_layer_norm.weight", {d});
        P.sln_b = pb.add("predictor.start_layer_norm.bias", {d});
        P.eln_g = pb.add("predictor.end_layer_norm.weight", {d});
        P.eln_b = pb.add("predictor.end_layer_norm.bias", {d});
    }
    P.s0w = pb.add("predictor.start_block.0.conv1d.weight", {d, 2 * d, 1});
    P.s0b = pb.add("predictor.start_block.0.conv1d.bias", {d});
    P.s1w = pb.add("predictor.start_block.2.conv1d.weight", {1, d, 1});
    P.s1b = pb.add("predictor.start_block.2.conv1d.bias", {1});
    P.e0w = pb.add("predictor.end_block.0.conv1d.weight", {d, 2 * d, 1});
    P.e0b = pb.add("predictor.end_block.0.conv1d.bias", {d});
    P.e1w = pb.add("predictor.end_block.2.conv1d.weight", {1, d, 1});
    P.e1b = pb.add("predictor.end_block.2.conv1d.bias", {1});
}

// ------------------------------------------------------------------------------------------------ weight packs
struct PackBuilder {
    vsl_handle_s* h;
    // forward pack of W (N x K, leading dim ld): Bm[k][c] = W[c][k]
    int fwd(int src, int N, int Kd, int ld) {
        const int dst = (int)h->pack_floats;
        h->pack_floats += (int64_t)pack_size(Kd, N);
        h->jobs.push_back(PackJob{src, dst, Kd, N, ld, 0, N, 0, 0});
        return dst;
    }
    // split packs (three bf16 planes, common.hpp pack3_index) of the same two operands
    // (kpad: the contraction extent is zero-padded to a multiple of it -- a K-chunked kernel then needs no tail guards)
    int fwd3(int src, int N, int Kd, int ld, int kpad = 16) {
        const int dst = (int)h->pack_floats, Kp = (Kd + kpad - 1) / kpad * kpad;
        h->pack_floats += (int64_t)((pack3_floats(Kp, N) + 3) & ~size_t(3));
        h->jobs.push_back(PackJob{src, dst, Kd, N, ld, 6, N, 0, 0, Kp, Kp});
        return dst;
    }
    int tr3(int src, int N, int Kd, int ld, int kpad = 16) {
        const int dst = (int)h->pack_floats, Np = (N + kpad - 1) / kpad * kpad;
        h->pack_floats += (int64_t)((pack3_floats(Np, Kd) + 3) & ~size_t(3));
        h->jobs.push_back(PackJob{src, dst, N, Kd, ld, 7, Kd, 0, 0, Np, Np});
        return dst;
    }
    // transpose pack of W (N x K): Bm[k = n][c] = W[n][c], ncols = K
    int tr(int src, int N, int Kd, int ld) {
        const int dst = (int)h->pack_floats;
        h->pack_floats += (int64_t)pack_size(N, Kd);
        h->jobs.push_back(PackJob{src, dst, N, Kd, ld, 1, Kd, 0, 0});
        return dst;
    }
};
void build_encoder_packs(PackBuilder& pk, vsl_handle_s* h, const EncP& e, EncPk& k) {
    for (int i = 0; i < 4; ++i) { k.pw_f[i] = pk.fwd(e.pw[i], D, D, D); k.pw_t[i] = pk.tr(e.pw[i], D, D, D); }
    // fused QKV operand: forward pack has 384 columns [q | k | v]; transpose pack has 384 contraction rows
    k.qkv_f = (int)h->pack_floats;
    h->pack_floats += (int64_t)pack_size(D, 3 * D);
    const int qkv[3] = {e.qw, e.kw, e.vw};
    for (int i = 0; i < 3; ++i) h->jobs.push_back(PackJob{qkv[i], k.qkv_f, D, D, D, 0, 3 * D, 0, i * D});
    k.qkv_t = (int)h->pack_floats;
    h->pack_floats += (int64_t)pack_size(3 * D, D);
    for (int i = 0; i < 3; ++i) h->jobs.push_back(PackJob{qkv[i], k.qkv_t, D, D, D, 1, D, i * D, 0});
    k.o_f = pk.fwd(e.ow, D, D, D);
    k.o_t = pk.tr(e.ow, D, D, D);
    // split packs (fp32 grade on the bf16 matrix cores)
    for (int i = 0; i < 4; ++i) { k.pw_f3[i] = pk.fwd3(e.pw[i], D, D, D); k.pw_t3[i] = pk.tr3(e.pw[i], D, D, D); }
    k.qkv_f3 = (int)h->pack_floats;
    h->pack_floats += (int64_t)((pack3_floats(D, 3 * D) + 3) & ~size_t(3));
    for (int i = 0; i < 3; ++i) h->jobs.push_back(PackJob{qkv[i], k.qkv_f3, D, D, D, 6, 3 * D, 0, i * D, D, D});
    k.qkv_t3 = (int)h->pack_floats;
    h->pack_floats += (int64_t)((pack3_floats(3 * D, D) + 3) & ~size_t(3));
    for (int i = 0; i < 3; ++i) h->jobs.push_back(PackJob{qkv[i], k.qkv_t3, D, D, D, 7, D, i * D, 0, 3 * D, D});
    k.o_f3 = pk.fwd3(e.ow, D, D, D);
    k.o_t3 = pk.tr3(e.ow, D, D, D);
}
void build_packs(vsl_handle_s* h) {
    const vsl_config& c = h->cfg;
    PackBuilder pk{h};
    ModelPk& K = h->K;
    const ModelP& P = h->P;
    K.va_f = pk.fwd(P.va_w, D, c.video_feature_dim, c.video_feature_dim);
    // bf16 throughput mode: the same weight rounded to bfloat16 in the operand layout of v_mfma_f32_32x32x16_bf16
    K.va_f16 = (int)h->pack_floats;
    h->pack_floats += (int64_t)((c.video_feature_dim + 15) / 16) * D * 16 / 2;
    h->jobs.push_back(PackJob{P.va_w, K.va_f16, c.video_feature_dim, D, c.video_feature_dim, 5, D, 0, 0});
    K.va_f3 = pk.fwd3(P.va_w, D, c.video_feature_dim, c.video_feature_dim, 128);      // split pack: fp32 grade on the bf16 matrix cores
    K.emb_f = pk.fwd(P.emb_w, D, c.word_dim + 100, c.word_dim + 100);
    K.emb_t = pk.tr(P.emb_w, D, c.word_dim + 100, c.word_dim + 100);
    K.emb_f3 = pk.fwd3(P.emb_w, D, c.word_dim + 100, c.word_dim + 100);
    {   // data-gradient operand with its column count padded to whole 128-column tiles (the pad columns are never stored)
        const int EWc = (c.word_dim + 100 + 127) / 128 * 128;
        K.emb_t3_cols = EWc;
        K.emb_t3 = (int)h->pack_floats;
        h->pack_floats += (int64_t)((pack3_floats(D, EWc) + 3) & ~size_t(3));
        h->jobs.push_back(PackJob{P.emb_w, K.emb_t3, D, c.word_dim + 100, c.word_dim + 100, 7, EWc, 0, 0, D, D});
    }
    build_encoder_packs(pk, h, P.fe, K.fe);
    if (c.predictor == 0) {
        for (int l = 0; l < 2; ++l) {
            K.l_f3[l] = pk.fwd3(P.l_wih[l], 4 * D, D, D);    // gi = x W_ih^T  : (R,128) x (128,512), split pack
            K.l_t[l] = pk.tr(P.l_wih[l], 4 * D, D, D);       // dx = dG W_ih   : (R,512) x (512,128)
            K.l_t3[l] = pk.tr3(P.l_wih[l], 4 * D, D, D, 128);
            for (int bw = 0; bw < 2; ++bw) {                 // W_hh in the register order of k_lstm1_fwd / k_lstm1_bwd
                (bw ? K.l_hb : K.l_hf)[l] = (int)h->pack_floats;
                h->jobs.push_back(PackJob{P.l_whh[l], (int)h->pack_floats, LSTM_IMG_FLOATS, 1, 0, 9 + bw, 1, 0, 0});
                h->pack_floats += LSTM_IMG_FLOATS;
            }
            if (l == 1) for (int bw = 0; bw < 2; ++bw) {     // the end LSTM's W_ih likewise: the projection workgroups of k_rnn_fwd / k_rnn_bwd
                (bw ? K.l_ib : K.l_if) = (int)h->pack_floats;
                h->jobs.push_back(PackJob{P.l_wih[1], (int)h->pack_floats, LSTM_IMG_FLOATS, 1, 0, 9 + bw, 1, 0, 0});
                h->pack_floats += LSTM_IMG_FLOATS;
            }
        }
        K.zero128 = (int)h->pack_floats;                    // a zero bias vector for the bias-less GEMM above
        h->pack_floats += D;
        h->jobs.push_back(PackJob{0, K.zero128, D, 1, 0, 4, 1, 0, 0});
    } else {
        build_encoder_packs(pk, h, P.pe, K.pe);
    }
    K.cqa_f = pk.fwd(P.cqa_w, D, 4 * D, 4 * D);
    K.cqa_t = pk.tr(P.cqa_w, D, 4 * D, 4 * D);
    K.cat1_f = pk.fwd(P.cat_w, D, D, 2 * D);          // first half of the (128, 256) CQConcatenate weight
    K.cat1_t = pk.tr(P.cat_w, D, D, 2 * D);
    K.s0_f = pk.fwd(P.s0w, D, 2 * D, 2 * D);
    K.s0_t = pk.tr(P.s0w, D, 2 * D, 2 * D);
    K.e0_f = pk.fwd(P.e0w, D, 2 * D, 2 * D);
    K.e0_t = pk.tr(P.e0w, D, 2 * D, 2 * D);
    // char-conv weights as one [ci][4 taps][100 channels] image; each job writes all 4 tap slots of its channels (zero
    // beyond the kernel width), so the image is fully defined every step
    K.ccw_img = (int)h->pack_floats;
    h->pack_floats += (int64_t)c.char_dim * 400;
    const int chn[4] = {10, 20, 30, 40};
    int oc0 = 0;
    for (int i = 0; i < 4; ++i) {
        PackJob j{P.ccw[i], K.ccw_img, chn[i] * c.char_dim * 4, 1, i + 1, 3, c.char_dim, 0, oc0};
        h->jobs.push_back(j);
        oc0 += chn[i];
    }
    // the same weights as the B operands of the embedding backward's dCe product, in lane order (kernels_fwd.hip, type 8)
    K.ccw_imgb = (int)h->pack_floats;
    const int ebn = ((c.char_dim + 15) / 16) * EB_IMG_Q * 64;
    h->pack_floats += ebn;
    oc0 = 0;
    for (int i = 0; i < 4; ++i) {
        PackJob j{P.ccw[i], K.ccw_imgb, ebn, 1, i + 1, 8, c.char_dim, chn[i], oc0};
        h->jobs.push_back(j);
        oc0 += chn[i];
    }
}

// ------------------------------------------------------------------------------------------------ plan
struct Bump {
    int64_t cur = 0;
    int64_t operator()(int64_t n) { const int64_t o = cur; cur += (n + 3) & ~int64_t(3); return o; }
};
void plan_encoder(Bump& al, EncWs& w, int Bn, int L, int H) {
    const int64_t R = (int64_t)Bn * L;
    w.R = (int)R; w.L = L;
    w.x0 = al(R * D);
    for (int i = 0; i < 4; ++i) { w.y[i] = al(R * D); w.u[i] = al(R * D); w.mask[i] = al(R * 4); }
    w.h1 = al(R * D); w.q = al(R * D); w.k = al(R * D); w.v = al(R * D);
    w.lse = al((int64_t)Bn * H * L);
    w.att = al(R * D); w.r = al(R * D); w.h2 = al(R * D); w.out = al(R * D);
}

// the handle whose call is enqueuing kernels on this thread (set for the duration of vsl_forward / vsl_backward)
static thread_local vsl_handle_s* g_cur = nullptr;
struct CallScope {
    explicit CallScope(vsl_handle_s* h) {
        h->sync_used = 0; h->stop_used = 0; h->ev_n = 0;
        for (bool& d : h->ev_dirty) d = false;
        h->stop_events = h->multi_stream;
        g_cur = h;
    }
    ~CallScope() { g_cur = nullptr; }
};

// vsl_loss / vsl_adamw_step: plain launches on the caller's stream; when the built-in profiler selects `name` they carry timing events too
struct ProfScope {
    vsl_handle_s* h;
    ProfScope(vsl_handle_s* h_, const char* name) : h(h_) {
        bool hit = false;
        if (h->prof_on) {
            const std::string& sel = h->prof_sel;
            const size_t n = strlen(name);
            for (size_t p = 0; !hit && p <= sel.size();) {
                size_t q = sel.find(',', p);
                if (q == std::string::npos) q = sel.size();
                hit = (q - p == 1 && sel[p] == '*') || (q - p == n && sel.compare(p, n, name) == 0);
                p = q + 1;
            }
        }
        if (hit) { h->stop_events = false; h->prof_name = name; g_cur = h; }
    }
    ~ProfScope() { if (g_cur == h && h->prof_name) { h->prof_name = nullptr; g_cur = nullptr; } }
};

struct Ctx {
    vsl_handle_s* h;
    Plan* p;
    const vsl_io* io;
    hipStream_t s;
    bool dry;                           // plan-building pass: record partial-slab allocations, launch nothing
    float* ws;
    size_t part_idx = 0;
    int64_t part_cur = 0;
    std::vector<SlabRec>* recs = nullptr;
    std::vector<WgradBatch> pend;       // weight-gradient batches waiting for ONE ordering point (wgrad_async / wgrad_flush)
    const vsl_loss_io* pend_loss = nullptr;   // vsl_io.fused_loss off the dependent chain: launched on the weight-gradient stream behind its first ordering point (loss_on_side)
    int pw_rows = 0;                    // rows per chunk of the NEXT enc_bwd's pointwise weight-gradient jobs (0 = WG_ROWS): the step's last batch, run_backward
    bool defer_w = false;
    const float* P(int off) const { return io->params + off; }
    const float* PK(int off) const { return ws + p->pack + off; }
    float* W(int64_t off) const { return ws + off; }
    // raw allocation in the partial arena
    int64_t part_alloc(int64_t n) {
        n = (n + 3) & ~int64_t(3);
        if (dry) { p->part_offs.push_back(part_cur); part_cur += n; return p->part_offs.back(); }
        return p->part_offs[part_idx++];
    }
    // register a reduction source: `src` is an ABSOLUTE workspace offset (partial arena or any saved buffer)
    void reg(int dst, int n, int64_t src, int nslabs, int ss, int rl = 0, int ds = 0, int vn = 0) {
        if (dry) recs->push_back(SlabRec{dst, n, nslabs, ss, rl ? rl : n, ds, vn ? vn : n, src});
    }
    // the common case: `nslabs` contiguous slabs of n floats for the parameter at dst
    float* slab(int dst, int n, int nslabs) {
        const int64_t o = part_alloc((int64_t)n * nslabs);
        reg(dst, n, p->partial + o, nslabs, n);
        return dry ? nullptr : ws + p->partial + o;
    }
    float* part_ptr(int64_t o) const { return dry ? nullptr : ws + p->partial + o; }
    // make stream `to` wait for everything enqueued so far on stream `from`
    void order(hipStream_t from, hipStream_t to) {
        if (dry || from == to) return;
        h->prof_wait(from, to);
        // rides on from's last kernel -- unless `from` itself was made to wait for something after that kernel (then a marker is recorded)
        if (hipEvent_t le = h->last_event(from)) { (void)hipStreamWaitEvent(to, le, 0); h->mark_waiting(to); return; }
        if (h->sync_used == h->sync_pool.size()) { hipEvent_t e; (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); h->sync_pool.push_back(e); }
        hipEvent_t e = h->sync_pool[h->sync_used++];
        (void)hipEventRecord(e, from);
        (void)hipStreamWaitEvent(to, e, 0);
        h->mark_waiting(to);
    }
    // one event record on `from`, two waiters
    void order2(hipStream_t from, hipStream_t to1, hipStream_t to2) {
        if (dry) return;
        h->prof_wait(from, to1);
        if (to2 != to1) h->prof_wait(from, to2);
        if (hipEvent_t le = h->last_event(from)) {
            if (to1 != from) { (void)hipStreamWaitEvent(to1, le, 0); h->mark_waiting(to1); }
            if (to2 != from && to2 != to1) { (void)hipStreamWaitEvent(to2, le, 0); h->mark_waiting(to2); }
            return;
        }
        if (h->sync_used == h->sync_pool.size()) { hipEvent_t e; (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); h->sync_pool.push_back(e); }
        hipEvent_t e = h->sync_pool[h->sync_used++];
        (void)hipEventRecord(e, from);
        if (to1 != from) { (void)hipStreamWaitEvent(to1, e, 0); h->mark_waiting(to1); }
        if (to2 != from && to2 != to1) { (void)hipStreamWaitEvent(to2, e, 0); h->mark_waiting(to2); }
    }
    hipStream_t side(int k) const { return (h->multi_stream && h->side[k]) ? h->side[k] : main; }
    hipStream_t main = nullptr;
    // built-in profiler: a selected LAUNCH hands its name to vsl_launch_events, which puts timing events on the kernel's own packet
    void pb(const char* name) {          // prof_sel: "*", one launch name, or a comma-separated list of names
        bool hit = false;
        if (h->prof_on) {
            const std::string& sel = h->prof_sel;
            const size_t n = strlen(name);
            for (size_t p = 0; !hit && p <= sel.size();) {
                size_t q = sel.find(',', p);
                if (q == std::string::npos) q = sel.size();
                hit = (q - p == 1 && sel[p] == '*') || (q - p == n && sel.compare(p, n, name) == 0);
                p = q + 1;
            }
        }
        h->prof_name = hit ? name : nullptr;
    }
    void pe(const char*) { h->prof_name = nullptr; }
    // elements one sample owns in the tensor dropped at `site` (the masks are keyed by the row-major element index)
    uint32_t site_elems(int site) const {
        const vsl_config& cf = h->cfg;
        const uint32_t T = p->T, Lq = p->Lq;
        if (site == 64) return T * cf.video_feature_dim;              // SITE_VIS
        if (site == 65) return Lq * cf.word_dim;                      // SITE_WORD
        if (site == 66) return Lq * p->Lc * cf.char_dim;              // SITE_CHAR
        if (site == 67) return T * D;                                 // SITE_CQ_C
        if (site == 68) return Lq * D;                                // SITE_CQ_Q
        const uint32_t L = (site >> 4) == 1 ? Lq : T;                 // encoder pass 1 = query
        if ((site & 15) != 5) return L * D;
        // 5 = attention probabilities (B, H, L, L); the kernels of L > 256 key their masks by key PAIRS (common.hpp drop_hash_odd): ceil(L / 2) hashes per row
        return L > 256 ? cf.num_heads * L * ((L + 1) / 2) : cf.num_heads * L * L;
    }
    Drop drop(int site) const {
        Drop d{0u, 0u, 1.0f, 0u};
        const float pr = h->cfg.drop_rate;
        if (io && io->training && pr > 0.f) {
            uint32_t x = (uint32_t)io->seed ^ ((uint32_t)(io->seed >> 32) * 0x9E3779B1u) ^ ((uint32_t)site * 0x85EBCA77u + 0x165667B1u);
            x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
            uint32_t k = x ^ 0x68E31DA4u;               // the site key: a second, independent hash of (step seed, site)
            k ^= k >> 15; k *= 0x2C1B3C6Du; k ^= k >> 12; k *= 0x297A2D39u; k ^= k >> 15;
            d.key = k;
            // keep = drop_hash(element, seed, key), element enters as element * 0x9E3779B1 + seed: a shard that starts at sample s of the global batch continues the
            // element counter at s * site_elems, which folds into the seed (no cost in the kernels)
            x += (uint32_t)io->sample_offset * site_elems(site) * 0x9E3779B1u;
            d.seed = x;
            d.thresh = (uint32_t)std::min(4294967295.0, (double)pr * 4294967296.0);
            d.scale = 1.0f / (1.0f - pr);
        }
        return d;
    }
};
#define LAUNCH(name, stmt) do { if (!c.dry) { c.pb(name); stmt; c.pe(name); } } while (0)

// Every cross-stream ordering point (event record on the producer + wait on the consumer) leaves a ~6-7 us bubble in the
// producer stream (rocprofv3 timeline), so weight-gradient batches -- nothing waits for them before the final reduction --
// are collected while `defer_w` is set and go to the side stream behind ONE event.
// the loss launch of vsl_io.fused_loss on the current stream (run_backward decides where: in front of the heads' backward when they read the seeds
// from memory, at the end of the main stream's work when they compute them)
void loss_on_side(Ctx& c) {
    const vsl_loss_io* fl = c.pend_loss;
    if (!fl || c.dry) return;
    c.pend_loss = nullptr;
    const vsl_io* io = c.io;
    LAUNCH("loss", launch_loss(io->start_logits, io->end_logits, io->h_score, fl->start_labels, fl->end_labels, fl->h_labels, io->v_mask, io->B, io->T,
                               fl->inv_batch, fl->mask_sum, fl->w_loc, fl->w_highlight, c.W(c.p->loss_scratch), fl->losses, fl->d_start_logits,
                               fl->d_end_logits, fl->d_h_score, c.s, c.h->loss_counter));
}
void wgrad_async(Ctx& c, hipStream_t sw, const WgradBatch& wb) {
    if (c.defer_w) { c.pend.push_back(wb); return; }
    hipStream_t keep = c.s;
    c.order(keep, sw);
    c.s = sw;
    LAUNCH("wgrad", launch_wgrad(wb, c.s));
    c.s = keep;
}
void wgrad_flush(Ctx& c, hipStream_t sw) {
    if (c.pend.empty()) return;
    hipStream_t keep = c.s;
    c.order(keep, sw);
    c.s = sw;
    for (const WgradBatch& wb : c.pend) LAUNCH("wgrad", launch_wgrad(wb, c.s));
    c.s = keep;
    c.pend.clear();
}
// after a fork the chain that finishes LAST should own the main stream: its join wait is then already satisfied (a wait on
// an event that has just been signalled costs 10 - 22 us).  At the headline shape the query side is the long one.
bool query_chain_is_longer(const Plan& p, bool forward) {
    // Forward: the video branch (VisualProjection 19 + conv block 35 + attention 24 us) and the query branch (embedding 30 + linear 17 +
    // conv block 25 + attention 11) are about even at the headline shape; the main stream keeps the video branch (swap: +0.6 %).
    // Backward: since the split-bf16 kernels of round 3 the QUERY side ends last (cq_bwd_d .. embed_bwd: 147 us of isolated kernels against
    // 133 for the video side), so it owns the main stream and the final join finds the video stream's event long signalled: 0.934-0.938 ->
    // 0.930 ms with optimizer (profiles/r03_notes.md).
    return !forward && p.T <= 128;          // longer videos: the video side's attention backward grows with T^2 and ends last again
}
// (Round 6 measured, same box, and dropped -- profiles/r06_notes.md: one wait instead of two on the joining stream by folding the third stream into
// the second, +0.5 %; the video chain on the main stream in the backward, +1 - 5 %; k_query_fwd held back behind the video pass' conv block, +2.5 %;
// k_query_bwd held back behind the video pass' attention backward, +0.3 %; the video pass' pointwise weight gradients on the weight-gradient stream, +3.5 %.)

// dropout site ids: encoder application `app` (0 video, 1 query, 2 predictor pass 1, 3 predictor pass 2) uses
// app * 16 + {0..3 conv layers, 4 LN1 out, 5 attention probs, 6 attention out, 7 LN2 out, 8 out_layer}
enum { SITE_VIS = 64, SITE_WORD = 65, SITE_CHAR = 66, SITE_CQ_C = 67, SITE_CQ_Q = 68 };

// ------------------------------------------------------------------------------------------------ forward
struct HeadTail { HeadArgs hs, he; const float *x, *vmask; };     // AttnBlockArgs::head_tail
void enc_fwd(Ctx& c, const EncP& P, const EncPk& K, const EncWs& w, const float* xin, const float* mask, int Bn, int app, const HeadTail* ht = nullptr) {
    const int R = w.R, L = w.L, H = c.h->cfg.num_heads;
    {
        // the whole conv block + LN1 / QKV in ONE launch (kernels_enc.hip: 12-row recomputed halo)
        CbFwdArgs a;
        memset(&a, 0, sizeof a);
        if (!c.dry) {
            a.xin = xin; a.pos = c.P(P.pos); a.x0_out = c.W(w.x0);
            for (int i = 0; i < 4; ++i) {
                a.ln_g[i] = c.P(P.lng[i]); a.ln_b[i] = c.P(P.lnb[i]); a.dw_w[i] = c.P(P.dw[i]);
                a.pw_b[i] = c.P(P.pwb[i]); a.y[i] = c.W(w.y[i]); a.u[i] = c.W(w.u[i]);
                a.relu_mask[i] = reinterpret_cast<uint32_t*>(c.W(w.mask[i])); a.dp[i] = c.drop(app * 16 + i);
            }
            a.qf = QkvFuse{c.P(P.ln1g), c.P(P.ln1b), c.P(P.qb), c.P(P.kb), c.P(P.vb), c.W(w.h1), c.W(w.q), c.W(w.k), c.W(w.v),
                           c.drop(app * 16 + 4)};
            for (int i = 0; i < 4; ++i) a.W3[i] = reinterpret_cast<const uint16_t*>(c.PK(K.pw_f3[i]));
            a.Wqkv3 = reinterpret_cast<const uint16_t*>(c.PK(K.qkv_f3));
            a.R = R; a.L = L;
        }
        LAUNCH("convblock_fwd", launch_convblock_fwd(a, c.s));
    }
    if (H == 8 && L <= 128) {      // longer sequences: K / V staged in LDS per 64 queries (k_attn_fwd + k_attn_out_fwd)
        AttnBlockArgs ab;
        memset(&ab, 0, sizeof ab);
        if (!c.dry) {
            ab = AttnBlockArgs{c.W(w.q), c.W(w.k), c.W(w.v), mask, c.W(w.y[3]), c.P(P.ln2g), c.P(P.ln2b), c.PK(K.o_f), c.P(P.ob),
                               c.W(w.att), c.W(w.lse), c.W(w.r), c.W(w.h2), c.W(w.out), L, 0,
                               c.drop(app * 16 + 5), c.drop(app * 16 + 6), c.drop(app * 16 + 7), c.drop(app * 16 + 8)};
            if (ht) { ab.head_tail = 1; ab.hs = ht->hs; ab.he = ht->he; ab.head_x = ht->x; ab.head_vmask = ht->vmask; }
        }
        LAUNCH("attn_block_fwd", launch_attn_block_fwd(ab, Bn, c.s));
        return;
    }
    LAUNCH("attn_fwd", launch_attn_fwd(c.W(w.q), c.W(w.k), c.W(w.v), mask, c.W(w.att), c.W(w.lse), Bn, L, H, 0, c.drop(app * 16 + 5), c.s));
    LAUNCH("attn_out_fwd", launch_attn_out_fwd(c.W(w.att), c.W(w.y[3]), c.P(P.ln2g), c.P(P.ln2b), c.PK(K.o_f), c.P(P.ob), c.W(w.r), c.W(w.h2),
                        c.W(w.out), R, c.drop(app * 16 + 6), c.drop(app * 16 + 7), c.drop(app * 16 + 8), c.s));
}

CharConvPtrs char_ptrs(const Ctx& c) {
    CharConvPtrs cc;
    for (int i = 0; i < 4; ++i) { cc.w[i] = c.P(c.h->P.ccw[i]); cc.b[i] = c.P(c.h->P.ccb[i]); }
    return cc;
}

// Time chunks of the rnn head's CHUNKED launches (batches the one-launch pipeline of kernels_lstm.hip does not take: 80 < B <= 256 and the
// 4-sample groups beyond; lengths in processing order): n equal chunks.  The second LSTM lags the first by one chunk and every chunk boundary is a
// cross-stream hop (~10-15 us on the chain: a stream that is already blocked on an event wakes up late), so n ~ sqrt(0.06 T): 3 chunks at
// T = 128, 4 at T = 256.  With the round-4 step (0.6 us) the choice hardly matters any more -- ms per step at B = 16, T = 128 for 1 / 2 / 3 / 4 / 6
// chunks: 0.823, 0.816, 0.817, 0.831, 0.869 (profiles/r04_rnn_chunk_sweep.txt): the hops eat what the overlap gains, which is why small batches
// go through k_rnn_fwd / k_rnn_bwd instead.  Measured and dropped earlier (profiles/r04_notes.md section 7): a shorter last chunk, the start
// LSTM's own input projection chunked, the start LSTM's dx GEMM chunk by chunk on the third stream.
std::vector<int> lstm_chunks(int T) {
    const int n = std::max(1, (int)std::lround(std::sqrt(0.06 * T)));
    std::vector<int> out;
    for (int i = 0; i < n; ++i) out.push_back((T * (i + 1)) / n - (T * i) / n);
    return out;
}

// Tag of one fused rnn launch's granules: unique per process and a NaN pattern (quiet NaN with a payload no arithmetic produces), so that
// nothing the caller's workspace may hold from earlier use -- activations, indices, older granules -- reads as a valid tag.
// The counter has 21 bits: after 2^21 fused launches (two per training step) the tags repeat.  `gen` = how often it has wrapped; a plan's granule
// buffers are cleared once per (workspace, generation) before a launch (rnn_granules_fresh), so a tag of generation g - 1 can never be read as
// one of generation g.  What remains an assumption: inside one generation the caller does not write its own data into the workspace range the
// plan uses for granules (include/vslnet_hip.h: the workspace belongs to the library between vsl_forward and vsl_backward of a step; across
// steps only its granule ranges must be left alone or be re-zeroed by a plan change).
static std::atomic<unsigned long long> g_rnn_launches{0};
unsigned rnn_epoch(long long* gen = nullptr) {
    // every tag carries the NaN pattern 0x7FE.....: counter value 0 gives 0x7FE00000, which is neither a cleared word (0) nor the tag of
    // the launch before or after it -- no remap (mapping it onto ...01 made it collide with the next launch's tag)
    const unsigned long long n = g_rnn_launches.fetch_add(1) + 1;
    if (gen) *gen = (long long)(n >> 21);
    return 0x7FE00000u | (unsigned)(n & 0x1FFFFFu);
}

// first use of a plan's granule buffers in this workspace, or the epoch counter has wrapped since: one memset on the launch stream
static void rnn_granules_fresh(Ctx& c, Plan& p, long long gen) {
    for (const Plan::GranSeen& g : p.gran_seen)
        if (g.ws == (const void*)c.ws && g.gen == gen) return;
    (void)hipMemsetAsync(c.W(p.h_gran), 0, (size_t)p.gran_floats * sizeof(float), c.s);
    p.gran_seen[p.gran_next] = Plan::GranSeen{c.ws, gen};
    p.gran_next = (p.gran_next + 1) & 3;
}

// Lq <= 32: the query branch runs as sample-local launches (kernels_query.hip) in both directions
// VSL_QUERY_FUSED: 0 = row-tile launches, 1 = both directions sample-local, 2 = forward only, 3 = backward only (A/B switch)
static int query_fused_mode() {
    static const int m = getenv("VSL_QUERY_FUSED") ? atoi(getenv("VSL_QUERY_FUSED")) : 1;
    return m;
}
static bool query_fused(const Ctx& c, bool forward = true) {
    const int m = query_fused_mode();
    const bool on = m == 1 || (forward ? m == 2 : m == 3);
    return on && query_fused_ok(c.p->Lq, c.h->cfg.num_heads, c.h->cfg.word_dim + 100);
}

void run_forward(Ctx& c) {
    const vsl_config& cf = c.h->cfg;
    const ModelP& P = c.h->P;
    const ModelPk& K = c.h->K;
    const Plan& p = *c.p;
    const vsl_io& io = *c.io;
    const int B = p.B, T = p.T, Lq = p.Lq, R = B * T, Rq = B * Lq;
    const int nj = (int)c.h->jobs.size(), njc = c.h->jobs_char, nj0 = c.h->jobs_first, nj1 = c.h->jobs_query;
    hipStream_t sq = c.side(0), sp = c.side(1);
    const bool split3 = sq != c.main && sp != c.main;
    hipEvent_t pack_first_ev = nullptr, pack_q_ev = nullptr;
    int pack_first_rec = -1, pack_q_rec = -1;
    if (split3) {
        // The main stream's pack carries the char-conv image as well, so the query stream's first kernel -- the word + char embedding, 160
        // workgroups x 125 KB of LDS -- starts right behind it and runs beside the remaining packs and VisualProjection, not beside the video
        // pass' conv block (which it kept off 160 CUs: 49 us in the step against 25 alone, profiles/r05_notes.md section 9); the Embedding
        // linear's operand is packed on the third stream meanwhile.
        LAUNCH("pack", launch_pack(io.params, c.W(p.pack), c.h->jobs_dev, njc + nj0, c.s));
        c.order2(c.main, sq, sp);
        pack_first_ev = nullptr;           // (the side streams already wait for this launch)
        pack_first_rec = c.h->prof_on ? (int)c.h->prof_recs.size() - 1 : -1;
        c.s = sp;
        LAUNCH("pack", launch_pack(io.params, c.W(p.pack), c.h->jobs_dev + njc + nj0, nj1, c.s));      // the Embedding linear's operand: the query stream waits for it behind embed_fwd
        pack_q_ev = c.h->last_event(sp);
        pack_q_rec = c.h->prof_on ? (int)c.h->prof_recs.size() - 1 : -1;
        LAUNCH("pack", launch_pack(io.params, c.W(p.pack), c.h->jobs_dev + njc + nj0 + nj1, nj - njc - nj0 - nj1, c.s));
        c.s = c.main;
    } else {
        LAUNCH("pack", launch_pack(io.params, c.W(p.pack), c.h->jobs_dev, nj, c.s));
    }
    // fork: the query branch (embedding + query encoder pass) runs beside the video branch
    if (!split3) c.order(c.main, sq);
    const bool qlong = query_chain_is_longer(p, true);  // the longer branch keeps the main stream (join wait already satisfied)
    c.s = qlong ? sq : c.main;
    if (io.video_features_bf16)      // bf16 throughput mode
        LAUNCH("vproj_fwd", launch_vproj_fwd_bf16(io.video_features_bf16, reinterpret_cast<const uint16_t*>(c.PK(K.va_f16)), c.P(P.va_b), c.W(p.vf), R,
                                                  cf.video_feature_dim, c.drop(SITE_VIS), c.s));
    else
        LAUNCH("vproj_fwd", launch_vproj_fwd3(io.video_features, reinterpret_cast<const uint16_t*>(c.PK(K.va_f3)), c.P(P.va_b), c.W(p.vf), R, cf.video_feature_dim,
                                              c.drop(SITE_VIS), c.s));
    enc_fwd(c, P.fe, K.fe, p.ve, c.W(p.vf), io.v_mask, B, 0);
    c.s = qlong ? c.main : sq;
    const bool wt = cf.word_table != 0;       // trainable word table: its rows 0, 1, 2.. are pad, unk, the vocabulary
    LAUNCH("embed_fwd", launch_embed_fwd(io.word_ids, io.char_ids, wt ? c.P(P.unk) : io.pad_vec, c.P(P.unk) + (wt ? cf.word_dim : 0), wt ? c.P(P.unk) + 2 * cf.word_dim : io.glove_vec, c.P(P.char_tab), char_ptrs(c), c.PK(K.ccw_img), c.W(p.E),
                     reinterpret_cast<int8_t*>(c.W(p.argpos)), Rq, p.Lc, cf.word_dim, cf.char_dim, c.drop(SITE_WORD),
                     c.drop(SITE_CHAR), c.s));
    if (split3 && !c.dry) {                // behind embed_fwd: the shared encoder's packs (main stream) and the Embedding linear's (third stream)
        for (int k = 0; k < 2; ++k) {
            hipEvent_t ev = k ? pack_q_ev : pack_first_ev;
            const int rec = k ? pack_q_rec : pack_first_rec;
            if (!ev) continue;
            (void)hipStreamWaitEvent(c.s, ev, 0); c.h->mark_waiting(c.s);
            if (rec >= 0) c.h->prof_pending.push_back({c.s, rec});
        }
    }   // the shared encoder's packs (main stream)
    if (query_fused(c)) {
        // the rest of the query branch in ONE sample-local launch (kernels_query.hip)
        QueryFwdArgs qa;
        memset(&qa, 0, sizeof qa);
        if (!c.dry) {
            const EncP& E = P.fe; const EncPk& EK = K.fe; const EncWs& w = p.qe;
            qa.E = c.W(p.E); qa.Wemb3 = reinterpret_cast<const uint16_t*>(c.PK(K.emb_f3)); qa.b_emb = c.P(P.emb_b); qa.qf = c.W(p.qf);
            qa.pos = c.P(E.pos); qa.x0 = c.W(w.x0);
            for (int i = 0; i < 4; ++i) {
                qa.W3[i] = reinterpret_cast<const uint16_t*>(c.PK(EK.pw_f3[i]));
                qa.ln_g[i] = c.P(E.lng[i]); qa.ln_b[i] = c.P(E.lnb[i]); qa.dw_w[i] = c.P(E.dw[i]); qa.pw_b[i] = c.P(E.pwb[i]);
                qa.y[i] = c.W(w.y[i]); qa.u[i] = c.W(w.u[i]); qa.relu_mask[i] = reinterpret_cast<uint32_t*>(c.W(w.mask[i])); qa.dp[i] = c.drop(16 + i);
            }
            qa.Wqkv3 = reinterpret_cast<const uint16_t*>(c.PK(EK.qkv_f3)); qa.Wo3 = reinterpret_cast<const uint16_t*>(c.PK(EK.o_f3));
            qa.ln1_g = c.P(E.ln1g); qa.ln1_b = c.P(E.ln1b); qa.bq = c.P(E.qb); qa.bk = c.P(E.kb); qa.bv = c.P(E.vb);
            qa.h1 = c.W(w.h1); qa.q = c.W(w.q); qa.k = c.W(w.k); qa.v = c.W(w.v); qa.d1 = c.drop(16 + 4);
            qa.mask = io.q_mask; qa.ln2_g = c.P(E.ln2g); qa.ln2_b = c.P(E.ln2b); qa.bo = c.P(E.ob);
            qa.att = c.W(w.att); qa.lse = c.W(w.lse); qa.r = c.W(w.r); qa.h2 = c.W(w.h2); qa.out = c.W(w.out);
            qa.d2 = c.drop(16 + 5); qa.d3 = c.drop(16 + 6); qa.d4 = c.drop(16 + 7); qa.d5 = c.drop(16 + 8);
            qa.EW = cf.word_dim + 100; qa.L = Lq; qa.b_off = 0;
        }
        LAUNCH("query_fwd", launch_query_fwd(qa, B, c.s));
    } else {
    if ((cf.word_dim + 100) % 16 == 0)
        LAUNCH("linear_fwd", launch_linear_fwd3(c.W(p.E), reinterpret_cast<const uint16_t*>(c.PK(K.emb_f3)), c.P(P.emb_b), c.W(p.qf), Rq, cf.word_dim + 100, c.s));
    else          // a width the 16-wide K steps of the split kernel do not tile: the fp32-input MFMA kernel (K streamed in chunks)
        LAUNCH("linear_fwd", launch_linear_fwd(c.W(p.E), c.PK(K.emb_f), c.P(P.emb_b), c.W(p.qf), Rq, cf.word_dim + 100, c.s));
    enc_fwd(c, P.fe, K.fe, p.qe, c.W(p.qf), io.q_mask, B, 1);
    }
    c.s = c.main;
    c.order(sq, c.main);                   // join
    if (split3) c.order(sp, c.main);       // the remaining packs (long done)
    // short videos, short queries: the column kernel (column softmax over the clips, M = S_col^T C, WeightedPool) is folded into its neighbours --
    // one launch and one boundary less on the dependent chain (VSL_CQ_FOLD=0: three launches)
    static const bool fold_on = !(getenv("VSL_CQ_FOLD") && getenv("VSL_CQ_FOLD")[0] == '0');
    const bool cq_fold = fold_on && cq_col_folds(T, Lq);
    const CqPoolArgs pool{c.P(P.pool_w), c.P(P.cat_w), c.P(P.cat_b), c.W(p.alpha), c.W(p.pooled), c.W(p.pb)};
    LAUNCH("cq_score", launch_cq_score(c.W(p.ve.out), c.W(p.qe.out), io.q_mask, c.P(P.w4C), c.P(P.w4Q), c.P(P.w4mlu), c.W(p.S), c.W(p.Srow), B, T,
                    Lq, 0, c.drop(SITE_CQ_C), c.drop(SITE_CQ_Q), c.s, cq_fold ? &pool : nullptr));
    // M = S_col^T C is produced as per-tile partials (the backward's cqP1 arena is free until then) and summed by cq_out
    if (!cq_fold)
        LAUNCH("cq_col", launch_cq_col(c.W(p.ve.out), c.W(p.qe.out), c.W(p.S), io.v_mask, io.q_mask, c.P(P.pool_w), c.P(P.cat_w), c.P(P.cat_b),
                      c.W(p.Scol), c.W(p.cqP1), c.W(p.alpha), c.W(p.pooled), c.W(p.pb), B, T, Lq, c.s));
    // cq_out also carries CQConcatenate + HighLightLayer + gating (row-local on the tile it produces)
    LAUNCH("cq_out", launch_cq_out(c.W(p.ve.out), c.W(p.qe.out), c.W(p.Srow), c.W(p.cqP1), c.W(p.M), c.PK(K.cqa_f), c.P(P.cqa_b), c.W(p.cat),
                  c.W(p.f1), c.PK(K.cat1_f), c.W(p.pb), c.P(P.hl_w), c.P(P.hl_b), io.v_mask, c.W(p.f2), io.h_score, c.W(p.gated), B, T, Lq,
                  c.s, cq_fold ? c.W(p.S) : nullptr, cq_fold ? c.W(p.Scol) : nullptr));
    if (cf.predictor == 0) {
        // rnn head (:341-343): start = LSTM_s(x) * mask ; end = LSTM_e(start) * mask ; no LayerNorm in front of the span blocks
        // The recurrence is latency bound (one sample per CU, 0.6 us per step).  Up to RNN_FUSED_MAX_B samples the whole head is ONE launch: start
        // LSTM, input projection and end LSTM as three workgroups per sample that hand each step over in-launch (k_rnn_fwd).  Larger batches
        // pipeline the two LSTMs in TIME CHUNKS over three streams: while the start LSTM runs chunk k + 1 on the main stream, a side stream
        // projects its chunk k (x W_ih^T of the end LSTM, a row-mapped GEMM) and another runs the end LSTM over it.  A chunk launch resumes from
        // the state the previous one saved for the backward (h_{t-1}, c_{t-1}).  Chunking: lstm_chunks().
        const std::vector<int> chunks = lstm_chunks(T);
        auto lstm = [&](int l, int t0, int t1) {
            const LstmWs& w = p.lstm[l];
            LAUNCH("lstm_fwd", launch_lstm_fwd(c.W(w.gi), c.P(P.l_whh[l]), c.PK(K.l_hf[l]), c.P(P.l_bih[l]), c.P(P.l_bhh[l]), io.v_mask, c.W(w.gates),
                                               c.W(w.cseq), c.W(w.tseq), c.W(w.hprev), c.W(w.out), B, T, c.s, t0, t1));
        };
        // gate projections gi = x W_ih^T: (rows,128) x (128,512), no bias, bf16x6 on the matrix cores
        auto gi = [&](int l, const float* x, int rows, int seg = 0, int off = 0) {
            LAUNCH("lstm_gi", launch_linear_fwd3(x, reinterpret_cast<const uint16_t*>(c.PK(K.l_f3[l])), nullptr, c.W(p.lstm[l].gi), rows, D, c.s, 4 * D, seg, T, off));
        };
        gi(0, c.W(p.gated), R);
        hipStream_t main_s = c.s;
        if (p.h_gran >= 0) {
            // one launch, three workgroups per sample (kernels_lstm.hip: k_rnn_fwd)
            RnnFwdArgs a;
            memset(&a, 0, sizeof a);
            a.gi0 = c.W(p.lstm[0].gi); a.Wih1 = c.PK(K.l_if); a.mask = io.v_mask;
            for (int l = 0; l < 2; ++l) {
                const LstmWs& w = p.lstm[l];
                a.Whh[l] = c.PK(K.l_hf[l]); a.bih[l] = c.P(P.l_bih[l]); a.bhh[l] = c.P(P.l_bhh[l]);
                a.gates[l] = c.W(w.gates); a.cseq[l] = c.W(w.cseq); a.tseq[l] = c.W(w.tseq); a.hprev[l] = c.W(w.hprev); a.out[l] = c.W(w.out);
            }
            a.h_gran = reinterpret_cast<unsigned long long*>(c.W(p.h_gran)); a.gi_gran = reinterpret_cast<unsigned long long*>(c.W(p.gi_gran));
            long long gen = 0;
            a.epoch = rnn_epoch(&gen); a.B = B; a.T = T;
            rnn_granules_fresh(c, *c.p, gen);
            LAUNCH("rnn_fwd", launch_rnn_fwd(a, c.s));
        } else if (chunks.size() < 2 || sq == main_s) {
            lstm(0, 0, T);
            gi(1, c.W(p.lstm[0].out), R);
            lstm(1, 0, T);
        } else {
            // Three streams (round 4): the end LSTM's input projection of chunk k (a 15 us GEMM) runs on the third stream behind the start
            // LSTM's chunk k, BESIDE the end LSTM's chunk k - 1 -- on the end LSTM's own stream it made that chain 59 us per chunk against 39
            // for the start LSTM, and the end LSTM finished 144 us after the start LSTM (profiles/r04_notes.md section 7).
            hipStream_t sg = c.side(1) != main_s ? c.side(1) : sq;
            int t0 = 0;
            for (int len : chunks) {
                const int t1 = t0 + len;
                lstm(0, t0, t1);
                c.order(main_s, sg);
                c.s = sg;
                gi(1, c.W(p.lstm[0].out), B * (t1 - t0), t1 - t0, t0);
                c.order(sg, sq);
                c.s = sq;
                lstm(1, t0, t1);
                c.s = main_s;
                t0 = t1;
            }
            c.order(sq, main_s);
        }
        HeadArgs hs{c.W(p.lstm[0].out), nullptr, nullptr, c.PK(K.s0_f), c.P(P.s0b), c.P(P.s1w), c.P(P.s1b), c.W(p.hid_s), nullptr, io.start_logits};
        HeadArgs he{c.W(p.lstm[1].out), nullptr, nullptr, c.PK(K.e0_f), c.P(P.e0b), c.P(P.e1w), c.P(P.e1b), c.W(p.hid_e), nullptr, io.end_logits};
        LAUNCH("head_fwd", launch_head_fwd(hs, he, c.W(p.gated), io.v_mask, R, c.s));
        return;
    }
    enc_fwd(c, P.pe, K.pe, p.p1, c.W(p.gated), io.v_mask, B, 2);
    HeadArgs hs{c.W(p.p1.out), c.P(P.sln_g), c.P(P.sln_b), c.PK(K.s0_f), c.P(P.s0b), c.P(P.s1w), c.P(P.s1b), c.W(p.hid_s),
                c.W(p.lnf_s), io.start_logits};
    HeadArgs he{c.W(p.p2.out), c.P(P.eln_g), c.P(P.eln_b), c.PK(K.e0_f), c.P(P.e0b), c.P(P.e1w), c.P(P.e1b), c.W(p.hid_e),
                c.W(p.lnf_e), io.end_logits};
    // T <= 128: the second pass' attention-block kernel goes on with both span heads on its tiles (one launch less on the dependent chain)
    static const bool heads_on = !(getenv("VSL_HEADS_FUSED") && getenv("VSL_HEADS_FUSED")[0] == '0');
    const bool heads_hosted = heads_on && cf.num_heads == 8 && attn_block_fwd_hosts_heads(T);
    HeadTail ht{hs, he, c.W(p.gated), io.v_mask};
    enc_fwd(c, P.pe, K.pe, p.p2, c.W(p.p1.out), io.v_mask, B, 3, heads_hosted ? &ht : nullptr);
    if (!heads_hosted) LAUNCH("head_fwd", launch_head_fwd(hs, he, c.W(p.gated), io.v_mask, R, c.s));
}

// ------------------------------------------------------------------------------------------------ backward
WgradJob wjob() { WgradJob j; memset(&j, 0, sizeof j); return j; }

// backward of one FeatureEncoder application: dy = grad wrt its output; writes grad wrt its input (dx0_out)
// the attention-output backward of encoder application `app` (its LayerNorm-2 partial slabs are allocated here: call once per application)
AttnOutBwdArgs attn_out_bwd_args(Ctx& c, const EncP& P, const EncPk& K, const EncWs& w, const float* dy, const float* dy2, int app) {
    const Plan& p = *c.p;
    const EncTmp& t = p.tmp[app];
    const int ntiles = (w.R + TILE_M - 1) / TILE_M;
    float* p_ln2g = c.slab(P.ln2g, D, ntiles);
    float* p_ln2b = c.slab(P.ln2b, D, ntiles);
    AttnOutBwdArgs a;
    memset(&a, 0, sizeof a);
    if (!c.dry) a = AttnOutBwdArgs{dy, dy2, c.W(w.r), c.P(P.ln2g), c.PK(K.o_t), c.W(t.go), c.W(t.dr), p_ln2g, p_ln2b, c.drop(app * 16 + 7), c.drop(app * 16 + 8)};
    return a;
}
// attn_out_done: the attention-output backward of this application already ran (fused into the span heads' kernel: run_backward)
// tail_ao / tail_cq: what the conv block's backward workgroups go on with on their tile of dx0 (CbBwdArgs::tail; the caller has checked
// convblock_bwd_hosts_tail and skips that launch)
struct LinTail { const uint16_t* WT3; float* dA; int K, Kc; };     // CbBwdArgs::tail == 3
void enc_bwd(Ctx& c, const EncP& P, const EncPk& K, const EncWs& w, const float* dy, const float* dy2, int64_t dx0_off,
             const float* mask, int Bn, int app, hipStream_t sw, WgradBatch* defer_pw = nullptr, bool attn_out_done = false,
             const AttnOutBwdArgs* tail_ao = nullptr, const CqcatBwdArgs* tail_cq = nullptr, const LinTail* tail_lin = nullptr) {
    // sw: stream of the early (out_layer / q,k,v) weight gradients.  The pointwise-conv batch goes to `sw` too unless the
    // caller asks for it back (defer_pw) to launch it on its own stream WITHOUT a cross-stream wait (each costs ~16 us).
    float* dx0_out = c.dry ? nullptr : c.W(dx0_off);
    const Plan& p = *c.p;
    const EncTmp& t = p.tmp[app];
    const int R = w.R, L = w.L, H = c.h->cfg.num_heads;
    const int ntiles = (R + TILE_M - 1) / TILE_M, nchunk = (R + WG_ROWS - 1) / WG_ROWS;
    if (!attn_out_done) {
        const AttnOutBwdArgs ao = attn_out_bwd_args(c, P, K, w, dy, dy2, app);
        LAUNCH("attn_out_bwd", launch_attn_out_bwd(ao.dy, ao.dy2, ao.r_in, ao.ln_g, ao.WTpack, ao.g_o, ao.dr, ao.p_lng, ao.p_lnb, R, ao.d4, ao.d5, c.s));
    }
    LAUNCH("attn_bwd", launch_attn_bwd(c.W(w.q), c.W(w.k), c.W(w.v), c.W(w.att), c.W(t.dr), c.W(w.lse), mask, c.W(t.dq), c.W(t.dk),
                           c.W(t.dv), Bn, L, H, 0, c.drop(app * 16 + 5), c.drop(app * 16 + 6), c.s));
    // whole tiles / sample tiles (and one dQ slab): the conv block's backward kernel computes its own incoming gradient from dq / dk / dv (CbBwdArgs::qk)
    static const bool fuse_on = !(getenv("VSL_QKV_FUSED") && getenv("VSL_QKV_FUSED")[0] == '0');
    const bool fuse_qkv = fuse_on && convblock_bwd_hosts_qkv(R, L) && attn_bwd_dq_slabs(L) == 1;
    const int nsl1 = fuse_qkv ? convblock_slabs(R, L) : ntiles;
    float* p_ln1g = c.slab(P.ln1g, D, nsl1);
    float* p_ln1b = c.slab(P.ln1b, D, nsl1);
    if (!fuse_qkv)
        LAUNCH("qkv_bwd", launch_qkv_bwd(c.W(t.dq), c.W(t.dk), c.W(t.dv), c.W(w.y[3]), c.W(t.dr), c.P(P.ln1g), reinterpret_cast<const uint16_t*>(c.PK(K.qkv_t3)),
                              c.W(t.ga), p_ln1g, p_ln1b, R, c.drop(app * 16 + 4), c.s, attn_bwd_dq_slabs(L)));
    {   // out_layer + fused q/k/v weight gradients: every input exists now -> side stream, beside the conv chain
        WgradBatch wb;
        memset(&wb, 0, sizeof wb);
        {
            WgradJob j = wjob();
            j.G[0] = c.dry ? nullptr : c.W(t.go); j.nG = 1; j.A[0] = c.dry ? nullptr : c.W(w.h2); j.nA = 1; j.K = D; j.R = R;
            j.out = c.slab(P.ow, D * D, nchunk);
            j.out_bias[0] = c.slab(P.ob, D, nchunk);
            wb.j[wb.n++] = j;
        }
        {
            WgradJob j = wjob();
            if (!c.dry) { j.G[0] = c.W(t.dq); j.G[1] = c.W(t.dk); j.G[2] = c.W(t.dv); j.A[0] = c.W(w.h1); }
            j.nG = 3; j.nA = 1; j.K = D; j.R = R;
            const int64_t o = c.part_alloc((int64_t)nchunk * 3 * D * D);
            c.reg(P.qw, D * D, p.partial + o, nchunk, 3 * D * D);
            c.reg(P.kw, D * D, p.partial + o + D * D, nchunk, 3 * D * D);
            c.reg(P.vw, D * D, p.partial + o + 2 * D * D, nchunk, 3 * D * D);
            j.out = c.part_ptr(o);
            j.out_bias[0] = c.slab(P.qb, D, nchunk);
            j.out_bias[1] = c.slab(P.kb, D, nchunk);
            j.out_bias[2] = c.slab(P.vb, D, nchunk);
            wb.j[wb.n++] = j;
        }
        wgrad_async(c, sw, wb);
    }
    float* g = c.dry ? nullptr : c.W(t.ga);
    {
        // the four layers in ONE launch (kernels_enc.hip: 12-row recomputed halo); slab order = the per-layer chain's
        CbBwdArgs a;
        memset(&a, 0, sizeof a);
        const int nsl = convblock_slabs(R, L);
        for (int i = 3; i >= 0; --i) {
            a.p_lng[i] = c.slab(P.lng[i], D, nsl);
            a.p_lnb[i] = c.slab(P.lnb[i], D, nsl);
            a.p_dw[i] = c.slab(P.dw[i], D * DWK, nsl);
        }
        if (tail_ao) { a.tail = 1; a.tail_ao = *tail_ao; }
        if (tail_cq) { a.tail = 2; a.tail_cq = *tail_cq; }
        if (tail_lin) { a.tail = 3; a.tail_lin_WT3 = tail_lin->WT3; a.tail_lin_dA = tail_lin->dA; a.tail_lin_K = tail_lin->K; a.tail_lin_Kc = tail_lin->Kc; }
        if (fuse_qkv) {
            a.qkv = 1;
            if (!c.dry) a.qk = QkvBwdFuse{c.W(t.dq), c.W(t.dk), c.W(t.dv), c.W(w.y[3]), c.W(t.dr), c.P(P.ln1g), reinterpret_cast<const uint16_t*>(c.PK(K.qkv_t3)),
                                          p_ln1g, p_ln1b, c.drop(app * 16 + 4)};
        }
        if (!c.dry) {
            a.dy = g; a.dx0 = dx0_out; a.R = R; a.L = L;
            for (int i = 0; i < 4; ++i) {
                a.x[i] = i > 0 ? c.W(w.y[i - 1]) : c.W(w.x0);
                a.relu_mask[i] = reinterpret_cast<const uint32_t*>(c.W(w.mask[i]));
                a.ln_g[i] = c.P(P.lng[i]); a.ln_b[i] = c.P(P.lnb[i]); a.dw_w[i] = c.P(P.dw[i]);
                a.dp[i] = c.drop(app * 16 + i); a.gz[i] = c.W(t.gz[i]);
                a.WT3[i] = reinterpret_cast<const uint16_t*>(c.PK(K.pw_t3[i]));
            }
        }
        LAUNCH("convblock_bwd", launch_convblock_bwd(a, c.s));
    }
    // pointwise-conv weight gradients (inputs complete only now); Wo / QKV were launched right after qkv_bwd
    {
        WgradBatch wb;
        memset(&wb, 0, sizeof wb);
        for (int i = 0; i < 4; ++i) {
            WgradJob j = wjob();
            if (!c.dry) { j.G[0] = c.W(t.gz[i]); j.A[0] = c.W(w.u[i]); }
            j.nG = 1; j.nA = 1; j.K = D; j.R = R;
            j.rows = c.pw_rows;
            const int nch_pw = wgrad_chunks(R, wgrad_rows(j));
            j.out = c.slab(P.pw[i], D * D, nch_pw);
            j.out_bias[0] = c.slab(P.pwb[i], D, nch_pw);
            wb.j[wb.n++] = j;
        }
        if (defer_pw) *defer_pw = wb;
        else wgrad_async(c, sw, wb);
    }
    // positional table (:202): dpos[t] = sum_b dx0[b, t] -- the per-sample rows of dx0 ARE the partial slabs
    c.reg(P.pos, c.h->cfg.max_pos_len * D, dx0_off, Bn, L * D, 0, 0, L * D);
}

// the query pass' backward as ONE sample-local launch (kernels_query.hip: k_query_bwd) + the weight-gradient jobs that read what it leaves:
// out_layer, fused q/k/v, the four pointwise convs (the Embedding linear's job is the caller's: it also owns dqf / E)
void query_bwd(Ctx& c, const float* dy, int64_t dx0_off, WgradBatch& jobs) {
    const vsl_config& cf = c.h->cfg;
    const ModelPk& MK = c.h->K;
    const EncP& P = c.h->P.fe;
    const EncPk& K = c.h->K.fe;
    const Plan& p = *c.p;
    const EncWs& w = p.qe;
    const EncTmp& t = p.tmp[1];
    const int B = p.B, Lq = p.Lq, R = w.R, nchunk = (R + WG_ROWS - 1) / WG_ROWS, app = 1;
    QueryBwdArgs a;
    memset(&a, 0, sizeof a);
    a.p_ln2g = c.slab(P.ln2g, D, B); a.p_ln2b = c.slab(P.ln2b, D, B);
    a.p_ln1g = c.slab(P.ln1g, D, B); a.p_ln1b = c.slab(P.ln1b, D, B);
    for (int i = 3; i >= 0; --i) {
        a.p_lng[i] = c.slab(P.lng[i], D, B);
        a.p_lnb[i] = c.slab(P.lnb[i], D, B);
        a.p_dw[i] = c.slab(P.dw[i], D * DWK, B);
    }
    if (!c.dry) {
        a.dout = dy; a.r = c.W(w.r); a.ln2_g = c.P(P.ln2g); a.WoT3 = reinterpret_cast<const uint16_t*>(c.PK(K.o_t3)); a.go = c.W(t.go);
        a.d2 = c.drop(app * 16 + 5); a.d3 = c.drop(app * 16 + 6); a.d4 = c.drop(app * 16 + 7); a.d5 = c.drop(app * 16 + 8);
        a.q = c.W(w.q); a.k = c.W(w.k); a.v = c.W(w.v); a.att = c.W(w.att); a.lse = c.W(w.lse); a.mask = c.io->q_mask;
        a.dq = c.W(t.dq); a.dk = c.W(t.dk); a.dv = c.W(t.dv);
        a.WqkvT3 = reinterpret_cast<const uint16_t*>(c.PK(K.qkv_t3)); a.y3 = c.W(w.y[3]); a.ln1_g = c.P(P.ln1g); a.d1 = c.drop(app * 16 + 4);
        for (int i = 0; i < 4; ++i) {
            a.WT3[i] = reinterpret_cast<const uint16_t*>(c.PK(K.pw_t3[i]));
            a.x[i] = i > 0 ? c.W(w.y[i - 1]) : c.W(w.x0);
            a.relu_mask[i] = reinterpret_cast<const uint32_t*>(c.W(w.mask[i]));
            a.ln_g[i] = c.P(P.lng[i]); a.ln_b[i] = c.P(P.lnb[i]); a.dw_w[i] = c.P(P.dw[i]);
            a.dp[i] = c.drop(app * 16 + i); a.gz[i] = c.W(t.gz[i]);
        }
        a.dx0 = c.W(dx0_off);
        a.WembT3 = reinterpret_cast<const uint16_t*>(c.PK(MK.emb_t3)); a.dE = c.W(p.dE);
        a.EW = cf.word_dim + 100; a.EWc = MK.emb_t3_cols; a.L = Lq; a.b_off = 0;
    }
    LAUNCH("query_bwd", launch_query_bwd(a, B, c.s));
    {
        WgradJob j = wjob();
        j.G[0] = c.dry ? nullptr : c.W(t.go); j.nG = 1; j.A[0] = c.dry ? nullptr : c.W(w.h2); j.nA = 1; j.K = D; j.R = R;
        j.out = c.slab(P.ow, D * D, nchunk);
        j.out_bias[0] = c.slab(P.ob, D, nchunk);
        jobs.j[jobs.n++] = j;
    }
    {
        WgradJob j = wjob();
        if (!c.dry) { j.G[0] = c.W(t.dq); j.G[1] = c.W(t.dk); j.G[2] = c.W(t.dv); j.A[0] = c.W(w.h1); }
        j.nG = 3; j.nA = 1; j.K = D; j.R = R;
        const int64_t o = c.part_alloc((int64_t)nchunk * 3 * D * D);
        c.reg(P.qw, D * D, p.partial + o, nchunk, 3 * D * D);
        c.reg(P.kw, D * D, p.partial + o + D * D, nchunk, 3 * D * D);
        c.reg(P.vw, D * D, p.partial + o + 2 * D * D, nchunk, 3 * D * D);
        j.out = c.part_ptr(o);
        j.out_bias[0] = c.slab(P.qb, D, nchunk);
        j.out_bias[1] = c.slab(P.kb, D, nchunk);
        j.out_bias[2] = c.slab(P.vb, D, nchunk);
        jobs.j[jobs.n++] = j;
    }
    for (int i = 0; i < 4; ++i) {
        WgradJob j = wjob();
        if (!c.dry) { j.G[0] = c.W(t.gz[i]); j.A[0] = c.W(w.u[i]); }
        j.nG = 1; j.nA = 1; j.K = D; j.R = R;
        j.out = c.slab(P.pw[i], D * D, nchunk);
        j.out_bias[0] = c.slab(P.pwb[i], D, nchunk);
        jobs.j[jobs.n++] = j;
    }
    c.reg(P.pos, cf.max_pos_len * D, dx0_off, B, Lq * D, 0, 0, Lq * D);     // positional table (:202): the per-sample rows of dx0 are the partial slabs
}

void run_backward(Ctx& c) {
    const vsl_config& cf = c.h->cfg;
    const ModelP& P = c.h->P;
    const ModelPk& K = c.h->K;
    const Plan& p = *c.p;
    const int B = p.B, T = p.T, Lq = p.Lq, R = B * T, Rq = B * Lq;
    const int ntiles = (R + TILE_M - 1) / TILE_M, nchunk = (R + WG_ROWS - 1) / WG_ROWS, nchunk_q = (Rq + WG_ROWS - 1) / WG_ROWS;
    const vsl_io* io = c.io;
    // streams: `main` carries the dependent dX chain; sw = every weight-gradient GEMM (they only feed the final
    // reduction); sq = the query-side chain (query encoder pass + embedding stack) once CQAttention's backward is done
    hipStream_t sw = c.dry ? nullptr : c.side(1), sq = c.dry ? nullptr : c.side(0);
    auto on_stream = [&](hipStream_t st, auto&& fn) { hipStream_t keep = c.s; c.order(keep, st); c.s = st; fn(); c.s = keep; };
    c.defer_w = !c.dry && cf.predictor == 1;
    bool hosted = false;                    // transformer head on whole tiles: two launches ride inside the conv block's backward kernels (below)
    float *p_hlw = nullptr, *p_hlb = nullptr;
    CqcatBwdArgs cqa;
    // ---- span heads
    HeadBwdArgs hs, he;
    memset(&hs, 0, sizeof hs);
    memset(&he, 0, sizeof he);
    const bool rnn = cf.predictor == 0;
    // vsl_io.fused_loss: the loss is part of this call.  Whole tiles + the caller's mask sum: k_loss_fused goes to the weight-gradient stream (nothing
    // on the dependent chain reads what it writes) and the consumers of the seeds -- the span heads' backward, the highlight layer's -- compute
    // them from the logits themselves.  Otherwise the loss launches first, on the caller's stream, and the seeds are read as they always were.
    const vsl_loss_io* fl = c.dry ? nullptr : io->fused_loss;
    const float *seed_s = nullptr, *seed_e = nullptr, *seed_h = nullptr;
    bool seeds_inline = false;
    HlSeed hlseed{nullptr, nullptr, 0.f, 0.f};
    if (!c.dry) {
        seed_s = fl ? fl->d_start_logits : io->d_start_logits; seed_e = fl ? fl->d_end_logits : io->d_end_logits; seed_h = fl ? fl->d_h_score : io->d_h_score;
        if (fl) {
            static const bool inline_on = !(getenv("VSL_LOSS_INLINE") && getenv("VSL_LOSS_INLINE")[0] == '0');
            seeds_inline = inline_on && T >= TILE_M && fl->mask_sum > 0.f && c.h->loss_counter != nullptr;       // (T >= 32: a row tile touches at most two samples)
            c.pend_loss = fl;
            if (!seeds_inline) loss_on_side(c);            // (here and now, on the caller's stream: the seeds are read from memory)
            else hlseed = HlSeed{fl->h_labels, io->v_mask, fl->w_highlight, fl->mask_sum};
        }
        hs = HeadBwdArgs{seed_s, c.W(p.hid_s), c.W(p.p1.out), rnn ? nullptr : c.P(P.sln_g), c.PK(K.s0_t), c.P(P.s1w), c.W(p.gz_s),
                         c.W(p.dfeat_s), c.W(p.dxh_s), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0};
        he = HeadBwdArgs{seed_e, c.W(p.hid_e), c.W(p.p2.out), rnn ? nullptr : c.P(P.eln_g), c.PK(K.e0_t), c.P(P.e1w), c.W(p.gz_e),
                         c.W(p.dfeat_e), c.W(p.dxh_e), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0};
        if (seeds_inline) {
            hs.logits = io->start_logits; hs.label = fl->start_labels; hs.cs = fl->w_loc * fl->inv_batch; hs.T = T;
            he.logits = io->end_logits; he.label = fl->end_labels; he.cs = fl->w_loc * fl->inv_batch; he.T = T;
        }
    }
    hs.p_b0 = c.slab(P.s0b, D, ntiles); hs.p_w1 = c.slab(P.s1w, D, ntiles); hs.p_b1 = c.slab(P.s1b, 1, ntiles);
    he.p_b0 = c.slab(P.e0b, D, ntiles); he.p_w1 = c.slab(P.e1w, D, ntiles); he.p_b1 = c.slab(P.e1b, 1, ntiles);
    if (!rnn) {
        hs.p_lng = c.slab(P.sln_g, D, ntiles); hs.p_lnb = c.slab(P.sln_b, D, ntiles);
        he.p_lng = c.slab(P.eln_g, D, ntiles); he.p_lnb = c.slab(P.eln_b, D, ntiles);
    }
    // transformer head: the end head's workgroups run the attention-output backward of the second predictor pass on their own tile of dfeat_e
    AttnOutBwdArgs ao2;
    if (!rnn) ao2 = attn_out_bwd_args(c, P.pe, K.pe, p.p2, nullptr, nullptr, 3);
    LAUNCH("head_bwd", launch_head_bwd(hs, he, R, c.s, rnn ? nullptr : &ao2));
    {
        WgradBatch wb;
        memset(&wb, 0, sizeof wb);
        for (int e = 0; e < 2; ++e) {
            WgradJob j = wjob();
            if (!c.dry) { j.G[0] = c.W(e ? p.gz_e : p.gz_s); j.A[0] = rnn ? c.W(p.lstm[e].out) : c.W(e ? p.lnf_e : p.lnf_s); j.A[1] = c.W(p.gated); }
            j.nG = 1; j.nA = 2; j.K = 2 * D; j.R = R;
            j.out = c.slab(e ? P.e0w : P.s0w, D * 2 * D, nchunk);
            wb.j[wb.n++] = j;
        }
        wgrad_async(c, sw, wb);
    }
    if (rnn) {
        // ---- rnn head: BPTT through the end LSTM, then the start LSTM (whose output also feeds the start span block).
        //      Pipelined in time chunks like the forward (run_forward): the end LSTM walks the chunks backwards on the main
        //      stream; behind it the side stream turns the chunk's gate gradients into the start LSTM's incoming gradient
        //      (dx = dG W_ih, row-mapped GEMM) and runs the start LSTM over the same chunk.
        const std::vector<int> chunks = lstm_chunks(T);
        auto bwd = [&](int l, int t0, int t1) {
            const LstmWs& w = p.lstm[l];
            const float* d1 = c.dry ? nullptr : c.W(l ? p.dfeat_e : p.dfeat_s);
            const float* d2 = (c.dry || l) ? nullptr : c.W(p.g_s1);
            LAUNCH("lstm_bwd", launch_lstm_bwd(d1, d2, io->v_mask, c.W(w.gates), c.W(w.cseq), c.W(w.tseq), c.P(P.l_whh[l]), c.PK(K.l_hb[l]), c.W(w.dG),
                                               B, T, c.s, c.W(w.carry), t0, t1));
        };
        // dx = dG W_ih : (rows,512) x (512,128), K-streamed GEMM kernel of the visual projection, no dropout, zero bias
        auto dx = [&](int l, int t0, int t1) {
            const LstmWs& w = p.lstm[l];
            const bool all = t0 == 0 && t1 == T;
            LAUNCH("lstm_dx", launch_vproj_fwd3(c.W(w.dG), reinterpret_cast<const uint16_t*>(c.PK(K.l_t3[l])), c.PK(K.zero128), c.W(l ? p.g_s1 : p.g_gated),
                                               all ? R : B * (t1 - t0), 4 * D, Drop{0u, 0u, 1.f}, c.s, all ? 0 : t1 - t0, T, t0));
        };
        const bool piped = !c.dry && chunks.size() >= 2 && sq != c.s;
        if (!c.dry) {
            if (p.h_gran >= 0) {
                RnnBwdArgs a;
                memset(&a, 0, sizeof a);
                a.dout[0] = c.W(p.dfeat_s); a.dout[1] = c.W(p.dfeat_e); a.mask = io->v_mask; a.Wih1 = c.PK(K.l_ib);
                for (int l = 0; l < 2; ++l) {
                    const LstmWs& w = p.lstm[l];
                    a.gates[l] = c.W(w.gates); a.cseq[l] = c.W(w.cseq); a.tseq[l] = c.W(w.tseq); a.Whh[l] = c.PK(K.l_hb[l]); a.dG[l] = c.W(w.dG);
                }
                a.dg_gran = reinterpret_cast<unsigned long long*>(c.W(p.dg_gran)); a.dx_gran = reinterpret_cast<unsigned long long*>(c.W(p.dx_gran));
                long long gen = 0;
                a.epoch = rnn_epoch(&gen); a.B = B; a.T = T;
                rnn_granules_fresh(c, *c.p, gen);
                LAUNCH("rnn_bwd", launch_rnn_bwd(a, c.s));
                dx(0, 0, T);
            } else if (!piped) {
                bwd(1, 0, T); dx(1, 0, T); bwd(0, 0, T); dx(0, 0, T);
            } else {
                hipStream_t main_s = c.s;
                int t1 = T;
                hipStream_t sg = sw != main_s ? sw : sq;      // the dx GEMM of a chunk beside the start LSTM's previous chunk (see run_forward)
                for (int len : chunks) {
                    const int t0 = t1 - len;
                    bwd(1, t0, t1);
                    c.order(main_s, sg);
                    c.s = sg;
                    dx(1, t0, t1);
                    c.order(sg, sq);
                    c.s = sq;
                    bwd(0, t0, t1);
                    c.s = main_s;
                    t1 = t0;
                }
                c.order(sq, main_s);
                dx(0, 0, T);
            }
        }
        for (int l = 1; l >= 0; --l) {
            const LstmWs& w = p.lstm[l];
            // weight gradients: dW_ih = dG^T x, dW_hh = dG^T h_prev, db_ih = db_hh = column sums of dG (two gate pairs per job)
            WgradBatch wb;
            memset(&wb, 0, sizeof wb);
            for (int which = 0; which < 2; ++which) {                     // 0: W_ih (+ biases), 1: W_hh
                for (int half = 0; half < 2; ++half) {
                    WgradJob j = wjob();
                    if (!c.dry) {
                        j.G[0] = c.W(w.dG) + (2 * half) * D; j.G[1] = c.W(w.dG) + (2 * half + 1) * D;
                        j.A[0] = which ? c.W(w.hprev) : (l ? c.W(p.lstm[0].out) : c.W(p.gated));
                    }
                    j.nG = 2; j.nA = 1; j.K = D; j.R = R; j.ldg = 4 * D;
                    const int dst = (which ? P.l_whh[l] : P.l_wih[l]) + half * 2 * D * D;
                    j.out = c.slab(dst, 2 * D * D, nchunk);
                    if (which == 0) {
                        for (int g = 0; g < 2; ++g) {
                            const int64_t o = c.part_alloc((int64_t)nchunk * D);
                            c.reg(P.l_bih[l] + (2 * half + g) * D, D, p.partial + o, nchunk, D);
                            c.reg(P.l_bhh[l] + (2 * half + g) * D, D, p.partial + o, nchunk, D);
                            j.out_bias[g] = c.part_ptr(o);
                        }
                    }
                    wb.j[wb.n++] = j;
                }
            }
            on_stream(sw, [&] { LAUNCH("wgrad", launch_wgrad(wb, c.s)); });
        }
    } else {
    // ---- predictor encoder, second pass (input = output of the first pass), then first pass
    // tile-local continuations inside the conv block's backward kernel (whole tiles): pass 2's conv block goes on with pass 1's attention-output
    // backward (dy = its dx0 + the start head's LayerNorm path), pass 1's with the CQConcatenate backward (dg0 = its dx0)
    hosted = convblock_bwd_hosts_tail(R, T);
    AttnOutBwdArgs ao1;
    if (hosted) ao1 = attn_out_bwd_args(c, P.pe, K.pe, p.p1, nullptr, c.dry ? nullptr : c.W(p.dfeat_s), 2);
    enc_bwd(c, P.pe, K.pe, p.p2, c.dry ? nullptr : c.W(p.dfeat_e), nullptr, p.g_s1, c.dry ? nullptr : io->v_mask, B, 3, sw, nullptr, true,
            hosted ? &ao1 : nullptr);
    wgrad_flush(c, sw);   // span heads + pass-2 weight gradients: one ordering point
    // grad wrt the first pass' output = (input grad of the second pass) + (LayerNorm path of the start head)
    if (hosted) {
        p_hlw = c.slab(P.hl_w, D, ntiles);
        p_hlb = c.slab(P.hl_b, 1, ntiles);
        memset(&cqa, 0, sizeof cqa);
        if (!c.dry) cqa = CqcatBwdArgs{nullptr, c.W(p.dxh_s), c.W(p.dxh_e), seed_h, c.W(p.f2), io->h_score, c.P(P.hl_w), c.PK(K.cat1_t), c.W(p.df2), c.W(p.df1), p_hlw, p_hlb,
                                       hlseed.h_lab, hlseed.vmask, hlseed.w_hl, hlseed.mask_sum};
    }
    enc_bwd(c, P.pe, K.pe, p.p1, c.dry ? nullptr : c.W(p.g_s1), c.dry ? nullptr : c.W(p.dfeat_s), p.g_gated,
            c.dry ? nullptr : io->v_mask, B, 2, sw, nullptr, hosted, nullptr, hosted ? &cqa : nullptr);
    }
    // ---- gating + highlight + CQConcatenate
    if (!hosted) {
        p_hlw = c.slab(P.hl_w, D, ntiles);
        p_hlb = c.slab(P.hl_b, 1, ntiles);
        LAUNCH("cqcat_bwd", launch_cqcat_bwd(c.W(p.g_gated), c.W(p.dxh_s), c.W(p.dxh_e), seed_h, c.W(p.f2), io->h_score, c.P(P.hl_w),
                                c.PK(K.cat1_t), c.W(p.df2), c.W(p.df1), p_hlw, p_hlb, R, c.s, hlseed.h_lab ? &hlseed : nullptr));
    }
    {
        WgradBatch wb;
        memset(&wb, 0, sizeof wb);
        {   // first half of the (128, 256) CQConcatenate weight: destination rows are strided by 256
            WgradJob j = wjob();
            if (!c.dry) { j.G[0] = c.W(p.df2); j.A[0] = c.W(p.f1); }
            j.nG = 1; j.nA = 1; j.K = D; j.R = R;
            const int64_t o = c.part_alloc((int64_t)nchunk * D * D);
            c.reg(P.cat_w, D * D, p.partial + o, nchunk, D * D, D, 2 * D);
            j.out = c.part_ptr(o);
            wb.j[wb.n++] = j;
        }
        {   // cqa_linear (128, 512)
            WgradJob j = wjob();
            if (!c.dry) { j.G[0] = c.W(p.df1); j.Afull = c.W(p.cat); }
            j.nG = 1; j.nA = 0; j.K = 4 * D; j.R = R;
            j.out = c.slab(P.cqa_w, D * 4 * D, nchunk);
            j.out_bias[0] = c.slab(P.cqa_b, D, nchunk);
            wb.j[wb.n++] = j;
        }
        wgrad_async(c, sw, wb);
        wgrad_flush(c, sw);                 // pass-1 + fusion weight gradients: one ordering point
        c.defer_w = false;
    }
    // ---- CQAttention + WeightedPool / pooled-bias backward: four tile-parallel kernels (kernels_bwd.hip)
    CqBwdArgs q;
    {
        const int ntile = (T + TILE_M - 1) / TILE_M;
        memset(&q, 0, sizeof q);
        if (!c.dry) {
            q.df1 = c.W(p.df1); q.df2 = c.W(p.df2); q.C = c.W(p.ve.out); q.Qf = c.W(p.qe.out); q.Srow = c.W(p.Srow);
            q.Scol = c.W(p.Scol); q.M = c.W(p.M); q.alpha = c.W(p.alpha); q.pooled = c.W(p.pooled); q.WcqaT = c.PK(K.cqa_t);
            q.w4C = c.P(P.w4C); q.w4Q = c.P(P.w4Q); q.w4mlu = c.P(P.w4mlu); q.pool_w = c.P(P.pool_w); q.Wcat = c.P(P.cat_w);
            q.dC = c.W(p.dC); q.dQ = c.W(p.dQtot); q.dSr = c.W(p.dSr); q.dSs = c.W(p.dSs);
            q.P1 = c.W(p.cqP1); q.P2 = c.W(p.cqP2); q.P3 = c.W(p.cqP3); q.P4 = c.W(p.cqP4); q.P5 = c.W(p.cqP5);
        }
        q.T = T; q.Lq = Lq; q.b_off = 0; q.dc = c.drop(SITE_CQ_C); q.dq = c.drop(SITE_CQ_Q);
        q.p_w4C = c.slab(P.w4C, D, B * ntile); q.p_w4mlu = c.slab(P.w4mlu, D, B * ntile);
        q.p_w4Q = c.slab(P.w4Q, D, B); q.p_pool = c.slab(P.pool_w, D, B); q.p_bcat = c.slab(P.cat_b, D, B);
        const int64_t o = c.part_alloc((int64_t)B * D * D);
        c.reg(P.cat_w + D, D * D, p.partial + o, B, D * D, D, 2 * D);          // second half of the (128, 256) weight
        q.p_W2 = c.part_ptr(o);
        LAUNCH("cq_bwd", launch_cq_bwd(q, B, c.s));
    }
    // early reduction (predictor / heads / CQ parameters): all their partials exist once the launches above are done
    // (the pooled-query parameters of k_cq_bwd_d belong to the late reduction: that kernel opens the query chain below).
    // ONE event for both consumers: the early reduction on sw and the fork of the query side onto sq
    c.order2(c.s, sw, sq);
    { hipStream_t keep = c.s; c.s = sw; LAUNCH("reduce", launch_reduce(c.ws, io->grads, p.segs_dev, p.blk2seg_dev, p.nblocks_early, p.sq_dev, c.s)); c.s = keep; }
    // data parallel: the predictor block of the gradient bucket is final now -- the caller's all-reduce of it can start (vslnet_hip.h)
    if (!c.dry && io->early_grads_event && cf.predictor == 1) (void)hipEventRecord((hipEvent_t)io->early_grads_event, sw);
    // from here the video side and the query side are independent; the longer one keeps the main stream
    hipStream_t main_s = c.s;
    const bool qlong = query_chain_is_longer(p, false);
    c.s = qlong ? sq : main_s;
    WgradBatch pw_video;
    memset(&pw_video, 0, sizeof pw_video);
    // The step's LAST weight-gradient batch (VisualProjection + the video pass' four pointwise gradients, sample-local query path) starts into an
    // empty chip: chunk rows that make it whole rounds of one-workgroup-per-CU chunks (Dv = 1024: 12 blocks x 32 chunks = 1.5 rounds -> 400 rows,
    // 252 workgroups; profiles/r06_notes.md section 8: -0.6 % against 256-row chunks).
    const int tail_rows = query_fused(c, false) ? wgrad_rows_whole_rounds(R, (cf.video_feature_dim + 127) / 128 + 4, c.h->cus) : WG_ROWS;
    c.pw_rows = tail_rows == WG_ROWS ? 0 : tail_rows;
    enc_bwd(c, P.fe, K.fe, p.ve, c.dry ? nullptr : c.W(p.dC), nullptr, p.dvf, c.dry ? nullptr : io->v_mask, B, 0, sw, &pw_video);
    c.pw_rows = 0;
    {   // tail of the video stream: VisualProjection + the video pass' pointwise weight gradients, back to back, no waits
        WgradBatch wb;
        memset(&wb, 0, sizeof wb);
        WgradJob j = wjob();
        j.rows = tail_rows == WG_ROWS ? 0 : tail_rows;
        const int nchunk = wgrad_chunks(R, tail_rows);      // (shadows the 256-row count: this job's slabs)
        if (!c.dry) {
            j.G[0] = c.W(p.dvf); j.Afull = io->video_features;
            if (io->video_features_bf16) { j.Afull = reinterpret_cast<const float*>(io->video_features_bf16); j.a_bf16 = 1; }
        }
        j.nG = 1; j.nA = 0; j.K = cf.video_feature_dim; j.R = R; j.drop_on_A = 1; j.dp = c.drop(SITE_VIS);
        j.out = c.slab(P.va_w, D * cf.video_feature_dim, nchunk);
        j.out_bias[0] = c.slab(P.va_b, D, nchunk);
        wb.j[wb.n++] = j;
        // sample-local query backward (kernels_query.hip): nothing on the video stream waits for the query chain any more -- the video pass'
        // pointwise weight gradients ride in THIS launch, the query side's own batch goes to the weight-gradient stream behind k_query_bwd.
        // (Measured and dropped: the video pass' pointwise batch on the weight-gradient stream instead, +3.5 %: profiles/r06_notes.md)
        if (query_fused(c, false)) { for (int i = 0; i < pw_video.n; ++i) wb.j[wb.n++] = pw_video.j[i]; }
        LAUNCH("wgrad", launch_wgrad(wb, c.s));
    }
    // ---- query pass, then the embedding stack (all on the other stream)
    c.s = qlong ? main_s : sq;
    LAUNCH("cq_bwd_d", launch_cq_bwd_query(q, B, c.s));     // dQ: only the query side consumes it
    WgradBatch pw_query;
    memset(&pw_query, 0, sizeof pw_query);
    const int EW = cf.word_dim + 100;
    // sample tiles (Lq <= 32): the query pass' conv-block backward goes on with the Embedding linear's data gradient on its rows of dx0
    const bool qfused = query_fused(c, false);
    const bool lin_hosted = qfused || (convblock_bwd_hosts_linear(Rq, Lq) && K.emb_t3_cols % 512 == 0);
    LinTail lt{nullptr, nullptr, EW, K.emb_t3_cols};
    if (lin_hosted && !c.dry) { lt.WT3 = reinterpret_cast<const uint16_t*>(c.PK(K.emb_t3)); lt.dA = c.W(p.dE); }
    if (qfused) query_bwd(c, c.dry ? nullptr : c.W(p.dQtot), p.dqf, pw_query);
    else
    enc_bwd(c, P.fe, K.fe, p.qe, c.dry ? nullptr : c.W(p.dQtot), nullptr, p.dqf, c.dry ? nullptr : io->q_mask, B, 1, sw, &pw_query, false, nullptr, nullptr,
            lin_hosted ? &lt : nullptr);
    {   // every remaining weight gradient (video + query pointwise convs, embedding linear) in ONE launch on the video
        // stream, beside the embedding backward that ends the query stream
        WgradBatch wb_tail = pw_video;
        if (qfused) wb_tail.n = 0;
        WgradJob j = wjob();
        if (!c.dry) { j.G[0] = c.W(p.dqf); j.Afull = c.W(p.E); }
        j.nG = 1; j.nA = 0; j.K = EW; j.R = Rq;
        j.out = c.slab(P.emb_w, D * EW, nchunk_q);
        j.out_bias[0] = c.slab(P.emb_b, D, nchunk_q);
        wb_tail.j[wb_tail.n++] = j;
        const bool fits = wb_tail.n + pw_query.n <= MAX_WJOBS;
        for (int i = 0; fits && i < pw_query.n; ++i) wb_tail.j[wb_tail.n++] = pw_query.j[i];
        hipStream_t qs = c.s, vs = qfused ? sw : (qlong ? sq : main_s);
        c.order(qs, vs);
        c.s = vs;
        LAUNCH("wgrad", launch_wgrad(wb_tail, c.s));
        if (!fits) LAUNCH("wgrad", launch_wgrad(pw_query, c.s));
        c.s = qs;
    }
    if (!lin_hosted)
        LAUNCH("linear_bwd_data", launch_linear_bwd_data3(c.W(p.dqf), reinterpret_cast<const uint16_t*>(c.PK(K.emb_t3)), c.W(p.dE), Rq, EW, K.emb_t3_cols, c.s));
    {
        const int ebc = embed_bwd_chunk(Rq, p.Lc, cf.char_dim), nce = (Rq + ebc - 1) / ebc;
        const int wtot = cf.char_dim * 300;
        const int64_t ow = c.part_alloc((int64_t)nce * wtot), ob = c.part_alloc((int64_t)nce * 100);
        const int ch[4] = {10, 20, 30, 40};
        int wo = 0, bo = 0;
        for (int i = 0; i < 4; ++i) {
            const int wn = ch[i] * cf.char_dim * (i + 1);
            c.reg(P.ccw[i], wn, p.partial + ow + wo, nce, wtot);
            c.reg(P.ccb[i], ch[i], p.partial + ob + bo, nce, 100);
            wo += wn; bo += ch[i];
        }
        float* p_tab = c.slab(P.char_tab, cf.char_size * cf.char_dim, nce);
        float* p_unk = cf.word_table ? nullptr : c.slab(P.unk, cf.word_dim, nce);
        LAUNCH("embed_bwd", launch_embed_bwd(c.W(p.dE), io->word_ids, io->char_ids, c.W(p.E), reinterpret_cast<const int8_t*>(c.W(p.argpos)),
                                c.P(P.char_tab), c.PK(K.ccw_imgb), c.part_ptr(ow), c.part_ptr(ob), p_tab, p_unk, Rq,
                                p.Lc, cf.word_dim, cf.char_dim, cf.char_size, c.drop(SITE_WORD), c.drop(SITE_CHAR), c.s));
    }
    if (cf.word_table && !c.dry)
        LAUNCH("word_table_bwd", launch_word_table_bwd(c.W(p.dE), io->word_ids, io->grads + P.unk, Rq, cf.word_size, cf.word_dim, c.drop(SITE_WORD), c.s));
    // the loss of vsl_io.fused_loss whose seeds the kernels computed themselves: 64 small workgroups behind the QUERY chain's last kernel (the main
    // stream at T <= 128), in the shadow of the join below (at the headline shape that chain ends ~30 us before the video stream's last weight-gradient
    // batch).  On the weight-gradient stream in front of its first batch it took CUs from the chain's 256-workgroup kernels and cost what it saved
    // (r06 notes 10)
    loss_on_side(c);
    c.s = main_s;
    c.order(sq, c.s);                      // join both side streams before the reduction
    c.order(sw, c.s);
    const int nlate = p.nblocks - p.nblocks_early;
    const vsl_fused_step* fs = c.dry ? nullptr : io->fused_step;
    if (!fs) {
        LAUNCH("reduce", launch_reduce(c.ws, io->grads, p.segs_dev, p.blk2seg_dev + 2 * p.nblocks_early, nlate, p.sq_dev + p.nblocks_early, c.s));
        return;
    }
    // the optimizer step inside the call (vsl_io.fused_step): one launch when every gradient leaves the reductions, the two launches otherwise
    const vsl_adamw& hp = fs->hp;
    const float bc1 = (float)(1.0 - std::pow((double)hp.beta1, (double)hp.step)), bc2s = (float)std::sqrt(1.0 - std::pow((double)hp.beta2, (double)hp.step));
    // ONE launch only on request (VSL_FUSED_TAIL=1, read per call): same-box it is level with the two launches (profiles/r06_notes.md section 7), and a
    // launch whose workgroups wait for each other is the wrong default for a process that may be given fewer CUs than the device reports
    const char* ft = getenv("VSL_FUSED_TAIL");
    const bool fuse = ft && atoi(ft) != 0;
    const int grid = fuse && p.sq_cover ? reduce_adamw_grid(nlate, c.h->tail_cap) : 0;
    if (grid > 0) {
        if (++c.h->tail_tag == 0) ++c.h->tail_tag;
        LAUNCH("reduce_adamw", launch_reduce_adamw(c.ws, io->grads, p.segs_dev, p.blk2seg_dev + 2 * p.nblocks_early, nlate, p.sq_dev + p.nblocks_early,
                                                   p.sq_dev, p.blk2seg_dev, p.nblocks_early, grid, c.h->tail_gran, c.h->tail_tag, fs->params,
                                                   fs->exp_avg, fs->exp_avg_sq, c.h->decay_dev, hp.lr, hp.beta1, hp.beta2, hp.eps,
                                                   hp.weight_decay, hp.clip_norm, bc1, bc2s, fs->grad_norm_out, hp.hf_order, c.s));
        return;
    }
    LAUNCH("reduce", launch_reduce(c.ws, io->grads, p.segs_dev, p.blk2seg_dev + 2 * p.nblocks_early, nlate, p.sq_dev + p.nblocks_early, c.s));
    LAUNCH("adamw", launch_adamw(fs->params, io->grads, fs->exp_avg, fs->exp_avg_sq, c.h->decay_dev, c.h->opt_scratch, c.h->param_floats, hp.lr,
                                 hp.beta1, hp.beta2, hp.eps, hp.weight_decay, hp.clip_norm, bc1, bc2s, fs->grad_norm_out, c.s, hp.hf_order,
                                 p.sq_cover ? p.sq_dev : nullptr, p.nblocks));
}

int build_plan(vsl_handle_s* h, int B, int T, int Lq, int Lc, Plan** out) {
    const vsl_config& cf = h->cfg;
    if (B <= 0 || T <= 0 || Lq <= 0) return fail("empty batch (B=%d T=%d Lq=%d)", B, T, Lq);
    if (T > cf.max_pos_len || Lq > cf.max_pos_len)
        return fail("sequence length (T=%d, Lq=%d) exceeds max_pos_len=%d: the positional table has no such row "
                    "(layers_t7.py:196 -- IndexError in the reference)", T, Lq, cf.max_pos_len);
    if (T > MAX_L) return fail("T=%d > %d clips not supported by the LDS-resident attention kernels", T, MAX_L);
    if ((int64_t)B * std::max(T, Lq) * D * 4 >= (int64_t(1) << 31))     // (the conv-block kernels address (R, 128) tensors with 32-bit byte offsets; 0x80000000 marks a dropped lane)
        return fail("B*T = %lld rows: an (R, 128) fp32 activation must stay below 2 GiB for the 32-bit buffer offsets of the conv-block kernels", (long long)B * std::max(T, Lq));
    if (Lq > MAX_LQ) return fail("Lq=%d > %d query words: the CQAttention kernels keep the whole query of a sample in LDS "
                                 "(the reference truncates at max_pos_len words, data_gen.py:188)", Lq, MAX_LQ);
    if (Lc < 4 || Lc > MAX_LC) return fail("Lc=%d must be in [4, %d] (the widest char conv has kernel 4, layers_t7.py:52)", Lc, MAX_LC);
    Plan* p = new Plan();
    p->B = B; p->T = T; p->Lq = Lq; p->Lc = Lc;
    const int64_t R = (int64_t)B * T, Rq = (int64_t)B * Lq;
    const int H = cf.num_heads, EW = cf.word_dim + 100;
    Bump al;
    p->pack = al(h->pack_floats);
    p->vf = al(R * D); p->E = al(Rq * EW); p->argpos = al((Rq * 100 + 3) / 4); p->qf = al(Rq * D);
    plan_encoder(al, p->ve, B, T, H); plan_encoder(al, p->qe, B, Lq, H);
    const bool rnn = cf.predictor == 0;
    if (rnn) {
        if (R * 4 * D >= (int64_t(1) << 31)) { delete p; return fail("B*T = %lld too large for the 32-bit offsets of the LSTM kernels", (long long)R); }
        p->p1.R = p->p2.R = (int)R; p->p1.L = p->p2.L = T;
        for (int l = 0; l < 2; ++l) {
            LstmWs& w = p->lstm[l];
            w.gi = al(R * 4 * D); w.gates = al(R * 4 * D); w.cseq = al(R * D); w.tseq = al(R * D); w.hprev = al(R * D); w.out = al(R * D); w.dG = al(R * 4 * D);
            w.carry = al((int64_t)B * 2 * D);
        }
        p->p1.out = p->lstm[0].out; p->p2.out = p->lstm[1].out;
        if (rnn_fused_ok(B)) {
            p->h_gran = al(R * 2 * D); p->gi_gran = al(R * 8 * D); p->dg_gran = al(R * 8 * D); p->dx_gran = al(R * 2 * D);
            p->gran_floats = (p->dx_gran + R * 2 * D) - p->h_gran;      // the four allocations are consecutive
        }
    } else {
        plan_encoder(al, p->p1, B, T, H); plan_encoder(al, p->p2, B, T, H);
    }
    p->S = al(R * Lq); p->Srow = al(R * Lq); p->Scol = al(R * Lq); p->M = al(Rq * D); p->alpha = al(Rq);
    p->pooled = al((int64_t)B * D); p->pb = al((int64_t)B * D); p->cat = al(R * 4 * D);
    p->f1 = al(R * D); p->f2 = al(R * D); p->gated = al(R * D);
    p->hid_s = al(R * D); p->hid_e = al(R * D); p->lnf_s = al(R * D); p->lnf_e = al(R * D);
    p->loss_scratch = al(5 * (int64_t)B + 8);
    p->gz_s = al(R * D); p->gz_e = al(R * D); p->dfeat_s = al(R * D); p->dfeat_e = al(R * D);
    p->dxh_s = al(R * D); p->dxh_e = al(R * D); p->g_s1 = al(R * D); p->g_gated = al(R * D);
    for (int ap = 0; ap < (rnn ? 2 : 4); ++ap) {
        const int64_t Ra = ap == 1 ? Rq : R;
        EncTmp& t = p->tmp[ap];
        t.dr = al(Ra * D); t.dq = al(Ra * D * attn_bwd_dq_slabs(ap == 1 ? Lq : T)); t.dk = al(Ra * D); t.dv = al(Ra * D);
        t.go = al(Ra * D);
        for (int i = 0; i < 4; ++i) t.gz[i] = al(Ra * D);
        t.ga = al(Ra * D);
    }
    p->df2 = al(R * D); p->df1 = al(R * D); p->dC = al(R * D);
    {
        const int64_t nt = (T + TILE_M - 1) / TILE_M;
        p->cqP1 = al((int64_t)B * nt * 2 * Lq * D); p->cqP2 = al((int64_t)B * nt * Lq); p->cqP3 = al((int64_t)B * nt * Lq);
        p->cqP4 = al((int64_t)B * nt * Lq * D); p->cqP5 = al((int64_t)B * nt * D);
    }
    p->dSr = al(R * Lq); p->dSs = al(R * Lq); p->dQtot = al(Rq * D); p->dvf = al(R * D); p->dqf = al(Rq * D); p->dE = al(Rq * EW);
    p->partial = al(0);
    // dry-run the backward to lay out the partial arena and collect the reduction table
    std::vector<SlabRec> recs;
    Ctx c{h, p, nullptr, nullptr, true, nullptr};
    c.recs = &recs;
    run_backward(c);
    p->partial_floats = c.part_cur;
    al(p->partial_floats);
    p->total = al.cur;
    if (p->total >= (int64_t(1) << 31)) { delete p; return fail("workspace of %lld floats exceeds the 2^31 offset range of the reduction table", (long long)p->total); }
    // group the records by destination (shared weights are written by up to 4 encoder applications)
    std::vector<ReduceSeg> segs;
    std::map<int, int> by_dst;
    for (const SlabRec& r : recs) {
        auto it = by_dst.find(r.dst);
        if (it == by_dst.end()) {
            ReduceSeg s;
            memset(&s, 0, sizeof s);
            s.dst = r.dst; s.n = r.n; s.rl = r.rl; s.ds = r.ds;
            by_dst[r.dst] = (int)segs.size();
            segs.push_back(s);
            it = by_dst.find(r.dst);
        }
        ReduceSeg& s = segs[it->second];
        if (s.n != r.n || s.nsrc >= 4) { delete p; return fail("internal: inconsistent partial slabs for param offset %d", r.dst); }
        s.src[s.nsrc] = (int)r.src; s.nslabs[s.nsrc] = r.nslabs; s.ss[s.nsrc] = r.ss; s.vn[s.nsrc] = r.vn;
        ++s.nsrc;
    }
    for (ReduceSeg& s : segs) {
        s.vec = (s.n % 4 == 0) && (s.rl % 4 == 0) && (s.ds % 4 == 0) && (s.dst % 4 == 0);
        for (int q = 0; q < s.nsrc; ++q) s.vec = s.vec && (s.src[q] % 4 == 0) && (s.ss[q] % 4 == 0) && (s.vn[q] % 4 == 0);
    }
    if (getenv("VSL_DEBUG_PLAN")) {
        int64_t tot = 0;
        for (const ReduceSeg& s : segs) {
            int64_t fl = 0; int ns = 0;
            for (int q = 0; q < s.nsrc; ++q) { fl += (int64_t)s.nslabs[q] * s.vn[q]; ns += s.nslabs[q]; }
            tot += fl;
            if (fl * 4 >= (256 << 10)) fprintf(stderr, "[plan] dst %8d n %7d nsrc %d slabs %5d  read %8.2f MiB\n", s.dst, s.n, s.nsrc, ns, fl * 4.0 / (1 << 20));
        }
        fprintf(stderr, "[plan] %zu segments, %.2f MiB of partial slabs read per step, partial arena %.2f MiB\n", segs.size(), tot * 4.0 / (1 << 20), p->partial_floats * 4.0 / (1 << 20));
    }
    // early / late split by destination: everything up to the end of the shared feature encoder's parameters is `late`
    // plus what k_cq_bwd_d produces on the query stream after the fork (w4Q, pooled-query weight, second half of / bias of
    // the CQConcatenate projection)
    const int late_end = h->P.w4C;       // params are laid out [embedding | visual | feature_encoder | cq... | predictor...]
    const int late_q[4] = {h->P.w4Q, h->P.pool_w, h->P.cat_b, h->P.cat_w + D};
    std::vector<int> blk;
    for (int pass = 0; pass < 2; ++pass) {
        for (size_t i = 0; i < segs.size(); ++i) {
            bool late = segs[i].dst < late_end;
            for (int d : late_q) late = late || segs[i].dst == d;
            if (late != (pass == 1)) continue;
            for (int o = 0; o < segs[i].n; o += 256) { blk.push_back((int)i); blk.push_back(o); }
        }
        if (pass == 0) p->nblocks_early = (int)blk.size() / 2;
    }
    p->nblocks = (int)blk.size() / 2;
    HIP_OK(hipMalloc(&p->segs_dev, segs.size() * sizeof(ReduceSeg)));
    HIP_OK(hipMalloc(&p->blk2seg_dev, blk.size() * sizeof(int)));
    HIP_OK(hipMemcpy(p->segs_dev, segs.data(), segs.size() * sizeof(ReduceSeg), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(p->blk2seg_dev, blk.data(), blk.size() * sizeof(int), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc(&p->sq_dev, (size_t)std::max(p->nblocks, 1) * sizeof(float)));
    {   // do the reduction blocks write every parameter gradient exactly once?  (then their sums of squares ARE the bucket's)
        std::vector<uint8_t> hit((size_t)h->param_floats, 0);
        bool once = true;
        for (const ReduceSeg& sg : segs)
            for (int i = 0; i < sg.n; ++i) {
                const int64_t d = (int64_t)sg.dst + (int64_t)(i / sg.rl) * sg.ds + (i % sg.rl);
                if (d < 0 || d >= h->param_floats || hit[(size_t)d]++) once = false;
            }
        for (const ParamInfo& pi : h->params)
            for (int64_t e = pi.off; e < pi.off + pi.numel; ++e) once = once && hit[(size_t)e] == 1;
        p->sq_cover = once;
    }
    *out = p;
    return 0;
}

void free_plan(Plan* p) {
    if (!p) return;
    if (p->segs_dev) (void)hipFree(p->segs_dev);
    if (p->blk2seg_dev) (void)hipFree(p->blk2seg_dev);
    if (p->sq_dev) (void)hipFree(p->sq_dev);
    delete p;
}

// Per-shape plans (workspace layout + reduction tables).  The collate narrows every batch to its own max T / Lq / Lc, so a
// long training run visits hundreds of shapes (Charades at batch 16: T in 40..128 x Lq in 4..10 x Lc in 5..12).  A plan is two small
// device tables (< 100 KB), a miss costs a rebuild, two hipMalloc and -- on eviction -- a device synchronise, so the cache holds every
// shape a realistic run sees (64 entries thrashed under shuffled real-data batches) and evicts the least recently used one
// beyond that.  (A forward and its backward use the same shape back to back, so a live plan is never evicted.)
constexpr size_t MAX_PLANS = 2048;
int get_plan(vsl_handle_s* h, int B, int T, int Lq, int Lc, Plan** out) {
    auto key = std::make_tuple(B, T, Lq, Lc);
    auto it = h->plans.find(key);
    if (it != h->plans.end()) { it->second->last_use = ++h->plan_clock; *out = it->second; return 0; }
    Plan* p = nullptr;
    if (int rc = build_plan(h, B, T, Lq, Lc, &p)) return rc;
    if (h->plans.size() >= MAX_PLANS) {
        auto victim = h->plans.begin();
        for (auto jt = h->plans.begin(); jt != h->plans.end(); ++jt)
            if (jt->second->last_use < victim->second->last_use) victim = jt;
        (void)hipDeviceSynchronize();            // its tables may still be read by an enqueued reduction
        if (h->sq_src == victim->second->sq_dev) { h->sq_src = nullptr; h->sq_grads = nullptr; }
        free_plan(victim->second);
        h->plans.erase(victim);
    }
    p->last_use = ++h->plan_clock;
    h->plans[key] = p;
    *out = p;
    return 0;
}

int check_io(vsl_handle_s* h, const vsl_io* io) {
    if (!h || !io) return fail("null handle / io");
    if (!io->params || (!h->cfg.word_table && (!io->pad_vec || !io->glove_vec)) || !io->word_ids || !io->char_ids || (!io->video_features && !io->video_features_bf16) ||
        !io->v_mask || !io->q_mask || !io->h_score || !io->start_logits || !io->end_logits || !io->workspace)
        return fail("vsl_io has a null device pointer");
    if (io->video_features_bf16 && (h->cfg.video_feature_dim % 8 != 0))
        return fail("bf16 features need video_feature_dim %% 8 == 0 (got %d)", h->cfg.video_feature_dim);
    return 0;
}

}  // namespace

namespace vsl {
void vsl_launch_events(hipStream_t s, hipEvent_t* start, hipEvent_t* stop) {
    vsl_handle_s* h = g_cur;
    if (!h || !(h->stop_events || h->prof_name)) return;
    hipEvent_t e;
    if (h->prof_name) {                      // profiled launch: timing events, start + stop, one record per kernel
        while (h->prof_pool.size() < h->prof_used + 2) { hipEvent_t t; (void)hipEventCreate(&t); h->prof_pool.push_back(t); }
        {
            vsl_handle_s::ProfRec rec{h->prof_name, h->prof_used, h->prof_used + 1, s,
                                      std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h->prof_t0).count(), {0, 0, 0, 0, 0, 0}, 0};
            const int me = (int)h->prof_recs.size();
            bool seen = false;
            for (auto& kv : h->prof_last) if (kv.first == s) { if (rec.nd < 6) rec.deps[rec.nd++] = kv.second; kv.second = me; seen = true; }
            if (!seen) h->prof_last.push_back({s, me});
            for (size_t i = 0; i < h->prof_pending.size();) {
                if (h->prof_pending[i].first == s) { if (rec.nd < 6) rec.deps[rec.nd++] = h->prof_pending[i].second; h->prof_pending.erase(h->prof_pending.begin() + i); }
                else ++i;
            }
            h->prof_recs.push_back(rec);
        }
        *start = h->prof_pool[h->prof_used];
        e = h->prof_pool[h->prof_used + 1];
        h->prof_used += 2;
    } else {
        if (h->stop_used == h->stop_pool.size()) { hipEvent_t t; (void)hipEventCreateWithFlags(&t, hipEventDisableTiming); h->stop_pool.push_back(t); }
        e = h->stop_pool[h->stop_used++];
    }
    *stop = e;
    if (h->stop_events) {                    // remember the stream's last kernel for Ctx::order
        int i = 0;
        while (i < h->ev_n && h->ev_stream[i] != s) ++i;
        if (i == h->ev_n) { if (h->ev_n == 4) return; h->ev_stream[h->ev_n++] = s; }
        h->ev_last[i] = e;
        h->ev_dirty[i] = false;              // the new kernel runs behind every wait enqueued before it
    }
}
}  // namespace vsl

// =================================================================================================== C ABI
extern "C" {

const char* vsl_last_error(void) { return g_err.c_str(); }

int vsl_create(const vsl_config* cfg, vsl_handle* out) {
    if (!cfg || !out) return fail("null argument");
    if (cfg->dim != D) return fail("configs.dim=%d: the HIP kernels are specialised for dim=128", cfg->dim);
    if (cfg->num_heads <= 0 || cfg->dim % cfg->num_heads != 0)
        return fail("The channels (%d) is not a multiple of attention heads (%d)", cfg->dim, cfg->num_heads);   // layers_t7.py:146
    if (cfg->dim / cfg->num_heads != HD) return fail("num_heads=%d: the attention kernels are specialised for head size 16 (8 heads)", cfg->num_heads);
    if (cfg->predictor != 0 && cfg->predictor != 1) return fail("predictor must be 0 ('rnn') or 1 ('transformer'), got %d", cfg->predictor);
    if (cfg->video_feature_dim <= 0 || cfg->video_feature_dim % 4) return fail("video_feature_dim=%d must be a positive multiple of 4 (rows are read as float4)", cfg->video_feature_dim);
    if ((cfg->word_dim + 100) % 8) return fail("word_dim + 100 = %d must be a multiple of 8", cfg->word_dim + 100);
    if (cfg->word_table != 0 && cfg->word_table != 1) return fail("word_table must be 0 (frozen vectors + unk_vec) or 1 (trainable table), got %d", cfg->word_table);
    if (cfg->word_table && (cfg->word_dim > 512 || cfg->word_size < 2)) return fail("trainable word table: word_dim=%d must be <= 512 and word_size=%d >= 2", cfg->word_dim, cfg->word_size);
    if (cfg->char_dim <= 0 || cfg->char_dim > 128) return fail("char_dim=%d must be in [1, 128]", cfg->char_dim);
    if (cfg->char_size <= 0 || cfg->char_size * cfg->char_dim > 65536) return fail("char table of %d x %d floats: every embedding-backward workgroup writes one partial copy, 65536 floats is the limit", cfg->char_size, cfg->char_dim);
    if (cfg->max_pos_len <= 0 || cfg->word_size < 2) return fail("bad max_pos_len / word_size");
    if (cfg->drop_rate < 0.f || cfg->drop_rate >= 1.f) return fail("drop_rate must be in [0, 1)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device available: libvslnet_hip needs an MI355X (gfx950)");
    vsl_handle_s* h = new vsl_handle_s();
    h->cfg = *cfg;
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&h->cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) h->cus = 0;
    }
    build_params(h);
    build_packs(h);
    {
        // Drop the pack jobs whose operand no kernel reads: the fp32 packs of the GEMMs that run as split products and the split packs of the
        // attention output projection (it stays on the fp32-input MFMA).
        // The regions stay in the pack buffer (offsets are fixed); k_pack just does a third less work every step.
        const ModelPk& K = h->K;
        std::vector<int> dead;
        auto enc = [&](const EncPk& e) {
            if (&e != &K.fe) { dead.push_back(e.o_f3); dead.push_back(e.o_t3); }      // (the feature encoder's: read by the sample-local query kernels, kernels_query.hip)
            for (int i = 0; i < 4; ++i) { dead.push_back(e.pw_f[i]); dead.push_back(e.pw_t[i]); }
            dead.push_back(e.qkv_f); dead.push_back(e.qkv_t);
        };
        enc(K.fe);
        if (cfg->predictor != 0) enc(K.pe);
        dead.push_back(K.va_f); dead.push_back(K.emb_t);
        if ((cfg->word_dim + 100) % 16 == 0) dead.push_back(K.emb_f);
        if (cfg->predictor == 0) for (int l = 0; l < 2; ++l) dead.push_back(K.l_t[l]);        // the rnn head's dx GEMM reads the split pack l_t3
        h->jobs.erase(std::remove_if(h->jobs.begin(), h->jobs.end(), [&](const PackJob& j) {
                          return std::find(dead.begin(), dead.end(), j.dst) != dead.end(); }), h->jobs.end());
    }
    if (hipMalloc(&h->loss_counter, sizeof(unsigned)) != hipSuccess || hipMemset(h->loss_counter, 0, sizeof(unsigned)) != hipSuccess) {
        delete h;
        return fail("hipMalloc of the loss counter failed");
    }
    {   // Three pack launches per forward (run_forward): what VisualProjection and the shared feature encoder's forward read goes first on the
        // main stream, the embedding stack's operands on the query stream, everything else (CQ fusion, heads, predictor encoder, every
        // data-gradient pack) on the weight-gradient stream, which is idle in the forward -- the step no longer opens with ONE 11 us launch
        // that packs 2.7 MB before its first kernel may start.
        const ModelPk& K = h->K;
        auto in = [](int dst, std::initializer_list<int> set) { return std::find(set.begin(), set.end(), dst) != set.end(); };
        auto cls = [&](const PackJob& j) {
            const EncPk& f = K.fe;
            if (j.dst == K.ccw_img) return 0;
            if (in(j.dst, {K.va_f, K.va_f16, K.va_f3, f.pw_f[0], f.pw_f[1], f.pw_f[2], f.pw_f[3], f.qkv_f, f.o_f, f.pw_f3[0], f.pw_f3[1], f.pw_f3[2], f.pw_f3[3],
                           f.qkv_f3, f.o_f3})) return 1;
            if (in(j.dst, {K.emb_f, K.emb_f3})) return 2;
            return 3;
        };
        std::stable_sort(h->jobs.begin(), h->jobs.end(), [&](const PackJob& a, const PackJob& b) { return cls(a) < cls(b); });
        for (const PackJob& j : h->jobs) { h->jobs_char += cls(j) == 0; h->jobs_first += cls(j) == 1; h->jobs_query += cls(j) == 2; }
    }
    if (hipMalloc(&h->jobs_dev, h->jobs.size() * sizeof(PackJob)) != hipSuccess ||
        hipMemcpy(h->jobs_dev, h->jobs.data(), h->jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice) != hipSuccess) {
        delete h;
        return fail("hipMalloc/hipMemcpy of the pack table failed");
    }
    {
        const char* e = getenv("VSL_MULTI_STREAM");
        h->multi_stream = !(e && e[0] == '0');
        if (h->multi_stream)
            for (int k = 0; k < 2; ++k) {
                // side(1) carries the weight gradients: nothing waits for them until the final reduction, so it runs at the
                // lowest priority the device offers and leaves CUs to the dependent chain
                int lo = 0, hi = 0;
                (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
                const int prio = k == 1 ? lo : 0;       // (the query stream's priority makes no difference either way: profiles/r03_notes.md)
                if (hipStreamCreateWithPriority(&h->side[k], hipStreamNonBlocking, prio) != hipSuccess) { h->side[k] = nullptr; (void)hipGetLastError(); }
            }
    }
    *out = h;
    return 0;
}

int vsl_destroy(vsl_handle h) {
    if (!h) return 0;
    for (int k = 0; k < 2; ++k) if (h->side[k]) (void)hipStreamDestroy(h->side[k]);
    for (hipEvent_t e : h->sync_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->stop_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->prof_pool) (void)hipEventDestroy(e);
    for (auto& kv : h->plans) {
        if (kv.second->segs_dev) (void)hipFree(kv.second->segs_dev);
        if (kv.second->blk2seg_dev) (void)hipFree(kv.second->blk2seg_dev);
        if (kv.second->sq_dev) (void)hipFree(kv.second->sq_dev);
        delete kv.second;
    }
    if (h->jobs_dev) (void)hipFree(h->jobs_dev);
    if (h->loss_counter) (void)hipFree(h->loss_counter);
    if (h->decay_dev) (void)hipFree(h->decay_dev);
    if (h->opt_scratch) (void)hipFree(h->opt_scratch);
    if (h->tail_gran) (void)hipFree(h->tail_gran);
    delete h;
    return 0;
}

int vsl_param_count(vsl_handle h) { return h ? (int)h->params.size() : 0; }
int64_t vsl_param_floats(vsl_handle h) { return h ? h->param_floats : 0; }
int vsl_param_info(vsl_handle h, int index, char* name, int name_cap, int64_t* offset, int64_t* numel, int32_t* ndim,
                   int64_t dims[4]) {
    if (!h || index < 0 || index >= (int)h->params.size()) return fail("bad parameter index");
    const ParamInfo& p = h->params[index];
    if (name && name_cap > 0) { strncpy(name, p.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (offset) *offset = p.off;
    if (numel) *numel = p.numel;
    if (ndim) *ndim = p.ndim;
    if (dims) for (int i = 0; i < 4; ++i) dims[i] = i < p.ndim ? p.dims[i] : 1;
    return 0;
}

int vsl_workspace_floats(vsl_handle h, int B, int T, int Lq, int Lc, int64_t* out) {
    if (!h || !out) return fail("null argument");
    Plan* p = nullptr;
    if (int rc = get_plan(h, B, T, Lq, Lc, &p)) return rc;
    *out = p->total;
    return 0;
}

int64_t vsl_workspace_offset(vsl_handle h, int B, int T, int Lq, int Lc, const char* name) {
    Plan* p = nullptr;
    if (!h || !name || get_plan(h, B, T, Lq, Lc, &p)) return -1;
    const std::string n = name;
    const std::pair<const char*, int64_t> tab[] = {
        {"video_affine", p->vf}, {"embedding_net", p->qf}, {"emb_concat", p->E}, {"venc", p->ve.out}, {"qenc", p->qe.out},
        {"venc_x0", p->ve.x0}, {"venc_conv0", p->ve.y[0]}, {"venc_conv3", p->ve.y[3]}, {"venc_q", p->ve.q}, {"venc_k", p->ve.k},
        {"venc_v", p->ve.v}, {"venc_att", p->ve.att}, {"venc_r", p->ve.r}, {"qenc_conv3", p->qe.y[3]}, {"qenc_att", p->qe.att},
        {"cq_score", p->S}, {"cq_srow", p->Srow}, {"cq_scol", p->Scol}, {"cq_M", p->M}, {"cq_attention", p->f1},
        {"cq_concat", p->f2}, {"gated", p->gated}, {"pred_s", p->p1.out}, {"pred_e", p->p2.out},
        {"d_gated_enc", p->g_gated}, {"d_gated_hs", p->dxh_s}, {"d_gated_he", p->dxh_e}, {"d_venc", p->dC}, {"d_qenc", p->dQtot}, {"d_video_affine", p->dvf}, {"d_embedding_net", p->dqf},
        {"d_pred_s", p->g_s1}, {"d_pred_s_head", p->dfeat_s}, {"d_cq_concat", p->df2}, {"d_cq_attention", p->df1}, {"d_emb_concat", p->dE}};
    for (auto& kv : tab) if (n == kv.first) return kv.second;
    // ReLU decisions of the forward (parity tests count sign flips against the oracle): "relu_<enc>_<layer>" = (R, 4) uint32
    // bit-masks of conv layer 0..3 of venc / qenc / p1 / p2 ; "hid_s" / "hid_e" = post-ReLU activations of the span heads
    if (n == "hid_s") return p->hid_s;
    if (n == "hid_e") return p->hid_e;
    const std::pair<const char*, const EncWs*> encs[] = {{"relu_venc_", &p->ve}, {"relu_qenc_", &p->qe}, {"relu_p1_", &p->p1}, {"relu_p2_", &p->p2}};
    // "<enc>_y<layer>": output of conv layer 0..3 of that pass (R, 128) ; "<enc>_x0": its input + positional rows
    const std::pair<const char*, const EncWs*> acts[] = {{"venc_y", &p->ve}, {"qenc_y", &p->qe}, {"p1_y", &p->p1}, {"p2_y", &p->p2}};
    for (auto& kv : acts) {
        const size_t len = strlen(kv.first);
        if (h->cfg.predictor == 0 && (kv.second == &p->p1 || kv.second == &p->p2)) continue;
        if (n.size() == len + 1 && n.compare(0, len, kv.first) == 0 && n[len] >= '0' && n[len] <= '3') return kv.second->y[n[len] - '0'];
        if (n.size() == len + 1 && n.compare(0, len - 1, kv.first, len - 1) == 0 && n.compare(len - 1, 2, "x0") == 0) return kv.second->x0;
    }
    for (auto& kv : encs) {
        const size_t len = strlen(kv.first);
        if (n.size() == len + 1 && n.compare(0, len, kv.first) == 0 && n[len] >= '0' && n[len] <= '3') {
            if (h->cfg.predictor == 0 && (kv.second == &p->p1 || kv.second == &p->p2)) return -1;
            return kv.second->mask[n[len] - '0'];
        }
    }
    return -1;
}

int vsl_forward(vsl_handle h, const vsl_io* io, void* hip_stream) {
    if (int rc = check_io(h, io)) return rc;
    Plan* p = nullptr;
    if (int rc = get_plan(h, io->B, io->T, io->Lq, io->Lc, &p)) return rc;
    Ctx c{h, p, io, (hipStream_t)hip_stream, false, io->workspace};
    c.main = c.s;
    CallScope scope(h);
    run_forward(c);
    HIP_OK(hipGetLastError());
    return 0;
}

int vsl_loss(vsl_handle h, const vsl_io* io, const vsl_loss_io* l, void* hip_stream) {
    if (!h || !io || !l) return fail("null argument");
    if (!l->start_labels || !l->end_labels || !l->h_labels || !l->losses) return fail("vsl_loss_io has a null pointer");
    if ((l->d_h_score == nullptr) != (l->d_start_logits == nullptr) || (l->d_h_score == nullptr) != (l->d_end_logits == nullptr))
        return fail("gradient seed outputs must be all set or all null");
    Plan* p = nullptr;
    if (int rc = get_plan(h, io->B, io->T, io->Lq, io->Lc, &p)) return rc;
    ProfScope ps(h, "loss");
    launch_loss(io->start_logits, io->end_logits, io->h_score, l->start_labels, l->end_labels, l->h_labels, io->v_mask, io->B,
                io->T, l->inv_batch, l->mask_sum, l->w_loc, l->w_highlight, io->workspace + p->loss_scratch, l->losses,
                l->d_start_logits, l->d_end_logits, l->d_h_score, (hipStream_t)hip_stream, h->loss_counter);
    HIP_OK(hipGetLastError());
    return 0;
}

// the optimizer's device-side constants (first use): decay flags, scratch of the two-kernel step, granules of the fused tail
static int ensure_opt_state(vsl_handle_s* h) {
    if (h->decay_dev) return 0;
    std::vector<uint8_t> mask((size_t)h->param_floats, 0);          // VSLNet_t7.py:9-13: no decay for bias / layer_norm / LayerNorm parameters
    for (const ParamInfo& p : h->params) {
        const bool nd = p.name.find("bias") != std::string::npos || p.name.find("layer_norm") != std::string::npos ||
                        p.name.find("LayerNorm") != std::string::npos;
        if (!nd) std::fill(mask.begin() + p.off, mask.begin() + p.off + p.numel, (uint8_t)1);
    }
    h->tail_cap = reduce_adamw_resident(h->cus);
    if (hipMalloc(&h->decay_dev, mask.size()) != hipSuccess || hipMalloc(&h->opt_scratch, (OPT_BLOCKS + 4) * sizeof(float)) != hipSuccess ||
        hipMalloc(&h->tail_gran, (size_t)h->tail_cap * sizeof(unsigned long long)) != hipSuccess)
        return fail("optimizer state: hipMalloc failed");
    if (hipMemcpy(h->decay_dev, mask.data(), mask.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(h->tail_gran, 0, (size_t)h->tail_cap * sizeof(unsigned long long)) != hipSuccess)      // tag 0 is never a launch's tag
        return fail("optimizer state: hipMemcpy failed");
    return 0;
}

int vsl_backward(vsl_handle h, const vsl_io* io, void* hip_stream) {
    if (int rc = check_io(h, io)) return rc;
    if (const vsl_loss_io* l = io->fused_loss) {
        if (!l->start_labels || !l->end_labels || !l->h_labels || !l->losses) return fail("vsl_backward: fused_loss has a null pointer");
        if (!l->d_start_logits || !l->d_end_logits || !l->d_h_score) return fail("vsl_backward: fused_loss needs the three gradient seed outputs");
        if (!io->grads) return fail("vsl_backward needs grads");
    } else
    if (!io->d_start_logits || !io->d_end_logits || !io->grads) return fail("vsl_backward needs d_start_logits, d_end_logits and grads");
    Plan* p = nullptr;
    if (int rc = get_plan(h, io->B, io->T, io->Lq, io->Lc, &p)) return rc;
    if (const vsl_fused_step* fs = io->fused_step) {
        if (!fs->params || !fs->exp_avg || !fs->exp_avg_sq) return fail("vsl_backward: fused_step needs params, exp_avg and exp_avg_sq");
        if (fs->hp.step < 1) return fail("vsl_backward: fused_step.hp.step must be >= 1 (got %d)", fs->hp.step);
        if (io->early_grads_event) return fail("vsl_backward: fused_step (single process) and early_grads_event (data parallel) exclude each other");
        if (int rc = ensure_opt_state(h)) return rc;
    }
    Ctx c{h, p, io, (hipStream_t)hip_stream, false, io->workspace};
    c.main = c.s;
    CallScope scope(h);
    run_backward(c);
    HIP_OK(hipGetLastError());
    h->sq_src = p->sq_cover ? p->sq_dev : nullptr; h->sq_n = p->nblocks; h->sq_grads = io->grads;
    return 0;
}

int vsl_abi_version(void) { return VSL_ABI_VERSION; }

uint64_t vsl_debug_rnn_launches(uint64_t n) { return (uint64_t)g_rnn_launches.exchange((unsigned long long)n); }

int64_t vsl_early_grad_offset(vsl_handle h) {
    if (!h) return -1;
    if (h->cfg.predictor != 1) return h->param_floats;
    int64_t off = h->param_floats;
    for (const ParamInfo& pi : h->params)
        if (pi.name.rfind("predictor.", 0) == 0) off = std::min<int64_t>(off, pi.off);
    return off;
}

int vsl_extract_index(vsl_handle h, const float* start_logits, const float* end_logits, int B, int T, int64_t* start_index,
                      int64_t* end_index, void* hip_stream) {
    if (!h || !start_logits || !end_logits || !start_index || !end_index) return fail("null argument");
    if (T > 8192) return fail("T too large");
    launch_extract_index(start_logits, end_logits, start_index, end_index, B, T, (hipStream_t)hip_stream);
    HIP_OK(hipGetLastError());
    return 0;
}

int vsl_adamw_step(vsl_handle h, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const vsl_adamw* hp,
                   float* grad_norm_out, void* hip_stream) {
    if (!h || !params || !grads || !exp_avg || !exp_avg_sq || !hp) return fail("vsl_adamw_step: null argument");
    if (hp->step < 1) return fail("vsl_adamw_step: step must be >= 1 (got %d)", hp->step);
    if (int rc = ensure_opt_state(h)) return rc;
    const double bc1 = 1.0 - std::pow((double)hp->beta1, (double)hp->step), bc2 = 1.0 - std::pow((double)hp->beta2, (double)hp->step);
    const float* sq = nullptr;
    if (hp->norm_from_backward) {
        if (h->sq_grads != grads) return fail("vsl_adamw_step: norm_from_backward is set but `grads` is not the bucket the last vsl_backward wrote");
        sq = h->sq_src;             // (null when some gradient of this configuration does not leave k_reduce: the k_sqsum pass then)
    }
    ProfScope ps(h, "adamw");
    launch_adamw(params, grads, exp_avg, exp_avg_sq, h->decay_dev, h->opt_scratch, h->param_floats, hp->lr, hp->beta1, hp->beta2,
                 hp->eps, hp->weight_decay, hp->clip_norm, (float)bc1, (float)std::sqrt(bc2), grad_norm_out, (hipStream_t)hip_stream,
                 hp->hf_order, sq, h->sq_n);
    if (hipGetLastError() != hipSuccess) return fail("vsl_adamw_step: launch failed");
    return 0;
}

int vsl_profile_select(vsl_handle h, const char* kernel) {
    if (!h) return fail("null handle");
    h->prof_on = kernel != nullptr && kernel[0] != 0;
    h->prof_sel = kernel ? kernel : "";
    h->prof_recs.clear();
    h->prof_used = 0;
    h->prof_last.clear(); h->prof_pending.clear();
    h->prof_t0 = std::chrono::steady_clock::now();
    return 0;
}

// one record of the ledger: the index-th profiled launch since vsl_profile_select, in enqueue order
int vsl_profile_launch(vsl_handle h, int index, char* name, int name_cap, int32_t* stream, double* start_us, double* stop_us, double* host_us,
                       int32_t* deps /* [6] */, int32_t* ndeps) {
    if (!h) return fail("null handle");
    if (index < 0 || index >= (int)h->prof_recs.size()) return 2;
    const auto& r = h->prof_recs[index];
    float t0 = 0.f, t1 = 0.f;
    const hipEvent_t ref = h->prof_pool[h->prof_recs[0].e0];
    if (hipEventSynchronize(h->prof_pool[r.e1]) != hipSuccess || hipEventElapsedTime(&t0, ref, h->prof_pool[r.e0]) != hipSuccess ||
        hipEventElapsedTime(&t1, ref, h->prof_pool[r.e1]) != hipSuccess)
        return fail("hipEventElapsedTime failed");
    if (name && name_cap > 0) { strncpy(name, r.name, name_cap - 1); name[name_cap - 1] = 0; }
    int si = 0;
    {   // stream index in order of first appearance
        std::vector<hipStream_t> seen;
        for (const auto& q : h->prof_recs) { if (std::find(seen.begin(), seen.end(), q.s) == seen.end()) seen.push_back(q.s); if (&q == &r) break; }
        si = (int)(std::find(seen.begin(), seen.end(), r.s) - seen.begin());
    }
    if (stream) *stream = si;
    if (start_us) *start_us = 1e3 * t0;
    if (stop_us) *stop_us = 1e3 * t1;
    if (host_us) *host_us = r.host_us;
    if (deps) for (int i = 0; i < 6; ++i) deps[i] = i < r.nd ? r.deps[i] : -1;
    if (ndeps) *ndeps = r.nd;
    return 0;
}

int vsl_profile_read(vsl_handle h, int index, char* name, int name_cap, double* total_ms, int32_t* count) {
    if (!h) return fail("null handle");
    // aggregate by kernel name (synchronises on the recorded events)
    std::vector<std::pair<std::string, std::pair<double, int>>> agg;
    for (auto& r : h->prof_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(h->prof_pool[r.e1]) != hipSuccess || hipEventElapsedTime(&ms, h->prof_pool[r.e0], h->prof_pool[r.e1]) != hipSuccess)
            return fail("hipEventElapsedTime failed");
        bool found = false;
        for (auto& a : agg) if (a.first == r.name) { a.second.first += ms; a.second.second++; found = true; break; }
        if (!found) agg.push_back({r.name, {ms, 1}});
    }
    if (index < 0 || index >= (int)agg.size()) return 2;       // end of list
    if (name && name_cap > 0) { strncpy(name, agg[index].first.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (total_ms) *total_ms = agg[index].second.first;
    if (count) *count = agg[index].second.second;
    return 0;
}

}  // extern "C"
