/* libvslnet_hip.so -- C ABI of the MI355X-native VSLNet forward/backward path.
 *
 * The reference (26hzhang/VSLNet) has no FFI / plugin interface: the hot path sits behind an ordinary Python class
 * (`model/VSLNet_t7.py:20-72`).  This ABI is therefore NEW; it is shaped so that the reference-side binding is one
 * thin `nn.Module` (vslnet_amd/model/VSLNet.py, loaded with ctypes -- see INTEGRATION.md).  Each entry point names the
 * reference interface it replaces.
 *
 * Conventions
 *   - plain C: opaque handle, plain pointers and sizes, no torch / C++ types in any signature;
 *   - every pointer inside `vsl_io` is a DEVICE pointer into caller-owned memory (PyTorch-ROCm tensors are used for
 *     storage only); the library owns nothing but the handle and its small per-shape plans;
 *   - all work is enqueued asynchronously on the caller's `hipStream_t` (passed as void*);
 *   - return value 0 = ok, non-zero = error; `vsl_last_error()` returns a thread-local message; nothing throws
 *     across the ABI;  one handle per device, not thread-safe per handle.
 *   - fp32 everywhere (the reference is IEEE fp32 end to end), ids/labels int64, masks fp32 0/1.
 */
#ifndef VSLNET_HIP_H
#define VSLNET_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vsl_handle_s* vsl_handle;

/* fields of the `configs` namespace read by VSLNet.__init__ (VSLNet_t7.py:24-38) */
typedef struct {
    int32_t dim;               /* configs.dim              (kernels are specialised for 128)              */
    int32_t num_heads;         /* configs.num_heads        (head size must be 16 -> 8 heads)              */
    int32_t max_pos_len;       /* configs.max_pos_len      (rows of the positional table; T, Lq <= it)    */
    int32_t video_feature_dim; /* configs.video_feature_dim (multiple of 4: 1024 I3D, 4096 / 500 C3D)         */
    int32_t word_dim;          /* configs.word_dim = 300   (word_dim + 100 must be a multiple of 8)       */
    int32_t char_dim;          /* configs.char_dim = 50    (<= 128)                                       */
    int32_t word_size;         /* configs.word_size  = rows of [pad; unk; glove]                          */
    int32_t char_size;         /* configs.char_size  = rows of the character table                        */
    int32_t predictor;         /* 0 = 'rnn' (DynamicRNN, layers_t7.py:302-313), 1 = 'transformer'         */
    float drop_rate;           /* configs.drop_rate                                                        */
    int32_t word_table;        /* 0: WordEmbedding(word_vectors=GloVe): frozen [pad; glove] + trainable unk_vec (layers_t7.py:29-34)
                                * 1: WordEmbedding(word_vectors=None): trainable nn.Embedding(word_size, word_dim, padding_idx=0)
                                *    (:36) -- the parameter "embedding_net.word_emb.word_emb.weight" replaces unk_vec and
                                *    vsl_io.pad_vec / glove_vec are ignored (may be NULL)                                  */
} vsl_config;

/* One forward / backward problem instance.  Replaces the argument list of VSLNet.forward (VSLNet_t7.py:52) plus
 * the tensors autograd keeps alive between forward and `total_loss.backward()` (main_t7.py:103-110). */
typedef struct {
    int32_t B, T, Lq, Lc;          /* batch, padded clips, padded query words, padded chars per word              */
    /* parameters */
    const float* params;           /* flat trainable parameter bucket, layout = vsl_param_info()                   */
    const float* pad_vec;          /* embedding_net.word_emb.pad_vec   (1, word_dim)  frozen                       */
    const float* glove_vec;        /* embedding_net.word_emb.glove_vec (word_size-2, word_dim) frozen              */
    /* batch (VSLNet_t7.py:52) */
    const int64_t* word_ids;       /* (B, Lq)                                                                      */
    const int64_t* char_ids;       /* (B, Lq, Lc)                                                                  */
    const float* video_features;   /* (B, T, video_feature_dim)                                                    */
    const float* v_mask;           /* (B, T)                                                                       */
    const float* q_mask;           /* (B, Lq)                                                                      */
    /* outputs (VSLNet_t7.py:62) */
    float* h_score;                /* (B, T)  exactly 0 at padded clips                                            */
    float* start_logits;           /* (B, T)  exactly -1e30 at padded clips                                        */
    float* end_logits;             /* (B, T)                                                                       */
    /* saved activations + backward temporaries; vsl_workspace_floats() floats, caller-owned, 16-byte aligned.      */
    /* Needs no initialisation, may be reused between shapes and handles.  The rnn head's in-launch hand-off buffers */
    /* live here: their tags are per-process NaN-patterned epochs (a 21-bit counter); the library clears a plan's   */
    /* hand-off range once per workspace pointer and again whenever the counter has wrapped.  A caller that writes  */
    /* its own data into a workspace between two steps of the SAME shape must hand over a different pointer (or     */
    /* re-create the handle): inside one generation a word of that range is assumed to be the library's own.        */
    float* workspace;
    /* dropout (nn.Dropout sites of layers_t7.py): counter-based masks, keyed by (seed, site, element)             */
    int32_t training;              /* 0: eval (no dropout)                                                         */
    uint64_t seed;                 /* change every step; the backward must see the forward's value                 */
    /* backward inputs / outputs */
    const float* d_h_score;        /* (B, T) dLoss/dh_score      (nullable = zeros)                                */
    const float* d_start_logits;   /* (B, T)                                                                        */
    const float* d_end_logits;     /* (B, T)                                                                        */
    float* grads;                  /* flat gradient bucket, same layout as `params` (overwritten, not accumulated) */
    /* data parallel: index of this shard's first sample in the GLOBAL batch.  The dropout element counters continue
     * from it, so for one seed the masks -- and with the global loss normalisers the summed gradient -- do not depend
     * on how many ranks the batch is split over (shards keep the global padded T / Lq / Lc).  0 = single process.  */
    int32_t sample_offset;
    /* bf16 THROUGHPUT mode (BASELINE configs[1] says "bf16"; the reference itself is fp32 end to end, so this is a separate mode
     * with its own tolerance, never the parity path): when non-NULL, the (B, T, video_feature_dim) features are read from this
     * bfloat16 copy instead of `video_features` (which may then be NULL) -- half the bytes of the only large HBM stream -- and
     * VisualProjection runs on the bf16 matrix cores (bf16 features x bf16-rounded weight, fp32 accumulate; the dropout scale is
     * applied to the fp32 sum).  Its weight gradient reads the same bf16 features (fp32 MFMA).  Everything else stays fp32. */
    const uint16_t* video_features_bf16;
    /* data parallel, optional (NULL = off): a caller-owned hipEvent_t that vsl_backward records as soon as the gradients of the
     * parameters at float offsets >= vsl_early_grad_offset() are FINAL in `grads` (the predictor block: about 55 % into the backward).
     * The caller can start their all-reduce on another stream behind this event while the rest of the backward runs; the remaining
     * offsets are final when vsl_backward's work on the caller's stream completes.  No reference counterpart (main_t7.py has no
     * distributed code); it is what makes the ONE exchange of SURVEY 8(e) overlap with the backward. */
    void* early_grads_event;
    /* single process, optional (NULL = off): the optimizer step of main_t7.py:111-113 applied by vsl_backward itself, behind its final
     * reduction (see vsl_fused_step below).  Not for data-parallel callers: their gradients are exchanged between the backward and the update. */
    const struct vsl_fused_step* fused_step;
    /* optional (NULL = off): vsl_loss folded into vsl_backward -- the call first does what vsl_loss(h, io, fused_loss) does (losses and the three
     * seeds are written exactly as vsl_loss writes them; complete when vsl_backward's work on the caller's stream is), then runs the backward
     * from those seeds; vsl_io.d_h_score / d_start_logits / d_end_logits are ignored.  For T >= 32 with the caller's mask_sum
     * the loss launch leaves the dependent chain: the span heads' and the highlight layer's backward compute their seeds from the logits
     * themselves (VSLNet_t7.py:67-72 and main_t7.py:107-110 as one call). */
    const struct vsl_loss_io* fused_loss;
} vsl_io;
/* Every struct of this header must be zero-initialised by the caller before the fields are set: new optional fields are appended, and
 * zero means "off".  vsl_abi_version() changes whenever a struct layout or an entry point's meaning changes; a binding checks it once. */
#define VSL_ABI_VERSION 8
int vsl_abi_version(void);

/* labels + weights for the fused loss (replaces compute_loss / compute_highlight_loss, VSLNet_t7.py:67-72, and the
 * combination `loc + highlight_lambda * hl` of main_t7.py:107).  Data-parallel callers pass the GLOBAL normalisers. */
typedef struct vsl_loss_io {
    const int64_t* start_labels;   /* (B)    */
    const int64_t* end_labels;     /* (B)    */
    const int64_t* h_labels;       /* (B, T) */
    float w_loc;                   /* weight of CE(start)+CE(end)      (main_t7.py: 1.0)                           */
    float w_highlight;             /* weight of the highlight loss     (main_t7.py: highlight_lambda = 5.0)        */
    float inv_batch;               /* 1 / global batch size (CrossEntropyLoss(mean), layers_t7.py:367-368)         */
    float mask_sum;                /* global sum(v_mask) (layers_t7.py:298); <= 0: use this batch's own sum        */
    float* losses;                 /* out, device, 4 floats: loc, highlight, w_loc*loc + w_hl*hl, mask_sum used     */
    float* d_h_score;              /* out (B, T), nullable: seeds for vsl_backward                                  */
    float* d_start_logits;         /* out (B, T)                                                                    */
    float* d_end_logits;           /* out (B, T)                                                                    */
} vsl_loss_io;

const char* vsl_last_error(void);

/* VSLNet.__init__ (VSLNet_t7.py:21-40): validates the config, fixes the flat parameter layout. */
int vsl_create(const vsl_config* cfg, vsl_handle* out);
int vsl_destroy(vsl_handle h);

/* flat parameter bucket layout; names and shapes are the reference's state_dict entries, trainable only
 * (pad_vec / glove_vec are passed separately in vsl_io). */
int vsl_param_count(vsl_handle h);
int vsl_param_info(vsl_handle h, int index, char* name, int name_cap, int64_t* offset, int64_t* numel, int32_t* ndim,
                   int64_t dims[4]);
int64_t vsl_param_floats(vsl_handle h);

/* number of floats the caller must provide in vsl_io.workspace for this shape */
int vsl_workspace_floats(vsl_handle h, int B, int T, int Lq, int Lc, int64_t* out);

/* VSLNet.forward (VSLNet_t7.py:52-62) */
int vsl_forward(vsl_handle h, const vsl_io* io, void* hip_stream);
/* compute_loss + compute_highlight_loss (VSLNet_t7.py:67-72) and their gradient seeds */
int vsl_loss(vsl_handle h, const vsl_io* io, const vsl_loss_io* l, void* hip_stream);
/* autograd backward of everything in vsl_forward (main_t7.py:110); needs the same io (workspace, seed) */
int vsl_backward(vsl_handle h, const vsl_io* io, void* hip_stream);
/* first float offset of the parameter block whose gradients are final when vsl_io.early_grads_event fires (== vsl_param_floats()
 * when the configuration has no such block: the rnn predictor) */
int64_t vsl_early_grad_offset(vsl_handle h);

/* test hook (tests/test_rnn_fused.py): sets the process-wide counter the fused rnn head derives its granule tags from -- 21 bits of it are the
 * tag, the bits above the generation for which a plan's granule buffers were last cleared -- so that a test can step across the wrap without two
 * million launches.  Returns the previous value.  No reference counterpart. */
uint64_t vsl_debug_rnn_launches(uint64_t n);
/* ConditionedPredictor.extract_index (layers_t7.py:355-363) */
int vsl_extract_index(vsl_handle h, const float* start_logits, const float* end_logits, int B, int T,
                      int64_t* start_index, int64_t* end_index, void* hip_stream);

/* Optimizer step on the flat buckets: clip_grad_norm_(grads, clip_norm) (main_t7.py:111) followed by AdamW with decoupled
 * weight decay (build_optimizer_and_scheduler, VSLNet_t7.py:8-17: no decay for names containing "bias", "layer_norm" or
 * "LayerNorm" -- the library derives that mask from its own parameter names).  Two kernels (one with norm_from_backward), no host synchronisation: the
 * global norm stays on the device.  Semantics are torch.optim.AdamW's (the reference's transformers.AdamW no longer
 * exists; SURVEY 8c "optimizer parity unpinned"):
 *     g   = grads * min(1, clip_norm / (||grads||_2 + 1e-6))            (clip_norm <= 0: no clipping)
 *     p  *= 1 - lr * weight_decay[param]
 *     m   = b1 m + (1 - b1) g ;  v = b2 v + (1 - b2) g^2
 *     p  -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
 * `step` is t (1-based).  exp_avg / exp_avg_sq are caller-owned buckets of vsl_param_floats() floats, zero at t = 1.
 * grad_norm_out (nullable): device float that receives the un-clipped global norm.  `grads` is not modified. */
typedef struct {
    float lr, beta1, beta2, eps, weight_decay, clip_norm;
    int32_t step;
    /* 0: torch.optim.AdamW ordering (above).  1: the historical transformers.AdamW the reference was written against
     * (VSLNet_t7.py:5,14; class removed from transformers 5.x, semantics from its published source):
     *     p -= lr * sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps) ;  then  p -= lr * weight_decay[param] * p          */
    int32_t hf_order;
    /* 1: the caller vouches that `grads` is exactly what the last vsl_backward of this handle left (no all-reduce, no scaling in
     * between: single-GPU training).  The global norm is then taken from the sums of squares the backward's final reduction
     * recorded per block instead of another pass over the bucket (one kernel less on the step's tail); the un-clipped norm agrees
     * with the two-kernel form to fp32 rounding (another summation order).  An error if `grads` is another buffer. */
    int32_t norm_from_backward;
} vsl_adamw;
int vsl_adamw_step(vsl_handle h, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                   const vsl_adamw* hp, float* grad_norm_out, void* hip_stream);
/* vsl_io.fused_step: vsl_backward followed by vsl_adamw_step(params, io->grads, exp_avg, exp_avg_sq, &hp, grad_norm_out) with
 * hp.norm_from_backward = 1, as ONE call (main_t7.py:110-113 without the host in between).  `params` is normally vsl_io.params itself: every
 * kernel that reads the weights is ordered in front of the update.  io->grads still receives the (unclipped) gradients.  The library issues
 * the final reduction and the update as two launches on the caller's stream; with VSL_FUSED_TAIL=1 in the environment they are ONE launch
 * whose workgroups hand their sums of squares to each other (same arithmetic, the norm summed in another fixed order) -- measured level with
 * the two launches, hence not the default; it needs every gradient to leave the final reduction (not word_table = 1). */
typedef struct vsl_fused_step {
    float* params;
    float* exp_avg;
    float* exp_avg_sq;
    vsl_adamw hp;
    float* grad_norm_out;      /* optional */
} vsl_fused_step;

/* workspace introspection for the parity tests: float offset of a named saved activation, -1 if unknown.
 * names: "video_affine", "embedding_net", "venc", "qenc", "cq_attention", "cq_concat", "gated", "pred_s", "pred_e" */
int64_t vsl_workspace_offset(vsl_handle h, int B, int T, int Lq, int Lc, const char* name);

/* per-kernel timing with HIP events recorded on the launch stream (no reference counterpart; feeds bench.py's roofline
 * object).  select: kernel launcher name ("wgrad", "vproj_fwd", ... or "*" for all, NULL/"" = off) and resets the
 * records; read: index-th aggregated record, returns 2 past the end. */
int vsl_profile_select(vsl_handle h, const char* kernel);
int vsl_profile_read(vsl_handle h, int index, char* name, int name_cap, double* total_ms, int32_t* count);
/* the same records one launch at a time, in enqueue order (tools/critical_path.py: the step's critical-path ledger): stream = index of
 * the launch's stream in order of first appearance, start / stop = the kernel's own dispatch-packet timestamps in us relative to the first
 * record's start, host = when the host enqueued it (us since vsl_profile_select), deps = indices of the launches it was ordered behind
 * (same-stream predecessor + every cross-stream ordering point in front of it).  Returns 2 past the end. */
int vsl_profile_launch(vsl_handle h, int index, char* name, int name_cap, int32_t* stream, double* start_us, double* stop_us,
                       double* host_us, int32_t* deps /* [6] */, int32_t* ndeps);

#ifdef __cplusplus
}
#endif
#endif
