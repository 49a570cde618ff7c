"""CPU oracle for the VSLNet forward/backward hot path (TEST INFRASTRUCTURE ONLY).

This file is a from-scratch *functional* restatement (plain torch-CPU fp32 tensor
ops, autograd for the backward) of the reference's PyTorch path:

    /root/reference/model/layers_t7.py   (layers)
    /root/reference/model/VSLNet_t7.py   (wiring, losses)

It is NOT the product: only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it, and only as the checker / the
timed CPU baseline.  The product path (`vslnet_amd`) never imports it and fails
loudly when the HIP library is missing.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function here
against fixtures in `tests/golden/` that were produced by importing the reference
itself in the build container (`oracle/make_golden.py`), to <= 2e-5.

All parameters are addressed by the reference's own `state_dict` key names
 so reference checkpoints can be fed straight in.
Each function cites the reference lines it restates.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

MASK_VALUE = -1e30
LN_EPS = 1e-6
CHAR_KERNELS = (1, 2, 3, 4)
CHAR_CHANNELS = (10, 20, 30, 40)


# Training-mode parity: nn.Dropout draws from torch's generator, the HIP path from a counter-based hash.  A test can hand
# the oracle the HIP path's masks: DROP_FORCED(call_number, shape, p) -> multiplier tensor (0 or 1/(1-p)); the dropout
# sites are numbered in call order (visual, word, char, 9 per encoder pass [conv 0-3, LN1, probabilities, attention output,
# LN2, out projection] for video / query, CQ context, CQ query, then the two predictor passes).
DROP_FORCED = None
DROP_CALLS = 0


def force_dropout(fn):
    global DROP_FORCED, DROP_CALLS
    DROP_FORCED, DROP_CALLS = fn, 0


TRAIN_MASK_SEED = 20260928


def train_mask(call, shape, p, seed=TRAIN_MASK_SEED):
    """Dropout multiplier of the seeded training-mode fixtures (oracle/make_golden.py TRAIN_CASES, tests/test_oracle_golden.py): call number
    `call` of a forward draws its mask from its own generator, in the (B, L, C) layout of the call site."""
    g = torch.Generator().manual_seed(seed + int(call))
    return (torch.rand(shape, generator=g) >= p).to(torch.float32) / (1.0 - p)


def _drop(x, p, training):
    """nn.Dropout(p) with inverted scaling (layers_t7.py: every nn.Dropout site)."""
    global DROP_CALLS
    if training and p > 0.0:
        if DROP_FORCED is not None:
            m = DROP_FORCED(DROP_CALLS, tuple(x.shape), p)
            DROP_CALLS += 1
            return x * m
        return F.dropout(x, p=p, training=True)
    return x


def mask_logits(x, mask, value=MASK_VALUE):
    """layers_t7.py:7-9 -- additive mask; masked entries become exactly -1e30 in fp32."""
    return x + (1.0 - mask.to(torch.float32)) * value


# Sum of |terms| of every Conv1D bias gradient of the last backward (id(bias tensor) -> (C,) tensor): db[c] = sum_rows dY[row, c] cancels
# structurally for some biases (the attention key bias: SURVEY 8a) and numerically on tiny batches; a test that wants an absolute gate
# proportional to what was actually added asks for these (tests/test_fuzz_parity.py).  None = not recording.
BIAS_TERMS = None


def record_bias_terms(on=True):
    global BIAS_TERMS
    BIAS_TERMS = {} if on else None


def bias_term_sums(P):
    """{parameter name: max over its elements of sum |dY|} for the biases seen since record_bias_terms()."""
    return {k: float(BIAS_TERMS[id(v)].max()) for k, v in P.items() if BIAS_TERMS is not None and id(v) in BIAS_TERMS}


def pointwise(x, w, b=None):
    """Conv1D with kernel 1 (layers_t7.py:12-22): y[..., o] = sum_i x[..., i] * w[o, i, 0] + b[o]."""
    y = torch.matmul(x, w[:, :, 0].t())
    if b is None:
        return y
    out = y + b
    if BIAS_TERMS is not None and out.requires_grad:
        def hook(g, key=id(b)):
            BIAS_TERMS[key] = BIAS_TERMS.get(key, 0) + g.detach().abs().reshape(-1, g.shape[-1]).sum(0)
        out.register_hook(hook)
    return out


def layer_norm(x, g, b):
    """nn.LayerNorm(dim, eps=1e-6) (layers_t7.py:129,152-153): biased variance over the last axis."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(var + LN_EPS) * g + b


# --------------------------------------------------------------------------------------
# embeddings / projection
# --------------------------------------------------------------------------------------
def visual_projection(P, vfeat, p, training):
    """VisualProjection.forward, layers_t7.py:111-115."""
    x = _drop(vfeat, p, training)
    return pointwise(x, P['video_affine.linear.conv1d.weight'], P['video_affine.linear.conv1d.bias'])


def word_embedding(P, word_ids, p, training):
    """WordEmbedding.forward, layers_t7.py:39-45: table = [pad; unk; glove] (word vectors given) or the trainable
    nn.Embedding(word_size, word_dim, padding_idx=0) of the word_vectors=None branch (:36, 43-44)."""
    if 'embedding_net.word_emb.word_emb.weight' in P:
        return _drop(F.embedding(word_ids, P['embedding_net.word_emb.word_emb.weight'], padding_idx=0), p, training)
    table = torch.cat([P['embedding_net.word_emb.pad_vec'], P['embedding_net.word_emb.unk_vec'],
                       P['embedding_net.word_emb.glove_vec']], dim=0)
    return _drop(table[word_ids], p, training)


def char_embedding(P, char_ids, p, training):
    """CharacterEmbedding.forward, layers_t7.py:62-72: lookup -> dropout -> 4x(1xk conv + ReLU + max over chars)."""
    emb = F.embedding(char_ids, P['embedding_net.char_emb.char_emb.weight'], padding_idx=0)   # (B, Lq, Lc, 50); row 0 gets no grad (:51)
    emb = _drop(emb, p, training)
    outs = []
    for i, k in enumerate(CHAR_KERNELS):
        w = P['embedding_net.char_emb.char_convs.%d.0.weight' % i]       # (c, 50, 1, k)
        b = P['embedding_net.char_emb.char_convs.%d.0.bias' % i]
        y = F.conv2d(emb.permute(0, 3, 1, 2), w, b)                       # (B, c, Lq, Lc-k+1)
        outs.append(torch.relu(y).max(dim=3).values.permute(0, 2, 1))     # (B, Lq, c)
    return torch.cat(outs, dim=-1)                                        # (B, Lq, 100)


def embedding(P, word_ids, char_ids, p, training):
    """Embedding.forward, layers_t7.py:83-88."""
    e = torch.cat([word_embedding(P, word_ids, p, training), char_embedding(P, char_ids, p, training)], dim=-1)
    return pointwise(e, P['embedding_net.linear.conv1d.weight'], P['embedding_net.linear.conv1d.bias'])


# --------------------------------------------------------------------------------------
# feature encoder
# --------------------------------------------------------------------------------------
def depthwise7(x, w):
    """nn.Conv1d(dim, dim, 7, groups=dim, padding=3, bias=False) on (B, L, C) (layers_t7.py:123-124).

    u[b,t,c] = sum_k w[c,0,k] * x[b,t+k-3,c], zero outside [0, L) -- padded rows are NOT masked."""
    k = w.shape[-1]
    return F.conv1d(x.transpose(1, 2), w, padding=k // 2, groups=w.shape[0]).transpose(1, 2)


# ReLU decisions of the last forward, in call order (conv layers of venc, qenc, [p1, p2], then the start / end span heads).
# The gradient is discontinuous where a pre-activation crosses zero, so the GPU parity tests compare these sign patterns
# with the ones the HIP path saved and only then decide which gradient gate applies (tests/helpers.py: relu_flips).
RELU_SIGNS = None
RELU_FORCED = None
RELU_FORCED_DEV = 0.0      # largest |pre-activation| on which a forced branch differed from the sign seen in that same forward (reported, not gated)
RELU_FORCED_RATIO = 0.0    # the same in units of the pre-activation's own fp32 noise scale: |z| / (2^-24 * (sum_k |a_k w_k| + |b|))
RELU_FORCED_COUNTS = []    # per ReLU site of the last forced forward: (overridden decisions, elements)


def record_relu_signs(on=True):
    global RELU_SIGNS
    RELU_SIGNS = [] if on else None


def force_relu_signs(masks):
    """Take the given branch (bool tensors, call order) at every ReLU of the next forward instead of sign(z): lets a test
    evaluate the oracle's gradient ON THE BRANCH THE GPU PATH TOOK when a pre-activation sits inside the forward noise."""
    global RELU_FORCED, RELU_FORCED_DEV, RELU_FORCED_RATIO, RELU_FORCED_COUNTS
    RELU_FORCED = list(masks) if masks is not None else None
    if masks is not None:
        RELU_FORCED_DEV = 0.0
        RELU_FORCED_RATIO = 0.0
        RELU_FORCED_COUNTS = []


def forced_relu_deviation():
    """After a forced forward: the largest |z| at which the forced decision contradicted sign(z) of THIS forward.  A test
    asserts it is inside the forward noise -- a forced branch is only legitimate where the pre-activation is ~0 (torch's own
    no_grad and autograd forwards already differ by ~1e-7 there: mkldnn picks different primitives)."""
    return RELU_FORCED_DEV


def forced_relu_counts():
    """After a forced forward: [(overridden decisions, elements)] per ReLU site, call order."""
    return list(RELU_FORCED_COUNTS)


def forced_relu_noise_ratio():
    """After a forced forward: max over the contradicted decisions of |z| / (u * S), u = 2^-24, S = sum_k |a_k w_k| + |b| of that
    pre-activation's own dot product (the quantity every fp32 summation-order error bound is proportional to).  tests/helpers.py
    (RELU_NOISE_KAPPA) derives the bound it is asserted against."""
    return RELU_FORCED_RATIO


def _relu(z, site, operands=None):
    """operands = (a, w, b) of z = pointwise(a, w, b): only read while a branch is being forced, for the noise scale S."""
    global RELU_FORCED_DEV, RELU_FORCED_RATIO
    if RELU_SIGNS is not None:
        RELU_SIGNS.append((site, (z.detach() > 0)))
    if RELU_FORCED is not None:
        m = RELU_FORCED.pop(0)
        bad = m != (z.detach() > 0)
        RELU_FORCED_COUNTS.append((int(bad.sum()), bad.numel()))
        dis = z.detach().abs()[bad]
        if dis.numel():
            RELU_FORCED_DEV = max(RELU_FORCED_DEV, float(dis.max()))
            if operands is not None:
                a, w, b = operands
                S = pointwise(a.detach().abs(), w.detach().abs(), None if b is None else b.detach().abs())
                RELU_FORCED_RATIO = max(RELU_FORCED_RATIO, float((dis / (S[bad] * 2.0 ** -24)).max()))
        return z * m.to(z.dtype)
    return torch.relu(z)


def conv_layer(x, ln_g, ln_b, dw_w, pw_w, pw_b, p, training):
    """One iteration of DepthwiseSeparableConvBlock.forward, layers_t7.py:133-139."""
    v = layer_norm(x, ln_g, ln_b)
    u = depthwise7(v, dw_w)
    z = pointwise(u, pw_w, pw_b)
    return _drop(_relu(z, 'conv', (u, pw_w, pw_b)), p, training) + x


def conv_block(P, pre, x, p, training, n_layers=4):
    """DepthwiseSeparableConvBlock.forward, layers_t7.py:131-140."""
    outs = []
    for i in range(n_layers):
        x = conv_layer(x,
                       P[pre + 'conv_block.layer_norms.%d.weight' % i], P[pre + 'conv_block.layer_norms.%d.bias' % i],
                       P[pre + 'conv_block.depthwise_separable_conv.%d.0.weight' % i],
                       P[pre + 'conv_block.depthwise_separable_conv.%d.1.weight' % i],
                       P[pre + 'conv_block.depthwise_separable_conv.%d.1.bias' % i], p, training)
        outs.append(x)
    return x, outs


def mha_block(P, pre, x, mask, n_heads, p, training, want=None):
    """MultiHeadAttentionBlock.forward, layers_t7.py:167-190."""
    a = pre + 'attention_block.'
    B, L, D = x.shape
    hd = D // n_heads
    h1 = _drop(layer_norm(x, P[a + 'layer_norm1.weight'], P[a + 'layer_norm1.bias']), p, training)
    q = pointwise(h1, P[a + 'query.conv1d.weight'], P[a + 'query.conv1d.bias'])
    k = pointwise(h1, P[a + 'key.conv1d.weight'], P[a + 'key.conv1d.bias'])
    v = pointwise(h1, P[a + 'value.conv1d.weight'], P[a + 'value.conv1d.bias'])
    qh = q.view(B, L, n_heads, hd).permute(0, 2, 1, 3)
    kh = k.view(B, L, n_heads, hd).permute(0, 2, 1, 3)
    vh = v.view(B, L, n_heads, hd).permute(0, 2, 1, 3)
    s = torch.matmul(qh, kh.transpose(-1, -2)) / math.sqrt(hd)            # scores scaled AFTER QK^T (:175)
    if mask is not None:
        s = mask_logits(s, mask[:, None, None, :])                         # keys only (:176-178)
    pr = _drop(torch.softmax(s, dim=-1), p, training)
    att = torch.matmul(pr, vh).permute(0, 2, 1, 3).reshape(B, L, D)
    r = _drop(att, p, training) + x
    h2 = _drop(layer_norm(r, P[a + 'layer_norm2.weight'], P[a + 'layer_norm2.bias']), p, training)
    o = pointwise(h2, P[a + 'out_layer.conv1d.weight'], P[a + 'out_layer.conv1d.bias'])
    y = _drop(o, p, training) + r
    if want is not None:
        want.update(q=q, k=k, v=v, att=att, r=r)
    return y


def feature_encoder(P, pre, x, mask, n_heads, p, training, want=None):
    """FeatureEncoder.forward, layers_t7.py:201-205 (pos-emb rows 0..L-1 added to every row, padded too)."""
    L = x.shape[1]
    x0 = x + P[pre + 'pos_embedding.position_embeddings.weight'][:L]
    c, couts = conv_block(P, pre, x0, p, training)
    y = mha_block(P, pre, c, mask, n_heads, p, training, want)
    if want is not None:
        want.update(x0=x0, conv_outs=couts)
    return y


# --------------------------------------------------------------------------------------
# fusion
# --------------------------------------------------------------------------------------
def cq_attention(P, ctx, qry, c_mask, q_mask, p, training, want=None):
    """CQAttention.forward + trilinear_attention, layers_t7.py:223-243."""
    cd, qd = _drop(ctx, p, training), _drop(qry, p, training)
    s0 = torch.matmul(cd, P['cq_attention.w4C'])                           # (B, T, 1)
    s1 = torch.matmul(qd, P['cq_attention.w4Q']).transpose(1, 2)           # (B, 1, Lq)
    s2 = torch.matmul(cd * P['cq_attention.w4mlu'], qd.transpose(1, 2))    # (B, T, Lq)
    score = s0 + s1 + s2
    s_row = torch.softmax(mask_logits(score, q_mask[:, None, :]), dim=2)   # over query words
    s_col = torch.softmax(mask_logits(score, c_mask[:, :, None]), dim=1)   # over clips
    c2q = torch.matmul(s_row, qry)
    q2c = torch.matmul(torch.matmul(s_row, s_col.transpose(1, 2)), ctx)    # materialises (B,T,T) like the reference
    cat = torch.cat([ctx, c2q, ctx * c2q, ctx * q2c], dim=2)
    out = pointwise(cat, P['cq_attention.cqa_linear.conv1d.weight'], P['cq_attention.cqa_linear.conv1d.bias'])
    if want is not None:
        want.update(score=score, s_row=s_row, s_col=s_col, c2q=c2q, q2c=q2c)
    return out


def weighted_pool(P, x, mask):
    """WeightedPool.forward, layers_t7.py:253-259."""
    alpha = torch.matmul(x, P['cq_concat.weighted_pool.weight'])           # (B, Lq, 1)
    alpha = torch.softmax(mask_logits(alpha, mask[:, :, None]), dim=1)
    return (x * alpha).sum(dim=1)                                          # (B, D)


def cq_concat(P, ctx, qry, q_mask):
    """CQConcatenate.forward, layers_t7.py:268-274."""
    pooled = weighted_pool(P, qry, q_mask)
    T = ctx.shape[1]
    cat = torch.cat([ctx, pooled[:, None, :].expand(-1, T, -1)], dim=2)
    return pointwise(cat, P['cq_concat.conv1d.conv1d.weight'], P['cq_concat.conv1d.conv1d.bias'])


def highlight(P, x, mask):
    """HighLightLayer.forward, layers_t7.py:282-289."""
    lg = pointwise(x, P['highlight_layer.conv1d.conv1d.weight'], P['highlight_layer.conv1d.conv1d.bias'])[..., 0]
    return torch.sigmoid(mask_logits(lg, mask))


def highlight_loss(scores, labels, mask, eps=1e-12):
    """HighLightLayer.compute_loss, layers_t7.py:291-299 (BCELoss clamps each log at -100)."""
    y = labels.to(torch.float32)
    w = torch.where(y == 0.0, y + 1.0, 2.0 * y)
    # BCELoss: -(y*max(log p,-100) + (1-y)*max(log(1-p),-100)); its backward is (p-y)/max(p*(1-p),1e-12), which is
    # what keeps exactly-0 scores at padded clips finite -- use torch's primitive so autograd matches.
    per = F.binary_cross_entropy(scores, y, reduction='none') * w
    m = mask.to(torch.float32)
    return (per * m).sum() / (m.sum() + eps)


# --------------------------------------------------------------------------------------
# conditioned predictor
# --------------------------------------------------------------------------------------
def lstm_layer(x, w_ih, w_hh, b_ih, b_hh):
    """nn.LSTM(dim, dim, 1 layer, batch_first, unidirectional), gate order i,f,g,o (layers_t7.py:305-306)."""
    B, L, D = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    gi = torch.matmul(x, w_ih.t()) + b_ih
    out = []
    for t in range(L):
        g = gi[:, t] + torch.matmul(h, w_hh.t()) + b_hh
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out.append(h)
    return torch.stack(out, dim=1)


def dynamic_rnn(P, pre, x, mask):
    """DynamicRNN.forward, layers_t7.py:308-313: LSTM over ALL padded steps, then output * mask."""
    out = lstm_layer(x, P[pre + 'lstm.weight_ih_l0'], P[pre + 'lstm.weight_hh_l0'],
                     P[pre + 'lstm.bias_ih_l0'], P[pre + 'lstm.bias_hh_l0'])
    return out * mask[:, :, None]


def span_head(P, name, feat, x):
    """start_block / end_block, layers_t7.py:328-337,349-350: Conv1D(2d->d) + ReLU + Conv1D(d->1)."""
    cat = torch.cat([feat, x], dim=2)
    z = pointwise(cat, P['predictor.%s_block.0.conv1d.weight' % name], P['predictor.%s_block.0.conv1d.bias' % name])
    return pointwise(_relu(z, 'head_' + name, (cat, P['predictor.%s_block.0.conv1d.weight' % name], P['predictor.%s_block.0.conv1d.bias' % name])),
                     P['predictor.%s_block.2.conv1d.weight' % name],
                     P['predictor.%s_block.2.conv1d.bias' % name])[..., 0]


def conditioned_predictor(P, x, mask, predictor, n_heads, p, training, want=None):
    """ConditionedPredictor.forward, layers_t7.py:340-353."""
    if predictor == 'rnn':
        s = dynamic_rnn(P, 'predictor.start_encoder.', x, mask)
        e = dynamic_rnn(P, 'predictor.end_encoder.', s, mask)
        sn, en = s, e
    else:
        s = feature_encoder(P, 'predictor.encoder.', x, mask, n_heads, p, training)
        e = feature_encoder(P, 'predictor.encoder.', s, mask, n_heads, p, training)   # same weights, pos-emb again
        sn = layer_norm(s, P['predictor.start_layer_norm.weight'], P['predictor.start_layer_norm.bias'])
        en = layer_norm(e, P['predictor.end_layer_norm.weight'], P['predictor.end_layer_norm.bias'])
    if want is not None:
        want.update(pred_s=s, pred_e=e)
    return mask_logits(span_head(P, 'start', sn, x), mask), mask_logits(span_head(P, 'end', en, x), mask)


def span_loss(start_logits, end_logits, start_labels, end_labels):
    """ConditionedPredictor.compute_cross_entropy_loss, layers_t7.py:365-369 (mean over the batch, twice)."""
    return F.cross_entropy(start_logits, start_labels) + F.cross_entropy(end_logits, end_labels)


def extract_index(start_logits, end_logits):
    """ConditionedPredictor.extract_index, layers_t7.py:355-363."""
    ps, pe = torch.softmax(start_logits, dim=1), torch.softmax(end_logits, dim=1)
    outer = torch.triu(ps[:, :, None] * pe[:, None, :], diagonal=0)
    return outer.max(dim=2).values.argmax(dim=1), outer.max(dim=1).values.argmax(dim=1)


# --------------------------------------------------------------------------------------
# whole model
# --------------------------------------------------------------------------------------
def forward(P, cfg, word_ids, char_ids, vfeat, v_mask, q_mask, training=False, want=None):
    """VSLNet.forward, VSLNet_t7.py:52-62.  `cfg` needs .num_heads .drop_rate .predictor."""
    p = float(cfg.drop_rate)
    H = int(cfg.num_heads)
    vf = visual_projection(P, vfeat, p, training)
    qf = embedding(P, word_ids, char_ids, p, training)
    wv = {} if want is not None else None
    wq = {} if want is not None else None
    ve = feature_encoder(P, 'feature_encoder.', vf, v_mask, H, p, training, wv)
    qe = feature_encoder(P, 'feature_encoder.', qf, q_mask, H, p, training, wq)
    wc = {} if want is not None else None
    f1 = cq_attention(P, ve, qe, v_mask, q_mask, p, training, wc)
    f2 = cq_concat(P, f1, qe, q_mask)
    h = highlight(P, f2, v_mask)
    gated = f2 * h[:, :, None]
    wp = {} if want is not None else None
    sl, el = conditioned_predictor(P, gated, v_mask, cfg.predictor, H, p, training, wp)
    if want is not None:
        want.update(video_affine=vf, embedding_net=qf, venc=ve, qenc=qe, cq_attention=f1, cq_concat=f2,
                    h_score=h, gated=gated, venc_parts=wv, qenc_parts=wq, cq_parts=wc, pred_parts=wp)
    return h, sl, el


def total_loss(P, cfg, batch, training=False, highlight_lambda=5.0):
    """The timed region of main_t7.py:103-107: forward + both losses, combined as loc + 5.0 * highlight."""
    h, sl, el = forward(P, cfg, batch['word_ids'], batch['char_ids'], batch['vfeats'], batch['v_mask'],
                        batch['q_mask'], training)
    hl = highlight_loss(h, batch['h_labels'], batch['v_mask'])
    loc = span_loss(sl, el, batch['s_labels'], batch['e_labels'])
    return loc + highlight_lambda * hl, (h, sl, el, hl, loc)


# --------------------------------------------------------------------------------------
# host-side batch helpers the callers own (not kernels, but part of the boundary contract)
# --------------------------------------------------------------------------------------
def convert_length_to_mask(lengths):
    """util/runner_utils_t7.py:48-52: (B, max(lengths)) float mask."""
    m = int(lengths.max().item())
    return (torch.arange(m)[None, :] < lengths[:, None]).float()


def highlight_labels(s_inds, e_inds, lens, max_len, extend=0.1):
    """util/data_loader_t7.py:37-52: span widened by round(0.1*len) each side, clipped to the valid clips."""
    B = len(s_inds)
    h = torch.zeros(B, max_len, dtype=torch.int64)
    for i in range(B):
        st, et = int(s_inds[i]), int(e_inds[i])
        ext = round(extend * float(et - st + 1))
        if ext > 0:
            st, et = max(0, st - ext), min(et + ext, int(lens[i]) - 1)
        h[i, st:et + 1] = 1
    return h


# --------------------------------------------------------------------------------------
# parameter / batch factories shared by tests, smoke and bench (synthetic data)
# --------------------------------------------------------------------------------------
def param_shapes(cfg):
    """state_dict key -> shape for the transformer / rnn variants."""
    d, Dv = cfg.dim, cfg.video_feature_dim
    S = {}
    if getattr(cfg, 'word_table', False):               # WordEmbedding(word_vectors=None), layers_t7.py:36
        S['embedding_net.word_emb.word_emb.weight'] = (cfg.word_size, cfg.word_dim)
    else:
        S['embedding_net.word_emb.pad_vec'] = (1, cfg.word_dim)
        S['embedding_net.word_emb.unk_vec'] = (1, cfg.word_dim)
        S['embedding_net.word_emb.glove_vec'] = (cfg.word_size - 2, cfg.word_dim)
    S['embedding_net.char_emb.char_emb.weight'] = (cfg.char_size, cfg.char_dim)
    for i, (k, c) in enumerate(zip(CHAR_KERNELS, CHAR_CHANNELS)):
        S['embedding_net.char_emb.char_convs.%d.0.weight' % i] = (c, cfg.char_dim, 1, k)
        S['embedding_net.char_emb.char_convs.%d.0.bias' % i] = (c,)
    S['embedding_net.linear.conv1d.weight'] = (d, cfg.word_dim + 100, 1)
    S['embedding_net.linear.conv1d.bias'] = (d,)
    S['video_affine.linear.conv1d.weight'] = (d, Dv, 1)
    S['video_affine.linear.conv1d.bias'] = (d,)

    def enc(pre):
        S[pre + 'pos_embedding.position_embeddings.weight'] = (cfg.max_pos_len, d)
        for i in range(4):
            S[pre + 'conv_block.depthwise_separable_conv.%d.0.weight' % i] = (d, 1, 7)
            S[pre + 'conv_block.depthwise_separable_conv.%d.1.weight' % i] = (d, d, 1)
            S[pre + 'conv_block.depthwise_separable_conv.%d.1.bias' % i] = (d,)
        for i in range(4):
            S[pre + 'conv_block.layer_norms.%d.weight' % i] = (d,)
            S[pre + 'conv_block.layer_norms.%d.bias' % i] = (d,)
        for n in ('query', 'key', 'value'):
            S[pre + 'attention_block.%s.conv1d.weight' % n] = (d, d, 1)
            S[pre + 'attention_block.%s.conv1d.bias' % n] = (d,)
        for n in ('layer_norm1', 'layer_norm2'):
            S[pre + 'attention_block.%s.weight' % n] = (d,)
            S[pre + 'attention_block.%s.bias' % n] = (d,)
        S[pre + 'attention_block.out_layer.conv1d.weight'] = (d, d, 1)
        S[pre + 'attention_block.out_layer.conv1d.bias'] = (d,)

    enc('feature_encoder.')
    S['cq_attention.w4C'] = (d, 1)
    S['cq_attention.w4Q'] = (d, 1)
    S['cq_attention.w4mlu'] = (1, 1, d)
    S['cq_attention.cqa_linear.conv1d.weight'] = (d, 4 * d, 1)
    S['cq_attention.cqa_linear.conv1d.bias'] = (d,)
    S['cq_concat.weighted_pool.weight'] = (d, 1)
    S['cq_concat.conv1d.conv1d.weight'] = (d, 2 * d, 1)
    S['cq_concat.conv1d.conv1d.bias'] = (d,)
    S['highlight_layer.conv1d.conv1d.weight'] = (1, d, 1)
    S['highlight_layer.conv1d.conv1d.bias'] = (1,)
    if cfg.predictor == 'rnn':
        for n in ('start', 'end'):
            S['predictor.%s_encoder.lstm.weight_ih_l0' % n] = (4 * d, d)
            S['predictor.%s_encoder.lstm.weight_hh_l0' % n] = (4 * d, d)
            S['predictor.%s_encoder.lstm.bias_ih_l0' % n] = (4 * d,)
            S['predictor.%s_encoder.lstm.bias_hh_l0' % n] = (4 * d,)
    else:
        enc('predictor.encoder.')
        for n in ('start', 'end'):
            S['predictor.%s_layer_norm.weight' % n] = (d,)
            S['predictor.%s_layer_norm.bias' % n] = (d,)
    for n in ('start', 'end'):
        S['predictor.%s_block.0.conv1d.weight' % n] = (d, 2 * d, 1)
        S['predictor.%s_block.0.conv1d.bias' % n] = (d,)
        S['predictor.%s_block.2.conv1d.weight' % n] = (1, d, 1)
        S['predictor.%s_block.2.conv1d.bias' % n] = (1,)
    return S


FROZEN = ('embedding_net.word_emb.pad_vec', 'embedding_net.word_emb.glove_vec')


def random_params(cfg, seed=12345, perturb=True):
    """Random-init weights of the reference architecture (init rules of VSLNet_t7.py:42-50 in distribution;
    `perturb` additionally randomises biases / LayerNorm so that tests exercise every term)."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    for k, shp in param_shapes(cfg).items():
        if k.endswith('pad_vec'):
            t = torch.zeros(shp)
        elif k.endswith('glove_vec') or 'position_embeddings' in k or k.endswith('char_emb.weight') or k.endswith('word_emb.weight'):
            t = torch.randn(shp, generator=g)
            if k.endswith('char_emb.weight') or k.endswith('word_emb.weight'):
                t[0] = 0.0                                                 # padding_idx=0 row (layers_t7.py:36, 51)
        elif ('layer_norm' in k) and k.endswith('weight'):
            t = torch.ones(shp) + (0.1 * torch.randn(shp, generator=g) if perturb else 0.0)
        elif k.endswith('bias') or 'lstm.bias' in k:
            t = 0.05 * torch.randn(shp, generator=g) if perturb else torch.zeros(shp)
        else:
            fan_out = shp[0] * (math.prod(shp[2:]) if len(shp) > 2 else 1)
            fan_in = shp[1] * (math.prod(shp[2:]) if len(shp) > 2 else 1) if len(shp) > 1 else shp[0]
            a = math.sqrt(6.0 / (fan_in + fan_out))
            t = (torch.rand(shp, generator=g) * 2 - 1) * a
        P[k] = t.to(torch.float32)
    return P


def make_cfg(**kw):
    base = dict(word_size=1002, char_size=40, dim=128, word_dim=300, char_dim=50, drop_rate=0.0,
                video_feature_dim=1024, num_heads=8, max_pos_len=128, predictor='transformer')
    base.update(kw)
    return SimpleNamespace(**base)


def synthetic_batch(cfg, B, T, Lq=20, Lc=10, seed=0, ragged=False):
    """SURVEY.md 8(d) synthetic inputs: N(0,1) features, uniform ids, full or ragged lengths + labels."""
    g = torch.Generator().manual_seed(seed)
    vfeats = torch.randn(B, T, cfg.video_feature_dim, generator=g)
    word_ids = torch.randint(2, cfg.word_size, (B, Lq), generator=g)
    char_ids = torch.randint(2, cfg.char_size, (B, Lq, Lc), generator=g)
    if ragged:
        lens = torch.randint(max(T // 2, 1), T + 1, (B,), generator=g)
        lens[0] = T
        qlens = torch.randint(min(3, Lq), Lq + 1, (B,), generator=g)
        qlens[-1] = Lq
        for b in range(B):
            vfeats[b, lens[b]:] = 0.0                                      # pad_video_seq pads with zeros
            word_ids[b, qlens[b]:] = 0
            char_ids[b, qlens[b]:] = 0
            # a few short words: trailing chars are PAD (id 0)
            char_ids[b, 0, max(Lc - 3, 1):] = 0
    else:
        lens = torch.full((B,), T, dtype=torch.int64)
    v_mask = convert_length_to_mask(lens)
    q_mask = (word_ids != 0).float()
    s = (torch.rand(B, generator=g) * (lens.float() / 2)).long()
    e = torch.minimum(s + (torch.rand(B, generator=g) * (lens.float() / 8 + 1)).long(), lens - 1)
    h_labels = highlight_labels(s, e, lens, T)
    return dict(vfeats=vfeats, lens=lens, word_ids=word_ids, char_ids=char_ids, v_mask=v_mask, q_mask=q_mask,
                s_labels=s, e_labels=e, h_labels=h_labels)
