"""Drop-in `VSLNet` module over the MI355X-native HIP path (mirror of /root/reference/model/VSLNet_t7.py:20-72).

    model = VSLNet(configs, word_vectors).to('cuda')
    h_score, start_logits, end_logits = model(word_ids, char_ids, video_features, v_mask, q_mask)
    loss = model.compute_loss(...) + 5.0 * model.compute_highlight_loss(...)
    loss.backward()

Same constructor, methods, `state_dict` keys and error behaviour as the reference; the arithmetic is ONE fused launch
sequence of hand-written gfx950 kernels reached through the C ABI (include/vslnet_hip.h).  All trainable parameters
live in one flat fp32 bucket (the nn.Parameters are views into it) and so do the gradients -- the layout data-parallel
training all-reduces as a single RCCL call (vslnet_amd/dp.py).
"""
import torch
import torch.nn as nn

from ..engine import Engine
from .layers import (Embedding, VisualProjection, FeatureEncoder, CQAttention, CQConcatenate, ConditionedPredictor,
                     HighLightLayer)


def build_optimizer_and_scheduler(model, configs):
    """VSLNet_t7.py:8-17: AdamW (no decay on bias / LayerNorm) + linear decay with warm-up.
    `transformers.AdamW` is gone from current transformers; torch.optim.AdamW with the historical HF defaults
    (eps 1e-6) is used instead -- optimizer parity is 'unpinned' (SURVEY 8c)."""
    no_decay = ('bias', 'layer_norm', 'LayerNorm')
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    groups = [{'params': [p for n, p in named if not any(k in n for k in no_decay)], 'weight_decay': 0.01},
              {'params': [p for n, p in named if any(k in n for k in no_decay)], 'weight_decay': 0.0}]
    optimizer = torch.optim.AdamW(groups, lr=configs.init_lr, eps=1e-6)
    total, warm = float(configs.num_train_steps), float(configs.num_train_steps * configs.warmup_proportion)

    def lr_lambda(step):
        if step < warm:
            return float(step) / max(1.0, warm)
        return max(0.0, (total - step) / max(1.0, total - warm))
    return optimizer, torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda)


class _ForwardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, word_ids, char_ids, vfeats, v_mask, q_mask, *params):
        eng = model._engine_for(vfeats.device)
        model._step += 1
        we = model.embedding_net.word_emb
        h, sl, el = eng.forward(model._flat, we.pad_vec.data if we.is_pretrained else None,
                                we.glove_vec.data if we.is_pretrained else None, word_ids.contiguous(), char_ids.contiguous(),
                                vfeats.contiguous().float(), v_mask.contiguous().float(), q_mask.contiguous().float(),
                                training=model.training, seed=(model._seed << 20) + model._step)
        ctx.model, ctx.token = model, model._step
        return h, sl, el

    @staticmethod
    def backward(ctx, d_h, d_sl, d_el):
        model = ctx.model
        if ctx.token != model._step:
            raise RuntimeError('VSLNet.backward: another forward ran since this graph was built; the HIP engine keeps the '
                               'saved activations of the latest forward only')
        # a fresh bucket per backward (autograd keeps what is returned): vsl_backward overwrites every element, so it is neither
        # zeroed nor copied
        g = model._engine.backward(d_h.contiguous(), d_sl.contiguous(), d_el.contiguous(), torch.empty_like(model._flat_grad))
        views = model._engine.views(g)
        return (None,) * 6 + tuple(views[n] for n in model._flat_names)


class _LossFn(torch.autograd.Function):
    """compute_loss / compute_highlight_loss through the fused loss kernel; `which` selects the returned term."""

    @staticmethod
    def forward(ctx, model, which, a, b, lab0, lab1, hlab, mask):
        eng = model._engine
        if which == 'loc':
            losses, d_h, d_sl, d_el = eng.loss(lab0, lab1, hlab, 1.0, 0.0, scores=a, start_logits=a, end_logits=b, v_mask=mask)
            ctx.save_for_backward(d_sl, d_el)
            return losses[0].clone()
        losses, d_h, d_sl, d_el = eng.loss(lab0, lab1, hlab, 0.0, 1.0, scores=a, start_logits=a, end_logits=a, v_mask=mask)
        ctx.save_for_backward(d_h)
        return losses[1].clone()

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        if len(saved) == 2:
            return None, None, saved[0] * g, saved[1] * g, None, None, None, None
        return None, None, saved[0] * g, None, None, None, None, None


class VSLNet(nn.Module):
    def __init__(self, configs, word_vectors):
        super().__init__()
        self.configs = configs
        d = configs.dim
        self.embedding_net = Embedding(num_words=configs.word_size, num_chars=configs.char_size, out_dim=d,
                                       word_dim=configs.word_dim, char_dim=configs.char_dim, word_vectors=word_vectors,
                                       drop_rate=configs.drop_rate)
        self.video_affine = VisualProjection(visual_dim=configs.video_feature_dim, dim=d, drop_rate=configs.drop_rate)
        self.feature_encoder = FeatureEncoder(dim=d, num_heads=configs.num_heads, kernel_size=7, num_layers=4,
                                              max_pos_len=configs.max_pos_len, drop_rate=configs.drop_rate)
        self.cq_attention = CQAttention(dim=d, drop_rate=configs.drop_rate)
        self.cq_concat = CQConcatenate(dim=d)
        self.highlight_layer = HighLightLayer(dim=d)
        self.predictor = ConditionedPredictor(dim=d, num_heads=configs.num_heads, drop_rate=configs.drop_rate,
                                              max_pos_len=configs.max_pos_len, predictor=configs.predictor)
        self.init_parameters()
        self._engine = None
        self._flat = None
        self._flat_grad = None
        self._flat_names = None
        self._step = 0
        self._seed = int(getattr(configs, 'seed', 12345))

    def init_parameters(self):
        """VSLNet_t7.py:42-50: Xavier-uniform conv/linear weights, zero biases, nn.LSTM.reset_parameters() for the rnn head.
        Same module order and the same initialisers as the reference, so the same torch seed gives bit-identical weights
        (pinned by tests/test_oracle_golden.py::test_init_matches_reference_under_the_same_seed)."""
        def init_weights(m):
            if isinstance(m, (nn.Conv2d, nn.Conv1d, nn.Linear)):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LSTM):
                m.reset_parameters()
        self.apply(init_weights)

    # ---- flat parameter bucket ------------------------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._engine = None            # device / dtype may have changed: rebuild lazily
        self._flat = None
        return out

    def _engine_for(self, device):
        if self._engine is not None and self._flat is not None:
            return self._engine
        pdev = next(self.parameters()).device
        if pdev.type != 'cuda':
            raise RuntimeError('VSLNet (vslnet_amd) runs on an MI355X only: move the module with .to("cuda") first; there is '
                               'no CPU fallback')
        eng = Engine(self.configs, device=pdev, word_table=not self.embedding_net.word_emb.is_pretrained)
        named = dict(self.named_parameters())
        flat = eng.new_flat()
        for n, o, k, shp in eng.layout:
            p = named[n]
            if tuple(p.shape) != tuple(shp):
                raise RuntimeError('parameter %s has shape %s, engine expects %s' % (n, tuple(p.shape), shp))
            flat[o:o + k].view(shp).copy_(p.data)
            p.data = flat[o:o + k].view(shp)       # the nn.Parameter now aliases the flat bucket
        self._engine, self._flat = eng, flat
        self._flat_grad = eng.new_flat()
        self._flat_names = [n for n, _, _, _ in eng.layout]
        self._flat_params = [named[n] for n in self._flat_names]
        return eng

    @property
    def flat_parameters(self):
        """(flat params, flat grads) buckets -- what the data-parallel wrapper all-reduces / the fused optimizer walks."""
        self._engine_for(None)
        return self._flat, self._flat_grad

    def state_dict_from_flat(self, host_flat):
        """`self.state_dict()` (same keys, order, shapes) with the trainable entries taken from a HOST copy of the flat bucket and the
        frozen ones (pad_vec, glove_vec) from a host copy made once -- the checkpoint writer's path (runner.CheckpointWriter.save_flat)."""
        from collections import OrderedDict
        eng = self._engine_for(None)
        views = {n: host_flat[o:o + k].view(shp) for n, o, k, shp in eng.layout}
        if getattr(self, '_frozen_host', None) is None:
            self._frozen_host = {k: v.detach().cpu() for k, v in self.state_dict().items() if k not in views}
        return OrderedDict((k, views[k] if k in views else self._frozen_host[k]) for k in self.state_dict().keys())

    # ---- reference API ---------------------------------------------------------------------------------------------
    def forward(self, word_ids, char_ids, video_features, v_mask, q_mask):
        self._engine_for(video_features.device)
        return _ForwardFn.apply(self, word_ids, char_ids, video_features, v_mask, q_mask, *self._flat_params)

    def extract_index(self, start_logits, end_logits):
        self._engine_for(start_logits.device)
        return self._engine.extract_index(start_logits.detach().contiguous(), end_logits.detach().contiguous())

    def compute_highlight_loss(self, scores, labels, mask):
        B = scores.shape[0]
        z = torch.zeros(B, dtype=torch.int64, device=scores.device)
        return _LossFn.apply(self, 'hl', scores.contiguous(), None, z, z, labels.contiguous().long(), mask.contiguous().float())

    def compute_loss(self, start_logits, end_logits, start_labels, end_labels):
        B, T = start_logits.shape
        zh = torch.zeros(B, T, dtype=torch.int64, device=start_logits.device)
        ones = torch.ones(B, T, dtype=torch.float32, device=start_logits.device)
        return _LossFn.apply(self, 'loc', start_logits.contiguous(), end_logits.contiguous(), start_labels.contiguous().long(),
                             end_labels.contiguous().long(), zh, ones)
