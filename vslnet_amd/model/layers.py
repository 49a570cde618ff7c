"""Host-side mirror of the reference's layer classes (/root/reference/model/layers_t7.py).

Same class names, constructor arguments and parameter names/shapes -- so `state_dict()` keys match the reference
 and reference `.t7` checkpoints load -- but these modules are PARAMETER CONTAINERS: their arithmetic
runs fused inside the hand-written HIP kernels behind `VSLNet.forward` (vslnet_amd/csrc), not layer by layer in
Python.  Calling a layer on its own raises: there is deliberately no eager/CPU fallback path.
"""
import torch
import torch.nn as nn

MASK_VALUE = -1e30


def mask_logits(inputs, mask, mask_value=MASK_VALUE):
    """layers_t7.py:7-9 (host helper; the kernels apply the same additive mask in fp32)."""
    return inputs + (1.0 - mask.type(torch.float32)) * mask_value


class _Fused(nn.Module):
    def forward(self, *a, **k):
        raise NotImplementedError('%s runs fused inside the HIP kernels of VSLNet.forward; it has no standalone '
                                  'eager implementation' % type(self).__name__)


class Conv1D(_Fused):
    """layers_t7.py:12-22 -- holds `conv1d.{weight (out, in, k), bias}`."""

    def __init__(self, in_dim, out_dim, kernel_size=1, stride=1, padding=0, bias=True):
        super().__init__()
        self.conv1d = nn.Conv1d(in_dim, out_dim, kernel_size, stride=stride, padding=padding, bias=bias)


class WordEmbedding(_Fused):
    """layers_t7.py:25-45: frozen [pad; glove] vectors + a trainable unk_vec when word vectors are given (what main_t7.py:83 does),
    a trainable nn.Embedding(num_words, word_dim, padding_idx=0) otherwise (vsl_config.word_table = 1)."""

    def __init__(self, num_words, word_dim, drop_rate, word_vectors=None):
        super().__init__()
        self.is_pretrained = word_vectors is not None
        if self.is_pretrained:
            self.pad_vec = nn.Parameter(torch.zeros(1, word_dim), requires_grad=False)
            unk = torch.empty(1, word_dim)
            nn.init.xavier_uniform_(unk)
            self.unk_vec = nn.Parameter(unk)
            self.glove_vec = nn.Parameter(torch.as_tensor(word_vectors, dtype=torch.float32).clone(), requires_grad=False)
        else:
            self.word_emb = nn.Embedding(num_words, word_dim, padding_idx=0)


class CharacterEmbedding(_Fused):
    """layers_t7.py:48-72."""

    def __init__(self, num_chars, char_dim, drop_rate):
        super().__init__()
        self.char_emb = nn.Embedding(num_chars, char_dim, padding_idx=0)
        self.char_convs = nn.ModuleList([
            nn.Sequential(nn.Conv2d(char_dim, c, kernel_size=(1, k), bias=True), nn.ReLU())
            for k, c in zip((1, 2, 3, 4), (10, 20, 30, 40))])


class Embedding(_Fused):
    """layers_t7.py:75-88."""

    def __init__(self, num_words, num_chars, word_dim, char_dim, drop_rate, out_dim, word_vectors=None):
        super().__init__()
        self.word_emb = WordEmbedding(num_words, word_dim, drop_rate, word_vectors=word_vectors)
        self.char_emb = CharacterEmbedding(num_chars, char_dim, drop_rate)
        self.linear = Conv1D(word_dim + 100, out_dim)


class PositionalEmbedding(_Fused):
    """layers_t7.py:91-102."""

    def __init__(self, num_embeddings, embedding_dim):
        super().__init__()
        self.position_embeddings = nn.Embedding(num_embeddings, embedding_dim)


class VisualProjection(_Fused):
    """layers_t7.py:105-115."""

    def __init__(self, visual_dim, dim, drop_rate=0.0):
        super().__init__()
        self.linear = Conv1D(visual_dim, dim)


class DepthwiseSeparableConvBlock(_Fused):
    """layers_t7.py:118-140."""

    def __init__(self, dim, kernel_size, drop_rate, num_layers=4):
        super().__init__()
        self.depthwise_separable_conv = nn.ModuleList([
            nn.Sequential(nn.Conv1d(dim, dim, kernel_size, groups=dim, padding=kernel_size // 2, bias=False),
                          nn.Conv1d(dim, dim, 1, bias=True), nn.ReLU()) for _ in range(num_layers)])
        self.layer_norms = nn.ModuleList([nn.LayerNorm(dim, eps=1e-6) for _ in range(num_layers)])


class MultiHeadAttentionBlock(_Fused):
    """layers_t7.py:143-190."""

    def __init__(self, dim, num_heads, drop_rate):
        super().__init__()
        assert dim % num_heads == 0, 'The channels (%d) is not a multiple of attention heads (%d)' % (dim, num_heads)
        self.head_size, self.num_heads, self.dim = dim // num_heads, num_heads, dim
        self.query, self.key, self.value = Conv1D(dim, dim), Conv1D(dim, dim), Conv1D(dim, dim)
        self.layer_norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.layer_norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.out_layer = Conv1D(dim, dim)


class FeatureEncoder(_Fused):
    """layers_t7.py:193-205."""

    def __init__(self, dim, num_heads, max_pos_len, kernel_size=7, num_layers=4, drop_rate=0.0):
        super().__init__()
        if kernel_size != 7 or num_layers != 4:
            raise NotImplementedError('the HIP encoder kernels are specialised for kernel_size=7, num_layers=4')
        self.pos_embedding = PositionalEmbedding(max_pos_len, dim)
        self.conv_block = DepthwiseSeparableConvBlock(dim, kernel_size, drop_rate, num_layers)
        self.attention_block = MultiHeadAttentionBlock(dim, num_heads, drop_rate)


class CQAttention(_Fused):
    """layers_t7.py:208-243."""

    def __init__(self, dim, drop_rate=0.0):
        super().__init__()
        w4C, w4Q, w4mlu = torch.empty(dim, 1), torch.empty(dim, 1), torch.empty(1, 1, dim)
        for w in (w4C, w4Q, w4mlu):
            nn.init.xavier_uniform_(w)
        self.w4C, self.w4Q, self.w4mlu = nn.Parameter(w4C), nn.Parameter(w4Q), nn.Parameter(w4mlu)
        self.cqa_linear = Conv1D(4 * dim, dim)


class WeightedPool(_Fused):
    """layers_t7.py:246-259."""

    def __init__(self, dim):
        super().__init__()
        w = torch.empty(dim, 1)
        nn.init.xavier_uniform_(w)
        self.weight = nn.Parameter(w)


class CQConcatenate(_Fused):
    """layers_t7.py:262-274."""

    def __init__(self, dim):
        super().__init__()
        self.weighted_pool = WeightedPool(dim)
        self.conv1d = Conv1D(2 * dim, dim)


class HighLightLayer(_Fused):
    """layers_t7.py:277-299."""

    def __init__(self, dim):
        super().__init__()
        self.conv1d = Conv1D(dim, 1)


class DynamicRNN(_Fused):
    """layers_t7.py:302-313: parameter container of the single-layer nn.LSTM(dim, dim) (gate order i,f,g,o, PyTorch's
    default uniform(-1/sqrt(dim), 1/sqrt(dim)) init -- VSLNet.init_parameters does not touch it).  The recurrence itself
    runs in k_lstm_fwd / k_lstm_bwd."""

    def __init__(self, dim):
        super().__init__()
        self.lstm = nn.LSTM(input_size=dim, hidden_size=dim, num_layers=1, bias=True, batch_first=True, bidirectional=False)


class ConditionedPredictor(_Fused):
    """layers_t7.py:316-369: rnn head (two DynamicRNN, the end one fed by the start one) or transformer head (one shared
    FeatureEncoder applied twice + LayerNorms), then the two span blocks."""

    def __init__(self, dim, num_heads, max_pos_len, drop_rate=0.0, predictor='rnn'):
        super().__init__()
        self.predictor = predictor
        if predictor == 'rnn':
            self.start_encoder = DynamicRNN(dim)
            self.end_encoder = DynamicRNN(dim)
        else:
            self.encoder = FeatureEncoder(dim, num_heads, max_pos_len, drop_rate=drop_rate)
            self.start_layer_norm = nn.LayerNorm(dim, eps=1e-6)
            self.end_layer_norm = nn.LayerNorm(dim, eps=1e-6)
        self.start_block = nn.Sequential(Conv1D(2 * dim, dim), nn.ReLU(), Conv1D(dim, 1))
        self.end_block = nn.Sequential(Conv1D(2 * dim, dim), nn.ReLU(), Conv1D(dim, 1))
