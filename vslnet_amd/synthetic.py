"""Synthetic batches of the reference's input layout: what `train_collate_fn`
(/root/reference/util/data_loader_t7.py:24-61) would hand to the model, generated directly on the device."""
from types import SimpleNamespace

import torch


def make_configs(**kw):
    """The fields VSLNet.__init__ reads from `configs` (VSLNet_t7.py:24-38) with the CLI defaults of main_t7.py:13-45."""
    base = dict(word_size=1002, char_size=40, dim=128, word_dim=300, char_dim=50, drop_rate=0.2,
                video_feature_dim=1024, num_heads=8, max_pos_len=128, predictor='transformer', seed=12345,
                init_lr=0.0001, warmup_proportion=0.0, num_train_steps=1000, highlight_lambda=5.0, clip_norm=1.0)
    base.update(kw)
    return SimpleNamespace(**base)


def convert_length_to_mask(lengths):
    """util/runner_utils_t7.py:48-52."""
    max_len = int(lengths.max().item())
    return (torch.arange(max_len, device=lengths.device).expand(lengths.shape[0], max_len) < lengths.unsqueeze(1)).float()


def highlight_labels(s_inds, e_inds, lens, max_len, extend=0.1):
    """Label construction of train_collate_fn (data_loader_t7.py:37-52): span widened by round(0.1 * span) clips."""
    B = len(s_inds)
    h = torch.zeros(B, max_len, dtype=torch.int64)
    for i in range(B):
        st, et = int(s_inds[i]), int(e_inds[i])
        ext = round(extend * float(et - st + 1))
        if ext > 0:
            st, et = max(0, st - ext), min(et + ext, int(lens[i]) - 1)
        h[i, st:et + 1] = 1
    return h


def synthetic_batch(configs, B, T, Lq=20, Lc=10, seed=0, device='cuda', ragged=False):
    g = torch.Generator().manual_seed(seed)
    vfeats = torch.randn(B, T, configs.video_feature_dim, generator=g)
    word_ids = torch.randint(2, configs.word_size, (B, Lq), generator=g)
    char_ids = torch.randint(2, configs.char_size, (B, Lq, Lc), generator=g)
    lens = torch.full((B,), T, dtype=torch.int64)
    if ragged:
        lens = torch.randint(max(T // 2, 1), T + 1, (B,), generator=g)
        lens[0] = T
        qlens = torch.randint(min(3, Lq), Lq + 1, (B,), generator=g)
        qlens[-1] = Lq
        for b in range(B):
            vfeats[b, lens[b]:] = 0.0
            word_ids[b, qlens[b]:] = 0
            char_ids[b, qlens[b]:] = 0
    s = (torch.rand(B, generator=g) * (lens.float() / 2)).long()
    e = torch.minimum(s + (torch.rand(B, generator=g) * (lens.float() / 8 + 1)).long(), lens - 1)
    out = dict(vfeats=vfeats, lens=lens, word_ids=word_ids, char_ids=char_ids, v_mask=convert_length_to_mask(lens),
               q_mask=(word_ids != 0).float(), s_labels=s, e_labels=e, h_labels=highlight_labels(s, e, lens, T))
    return {k: v.to(device).contiguous() for k, v in out.items()}
