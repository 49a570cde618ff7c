"""Batch pipeline of the VSLNet path: what sits between the processed dataset and `VSLNet.forward`.

Mirrors the behaviour of the reference's `util/data_util.py` (padding :117-159, feature resampling :59-73, index <-> time
:92-114, feature loading :44-56) and `util/data_loader_t7.py` (Dataset :8-21, collate functions :24-81, loaders :84-95),
written array-at-a-time instead of list-at-a-time: a batch is assembled straight into (optionally pinned) tensors, so the
H2D copy of the (B, T, Dv) feature block can be asynchronous.  Pinned by tests/test_data_pipeline.py against fixtures
generated from the reference's functions (oracle/make_golden.py: run_host_pipeline).

Dataset generation itself (tokenisation with nltk, GloVe filtering: util/data_gen.py) is out of scope; `load_dataset` reads
the pickle the reference writes (`<save_dir>/<task>_<fv>_<max_pos_len>.pkl`, data_gen.py:196-244) or builds a synthetic,
learnable dataset of the same record format for `--task synthetic`.
"""
import glob
import os
import pickle

import numpy as np
import torch
import torch.utils.data


# ------------------------------------------------------------------------------------------------ index <-> time
def index_to_time(start_index, end_index, num_units, duration):
    """data_util.py:109-114 -- clip i covers [i, i + 1) * duration / num_units, evaluated in float32 like the reference."""
    unit = np.float32
    start = (np.arange(0, num_units).astype(unit) * duration / float(num_units))[start_index]
    end = (np.arange(1, num_units + 1).astype(unit) * duration / float(num_units))[end_index]
    return start, end


def time_to_index(start_time, end_time, num_units, duration):
    """data_util.py:98-106 -- the (start, end) clip pair whose interval has the highest IoU with [start_time, end_time];
    first maximum in row-major order, like np.argmax over the reference's candidate grid.  Returns (s, e, overlaps)."""
    s_t = (np.arange(0, num_units).astype(np.float32) / float(num_units) * duration).astype(np.float64)
    e_t = (np.arange(1, num_units + 1).astype(np.float32) / float(num_units) * duration).astype(np.float64)
    inter = np.maximum(0.0, np.minimum(e_t[None, :], end_time) - np.maximum(s_t[:, None], start_time))
    union = np.maximum(1e-12, np.maximum(e_t[None, :], end_time) - np.minimum(s_t[:, None], start_time))
    overlaps = inter / union
    flat = int(np.argmax(overlaps))
    return flat // num_units, flat % num_units, overlaps


def resample_features(feature, max_num_clips):
    """visual_feature_sampling, data_util.py:59-73: videos longer than max_num_clips are mean-pooled over
    max_num_clips nearly equal windows (window i = rows round(i n / m) .. round((i+1) n / m), at least one row)."""
    n = feature.shape[0]
    if n <= max_num_clips:
        return feature
    edges = np.round(np.arange(0, max_num_clips + 1, 1.0) / max_num_clips * n).astype(np.int32)
    edges = np.minimum(edges, n - 1)
    lo, hi = edges[:-1], np.maximum(edges[1:], edges[:-1] + 1)          # an empty window takes its single start row
    return np.stack([feature[a:b].mean(axis=0) if b - a > 1 else feature[a] for a, b in zip(lo, hi)])


def load_video_features(root, max_position_length):
    """data_util.py:44-56: {video id: (n_clips, Dv) float32}, resampled to at most max_position_length clips."""
    out = {}
    for path in sorted(glob.glob(os.path.join(root, '*.npy'))):
        vid = os.path.basename(path).split('.')[0]
        feat = np.load(path)
        out[vid] = feat if max_position_length is None else resample_features(feat, max_position_length)
    return out


# ------------------------------------------------------------------------------------------------ padding
def pad_words(word_ids):
    """pad_seq (data_util.py:117-128) for a batch of word-id lists -> (B, Lq) int64, 0 = <PAD>."""
    width = max(len(w) for w in word_ids)
    out = np.zeros((len(word_ids), width), dtype=np.int64)
    for i, w in enumerate(word_ids):
        out[i, :len(w)] = w
    return out


def pad_chars(char_ids):
    """pad_char_seq (data_util.py:131-143): (B, Lq, Lc) int64 with Lq = longest query, Lc = longest word of the batch."""
    lq = max(len(q) for q in char_ids)
    lc = max(len(w) for q in char_ids for w in q)
    out = np.zeros((len(char_ids), lq, lc), dtype=np.int64)
    for i, q in enumerate(char_ids):
        for j, w in enumerate(q):
            out[i, j, :len(w)] = w
    return out


def pad_videos(features, out=None):
    """pad_video_seq (data_util.py:146-159): zero-pad to the longest video of the batch -> ((B, T, Dv) float32, lens)."""
    lens = np.array([f.shape[0] for f in features], dtype=np.int64)
    T, dv = int(lens.max()), features[0].shape[1]
    if out is None:
        out = np.zeros((len(features), T, dv), dtype=np.float32)
    else:
        out[...] = 0
    for i, f in enumerate(features):
        out[i, :f.shape[0]] = f
    return out, lens


REF_EXTEND = 0.1      # train_collate_fn hard-codes the highlight extension (data_loader_t7.py:42); --extend is accepted and ignored, like the reference


def highlight_targets(s_inds, e_inds, lens, max_len, extend=0.1):
    """train_collate_fn, data_loader_t7.py:41-52: 1 on the target span widened by round(extend * span) clips on both
    sides (Python round = half-to-even, like np.rint), clipped to the video."""
    s, e, lens = (np.asarray(x, dtype=np.int64) for x in (s_inds, e_inds, lens))
    ext = np.rint(extend * (e - s + 1).astype(np.float64)).astype(np.int64)
    lo = np.where(ext > 0, np.maximum(0, s - ext), s)
    hi = np.where(ext > 0, np.minimum(e + ext, lens - 1), e)
    pos = np.arange(max_len)[None, :]
    return ((pos >= lo[:, None]) & (pos <= hi[:, None])).astype(np.int64)


# ------------------------------------------------------------------------------------------------ dataset / collate
class VideoQueryDataset(torch.utils.data.Dataset):
    """data_loader_t7.py:8-21: record -> (record, (n, Dv) features, word ids, char ids, s_ind, e_ind)."""

    def __init__(self, dataset, video_features):
        self.dataset, self.video_features = dataset, video_features

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, index):
        r = self.dataset[index]
        return r, self.video_features[r['vid']], r['w_ids'], r['c_ids'], int(r['s_ind']), int(r['e_ind'])


def _pin(t, pin):
    return t.pin_memory() if pin and torch.cuda.is_available() else t


def collate_test(items, pin=False):
    """test_collate_fn, data_loader_t7.py:64-81 -> (records, vfeats, vfeat_lens, word_ids, char_ids)."""
    records, feats, words, chars = zip(*[(it[0], it[1], it[2], it[3]) for it in items])
    vfeats, lens = pad_videos(feats)
    return (records, _pin(torch.from_numpy(vfeats), pin), torch.from_numpy(lens), torch.from_numpy(pad_words(words)),
            torch.from_numpy(pad_chars(chars)))


def collate_train(items, pin=False, extend=0.1):
    """train_collate_fn, data_loader_t7.py:24-61 -> (records, vfeats, vfeat_lens, word_ids, char_ids, s_labels, e_labels,
    h_labels)."""
    records, vfeats, lens, word_ids, char_ids = collate_test(items, pin)
    s = np.array([it[4] for it in items], dtype=np.int64)
    e = np.array([it[5] for it in items], dtype=np.int64)
    h = highlight_targets(s, e, lens.numpy(), int(lens.max()), extend)
    return records, vfeats, lens, word_ids, char_ids, torch.from_numpy(s), torch.from_numpy(e), torch.from_numpy(h)


def get_train_loader(dataset, video_features, configs, pin=False, generator=None):
    """data_loader_t7.py:84-88 (shuffled, 0 workers)."""
    return torch.utils.data.DataLoader(VideoQueryDataset(dataset, video_features), batch_size=configs.batch_size, shuffle=True,
                                       collate_fn=lambda b: collate_train(b, pin, REF_EXTEND), generator=generator)


def get_test_loader(dataset, video_features, configs, pin=False):
    """data_loader_t7.py:91-95."""
    return torch.utils.data.DataLoader(VideoQueryDataset(dataset, video_features), batch_size=configs.batch_size, shuffle=False,
                                       collate_fn=lambda b: collate_test(b, pin))


# ------------------------------------------------------------------------------------------------ HBM-resident splits
class ResidentSplit:
    """One split of the processed dataset held ENTIRELY in device memory.

    The reference re-collates every batch on the host and copies the (B, T, Dv) feature block (32 MiB at the Charades
    shape) to the device each step (data_loader_t7.py:24-81, main_t7.py:96-99); with a sub-millisecond model step that
    pipeline is the whole step time.  A MI355X has 288 GB of HBM and the reference's processed benchmarks are a few GB
    (Charades-STA I3D at max_pos_len 128: ~5 GB), so the MI355X-first layout is: every video's (already resampled)
    feature matrix zero-padded into ONE (n_videos, Tmax, Dv) tensor, every record's word / char ids, span and highlight
    row padded to the split-wide maxima, all uploaded once.  A batch is then five device gathers out of views narrowed to
    the BATCH maxima (so the tensors have exactly the shapes and contents train_collate_fn / test_collate_fn produce),
    and the host only draws the permutation and three integer maxima per step.

    Iterating yields the reference's tuples (records, vfeats, vfeat_lens[host], word_ids, char_ids[, s, e, h]) with device
    tensors -- `runner.eval_test` and the module-API loop take them unchanged.  `shards(rank, world)` yields the dicts the
    fused data-parallel loop consumes (only this rank's rows are gathered; widths and normalisers stay global; `row0` = index
    of the shard's first sample in the global batch, the `sample_offset` of the dropout counters)."""

    def __init__(self, records, video_features, configs, device, train, generator=None):
        self.records, self.device, self.train = list(records), torch.device(device), train
        self.batch_size, self.generator = configs.batch_size, generator
        extend = REF_EXTEND            # data_loader_t7.py:42 hard-codes 0.1; its --extend flag (main_t7.py:40) is dead
        n = len(self.records)
        vids = sorted({r['vid'] for r in self.records})
        vpos = {v: i for i, v in enumerate(vids)}
        vlen = np.array([video_features[v].shape[0] for v in vids], dtype=np.int64)
        tmax, dv = int(vlen.max()), int(video_features[vids[0]].shape[1])
        need = len(vids) * tmax * dv * 4
        if self.device.type == 'cuda':
            free, _ = torch.cuda.mem_get_info(self.device)
            if need > 0.8 * free:
                raise MemoryError('the %s split needs %.1f GiB of device memory for its features (%d videos x %d clips x %d), %.1f GiB '
                                  'are free: use --data loader' % ('train' if train else 'test', need / 2 ** 30, len(vids), tmax, dv, free / 2 ** 30))
        self.feats = torch.zeros((len(vids), tmax, dv), dtype=torch.float32, device=self.device)
        for i, v in enumerate(vids):                                       # one upload per video, once
            self.feats[i, :vlen[i]] = torch.from_numpy(np.ascontiguousarray(video_features[v], dtype=np.float32)).to(self.device)
        self.vid_of = np.array([vpos[r['vid']] for r in self.records], dtype=np.int64)
        self.lens = vlen[self.vid_of]                                      # clips per RECORD (host copy: widths, normalisers)
        self.nwords = np.array([len(r['w_ids']) for r in self.records], dtype=np.int64)
        self.nchars = np.array([max(len(w) for w in r['c_ids']) for r in self.records], dtype=np.int64)
        words = np.zeros((n, int(self.nwords.max())), dtype=np.int64)
        chars = np.zeros((n, int(self.nwords.max()), int(self.nchars.max())), dtype=np.int64)
        for i, r in enumerate(self.records):
            words[i, :len(r['w_ids'])] = r['w_ids']
            for j, w in enumerate(r['c_ids']):
                chars[i, j, :len(w)] = w
        up = lambda a: torch.from_numpy(a).to(self.device)
        self.d_vid, self.d_lens, self.d_words, self.d_chars = up(self.vid_of), up(self.lens), up(words), up(chars)
        if train:
            s = np.array([int(r['s_ind']) for r in self.records], dtype=np.int64)
            e = np.array([int(r['e_ind']) for r in self.records], dtype=np.int64)
            self.d_s, self.d_e = up(s), up(e)
            self.d_h = up(highlight_targets(s, e, self.lens, tmax, extend))      # the row of a record does not depend on its batch

    def __len__(self):
        return (len(self.records) + self.batch_size - 1) // self.batch_size

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in vars(self).values() if torch.is_tensor(t))

    def _order(self, order=None):
        n = len(self.records)
        if order is not None:
            return np.asarray(order, dtype=np.int64)
        if not self.train:
            return np.arange(n, dtype=np.int64)
        return torch.randperm(n, generator=self.generator).numpy()         # shuffle=True of data_loader_t7.py:86

    def _gather(self, idx_dev, T, Lq, Lc):
        vf = torch.index_select(self.feats[:, :T], 0, self.d_vid[idx_dev])
        out = [vf, torch.index_select(self.d_words[:, :Lq], 0, idx_dev), torch.index_select(self.d_chars[:, :Lq, :Lc], 0, idx_dev)]
        if self.train:
            out += [self.d_s[idx_dev], self.d_e[idx_dev], torch.index_select(self.d_h[:, :T], 0, idx_dev)]
        return out

    def shards(self, rank=0, world=1, order=None):
        """One dict per GLOBAL batch: this rank's rows on the device + what the losses need about the whole batch."""
        order = self._order(order)
        d_order = torch.from_numpy(order).to(self.device)                  # the only per-epoch upload
        bs = self.batch_size
        for a in range(0, len(order), bs):
            idx = order[a:a + bs]
            lens = self.lens[idx]
            T, Lq, Lc = int(lens.max()), int(self.nwords[idx].max()), int(self.nchars[idx].max())
            base, rem = divmod(len(idx), world)
            lo = rank * base + min(rank, rem)
            hi = lo + base + (1 if rank < rem else 0)
            idx_dev = d_order[a + lo:a + hi]
            g = self._gather(idx_dev, T, Lq, Lc)
            l_dev = self.d_lens[idx_dev]
            v_mask = (torch.arange(T, device=self.device)[None, :] < l_dev[:, None]).float()
            b = {'records': [self.records[i] for i in idx[lo:hi]], 'lens_global': lens, 'row0': lo, 'vfeats': g[0], 'v_mask': v_mask,
                 'word_ids': g[1], 'char_ids': g[2]}
            if self.train:
                b.update(s_labels=g[3], e_labels=g[4], h_labels=g[5])
            yield b

    def __iter__(self):
        for b in self.shards(0, 1):
            head = (b['records'], b['vfeats'], torch.from_numpy(b['lens_global']), b['word_ids'], b['char_ids'])
            yield head + ((b['s_labels'], b['e_labels'], b['h_labels']) if self.train else ())


def loader_shards(loader, device, rank=0, world=1):
    """The same dicts as ResidentSplit.shards from a host-side DataLoader (collate on the host, H2D copy per step)."""
    for batch in loader:
        records, vfeats, lens = batch[0], batch[1], batch[2]
        n = vfeats.shape[0]
        base, rem = divmod(n, world)
        lo = rank * base + min(rank, rem)
        sl = slice(lo, lo + base + (1 if rank < rem else 0))
        mv = lambda t: t[sl].to(device, non_blocking=True).contiguous()
        l_dev = lens[sl].to(device, non_blocking=True)
        T = int(lens.max())
        b = {'records': list(records[sl]), 'lens_global': lens.numpy(), 'row0': lo, 'vfeats': mv(vfeats),
             'v_mask': (torch.arange(T, device=device)[None, :] < l_dev[:, None]).float(),
             'word_ids': mv(batch[3]), 'char_ids': mv(batch[4])}
        if len(batch) > 5:
            b.update(s_labels=mv(batch[5]), e_labels=mv(batch[6]), h_labels=mv(batch[7]))
        yield b


# ------------------------------------------------------------------------------------------------ datasets
def dataset_path(configs):
    """Where the reference's gen_or_load_dataset keeps its pickle (data_gen.py:201-206)."""
    parts = [configs.task, configs.fv, str(configs.max_pos_len)] + ([configs.suffix] if getattr(configs, 'suffix', None) else [])
    return os.path.join(configs.save_dir, '_'.join(parts) + '.pkl')


def synthetic_dataset(configs, n_train=512, n_test=128, n_words=200, n_chars=30, seed=0):
    """A dataset in the reference's record format (data_gen.py:178-181, 236-239) whose answer is recoverable from the
    inputs: the first word of every query names one of 16 'events'; the clips inside the target span carry that event's
    signature in their features.  Used by `--task synthetic`, the end-to-end tests and the convergence check."""
    rs = np.random.RandomState(seed)
    dv, tmax = configs.video_feature_dim, configs.max_pos_len
    n_events = 16
    signature = rs.randn(n_events, dv).astype(np.float32)
    feats, sets = {}, []
    for split, n in (('train', n_train), ('test', n_test)):
        recs = []
        for i in range(n):
            vid = '%s%05d' % (split[:2], i)
            L = int(rs.randint(max(8, tmax // 2), tmax + 1))
            ev = int(rs.randint(n_events))
            s = int(rs.randint(0, L - 2))
            e = int(min(L - 1, s + rs.randint(1, max(2, L // 3))))
            f = rs.randn(L, dv).astype(np.float32) * 0.5
            f[s:e + 1] += signature[ev]
            feats[vid] = f
            dur = float(L) * 1.5
            nw = int(rs.randint(3, 9))
            w = [2 + ev] + [int(x) for x in rs.randint(2 + n_events, n_words, size=nw - 1)]
            c = [[2 + (wi * 7 + k) % (n_chars - 2) for k in range(1 + wi % 6)] for wi in w]
            st, et = index_to_time(s, e, L, dur)
            recs.append({'sample_id': len(recs), 'vid': vid, 's_time': float(st), 'e_time': float(et), 'duration': dur,
                         'words': ['w%d' % x for x in w], 's_ind': s, 'e_ind': e, 'v_len': L, 'w_ids': w, 'c_ids': c})
        sets.append(recs)
    vectors = rs.randn(n_words - 2, configs.word_dim).astype(np.float32) * 0.3
    dataset = {'train_set': sets[0], 'val_set': None, 'test_set': sets[1], 'word_dict': None, 'char_dict': None,
               'word_vector': vectors, 'n_train': n_train, 'n_val': 0, 'n_test': n_test, 'n_words': n_words, 'n_chars': n_chars}
    return dataset, feats


# limits of the HIP engine (vslnet_amd/csrc/common.hpp: MAX_LQ, MAX_LC; api.hip vsl_create): checked against the WHOLE dataset
# before the first step, so that one long query or token cannot abort a run in the middle of an epoch
ENGINE_MAX_WORDS, ENGINE_MAX_CHARS, ENGINE_MAX_CHAR_DIM = 128, 40, 128


def validate_dataset(dataset, configs):
    """Host-side checks the reference gets for free from nn.Embedding's IndexError: every id inside its table, every length
    inside the engine's limits.  Raises ValueError naming the first offending record."""
    n_words, n_chars = int(dataset['n_words']), int(dataset['n_chars'])
    if int(getattr(configs, 'char_dim', 50)) > ENGINE_MAX_CHAR_DIM:
        raise ValueError('--char_dim %d: the HIP embedding kernels stage character rows in LDS with a row stride of at most %d floats'
                         % (configs.char_dim, ENGINE_MAX_CHAR_DIM))
    for split in ('train_set', 'val_set', 'test_set'):
        for r in dataset.get(split) or []:
            w = np.asarray(r['w_ids'], dtype=np.int64)
            where = '%s record %s (vid %s)' % (split, r.get('sample_id', '?'), r.get('vid', '?'))
            if len(w) == 0 or len(w) > min(ENGINE_MAX_WORDS, int(configs.max_pos_len)):
                raise ValueError('%s: %d query words; supported: 1 .. min(max_pos_len=%d, %d) (the CQAttention kernels keep a whole '
                                 'query in LDS)' % (where, len(w), configs.max_pos_len, ENGINE_MAX_WORDS))
            if w.min() < 0 or w.max() >= n_words:
                raise ValueError('%s: word id %d outside [0, %d)' % (where, int(w.max() if w.max() >= n_words else w.min()), n_words))
            for c in r['c_ids']:
                c = np.asarray(c, dtype=np.int64)
                if len(c) > ENGINE_MAX_CHARS:
                    raise ValueError('%s: a token of %d characters; supported: <= %d' % (where, len(c), ENGINE_MAX_CHARS))
                if len(c) and (c.min() < 0 or c.max() >= n_chars):
                    raise ValueError('%s: char id outside [0, %d)' % (where, n_chars))
            if 's_ind' in r and not (0 <= int(r['s_ind']) <= int(r['e_ind'])):
                raise ValueError('%s: span labels (%s, %s)' % (where, r['s_ind'], r['e_ind']))


def load_dataset(configs):
    """-> (dataset dict in the layout of data_gen.py:236-239, {vid: features}).  ValueError for an unknown task or a
    missing processed dataset, like the reference (data_gen.py:229, main_t7.py:134)."""
    if configs.task == 'synthetic':
        dataset, feats = synthetic_dataset(configs, getattr(configs, 'synthetic_train', 512), getattr(configs, 'synthetic_test', 128),
                                           seed=configs.seed)
        validate_dataset(dataset, configs)
        return dataset, feats
    if configs.task not in ('charades', 'activitynet', 'tacos'):
        raise ValueError('Unknown task {}!!!'.format(configs.task))
    path = dataset_path(configs)
    if not os.path.exists(path):
        raise ValueError('processed dataset %s not found: generate it with the reference\'s util/data_gen.py (needs nltk and the '
                         'GloVe file; dataset generation is outside this build), or use --task synthetic' % path)
    with open(path, 'rb') as f:
        dataset = pickle.load(f)
    feature_dir = os.path.join('data', 'features', configs.task, configs.fv)
    features = load_video_features(feature_dir, configs.max_pos_len)
    if not features:
        raise ValueError('no *.npy video features under %s' % feature_dir)
    validate_dataset(dataset, configs)
    return dataset, features
