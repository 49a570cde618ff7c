"""Evaluation loop, metrics and checkpoint housekeeping of the VSLNet path.

Mirrors `util/runner_utils_t7.py`: set_th_config :11-18, checkpoint helpers :21-45, convert_length_to_mask :48-52, IoU
metrics :55-68, eval_test :71-101.  Span extraction runs in the library's scan kernel (`VSLNet.extract_index` ->
vsl_extract_index), the metrics are host arithmetic on a few floats per sample."""
import glob
import os
import random

import numpy as np
import torch

from .data import index_to_time
from .synthetic import convert_length_to_mask     # noqa: F401  (re-exported: runner_utils_t7.py:48-52)


def set_th_config(seed):
    """runner_utils_t7.py:11-18 (the cudnn switches have no ROCm meaning and are left alone)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def _step_of(path, suffix):
    return int(os.path.basename(path).split('_')[1][:-(len(suffix) + 1)])


def filter_checkpoints(model_dir, suffix='t7', max_to_keep=5):
    """runner_utils_t7.py:21-32: keep the max_to_keep checkpoints with the highest step in `<name>_<step>.<suffix>`."""
    paths = sorted(glob.glob(os.path.join(model_dir, '*.{}'.format(suffix))), key=lambda p: _step_of(p, suffix))
    for p in paths[:max(0, len(paths) - max_to_keep)]:
        os.remove(p)


class _LazyState:
    """state_dict of a flat-bucket module from a host copy of the bucket (built on the worker thread)."""

    def __init__(self, model, host_flat, free=None):
        self.model, self.host, self.free = model, host_flat, free

    def build(self):
        try:
            return self.model.state_dict_from_flat(self.host.clone())    # a private copy ...
        finally:
            if self.free is not None:
                self.free.set()                                          # ... after which the pinned buffer may be reused (save_flat waits for this)


class CheckpointWriter:
    """`torch.save(model.state_dict(), path)` + `filter_checkpoints` (main_t7.py:125-126) off the training loop: the state is copied to
    pinned host memory on a side stream (the step's kernels keep running), a worker thread writes the file and prunes old ones.
    The reference saves whenever r1i7 >= the best so far, i.e. at every evaluation while the metric ties; a synchronous save costs
    ~14 ms (D2H of the state, pickle, file write) = 15 training steps at the headline shape (tools/e2e_rate.py)."""

    def __init__(self, device):
        import queue
        import threading
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == 'cuda' else None
        self.q = queue.Queue()
        self.err = None
        self.thread = threading.Thread(target=self._work, daemon=True)
        self.thread.start()

    def _work(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            state, event, path, model_dir, suffix, keep = item
            try:
                if event is not None:
                    event.synchronize()
                if isinstance(state, _LazyState):
                    state = state.build()
                tmp = path + '.tmp'                      # a crash / Ctrl-C mid-write must not leave a truncated *.t7 for get_last_checkpoint
                torch.save(state, tmp)
                os.replace(tmp, path)
                filter_checkpoints(model_dir, suffix=suffix, max_to_keep=keep)
            except Exception as e:                       # raised by the next save() / save_flat() / close()
                self.err = e
                if isinstance(state, _LazyState) and state.free is not None:
                    state.free.set()

    def _raise_pending(self):
        """A failed write stops the run at the next save, like the reference's synchronous torch.save would (main_t7.py:125)."""
        if self.err is not None:
            err, self.err = self.err, None
            raise err

    def save(self, state_dict, path, model_dir, suffix='t7', max_to_keep=3):
        self._raise_pending()
        if self.stream is None:
            self.q.put(({k: v.clone() for k, v in state_dict.items()}, None, path, model_dir, suffix, max_to_keep))
            return
        self.stream.wait_stream(torch.cuda.current_stream(self.device))          # the values as of this point of the training stream
        with torch.cuda.stream(self.stream):
            host = {}
            for k, v in state_dict.items():
                h = torch.empty(v.shape, dtype=v.dtype, device='cpu', pin_memory=True)
                h.copy_(v, non_blocking=True)
                host[k] = h
            ev = torch.cuda.Event()
            ev.record(self.stream)
        # the optimizer must not overwrite the parameters before the copies have read them
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        self.q.put((host, ev, path, model_dir, suffix, max_to_keep))

    def save_flat(self, model, path, model_dir, suffix='t7', max_to_keep=3):
        """The same for a module whose trainable parameters live in ONE flat bucket (vslnet_amd.model.VSLNet): a device-side snapshot of the
        bucket on the training stream (4.4 MB, microseconds; the optimizer may go on at once), ONE device-to-host copy of it on the side
        stream, and the worker thread rebuilds the reference's state_dict (same keys, shapes, order) from views of the host copy.  The
        per-tensor path above costs ~110 small copies the training stream has to wait for (3.4 ms per checkpoint)."""
        import threading
        self._raise_pending()
        flat, _ = model.flat_parameters
        snap = flat.clone()
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        if getattr(self, '_pin', None) is None or self._pin[0].numel() != snap.numel():
            self._pin = [torch.empty(snap.shape, dtype=snap.dtype, device='cpu', pin_memory=True) for _ in range(2)]
            self._pin_free = [threading.Event(), threading.Event()]
            for e in self._pin_free:
                e.set()
            self._pin_i = 0
        self._pin_i ^= 1
        # two buffers: the worker may still be writing the previous checkpoint.  Back-pressure: a buffer is reused only after the worker
        # has taken its private copy of it (blocks only when the writer lags two saves behind -- a slow disk, tiny epochs)
        free = self._pin_free[self._pin_i]
        free.wait()
        self._raise_pending()
        free.clear()
        with torch.cuda.stream(self.stream):
            host = self._pin[self._pin_i]
            host.copy_(snap, non_blocking=True)
            snap.record_stream(self.stream)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.q.put((_LazyState(model, host, free), ev, path, model_dir, suffix, max_to_keep))

    def close(self):
        self.q.put(None)
        self.thread.join()
        self._raise_pending()


def get_last_checkpoint(model_dir, suffix='t7'):
    """runner_utils_t7.py:35-45."""
    paths = glob.glob(os.path.join(model_dir, '*.{}'.format(suffix)))
    if not paths:
        raise ValueError('no *.%s checkpoint in %s' % (suffix, model_dir))
    return max(paths, key=lambda p: _step_of(p, suffix))


def calculate_iou(i0, i1):
    """runner_utils_t7.py:64-68: temporal IoU of two [start, end] intervals (hull as the union), floored at 0."""
    hull = max(i0[1], i1[1]) - min(i0[0], i1[0])
    inter = min(i0[1], i1[1]) - max(i0[0], i1[0])
    return max(0.0, 1.0 * inter / hull)


def calculate_iou_accuracy(ious, threshold):
    """runner_utils_t7.py:55-61: percentage of samples with IoU >= threshold."""
    ious = np.asarray(ious, dtype=np.float64)
    return float((ious >= threshold).sum()) / float(len(ious)) * 100.0


def summarise(ious, epoch=None, global_step=None):
    """The metric block of eval_test (runner_utils_t7.py:90-101) -> (r1i3, r1i5, r1i7, mIoU, score string)."""
    r1i3, r1i5, r1i7 = (calculate_iou_accuracy(ious, t) for t in (0.3, 0.5, 0.7))
    mi = float(np.mean(ious) * 100.0)
    text = 'Epoch {}, Step {}:\n'.format(epoch, global_step)
    text += 'Rank@1, IoU=0.3: {:.2f}\tRank@1, IoU=0.5: {:.2f}\tRank@1, IoU=0.7: {:.2f}\tmean IoU: {:.2f}\n'.format(r1i3, r1i5, r1i7, mi)
    return r1i3, r1i5, r1i7, mi, text


def eval_test(model, data_loader, device, mode='test', epoch=None, global_step=None):
    """runner_utils_t7.py:71-101.  The batch is moved with non-blocking copies (pinned staging when the loader pins), the
    indices of the whole batch come back in one D2H copy."""
    del mode
    ious = []
    with torch.no_grad():
        for records, vfeats, vfeat_lens, word_ids, char_ids in data_loader:
            vfeats, vfeat_lens = vfeats.to(device, non_blocking=True), vfeat_lens.to(device, non_blocking=True)
            word_ids, char_ids = word_ids.to(device, non_blocking=True), char_ids.to(device, non_blocking=True)
            query_mask = (word_ids != 0).float()
            video_mask = convert_length_to_mask(vfeat_lens)
            _, start_logits, end_logits = model(word_ids, char_ids, vfeats, video_mask, query_mask)
            s_idx, e_idx = model.extract_index(start_logits, end_logits)
            idx = torch.stack([s_idx, e_idx]).cpu().numpy()
            for r, si, ei in zip(records, idx[0], idx[1]):
                st, et = index_to_time(int(si), int(ei), r['v_len'], r['duration'])
                ious.append(calculate_iou([st, et], [r['s_time'], r['e_time']]))
    return summarise(ious, epoch, global_step)
