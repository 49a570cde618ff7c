"""ctypes binding of libvslnet_hip.so (include/vslnet_hip.h) + a thin engine object.

PyTorch-ROCm tensors are used for STORAGE ONLY: every pointer handed to the library is `tensor.data_ptr()` of a
caller-owned CUDA(HIP) tensor, and all kernels are enqueued on torch's current stream.  There is no CPU fallback:
if the shared library is missing or no MI355X is visible, construction raises.
"""
import ctypes as C
import os
from types import SimpleNamespace

import torch

from . import build as _build

_LIB = None


class vsl_config(C.Structure):
    _fields_ = [('dim', C.c_int32), ('num_heads', C.c_int32), ('max_pos_len', C.c_int32),
                ('video_feature_dim', C.c_int32), ('word_dim', C.c_int32), ('char_dim', C.c_int32),
                ('word_size', C.c_int32), ('char_size', C.c_int32), ('predictor', C.c_int32), ('drop_rate', C.c_float),
                ('word_table', C.c_int32)]


class vsl_io(C.Structure):
    _fields_ = [('B', C.c_int32), ('T', C.c_int32), ('Lq', C.c_int32), ('Lc', C.c_int32),
                ('params', C.c_void_p), ('pad_vec', C.c_void_p), ('glove_vec', C.c_void_p),
                ('word_ids', C.c_void_p), ('char_ids', C.c_void_p), ('video_features', C.c_void_p),
                ('v_mask', C.c_void_p), ('q_mask', C.c_void_p),
                ('h_score', C.c_void_p), ('start_logits', C.c_void_p), ('end_logits', C.c_void_p),
                ('workspace', C.c_void_p), ('training', C.c_int32), ('seed', C.c_uint64),
                ('d_h_score', C.c_void_p), ('d_start_logits', C.c_void_p), ('d_end_logits', C.c_void_p),
                ('grads', C.c_void_p), ('sample_offset', C.c_int32), ('video_features_bf16', C.c_void_p),
                ('early_grads_event', C.c_void_p),
                ('fused_step', C.c_void_p),        # (NULL unless Engine.backward(fused_step=...): an older build loaded for an A/B run reads zeros here)
                ('fused_loss', C.c_void_p)]        # (NULL unless the last loss() was lazy=True)


class vsl_loss_io(C.Structure):
    _fields_ = [('start_labels', C.c_void_p), ('end_labels', C.c_void_p), ('h_labels', C.c_void_p),
                ('w_loc', C.c_float), ('w_highlight', C.c_float), ('inv_batch', C.c_float), ('mask_sum', C.c_float),
                ('losses', C.c_void_p), ('d_h_score', C.c_void_p), ('d_start_logits', C.c_void_p),
                ('d_end_logits', C.c_void_p)]


class vsl_adamw(C.Structure):
    _fields_ = [('lr', C.c_float), ('beta1', C.c_float), ('beta2', C.c_float), ('eps', C.c_float), ('weight_decay', C.c_float),
                ('clip_norm', C.c_float), ('step', C.c_int32), ('hf_order', C.c_int32), ('norm_from_backward', C.c_int32)]


class vsl_fused_step(C.Structure):
    _fields_ = [('params', C.c_void_p), ('exp_avg', C.c_void_p), ('exp_avg_sq', C.c_void_p), ('hp', vsl_adamw), ('grad_norm_out', C.c_void_p)]


class VslError(RuntimeError):
    pass


ABI_SYMBOLS = ['vsl_last_error', 'vsl_create', 'vsl_destroy', 'vsl_param_count', 'vsl_param_info', 'vsl_param_floats',
               'vsl_workspace_floats', 'vsl_forward', 'vsl_loss', 'vsl_backward', 'vsl_extract_index',
               'vsl_adamw_step', 'vsl_workspace_offset', 'vsl_profile_select', 'vsl_profile_read', 'vsl_profile_launch', 'vsl_abi_version',
               'vsl_early_grad_offset', 'vsl_debug_rnn_launches']
ABI_VERSION = 8                                     # include/vslnet_hip.h: VSL_ABI_VERSION


def load_library():
    """dlopen libvslnet_hip.so (building it in-tree with hipcc when stale/missing).  Raises if that fails."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.environ.get('VSLNET_HIP_LIB')        # A/B runs: an alternative build of the same ABI
    if not path:
        path = _build.LIB
        if not os.path.exists(path) or _build._stale():
            # several ranks of one node may get here together (torch.distributed.run): one builds, the others wait for it
            import fcntl
            os.makedirs(_build.LIBDIR, exist_ok=True)
            with open(os.path.join(_build.LIBDIR, '.build.lock'), 'w') as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                try:
                    if not os.path.exists(path) or _build._stale():
                        path = _build.build()
                finally:
                    fcntl.flock(lock, fcntl.LOCK_UN)
    lib = C.CDLL(path)
    explicit = bool(os.environ.get('VSLNET_HIP_LIB'))     # an A/B baseline built from an older revision (tools/build_base.py) may predate the check:
    #                                                       new vsl_io fields are appended, so an older library simply does not read them
    if not explicit and (not hasattr(lib, 'vsl_abi_version') or lib.vsl_abi_version() != ABI_VERSION):
        raise VslError('%s implements another ABI version than this binding (%d): rebuild it (python -m vslnet_amd.build --force)' % (path, ABI_VERSION))
    if hasattr(lib, 'vsl_debug_rnn_launches'):
        lib.vsl_debug_rnn_launches.argtypes = [C.c_uint64]
        lib.vsl_debug_rnn_launches.restype = C.c_uint64
    if hasattr(lib, 'vsl_early_grad_offset'):
        lib.vsl_early_grad_offset.argtypes = [C.c_void_p]
        lib.vsl_early_grad_offset.restype = C.c_int64
    lib.vsl_last_error.restype = C.c_char_p
    lib.vsl_create.argtypes = [C.POINTER(vsl_config), C.POINTER(C.c_void_p)]
    lib.vsl_destroy.argtypes = [C.c_void_p]
    lib.vsl_param_count.argtypes = [C.c_void_p]
    lib.vsl_param_floats.argtypes = [C.c_void_p]
    lib.vsl_param_floats.restype = C.c_int64
    lib.vsl_param_info.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    lib.vsl_workspace_floats.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]
    lib.vsl_forward.argtypes = [C.c_void_p, C.POINTER(vsl_io), C.c_void_p]
    lib.vsl_loss.argtypes = [C.c_void_p, C.POINTER(vsl_io), C.POINTER(vsl_loss_io), C.c_void_p]
    lib.vsl_backward.argtypes = [C.c_void_p, C.POINTER(vsl_io), C.c_void_p]
    lib.vsl_extract_index.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.vsl_adamw_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(vsl_adamw), C.c_void_p, C.c_void_p]
    lib.vsl_workspace_offset.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p]
    lib.vsl_workspace_offset.restype = C.c_int64
    lib.vsl_profile_select.argtypes = [C.c_void_p, C.c_char_p]
    lib.vsl_profile_read.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    if hasattr(lib, 'vsl_profile_launch'):       # (absent from an older build loaded through VSLNET_HIP_LIB for an A/B run)
        lib.vsl_profile_launch.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                           C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    _LIB = lib
    return lib


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t, dtype, shape, name):
    if t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_cuda or not t.is_contiguous():
        raise ValueError('%s: expected contiguous cuda %s %s, got %s %s on %s' %
                         (name, dtype, tuple(shape), t.dtype, tuple(t.shape), t.device))


_HW_QUEUE_WARNED = False


def check_hw_queues():
    """A process group over RCCL brings its own streams; with fewer than 8 hardware queues the library's two side streams then share a
    queue with another stream and the fork / join overlap of the step is silently lost (measured: 1.25 -> 1.51 ms per step).  The
    variable only counts if it was in the environment before the process initialised HIP -- the library cannot enforce that, so it refuses
    to run in the state it can detect AND that costs: an nccl (= RCCL) group of more than one rank with GPU_MAX_HW_QUEUES below 8 or set too
    late.  Anything else -- a gloo group, one rank, an embedding application that initialised HIP first -- is a performance note, printed once;
    VSL_ALLOW_FEW_HW_QUEUES=1 turns the refusal into that note as well."""
    global _HW_QUEUE_WARNED
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return
    import vslnet_amd
    try:
        q = int(os.environ.get('GPU_MAX_HW_QUEUES', '0'))
    except ValueError:
        q = 0
    if q >= 8 and not vslnet_amd.QUEUES_SET_LATE:
        return
    msg = ('GPU_MAX_HW_QUEUES=%s%s while a torch.distributed process group exists: the step needs 8 hardware queues beside '
           "RCCL's streams.  Export GPU_MAX_HW_QUEUES=8 before the process starts (or import vslnet_amd before the first CUDA call)"
           % (os.environ.get('GPU_MAX_HW_QUEUES'), ' (set after HIP was initialised)' if vslnet_amd.QUEUES_SET_LATE else ''))
    rccl_multi = dist.get_backend() == 'nccl' and dist.get_world_size() > 1
    if rccl_multi and os.environ.get('VSL_ALLOW_FEW_HW_QUEUES') != '1':
        raise VslError(msg)
    if not _HW_QUEUE_WARNED:
        _HW_QUEUE_WARNED = True
        import warnings
        warnings.warn('vslnet_amd: ' + msg + ' -- continuing (%s)' % ('VSL_ALLOW_FEW_HW_QUEUES=1' if rccl_multi else 'no multi-rank RCCL group: nothing to lose'))


class Engine:
    """One `vsl_handle` + caller-owned buffers.  Mirrors what VSLNet.__init__ / forward / backward need."""

    def __init__(self, configs, device=None, word_table=None):
        """`word_table=True` (or configs.word_table): WordEmbedding(word_vectors=None), the trainable nn.Embedding branch
        (layers_t7.py:36); forward() then takes pad_vec = glove_vec = None."""
        if not torch.cuda.is_available():
            raise VslError('vslnet_amd needs an MI355X (gfx950) visible to PyTorch-ROCm; there is no CPU fallback')
        check_hw_queues()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.lib = load_library()
        # vsl_io.fused_step (ABI 8): an older build loaded through VSLNET_HIP_LIB for an A/B run takes the two calls instead
        self.fused_tail = hasattr(self.lib, 'vsl_abi_version') and self.lib.vsl_abi_version() >= 8
        pred = {'rnn': 0, 'transformer': 1}.get(configs.predictor)
        if pred is None:
            raise ValueError('unknown predictor %r' % (configs.predictor,))
        self.cfg = vsl_config(int(configs.dim), int(configs.num_heads), int(configs.max_pos_len),
                              int(configs.video_feature_dim), int(configs.word_dim), int(configs.char_dim),
                              int(configs.word_size), int(configs.char_size), pred, float(configs.drop_rate),
                              int(bool(getattr(configs, 'word_table', False) if word_table is None else word_table)))
        self.configs = configs
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            if self.lib.vsl_create(C.byref(self.cfg), C.byref(h)):
                msg = self.lib.vsl_last_error().decode()
                if 'not implemented' in msg:
                    raise NotImplementedError(msg)
                if 'not a multiple of attention heads' in msg:
                    raise AssertionError(msg)               # same condition the reference asserts (layers_t7.py:146)
                raise VslError(msg)
        self.h = h
        self.param_floats = int(self.lib.vsl_param_floats(h))
        self.layout = []                                    # (name, offset, numel, shape)
        name = C.create_string_buffer(256)
        off, num, nd = C.c_int64(), C.c_int64(), C.c_int32()
        dims = (C.c_int64 * 4)()
        for i in range(self.lib.vsl_param_count(h)):
            self._call(self.lib.vsl_param_info(h, i, name, 256, C.byref(off), C.byref(num), C.byref(nd), dims))
            self.layout.append((name.value.decode(), off.value, num.value, tuple(dims[j] for j in range(nd.value))))
        self._ws, self._grown = None, False
        self._last = None
        self._pending_loss = None

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                self.lib.vsl_destroy(self.h)
        except Exception:
            pass

    def _call(self, rc):
        if rc:
            msg = self.lib.vsl_last_error().decode()
            if 'exceeds max_pos_len' in msg:
                raise IndexError(msg)                       # the reference raises IndexError from nn.Embedding here
            raise VslError(msg)

    # ---- flat parameter bucket helpers -------------------------------------------------------------------
    def new_flat(self):
        return torch.zeros(self.param_floats, dtype=torch.float32, device=self.device)

    def views(self, flat):
        return {n: flat[o:o + k].view(shp) for n, o, k, shp in self.layout}

    def workspace(self, B, T, Lq, Lc):
        """ONE caller-owned workspace, grown to the largest shape seen: its contents only live from a forward to the
        backward of the same batch, and the collate narrows every batch to its own max T / Lq / Lc, so real-data training
        visits hundreds of shapes -- a workspace per shape would grow without bound beside the HBM-resident dataset."""
        n = C.c_int64()
        with torch.cuda.device(self.device):
            self._call(self.lib.vsl_workspace_floats(self.h, B, T, Lq, Lc, C.byref(n)))
        if self._ws is None or self._ws.numel() < n.value:
            # drop EVERY reference to the old one first (the last forward's io / keep-alive tuple hold it too): old and new must not
            # coexist at the growth peak beside an HBM-resident dataset
            self._ws = None
            self._last = self._last_ws = self._keep = None
            self._ws = torch.empty(int(n.value * 1.25) if self._grown else n.value, dtype=torch.float32, device=self.device)
            self._grown = True
        return self._ws

    def ws_view(self, name, shape):
        """Saved activation `name` of the LAST forward as a tensor view (parity tests)."""
        io = self._last
        off = self.lib.vsl_workspace_offset(self.h, io.B, io.T, io.Lq, io.Lc, name.encode())
        if off < 0:
            raise KeyError(name)
        n = 1
        for s in shape:
            n *= s
        return self._last_ws[off:off + n].view(shape)

    # ---- the three calls ------------------------------------------------------------------------------------
    def forward(self, flat, pad_vec, glove_vec, word_ids, char_ids, vfeats, v_mask, q_mask, training=False, seed=0,
                sample_offset=0):
        """`sample_offset`: index of this shard's first sample in the global batch (data parallel; vslnet_hip.h).
        `vfeats` in torch.bfloat16 selects the bf16 THROUGHPUT mode (vsl_io.video_features_bf16): bf16 features in HBM and a
        bf16-MFMA VisualProjection; everything downstream stays fp32.  Not the parity path.
        (Round 6 removed vsl_io.arithmetic, the one-product mode of VisualProjection and the weight gradients: it measured +0.5 % for a 5e-2
        logit tolerance, and its measured ceiling -- every conv-block GEMM as one product too -- was 1.07 x; profiles/r05_notes.md section 9.)"""
        B, T, Dv = vfeats.shape
        bf16 = vfeats.dtype == torch.bfloat16
        Lq, Lc = char_ids.shape[1], char_ids.shape[2]
        _chk(flat, torch.float32, (self.param_floats,), 'params')
        _chk(word_ids, torch.int64, (B, Lq), 'word_ids')
        _chk(char_ids, torch.int64, (B, Lq, Lc), 'char_ids')
        _chk(vfeats, torch.bfloat16 if bf16 else torch.float32, (B, T, self.cfg.video_feature_dim), 'video_features')
        _chk(v_mask, torch.float32, (B, T), 'v_mask')
        _chk(q_mask, torch.float32, (B, Lq), 'q_mask')
        if self.cfg.word_table:
            pad_vec = glove_vec = None                  # rows 0, 1, 2.. of the trainable table inside `flat` stand in
        else:
            _chk(pad_vec, torch.float32, (1, self.cfg.word_dim), 'pad_vec')
            _chk(glove_vec, torch.float32, (self.cfg.word_size - 2, self.cfg.word_dim), 'glove_vec')
        ws = self.workspace(B, T, Lq, Lc)
        out = torch.empty(3, B, T, dtype=torch.float32, device=self.device)
        io = vsl_io()
        io.B, io.T, io.Lq, io.Lc = B, T, Lq, Lc
        io.params, io.pad_vec, io.glove_vec = _ptr(flat), _ptr(pad_vec), _ptr(glove_vec)
        io.word_ids, io.char_ids = _ptr(word_ids), _ptr(char_ids)
        io.video_features, io.video_features_bf16 = (None, _ptr(vfeats)) if bf16 else (_ptr(vfeats), None)
        io.v_mask, io.q_mask = _ptr(v_mask), _ptr(q_mask)
        io.h_score, io.start_logits, io.end_logits = _ptr(out[0]), _ptr(out[1]), _ptr(out[2])
        io.workspace = _ptr(ws)
        io.training, io.seed, io.sample_offset = int(bool(training)), int(seed) & 0xFFFFFFFFFFFFFFFF, int(sample_offset)
        self._flush_loss()
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self._call(self.lib.vsl_forward(self.h, C.byref(io), stream))
        self._last, self._last_ws = io, ws
        # keep every tensor the io struct points to alive until the backward
        self._keep = (flat, pad_vec, glove_vec, word_ids, char_ids, vfeats, v_mask, q_mask, out, ws)
        return out[0], out[1], out[2]

    def loss(self, start_labels, end_labels, h_labels, w_loc=1.0, w_highlight=5.0, inv_batch=None, mask_sum=0.0,
             want_grads=True, scores=None, start_logits=None, end_logits=None, v_mask=None, lazy=False):
        """Fused compute_loss + compute_highlight_loss.  By default on the outputs of the LAST forward; explicit
        (B, T) tensors may be passed instead.  Returns (losses[4], d_h, d_sl, d_el).
        `lazy=True` (training steps: what `main.train` and `bench.py` do): nothing is launched here -- the loss becomes part of the next
        `backward()` on the returned seeds (vsl_io.fused_loss: for T >= 32 the loss kernel then leaves the dependent chain and the consumers of
        the seeds compute them from the logits); `losses` and the seeds are valid once that backward's work is.  Any other use of the engine
        in between (another forward / loss / a backward on other seeds) issues the pending loss first."""
        io = self._last
        if scores is not None:
            B, T = scores.shape
            for t, nm in ((scores, 'scores'), (start_logits, 'start_logits'), (end_logits, 'end_logits'), (v_mask, 'mask')):
                _chk(t, torch.float32, (B, T), nm)
            io2 = vsl_io()
            C.memmove(C.byref(io2), C.byref(io), C.sizeof(vsl_io))
            if (B, T) != (io.B, io.T):
                raise ValueError('loss inputs (%d, %d) do not match the last forward (%d, %d)' % (B, T, io.B, io.T))
            io2.h_score, io2.start_logits, io2.end_logits, io2.v_mask = _ptr(scores), _ptr(start_logits), _ptr(end_logits), _ptr(v_mask)
            io = io2
        B, T = io.B, io.T
        _chk(start_labels, torch.int64, (B,), 'start_labels')
        _chk(end_labels, torch.int64, (B,), 'end_labels')
        _chk(h_labels, torch.int64, (B, T), 'h_labels')
        losses = torch.empty(4, dtype=torch.float32, device=self.device)
        d = torch.empty(3, B, T, dtype=torch.float32, device=self.device) if want_grads else None
        l = vsl_loss_io()
        l.start_labels, l.end_labels, l.h_labels = _ptr(start_labels), _ptr(end_labels), _ptr(h_labels)
        l.w_loc, l.w_highlight = float(w_loc), float(w_highlight)
        l.inv_batch = float(1.0 / B if inv_batch is None else inv_batch)
        l.mask_sum = float(mask_sum)
        l.losses = _ptr(losses)
        if want_grads:
            l.d_h_score, l.d_start_logits, l.d_end_logits = _ptr(d[0]), _ptr(d[1]), _ptr(d[2])
        self._flush_loss()
        if lazy and want_grads and scores is None and self.fused_tail:
            self._pending_loss = (l, io, (start_labels, end_labels, h_labels, losses, d))
            return losses, d[0], d[1], d[2]
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self._call(self.lib.vsl_loss(self.h, C.byref(io), C.byref(l), stream))
        return (losses,) + ((d[0], d[1], d[2]) if want_grads else (None, None, None))

    def _flush_loss(self):
        """Issue a pending lazy loss() as its own launch (the caller did not go on to backward() on its seeds)."""
        pend, self._pending_loss = getattr(self, '_pending_loss', None), None
        if pend is not None:
            l, io, _ = pend
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            self._call(self.lib.vsl_loss(self.h, C.byref(io), C.byref(l), stream))

    def early_grad_offset(self):
        """First float offset of the block of `grads` that is final when `backward(..., early_event=)` fires its event."""
        return int(self.lib.vsl_early_grad_offset(self.h))

    def backward(self, d_h, d_sl, d_el, grads, early_event=None, fused_step=None):
        """Backward of the LAST forward; writes the flat gradient bucket `grads` (same layout as the params).
        `early_event` (torch.cuda.Event, data parallel): recorded by the library as soon as grads[early_grad_offset():] is final.
        `fused_step` (single process; `FlatAdamW.fused_step()`): dict(flat, exp_avg, exp_avg_sq, lr, step, betas, eps, weight_decay, clip_norm,
        hf_order[, grad_norm_out]) -- this optimizer step is applied inside the backward's last launch (final reduction + clip + AdamW in one
        kernel); equivalent to backward() followed by adamw_step(..., norm_from_backward=True)."""
        io = self._last
        B, T = io.B, io.T
        pend = getattr(self, '_pending_loss', None)
        fused_loss = None
        if pend is not None:
            d = pend[2][4]
            if pend[1] is io and d_h is not None and d_h.data_ptr() == d[0].data_ptr() and d_sl.data_ptr() == d[1].data_ptr() and d_el.data_ptr() == d[2].data_ptr():
                fused_loss, self._pending_loss = pend, None          # the lazy loss rides in this call
            else:
                self._flush_loss()
        if d_h is not None:
            _chk(d_h, torch.float32, (B, T), 'd_h_score')
        _chk(d_sl, torch.float32, (B, T), 'd_start_logits')
        _chk(d_el, torch.float32, (B, T), 'd_end_logits')
        _chk(grads, torch.float32, (self.param_floats,), 'grads')
        io.d_h_score, io.d_start_logits, io.d_end_logits, io.grads = _ptr(d_h), _ptr(d_sl), _ptr(d_el), _ptr(grads)
        io.early_grads_event = None
        if early_event is not None:
            early_event.record(torch.cuda.current_stream(self.device))     # creates the handle lazily; harmless: recorded again by the library
            io.early_grads_event = C.c_void_p(early_event.cuda_event)
        io.fused_loss = C.cast(C.pointer(fused_loss[0]), C.c_void_p) if fused_loss is not None else None
        io.fused_step = None
        fs = None
        if fused_step is not None:
            if early_event is not None:
                raise ValueError('fused_step is the single-process step; early_event belongs to the data-parallel exchange')
            n = self.param_floats
            for nm in ('flat', 'exp_avg', 'exp_avg_sq'):
                _chk(fused_step[nm], torch.float32, (n,), nm)
            b = fused_step.get('betas', (0.9, 0.999))
            fs = vsl_fused_step()
            fs.params, fs.exp_avg, fs.exp_avg_sq = _ptr(fused_step['flat']), _ptr(fused_step['exp_avg']), _ptr(fused_step['exp_avg_sq'])
            fs.hp = vsl_adamw(float(fused_step['lr']), float(b[0]), float(b[1]), float(fused_step.get('eps', 1e-6)),
                              float(fused_step.get('weight_decay', 0.01)), float(fused_step.get('clip_norm', 1.0) or 0.0), int(fused_step['step']),
                              int(bool(fused_step.get('hf_order', False))), 1)
            gn = fused_step.get('grad_norm_out')
            fs.grad_norm_out = _ptr(gn) if gn is not None else None
            io.fused_step = C.cast(C.pointer(fs), C.c_void_p)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        try:
            self._call(self.lib.vsl_backward(self.h, C.byref(io), stream))
        finally:
            io.fused_step = io.fused_loss = None          # (`fs` / the loss struct live until here; the library reads them during the call only)
        return grads

    def profile_select(self, kernel):
        """Time launches of `kernel` ('*' = all, None = off) with HIP events on the launch stream."""
        self._call(self.lib.vsl_profile_select(self.h, None if kernel is None else kernel.encode()))

    def profile_read(self):
        """-> {kernel: (total_ms, launches)} since the last profile_select."""
        out, name, ms, n = {}, C.create_string_buffer(64), C.c_double(), C.c_int32()
        i = 0
        while True:
            rc = self.lib.vsl_profile_read(self.h, i, name, 64, C.byref(ms), C.byref(n))
            if rc == 2:
                break
            self._call(rc)
            out[name.value.decode()] = (ms.value, n.value)
            i += 1
        return out

    def profile_launches(self):
        """-> [dict(name, stream, start_us, stop_us, host_us, deps)] of every profiled launch since the last profile_select, in enqueue
        order (tools/critical_path.py)."""
        out, name = [], C.create_string_buffer(64)
        st, t0, t1, th, nd = C.c_int32(), C.c_double(), C.c_double(), C.c_double(), C.c_int32()
        deps = (C.c_int32 * 6)()
        i = 0
        while True:
            rc = self.lib.vsl_profile_launch(self.h, i, name, 64, C.byref(st), C.byref(t0), C.byref(t1), C.byref(th), deps, C.byref(nd))
            if rc == 2:
                break
            self._call(rc)
            out.append(dict(name=name.value.decode(), stream=st.value, start_us=t0.value, stop_us=t1.value, host_us=th.value,
                            deps=[deps[k] for k in range(nd.value)]))
            i += 1
        return out

    def adamw_step(self, flat, grads, exp_avg, exp_avg_sq, lr, step, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01,
                   clip_norm=1.0, grad_norm_out=None, hf_order=False, norm_from_backward=False):
        """clip_grad_norm_ + AdamW on the flat buckets (main_t7.py:111-112), two kernels, no host synchronisation.
        `step` is 1-based; `grad_norm_out` (optional 1-element device tensor) receives the un-clipped global norm.
        `norm_from_backward=True`: `grads` is untouched since this engine's last backward() (single-GPU training, no exchange): the
        norm comes from the sums of squares the backward's final reduction recorded -- one kernel less."""
        n = self.param_floats
        for t, nm in ((flat, 'flat'), (grads, 'grads'), (exp_avg, 'exp_avg'), (exp_avg_sq, 'exp_avg_sq')):
            _chk(t, torch.float32, (n,), nm)
        hp = vsl_adamw(float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), float(clip_norm or 0.0), int(step),
                       int(bool(hf_order)), int(bool(norm_from_backward)))
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self._call(self.lib.vsl_adamw_step(self.h, _ptr(flat), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), C.byref(hp),
                                           _ptr(grad_norm_out) if grad_norm_out is not None else None, stream))

    def extract_index(self, start_logits, end_logits):
        B, T = start_logits.shape
        _chk(start_logits, torch.float32, (B, T), 'start_logits')
        _chk(end_logits, torch.float32, (B, T), 'end_logits')
        idx = torch.empty(2, B, dtype=torch.int64, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self._call(self.lib.vsl_extract_index(self.h, _ptr(start_logits), _ptr(end_logits), B, T, _ptr(idx[0]),
                                              _ptr(idx[1]), stream))
        return idx[0], idx[1]


def flat_from_state_dict(engine, sd):
    """Pack a reference-style state_dict (name -> tensor) into the engine's flat parameter bucket."""
    flat = engine.new_flat()
    for n, o, k, shp in engine.layout:
        flat[o:o + k] = sd[n].reshape(-1).to(device=engine.device, dtype=torch.float32)
    return flat
