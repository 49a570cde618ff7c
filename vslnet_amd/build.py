"""Builds libvslnet_hip.so (hand-written HIP for gfx950) in-tree with hipcc.  No GPU needed to build."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libvslnet_hip.so')
SOURCES = ['kernels_fwd.hip', 'kernels_bwd.hip', 'kernels_enc.hip', 'kernels_wgrad.hip', 'kernels_split.hip', 'kernels_lstm.hip', 'kernels_query.hip', 'api.hip']
HEADERS = ['common.hpp', 'launch.hpp', os.path.join('..', '..', 'include', 'vslnet_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-Wno-pass-failed']
# per-file flags.  kernels_wgrad.hip: the split-bf16 loop is hand-scheduled scalar fp32 code; SLP vectorisation turns its subtractions into
# v_pk_add_f32 + v_mov packing, which is slower beside MFMAs (MI355X_MICROARCH.md, price of fillers)
FILE_FLAGS = {'kernels_wgrad.hip': ['-fno-slp-vectorize'], 'kernels_split.hip': ['-fno-slp-vectorize'], 'kernels_enc.hip': ['-fno-slp-vectorize'], 'kernels_query.hip': ['-fno-slp-vectorize']}


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, csrc=None, out=None, stamps=False):
    """Compile every HIP translation unit for gfx950 and link the shared library.  Returns its path.
    `csrc` / `out` build another source tree into another file (baseline builds for same-box A/B runs).
    `stamps=True` builds lib/libvslnet_hip_stamps.so with the in-kernel phase stamps compiled in (-DVSL_STAMPS; run with
    VSLNET_HIP_LIB=<that file> VSL_DEBUG_TIMING=1): the product library carries none, a disabled stamp still costs a memory round trip."""
    if stamps and out is None:
        out = os.path.join(LIBDIR, 'libvslnet_hip_stamps.so')
    src_dir, lib = csrc or CSRC, out or LIB
    if csrc is None and out is None and not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    tag = '' if out is None else '.' + os.path.basename(out)

    def cc(src):
        obj = os.path.join(LIBDIR, src.replace('.hip', tag + '.o'))
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(src, []) + (['-DVSL_STAMPS'] if stamps else []) + os.environ.get('VSL_EXTRA_HIPCC_FLAGS', '').split() + ['-c', os.path.join(src_dir, src), '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr[-4000:]))
        if verbose and r.stderr:
            sys.stderr.write(r.stderr)
        return obj

    srcs = [s for s in SOURCES if os.path.exists(os.path.join(src_dir, s))]      # (a baseline revision may predate a translation unit)
    with ThreadPoolExecutor(len(srcs)) as ex:
        objs = list(ex.map(cc, srcs))
    r = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', lib], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stderr[-4000:])
    return lib


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True, stamps='--stamps' in sys.argv))
