"""Build vslnet_amd/lib/libvslnet_hip_<name>.so (default name: base) from the sources of a git revision (default HEAD), for same-box A/B runs:
    python tools/build_base.py <rev> [name]
    VSLNET_HIP_LIB=vslnet_amd/lib/libvslnet_hip_base.so python bench.py ...   vs   python bench.py ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rev = sys.argv[1] if len(sys.argv) > 1 else 'HEAD'
name = sys.argv[2] if len(sys.argv) > 2 else 'base'          # -> vslnet_amd/lib/libvslnet_hip_<name>.so
tmp = '/tmp/vsl_base_src'
files = ['vslnet_amd/csrc/api.hip', 'vslnet_amd/csrc/kernels_fwd.hip', 'vslnet_amd/csrc/kernels_bwd.hip', 'vslnet_amd/csrc/kernels_enc.hip',
         'vslnet_amd/csrc/kernels_wgrad.hip', 'vslnet_amd/csrc/kernels_split.hip', 'vslnet_amd/csrc/kernels_lstm.hip', 'vslnet_amd/csrc/kernels_query.hip', 'vslnet_amd/csrc/common.hpp', 'vslnet_amd/csrc/launch.hpp', 'vslnet_amd/csrc/tile_bodies.hpp', 'include/vslnet_hip.h']
for f in files:
    os.makedirs(os.path.dirname(os.path.join(tmp, f)), exist_ok=True)
    try:
        data = subprocess.check_output(['git', 'show', '%s:%s' % (rev, f)], cwd=ROOT, stderr=subprocess.DEVNULL)
    except subprocess.CalledProcessError:       # the revision predates this file
        if os.path.exists(os.path.join(tmp, f)):
            os.remove(os.path.join(tmp, f))
        continue
    open(os.path.join(tmp, f), 'wb').write(data)
from vslnet_amd import build  # noqa: E402
print(build.build(csrc=os.path.join(tmp, 'vslnet_amd/csrc'), out=os.path.join(ROOT, 'vslnet_amd/lib/libvslnet_hip_%s.so' % name)))
