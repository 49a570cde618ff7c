import os as _os

# The step forks work onto two side streams; together with torch's stream and RCCL's own streams that exceeds ROCm's default
# of 4 hardware queues per process, which silently serialises the streams (measured: +20 % step time once a process group
# exists).  Only effective if this package is imported before the first CUDA call of the process.
# QUEUES_SET_LATE: the variable was absent AND the process had already initialised HIP when this package was imported, so the default
# below cannot take effect any more; Engine() refuses to run beside a process group in that state (vslnet_amd.engine.check_hw_queues).
import sys as _sys
QUEUES_SET_LATE = False
if 'GPU_MAX_HW_QUEUES' not in _os.environ:
    _t = _sys.modules.get('torch')
    QUEUES_SET_LATE = bool(_t is not None and _t.cuda.is_initialized())
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
