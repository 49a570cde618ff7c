

import os as _os

# The step forks work onto two side streams; together with torch's stream and RCCL's own streams that exceeds ROCm's default
# of 4 hardware queues per process, which silently serialises the streams (measured: +20 % step time once a process group
# exists).  Only effective if this package is imported before the first CUDA call of the process.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
