"""MI355X-native SmartVidCrop saliency-to-crop hot path (drop-in for bmezaris/RetargetVid's
smartVidCrop.py entry points).  See DESIGN.md and INTEGRATION.md."""
__version__ = '0.1.0'

import os as _os

# Videos (and batches) in flight live on HIP streams of their own; ROCm gives a process four hardware queues unless told
# otherwise, so a fifth stream shares a queue with another one and the two run one after the other (measured: config 3 with
# four worker streams 1.53 -> 1.33 s per 100 videos, the pipelined bench 2.3 -> 1.25 ms per step).  Read by the HIP runtime
# when it starts, i.e. this import must come before the first GPU call of the process; a value already set wins.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
