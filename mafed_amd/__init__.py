"""mafed_amd -- MI355X-native (gfx950) implementation of the MAFED per-step training hot path.

Host side (Python on PyTorch-ROCm for device memory / streams / torch.distributed) mirrors the reference's
plugin surface for this path (``CLStrategy`` / ``CLMethod`` registry, ``model_architecture``, HF-style
``model(**batch)`` outputs, a ``Trainer.step()`` in the reference's Lightning hook order); the arithmetic runs in
hand-written HIP kernels behind the C-ABI of ``include/mafed_hip.h`` (``libmafed_hip.so``).  There is no CPU path.
"""
__version__ = "0.1.0"

import os as _os

# HIP runtime option (read when the runtime initialises; a value the user has set wins): kernel arguments in device memory, which
# shortens the launch-to-launch latency the hand-scheduled step is sensitive to
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")   # the runtime's default, pinned: the stream plan of the step assumes it (5 queues: +20 % step time)

from mafed_amd.methods import CLMethod, CLStrategy, ER, EWC, FeatureDistillation, Naive  # noqa: F401
from mafed_amd.model import VLPythiaConfig, VLPythiaForCausalLM, model_architecture  # noqa: F401
from mafed_amd.optim import FlatAdamW, get_linear_schedule_with_warmup  # noqa: F401
from mafed_amd.trainer import Trainer  # noqa: F401
