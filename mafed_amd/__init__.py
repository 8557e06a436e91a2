"""mafed_amd -- MI355X-native (gfx950) implementation of the MAFED per-step training hot path.

Host side (Python on PyTorch-ROCm for device memory / streams / torch.distributed) mirrors the reference's
plugin surface for this path (``CLStrategy`` / ``CLMethod`` registry, ``model_architecture``, HF-style
``model(**batch)`` outputs, a ``Trainer.step()`` in the reference's Lightning hook order); the arithmetic runs in
hand-written HIP kernels behind the C-ABI of ``include/mafed_hip.h`` (``libmafed_hip.so``).  There is no CPU path.
"""
__version__ = "0.1.0"

from mafed_amd.runtime_env import apply_recommended_runtime_env, runtime_env  # noqa: F401  (opt-in; importing the package sets nothing)
from mafed_amd.methods import CLMethod, CLStrategy, ER, EWC, FeatureDistillation, Naive  # noqa: F401
from mafed_amd.model import VLPythiaConfig, VLPythiaForCausalLM, model_architecture  # noqa: F401
from mafed_amd.optim import FlatAdamW, get_linear_schedule_with_warmup  # noqa: F401
from mafed_amd.trainer import Trainer  # noqa: F401
