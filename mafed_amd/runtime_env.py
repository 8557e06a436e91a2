"""HIP runtime options the step was tuned with -- an explicit opt-in, never applied by importing the package.

The HIP runtime reads both when it initialises (first HIP call of the process), so they must be in the environment before anything
touches the GPU; they apply to every HIP user of the process, which is why a library import must not set them behind the caller's back.

    import mafed_amd
    mafed_amd.apply_recommended_runtime_env()      # before the first torch.cuda call; a value already exported wins

``HIP_FORCE_DEV_KERNARG=1``  kernel arguments in device memory: shorter launch-to-launch dependency latency.
``GPU_MAX_HW_QUEUES=4``      the stream plan of the step (main | teacher + dW | dW | optimiser + loader) assumes four hardware queues;
                             tuned with ONE rank per process and no RCCL streams -- re-measure with --hw-queues under N > 1.
"""
from __future__ import annotations

import os
import warnings
from typing import Dict, Optional

RECOMMENDED = {"HIP_FORCE_DEV_KERNARG": "1", "GPU_MAX_HW_QUEUES": "4"}


def _hip_initialised() -> bool:
    try:
        import torch
        return bool(torch.cuda.is_initialized())
    except Exception:
        return False


def apply_recommended_runtime_env(overrides: Optional[Dict[str, str]] = None) -> Dict[str, Optional[str]]:
    """Sets the options above unless the variable is already exported (``overrides`` replaces a recommended value and does win over
    the environment).  Warns -- and changes nothing the runtime will see -- when HIP is already initialised.  Returns runtime_env()."""
    late = _hip_initialised()
    want = dict(RECOMMENDED)
    for k, v in (overrides or {}).items():
        want[k] = str(v)
    for k, v in want.items():
        forced = overrides is not None and k in overrides
        if late and (forced or k not in os.environ):
            warnings.warn(f"mafed_amd: {k}={v} requested after the HIP runtime initialised -- it has no effect in this process", RuntimeWarning,
                          stacklevel=2)
            continue
        if forced:
            os.environ[k] = v
        else:
            os.environ.setdefault(k, v)
    return runtime_env()


def runtime_env() -> Dict[str, Optional[str]]:
    """What the process environment holds for the tuned options (None = the runtime's default)."""
    keys = list(RECOMMENDED) + ["NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS", "HSA_ENABLE_IPC_MODE_LEGACY"]
    return {k: os.environ.get(k) for k in keys}
