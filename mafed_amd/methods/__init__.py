"""CL-method registry of the path (reference: mafed/methods/__init__.py:6-11).

``ewc`` is a different method from MAFED; it is built as the first "next" row.
"""
from mafed_amd.methods.base import CLStrategy, Naive
from mafed_amd.methods.distillation import FeatureDistillation
from mafed_amd.methods.distillation_loss_weights import DistillationWeights
from mafed_amd.methods.ewc import EWC
from mafed_amd.methods.memory import HBMReplayBuffer
from mafed_amd.methods.replay import ER

CLMethod = {
    "naive": Naive,
    "ewc": EWC,
    "replay": ER,
    "featdistill": FeatureDistillation,
}
