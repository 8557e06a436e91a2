"""CL-method registry of the path (reference: mafed/methods/__init__.py:6-11).

``ewc`` is registered upstream but is a different method, outside the MAFED hot path.
"""
from mafed_amd.methods.base import CLStrategy, Naive
from mafed_amd.methods.distillation import FeatureDistillation
from mafed_amd.methods.distillation_loss_weights import DistillationWeights
from mafed_amd.methods.memory import HBMReplayBuffer
from mafed_amd.methods.replay import ER

CLMethod = {
    "naive": Naive,
    "replay": ER,
    "featdistill": FeatureDistillation,
}
