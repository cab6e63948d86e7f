"""patchmatchnet_amd -- MI355X-native learned-PatchMatch depth inference behind the PatchmatchNet interface.

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + C ABI, see include/pmn_hip.h), ``ops`` (checked
tensor wrappers), and the host-side mirror of the reference's module interface (``net``, ``patchmatch``, ``module``).
"""
from ._lib import PmnError, build, lib  # noqa: F401
from .module import differentiable_warping, depth_regression, is_empty  # noqa: F401
from .net import FeatureNet, PatchmatchNet, Refinement, adjust_image_dims  # noqa: F401
from .patchmatch import PatchMatch  # noqa: F401

__all__ = ["PatchmatchNet", "PatchMatch", "FeatureNet", "Refinement", "differentiable_warping", "depth_regression",
           "is_empty", "adjust_image_dims", "build", "lib", "PmnError"]
