"""pysfm_amd - the bundle-adjustment inner loop of alexflint/pysfm on MI355X.

    from pysfm_amd import Bundle, BundleAdjuster, sensor_model
    ba = BundleAdjuster(bundle)          # same API as pysfm's bundle_adjuster.BundleAdjuster
    ba.optimize(max_steps=25)

All arithmetic runs in hand-written HIP kernels (pysfm_amd/csrc) behind the C ABI
of include/pysfm_ba.h; importing the package does not need a GPU, using it does.
"""
from . import sensor_model                                     # noqa: F401
from .bundle import Bundle, Camera, Track                      # noqa: F401
from .bundle_adjuster import BundleAdjuster, NormalEquationsIllconditioned  # noqa: F401

__all__ = ['Bundle', 'Camera', 'Track', 'BundleAdjuster', 'NormalEquationsIllconditioned', 'sensor_model']
