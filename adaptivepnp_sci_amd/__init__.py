"""adaptivepnp_sci_amd -- MI355X-native adaptive plug-and-play ADMM reconstruction for Bayer video
snapshot compressive imaging: hand-written HIP kernels (gfx950) behind a C ABI (include/scipnp.h),
driven through drop-in equivalents of the reference's solver entry points.

    from adaptivepnp_sci_amd import twoStageAdmm_denoise_bayer, admm_denoise_bayer_demosaic_pre

Importing the package does not touch the GPU; the first solver/op call loads libscipnp.so and
raises if the library or the device is missing (there is no CPU fallback).
"""
from .solver import (admm_denoise, admm_denoise_bayer_demosaic_pre, admm_denoise_gray, gap_denoise,  # noqa: F401
                     twoStageAdmm_denoise_bayer)
from .denoisers import fastdvdnet_denoiser_full_tensor_v2, ffdnet_rgb_denoise_full_tensor, test_ddnet  # noqa: F401
from .ddnet import DDnet  # noqa: F401
from .fastdvd import FastDVDnet  # noqa: F401
from .nets import FFDNet  # noqa: F401

__version__ = '0.6.0'
