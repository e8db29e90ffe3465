"""geniconet_amd: MI355X-native icosahedral hex-convolution encoder/decoder hot path of GenIcoNet.

Python host code over libicn.so (hand-written gfx950 HIP kernels behind the C ABI of include/icn.h).
"""
from . import _lib, geometry  # noqa: F401
from .ico_conv import IcoConvS2S, IcoUpsampleS2S, ico_conv, ico_upsample  # noqa: F401

__all__ = ['IcoConvS2S', 'IcoUpsampleS2S', 'ico_conv', 'ico_upsample', 'geometry']
