"""icocnn.ico_conv: IcoConvS2S, IcoUpsampleS2S (see geniconet_amd.ico_conv)."""
from geniconet_amd.ico_conv import IcoConvS2S, IcoUpsampleS2S, ico_conv, ico_upsample  # noqa: F401
