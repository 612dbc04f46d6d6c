"""MI355X-native appearance-motion memory-consistency path (AMMC-Net, AAAI 2021).

`unet` mirrors the reference's model interface (Code/models/unet.py); the
compute lives in libammc_hip.so (csrc/, C ABI in include/ammc_hip.h).
"""
from .unet import (UNet, UNetMem_v7, Quantize_topk, bridge, double_conv, down,  # noqa: F401
                   enc_quan_dec_res_topk, enc_quan_dec_topk, get_twostream, get_unet,
                   get_unet_vq_topk_res, inconv, twostream, up)

from .discriminator import PixelDiscriminator  # noqa: F401
from .flownet import FlowNet2SD  # noqa: F401

__all__ = ["FlowNet2SD", "PixelDiscriminator", "UNet", "UNetMem_v7", "Quantize_topk", "bridge", "double_conv", "down", "enc_quan_dec_res_topk",
           "enc_quan_dec_topk", "get_twostream", "get_unet", "get_unet_vq_topk_res", "inconv", "twostream", "up"]
