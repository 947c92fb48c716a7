"""MI355X-native bi-TAI hot path (drop-in for the bi-TAI path of MichiganCOG/video-frame-inpainting).

Layout (mirrors the reference's modules for this path):
  csrc/ + libtai_sepconv.so   HIP kernels behind the C ABI of include/tai_sepconv.h
  separable_convolution       the autograd op (reference src/separable_convolution/SeparableConvolution.py)
  mcnet, tai, create_model    the model (reference src/models/{mcnet,tai}/, create_model.py)
  environments, options       eval / training environments and flags (reference src/environments, src/options)
  losses, sn_discriminator    training-path losses (reference src/losses, src/discriminators)
  metrics, synthetic          PSNR/SSIM definition of the reference; seeded synthetic clips
  graph, parallel             hipGraph capture; clip-sharded data parallelism over RCCL
"""
from .create_model import create_model, supported_model_keys  # noqa: F401
from .separable_convolution import SeparableConvolution  # noqa: F401
from .tai import TAIFillInModel  # noqa: F401
from .mcnet import MCNetFillInModel  # noqa: F401

__version__ = '0.1.0'
