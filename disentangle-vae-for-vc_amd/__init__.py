"""disentangle-vae-for-vc_amd — the MI355X-native training hot path of the disentangled VAE for voice
conversion (reference: v-manhlt3/Disentangle-VAE-for-VC, model/disentangled_vae.py + the step loop of
model/variational_base_vae.py), behind the reference's own Python class surface.

The directory name is not a Python identifier; import it as `dvae_amd` (repo-root shim `dvae_amd.py`).

  csrc/ + libdvae_hip.so   hand-written HIP/CDNA4 kernels behind a C ABI (include/dvae_hip.h)
  _lib.py                  ctypes binding (fails loudly if the library is missing)
  ops.py                   autograd routing between the forward/backward kernels
  model/                   DisentangledVAE, ConvolutionalMulVAE, VariationalBaseModelVAE (reference API)
  optim.py                 flat-buffer Adam (one HBM-bound launch)
  ddp.py                   bucketed RCCL all-reduce of the flat gradient buffer, overlapped with backward
  data.py                  SpeechDatasetGVAE semantics + synthetic generators
"""
from . import _lib  # noqa: F401
from .model.disentangled_vae import (ConvNorm, ConvolutionalMulVAE, DisentangledVAE, LinearNorm, Postnet,  # noqa: F401
                                     init_weights)
from .model.variational_base_vae import VariationalBaseModelVAE  # noqa: F401
from .optim import FlatAdam  # noqa: F401

__version__ = "0.1.0"
