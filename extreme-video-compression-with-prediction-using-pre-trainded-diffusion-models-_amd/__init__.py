"""evc_amd -- MI355X-native decode hot path of "Extreme Video Compression With Prediction Using
Pre-trained Diffusion Models": conditional video-diffusion sampler (NCSN++ "unetmore" score network +
DDPM / DDIM / F-PNDM loops) and the ELIC key-frame decoder, as hand-written HIP kernels for gfx950
behind a C ABI (include/evc_hip.h, include/evc_rans.h), driven by thin Python host code.

Import through the repo-root alias module: ``import evc_amd``.  There is no CPU fallback: every
compute entry point raises if ``libevc_hip.so`` is missing or no gfx950 device is visible.
"""
__version__ = "0.1.0"
