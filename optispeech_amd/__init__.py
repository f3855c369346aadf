"""optispeech_amd: MI355X-native (gfx950) implementation of the OptiSpeech ConvNeXt training-step and
``synthesise`` hot path behind the reference's Python API.  All arithmetic on the path runs in
hand-written HIP kernels (optispeech_amd/csrc) reached through the C ABI in include/osp.h."""
__version__ = "0.1.0"
