from .optispeech import OptiSpeech  # noqa: F401
