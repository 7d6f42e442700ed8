"""loco-asr_amd -- MI355X-native SpeechT5 speech-encoder embedding path of keya-dialog/LoCo-ASR.

The directory name carries a hyphen (it is fixed by the build contract), so import it with
``importlib.import_module("loco-asr_amd")`` or through the alias module ``loco_asr_amd`` at the
repository root.  Importing the package is cheap and works without a GPU; constructing the encoder
loads the HIP library (``libloco_asr.so``) and fails loudly if it has not been built.
"""
from . import synth  # noqa: F401
from ._lib import LIB_PATH, LocoError  # noqa: F401
from .encoder import (BaseModelOutput, Pack, SpeechT5EncoderWithSpeechPrenetMI355X,  # noqa: F401
                      SpeechT5ForSpeechToTextMI355X, sinusoid_table)
from .feature_extractor import BatchFeature, SpeechT5FeatureExtractorMI355X  # noqa: F401
from .intent_head import IntentClassifierMI355X  # noqa: F401
from .text_encoder import (SpeechT5EncoderWithTextPrenetMI355X, SpeechT5ForTextToSpeechMI355X,  # noqa: F401
                           scaled_positional_table)

__all__ = ["synth", "LIB_PATH", "LocoError", "BaseModelOutput", "Pack", "SpeechT5EncoderWithSpeechPrenetMI355X",
           "SpeechT5ForSpeechToTextMI355X", "sinusoid_table", "BatchFeature", "SpeechT5FeatureExtractorMI355X", "IntentClassifierMI355X",
           "SpeechT5EncoderWithTextPrenetMI355X", "SpeechT5ForTextToSpeechMI355X", "scaled_positional_table"]
