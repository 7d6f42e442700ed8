"""Importable alias for the hyphenated package directory ``loco-asr_amd/``:

    import loco_asr_amd as la
    enc = la.SpeechT5EncoderWithSpeechPrenetMI355X()
"""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module("loco-asr_amd")
