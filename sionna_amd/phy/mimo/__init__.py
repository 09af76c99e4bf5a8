"""MIMO stream bookkeeping and linear equalisation/detection (mirror of ``sionna.phy.mimo``
for the hot path: StreamManagement, lmmse_equalizer, LinearDetector("lmmse"))."""
from .stream_management import StreamManagement
from .equalization import lmmse_equalizer, zf_equalizer, mf_equalizer
from .detection import LinearDetector, MMSEPICDetector, EPDetector, KBestDetector, MaximumLikelihoodDetector
