"""MIMO stream bookkeeping and linear equalisation/detection (mirror of ``sionna.phy.mimo``
for the hot path: StreamManagement, lmmse_equalizer, LinearDetector("lmmse"))."""
from .stream_management import StreamManagement
from .equalization import lmmse_matrix, lmmse_equalizer, zf_equalizer, mf_equalizer
from .utils import (whiten_channel, complex2real_vector, real2complex_vector, complex2real_matrix, real2complex_matrix,
                    complex2real_covariance, real2complex_covariance, complex2real_channel, real2complex_channel)
from .detection import LinearDetector, MMSEPICDetector, EPDetector, KBestDetector, MaximumLikelihoodDetector
