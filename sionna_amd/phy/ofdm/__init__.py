"""OFDM resource grid, pilots, channel estimation and equalisation (mirror of
``sionna.phy.ofdm`` for the hot path)."""
from .pilot_pattern import PilotPattern, EmptyPilotPattern, KroneckerPilotPattern
from .resource_grid import ResourceGrid, ResourceGridMapper, ResourceGridDemapper, RemoveNulledSubcarriers
from .channel_estimation import (LSChannelEstimator, NearestNeighborInterpolator, LinearInterpolator, LMMSEInterpolator,
                                 tdl_freq_cov_mat, tdl_time_cov_mat)
from .equalization import OFDMEqualizer, LMMSEEqualizer, ZFEqualizer, MFEqualizer
from .detection import LinearDetector, MMSEPICDetector, EPDetector, KBestDetector, MaximumLikelihoodDetector, \
    MaximumLikelihoodDetectorWithPrior
from .modulator import OFDMModulator, OFDMDemodulator
