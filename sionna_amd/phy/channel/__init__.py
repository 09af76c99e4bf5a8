"""Channel models of the hot path (mirror of ``sionna.phy.channel``)."""
from .awgn import AWGN
from .utils import subcarrier_frequencies, cir_to_ofdm_channel
from .ofdm_channel import GenerateOFDMChannel, ApplyOFDMChannel, OFDMChannel, RayleighBlockFading
from .time_channel import (time_lag_discrete_time_channel, cir_to_time_channel, GenerateTimeChannel,
                           ApplyTimeChannel, TimeChannel)
from .flat_fading_channel import GenerateFlatFadingChannel, ApplyFlatFadingChannel, FlatFadingChannel
from .spatial_correlation import SpatialCorrelation, KroneckerModel, PerColumnModel, exp_corr_mat, one_ring_corr_mat
from . import tr38901
