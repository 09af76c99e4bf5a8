"""Channel models of the hot path (mirror of ``sionna.phy.channel``)."""
from .awgn import AWGN
from .utils import subcarrier_frequencies, cir_to_ofdm_channel
from .ofdm_channel import GenerateOFDMChannel, ApplyOFDMChannel, OFDMChannel, RayleighBlockFading
from . import tr38901
