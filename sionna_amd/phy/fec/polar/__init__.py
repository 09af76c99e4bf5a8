"""Polar codes of the hot path (mirror of ``sionna.phy.fec.polar``)."""
from .utils import generate_5g_ranking
from .encoding import PolarEncoder, Polar5GEncoder
from .decoding import PolarSCDecoder, PolarSCLDecoder, PolarBPDecoder, Polar5GDecoder
