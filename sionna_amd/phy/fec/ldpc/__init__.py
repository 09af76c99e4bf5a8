"""5G-NR LDPC encoder and BP decoders (mirror of ``sionna.phy.fec.ldpc``, reference fec/ldpc/__init__.py:7-10)."""
from .encoding import LDPC5GEncoder
from .decoding import LDPCBPDecoder, LDPC5GDecoder
from .custom import (cn_update_minsum, cn_update_phi, cn_update_tanh, cn_update_offset_minsum, vn_update_sum,
                     EXITCallback, DecoderStatisticsCallback, WeightedBPCallback, RaggedMessages)
