"""5G-NR LDPC encoder and BP decoders (mirror of ``sionna.phy.fec.ldpc``)."""
from .encoding import LDPC5GEncoder
from .decoding import LDPCBPDecoder, LDPC5GDecoder
