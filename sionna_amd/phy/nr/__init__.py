"""5G NR transport-block chain (mirror of the ``TBEncoder`` / ``TBDecoder`` part of ``sionna.phy.nr``)."""
from .utils import generate_prng_seq, calculate_num_coded_bits, calculate_tb_size
from .tb_encoder import TBEncoder
from .tb_decoder import TBDecoder
