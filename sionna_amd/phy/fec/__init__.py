"""Forward error correction blocks of the hot path (LDPC 5G, Polar 5G, CRC, scrambling,
row/column interleaving, generic linear encoder, code utilities)."""
from . import ldpc
from . import polar, crc, scrambling, interleaving, linear, utils
