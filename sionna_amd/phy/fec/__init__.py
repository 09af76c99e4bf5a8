"""Forward error correction blocks of the hot path (LDPC 5G, Polar 5G, CRC, scrambling,
row/column interleaving)."""
from . import ldpc
from . import polar, crc, scrambling, interleaving
