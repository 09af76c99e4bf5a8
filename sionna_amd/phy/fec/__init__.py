"""Forward error correction blocks of the hot path (LDPC 5G; Polar/CRC follow)."""
from . import ldpc
from . import polar, crc
