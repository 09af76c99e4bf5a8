"""``sionna_amd.phy`` - host-side mirror of the ``sionna.phy`` API for the MI355X hot path."""
from .config import config, dtypes
from .block import Block, Object, Tensor
from . import mapping, utils, channel, mimo, ofdm
from . import fec
from . import nr
from .fec import ldpc
