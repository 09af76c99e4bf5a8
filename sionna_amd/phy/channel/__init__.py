"""Channel models of the hot path (mirror of ``sionna.phy.channel``)."""
from .awgn import AWGN
