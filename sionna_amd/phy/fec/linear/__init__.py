"""Generic binary linear block codes (mirror of ``sionna.phy.fec.linear``: encoders only)."""
from .encoding import LinearEncoder, AllZeroEncoder
