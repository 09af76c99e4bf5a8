"""3GPP TR 38.901 link-level channel models of the hot path (TDL)."""
from .tdl import TDL
