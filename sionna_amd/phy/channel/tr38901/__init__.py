"""3GPP TR 38.901 link-level channel models (TDL, CDL) and antenna models."""
from .tdl import TDL
from .antenna import AntennaElement, AntennaPanel, PanelArray, Antenna, AntennaArray
from .cdl import CDL
