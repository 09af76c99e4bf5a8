"""Antenna models of 3GPP TR 38.901 Sec. 7.3 - mirrors of ``AntennaElement``, ``AntennaPanel``,
``PanelArray``, ``Antenna`` and ``AntennaArray`` (reference src/sionna/phy/channel/tr38901/
antenna.py:14-743).  Pure geometry / pattern descriptions on the host (NumPy); the channel models
tabulate the fields they need once per instance."""
import numpy as np

from ...block import Object

PI = np.pi
SPEED_OF_LIGHT = 299792458.0


class AntennaElement(Object):
    """``AntennaElement(pattern="omni"|"38.901", slant_angle=0.0)``; ``field(theta, phi)`` returns the
    vertical / horizontal field components (7.3-4/5) for zenith ``theta`` and azimuth ``phi`` [rad]."""

    def __init__(self, pattern, slant_angle=0.0, precision=None):
        super().__init__(precision=precision)
        assert pattern in ["omni", "38.901"], "The radiation_pattern must be one of [\"omni\", \"38.901\"]."
        self._pattern, self._slant_angle = pattern, float(slant_angle)

    pattern = property(lambda self: self._pattern)
    slant_angle = property(lambda self: self._slant_angle)

    def radiation_pattern(self, theta, phi):
        """Linear power gain; "38.901" = Table 7.3-1 (65 deg beamwidths, 30 dB floor, 8 dBi)."""
        theta, phi = np.asarray(theta, np.float64), np.asarray(phi, np.float64)
        if self._pattern == "omni":
            return np.ones_like(theta)
        bw = 65 / 180 * PI
        a_v = -np.minimum(12 * ((theta - PI / 2) / bw) ** 2, 30)
        a_h = -np.minimum(12 * (phi / bw) ** 2, 30)
        return 10 ** ((-np.minimum(-(a_v + a_h), 30) + 8) / 10)

    def field(self, theta, phi):
        a = np.sqrt(self.radiation_pattern(theta, phi))
        return a * np.cos(self._slant_angle), a * np.sin(self._slant_angle)

    def compute_gain(self):
        """(gain, directivity) in dB by numerical integration on a 1-degree grid (antenna.py:149-170)."""
        theta, phi = np.linspace(0.0, PI, 181), np.linspace(-PI, PI, 361)
        pg, tg = np.meshgrid(phi, theta)
        ft, fp = self.field(tg, pg)
        u = ft ** 2 + fp ** 2
        po = np.sum(u * np.sin(tg) * (theta[1] - theta[0]) * (phi[1] - phi[0]))
        return 10 * np.log10(np.max(u)), 10 * np.log10(np.max(u / (po / (4 * PI))))


class AntennaPanel(Object):
    """Element positions of one panel in multiples of the wavelength, LCS (panel in the y-z plane),
    column-major element order; dual polarisation duplicates the positions."""

    def __init__(self, num_rows, num_cols, polarization, vertical_spacing, horizontal_spacing, precision=None):
        super().__init__(precision=precision)
        assert polarization in ("single", "dual"), "polarization must be either 'single' or 'dual'"
        self._num_rows, self._num_cols, self._polarization = int(num_rows), int(num_cols), polarization
        self._vertical_spacing, self._horizontal_spacing = float(vertical_spacing), float(horizontal_spacing)
        j, i = np.meshgrid(np.arange(num_cols), np.arange(num_rows), indexing="ij")      # column-major flattening
        pos = np.stack([np.zeros(i.size), j.reshape(-1) * horizontal_spacing, -i.reshape(-1) * vertical_spacing], axis=1)
        pos += [0, -(num_cols - 1) * horizontal_spacing / 2, (num_rows - 1) * vertical_spacing / 2]
        self._ant_pos = np.concatenate([pos, pos]) if polarization == "dual" else pos

    ant_pos = property(lambda self: self._ant_pos)
    num_rows = property(lambda self: self._num_rows)
    num_cols = property(lambda self: self._num_cols)
    polarization = property(lambda self: self._polarization)
    vertical_spacing = property(lambda self: self._vertical_spacing)
    horizontal_spacing = property(lambda self: self._horizontal_spacing)


class PanelArray(Object):
    """``PanelArray(num_rows_per_panel, num_cols_per_panel, polarization, polarization_type,
    antenna_pattern, carrier_frequency, num_rows=1, num_cols=1, panel_vertical_spacing=None,
    panel_horizontal_spacing=None, element_vertical_spacing=None, element_horizontal_spacing=None)``
    (antenna.py:281-655); positions in metres."""

    def __init__(self, num_rows_per_panel, num_cols_per_panel, polarization, polarization_type, antenna_pattern,
                 carrier_frequency, num_rows=1, num_cols=1, panel_vertical_spacing=None, panel_horizontal_spacing=None,
                 element_vertical_spacing=None, element_horizontal_spacing=None, precision=None):
        super().__init__(precision=precision)
        assert polarization in ("single", "dual"), "polarization must be either 'single' or 'dual'"
        ev = 0.5 if element_vertical_spacing is None else element_vertical_spacing
        eh = 0.5 if element_horizontal_spacing is None else element_horizontal_spacing
        pv = (num_rows_per_panel - 1) * ev + 0.5 if panel_vertical_spacing is None else panel_vertical_spacing
        ph = (num_cols_per_panel - 1) * eh + 0.5 if panel_horizontal_spacing is None else panel_horizontal_spacing
        assert ph > (num_cols_per_panel - 1) * eh, "Pannel horizontal spacing must be larger than the panel width"
        assert pv > (num_rows_per_panel - 1) * ev, "Pannel vertical spacing must be larger than panel height"
        self._num_rows, self._num_cols = int(num_rows), int(num_cols)
        self._num_rows_per_panel, self._num_cols_per_panel = int(num_rows_per_panel), int(num_cols_per_panel)
        self._polarization, self._polarization_type = polarization, polarization_type
        self._panel_vertical_spacing, self._panel_horizontal_spacing = pv, ph
        self._element_vertical_spacing, self._element_horizontal_spacing = ev, eh
        p = 1 if polarization == "single" else 2
        self._num_panels = self._num_rows * self._num_cols
        self._num_panel_ant = self._num_rows_per_panel * self._num_cols_per_panel * p
        self._num_ant = self._num_panels * self._num_panel_ant
        self._lambda_0 = SPEED_OF_LIGHT / carrier_frequency
        if polarization == "single":
            assert polarization_type in ["V", "H"], "For single polarization, polarization_type must be 'V' or 'H'"
            slant = 0 if polarization_type == "V" else PI / 2
            self._ant_pol1, self._ant_pol2 = AntennaElement(antenna_pattern, slant, self.precision), None
        else:
            assert polarization_type in ["VH", "cross"], "For dual polarization, polarization_type must be 'VH' or 'cross'"
            slant = 0 if polarization_type == "VH" else -PI / 4
            self._ant_pol1 = AntennaElement(antenna_pattern, slant, self.precision)
            self._ant_pol2 = AntennaElement(antenna_pattern, slant + PI / 2, self.precision)
        panel = AntennaPanel(num_rows_per_panel, num_cols_per_panel, polarization, ev, eh, self.precision).ant_pos
        jj, ii = np.meshgrid(np.arange(num_cols), np.arange(num_rows), indexing="ij")    # panels column by column
        offs = np.stack([np.zeros(ii.size), jj.reshape(-1) * ph, -ii.reshape(-1) * pv], axis=1)
        pos = (panel[None, :, :] + offs[:, None, :]).reshape(-1, 3)
        pos += [0, -(num_cols - 1) * ph / 2, (num_rows - 1) * pv / 2]
        self._ant_pos = pos * self._lambda_0
        ind = np.arange(self._num_ant).reshape(self._num_panels * p, -1)                  # per panel: pol 1 block, pol 2 block
        self._ant_ind_pol1 = ind[::p].reshape(-1)
        self._ant_ind_pol2 = ind[1::2].reshape(-1) if p == 2 else np.array([], np.int64)

    num_rows = property(lambda self: self._num_rows)
    num_cols = property(lambda self: self._num_cols)
    num_rows_per_panel = property(lambda self: self._num_rows_per_panel)
    num_cols_per_panel = property(lambda self: self._num_cols_per_panel)
    polarization = property(lambda self: self._polarization)
    polarization_type = property(lambda self: self._polarization_type)
    panel_vertical_spacing = property(lambda self: self._panel_vertical_spacing)
    panel_horizontal_spacing = property(lambda self: self._panel_horizontal_spacing)
    element_vertical_spacing = property(lambda self: self._element_vertical_spacing)
    element_horizontal_spacing = property(lambda self: self._element_horizontal_spacing)
    num_panels = property(lambda self: self._num_panels)
    num_panels_ant = property(lambda self: self._num_panel_ant)
    num_ant = property(lambda self: self._num_ant)
    ant_pol1 = property(lambda self: self._ant_pol1)
    ant_pos = property(lambda self: self._ant_pos)
    ant_ind_pol1 = property(lambda self: self._ant_ind_pol1)
    ant_pos_pol1 = property(lambda self: self._ant_pos[self._ant_ind_pol1])

    @property
    def ant_pol2(self):
        assert self._polarization == "dual", "This property is not defined with single polarization"
        return self._ant_pol2

    @property
    def ant_ind_pol2(self):
        assert self._polarization == "dual", "This property is not defined with single polarization"
        return self._ant_ind_pol2

    @property
    def ant_pos_pol2(self):
        assert self._polarization == "dual", "This property is not defined with single polarization"
        return self._ant_pos[self._ant_ind_pol2]

    def pol_index(self):
        """[num_ant] 0 / 1: polarisation of every element."""
        pol = np.zeros(self._num_ant, np.int32)
        pol[self._ant_ind_pol2] = 1
        return pol


class Antenna(PanelArray):
    """``Antenna(polarization, polarization_type, antenna_pattern, carrier_frequency)``: single element
    (two co-located ones for dual polarisation)."""

    def __init__(self, polarization, polarization_type, antenna_pattern, carrier_frequency, precision=None):
        super().__init__(1, 1, polarization, polarization_type, antenna_pattern, carrier_frequency, precision=precision)


class AntennaArray(PanelArray):
    """``AntennaArray(num_rows, num_cols, polarization, polarization_type, antenna_pattern,
    carrier_frequency, vertical_spacing=None, horizontal_spacing=None)``: one panel."""

    def __init__(self, num_rows, num_cols, polarization, polarization_type, antenna_pattern, carrier_frequency,
                 vertical_spacing=None, horizontal_spacing=None, precision=None):
        super().__init__(num_rows, num_cols, polarization, polarization_type, antenna_pattern, carrier_frequency,
                         element_vertical_spacing=vertical_spacing, element_horizontal_spacing=horizontal_spacing,
                         precision=precision)
