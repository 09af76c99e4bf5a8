"""``mimo.LinearDetector`` - equaliser followed by a demapper (reference
src/sionna/phy/mimo/detection.py:24-143).  Only the LMMSE equaliser with bit output is on the
hot path."""
import torch

from ..block import Block
from ..mapping import Demapper, Constellation
from .equalization import lmmse_equalizer


class LinearDetector(Block):
    def __init__(self, equalizer, output, demapping_method, constellation_type=None, num_bits_per_symbol=None,
                 constellation=None, hard_out=False, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        if output != "bit":
            raise NotImplementedError("LinearDetector: only output='bit' is on the MI355X hot path")
        if equalizer == "lmmse":
            self._equalizer = lmmse_equalizer
        elif callable(equalizer):
            self._equalizer = equalizer
        else:
            raise NotImplementedError(f"LinearDetector: equalizer '{equalizer}' is outside the hot path (lmmse only)")
        self._constellation = Constellation.check_or_create(
            constellation_type=constellation_type, num_bits_per_symbol=num_bits_per_symbol,
            constellation=constellation, precision=precision)
        self._demapper = Demapper(demapping_method, constellation=self._constellation, hard_out=hard_out,
                                  precision=precision)

    def call(self, y, h, s):
        x_hat, no_eff = self._equalizer(y, h, s)
        z = self._demapper(x_hat, no_eff)                     # [..., K*m]
        m = self._constellation.num_bits_per_symbol
        return z.reshape(tuple(x_hat.shape) + (m,))           # [..., K, m] (detection.py:139-143)
