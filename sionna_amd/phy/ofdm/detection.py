"""``ofdm.LinearDetector`` - mirror of reference src/sionna/phy/ofdm/detection.py:740-847
(-> ``OFDMDetector.call`` :289-317 -> ``mimo.LinearDetector``): fused LMMSE equaliser followed by
the LLR demapper with the per-symbol effective noise variance."""
from ..block import Block
from ..mapping import Demapper, Constellation
from .equalization import LMMSEEqualizer


class LinearDetector(Block):
    def __init__(self, equalizer, output, demapping_method, resource_grid, stream_management,
                 constellation_type=None, num_bits_per_symbol=None, constellation=None, hard_out=False,
                 precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        if equalizer != "lmmse":
            raise NotImplementedError(f"LinearDetector: equalizer '{equalizer}' is outside the hot path (lmmse only)")
        if output != "bit":
            raise NotImplementedError("LinearDetector: only output='bit' is on the MI355X hot path")
        self._eq = LMMSEEqualizer(resource_grid, stream_management, precision=precision)
        self._constellation = Constellation.check_or_create(
            constellation_type=constellation_type, num_bits_per_symbol=num_bits_per_symbol,
            constellation=constellation, precision=precision)
        self._demapper = Demapper(demapping_method, constellation=self._constellation, hard_out=hard_out,
                                  precision=precision)

    def call(self, y, h_hat, err_var, no):
        x_hat, no_eff = self._eq(y, h_hat, err_var, no)
        return self._demapper(x_hat, no_eff)          # [batch, num_tx, num_streams, num_data_symbols*m]
