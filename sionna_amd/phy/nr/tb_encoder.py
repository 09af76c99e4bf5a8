"""``TBEncoder`` - 5G NR transport-block encoding (38.212 Sec. 6.2): TB CRC, code-block
segmentation + CB CRC, LDPC encoding and rate matching per code block, bit interleaving, code-block
concatenation, scrambling.  Mirror of reference src/sionna/phy/nr/tb_encoder.py:15-435; every stage
runs on the HIP blocks of this package (``samd_crc_f32``, ``samd_ldpc5g_encode_f32``,
``samd_gather3``, ``samd_scramble_f32``)."""
import numpy as np
import torch

from ... import _ffi
from ..block import Block, wrap
from ..fec.crc import CRCEncoder
from ..fec.ldpc import LDPC5GEncoder
from ..fec.scrambling import TB5GScrambler
from .utils import calculate_tb_size


class TBEncoder(Block):
    """``TBEncoder(target_tb_size, num_coded_bits, target_coderate, num_bits_per_symbol, num_layers=1,
    n_rnti=1, n_id=1, channel_type="PUSCH", codeword_index=0, use_scrambler=True, verbose=False)``
    ``(bits [..., k] or [..., num_tx, k]) -> [..., n]``."""

    def __init__(self, target_tb_size, num_coded_bits, target_coderate, num_bits_per_symbol, num_layers=1, n_rnti=1,
                 n_id=1, channel_type="PUSCH", codeword_index=0, use_scrambler=True, verbose=False, precision=None,
                 **kwargs):
        kwargs.pop("output_dtype", None)
        super().__init__(precision=precision, **kwargs)
        assert isinstance(use_scrambler, bool), "use_scrambler must be bool."
        assert isinstance(verbose, bool), "verbose must be bool."
        assert channel_type in ("PDSCH", "PUSCH"), "Unsupported channel_type."
        assert target_tb_size % 1 == 0, "target_tb_size must be int."
        assert num_coded_bits % 1 == 0, "num_coded_bits must be int."
        assert 0. < target_coderate <= 948 / 1024, "target_coderate must be in range(0,0.925)."
        assert num_bits_per_symbol % 1 == 0, "num_bits_per_symbol must be int."
        assert num_layers % 1 == 0, "num_layers must be int."
        if channel_type == "PDSCH":
            assert codeword_index in (0, 1), "codeword_index must be 0 or 1."
        else:
            assert codeword_index == 0, 'codeword_index must be 0 for "PUSCH".'
        self._use_scrambler, self._verbose, self._channel_type = use_scrambler, verbose, channel_type
        self._target_tb_size, self._num_coded_bits = int(target_tb_size), int(num_coded_bits)
        self._target_coderate = float(target_coderate)
        self._num_bits_per_symbol, self._num_layers = int(num_bits_per_symbol), int(num_layers)
        self._codeword_index = int(codeword_index)
        if isinstance(n_rnti, (list, tuple)):
            assert isinstance(n_id, (list, tuple)), "n_id must be also a list."
            assert len(n_rnti) == len(n_id), "n_id and n_rnti must be of same length."
            self._n_rnti, self._n_id = list(n_rnti), list(n_id)
        else:
            self._n_rnti, self._n_id = [n_rnti], [n_id]
        for lst, name in ((self._n_rnti, "n_rnti"), (self._n_id, "n_id")):
            for idx, v in enumerate(lst):
                assert v % 1 == 0, f"{name} must be int."
                lst[idx] = int(v)
        self._num_tx = len(self._n_id)

        (self._tb_size, self._cb_size, self._num_cbs, self._tb_crc_length, self._cb_crc_length,
         self._cw_lengths) = calculate_tb_size(target_tb_size=self._target_tb_size, num_coded_bits=self._num_coded_bits,
                                               target_coderate=self._target_coderate,
                                               modulation_order=self._num_bits_per_symbol,
                                               num_layers=self._num_layers, verbose=verbose)
        assert self._tb_size <= self._tb_crc_length + np.sum(self._cw_lengths), "Invalid TB parameters."
        self._k_padding = self._tb_size - self._target_tb_size
        if self._tb_size != self._target_tb_size:
            print(f"Note: actual tb_size={self._tb_size} is slightly different than requested "
                  f"target_tb_size={self._target_tb_size} due to quantization. Internal zero padding will be applied.")
        self._coderate = self._tb_size / self._num_coded_bits
        # the stages work on bits carried as float32 (exact); with precision="double" the block's output is cast (block.py::_bits)
        self._tb_crc_encoder = CRCEncoder("CRC16" if self._tb_crc_length == 16 else "CRC24A", precision="single")
        self._cb_crc_encoder = CRCEncoder("CRC24B", precision="single") if self._cb_crc_length == 24 else None
        # lists -> one stream per entry on axis -2 (tb_encoder.py:232-240)
        self._scrambler = TB5GScrambler(n_rnti=self._n_rnti, n_id=self._n_id, binary=True, channel_type=channel_type,
                                        codeword_index=codeword_index, precision="single") if use_scrambler else None
        lmin, lmax = int(np.min(self._cw_lengths)), int(np.max(self._cw_lengths))
        # num_bits_per_symbol=1 deactivates the encoder's own interleaver (tb_encoder.py:243-246)
        self._encoder = LDPC5GEncoder(self._cb_size, lmax, num_bits_per_symbol=1, precision="single")
        perm_short, _ = self._encoder.generate_out_int(lmin, self._num_bits_per_symbol)
        perm_long, _ = self._encoder.generate_out_int(lmax, self._num_bits_per_symbol)
        perm, punc, pos = [], [], 0
        for l in self._cw_lengths:                                     # tb_encoder.py:252-283
            if l == lmin:
                perm.append(np.asarray(perm_short) + pos)
                punc.append(np.arange(pos + lmin, pos + lmax))
                pos += lmax
            elif l == lmax:
                perm.append(np.asarray(perm_long) + pos)
                pos += l
            else:
                raise ValueError("Invalid cw_lengths.")
        self._output_perm = np.concatenate(perm + punc).astype(np.int32)
        self._output_perm_inv = np.argsort(self._output_perm).astype(np.int32)
        self._perm_dev = None

    tb_size = property(lambda self: self._tb_size)
    k = property(lambda self: self._target_tb_size)
    k_padding = property(lambda self: self._k_padding)
    n = property(lambda self: int(np.sum(self._cw_lengths)))
    num_cbs = property(lambda self: self._num_cbs)
    coderate = property(lambda self: self._coderate)
    ldpc_encoder = property(lambda self: self._encoder)
    scrambler = property(lambda self: self._scrambler)
    tb_crc_encoder = property(lambda self: self._tb_crc_encoder)
    cb_crc_encoder = property(lambda self: self._cb_crc_encoder)
    num_tx = property(lambda self: self._num_tx)
    cw_lengths = property(lambda self: self._cw_lengths)
    output_perm_inv = property(lambda self: self._output_perm_inv)

    def build(self, input_shapes):
        assert input_shapes[-1] == self.k, f"Invalid input shape. Expected TB length is {self.k}."

    def call(self, inputs):
        u = _ffi.to_device(inputs, torch.float32)
        assert u.shape[-1] == self.k, f"Invalid input shape. Expected TB length is {self.k}."
        shape = tuple(u.shape)
        u = u.reshape(-1, self._num_tx, self.k)
        if self._k_padding > 0:                                        # zero padding to the quantised TB size
            u = torch.cat([u, torch.zeros(u.shape[:-1] + (self._k_padding,), dtype=u.dtype, device=u.device)], dim=-1)
        u_crc = self._tb_crc_encoder(u)
        u_cb = u_crc.reshape(-1, self._num_tx, self._num_cbs, self._cb_size - self._cb_crc_length)
        if self._cb_crc_encoder is not None:
            u_cb = self._cb_crc_encoder(u_cb)
        c_cb = self._encoder(u_cb)
        lmax = int(np.max(self._cw_lengths))
        c = c_cb.reshape(-1, self._num_cbs * lmax).contiguous()
        if self._perm_dev is None:
            self._perm_dev = (_ffi.to_device(self._output_perm[:self.n].copy(), torch.int32),
                              _ffi.to_device(np.zeros(1, np.int32), torch.int32))
        perm, zero = self._perm_dev
        out = torch.empty((c.shape[0], self.n), dtype=torch.float32, device=c.device)
        if c.shape[0]:                                                 # interleave + puncture in one gather
            _ffi.check(_ffi.lib().samd_gather3(_ffi.ptr(c), _ffi.ptr(zero), _ffi.ptr(perm), c.shape[0], 1,
                                               self._num_cbs * lmax, 1, self.n, 1, _ffi.ptr(out), _ffi.stream()),
                       "TBEncoder interleaver")
        c = out.reshape(-1, self._num_tx, self.n)
        if self._scrambler is not None:
            c = self._scrambler(c)
        return wrap(self._bits(c.reshape(shape[:-1] + (self.n,))))
