"""``TBDecoder`` - 5G NR transport-block decoding: descrambling, de-interleaving / rate recovery,
LDPC decoding per code block, CB / TB CRC removal and TB CRC check.  Mirror of reference
src/sionna/phy/nr/tb_decoder.py:15-213."""
import numpy as np
import torch

from ... import _ffi
from ..block import Block, wrap
from ..fec.crc import CRCDecoder
from ..fec.ldpc import LDPC5GDecoder
from ..fec.scrambling import Descrambler
from .tb_encoder import TBEncoder


class TBDecoder(Block):
    """``TBDecoder(encoder, num_bp_iter=20, cn_update="boxplus-phi", vn_update="sum")(llr [..., n])``
    ``-> (b_hat [..., k], tb_crc_status [...])``."""

    def __init__(self, encoder, num_bp_iter=20, cn_update="boxplus-phi", vn_update="sum", precision=None, **kwargs):
        kwargs.pop("output_dtype", None)
        super().__init__(precision=precision, **kwargs)
        assert isinstance(encoder, TBEncoder), "encoder must be TBEncoder."
        self._tb_encoder = encoder
        self._num_cbs = encoder.num_cbs
        self._decoder = LDPC5GDecoder(encoder=encoder.ldpc_encoder, num_iter=num_bp_iter, cn_update=cn_update,
                                      vn_update=vn_update, hard_out=True, return_infobits=True, precision=precision)
        self._descrambler = Descrambler(encoder.scrambler, binary=False, precision=precision) \
            if encoder.scrambler is not None else None
        self._tb_crc_decoder = CRCDecoder(encoder.tb_crc_encoder, precision=precision)
        self._cb_crc_decoder = CRCDecoder(encoder.cb_crc_encoder, precision=precision) \
            if encoder.cb_crc_encoder is not None else None
        self._perm_dev = None

    tb_size = property(lambda self: self._tb_encoder.tb_size)
    k = property(lambda self: self._tb_encoder.tb_size)
    n = property(lambda self: self._tb_encoder.n)

    def build(self, input_shapes):
        assert input_shapes[-1] == self.n, f"Invalid input shape. Expected input length is {self.n}."

    def call(self, inputs):
        enc = self._tb_encoder
        llr = _ffi.to_device(inputs, self.rdtype)
        assert llr.shape[-1] == self.n, f"Invalid input shape. Expected input length is {self.n}."
        shape = tuple(llr.shape)
        llr = llr.reshape(-1, enc.num_tx, enc.n)
        if self._descrambler is not None:
            llr = self._descrambler(llr)
        llr = llr.reshape(-1, enc.n)
        n_full = enc.ldpc_encoder.n * enc.num_cbs
        # zero filler LLRs for the punctured tail + inverse interleaver (tb_decoder.py:161-173): one gather
        # from [llr | 0] with the inverse permutation
        if self._perm_dev is None:
            self._perm_dev = (_ffi.to_device(enc.output_perm_inv, torch.int32),
                              _ffi.to_device(np.zeros(1, np.int32), torch.int32))
        perm, zero = self._perm_dev
        if n_full > enc.n:
            llr = torch.cat([llr, torch.zeros((llr.shape[0], n_full - enc.n), dtype=llr.dtype, device=llr.device)], dim=-1)
        llr = llr.contiguous()
        llr_int = torch.empty_like(llr)
        if llr.shape[0]:
            _ffi.check(_ffi.lib().samd_gather3(_ffi.ptr(llr), _ffi.ptr(zero), _ffi.ptr(perm), llr.shape[0], 1, n_full, 1,
                                               n_full, llr.element_size() // 4, _ffi.ptr(llr_int), _ffi.stream()), "TBDecoder deinterleaver")
        llr_cb = llr_int.reshape(-1, enc.num_tx, self._num_cbs, enc.ldpc_encoder.n)
        u_hat_cb = self._decoder(llr_cb)
        if self._cb_crc_decoder is not None:
            u_hat_cb, _ = self._cb_crc_decoder(u_hat_cb)
        u_hat_tb = u_hat_cb.reshape(-1, enc.num_tx, self.tb_size + enc.tb_crc_encoder.crc_length)
        u_hat, status = self._tb_crc_decoder(u_hat_tb)
        u_hat = u_hat.reshape(shape[:-1] + (self.tb_size,))
        status = status.reshape(shape[:-1])
        if enc.k_padding > 0:
            u_hat = u_hat[..., :-enc.k_padding].contiguous()
        return wrap(u_hat), status
