"""CPU validation of the QC kernel algorithms (tests/kernel_models.py) against the oracle."""
import numpy as np
import pytest

from oracle.ldpc5g import LDPC5GCode
from oracle import ldpc_bp as obp
from sionna_amd.phy.fec.ldpc import LDPC5GEncoder, LDPC5GDecoder
from tests.kernel_models import encode_qc_model, decode_onchip_model

CODES = [(64, 128, None, None), (200, 600, None, 2), (1024, 2048, "bg1", None), (500, 1000, None, 4),
         (2816, 8448, "bg1", 6), (1347, 1554, None, None), (8448, 25344, None, None), (3840, 4800, "bg2", None)]


@pytest.mark.parametrize("k,n,bg,m", CODES)
def test_encoder_model_matches_oracle(k, n, bg, m):
    enc = LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg=bg)
    code = LDPC5GCode(k, n, num_bits_per_symbol=m, bg=bg)
    u = np.random.default_rng(k + n).integers(0, 2, (6, k)).astype(np.float32)
    assert np.array_equal(encode_qc_model(enc, u), code.encode(u))


@pytest.mark.parametrize("k,n,bg,m,offset", [(64, 128, None, None, 0.0), (200, 600, None, 2, 0.5),
                                             (100, 200, "bg1", None, 0.0), (400, 480, None, 4, 0.5)])
def test_onchip_decoder_model_bit_exact(k, n, bg, m, offset):
    enc = LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg=bg)
    cn = "offset-minsum" if offset else "minsum"
    dec = LDPC5GDecoder(enc, cn_update=cn, hard_out=False, return_infobits=False)
    code = LDPC5GCode(k, n, num_bits_per_symbol=m, bg=bg)
    odec = obp.LDPC5GDecoder(code, cn_update=cn, hard_out=False, return_infobits=False)
    rng = np.random.default_rng(11)
    u = rng.integers(0, 2, (3, k)).astype(np.float32)
    c = code.encode(u)
    llr = ((2 * c - 1) * 2 + rng.normal(scale=2.0, size=c.shape)).astype(np.float32)
    llr[0, :5] = 0.0                                    # exact zeros / ties
    llr[1] = np.round(llr[1])                           # many duplicate magnitudes
    for it in (0, 1, 3):
        x_model = decode_onchip_model(dec, llr, it, offset)
        ref = obp.LDPCBPDecoder(odec.pcm, cn_update=cn, hard_out=False, num_iter=it)
        x_ref = -ref.decode(odec.rate_recover(llr))     # internal LLR sign
        assert np.array_equal(x_model, x_ref), f"iter {it}"
