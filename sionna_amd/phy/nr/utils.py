"""5G NR helpers of the transport-block chain - mirror of reference src/sionna/phy/nr/utils.py
(``generate_prng_seq`` :14-78, ``calculate_num_coded_bits`` :374-471, ``calculate_tb_size``
:473-805).  Scalar arguments (the TB blocks only need scalars); init-time host arithmetic in the
reference's float32."""
import numpy as np
import torch

from ... import _ffi

# 38.214 Table 5.1.3.2-1
_TAB51321 = np.array([-1, 24, 32, 40, 48, 56, 64, 72, 80, 88, 96, 104, 112, 120, 128, 136, 144, 152, 160, 168, 176, 184, 192,
                      208, 224, 240, 256, 272, 288, 304, 320, 336, 352, 368, 384, 408, 432, 456, 480, 504, 528, 552, 576,
                      608, 640, 672, 704, 736, 768, 808, 848, 888, 928, 984, 1032, 1064, 1128, 1160, 1192, 1224, 1256,
                      1288, 1320, 1352, 1416, 1480, 1544, 1608, 1672, 1736, 1800, 1864, 1928, 2024, 2088, 2152, 2216,
                      2280, 2408, 2472, 2536, 2600, 2664, 2728, 2792, 2856, 2976, 3104, 3240, 3368, 3496, 3624, 3752,
                      3824], np.float32)


def generate_prng_seq(length, c_init):
    """38.211 Sec. 5.2.1 length-31 Gold sequence as a NumPy array of 0/1 floats (utils.py:14-78);
    produced by ``samd_nr_prng_seq_f32``."""
    assert length % 1 == 0 and int(length) > 0, "length must be a positive integer."
    assert c_init % 1 == 0, "c_init must be integer."
    assert 0 <= int(c_init) < 2 ** 32, "c_init must be in [0, 2^32-1]."
    out = torch.empty(int(length), dtype=torch.float32, device=_ffi.device())
    _ffi.check(_ffi.lib().samd_nr_prng_seq_f32(int(c_init), int(length), _ffi.ptr(out), _ffi.stream()),
               "generate_prng_seq")
    return out.cpu().numpy()


def calculate_num_coded_bits(modulation_order, num_prbs, num_ofdm_symbols, num_dmrs_per_prb, num_layers=1, num_ov=0,
                             tb_scaling=1.0, precision=None):
    """Coded bits that fit into a slot (utils.py:374-471)."""
    assert 1 <= num_ofdm_symbols <= 14, "num_ofdm_symbols must be in [1, 14]."
    assert 1 <= num_prbs <= 275, "num_prbs must be in [1, 275]."
    assert tb_scaling in (0.25, 0.5, 1.0), "tb_scaling must be 0.25, 0.5, or 1.0."
    n_re_per_prb = min(156, 12 * int(num_ofdm_symbols) - int(num_dmrs_per_prb) - int(num_ov))
    return int(np.float32(tb_scaling) * np.float32(n_re_per_prb * int(num_prbs) * int(modulation_order) * int(num_layers)))


def calculate_tb_size(modulation_order, target_coderate, target_tb_size=None, num_coded_bits=None, num_prbs=None,
                      num_ofdm_symbols=None, num_dmrs_per_prb=None, num_layers=1, num_ov=0, tb_scaling=1.0,
                      return_cw_length=True, verbose=False, precision=None):
    """Transport block size of 38.214 Sec. 5.1.3.2 / 6.1.4.2 (utils.py:473-805).
    Returns ``(tb_size, cb_size, num_cb, tb_crc_length, cb_crc_length[, cw_length])``."""
    f = np.float32
    if num_coded_bits is None:
        assert num_prbs is not None and num_ofdm_symbols is not None and num_dmrs_per_prb is not None, \
            "If num_coded_bits is None then num_prbs, num_ofdm_symbols, num_dmrs_per_prb must be specified."
        num_coded_bits = calculate_num_coded_bits(modulation_order, num_prbs, num_ofdm_symbols, num_dmrs_per_prb,
                                                  num_layers, num_ov, tb_scaling)
    num_coded_bits, num_layers, modulation_order = int(num_coded_bits), int(num_layers), int(modulation_order)
    assert num_coded_bits % num_layers == 0, "num_coded_bits must be a multiple of num_layers."
    if target_tb_size is not None:
        t = f(target_tb_size)
        assert t < f(num_coded_bits), "target_tb_size must be less than num_coded_bits."
    else:
        t = f(target_coderate) * f(num_coded_bits)
    if t <= 3824:
        n = max(f(3.0), f(np.floor(np.log(t) / f(np.log(2.0))) - 6))
        n_info_q = max(f(24.0), f(f(2) ** n * np.floor(t / f(2) ** n)))
    else:
        n = np.floor(np.log(t - f(24)) / np.log(f(2.0))) - f(5.)
        n_info_q = max(f(3840.0), f(f(2) ** n * np.round((t - f(24)) / f(2) ** n)))
    if n_info_q <= 3824:
        num_cb = 1
        ge = _TAB51321 >= n_info_q
        ind = int(np.argmax(np.cumsum(1 - 2 * ge.astype(np.float32))))
        tb_size = int(_TAB51321[min(ind + 1, len(_TAB51321) - 1)])
    else:
        if target_coderate <= 1 / 4:
            num_cb = int(np.ceil((n_info_q + f(24)) / f(3816)))
        elif n_info_q > 8424:
            num_cb = int(np.ceil((n_info_q + f(24)) / f(8424)))
        else:
            num_cb = 1
        tb_size = int(f(8) * f(num_cb) * np.ceil((n_info_q + f(24)) / (f(8) * f(num_cb))) - f(24))
    tb_crc_length = 24 if tb_size > 3824 else 16
    cb_crc_length = 24 if num_cb > 1 else 0
    cb_size = int((tb_size + tb_crc_length) / num_cb) + cb_crc_length
    if verbose:
        print(f"Modulation order: {modulation_order}")
        if target_coderate is not None:
            print(f"Target coderate: {target_coderate:.3f}")
        print(f"Effective coderate: {tb_size / num_coded_bits:.3f}")
        print(f"Number of layers: {num_layers}")
        print("------------------")
        print(f"Info bits per TB: {tb_size}")
        print(f"TB CRC length: {tb_crc_length}")
        print(f"Total number of coded TB bits: {num_coded_bits}")
        print("------------------")
        print(f"Info bits per CB: {cb_size}")
        print(f"Number of CBs: {num_cb}")
        print(f"CB CRC length: {cb_crc_length}")
    if not return_cw_length:
        return tb_size, cb_size, num_cb, tb_crc_length, cb_crc_length
    q = num_layers * modulation_order
    num_last = int(num_coded_bits / q) % num_cb
    len_last = q * int(np.ceil(num_coded_bits / (q * num_cb)))
    len_first = q * int(np.floor(num_coded_bits / (q * num_cb)))
    cw_length = np.array([len_first] * (num_cb - num_last) + [len_last] * num_last, np.int64)
    return tb_size, cb_size, num_cb, tb_crc_length, cb_crc_length, cw_length
