"""Oracle (NumPy, CPU): SNR helper, hard decisions, error counting, AWGN and the RNG spec.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Restates (paths relative to
/root/reference/src/sionna/phy):

* ebnodb2no            utils/misc.py:171-251
* hard_decisions       utils/misc.py:254-271      (strict ``llr > 0``)
* count_errors & co    utils/metrics.py:9-144
* complex_normal/AWGN  utils/misc.py:19-54, channel/awgn.py:63-78

RNG: the reference draws from ``tf.random.Generator`` (Philox) which cannot be reproduced
without TensorFlow ("parity unpinned" for random streams).  The build
therefore defines its own counter-based stream - Philox4x32-10 (Salmon et al., SC'11;
the public Random123 constants) keyed by (seed, call counter) and indexed by element -
and this file is its executable specification; the HIP kernels must reproduce the
integer stream bit-exactly and the derived normals to float32 rounding.
"""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  All inputs uint32 arrays (broadcastable)."""
    c0, c1, c2, c3 = [np.asarray(c, np.uint32) for c in (c0, c1, c2, c3)]
    k0 = np.asarray(k0, np.uint32)
    k1 = np.asarray(k1, np.uint32)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = (k0 + _W0).astype(np.uint32)
            k1 = (k1 + _W1).astype(np.uint32)
    return c0, c1, c2, c3


def philox_block(seed, call, num_blocks):
    """4 x uint32 for block indices 0..num_blocks-1 of stream (seed, call).

    counter = (block_lo, block_hi, call_lo, call_hi), key = (seed_lo, seed_hi).
    """
    idx = np.arange(num_blocks, dtype=np.uint64)
    lo = (idx & _MASK).astype(np.uint32)
    hi = (idx >> np.uint64(32)).astype(np.uint32)
    seed, call = int(seed) & (2 ** 64 - 1), int(call) & (2 ** 64 - 1)
    return philox4x32_10(lo, hi, np.uint32(call & 0xFFFFFFFF), np.uint32(call >> 32),
                         np.uint32(seed & 0xFFFFFFFF), np.uint32(seed >> 32))


def random_bits(seed, call, n):
    """BinarySource stream: element i = bit 0 of word (i%4) of block (i//4)."""
    nb = (n + 3) // 4
    w = np.stack(philox_block(seed, call, nb), axis=1).reshape(-1)[:n]
    return (w & np.uint32(1)).astype(np.float32)


def _u01(x):
    """uint32 -> float32 in (0,1): (x>>8)*2^-24 + 2^-25 (exact in float32)."""
    return (x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24) + np.float32(2.0 ** -25)


def complex_normal(seed, call, n, var=1.0):
    """n complex64 samples CN(0,var): block b gives samples 2b (words 0,1) and 2b+1 (2,3).

    Box-Muller per sample: r = sqrt(-2 ln u_a), t = 2 pi u_b, re = r cos t, im = r sin t,
    scaled by sqrt(var/2) (utils/misc.py:45-52).
    """
    nb = (n + 1) // 2
    w0, w1, w2, w3 = philox_block(seed, call, nb)
    ua = np.stack([_u01(w0), _u01(w2)], axis=1).reshape(-1)[:n]
    ub = np.stack([_u01(w1), _u01(w3)], axis=1).reshape(-1)[:n]
    r = np.sqrt(np.float32(-2.0) * np.log(ua))
    t = np.float32(6.283185307179586) * ub
    s = np.sqrt(np.float32(var) / np.float32(2))
    return ((r * np.cos(t)) * s + 1j * ((r * np.sin(t)) * s)).astype(np.complex64)


def awgn(x, no, seed, call):
    """channel/awgn.py:63-78 with the build's RNG stream; x flattened in C order."""
    x = np.asarray(x, np.complex64)
    no = np.broadcast_to(np.asarray(no, np.float32), x.shape).reshape(-1)
    w = complex_normal(seed, call, x.size, 1.0)
    # noise = w * sqrt(no): complex_normal(var=1) scaled per element (awgn.py:72-76)
    return (x.reshape(-1) + w * np.sqrt(no).astype(np.float32)).reshape(x.shape).astype(np.complex64)


def ebnodb2no(ebno_db, num_bits_per_symbol, coderate, resource_grid=None):
    """utils/misc.py:171-251 (float32 arithmetic like the reference's default precision)."""
    f = np.float32
    ebno = np.power(f(10), f(ebno_db) / f(10), dtype=np.float32)
    energy_per_symbol = 1.
    if resource_grid is not None:
        energy_per_symbol /= resource_grid.num_streams_per_tx
        cp_overhead = resource_grid.cyclic_prefix_length / resource_grid.fft_size
        num_syms = (resource_grid.num_ofdm_symbols * (1 + cp_overhead)
                    * resource_grid.num_effective_subcarriers)
        energy_per_symbol *= num_syms / resource_grid.num_data_symbols
    return f(1) / (ebno * f(coderate) * f(num_bits_per_symbol) / f(energy_per_symbol))


def hard_decisions(llr):
    """utils/misc.py:254-271"""
    return (llr > 0).astype(llr.dtype)


def count_errors(b, b_hat):
    """utils/metrics.py:94-117"""
    return int(np.sum(b != b_hat))


def count_block_errors(b, b_hat):
    """utils/metrics.py:119-144"""
    return int(np.sum(np.any(b != b_hat, axis=-1)))
