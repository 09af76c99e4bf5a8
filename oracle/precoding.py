"""CPU oracle of the regularised zero-forcing precoder (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates ``rzf_precoding_matrix`` / ``rzf_precoder`` (reference src/sionna/phy/mimo/precoding.py:12-244) and
``RZFPrecoder.call`` with its effective channel (ofdm/precoding.py:81-177) in complex128.  No HIP kernel exists for this
row yet (the downlink tables of MIMO_OFDM_Transmissions_over_CDL.ipynb need it); the oracle and its
reference-executed fixture (tests/golden/precoding_ref_golden.npz, tools/gen_precoding_ref_golden.py) are the target that
kernel will be held to."""
import numpy as np

C = np.complex128


def rzf_precoding_matrix(h, alpha=0.):
    """precoding.py:70-88: G = V D, V = H^H (H H^H + alpha I)^-1, D = diag(1 / ||v_k||); h [..., K, M] -> [..., M, K]."""
    h = np.asarray(h, C)
    K = h.shape[-2]
    a = h @ np.conj(np.swapaxes(h, -1, -2)) + np.asarray(alpha, np.float64)[..., None, None] * np.eye(K)
    g = np.conj(np.swapaxes(np.linalg.solve(a, h), -1, -2))
    norm = np.sqrt(np.sum(np.abs(g) ** 2, axis=-2, keepdims=True))
    return np.where(norm == 0, 0, g / np.where(norm == 0, 1, norm))


def rzf_precoder(x, h, alpha=0.):
    """precoding.py:157-244: (G x, G) for x [..., K], h [..., K, M]."""
    g = rzf_precoding_matrix(h, alpha)
    return (g @ np.asarray(x, C)[..., None])[..., 0], g


def ofdm_rzf_precoder(x, h, precoding_ind, nulled_remove, alpha=0.):
    """RZFPrecoder.call (ofdm/precoding.py:118-177): x [B, tx, streams, T, fft], h [B, rx, rxant, tx, txant, T, fft],
    precoding_ind [tx, streams-worth of receivers] (StreamManagement.precoding_ind), nulled_remove: indices of the effective
    subcarriers (RemoveNulledSubcarriers) -> (x_precoded [B, tx, txant, T, fft], h_eff [B, rx, rxant, tx, streams, T, F_eff])."""
    x, h = np.asarray(x, C), np.asarray(h, C)
    B, ntx, ns, T, F = x.shape
    xp = np.transpose(x, [0, 1, 3, 4, 2])                                         # [B, tx, T, F, streams]
    h_pc = np.transpose(h, [3, 1, 2, 4, 5, 6, 0])                                 # [tx, rx, rxant, txant, T, F, B]
    hd = np.stack([h_pc[t][np.asarray(precoding_ind)[t]] for t in range(ntx)])   # [tx, nrx_t, rxant, txant, T, F, B]
    hd = hd.reshape((ntx, -1) + hd.shape[3:])                                     # [tx, K, txant, T, F, B]
    hd = np.transpose(hd, [5, 0, 3, 4, 1, 2])                                     # [B, tx, T, F, K, txant]
    xo, g = rzf_precoder(xp, hd, alpha)                                           # [B, tx, T, F, txant], [B, tx, T, F, txant, K]
    xo = np.transpose(xo, [0, 1, 4, 2, 3])
    hh = np.transpose(h, [0, 1, 3, 5, 6, 2, 4])                                   # [B, rx, tx, T, F, rxant, txant]
    h_eff = hh @ g[:, None]                                                       # [B, rx, tx, T, F, rxant, K]
    h_eff = np.transpose(h_eff, [0, 1, 5, 2, 6, 3, 4])[..., np.asarray(nulled_remove)]
    return xo, h_eff
