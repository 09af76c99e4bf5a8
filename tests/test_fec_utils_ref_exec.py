"""Host-side code utilities (sionna_amd/phy/fec/utils.py) against the reference's OWN ``fec/utils.py`` executed under the
NumPy stand-in for TensorFlow (tests/golden/fec_utils_ref_golden.npz, tools/gen_fec_utils_ref_golden.py): J-function pair,
``llr2mi``, bit / integer conversions, ``int_mod_2``, the example parity-check matrices, ``pcm2gm`` / ``gm2pcm`` /
``make_systematic`` (same matrices and column swaps bit for bit), and the parameters ``GaussianPriorSource`` derives from
``no`` / ``mi``."""
import os

import numpy as np
import pytest

from sionna_amd.phy.fec import utils as u

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "fec_utils_ref_golden.npz"))


def test_j_functions_and_llr2mi():
    # the reference evaluates these in float32 (TF), this build in float64 on the host: 1e-5 of scale
    assert np.allclose(u.j_fun(G["mu"]), G["j_fun"], rtol=2e-5, atol=2e-6)
    assert np.allclose(u.j_fun_inv(G["mi"]), G["j_fun_inv"], rtol=1e-4, atol=1e-5)
    assert np.isclose(u.llr2mi(G["llr"]), G["llr2mi"], rtol=1e-5)
    assert np.isclose(u.llr2mi(G["llr"], G["s"]), G["llr2mi_s"], rtol=1e-5)
    assert np.allclose(u.llr2mi(G["llr"], G["s"], reduce_dims=False), G["llr2mi_rows"], rtol=1e-5)


def test_bit_and_integer_conversions():
    assert [u.bin2int(list(b)) for b in G["bits"]] == list(G["bin2int"])
    assert np.array_equal(np.asarray(u.bin2int_tf(G["bits"])), G["bin2int_tf"])
    assert np.array_equal(np.array([u.int2bin(int(v), 12) for v in G["ints"]]), G["int2bin"])
    assert np.array_equal(np.asarray(u.int2bin_tf(G["ints"].astype(np.int32), 12)), G["int2bin_tf"])
    assert np.array_equal(np.asarray(u.int_mod_2(G["mod2_in"]), np.float32), G["mod2"])
    assert np.array_equal(np.asarray(u.int_mod_2(np.arange(-4, 5))), np.arange(-4, 5) & 1)


@pytest.mark.parametrize("i", range(4))
def test_example_codes_and_systematic_forms(i):
    pcm, k, n, r = u.load_parity_check_examples(i)
    assert tuple(pcm.shape) == tuple(G[f"pcm{i}_shape"]) and [int(pcm.sum()), k, n] == list(G[f"pcm{i}_sum"])
    gm = u.pcm2gm(pcm)
    assert np.array_equal(np.packbits(np.asarray(gm).astype(np.uint8), axis=1), G[f"gm{i}"])
    assert bool(u.verify_gm_pcm(gm, pcm)) == bool(G[f"gm{i}_ok"])
    msys, swaps = u.make_systematic(pcm, is_pcm=True)
    assert np.array_equal(np.packbits(np.asarray(msys).astype(np.uint8), axis=1), G[f"sys{i}"])
    assert np.array_equal(np.array(swaps, np.int64).reshape(-1, 2), G[f"swaps{i}"])
    assert np.array_equal(np.packbits(np.asarray(u.gm2pcm(gm)).astype(np.uint8), axis=1), G[f"pcm_back{i}"])


def test_gaussian_prior_source_parameters():
    """N(-mu, 2 mu) with mu = 2 / no, or mu = J^-1(mi) (fec/utils.py:16-113): the moments of 400 k reference draws."""
    for is_mi, v, mean, std in G["gps"]:
        mu = float(u.j_fun_inv(max(v, 1e-7))) if is_mi else 2.0 / max(v, 1e-7)
        assert abs(-mu - mean) < 4 * std / np.sqrt(4e5) + 1e-3 * mu and abs(np.sqrt(2 * mu) - std) < 0.01 * std
