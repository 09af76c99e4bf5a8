"""Drop-in proof (north star: "drops into existing BER notebooks"): model code written the way the reference's
tutorials write it - ``import sionna.phy``, ``import tensorflow as tf``, a ``Block`` whose ``call`` is decorated with
``@tf.function()``, ``tf.constant`` scalars, ``PlotBER.simulate`` - runs unchanged on sionna_amd after
``install_as_sionna(tf_shim=True)``.  The model below follows the structure of the ``System_Model`` cell of
tutorials/phy/5G_Channel_Coding_Polar_vs_LDPC_Codes.ipynb (source -> encoder -> QAM mapper -> AWGN -> demapper ->
decoder, Eb/N0 or Es/N0), exercised with the notebook's three 5G schemes: LDPC BP-20, Polar SC, Polar SCL-8."""
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def notebook_env():
    import matplotlib
    matplotlib.use("Agg")
    import sionna_amd
    before = {k: sys.modules.get(k) for k in list(sys.modules) if k == "tensorflow" or k == "sionna" or k.startswith("sionna.")}
    had_tf = "tensorflow" in sys.modules
    sionna_amd.install_as_sionna(tf_shim=True)
    yield
    for k in [k for k in sys.modules if k == "sionna" or k.startswith("sionna.")]:
        if k not in before:
            del sys.modules[k]
    if not had_tf:
        sys.modules.pop("tensorflow", None)


def test_polar_vs_ldpc_notebook_model(notebook_env):
    # ---- the notebook's import cell
    import sionna.phy
    import tensorflow as tf
    gpus = tf.config.list_physical_devices('GPU')
    if gpus:
        tf.config.experimental.set_memory_growth(gpus[0], True)
    tf.get_logger().setLevel('ERROR')
    sionna.phy.config.seed = 42
    from sionna.phy import Block
    from sionna.phy.mapping import Constellation, Mapper, Demapper, BinarySource
    from sionna.phy.fec.polar import PolarEncoder, Polar5GEncoder, PolarSCLDecoder, Polar5GDecoder
    from sionna.phy.fec.ldpc import LDPC5GEncoder, LDPC5GDecoder
    from sionna.phy.fec.polar.utils import generate_5g_ranking, generate_rm_code
    from sionna.phy.utils import count_block_errors, ebnodb2no, PlotBER
    from sionna.phy.channel import AWGN

    # ---- the notebook's model cell
    class SystemModel(Block):
        def __init__(self, k, n, num_bits_per_symbol, encoder, decoder, demapping_method="app", sim_esno=False,
                     cw_estimates=False):
            super().__init__()
            self.k, self.n, self.sim_esno, self.cw_estimates = k, n, sim_esno, cw_estimates
            self.num_bits_per_symbol = num_bits_per_symbol
            self.source = BinarySource()
            self.constellation = Constellation("qam", num_bits_per_symbol=self.num_bits_per_symbol)
            self.mapper = Mapper(constellation=self.constellation)
            self.demapper = Demapper(demapping_method, constellation=self.constellation)
            self.channel = AWGN()
            self.encoder, self.decoder = encoder, decoder

        @tf.function()
        def call(self, batch_size, ebno_db):
            if self.sim_esno:
                no = ebnodb2no(ebno_db, num_bits_per_symbol=1, coderate=1)
            else:
                no = ebnodb2no(ebno_db, num_bits_per_symbol=self.num_bits_per_symbol, coderate=self.k / self.n)
            u = self.source([batch_size, self.k])
            c = self.encoder(u)
            y = self.channel(self.mapper(c), no)
            u_hat = self.decoder(self.demapper(y, no))
            return (c, u_hat) if self.cw_estimates else (u, u_hat)

    # ---- the notebook's code list (5G schemes) and simulation loop
    k, n = 64, 128
    codes_under_test = []
    enc = LDPC5GEncoder(k=k, n=n)
    codes_under_test.append([enc, LDPC5GDecoder(enc, num_iter=20), "5G LDPC BP-20"])
    enc = Polar5GEncoder(k=k, n=n)
    codes_under_test.append([enc, Polar5GDecoder(enc, dec_type="SC"), "5G Polar+CRC SC"])
    enc = Polar5GEncoder(k=k, n=n)
    codes_under_test.append([enc, Polar5GDecoder(enc, dec_type="SCL", list_size=8), "5G Polar+CRC SCL-8"])
    f, _, _, _, _ = generate_rm_code(3, 7)                             # equals k=64 and n=128
    codes_under_test.append([PolarEncoder(f, n), PolarSCLDecoder(f, n, list_size=8), "Reed Muller (RM) SCL-8"])
    assert len(generate_5g_ranking(k, n)[0]) == n - k

    ber_plot128 = PlotBER(f"Performance of Short Length Codes (k={k}, n={n})")
    ebno_db = np.arange(0, 5, 1.0)
    results = {}
    for code in codes_under_test:
        model = SystemModel(k=k, n=n, num_bits_per_symbol=2, encoder=code[0], decoder=code[1])
        ber, bler = ber_plot128.simulate(tf.function(model, jit_compile=True) if code[2].startswith("Reed") else model,
                                         ebno_dbs=ebno_db, legend=code[2], max_mc_iter=4,
                                         num_target_block_errors=200, batch_size=2000, soft_estimates=False,
                                         early_stop=True, show_fig=False, add_bler=True,
                                         forward_keyboard_interrupt=True, verbose=False)
        results[code[2]] = (ber.numpy(), bler.numpy())
    fig_ax = ber_plot128(ylim=(1e-5, 1), show_bler=False)
    assert fig_ax is not None and len(ber_plot128.legend) == 8 and ber_plot128.is_bler == [False, True] * 4
    ber_plot128(ylim=(1e-5, 1), show_ber=False)
    import matplotlib.pyplot as plt
    plt.close("all")

    for name, (ber, bler) in results.items():
        assert ber[0] > 1e-3 and bler[0] > 1e-2, name                  # errors at 0 dB
        sim = bler > 0
        assert np.all(np.diff(bler[sim]) < 0), (name, bler)            # waterfall
    # list decoding beats successive cancellation, as in the notebook's figure
    assert results["5G Polar+CRC SCL-8"][1][2] < results["5G Polar+CRC SC"][1][2]

    # ---- the notebook's threshold-search idiom: tf.constant scalars, count_block_errors, .numpy()
    enc = Polar5GEncoder(k=32, n=160)
    model = SystemModel(k=32, n=160, num_bits_per_symbol=2, encoder=enc,
                        decoder=Polar5GDecoder(enc, dec_type="SCL", list_size=8), sim_esno=True)
    u, u_hat = model(tf.constant(500, tf.int32), tf.constant(-2.0, tf.float32))
    nerr = count_block_errors(u, u_hat)
    assert u.numpy().shape == (500, 32) and 0 <= int(nerr) <= 500
    u, u_hat = model(tf.constant(500, tf.int32), tf.constant(6.0, tf.float32))
    assert int(count_block_errors(u, u_hat)) == 0
