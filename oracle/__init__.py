"""CPU oracle: a restatement of the reference algorithms of the hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``sionna_amd/`` imports this package; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do,
and there only as the checker / the timed CPU baseline, never as the product path.

Every function cites the reference file:line it follows (paths relative to
``/root/reference/src/sionna/phy``).  The reference itself (TensorFlow) cannot be
imported in this environment, so the oracle is pinned
against the reference's own golden vectors and embedded NumPy test formulas instead:

* LDPC encoder     -> 28 golden generator matrices ``test/codes/ldpc/k*_n*_G.npy``
                      (fixtures derived from them: ``tests/golden/ldpc_enc_golden.npz``)
* BP node updates  -> per-node formulas of ``test/unit/fec/test_ldpc_decoding.py:400-655``
* demapper         -> ``test/unit/mapping/test_mapping.py:175-199`` (scipy logsumexp)
* CRC / Polar      -> ``test/codes/crc/*.npy``, ``test/codes/polar/*.npy``

Since round 4 the oracle is ALSO pinned against the reference's own source files EXECUTED
here under a NumPy stand-in for TensorFlow (``tools/ref_exec``; fixtures
``tests/golden/*_ref_*``, tests ``tests/test_oracle_ref_exec*.py``,
``tests/test_sim_ber_ref_exec.py``, ``tests/test_fec_utils_ref_exec.py``): BP node updates and
whole decoders (min-sum family, VN update, state, layered: bit for bit), 5G LDPC / Polar
encoders and Polar SC / SCL / hybrid decisions (bit for bit), the Polar BP decoder (``polar_bp.py``: soft outputs bit
for bit on NumPy's exp / log, ``tests/test_oracle_ref_exec_polar_bp.py``), mapper / demapper, the symbol-domain blocks and
the ``output="symbol"`` forms of the EP / K-Best / MMSE-PIC detectors (``tests/test_oracle_ref_exec_symbol.py``), MIMO
equalisers, OFDM modulator / demodulator / time channel, LS estimators, OFDM detectors, the
IDD chain, ``sim_ber`` (bit for bit), TDL / CDL generators (parameters exactly, realisations
statistically), and against the BER / BLER tables the reference publishes in its notebooks
(``tests/test_gpu_ber_reference.py``).

Still "parity unpinned" by construction: the VALUES of random streams (bits, noise, channel
draws, pilot symbols) - the reference takes them from TensorFlow's generators, this build from
its own Philox specification (``oracle/utils.py``) - and TensorFlow's own last bits of exp / log
in the boxplus rules (``oracle/ldpc_bp.c`` defines its arithmetic; 1e-5 against the reference's
formulas on NumPy's exp / log).
"""
