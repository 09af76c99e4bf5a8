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

Parity status: LMMSE, TDL, AWGN and end-to-end BER are "parity unpinned" by value in the
reference (statistical / smoke tests only); for those the oracle
restates the formulas and is checked by invariants.
"""
