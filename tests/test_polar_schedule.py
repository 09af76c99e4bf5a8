"""Host logic of the Polar decoding schedule (sionna_amd/phy/fec/polar/decoding.py): the fused records handed to the
kernels expand to exactly the operations of the plain schedule, which follows the reference recursion
(src/sionna/phy/fec/polar/decoding.py:919-1005)."""
import numpy as np
import pytest

from sionna_amd.phy.fec.polar import decoding as pd
from sionna_amd.phy.fec.polar.utils import generate_5g_ranking


def _frozen(k, n):
    frozen, _ = generate_5g_ranking(k, n)
    ind = np.zeros(n, int)
    ind[frozen] = 1
    return ind


def _expand(ops, frozen_ind):
    """SUBTREE records back into the plain operations (what the kernels execute for them)."""
    out = []

    def node(start, s, side, fast):
        size = 1 << s
        if s == 0:
            out.append((pd.OP_LEAF, 0, side, -1 - start if frozen_ind[start] else start))
            return
        blk = frozen_ind[start:start + size]
        if fast and blk.sum() == size:
            out.append((pd.OP_RATE0, s, side, 0))
            return
        if fast and blk[-1] == 0 and blk[:-1].sum() == size - 1:
            out.append((pd.OP_REP, s, side, start + size - 1))
            return
        out.append((pd.OP_F, s, 0, 0))
        node(start, s - 1, 0, fast)
        out.append((pd.OP_G, s, 0, 0))
        node(start + size // 2, s - 1, 1, fast)
        out.append((pd.OP_COMBINE, s - 1, side, 0))

    for op, a0, a1, a2 in ops:
        if op == pd.OP_SUBTREE:
            node(int(a2) & 4095, int(a0), int(a1), bool(int(a2) & 4096))
        else:
            out.append((int(op), int(a0), int(a1), int(a2)))
    return np.asarray(out, np.int32)


@pytest.mark.parametrize("k,n", [(523, 1024), (100, 256), (40, 64), (300, 512), (16, 32)])
@pytest.mark.parametrize("fast", [True, False])
def test_fused_records_expand_to_the_plain_schedule(k, n, fast):
    ind = _frozen(k, n)
    plain = pd.build_schedule(ind, use_fast=fast)
    fused = pd.fuse_schedule(plain)
    assert np.array_equal(_expand(fused, ind), plain)
    assert len(fused) <= len(plain)
    for r in (1, 2, 3, 4, 5):
        if n < (1 << (r + 1)):
            continue
        sub = pd.build_schedule(ind, use_fast=fast, subtree_stage=r)
        assert np.array_equal(_expand(sub, ind), plain), (r, fast)
        recs = sub[sub[:, 0] == pd.OP_SUBTREE]
        assert np.all(recs[:, 1] == r) and np.all((recs[:, 3] & 4095) % (1 << r) == 0)
        assert np.all((recs[:, 3] >= 4096) == fast)


def test_packing_round_trip():
    ind = _frozen(523, 1024)
    sub = pd.build_schedule(ind, subtree_stage=4)
    packed = pd.pack_schedule(sub)
    assert packed.dtype == np.int32 and len(packed) == len(pd.fuse_schedule(sub))
    op, stage, side = packed & 7, (packed >> 3) & 15, (packed >> 7) & 1
    a2 = ((packed >> 8) & 0xFFF) - 2048
    flag = (packed >> 20) & 1
    assert np.array_equal(op, sub[:, 0]) and np.array_equal(side, sub[:, 2])
    m = sub[:, 0] == pd.OP_SUBTREE
    assert np.array_equal(stage[m], sub[m, 1]) and np.array_equal(a2[m], sub[m, 3] & 4095) and np.all(flag[m] == 1)
    assert np.array_equal(a2[~m], sub[~m, 3]) and np.all(flag[~m] == 0)
    assert op[-1] == pd.OP_END
    # the rate-1/2 n = 1024 schedule: 2161 plain operations, 385 with stage-3 subtrees, 213 with stage-4 subtrees
    assert len(pd.build_schedule(ind)) == 2161 and len(pd.build_schedule(ind, subtree_stage=3)) == 385
    assert len(sub) == 213
