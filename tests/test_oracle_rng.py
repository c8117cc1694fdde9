"""CPU: the noise-generator oracle (oracle/rng.py) against Philox4x32-10's published known-answer vectors (Random123 kat_vectors:
`philox4x32 10` rows -- all-zero, all-ones and the pi-digits counter / key), plus the statistical sanity of its two output kinds."""
import numpy as np

from oracle import rng


def test_philox4x32_10_known_answers():
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = rng.philox4x32_10(*[[c] for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want


def test_normal_and_keep_mask_statistics_and_counter_layout():
    x = rng.normal(256, 128, seed=7, step=3)
    assert abs(x.mean()) < 0.02 and abs(x.std() - 1.0) < 0.02 and np.isfinite(x).all()
    m = rng.keep_mask(256, 1024, 0.2, seed=7, step=3, stream=2)
    assert set(np.unique(m)) == {np.float32(0.0), np.float32(1.25)} and abs((m > 0).mean() - 0.8) < 0.01
    # a sample's numbers depend on its GLOBAL index only: two ranks of 4 draw what one rank of 8 draws
    a = rng.normal(8, 33, seed=1, step=5)
    b = np.concatenate([rng.normal(4, 33, seed=1, step=5, sample0=0), rng.normal(4, 33, seed=1, step=5, sample0=4)])
    assert np.array_equal(a, b)
    # step / stream / seed decorrelate
    assert not np.array_equal(a, rng.normal(8, 33, seed=1, step=6)) and not np.array_equal(a, rng.normal(8, 33, seed=2, step=5))
    assert not np.array_equal(rng.keep_mask(8, 64, 0.5, 1, stream=0), rng.keep_mask(8, 64, 0.5, 1, stream=1))
