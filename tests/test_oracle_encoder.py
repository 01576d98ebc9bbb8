"""Oracle vs the reference's own encoder outputs (tests/golden/encoder_golden.npz,
made by shim-importing /root/reference/src/ann_solo/spectrum.py)."""
import numpy as np


def test_murmur_kats(O, golden):
    g = golden('encoder_golden.npz')
    for key, seed, want in zip(g['murmur_kat_key'], g['murmur_kat_seed'], g['murmur_kat_hash']):
        assert O.murmur3_32(bytes.fromhex(str(key)), int(seed)) == int(want)


def test_hash_idx_and_get_dim(O, golden):
    g = golden('encoder_golden.npz')
    assert [O.hash_idx(int(b), 800) for b in g['bins']] == g['hashes'].tolist()
    assert [O.hash_idx(int(b), 64) for b in g['bins']] == g['hashes64'].tolist()
    for args, want in zip(g['dim_args'], g['dims']):
        n, s, e = O.get_dim(*args)
        assert n == int(want[0]) and s == want[1] and e == want[2]
    assert O.get_dim(11, 2010, 0.04) == (49976, 10.96, 2010.0)


def test_bin_idx_matches_numpy_floor_division(O, golden):
    g = golden('encoder_golden.npz')
    got = np.array([O.bin_idx(m, 10.96, 0.04) for m in g['mz']], np.int64)
    assert np.array_equal(got, g['bin_idx'])
    # the 0.12 // 0.04 == 2.0 quirk
    assert O.lib().orc_npy_floor_divide(0.12, 0.04) == 2.0


def test_vectors_match_reference(O, golden):
    g = golden('encoder_golden.npz')
    for key, bin_size, hl, norm in (('vec', 0.04, 800, True), ('vec_nonorm', 0.04, 800, False),
                                    ('vec_h64', 0.05, 64, True)):
        mb = O.get_dim(11, 2010, bin_size)[1]
        got = O.encode_batch(g['mz'], g['intensity'], g['offsets'], mb, bin_size, hl, 42, norm)
        want = g[key]
        # identical support (integer work: bit-exact) ...
        assert np.array_equal(got != 0, want != 0)
        if not norm:
            assert np.array_equal(got, want)          # fp32 adds in peak order: bit-exact
        else:
            # np.linalg.norm is a BLAS sdot whose summation order is build-specific;
            # the oracle's canonical norm agrees to a few ulp.
            np.testing.assert_allclose(got, want, rtol=4e-7, atol=0)
            nrm = np.sqrt((got.astype(np.float64) ** 2).sum(1))
            assert np.all(np.abs(nrm - 1) < 1e-6)
