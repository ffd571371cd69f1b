"""Known-answer tests of the oracle's Philox4x32-10 restatement (CPU)."""
import numpy as np

from oracle import gse_oracle as orc


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 with 10 rounds
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        got = orc.philox4x32_10(np.array(ctr, dtype=np.uint32), np.array(key, dtype=np.uint32))
        assert tuple(int(v) for v in got) == want


def test_words_to_draw_statistics():
    sp, _ = orc.bioreactor_noise()
    words = orc.rng_words8(seed=1234567, step=3, purpose=orc.RNG_PURPOSE["draw"], elements=np.arange(200000))
    x, comp = orc.words_to_draw(words, sp)
    frac = np.bincount(comp, minlength=2) / len(comp)
    assert abs(frac[0] - 0.75) < 0.005
    for d in range(2):
        sel = x[comp == d]
        var = sel.var(axis=0)
        np.testing.assert_allclose(var, np.diag(sp.covariances[d]), rtol=0.03)
        assert np.all(np.abs(sel.mean(axis=0)) < 4 * np.sqrt(np.diag(sp.covariances[d]) / len(sel)))


def test_gsukf_predict_noise_statistics():
    """The Gaussian-sum filter's noise mapping (sigma points in pairs sharing three Philox blocks): every point's draw has the
    mixture's component frequencies and moments, and the two halves of a pair -- which share one Box-Muller pair and one
    block of selectors -- are uncorrelated, as are neighbouring pairs and steps."""
    sp, _ = orc.bioreactor_noise()
    n = 20000
    x, comp = orc.gsukf_predict_noise(seed=987654321, step=5, n_gaussians=n, mixture=sp)
    assert x.shape == (n, 11, 5) and comp.shape == (n, 11)
    for j in range(11):
        frac = np.mean(comp[:, j] == 0)
        assert abs(frac - 0.75) < 0.015, (j, frac)
        for d in range(2):
            sel = x[comp[:, j] == d, j]
            np.testing.assert_allclose(sel.var(axis=0), np.diag(sp.covariances[d]), rtol=0.12, err_msg=str((j, d)))
            assert np.all(np.abs(sel.mean(axis=0)) < 5 * np.sqrt(np.diag(sp.covariances[d]) / len(sel)))
    # standardised draws: correlations between the points of a Gaussian stay at the sampling noise (1 / sqrt(n) = 0.007)
    sd = np.sqrt(np.array([np.diag(c) for c in sp.covariances]))[comp]            # (n, 11, 5)
    zs = (x / sd).reshape(n, 55)
    corr = np.corrcoef(zs, rowvar=False)
    off = corr - np.eye(55)
    assert np.max(np.abs(off)) < 0.04, float(np.max(np.abs(off)))
    assert abs(np.corrcoef(comp[:, 0], comp[:, 1])[0, 1]) < 0.03                     # the pair's selectors: words 3 and 2 of B
    y, _ = orc.gsukf_predict_noise(seed=987654321, step=6, n_gaussians=n, mixture=sp)
    assert abs(np.corrcoef(x[:, 3, 0], y[:, 3, 0])[0, 1]) < 0.03
