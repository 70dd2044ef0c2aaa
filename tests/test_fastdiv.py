"""The multiply-shift division of the kernels' index prologues (fusion_gcn_amd/csrc/fgcn_common.hpp, FastDiv / make_fastdiv / fastdiv): the
host restatement of its two formulas must equal the integer quotient for every dividend below 2^29 -- checked at the ends of every quotient
step for the divisors the launchers pass (joints per frame, rows per sample, frames per view) and on random pairs."""
import re
from pathlib import Path

import numpy as np

HDR = Path(__file__).resolve().parents[1] / "fusion_gcn_amd" / "csrc" / "fgcn_common.hpp"
LIMIT = 1 << 29


def make_fastdiv(d: int):
    L = 0
    while (1 << L) < d:
        L += 1
    p = 29 + L
    return ((1 << p) + d - 1) // d, p


def test_header_states_the_formulas_restated_here():
    text = HDR.read_text()
    assert re.search(r"f\.p = 29 \+ L;", text) and re.search(r"f\.m = \(unsigned\)\(\(\(1ull << f\.p\) \+ d - 1\) / d\);", text)
    assert re.search(r"\(\(unsigned long long\)n \* f\.m\) >> f\.p", text)


def test_multiplier_fits_32_bits_and_the_product_64():
    for d in list(range(1, 5000)) + [25 * 300, 25 * 300 * 2, 27 * 150, 1956, 65535, (1 << 20) + 1, LIMIT - 1]:
        m, p = make_fastdiv(d)
        assert m < (1 << 32) and p <= 29 + 29 and (LIMIT - 1) * m < (1 << 64)


def test_quotients_are_exact_below_2_pow_29():
    rng = np.random.default_rng(5)
    divisors = list(range(1, 200)) + [300, 652, 1875, 1956, 3750, 7500, 15000, 25 * 300 * 2, 65537, 999983] + rng.integers(1, 1 << 22, 300).tolist()
    for d in divisors:
        m, p = make_fastdiv(int(d))
        qmax = (LIMIT - 1) // d
        qs = np.unique(np.concatenate([np.arange(0, min(qmax, 2000) + 1), rng.integers(0, qmax + 1, 4000), [qmax]])).astype(object)
        for off in (0, d - 1):                                   # both ends of every quotient step
            n = qs * d + off
            n = n[n < LIMIT]
            got = (n * m) >> p
            assert (got == n // d).all(), d
    n = rng.integers(0, LIMIT, 200000).astype(object)
    d = rng.integers(1, 1 << 20, 200000).astype(object)
    for ni, di in zip(n[:20000], d[:20000]):
        m, p = make_fastdiv(int(di))
        assert (int(ni) * m) >> p == int(ni) // int(di)
