"""The statistical bar of the mask-parity tests (tests/tools_metrics.py): thresholds and intervals quoted in DESIGN.md."""
from tools_metrics import P0, binom_min_successes, wilson


def test_binomial_bar_arithmetic():
    """Rejection thresholds at the 1 % level for p0 = 46/48 (the reference's own rate under another fp32 summation order):
    3 windows -> 2, 6 -> 4, 16 -> 13, 32 -> 28."""
    assert abs(P0 - 0.9583) < 1e-4
    assert [binom_min_successes(n) for n in (3, 6, 16, 32)] == [2, 4, 13, 28]
    lo, hi = wilson(31, 32)
    assert 0.84 < lo < 0.85 and 0.99 < hi < 1.0
    lo, hi = wilson(16, 16)
    assert 0.80 < lo < 0.81 and hi == 1.0
