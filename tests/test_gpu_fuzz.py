"""GPU: the randomised parity sweep of tests/fuzz_parity.py as a collected, fixed-seed test -- random
(F, d, B, T, likelihood, link, S, id width, skew) configurations, kernels vs the fp64 row-wise oracle: loss,
predictions, every gradient, prediction launches, the fused backward+Adam forms.  Both forward kernels
(k_fwd2, the task-stream form F = 2 problems with d >= 20 take, and k_fwd) are covered: the second run forces k_fwd,
the third k_fwd2 wherever it is defined (small d included)."""
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,n,force", [(0, 220, None), (1, 120, "1"), (2, 120, "2")])
def test_fuzz_parity_fixed_seed(seed, n, force, monkeypatch):
    import fuzz_parity
    if force:
        monkeypatch.setenv("VFM_FWD_KERNEL", force)
    bad, worst = fuzz_parity.sweep(n, seed, verbose=False)
    assert not bad, bad[:3]
    assert worst["loss"] < 1e-4 and worst["pred"] < 2e-4
