"""GPU: `VFM.fit` at the ML-20M shape learns like the REFERENCE does.  tests/golden/bigfit_ml20m_d128.json holds the
per-epoch curves of the reference's own `CF` class (tools/make_bigfit_golden.py: ids of the ML-20M ranges, d = 128, 16
batches of 100,000 rows, lr = 1/17 by the reference's rule vfm-torch.py:92, 2 epochs of the loop :347-417, three sampler
seeds) on data tests/golden_util.py::bigfit_data regenerates here.  The GPU run starts from the same initial weights
(same init seed and RNG order: test_init_matches_reference_seed) but draws its eps from the in-kernel Philox stream, so
the comparison is statistical: every per-epoch figure of every GPU seed must lie within the reference's mean +- (4 x its
spread over seeds + 0.5 %).

This also answers why round 2's big-shape demo (profiles/r02_fit_demo.txt) showed a flat train RMSE of 2.29 and an
`rmse_of_last` of 10: the reference does exactly that at lr = 1/17 from an N(0,1) initialisation (fixture: train RMSE
2.30 -> 2.27, test RMSE of last mean 11.0 -> 10.5 over the same two epochs)."""
import json
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN, BIGFIT, bigfit_data

pytestmark = pytest.mark.gpu


def test_fit_matches_the_reference_learning_curves_at_ml20m_shape():
    from vae_amd.model import VFM
    ref = json.load(open(os.path.join(GOLDEN, "bigfit_ml20m_d128.json")))
    c = BIGFIT
    Xtr, ytr, Xte, yte = bigfit_data()
    assert len(ytr) == ref["nb_train"] and len(yte) == ref["nb_test"]
    keys = ("elbo", "train_rmse", "test_rmse", "test_rmse_all", "test_rmse_of_last", "test_rmse_of_mean")
    want = {k: np.array([[e[k] for e in run["epochs"]] for run in ref["runs"]]) for k in keys}      # [seed, epoch]
    got = {k: [] for k in keys}
    for seed in (101, 102, 103):
        torch.manual_seed(ref["init_seed"])
        m = VFM(c["N"], c["M"], c["d"], device="cuda", rng_seed=seed)
        h = m.fit(Xtr, ytr, n_epochs=c["n_epochs"], batch_size=c["batch"], X_test=Xte, y_test=yte, verbose=False)
        assert abs(m.lr - ref["lr"]) < 1e-12
        got["elbo"].append(h["elbo"]); got["train_rmse"].append(h["train_rmse"])
        for k, hk in (("test_rmse", "rmse"), ("test_rmse_all", "rmse_all"), ("test_rmse_of_last", "rmse_of_last"),
                      ("test_rmse_of_mean", "rmse_of_mean")):
            got[k].append([t[hk] for t in h["test"]])
    for k in keys:
        g, w = np.array(got[k]), want[k]
        mean, spread = w.mean(axis=0), w.std(axis=0)
        tol = 4 * spread + 5e-3 * np.abs(mean)
        assert (np.abs(g - mean) <= tol).all(), (k, g.tolist(), mean.tolist(), tol.tolist())
    # and both learn the same little in two epochs at this learning rate: the ELBO falls by > 3x, the RMSEs barely move
    assert want["elbo"][:, 1].mean() < 0.35 * want["elbo"][:, 0].mean()
    assert np.array(got["elbo"])[:, 1].mean() < 0.35 * np.array(got["elbo"])[:, 0].mean()
