#!/usr/bin/env python3
"""Per-epoch learning curves of the REFERENCE's own `CF` class at the ML-20M shape, for the statistical end-to-end check of
`VFM.fit` (tests/test_gpu_bigfit.py): ids of the ML-20M ranges (N = 138,493 users, M = 26,744 items), d = 128, a
1.6 M-triple training slice in 16 batches of 100,000 rows (the reference's BATCH_SIZE, vfm-torch.py:77), lr by the
reference's rule (:92), 2 epochs of the loop :347-370 with the end-of-epoch block :378-417, once per sampler seed.

Runs ONLY in the build container (needs /root/reference; `CF` is lifted with `ast` by tools/make_golden.py's helper --
nothing of the reference is copied).  What is stored is DATA: per seed and epoch the mean ELBO loss over the epoch's
batches, the train RMSE of the clipped sampled predictions (:379-381) and the four test RMSEs (:402-417).  The training
data itself is not stored: tests/golden_util.py::bigfit_data regenerates it from a seeded torch CPU generator.

usage: python tools/make_bigfit_golden.py      (~3 minutes on 8 cores; writes tests/golden/bigfit_ml20m_d128.json)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from torch import nn, distributions
from golden_util import bigfit_data, BIGFIT
import make_golden as MG


def main():
    N, M, d, B, n_epochs = BIGFIT["N"], BIGFIT["M"], BIGFIT["d"], BIGFIT["batch"], BIGFIT["n_epochs"]
    Xtr, ytr, Xte, yte = bigfit_data()
    nb_train = len(ytr)
    nb_occ = torch.bincount(Xtr.flatten(), minlength=N + M)          # vfm-torch.py:89
    lr = 1 / (1 + nb_train // B)                                      # :92
    ns = {"torch": torch, "nn": nn, "distributions": distributions, "np": np}
    CF, span = MG.lift_cf(ns)
    out = {"reference": "vfm-torch.py class CF lines %d-%d, loop :347-417" % span, "N": N, "M": M, "d": d, "batch": B,
           "nb_train": nb_train, "nb_test": len(yte), "lr": lr, "n_epochs": n_epochs, "init_seed": 42, "runs": []}
    for seed in BIGFIT["sampler_seeds"]:
        model = MG.make_model(ns, CF, N, M, d, nb_occ, "reg", torch.float32, 42)
        opt = torch.optim.Adam(model.parameters(), lr=lr)             # :339
        torch.manual_seed(seed)
        all_preds, run = [], {"sampler_seed": seed, "epochs": []}
        for epoch in range(n_epochs):
            t0 = time.time()
            losses, pred = [], []
            for lo in range(0, nb_train, B):                          # :351-370
                x, y = Xtr[lo:lo + B], ytr[lo:lo + B]
                lik, _, _, kl = model(x)
                loss = -lik.log_prob(y).mean() * nb_train + kl
                opt.zero_grad()
                loss.backward()
                opt.step()
                losses.append(float(loss))
                pred.append(lik.mean.squeeze().detach().numpy())
            p = np.clip(np.concatenate(pred), 1, 5)                   # :378-381
            model.save_weights()
            rec = {"elbo": float(np.mean(losses)), "train_rmse": float(np.sqrt(np.mean((p - ytr.numpy()) ** 2)))}
            with torch.no_grad():
                lik, last, mean, _ = model(Xte)                       # :402-417
            yp = lik.mean.squeeze().detach().numpy().clip(1, 5)
            all_preds.append(yp)
            rm = lambda a: float(np.sqrt(np.mean((np.asarray(a) - yte.numpy()) ** 2)))
            rec.update(test_rmse=rm(yp), test_rmse_all=rm(np.mean(all_preds, axis=0)), test_rmse_of_last=rm(last),
                       test_rmse_of_mean=rm(np.clip(mean, 1, 5)))
            run["epochs"].append(rec)
            print(seed, epoch, rec, "%.0f s" % (time.time() - t0), flush=True)
        out["runs"].append(run)
    out["mean_predictor_test_rmse"] = float(np.sqrt(np.mean((float(ytr.mean()) - yte.numpy()) ** 2)))
    with open(os.path.join(ROOT, "tests", "golden", "bigfit_ml20m_d128.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
