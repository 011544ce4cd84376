#!/usr/bin/env python3
"""BASELINE configs[0] end to end: the reference's toy `fraction` data set (536 students x 20 questions, binary
outcomes; tests/golden/fraction/data.csv is the reference's data/fraction/data.csv), d = 5, `output='class'`, trained
with the reference's loop (vfm-torch.py:347-422: one batch per epoch, lr = 1/(1 + nb_train // batch), dense Adam) through
the HIP path -- what a user of `vfm-torch.py` runs after switching to this package:

    python examples/fit_fraction.py [n_epochs]          (needs an MI355X; vae_amd has no CPU fallback)

Prints the per-epoch ELBO / train AUC the reference prints, the test AUC / MAP of its evaluation block, and the
posterior-predictive mean + logit variance of a few test pairs (`predict_samples`: what the paper's preference-elicitation
use case consumes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vae_amd.model import VFM
from vae_amd.data import load_fraction


def main():
    n_epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    N, M, X_train, X_test, y_train, y_test = load_fraction(os.path.join(ROOT, "tests", "golden", "fraction"))
    torch.manual_seed(42)                                   # same init order as CF.__init__ (vfm-torch.py:136-153)
    model = VFM(N, M, embedding_size=5, output="class", device="cuda")
    hist = model.fit(X_train, y_train, n_epochs=n_epochs, batch_size=100000, X_test=X_test, y_test=y_test,
                     display_every=max(1, n_epochs // 6), verbose=True)
    print("final test metrics:", hist["test"][-1])
    unc = model.predict_samples(X_test[:5], n_samples=50)
    for row, p, v in zip(X_test[:5].tolist(), unc["mean"].tolist(), unc["logits_var"].tolist()):
        print(f"student {row[0]:4d} question {row[1] - N:2d}: P(correct) = {p:.3f}, logit variance = {v:.3f}")


if __name__ == "__main__":
    main()
