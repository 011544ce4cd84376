"""Helpers to read the golden fixtures written by tools/make_golden.py."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

SINGLE_CASES = ["quirk_reg_d8", "fraction_class_d5", "ml100k_reg_d20", "ml100k_class_d20",
                "ml20m_reg_d128", "dup_reg_d12", "dup_class_d12"]
F64_CASES = ["quirk_reg_d8", "fraction_class_d5", "ml100k_reg_d20", "dup_reg_d12"]
# the reference's other two globals of the path (vfm-torch.py:19,125-126): S > 1 samples, LINK = softplus
VARIANT_CASES = ["multi_reg_d8_s3", "softplus_reg_d8", "softplus_multi_class_d8_s2", "dup_multi_reg_d12_s2"]

PARAM_KEYS = ("alpha", "global_bias_mean", "global_bias_scale", "bias_params", "entity_params")


class Case:
    """One single-step fixture with dense tables rebuilt (fixtures of big tables
    only store the rows the batch touches)."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.z = z
        self.name = name
        self.N, self.M, self.d = int(z["N"]), int(z["M"]), int(z["d"])
        self.T = self.N + self.M
        self.nb_train = int(z["nb_train"])
        self.output = str(z["output"])
        self.x, self.y, self.nb_occ = z["x"], z["y"], z["nb_occ"]
        self.uniq = z["uniq"]
        self.sparse = "sparse_rows" in z.files
        self.n_samples = int(z["n_samples"]) if "n_samples" in z.files else 1
        self.link = str(z["link"]) if "link" in z.files else "abs"
        self.group_hi = np.array([self.N + 1, self.T], dtype=np.int64)   # `<= N` quirk
        self.group_n = np.array([self.N, self.M], dtype=np.float64)

    def params(self, dtype=np.float32):
        P = {k: self.z["p_" + k].astype(dtype) for k in PARAM_KEYS}
        if self.sparse:
            rng = np.random.default_rng(123)
            for k, w in (("bias_params", 2), ("entity_params", 2 * self.d)):
                full = rng.standard_normal((self.T, w)).astype(dtype)
                full[self.uniq] = P[k]
                P[k] = full
        return P

    def eps(self, tag="f32", dtype=None):
        """eps0[1], eps_w[T], eps_v[T,d] scattered to entity-id order ([S], [S,T], [S,T,d] for
        fixtures with S > 1 variational samples)."""
        z = self.z
        dt = dtype or z[f"{tag}_eps_v"].dtype
        lead = () if self.n_samples == 1 else (self.n_samples,)
        ew = np.zeros(lead + (self.T,), dt)
        ev = np.zeros(lead + (self.T, self.d), dt)
        ew[..., self.uniq] = z[f"{tag}_eps_w"]
        ev[..., self.uniq, :] = z[f"{tag}_eps_v"]
        return z[f"{tag}_eps0"].astype(dt), ew, ev

    def eps_uniq(self, tag="f32"):
        z = self.z
        return z[f"{tag}_eps0"], z[f"{tag}_eps_w"], z[f"{tag}_eps_v"]

    def expected(self, key, tag="f32"):
        v = self.z[f"{tag}_{key}"]
        if self.sparse and key in ("g_bias_params", "g_entity_params"):
            full = np.zeros((self.T, v.shape[1]), v.dtype)
            full[self.uniq] = v
            return full
        return v


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = max(np.abs(b).max(), 1e-30)
    return np.abs(a - b).max() / den


# ---- the big-shape learning-curve fixture (tools/make_bigfit_golden.py, tests/test_gpu_bigfit.py)
BIGFIT = {"N": 138493, "M": 26744, "d": 128, "batch": 100000, "n_train": 1600000, "n_test": 200000, "n_epochs": 2,
          "sampler_seeds": [0, 1, 2]}


def bigfit_data():
    """Synthetic low-rank ratings with ML-20M id ranges from a seeded torch CPU generator (the same values wherever this
    runs): y = clip(3.2 + <U_u, V_i> + 0.3 noise, 1, 5).  Returns (X_train, y_train, X_test, y_test), ids int64."""
    import torch
    c = BIGFIT
    g = torch.Generator().manual_seed(20260101)
    k = 8
    U, V = torch.randn(c["N"], k, generator=g) * 0.7, torch.randn(c["M"], k, generator=g) * 0.7
    n = c["n_train"] + c["n_test"]
    u = torch.randint(0, c["N"], (n,), generator=g)
    i = torch.randint(0, c["M"], (n,), generator=g)
    y = (3.2 + (U[u] * V[i]).sum(1) + 0.3 * torch.randn(n, generator=g)).clamp(1, 5)
    X = torch.stack([u, i + c["N"]], 1)
    nt = c["n_train"]
    return X[:nt].contiguous(), y[:nt].contiguous(), X[nt:].contiguous(), y[nt:].contiguous()
