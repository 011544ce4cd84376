"""Data loaders with the reference's conventions (prepare.py) + synthetic generators for the
benchmark configurations (BASELINE.json `configs`).

File formats (prepare.py:10-37): `data/<DATA>/data.csv` with columns
`user,item[,shifted_item],rating|outcome`; `trainval.csv` / `test.csv` with an `index` column.
Item ids are shifted by N on disk (`shifted_item = item + N`, prepare.py:47); when the column is
missing the reference falls back to the unshifted `item` (prepare.py:20-21).
"""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np


def load_data(DATA, output_type="reg", root="data"):
    """Same contract as the reference's `load_data` (prepare.py:10-37):
    returns (N, M, X_train, X_test, y_train, y_test, i)."""
    import pandas as pd
    if DATA == "fr_en":
        columns = ["format", "shifted_item", "user"]
        outcome = "outcome"
    else:
        columns = ["user", "shifted_item"]
        outcome = "rating" if output_type == "reg" else "outcome"
    path = Path(root) / DATA
    df = pd.read_csv(path / "data.csv")
    if "shifted_item" not in df.columns:
        df["shifted_item"] = df["item"]
    i = {"trainval": pd.read_csv(path / "trainval.csv")["index"].tolist(),
         "test": pd.read_csv(path / "test.csv")["index"].tolist()}
    tr = df.loc[i["trainval"], [outcome] + columns]
    te = df.loc[i["test"], [outcome] + columns]
    return (df["user"].nunique(), df["item"].nunique(), tr[columns].to_numpy(), te[columns].to_numpy(),
            tr[outcome].to_numpy(), te[outcome].to_numpy(), i)


def prepare_data(DATA, is_classification, root="data"):
    """prepare.py:39-64: re-index users / items to 0..N-1 / 0..M-1, add `shifted_item = item + N`
    and `outcome = rating >= 4`, rewrite data.csv and export libFM text files
    `<outcome> <user>:1 <shifted_item>:1`."""
    import pandas as pd
    outcome = "outcome" if is_classification else "rating"
    path = Path(root) / DATA
    df = pd.read_csv(path / "data.csv")
    df["user"] = np.unique(df["user"], return_inverse=True)[1]
    df["item"] = np.unique(df["item"], return_inverse=True)[1]
    df["shifted_item"] = df["item"] + df["user"].nunique()
    i = {k: pd.read_csv(path / f"{k}.csv")["index"].tolist() for k in ("trainval", "test")}
    df["outcome"] = (df["rating"] >= 4).astype(int)
    df.to_csv(path / "data.csv", index=False)
    for name in ("trainval", "test"):
        out = path / f"{DATA}.{name}_libfm"
        if not out.is_file():
            rows = df.loc[i[name], ["user", "shifted_item", outcome]].to_numpy()
            with open(out, "w") as f:
                for u, it, o in rows:
                    f.write("{:d} {:d}:1 {:d}:1\n".format(int(o), int(u), int(it)))
    return df


def load_libfm(train_path, test_path):
    """The `movie100` branch of the reference (vfm-torch.py:31-57): libFM text files with lines
    `<outcome> <user>:1 <item>:1` (as written by `prepare_data`).  Returns
    (N, M, X_train, X_test, y_train, y_test) with N / M counted over train + test like the reference
    (:48-50); ids are used as they appear in the files (already shifted when exported by prepare_data)."""
    import pandas as pd

    def read(path):
        df = pd.read_csv(path, names=("outcome", "user", "item"), sep=" ")
        df["user"] = df["user"].map(lambda t: t[:-2])          # strip the ":1" (vfm-torch.py:38-39)
        df["item"] = df["item"].map(lambda t: t[:-2])
        return df.astype(int)

    tr, te = read(train_path), read(test_path)
    both = pd.concat((tr, te), axis=0)
    N, M = int(both["user"].nunique()), int(both["item"].nunique())
    return (N, M, tr[["user", "item"]].to_numpy(), te[["user", "item"]].to_numpy(),
            tr["outcome"].to_numpy(), te["outcome"].to_numpy())


def load_ratings_frame(df, user_col="userId", item_col="movieId", rating_col="rating", test_size=0.2,
                       seed=None):
    """The `movielens` / parquet branches of the reference (vfm-torch.py:60-73,96-117): re-index users
    and items with np.unique, shift item ids by N, shuffle-split 80/20 (the reference's
    train_test_split(shuffle=True) is unseeded; pass `seed` for reproducibility).
    Returns (N, M, X_train, X_test, y_train, y_test)."""
    user = np.unique(df[user_col], return_inverse=True)[1]
    item = np.unique(df[item_col], return_inverse=True)[1]
    N, M = int(user.max()) + 1, int(item.max()) + 1
    X = np.stack([user, item + N], 1).astype(np.int64)
    y = np.asarray(df[rating_col], dtype=np.float32)
    perm = np.random.default_rng(seed).permutation(len(y))
    n_te = int(round(test_size * len(y)))
    te, tr = perm[:n_te], perm[n_te:]
    return N, M, X[tr], X[te], y[tr], y[te]


def read_config(path, df=None):
    """(nb_users, nb_items) of a data directory the way the TF sibling takes them (vfm.py:97-104): from the
    optional `config.yml` (keys `nb_users`, `nb_items`), else guessed from the data as 1 + the largest id."""
    cfg = Path(path) / "config.yml"
    if cfg.is_file():
        import yaml
        with open(cfg) as f:
            c = yaml.safe_load(f)
        return int(c["nb_users"]), int(c["nb_items"])
    if df is None:
        import pandas as pd
        df = pd.read_csv(Path(path) / "data.csv")
    return 1 + int(df["user"].max()), 1 + int(df["item"].max())


def load_fraction(path, test_size=0.2, seed=0):
    """The shipped toy set `data/fraction/data.csv` (536 users x 20 items, binary `outcome`).
    It has no split files, no `shifted_item` column and no `config.yml`, so: N, M come from `read_config`
    (1 + max id, vfm.py:102-104: 536, 20), item ids are shifted by N here (vfm.py:105) and the
    split is a seeded 80/20 shuffle (the TF sibling falls back to train_test_split, vfm.py:211-212).
    Rows whose entities never occur in the training part are dropped from the test part.
    Returns (N, M, X_train, X_test, y_train, y_test)."""
    import pandas as pd
    df = pd.read_csv(os.path.join(path, "data.csv"))
    N, M = read_config(path, df)          # config.yml if the directory has one (vfm.py:97-104), else 1 + max id
    X = np.stack([df.user.to_numpy(), df.item.to_numpy() + N], 1).astype(np.int64)
    y = df.outcome.to_numpy().astype(np.float32)
    perm = np.random.default_rng(seed).permutation(len(y))
    n_te = int(round(test_size * len(y)))
    te, tr = perm[:n_te], perm[n_te:]
    occ = np.bincount(X[tr].reshape(-1), minlength=N + M)
    te = te[(occ[X[te]] > 0).all(1)]
    return N, M, X[tr], X[te], y[tr], y[te]


def synthetic_triples(field_sizes, n, seed=0, output="reg", device="cpu", zipf=None):
    """Uniform-random (user, item+N, ...) ids for F fields with consecutive id ranges and
    ratings `randint(1,6)` ('reg') or Bernoulli(0.25) labels ('class') -- the synthetic inputs of
    SURVEY.md 8(d).  Generated with a seeded torch.Generator on `device`."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    cols, lo = [], 0
    for f, sz in enumerate(field_sizes):
        if zipf and f > 0:
            # Zipf-like popularity on the non-user fields (contention study)
            w = 1.0 / torch.arange(1, sz + 1, dtype=torch.float64, device=device) ** zipf
            ids = torch.multinomial(w, n, replacement=True, generator=g)
        else:
            ids = torch.randint(0, sz, (n,), generator=g, device=device)
        cols.append(ids + lo)
        lo += sz
    X = torch.stack(cols, 1).contiguous()
    if output == "reg":
        y = torch.randint(1, 6, (n,), generator=g, device=device).to(torch.float32)
    else:
        y = (torch.rand(n, generator=g, device=device) < 0.25).to(torch.float32)
    return X, y
