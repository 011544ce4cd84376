"""CPU: the data loaders keep the reference's conventions (prepare.py:10-64) and the product package
never touches the oracle."""
import os
import re

import numpy as np
import pandas as pd
import pytest

from vae_amd import data as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_dataset(root, name="toy", shifted=True):
    path = root / name
    path.mkdir(parents=True)
    g = np.random.default_rng(0)
    n, N, M = 60, 7, 5
    df = pd.DataFrame({"user": g.integers(0, N, n), "item": g.integers(0, M, n), "rating": g.integers(1, 6, n)})
    df.loc[:N - 1, "user"] = np.arange(N)          # every user / item present
    df.loc[:M - 1, "item"] = np.arange(M)
    if shifted:
        df["shifted_item"] = df["item"] + N
    df["outcome"] = (df["rating"] >= 4).astype(int)
    df.to_csv(path / "data.csv", index=False)
    idx = g.permutation(n)
    pd.DataFrame({"index": idx[:45]}).to_csv(path / "trainval.csv", index=False)
    pd.DataFrame({"index": idx[45:]}).to_csv(path / "test.csv", index=False)
    return df, idx, N, M


def test_load_data_contract(tmp_path):
    df, idx, N, M = _write_dataset(tmp_path)
    n, m, Xtr, Xte, ytr, yte, i = D.load_data("toy", "reg", root=tmp_path)
    assert (n, m) == (N, M) and Xtr.shape == (45, 2) and Xte.shape == (15, 2)
    assert i["trainval"] == idx[:45].tolist() and i["test"] == idx[45:].tolist()
    assert np.array_equal(Xtr[:, 0], df.loc[idx[:45], "user"]) and np.array_equal(Xtr[:, 1], df.loc[idx[:45], "item"] + N)
    assert np.array_equal(ytr, df.loc[idx[:45], "rating"])
    _, _, _, _, yc, _, _ = D.load_data("toy", "class", root=tmp_path)
    assert set(np.unique(yc)) <= {0, 1}


def test_load_data_falls_back_to_unshifted_item(tmp_path):
    """prepare.py:20-21: without a `shifted_item` column the unshifted item id is used."""
    df, idx, N, M = _write_dataset(tmp_path, shifted=False)
    _, _, Xtr, _, _, _, _ = D.load_data("toy", "reg", root=tmp_path)
    assert np.array_equal(Xtr[:, 1], df.loc[idx[:45], "item"])


def test_prepare_data_reindexes_and_exports_libfm(tmp_path):
    df, idx, N, M = _write_dataset(tmp_path, shifted=False)
    raw = pd.read_csv(tmp_path / "toy" / "data.csv")
    raw["user"] = raw["user"] * 10 + 3                  # non-contiguous raw ids
    raw.drop(columns=["outcome"]).to_csv(tmp_path / "toy" / "data.csv", index=False)
    out = D.prepare_data("toy", False, root=tmp_path)
    assert out["user"].max() == N - 1 and (out["shifted_item"] == out["item"] + N).all()
    assert (out["outcome"] == (out["rating"] >= 4).astype(int)).all()
    lines = open(tmp_path / "toy" / "toy.trainval_libfm").read().splitlines()
    assert len(lines) == 45 and re.fullmatch(r"\d+ \d+:1 \d+:1", lines[0])
    r0 = out.loc[idx[0]]
    assert lines[0] == f"{int(r0.rating)} {int(r0.user)}:1 {int(r0.shifted_item)}:1"


def test_load_fraction_fixture():
    N, M, Xtr, Xte, ytr, yte = D.load_fraction(os.path.join(ROOT, "tests", "golden", "fraction"))
    assert (N, M) == (536, 20) and len(ytr) + len(yte) <= 10720 and len(ytr) == 8576
    assert Xtr[:, 0].max() < N and Xtr[:, 1].min() >= N and Xtr[:, 1].max() < N + M
    assert set(np.unique(ytr)) == {0.0, 1.0}


def test_synthetic_triples_ranges_and_determinism():
    X, y = D.synthetic_triples([10, 20, 5], 1000, seed=3)
    X2, y2 = D.synthetic_triples([10, 20, 5], 1000, seed=3)
    assert (X == X2).all() and (y == y2).all()
    assert X[:, 0].max() < 10 and X[:, 1].min() >= 10 and X[:, 1].max() < 30 and X[:, 2].min() >= 30 and X[:, 2].max() < 35
    assert y.min() >= 1 and y.max() <= 5
    Xz, _ = D.synthetic_triples([10, 2000], 5000, seed=1, zipf=1.1)
    counts = np.bincount(Xz[:, 1].numpy() - 10, minlength=2000)
    assert counts[:10].sum() > 5 * counts[-10:].sum()           # popular head


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "vae_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, re.M), f
                assert "vfm_oracle" not in txt, f


def test_libfm_round_trip(tmp_path):
    """prepare_data's libFM export (prepare.py:58-62) read back by the movie100-branch loader."""
    df, idx, N, M = _write_dataset(tmp_path, shifted=False)
    out = D.prepare_data("toy", True, root=tmp_path)
    n, m, Xtr, Xte, ytr, yte = D.load_libfm(tmp_path / "toy" / "toy.trainval_libfm", tmp_path / "toy" / "toy.test_libfm")
    assert (n, m) == (N, M) and Xtr.shape == (45, 2) and Xte.shape == (15, 2)
    assert np.array_equal(Xtr[:, 0], out.loc[idx[:45], "user"]) and np.array_equal(Xtr[:, 1], out.loc[idx[:45], "shifted_item"])
    assert np.array_equal(ytr, out.loc[idx[:45], "outcome"])


def test_load_ratings_frame_reindexes_and_splits():
    g = np.random.default_rng(1)
    df = pd.DataFrame({"userId": g.choice([3, 17, 99, 1000], 200), "movieId": g.choice([5, 6, 70, 800, 9000], 200),
                       "rating": g.integers(1, 11, 200) / 2})
    N, M, Xtr, Xte, ytr, yte = D.load_ratings_frame(df, seed=0)
    assert (N, M) == (4, 5) and len(ytr) == 160 and len(yte) == 40
    X = np.concatenate([Xtr, Xte])
    assert X[:, 0].min() == 0 and X[:, 0].max() == 3 and X[:, 1].min() == 4 and X[:, 1].max() == 8
    assert sorted(np.concatenate([ytr, yte]).tolist()) == sorted(df["rating"].astype(np.float32).tolist())


def test_loaders_against_the_reference_fixtures(tmp_path):
    """tests/golden/loader/ was produced by running the reference's own prepare.prepare_data / load_data
    (tools/make_loader_golden.py) on the committed raw files: the same raw files through vae_amd.data must give
    byte-identical data.csv / libFM exports and identical load_data results."""
    import shutil
    G = os.path.join(ROOT, "tests", "golden", "loader")
    d = tmp_path / "toy"
    d.mkdir()
    shutil.copy(os.path.join(G, "raw_data.csv"), d / "data.csv")
    for f in ("trainval.csv", "test.csv"):
        shutil.copy(os.path.join(G, f), d / f)
    exp = np.load(os.path.join(G, "expected_load_data.npz"))
    # load_data on the raw file: the `shifted_item` fallback (prepare.py:20-21)
    N, M, Xtr, Xte, ytr, yte, i = D.load_data("toy", "reg", root=tmp_path)
    assert (N, M) == (int(exp["raw_N"]), int(exp["raw_M"]))
    assert np.array_equal(Xtr, exp["raw_X_train"]) and np.array_equal(Xte, exp["raw_X_test"])
    assert np.array_equal(ytr, exp["raw_y_train"])
    # prepare_data: re-indexing, shifted_item, outcome, libFM exports (prepare.py:39-64)
    D.prepare_data("toy", False, root=tmp_path)
    assert open(d / "data.csv").read() == open(os.path.join(G, "expected_data.csv")).read()
    for name in ("trainval", "test"):
        assert open(d / f"toy.{name}_libfm").read() == open(os.path.join(G, f"expected.{name}_libfm")).read()
    for ot in ("reg", "class"):
        N, M, Xtr, Xte, ytr, yte, i = D.load_data("toy", ot, root=tmp_path)
        assert (N, M) == (int(exp[f"{ot}_N"]), int(exp[f"{ot}_M"]))
        assert np.array_equal(Xtr, exp[f"{ot}_X_train"]) and np.array_equal(Xte, exp[f"{ot}_X_test"])
        assert np.array_equal(ytr, exp[f"{ot}_y_train"]) and np.array_equal(yte, exp[f"{ot}_y_test"])
        assert i["trainval"] == exp[f"{ot}_i_trainval"].tolist() and i["test"] == exp[f"{ot}_i_test"].tolist()
    # the libFM export read back (the `movie100` branch, vfm-torch.py:31-57)
    n2, m2, Ltr, Lte, ly, _ = D.load_libfm(d / "toy.trainval_libfm", d / "toy.test_libfm")
    assert np.array_equal(Ltr, exp["reg_X_train"]) and np.array_equal(ly, exp["reg_y_train"])


def test_read_config_yml_and_fallback(tmp_path):
    """config.yml (nb_users / nb_items) as the TF sibling reads it (vfm.py:97-104), else 1 + the largest id."""
    pd.DataFrame({"user": [0, 2, 5], "item": [1, 1, 3], "outcome": [1, 0, 1]}).to_csv(tmp_path / "data.csv", index=False)
    assert D.read_config(tmp_path) == (6, 4)
    (tmp_path / "config.yml").write_text("nb_users: 10\nnb_items: 7\n")
    assert D.read_config(tmp_path) == (10, 7)
    N, M, Xtr, Xte, ytr, yte = D.load_fraction(str(tmp_path), test_size=0.0)
    assert (N, M) == (10, 7) and Xtr[:, 1].min() >= 10             # items shifted by the configured N
    frac = os.path.join(ROOT, "tests", "golden", "fraction")
    assert D.read_config(frac) == (536, 20)                        # the shipped toy set has no config.yml
