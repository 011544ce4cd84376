#!/usr/bin/env python3
"""Generate the loader fixtures by running the reference's own `prepare.py` (build container only).

A small raw data set -- non-contiguous user / item ids, ratings 1..5, trainval / test index files, the
layout `prepare.py` expects under `data/<DATA>/` -- is written to a scratch directory; the reference's
`prepare_data(DATA, is_classification)` (prepare.py:39-64) and `load_data(DATA, output_type)` (:10-37) are
imported from /root/reference and run on it.  What is committed under tests/golden/loader/ is DATA only:
the raw input files, the files the reference wrote (re-indexed data.csv, the two libFM exports) and the
arrays `load_data` returned.  tests/test_data_cpu.py feeds the same raw files to vae_amd.data and compares.
"""
import os
import shutil
import sys
import tempfile

import numpy as np
import pandas as pd

REF = os.environ.get("VFM_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "loader")


def main():
    sys.path.insert(0, REF)
    import prepare as ref                      # the reference module itself (imports only at top level)
    g = np.random.default_rng(12)
    n = 400
    users = g.choice([3, 7, 8, 15, 21, 22, 40, 41, 57, 90, 91, 120], n)          # raw, non-contiguous ids
    items = g.choice([1000, 1003, 1004, 1010, 1500, 1501, 1777, 2000, 2048], n)
    rating = g.integers(1, 6, n)
    raw = pd.DataFrame({"user": users, "item": items, "rating": rating})
    perm = g.permutation(n)
    tr, te = np.sort(perm[:320]), np.sort(perm[320:])
    os.makedirs(OUT, exist_ok=True)
    raw.to_csv(os.path.join(OUT, "raw_data.csv"), index=False)
    pd.DataFrame({"index": tr}).to_csv(os.path.join(OUT, "trainval.csv"), index=False)
    pd.DataFrame({"index": te}).to_csv(os.path.join(OUT, "test.csv"), index=False)

    tmp = tempfile.mkdtemp()
    cwd = os.getcwd()
    try:
        d = os.path.join(tmp, "data", "toy")
        os.makedirs(d)
        shutil.copy(os.path.join(OUT, "raw_data.csv"), os.path.join(d, "data.csv"))
        for f in ("trainval.csv", "test.csv"):
            shutil.copy(os.path.join(OUT, f), os.path.join(d, f))
        os.chdir(tmp)                                           # prepare.py reads Path('data') / DATA
        ref.prepare_data("toy", False)                          # prepare.py:39-64
        shutil.copy(os.path.join(d, "data.csv"), os.path.join(OUT, "expected_data.csv"))
        for name in ("trainval", "test"):
            shutil.copy(os.path.join(d, f"toy.{name}_libfm"), os.path.join(OUT, f"expected.{name}_libfm"))
        rec = {}
        for ot in ("reg", "class"):
            N, M, Xtr, Xte, ytr, yte, i = ref.load_data("toy", ot)    # prepare.py:10-37
            rec.update({f"{ot}_N": N, f"{ot}_M": M, f"{ot}_X_train": Xtr, f"{ot}_X_test": Xte,
                        f"{ot}_y_train": ytr, f"{ot}_y_test": yte,
                        f"{ot}_i_trainval": np.array(i["trainval"]), f"{ot}_i_test": np.array(i["test"])})
        # the fallback of load_data when `shifted_item` is missing (prepare.py:20-21): the raw file
        shutil.copy(os.path.join(OUT, "raw_data.csv"), os.path.join(d, "data.csv"))
        N, M, Xtr, Xte, ytr, yte, i = ref.load_data("toy", "reg")
        rec.update({"raw_N": N, "raw_M": M, "raw_X_train": Xtr, "raw_X_test": Xte, "raw_y_train": ytr})
        np.savez_compressed(os.path.join(OUT, "expected_load_data.npz"), **rec)
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
