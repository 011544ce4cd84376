"""GPU: vfm_build_index (the stable radix sort of csrc/vfm_index.hip) against numpy's stable argsort -- the
inverted index must be EXACTLY the stable counting sort (row order inside every entity's list), the heavy
lists and their work items exactly the specified cut, for every id width, skew and size class."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(x, T, F, L, thr=None):
    thr = L if thr is None else thr
    flat = x.reshape(-1).astype(np.int64)
    order = np.argsort(flat, kind="stable")
    occ_rows = (order // F).astype(np.int32)
    counts = np.bincount(flat, minlength=T)
    occ_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    heavy = np.nonzero(counts > thr)[0]          # thr = vfm_heavy_threshold(L): longer lists are pre-reduced, in items of L
    items = []
    for slot, e in enumerate(heavy):
        b, c = occ_ptr[e], counts[e]
        for o in range(b, b + c, L):
            items.append((slot, o, min(o + L, b + c), 0))
    return occ_ptr, occ_rows, heavy.astype(np.int32), np.array(items, np.int32).reshape(-1, 4)


CASES = [  # (B, field sizes, zipf exponent or None, id dtype)
    (1, [3, 4], None, torch.int64),
    (7, [1, 1], None, torch.int32),                 # every row the same two entities
    (2049, [50, 30], None, torch.int64),            # just over one sort tile per column
    (5000, [300, 70000], 1.1, torch.int32),         # 17-bit keys (3 passes), skewed items: heavy lists
    (100000, [138493, 26744], None, torch.int64),   # cfg3
    (40000, [5, 4], None, torch.int64),             # 9 entities: every list is heavy, thousands of work items
    (2048, [31250] * 32, None, torch.int64),        # cfg5: F = 32, T = 10^6 (20-bit keys)
    (300, [2 ** 21 + 5, 9], None, torch.int64),     # 22-bit keys, nearly empty table
    (300000, [138493, 26744], 1.05, torch.int32),   # 293 sort tiles: the per-digit scan kernel (more than 256 tiles)
    (263000, [90, 26744], None, torch.int64),       # 257 tiles, one column with very long lists
    (60000, [300, 40000], 1.3, torch.int64),        # many rows per entity, a small part of a large table: the lowered heavy threshold
]


@pytest.mark.parametrize("B,sizes,zipf,dtype", CASES)
def test_index_equals_stable_sort(B, sizes, zipf, dtype):
    from vae_amd import ops, _lib
    from vae_amd.data import synthetic_triples
    dev = torch.device("cuda:0")
    F, T = len(sizes), int(sum(sizes))
    X, y = synthetic_triples(sizes, B, seed=B % 97, device=dev, zipf=zipf)
    X = X.to(dtype).contiguous()
    spec = ops.Spec(T=T, F=F, d=8, group_hi=tuple(int(v) for v in np.cumsum(sizes)), group_n=tuple(float(s) for s in sizes),
                    likelihood=_lib.LIK_NORMAL)
    inv_occ = torch.ones(T, dtype=torch.float32, device=dev)
    plan = ops.BatchPlan(spec, X, y, inv_occ)
    assert plan.heavy_list == _lib.heavy_list_for(B * F, T) and 8 <= plan.heavy_list <= 64
    # (many rows per entity on a large table: the heavy lists are rebuilt with a quarter of the item length as threshold)
    lowered = plan.B >= 4 * plan.U and 4 * plan.U <= T and T >= 8192
    thr = int(_lib.load().vfm_heavy_threshold(plan.heavy_list)) if lowered else plan.heavy_list
    assert thr == (max(8, plan.heavy_list // 4) if lowered else plan.heavy_list) and getattr(plan, "heavy_threshold", plan.heavy_list) == thr
    ptr_want, rows_want, heavy_want, items_want = _ref(X.cpu().numpy(), T, F, plan.heavy_list, thr)
    assert np.array_equal(plan.occ_ptr.cpu().numpy(), ptr_want)
    assert np.array_equal(plan.occ_rows.cpu().numpy(), rows_want)
    if len(heavy_want):
        assert np.array_equal(plan.heavy[0].cpu().numpy(), heavy_want)
        assert np.array_equal(plan.heavy[1].cpu().numpy(), items_want)
        assert plan.heavy[2].numel() == (len(heavy_want) + len(items_want)) * (4 + 8)
    else:
        assert plan.heavy is None
    uniq = np.unique(X.cpu().numpy())
    assert plan.U == len(uniq) and np.array_equal(plan.touched_ids().cpu().numpy(), uniq)
    # W (vfm_batch_norms) with inv_occ = 1 is the row count of every column
    assert np.allclose(plan.W.cpu().numpy(), B)


def test_empty_shard_and_bad_ids():
    from vae_amd import ops, _lib
    dev = torch.device("cuda:0")
    spec = ops.Spec(T=10, F=2, d=8, group_hi=(5, 10), group_n=(5.0, 5.0), likelihood=_lib.LIK_NORMAL)
    inv_occ = torch.ones(10, dtype=torch.float32, device=dev)
    empty = ops.BatchPlan(spec, torch.zeros(0, 2, dtype=torch.int64, device=dev), torch.zeros(0, device=dev), inv_occ)
    assert empty.occ_ptr.cpu().tolist() == [0] * 11 and empty.occ_rows.numel() == 0 and empty.heavy is None
    x = torch.tensor([[0, 5], [1, 10], [-1, 6]], device=dev)
    with pytest.raises(IndexError):
        ops.BatchPlan(spec, x, torch.zeros(3, device=dev), inv_occ)
    with pytest.raises(IndexError):
        ops.BatchPlan(spec, x, None, None)                       # prediction plan: same check, no index
    p = ops.BatchPlan(spec, x, torch.zeros(3, device=dev), inv_occ, validate=False)    # out-of-range ids index as id 0
    # keys by position: 0, 5, 1, (10 -> 0), (-1 -> 0), 6
    assert p.occ_ptr.cpu().tolist() == [0, 3, 4, 4, 4, 4, 5, 6, 6, 6, 6]
    assert p.occ_rows.cpu().tolist() == [0, 1, 2, 1, 0, 2]


def test_index_is_deterministic_and_plan_cost_is_one_readback():
    """Two builds of the same batch give identical buffers (no atomics between workgroups: nothing depends on
    scheduling), and the build itself enqueues without a host synchronisation until its single readback."""
    from vae_amd import ops, _lib
    from vae_amd.data import synthetic_triples
    dev = torch.device("cuda:0")
    sizes = [1000, 200]
    X, y = synthetic_triples(sizes, 30000, seed=5, device=dev, zipf=1.2)
    spec = ops.Spec(T=1200, F=2, d=16, group_hi=(1000, 1200), group_n=(1000.0, 200.0), likelihood=_lib.LIK_NORMAL)
    inv_occ = torch.ones(1200, dtype=torch.float32, device=dev)
    a, b = ops.BatchPlan(spec, X, y, inv_occ), ops.BatchPlan(spec, X, y, inv_occ)
    assert torch.equal(a.occ_ptr, b.occ_ptr) and torch.equal(a.occ_rows, b.occ_rows)
    assert torch.equal(a.heavy[0], b.heavy[0]) and torch.equal(a.heavy[1], b.heavy[1])


def test_union_rows_of_two_batches():
    """vfm_union_rows: the sorted entities of batch A or batch B (the rows of the look-ahead step), via BatchPlan."""
    from vae_amd import ops
    from vae_amd.data import synthetic_triples
    dev = torch.device("cuda:0")
    sizes = [5000, 300]
    T = sum(sizes)
    spec = ops.Spec(T=T, F=2, d=8, group_hi=(5001, T), group_n=(5000.0, 300.0), likelihood=0, nb_train=10 ** 5)
    Xa, _ = synthetic_triples(sizes, 700, seed=1, device=dev)
    Xb, _ = synthetic_triples(sizes, 1300, seed=2, device=dev)
    pa, pb = ops.BatchPlan(spec, Xa, None, None), ops.BatchPlan(spec, Xb, None, None)
    rows = pa.lookahead_rows(pb)
    want = np.union1d(Xa.cpu().numpy().reshape(-1), Xb.cpu().numpy().reshape(-1))
    assert rows.dtype == torch.int32 and np.array_equal(rows.cpu().numpy(), want)
    assert pa.lookahead_rows(pb) is rows or torch.equal(pa.lookahead_rows(pb), rows)      # cached per pair
    assert np.array_equal(pb.lookahead_rows(pa).cpu().numpy(), want)
    empty = ops.BatchPlan(spec, Xa[:0], None, None)
    assert np.array_equal(pa.lookahead_rows(empty).cpu().numpy(), np.unique(Xa.cpu().numpy().reshape(-1)))
