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
    (3000, [200, 300], 1.2, torch.int64),           # 9-bit keys: ONE radix pass (the first pass is the last)
    (5000, [1024, 1024], None, torch.int32),        # T a multiple of the compaction chunk: occ_ptr[T] has no thread of its own
    (70000, [2, 3], None, torch.int64),             # 3-bit keys, 69 sort tiles in three tile groups, five enormous lists
    (1048576, [138493, 26744], None, torch.int64),  # the bench's largest batch: 1,024 sort tiles in 32 groups, 646 compaction chunks
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
    # W (formed inside the index build for F <= 4, by vfm_batch_norms' kernels otherwise) with inv_occ = 1 is the row count of every column
    assert np.allclose(plan.W.cpu().numpy(), B)
    # the most work items one heavy entity has (vfm_index_t.max_items), from the build's own readback
    if len(heavy_want):
        assert plan.heavy_max_items == int(np.bincount(items_want[:, 0]).max())
    if F == 2:      # the entity in the other column of every occurrence
        xn = X.cpu().numpy().astype(np.int64)
        order = np.argsort(xn.reshape(-1), kind="stable")
        assert np.array_equal(plan.occ_other.cpu().numpy()[: 2 * B], xn.reshape(-1)[order ^ 1].astype(np.int32))
    assert int(plan.status.item()) == 0


def test_normalisers_from_the_index_build_equal_the_stand_alone_kernels():
    """W_f = sum_r 1/occ(x_rf) (vfm-torch.py:305-306) comes out of the index build's own launches (fp64, fixed order) for up
    to four fields: == vfm_batch_norms (fp64 partial sums + fixed-point integer atomics) to 1e-9, == numpy's fp64 sum, and
    bitwise reproducible."""
    import ctypes as C
    from vae_amd import ops, _lib
    from vae_amd.data import synthetic_triples
    dev = torch.device("cuda:0")
    for sizes, B in (([5000, 300], 70001), ([40, 30, 20], 5000), ([10, 10, 10, 10], 333), ([9] * 5, 4000)):
        F, T = len(sizes), int(sum(sizes))
        X, y = synthetic_triples(sizes, B, seed=3, device=dev, zipf=1.1)
        occ = torch.randint(1, 900, (T,), device=dev)
        inv_occ = ops.inv_occ_from_counts(occ)
        spec = ops.Spec(T=T, F=F, d=8, group_hi=tuple(int(v) for v in np.cumsum(sizes)), group_n=tuple(float(v) for v in sizes), likelihood=0)
        a, b = ops.BatchPlan(spec, X, y, inv_occ), ops.BatchPlan(spec, X, y, inv_occ)
        assert torch.equal(a.W, b.W)
        alone = torch.empty(F, dtype=torch.float64, device=dev)
        p = ops._problem(spec, B, B, 64)
        _lib.check(_lib.load().vfm_batch_norms(C.byref(p), _lib.ptr(X), _lib.ptr(inv_occ), _lib.ptr(alone),
                                               _lib.current_stream_ptr(dev)), "vfm_batch_norms")
        want = inv_occ.double().cpu().numpy()[X.cpu().numpy()].sum(0)
        assert np.allclose(a.W.cpu().numpy(), want, rtol=1e-12) and np.allclose(alone.cpu().numpy(), want, rtol=1e-9)


def test_a_corrupted_index_is_clamped_and_reported_not_followed():
    """The kernels that walk an inverted index bound what they read: an occ_ptr that is not monotone, list
    offsets past B*F, row numbers outside [0, B) -- a freed or overwritten plan -- give a FINITE run (no hang, no fault), a
    non-zero vfm_index_t.status, and `BatchPlan.check_status()` raises.  Every backward family: gradients, fused Adam,
    statistics, the pipelined form with its heavy-list pre-reduction, the variants' backward."""
    from vae_amd import ops, _lib
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    from vae_amd.variants import variant_forward, variant_backward
    dev = torch.device("cuda:0")
    sizes, d, B = [300, 40], 32, 4000
    X, y = synthetic_triples(sizes, B, seed=2, device=dev, zipf=1.2)
    torch.manual_seed(1)
    m = VFM(field_sizes=sizes, embedding_size=d, device=dev, rng_seed=5)
    m.set_training_data(X, nb_train=B)
    m._ensure_opt_state()
    m._set_moment_form(True)
    ent, bia, scal = m._views(m._flat)
    mv, vv = m._views(m._adam_m), m._views(m._adam_v)
    g = torch.Generator(device="cpu").manual_seed(7)

    def broken(kind):
        plan = m.plan(X, y)
        assert plan.heavy is not None and int(plan.status.item()) == 0
        if kind == "ptr":          # not monotone, negative, far past the end
            bad = torch.randint(-5, 3 * B * 2, (m.T + 1,), generator=g).to(torch.int32)
            plan.occ_ptr.copy_(bad.to(dev))
        elif kind == "rows":       # row numbers outside the batch
            plan.occ_rows.copy_(torch.randint(-B, 5 * B, (2 * B,), generator=g).to(torch.int32).to(dev))
        elif kind == "items":      # work items naming lists that do not exist
            hid, items, acc = plan.heavy
            items.copy_(torch.randint(-7, 4 * B, tuple(items.shape), generator=g).to(torch.int32).to(dev))
        elif kind == "other":      # entities outside the table in the other-column array (the pipelined walk)
            plan.occ_other.copy_(torch.randint(-9, 50 * m.T, (2 * B,), generator=g).to(torch.int32).to(dev))
        return plan

    for kind in ("ptr", "rows", "items"):
        plan = broken(kind)
        st = ops.elbo_forward(plan, ent, bia, scal, m.inv_occ, seed=5, step=1)
        ops.elbo_finalize(st, scal)
        ops.elbo_backward(plan, st, ent.clone(), bia.clone(), scal.clone(), m.inv_occ, torch.ones(1, device=dev))
        ops.elbo_backward_adam(plan, st, ent.clone(), bia.clone(), scal.clone(), m.inv_occ, mv, vv, 0.01, 1,
                               loss_out=torch.zeros(3, device=dev), scaled_moments=True)
        torch.cuda.synchronize()            # (returns: nothing hung)
        assert int(plan.status.item()) > 0, kind
        with pytest.raises(_lib.VfmLibraryError):
            plan.check_status()
    for kind in ("other", "rows"):          # the pipelined step's backward: gathers the other entity's record per occurrence
        plan = broken(kind)
        zrec = torch.zeros(m.T, ops.record_len(d), device=dev)
        ops.sample_records(plan, ent, bia, m.inv_occ, zrec, 5, 1)
        st = ops.elbo_forward_records(plan, zrec, scal, 5, 1, torch.empty(B, device=dev), torch.empty(B, device=dev), m._partials)
        ops.elbo_backward_adam_pipe(plan, st, zrec, None, None, 2, ent.clone(), bia.clone(), scal.clone(), m.inv_occ, mv, vv, 0.01, 1,
                                    torch.zeros(3, device=dev))
        torch.cuda.synchronize()
        assert int(plan.status.item()) > 0, kind
    plan = broken("ptr")                    # the sibling scripts' objective (closed form), lane-group kernels
    spec = plan.spec
    fw = variant_forward(plan, "closed_form", ent, bia, scal, m.inv_occ, seed=5, step=1)
    variant_backward(plan, fw, ent, bia, scal, m.inv_occ, torch.ones(1, device=dev))
    torch.cuda.synchronize()
    assert int(plan.status.item()) > 0


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
