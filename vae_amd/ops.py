"""Host-side operators over the C ABI (include/vfm_hip.h).  Device memory, streams and
autograd come from PyTorch-ROCm; all arithmetic of the hot path runs in the HIP kernels.
The per-step entry points are called through `torch.ops.vfm_hip.*` (the TORCH_LIBRARY shim
`csrc/vfm_torch_ops.cpp`, which forwards raw pointers + the current HIP stream to the C ABI); the
once-per-dataset helpers and the tests of the raw ABI use ctypes on the same library.

`BatchPlan`  -- everything about one batch that does not depend on the parameters: the ids,
               targets, the batch normalisers W (vfm-torch.py:305-306) and the inverted index
               (entity -> rows) the backward kernel walks.  The reference's loader does not
               shuffle (vfm-torch.py:121-122), so `fit()` builds each plan once and reuses it
               every epoch.
`elbo_forward / elbo_backward` -- thin launch wrappers.
`ElboFunction` -- torch.autograd.Function: the drop-in for `CF.forward` + loss line
               (vfm-torch.py:189-324,359) + autograd (:368-369).
"""
from __future__ import annotations

import contextlib
import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import Problem, check, ptr, current_stream_ptr

FLAG_NO_PRIOR_TERMS = 1
FLAG_EPS_ZERO = 2
FLAG_SPARSE_ADAM = 4
FLAG_LINK_SOFTPLUS = 16
FLAG_SCALED_MOMENTS = 32
FLAG_ROWS_TOUCHED = 128
FLAG_SHARE_GPU = 2048
MOMENT_PERIOD = 128
MAX_SAMPLES = 64
_I63 = (1 << 63) - 1


@dataclass
class Spec:
    """Static description of the model / data set (the globals the reference's CF reads:
    N, M, nb_occ, EMBEDDING_SIZE, vfm-torch.py:18,87-89)."""
    T: int
    F: int
    d: int
    group_hi: tuple      # exclusive id upper bound of each of the F groups
    group_n: tuple       # n_g multipliers (N, M in the reference)
    likelihood: int      # _lib.LIK_NORMAL / LIK_BERNOULLI
    nb_train: int = 1
    n_samples: int = 1   # N_VARIATIONAL_SAMPLES (vfm-torch.py:19)
    link: str = "abs"    # LINK (vfm-torch.py:125-126): "abs" (the one in effect) or "softplus"

    def __post_init__(self):
        if self.link not in ("abs", "softplus"):
            raise ValueError("link must be 'abs' or 'softplus'")
        if not 1 <= int(self.n_samples) <= MAX_SAMPLES:
            raise ValueError(f"n_samples must be in [1,{MAX_SAMPLES}]")

    @property
    def link_flag(self) -> int:
        return FLAG_LINK_SOFTPLUS if self.link == "softplus" else 0


def _problem(spec: Spec, B: int, B_global: int, id_bits: int, seed: int = 0, step: int = 0,
             flags: int = 0, wrec=None, dev_step=None) -> Problem:
    p = Problem()
    if wrec is not None:
        p.wrec = wrec.data_ptr()
    if dev_step is not None:
        p.dev_step = dev_step.data_ptr()
    p.B, p.B_global, p.T, p.nb_train = B, B_global, spec.T, spec.nb_train
    p.F, p.d, p.likelihood, p.id_bits = spec.F, spec.d, spec.likelihood, id_bits
    p.n_samples, p.flags = int(spec.n_samples), flags | spec.link_flag
    for g in range(spec.F):
        p.group_hi[g] = int(spec.group_hi[g])
        p.group_n[g] = float(spec.group_n[g])
    p.seed, p.step = seed & (2 ** 64 - 1), step & (2 ** 62 - 1)
    return p


def _need_cuda(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise _lib.VfmLibraryError(
            f"{name} is on {t.device}: vae_amd runs on an MI355X only (no CPU fallback)")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


def inv_occ_from_counts(nb_occ: torch.Tensor) -> torch.Tensor:
    """1/nb_occ as fp32 (vfm-torch.py:89,298-306)."""
    _need_cuda(nb_occ, "nb_occ")
    assert nb_occ.dtype == torch.int64
    out = torch.empty(nb_occ.numel(), dtype=torch.float32, device=nb_occ.device)
    lib = _lib.load()
    check(lib.vfm_inv_occ_f32(ptr(nb_occ), ptr(out), nb_occ.numel(), current_stream_ptr(nb_occ.device)),
          "vfm_inv_occ_f32")
    return out


_WS = {}
# The scratch buffer below is shared by every plan built on a (device, stream): safe because a build's launches are
# stream-ordered -- as long as they are enqueued as ONE uninterrupted sequence.  ctypes releases the GIL inside the C call,
# so two host threads building plans on the same stream could interleave their launches (and their pinned readback
# slices): every enqueue that uses the shared scratch / the pinned chunk runs under this lock.
_ENQUEUE_LOCK = __import__("threading").RLock()


def _index_workspace(n_int32: int, dev, stream: int) -> torch.Tensor:
    """One scratch buffer per (device, stream) for vfm_build_index (stream-ordered reuse; grown on demand).  `stream`: the
    raw handle of the stream the build is enqueued on."""
    key = (dev.index, stream)
    t = _WS.get(key)
    if t is None or t.numel() < n_int32:
        with torch.cuda.stream(_stream_obj(dev, stream)):       # (from the pool of the stream that writes it)
            t = _WS[key] = torch.empty(n_int32, dtype=torch.int32, device=dev)
    return t


_ARENAS = {}


def _side_arena(dev, stream, numel: int) -> torch.Tensor:
    """An uninitialised int32 buffer from the allocator pool of `stream` (a side stream of plan builds).  Eight at a time:
    entering and leaving a stream context costs the host ~25 us, once per plan it was a third of the streamed loop's
    enqueue time; the buffers of one (stream, size) wait here until a build takes them (sizes repeat: B, F, T fix them)."""
    key = (dev.index, stream.cuda_stream, numel)
    stash = _ARENAS.get(key)
    if not stash:
        if len(_ARENAS) > 8:             # (batches of many different sizes: drop the other stashes)
            _ARENAS.clear()
        with torch.cuda.stream(stream):
            stash = _ARENAS[key] = [torch.empty(numel, dtype=torch.int32, device=dev) for _ in range(8)]
    return stash.pop()


_STREAMS = {}


def _stream_obj(dev, raw: int):
    """The torch Stream object of a raw stream handle (cached: torch.cuda.current_stream() costs 4.5 us a call)."""
    s = _STREAMS.get((dev.index, raw))
    if s is None:
        s = torch.cuda.current_stream(dev)
        if s.cuda_stream != raw:                     # (a stream that is not current: wrap the handle)
            s = torch.cuda.ExternalStream(raw, device=dev)
        if len(_STREAMS) > 64:
            _STREAMS.clear()
        _STREAMS[(dev.index, raw)] = s
    return s


_SIZES = {}


def _index_sizes(B: int, F: int, T: int):
    """(heavy-list length, workspace bytes) of a build: two library calls, cached per shape."""
    hit = _SIZES.get((B, F, T))
    if hit is None:
        lib = _lib.load()
        hit = (_lib.heavy_list_for(B * F, T), int(lib.vfm_index_workspace_bytes(B, F, T)))
        if len(_SIZES) > 256:
            _SIZES.clear()
        _SIZES[(B, F, T)] = hit
    return hit


_PINNED = {"chunk": None, "used": 0}


def _pinned_ints(n: int) -> torch.Tensor:
    """n int32 of pinned host memory for a small asynchronous readback.  Slices of 4 KiB chunks: one pinned allocation
    (a driver call of ~0.1 ms) serves a thousand plans instead of one each; a chunk lives as long as a slice of it."""
    if _PINNED["chunk"] is None or _PINNED["used"] + n > _PINNED["chunk"].numel():
        _PINNED["chunk"] = torch.empty(max(1024, n), dtype=torch.int32, pin_memory=True)
        _PINNED["used"] = 0
    t = _PINNED["chunk"][_PINNED["used"]:_PINNED["used"] + n]
    _PINNED["used"] += n
    return t


def _carve(total_dev, sizes):
    """Views of the given int32 sizes into one allocation (each 16-byte aligned)."""
    out, off = [], 0
    for n in sizes:
        out.append(total_dev[off:off + n])
        off += (n + 3) & ~3
    return out


class BatchPlan:
    """Parameter-independent state of one batch (or one rank's row shard of it)."""

    def __init__(self, spec: Spec, x: torch.Tensor, y: Optional[torch.Tensor],
                 inv_occ: Optional[torch.Tensor], B_global: Optional[int] = None,
                 build_index: bool = True, validate: bool = True, process_group=None, defer_readback: bool = False,
                 stream=None):
        """defer_readback: enqueue the index build and return; its 32-byte readback (and the IndexError an id out
        of range raises) happens when the plan is first used -- a loop that builds many plans then never waits.
        stream: a torch Stream to enqueue the build on instead of the current one (PlanStream)."""
        _need_cuda(x, "x")
        if x.dim() != 2 or x.shape[1] != spec.F:
            raise ValueError(f"x must be [B,{spec.F}], got {tuple(x.shape)}")
        if x.dtype not in (torch.int64, torch.int32):
            raise TypeError("x must be int64 or int32")
        self.spec = spec
        self.x = x
        self.B = x.shape[0]
        self.B_global = self.B if B_global is None else int(B_global)
        self.id_bits = 64 if x.dtype == torch.int64 else 32
        self.y = None
        if y is not None:
            _need_cuda(y, "y")
            self.y = y.to(torch.float32).contiguous()
            if self.y.shape != (self.B,):
                raise ValueError("y must be [B]")
        self.W = None
        self.occ_ptr = None
        self._occ_rows = self._status = self._occ_other = self._lay = None
        self._ready = None               # (event, stream) of a build enqueued on another stream (PlanStream): see use_on_current
        self._arena = None
        self._heavy = self._pend = None
        self._U = 0
        self._touched = None
        self._checked = False
        dev = x.device
        lib = _lib.load()
        want_w = inv_occ is not None and y is not None
        if want_w and not (build_index and y is not None):
            self.W = torch.empty(spec.F, dtype=torch.float64, device=dev)
        if build_index and y is not None:
            # the batch normalisers W (vfm-torch.py:305-306) come out of the index build's own launches (the ids are read once)
            self.build_index(validate=validate, defer=defer_readback, inv_occ=inv_occ if want_w else None, stream=stream)
        elif want_w:
            p = _problem(spec, self.B, self.B_global, self.id_bits)
            check(lib.vfm_batch_norms(C.byref(p), ptr(x), ptr(inv_occ), ptr(self.W),
                                      current_stream_ptr(dev)), "vfm_batch_norms")
        if want_w and process_group is not None:
            from .dist import sum_normalisers
            sum_normalisers(self.W, process_group)
        if build_index and y is not None:
            pass                                                          # (the build's one readback carries the id check too)
        elif validate and self.B > 0:
            # nn.Embedding would raise IndexError (vfm-torch.py:207): one readback
            if bool(((x < 0) | (x >= spec.T)).any()):
                raise IndexError(f"entity id out of range [0,{spec.T}): min {int(x.min())}, max {int(x.max())}")
            self._checked = True

    def build_index(self, validate: bool = False, defer: bool = False, inv_occ=None, stream=None):
        """Inverted index entity -> batch rows: `vfm_build_index` (a stable radix sort in HIP, csrc/vfm_index.hip;
        what the reference gets from torch.unique, vfm-torch.py:190-192).  Launch-only except for ONE small
        readback at the end: (ids out of range, number of heavy lists, number of their work items, entities in the
        batch, most work items of one entity).  inv_occ: also fill self.W in the same launches.  stream: a torch Stream
        other than the current one to enqueue on (the buffers are then marked as in use there)."""
        spec = self.spec
        n = self.B * spec.F
        L, nbytes = _index_sizes(self.B, spec.F, spec.T)
        self.heavy_list = L
        if nbytes < 0:
            raise ValueError("batch too large for a 32-bit index (B * F must stay below 2^31)")
        with _ENQUEUE_LOCK:
            self._build_index_locked(_lib.load(), n, L, nbytes, validate, inv_occ, stream)
        if not defer:
            self._finish()

    def _build_index_locked(self, lib, n, L, nbytes, validate, inv_occ=None, stream=None):
        spec, dev = self.spec, self.x.device
        raw = stream.cuda_stream if stream is not None else _lib.raw_stream(dev)
        ws = _index_workspace(nbytes // 4 + 4, dev, raw)  # scratch of the build only: shared by all plans of a (device, stream)
        cap_h, cap_i = n // L + 1, 2 * n // L + 2
        # ONE device allocation per plan, carved into the index's arrays (seven allocations per plan made the first
        # build of a training set's plans twice as long as the kernels themselves); the views are made when first asked for
        # (the last piece: W [F] fp64, when the build makes the normalisers -- one allocation, one stream to record)
        sizes = (spec.T + 1, max(n, 1), cap_h, 4 * cap_i, 8, max(min(n, spec.T), 1), max(n, 1) if spec.F == 2 else 0,
                 2 * spec.F if inv_occ is not None else 0)
        offs, o = [], 0
        for v in sizes:
            offs.append(o)
            o += (v + 3) & ~3
        # A build on a side stream allocates from THAT stream's pool.  (From the current stream's pool a block may be one a
        # dropped plan has just returned -- free for the current stream, whose queued steps still read it, but the side stream
        # would start overwriting it at once: a build enqueued with fork=False is not ordered behind those steps.  Found by
        # tests/test_gpu_full_size.py::test_cfg3_plans_built_inside_the_loop_follow_the_resident_plans_run: losses drifting
        # from the 10th step on, equal with a device sync before every step.)  The consumer's stream is recorded when the plan
        # is first used there (use_on_current), so the block is not handed out again under ITS queued work either.
        arena = self._arena = torch.empty(o, dtype=torch.int32, device=dev) if stream is None else _side_arena(dev, stream, o)
        self._lay = (offs, sizes, cap_i)
        if inv_occ is not None:
            self.W = arena[offs[7]: offs[7] + 2 * spec.F].view(torch.float64)
        base = arena.data_ptr()
        P = [C.c_void_p(base + 4 * q) for q in offs]
        host = _pinned_ints(8)
        check(lib.vfm_build_index(self.B, spec.F, spec.T, self.id_bits, ptr(self.x), ptr(ws), P[0], P[1], L, P[2], cap_h, P[3],
                                  cap_i, P[5], P[6] if spec.F == 2 else None, ptr(inv_occ),
                                  ptr(self.W) if inv_occ is not None else None, P[4], C.c_void_p(host.data_ptr()),
                                  C.c_void_p(raw)), "vfm_build_index")
        self.occ_ptr = arena[: spec.T + 1]
        # the plan build's one readback (32 bytes into pinned host memory, enqueued by the library) is complete when `done` is
        done = torch.cuda.Event()
        done.record(stream if stream is not None else _stream_obj(dev, raw))
        self._pend = (host, done, validate)

    def _view(self, k):
        offs, sizes, _ = self._lay
        return self._arena[offs[k]: offs[k] + sizes[k]]

    @property
    def occ_rows(self):
        if self._occ_rows is None and self._arena is not None:
            self._occ_rows = self._view(1)[: self.B * self.spec.F]
        return self._occ_rows

    @property
    def status(self):
        """counts[5]: the word the kernels that walk this index report clamped entries in (vfm_index_t.status); zeroed by the build."""
        if self._status is None and self._arena is not None:
            self._status = self._view(4)[5:6]
        return self._status

    @property
    def occ_other(self):
        """Two fields: the entity in the other column of every occurrence (the pipelined step gathers its sample)."""
        if self._occ_other is None and self._arena is not None and self.spec.F == 2:
            self._occ_other = self._view(6)
        return self._occ_other

    def _finish(self):
        """Complete a deferred index build: read (bad ids, heavy lists, work items, entities) and size the lists."""
        if self._pend is None:
            return
        self.use_on_current()            # (a build enqueued on a side stream: whatever is launched from here on comes after it)
        host, done, validate = self._pend
        self._pend = None
        done.synchronize()
        n_bad, n_heavy, n_items, n_touched, max_items = (int(v) for v in host.tolist()[:5])
        spec, dev = self.spec, self.x.device
        hid, items, touched = self._view(2), self._view(3).view(self._lay[2], 4), self._view(5)
        self._U = n_touched                              # entities in the batch
        self._touched = touched[:n_touched]
        # many rows per entity on a large table (rows in the data files' order: a few hundred users' consecutive ratings):
        # the heavy lists once more with a lower threshold -- lists of 17..64 rows become work items of the pre-reduction
        # (eight occurrences in flight) instead of being walked by one lane group of the main kernel, two at a time
        # (vfm_rebuild_heavy in include/vfm_hip.h: backward 103 -> 94 us on that shape; 3 % slower where users are spread
        # over the table, so only here)
        thr = int(_lib.load().vfm_heavy_threshold(self.heavy_list))
        # (... and where the batch covers the table -- 10^6 rows at the ML-20M shape -- the main kernel has 165 K rows to
        #  hide those walks behind: lowering the threshold there cost 10 %)
        if (n_touched > 0 and self.B >= 4 * n_touched and 4 * n_touched <= spec.T and spec.T >= 8192
                and thr < self.heavy_list and not n_bad):
            n, L = self.B * spec.F, self.heavy_list
            cap_h, cap_i = n // thr + 1, n // L + n // thr + 2
            arena = torch.empty(((cap_h + 3) & ~3) + 4 * cap_i + 8, dtype=torch.int32, device=dev)
            hid, items, counts2 = arena[:cap_h], arena[(cap_h + 3) & ~3:((cap_h + 3) & ~3) + 4 * cap_i].view(cap_i, 4), arena[-8:]
            lib = _lib.load()
            with _ENQUEUE_LOCK:
                ws = _index_workspace(int(lib.vfm_index_workspace_bytes(self.B, spec.F, spec.T)) // 4 + 4, dev, _lib.raw_stream(dev))
                check(lib.vfm_rebuild_heavy(spec.T, ptr(self.occ_ptr), ptr(ws), L, thr, ptr(hid), cap_h, ptr(items), cap_i,
                                            ptr(counts2), current_stream_ptr(dev)), "vfm_rebuild_heavy")
                _, n_heavy, n_items, _, max_items = (int(v) for v in counts2.tolist()[:5])      # (a second small readback, these plans only)
            self.heavy_threshold = thr
        if validate and n_bad and not self._checked:
            # nn.Embedding would raise IndexError (vfm-torch.py:207)
            raise IndexError(f"entity id out of range [0,{spec.T}): min {int(self.x.min())}, max {int(self.x.max())}")
        self._checked = True
        # long lists (skewed data, small tables): cut in work items of at most `heavy_list` occurrences, pre-reduced
        # by their own lane groups in every backward call (vfm_index_t in include/vfm_hip.h)
        self._heavy = None
        # the most work items one entity has (from the build's own readback): at most VFM_HEAVY_DIRECT and the backward
        # leaves its k_heavy_sum launch out (vfm_index_t.max_items); with it the work-item length and the heavy threshold
        # the lists were built with (vfm_index_t.heavy_list / heavy_threshold: the small-table step re-derives the items)
        self.heavy_max_items = max_items if n_heavy > 0 else 0
        self._heavy_meta = torch.tensor([self.heavy_max_items, self.heavy_list, getattr(self, "heavy_threshold", self.heavy_list)],
                                        dtype=torch.int32)
        if n_heavy > 0:
            rec = 4 + (spec.d + 3) // 4 * 4
            self._heavy = (hid[:n_heavy], items[:n_items],       # scratch: entity records + work-item records
                           torch.zeros(spec.n_samples * (n_heavy + n_items) * rec, dtype=torch.float32, device=dev))

    @property
    def U(self) -> int:
        """Number of distinct entities in the batch."""
        if self.occ_ptr is None:
            self.build_index()
        self._finish()
        return self._U

    @property
    def heavy(self):
        self._finish()
        return self._heavy

    def touched_ids(self) -> torch.Tensor:
        """Sorted ids of the entities this batch contains (int32; made by the index build)."""
        if self.occ_ptr is None:
            self.build_index()
        self._finish()
        return self._touched

    def lookahead_rows(self, next_plan: "BatchPlan") -> torch.Tensor:
        """Sorted ids of the entities of this batch or of `next_plan`'s (int32): the rows the look-ahead step visits.
        Made once per pair of plans (`vfm_union_rows`; one 4-byte readback) and kept with this plan -- like the plan
        itself it depends on the ids only, so a training loop that revisits its batches builds it once."""
        pairs = self.__dict__.setdefault("_pairs", {})
        hit = pairs.get(id(next_plan))
        if hit is not None and hit[0]() is next_plan:
            if hit[2] is not None:               # deferred count: first use
                host, done, src = hit[2]
                done.synchronize()               # (host-side: the list is complete -- whatever is enqueued from here on, on
                                                 #  any stream, comes after it)
                cur = _stream_obj(self.x.device, _lib.raw_stream(self.x.device))
                if src.cuda_stream != cur.cuda_stream:
                    hit[1].record_stream(cur)    # (allocated on the build's stream, read on this one)
                pairs[id(next_plan)] = (hit[0], hit[1][:int(host.item())], None)
            return pairs[id(next_plan)][1]
        self.prepare_lookahead(next_plan)
        return self.lookahead_rows(next_plan)

    def prepare_lookahead(self, next_plan: "BatchPlan", stream=None) -> None:
        """Enqueue the build of `lookahead_rows(next_plan)` without waiting for its count (stream: a torch Stream other than
        the current one to enqueue on)."""
        import weakref
        pairs = self.__dict__.setdefault("_pairs", {})
        hit = pairs.get(id(next_plan))
        if hit is not None and hit[0]() is next_plan:
            return
        dev, T = self.x.device, self.spec.T
        for q in (self, next_plan):
            if q.occ_ptr is None:
                q.build_index(defer=True, stream=stream)
        lib = _lib.load()
        raw = stream.cuda_stream if stream is not None else _lib.raw_stream(dev)
        with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):     # (from the build stream's pool: see _build_index_locked)
            buf = torch.empty((min(T, (self.B + next_plan.B) * self.spec.F) or 1) + 4, dtype=torch.int32, device=dev)
        rows, count = buf[:-4], buf[-4:]
        with _ENQUEUE_LOCK:
            ws = _index_workspace(int(lib.vfm_union_workspace_bytes(T)) // 4 + 4, dev, raw)
            host = _pinned_ints(1)
            check(lib.vfm_union_rows(T, ptr(self.occ_ptr), ptr(next_plan.occ_ptr), ptr(ws), ptr(rows), ptr(count),
                                     C.c_void_p(host.data_ptr()), C.c_void_p(raw)), "vfm_union_rows")
            done = torch.cuda.Event()
            src = stream if stream is not None else _stream_obj(dev, raw)
            done.record(src)
        pairs[id(next_plan)] = (weakref.ref(next_plan), rows, (host, done, src))
        if len(pairs) > 4:
            pairs.pop(next(iter(pairs)))

    def use_on_current(self):
        """A plan built on a side stream (PlanStream) is about to be used on the current one: that stream waits for the
        build (an event, no host sync) and the caching allocator is told the buffers are in use there."""
        if self._ready is None:
            return
        ev, src = self._ready
        dev = self.x.device
        raw = _lib.raw_stream(dev)
        if raw != src.cuda_stream:
            cur = _stream_obj(dev, raw)
            cur.wait_event(ev)
            if self._arena is not None:
                self._arena.record_stream(cur)      # (allocated on the build's stream, used on this one from here on; W is a piece of it)
        self._ready = None

    def check_status(self):
        """Raise if a kernel that walked this plan's index had to clamp an entry (a row number, a list offset or an entity
        outside its range: a corrupted or stale index -- the kernels never follow such an entry, they count it in
        `vfm_index_t.status`).  One 4-byte readback: `fit()` calls it once per epoch."""
        if self.occ_ptr is not None and int(self.status.item()) != 0:
            raise _lib.VfmLibraryError("corrupted inverted index: a backward kernel clamped entries of this plan's index "
                                       "(vfm_index_t.status != 0); the step's results are not to be trusted")

    def index_tensors(self, with_touched: bool = False):
        """What the backward-family ops take as `index` (with_touched: + the batch's entities as a list, for the
        touched-rows step of the lazy exact Adam mode)."""
        if self.occ_ptr is None:
            self.build_index()
        self._finish()
        base = [self.occ_ptr, self.occ_rows, self.status]
        if self.heavy is not None:
            base = base + list(self.heavy)
        if with_touched:
            base = base + [self._touched]
        return base + [self._heavy_meta]       # (host tensor: vfm_index_t.max_items, heavy_list, heavy_threshold)


def _index_struct(plan: BatchPlan) -> "_lib.Index":
    """`vfm_index_t` of a plan (the ctypes callers; the torch.ops shim builds its own from `index_tensors()`)."""
    if plan.occ_ptr is None:
        plan.build_index()
    plan._finish()
    ix = _lib.Index()
    ix.occ_ptr, ix.occ_rows, ix.status = plan.occ_ptr.data_ptr(), plan.occ_rows.data_ptr(), plan.status.data_ptr()
    ix.heavy_list, ix.heavy_threshold = plan.heavy_list, getattr(plan, "heavy_threshold", plan.heavy_list)
    if plan.heavy is not None:
        hid, items, acc = plan.heavy
        ix.heavy_ids, ix.heavy_items, ix.heavy_acc = hid.data_ptr(), items.data_ptr(), acc.data_ptr()
        ix.n_heavy, ix.n_items = hid.numel(), items.numel() // 4
        ix.max_items = plan.heavy_max_items
    return ix


class PlanStream:
    """A side stream for plan builds: the index of batch t+1 (and the look-ahead row list of the pair) is built WHILE step t
    runs -- what a caller that streams or shuffles its batches needs, since nothing of a plan can then be reused (the
    reference itself pays torch.unique x3 inside every step, vfm-torch.py:190-192).  Fork / join by events only: the side
    stream waits for the caller's stream at the point of the call (the ids may have just been produced there), and the plan
    carries the event its first user waits for (`BatchPlan.use_on_current`).  The builds are small latency-bound launches
    (~40 us of GPU time per batch at the ML-20M shape) that fit beside the bandwidth-bound step kernels."""

    def __init__(self, device, n_streams: Optional[int] = None):
        import os
        self.device = torch.device(device)
        n = int(os.environ.get("VFM_PLAN_STREAMS", "2")) if n_streams is None else int(n_streams)
        prio = int(os.environ.get("VFM_PLAN_PRIORITY", "-1"))
        # builds alternate between the streams: a build is a chain of seven dependent small launches, each waiting for
        # free workgroup slots beside the step's kernels -- on ONE stream the chain of batch t+3 also waits for the chain
        # of batch t+2 to end, and the loop runs at the pace of that one queue
        self.streams = [torch.cuda.Stream(self.device, priority=prio) for _ in range(max(1, n))]
        self._turn = 0

    @property
    def stream(self):
        return self.streams[self._turn % len(self.streams)]

    def wait_current(self):
        """Every side stream waits (an event, no host sync) for what is enqueued so far on the current stream: call once
        after producing long-lived inputs (a training set moved to the device, sorted) and before a run of `fork=False`
        builds that read them."""
        ev = torch.cuda.Event()
        ev.record(_stream_obj(self.device, _lib.raw_stream(self.device)))
        for st in self.streams:
            st.wait_event(ev)

    def build(self, make, pair_with: Optional[BatchPlan] = None, fork: bool = True) -> BatchPlan:
        """`make(stream)` -> BatchPlan whose build is enqueued on `stream` (BatchPlan(..., defer_readback=True, stream=stream));
        `pair_with`: also enqueue the look-ahead row list of (pair_with, new plan) there.  fork: the side stream first waits
        for everything enqueued so far on the current stream -- needed when the ids / targets were produced there just now;
        with long-lived inputs (a resident training set) fork=False lets the build start at once instead of behind the steps
        already in flight (with the fork the caller's NEXT step found the build only just started: 0.25 instead of 0.21 ms per
        step at the ML-20M shape)."""
        self._turn += 1
        st = self.stream
        if fork:
            main = _stream_obj(self.device, _lib.raw_stream(self.device))
            ev0 = torch.cuda.Event()
            ev0.record(main)
            st.wait_event(ev0)
        plan = make(st)
        if pair_with is not None:
            if pair_with._ready is not None and pair_with._ready[1] is not st:      # (built on the other side stream)
                st.wait_event(pair_with._ready[0])
            pair_with.prepare_lookahead(plan, stream=st)
        ev = torch.cuda.Event()
        ev.record(st)
        plan._ready = (ev, st)
        return plan


@dataclass
class FwdState:
    pred: torch.Tensor          # [B] unscaled prediction / logit ([S,B] when n_samples = S > 1)
    partials: torch.Tensor      # [8] fp64
    sumz: Optional[torch.Tensor]
    grow: Optional[torch.Tensor]
    problem: Problem
    eps: Optional[tuple]


def elbo_forward(plan: BatchPlan, entity_params, bias_params, scalars, inv_occ, *, eps=None,
                 seed=0, step=0, train=True, flags=0, out_pred=None, out_sumz=None,
                 out_grow=None, out_partials=None, wrec=None, dev_step=None) -> FwdState:
    """Launch vfm_elbo_fwd_f32 on the current stream.  `eps` = (eps_entity[T,d], eps_bias[T],
    eps_global[1]) tables indexed by entity id ([S,T,d], [S,T], [S] with S = spec.n_samples > 1), or
    None for the in-kernel Philox stream."""
    spec = plan.spec
    dev = plan.x.device
    for t, n in ((entity_params, "entity_params"), (bias_params, "bias_params"), (scalars, "scalars")):
        _need_cuda(t, n)
    if entity_params.shape != (spec.T, 2 * spec.d) or bias_params.shape != (spec.T, 2):
        raise ValueError("table shapes do not match the spec")
    have_y = plan.y is not None
    train = train and have_y
    B, S = plan.B, int(spec.n_samples)
    pshape = (B,) if S == 1 else (S, B)
    pred = out_pred if out_pred is not None else torch.empty(pshape, dtype=torch.float32, device=dev)
    partials = out_partials if out_partials is not None else torch.empty(
        _lib.PARTIALS_LEN, dtype=torch.float64, device=dev)
    sumz = grow = None
    if train:
        sumz = out_sumz if out_sumz is not None else torch.empty(S * B, spec.d, dtype=torch.float32, device=dev)
        grow = out_grow if out_grow is not None else torch.empty(B, dtype=torch.float32, device=dev)
    p = _problem(spec, B, plan.B_global, plan.id_bits, seed, step, flags)
    e = eps if eps is not None else (None, None, None)
    _lib.ops().elbo_fwd(plan.x, plan.y, entity_params, bias_params, inv_occ if have_y else None, scalars,
                        plan.W if have_y else None, e[0], e[1], e[2], pred, partials, sumz, grow,
                        list(spec.group_hi), list(spec.group_n), spec.nb_train, plan.B_global,
                        spec.likelihood, p.flags, p.seed & _I63, p.step, S,
                        wrec if (train and eps is None) else None, dev_step)
    return FwdState(pred, partials, sumz, grow, p, eps)


def elbo_finalize(st: FwdState, scalars, out=None) -> torch.Tensor:
    """loss[3] = (loss, likelihood term, KL term) from (rank-summed) partials."""
    loss = out if out is not None else torch.empty(3, dtype=torch.float32, device=scalars.device)
    p = st.problem
    _lib.ops().elbo_finalize(st.partials, scalars, loss, p.nb_train, p.B_global, p.flags, p.n_samples)
    return loss


def elbo_backward(plan: BatchPlan, st: FwdState, entity_params, bias_params, scalars, inv_occ,
                  grad_out, g_entity=None, g_bias=None, g_scalars=None):
    """Launch vfm_elbo_bwd_f32: dense gradients of both tables + the three scalars."""
    dev = plan.x.device
    if plan.occ_ptr is None:
        plan.build_index()
    if g_entity is None:
        g_entity = torch.empty_like(entity_params)
    if g_bias is None:
        g_bias = torch.empty_like(bias_params)
    if g_scalars is None:
        g_scalars = torch.empty(3, dtype=torch.float32, device=dev)
    e = st.eps if st.eps is not None else (None, None, None)
    p, spec = st.problem, plan.spec
    _lib.ops().elbo_bwd(plan.index_tensors(), entity_params, bias_params, inv_occ, scalars, plan.W,
                        e[0], e[1], e[2], st.sumz, st.grow, st.partials, grad_out, g_entity, g_bias, g_scalars,
                        spec.F, list(spec.group_hi), list(spec.group_n), p.nb_train, p.B_global, p.likelihood,
                        p.flags, p.seed & _I63, p.step, p.n_samples)
    return g_entity, g_bias, g_scalars


def elbo_backward_adam(plan: BatchPlan, st: FwdState, entity_params, bias_params, scalars, inv_occ,
                       m_views, v_views, lr, step, beta1=0.9, beta2=0.999, eps_adam=1e-8, loss_out=None,
                       sparse=False, scaled_moments=False, rows=None, wrec=None, dev_step=None):
    """Launch vfm_elbo_bwd_adam_f32: backward + dense Adam in one kernel (single rank).
    m_views / v_views = (entity, bias, scalars) moment tensors.  `loss_out` (3 floats): also do the
    work of elbo_finalize in the same launch.  `scaled_moments`: the buffers are in the scaled form of
    VFM_FLAG_SCALED_MOMENTS (see include/vfm_hip.h; `moments_rescale` converts).  `rows`: None = all
    table rows; "touched" = only the rows of the batch (+ scalars, loss): the row-list step of the lazy exact Adam."""
    rows_flag = {None: 0, "touched": FLAG_ROWS_TOUCHED}[rows]
    if plan.occ_ptr is None:
        plan.build_index()
    e = st.eps if st.eps is not None else (None, None, None)
    p, spec = st.problem, plan.spec
    _lib.ops().elbo_bwd_adam(plan.index_tensors(with_touched=rows == "touched"), entity_params, bias_params, scalars,
                             inv_occ, plan.W,
                             e[0], e[1], e[2], st.sumz, st.grow, st.partials, m_views[0], v_views[0],
                             m_views[1], v_views[1], m_views[2], v_views[2], spec.F, list(spec.group_hi),
                             list(spec.group_n), p.nb_train, p.B_global, p.likelihood,
                             p.flags | (FLAG_SPARSE_ADAM if sparse else 0) |
                             (FLAG_SCALED_MOMENTS if scaled_moments else 0) | rows_flag,
                             p.seed & _I63, p.step, lr, beta1, beta2, eps_adam, int(step), loss_out, p.n_samples,
                             wrec, dev_step)


FLAG_ZREC = 1024


def record_len(d: int) -> int:
    """Floats per entity in a sample-record table of the pipelined step: (w, weighted KL, 0, 0 | z[0..d-1])."""
    return 4 + d


def pipeline_supported(spec: Spec) -> bool:
    """The software-pipelined step covers the reference's own shape: two fields, one sample, d % 4 == 0, d <= 512."""
    return spec.F == 2 and int(spec.n_samples) == 1 and spec.d % 4 == 0 and spec.d <= 512 and spec.link == "abs"


def sample_records(plan: BatchPlan, entity_params, bias_params, inv_occ, zrec, seed, step):
    """vfm_sample_records_f32: the sample records of `plan`'s entities for Philox step `step`, from the tables."""
    p = _problem(plan.spec, plan.B, plan.B_global, plan.id_bits, seed, step, 0)
    ids = plan.touched_ids()
    check(_lib.load().vfm_sample_records_f32(C.byref(p), ptr(ids), ids.numel(), ptr(entity_params), ptr(bias_params),
                                             ptr(inv_occ), ptr(plan.W), ptr(zrec), current_stream_ptr(zrec.device)),
          "vfm_sample_records_f32")


def elbo_forward_records(plan: BatchPlan, zrec, scalars, seed, step, out_pred, out_grow, out_partials, dev_step=None, flags=0) -> FwdState:
    """The forward of the pipelined step (VFM_FLAG_ZREC): a gather of this step's sample records; no sumz."""
    spec = plan.spec
    p = _problem(spec, plan.B, plan.B_global, plan.id_bits, seed, step, FLAG_ZREC | flags, dev_step=dev_step)
    check(_lib.load().vfm_elbo_fwd_f32(C.byref(p), ptr(plan.x), ptr(plan.y), ptr(zrec), None, None, ptr(scalars), None,
                                       None, None, None, ptr(out_pred), ptr(out_partials), None, ptr(out_grow),
                                       current_stream_ptr(zrec.device)), "vfm_elbo_fwd_f32 (records)")
    p.flags = spec.link_flag | (flags & FLAG_SHARE_GPU)
    return FwdState(out_pred, out_partials, None, out_grow, p, None)


def elbo_backward_adam_pipe(plan: BatchPlan, st: FwdState, zrec, zrec_next, next_plan, next_step, entity_params,
                            bias_params, scalars, inv_occ, m_views, v_views, lr, step, loss_out, beta1=0.9, beta2=0.999,
                            eps_adam=1e-8, scaled_moments=True, wrec=None, dev_step=None, last_step=None, step_tab=None,
                            listed=True, la_next=None):
    """vfm_elbo_bwd_adam_pipe_f32: loss + backward + dense Adam, gathering samples from `zrec`, and (next_plan given)
    writing the records of `next_plan`'s entities for Philox step `next_step` into `zrec_next`."""
    ix = _index_struct(plan)
    ix.occ_other = plan.occ_other.data_ptr()
    pipe = _lib.Pipe()
    pipe.zrec = zrec.data_ptr()
    if next_plan is not None:
        if next_plan.occ_ptr is None:
            next_plan.build_index()
        pipe.zrec_next, pipe.next_occ_ptr, pipe.next_W = zrec_next.data_ptr(), next_plan.occ_ptr.data_ptr(), next_plan.W.data_ptr()
        pipe.next_step = int(next_step)
    if last_step is not None:          # look-ahead form: visit the rows of this batch and of `la_next` only
        if la_next.occ_ptr is None:
            la_next.build_index()
        pipe.last_step, pipe.step_tab = last_step.data_ptr(), step_tab.data_ptr()
        pipe.next_occ_ptr = la_next.occ_ptr.data_ptr()
        if next_plan is not None and next_plan is not la_next:
            raise ValueError("pipelined look-ahead step: the records are prepared for the batch the look-ahead names")
        if listed:
            rows = plan.lookahead_rows(la_next)
            if rows.numel() > 0:
                ix.touched_ids, ix.n_touched = rows.data_ptr(), rows.numel()
    p = st.problem
    p.flags = plan.spec.link_flag | (FLAG_SCALED_MOMENTS if scaled_moments else 0) | (p.flags & FLAG_SHARE_GPU)
    p.wrec = wrec.data_ptr() if wrec is not None else None
    p.dev_step = dev_step.data_ptr() if dev_step is not None else None
    check(_lib.load().vfm_elbo_bwd_adam_pipe_f32(
        C.byref(p), C.byref(ix), C.byref(pipe), ptr(entity_params), ptr(bias_params), ptr(scalars), ptr(inv_occ),
        ptr(plan.W), ptr(st.grow), ptr(st.partials), ptr(m_views[0]), ptr(v_views[0]), ptr(m_views[1]), ptr(v_views[1]),
        ptr(m_views[2]), ptr(v_views[2]), lr, beta1, beta2, eps_adam, int(step), ptr(loss_out),
        current_stream_ptr(zrec.device)), "vfm_elbo_bwd_adam_pipe_f32")


def elbo_backward_adam_lookahead(plan: BatchPlan, st: FwdState, next_plan: BatchPlan, entity_params, bias_params,
                                 scalars, inv_occ, m_views, v_views, lr, step, loss_out, last_step, step_tab,
                                 beta1=0.9, beta2=0.999, eps_adam=1e-8, listed=True, wrec=None, dev_step=None):
    """vfm_elbo_bwd_adam_lookahead_f32: the fused dense step visiting only the rows of this batch and of the next
    (listed: as a list made once per pair of plans; else the kernel classifies all T rows itself)."""
    ix = _index_struct(plan)
    if listed:
        rows = plan.lookahead_rows(next_plan)
        if rows.numel() > 0:
            ix.touched_ids, ix.n_touched = rows.data_ptr(), rows.numel()
    if next_plan.occ_ptr is None:
        next_plan.build_index()
    p = st.problem
    share = p.flags & FLAG_SHARE_GPU
    p.flags = plan.spec.link_flag | FLAG_SCALED_MOMENTS | share
    p.wrec = wrec.data_ptr() if wrec is not None else None
    p.dev_step = dev_step.data_ptr() if dev_step is not None else None
    check(_lib.load().vfm_elbo_bwd_adam_lookahead_f32(
        C.byref(p), C.byref(ix), ptr(entity_params), ptr(bias_params), ptr(scalars), ptr(inv_occ), ptr(plan.W),
        ptr(st.sumz), ptr(st.grow), ptr(st.partials), ptr(m_views[0]), ptr(v_views[0]), ptr(m_views[1]), ptr(v_views[1]),
        ptr(m_views[2]), ptr(v_views[2]), lr, beta1, beta2, eps_adam, int(step), ptr(loss_out), ptr(last_step),
        ptr(next_plan.occ_ptr), ptr(step_tab), current_stream_ptr(entity_params.device)),
        "vfm_elbo_bwd_adam_lookahead_f32")
    p.flags = plan.spec.link_flag | share


def adam_catchup(entity_params, bias_params, m_views, v_views, last_step, ids, lr_of_step, upto, mark,
                 beta1=0.9, beta2=0.999, eps_adam=1e-8, wrec=None):
    """Launch vfm_adam_catchup_f32 (lazy exact dense Adam): replay the skipped zero-gradient updates of steps
    last_step[e]+1 .. upto on the rows `ids` (int32 tensor; None = all rows), then stamp them with `mark`."""
    T, d = entity_params.shape[0], entity_params.shape[1] // 2
    n = T if ids is None else ids.numel()
    need = (int(upto) - 1) % MOMENT_PERIOD + 1 if upto > 0 else 0     # steps of upto's moment period, up to upto
    vals = [float(v) for v in lr_of_step] + [0.0] * max(0, need - len(lr_of_step))    # (placeholders: never replayed --
    lrs = (C.c_float * max(len(vals), 1))(*vals)                                        #  every row is past those steps)
    check(_lib.load().vfm_adam_catchup_f32(ptr(entity_params), ptr(bias_params), ptr(m_views[0]), ptr(v_views[0]),
                                           ptr(m_views[1]), ptr(v_views[1]), ptr(last_step), ptr(ids), n, T, d, lrs,
                                           len(vals), beta1, beta2, eps_adam, int(upto), int(mark), ptr(wrec),
                                           current_stream_ptr(entity_params.device)), "vfm_adam_catchup_f32")


def wrec_build(bias_params, inv_occ, wrec):
    """vfm_wrec_build_f32: the packed first-order records (mu_w, s_w, 1/occ, 0) of every entity (vfm_problem_t.wrec)."""
    check(_lib.load().vfm_wrec_build_f32(ptr(bias_params), ptr(inv_occ), bias_params.shape[0], ptr(wrec),
                                         current_stream_ptr(wrec.device)), "vfm_wrec_build_f32")


class StepState:
    """Device-resident step state (`vfm_dev_step_t` + its table of per-step Adam constants): what lets a captured HIP
    graph of a training step be REPLAYED -- the kernels read the Philox step and the Adam constants from here and
    advance the counters themselves (include/vfm_hip.h).  The table is filled on the host by `vfm_step_consts` (the
    arithmetic of the host-argument path, so a replayed trajectory is bitwise the eager one).  For callers that capture
    their own graphs (`VFM.fit` launches eagerly: tests/test_gpu_replay.py is the worked example)."""
    WINDOW = 4096          # Adam steps per table fill

    def __init__(self, device):
        self.dev = torch.zeros(8, dtype=torch.int64, device=device)                   # vfm_dev_step_t
        self.tab = torch.zeros(self.WINDOW, 16, dtype=torch.float32, device=device)   # vfm_step_consts_t [WINDOW]
        self.first, self.length, self.key = 0, 0, None
        self.counters = None             # (philox_step, adam_step) the device holds if every launch since was a replay

    def cover(self, adam_step: int, lr: float, beta1: float, beta2: float, eps: float, scaled: bool):
        """Make sure the table holds the constants of `adam_step` (the next update) for these hyper-parameters."""
        key = (float(lr), float(beta1), float(beta2), float(eps), bool(scaled))
        if key == self.key and self.first <= adam_step < self.first + self.length:
            return
        lib = _lib.load()
        host = (_lib.StepConsts * self.WINDOW)()
        for i in range(self.WINDOW):
            check(lib.vfm_step_consts(key[0], key[1], key[2], key[3], adam_step + i, int(scaled), C.byref(host[i])),
                  "vfm_step_consts")
        raw = torch.frombuffer(bytearray(bytes(host)), dtype=torch.float32).view(self.WINDOW, 16)
        self.tab.copy_(raw)               # (synchronous copy from pageable memory: the host buffer may go right after)
        hdr = _lib.DevStep()
        hdr.tab_first, hdr.tab_len, hdr.tab = adam_step, self.WINDOW, self.tab.data_ptr()
        hdr.error = int(self.dev[7].item())       # (sticky: a table miss of earlier replays survives the refill)
        self.dev.copy_(torch.frombuffer(bytearray(bytes(hdr)), dtype=torch.int64))
        self.first, self.length, self.key, self.counters = adam_step, self.WINDOW, key, None

    def set(self, philox_step: int, adam_step: int):
        if self.counters != (philox_step, adam_step):
            check(_lib.load().vfm_dev_step_set(ptr(self.dev), int(philox_step), int(adam_step),
                                               current_stream_ptr(self.dev.device)), "vfm_dev_step_set")
            self.counters = (philox_step, adam_step)

    def advanced(self):
        """A replayed step ran: the kernels moved both counters on by one."""
        self.counters = (self.counters[0] + 1, self.counters[1] + 1)

    def error(self) -> bool:
        return bool(int(self.dev[7].item()))


def exchange_record_len(d: int) -> int:
    """Floats per entity in the multi-rank exchange buffer: (sum grow, count, 0, 0 | A_e[0..d-1])."""
    return 4 + (d + 3) // 4 * 4


def elbo_backward_acc(plan: BatchPlan, st: FwdState, acc, sums, e_lo=0, e_hi=0):
    """Multi-rank backward, stage 1: this shard's sufficient statistics of the gradient for the
    entities [e_lo, e_hi) -- per entity (sum grow_r, count, A_e = sum grow_r*sumz_r), about half the
    bytes of the gradient; the caller sums them over ranks."""
    if plan.occ_ptr is None:
        plan.build_index()
    spec = plan.spec
    _lib.ops().elbo_bwd_acc(plan.index_tensors(), st.sumz, st.grow, st.partials, acc, sums,
                            spec.T, spec.F, spec.d, e_lo, e_hi)


def elbo_apply_adam(plan: BatchPlan, st: FwdState, acc, sums, entity_params, bias_params, scalars,
                    inv_occ, m_views, v_views, lr, step, beta1=0.9, beta2=0.999, eps_adam=1e-8, e_lo=0, e_hi=0,
                    scaled_moments=False):
    """Multi-rank backward, stage 2: gradient epilogue + dense Adam from the rank-summed statistics of
    the entities [e_lo, e_hi) (the chunk that ends at T also updates the three scalars)."""
    e = st.eps if st.eps is not None else (None, None, None)
    p, spec = st.problem, plan.spec
    _lib.ops().elbo_apply_adam(acc, sums, entity_params, bias_params, scalars, inv_occ, plan.W,
                               e[0], e[1], e[2], m_views[0], v_views[0], m_views[1], v_views[1], m_views[2],
                               v_views[2], spec.F, list(spec.group_hi), list(spec.group_n), p.nb_train,
                               p.B_global, p.likelihood,
                               (p.flags & ~FLAG_NO_PRIOR_TERMS) | (FLAG_SCALED_MOMENTS if scaled_moments else 0),
                               p.seed & _I63, p.step,
                               lr, beta1, beta2, eps_adam, int(step), e_lo, e_hi)


def elbo_backward_acc_rows(plan: BatchPlan, st: FwdState, row_ids, acc, sums):
    """vfm_elbo_bwd_acc_rows_f32: this shard's gradient statistics of the listed rows (sorted int32 ids), written
    COMPACTLY: record i of `acc` belongs to row_ids[i] (zeros for listed rows this shard does not contain)."""
    ix = _index_struct(plan)
    spec = plan.spec
    p = _problem(spec, plan.B, plan.B, 64)
    check(_lib.load().vfm_elbo_bwd_acc_rows_f32(C.byref(p), C.byref(ix), ptr(row_ids), row_ids.numel(), ptr(st.sumz),
                                                ptr(st.grow), ptr(st.partials), ptr(acc), ptr(sums),
                                                current_stream_ptr(acc.device)), "vfm_elbo_bwd_acc_rows_f32")


def elbo_apply_adam_rows(plan: BatchPlan, st: FwdState, acc, sums, row_ids, entity_params, bias_params, scalars, inv_occ,
                         m_views, v_views, lr, step, move_scalars: bool, compact: bool = False, beta1=0.9, beta2=0.999,
                         eps_adam=1e-8, wrec=None):
    """vfm_elbo_apply_adam_rows_f32: the apply stage over a sorted int32 list of rows (the multi-rank step's lazy exact
    Adam: the rows some rank's shard contains; scaled moments).  move_scalars: this launch also updates the three scalars;
    compact: `acc` is the compact buffer (record i <-> row_ids[i]) instead of the dense table."""
    p0, spec = st.problem, plan.spec
    p = _problem(spec, 0, p0.B_global, 64, p0.seed, p0.step, (p0.flags & ~FLAG_NO_PRIOR_TERMS) | FLAG_SCALED_MOMENTS, wrec=wrec)
    p.e_lo, p.e_hi = 0, (0 if move_scalars else 1)
    check(_lib.load().vfm_elbo_apply_adam_rows_f32(
        C.byref(p), ptr(acc), ptr(sums), ptr(row_ids), row_ids.numel(), int(bool(compact)), ptr(entity_params), ptr(bias_params),
        ptr(scalars),
        ptr(inv_occ), ptr(plan.W), ptr(m_views[0]), ptr(v_views[0]), ptr(m_views[1]), ptr(v_views[1]), ptr(m_views[2]),
        ptr(v_views[2]), lr, beta1, beta2, eps_adam, int(step), current_stream_ptr(entity_params.device)),
        "vfm_elbo_apply_adam_rows_f32")


def moments_rescale(m, v, step, to_scaled, beta1=0.9, beta2=0.999):
    """Convert flat Adam moment buffers between the plain and the scaled form (`step` = Adam steps so far)."""
    _lib.ops().moments_rescale(m, v, beta1, beta2, int(step), bool(to_scaled))


def adam_step(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
    """In-place dense Adam (torch.optim.Adam defaults, vfm-torch.py:339,370) on one flat tensor."""
    for t in (p, g, m, v):
        _need_cuda(t, "adam tensor")
    _lib.ops().adam(p, g, m, v, lr, beta1, beta2, eps, int(step))


def philox_eps(spec: Spec, seed: int, step: int, device):
    """The eps tables the kernels generate for (seed, step) -- test helper.  [T,d], [T], [1]; with
    S = spec.n_samples > 1: [S,T,d], [S,T], [S]."""
    S = int(spec.n_samples)
    lead = () if S == 1 else (S,)
    ee = torch.empty(*lead, spec.T, spec.d, dtype=torch.float32, device=device)
    eb = torch.empty(*lead, spec.T, dtype=torch.float32, device=device)
    eg = torch.empty(S, dtype=torch.float32, device=device)
    p = _problem(spec, 0, 0, 64, seed, step)
    lib = _lib.load()
    check(lib.vfm_philox_eps_f32(C.byref(p), ptr(ee), ptr(eb), ptr(eg), current_stream_ptr(ee.device)),
          "vfm_philox_eps_f32")
    return ee, eb, eg


class ElboFunction(torch.autograd.Function):
    """loss, pred = ElboFunction.apply(entity_params, bias_params, alpha, m0, s0, plan, inv_occ,
    eps, seed, step).  Differentiable in the five parameters (vfm-torch.py:136-138,152-153)."""

    @staticmethod
    def forward(ctx, entity_params, bias_params, alpha, global_bias_mean, global_bias_scale,
                plan, inv_occ, eps, seed, step):
        scalars = torch.cat([alpha.detach().reshape(1), global_bias_mean.detach().reshape(1),
                             global_bias_scale.detach().reshape(1)]).contiguous()
        ent, bia = entity_params.detach(), bias_params.detach()
        st = elbo_forward(plan, ent, bia, scalars, inv_occ, eps=eps, seed=seed, step=step, train=True)
        loss3 = elbo_finalize(st, scalars)
        ctx.plan, ctx.st, ctx.inv_occ, ctx.scalars = plan, st, inv_occ, scalars
        ctx.save_for_backward(ent, bia)
        ctx.mark_non_differentiable(st.pred, loss3)
        ctx.set_materialize_grads(True)
        return loss3[0:1].clone(), st.pred, loss3

    @staticmethod
    def backward(ctx, g_loss, _g_pred, _g_loss3):
        ent, bia = ctx.saved_tensors
        gout = g_loss.to(torch.float32).reshape(1).contiguous()
        g_ent, g_bias, g_sc = elbo_backward(ctx.plan, ctx.st, ent, bia, ctx.scalars, ctx.inv_occ, gout)
        return (g_ent, g_bias, g_sc[0:1], g_sc[1:2], g_sc[2:3], None, None, None, None, None)
