#!/usr/bin/env python3
"""bench.py -- rating-triples/s of the Variational-FM ELBO training step on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 under torch.distributed.run, one
rank per GPU, RCCL).  One "step" = one full iteration of the reference loop body
(vfm-torch.py:351-370) over one batch: fused forward kernel, loss, backward kernel, (all-reduce
of the gradients when N>1,) dense Adam over every parameter.  Inputs (ids, targets, occurrence
counts, per-batch normalisers and inverted index) are resident in HBM before the timed region.

Workload (BASELINE.json configs[2]): synthetic ML-20M-shape triples, N=138,493 users x M=26,744
items, d=128, global batch B=100,000 rows (the reference's BATCH_SIZE, vfm-torch.py:77),
uniform-random ids, ratings randint(1,6), eps generated in-kernel (Philox).
N>1 (BASELINE configs[3]): the north star's pattern only -- the batch row-sharded over the ranks, tables replicated, ONE
all-reduce per step -- in its three forms (`stats`: the gradient's sufficient statistics, compacted to the entities
some rank's shard contains; `grads`: the flat gradient buffer; `rows`: every row's dloss/dpred), each timed, the fastest one
is the headline (`candidates` lists all, each with the rows its forward / backward handle per rank and the table rows it
updates per rank).  Headline regime: STRONG (SURVEY cfg4: the 100,000-row batch split by rows); the same mode in
the WEAK regime (100,000 rows per rank) under the key `weak`.  Every candidate is probed first and dropped if it projects
past --mode-budget-s or past what is left of --total-budget-s; a watchdog ends a run that hangs in a collective after
--wall-limit-s (480 s: below the 600 s the round-end driver gives a run).

Prints ONE JSON line (rank 0).  `roofline` describes the kernel that takes the most time per step (its `elbo_fwd_kernel`
entry is the fused forward ELBO kernel; `frac_fwd_8d` / `frac_K_8d` are SURVEY 8(d)'s own figures against the 0.60
target); `kernels` lists all of them (HIP events on every 5th step of the timed region); `sustained` = 2,000
more steps without events; `streamed` = the regime of a caller that cannot keep plans (streamed / shuffled batches): every
step's plan -- inverted index, normalisers, look-ahead row list: what replaces the reference's torch.unique x3,
vfm-torch.py:190-192 -- is built INSIDE the timed region, a few steps ahead on side streams (`--plans stream` makes that the
headline region).  `cpu_baseline` = the reference-shaped torch-CPU restatement
(oracle/vfm_oracle.py, pinned to the reference by tests/golden; the reference's two unused per-row lookups included)
timed on this node's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ~6.3 TB/s

WORKLOADS = {
    # name: (field sizes, d, batch, nb_train, output)
    "ml20m_d128": ([138493, 26744], 128, 100000, 16000210, "reg"),
    "ml100k_d20": ([943, 1682], 20, 80000, 80000, "reg"),
    "criteo_d256": ([31250] * 32, 256, 2048, 1 << 22, "class"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="ml20m_d128", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="rows per step of the GLOBAL batch (default: the workload's)")
    ap.add_argument("--n-batches", type=int, default=16, help="distinct batches cycled through")
    ap.add_argument("--id32", action="store_true", help="int32 ids instead of the reference's int64")
    ap.add_argument("--no-sort", action="store_true",
                    help="keep the generated row order (default: rows of each batch ordered by item id, "
                         "as VFM.fit(sort_within_batch=True) does; loss and gradients are invariant)")
    ap.add_argument("--exchange", default="north-star", choices=["north-star", "stats", "grads", "rows"],
                    help="N>1, what the ranks exchange per step (DESIGN.md section 6).  north-star (default): the row-sharded "
                         "batch with ONE all-reduce per step, in its three forms -- `stats` (sufficient statistics of the "
                         "gradient), `grads` (the literal gradient) and `rows` (every row's dloss/dpred; two fields) -- each "
                         "timed, the fastest one is the headline.")
    ap.add_argument("--plans", default="resident", choices=["resident", "stream"],
                    help="N=1.  resident (default): plans are built once per batch before the timed region and reused (the "
                         "reference's loader does not shuffle, vfm-torch.py:121-122); stream: the timed region builds every plan "
                         "it uses, a few steps ahead (VFM.plan_prefetch_depth) on side streams -- the regime of a caller that streams or shuffles batches. "
                         "Whatever this says, the line carries both figures (`ms_per_step` / `streamed`).")
    ap.add_argument("--streamed-steps", type=int, default=400, help="steps of the extra `streamed` region (0: none)")
    ap.add_argument("--scaling", default="both", choices=["both", "strong", "weak"],
                    help="N>1: strong = the workload's global batch split by rows over the ranks (SURVEY cfg4; the headline), "
                         "weak = that many rows PER rank; both = the two in one run (weak under the key `weak`)")
    ap.add_argument("--strong", action="store_true", help="same as --scaling strong")
    ap.add_argument("--mode-budget-s", type=float, default=90.0,
                    help="N>1: a candidate whose 3-step probe projects more than this many seconds for warm-up + timed steps "
                         "is skipped (recorded, never retried)")
    ap.add_argument("--wall-limit-s", type=float, default=480.0,
                    help="watchdog: if the whole run is still going after this many seconds (a hung collective), every rank "
                         "prints what it was doing and exits with code 3 (the round-end driver gives a run 600 s)")
    ap.add_argument("--total-budget-s", type=float, default=300.0,
                    help="N>1: what init + all candidates + both regimes may take; a candidate whose probe projects past what "
                         "is left of it is skipped")
    ap.add_argument("--dim", type=int, default=0, help="embedding size instead of the workload's (shape sweeps)")
    ap.add_argument("--zipf", type=float, default=0.0, help="Zipf exponent of the item popularity (0 = uniform ids)")
    ap.add_argument("--user-order", action="store_true",
                    help="generated rows sorted by user id before they are cut into batches -- the order of the reference's data "
                         "files (a batch then holds the consecutive ratings of a few hundred users instead of 100,000 random pairs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=25.0)
    ap.add_argument("--no-events", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--event-every", type=int, default=5,
                    help="record the per-kernel HIP events on every n-th step of the timed region (5 is coprime with the 16 batches and the "
                         "128-step moment period, so no step kind is over-sampled; recording them on every "
                         "step costs the launch-bound shapes ~10 us per step: ML-100K shape 60 -> 45 us)")
    ap.add_argument("--sync-each-step", action="store_true", help="debug: host sync after every step")
    ap.add_argument("--unfused", action="store_true", help="separate backward and Adam kernels (as N>1 does)")
    ap.add_argument("--fwd-eps", default="philox", choices=["philox", "table"],
                    help="A/B: eps of the forward / backward from tables (one fixed draw, as the parity tests feed "
                         "the reference's recorded draws) instead of the in-kernel Philox stream")
    ap.add_argument("--pipeline", default="auto", choices=["auto", "on", "off"],
                    help="the software-pipelined step (the backward of step n writes the sample records of batch n+1, the "
                         "forward gathers records: no sampling, no sumz); auto: from 2 rows per distinct entity")
    ap.add_argument("--lookahead", default="on", choices=["on", "scan", "off"],
                    help="look-ahead lazy exact Adam: the fused step skips the rows that are neither in this batch nor in "
                         "the next (bitwise the dense trajectory); scan = the kernel classifies all table rows itself instead "
                         "of walking the pair's row list; off = every row every step")
    ap.add_argument("--sustained-steps", type=int, default=2000)
    ap.add_argument("--settle-steps", type=int, default=300,
                    help="untimed training steps before the warm-up of a commanded region (0: none); the same region "
                         "without them is timed first and reported as settle.cold_ms_per_step")
    ap.add_argument("--settle-max-s", type=float, default=1.0, help="wall-clock bound of the settling steps")
    ap.add_argument("--no-wrec", action="store_true", help="A/B: without the packed first-order records")
    ap.add_argument("--no-regions", action="store_true", help="skip the short unfused run that measures region K")
    ap.add_argument("--lazy-adam", default="auto", choices=["auto", "on", "off"],
                    help="lazy exact dense Adam (rows outside the batch are skipped and replayed later, bitwise the dense "
                         "trajectory); auto: when a batch touches less than 35 %% of the table")
    ap.add_argument("--plain-moments", action="store_true",
                    help="A/B: keep the Adam moments in the plain form (no VFM_FLAG_SCALED_MOMENTS)")
    args = ap.parse_args()
    if args.strong:
        args.scaling = "strong"
    args.event_every = max(1, min(args.event_every, args.steps // 5))       # (at least five sampled steps in a short region)

    import threading
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # VFM_BENCH_FORCE_GROUP=1 at N = 1: run the MULTI-RANK path (process group, candidates, regimes) over a communicator
    # of world size 1 -- on a one-GPU box the only way to take bench.py's N > 1 code through RCCL itself (which refuses
    # two ranks on one device).  A rehearsal of the code path, labelled as such; not a measurement of anything.
    forced = world == 1 and os.environ.get("VFM_BENCH_FORCE_GROUP") == "1"
    # stdout carries ONE line, the JSON: whatever libraries print there (RCCL's version banner at communicator creation
    # goes to fd 1 in this image) is sent to stderr instead
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    multi = world > 1 or forced
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")

    # watchdog: a collective that never returns would otherwise cost the caller its whole time limit
    doing = ["start"]
    t_begin = time.perf_counter()

    def watchdog():
        while True:
            time.sleep(5.0)
            if time.perf_counter() - t_begin > args.wall_limit_s:
                print(json.dumps({"error": "bench.py wall limit", "rank": rank, "seconds": args.wall_limit_s,
                                  "was_doing": doing[0]}), file=sys.stderr, flush=True)
                os._exit(3)
    threading.Thread(target=watchdog, daemon=True).start()

    ndev = torch.cuda.device_count()
    backend = os.environ.get("VFM_BENCH_BACKEND", "nccl")    # nccl = RCCL; "gloo" only to rehearse on one GPU
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    pg = None
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        doing[0] = "init_process_group"
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        pg = dist.group.WORLD

    # what the collective library itself reports: N ranks, and a 1-element all-reduce that must give N
    comm_check = None
    if multi:
        doing[0] = "comm_check all_reduce"
        one = torch.ones(1, device=dev)
        dist.all_reduce(one, group=pg)
        try:        # what the collective library says about itself (RCCL reports through torch's nccl binding)
            lib_version = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
        except Exception as exc:
            lib_version = "unavailable: %s" % type(exc).__name__
        comm_check = {"backend": backend, "library": "RCCL (torch.distributed backend 'nccl' on ROCm)" if backend == "nccl" else backend,
                      "library_version": lib_version, "world_size": dist.get_world_size(pg),
                      "allreduce_of_ones": float(one.item()), "device_of_rank0": torch.cuda.get_device_name(dev)}
        assert comm_check["world_size"] == world and comm_check["allreduce_of_ones"] == float(world), comm_check

    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples

    sizes, d, B_global, nb_train, output = WORKLOADS[args.workload]
    if args.dim:
        d = args.dim
    if args.batch:
        B_global = args.batch
    F = len(sizes)
    nbt = max(1, min(args.n_batches, args.steps + args.warmup))

    torch.manual_seed(42)
    model = VFM(field_sizes=sizes, embedding_size=d, output=output, device=dev, rng_seed=1234)
    model.scaled_moments = not args.plain_moments
    model.lazy_adam = {"auto": "auto", "on": True, "off": False}[args.lazy_adam]
    model.pipeline = {"auto": "auto", "on": True, "off": False}[args.pipeline]
    model.lookahead = args.lookahead != "off"
    model.lookahead_list = args.lookahead == "on"
    model.use_wrec = not args.no_wrec
    eps_tables = None
    if args.fwd_eps == "table":
        from vae_amd import ops as _ops
        eps_tables = _ops.philox_eps(model.spec(), seed=1, step=1, device=dev)

    def barrier():
        torch.cuda.synchronize()
        if multi:
            dist.barrier(group=pg)
            torch.cuda.synchronize()

    def max_over_ranks(v):
        if multi:
            t = torch.tensor([v], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=pg)
            v = float(t.item())
        return v

    class Setup:
        """Synthetic data of one scaling regime resident on the device (this rank's B rows of nbt global batches of
        B * world rows), the occurrence counts of the (virtual) training set, and -- per exchange mode -- the plans."""

        def __init__(self, B):
            self.B = B
            self.X, self.y = synthetic_triples(sizes, nbt * B, seed=1000 + rank, output=output, device=dev,
                                               zipf=args.zipf if args.zipf > 0 else None)
            if args.user_order:      # the data files' order (ML-20M's ratings.csv is sorted by user): a batch = few users' runs
                o_ = torch.argsort(self.X[:, 0], stable=True)
                self.X, self.y = self.X[o_].contiguous(), self.y[o_].contiguous()
            if args.id32:
                self.X = self.X.to(torch.int32)
            # expected counts of nb_train uniform triples, at least the counts seen in the generated rows (keeps
            # 1/occ finite for every touched id)
            occ = torch.bincount(self.X.reshape(-1).to(torch.int64), minlength=model.T)
            if multi:
                dist.all_reduce(occ, group=pg)
            scale = max(1.0, nb_train / float(nbt * B * world))
            self.occ = torch.clamp((occ.to(torch.float64) * scale).round().to(torch.int64), min=1)
            self.lr = 1.0 / (1 + nb_train // (B * world))       # vfm-torch.py:92
            self.Xg = self.yg = None
            self.plans = {}
            self.plan_build, self.plan_build_warm = {}, {}

        def activate(self):
            model.sync_lazy()
            model.set_training_data(self.X, nb_train=nb_train, nb_occ=self.occ)
            model.lr = self.lr

        def build_plans(self, mode):
            """The per-batch work OUTSIDE the timed step: batch normalisers W (k_norms), the inverted index
            (vfm_build_index: radix sort) and its one readback.  The reference pays torch.unique x3 inside every
            step (vfm-torch.py:190-192); here a plan is built once per batch and reused every epoch (the loader
            does not shuffle, :121-122), so its cost is reported separately and amortised over the 50 epochs."""
            key = mode if multi else "single"
            if key in self.plans:
                return self.plans[key]
            B = self.B
            ps, batches = [], []
            for i in range(nbt):
                xb, yb = self.X[i * B:(i + 1) * B], self.y[i * B:(i + 1) * B]
                if not args.no_sort:
                    o = torch.argsort(xb[:, -1], stable=True)
                    xb, yb = xb[o].contiguous(), yb[o].contiguous()
                batches.append((xb, yb))
            model.exchange = mode
            model.plan(batches[0][0], batches[0][1], B_global=B * world, process_group=pg)      # (allocator / module warm-up)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for xb, yb in batches:         # as fit() does: the index builds are enqueued back to back, their 16-byte
                ps.append(model.plan(xb, yb, B_global=B * world, process_group=pg, defer_readback=True))     # readbacks
            if not multi and model.lookahead and model.lookahead_list:
                for i_, p_ in enumerate(ps):   # the look-ahead step's row lists (this batch + the next one), also per batch
                    p_.prepare_lookahead(ps[(i_ + 1) % len(ps)])
            for p_ in ps:                  # are collected afterwards
                p_.U
            if not multi and model.lookahead and model.lookahead_list:
                for i_, p_ in enumerate(ps):   # (the row lists' deferred counts too: no readback is left for the timed region)
                    if model._lookahead_pays(p_, ps[(i_ + 1) % len(ps)]):
                        p_.lookahead_rows(ps[(i_ + 1) % len(ps)])
            torch.cuda.synchronize()
            self.plan_build[key] = (time.perf_counter() - t0) / nbt * 1e3
            if not multi:       # the same once more (discarded): the allocator now has the buffers -- what a loop that
                t0 = time.perf_counter()                 # rebuilds its plans every epoch (shuffled batches) pays per batch
                again = [model.plan(xb, yb, B_global=B * world, process_group=pg, defer_readback=True) for xb, yb in batches]
                for p_ in again:
                    p_.U
                torch.cuda.synchronize()
                self.plan_build_warm[key] = (time.perf_counter() - t0) / nbt * 1e3
                del again
            self.batches = batches
            self.plans[key] = (ps, [p.U for p in ps])
            return self.plans[key]

    step_no = [0]                     # batches are cycled through in order across warm-up and timed regions

    def run_streamed(setup, n):
        """n steps of a caller that keeps NO plan: the plan of batch t + depth (index, normalisers) is built on a side stream
        while step t runs -- everything that replaces the reference's
        torch.unique x3 (vfm-torch.py:190-192) is inside the loop."""
        bt = setup.batches
        t0 = step_no[0]
        D = max(2, int(model.plan_prefetch_depth))
        model.plan_streams().wait_current()
        q = [model.plan(*bt[t0 % nbt], defer_readback=True)] + [model.plan_async(*bt[(t0 + k) % nbt]) for k in range(1, D)]
        trace = os.environ.get("VFM_BENCH_TRACE")      # debug: host time of every 50 steps of the streamed loop
        tt = time.perf_counter()
        for i in range(n):       # D plans in hand: this batch's, the next one's (named to the step), the ones being built
            s = step_no[0]
            step_no[0] += 1
            model.train_step(q[0], next_plan=q[1], fused=not args.unfused, prefetch=bt[(s + D) % nbt] + (False,))   # (resident data: no fork)
            q.pop(0)
            q.append(model.prefetched)
            if trace and i % 50 == 49:
                print(f"[trace] steps ..{s}: host {(time.perf_counter() - tt) / 50 * 1e3:.4f} ms/step, adam_t {model._adam_t}, "
                      f"reserved {torch.cuda.memory_reserved() >> 20} MiB", file=sys.stderr, flush=True)
                tt = time.perf_counter()

    # HIP timing events of the commanded region are CREATED before it (measure() fills the pool and records each once: torch
    # makes the hipEvent at the first record).  Creating them inside cost the first process of a session ONE stall of 15-20 ms
    # around the 80th event -- the runtime growing its pool of profiling signals -- i.e. 0.222-0.256 instead of 0.191-0.195 ms
    # per step over 500 steps; later processes on the box did not show it (tools/first_run_ab.sh, VFM_BENCH_DUMP_EVENTS).
    event_pool = []

    def fill_event_pool(n):
        while len(event_pool) < n:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            event_pool.append(e)
        torch.cuda.synchronize()

    def run(plans, n, events, streamed=None):
        if streamed is not None:
            return run_streamed(streamed, n)
        for _ in range(n):
            s = step_no[0]
            step_no[0] += 1
            mark = None
            if events is not None and s % max(1, args.event_every) == 0:
                ev = {}
                events.append(ev)

                def mark(name, ev=ev):
                    e = event_pool.pop() if event_pool else torch.cuda.Event(enable_timing=True)
                    e.record()
                    ev[name] = e
            model.train_step(plans[s % nbt], process_group=pg, mark=mark, fused=not args.unfused, eps=eps_tables,
                             next_plan=plans[(s + 1) % nbt] if (not multi or model.exchange == "rows") else None)
            if args.sync_each_step:
                torch.cuda.synchronize()

    def measure(setup, mode, steps, warmup, with_events, streamed=False, probe=False, settle=0):
        """warm-up, then `steps` timed steps bracketed by barrier + synchronize on both sides, MAX over ranks.  With
        `probe`, three steps are timed first and the candidate is dropped if it projects past --mode-budget-s."""
        doing[0] = f"measure {mode} B={setup.B}"
        model.exchange = mode
        plans, uniq = setup.build_plans(mode)
        if probe:
            run(plans, 2, None)
            barrier()
            t0 = time.perf_counter()
            run(plans, 3, None)
            barrier()
            per = max_over_ranks(time.perf_counter() - t0) / 3
            left = max_over_ranks(args.total_budget_s - (time.perf_counter() - t_begin))     # (every rank takes the same decision)
            need = per * (steps + warmup + (settle if settle else 0) + 2 * min(steps, 20))
            if need > args.mode_budget_s or need > left:
                return {"skipped": f"probe: {per * 1e3:.2f} ms per step projects {need:.1f} s: past --mode-budget-s "
                                   f"{args.mode_budget_s} or the {left:.1f} s left of --total-budget-s {args.total_budget_s}"}
        stm = setup if streamed else None          # (run(): the streamed loop when a Setup is given)
        cold = None
        if settle > 0:
            # The chip needs 10-50 ms of THIS work after an idle stretch before the backward kernel reaches its steady
            # duration (tools/transient_probe.py: 163 -> 151 us over the first 60-260 steps, again after 0.3 s of
            # idleness, not after a reload of the same values without one).  A commanded region of 20 steps is 4 ms,
            # so it is preceded by untimed settling steps -- real training steps, like the warm-up's -- and the same
            # region WITHOUT them is timed first and reported beside the headline as `settle.cold_ms_per_step`.
            kc = min(steps, 20)
            run(plans, warmup, None, stm)
            barrier()
            t0 = time.perf_counter()
            run(plans, kc, None, stm)
            barrier()
            cold = {"steps": kc, "ms_per_step": max_over_ranks(time.perf_counter() - t0) / kc * 1e3}
            done, t0 = 0, time.perf_counter()
            while done < settle:
                run(plans, 50, None, stm)
                done += 50
                torch.cuda.synchronize()
                if max_over_ranks(time.perf_counter() - t0) >= args.settle_max_s:     # (every rank takes the same decision)
                    break
            cold["settle_steps"] = done
        if with_events:
            fill_event_pool(6 * (steps // max(1, args.event_every) + 2))
        run(plans, warmup, None, stm)
        barrier()
        events = [] if with_events else None
        t0 = time.perf_counter()
        run(plans, steps, events, stm)
        t_host = time.perf_counter() - t0          # host time to enqueue all steps (before the sync)
        lazy_kind = model._lazy_kind               # which lazy exact form the timed steps ran in (None: every row every step)
        model.sync_lazy()                          # rows still lagging get their skipped updates INSIDE the timed region
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        xbytes = None
        if multi:      # what ONE step's exchange carried (the last step's; bytes per rank into the all-reduce)
            xbytes = {"stats": 4 * int(model._exchanged_floats), "grads": 4 * int(model._gflat.numel()),
                      "rows": 4 * int(model._exchanged_floats)}.get(mode)
        return {"dt": dt, "t_host": t_host, "events": events, "lazy_kind": lazy_kind, "plans": plans, "cold": cold,
                "U": sum(uniq) / len(uniq), "B": setup.B, "steps": steps, "mode": mode, "exchange_bytes": xbytes}

    # ------------------------------------------------------------------ what to run
    exchange_note = None
    if not multi:
        regimes = [("strong", B_global)]
        cands = ["single"]
    else:
        regimes = [(r, (B_global + world - 1) // world if r == "strong" else B_global)
                   for r in (("strong", "weak") if args.scaling == "both" else (args.scaling,))]
        if args.exchange == "north-star":
            cands = ["stats", "grads"]
            from vae_amd.dist import rows_supported
            if rows_supported(model.spec()):
                cands.append("rows")
        else:
            cands = [args.exchange]
    init = model._flat.clone()
    model._ensure_opt_state()

    def reset_state():
        model.sync_lazy()
        model._flat.copy_(init)
        model._adam_m.zero_(); model._adam_v.zero_()
        model._adam_t, model.global_step, model._moments_scaled = 0, 0, False
        model.params_changed()

    # the box's streaming rate (device-to-device copies of 1 GiB) -- measured before the timed regions, it touches
    # nothing of the model
    copy_gbs = stream_copy_rate(dev)
    results = {}          # regime -> {mode: measurement}
    head = None           # the headline measurement: first regime, fastest north-star candidate
    for reg, B in regimes:
        setup = Setup(B)
        setup.activate()
        results[reg] = {}
        use = cands if reg == regimes[0][0] else [head["mode"]]       # the other regime: the headline's mode only
        for mode in use:
            reset_state()
            try:
                stream_head = not multi and args.plans == "stream"
                m = measure(setup, mode, args.steps, args.warmup, with_events=not args.no_events and not stream_head,
                            streamed=stream_head, probe=multi, settle=args.settle_steps)
            except Exception as exc:      # communication-pattern fallback only (never a compute fallback)
                if len(use) == 1:
                    raise
                m = {"skipped": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
                print("[bench] %s failed: %s" % (mode, m["skipped"]), file=sys.stderr)
            m["setup"] = setup
            results[reg][mode] = m
        ok = {k: v for k, v in results[reg].items() if "dt" in v}
        if not ok:
            raise SystemExit("[bench] no exchange pattern ran: " + str({k: v.get("skipped") for k, v in results[reg].items()}))
        if head is None:
            head = min(ok.values(), key=lambda v: v["dt"])
    model.exchange = head["mode"] if multi else model.exchange
    setup, B, U, dt, plans = head["setup"], head["B"], head["U"], head["dt"], head["plans"]
    events, lazy_kind, t_host = head["events"], head["lazy_kind"], head["t_host"]
    loss = float(model._gflat[model._n_flat].item())
    nan_params = int(torch.isnan(model._flat).sum().item())

    # ---- N = 1: figures next to the commanded region: `sustained` = the same steps over a long region without events;
    # `streamed` = the regime in which no plan is kept (every step's plan built inside the timed region, a few steps ahead on a
    # side stream) -- with the GPU time of one plan build, measured alone, beside it
    # (the auxiliary regions run AFTER the commanded one and must not cost the line: a Python error in one of them is
    #  reported in its key and on stderr, and the run goes on)
    sustained = streamed = None

    def aux(name, fn):
        try:
            return fn()
        except Exception as exc:
            print("[bench] %s region failed: %s: %s" % (name, type(exc).__name__, str(exc)[:300]), file=sys.stderr)
            return {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}

    def sustained_region():
        setup.activate()
        m = measure(setup, "single", args.sustained_steps, 4, with_events=False)
        return {"steps": args.sustained_steps, "ms_per_step": round(m["dt"] / args.sustained_steps * 1e3, 4),
                "triples_per_s": round(args.sustained_steps * B / m["dt"], 1),
                "host_enqueue_ms_per_step": round(m["t_host"] / args.sustained_steps * 1e3, 4)}
    if not multi and args.sustained_steps > 0:
        sustained = aux("sustained", sustained_region)

    def streamed_region():
        setup.activate()
        m = measure(setup, "single", args.streamed_steps, nbt + 4, with_events=False, streamed=True)
        # one plan build (index + normalisers) alone on the chip: HIP events around builds enqueued back to back on the current
        # stream -- the larger of the build's GPU time and the host time to enqueue it
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        keep_ = [model.plan(*setup.batches[i_ % nbt], defer_readback=True) for i_ in range(nbt)]
        e1.record()
        torch.cuda.synchronize()
        del keep_
        return {"steps": args.streamed_steps, "ms_per_step": round(m["dt"] / args.streamed_steps * 1e3, 4),
                "triples_per_s": round(args.streamed_steps * B / m["dt"], 1),
                "host_enqueue_ms_per_step": round(m["t_host"] / args.streamed_steps * 1e3, 4),
                "plan_build_gpu_us_alone": round(e0.elapsed_time(e1) / nbt * 1e3, 2),
                "prefetch_depth": int(model.plan_prefetch_depth), "step_lead": int(model.step_lead),
                "note": "every step builds the plan of the batch `prefetch_depth` steps ahead (inverted index + batch normalisers; the look-ahead "
                        "kernel classifies the table rows itself) on a side stream inside the timed region; nothing of a plan is reused; "
                        "plan_build_gpu_us_alone = max(GPU time, host enqueue time) of one build with nothing beside it"}
    if not multi and args.streamed_steps > 0 and not args.unfused and eps_tables is None:
        streamed = aux("streamed", streamed_region)
    if not multi and events is None and not args.no_events:
        # (--plans stream: the headline region carries no per-kernel events) per-kernel durations from a resident-plans pass
        m = measure(setup, "single", min(100, args.steps), 4, with_events=True)
        events, lazy_kind = m["events"], m["lazy_kind"]

    # ---- per-kernel durations from the HIP events recorded inside the timed region + the roofline object
    args._plans = plans
    kern, roof = kernel_report(events, model, args, world if not forced else 2, B, d, F, U, lazy_kind)
    if roof is not None:
        roof["box_stream_copy_GBs"] = round(copy_gbs, 1)
        roof["frac_of_box_stream_copy"] = round(roof["achieved"] / copy_gbs, 4)

    # ---- regions of SURVEY 8(d): F = forward only, K = forward + loss + backward kernels (no optimizer), S = the
    # full step.  The timed step fuses the backward with Adam, so K comes from a short run of the UNFUSED step
    # (separate k_bwd writing the dense gradient, then k_adam) after the timed region.
    regions = None

    def regions_region():
        ev2 = []
        for s_ in range(5):
            model.train_step(plans[s_ % nbt], fused=False, eps=eps_tables)
        nreg = min(40, max(10, args.steps))
        for s_ in range(nreg):
            ev = {}
            ev2.append(ev)

            def mark(name, ev=ev):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev[name] = e
            model.train_step(plans[s_ % nbt], fused=False, mark=mark, eps=eps_tables)
        torch.cuda.synchronize()
        order = list(ev2[0].keys())
        per = {k: [] for k in order[1:]}
        for ev in ev2:
            for a_, b_ in zip(order[:-1], order[1:]):
                per[b_].append(ev[a_].elapsed_time(ev[b_]))
        un = {k: sorted(v)[len(v) // 2] * 1e3 for k, v in per.items()}        # medians (one-off stalls do not count)
        K_us = un["fwd"] + un["finalize"] + un["bwd"]
        regions_ = {"F_us": kern["fwd"]["avg_us"], "K_us": round(K_us, 2),
                   "S_us": round(dt / args.steps * 1e6, 2),
                   "K_triples_per_s": round(B / (K_us * 1e-6), 1),
                   "unfused_step_us": {k: round(v, 2) for k, v in un.items()},
                   "note": "F and S from the timed region; K = k_fwd + k_finalize + k_bwd (dense gradient written), medians "
                           "over %d unfused steps run after it" % nreg}
        if roof is not None and "bwd" in un:
            # SURVEY 8(d)'s own figures, first-class: the forward kernel and region K on ITS algorithmic bytes
            # (bytes_fwd = U(8d+16) + B(8F+8); bytes_K = 2 bytes_fwd + U(8d+8)) against the 8 TB/s peak
            idb = 4 if args.id32 else 8
            b_fwd = U * (8 * d + 16) + B * (idb * F + 8)
            b_K = 2 * b_fwd + U * (8 * d + 8)
            roof["frac_fwd_8d"] = round(b_fwd / (kern["fwd"]["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            roof["frac_K_8d"] = round(b_K / (K_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            roof["bytes_fwd_8d"], roof["bytes_K_8d"] = int(b_fwd), int(b_K)
            roof["target_8d"] = 0.60
        return regions_
    if not multi and kern and not args.unfused and not args.no_regions:
        regions = aux("regions", regions_region)

    # ---- CPU baseline (rank 0, N=1): reference-shaped restatement on the host cores
    cpu = None
    if rank == 0 and not multi and not args.no_cpu_baseline:
        cpu = aux("cpu_baseline", lambda: cpu_baseline(sizes, d, B, nb_train, output, plans[0], setup.occ, args.cpu_seconds))

    if rank == 0:
        def line_of(m, reg):
            # what ONE rank's kernels handle per step in this form: a `rows` headline must not read as a data-parallel backward
            rows_b = m["B"] * world if m["mode"] == "rows" else m["B"]
            return {"scaling": reg, "exchange": m["mode"] if multi else None, "batch_per_gpu": m["B"],
                    "global_batch": m["B"] * world, "ms_per_step": round(m["dt"] / m["steps"] * 1e3, 4),
                    "value": round(m["steps"] * m["B"] * world / m["dt"], 1),
                    "host_enqueue_ms_per_step": round(m["t_host"] / m["steps"] * 1e3, 4),
                    "allreduce_bytes_per_step": m.get("exchange_bytes"),
                    "rows_forward_per_rank": m["B"], "rows_backward_per_rank": rows_b,
                    "table_rows_updated_per_rank": model.T,
                    "per_rank_work": PARALLELISM[m["mode"] if multi else "single"]}
        PARALLELISM = {
            "single": "one rank: the whole batch",
            "stats": "forward + statistics backward row-sharded (B/N rows per rank); Adam replicated (every rank updates all table rows)",
            "grads": "forward + gradient backward row-sharded (B/N rows per rank); Adam replicated (every rank updates all table rows)",
            "rows": "forward row-sharded (B/N rows per rank); backward + Adam REPLICATED (every rank walks all B rows of the global "
                    "batch and updates all table rows)"}
        head_reg = regimes[0][0]
        cand_lines = {reg: {k: (line_of(v, reg) if "dt" in v else {"skipped": v["skipped"]}) for k, v in rs.items()}
                      for reg, rs in results.items()}
        value = args.steps * B * world / dt
        piped = bool(model._zrec is not None and not multi)
        out = {
            "metric": "rating-triples/sec (ELBO step) at d=%d" % d,
            "value": round(value, 1), "unit": "triples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "host_enqueue_ms_per_step": round(t_host / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": head_reg if multi else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": args.workload, "fields": F, "entities": model.T, "d": d,
                       "batch_per_gpu": B, "global_batch": B * world, "nb_train": nb_train,
                       "ids": "int32" if args.id32 else "int64",
                       "eps": "philox-in-kernel" if eps_tables is None else "tables (fixed draw, A/B)",
                       "row_source_order": "sorted by user (the data files' order)" if args.user_order else "as generated", "id_distribution": ("zipf(%.2f) items" % args.zipf) if args.zipf > 0 else "uniform",
                       "likelihood": output, "unique_entities_per_batch": round(U, 1),
                       "row_order": "as-generated" if args.no_sort else "sorted by last id column inside each batch",
                       "adam_moments": "scaled form (untouched rows do not write m, v)" if (
                           model.scaled_moments and not multi and not args.unfused) else "plain",
                       "lazy_exact_adam": {"list": "rows of the batch only + replay pass", "la": "look-ahead (this batch + next batch)",
                                           None: False}[lazy_kind],
                       "pipelined_step": piped, "packed_first_order_records": bool(model.use_wrec and not multi and not args.unfused),
                       "plans": ("built inside the timed region, a few steps ahead on side streams" if (not multi and args.plans == "stream")
                                 else "resident: built once per batch before the timed region, reused every epoch (vfm-torch.py:121-122: no shuffling)"),
                       "exchange": model.exchange if multi else None, "exchange_note": exchange_note,
                       "step": ({"stats": "fwd+loss+bwd_acc+allreduce(stats)+apply_adam",
                                 "grads": "fwd+loss+bwd+allreduce(grads)+dense-adam",
                                 "rows": "sample(records of all the batch's entities)+fwd(own rows)+allreduce(row gradients: B_global+8 doubles)"
                                         "+fused(loss+bwd+dense-adam) over the whole batch on every rank"}[model.exchange]
                                if multi else "fwd+loss+bwd+dense-adam") if
                               (multi or args.unfused) else ("fwd(records)+loss+fused(bwd+dense-adam+next batch's sampling)" if
                                                                 piped else "fwd+loss+fused(bwd+dense-adam)"),
                       "parallelism": (f"x{world}, tables replicated, one all-reduce per step: forward row-sharded ({B} of {B * world} rows per "
                                       f"rank); backward + Adam replicated (all {B * world} rows and all table rows on every rank)")
                       if (multi and model.exchange == "rows") else
                       f"row-sharded dp{world}: {B} of {B * world} rows per rank in forward and backward, tables replicated (Adam over all "
                       f"table rows on every rank), one all-reduce per step"},
            "roofline": roof, "kernels": kern, "kernel_events_on_every_nth_step": max(1, args.event_every), "regions": regions,
            "sustained": sustained, "streamed": streamed,
            "ms_per_step_streamed": streamed.get("ms_per_step") if streamed else None,
            "settle": None if not head.get("cold") else {
                "untimed_steps_before_warmup": head["cold"]["settle_steps"],
                "cold_ms_per_step": round(head["cold"]["ms_per_step"], 4), "cold_steps": head["cold"]["steps"],
                "note": "the same warm-up + region timed first, WITHOUT the settling steps (the chip reaches the steady "
                        "duration of the backward kernel 10-50 ms of this work after an idle stretch: tools/transient_probe.py)"},
            "candidates": cand_lines if multi else None,
            "weak": (lambda w: line_of(min(w, key=lambda v: v["dt"]), "weak") if w else None)(
                [v for v in results.get("weak", {}).values() if "dt" in v]) if multi and head_reg != "weak" else None,
            "plan_build_ms_per_batch": {k: round(v, 4) for k, v in setup.plan_build.items()},
            "plan_build_rebuilt_ms_per_batch": {k: round(v, 4) for k, v in setup.plan_build_warm.items()},
            "plan_build_amortised_us_per_step_at_50_epochs": {k: round(v * 1e3 / 50, 3) for k, v in setup.plan_build.items()},
            "comm_check": comm_check, "forced_single_rank_group": forced or None,
            "cpu_baseline": cpu, "final_loss": loss, "nan_params": nan_params,
        }
        print(json.dumps(out), file=json_out, flush=True)
    doing[0] = "destroy_process_group"
    if multi:
        dist.destroy_process_group()


def kernel_report(events, model, args, world, B, d, F, U, lazy_kind=None):
    """Per-kernel averages of the HIP events recorded inside the timed region, each with its ALGORITHMIC bytes
     and the resulting GB/s, and the `roofline` object of the kernel that takes most time."""
    kern = {}
    roof = None
    if events:
        acc = {}                                  # marks in launch order: start, fwd, finalize, ...
        dump = os.environ.get("VFM_BENCH_DUMP_EVENTS")       # debug: every sampled step's durations, one line each
        lines = []
        for ev in events:
            order = list(ev.keys())
            row = {}
            for a_, b_ in zip(order[:-1], order[1:]):
                row[b_] = ev[a_].elapsed_time(ev[b_])
                acc[b_] = acc.get(b_, 0.0) + row[b_]
            lines.append(" ".join(f"{k} {v * 1e3:.1f}" for k, v in row.items()))
        if dump:
            open(dump, "w").write("\n".join(lines) + "\n")
        n_params = model._n_flat
        # ALGORITHMIC bytes per launch (fp32 params, ids as given, eps in-kernel):
        idb = 4 if args.id32 else 8
        d_k, B_k = d, B
        rows_mode = world > 1 and model.exchange == "rows"
        if rows_mode:         # per rank: own rows in the forward, ALL rows (and their entities: U_all) in the sampling pass and the backward
            U_all = sum(p_.__dict__["_gplan"][1].U for p_ in getattr(args, "_plans", []) if "_gplan" in p_.__dict__) / max(1, len(getattr(args, "_plans", []))) or U
        bytes_fwd = U * (8 * d_k + 16) + B_k * (idb * F + 8)        # touched rows once + ids, y, pred
        bytes_bwd = bytes_fwd + U * (8 * d_k + 8)                   # re-read + one write per touched row
        look = lazy_kind == "la"  # look-ahead lazy exact Adam: rows in neither this nor the next batch are skipped
        lazy = "catchup" in acc and not look   # lazy exact dense Adam: only the batch's rows are read / written per step
        piped = model._zrec is not None and world == 1 and "sample_rec" not in acc     # software-pipelined step (steady state)
        if piped:
            # forward = gather of sample records: each touched record once (4d + 16 B) + ids, y, pred, grow
            bytes_fwd = U * (4 * d_k + 16) + B_k * (idb * F + 12)
        alg = {"fwd": bytes_fwd, "bwd": bytes_bwd, "adam": 28.0 * n_params, "finalize": 0.0,
               # replay of the skipped updates on the batch's rows: read p, m, v, write p
               "catchup": 16.0 * U * (2 * d_k + 2) + 8.0 * U,
               "allreduce": 4.0 * n_params,
               # fused backward+Adam: gradients stay on chip -> per-row inputs + Adam state traffic
               # (scaled moments: the rows a batch does not touch read p, m, v and write p only: 16 B/param)
               "sample_rec": U * (8 * d_k + 16) + U * (4 * d_k + 16),
               # look-ahead: rows of the batch 24 B/parameter, rows of the next batch only 16, the others nothing
               "bwd_adam": (B_k * (idb * F + 8) + U * 16 + (2 * d_k + 2) * (24.0 * U + 16.0 * U * (1.0 - U / model.T)))
               if look else (B_k * (idb * F + 8) + U * 16 + 24.0 * U * (2 * d_k + 2)) if lazy else (
                           B_k * (idb * F + 8) + U * 16 + 24.0 * n_params -
                           (8.0 * (model.T - U) * (2 * d_k + 2) if (model.scaled_moments and not model.sparse_adam) else 0.0)
                           # pipelined: + the next batch's records written, the other entities' records read once
                           + (2.0 * U * (4 * d_k + 16) if piped else 0.0)),
               # staged multi-rank form: statistics [T,d+2] written / all-reduced / read + Adam state
               "bwd_acc": B * F * (4 * d + 8) + 4.0 * model.T * (d + 2),
               # all-reduce of the statistics overlapped with the epilogue + dense Adam kernels
               "exchange_apply_adam": 4.0 * model.T * (d + 2) + 24.0 * n_params,
               }
        alg["allreduce"] = 4.0 * n_params
        if rows_mode:
            Bg = args._plans[0].B_global
            alg["fwd"] = U * (4 * d + 16) + B * (idb * F + 12)                      # the own rows' records once + ids, y, pred, grow
            alg["sample_rec"] = U_all * (8 * d + 16) + U_all * (4 * d + 16)
            alg["exchange"] = 8.0 * (Bg + 8)
            alg["bwd_adam"] = (Bg * (idb * F + 8) + U_all * 16 + 24.0 * n_params
                               - (8.0 * (model.T - U_all) * (2 * d + 2) if model.scaled_moments else 0.0) + U_all * (4 * d + 16))
        forced = os.environ.get("VFM_FWD_KERNEL", "0")        # (use_fwd2 in csrc/vfm_abi.hip: k_fwd2 from d = 20 on)
        fwd2 = F == 2 and d % 4 == 0 and d <= 512 and model.n_samples == 1 and model.link == "abs" and \
            forced != "1" and (d >= 20 or forced == "2")
        names = {"fwd": ("k_fwd2<ZREC> (pipelined step: gather of this step's sample records -> FM -> ELBO; the records were "
                         "written by the previous step's fused backward)") if piped else
                        ("k_fwd2 (task stream: gather->reparam->FM->ELBO, a repeated id of the sorted column sampled "
                         "once per run)" if fwd2 else "k_fwd (gather->reparam->FM->ELBO)"), "bwd": "k_bwd (entity-centric gradients)",
                 "bwd_adam": "k_bwd<ADAM> (gradients + dense Adam fused)" + (", rows of the batch only (lazy exact Adam)" if lazy else "")
                             + (" + sampling of the next batch's entities (pipelined step)" if piped else "")
                             + (", look-ahead lazy exact form (rows in neither this batch nor the next are skipped)" if look else ""),
                 "adam": "k_adam (dense Adam)",
                 "sample_rec": "k_sample_rec (sample records of a batch from the tables: first step of a pipelined run)",
                 "catchup": "k_adam_catchup (lazy exact dense Adam: replay of the skipped zero-gradient updates on the batch's rows)",
                 "finalize": "k_finalize", "allreduce": "RCCL all-reduce of the flat exchange buffer",
                 "exchange": "RCCL all-reduce of every row's dloss/dpred + the six ELBO sums (B_global + 8 doubles), with its packing",
                 "bwd_acc": "k_bwd<ACC> (gradient statistics of the shard)",
                 "exchange_apply_adam": "RCCL all-reduce of the statistics, chunk-overlapped with "
                                        "k_bwd<APPLY,ADAM> (epilogue + dense Adam)"}
        if rows_mode:
            names["fwd"] = "k_fwd2<ZREC> (the rank's own rows: gather of the sample records -> FM -> ELBO)"
            names["sample_rec"] = "k_sample_rec (records of ALL entities of the global batch, from this rank's replica of the tables)"
            names["bwd_adam"] = "k_bwd<ADAM, records> (loss + gradients + dense Adam over the WHOLE batch, on every rank)"
        for k, ms in acc.items():
            us = ms / len(events) * 1e3
            gbs = alg[k] / (us * 1e-6) / 1e9 if us > 0 else 0.0
            kern[k] = {"kernel": names[k], "avg_us": round(us, 2), "alg_bytes": int(alg[k]),
                       "achieved_GBs": round(gbs, 1), "frac_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
        cand = [k for k in kern if k in ("fwd", "bwd", "bwd_adam", "bwd_acc", "adam", "catchup")]
        dom = max(cand, key=lambda k: kern[k]["avg_us"])
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "latest_traffic.json")
        same_cmd = (args.workload == "ml20m_d128" and not args.dim and B == 100000 and not args.no_sort and world == 1 and not piped
                    and look and not getattr(args, "zipf", None) and args.fwd_eps == "philox"
                    and not os.environ.get("VFM_FWD_KERNEL") and not os.environ.get("VFM_FWD_AB_NORNG"))
        if os.path.exists(tpath) and same_cmd:
            # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
            # (counters cannot be read from inside the process; see profiles/README.md)
            tj = json.load(open(tpath))
            from vae_amd.build import sources_digest
            if tj.get("_csrc_sha1") not in (None, sources_digest()):
                tj = {}          # the kernels changed since those passes: no figure rather than a stale one
                traffic_src = "profiles/latest_traffic.json is older than vae_amd/csrc (re-run tools/profile_round.sh)"
            if dom in tj:
                traffic, traffic_src = tj[dom]["hbm_bytes_per_launch"], "profiles/latest_traffic.json"
            for k in kern:
                if k in tj:
                    kern[k]["hbm_bytes_pmc"] = tj[k]["hbm_bytes_per_launch"]
        def on_traffic(k):       # fraction of peak on the bytes the PMC passes counted (not the algorithmic ones)
            t = kern[k].get("hbm_bytes_pmc")
            return round(t / (kern[k]["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if t and kern[k]["avg_us"] > 0 else None
        roof = {"kernel": kern[dom]["kernel"], "bound": "hbm", "achieved": kern[dom]["achieved_GBs"],
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kern[dom]["frac_hbm_peak"], "traffic": traffic,
                "traffic_source": traffic_src,
                "avg_us": kern[dom]["avg_us"], "alg_bytes": kern[dom]["alg_bytes"],
                "frac_on_traffic": on_traffic(dom),
                "elbo_fwd_kernel": {"achieved": kern["fwd"]["achieved_GBs"], "frac": kern["fwd"]["frac_hbm_peak"],
                                    "avg_us": kern["fwd"]["avg_us"], "alg_bytes": kern["fwd"]["alg_bytes"],
                                    "traffic": kern["fwd"].get("hbm_bytes_pmc"), "frac_on_traffic": on_traffic("fwd")}}
    return kern, roof


def stream_copy_rate(dev):
    """This box's streaming rate (device-to-device copy of 1 GiB): an honest denominator next to the 8 TB/s
    spec peak."""
    import torch
    src = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    dst = torch.empty_like(src)
    dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10):
        dst.copy_(src)
    e0.record()
    for _ in range(10):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    copy_gbs = 10 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst
    return copy_gbs


def cpu_baseline(sizes, d, B, nb_train, output, plan, occ, budget_s):
    """Time the oracle's reference-shaped step (same op graph as vfm-torch.py:189-324,359,368-370, INCLUDING the two
    unused per-row lookups of :204-205 the reference pays for) on the host cores.  torch's intra-op threading does not scale to all logical CPUs of the node for this op mix, so the
    thread count is swept (one warm-up + two steps each) and the best count is then timed for the rest of the
    budget: the reported value is the best the host does, with the count stated."""
    import torch
    from oracle import vfm_oracle as O
    if len(sizes) != 2:
        return None
    N, M = sizes
    x = plan.x.to(torch.int64).cpu()
    y = plan.y.cpu()
    occ_c = occ.cpu()
    ncpu = os.cpu_count() or 1
    t_start = time.perf_counter()

    def fresh():
        torch.manual_seed(42)
        P = O.make_params(N + M, d)
        return P, torch.optim.Adam(list(P.values()), lr=1.0 / (1 + nb_train // B))

    def time_steps(P, opt, n_max, budget):
        n, t0 = 0, time.perf_counter()
        while True:
            O.reference_shaped_step(P, opt, x, y, occ_c, N, M, nb_train, output, dead_gathers=True)
            n += 1
            el = time.perf_counter() - t0
            if el >= budget or n >= n_max:
                return n, el

    keep = torch.get_num_threads()
    sweep = {}
    for nt in sorted({t for t in (8, 16, 32, 64, ncpu // 2, ncpu) if 1 <= t <= ncpu}):
        if time.perf_counter() - t_start > 0.6 * budget_s and sweep:
            break
        torch.set_num_threads(nt)
        P, opt = fresh()
        O.reference_shaped_step(P, opt, x, y, occ_c, N, M, nb_train, output, dead_gathers=True)       # warm-up
        n, el = time_steps(P, opt, 2, 1e9)
        sweep[nt] = round(n * B / el, 1)
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    P, opt = fresh()
    O.reference_shaped_step(P, opt, x, y, occ_c, N, M, nb_train, output, dead_gathers=True)           # warm-up
    n, el = time_steps(P, opt, 50, max(2.0, budget_s - (time.perf_counter() - t_start)))
    torch.set_num_threads(keep)
    return {"value": round(n * B / el, 1), "unit": "triples/s", "cores": best,
            "kind": "port", "ms_per_step": round(el / n * 1e3, 2),
            "thread_sweep_triples_per_s": {str(k): v for k, v in sweep.items()},
            "sample": f"{n} full steps (fwd+loss+bwd+Adam) of the same batch shape B={B}, d={d}, "
                      f"T={N + M} with the torch-CPU reference-shaped restatement (oracle/vfm_oracle.py) on the "
                      f"best thread count of a sweep (1 warm-up + 2 steps per count); host has {ncpu} logical CPUs"}


if __name__ == "__main__":
    main()
