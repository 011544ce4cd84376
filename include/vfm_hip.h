/*
 * vfm_hip.h -- C ABI of libvfm_hip.so: the MI355X (gfx950) Variational-FM ELBO step.
 *
 * Drop-in boundary.  The reference (jilljenn/vae) has no FFI / operator API; the one
 * boundary of its hot path is the Python call `CF.forward(x)` (vfm-torch.py:189) whose
 * results are combined into the loss at vfm-torch.py:359 and differentiated at :368-369.
 * Every entry point below replaces a slice of that call; the slice is cited per function.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer except `vfm_problem_t*` is DEVICE memory
 *    owned by the caller; the library never allocates, frees or keeps a pointer past return.
 *  - launch-only: no host synchronisation inside, safe under hipGraph capture, re-entrant
 *    (the only state is a thread-local error string -- and, per device, ONE side stream with two events,
 *    created the first time a fused backward meets an index with heavy lists in a large table: the
 *    pre-reduction of those lists then runs beside the main kernel, fork / join by events on the caller's
 *    stream -- that enqueue is serialised per device by a mutex, so host threads driving different streams of one
 *    device may call concurrently; env VFM_HEAVY_OVERLAP=0 keeps everything on the caller's stream).
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *  - return 0 on success, otherwise a negative VFM_E_* code or a positive hipError_t;
 *    `vfm_last_error()` describes the last failure on the calling thread.
 *  - all floating point is fp32 (as the reference), reductions accumulate in fp64.
 *
 * Data layout in HBM (identical to the reference's nn.Embedding tables, so state_dicts
 * line up, vfm-torch.py:152-153):
 *    entity_params [T, 2d] row-major: row e = [ mu_v(0..d-1) | s_v(0..d-1) ]  (8d bytes)
 *    bias_params   [T, 2]           : row e = [ mu_w, s_w ]
 *    scalars       [3]              : alpha, global_bias_mean, global_bias_scale
 *    x             [B, F] row-major entity ids, int64 (reference: torch.LongTensor) or int32
 *    link function of the scale parameters: sigma = |s| (vfm-torch.py:126, the assignment that wins)
 *    or, with VFM_FLAG_LINK_SOFTPLUS, sigma = softplus(s) (:125) -- the reference's global LINK.
 */
#ifndef VFM_HIP_H
#define VFM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version 5.  Every struct that crosses the boundary by pointer (vfm_problem_t, vfm_index_t, vfm_pipe_t) starts with
 * (struct_size, abi_version): the caller sets them to sizeof(the struct it was COMPILED against) and VFM_ABI_VERSION
 * (VFM_STRUCT_INIT does both after zeroing the struct), and every entry point that takes the struct returns VFM_E_INVALID
 * when they differ from the library's own -- a binding built against another layout is refused instead of being read past
 * its end (version 4 grew vfm_index_t by a field without a version bump; a 72-byte caller struct was then over-read). */
#define VFM_ABI_VERSION 5
#define VFM_MAX_FIELDS 64
#define VFM_STRUCT_INIT(s) do { memset(&(s), 0, sizeof(s)); (s).struct_size = (uint32_t)sizeof(s); \
                                (s).abi_version = VFM_ABI_VERSION; } while (0)      /* (needs <string.h>) */

#define VFM_E_INVALID (-1)   /* bad argument (shape, null pointer, unsupported size)   */
#define VFM_E_UNSUPPORTED (-2)

#define VFM_LIK_NORMAL 0     /* 'reg'  : Normal(pred, 1/sqrt|alpha|)  vfm-torch.py:268  */
#define VFM_LIK_BERNOULLI 1  /* 'class': Bernoulli(logits=pred)       vfm-torch.py:270  */

/* vfm_problem_t.flags */
#define VFM_FLAG_NO_PRIOR_TERMS 1 /* this rank leaves out the terms that do not depend on batch
                                     rows -- KL(q(w0)||N(0,1)) in the loss and its gradient wrt
                                     global_bias_mean/scale -- so that SUMMING loss and scalar
                                     gradients over ranks counts them exactly once                */
#define VFM_FLAG_SPARSE_ADAM 4    /* vfm_elbo_bwd_adam_f32 only, OPT-IN, changes results: rows that are not
                                     in the batch are skipped (no momentum drift), unlike the reference's
                                     dense Adam (vfm-torch.py:339); bias corrections use the global step   */
#define VFM_FLAG_SCALED_MOMENTS 32 /* vfm_elbo_bwd_adam_f32 / vfm_elbo_apply_adam_f32: the moment buffers hold
                                     m / beta1^k and v / beta2^k, k = t mod VFM_MOMENT_PERIOD after Adam step t
                                     (k = 0: the plain moments).  The decay of a row without gradient is then
                                     implicit, so the rows a batch does not touch read their moments but do not
                                     write them back: 16 instead of 24 bytes per parameter.  Same dense Adam
                                     (every row moves every step); results equal the plain form up to fp32
                                     rounding.  Every VFM_MOMENT_PERIOD-th step writes the plain moments for all
                                     rows.  vfm_moments_rescale_f32 converts a buffer between the two forms.   */
#define VFM_MOMENT_PERIOD 128
#define VFM_FLAG_ROWS_TOUCHED 128  /* vfm_elbo_bwd_adam_f32: handle ONLY the table rows the batch contains (+ the three scalars,
                                     the loss) -- with vfm_index_t.touched_ids as a list: the row-list step of the lazy exact
                                     Adam (the other rows are replayed later by vfm_adam_catchup_f32)            */
#define VFM_FLAG_ZREC 1024        /* vfm_elbo_fwd_f32 (two fields, one sample, d % 4 == 0, d <= 512, training): `entity_params`
                                     holds this step's SAMPLES, one record (w, weighted KL, 0, 0 | z[0..d-1]) of 4 + d
                                     floats per ENTITY ID (vfm_sample_records_f32 / vfm_elbo_bwd_adam_pipe_f32 write them);
                                     x holds entity ids as usual.  Nothing is sampled and no sumz is written (the
                                     pipelined backward gathers the samples themselves): bias_params, inv_occ, W, sumz
                                     may be NULL.  Half the gather bytes of the (mu | s) rows, no RNG.  |.| link only. */
#define VFM_FLAG_SHARE_GPU 2048   /* the caller runs other work BESIDE this step -- the next batches' index builds on a side stream
                                     (vfm_build_index) -- so the two-field forward and the fused backward launch 15/16 of the
                                     workgroups the chip can hold instead of filling every slot: small kernels of the other
                                     stream then find room at once (ML-20M shape, plans built inside the loop: 0.245 -> 0.229 ms
                                     per step; with nothing beside it the step itself is 2 % slower, hence a flag)            */
#define VFM_FLAG_EPS_ZERO 2       /* eps = 0 everywhere: deterministic prediction from the
                                     posterior means (vfm-torch.py:248-259)                       */
#define VFM_FLAG_LINK_SOFTPLUS 16 /* LINK = softplus instead of |.| (vfm-torch.py:125-126; applies to alpha,
                                     global_bias_scale and both tables' scale halves); give it to every call
                                     of a step                                                            */

/* indices into the fp64 `partials` vector written by vfm_elbo_fwd_f32 */
#define VFM_P_LL 0      /* sum_r log p(y_r | pred_r)                                    */
#define VFM_P_KL 1      /* sum over occurrences of KL_e/occ(e) * n_g/W_g (kl_rescaled)  */
#define VFM_P_G 2       /* sum_r dloss/dpred_r                                          */
#define VFM_P_ALPHA 3   /* Normal likelihood.  In a workgroup SLOT: sum_r (y-pred)^2/2 of its rows -- positive terms only;
                           slot entry [6] holds the number of terms.  After the reduction (partials[3]):
                           sum_r (y-pred)^2/2 - n_terms/(2|alpha|), the difference formed ONCE, in fp64 -- the gradient of
                           alpha is a cancelling sum and a per-row fp32 difference lost three digits of it */
#define VFM_P_BADID 4   /* number of ids outside [0,T) met (they are clamped to 0)      */
#define VFM_P_GE0 5     /* n_samples > 1: sum_s eps0^s * sum_r dloss/dpred[s,r]         */
#define VFM_P_REDUCED 6 /* 1.0 once the slots were reduced into [0..5] (vfm_elbo_finalize_f32, or the fused
                           backward given `loss`); 0.0 after a forward.  A backward call that needs the sums
                           and finds 0.0 here produces NaN scalar gradients instead of using stale sums */
#define VFM_N_PARTIALS 8
/* `partials` is a caller-owned fp64 workspace of VFM_PARTIALS_LEN entries: [0..7] the sums
 * above (valid after vfm_elbo_finalize_f32; [7] = number of forward blocks), followed by one
 * 8-entry slot per forward workgroup (written with plain stores: deterministic sums, no atomics;
 * slot entries [0..5] as VFM_P_*, [6] = VFM_SLOT_NTERMS: Normal-likelihood terms summed into the slot's [3]) */
#define VFM_SLOT_NTERMS 6
#define VFM_MAX_FWD_BLOCKS 4096
#define VFM_PARTIALS_LEN (VFM_N_PARTIALS * (1 + VFM_MAX_FWD_BLOCKS))

/* ---- Device-resident step state: what makes a captured training step REPLAYABLE --------------------------------
 * Everything that changes from one training step to the next -- the Philox step of the eps stream, the Adam step count
 * and the constants derived from it (bias corrections, the scaled-moment factors, the learning rate) -- normally
 * arrives as host-side arguments, which a captured HIP graph would freeze.  With `vfm_problem_t.dev_step` set, the
 * forward and the fused backward+Adam entry points read all of it from DEVICE memory instead and advance it themselves
 * (the host-side `step`, `lr`, `beta*`, `seed`-independent arguments of those calls are then ignored; `p->step` too):
 *   forward   : eps stream step = dev_step->philox_step; hands (philox_step, adam_step) over to the backward
 *   backward  : constants = dev_step->tab[adam_step - tab_first]; after the last read of the step, the launch that
 *               forms the loss sets philox_step += 1, adam_step += 1 -- the next replay runs the next step.
 * So one graph captured per (batch, next batch) pair replays for every epoch (vfm-torch.py:351-370 is the loop).
 * The table is made on the HOST by vfm_step_consts (the same arithmetic the host-argument path uses: a replayed
 * trajectory is bitwise the eager one) and uploaded by the caller; `error` becomes non-zero when a step found no
 * table entry (the step then runs with the nearest entry: results are wrong, the flag says so). */
typedef struct vfm_step_consts {      /* constants of ONE Adam step (64 bytes) */
  float step_size, bc2_sqrt;          /* lr / (1 - beta1^t), sqrt(1 - beta2^t)                                   */
  float a1, q2, c1, c2, s1, s2;       /* scaled-moment form (VFM_FLAG_SCALED_MOMENTS), see csrc/vfm_args.hpp      */
  int32_t store_true;                 /* this step ends a moment period: the plain moments are written back       */
  int32_t k;                          /* position of the step in its moment period, 1..VFM_MOMENT_PERIOD          */
  int32_t scaled;                     /* the entry was made for the scaled-moment form                            */
  int32_t reserved;
  float lr, beta1, beta2, eps;
} vfm_step_consts_t;
typedef struct vfm_dev_step {         /* DEVICE memory, caller-owned, 64 bytes, 16-byte aligned */
  uint64_t philox_step;               /* eps stream step of the NEXT forward                                      */
  int64_t adam_step;                  /* 1-based count of the NEXT Adam update                                    */
  uint64_t philox_step_bwd;           /* hand-over written by the forward of the running step                     */
  int64_t adam_step_bwd;
  int64_t tab_first, tab_len;         /* tab[i] holds the constants of Adam step tab_first + i                    */
  const vfm_step_consts_t* tab;       /* DEVICE pointer                                                           */
  int64_t error;
} vfm_dev_step_t;
/* HOST helper: the constants of Adam step `step` (1-based) as every fused entry point forms them from its host
 * arguments (torch.optim.Adam's bias corrections and step size for the defaults of vfm-torch.py:339, lr of :92).  scaled != 0: for VFM_FLAG_SCALED_MOMENTS.  Returns 0, or VFM_E_* (beta^k underflow). */
int vfm_step_consts(float lr, float beta1, float beta2, float eps_adam, int64_t step, int32_t scaled,
                    vfm_step_consts_t* out);
/* Sets (philox_step, adam_step) of a device step state (one tiny launch; stream-ordered). */
int vfm_dev_step_set(vfm_dev_step_t* dev_step, uint64_t philox_step, int64_t adam_step, void* stream);

/* Problem description shared by all kernels of one step (host memory). */
typedef struct vfm_problem {
  uint32_t struct_size;   /* sizeof(vfm_problem_t) of the caller's build; checked by every entry point (VFM_STRUCT_INIT) */
  uint32_t abi_version;   /* VFM_ABI_VERSION of the caller's build                                                       */
  int64_t B;          /* rows handled by THIS call (this rank's shard of the batch)       */
  int64_t B_global;   /* rows of the whole batch over all ranks (loss uses nb_train/B_global) */
  int64_t T;          /* rows of the two tables (N + M in the reference)                 */
  int64_t nb_train;   /* vfm-torch.py:91                                                 */
  int32_t F;          /* fields per row (2 in the reference: user, item)                 */
  int32_t d;          /* embedding size                                                  */
  int32_t likelihood; /* VFM_LIK_*                                                       */
  int32_t id_bits;    /* 64 or 32: element type of x                                     */
  int32_t n_samples;  /* variational samples S in [1,64] (N_VARIATIONAL_SAMPLES, vfm-torch.py:19);
                         S > 1: forward, finalize, vfm_elbo_bwd_f32 and vfm_elbo_bwd_adam_f32 only   */
  int32_t flags;      /* VFM_FLAG_*                                                      */
  /* id groups for the KL re-weighting (vfm-torch.py:314-317): entity e belongs to the first
   * g with e < group_hi[g]; its KL is scaled by group_n[g] / W[g].  There are F groups and
   * W[f] is the normaliser of column f.  Reference: group_hi = {N+1, N+M} (the `<= N` test
   * of :316 puts item id N into the user group), group_n = {N, M}.                      */
  int64_t group_hi[VFM_MAX_FIELDS];
  double group_n[VFM_MAX_FIELDS];
  /* counter-based RNG key, used when the eps tables are NULL: eps(e,k) =
   * BoxMuller(Philox4x32-10(key = seed, ctr = (k/8, e, step))).  One draw per ENTITY per
   * step, shared by every row that contains it (vfm-torch.py:207-208,238-245).          */
  uint64_t seed;
  uint64_t step;
  /* entity range [e_lo, e_hi) handled by a backward-family call (e_hi == 0: up to T).  Lets a
   * multi-rank step cut the table in chunks and overlap the exchange of one chunk with the kernels
   * of its neighbours.  The forward ignores it. */
  int64_t e_lo, e_hi;
  /* optional device-side extensions (NULL = off) */
  vfm_dev_step_t* dev_step;   /* replayable step: see vfm_dev_step_t above.  Honoured by vfm_elbo_fwd_f32 (training, Philox
                                 eps), vfm_elbo_bwd_adam_f32, _lookahead_f32 and _pipe_f32; other entry points reject it.  */
  float* wrec;                /* [T,4] packed first-order records (mu_w, s_w, 1/occ, 0) per entity -- a CACHE of
                                 bias_params and inv_occ (vfm_wrec_build_f32 fills it).  The two-field training forward
                                 reads ONE 16-byte record per sampling task instead of an 8-byte row of bias_params and a
                                 4-byte entry of inv_occ (two cache lines for 12 bytes); every fused backward+Adam form and
                                 vfm_adam_catchup_f32 given the pointer refresh (mu_w, s_w) of the rows they update, so the
                                 cache stays coherent across fused steps.  Any other writer of bias_params invalidates it:
                                 the caller rebuilds (or passes NULL).  bias_params stays the parameter (state_dict layout
                                 of vfm-torch.py:152 unchanged).                                                          */
} vfm_problem_t;
/* wrec[e] = (bias_params[e][0], bias_params[e][1], inv_occ[e], 0): what the forward's gathers of vfm-torch.py:207
 * (`self.bias_params(uniq_entities)`) and the `nb_occ[uniq]` divisors of :298-306 read per entity, in one 16-byte record. */
int vfm_wrec_build_f32(const float* bias_params, const float* inv_occ, int64_t T, float* wrec, void* stream);

/* Inverted index of one batch (entity -> batch rows), built once per batch by the caller:
 *   occ_ptr [T+1] offsets into occ_rows [B*F] (row numbers sorted by entity id, stable).
 * Long lists (skewed data: a popular item can own 10% of the rows; small tables: every entity in dozens of
 * rows) would serialise on one lane group, so entities with more occurrences than the index's heavy-list
 * length (any length >= VFM_HEAVY_MIN works: the kernels look an entity up in `heavy_ids` whenever its list is
 * longer than VFM_HEAVY_MIN) are also listed in `heavy_ids` (sorted) and their lists cut in work items
 * `heavy_items` [n_items,4] = (slot in heavy_ids, begin, end, 0) of at most that many occurrences, in list
 * order (vfm_build_index makes all of this).  Every backward call first reduces the work items (one lane
 * group each, plain stores) and then adds each entity's items in a fixed order (entities of more than
 * VFM_HEAVY_DIRECT items: a kernel of their own; the others: inside the main kernel) -- no atomics --
 * in the scratch table `heavy_acc` [n_samples, n_heavy + n_items, 4 + round4(d)] (overwritten per call).
 * n_heavy == 0: all three may be NULL. */
#define VFM_HEAVY_LIST 64
#define VFM_HEAVY_MIN 8
#define VFM_HEAVY_DIRECT 8      /* an entity of at most this many work items is summed by the main kernel itself */
typedef struct vfm_index {
  uint32_t struct_size;   /* sizeof(vfm_index_t) of the caller's build (VFM_STRUCT_INIT) */
  uint32_t abi_version;
  const int32_t* occ_ptr;
  const int32_t* occ_rows;
  const int32_t* heavy_ids;
  const int32_t* heavy_items;
  float* heavy_acc;
  int32_t n_heavy;
  int32_t n_items;
  /* optional: the sorted ids of the entities the batch contains.  Given together with VFM_FLAG_ROWS_TOUCHED,
   * vfm_elbo_bwd_adam_f32 walks this list instead of scanning all T table rows (lazy Adam step). */
  const int32_t* touched_ids;
  int64_t n_touched;
  /* optional, two fields: occ_other[i] = the entity in the other column of row occ_rows[i] (vfm_elbo_bwd_adam_pipe_f32) */
  const int32_t* occ_other;
  /* optional: the largest number of work items ONE heavy entity has (0 = not known).  When it is known to be at most
   * VFM_HEAVY_DIRECT the main kernel adds every entity's item records itself and the k_heavy_sum launch -- which would
   * find nothing to do -- is left out (ML-100K shape: one launch of three per backward). */
  int32_t max_items;
  /* optional: what the heavy lists were built with -- the work-item length (vfm_build_index's `heavy_list`) and the
   * occurrence count above which an entity is heavy (the same, or vfm_rebuild_heavy's `threshold`); 0 = not known.  With
   * both known, a SMALL table (T (2d + 2) 4 <= 2 MB: the ML-100K shape) takes the fused dense step through ONE launch that
   * re-derives the work items from the lists themselves (one wave per table row, csrc/vfm_bwd_small.hpp) instead of the
   * pre-reduction kernels + the row kernel: bitwise the same step, 31 -> ~10 us of backward at that shape. */
  int32_t heavy_list, heavy_threshold;
  /* optional DEVICE word (e.g. vfm_build_index's counts[5]; zero it once): the kernels that walk the index BOUND what they
   * read -- a list length outside [0, B*F], a row number outside [0, B) or an entity outside [0, T) is clamped -- and add 1
   * here whenever a clamp fires, so a corrupted index shows up at the caller's next look at this word (vae_amd: end of every
   * epoch, BatchPlan.check_status) as an error instead of a hang or a fault.  NULL: clamps still apply, nothing is reported. */
  int32_t* status;
} vfm_index_t;

int vfm_abi_version(void);
const char* vfm_last_error(void);

/* Builds the inverted index of one batch on the GPU (what the reference gets from torch.unique(x,
 * return_inverse, return_counts), vfm-torch.py:190-192, restated as entity -> rows): a stable radix sort
 * of the B*F (entity id, position) pairs -- an entity's rows keep their row order, so every sum the
 * backward forms over them has a fixed order.  Launch-only, no atomics between workgroups.
 *   x         [B,F] ids (id_bits 32 or 64)
 *   ws        workspace of vfm_index_workspace_bytes(B, F, T) bytes (16-byte aligned), scratch
 *   occ_ptr   [T+1], occ_rows [B*F]                       (vfm_index_t)
 *   heavy_list  entities with MORE occurrences than this get an entry in heavy_ids (id order) and
 *             ceil(count / heavy_list) work items (slot, begin, end, 0) in heavy_items [cap_items,4];
 *             capacities that always suffice: cap_heavy = B*F / heavy_list + 1, cap_items = 2*B*F / heavy_list + 2
 *   touched_ids  NULL, or room for min(B*F, T) int32: receives the sorted ids of the entities the batch
 *             contains (the row list of the lazy Adam step, vfm_index_t.touched_ids)
 *   occ_other NULL, or (F == 2 only) [B*2] int32: for every entry of occ_rows the entity in the OTHER column of
 *             that row (vfm_index_t.occ_other: the software-pipelined step gathers that entity's sample)
 *   inv_occ, W  both NULL, or the [T] reciprocal occurrence counts and room for F doubles: the batch normalisers
 *             W[f] = sum_r inv_occ[x[r,f]] (what vfm_batch_norms computes; vfm-torch.py:305-306) come out of the same
 *             launches (the ids are read once).  Fixed summation order: bitwise reproducible.
 *   counts    [8] int32, DEVICE: (ids outside [0,T) met -- they are indexed as id 0, like the forward
 *             clamps them --, n_heavy, n_items, n_touched = entities in the batch, max_items = the most work items one
 *             heavy entity has (vfm_index_t.max_items), 0 = a zeroed word for vfm_index_t.status, 0, 0): the caller reads
 *             them back once to fill vfm_index_t
 *   counts_host  NULL, or 8 int32 of PINNED HOST memory: the library enqueues that readback itself (an asynchronous copy
 *             behind the last kernel; the caller synchronises with the stream or an event of its own before reading)
 * Launches (B*F <= 2^31 keys, T < 2^32): one memset, key extraction (+ the first pass's digit counts, W), one stable scatter
 * per ceil(log2 T / 9) radix passes (each also counts the next pass's digits and, once, the occurrences per entity), and two
 * compaction launches -- six at the ML-20M shape (was sixteen). */
int64_t vfm_index_workspace_bytes(int64_t B, int32_t F, int64_t T);
/* The heavy lists of an index once more, with a LOWER threshold than their work-item length: entities with more than
 * `threshold` occurrences (VFM_HEAVY_MIN <= threshold <= heavy_list) become heavy, their lists still cut in items of up to
 * heavy_list.  For batches with many rows per entity (rows in the data files' order: a few hundred users' consecutive
 * ratings) the walk of a 17..64-row list by ONE lane group, two occurrences in flight, is what the fused backward waits
 * for; pre-reduced, such a list is one work item with eight in flight (data-file order at the ML-20M shape: backward 103 ->
 * 94 us, step 0.130 -> 0.112 ms; with users spread over the table it costs 3 %, with 10^6 rows covering the table 10 %, so
 * the caller decides: vae_amd does it for plans with B >= 4 U and U <= T / 4 on tables of >= 8,192 rows).  occ_ptr: of the built index; ws: a workspace as for
 * vfm_build_index; counts [8]: (0, n_heavy, n_items, n_touched, max_items, 0, 0, 0) again.  Capacities that always suffice:
 * cap_heavy = B*F / threshold + 1, cap_items = B*F / heavy_list + B*F / threshold + 2.
 * vfm_heavy_threshold(heavy_list) = the threshold vae_amd uses: heavy_list / 4, at least VFM_HEAVY_MIN. */
int vfm_rebuild_heavy(int64_t T, const int32_t* occ_ptr, void* ws, int32_t heavy_list, int32_t threshold, int32_t* heavy_ids,
                      int64_t cap_heavy, int32_t* heavy_items, int64_t cap_items, int32_t* counts /*[8]*/, void* stream);
int32_t vfm_heavy_threshold(int32_t heavy_list);
/* The heavy-list length to build an index with: VFM_HEAVY_LIST when the table has at least VFM_HEAVY_UNITS rows
 * (one lane group per row already fills the chip and only the really long lists need cutting), else about
 * n_occ / VFM_HEAVY_UNITS occurrences per work item, never below VFM_HEAVY_MIN (small tables -- ML-100K shape:
 * 2,625 entities in ~60 rows each -- get their parallelism from the work items).  env VFM_HEAVY_LIST overrides. */
#define VFM_HEAVY_UNITS 8192
int32_t vfm_heavy_list_for(int64_t n_occ, int64_t T);
int vfm_build_index(int64_t B, int32_t F, int64_t T, int32_t id_bits, const void* x, void* ws, int32_t* occ_ptr,
                    int32_t* occ_rows, int32_t heavy_list, int32_t* heavy_ids, int64_t cap_heavy,
                    int32_t* heavy_items, int64_t cap_items, int32_t* touched_ids, int32_t* occ_other,
                    const float* inv_occ, double* W, int32_t* counts, int32_t* counts_host, void* stream);

/* inv_occ[e] = 1 / nb_occ[e]   (nb_occ = bincount of the training ids, vfm-torch.py:89;
 * used as `nb_occ[uniq]` divisors at :298-306,315).  Done once per training set. */
int vfm_inv_occ_f32(const int64_t* nb_occ, float* inv_occ, int64_t T, void* stream);

/* W[f] = sum_r inv_occ[x[r,f]] -- the user/item normalisers of vfm-torch.py:305-306 in
 * row-wise form.  `W` (fp64 [F]) is overwritten; the sums are bitwise reproducible (fixed order).
 * Parameter-free: depends on the
 * batch only, so it can be computed once per batch and cached over epochs.  With several
 * ranks the caller sums W over ranks before the forward call. */
int vfm_batch_norms(const vfm_problem_t* p, const void* x, const float* inv_occ, double* W,
                    void* stream);

/* Forward of the fused hot path (vfm-torch.py:189-324 without torch.unique, plus the
 * per-row terms of the loss :359): gathers, reparameterised sample z = mu + |s| * eps,
 * FM prediction  pred_r = w0 + sum_f w_f + 1/2 sum_k[(sum_f z_fk)^2 - sum_f z_fk^2],
 * log-likelihood, KL-to-prior with occurrence re-weighting.
 *  in : x [B,F], y [B] (NULL => prediction only: no likelihood / KL / training state; then
 *       inv_occ, W, sumz, grow may be NULL too.  With y, all of them are required.)
 *       W [F] batch-global normalisers, eps_* tables indexed BY ENTITY ID
 *       (eps_entity [T,d], eps_bias [T], eps_global [1]) or all three NULL => Philox.
 *  out: pred [S,B]    unscaled prediction (logit for Bernoulli) of every variational sample
 *       partials [VFM_PARTIALS_LEN] fp64 workspace: per-workgroup partial sums; call
 *       vfm_elbo_finalize_f32 to reduce them into partials[0..7] (see VFM_P_*)
 *       sumz [S,B,d] and grow [B]: training state for the backward call:
 *       sumz[s,r,k] = sum_f z^s[x_rf,k],  grow[r] = sum_s dloss/dpred[s,r].
 *  S = n_samples > 1 (vfm-torch.py:238-245,265): per sample its own eps (tables: eps_entity [S,T,d],
 *       eps_bias [S,T], eps_global [S]); the entity terms are averaged over the samples BEFORE the
 *       likelihood, the global bias is not: pred[s,r] = w0^s + mean_s'(sum_f w^s' + FM(z^s')), and the
 *       loss averages the log-likelihood over S*B (:359).  One launch per sample.
 *  eps: Philox mode draws, per entity and step, d embedding normals + 1 first-order-weight
 *       normal from Philox4x32-10(ctr = (k/8, e, step), key = seed) (see vfm_philox_eps_f32).  */
int vfm_elbo_fwd_f32(const vfm_problem_t* p, const void* x, const float* y,
                     const float* entity_params, const float* bias_params,
                     const float* inv_occ, const float* scalars, const double* W,
                     const float* eps_entity, const float* eps_bias, const float* eps_global,
                     float* pred, double* partials, float* sumz, float* grow, void* stream);

/* Reduces the per-workgroup slots of `partials` into partials[0..7] and forms
 * loss[0] = -(nb_train/B_global) * partials[LL] + KL(q(w0)||N(0,1)) + partials[KL]
 * (vfm-torch.py:322,359), loss[1] = the likelihood term, loss[2] = the KL term.  NaN when
 * partials[BADID] != 0.  MUST run after vfm_elbo_fwd_f32 and before either backward call (they
 * read partials[G], partials[ALPHA]).  With several ranks every rank finalises its own shard
 * (VFM_FLAG_NO_PRIOR_TERMS on all ranks but one) and loss / scalar gradients are summed. */
int vfm_elbo_finalize_f32(const vfm_problem_t* p, double* partials, const float* scalars,
                          float* loss, void* stream);

/* Backward (replaces autograd through vfm-torch.py:189-324,359; :368-369).  Entity-centric:
 * one lane group per table row e sums grow[r] * sumz[r,:] over the rows that contain e
 * (inverted index `idx`), then writes the DENSE gradient row (zeros for rows not in the batch, like
 * the reference's dense nn.Embedding gradients).  No atomics: bitwise reproducible for a
 * fixed index, heavy lists included.  grad_out [1] = dL/dloss (device).  g_scalars [3] = grads of alpha,
 * global_bias_mean, global_bias_scale from the (rank-summed) partials. */
int vfm_elbo_bwd_f32(const vfm_problem_t* p, const vfm_index_t* idx,
                     const float* entity_params, const float* bias_params,
                     const float* inv_occ, const float* scalars, const double* W,
                     const float* eps_entity, const float* eps_bias, const float* eps_global,
                     const float* sumz, const float* grow, const double* partials,
                     const float* grad_out, float* g_entity, float* g_bias, float* g_scalars,
                     void* stream);

/* Backward with the dense Adam update fused in (single rank): the same gradient as
 * vfm_elbo_bwd_f32 with grad_out = 1, applied at once to (param, m, v) of every table row and of
 * the three scalars with torch.optim.Adam's update (defaults of vfm-torch.py:339; `step` is the
 * 1-based update count).  The gradient never reaches HBM.  In place on entity_params, bias_params,
 * scalars and the six moment buffers (same shapes as their parameters).  Rows not in the batch
 * still move through their momentum, exactly like the reference's dense Adam.
 * `loss` (3 floats) may be NULL; when given, the launch also does the work of
 * vfm_elbo_finalize_f32 (slot reduction + loss triple), saving that launch. */
int vfm_elbo_bwd_adam_f32(const vfm_problem_t* p, const vfm_index_t* idx,
                          float* entity_params, float* bias_params, float* scalars,
                          const float* inv_occ, const double* W,
                          const float* eps_entity, const float* eps_bias, const float* eps_global,
                          const float* sumz, const float* grow, double* partials,
                          float* m_entity, float* v_entity, float* m_bias, float* v_bias,
                          float* m_scalars, float* v_scalars,
                          float lr, float beta1, float beta2, float eps_adam, int64_t step, float* loss,
                          void* stream);

/* Multi-rank form of the backward (row-sharded batch): instead of the [T,2d] gradient, the ranks
 * exchange its SUFFICIENT STATISTICS, which are sums over batch rows and half the size:
 *   acc [T, 4 + round4(d)] : one record per entity e
 *         (sum of grow_r over the shard's rows containing e, number of occurrences of e, 0, 0,
 *          A_e[0..d-1] = sum of grow_r * sumz_r over those rows)
 *   sums [2] : (sum_r grow_r, alpha term) of the shard -- from the finalised `partials`
 * vfm_elbo_bwd_acc_f32 writes them (dense, zeros for entities not in the shard).  The caller sums
 * acc and sums over ranks (one all-reduce of a flat buffer holding both, or one per entity chunk
 * [e_lo, e_hi) to overlap communication with the neighbouring chunks' kernels), then every rank
 * calls vfm_elbo_apply_adam_f32: gradient epilogue (eps regeneration, KL part, |.| link) + dense Adam
 * from the global statistics -- the replicas stay identical.  No VFM_FLAG_NO_PRIOR_TERMS here: the
 * row-independent terms are added once from the global sums. */
int vfm_elbo_bwd_acc_f32(const vfm_problem_t* p, const vfm_index_t* idx,
                         const float* sumz, const float* grow, const double* partials, float* acc,
                         float* sums, void* stream);
int vfm_elbo_apply_adam_f32(const vfm_problem_t* p, const float* acc, const float* sums,
                            float* entity_params, float* bias_params, float* scalars, const float* inv_occ,
                            const double* W, const float* eps_entity, const float* eps_bias,
                            const float* eps_global, float* m_entity, float* v_entity, float* m_bias,
                            float* v_bias, float* m_scalars, float* v_scalars, float lr, float beta1,
                            float beta2, float eps_adam, int64_t step, void* stream);
/* The apply stage over a LIST of rows -- the multi-rank step's lazy exact dense Adam (replaces, per rank of a
 * data-parallel run, `optimizer.step()` of vfm-torch.py:370 on the gradients of :368-369 summed over ranks; the
 * reference itself is single-process).  With the statistics exchange
 * compacted to the entities some rank's shard contains (the caller knows that set: it is what it all-reduced), every
 * other row of the table has a zero gradient on EVERY rank this step, so -- as in the single-rank lazy forms -- its
 * update can wait: the caller (1) replays what the listed rows skipped with vfm_adam_catchup_f32 (mark = this step)
 * BEFORE the forward reads them, (2) calls this with the same list after the all-reduce: gradient epilogue + Adam on
 * the listed rows only (sorted ids; compact_records == 0: their records at acc + id * (4 + round4(d)) of the DENSE
 * statistics table; != 0: acc is a COMPACT buffer, record i belongs to row_ids[i] -- what vfm_elbo_bwd_acc_rows_f32
 * writes and the compacted exchange all-reduces, so no dense table is involved at all), and
 * (3) on the last step of every moment period brings all rows up to date and runs vfm_elbo_apply_adam_f32.  Identical
 * lists on every rank keep the replicas bit-identical; the trajectory is the dense one (same kernel instance, same
 * arithmetic per row).  VFM_FLAG_SCALED_MOMENTS required; Philox eps; the launch whose p->e_hi is 0 or T also moves
 * the three scalars (give the other launches of a chunked run any 0 < e_hi < T). */
int vfm_elbo_bwd_acc_rows_f32(const vfm_problem_t* p, const vfm_index_t* idx, const int32_t* row_ids, int64_t n_rows,
                              const float* sumz, const float* grow, const double* partials, float* acc,
                              float* sums, void* stream);      /* statistics of the listed rows, record i <-> row_ids[i] */
int vfm_elbo_apply_adam_rows_f32(const vfm_problem_t* p, const float* acc, const float* sums, const int32_t* row_ids,
                                 int64_t n_rows, int32_t compact_records, float* entity_params, float* bias_params, float* scalars,
                                 const float* inv_occ, const double* W, float* m_entity, float* v_entity, float* m_bias,
                                 float* v_bias, float* m_scalars, float* v_scalars, float lr, float beta1, float beta2,
                                 float eps_adam, int64_t step, void* stream);
/* Dense Adam step, torch.optim.Adam defaults and single-tensor op order (vfm-torch.py:339,370:
 * betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad).  `step` is the 1-based count of
 * this update (bias corrections are formed on the host in fp64).  In place on p, m, v; all four
 * pointers 16-byte aligned. */
int vfm_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                 float beta2, float eps, int64_t step, void* stream);

/* Software-pipelined step (two fields, one sample, single rank, Philox eps).  The forward of a step needs, per
 * entity of its batch, one sample z = mu + sigma*eps, the sampled first-order weight and the entity's KL -- all
 * functions of the entity's OWN row.  The fused backward of the PREVIOUS step has every updated row in registers
 * anyway (dense Adam), so it writes these records for the rows of the next batch there (51 MB at ML-20M shape
 * instead of the forward re-reading 130 MB of table rows and drawing 16 M normals), and the forward shrinks to a
 * gather of records (VFM_FLAG_ZREC).  The backward in turn gathers the OTHER entity's sample from the records
 * instead of the per-row `sumz` the plain forward writes (no sumz at all).  Same draws (Philox counters of the
 * step the record is for), same arithmetic: the trajectory equals the plain step's up to summation order.
 *   zrec, zrec_next  [T, 4 + d] fp32 record tables, indexed by entity id (double buffer, caller-owned)
 *   vfm_sample_records_f32   writes the records of the listed entities from the tables (first step of a run):
 *                            p->step = the step the records are for; W = that batch's normalisers
 *   vfm_elbo_bwd_adam_pipe_f32  = vfm_elbo_bwd_adam_f32 (plain dense step; no row flags) with sumz replaced by
 *                            pipe->zrec + idx->occ_other, and -- when pipe->zrec_next != NULL -- the records of
 *                            step pipe->next_step for every entity e with next_occ_ptr[e+1] != next_occ_ptr[e]
 *                            (the next batch's inverted-index offsets), weighted with next_W.
 * Look-ahead form (pipe->last_step != NULL; VFM_FLAG_SCALED_MOMENTS; not on the last step of a moment period): as
 * vfm_elbo_bwd_adam_lookahead_f32 -- a row in neither this batch nor the next (next_occ_ptr, required) is skipped, a
 * visited row first replays the zero-gradient updates it skipped (step_tab: the period's per-step constants, kept by the
 * kernel) -- with idx->touched_ids, if given, as the list of rows to visit.  The rows of the next batch are then current
 * when their records are written, which is all the next forward reads. */
typedef struct vfm_pipe {
  uint32_t struct_size;   /* sizeof(vfm_pipe_t) of the caller's build (VFM_STRUCT_INIT) */
  uint32_t abi_version;
  const float* zrec;
  float* zrec_next;
  const int32_t* next_occ_ptr;
  const double* next_W;
  uint64_t next_step;
  int32_t* last_step;           /* [T] last Adam step applied to each row; NULL = every row every step */
  float* step_tab;              /* [2 * (VFM_MOMENT_PERIOD + 1)] floats, as for vfm_elbo_bwd_adam_lookahead_f32 */
} vfm_pipe_t;
int vfm_sample_records_f32(const vfm_problem_t* p, const int32_t* ids, int64_t n, const float* entity_params,
                           const float* bias_params, const float* inv_occ, const double* W, float* zrec, void* stream);
int vfm_elbo_bwd_adam_pipe_f32(const vfm_problem_t* p, const vfm_index_t* idx, const vfm_pipe_t* pipe,
                               float* entity_params, float* bias_params, float* scalars,
                               const float* inv_occ, const double* W, const float* grow, double* partials,
                               float* m_entity, float* v_entity, float* m_bias, float* v_bias,
                               float* m_scalars, float* v_scalars,
                               float lr, float beta1, float beta2, float eps_adam, int64_t step, float* loss,
                               void* stream);

/* Lazy EXACT dense Adam for sparse-touch regimes (Criteo shape: a batch touches 6 % of the table).  With
 * VFM_FLAG_SCALED_MOMENTS a row without gradient keeps its stored moments and its parameters move by a function
 * of (p, m, v) and the step's constants alone -- so the caller may SKIP such rows (vfm_elbo_bwd_adam_f32 with
 * VFM_FLAG_ROWS_TOUCHED) and replay the skipped zero-gradient updates later with this call, bitwise as the
 * dense kernel would have applied them (same fp32 operations, same order):
 *   for every listed row e (ids [n] int32, or NULL = all T rows, n == T): apply the updates of Adam steps
 *   last_step[e]+1 .. upto, then last_step[e] = mark (>= upto: `mark = upto + 1` when the caller is about to
 *   apply step upto+1 to exactly these rows).
 * lr_of_step (HOST memory, n_lr entries): the learning rate of the 1st, 2nd, ... step of the moment period containing
 * `upto` (steps P+1 .. upto, P = floor((upto-1) / VFM_MOMENT_PERIOD) * VFM_MOMENT_PERIOD); n_lr >= upto - P, fewer is
 * VFM_E_INVALID (the table is never read past n_lr).  wrec: NULL, or the packed first-order records to refresh
 * (vfm_problem_t.wrec).  A row must not lag across
 * a period boundary: at the last step of every period (step % VFM_MOMENT_PERIOD == 0) the caller brings ALL
 * rows up to date and runs the dense call (which writes the plain moments for all rows).
 * Call it on a batch's rows before the forward of a step, and on all rows before predicting / saving. */
int vfm_adam_catchup_f32(float* entity_params, float* bias_params, const float* m_entity, const float* v_entity,
                         const float* m_bias, const float* v_bias, int32_t* last_step, const int32_t* ids, int64_t n,
                         int64_t T, int32_t d, const float* lr_of_step, int64_t n_lr, float beta1, float beta2,
                         float eps_adam, int64_t upto, int64_t mark, float* wrec, void* stream);

/* Look-ahead form of the lazy exact Adam (mid-range touch fractions, e.g. ML-20M shape at B = 100 K: 59 % of the
 * rows per batch): the fused dense step visits a row only if it is in THIS batch (gradient update) or in the NEXT one
 * (`next_occ_ptr`: that batch's inverted-index offsets; the row replays whatever it skipped plus this step's
 * zero-gradient update, so the next forward reads current parameters) -- a row in neither waits.  With independent
 * batches touching a share f of the rows each, (1-f)^2 of the table is not read or written at all.  No separate
 * catch-up pass: the replay runs inside the kernel, constants of the period's steps in `step_tab`
 * (2 * (VFM_MOMENT_PERIOD + 1) floats, device; every look-ahead call records its own step there).  Bitwise the dense
 * trajectory.  Requirements: the rows of THIS batch are up to date through step-1 (they are if the previous step was a
 * look-ahead call naming this batch, or after vfm_adam_catchup_f32 on them); `last_step` as in vfm_adam_catchup_f32;
 * not for the last step of a moment period (step % VFM_MOMENT_PERIOD == 0: bring all rows up to date and run the dense
 * call).  With idx->touched_ids / n_touched set to the list vfm_union_rows makes of the two batches, the kernel walks
 * that list instead of classifying all T table rows (every lane group then has work in every iteration).  Same
 * arguments as vfm_elbo_bwd_adam_f32 otherwise (Philox eps). */
/* rows [count] = the sorted ids of the entities that occur in batch A or in batch B (their inverted indexes' occ_ptr
 * arrays), count[0] (DEVICE int32) their number; ws = vfm_union_workspace_bytes(T) bytes of scratch.  Room for
 * min(T, n_occ_A + n_occ_B) ids always suffices.  Launch-only, no atomics: two calls give the same list.  count_host: NULL, or one int32 of
 * PINNED HOST memory the count is copied to asynchronously (as vfm_build_index's counts_host). */
int64_t vfm_union_workspace_bytes(int64_t T);
int vfm_union_rows(int64_t T, const int32_t* occ_ptr_a, const int32_t* occ_ptr_b, void* ws, int32_t* rows, int32_t* count,
                   int32_t* count_host, void* stream);
int vfm_elbo_bwd_adam_lookahead_f32(const vfm_problem_t* p, const vfm_index_t* idx,
                                    float* entity_params, float* bias_params, float* scalars,
                                    const float* inv_occ, const double* W,
                                    const float* sumz, const float* grow, double* partials,
                                    float* m_entity, float* v_entity, float* m_bias, float* v_bias,
                                    float* m_scalars, float* v_scalars,
                                    float lr, float beta1, float beta2, float eps_adam, int64_t step, float* loss,
                                    int32_t* last_step, const int32_t* next_occ_ptr, float* step_tab, void* stream);

/* Convert Adam moment buffers (n floats each) between the plain form and the scaled form of
 * VFM_FLAG_SCALED_MOMENTS, `step` = number of Adam steps applied so far: to_scaled != 0 divides m by
 * beta1^(step mod VFM_MOMENT_PERIOD) and v by beta2^(...), to_scaled == 0 multiplies. */
int vfm_moments_rescale_f32(float* m, float* v, int64_t n, float beta1, float beta2, int64_t step, int32_t to_scaled,
                            void* stream);

/* ELBO variants of the reference's sibling scripts, as one general (any F, d <= 1024, single rank, |.| link,
 * one sample) forward / backward pair -- same access pattern as the fused kernels, a different epilogue
 * (csrc/vfm_variants8.hpp for d % 8 == 0: lane groups, 8 coordinates per lane and Philox call, pipelined gathers;
 * csrc/vfm_variants.hip: the scalar pair for every other d):
 *   objective VFM_OBJ_SAMPLED      the sampled ELBO of vfm-torch.py:189-324,359
 *             VFM_OBJ_CLOSED_FORM  the closed-form expected log-likelihood of vfm-tomasrch.py:369-451 (no
 *                                  sampling; Normal likelihood): per row 1/2 log|alpha| - |alpha|/2 ((y - y_bar)^2 + T_n)
 *   priors    NULL = N(0,1) (vfm-torch.py:162-164), else the LEARNABLE GROUP PRIORS of vfm-tomasrch.py:206-290,
 *             flat [mean0, scale0 | mean_w[G] | scale_w[G] | mean_v[G,d] | scale_v[G,d]] (G = F groups =
 *             columns; sigma = |scale|); g_priors receives their gradient in the same layout
 *   values    NULL = entity ids only, else [B,F] feature values (sparse features with values != 1,
 *             vfm.py:483-509): pred = w0 + sum_f v_f w_f + 1/2 sum_k[(sum_f v_f z_fk)^2 - sum_f v_f^2 z_fk^2]
 * vfm_variant_fwd_f32: pred [B] (the sampled prediction, or y_bar), `state` [B, d] (sampled) / [B, 3d] (closed
 *   form) and grow [B] for the backward, partials workspace as vfm_elbo_fwd_f32; with `loss` (3 floats) also the
 *   loss triple (loss, likelihood term, KL term incl. the global bias' KL to ITS prior).  y == NULL: prediction only.
 * vfm_variant_bwd_f32: dense gradients of both tables, the three scalars and (with priors) the priors;
 *   occ_pos_ws = vfm_variant_workspace_elems(B, F, d) 4-byte elements of scratch (16-byte aligned).  Prior
 *   gradients: d % 8 == 0: per-workgroup partial rows added in a fixed order (reproducible); other d: float
 *   atomics (order-dependent in the last bits). */
#define VFM_OBJ_SAMPLED 0
#define VFM_OBJ_CLOSED_FORM 1
int vfm_variant_fwd_f32(const vfm_problem_t* p, int32_t objective, const void* x, const float* values, const float* y,
                        const float* entity_params, const float* bias_params, const float* inv_occ,
                        const float* scalars, const double* W, const float* priors, const float* eps_entity,
                        const float* eps_bias, const float* eps_global, float* pred, double* partials, float* state,
                        float* grow, float* loss, void* stream);
int64_t vfm_variant_workspace_elems(int64_t B, int32_t F, int32_t d);
int vfm_variant_bwd_f32(const vfm_problem_t* p, int32_t objective, const vfm_index_t* idx, int32_t* occ_pos_ws,
                        const void* x, const float* values, const float* entity_params, const float* bias_params,
                        const float* inv_occ, const float* scalars, const double* W, const float* priors,
                        const float* eps_entity, const float* eps_bias, const float* eps_global, const float* state,
                        const float* grow, const double* partials, const float* grad_out, float* g_entity,
                        float* g_bias, float* g_scalars, float* g_priors, void* stream);

/* Debug / test helper: write the eps the kernels would generate from (seed, step) into
 * tables (eps_entity [S,T,d], eps_bias [S,T], eps_global [S]; S = n_samples). */
int vfm_philox_eps_f32(const vfm_problem_t* p, float* eps_entity, float* eps_bias,
                       float* eps_global, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VFM_HIP_H */
