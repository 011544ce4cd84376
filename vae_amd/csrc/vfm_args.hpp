// vfm_args.hpp -- kernel argument structs and the host-side declarations shared by the translation
// units of libvfm_hip.so:
//   vfm_abi.hip  C ABI (include/vfm_hip.h), argument checks, small / Adam / shard glue kernels, k_heavy
//   vfm_fwd.hip  k_fwd instances + dispatch            (compiled once per link function, -DVFM_LINK=0|1)
//   vfm_fwd2.hip k_fwd2 instances + dispatch           (two fields, task-stream form; |.| link)
//   vfm_fwd2m.hip k_fwd2m instances + dispatch         (the same with 2..4 variational samples inside the kernel; |.| link)
//   vfm_fwdg.hip k_fwdg instances + dispatch           (general F, a row's fields split over lane groups; |.| link)
//   vfm_bwd.hip  k_bwd instances + dispatch            (compiled per link function and per family: -DVFM_BWD_PART=0 the
//                                                       gradient / statistics forms, =1 the fused-Adam forms)
// The softplus link (vfm-torch.py:125, overwritten by :126 in the reference itself) runs through the GENERAL kernels only
// (k_fwd, k_bwd without the pipelined forms): the specialised forward kernels and the record step exist for |.|.
// Splitting keeps every kernel family in its own object (built in parallel by vae_amd/build.py); a
// kernel is always launched from the unit that defines it, so no relocatable device code is needed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vfm_hip.h"

#define VFM_INTERNAL __attribute__((visibility("hidden")))

namespace vfm {

constexpr int BLOCK = 256;
constexpr float LOG_SQRT_2PI = 0.918938533204672742f;
constexpr float LN2 = 0.693147180559945309f;
constexpr float LOG2E = 1.4426950408889634f;

struct RngKey {
  uint32_t seed_lo, seed_hi, step_lo, step_hi;   // step_hi carries the sample index in bits 16.. (S > 1)
  uint32_t chunk_off;   // global index of local chunk 0 (4 coordinates each; dimension-sharded mode), even
};

// Kernel arguments (by value)
struct KArgs {
  int64_t B, T;
  int64_t e_lo, e_hi;   // entity range of a backward launch (chunked multi-rank pipeline)
  int32_t F, d, lik, id64, G, flags;
  int32_t S, sample;    // variational samples: S of them; the forward runs one launch per sample
  int32_t row_filter;   // fused backward+Adam: 0 all rows, 2 rows in the batch only,
                        // 3 heavy entities only (listed), 4 all but the heavy entities
  float inv_S;
  float ll_scale;  // nb_train / (B_global * S)
  double ll_scale_d;
  RngKey key;
  const void* x;
  const float* y;
  const float* entity;
  const float* bias;
  const float* inv_occ;
  const float* scalars;
  const double* W;
  const float* eps_entity;
  const float* eps_bias;
  const float* eps_global;
  vfm_dev_step_t* dev;  // non-NULL: the step-dependent values (Philox step, Adam constants) live in device memory (replayable step)
  float* wrec;          // non-NULL: packed first-order records [T,4] = (mu_w, s_w, 1/occ, 0) (vfm_problem_t.wrec)
  int64_t group_hi[VFM_MAX_FIELDS];
  double group_n[VFM_MAX_FIELDS];
};

struct FwdOut {
  float* pred;
  double* partials;
  float* sumz;
  float* grow;
};

struct BwdArgs {
  const int32_t* occ_ptr;
  const int32_t* occ_rows;
  const float* sumz;
  const float* grow;
  double* partials;
  const float* grad_out;
  float* g_entity;
  float* g_bias;
  float* g_scalars;
  float* loss;   // non-NULL: this launch also reduces the forward's partial slots and forms the loss
  // staged (multi-rank) form: sufficient statistics of the gradient, exchanged instead of the gradient
  float* acc;    // [T, 4 + round4(d)] record per entity: (sum_r grow_r, occurrences, 0, 0 | A_e[0..d-1]),
                 //   A_e = sum_r grow_r * sumz_r      (STAGE_ACC writes, STAGE_APPLY reads)
  float* sums;   // [2]    (sum_r grow_r over all rows, alpha term)
  // entities whose occurrence list is longer than VFM_HEAVY_LIST: pre-reduced by k_heavy
  const int32_t* heavy_ids;   // [n_heavy] sorted
  const float* heavy_acc;     // per sample: n_heavy entity records (sum grow, count, 0, 0 | A_e) [+ the work items' records]
  int32_t n_heavy;
  int32_t heavy_stride;       // records per sample in heavy_acc (n_heavy + n_items)
  // fused-Adam instances: the sorted ids of the rows to visit; NULL = scan all table rows.  The lazy exact-Adam
  // step (VFM_FLAG_ROWS_TOUCHED + vfm_index_t.touched_ids) walks this list.
  const int32_t* row_ids;
  int64_t n_rows;
  // software-pipelined step (PIPE instances, two fields): the walk gathers the OTHER entity's sample from this
  // step's records instead of a sumz row, and rows of the NEXT batch get their next-step record written after
  // their Adam update (vfm_elbo_bwd_adam_pipe_f32)
  const float* zrec;
  const int32_t* occ_other;
  float* zrec_next;               // NULL: nothing to prepare
  const int32_t* next_occ_ptr;    // inverted-index offsets of the next batch: membership test
  const double* next_W;           // its normalisers
  RngKey next_key;                // its Philox key (step + 1)
  // look-ahead lazy Adam (LA instances, vfm_elbo_bwd_adam_lookahead_f32): a row that is neither in this batch nor
  // in the next (next_occ_ptr) is skipped; a row that was skipped replays its zero-gradient updates (the constants
  // of the period's steps live in step_tab) before this step's update
  int32_t* last_step;             // [T] last Adam step applied to each row
  float2* step_tab;               // [VFM_MOMENT_PERIOD + 1] (a1, q2) of the k-th step of the moment period
  int32_t la_step, la_k;          // this Adam step and its position in the period
  // multi-rank stages with a row list: 1 = the statistics records form a COMPACT buffer, record i belongs to row_ids[i]
  // (what the compacted exchange all-reduces); 0 = records sit in the dense [T] table at the entity's own index
  int32_t rec_by_slot;
  // bounds of what the index may name (a corrupted index is clamped, never followed): list lengths in [0, n_occ], row
  // numbers in [0, B); `status` (device, may be NULL) counts the clamps that fired (vfm_index_t.status)
  int32_t n_occ;
  int32_t* status;
};

struct AdamArgs {
  float* m_entity; float* v_entity; float* m_bias; float* v_bias; float* m_scal; float* v_scal;
  float b1, b2, eps, step_size, bc2_sqrt;
  // VFM_FLAG_SCALED_MOMENTS: the buffers hold m / b1^k, v / b2^k; s1 = b1^k, s2 = b2^k of THIS step,
  // c1 = (1-b1)/s1, c2 = (1-b2)/s2; store_true: this step ends a period (write the true moments back)
  int32_t scaled, store_true;
  float s1, s2, c1, c2, inv_bc2_sqrt;
  // scaled form, folded per-step constants: a1 = step_size * b1^k (times the stored moment = step_size * m),
  // q2 = sqrt(b2^k) / sqrt(bc2) (times sqrt of the stored second moment = sqrt(v) / sqrt(bc2))
  float a1, q2;
};

// ---- replayable step (vfm_dev_step_t): device-side reads of what the host arguments otherwise carry ----
// forward: the eps stream step; block 0 hands (philox_step, adam_step) over to the backward launches of the same step
__device__ __forceinline__ RngKey key_of_step(const KArgs& a, bool writer) {
  RngKey key = a.key;
  if (a.dev) {
    const uint64_t s = a.dev->philox_step;
    key.step_lo = (uint32_t)s; key.step_hi = (uint32_t)(s >> 32);
    if (writer) { a.dev->philox_step_bwd = s; a.dev->adam_step_bwd = a.dev->adam_step; }
  }
  return key;
}
// backward: eps step + the Adam constants of the step from the device table; `bump`: this thread advances the state
// (the launch that forms the loss, after which nothing of this step reads philox_step / adam_step again)
__device__ __forceinline__ void load_dev_step(const KArgs& a, RngKey& key, AdamArgs& ad, int32_t& la_step, int32_t& la_k,
                                              RngKey& next_key, bool bump) {
  if (!a.dev) return;
  const uint64_t s = a.dev->philox_step_bwd;
  const int64_t t = a.dev->adam_step_bwd;
  key.step_lo = (uint32_t)s; key.step_hi = (uint32_t)(s >> 32);
  next_key.step_lo = (uint32_t)(s + 1); next_key.step_hi = (uint32_t)((s + 1) >> 32);
  int64_t i = t - a.dev->tab_first;
  const bool miss = i < 0 || i >= a.dev->tab_len;
  if (miss) i = i < 0 ? 0 : a.dev->tab_len - 1;
  const vfm_step_consts_t c = a.dev->tab[i];
  ad.step_size = c.step_size; ad.bc2_sqrt = c.bc2_sqrt; ad.inv_bc2_sqrt = 1.0f / c.bc2_sqrt;
  ad.a1 = c.a1; ad.q2 = c.q2; ad.c1 = c.c1; ad.c2 = c.c2; ad.s1 = c.s1; ad.s2 = c.s2;
  ad.store_true = c.store_true; ad.scaled = c.scaled;
  ad.b1 = c.beta1; ad.b2 = c.beta2; ad.eps = c.eps;
  la_step = (int32_t)t; la_k = c.k;
  if (bump) {
    a.dev->philox_step = s + 1; a.dev->adam_step = t + 1;
    if (miss) a.dev->error = 1;
  }
}

// lane-group shape of the row kernels for an embedding size (see pick_shape in vfm_abi.hip)
struct Shape {
  int lpe, cpl, vec;
};

enum { EPS_PHILOX = 0, EPS_TABLE = 1, EPS_ZERO = 2, EPS_ZREC = 4 };
enum { MODE_PREDICT = 0, MODE_TRAIN = 1 };
enum { STAGE_FULL = 0, STAGE_ACC = 1, STAGE_APPLY = 2 };
enum { LINK_ABS = 0, LINK_SOFTPLUS = 1 };
constexpr int MAX_SAMPLES = 64;

// thread-local error string behind vfm_last_error() (defined in vfm_abi.hip)
VFM_INTERNAL int fail(int code, const char* msg);
VFM_INTERNAL int fail_hip(hipError_t e, const char* where);
VFM_INTERNAL int env_int(const char* name, int dflt);
// W[f] = sum_r inv_occ[x[r,f]] with the stand-alone kernels (vfm_abi.hip)
VFM_INTERNAL int launch_norms(int64_t B, int F, int64_t T, int id_bits, const void* x, const float* inv_occ, double* W,
                              hipStream_t st);

// per-link-function launchers (vfm_fwd.hip / vfm_bwd.hip, one object per link; vfm_bwd.hip in two parts: `adam` 0 / 10
// (gradients, statistics) in part 0, the fused-Adam forms 1 / 2 / 11 in part 1)
#define VFM_DECLARE_LAUNCHERS(SUFFIX)                                                                          \
  VFM_INTERNAL int launch_fwd_##SUFFIX(const Shape& s, int eps, int mode, int ff, KArgs& a, const FwdOut& o,   \
                                       hipStream_t st);                                                        \
  VFM_INTERNAL int launch_bwd0_##SUFFIX(const Shape& s, int eps, int adam, KArgs& a, const BwdArgs& b,         \
                                        const AdamArgs& ad, hipStream_t st);                                   \
  VFM_INTERNAL int launch_bwd1_##SUFFIX(const Shape& s, int eps, int adam, KArgs& a, const BwdArgs& b,         \
                                        const AdamArgs& ad, hipStream_t st);
VFM_DECLARE_LAUNCHERS(abs)
VFM_DECLARE_LAUNCHERS(softplus)
// the small-table form of the fused dense step (vfm_bwd_small.hpp): L = work-item length, THR = heavy threshold of the index
VFM_INTERNAL int launch_bwd_small_abs(const Shape& s, KArgs& a, const BwdArgs& b, const AdamArgs& ad, int L, int THR, hipStream_t st);
VFM_INTERNAL int launch_bwd_small_softplus(const Shape& s, KArgs& a, const BwdArgs& b, const AdamArgs& ad, int L, int THR, hipStream_t st);
// |.| link only: the specialised forward kernels and the record sampler of the pipelined step
VFM_INTERNAL int launch_fwd2_abs(int eps, int mode, KArgs& a, const FwdOut& o, hipStream_t st);
VFM_INTERNAL int launch_fwd2m_abs(int eps, int mode, KArgs& a, const FwdOut& o, hipStream_t st);
VFM_INTERNAL int launch_fwdg_abs(int eps, int mode, KArgs& a, const FwdOut& o, hipStream_t st);
VFM_INTERNAL int launch_sample_rec_abs(const Shape& s, KArgs& a, const int32_t* ids, int n, float* zrec, hipStream_t st);

// (d = 4 and d = 8 run the 4-lane shape with idle lanes: two instance families fewer in every kernel)
#define VFM_FOR_SHAPES(X)                                                                      \
  X(4, 1, 4) X(8, 1, 4) X(16, 1, 4) X(32, 1, 4) X(64, 1, 4) X(64, 2, 4) \
  X(64, 4, 4) X(8, 1, 1) X(64, 1, 1) X(64, 4, 1)

}  // namespace vfm
