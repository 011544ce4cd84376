// vfm_abi.hip -- the C ABI of libvfm_hip.so (include/vfm_hip.h): argument checks, kernel-argument
// set-up and dispatch for the hand-written gfx950 (MI355X, CDNA4) kernels of the Variational-FM ELBO
// step.  Wave = 64 lanes; no MFMA: the path is gather + elementwise + reduction and is bounded by
// HBM / Infinity-Cache bandwidth.
//
// Replaces the per-batch body of the reference: vfm-torch.py:189-324 (CF.forward), the loss
// line :359 and autograd through them (:368-369).  Math: SURVEY.md Appendix A.
//
// Kernels (details at each definition; translation units: vfm_args.hpp)
//   k_fwd     : a *lane group* of LPE lanes owns one batch row, each lane CPL chunks of VEC
//               coordinates; rows of a workgroup are contiguous; ids -> table rows -> arithmetic are
//               software-pipelined in registers; FM reduction with DPP / permlane swaps; per-block
//               partial sums to private slots (no atomics).                             [vfm_fwd.hip]
//   k_finalize: adds the slots, forms the loss (also foldable into the fused backward).
//   k_bwd     : entity-centric.  A lane group owns one TABLE row e, sums grow[r]*sumz[r,:] over the
//               batch rows containing e (inverted index) and either stores the dense gradient row,
//               applies dense Adam in place (ADAM), or -- multi-rank -- stores / consumes the
//               gradient's sufficient statistics (STAGE_ACC / STAGE_APPLY).             [vfm_bwd.hip]
//   k_heavy   : parallel pre-reduction of occurrence lists longer than VFM_HEAVY_LIST (skewed data).
//   k_adam    : dense Adam on a flat buffer (unfused path).  k_norms, k_inv_occ: per-batch / per-dataset
//               normalisers.  k_philox_dump: the eps stream, for tests.
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <mutex>

#include "vfm_args.hpp"

namespace vfm {

namespace {
thread_local char g_err[512] = "";
}

int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
int fail_hip(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
  return (int)e;
}
int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

namespace {

#include "vfm_rng.hpp"
#include "vfm_common.hpp"
#include "vfm_reduce.hpp"
#include "vfm_small_kernels.hpp"
#include "vfm_heavy.hpp"
#include "vfm_adam.hpp"

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
bool pick_shape(int d, Shape* s) {
  auto pow2ceil = [](int v) { int p = 1; while (p < v) p <<= 1; return p; };
  if (d % 4 == 0) {
    const int C = d / 4;
    s->vec = 4;
    s->lpe = pow2ceil(C) < 64 ? (pow2ceil(C) < 4 ? 4 : pow2ceil(C)) : 64;      // (d = 4, 8: the 4-lane shape, idle lanes)
    const int cpl = (C + s->lpe - 1) / s->lpe;
    s->cpl = cpl <= 1 ? 1 : (cpl <= 2 ? 2 : 4);
    return cpl <= 4;
  }
  s->vec = 1;
  if (d <= 8) { s->lpe = 8; s->cpl = 1; return true; }
  if (d <= 64) { s->lpe = 64; s->cpl = 1; return true; }
  s->lpe = 64; s->cpl = 4;
  return d <= 256;
}

// a struct built against another layout is refused, not read past its end (VFM_STRUCT_INIT in include/vfm_hip.h)
int check_struct(uint32_t size, uint32_t abi, size_t want, const char* name) {
  if (size == (uint32_t)want && abi == (uint32_t)VFM_ABI_VERSION) return 0;
  snprintf(g_err, sizeof(g_err), "%s: struct_size %u / abi_version %u, this library expects %zu / %d (set them with "
           "VFM_STRUCT_INIT; a binding generated from another version of include/vfm_hip.h must be regenerated)",
           name, size, abi, want, VFM_ABI_VERSION);
  return VFM_E_INVALID;
}

int check_problem(const vfm_problem_t* p, bool dev_ok = false) {
  if (!p) return fail(VFM_E_INVALID, "problem is NULL");
  if (int rc = check_struct(p->struct_size, p->abi_version, sizeof(vfm_problem_t), "vfm_problem_t")) return rc;
  if (p->dev_step && !dev_ok)
    return fail(VFM_E_UNSUPPORTED, "vfm_problem_t.dev_step: only the training forward and the fused backward+Adam entry points read the step from device memory");
  if (p->dev_step && p->n_samples != 1) return fail(VFM_E_UNSUPPORTED, "vfm_problem_t.dev_step: one variational sample");
  if (p->dev_step && (((uintptr_t)p->dev_step) & 15) != 0) return fail(VFM_E_INVALID, "vfm_problem_t.dev_step must be 16-byte aligned");
  if (p->wrec && (((uintptr_t)p->wrec) & 15) != 0) return fail(VFM_E_INVALID, "vfm_problem_t.wrec must be 16-byte aligned");
  if (p->B < 0 || p->T <= 0 || p->T > 0xFFFFFFFELL) return fail(VFM_E_INVALID, "bad B or T");
  if (p->F < 1 || p->F > VFM_MAX_FIELDS) return fail(VFM_E_INVALID, "F out of range [1,64]");
  if (p->d < 1) return fail(VFM_E_INVALID, "d < 1");
  if (p->id_bits != 32 && p->id_bits != 64) return fail(VFM_E_INVALID, "id_bits must be 32 or 64");
  if (p->likelihood != VFM_LIK_NORMAL && p->likelihood != VFM_LIK_BERNOULLI)
    return fail(VFM_E_INVALID, "unknown likelihood");
  if (p->n_samples < 1 || p->n_samples > MAX_SAMPLES) return fail(VFM_E_INVALID, "n_samples out of range [1,64]");
  if (p->n_samples > 1 && (p->step >> 48) != 0) return fail(VFM_E_INVALID, "step must stay below 2^48 when n_samples > 1");
  if (p->B_global < p->B) return fail(VFM_E_INVALID, "B_global < B");
  if (p->B * (int64_t)p->F > 0x7FFFFFFFLL) return fail(VFM_E_INVALID, "B*F exceeds int32 index range");
  if (p->e_lo < 0 || p->e_lo > p->T || (p->e_hi != 0 && p->e_hi < p->e_lo)) return fail(VFM_E_INVALID, "bad entity range");
  if (p->flags & ~(VFM_FLAG_NO_PRIOR_TERMS | VFM_FLAG_EPS_ZERO | VFM_FLAG_SPARSE_ADAM | VFM_FLAG_LINK_SOFTPLUS |
                   VFM_FLAG_SCALED_MOMENTS | VFM_FLAG_ROWS_TOUCHED | VFM_FLAG_ZREC | VFM_FLAG_SHARE_GPU))
    return fail(VFM_E_INVALID, "unknown bit in vfm_problem_t.flags");
  Shape s;
  if (!pick_shape(p->d, &s)) return fail(VFM_E_UNSUPPORTED, "embedding size d not supported (d%4==0: d<=1024, else d<=256)");
  return 0;
}

bool softplus(const vfm_problem_t* p) { return (p->flags & VFM_FLAG_LINK_SOFTPLUS) != 0; }

// the multi-rank stages exchange single-sample statistics
int single_sample_only(const vfm_problem_t* p, const char* who) {
  if (p->n_samples == 1) return 0;
  snprintf(g_err, sizeof(g_err), "%s: n_samples > 1 is not supported by this entry point", who);
  return VFM_E_UNSUPPORTED;
}


// eps source of a call: VFM_FLAG_EPS_ZERO > tables > Philox
int eps_mode(const vfm_problem_t* p, const float* ee, const float* eb, const float* eg, int* mode) {
  const int neps = (ee != nullptr) + (eb != nullptr) + (eg != nullptr);
  if (neps != 0 && neps != 3) return fail(VFM_E_INVALID, "give all three eps tables or none");
  *mode = (p->flags & VFM_FLAG_EPS_ZERO) ? EPS_ZERO : (neps == 3 ? EPS_TABLE : EPS_PHILOX);
  return 0;
}

// `sample`: the variational sample a forward launch handles (S > 1: one launch per sample; the eps
// tables of sample s are the s-th [T,d] / [T] blocks).  Backward launches take sample 0 = the base.
KArgs make_args(const vfm_problem_t* p, const void* x, const float* y, const float* entity,
                const float* bias, const float* inv_occ, const float* scalars, const double* W,
                const float* ee, const float* eb, const float* eg, int sample = 0) {
  KArgs a;
  memset(&a, 0, sizeof(a));
  a.S = p->n_samples; a.sample = sample; a.inv_S = 1.0f / (float)p->n_samples;
  if (sample > 0) {
    if (ee) ee += (size_t)sample * (size_t)p->T * (size_t)p->d;
    if (eb) eb += (size_t)sample * (size_t)p->T;
  }
  a.B = p->B; a.T = p->T; a.F = p->F; a.d = p->d; a.lik = p->likelihood;
  a.e_lo = p->e_lo; a.e_hi = (p->e_hi > 0 && p->e_hi < p->T) ? p->e_hi : p->T;
  a.id64 = p->id_bits == 64; a.G = p->F; a.flags = p->flags;
  a.ll_scale_d = (double)p->nb_train / ((double)(p->B_global > 0 ? p->B_global : 1) * (double)p->n_samples);
  a.ll_scale = (float)a.ll_scale_d;
  a.key.seed_lo = (uint32_t)p->seed; a.key.seed_hi = (uint32_t)(p->seed >> 32);
  a.key.step_lo = (uint32_t)p->step; a.key.step_hi = (uint32_t)(p->step >> 32);
  a.key.chunk_off = 0;
  a.x = x; a.y = y; a.entity = entity; a.bias = bias; a.inv_occ = inv_occ; a.scalars = scalars;
  a.W = W; a.eps_entity = ee; a.eps_bias = eb; a.eps_global = eg;
  a.dev = p->dev_step; a.wrec = p->wrec;
  for (int g = 0; g < p->F; ++g) { a.group_hi[g] = p->group_hi[g]; a.group_n[g] = p->group_n[g]; }
  return a;
}

// Which forward kernel: two fields, one variational sample, d % 4 == 0 and d <= 512 run as a stream of
// sampling tasks (k_fwd2: an id repeated in consecutive rows of the second column is sampled once per run);
// everything else -- general F, S > 1, the multi-rank forms that feed the kernel slots / partial row values --
// runs k_fwd.  VFM_FWD_KERNEL=1 forces k_fwd (A/B runs, tests of both kernels).
bool use_fwd2(const vfm_problem_t* p, int eps, bool multi = false) {
  if (p->F != 2 || (p->d & 3) != 0 || p->d > 512) return false;
  if (multi ? (p->n_samples < 2 || p->n_samples > 4 || eps == EPS_ZERO) : p->n_samples != 1) return false;
  if (softplus(p)) return false;          // (the softplus link runs through the general kernel: vfm_args.hpp)
  // below d = 20 a lane group is two lanes wide and the row-parallel k_fwd is the faster kernel (d = 16, 800 K rows:
  // 46.6 vs 60.1 us; from d = 20 on k_fwd2 wins: 17.7 vs 20.8 at d = 20, 40.2 vs 43.2 at d = 128).  VFM_FWD_KERNEL=2
  // forces k_fwd2 wherever it is defined, =1 k_fwd (A/B runs, tests); the record gather always runs in k_fwd2.
  const int forced = env_int("VFM_FWD_KERNEL", 0);
  if (forced == 1) return false;
  return forced == 2 || eps == EPS_ZREC || multi || p->d >= 20;
}

// General number of fields (F != 2), one sample, d % 4 == 0, 16 <= d <= 512, eps from Philox (or zero, predictions):
// k_fwdg -- the fields of a row split over lane groups, 8 coordinates per lane (vfm_fwdg.hpp).  Table eps, S > 1 and
// the multi-rank forms (slots / partial row values / no first-order weights) stay with k_fwd; VFM_FWD_KERNEL=1 forces it.
bool use_fwdg(const vfm_problem_t* p, int eps, int mode) {
  if (p->F == 2 || (p->d & 3) != 0 || p->d < 16 || p->d > 512 || p->n_samples != 1) return false;
  if ((p->flags & VFM_FLAG_ZREC) || softplus(p)) return false;
  if (!(eps == EPS_PHILOX || (eps == EPS_ZERO && mode == MODE_PREDICT))) return false;
  return env_int("VFM_FWD_KERNEL", 0) != 1;
}

int dispatch_fwd(const vfm_problem_t* p, const Shape& s, int eps, int mode, int ff, KArgs& a, const FwdOut& o,
                 hipStream_t st) {
  if (use_fwdg(p, eps, mode)) return launch_fwdg_abs(eps, mode, a, o, st);
  if (use_fwd2(p, eps)) {
    // VFM_FWD_AB_NORNG=1 (profiling only, wrong results): eps = 0 in the training forward, i.e. the kernel
    // without its Philox / Box-Muller arithmetic
    if (mode == MODE_TRAIN && eps == EPS_PHILOX && env_int("VFM_FWD_AB_NORNG", 0) == 1) eps = EPS_ZERO;
    return launch_fwd2_abs(eps, mode, a, o, st);
  }
  return softplus(p) ? launch_fwd_softplus(s, eps, mode, ff, a, o, st) : launch_fwd_abs(s, eps, mode, ff, a, o, st);
}
int dispatch_bwd(const vfm_problem_t* p, const Shape& s, int eps, int adam, KArgs& a, const BwdArgs& b,
                 const AdamArgs& ad, hipStream_t st) {
  // (the gradient / statistics forms and the fused-Adam forms sit in two translation units per link: vfm_bwd.hip)
  if (adam == 0 || adam == 10)
    return softplus(p) ? launch_bwd0_softplus(s, eps, adam, a, b, ad, st) : launch_bwd0_abs(s, eps, adam, a, b, ad, st);
  return softplus(p) ? launch_bwd1_softplus(s, eps, adam, a, b, ad, st) : launch_bwd1_abs(s, eps, adam, a, b, ad, st);
}

// what every backward-family launch starts from: the index, the bounds of what it may name, its status word
BwdArgs bwd_args(const vfm_problem_t* p, const vfm_index_t* idx) {
  BwdArgs b;
  memset(&b, 0, sizeof(b));
  if (idx) { b.occ_ptr = idx->occ_ptr; b.occ_rows = idx->occ_rows; b.status = idx->status; }
  b.n_occ = (int32_t)(p->B * (int64_t)p->F);
  return b;
}

template <int LPE, int CPL, int VEC>
int launch_heavy_t(const vfm_problem_t* p, const vfm_index_t* idx, const float* sumz, const float* grow, float* hacc, int d,
                   hipStream_t st, const float* zrec = nullptr) {
  constexpr int GPB = BLOCK / LPE;
  const size_t xs = 4 + (((size_t)d + 3) & ~(size_t)3);
  float* item_acc = hacc + xs * (size_t)idx->n_heavy;            // item records sit behind the entity records
  int64_t nb = ((int64_t)idx->n_items + GPB - 1) / GPB;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL((k_heavy<LPE, CPL, VEC>), dim3((unsigned)nb), dim3(BLOCK), 0, st, idx->heavy_items,
                     (int)idx->n_items, idx->occ_rows, sumz, grow, item_acc, d, zrec ? idx->occ_other : nullptr, zrec, hacc,
                     (int)idx->n_heavy, (int)(p->B * (int64_t)p->F), (int)p->B, p->T, idx->status);
  if (idx->max_items > 0 && idx->max_items <= VFM_HEAVY_DIRECT) return 0;      // (every entity's items are added by the main kernel)
  nb = idx->n_heavy;                                             // one workgroup per heavy entity
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL((k_heavy_sum<LPE, CPL, VEC>), dim3((unsigned)nb), dim3(BLOCK), 0, st, idx->heavy_items,
                     (int)idx->n_items, (int)idx->n_heavy, item_acc, hacc, d);
  return 0;
}

// pre-reduce the long occurrence lists (if the index has any) and point the main kernel at the result
int run_heavy(const vfm_problem_t* p, const vfm_index_t* idx, const float* sumz, const float* grow,
              hipStream_t st, BwdArgs* b, const float* zrec = nullptr) {
  b->heavy_ids = nullptr; b->heavy_acc = nullptr; b->n_heavy = 0; b->heavy_stride = 0;
  if (idx->n_heavy <= 0 || idx->n_items <= 0) return 0;
  if (!idx->heavy_ids || !idx->heavy_items || !idx->heavy_acc)
    return fail(VFM_E_INVALID, "index: heavy_ids / heavy_items / heavy_acc missing");
  if ((int64_t)idx->n_heavy > p->T || (int64_t)idx->n_items > p->B * (int64_t)p->F + idx->n_heavy)
    return fail(VFM_E_INVALID, "index: more heavy entities than table rows, or more work items than occurrences");
  const size_t xs = 4 + (((size_t)p->d + 3) & ~(size_t)3);
  const size_t per_sample = xs * ((size_t)idx->n_heavy + (size_t)idx->n_items);   // records of one sample
  Shape s;
  pick_shape(p->d, &s);
  for (int sm = 0; sm < p->n_samples; ++sm) {
    const float* sz = sumz ? sumz + (size_t)sm * (size_t)p->B * (size_t)p->d : nullptr;
    float* hacc = idx->heavy_acc + (size_t)sm * per_sample;
#define X(L_, C_, V_) \
    if (s.lpe == L_ && s.cpl == C_ && s.vec == V_) launch_heavy_t<L_, C_, V_>(p, idx, sz, grow, hacc, p->d, st, zrec);
    VFM_FOR_SHAPES(X)
#undef X
  }
  b->heavy_ids = idx->heavy_ids; b->heavy_acc = idx->heavy_acc; b->n_heavy = idx->n_heavy;
  b->heavy_stride = idx->n_heavy + idx->n_items;
  return 0;
}

// ---- overlap of the long-list pre-reduction with the main kernel (skewed data, large tables) ----
// k_heavy -> k_heavy_sum are two small latency-bound kernels the main kernel would otherwise wait for (Zipf(1.1) items
// at cfg3: 20 + 14 us in front of a 150 us kernel).  They run on a side stream of the library's own (one per device,
// created on first use, never destroyed) while the main kernel handles every entity BUT the heavy ones on the
// caller's stream; the heavy entities follow in a second, listed launch once the pre-reduction has finished.
// Ordering is by events only; the caller's stream stays the only thing the caller has to synchronise with.
// Thread safety: the side stream and its event pair are shared by every caller of a device, so the whole fork / join
// enqueue (record e1 .. wait e2) runs under `mu` -- two host threads driving different streams of one device cannot
// interleave their records and waits (an event waited on always carries the record of the same call).
struct Side { hipStream_t s = nullptr; hipEvent_t e1 = nullptr, e2 = nullptr; bool ok = false, tried = false; std::mutex mu; };
Side* side_of_device() {
  static Side sides[64];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  Side& sd = sides[dev];
  if (!sd.tried) {
    sd.tried = true;
    sd.ok = hipStreamCreateWithFlags(&sd.s, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&sd.e1, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&sd.e2, hipEventDisableTiming) == hipSuccess;
  }
  return sd.ok ? &sd : nullptr;
}
// worth it when the heavy entities are a small part of the rows the main kernel visits (otherwise nothing is hidden)
bool heavy_overlap(const vfm_problem_t* p, const vfm_index_t* idx) {
  return idx->n_heavy > 0 && idx->n_items > 0 && (int64_t)idx->n_heavy * 16 < p->T && p->n_samples == 1 &&
         env_int("VFM_HEAVY_OVERLAP", 1) != 0;
}

// fused backward + Adam with the pre-reduction overlapped: `b` without heavy fields yet; rows = b.row_ids (or all)
int bwd_adam_overlapped(const vfm_problem_t* p, const vfm_index_t* idx, const Shape& s, int eps, int adam, KArgs& a, BwdArgs& b,
                        const AdamArgs& ad, const float* sumz, const float* grow, hipStream_t st, Side* sd) {
  std::lock_guard<std::mutex> lk(sd->mu);
  hipError_t e = hipEventRecord(sd->e1, st);
  if (e == hipSuccess) e = hipStreamWaitEvent(sd->s, sd->e1, 0);
  if (e != hipSuccess) return fail_hip(e, "heavy-list overlap: events");
  if (int rc = run_heavy(p, idx, sumz, grow, sd->s, &b)) return rc;
  e = hipEventRecord(sd->e2, sd->s);
  if (e != hipSuccess) return fail_hip(e, "heavy-list overlap: events");
  const int32_t keep = a.row_filter;
  a.row_filter = 4;                                   // everything but the heavy entities (+ loss, scalars)
  if (int rc = dispatch_bwd(p, s, eps, adam, a, b, ad, st)) return rc;
  e = hipStreamWaitEvent(st, sd->e2, 0);
  if (e != hipSuccess) return fail_hip(e, "heavy-list overlap: events");
  BwdArgs bh = b;
  bh.row_ids = idx->heavy_ids; bh.n_rows = idx->n_heavy;
  a.row_filter = 3;                                   // the heavy entities, listed
  const int rc = dispatch_bwd(p, s, eps, adam, a, bh, ad, st);
  a.row_filter = keep;
  return rc;
}

// The fused dense step of a SMALL table in one launch (csrc/vfm_bwd_small.hpp): every row, dense Adam, Philox eps, one
// sample, d % 4 == 0 up to 256, and an index that says what its heavy lists were built with.  VFM_BWD_SMALL=0: A/B, tests.
bool small_table_step(const vfm_problem_t* p, const vfm_index_t* idx, int eps, int rows_flags) {
  if (eps != EPS_PHILOX || rows_flags != 0 || p->n_samples != 1 || (p->flags & VFM_FLAG_SPARSE_ADAM)) return false;
  if ((p->d & 3) != 0 || p->d > 256) return false;
  if (idx->heavy_list < VFM_HEAVY_MIN || idx->heavy_threshold < VFM_HEAVY_MIN || idx->heavy_threshold > idx->heavy_list) return false;
  if (p->T * (2 * (int64_t)p->d + 2) * 4 > (2LL << 20)) return false;
  return env_int("VFM_BWD_SMALL", 1) != 0;
}

int check_index(const vfm_problem_t* p, const vfm_index_t* idx, const char* who) {
  if (!idx) { snprintf(g_err, sizeof(g_err), "%s: inverted index missing", who); return VFM_E_INVALID; }
  if (int rc = check_struct(idx->struct_size, idx->abi_version, sizeof(vfm_index_t), "vfm_index_t")) return rc;
  if (!idx->occ_ptr || (p->B > 0 && !idx->occ_rows)) {
    snprintf(g_err, sizeof(g_err), "%s: inverted index missing", who);
    return VFM_E_INVALID;
  }
  if (idx->n_heavy < 0 || idx->n_items < 0 || idx->n_touched < 0 || idx->max_items < 0) {
    snprintf(g_err, sizeof(g_err), "%s: negative count in vfm_index_t", who);
    return VFM_E_INVALID;
  }
  return 0;
}

int after_launch(const char* where) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, where);
  return 0;
}

void adam_consts(float lr, float beta1, float beta2, int64_t step, float* step_size, float* bc2_sqrt) {
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  *step_size = (float)((double)lr / bc1);
  *bc2_sqrt = (float)sqrt(bc2);
}

// the two folded constants of a scaled-form step (one definition: the dense kernel and the catch-up table use it)
void scaled_step_consts(float step_size, float bc2_sqrt, double s1, double s2, float* a1, float* q2) {
  *a1 = (float)((double)step_size * s1);
  *q2 = (float)(sqrt(s2) / (double)bc2_sqrt);
}

// VFM_FLAG_SCALED_MOMENTS: scale factors of Adam step `step` (k = position inside the period, 1..R)
int scaled_moment_consts(const vfm_problem_t* p, float beta1, float beta2, int64_t step, AdamArgs* ad) {
  ad->scaled = 0; ad->store_true = 0; ad->s1 = ad->s2 = 1.f; ad->c1 = 1.f - beta1; ad->c2 = 1.f - beta2;
  ad->inv_bc2_sqrt = 1.0f / ad->bc2_sqrt;
  ad->a1 = ad->step_size; ad->q2 = ad->inv_bc2_sqrt;
  if (!(p->flags & VFM_FLAG_SCALED_MOMENTS)) return 0;
  if (p->flags & VFM_FLAG_SPARSE_ADAM)
    return fail(VFM_E_INVALID, "VFM_FLAG_SCALED_MOMENTS and VFM_FLAG_SPARSE_ADAM exclude each other");
  const int64_t k = (step - 1) % VFM_MOMENT_PERIOD + 1;
  const double s1 = pow((double)beta1, (double)k), s2 = pow((double)beta2, (double)k);
  if (!(s1 > 1e-30) || !(s2 > 1e-30))
    return fail(VFM_E_UNSUPPORTED, "VFM_FLAG_SCALED_MOMENTS: beta^128 underflows (use the plain moments for such betas)");
  ad->scaled = 1; ad->store_true = (k == VFM_MOMENT_PERIOD);
  ad->s1 = (float)s1; ad->s2 = (float)s2;
  ad->c1 = (float)((1.0 - (double)beta1) / s1); ad->c2 = (float)((1.0 - (double)beta2) / s2);
  scaled_step_consts(ad->step_size, ad->bc2_sqrt, s1, s2, &ad->a1, &ad->q2);
  return 0;
}

}  // namespace
}  // namespace vfm

using namespace vfm;

// the stand-alone batch normalisers (vfm_batch_norms; vfm_build_index with more than four fields)
int vfm::launch_norms(int64_t B, int F, int64_t T, int id_bits, const void* x, const float* inv_occ, double* W, hipStream_t st) {
  hipLaunchKernelGGL(k_zero_f64, dim3(1), dim3(64), 0, st, W, F);      // (all-zero bits: 0.0 and integer 0)
  if (B == 0) return after_launch("vfm_batch_norms");                  // empty shard: W = 0
  int64_t nbx = (B + BLOCK * 8 - 1) / (BLOCK * 8);
  if (nbx > 1024) nbx = 1024;
  hipLaunchKernelGGL(k_norms, dim3((unsigned)nbx, (unsigned)F), dim3(BLOCK), 0, st, x, (int)(id_bits == 64), inv_occ, B, F, T,
                     reinterpret_cast<unsigned long long*>(W));
  hipLaunchKernelGGL(k_norms_fix, dim3(1), dim3(64), 0, st, W, F);
  return after_launch("vfm_batch_norms");
}

extern "C" {

int vfm_abi_version(void) { return VFM_ABI_VERSION; }
const char* vfm_last_error(void) { return g_err; }

int vfm_step_consts(float lr, float beta1, float beta2, float eps_adam, int64_t step, int32_t scaled,
                    vfm_step_consts_t* out) {
  if (!out || step < 1) return fail(VFM_E_INVALID, "vfm_step_consts: bad argument");
  vfm_problem_t p;
  memset(&p, 0, sizeof(p));
  p.flags = scaled ? VFM_FLAG_SCALED_MOMENTS : 0;
  AdamArgs ad;
  memset(&ad, 0, sizeof(ad));
  ad.b1 = beta1; ad.b2 = beta2; ad.eps = eps_adam;
  adam_consts(lr, beta1, beta2, step, &ad.step_size, &ad.bc2_sqrt);         // (exactly what the host-argument path does)
  if (int rc = scaled_moment_consts(&p, beta1, beta2, step, &ad)) return rc;
  memset(out, 0, sizeof(*out));
  out->step_size = ad.step_size; out->bc2_sqrt = ad.bc2_sqrt; out->a1 = ad.a1; out->q2 = ad.q2;
  out->c1 = ad.c1; out->c2 = ad.c2; out->s1 = ad.s1; out->s2 = ad.s2;
  out->store_true = ad.store_true; out->k = (int32_t)((step - 1) % VFM_MOMENT_PERIOD + 1); out->scaled = ad.scaled;
  out->lr = lr; out->beta1 = beta1; out->beta2 = beta2; out->eps = eps_adam;
  return 0;
}

int vfm_dev_step_set(vfm_dev_step_t* dev_step, uint64_t philox_step, int64_t adam_step, void* stream) {
  if (!dev_step || adam_step < 1) return fail(VFM_E_INVALID, "vfm_dev_step_set: bad argument");
  hipLaunchKernelGGL(k_dev_step_set, dim3(1), dim3(64), 0, (hipStream_t)stream, dev_step, philox_step, adam_step);
  return after_launch("vfm_dev_step_set");
}

int vfm_wrec_build_f32(const float* bias_params, const float* inv_occ, int64_t T, float* wrec, void* stream) {
  if (!bias_params || !inv_occ || !wrec || T <= 0 || (((uintptr_t)wrec) & 15) != 0)
    return fail(VFM_E_INVALID, "vfm_wrec_build_f32: bad argument");
  const int grid = (int)((T + 255) / 256 < 2048 ? (T + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_wrec_build, dim3(grid), dim3(256), 0, (hipStream_t)stream, bias_params, inv_occ, T, wrec);
  return after_launch("vfm_wrec_build_f32");
}

int vfm_inv_occ_f32(const int64_t* nb_occ, float* inv_occ, int64_t T, void* stream) {
  if (!nb_occ || !inv_occ || T <= 0) return fail(VFM_E_INVALID, "vfm_inv_occ_f32: bad argument");
  const int grid = (int)((T + 255) / 256 < 2048 ? (T + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_inv_occ, dim3(grid), dim3(256), 0, (hipStream_t)stream, nb_occ, inv_occ, T);
  return after_launch("vfm_inv_occ_f32");
}

int vfm_batch_norms(const vfm_problem_t* p, const void* x, const float* inv_occ, double* W,
                    void* stream) {
  if (int rc = check_problem(p)) return rc;
  if (!W || (p->B > 0 && (!x || !inv_occ))) return fail(VFM_E_INVALID, "vfm_batch_norms: NULL pointer");
  return launch_norms(p->B, p->F, p->T, p->id_bits, x, inv_occ, W, (hipStream_t)stream);
}

int vfm_elbo_fwd_f32(const vfm_problem_t* p, const void* x, const float* y,
                     const float* entity_params, const float* bias_params,
                     const float* inv_occ, const float* scalars, const double* W,
                     const float* eps_entity, const float* eps_bias, const float* eps_global,
                     float* pred, double* partials, float* sumz, float* grow, void* stream) {
  if (int rc = check_problem(p, true)) return rc;
  if (!partials) return fail(VFM_E_INVALID, "vfm_elbo_fwd_f32: partials is NULL");
  if (p->dev_step && (eps_entity || eps_bias || eps_global || (p->flags & VFM_FLAG_EPS_ZERO)))
    return fail(VFM_E_UNSUPPORTED, "vfm_elbo_fwd_f32: dev_step goes with the in-kernel Philox eps stream only");
  hipStream_t st0 = (hipStream_t)stream;
  if (p->B == 0) {  // empty shard (a rank without rows): zero sums, zero blocks; buffers may be NULL
    hipLaunchKernelGGL(k_zero_f64, dim3(1), dim3(64), 0, st0, partials, (int)VFM_N_PARTIALS);
    return after_launch("vfm_elbo_fwd_f32");
  }
  const bool zrec = (p->flags & VFM_FLAG_ZREC) != 0;
  if (zrec) {       // the forward of the software-pipelined step: a gather of this step's sample records
    if (!use_fwd2(p, EPS_ZREC)) return fail(VFM_E_UNSUPPORTED, "VFM_FLAG_ZREC: two fields, one sample, d % 4 == 0, d <= 512, |.| link");
    if (!x || !y || !entity_params || !scalars || !pred || !grow || eps_entity || eps_bias || eps_global)
      return fail(VFM_E_INVALID, "vfm_elbo_fwd_f32 (VFM_FLAG_ZREC): x, y, the record table, scalars, pred, grow; no eps tables");
    KArgs a = make_args(p, x, y, entity_params, nullptr, nullptr, scalars, nullptr, nullptr, nullptr, nullptr, 0);
    FwdOut o{pred, partials, nullptr, grow};
    if (int rc = launch_fwd2_abs(EPS_ZREC, MODE_TRAIN, a, o, (hipStream_t)stream)) return rc;
    return after_launch("vfm_elbo_fwd_f32");
  }
  if (!x || !entity_params || !bias_params || !scalars || !pred)
    return fail(VFM_E_INVALID, "vfm_elbo_fwd_f32: NULL pointer");
  const bool train = y != nullptr;
  if (train && (!inv_occ || !W || !sumz || !grow))
    return fail(VFM_E_INVALID, "vfm_elbo_fwd_f32: y given, so inv_occ, W, sumz and grow are required");
  if (!train && (sumz || grow)) return fail(VFM_E_INVALID, "vfm_elbo_fwd_f32: sumz / grow need y");
  int eps;
  if (int rc = eps_mode(p, eps_entity, eps_bias, eps_global, &eps)) return rc;
  if (train && eps == EPS_ZERO) return fail(VFM_E_UNSUPPORTED, "vfm_elbo_fwd_f32: VFM_FLAG_EPS_ZERO is prediction-only");
  hipStream_t st = (hipStream_t)stream;
  FwdOut o{pred, partials, sumz, grow};
  Shape s;
  pick_shape(p->d, &s);
  if (use_fwd2(p, eps, true)) {        // two fields, 2..4 samples: ONE launch, the sample loop runs inside the kernel
    KArgs a = make_args(p, x, y, entity_params, bias_params, inv_occ, scalars, W, eps_entity, eps_bias, eps_global, 0);
    const int md = train ? MODE_TRAIN : MODE_PREDICT;
    if (int rc = launch_fwd2m_abs(eps, md, a, o, st)) return rc;
    return after_launch("vfm_elbo_fwd_f32");
  }
  for (int sm = 0; sm < p->n_samples; ++sm) {      // one launch per variational sample (vfm_fwd.hpp)
    KArgs a = make_args(p, x, y, entity_params, bias_params, inv_occ, scalars, W, eps_entity, eps_bias,
                        eps_global, sm);
    if (int rc = dispatch_fwd(p, s, eps, train ? MODE_TRAIN : MODE_PREDICT, p->F == 2 ? 2 : 0, a, o, st)) return rc;
  }
  return after_launch("vfm_elbo_fwd_f32");
}

int vfm_elbo_finalize_f32(const vfm_problem_t* p, double* partials, const float* scalars,
                          float* loss, void* stream) {
  if (int rc = check_problem(p)) return rc;
  if (!partials || !scalars || !loss) return fail(VFM_E_INVALID, "vfm_elbo_finalize_f32: NULL pointer");
  const double ll_scale = (double)p->nb_train / ((double)(p->B_global > 0 ? p->B_global : 1) * (double)p->n_samples);
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, (hipStream_t)stream, partials, scalars, ll_scale,
                     (int)p->flags, loss);
  return after_launch("vfm_elbo_finalize_f32");
}

int vfm_elbo_bwd_f32(const vfm_problem_t* p, const vfm_index_t* idx,
                     const float* entity_params, const float* bias_params,
                     const float* inv_occ, const float* scalars, const double* W,
                     const float* eps_entity, const float* eps_bias, const float* eps_global,
                     const float* sumz, const float* grow, const double* partials,
                     const float* grad_out, float* g_entity, float* g_bias, float* g_scalars,
                     void* stream) {
  if (int rc = check_problem(p)) return rc;
  if (int rc = check_index(p, idx, "vfm_elbo_bwd_f32")) return rc;
  if (!entity_params || !bias_params || !inv_occ || !scalars || !W || !partials || !grad_out ||
      !g_entity || !g_bias || !g_scalars || (p->B > 0 && (!sumz || !grow)))
    return fail(VFM_E_INVALID, "vfm_elbo_bwd_f32: NULL pointer");
  int eps;
  if (int rc = eps_mode(p, eps_entity, eps_bias, eps_global, &eps)) return rc;
  if (eps == EPS_ZERO) return fail(VFM_E_UNSUPPORTED, "vfm_elbo_bwd_f32: VFM_FLAG_EPS_ZERO is prediction-only");
  KArgs a = make_args(p, nullptr, nullptr, entity_params, bias_params, inv_occ, scalars, W, eps_entity,
                      eps_bias, eps_global);
  BwdArgs b = bwd_args(p, idx);
  b.sumz = sumz; b.grow = grow; b.partials = const_cast<double*>(partials); b.grad_out = grad_out;
  b.g_entity = g_entity; b.g_bias = g_bias; b.g_scalars = g_scalars;
  if (int rc = run_heavy(p, idx, sumz, grow, (hipStream_t)stream, &b)) return rc;
  AdamArgs ad;
  memset(&ad, 0, sizeof(ad));
  Shape s;
  pick_shape(p->d, &s);
  if (int rc = dispatch_bwd(p, s, eps, 0, a, b, ad, (hipStream_t)stream)) return rc;
  return after_launch("vfm_elbo_bwd_f32");
}

int vfm_elbo_bwd_adam_f32(const vfm_problem_t* p, const vfm_index_t* idx,
                          float* entity_params, float* bias_params, float* scalars,
                          const float* inv_occ, const double* W,
                          const float* eps_entity, const float* eps_bias, const float* eps_global,
                          const float* sumz, const float* grow, double* partials,
                          float* m_entity, float* v_entity, float* m_bias, float* v_bias,
                          float* m_scalars, float* v_scalars,
                          float lr, float beta1, float beta2, float eps_adam, int64_t step, float* loss,
                          void* stream) {
  if (int rc = check_problem(p, true)) return rc;
  if (int rc = check_index(p, idx, "vfm_elbo_bwd_adam_f32")) return rc;
  if (p->dev_step) {      // replayable step: lr / betas / step come from the device table (the host values are ignored)
    if ((p->flags & VFM_FLAG_SPARSE_ADAM) || eps_entity || eps_bias || eps_global)
      return fail(VFM_E_UNSUPPORTED, "vfm_elbo_bwd_adam_f32: dev_step excludes VFM_FLAG_SPARSE_ADAM and eps tables");
    step = 1; lr = 0.f;
  }
  const int rows_flags = p->flags & VFM_FLAG_ROWS_TOUCHED;
  if (rows_flags && (p->flags & VFM_FLAG_SPARSE_ADAM))
    return fail(VFM_E_INVALID, "vfm_elbo_bwd_adam_f32: VFM_FLAG_ROWS_TOUCHED does not combine with VFM_FLAG_SPARSE_ADAM");
  if (!entity_params || !bias_params || !scalars || !m_entity ||
      !v_entity || !m_bias || !v_bias || !m_scalars || !v_scalars || step < 1 ||
      !inv_occ || !W || !partials || (p->B > 0 && (!sumz || !grow)))
    return fail(VFM_E_INVALID, "vfm_elbo_bwd_adam_f32: bad argument");
  if (p->flags & VFM_FLAG_NO_PRIOR_TERMS)
    return fail(VFM_E_INVALID, "vfm_elbo_bwd_adam_f32: single-rank only (gradients never leave the kernel)");
  int eps;
  if (int rc = eps_mode(p, eps_entity, eps_bias, eps_global, &eps)) return rc;
  if (eps == EPS_ZERO) return fail(VFM_E_UNSUPPORTED, "vfm_elbo_bwd_adam_f32: VFM_FLAG_EPS_ZERO is prediction-only");
  KArgs a = make_args(p, nullptr, nullptr, entity_params, bias_params, inv_occ, scalars, W, eps_entity,
                      eps_bias, eps_global);
  a.row_filter = rows_flags == VFM_FLAG_ROWS_TOUCHED ? 2 : 0;
  BwdArgs b = bwd_args(p, idx);
  b.sumz = sumz; b.grow = grow; b.partials = partials; b.loss = loss;
  if (small_table_step(p, idx, eps, rows_flags)) {       // one launch: no pre-reduction kernels, a wave per table row
    AdamArgs ad{m_entity, v_entity, m_bias, v_bias, m_scalars, v_scalars, beta1, beta2, eps_adam, 0.f, 0.f};
    adam_consts(lr, beta1, beta2, step, &ad.step_size, &ad.bc2_sqrt);
    if (int rc = scaled_moment_consts(p, beta1, beta2, step, &ad)) return rc;
    Shape s;
    pick_shape(p->d, &s);
    if (int rc = softplus(p) ? launch_bwd_small_softplus(s, a, b, ad, idx->heavy_list, idx->heavy_threshold, (hipStream_t)stream)
                             : launch_bwd_small_abs(s, a, b, ad, idx->heavy_list, idx->heavy_threshold, (hipStream_t)stream))
      return rc;
    return after_launch("vfm_elbo_bwd_adam_f32");
  }
  Side* sd = (!(p->flags & VFM_FLAG_SPARSE_ADAM) && heavy_overlap(p, idx)) ? side_of_device() : nullptr;
  if (!sd)
    if (int rc = run_heavy(p, idx, sumz, grow, (hipStream_t)stream, &b)) return rc;
  AdamArgs ad{m_entity, v_entity, m_bias, v_bias, m_scalars, v_scalars, beta1, beta2, eps_adam, 0.f, 0.f};
  adam_consts(lr, beta1, beta2, step, &ad.step_size, &ad.bc2_sqrt);
  if (int rc = scaled_moment_consts(p, beta1, beta2, step, &ad)) return rc;
  Shape s;
  pick_shape(p->d, &s);
  int adam = (p->flags & VFM_FLAG_SPARSE_ADAM) ? 2 : 1;
  if (rows_flags == VFM_FLAG_ROWS_TOUCHED && idx->touched_ids && idx->n_touched <= p->T && env_int("VFM_ROWS_LIST", 1) != 0) {
    // the rows of the batch as a LIST (lazy exact Adam): the kernel walks it instead of scanning all T rows -- the
    // same instance as the dense step, so both agree bit for bit on the rows they share
    b.row_ids = idx->touched_ids; b.n_rows = idx->n_touched;
    a.row_filter = 0;
  }
  if (sd && a.row_filter == 0) {
    if (int rc = bwd_adam_overlapped(p, idx, s, eps, adam, a, b, ad, sumz, grow, (hipStream_t)stream, sd)) return rc;
    return after_launch("vfm_elbo_bwd_adam_f32");
  }
  if (sd)       // (a row filter of the caller's: the plain order)
    if (int rc = run_heavy(p, idx, sumz, grow, (hipStream_t)stream, &b)) return rc;
  if (int rc = dispatch_bwd(p, s, eps, adam, a, b, ad, (hipStream_t)stream)) return rc;
  return after_launch("vfm_elbo_bwd_adam_f32");
}

static int bwd_acc_impl(const vfm_problem_t* p, const vfm_index_t* idx, const int32_t* row_ids, int64_t n_rows,
                        const float* sumz, const float* grow, const double* partials, float* acc, float* sums, void* stream,
                        const char* who) {
  if (int rc = check_problem(p)) return rc;
  if (int rc = single_sample_only(p, who)) return rc;
  if (int rc = check_index(p, idx, who)) return rc;
  if (!partials || !acc || !sums || (p->B > 0 && (!sumz || !grow)))
    return fail(VFM_E_INVALID, "vfm_elbo_bwd_acc_f32: NULL pointer");
  KArgs a = make_args(p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  BwdArgs b = bwd_args(p, idx);
  b.sumz = sumz; b.grow = grow; b.partials = const_cast<double*>(partials); b.acc = acc; b.sums = sums;
  if (row_ids) {        // the listed rows only, their records written COMPACTLY (record i <-> row_ids[i])
    if (n_rows < 0 || n_rows > p->T || p->e_lo != 0 || (p->e_hi != 0 && p->e_hi != p->T))
      return fail(VFM_E_INVALID, "vfm_elbo_bwd_acc_rows_f32: 0 <= n_rows <= T, whole entity range");
    b.row_ids = row_ids; b.n_rows = n_rows; b.rec_by_slot = 1;
  }
  // (the pre-reduction covers whole lists, so with several entity chunks it runs with the first one)
  if (p->e_lo == 0)
    if (int rc = run_heavy(p, idx, sumz, grow, (hipStream_t)stream, &b)) return rc;
  if (p->e_lo != 0 && idx->n_heavy > 0) {
    b.heavy_ids = idx->heavy_ids; b.heavy_acc = idx->heavy_acc; b.n_heavy = idx->n_heavy;
    b.heavy_stride = idx->n_heavy + idx->n_items;
  }
  AdamArgs ad;
  memset(&ad, 0, sizeof(ad));
  Shape s;
  pick_shape(p->d, &s);
  if (int rc = dispatch_bwd(p, s, EPS_ZERO, 10, a, b, ad, (hipStream_t)stream)) return rc;
  return after_launch(who);
}

int vfm_elbo_bwd_acc_f32(const vfm_problem_t* p, const vfm_index_t* idx,
                         const float* sumz, const float* grow, const double* partials, float* acc,
                         float* sums, void* stream) {
  return bwd_acc_impl(p, idx, nullptr, 0, sumz, grow, partials, acc, sums, stream, "vfm_elbo_bwd_acc_f32");
}

int vfm_elbo_bwd_acc_rows_f32(const vfm_problem_t* p, const vfm_index_t* idx, const int32_t* row_ids, int64_t n_rows,
                              const float* sumz, const float* grow, const double* partials, float* acc,
                              float* sums, void* stream) {
  static const int32_t none = 0;
  if (!row_ids && n_rows != 0) return fail(VFM_E_INVALID, "vfm_elbo_bwd_acc_rows_f32: row_ids is NULL");
  return bwd_acc_impl(p, idx, row_ids ? row_ids : &none, n_rows, sumz, grow, partials, acc, sums, stream,
                      "vfm_elbo_bwd_acc_rows_f32");
}

static int apply_adam_impl(const vfm_problem_t* p, const float* acc, const float* sums,
                           float* entity_params, float* bias_params, float* scalars, const float* inv_occ,
                           const double* W, const float* eps_entity, const float* eps_bias,
                           const float* eps_global, float* m_entity, float* v_entity, float* m_bias,
                           float* v_bias, float* m_scalars, float* v_scalars, float lr, float beta1,
                           float beta2, float eps_adam, int64_t step, const int32_t* row_ids, int64_t n_rows,
                           int32_t compact, void* stream, const char* who) {
  if (int rc = check_problem(p)) return rc;
  if (int rc = single_sample_only(p, who)) return rc;
  if (!acc || !sums || !entity_params || !bias_params || !scalars || !inv_occ || !W || !m_entity ||
      !v_entity || !m_bias || !v_bias || !m_scalars || !v_scalars || step < 1)
    return fail(VFM_E_INVALID, "vfm_elbo_apply_adam_f32: bad argument");
  int eps;
  if (int rc = eps_mode(p, eps_entity, eps_bias, eps_global, &eps)) return rc;
  if (eps == EPS_ZERO) return fail(VFM_E_UNSUPPORTED, "vfm_elbo_apply_adam_f32: VFM_FLAG_EPS_ZERO is prediction-only");
  KArgs a = make_args(p, nullptr, nullptr, entity_params, bias_params, inv_occ, scalars, W, eps_entity, eps_bias,
                      eps_global);
  BwdArgs b = bwd_args(p, nullptr);
  b.acc = const_cast<float*>(acc); b.sums = const_cast<float*>(sums);
  if (row_ids) {
    if (n_rows < 0 || n_rows > p->T) return fail(VFM_E_INVALID, "vfm_elbo_apply_adam_rows_f32: 0 <= n_rows <= T");
    b.row_ids = row_ids; b.n_rows = n_rows; b.rec_by_slot = compact ? 1 : 0;
  }
  AdamArgs ad{m_entity, v_entity, m_bias, v_bias, m_scalars, v_scalars, beta1, beta2, eps_adam, 0.f, 0.f};
  adam_consts(lr, beta1, beta2, step, &ad.step_size, &ad.bc2_sqrt);
  if (int rc = scaled_moment_consts(p, beta1, beta2, step, &ad)) return rc;
  if (row_ids && (!ad.scaled || ad.store_true))
    return fail(VFM_E_INVALID, "vfm_elbo_apply_adam_rows_f32: scaled moments only, and not on the last step of a moment period "
                               "(bring every row up to date and run vfm_elbo_apply_adam_f32)");
  Shape s;
  pick_shape(p->d, &s);
  if (int rc = dispatch_bwd(p, s, eps, 11, a, b, ad, (hipStream_t)stream)) return rc;
  return after_launch(who);
}

int vfm_elbo_apply_adam_f32(const vfm_problem_t* p, const float* acc, const float* sums,
                            float* entity_params, float* bias_params, float* scalars, const float* inv_occ,
                            const double* W, const float* eps_entity, const float* eps_bias,
                            const float* eps_global, float* m_entity, float* v_entity, float* m_bias,
                            float* v_bias, float* m_scalars, float* v_scalars, float lr, float beta1,
                            float beta2, float eps_adam, int64_t step, void* stream) {
  return apply_adam_impl(p, acc, sums, entity_params, bias_params, scalars, inv_occ, W, eps_entity, eps_bias, eps_global,
                         m_entity, v_entity, m_bias, v_bias, m_scalars, v_scalars, lr, beta1, beta2, eps_adam, step,
                         nullptr, 0, 0, stream, "vfm_elbo_apply_adam_f32");
}

int vfm_elbo_apply_adam_rows_f32(const vfm_problem_t* p, const float* acc, const float* sums, const int32_t* row_ids,
                                 int64_t n_rows, int32_t compact_records, float* entity_params, float* bias_params, float* scalars,
                                 const float* inv_occ, const double* W, float* m_entity, float* v_entity, float* m_bias,
                                 float* v_bias, float* m_scalars, float* v_scalars, float lr, float beta1, float beta2,
                                 float eps_adam, int64_t step, void* stream) {
  if (!row_ids && n_rows != 0) return fail(VFM_E_INVALID, "vfm_elbo_apply_adam_rows_f32: row_ids is NULL");
  static const int32_t none = 0;
  return apply_adam_impl(p, acc, sums, entity_params, bias_params, scalars, inv_occ, W, nullptr, nullptr, nullptr,
                         m_entity, v_entity, m_bias, v_bias, m_scalars, v_scalars, lr, beta1, beta2, eps_adam, step,
                         row_ids ? row_ids : &none, n_rows, compact_records, stream,
                         "vfm_elbo_apply_adam_rows_f32");
}

int vfm_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                 float beta2, float eps, int64_t step, void* stream) {
  if (!p || !g || !m || !v || n < 0 || step < 1) return fail(VFM_E_INVALID, "vfm_adam_f32: bad argument");
  if (n == 0) return 0;
  if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) != 0)
    return fail(VFM_E_INVALID, "vfm_adam_f32: pointers must be 16-byte aligned");
  float step_size, bc2_sqrt;
  adam_consts(lr, beta1, beta2, step, &step_size, &bc2_sqrt);
  const int64_t n4 = n / 4;
  int64_t nb = (n4 + BLOCK - 1) / BLOCK;
  if (nb < 1) nb = 1;
  const int grid = (int)(nb < 4096 ? nb : 4096);
  hipLaunchKernelGGL(k_adam, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, p, g, m, v, n4, n, beta1,
                     beta2, eps, step_size, bc2_sqrt);
  return after_launch("vfm_adam_f32");
}

int vfm_sample_records_f32(const vfm_problem_t* p, const int32_t* ids, int64_t n, const float* entity_params,
                           const float* bias_params, const float* inv_occ, const double* W, float* zrec, void* stream) {
  if (int rc = check_problem(p)) return rc;
  if (int rc = single_sample_only(p, "vfm_sample_records_f32")) return rc;
  if ((p->d & 3) != 0 || softplus(p)) return fail(VFM_E_UNSUPPORTED, "vfm_sample_records_f32: d % 4 == 0, |.| link");
  if (n < 0 || n > 0x7FFFFFFFLL || (n > 0 && (!ids || !entity_params || !bias_params || !inv_occ || !W || !zrec)))
    return fail(VFM_E_INVALID, "vfm_sample_records_f32: bad argument");
  if (n == 0) return 0;
  KArgs a = make_args(p, nullptr, nullptr, entity_params, bias_params, inv_occ, nullptr, W, nullptr, nullptr, nullptr);
  Shape s;
  pick_shape(p->d, &s);
  if (int rc = launch_sample_rec_abs(s, a, ids, (int)n, zrec, (hipStream_t)stream)) return rc;
  return after_launch("vfm_sample_records_f32");
}

int vfm_elbo_bwd_adam_pipe_f32(const vfm_problem_t* p, const vfm_index_t* idx, const vfm_pipe_t* pipe,
                               float* entity_params, float* bias_params, float* scalars,
                               const float* inv_occ, const double* W, const float* grow, double* partials,
                               float* m_entity, float* v_entity, float* m_bias, float* v_bias,
                               float* m_scalars, float* v_scalars,
                               float lr, float beta1, float beta2, float eps_adam, int64_t step, float* loss,
                               void* stream) {
  if (int rc = check_problem(p, true)) return rc;
  if (p->dev_step) { step = 1; lr = 0.f; }      // (replayable step: the device table carries them; pipe->next_step too)
  if (int rc = single_sample_only(p, "vfm_elbo_bwd_adam_pipe_f32")) return rc;
  if (int rc = check_index(p, idx, "vfm_elbo_bwd_adam_pipe_f32")) return rc;
  if (p->F != 2 || (p->d & 3) != 0 || p->d > 512)
    return fail(VFM_E_UNSUPPORTED, "vfm_elbo_bwd_adam_pipe_f32: two fields, d % 4 == 0, d <= 512");
  if (p->flags & ~(VFM_FLAG_SCALED_MOMENTS | VFM_FLAG_SHARE_GPU))
    return fail(VFM_E_UNSUPPORTED, "vfm_elbo_bwd_adam_pipe_f32: only VFM_FLAG_SCALED_MOMENTS (the record step exists for the |.| link)");
  if (!pipe) return fail(VFM_E_INVALID, "vfm_elbo_bwd_adam_pipe_f32: pipe is NULL");
  if (int rc = check_struct(pipe->struct_size, pipe->abi_version, sizeof(vfm_pipe_t), "vfm_pipe_t")) return rc;
  if (!pipe->zrec || (p->B > 0 && !idx->occ_other) ||
      (pipe->zrec_next && (!pipe->next_occ_ptr || !pipe->next_W)))
    return fail(VFM_E_INVALID, "vfm_elbo_bwd_adam_pipe_f32: pipe / idx->occ_other incomplete");
  if (pipe->last_step) {      // look-ahead form
    if (!pipe->next_occ_ptr || !pipe->step_tab || !(p->flags & VFM_FLAG_SCALED_MOMENTS))
      return fail(VFM_E_INVALID, "vfm_elbo_bwd_adam_pipe_f32 (look-ahead form): next_occ_ptr, step_tab and VFM_FLAG_SCALED_MOMENTS");
    if (!p->dev_step && step % VFM_MOMENT_PERIOD == 0)
      return fail(VFM_E_INVALID, "vfm_elbo_bwd_adam_pipe_f32 (look-ahead form): the last step of a moment period is a dense one");
  }
  if (!entity_params || !bias_params || !scalars || !m_entity || !v_entity || !m_bias || !v_bias || !m_scalars ||
      !v_scalars || step < 1 || !inv_occ || !W || !partials || (p->B > 0 && !grow))
    return fail(VFM_E_INVALID, "vfm_elbo_bwd_adam_pipe_f32: bad argument");
  KArgs a = make_args(p, nullptr, nullptr, entity_params, bias_params, inv_occ, scalars, W, nullptr, nullptr, nullptr);
  BwdArgs b = bwd_args(p, idx);
  b.grow = grow; b.partials = partials; b.loss = loss;
  if (int rc = run_heavy(p, idx, nullptr, grow, (hipStream_t)stream, &b, pipe->zrec)) return rc;
  b.zrec = pipe->zrec; b.occ_other = idx->occ_other;
  b.zrec_next = pipe->zrec_next; b.next_occ_ptr = pipe->next_occ_ptr; b.next_W = pipe->next_W;
  b.next_key = a.key;
  b.next_key.step_lo = (uint32_t)pipe->next_step; b.next_key.step_hi = (uint32_t)(pipe->next_step >> 32);
  if (pipe->last_step) {
    b.last_step = pipe->last_step; b.step_tab = reinterpret_cast<float2*>(pipe->step_tab);
    b.la_step = (int32_t)step; b.la_k = (int32_t)((step - 1) % VFM_MOMENT_PERIOD + 1);
    if (idx->touched_ids && idx->n_touched > 0) {      // the rows to visit, listed (vfm_union_rows)
      if (idx->n_touched > p->T) return fail(VFM_E_INVALID, "vfm_elbo_bwd_adam_pipe_f32: more listed rows than table rows");
      b.row_ids = idx->touched_ids; b.n_rows = idx->n_touched;
    }
  }
  AdamArgs ad{m_entity, v_entity, m_bias, v_bias, m_scalars, v_scalars, beta1, beta2, eps_adam, 0.f, 0.f};
  adam_consts(lr, beta1, beta2, step, &ad.step_size, &ad.bc2_sqrt);
  if (int rc = scaled_moment_consts(p, beta1, beta2, step, &ad)) return rc;
  Shape s;
  pick_shape(p->d, &s);
  if (int rc = dispatch_bwd(p, s, EPS_PHILOX, 1, a, b, ad, (hipStream_t)stream)) return rc;
  return after_launch("vfm_elbo_bwd_adam_pipe_f32");
}

int vfm_elbo_bwd_adam_lookahead_f32(const vfm_problem_t* p, const vfm_index_t* idx,
                                    float* entity_params, float* bias_params, float* scalars,
                                    const float* inv_occ, const double* W,
                                    const float* sumz, const float* grow, double* partials,
                                    float* m_entity, float* v_entity, float* m_bias, float* v_bias,
                                    float* m_scalars, float* v_scalars,
                                    float lr, float beta1, float beta2, float eps_adam, int64_t step, float* loss,
                                    int32_t* last_step, const int32_t* next_occ_ptr, float* step_tab, void* stream) {
  if (int rc = check_problem(p, true)) return rc;
  if (p->dev_step) { step = 1; lr = 0.f; }      // (replayable step: the caller keeps period-ending steps out of its graphs)
  if (int rc = single_sample_only(p, "vfm_elbo_bwd_adam_lookahead_f32")) return rc;
  if (int rc = check_index(p, idx, "vfm_elbo_bwd_adam_lookahead_f32")) return rc;
  if ((p->flags & ~(VFM_FLAG_LINK_SOFTPLUS | VFM_FLAG_SHARE_GPU)) != VFM_FLAG_SCALED_MOMENTS)
    return fail(VFM_E_INVALID, "vfm_elbo_bwd_adam_lookahead_f32: VFM_FLAG_SCALED_MOMENTS (and optionally the link flag) only");
  if (!entity_params || !bias_params || !scalars || !m_entity || !v_entity || !m_bias || !v_bias || !m_scalars ||
      !v_scalars || step < 1 || !inv_occ || !W || !partials || (p->B > 0 && (!sumz || !grow)) || !last_step ||
      !next_occ_ptr || !step_tab)
    return fail(VFM_E_INVALID, "vfm_elbo_bwd_adam_lookahead_f32: bad argument");
  if (step % VFM_MOMENT_PERIOD == 0)
    return fail(VFM_E_INVALID, "vfm_elbo_bwd_adam_lookahead_f32: the last step of a moment period is a dense one "
                               "(bring every row up to date, then vfm_elbo_bwd_adam_f32)");
  KArgs a = make_args(p, nullptr, nullptr, entity_params, bias_params, inv_occ, scalars, W, nullptr, nullptr, nullptr);
  BwdArgs b = bwd_args(p, idx);
  b.sumz = sumz; b.grow = grow; b.partials = partials; b.loss = loss;
  Side* sd = heavy_overlap(p, idx) ? side_of_device() : nullptr;
  if (!sd)
    if (int rc = run_heavy(p, idx, sumz, grow, (hipStream_t)stream, &b)) return rc;
  b.last_step = last_step; b.next_occ_ptr = next_occ_ptr; b.step_tab = reinterpret_cast<float2*>(step_tab);
  if (idx->touched_ids && idx->n_touched > 0) {      // the rows to visit, listed (vfm_union_rows): no scan over the table
    if (idx->n_touched > p->T) return fail(VFM_E_INVALID, "vfm_elbo_bwd_adam_lookahead_f32: more listed rows than table rows");
    b.row_ids = idx->touched_ids; b.n_rows = idx->n_touched;
  }
  b.la_step = (int32_t)step; b.la_k = (int32_t)((step - 1) % VFM_MOMENT_PERIOD + 1);
  AdamArgs ad{m_entity, v_entity, m_bias, v_bias, m_scalars, v_scalars, beta1, beta2, eps_adam, 0.f, 0.f};
  adam_consts(lr, beta1, beta2, step, &ad.step_size, &ad.bc2_sqrt);
  if (int rc = scaled_moment_consts(p, beta1, beta2, step, &ad)) return rc;
  Shape s;
  pick_shape(p->d, &s);
  if (sd) {
    if (int rc = bwd_adam_overlapped(p, idx, s, EPS_PHILOX, 1, a, b, ad, sumz, grow, (hipStream_t)stream, sd)) return rc;
    return after_launch("vfm_elbo_bwd_adam_lookahead_f32");
  }
  if (int rc = dispatch_bwd(p, s, EPS_PHILOX, 1, a, b, ad, (hipStream_t)stream)) return rc;
  return after_launch("vfm_elbo_bwd_adam_lookahead_f32");
}

int vfm_adam_catchup_f32(float* entity_params, float* bias_params, const float* m_entity, const float* v_entity,
                         const float* m_bias, const float* v_bias, int32_t* last_step, const int32_t* ids, int64_t n,
                         int64_t T, int32_t d, const float* lr_of_step, int64_t n_lr, float beta1, float beta2,
                         float eps_adam, int64_t upto, int64_t mark, float* wrec, void* stream) {
  if (!entity_params || !bias_params || !m_entity || !v_entity || !m_bias || !v_bias || !last_step || n < 0 || T < 1 ||
      d < 1 || upto < 0 || mark < upto || mark > 0x7FFFFFFFLL || (ids == nullptr && n != T))
    return fail(VFM_E_INVALID, "vfm_adam_catchup_f32: bad argument (ids == NULL means all T rows: n == T)");
  if (n == 0 || upto == 0) {          // nothing to replay (before the first step); still stamp the rows
    if (n == 0) return 0;
  }
  // the steps of the moment period that contains `upto`: period_start + 1 .. upto
  const int64_t pstart = upto > 0 ? ((upto - 1) / VFM_MOMENT_PERIOD) * VFM_MOMENT_PERIOD : 0;
  const int kmax = (int)(upto - pstart);
  if (kmax > 0 && (!lr_of_step || n_lr < kmax))
    return fail(VFM_E_INVALID, "vfm_adam_catchup_f32: lr_of_step (host) needs one entry per step of the period up to `upto` (n_lr too small)");
  if (wrec && (((uintptr_t)wrec) & 15) != 0) return fail(VFM_E_INVALID, "vfm_adam_catchup_f32: wrec must be 16-byte aligned");
  CatchTab tab;
  memset(&tab, 0, sizeof(tab));
  for (int k = 1; k <= kmax; ++k) {   // exactly the constants vfm_elbo_bwd_adam_f32 forms for that step
    float step_size, bc2_sqrt;
    adam_consts(lr_of_step[k - 1], beta1, beta2, pstart + k, &step_size, &bc2_sqrt);
    const double s1 = pow((double)beta1, (double)k), s2 = pow((double)beta2, (double)k);
    if (!(s1 > 1e-30) || !(s2 > 1e-30)) return fail(VFM_E_UNSUPPORTED, "vfm_adam_catchup_f32: beta^k underflows");
    float a1, q2;
    scaled_step_consts(step_size, bc2_sqrt, s1, s2, &a1, &q2);
    tab.c[k] = make_float4(a1, q2, 0.f, 0.f);
  }
  int64_t nb = (n + BLOCK / 64 - 1) / (BLOCK / 64);
  if (nb > 8192) nb = 8192;
  if ((d & 1) == 0)
    hipLaunchKernelGGL(k_adam_catchup<4>, dim3((unsigned)nb), dim3(BLOCK), 0, (hipStream_t)stream, entity_params, bias_params,
                       m_entity, v_entity, m_bias, v_bias, last_step, ids, n, (int)d, (int32_t)pstart, (int32_t)upto,
                       (int32_t)mark, eps_adam, tab, wrec);
  else
    hipLaunchKernelGGL(k_adam_catchup<1>, dim3((unsigned)nb), dim3(BLOCK), 0, (hipStream_t)stream, entity_params, bias_params,
                       m_entity, v_entity, m_bias, v_bias, last_step, ids, n, (int)d, (int32_t)pstart, (int32_t)upto,
                       (int32_t)mark, eps_adam, tab, wrec);
  return after_launch("vfm_adam_catchup_f32");
}

int vfm_moments_rescale_f32(float* m, float* v, int64_t n, float beta1, float beta2, int64_t step, int32_t to_scaled,
                            void* stream) {
  if (!m || !v || n < 0 || step < 0) return fail(VFM_E_INVALID, "vfm_moments_rescale_f32: bad argument");
  const int64_t k = step % VFM_MOMENT_PERIOD;
  if (k == 0 || n == 0) return 0;       // at a period boundary both forms coincide
  double c1 = pow((double)beta1, (double)k), c2 = pow((double)beta2, (double)k);
  if (!(c1 > 1e-30) || !(c2 > 1e-30)) return fail(VFM_E_UNSUPPORTED, "vfm_moments_rescale_f32: beta^k underflows");
  if (to_scaled) { c1 = 1.0 / c1; c2 = 1.0 / c2; }
  int64_t nb = (n + BLOCK - 1) / BLOCK;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(k_rescale2, dim3((unsigned)nb), dim3(BLOCK), 0, (hipStream_t)stream, m, v, n, (float)c1, (float)c2);
  return after_launch("vfm_moments_rescale_f32");
}

int vfm_philox_eps_f32(const vfm_problem_t* p, float* eps_entity, float* eps_bias, float* eps_global,
                       void* stream) {
  if (int rc = check_problem(p)) return rc;
  if (!eps_entity || !eps_bias || !eps_global) return fail(VFM_E_INVALID, "vfm_philox_eps_f32: NULL pointer");
  for (int sm = 0; sm < p->n_samples; ++sm) {     // tables of sample s: the s-th [T,d] / [T] / [1] blocks
    KArgs a = make_args(p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                        nullptr, sm);
    hipLaunchKernelGGL(k_philox_dump, dim3(1024), dim3(256), 0, (hipStream_t)stream, a,
                       eps_entity + (size_t)sm * (size_t)p->T * (size_t)p->d, eps_bias + (size_t)sm * (size_t)p->T,
                       eps_global + sm);
  }
  return after_launch("vfm_philox_eps_f32");
}

}  // extern "C"
