// vfm_bwd.hip -- k_bwd (entity-centric gradients / fused Adam / multi-rank stages) and k_sample_rec
// instances with their dispatch.  Compiled once per link function -- -DVFM_LINK=0 (|.|) / 1 (softplus) -- and per
// kernel family: -DVFM_BWD_PART=0 the gradient and statistics forms (`adam` 0, 10), =1 the fused-Adam forms (1, 2, 11;
// with the pipelined / look-ahead instances and the record sampler, |.| link only for the pipelined ones): four objects
// of similar compile time instead of two long ones.
// gfx950 only, wave = 64.  See vfm_args.hpp for how libvfm_hip.so is split into translation units.
#include <math.h>

#include "vfm_args.hpp"

#ifndef VFM_LINK
#error "compile with -DVFM_LINK=0 (abs) or -DVFM_LINK=1 (softplus)"
#endif
#ifndef VFM_BWD_PART
#error "compile with -DVFM_BWD_PART=0 (gradient / statistics forms) or =1 (fused-Adam forms)"
#endif

namespace vfm {
namespace {

#include "vfm_rng.hpp"
#include "vfm_common.hpp"
#include "vfm_reduce.hpp"
#include "vfm_bwd.hpp"
#include "vfm_sample.hpp"
#if VFM_BWD_PART == 1
#include "vfm_bwd_small.hpp"
#endif

constexpr int LINK = VFM_LINK;

template <int LPE, int CPL, int VEC, int EPS, int ADAM, int STAGE = STAGE_FULL>
int launch_bwd_t(KArgs& a, const BwdArgs& b, const AdamArgs& ad, hipStream_t st) {
  constexpr int GPB = BLOCK / LPE;
  const int per_cu = env_int("VFM_BWD_BLOCKS_PER_CU", 8);
  int64_t nb = (a.e_hi - a.e_lo + GPB - 1) / GPB;
  if (ADAM != 0 && b.row_ids) nb = (b.n_rows + GPB - 1) / GPB;     // the listed rows only
  const int64_t cap = 256LL * per_cu;
  if (nb > cap) nb = cap;
  // VFM_FLAG_SHARE_GPU: 15/16 of the 1,024 workgroups the chip holds (126 VGPRs: four per CU), so that another stream's small
  // kernels (256 threads, <= 128 VGPRs: csrc/vfm_index.hip) find a free slot on 64 CUs while this one runs
  { const int share = env_int("VFM_SHARE_CAP", 960); if ((a.flags & VFM_FLAG_SHARE_GPU) && share > 0 && nb > share) nb = share; }
  { const int forced = env_int("VFM_BWD_GRID", 0); if (forced > 0 && nb > forced) nb = forced; }      // (A/B runs)
  if (nb < 1) nb = 1;
  if constexpr (STAGE == STAGE_FULL && ADAM == 1 && EPS == EPS_PHILOX && VEC == 4 && LINK == LINK_ABS) {
    if (b.zrec != nullptr) {      // software-pipelined step: gathers samples, writes the next step's records
      if (a.S > 1) return fail(VFM_E_UNSUPPORTED, "pipelined step: one variational sample");
      if (b.last_step != nullptr)     // ... in the look-ahead form: rows in neither this batch nor the next are skipped
        hipLaunchKernelGGL((k_bwd<LPE, CPL, VEC, EPS, ADAM, STAGE, LINK, false, true, true>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, b, ad);
      else
        hipLaunchKernelGGL((k_bwd<LPE, CPL, VEC, EPS, ADAM, STAGE, LINK, false, true>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, b, ad);
      return 0;
    }
  }
  if (b.zrec != nullptr) return fail(VFM_E_UNSUPPORTED, "pipelined step: fused Adam, Philox eps, d % 4 == 0, |.| link only");
  if constexpr (STAGE == STAGE_FULL && ADAM == 1 && EPS == EPS_PHILOX) {
    if (b.last_step != nullptr) {     // look-ahead lazy Adam
      if (a.S > 1) return fail(VFM_E_UNSUPPORTED, "look-ahead lazy Adam: one variational sample");
      hipLaunchKernelGGL((k_bwd<LPE, CPL, VEC, EPS, ADAM, STAGE, LINK, false, false, true>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, b, ad);
      return 0;
    }
  }
  if (b.last_step != nullptr) return fail(VFM_E_UNSUPPORTED, "look-ahead lazy Adam: fused Adam with Philox eps only");
  if constexpr (STAGE == STAGE_FULL) {
    if (a.S > 1) {       // variational samples: the instance with the per-sample walk
      hipLaunchKernelGGL((k_bwd<LPE, CPL, VEC, EPS, ADAM, STAGE, LINK, true>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, b, ad);
      return 0;
    }
  }
  hipLaunchKernelGGL((k_bwd<LPE, CPL, VEC, EPS, ADAM, STAGE, LINK, false>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, b, ad);
  return 0;
}

// adam: 0 gradients, 1 dense Adam fused, 2 row-sparse Adam fused, 10 statistics (STAGE_ACC), 11 apply (STAGE_APPLY)
template <int LPE, int CPL, int VEC>
int launch_bwd_s(int eps, int adam, KArgs& a, const BwdArgs& b, const AdamArgs& ad, hipStream_t st) {
#if VFM_BWD_PART == 0
  if (eps == EPS_PHILOX && adam == 0) return launch_bwd_t<LPE, CPL, VEC, EPS_PHILOX, 0>(a, b, ad, st);
  if (eps == EPS_TABLE && adam == 0) return launch_bwd_t<LPE, CPL, VEC, EPS_TABLE, 0>(a, b, ad, st);
  if (adam == 10) return launch_bwd_t<LPE, CPL, VEC, EPS_ZERO, 0, STAGE_ACC>(a, b, ad, st);
#else
  if (eps == EPS_PHILOX && adam == 1) return launch_bwd_t<LPE, CPL, VEC, EPS_PHILOX, 1>(a, b, ad, st);
  if (eps == EPS_TABLE && adam == 1) return launch_bwd_t<LPE, CPL, VEC, EPS_TABLE, 1>(a, b, ad, st);
  if (eps == EPS_PHILOX && adam == 2) return launch_bwd_t<LPE, CPL, VEC, EPS_PHILOX, 2>(a, b, ad, st);
  if (eps == EPS_TABLE && adam == 2) return launch_bwd_t<LPE, CPL, VEC, EPS_TABLE, 2>(a, b, ad, st);
  if (eps == EPS_PHILOX && adam == 11) return launch_bwd_t<LPE, CPL, VEC, EPS_PHILOX, 1, STAGE_APPLY>(a, b, ad, st);
  if (eps == EPS_TABLE && adam == 11) return launch_bwd_t<LPE, CPL, VEC, EPS_TABLE, 1, STAGE_APPLY>(a, b, ad, st);
#endif
  return fail(VFM_E_UNSUPPORTED, "backward: unsupported eps source / form in this translation unit");
}

int dispatch_bwd(const Shape& s, int eps, int adam, KArgs& a, const BwdArgs& b, const AdamArgs& ad,
                 hipStream_t st) {
#define X(L_, C_, V_) \
  if (s.lpe == L_ && s.cpl == C_ && s.vec == V_) return launch_bwd_s<L_, C_, V_>(eps, adam, a, b, ad, st);
  VFM_FOR_SHAPES(X)
#undef X
  return fail(VFM_E_UNSUPPORTED, "no kernel instance for this embedding size");
}

#if VFM_BWD_PART == 1 && VFM_LINK == 0
int dispatch_sample_rec(const Shape& s, KArgs& a, const int32_t* ids, int n, float* zrec, hipStream_t st) {
#define X(L_, C_, V_)                                                                                   \
  if (s.lpe == L_ && s.cpl == C_ && s.vec == V_) {                                                      \
    constexpr int GPB = BLOCK / L_;                                                                     \
    int64_t nb = ((int64_t)n + GPB - 1) / GPB;                                                          \
    if (nb > 65535) nb = 65535;                                                                         \
    hipLaunchKernelGGL((k_sample_rec<L_, C_, V_, LINK>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, ids, n, zrec); \
    return 0;                                                                                           \
  }
  VFM_FOR_SHAPES(X)
#undef X
  return fail(VFM_E_UNSUPPORTED, "no kernel instance for this embedding size");
}

#endif

}  // namespace

#if VFM_BWD_PART == 1 && VFM_LINK == 0
int launch_sample_rec_abs(const Shape& s, KArgs& a, const int32_t* ids, int n, float* zrec, hipStream_t st) {
  return dispatch_sample_rec(s, a, ids, n, zrec, st);
}
#endif

#if VFM_BWD_PART == 1
namespace {
int dispatch_bwd_small(const Shape& s, KArgs& a, const BwdArgs& b, const AdamArgs& ad, int L, int THR, hipStream_t st) {
  if (s.vec != 4 || s.cpl != 1) return fail(VFM_E_UNSUPPORTED, "small-table backward: d % 4 == 0, d <= 256");
  int64_t nb = (a.T + BLOCK / 64 - 1) / (BLOCK / 64);       // one wave per table row
  if (nb > 256 * 16) nb = 256 * 16;
#define X(L_)                                                                                                              \
  if (s.lpe == L_) {                                                                                                       \
    hipLaunchKernelGGL((k_bwd_small<L_, LINK>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, b, ad, L, THR);                 \
    return 0;                                                                                                              \
  }
  X(4) X(8) X(16) X(32) X(64)
#undef X
  return fail(VFM_E_UNSUPPORTED, "small-table backward: no instance for this embedding size");
}
}  // namespace
#if VFM_LINK == 0
int launch_bwd_small_abs(const Shape& s, KArgs& a, const BwdArgs& b, const AdamArgs& ad, int L, int THR, hipStream_t st) {
  return dispatch_bwd_small(s, a, b, ad, L, THR, st);
}
#else
int launch_bwd_small_softplus(const Shape& s, KArgs& a, const BwdArgs& b, const AdamArgs& ad, int L, int THR, hipStream_t st) {
  return dispatch_bwd_small(s, a, b, ad, L, THR, st);
}
#endif
#endif

#define VFM_CAT3(a, b, c) a##b##c
#define VFM_BWD_NAME(part, suffix) VFM_CAT3(launch_bwd, part, suffix)
#if VFM_LINK == 0
int VFM_BWD_NAME(VFM_BWD_PART, _abs)(const Shape& s, int eps, int adam, KArgs& a, const BwdArgs& b, const AdamArgs& ad, hipStream_t st) {
  return dispatch_bwd(s, eps, adam, a, b, ad, st);
}
#else
int VFM_BWD_NAME(VFM_BWD_PART, _softplus)(const Shape& s, int eps, int adam, KArgs& a, const BwdArgs& b, const AdamArgs& ad, hipStream_t st) {
  return dispatch_bwd(s, eps, adam, a, b, ad, st);
}
#endif

}  // namespace vfm
