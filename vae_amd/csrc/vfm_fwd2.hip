// vfm_fwd2.hip -- k_fwd2 instances (the two-field forward as a stream of sampling tasks, vfm_fwd2.hpp) and
// their dispatch.  Compiled once per link function: -DVFM_LINK=0 (|.|, vfm-torch.py:126) and -DVFM_LINK=1
// (softplus, :125).  gfx950 only, wave = 64.
#include <math.h>

#include "vfm_args.hpp"

#ifndef VFM_LINK
#error "compile with -DVFM_LINK=0 (abs) or -DVFM_LINK=1 (softplus)"
#endif

namespace vfm {
namespace {

#include "vfm_rng.hpp"
#include "vfm_common.hpp"
typedef float v2f __attribute__((ext_vector_type(2)));
#include "vfm_fwd2.hpp"

constexpr int LINK = VFM_LINK;

template <int LPE, bool FULL, int EPS, int MODE>
int launch_fwd2_t(KArgs& a, const FwdOut& o, hipStream_t st) {
  constexpr int GPB = BLOCK / LPE;
  // enough groups that every SIMD holds several waves; each group then owns a contiguous range of rows
  const int per_cu = env_int("VFM_FWD2_BLOCKS_PER_CU", 4);
  int64_t nb = (a.B + GPB - 1) / GPB;
  int64_t cap = 256LL * per_cu;
  if (cap > VFM_MAX_FWD_BLOCKS) cap = VFM_MAX_FWD_BLOCKS;
  if (nb > cap) nb = cap;
  { const int share = env_int("VFM_SHARE_CAP", 960); if ((a.flags & VFM_FLAG_SHARE_GPU) && share > 0 && nb > share) nb = share; }       // (leave workgroup slots to another stream: include/vfm_hip.h)
  { const int forced = env_int("VFM_FWD2_GRID", 0); if (forced > 0 && nb > forced) nb = forced; }      // (A/B runs)
  if (nb < 1) nb = 1;
  if constexpr (EPS == EPS_PHILOX && MODE == MODE_TRAIN) {
    if (a.wrec) {       // the packed first-order records are given: one 16-byte record per task (vfm_problem_t.wrec)
      if (a.id64)
        hipLaunchKernelGGL((k_fwd2<LPE, FULL, EPS, MODE, true, LINK, true>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, o);
      else
        hipLaunchKernelGGL((k_fwd2<LPE, FULL, EPS, MODE, false, LINK, true>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, o);
      return 0;
    }
  }
  if (a.id64)
    hipLaunchKernelGGL((k_fwd2<LPE, FULL, EPS, MODE, true, LINK>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, o);
  else
    hipLaunchKernelGGL((k_fwd2<LPE, FULL, EPS, MODE, false, LINK>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, o);
  return 0;
}

template <int LPE, bool FULL>
int launch_fwd2_s(int eps, int mode, KArgs& a, const FwdOut& o, hipStream_t st) {
#define FWD2(E_, M_) \
  if (eps == E_ && mode == M_) return launch_fwd2_t<LPE, FULL, E_, M_>(a, o, st);
  FWD2(EPS_PHILOX, MODE_TRAIN) FWD2(EPS_TABLE, MODE_TRAIN) FWD2(EPS_ZERO, MODE_TRAIN) FWD2(EPS_ZREC, MODE_TRAIN)
  FWD2(EPS_PHILOX, MODE_PREDICT) FWD2(EPS_TABLE, MODE_PREDICT) FWD2(EPS_ZERO, MODE_PREDICT)
#undef FWD2
  return fail(VFM_E_UNSUPPORTED, "forward (task stream): unsupported eps source / mode combination");
}

int dispatch_fwd2(int eps, int mode, KArgs& a, const FwdOut& o, hipStream_t st) {
  const int C = a.d / 4, P = (C + 1) / 2;          // lane p owns the chunk pair (2p, 2p+1)
  int lpe = 1;
  while (lpe < P) lpe <<= 1;
  const bool full = a.d == 8 * lpe;
#define X(L_)                                                                       \
  if (lpe == L_) return full ? launch_fwd2_s<L_, true>(eps, mode, a, o, st)         \
                             : launch_fwd2_s<L_, false>(eps, mode, a, o, st);
  X(1) X(2) X(4) X(8) X(16) X(32) X(64)
#undef X
  return fail(VFM_E_UNSUPPORTED, "forward (task stream): embedding size above 512");
}

}  // namespace

#if VFM_LINK == 0
int launch_fwd2_abs(int eps, int mode, KArgs& a, const FwdOut& o, hipStream_t st) {
  return dispatch_fwd2(eps, mode, a, o, st);
}
#else
int launch_fwd2_softplus(int eps, int mode, KArgs& a, const FwdOut& o, hipStream_t st) {
  return dispatch_fwd2(eps, mode, a, o, st);
}
#endif

}  // namespace vfm
