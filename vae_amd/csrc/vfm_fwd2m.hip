// vfm_fwd2m.hip -- k_fwd2m instances (the two-field task-stream forward with 2..4 variational samples inside the
// kernel, vfm_fwd2m.hpp) and their dispatch.  Compiled once per link function.  gfx950 only, wave = 64.
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
#include "vfm_fwd2m.hpp"

constexpr int LINK = VFM_LINK;

template <int LPE, bool FULL, int EPS, int MODE>
int launch_fwd2m_t(KArgs& a, const FwdOut& o, hipStream_t st) {
  constexpr int GPB = BLOCK / LPE;
  const int per_cu = env_int("VFM_FWD2_BLOCKS_PER_CU", 4);
  int64_t nb = (a.B + GPB - 1) / GPB;
  int64_t cap = 256LL * per_cu;
  if (cap > VFM_MAX_FWD_BLOCKS) cap = VFM_MAX_FWD_BLOCKS;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  if (a.id64)
    hipLaunchKernelGGL((k_fwd2m<LPE, FULL, EPS, MODE, true, LINK>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, o);
  else
    hipLaunchKernelGGL((k_fwd2m<LPE, FULL, EPS, MODE, false, LINK>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, o);
  return 0;
}

template <int LPE, bool FULL>
int launch_fwd2m_s(int eps, int mode, KArgs& a, const FwdOut& o, hipStream_t st) {
  if (eps == EPS_PHILOX && mode == MODE_TRAIN) return launch_fwd2m_t<LPE, FULL, EPS_PHILOX, MODE_TRAIN>(a, o, st);
  if (eps == EPS_TABLE && mode == MODE_TRAIN) return launch_fwd2m_t<LPE, FULL, EPS_TABLE, MODE_TRAIN>(a, o, st);
  if (eps == EPS_PHILOX && mode == MODE_PREDICT) return launch_fwd2m_t<LPE, FULL, EPS_PHILOX, MODE_PREDICT>(a, o, st);
  if (eps == EPS_TABLE && mode == MODE_PREDICT) return launch_fwd2m_t<LPE, FULL, EPS_TABLE, MODE_PREDICT>(a, o, st);
  return fail(VFM_E_UNSUPPORTED, "forward (task stream, S > 1): unsupported eps source / mode combination");
}

int dispatch_fwd2m(int eps, int mode, KArgs& a, const FwdOut& o, hipStream_t st) {
  if (a.S < 2 || a.S > FWD2M_MAXS) return fail(VFM_E_UNSUPPORTED, "forward (task stream, S > 1): 2 to 4 samples");
  const int C = a.d / 4, P = (C + 1) / 2;
  int lpe = 1;
  while (lpe < P) lpe <<= 1;
  const bool full = a.d == 8 * lpe;
#define X(L_)                                                                        \
  if (lpe == L_) return full ? launch_fwd2m_s<L_, true>(eps, mode, a, o, st)         \
                             : launch_fwd2m_s<L_, false>(eps, mode, a, o, st);
  X(1) X(2) X(4) X(8) X(16) X(32) X(64)
#undef X
  return fail(VFM_E_UNSUPPORTED, "forward (task stream): embedding size above 512");
}

}  // namespace

#if VFM_LINK == 0
int launch_fwd2m_abs(int eps, int mode, KArgs& a, const FwdOut& o, hipStream_t st) {
  return dispatch_fwd2m(eps, mode, a, o, st);
}
#else
int launch_fwd2m_softplus(int eps, int mode, KArgs& a, const FwdOut& o, hipStream_t st) {
  return dispatch_fwd2m(eps, mode, a, o, st);
}
#endif

}  // namespace vfm
