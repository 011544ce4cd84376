// vfm_fwdg.hip -- k_fwdg instances (the general-F forward with a row's fields split over lane groups, vfm_fwdg.hpp)
// and their dispatch.  Compiled once per link function: -DVFM_LINK=0 (|.|, vfm-torch.py:126) and -DVFM_LINK=1
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
#include "vfm_fwdg.hpp"

constexpr int LINK = VFM_LINK;

template <int LPE, bool FULL, int EPS, int MODE, bool WREC>
int launch_fwdg_t(KArgs& a, const FwdOut& o, hipStream_t st) {
  constexpr int GPB = BLOCK / LPE;
  // segments per row: from (F, d) only, so a row is summed in the same order whatever batch it arrives in
  int ns = 1;
  while (ns < GPB && ns * 8 < a.F) ns <<= 1;
  const int seg = (a.F + ns - 1) / ns;
  const int rpb = GPB / ns;
  int64_t nb = (a.B + rpb - 1) / rpb;
  if (nb > VFM_MAX_FWD_BLOCKS) nb = VFM_MAX_FWD_BLOCKS;
  if (nb < 1) nb = 1;
  if (a.id64)
    hipLaunchKernelGGL((k_fwdg<LPE, FULL, EPS, MODE, true, LINK, WREC>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, o, ns, seg);
  else
    hipLaunchKernelGGL((k_fwdg<LPE, FULL, EPS, MODE, false, LINK, WREC>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, o, ns, seg);
  return 0;
}

template <int LPE, bool FULL>
int launch_fwdg_s(int eps, int mode, KArgs& a, const FwdOut& o, hipStream_t st) {
  if (eps == EPS_PHILOX && mode == MODE_TRAIN)
    return a.wrec ? launch_fwdg_t<LPE, FULL, EPS_PHILOX, MODE_TRAIN, true>(a, o, st)
                  : launch_fwdg_t<LPE, FULL, EPS_PHILOX, MODE_TRAIN, false>(a, o, st);
  if (eps == EPS_PHILOX && mode == MODE_PREDICT) return launch_fwdg_t<LPE, FULL, EPS_PHILOX, MODE_PREDICT, false>(a, o, st);
  if (eps == EPS_ZERO && mode == MODE_PREDICT) return launch_fwdg_t<LPE, FULL, EPS_ZERO, MODE_PREDICT, false>(a, o, st);
  return fail(VFM_E_UNSUPPORTED, "forward (split rows): unsupported eps source / mode combination");
}

int dispatch_fwdg(int eps, int mode, KArgs& a, const FwdOut& o, hipStream_t st) {
  const int C = a.d / 4, P = (C + 1) / 2;          // lane p owns the chunk pair (2p, 2p+1)
  int lpe = 2;
  while (lpe < P) lpe <<= 1;
  const bool full = a.d == 8 * lpe;
#define X(L_)                                                                       \
  if (lpe == L_) return full ? launch_fwdg_s<L_, true>(eps, mode, a, o, st)         \
                             : launch_fwdg_s<L_, false>(eps, mode, a, o, st);
  X(2) X(4) X(8) X(16) X(32) X(64)
#undef X
  return fail(VFM_E_UNSUPPORTED, "forward (split rows): embedding size above 512");
}

}  // namespace

#if VFM_LINK == 0
int launch_fwdg_abs(int eps, int mode, KArgs& a, const FwdOut& o, hipStream_t st) {
  return dispatch_fwdg(eps, mode, a, o, st);
}
#else
int launch_fwdg_softplus(int eps, int mode, KArgs& a, const FwdOut& o, hipStream_t st) {
  return dispatch_fwdg(eps, mode, a, o, st);
}
#endif

}  // namespace vfm
