// vfm_fwd.hip -- k_fwd instances (gather -> reparameterised sample -> FM -> ELBO) and their dispatch.
// Compiled once per link function: -DVFM_LINK=0 (|.|, vfm-torch.py:126) and -DVFM_LINK=1 (softplus, :125).
// gfx950 only, wave = 64.  See vfm_args.hpp for how libvfm_hip.so is split into translation units.
#include <math.h>

#include "vfm_args.hpp"

#ifndef VFM_LINK
#error "compile with -DVFM_LINK=0 (abs) or -DVFM_LINK=1 (softplus)"
#endif

namespace vfm {
namespace {

#include "vfm_rng.hpp"
#include "vfm_common.hpp"
#include "vfm_fwd.hpp"

constexpr int LINK = VFM_LINK;

// ---- forward dispatch: shape x eps source x mode x (F == 2 ?) ----
template <int LPE, int CPL, int VEC, int EPS, int MODE, int FF, bool ID64>
int launch_fwd_t(KArgs& a, const FwdOut& o, hipStream_t st) {
  constexpr int GPB = BLOCK / LPE;
  // persistent-ish grid: enough groups that each owns a few rows (pipelined), capped at
  // VFM_FWD_BLOCKS_PER_CU resident workgroups on each of the 256 CUs
  const int per_cu = env_int("VFM_FWD_BLOCKS_PER_CU", 4);
  int64_t nb = (a.B + GPB - 1) / GPB;
  int64_t cap = 256LL * per_cu;
  if (cap > VFM_MAX_FWD_BLOCKS) cap = VFM_MAX_FWD_BLOCKS;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL((k_fwd<LPE, CPL, VEC, EPS, MODE, FF, ID64, LINK>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, o);
  return 0;
}

template <int LPE, int CPL, int VEC>
int launch_fwd_s(int eps, int mode, int ff, KArgs& a, const FwdOut& o, hipStream_t st) {
#define FWD(E_, M_)                                                                            \
  if (eps == E_ && mode == M_) {                                                               \
    if constexpr (VEC == 4) {                                                                  \
      if constexpr (LPE >= 2) {                                                                \
        if (ff == 2 && a.id64) return launch_fwd_t<LPE, CPL, VEC, E_, M_, 2, true>(a, o, st);  \
        if (ff == 2) return launch_fwd_t<LPE, CPL, VEC, E_, M_, 2, false>(a, o, st);           \
      }                                                                                        \
    }                                                                                          \
    return launch_fwd_t<LPE, CPL, VEC, E_, M_, 0, true>(a, o, st);                             \
  }
  FWD(EPS_PHILOX, MODE_TRAIN) FWD(EPS_TABLE, MODE_TRAIN)
  FWD(EPS_PHILOX, MODE_PREDICT) FWD(EPS_TABLE, MODE_PREDICT) FWD(EPS_ZERO, MODE_PREDICT)
#undef FWD
  return fail(VFM_E_UNSUPPORTED, "forward: unsupported eps source / mode combination");
}

int dispatch_fwd(const Shape& s, int eps, int mode, int ff, KArgs& a, const FwdOut& o, hipStream_t st) {
#define X(L_, C_, V_) \
  if (s.lpe == L_ && s.cpl == C_ && s.vec == V_) return launch_fwd_s<L_, C_, V_>(eps, mode, ff, a, o, st);
  VFM_FOR_SHAPES(X)
#undef X
  return fail(VFM_E_UNSUPPORTED, "no kernel instance for this embedding size");
}

}  // namespace

#if VFM_LINK == 0
int launch_fwd_abs(const Shape& s, int eps, int mode, int ff, KArgs& a, const FwdOut& o, hipStream_t st) {
  return dispatch_fwd(s, eps, mode, ff, a, o, st);
}
#else
int launch_fwd_softplus(const Shape& s, int eps, int mode, int ff, KArgs& a, const FwdOut& o, hipStream_t st) {
  return dispatch_fwd(s, eps, mode, ff, a, o, st);
}
#endif

}  // namespace vfm
