// vfm_reduce.hpp -- reduction of the forward's per-workgroup partial sums + the loss triple.
// Included inside `namespace vfm { namespace {` of vfm_abi.hip (k_finalize) and vfm_bwd.hip (folded
// into the fused backward).
#pragma once

// Reduce the forward's per-workgroup slots into partials[0..4] and form the loss triple.  Called by
// all BLOCK threads of ONE workgroup; the totals are valid in thread 0 (and in memory) afterwards.
__device__ __forceinline__ void reduce_slots_and_loss(double* __restrict__ partials,
                                                      const float* __restrict__ scalars, double ll_scale,
                                                      int flags, float* __restrict__ loss, double (*sh)[BLOCK / 64] /* [7] */,
                                                      double (&tot)[6]) {
  const int nblk = (int)partials[7];
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};                // [6]: Normal-likelihood terms behind the slots' alpha sums
  for (int b = threadIdx.x; b < nblk; b += BLOCK) {
    const double* slot = partials + VFM_N_PARTIALS * (1 + (size_t)b);
#pragma unroll
    for (int i = 0; i < 7; ++i) acc[i] += slot[i];
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc[i] += __shfl_xor(acc[i], m, 64);
    if ((threadIdx.x & 63) == 0) sh[i][threadIdx.x >> 6] = acc[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    tot[i] = 0;
    for (int w = 0; w < BLOCK / 64; ++w) tot[i] += sh[i][w];
  }
  {   // dloss/d|alpha| (up to nb_train / B): sum (y - pred)^2 / 2 - n / (2 |alpha|), the difference formed once, in fp64
    double nterms = 0;
    for (int w = 0; w < BLOCK / 64; ++w) nterms += sh[VFM_SLOT_NTERMS][w];
    const double al = scalars[0];
    const double aabs = (flags & VFM_FLAG_LINK_SOFTPLUS) ? fmax(al, 0.0) + log1p(exp(-fabs(al))) : fabs(al);
    if (nterms > 0.0) tot[VFM_P_ALPHA] -= 0.5 * nterms / aabs;
  }
  if (threadIdx.x != 0) return;
#pragma unroll
  for (int i = 0; i < 6; ++i) partials[i] = tot[i];
  partials[VFM_P_REDUCED] = 1.0;
  const double m0 = scalars[1], s0 = scalars[2];
  // sigma_0 = link(s0): |s0| or softplus(s0)
  const double sg0 = (flags & VFM_FLAG_LINK_SOFTPLUS) ? fmax(s0, 0.0) + log1p(exp(-fabs(s0))) : fabs(s0);
  const double kl0 = (flags & VFM_FLAG_NO_PRIOR_TERMS)
                         ? 0.0
                         : 0.5 * (sg0 * sg0 + m0 * m0 - 1.0) - log(fmax(sg0, (double)SIGMA_MIN));
  const double nll = -ll_scale * tot[VFM_P_LL];
  const double kl = kl0 + tot[VFM_P_KL];
  const bool bad = tot[VFM_P_BADID] != 0.0;
  const float nanv = __builtin_nanf("");
  loss[0] = bad ? nanv : (float)(nll + kl);
  loss[1] = bad ? nanv : (float)nll;
  loss[2] = bad ? nanv : (float)kl;
}

