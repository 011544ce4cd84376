// vfm_small_kernels.hpp -- k_zero_f64, k_inv_occ, k_norms, slot reduction + k_finalize.
// Included inside `namespace vfm { namespace {` of vfm_abi.hip.
#pragma once

// ---------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------
__global__ void k_zero_f64(double* __restrict__ p, int n) {
  if ((int)threadIdx.x < n) p[threadIdx.x] = 0.0;
}

__global__ void k_inv_occ(const int64_t* __restrict__ occ, float* __restrict__ inv, int64_t T) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < T;
       i += (int64_t)gridDim.x * blockDim.x)
    inv[i] = 1.0f / (float)occ[i];
}

// W[f] = sum_r inv_occ[x[r,f]], bitwise reproducible AND parallel: every workgroup sums its rows of column f
// in fp64 in a fixed order, converts the partial sum to 2^-31 fixed point and adds it with a 64-bit INTEGER
// atomic (integer addition is associative, so the order in which workgroups arrive does not matter; 1/occ <= 1
// and B < 2^31 keep the sum below 2^62; the rounding is <= 2^-32 per workgroup, 1e-10 of W).  k_norms_fix
// turns the accumulators into doubles in place.  Once per batch, outside the step.
constexpr double NORMS_SCALE = 2147483648.0;   // 2^31
__global__ __launch_bounds__(BLOCK) void k_norms(const void* __restrict__ x, int id64,
                                                 const float* __restrict__ inv_occ, int64_t B, int F,
                                                 int64_t T, unsigned long long* __restrict__ Wfix) {
  __shared__ double sh[BLOCK / 64];
  const int f = blockIdx.y;
  double acc = 0.0;
  for (int64_t r = blockIdx.x * (int64_t)BLOCK + threadIdx.x; r < B; r += (int64_t)gridDim.x * BLOCK) {
    const int64_t o = r * F + f;
    const int64_t id = id64 ? ((const int64_t*)x)[o] : (int64_t)((const int32_t*)x)[o];
    if (id >= 0 && id < T) acc += (double)inv_occ[id];
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < BLOCK / 64; ++w) t += sh[w];
    atomicAdd(&Wfix[f], (unsigned long long)(long long)__double2ll_rn(t * NORMS_SCALE));
  }
}

__global__ void k_norms_fix(double* __restrict__ W, int F) {
  if ((int)threadIdx.x < F) {
    const long long v = reinterpret_cast<const long long*>(W)[threadIdx.x];
    W[threadIdx.x] = (double)v / NORMS_SCALE;
  }
}

__global__ __launch_bounds__(BLOCK) void k_finalize(double* __restrict__ partials,
                                                    const float* __restrict__ scalars, double ll_scale,
                                                    int flags, float* __restrict__ loss) {
  __shared__ double sh[6][BLOCK / 64];
  double tot[6];
  reduce_slots_and_loss(partials, scalars, ll_scale, flags, loss, sh, tot);
}

// Dimension-sharded mode, after the all-reduce of the row values: pred[r] <- w0 + pred[r], the likelihood
// terms and grow[r] = dloss/dpred_r, per-workgroup sums into `partials` slots laid out like the forward's
// (the KL slot of workgroup 0 = the sum of pred[B .. B + VFM_MAX_FWD_BLOCKS), the forward workgroups' KL shares
// summed over ranks; NaN there -- ids out of range on some rank -- makes the loss NaN).  Same arithmetic per row as finish_row in vfm_fwd.hpp.
__global__ __launch_bounds__(BLOCK) void k_lik(const KArgs a, const FwdOut out) {
  __shared__ float sh_red[6 * 4];
  __shared__ double sh_kl[BLOCK / 64];
  const bool softplus = (a.flags & VFM_FLAG_LINK_SOFTPLUS) != 0;
  const float alpha = a.scalars[0], m0 = a.scalars[1], s0 = a.scalars[2];
  const float aabs = softplus ? link_f<LINK_SOFTPLUS>(alpha) : link_f<LINK_ABS>(alpha);
  float e0;
  if (a.eps_global) {
    e0 = a.eps_global[0];
  } else {
    float n[8], nb;
    normal8b(a.key, 0xFFFFFFFFu, 0u, n, nb);
    e0 = n[0];
  }
  const float w0 = fmaf(softplus ? link_f<LINK_SOFTPLUS>(s0) : link_f<LINK_ABS>(s0), e0, m0);
  const float half_log_a = 0.5f * LN2 * __builtin_amdgcn_logf(aabs);
  float tot[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int64_t r = blockIdx.x * (int64_t)BLOCK + threadIdx.x; r < a.B; r += (int64_t)gridDim.x * BLOCK) {
    const float pred = w0 + out.pred[r];
    const float y = a.y[r];
    float ll, dll, at;
    lik_terms(a.lik, y, pred, aabs, half_log_a, ll, dll, at);
    const float g = -a.ll_scale * dll;
    out.pred[r] = pred;
    out.grow[r] = g;
    tot[0] += ll; tot[2] += g; tot[3] += at;
  }
  block_sum<6>(tot, sh_red);
  if (threadIdx.x == 0) {
    double* slot = out.partials + VFM_N_PARTIALS * (1 + (size_t)blockIdx.x);
#pragma unroll
    for (int i = 0; i < 6; ++i) slot[i] = (double)tot[i];
    if (blockIdx.x == 0) {
      out.partials[7] = (double)gridDim.x;
      out.partials[VFM_P_REDUCED] = 0.0;
    }
  }
  if (blockIdx.x == 0) {     // the KL term: the forward workgroups' shares, summed over ranks by the all-reduce
    double kl = 0.0;         // (fixed order: reproducible)
    for (int b = threadIdx.x; b < VFM_MAX_FWD_BLOCKS; b += BLOCK) kl += (double)out.pred[a.B + b];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) kl += __shfl_xor(kl, m, 64);
    if ((threadIdx.x & 63) == 0) sh_kl[threadIdx.x >> 6] = kl;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int w = 0; w < BLOCK / 64; ++w) t += sh_kl[w];
      out.partials[VFM_N_PARTIALS * 1 + VFM_P_KL] = t;
    }
  }
}

// ws[0] <- sum of ws[1 .. (int)ws[0]]  (per-block partial sums written by a preceding kernel)
__global__ __launch_bounds__(BLOCK) void k_sum_slots(double* __restrict__ ws) {
  __shared__ double sh[BLOCK / 64];
  const int n = (int)ws[0];
  double acc = 0;
  for (int i = threadIdx.x; i < n; i += BLOCK) acc += ws[1 + i];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int w = 0; w < BLOCK / 64; ++w) t += sh[w];
    ws[0] = t;
  }
}

// replayable step: set the device-resident counters (vfm_dev_step_set)
__global__ void k_dev_step_set(vfm_dev_step_t* dev, uint64_t philox_step, int64_t adam_step) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    dev->philox_step = philox_step; dev->adam_step = adam_step;
    dev->philox_step_bwd = philox_step; dev->adam_step_bwd = adam_step;
  }
}

// packed first-order records (mu_w, s_w, 1/occ, 0) per entity (vfm_problem_t.wrec)
__global__ __launch_bounds__(256) void k_wrec_build(const float* __restrict__ bias, const float* __restrict__ inv_occ, int64_t T,
                                                    float* __restrict__ wrec) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < T; e += (int64_t)gridDim.x * blockDim.x) {
    const float2 th = *reinterpret_cast<const float2*>(bias + 2 * e);
    *reinterpret_cast<float4*>(wrec + 4 * e) = make_float4(th.x, th.y, inv_occ[e], 0.f);
  }
}
