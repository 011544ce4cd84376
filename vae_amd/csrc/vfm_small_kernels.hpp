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
  __shared__ double sh[7][BLOCK / 64];
  double tot[6];
  reduce_slots_and_loss(partials, scalars, ll_scale, flags, loss, sh, tot);
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
