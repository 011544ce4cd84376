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

// W[f] = sum_r inv_occ[x[r,f]]: one workgroup per column f, fp64, fixed summation order (thread-strided
// partial sums, xor-shuffle tree inside each wave, waves added in order) -- bitwise reproducible, so the
// KL scale n_g / W_g and everything downstream of it is too.  Once per batch, outside the step.
constexpr int NORMS_BLOCK = 1024;
__global__ __launch_bounds__(NORMS_BLOCK) void k_norms(const void* __restrict__ x, int id64,
                                                       const float* __restrict__ inv_occ, int64_t B, int F,
                                                       int64_t T, double* __restrict__ W) {
  __shared__ double sh[NORMS_BLOCK / 64];
  const int f = blockIdx.x;
  double acc = 0.0;
  auto id_of = [&](int64_t r) -> int64_t {
    const int64_t o = r * F + f;
    return id64 ? ((const int64_t*)x)[o] : (int64_t)((const int32_t*)x)[o];
  };
  constexpr int UN = 8;                  // independent id -> 1/occ chains in flight per thread
  int64_t r = threadIdx.x;
  for (; r + (UN - 1) * (int64_t)NORMS_BLOCK < B; r += UN * (int64_t)NORMS_BLOCK) {
    int64_t id[UN];
    float v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) id[u] = id_of(r + u * (int64_t)NORMS_BLOCK);
#pragma unroll
    for (int u = 0; u < UN; ++u) v[u] = (id[u] >= 0 && id[u] < T) ? inv_occ[id[u]] : 0.f;
#pragma unroll
    for (int u = 0; u < UN; ++u) acc += (double)v[u];
  }
  for (; r < B; r += NORMS_BLOCK) {
    const int64_t id = id_of(r);
    if (id >= 0 && id < T) acc += (double)inv_occ[id];
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < NORMS_BLOCK / 64; ++w) t += sh[w];
    W[f] = t;
  }
}

__global__ __launch_bounds__(BLOCK) void k_finalize(double* __restrict__ partials,
                                                    const float* __restrict__ scalars, double ll_scale,
                                                    int flags, float* __restrict__ loss) {
  __shared__ double sh[6][BLOCK / 64];
  double tot[6];
  reduce_slots_and_loss(partials, scalars, ll_scale, flags, loss, sh, tot);
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
