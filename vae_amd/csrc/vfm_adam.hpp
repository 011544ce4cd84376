// vfm_adam.hpp -- k_adam (dense Adam on a flat buffer), k_philox_dump.
// Included inside `namespace vfm { namespace {` of vfm_abi.hip.
#pragma once

// ---------------------------------------------------------------------------------------
// dense Adam (torch.optim.Adam defaults of vfm-torch.py:339,370; single-tensor op order:
// lerp / mul+addcmul / sqrt / div / add eps / addcdiv)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_adam(float* __restrict__ p, const float* __restrict__ g,
                                                float* __restrict__ m, float* __restrict__ v,
                                                int64_t n4, int64_t n, float b1, float b2, float eps,
                                                float step_size, float bc2_sqrt) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = blockIdx.x * (int64_t)BLOCK + threadIdx.x; i < n4; i += stride) {
    v4f gg = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(g) + i);
    v4f mm = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(m) + i);
    v4f vv = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(v) + i);
    float4 pp = reinterpret_cast<float4*>(p)[i];
    float* G = (float*)&gg; float* M = (float*)&mm; float* V = (float*)&vv; float* P = (float*)&pp;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      M[t] = M[t] + (G[t] - M[t]) * (1.0f - b1);
      V[t] = V[t] * b2 + ((1.0f - b2) * G[t]) * G[t];
      const float denom = __fsqrt_rn(V[t]) / bc2_sqrt + eps;
      P[t] = P[t] + (-step_size * M[t]) / denom;
    }
    __builtin_nontemporal_store(mm, reinterpret_cast<v4f*>(m) + i);
    __builtin_nontemporal_store(vv, reinterpret_cast<v4f*>(v) + i);
    reinterpret_cast<float4*>(p)[i] = pp;
  }
  // tail (n % 4 elements)
  const int64_t i = n4 * 4 + blockIdx.x * (int64_t)BLOCK + threadIdx.x;
  if (i < n) {
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
    const float vi = v[i] * b2 + ((1.0f - b2) * gi) * gi;
    m[i] = mi; v[i] = vi;
    p[i] = p[i] + (-step_size * mi) / (__fsqrt_rn(vi) / bc2_sqrt + eps);
  }
}

// m *= c1, v *= c2 (conversion between the plain and the scaled moment representation)
__global__ __launch_bounds__(BLOCK) void k_rescale2(float* __restrict__ m, float* __restrict__ v, int64_t n, float c1,
                                                    float c2) {
  for (int64_t i = blockIdx.x * (int64_t)BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
    m[i] *= c1;
    v[i] *= c2;
  }
}

// eps dump (tests)
__global__ void k_philox_dump(const KArgs a, float* eps_entity, float* eps_bias, float* eps_global) {
  const RngKey key = key_of_sample(a.key, a.sample);
  const int64_t n8 = ((int64_t)a.d + 7) / 8;
  const int64_t total = a.T * n8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i / n8;
    const int p = (int)(i % n8);
    float n[8], nb;
    normal8b(key, (uint32_t)e, (uint32_t)p, n, nb);
    for (int t = 0; t < 8; ++t)
      if (p * 8 + t < a.d) eps_entity[e * a.d + p * 8 + t] = n[t];
    if (p == 0) eps_bias[e] = nb;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float n[8], nb;
    normal8b(key, 0xFFFFFFFFu, 0u, n, nb);
    eps_global[0] = n[0];
  }
}
