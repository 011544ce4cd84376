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

// ---------------------------------------------------------------------------------------
// Lazy EXACT dense Adam (sparse-touch regimes: Criteo shape, 6 % of the rows in a batch).  Dense Adam moves
// every row every step; a row WITHOUT gradient moves by a function of its own (p, m, v) and the step's
// constants only, and in the scaled-moment form (VFM_FLAG_SCALED_MOMENTS) its stored moments do not change
// at all.  So such a row can be skipped and REPLAYED later: k_adam_catchup applies the zero-gradient updates
// of the steps last_step[e]+1 .. upto to the listed rows, with the same fp32 operations in the same order as
// k_bwd<ADAM> (adam_update, scaled form, g = 0) -- bitwise the dense trajectory.  The constants of the steps
// of the current moment period travel by value as a kernel argument (2 KB).  One wave per row.
// ---------------------------------------------------------------------------------------
struct CatchTab {
  float4 c[VFM_MOMENT_PERIOD + 1];      // c[k] = (a1 = step_size b1^k, q2 = sqrt(b2^k) / sqrt(bc2), -, -) of the period's k-th step
};

// one replayed zero-gradient step, given r = sqrt(stored second moment) -- the scaled branch of adam_update (vfm_bwd.hpp)
__device__ __forceinline__ float catchup_one(float p, float m, float r, const float4 c, float eps) {
  const float denom = fmaf(r, c.y, eps);
  return fmaf(-c.x * m, __builtin_amdgcn_rcpf(denom), p);
}

template <int VEC>        // 4: rows of 2d floats are 16-byte aligned (d even); 1: any d
__global__ __launch_bounds__(BLOCK) void k_adam_catchup(float* __restrict__ entity, float* __restrict__ bias,
                                                        const float* __restrict__ m_entity, const float* __restrict__ v_entity,
                                                        const float* __restrict__ m_bias, const float* __restrict__ v_bias,
                                                        int32_t* __restrict__ last_step, const int32_t* __restrict__ ids,
                                                        int64_t n, int d, int32_t pstart, int32_t upto, int32_t mark,
                                                        float eps, const CatchTab tab, float* __restrict__ wrec) {
  const int lane = threadIdx.x & 63;
  // The period's step constants live in REGISTERS, two entries per lane (k = 1 + lane, 65 + lane); a replayed step
  // fetches its pair with v_readlane (k is wave-uniform).  Indexing the by-value table with k inside the loop made
  // every replayed step wait for a vector load from the kernarg segment (first version: 222 us at the Criteo shape).
  const float4 c_lo = tab.c[1 + lane], c_hi = tab.c[(65 + lane) <= VFM_MOMENT_PERIOD ? 65 + lane : VFM_MOMENT_PERIOD];
  const int a_lo = __float_as_int(c_lo.x), q_lo = __float_as_int(c_lo.y);
  const int a_hi = __float_as_int(c_hi.x), q_hi = __float_as_int(c_hi.y);
  auto consts = [&](int k) -> float4 {              // k in 1 .. VFM_MOMENT_PERIOD, wave-uniform
    const int i = k - 1;
    float4 c;
    if (i < 64) {
      c.x = __int_as_float(__builtin_amdgcn_readlane(a_lo, i)); c.y = __int_as_float(__builtin_amdgcn_readlane(q_lo, i));
    } else {
      c.x = __int_as_float(__builtin_amdgcn_readlane(a_hi, i - 64)); c.y = __int_as_float(__builtin_amdgcn_readlane(q_hi, i - 64));
    }
    c.z = 0.f; c.w = 0.f;
    return c;
  };
  const int64_t nw = (int64_t)gridDim.x * (BLOCK / 64);
  const int np = (2 * d) / VEC;                    // pieces of an entity row
  for (int64_t i = (int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); i < n; i += nw) {
    const int64_t e = ids ? (int64_t)ids[i] : i;
    int k0 = __builtin_amdgcn_readfirstlane(last_step[e] - pstart);
    const int k1 = upto - pstart;
    if (k0 < 0) k0 = 0;                            // (never: a period boundary brings every row up to date)
    if (k1 > k0) {
      const size_t ro = (size_t)e * (2 * (size_t)d);
      // bias pair of the row: lane 0, loaded first and replayed inside the first pass' step loop
      float2 pb = make_float2(0.f, 0.f), mb = pb;
      float rbx = 1.f, rby = 1.f;
      if (lane == 0) {
        pb = *reinterpret_cast<float2*>(bias + 2 * (size_t)e);
        mb = *reinterpret_cast<const float2*>(m_bias + 2 * (size_t)e);
        const float2 vb = *reinterpret_cast<const float2*>(v_bias + 2 * (size_t)e);
        rbx = __builtin_amdgcn_sqrtf(vb.x); rby = __builtin_amdgcn_sqrtf(vb.y);
      }
      // two pieces per lane and pass: 2 VEC independent update chains under one fetch of the constants
      for (int j0 = 0; j0 < np; j0 += 128) {
        const int j = j0 + lane;
        const bool h0 = j < np, h1 = j + 64 < np;
        Chunk<VEC> p0, p1, m0, m1, r0, r1;
#pragma unroll
        for (int t = 0; t < VEC; ++t) { p0.v[t] = p1.v[t] = m0.v[t] = m1.v[t] = 0.f; r0.v[t] = r1.v[t] = 1.f; }
        if (h0) {
          p0 = ld_chunk<VEC>(entity + ro + (size_t)j * VEC);
          m0 = ld_chunk_nt<VEC>(m_entity + ro + (size_t)j * VEC);
          r0 = ld_chunk_nt<VEC>(v_entity + ro + (size_t)j * VEC);
        }
        if (h1) {
          p1 = ld_chunk<VEC>(entity + ro + (size_t)(j + 64) * VEC);
          m1 = ld_chunk_nt<VEC>(m_entity + ro + (size_t)(j + 64) * VEC);
          r1 = ld_chunk_nt<VEC>(v_entity + ro + (size_t)(j + 64) * VEC);
        }
#pragma unroll
        for (int t = 0; t < VEC; ++t) { r0.v[t] = __builtin_amdgcn_sqrtf(r0.v[t]); r1.v[t] = __builtin_amdgcn_sqrtf(r1.v[t]); }
        const bool with_bias = j0 == 0;
        for (int k = k0 + 1; k <= k1; ++k) {
          const float4 c = consts(k);
#pragma unroll
          for (int t = 0; t < VEC; ++t) {
            p0.v[t] = catchup_one(p0.v[t], m0.v[t], r0.v[t], c, eps);
            p1.v[t] = catchup_one(p1.v[t], m1.v[t], r1.v[t], c, eps);
          }
          if (with_bias) {                         // (uniform; lanes other than 0 carry zeros)
            pb.x = catchup_one(pb.x, mb.x, rbx, c, eps); pb.y = catchup_one(pb.y, mb.y, rby, c, eps);
          }
        }
        if (h0) st_chunk<VEC>(entity + ro + (size_t)j * VEC, p0);
        if (h1) st_chunk<VEC>(entity + ro + (size_t)(j + 64) * VEC, p1);
      }
      if (lane == 0) {
        *reinterpret_cast<float2*>(bias + 2 * (size_t)e) = pb;
        if (wrec) *reinterpret_cast<float2*>(wrec + 4 * (size_t)e) = pb;       // the packed first-order record follows
      }
    }
    if (lane == 0) last_step[e] = mark;
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
