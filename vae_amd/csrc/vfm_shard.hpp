// vfm_shard.hpp -- kernels of the entity-sharded multi-rank mode (tables partitioned by e mod N).
// Part of vfm_kernels.hip (one translation unit; included inside its anonymous namespace).
#pragma once

// Owner side, before the forward: sample the requested entities once.  ids[i] is a global entity id
// owned by this rank; record i of `out` becomes (w, 0, 0, 0 | z[0..d-1]) with z = mu + |s| eps,
// w = mu_w + |s_w| eps_w -- what the requesting ranks' forward (EPS_ZPRE) consumes.  A lane group per
// record, same chunking as the other kernels.
template <int LPE, int CPL, int VEC, int EPS>
__global__ __launch_bounds__(BLOCK) void k_sample(const KArgs a, const int32_t* __restrict__ ids, int n,
                                                  float* __restrict__ out) {
  constexpr int GPB = BLOCK / LPE;
  const int lig = threadIdx.x % LPE;
  const int d = a.d;
  const int C = (d + VEC - 1) / VEC;
  const int64_t xs = 4 + (((int64_t)d + 3) & ~(int64_t)3);
  for (int i = blockIdx.x * GPB + threadIdx.x / LPE; i < n; i += gridDim.x * GPB) {
    const uint32_t e = (uint32_t)ids[i];
    const float* row = a.entity + (size_t)e * (2 * (size_t)d);
    float* rec = out + (size_t)i * xs;
    float epw = 0.f;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int j = lig + c * LPE;
      if (j < C) {
        const Chunk<VEC> mu = ld_chunk<VEC>(row + (size_t)j * VEC);
        const Chunk<VEC> s = ld_chunk<VEC>(row + d + (size_t)j * VEC);
        Chunk<VEC> ep, z;
        if constexpr (EPS == EPS_TABLE) {
          ep = ld_chunk<VEC>(a.eps_entity + (size_t)e * d + (size_t)j * VEC);
        } else {
          float nb;
          eps_of_chunk<VEC>(a.key, e, j, ep.v, nb);
          if (c == 0) epw = nb;
        }
#pragma unroll
        for (int t = 0; t < VEC; ++t) z.v[t] = fmaf(fabsf(s.v[t]), ep.v[t], mu.v[t]);
        st_chunk<VEC>(rec + 4 + (size_t)j * VEC, z);
      }
    }
    if (lig == 0) {
      if constexpr (EPS == EPS_TABLE) epw = a.eps_bias[e];
      const float2 th = *reinterpret_cast<const float2*>(a.bias + 2 * (size_t)e);
      *reinterpret_cast<float4*>(rec) = make_float4(fmaf(fabsf(th.y), epw, th.x), 0.f, 0.f, 0.f);
    }
  }
}

// Owner side, after the backward exchange: dst[idx[i]] += src[i] for n records of `xs` floats.  The
// ids inside one source rank's list are unique, so there are no conflicts within a call; the caller
// adds the source ranks one after the other (deterministic sums).
__global__ __launch_bounds__(BLOCK) void k_records_add(float* __restrict__ dst, const int32_t* __restrict__ idx,
                                                       const float* __restrict__ src, int64_t n, int xs4) {
  const int64_t total = n * xs4;
  for (int64_t t = blockIdx.x * (int64_t)BLOCK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLOCK) {
    const int64_t i = t / xs4;
    const int k = (int)(t % xs4);
    float4* d4 = reinterpret_cast<float4*>(dst) + (int64_t)idx[i] * xs4 + k;
    const float4 s4 = reinterpret_cast<const float4*>(src)[t];
    float4 v = *d4;
    v.x += s4.x; v.y += s4.y; v.z += s4.z; v.w += s4.w;
    *d4 = v;
  }
}

// The same with float atomics: all source ranks' records in ONE launch (ids may repeat across sources).
__global__ __launch_bounds__(BLOCK) void k_records_add_atomic(float* __restrict__ dst, const int32_t* __restrict__ idx,
                                                              const float* __restrict__ src, int64_t n, int xs) {
  const int64_t total = n * xs;
  for (int64_t t = blockIdx.x * (int64_t)BLOCK + threadIdx.x; t < total; t += (int64_t)gridDim.x * BLOCK) {
    const int64_t i = t / xs;
    const int k = (int)(t % xs);
    if (k == 2 || k == 3) continue;                    // padding words of a record
    const float v = src[t];
    if (v != 0.f) atomicAdd(dst + (int64_t)idx[i] * xs + k, v);
  }
}

// small[2..4] <- (nll, KL(q(w0)) share, KL share of the owned entities); one thread
__global__ void k_shard_pack(float* __restrict__ small, const float* __restrict__ loss_local,
                             const double* __restrict__ kl_ws) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    small[2] = loss_local[1];
    small[3] = loss_local[2];
    small[4] = (float)kl_ws[0];
  }
}

// loss3 <- (nll + kl, nll, kl) from the rank-summed small vector; one thread
__global__ void k_shard_loss(const float* __restrict__ small, float* __restrict__ loss3) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float nll = small[2], kl = small[3] + small[4];
    loss3[0] = nll + kl; loss3[1] = nll; loss3[2] = kl;
  }
}
