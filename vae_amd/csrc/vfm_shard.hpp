// vfm_shard.hpp -- glue kernels of the entity-sharded multi-rank mode (tables partitioned by e mod N;
// k_sample: vfm_sample.hpp).  Included inside `namespace vfm { namespace {` of vfm_abi.hip.
#pragma once

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
