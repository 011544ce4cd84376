// vfm_kernels.hip -- hand-written gfx950 (MI355X, CDNA4) kernels + C ABI for the
// Variational-FM ELBO step.  Wave = 64 lanes; no MFMA: the path is gather + elementwise +
// reduction and is bounded by HBM / Infinity-Cache bandwidth.
//
// Replaces the per-batch body of the reference: vfm-torch.py:189-324 (CF.forward), the loss
// line :359 and autograd through them (:368-369).  Math: SURVEY.md Appendix A.
//
// Work decomposition
//   forward : a *lane group* of LPE lanes owns one batch row; each lane owns CPL chunks of
//             VEC consecutive embedding coordinates.  A 256-thread workgroup stages a tile of
//             rows in LDS first (one thread per (row, field) occurrence: id, bias sample,
//             KL weight), then the lane groups gather the 8d-byte table rows with 16-byte
//             loads, form z = mu + |s| eps in registers, and reduce the FM term across the
//             group with wave shuffles.
//   backward: entity-centric.  A lane group owns one TABLE row e and sums grow[r]*sumz[r,:]
//             over the batch rows containing e (inverted index), then writes the dense
//             gradient row once.  No atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

#include "vfm_hip.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
int fail_hip(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
  return (int)e;
}

constexpr int BLOCK = 256;
constexpr int OCC_CAP = 1024;  // (row, field) occurrences staged per LDS tile
constexpr float LOG_SQRT_2PI = 0.918938533204672742f;
constexpr float LN2 = 0.693147180559945309f;

// ---------------------------------------------------------------------------------------
// Counter-based RNG: Philox4x32-10 (Salmon et al. 2011) + Box-Muller on the hardware
// transcendental units.  One call yields the 4 standard normals of coordinates 4j..4j+3 of
// entity e at step `step`: every row (and every rank) that touches e sees the same draw.
// ---------------------------------------------------------------------------------------
struct RngKey {
  uint32_t seed_lo, seed_hi, step_lo, step_hi;
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
  const float u1 = ((float)(a >> 8) + 0.5f) * 5.9604644775390625e-8f;  // (0,1)
  const float u2 = (float)(b >> 8) * 5.9604644775390625e-8f;           // [0,1) revolutions
  const float r = __builtin_amdgcn_sqrtf(-2.0f * LN2 * __builtin_amdgcn_logf(u1));
  n0 = r * __builtin_amdgcn_cosf(u2);  // v_cos_f32 / v_sin_f32 take revolutions
  n1 = r * __builtin_amdgcn_sinf(u2);
}

enum { TAG_ENTITY = 0, TAG_BIAS = 1, TAG_GLOBAL = 2 };

__device__ __forceinline__ void normal4(const RngKey& k, uint32_t e, uint32_t j, uint32_t tag,
                                        float n[4]) {
  uint32_t o[4];
  philox4x32_10(j, e, k.step_lo, k.step_hi ^ (tag << 30), k.seed_lo, k.seed_hi, o);
  box_muller(o[0], o[1], n[0], n[1]);
  box_muller(o[2], o[3], n[2], n[3]);
}

// ---------------------------------------------------------------------------------------
// Kernel arguments (by value)
// ---------------------------------------------------------------------------------------
struct KArgs {
  int64_t B, T;
  int32_t F, d, lik, id64, G, TR, flags;
  float ll_scale;  // nb_train / B_global
  RngKey key;
  const void* x;
  const float* y;
  const float* entity;
  const float* bias;
  const float* inv_occ;
  const float* scalars;
  const double* W;
  const float* eps_entity;
  const float* eps_bias;
  const float* eps_global;
  int64_t group_hi[VFM_MAX_FIELDS];
  double group_n[VFM_MAX_FIELDS];
};

struct FwdOut {
  float* pred;
  double* partials;
  float* sumz;
  float* grow;
};

struct BwdArgs {
  const int32_t* occ_ptr;
  const int32_t* occ_rows;
  const float* sumz;
  const float* grow;
  const double* partials;
  const float* grad_out;
  float* g_entity;
  float* g_bias;
  float* g_scalars;
};

template <int VEC>
struct Chunk {
  float v[VEC];
};

template <int VEC>
__device__ __forceinline__ Chunk<VEC> ld_chunk(const float* p) {
  Chunk<VEC> c;
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    c.v[0] = t.x; c.v[1] = t.y; c.v[2] = t.z; c.v[3] = t.w;
  } else {
    c.v[0] = *p;
  }
  return c;
}

template <int VEC>
__device__ __forceinline__ void st_chunk(float* p, const Chunk<VEC>& c) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(c.v[0], c.v[1], c.v[2], c.v[3]);
  } else {
    *p = c.v[0];
  }
}

// eps for chunk j (coordinates j*VEC ..) of entity e: table or Philox
template <int VEC>
__device__ __forceinline__ Chunk<VEC> eps_chunk(const KArgs& a, uint32_t e, int j) {
  if (a.flags & VFM_FLAG_EPS_ZERO) {
    Chunk<VEC> z;
#pragma unroll
    for (int t = 0; t < VEC; ++t) z.v[t] = 0.f;
    return z;
  }
  if (a.eps_entity) return ld_chunk<VEC>(a.eps_entity + (size_t)e * a.d + (size_t)j * VEC);
  Chunk<VEC> c;
  float n[4];
  if constexpr (VEC == 4) {
    normal4(a.key, e, (uint32_t)j, TAG_ENTITY, n);
    c.v[0] = n[0]; c.v[1] = n[1]; c.v[2] = n[2]; c.v[3] = n[3];
  } else {
    normal4(a.key, e, (uint32_t)j >> 2, TAG_ENTITY, n);
    c.v[0] = n[j & 3];
  }
  return c;
}

__device__ __forceinline__ float eps_bias_of(const KArgs& a, uint32_t e) {
  if (a.flags & VFM_FLAG_EPS_ZERO) return 0.f;
  if (a.eps_bias) return a.eps_bias[e];
  float n[4];
  normal4(a.key, e, 0u, TAG_BIAS, n);
  return n[0];
}

__device__ __forceinline__ float eps_global_of(const KArgs& a) {
  if (a.flags & VFM_FLAG_EPS_ZERO) return 0.f;
  if (a.eps_global) return a.eps_global[0];
  float n[4];
  normal4(a.key, 0xFFFFFFFFu, 0u, TAG_GLOBAL, n);
  return n[0];
}

__device__ __forceinline__ float kl_std_normal(float mu, float sg) {
  // KL(N(mu, sg) || N(0,1)) = 1/2 (sg^2 + mu^2 - 1) - log sg   (torch kl.py _kl_normal_normal)
  return 0.5f * (sg * sg + mu * mu - 1.0f) - LN2 * __builtin_amdgcn_logf(sg);
}

__device__ __forceinline__ float signf(float s) { return (s > 0.f) ? 1.f : ((s < 0.f) ? -1.f : 0.f); }

template <int W>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int m = W / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

__device__ __forceinline__ int group_index(const int64_t* hi, int G, int64_t id) {
  int g = 0;
  while (g < G - 1 && id >= hi[g]) ++g;
  return g;
}

// sum NV per-thread values over the block, thread 0 gets the totals
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* sh /* [NV * 4] */) {
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = group_sum<64>(v[i]);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) sh[i * 4 + wave] = v[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = sh[i * 4] + sh[i * 4 + 1] + sh[i * 4 + 2] + sh[i * 4 + 3];
  }
}

// ---------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------
__global__ void k_inv_occ(const int64_t* __restrict__ occ, float* __restrict__ inv, int64_t T) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < T;
       i += (int64_t)gridDim.x * blockDim.x)
    inv[i] = 1.0f / (float)occ[i];
}

__global__ __launch_bounds__(BLOCK) void k_norms(const void* __restrict__ x, int id64,
                                                 const float* __restrict__ inv_occ, int64_t n_occ,
                                                 int F, int64_t T, double* __restrict__ W) {
  __shared__ float sh[VFM_MAX_FIELDS];
  if (threadIdx.x < VFM_MAX_FIELDS) sh[threadIdx.x] = 0.f;
  __syncthreads();
  // each thread walks occurrences o = t, t + stride...; stride is a multiple of F so the
  // field of a thread is fixed
  const int64_t stride0 = (int64_t)gridDim.x * BLOCK;
  const int64_t stride = (stride0 + F - 1) / F * F;
  const int64_t t = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
  float acc = 0.f;
  for (int64_t o = t; o < n_occ; o += stride) {
    int64_t id = id64 ? ((const int64_t*)x)[o] : (int64_t)((const int32_t*)x)[o];
    if (id >= 0 && id < T) acc += inv_occ[id];
  }
  if (acc != 0.f) atomicAdd(&sh[t % F], acc);
  __syncthreads();
  if (threadIdx.x < F) atomicAdd(&W[threadIdx.x], (double)sh[threadIdx.x]);
}

__global__ void k_finalize(const double* __restrict__ partials, const float* __restrict__ scalars,
                           double ll_scale, int flags, float* __restrict__ loss) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double m0 = scalars[1], s0 = scalars[2];
  const double kl0 = (flags & VFM_FLAG_NO_PRIOR_TERMS)
                         ? 0.0
                         : 0.5 * (s0 * s0 + m0 * m0 - 1.0) - log(fabs(s0));
  const double nll = -ll_scale * partials[VFM_P_LL];
  const double kl = kl0 + partials[VFM_P_KL];
  const bool bad = partials[VFM_P_BADID] != 0.0;
  const float nanv = __builtin_nanf("");
  loss[0] = bad ? nanv : (float)(nll + kl);
  loss[1] = bad ? nanv : (float)nll;
  loss[2] = bad ? nanv : (float)kl;
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
template <int LPE, int CPL, int VEC>
__global__ __launch_bounds__(BLOCK) void k_fwd(const KArgs a, const FwdOut out) {
  constexpr int GPB = BLOCK / LPE;
  __shared__ uint32_t sh_id[OCC_CAP];
  __shared__ float sh_c[OCC_CAP];
  __shared__ float sh_w[OCC_CAP];
  __shared__ float sh_cs[VFM_MAX_FIELDS];
  __shared__ int64_t sh_hi[VFM_MAX_FIELDS];
  __shared__ float sh_red[5 * 4];

  const int tid = threadIdx.x;
  const int lig = tid % LPE;
  const int gi = tid / LPE;
  const int F = a.F, d = a.d;
  const int C = (d + VEC - 1) / VEC;
  const bool have_y = a.y != nullptr;
  const bool train = out.sumz != nullptr;

  if (tid < a.G) {
    sh_cs[tid] = have_y ? (float)(a.group_n[tid] / a.W[tid]) : 0.f;
    sh_hi[tid] = a.group_hi[tid];
  }
  const float alpha = a.scalars[0], m0 = a.scalars[1], s0 = a.scalars[2];
  const float aabs = fabsf(alpha);
  const float w0 = m0 + fabsf(s0) * eps_global_of(a);
  const float half_log_a = 0.5f * LN2 * __builtin_amdgcn_logf(aabs);

  float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // ll, kl, g, alpha-term, bad ids

  const int64_t ntiles = (a.B + a.TR - 1) / a.TR;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
    const int64_t row0 = tile * a.TR;
    const int nrows = (int)((a.B - row0 < a.TR) ? (a.B - row0) : a.TR);
    const int nocc = nrows * F;
    // ---- stage the tile: one thread per (row, field) occurrence ----
    for (int o = tid; o < nocc; o += BLOCK) {
      const int64_t go = row0 * F + o;
      int64_t id = a.id64 ? ((const int64_t*)a.x)[go] : (int64_t)((const int32_t*)a.x)[go];
      if (id < 0 || id >= a.T) { acc[4] += 1.f; id = 0; }
      const uint32_t e = (uint32_t)id;
      const float2 th = *reinterpret_cast<const float2*>(a.bias + 2 * (size_t)e);
      const float sg = fabsf(th.y);
      const float w = th.x + sg * eps_bias_of(a, e);
      float c = 0.f;
      if (have_y) {
        c = sh_cs[group_index(sh_hi, a.G, id)] * a.inv_occ[e];
        acc[1] += c * kl_std_normal(th.x, sg);
      }
      sh_id[o] = e;
      sh_c[o] = c;
      sh_w[o] = w;
    }
    __syncthreads();
    // ---- one lane group per row ----
    for (int rr = gi; rr < nrows; rr += GPB) {
      const int64_t r = row0 + rr;
      Chunk<VEC> sz[CPL];
#pragma unroll
      for (int i = 0; i < CPL; ++i)
#pragma unroll
        for (int t = 0; t < VEC; ++t) sz[i].v[t] = 0.f;
      float zz = 0.f, bsum = 0.f, klacc = 0.f;
      for (int f = 0; f < F; ++f) {
        const uint32_t e = sh_id[rr * F + f];
        const float c = sh_c[rr * F + f];
        bsum += sh_w[rr * F + f];
        const float* row = a.entity + (size_t)e * (2 * (size_t)d);
        float klv = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          const int j = lig + i * LPE;
          if (j < C) {
            const Chunk<VEC> mu = ld_chunk<VEC>(row + (size_t)j * VEC);
            const Chunk<VEC> s = ld_chunk<VEC>(row + d + (size_t)j * VEC);
            const Chunk<VEC> ep = eps_chunk<VEC>(a, e, j);
#pragma unroll
            for (int t = 0; t < VEC; ++t) {
              const float sg = fabsf(s.v[t]);
              const float z = fmaf(sg, ep.v[t], mu.v[t]);
              sz[i].v[t] += z;
              zz = fmaf(z, z, zz);
              if (have_y) klv += kl_std_normal(mu.v[t], sg);
            }
          }
        }
        klacc = fmaf(c, klv, klacc);
      }
      float q = -zz;
#pragma unroll
      for (int i = 0; i < CPL; ++i)
#pragma unroll
        for (int t = 0; t < VEC; ++t) q = fmaf(sz[i].v[t], sz[i].v[t], q);
      q = 0.5f * group_sum<LPE>(q);
      acc[1] += klacc;
      const float pred = w0 + bsum + q;
      float g = 0.f;
      if (lig == 0) {
        out.pred[r] = pred;
        if (have_y) {
          const float y = a.y[r];
          float ll, dll;
          if (a.lik == VFM_LIK_NORMAL) {
            const float diff = y - pred;
            ll = -0.5f * aabs * diff * diff + half_log_a - LOG_SQRT_2PI;
            dll = aabs * diff;
            acc[3] += 0.5f * diff * diff - 0.5f / aabs;
          } else {
            const float ax = fabsf(pred);
            const float e1 = __expf(-ax);
            ll = y * pred - (fmaxf(pred, 0.f) + log1pf(e1));
            const float sig = (pred >= 0.f) ? 1.f / (1.f + e1) : e1 / (1.f + e1);
            dll = y - sig;
          }
          g = -a.ll_scale * dll;
          acc[0] += ll;
          acc[2] += g;
          if (train) out.grow[r] = g;
        }
      }
      if (train) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          const int j = lig + i * LPE;
          if (j < C) st_chunk<VEC>(out.sumz + (size_t)r * d + (size_t)j * VEC, sz[i]);
        }
      }
    }
  }
  block_sum<5>(acc, sh_red);
  if (tid == 0) {
    if (acc[0] != 0.f) atomicAdd(&out.partials[VFM_P_LL], (double)acc[0]);
    if (acc[1] != 0.f) atomicAdd(&out.partials[VFM_P_KL], (double)acc[1]);
    if (acc[2] != 0.f) atomicAdd(&out.partials[VFM_P_G], (double)acc[2]);
    if (acc[3] != 0.f) atomicAdd(&out.partials[VFM_P_ALPHA], (double)acc[3]);
    if (acc[4] != 0.f) atomicAdd(&out.partials[VFM_P_BADID], (double)acc[4]);
  }
}

// ---------------------------------------------------------------------------------------
// backward (entity-centric, dense gradient rows, no atomics)
// ---------------------------------------------------------------------------------------
template <int LPE, int CPL, int VEC>
__global__ __launch_bounds__(BLOCK) void k_bwd(const KArgs a, const BwdArgs b) {
  constexpr int GPB = BLOCK / LPE;
  __shared__ float sh_cs[VFM_MAX_FIELDS];
  __shared__ int64_t sh_hi[VFM_MAX_FIELDS];
  const int tid = threadIdx.x;
  const int lig = tid % LPE;
  const int gi = tid / LPE;
  const int d = a.d;
  const int C = (d + VEC - 1) / VEC;
  if (tid < a.G) {
    sh_cs[tid] = (float)(a.group_n[tid] / a.W[tid]);
    sh_hi[tid] = a.group_hi[tid];
  }
  __syncthreads();
  const float gout = b.grad_out[0];

  if (blockIdx.x == 0 && tid == 0) {
    const float alpha = a.scalars[0], m0 = a.scalars[1], s0 = a.scalars[2];
    const float sum_g = (float)b.partials[VFM_P_G];
    const float e0 = eps_global_of(a);
    const float as0 = fabsf(s0);
    b.g_scalars[0] = (a.lik == VFM_LIK_NORMAL)
                         ? gout * signf(alpha) * a.ll_scale * (float)b.partials[VFM_P_ALPHA]
                         : 0.f;
    const float prior = (a.flags & VFM_FLAG_NO_PRIOR_TERMS) ? 0.f : 1.f;
    b.g_scalars[1] = gout * (sum_g + prior * m0);
    b.g_scalars[2] = gout * signf(s0) * (e0 * sum_g + prior * (as0 - 1.0f / as0));
  }

  for (int64_t e = (int64_t)blockIdx.x * GPB + gi; e < a.T; e += (int64_t)gridDim.x * GPB) {
    const int beg = b.occ_ptr[e], end = b.occ_ptr[e + 1];
    float* grow_e = b.g_entity + (size_t)e * (2 * (size_t)d);
    if (beg == end) {  // entity not in the batch: dense zero row (vfm-torch.py:152-153 dense grads)
      Chunk<VEC> zc;
#pragma unroll
      for (int t = 0; t < VEC; ++t) zc.v[t] = 0.f;
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int j = lig + i * LPE;
        if (j < C) {
          st_chunk<VEC>(grow_e + (size_t)j * VEC, zc);
          st_chunk<VEC>(grow_e + d + (size_t)j * VEC, zc);
        }
      }
      if (lig == 0) *reinterpret_cast<float2*>(b.g_bias + 2 * (size_t)e) = make_float2(0.f, 0.f);
      continue;
    }
    Chunk<VEC> A[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
      for (int t = 0; t < VEC; ++t) A[i].v[t] = 0.f;
    float gs = 0.f;
    for (int o = beg; o < end; ++o) {
      const int r = b.occ_rows[o];
      const float g = b.grow[r];
      gs += g;
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int j = lig + i * LPE;
        if (j < C) {
          const Chunk<VEC> sv = ld_chunk<VEC>(b.sumz + (size_t)r * d + (size_t)j * VEC);
#pragma unroll
          for (int t = 0; t < VEC; ++t) A[i].v[t] = fmaf(g, sv.v[t], A[i].v[t]);
        }
      }
    }
    const float c = sh_cs[group_index(sh_hi, a.G, e)] * a.inv_occ[e] * (float)(end - beg);
    const float* row = a.entity + (size_t)e * (2 * (size_t)d);
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int j = lig + i * LPE;
      if (j < C) {
        const Chunk<VEC> mu = ld_chunk<VEC>(row + (size_t)j * VEC);
        const Chunk<VEC> s = ld_chunk<VEC>(row + d + (size_t)j * VEC);
        const Chunk<VEC> ep = eps_chunk<VEC>(a, (uint32_t)e, j);
        Chunk<VEC> gm, gv;
#pragma unroll
        for (int t = 0; t < VEC; ++t) {
          const float sg = fabsf(s.v[t]);
          const float z = fmaf(sg, ep.v[t], mu.v[t]);
          const float gz = A[i].v[t] - z * gs;  // sum_r g_r (sumz_rk - z_ek)
          gm.v[t] = gout * (gz + c * mu.v[t]);
          gv.v[t] = gout * signf(s.v[t]) * (gz * ep.v[t] + c * (sg - 1.0f / sg));
        }
        st_chunk<VEC>(grow_e + (size_t)j * VEC, gm);
        st_chunk<VEC>(grow_e + d + (size_t)j * VEC, gv);
      }
    }
    if (lig == 0) {
      const float2 th = *reinterpret_cast<const float2*>(a.bias + 2 * (size_t)e);
      const float sg = fabsf(th.y);
      const float ew = eps_bias_of(a, (uint32_t)e);
      *reinterpret_cast<float2*>(b.g_bias + 2 * (size_t)e) =
          make_float2(gout * (gs + c * th.x),
                      gout * signf(th.y) * (gs * ew + c * (sg - 1.0f / sg)));
    }
  }
}

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
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float4 pp = reinterpret_cast<float4*>(p)[i];
    float* G = (float*)&gg; float* M = (float*)&mm; float* V = (float*)&vv; float* P = (float*)&pp;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      M[t] = M[t] + (G[t] - M[t]) * (1.0f - b1);
      V[t] = V[t] * b2 + ((1.0f - b2) * G[t]) * G[t];
      const float denom = __fsqrt_rn(V[t]) / bc2_sqrt + eps;
      P[t] = P[t] + (-step_size * M[t]) / denom;
    }
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
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

// eps dump (tests)
__global__ void k_philox_dump(const KArgs a, float* eps_entity, float* eps_bias, float* eps_global) {
  const int64_t n4 = ((int64_t)a.d + 3) / 4;
  const int64_t total = a.T * n4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i / n4;
    const int j = (int)(i % n4);
    float n[4];
    normal4(a.key, (uint32_t)e, (uint32_t)j, TAG_ENTITY, n);
    for (int t = 0; t < 4; ++t)
      if (j * 4 + t < a.d) eps_entity[e * a.d + j * 4 + t] = n[t];
    if (j == 0) {
      normal4(a.key, (uint32_t)e, 0u, TAG_BIAS, n);
      eps_bias[e] = n[0];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float n[4];
    normal4(a.key, 0xFFFFFFFFu, 0u, TAG_GLOBAL, n);
    eps_global[0] = n[0];
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
struct Shape {
  int lpe, cpl, vec;
};

bool pick_shape(int d, Shape* s) {
  auto pow2ceil = [](int v) { int p = 1; while (p < v) p <<= 1; return p; };
  if (d % 4 == 0) {
    const int C = d / 4;
    s->vec = 4;
    s->lpe = pow2ceil(C) < 64 ? pow2ceil(C) : 64;
    const int cpl = (C + s->lpe - 1) / s->lpe;
    s->cpl = cpl <= 1 ? 1 : (cpl <= 2 ? 2 : 4);
    return cpl <= 4;
  }
  s->vec = 1;
  if (d <= 8) { s->lpe = 8; s->cpl = 1; return true; }
  if (d <= 64) { s->lpe = 64; s->cpl = 1; return true; }
  s->lpe = 64; s->cpl = 4;
  return d <= 256;
}

int check_problem(const vfm_problem_t* p) {
  if (!p) return fail(VFM_E_INVALID, "problem is NULL");
  if (p->B < 0 || p->T <= 0 || p->T > 0xFFFFFFFELL) return fail(VFM_E_INVALID, "bad B or T");
  if (p->F < 1 || p->F > VFM_MAX_FIELDS) return fail(VFM_E_INVALID, "F out of range [1,64]");
  if (p->d < 1) return fail(VFM_E_INVALID, "d < 1");
  if (p->id_bits != 32 && p->id_bits != 64) return fail(VFM_E_INVALID, "id_bits must be 32 or 64");
  if (p->likelihood != VFM_LIK_NORMAL && p->likelihood != VFM_LIK_BERNOULLI)
    return fail(VFM_E_INVALID, "unknown likelihood");
  if (p->n_samples != 1) return fail(VFM_E_UNSUPPORTED, "only n_samples == 1 is supported");
  if (p->B_global < p->B) return fail(VFM_E_INVALID, "B_global < B");
  if (p->B * (int64_t)p->F > 0x7FFFFFFFLL) return fail(VFM_E_INVALID, "B*F exceeds int32 index range");
  Shape s;
  if (!pick_shape(p->d, &s)) return fail(VFM_E_UNSUPPORTED, "embedding size d not supported (d%4==0: d<=1024, else d<=256)");
  return 0;
}

KArgs make_args(const vfm_problem_t* p, const void* x, const float* y, const float* entity,
                const float* bias, const float* inv_occ, const float* scalars, const double* W,
                const float* ee, const float* eb, const float* eg) {
  KArgs a;
  memset(&a, 0, sizeof(a));
  a.B = p->B; a.T = p->T; a.F = p->F; a.d = p->d; a.lik = p->likelihood;
  a.id64 = p->id_bits == 64; a.G = p->F; a.flags = p->flags;
  a.ll_scale = (float)((double)p->nb_train / (double)(p->B_global > 0 ? p->B_global : 1));
  a.key.seed_lo = (uint32_t)p->seed; a.key.seed_hi = (uint32_t)(p->seed >> 32);
  a.key.step_lo = (uint32_t)p->step; a.key.step_hi = (uint32_t)(p->step >> 32) & 0x3FFFFFFFu;
  a.x = x; a.y = y; a.entity = entity; a.bias = bias; a.inv_occ = inv_occ; a.scalars = scalars;
  a.W = W; a.eps_entity = ee; a.eps_bias = eb; a.eps_global = eg;
  for (int g = 0; g < p->F; ++g) { a.group_hi[g] = p->group_hi[g]; a.group_n[g] = p->group_n[g]; }
  return a;
}

template <template <int, int, int> class L, typename... Args>
int dispatch(const Shape& s, Args&&... args) {
#define CASE(L_, C_, V_) \
  if (s.lpe == L_ && s.cpl == C_ && s.vec == V_) return L<L_, C_, V_>::run(args...);
  CASE(1, 1, 4) CASE(2, 1, 4) CASE(4, 1, 4) CASE(8, 1, 4) CASE(16, 1, 4) CASE(32, 1, 4)
  CASE(64, 1, 4) CASE(64, 2, 4) CASE(64, 4, 4)
  CASE(8, 1, 1) CASE(64, 1, 1) CASE(64, 4, 1)
#undef CASE
  return fail(VFM_E_UNSUPPORTED, "no kernel instance for this embedding size");
}

template <int LPE, int CPL, int VEC>
struct LaunchFwd {
  static int run(KArgs& a, const FwdOut& o, hipStream_t st) {
    constexpr int GPB = BLOCK / LPE;
    // tile rows: fill the LDS tile, but keep >= ~2048 tiles so every CU of the 8 XCDs gets work
    int64_t tr = OCC_CAP / a.F;
    const int64_t want = (a.B + 2047) / 2048;
    if (tr > want) tr = want;
    if (tr < GPB) tr = GPB;
    if (tr * a.F > OCC_CAP) tr = OCC_CAP / a.F;
    if (tr < 1) tr = 1;
    a.TR = (int)tr;
    const int64_t ntiles = (a.B + tr - 1) / tr;
    const int grid = (int)(ntiles < 2048 ? (ntiles > 0 ? ntiles : 1) : 2048);
    hipLaunchKernelGGL((k_fwd<LPE, CPL, VEC>), dim3(grid), dim3(BLOCK), 0, st, a, o);
    return 0;
  }
};

template <int LPE, int CPL, int VEC>
struct LaunchBwd {
  static int run(KArgs& a, const BwdArgs& b, hipStream_t st) {
    constexpr int GPB = BLOCK / LPE;
    const int64_t nb = (a.T + GPB - 1) / GPB;
    const int grid = (int)(nb < 4096 ? nb : 4096);
    hipLaunchKernelGGL((k_bwd<LPE, CPL, VEC>), dim3(grid), dim3(BLOCK), 0, st, a, b);
    return 0;
  }
};

int after_launch(const char* where) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, where);
  return 0;
}

}  // namespace

extern "C" {

int vfm_abi_version(void) { return VFM_ABI_VERSION; }
const char* vfm_last_error(void) { return g_err; }

int vfm_inv_occ_f32(const int64_t* nb_occ, float* inv_occ, int64_t T, void* stream) {
  if (!nb_occ || !inv_occ || T <= 0) return fail(VFM_E_INVALID, "vfm_inv_occ_f32: bad argument");
  const int grid = (int)((T + 255) / 256 < 2048 ? (T + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_inv_occ, dim3(grid), dim3(256), 0, (hipStream_t)stream, nb_occ, inv_occ, T);
  return after_launch("vfm_inv_occ_f32");
}

int vfm_batch_norms(const vfm_problem_t* p, const void* x, const float* inv_occ, double* W,
                    void* stream) {
  if (int rc = check_problem(p)) return rc;
  if (!x || !inv_occ || !W) return fail(VFM_E_INVALID, "vfm_batch_norms: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(W, 0, sizeof(double) * p->F, st);
  if (e != hipSuccess) return fail_hip(e, "vfm_batch_norms memset");
  const int64_t n_occ = p->B * p->F;
  if (n_occ == 0) return 0;
  const int64_t nb = (n_occ + BLOCK - 1) / BLOCK;
  const int grid = (int)(nb < 1024 ? nb : 1024);
  hipLaunchKernelGGL(k_norms, dim3(grid), dim3(BLOCK), 0, st, x, (int)(p->id_bits == 64), inv_occ,
                     n_occ, (int)p->F, p->T, W);
  return after_launch("vfm_batch_norms");
}

int vfm_elbo_fwd_f32(const vfm_problem_t* p, const void* x, const float* y,
                     const float* entity_params, const float* bias_params,
                     const float* inv_occ, const float* scalars, const double* W,
                     const float* eps_entity, const float* eps_bias, const float* eps_global,
                     float* pred, double* partials, float* sumz, float* grow, void* stream) {
  if (int rc = check_problem(p)) return rc;
  if (!x || !entity_params || !bias_params || !scalars || !pred || !partials)
    return fail(VFM_E_INVALID, "vfm_elbo_fwd_f32: NULL pointer");
  if (y && (!inv_occ || !W)) return fail(VFM_E_INVALID, "vfm_elbo_fwd_f32: y given but inv_occ / W NULL");
  if ((sumz == nullptr) != (grow == nullptr) || (sumz && !y))
    return fail(VFM_E_INVALID, "vfm_elbo_fwd_f32: sumz and grow go together and need y");
  const int neps = (eps_entity != nullptr) + (eps_bias != nullptr) + (eps_global != nullptr);
  if (neps != 0 && neps != 3) return fail(VFM_E_INVALID, "vfm_elbo_fwd_f32: give all three eps tables or none");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(partials, 0, sizeof(double) * VFM_N_PARTIALS, st);
  if (e != hipSuccess) return fail_hip(e, "vfm_elbo_fwd_f32 memset");
  if (p->B == 0) return 0;
  KArgs a = make_args(p, x, y, entity_params, bias_params, inv_occ, scalars, W, eps_entity, eps_bias,
                      eps_global);
  FwdOut o{pred, partials, sumz, grow};
  Shape s;
  pick_shape(p->d, &s);
  if (int rc = dispatch<LaunchFwd>(s, a, o, st)) return rc;
  return after_launch("vfm_elbo_fwd_f32");
}

int vfm_elbo_finalize_f32(const vfm_problem_t* p, const double* partials, const float* scalars,
                          float* loss, void* stream) {
  if (int rc = check_problem(p)) return rc;
  if (!partials || !scalars || !loss) return fail(VFM_E_INVALID, "vfm_elbo_finalize_f32: NULL pointer");
  const double ll_scale = (double)p->nb_train / (double)(p->B_global > 0 ? p->B_global : 1);
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, (hipStream_t)stream, partials, scalars, ll_scale,
                     (int)p->flags, loss);
  return after_launch("vfm_elbo_finalize_f32");
}

int vfm_elbo_bwd_f32(const vfm_problem_t* p, const int32_t* occ_ptr, const int32_t* occ_rows,
                     const float* entity_params, const float* bias_params,
                     const float* inv_occ, const float* scalars, const double* W,
                     const float* eps_entity, const float* eps_bias, const float* eps_global,
                     const float* sumz, const float* grow, const double* partials,
                     const float* grad_out, float* g_entity, float* g_bias, float* g_scalars,
                     void* stream) {
  if (int rc = check_problem(p)) return rc;
  if (!occ_ptr || !occ_rows || !entity_params || !bias_params || !inv_occ || !scalars || !W ||
      !sumz || !grow || !partials || !grad_out || !g_entity || !g_bias || !g_scalars)
    return fail(VFM_E_INVALID, "vfm_elbo_bwd_f32: NULL pointer");
  const int neps = (eps_entity != nullptr) + (eps_bias != nullptr) + (eps_global != nullptr);
  if (neps != 0 && neps != 3) return fail(VFM_E_INVALID, "vfm_elbo_bwd_f32: give all three eps tables or none");
  KArgs a = make_args(p, nullptr, nullptr, entity_params, bias_params, inv_occ, scalars, W, eps_entity,
                      eps_bias, eps_global);
  BwdArgs b{occ_ptr, occ_rows, sumz, grow, partials, grad_out, g_entity, g_bias, g_scalars};
  Shape s;
  pick_shape(p->d, &s);
  if (int rc = dispatch<LaunchBwd>(s, a, b, (hipStream_t)stream)) return rc;
  return after_launch("vfm_elbo_bwd_f32");
}

int vfm_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                 float beta2, float eps, int64_t step, void* stream) {
  if (!p || !g || !m || !v || n < 0 || step < 1) return fail(VFM_E_INVALID, "vfm_adam_f32: bad argument");
  if (n == 0) return 0;
  if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) != 0)
    return fail(VFM_E_INVALID, "vfm_adam_f32: pointers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const int64_t n4 = n / 4;
  int64_t nb = (n4 + BLOCK - 1) / BLOCK;
  if (nb < 1) nb = 1;
  const int grid = (int)(nb < 4096 ? nb : 4096);
  hipLaunchKernelGGL(k_adam, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, p, g, m, v, n4, n, beta1,
                     beta2, eps, step_size, bc2_sqrt);
  return after_launch("vfm_adam_f32");
}

int vfm_philox_eps_f32(const vfm_problem_t* p, float* eps_entity, float* eps_bias, float* eps_global,
                       void* stream) {
  if (int rc = check_problem(p)) return rc;
  if (!eps_entity || !eps_bias || !eps_global) return fail(VFM_E_INVALID, "vfm_philox_eps_f32: NULL pointer");
  KArgs a = make_args(p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                      nullptr);
  hipLaunchKernelGGL(k_philox_dump, dim3(1024), dim3(256), 0, (hipStream_t)stream, a, eps_entity, eps_bias,
                     eps_global);
  return after_launch("vfm_philox_eps_f32");
}

}  // extern "C"
