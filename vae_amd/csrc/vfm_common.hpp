// vfm_common.hpp -- kernel argument structs, chunk loads/stores, lane-group reductions, KL helpers.
// Included inside `namespace vfm { namespace {` of a translation unit (argument structs: vfm_args.hpp).
#pragma once

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

// streaming (non-temporal) forms for data that is read / written once per step and is far larger
// than the caches (Adam moments, dense gradient rows).  Measured at cfg3 (same box, A/B): with the
// moments streamed `nt` the fused backward+Adam kernel takes 200 us instead of 228 us, and the NEXT
// forward 41.7 us instead of 47.3 us -- the 340 MB of moments no longer evict the 169 MB parameter
// table and the 51 MB sumz buffer from the 256 MB Infinity Cache.
typedef float v4f __attribute__((ext_vector_type(4)));
template <int VEC>
__device__ __forceinline__ Chunk<VEC> ld_chunk_nt(const float* p) {
  Chunk<VEC> c;
  if constexpr (VEC == 4) {
    const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    c.v[0] = t.x; c.v[1] = t.y; c.v[2] = t.z; c.v[3] = t.w;
  } else {
    c.v[0] = __builtin_nontemporal_load(p);
  }
  return c;
}

template <int VEC>
__device__ __forceinline__ void st_chunk(float* p, const Chunk<VEC>& c);

template <int VEC>
__device__ __forceinline__ void st_chunk_nt(float* p, const Chunk<VEC>& c) {
  if constexpr (VEC == 4) {
    const v4f t = {c.v[0], c.v[1], c.v[2], c.v[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
  } else {
    __builtin_nontemporal_store(c.v[0], p);
  }
}

template <int VEC>
__device__ __forceinline__ void st_chunk(float* p, const Chunk<VEC>& c) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(c.v[0], c.v[1], c.v[2], c.v[3]);
  } else {
    *p = c.v[0];
  }
}

// Guard for the |.| link (vfm-torch.py:126): a scale parameter that an Adam update lands on
// EXACTLY 0.0f makes -log|s| and 1/|s| infinite (the reference would raise in
// Normal(scale=0) / produce NaN).  With 2*10^7 scale parameters and lr-sized steps this exact
// cancellation does happen within ~100 steps at ML-20M shape, so: log and 1/sigma use
// max(|s|, SIGMA_MIN) and sign(0) := +1.  Identical to the reference wherever the reference is finite
// and |s| >= SIGMA_MIN.
constexpr float SIGMA_MIN = 1e-12f;

__device__ __forceinline__ float kl_std_normal(float mu, float sg) {
  // KL(N(mu, sg) || N(0,1)) = 1/2 (sg^2 + mu^2 - 1) - log sg   (torch kl.py _kl_normal_normal)
  return 0.5f * (sg * sg + mu * mu - 1.0f) - LN2 * __builtin_amdgcn_logf(fmaxf(sg, SIGMA_MIN));
}

__device__ __forceinline__ float inv_sigma(float sg) { return 1.0f / fmaxf(sg, SIGMA_MIN); }

__device__ __forceinline__ float signf(float s) { return (s < 0.f) ? -1.f : 1.f; }

// Link function of the scale parameters (the reference's global LINK, vfm-torch.py:125-126):
//   LINK_ABS      sigma = |s|            dsigma/ds = sign(s)       (the assignment that wins, :126)
//   LINK_SOFTPLUS sigma = log(1 + e^s)   dsigma/ds = sigmoid(s)    (:125; vfm.py:88)
// softplus on the hardware exp2 / log2 units; log1p(t) = log(u) * t / (u - 1), u = 1 + t (exact
// where 1 + t rounds to 1) keeps sigma accurate for very negative s.
template <int LINK>
__device__ __forceinline__ float link_f(float s) {
  if constexpr (LINK == LINK_ABS) {
    return fabsf(s);
  } else {
    const float t = __builtin_amdgcn_exp2f(-LOG2E * fabsf(s));
    const float u = 1.0f + t;
    const float l1p = (u == 1.0f) ? t : LN2 * __builtin_amdgcn_logf(u) * (t / (u - 1.0f));
    return fmaxf(s, 0.f) + l1p;
  }
}
template <int LINK>
__device__ __forceinline__ float dlink_f(float s) {
  if constexpr (LINK == LINK_ABS) {
    return signf(s);
  } else {
    const float t = __builtin_amdgcn_exp2f(-LOG2E * fabsf(s));
    const float inv = 1.0f / (1.0f + t);
    return (s >= 0.f) ? inv : t * inv;
  }
}

// all-reduce (sum) over aligned groups of W lanes, on the VALU: DPP row operations inside a
// 16-lane row, v_permlane16/32_swap (gfx950) across rows -- no LDS round trips.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

template <int W>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (W >= 2) v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
  if constexpr (W >= 4) v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
  if constexpr (W >= 8) v += dpp_f<0x141>(v);   // row_half_mirror
  if constexpr (W >= 16) v += dpp_f<0x140>(v);  // row_mirror
  if constexpr (W >= 32) {
    const int iv = __builtin_bit_cast(int, v);
    const auto r = __builtin_amdgcn_permlane16_swap(iv, iv, false, false);
    v = __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
  }
  if constexpr (W >= 64) {
    const int iv = __builtin_bit_cast(int, v);
    const auto r = __builtin_amdgcn_permlane32_swap(iv, iv, false, false);
    v = __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
  }
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

// log-likelihood of one (row, sample) and its derivative wrt the prediction.  aterm: the POSITIVE half of the row's
// share of dloss/d|alpha| = (y - pred)^2 / 2 - 1 / (2 |alpha|): the lanes add up positive terms only, the constant
// n_terms / (2 |alpha|) is taken off the total once, in fp64, when the slots are reduced (VFM_P_ALPHA, vfm_reduce.hpp).
// (Per row, in fp32, the difference cancels -- at the optimum the sum is zero -- and the gradient of alpha came out
//  with three digits fewer than every other number of the step.)
__device__ __forceinline__ void lik_terms(int lik, float y, float pred, float aabs, float half_log_a, float& ll,
                                          float& dll, float& aterm) {
  if (lik == VFM_LIK_NORMAL) {
    const float diff = y - pred;
    ll = -0.5f * aabs * diff * diff + half_log_a - LOG_SQRT_2PI;
    dll = aabs * diff;
    aterm = 0.5f * diff * diff;
  } else {
    // log-sigmoid on the hardware exp2/log2 units: softplus(x) = max(x,0) + ln(1 + e^-|x|)
    const float e1 = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(pred));
    ll = y * pred - (fmaxf(pred, 0.f) + LN2 * __builtin_amdgcn_logf(1.0f + e1));
    const float inv = __builtin_amdgcn_rcpf(1.0f + e1);
    // y - sigmoid(pred) without forming a sigmoid near 1: for pred >= 0, 1 - sigmoid = e1 / (1 + e1), so
    // y - sigmoid = (y - 1) + e1 / (1 + e1) -- exact for y = 1, where  1 - 0.9995  in fp32 kept three digits (the worst
    // scalar gradient of the 2,500-configuration fuzz: 0.236 of its summation bound, a saturated row at B = 2)
    const float t = e1 * inv;
    dll = (pred >= 0.f) ? (y - 1.0f) + t : y - t;
    aterm = 0.f;
  }
}


// what a forward workgroup writes into its slot's VFM_SLOT_NTERMS entry: the number of Normal-likelihood terms the
// LAUNCH summed into the slots' VFM_P_ALPHA entries (B rows x S samples), carried by workgroup 0 alone
__device__ __forceinline__ double slot_nterms(const KArgs& a, bool train_final) {
  return (train_final && blockIdx.x == 0 && a.lik == VFM_LIK_NORMAL) ? (double)a.B * (double)a.S : 0.0;
}
