// vfm_common.hpp -- kernel argument structs, chunk loads/stores, lane-group reductions, KL helpers.
// Part of vfm_kernels.hip (one translation unit; included inside its anonymous namespace).
#pragma once

// ---------------------------------------------------------------------------------------
// Kernel arguments (by value)
// ---------------------------------------------------------------------------------------
struct KArgs {
  int64_t B, T;
  int64_t e_lo, e_hi;   // entity range of a backward launch (chunked multi-rank pipeline)
  int32_t own_mod, own_rank;   // entity-sharded apply: this rank owns e = own_rank (mod own_mod)
  int32_t F, d, lik, id64, G, flags;
  float ll_scale;  // nb_train / B_global
  double ll_scale_d;
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
  double* partials;
  const float* grad_out;
  float* g_entity;
  float* g_bias;
  float* g_scalars;
  float* loss;   // non-NULL: this launch also reduces the forward's partial slots and forms the loss
  // staged (multi-rank) form: sufficient statistics of the gradient, exchanged instead of the gradient
  float* acc;    // [T, 4 + round4(d)] record per entity: (sum_r grow_r, occurrences, 0, 0 | A_e[0..d-1]),
                 //   A_e = sum_r grow_r * sumz_r      (STAGE_ACC writes, STAGE_APPLY reads)
  float* sums;   // [2]    (sum_r grow_r over all rows, alpha term)
  // entities whose occurrence list is longer than VFM_HEAVY_LIST: pre-reduced by k_heavy
  const int32_t* heavy_ids;   // [n_heavy] sorted
  const float* heavy_acc;     // [n_heavy, 4 + round4(d)] records (sum grow, count, 0, 0 | A_e)
  int32_t n_heavy;
  double* kl_slots;           // STAGE_APPLY: [0] = blocks, [1 + b] = block b's sum of c_e * KL_e (NULL: not wanted)
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
