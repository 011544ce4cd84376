// vfm_kernels.hip -- hand-written gfx950 (MI355X, CDNA4) kernels + C ABI for the
// Variational-FM ELBO step.  Wave = 64 lanes; no MFMA: the path is gather + elementwise +
// reduction and is bounded by HBM / Infinity-Cache bandwidth.
//
// Replaces the per-batch body of the reference: vfm-torch.py:189-324 (CF.forward), the loss
// line :359 and autograd through them (:368-369).  Math: SURVEY.md Appendix A.
//
// Kernels (details at each definition)
//   k_fwd     : a *lane group* of LPE lanes owns one batch row, each lane CPL chunks of VEC
//               coordinates; rows of a workgroup are contiguous; ids -> table rows -> arithmetic are
//               software-pipelined in registers; FM reduction with DPP / permlane swaps; per-block
//               partial sums to private slots (no atomics).
//   k_finalize: adds the slots, forms the loss (also foldable into the fused backward).
//   k_bwd     : entity-centric.  A lane group owns one TABLE row e, sums grow[r]*sumz[r,:] over the
//               batch rows containing e (inverted index) and either stores the dense gradient row,
//               applies dense Adam in place (ADAM), or -- multi-rank -- stores / consumes the
//               gradient's sufficient statistics (STAGE_ACC / STAGE_APPLY).
//   k_heavy   : parallel pre-reduction of occurrence lists longer than VFM_HEAVY_LIST (skewed data).
//   k_adam    : dense Adam on a flat buffer (unfused path).  k_norms, k_inv_occ: per-batch / per-dataset
//               normalisers.  k_philox_dump: the eps stream, for tests.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>

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
constexpr float LOG_SQRT_2PI = 0.918938533204672742f;
constexpr float LN2 = 0.693147180559945309f;

// ---------------------------------------------------------------------------------------
// Counter-based RNG: Philox4x32-10 (Salmon et al. 2011) + Box-Muller on the hardware
// transcendental units (v_log_f32 / v_sqrt_f32 / v_sin_f32 / v_cos_f32).  The draw of entity e at
// step `step` depends on (seed, step, e, coordinate) only: every row -- and every rank -- that
// touches e regenerates the same eps, no eps tensor is ever stored.
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

// Box-Muller pair from RB radius bits and AB angle bits (v_sin / v_cos take revolutions).  Both
// uniforms sit at bin centres: u1 in (0,1) so log is finite, and no angle lands exactly on an axis.
template <int RB, int AB>
__device__ __forceinline__ void box_muller_bits(uint32_t f, float& n0, float& n1) {
  const float u1 = fmaf((float)((f >> AB) & ((1u << RB) - 1u)), 1.0f / (float)(1u << RB), 0.5f / (float)(1u << RB));
  const float u2 = fmaf((float)(f & ((1u << AB) - 1u)), 1.0f / (float)(1u << AB), 0.5f / (float)(1u << AB));
  const float r = __builtin_amdgcn_sqrtf(-2.0f * LN2 * __builtin_amdgcn_logf(u1));
  n0 = r * __builtin_amdgcn_cosf(u2);
  n1 = r * __builtin_amdgcn_sinf(u2);
}

// One Philox4x32-10 call with counter (p, e, step_lo, step_hi) and key (seed_lo, seed_hi) yields
// 128 bits = four 26-bit fields (16-bit radius, 10-bit angle) + one 24-bit field (16 + 8):
//   n[0..7] : eps of embedding coordinates 8p .. 8p+7 of entity e at this step (four Box-Muller pairs)
//   nb      : eps of the entity's first-order weight (first normal of the fifth pair), used for p == 0
// The global-bias eps is n[0] of the reserved id e = 0xFFFFFFFF, p = 0 (one draw per entity per
// step, as the reference's per-unique-entity rsample, vfm-torch.py:207-208,238-245).
__device__ __forceinline__ void normal8b(const RngKey& k, uint32_t e, uint32_t p, float n[8], float& nb) {
  uint32_t o[4];
  philox4x32_10(p, e, k.step_lo, k.step_hi, k.seed_lo, k.seed_hi, o);
  box_muller_bits<16, 10>(o[0], n[0], n[1]);                                        // o0[25:0]
  box_muller_bits<16, 10>(__builtin_amdgcn_alignbit(o[1], o[0], 26), n[2], n[3]);   // o1[19:0] : o0[31:26]
  box_muller_bits<16, 10>(__builtin_amdgcn_alignbit(o[2], o[1], 20), n[4], n[5]);   // o2[13:0] : o1[31:20]
  box_muller_bits<16, 10>(__builtin_amdgcn_alignbit(o[3], o[2], 14), n[6], n[7]);   // o3[7:0]  : o2[31:14]
  float unused;
  box_muller_bits<16, 8>(o[3] >> 8, nb, unused);                                    // o3[31:8]
}

// eps of chunk j (VEC coordinates from j*VEC) for the lane that owns it, plus the bias eps
template <int VEC>
__device__ __forceinline__ void eps_of_chunk(const RngKey& k, uint32_t e, int j, float (&ep)[VEC], float& nb) {
  float n[8];
  if constexpr (VEC == 4) {
    normal8b(k, e, (uint32_t)j >> 1, n, nb);
    const bool odd = j & 1;
#pragma unroll
    for (int t = 0; t < 4; ++t) ep[t] = odd ? n[4 + t] : n[t];
  } else {
    normal8b(k, e, (uint32_t)j >> 3, n, nb);
    float v = n[0];
#pragma unroll
    for (int t = 1; t < 8; ++t) v = ((j & 7) == t) ? n[t] : v;
    ep[0] = v;
  }
}

// ---------------------------------------------------------------------------------------
// Kernel arguments (by value)
// ---------------------------------------------------------------------------------------
struct KArgs {
  int64_t B, T;
  int64_t e_lo, e_hi;   // entity range of a backward launch (chunked multi-rank pipeline)
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

__global__ __launch_bounds__(BLOCK) void k_norms(const void* __restrict__ x, int id64,
                                                 const float* __restrict__ inv_occ, int64_t n_occ,
                                                 int F, int64_t T, double* __restrict__ W) {
  __shared__ float sh[VFM_MAX_FIELDS];
  if (threadIdx.x < VFM_MAX_FIELDS) sh[threadIdx.x] = 0.f;
  __syncthreads();
  // each thread walks occurrences o = t, t + stride...; stride is a multiple of F so the
  // field of a thread is fixed
  const int64_t stride0 = (int64_t)gridDim.x * BLOCK;
  const int64_t stride = stride0 / F * F;   // (threads t >= stride stay idle)
  const int64_t t = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
  float acc = 0.f;
  for (int64_t o = (t < stride ? t : n_occ); o < n_occ; o += stride) {
    int64_t id = id64 ? ((const int64_t*)x)[o] : (int64_t)((const int32_t*)x)[o];
    if (id >= 0 && id < T) acc += inv_occ[id];
  }
  if (acc != 0.f) atomicAdd(&sh[t % F], acc);
  __syncthreads();
  if (threadIdx.x < F) atomicAdd(&W[threadIdx.x], (double)sh[threadIdx.x]);
}

// Reduce the forward's per-workgroup slots into partials[0..4] and form the loss triple.  Called by
// all BLOCK threads of ONE workgroup; the totals are valid in thread 0 (and in memory) afterwards.
__device__ __forceinline__ void reduce_slots_and_loss(double* __restrict__ partials,
                                                      const float* __restrict__ scalars, double ll_scale,
                                                      int flags, float* __restrict__ loss, double (*sh)[BLOCK / 64],
                                                      double (&tot)[5]) {
  const int nblk = (int)partials[7];
  double acc[5] = {0, 0, 0, 0, 0};
  for (int b = threadIdx.x; b < nblk; b += BLOCK) {
    const double* slot = partials + VFM_N_PARTIALS * (1 + (size_t)b);
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[i] += slot[i];
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc[i] += __shfl_xor(acc[i], m, 64);
    if ((threadIdx.x & 63) == 0) sh[i][threadIdx.x >> 6] = acc[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    tot[i] = 0;
    for (int w = 0; w < BLOCK / 64; ++w) tot[i] += sh[i][w];
  }
  if (threadIdx.x != 0) return;
#pragma unroll
  for (int i = 0; i < 5; ++i) partials[i] = tot[i];
  const double m0 = scalars[1], s0 = scalars[2];
  const double kl0 = (flags & VFM_FLAG_NO_PRIOR_TERMS)
                         ? 0.0
                         : 0.5 * (s0 * s0 + m0 * m0 - 1.0) - log(fmax(fabs(s0), (double)SIGMA_MIN));
  const double nll = -ll_scale * tot[VFM_P_LL];
  const double kl = kl0 + tot[VFM_P_KL];
  const bool bad = tot[VFM_P_BADID] != 0.0;
  const float nanv = __builtin_nanf("");
  loss[0] = bad ? nanv : (float)(nll + kl);
  loss[1] = bad ? nanv : (float)nll;
  loss[2] = bad ? nanv : (float)kl;
}

__global__ __launch_bounds__(BLOCK) void k_finalize(double* __restrict__ partials,
                                                    const float* __restrict__ scalars, double ll_scale,
                                                    int flags, float* __restrict__ loss) {
  __shared__ double sh[5][BLOCK / 64];
  double tot[5];
  reduce_slots_and_loss(partials, scalars, ll_scale, flags, loss, sh, tot);
}

// ---------------------------------------------------------------------------------------
// forward
//
// A lane group of LPE lanes owns one batch row at a time (rows are dealt round-robin over all
// groups of the grid); lane `lig` owns chunks j = lig + i*LPE (i < CPL) of VEC coordinates.
// Nothing is shared between groups: no LDS staging, no barrier in the row loop.  The dependent
// chain per row is  ids -> table rows  and it is software-pipelined three deep:
//     ids of row i+2  |  table-row loads of row i+1 (registers)  |  arithmetic of row i
// so that every wave keeps 8d*F bytes per row in flight while the Philox / Box-Muller /
// KL arithmetic of the previous row runs.  FF = 2 keeps both fields of a row in registers
// (the reference's user/item case); FF = 0 streams a runtime number of fields.
// ---------------------------------------------------------------------------------------
enum { EPS_PHILOX = 0, EPS_TABLE = 1, EPS_ZERO = 2 };
enum { MODE_PREDICT = 0, MODE_TRAIN = 1 };

template <int CPL, int VEC, int EPS>
struct FieldRegs {            // everything one (row, field) occurrence needs, in registers
  uint32_t e;
  Chunk<VEC> mu[CPL], s[CPL], ep[CPL];
  float2 th;                  // bias row (mu_w, s_w)
  float io;                   // 1/occ
  float epw;                  // bias eps (table mode)
};

template <int LPE, int CPL, int VEC, int EPS, int MODE>
__device__ __forceinline__ void load_field(const KArgs& a, uint32_t e, int lig, int C,
                                           FieldRegs<CPL, VEC, EPS>& R) {
  const int d = a.d;
  R.e = e;
  const float* row = a.entity + (size_t)e * (2 * (size_t)d);
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    // lanes past the last chunk re-load the last chunk (same lines, no branch); consume_field masks them
    int j = lig + i * LPE;
    j = j < C ? j : C - 1;
    R.mu[i] = ld_chunk<VEC>(row + (size_t)j * VEC);
    R.s[i] = ld_chunk<VEC>(row + d + (size_t)j * VEC);
    if constexpr (EPS == EPS_TABLE) R.ep[i] = ld_chunk<VEC>(a.eps_entity + (size_t)e * d + (size_t)j * VEC);
  }
  R.th = *reinterpret_cast<const float2*>(a.bias + 2 * (size_t)e);
  if constexpr (MODE == MODE_TRAIN) R.io = a.inv_occ[e];
  if constexpr (EPS == EPS_TABLE) R.epw = a.eps_bias[e];
}

// Raw id of occurrence `pos` (not inspected here: looking at the value would force a wait on
// every load in flight; the range check happens one pipeline stage later, in check_id).
template <bool ID64>
struct RawId { uint32_t lo, hi; };

template <bool ID64>
__device__ __forceinline__ RawId<ID64> load_raw_id(const KArgs& a, int64_t pos) {
  RawId<ID64> r;
  if constexpr (ID64) {
    const uint2 v = reinterpret_cast<const uint2*>(a.x)[pos];
    r.lo = v.x; r.hi = v.y;
  } else {
    r.lo = reinterpret_cast<const uint32_t*>(a.x)[pos];
    r.hi = (r.lo >> 31) ? 0xFFFFFFFFu : 0u;   // sign extension of an int32 id
  }
  return r;
}

template <bool ID64>
__device__ __forceinline__ uint32_t check_id(const KArgs& a, const RawId<ID64>& r, float& bad) {
  const bool ok = (r.hi == 0u) && ((int64_t)r.lo < a.T);
  if (!ok) bad += 1.f;
  return ok ? r.lo : 0u;
}

// per-row running sums of one lane
template <int CPL, int VEC>
struct RowAcc {
  Chunk<VEC> sz[CPL];
  float zz, part, kl;
  __device__ __forceinline__ void reset() {
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
      for (int t = 0; t < VEC; ++t) sz[i].v[t] = 0.f;
    zz = 0.f; part = 0.f; kl = 0.f;
  }
};

typedef float v2f __attribute__((ext_vector_type(2)));

// z = mu + |s| eps for one chunk, FM partial sums and the KL polynomial / log parts.
// VEC == 4 uses packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two coordinates per
// instruction).  `valid` masks the lanes past the last chunk (they hold a re-loaded copy).
template <int VEC, int MODE>
__device__ __forceinline__ void chunk_math(const Chunk<VEC>& mu, const Chunk<VEC>& s, const float (&ep)[VEC],
                                           bool valid, Chunk<VEC>& sz, float& zz, float& klv) {
  if constexpr (VEC == 4) {
    v2f zq = {0.f, 0.f}, kq = {0.f, 0.f};
    float lg = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const v2f m2 = {mu.v[2 * h], mu.v[2 * h + 1]};
      const v2f g2 = {fabsf(s.v[2 * h]), fabsf(s.v[2 * h + 1])};
      const v2f e2 = {ep[2 * h], ep[2 * h + 1]};
      const v2f z2 = g2 * e2 + m2;
      v2f a2 = {sz.v[2 * h], sz.v[2 * h + 1]};
      a2 = valid ? a2 + z2 : a2;
      sz.v[2 * h] = a2.x; sz.v[2 * h + 1] = a2.y;
      zq = z2 * z2 + zq;
      if constexpr (MODE == MODE_TRAIN) {
        kq = g2 * g2 + kq;
        kq = m2 * m2 + kq;
        // log s0 + log s1 = log(s0 * s1): one v_log_f32 per pair (the clamped product stays >= 1e-24)
        lg += __builtin_amdgcn_logf(fmaxf(g2.x, SIGMA_MIN) * fmaxf(g2.y, SIGMA_MIN));
      }
    }
    zz += valid ? zq.x + zq.y : 0.f;
    if constexpr (MODE == MODE_TRAIN) klv += valid ? fmaf(0.5f, kq.x + kq.y, fmaf(-LN2, lg, -2.0f)) : 0.f;
  } else {
    const float sg = fabsf(s.v[0]);
    const float z = valid ? fmaf(sg, ep[0], mu.v[0]) : 0.f;
    sz.v[0] += z;
    zz = fmaf(z, z, zz);
    if constexpr (MODE == MODE_TRAIN) klv += valid ? kl_std_normal(mu.v[0], sg) : 0.f;
  }
}

// first-order weight of one occurrence (the lane that owns it): sample + KL
template <int MODE>
__device__ __forceinline__ void bias_math(const float2 th, float epw, bool owner, float& part, float& klv) {
  const float sgw = fabsf(th.y);
  part += owner ? fmaf(sgw, epw, th.x) : 0.f;
  if constexpr (MODE == MODE_TRAIN) klv += owner ? kl_std_normal(th.x, sgw) : 0.f;
}

// arithmetic of one occurrence (generic path): every lane draws its own chunk's eps
template <int LPE, int CPL, int VEC, int EPS, int MODE>
__device__ __forceinline__ void consume_field(const KArgs& a, const FieldRegs<CPL, VEC, EPS>& R, int lig,
                                              int C, float cs, RowAcc<CPL, VEC>& acc) {
  float klv = 0.f;
  float epw = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    int j = lig + i * LPE;
    const bool valid = j < C;
    j = valid ? j : C - 1;
    float ep[VEC];
    if constexpr (EPS == EPS_TABLE) {
#pragma unroll
      for (int t = 0; t < VEC; ++t) ep[t] = R.ep[i].v[t];
    } else if constexpr (EPS == EPS_ZERO) {
#pragma unroll
      for (int t = 0; t < VEC; ++t) ep[t] = 0.f;
    } else {
      float nb;
      eps_of_chunk<VEC>(a.key, R.e, j, ep, nb);
      if (i == 0) epw = nb;   // only the lane that owns coordinate 0 (lig == 0) uses it
    }
    chunk_math<VEC, MODE>(R.mu[i], R.s[i], ep, valid, acc.sz[i], acc.zz, klv);
  }
  if constexpr (EPS == EPS_TABLE) epw = R.epw;
  bias_math<MODE>(R.th, epw, lig == 0, acc.part, klv);
  if constexpr (MODE == MODE_TRAIN) acc.kl = fmaf(cs * R.io, klv, acc.kl);
}

// arithmetic of a two-field row (VEC == 4): ONE Philox call per lane serves both fields.  Lanes
// pair up (2m, 2m+1): the even lane draws the 8 normals of chunks (2m, 2m+1) of field 0's entity,
// the odd lane those of field 1's entity, and they exchange one half over DPP (quad_perm
// [1,0,3,2]).  Lane 0 / lane 1 own the first-order weights of field 0 / field 1 (their calls have
// p == 0 and carry the bias normal).
template <int LPE, int CPL, int EPS, int MODE>
__device__ __forceinline__ void consume_row2(const KArgs& a, const FieldRegs<CPL, 4, EPS>& R0,
                                             const FieldRegs<CPL, 4, EPS>& R1, int lig, int C, float cs0,
                                             float cs1, RowAcc<CPL, 4>& acc) {
  static_assert(LPE >= 2, "lane pairing needs at least two lanes per row");
  const bool odd = lig & 1;
  float kl0 = 0.f, kl1 = 0.f, epw = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    int j = lig + i * LPE;
    const bool valid = j < C;
    j = valid ? j : C - 1;
    float ep0[4], ep1[4];
    if constexpr (EPS == EPS_TABLE) {
#pragma unroll
      for (int t = 0; t < 4; ++t) { ep0[t] = R0.ep[i].v[t]; ep1[t] = R1.ep[i].v[t]; }
    } else if constexpr (EPS == EPS_ZERO) {
#pragma unroll
      for (int t = 0; t < 4; ++t) { ep0[t] = 0.f; ep1[t] = 0.f; }
    } else {
      float n[8], nb;
      // pair index of chunk j is j >> 1 (LPE is even, so both lanes of a pair agree on it)
      normal8b(a.key, odd ? R1.e : R0.e, (uint32_t)j >> 1, n, nb);
      if (i == 0) epw = nb;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float send = odd ? n[t] : n[4 + t];     // what the partner lane needs from me
        const float recv = dpp_f<0xB1>(send);
        ep0[t] = odd ? recv : n[t];                   // field 0, my chunk
        ep1[t] = odd ? n[4 + t] : recv;               // field 1, my chunk
      }
    }
    chunk_math<4, MODE>(R0.mu[i], R0.s[i], ep0, valid, acc.sz[i], acc.zz, kl0);
    chunk_math<4, MODE>(R1.mu[i], R1.s[i], ep1, valid, acc.sz[i], acc.zz, kl1);
  }
  if constexpr (EPS == EPS_TABLE) epw = odd ? R1.epw : R0.epw;
  float klb = 0.f;
  bias_math<MODE>(odd ? R1.th : R0.th, epw, lig < 2, acc.part, klb);
  if constexpr (MODE == MODE_TRAIN) {
    const float c0 = cs0 * R0.io, c1 = cs1 * R1.io;
    acc.kl = fmaf(c0, kl0, fmaf(c1, kl1, fmaf(odd ? c1 : c0, klb, acc.kl)));
  }
}

// finish a row: FM reduction over the group, likelihood, outputs
template <int LPE, int CPL, int VEC, int MODE>
__device__ __forceinline__ void finish_row(const KArgs& a, const FwdOut& out, int64_t r, int lig, int C,
                                           float w0, float aabs, float half_log_a, float y,
                                           RowAcc<CPL, VEC>& acc, float (&tot)[5]) {
  float q = -acc.zz;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    float qi = 0.f;
#pragma unroll
    for (int t = 0; t < VEC; ++t) qi = fmaf(acc.sz[i].v[t], acc.sz[i].v[t], qi);
    q += (lig + i * LPE < C) ? qi : 0.f;
  }
  const float pred = w0 + group_sum<LPE>(fmaf(0.5f, q, acc.part));
  if constexpr (MODE == MODE_TRAIN) {
    tot[1] += acc.kl;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int j = lig + i * LPE;
      if (j < C) st_chunk<VEC>(out.sumz + (size_t)r * a.d + (size_t)j * VEC, acc.sz[i]);
    }
  }
  if (lig == 0) {
    out.pred[r] = pred;
    if constexpr (MODE == MODE_TRAIN) {
      float ll, dll;
      if (a.lik == VFM_LIK_NORMAL) {
        const float diff = y - pred;
        ll = -0.5f * aabs * diff * diff + half_log_a - LOG_SQRT_2PI;
        dll = aabs * diff;
        tot[3] += 0.5f * diff * diff - 0.5f / aabs;
      } else {
        // log-sigmoid on the hardware exp2/log2 units: softplus(x) = max(x,0) + ln(1 + e^-|x|)
        const float e1 = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(pred));
        ll = y * pred - (fmaxf(pred, 0.f) + LN2 * __builtin_amdgcn_logf(1.0f + e1));
        const float inv = __builtin_amdgcn_rcpf(1.0f + e1);
        dll = y - ((pred >= 0.f) ? inv : e1 * inv);
      }
      const float g = -a.ll_scale * dll;
      tot[0] += ll;
      tot[2] += g;
      out.grow[r] = g;
    }
  }
}

template <int LPE, int CPL, int VEC, int EPS, int MODE, int FF, bool ID64>
__global__ __launch_bounds__(BLOCK) void k_fwd(const KArgs a, const FwdOut out) {
  constexpr int GPB = BLOCK / LPE;
  __shared__ float sh_cs[VFM_MAX_FIELDS];
  __shared__ int64_t sh_hi[VFM_MAX_FIELDS];
  __shared__ float sh_red[5 * 4];

  const int tid = threadIdx.x;
  const int lig = tid % LPE;
  const int F = (FF > 0) ? FF : a.F;
  const int C = (a.d + VEC - 1) / VEC;

  if (MODE == MODE_TRAIN && tid < a.G) {
    sh_cs[tid] = (float)(a.group_n[tid] / a.W[tid]);
    sh_hi[tid] = a.group_hi[tid];
  }
  const float alpha = a.scalars[0], m0 = a.scalars[1], s0 = a.scalars[2];
  const float aabs = fabsf(alpha);
  float e0 = 0.f;
  if constexpr (EPS == EPS_TABLE) e0 = a.eps_global[0];
  if constexpr (EPS == EPS_PHILOX) {
    float n[8], nb;
    normal8b(a.key, 0xFFFFFFFFu, 0u, n, nb);
    e0 = n[0];
  }
  const float w0 = fmaf(fabsf(s0), e0, m0);
  const float half_log_a = 0.5f * LN2 * __builtin_amdgcn_logf(aabs);
  if (MODE == MODE_TRAIN) __syncthreads();

  float tot[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // ll, kl, g, alpha-term, bad ids
  // Each workgroup owns a CONTIGUOUS chunk of rows (its GPB lane groups interleave inside it): with
  // the rows of a batch ordered by item id, the rows that share an item row are then gathered by the
  // same CU at about the same time and hit L1 / the XCD's L2 instead of HBM.
  int64_t rpb = (a.B + gridDim.x - 1) / gridDim.x;
  rpb = (rpb + GPB - 1) / GPB * GPB;
  const int64_t rbeg = (int64_t)blockIdx.x * rpb;
  const int64_t rend = (rbeg + rpb < a.B) ? rbeg + rpb : a.B;    // this block's rows: [rbeg, rend)
  const int64_t ngroups = GPB;                                    // row stride of a lane group
  const int64_t g0 = rbeg + tid / LPE;
  const int64_t Bm1 = rend - 1;

  if constexpr (FF == 2 && VEC == 4 && LPE >= 2) {
    // ---- two fields per row, both in registers; double buffer across rows.  Rows past the end
    // are clamped to the last row for the (harmless, branch-free) prefetches. ----
    float cs0 = 0.f, cs1 = 0.f;
    int64_t hi0 = 0;
    if constexpr (MODE == MODE_TRAIN) { cs0 = sh_cs[0]; cs1 = sh_cs[1]; hi0 = sh_hi[0]; }
    FieldRegs<CPL, VEC, EPS> A0, A1, B0, B1;
    float yA = 0.f, yB = 0.f;
    int64_t r = g0;
    if (r < rend) {
      RawId<ID64> i0 = load_raw_id<ID64>(a, r * 2), i1 = load_raw_id<ID64>(a, r * 2 + 1);
      const int64_t r1 = (r + ngroups < rend) ? r + ngroups : Bm1;
      RawId<ID64> n0 = load_raw_id<ID64>(a, r1 * 2), n1 = load_raw_id<ID64>(a, r1 * 2 + 1);
      load_field<LPE, CPL, VEC, EPS, MODE>(a, check_id<ID64>(a, i0, tot[4]), lig, C, A0);
      load_field<LPE, CPL, VEC, EPS, MODE>(a, check_id<ID64>(a, i1, tot[4]), lig, C, A1);
      if constexpr (MODE == MODE_TRAIN) yA = a.y[r];
      RowAcc<CPL, VEC> acc;
      while (true) {
        // stage 1: table rows of row r+ng into B (ids arrived a stage ago), ids of row r+2ng
        int64_t rn = r + ngroups;
        {
          const int64_t rc = rn < rend ? rn : Bm1;
          const int64_t r2 = (rn + ngroups < rend) ? rn + ngroups : Bm1;
          const bool live = rn < rend;
          float badn = 0.f;
          const uint32_t e0n = check_id<ID64>(a, n0, badn), e1n = check_id<ID64>(a, n1, badn);
          if (live) tot[4] += badn;
          n0 = load_raw_id<ID64>(a, r2 * 2);
          n1 = load_raw_id<ID64>(a, r2 * 2 + 1);
          load_field<LPE, CPL, VEC, EPS, MODE>(a, e0n, lig, C, B0);
          load_field<LPE, CPL, VEC, EPS, MODE>(a, e1n, lig, C, B1);
          if constexpr (MODE == MODE_TRAIN) yB = a.y[rc];
        }
        // stage 2: arithmetic of row r from A while B's loads are in flight
        acc.reset();
        consume_row2<LPE, CPL, EPS, MODE>(a, A0, A1, lig, C, ((int64_t)A0.e < hi0) ? cs0 : cs1,
                                          ((int64_t)A1.e < hi0) ? cs0 : cs1, acc);
        finish_row<LPE, CPL, VEC, MODE>(a, out, r, lig, C, w0, aabs, half_log_a, yA, acc, tot);
        r = rn;
        if (r >= rend) break;
        // the same with the roles of A and B swapped (static register naming, no copies)
        rn = r + ngroups;
        {
          const int64_t rc = rn < rend ? rn : Bm1;
          const int64_t r2 = (rn + ngroups < rend) ? rn + ngroups : Bm1;
          const bool live = rn < rend;
          float badn = 0.f;
          const uint32_t e0n = check_id<ID64>(a, n0, badn), e1n = check_id<ID64>(a, n1, badn);
          if (live) tot[4] += badn;
          n0 = load_raw_id<ID64>(a, r2 * 2);
          n1 = load_raw_id<ID64>(a, r2 * 2 + 1);
          load_field<LPE, CPL, VEC, EPS, MODE>(a, e0n, lig, C, A0);
          load_field<LPE, CPL, VEC, EPS, MODE>(a, e1n, lig, C, A1);
          if constexpr (MODE == MODE_TRAIN) yA = a.y[rc];
        }
        acc.reset();
        consume_row2<LPE, CPL, EPS, MODE>(a, B0, B1, lig, C, ((int64_t)B0.e < hi0) ? cs0 : cs1,
                                          ((int64_t)B1.e < hi0) ? cs0 : cs1, acc);
        finish_row<LPE, CPL, VEC, MODE>(a, out, r, lig, C, w0, aabs, half_log_a, yB, acc, tot);
        r = rn;
        if (r >= rend) break;
      }
    }
  } else {
    // ---- runtime number of fields: stream the occurrences (r, f), double buffer across them ----
    auto raw = [&](int64_t pos) -> RawId<true> {
      RawId<true> v;
      if (a.id64) { const uint2 t = reinterpret_cast<const uint2*>(a.x)[pos]; v.lo = t.x; v.hi = t.y; }
      else { v.lo = reinterpret_cast<const uint32_t*>(a.x)[pos]; v.hi = (v.lo >> 31) ? 0xFFFFFFFFu : 0u; }
      return v;
    };
    auto cs_of = [&](uint32_t e, int fcol) -> float {
      if constexpr (MODE != MODE_TRAIN) return 0.f;
      const int64_t id = (int64_t)e;
      const int64_t lo = fcol > 0 ? sh_hi[fcol - 1] : 0;
      if (id >= lo && id < sh_hi[fcol]) return sh_cs[fcol];   // the usual case: column f <-> group f
      return sh_cs[group_index(sh_hi, a.G, id)];
    };
    FieldRegs<CPL, VEC, EPS> A, Bq;
    RowAcc<CPL, VEC> acc;
    int64_t r = g0;
    int f = 0;
    if (r < rend) {
      const int64_t last = rend * F - 1;
      load_field<LPE, CPL, VEC, EPS, MODE>(a, check_id<true>(a, raw(r * F), tot[4]), lig, C, A);
      // position of the occurrence after the current one (clamped), its id prefetched
      auto next_pos = [&](int64_t rr, int ff, int64_t& rn, int& fn) {
        fn = ff + 1; rn = rr;
        if (fn == F) { fn = 0; rn = rr + ngroups; }
      };
      int64_t rn; int fn;
      next_pos(r, f, rn, fn);
      RawId<true> nid = raw(rn < rend ? rn * F + fn : last);
      acc.reset();
      while (true) {
        // stage 1: table row of the next occurrence, id of the one after
        {
          const bool live = rn < rend;
          float badn = 0.f;
          const uint32_t en = check_id<true>(a, nid, badn);
          if (live) tot[4] += badn;
          int64_t r2; int f2;
          next_pos(rn, fn, r2, f2);
          nid = raw(r2 < rend ? r2 * F + f2 : last);
          load_field<LPE, CPL, VEC, EPS, MODE>(a, en, lig, C, Bq);
        }
        consume_field<LPE, CPL, VEC, EPS, MODE>(a, A, lig, C, cs_of(A.e, f), acc);
        if (f == F - 1) {
          float y = 0.f;
          if constexpr (MODE == MODE_TRAIN) y = a.y[r];
          finish_row<LPE, CPL, VEC, MODE>(a, out, r, lig, C, w0, aabs, half_log_a, y, acc, tot);
          acc.reset();
        }
        r = rn; f = fn;
        if (r >= rend) break;
        next_pos(r, f, rn, fn);
        {
          const bool live = rn < rend;
          float badn = 0.f;
          const uint32_t en = check_id<true>(a, nid, badn);
          if (live) tot[4] += badn;
          int64_t r2; int f2;
          next_pos(rn, fn, r2, f2);
          nid = raw(r2 < rend ? r2 * F + f2 : last);
          load_field<LPE, CPL, VEC, EPS, MODE>(a, en, lig, C, A);
        }
        consume_field<LPE, CPL, VEC, EPS, MODE>(a, Bq, lig, C, cs_of(Bq.e, f), acc);
        if (f == F - 1) {
          float y = 0.f;
          if constexpr (MODE == MODE_TRAIN) y = a.y[r];
          finish_row<LPE, CPL, VEC, MODE>(a, out, r, lig, C, w0, aabs, half_log_a, y, acc, tot);
          acc.reset();
        }
        r = rn; f = fn;
        if (r >= rend) break;
        next_pos(r, f, rn, fn);
      }
    }
  }
  // per-block partial sums go to the block's own slot (plain stores: no same-address atomics --
  // 5 fp64 atomics from each of ~10^3 blocks finishing together serialised for tens of
  // microseconds -- and the sums become bitwise reproducible); k_finalize adds the slots up.
  block_sum<5>(tot, sh_red);
  if (tid == 0) {
    double* slot = out.partials + VFM_N_PARTIALS * (1 + (size_t)blockIdx.x);
#pragma unroll
    for (int i = 0; i < 5; ++i) slot[i] = (double)tot[i];
    if (blockIdx.x == 0) out.partials[7] = (double)gridDim.x;
  }
}

// ---------------------------------------------------------------------------------------
// backward (entity-centric, dense gradient rows, no atomics) -- optionally with the dense Adam
// update fused in (ADAM = 1): the gradient row never leaves registers.
//
// A lane group owns one TABLE row e: it sums grow[r] * sumz[r,:] over the batch rows that contain
// e (inverted index occ_ptr / occ_rows), adds the KL part, and either stores the dense gradient
// row (zeros when e is not in the batch: the reference's nn.Embedding gradients are dense) or
// applies torch.optim.Adam's update to (p, m, v) of that row in place.  Only e's own parameters
// are read, so the in-place update is race free.  All loads that do not depend on the index
// chain (own row, Adam moments, next entity's offsets) are issued before walking it.
// ---------------------------------------------------------------------------------------
struct AdamArgs {
  float* m_entity; float* v_entity; float* m_bias; float* v_bias; float* m_scal; float* v_scal;
  float b1, b2, eps, step_size, bc2_sqrt;
};

__device__ __forceinline__ float adam_update(float p, float g, float& m, float& v, const AdamArgs& ad) {
  m = m + (g - m) * (1.0f - ad.b1);
  v = v * ad.b2 + ((1.0f - ad.b2) * g) * g;
  const float denom = __fsqrt_rn(v) / ad.bc2_sqrt + ad.eps;
  return p + (-ad.step_size * m) / denom;
}

// Skewed batches (a popular item can own 10^4 of the 10^5 rows): an occurrence list longer than
// VFM_HEAVY_LIST is cut in chunks of that length (work items built with the index), each walked by
// its own lane group here and added -- a few float atomics per chunk -- into the entity's record of a
// small scratch table; the main kernel then reads that record instead of walking the list.  Without
// it one lane group serialises the whole list (Zipf(1.1) items: 3.1 ms instead of 0.2 ms).
template <int LPE, int CPL, int VEC>
__global__ __launch_bounds__(BLOCK) void k_heavy(const int32_t* __restrict__ items, int n_items,
                                                 const int32_t* __restrict__ occ_rows,
                                                 const float* __restrict__ sumz, const float* __restrict__ grow,
                                                 float* __restrict__ heavy_acc, int d) {
  constexpr int GPB = BLOCK / LPE;
  const int lig = threadIdx.x % LPE;
  const int C = (d + VEC - 1) / VEC;
  const int64_t xs = 4 + (((int64_t)d + 3) & ~(int64_t)3);
  for (int it = blockIdx.x * GPB + threadIdx.x / LPE; it < n_items; it += gridDim.x * GPB) {
    const int slot = items[4 * it], beg = items[4 * it + 1], end = items[4 * it + 2];
    Chunk<VEC> A[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
      for (int t = 0; t < VEC; ++t) A[i].v[t] = 0.f;
    float gs = 0.f;
    for (int o = beg; o < end; o += 4) {      // four occurrences in flight
      int r[4]; float g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool ok = o + u < end;
        r[u] = occ_rows[ok ? o + u : beg];
        g[u] = ok ? grow[r[u]] : 0.f;
        gs += g[u];
      }
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int j = lig + i * LPE;
        if (j < C) {
          Chunk<VEC> sv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) sv[u] = ld_chunk<VEC>(sumz + (size_t)r[u] * d + (size_t)j * VEC);
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < VEC; ++t) A[i].v[t] = fmaf(g[u], sv[u].v[t], A[i].v[t]);
        }
      }
    }
    float* rec = heavy_acc + (size_t)slot * xs;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int j = lig + i * LPE;
      if (j < C) {
#pragma unroll
        for (int t = 0; t < VEC; ++t) atomicAdd(rec + 4 + (size_t)j * VEC + t, A[i].v[t]);
      }
    }
    if (lig == 0) { atomicAdd(rec, gs); atomicAdd(rec + 1, (float)(end - beg)); }
  }
}

__device__ __forceinline__ int heavy_slot_of(const int32_t* ids, int n, int e) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (ids[mid] < e) lo = mid + 1; else hi = mid;
  }
  return (n > 0 && ids[lo] == e) ? lo : -1;
}

enum { STAGE_FULL = 0, STAGE_ACC = 1, STAGE_APPLY = 2 };

template <int LPE, int CPL, int VEC, int EPS, int ADAM, int STAGE>
__global__ __launch_bounds__(BLOCK) void k_bwd(const KArgs a, const BwdArgs b, const AdamArgs ad) {
  constexpr int GPB = BLOCK / LPE;
  __shared__ float sh_cs[VFM_MAX_FIELDS];
  __shared__ int64_t sh_hi[VFM_MAX_FIELDS];
  __shared__ double sh_fin[5][BLOCK / 64];
  const int tid = threadIdx.x;
  const int lig = tid % LPE;
  const int d = a.d;
  const int C = (d + VEC - 1) / VEC;
  if (STAGE != STAGE_ACC && tid < a.G) {
    sh_cs[tid] = (float)(a.group_n[tid] / a.W[tid]);
    sh_hi[tid] = a.group_hi[tid];
  }
  __syncthreads();
  const float gout = (ADAM || STAGE == STAGE_ACC) ? 1.0f : b.grad_out[0];

  double fin[5] = {0, 0, 0, 0, 0};
  const bool fold = STAGE == STAGE_FULL && b.loss != nullptr;   // uniform: fold vfm_elbo_finalize_f32 in
  if (blockIdx.x == 0 && fold)
    reduce_slots_and_loss(b.partials, a.scalars, a.ll_scale_d, a.flags, b.loss, sh_fin, fin);
  if (STAGE == STAGE_ACC && blockIdx.x == 0 && tid == 0 && a.e_lo == 0) {
    b.sums[0] = (float)b.partials[VFM_P_G];       // this rank's row sums, to be summed over ranks
    b.sums[1] = (float)b.partials[VFM_P_ALPHA];
  }
  if (STAGE != STAGE_ACC && blockIdx.x == 0 && tid == 0 && a.e_hi == a.T) {   // (last chunk of a chunked run)
    const float alpha = a.scalars[0], m0 = a.scalars[1], s0 = a.scalars[2];
    const float sum_g = (STAGE == STAGE_APPLY) ? b.sums[0] : (float)(fold ? fin[VFM_P_G] : b.partials[VFM_P_G]);
    const float sum_a = (STAGE == STAGE_APPLY) ? b.sums[1]
                                               : (float)(fold ? fin[VFM_P_ALPHA] : b.partials[VFM_P_ALPHA]);
    float e0 = 0.f;
    if constexpr (EPS == EPS_TABLE) e0 = a.eps_global[0];
    if constexpr (EPS == EPS_PHILOX) {
      float n[8], nb;
      normal8b(a.key, 0xFFFFFFFFu, 0u, n, nb);
      e0 = n[0];
    }
    const float as0 = fabsf(s0);
    const float prior = (a.flags & VFM_FLAG_NO_PRIOR_TERMS) ? 0.f : 1.f;
    const float ga = (a.lik == VFM_LIK_NORMAL)
                         ? gout * signf(alpha) * a.ll_scale * sum_a : 0.f;
    const float gm = gout * (sum_g + prior * m0);
    const float gs = gout * signf(s0) * (e0 * sum_g + prior * (as0 - inv_sigma(as0)));
    if constexpr (ADAM) {
      float* sc = const_cast<float*>(a.scalars);
      const float gg[3] = {ga, gm, gs};
      for (int i = 0; i < 3; ++i) {
        float m = ad.m_scal[i], v = ad.v_scal[i];
        // alpha has no gradient under the Bernoulli likelihood (reference: grad None, Adam skips it)
        if (i == 0 && a.lik != VFM_LIK_NORMAL) continue;
        sc[i] = adam_update(sc[i], gg[i], m, v, ad);
        ad.m_scal[i] = m; ad.v_scal[i] = v;
      }
    } else {
      b.g_scalars[0] = ga; b.g_scalars[1] = gm; b.g_scalars[2] = gs;
    }
  }

  const int64_t stride = (int64_t)gridDim.x * GPB;
  const int64_t xs = 4 + (((int64_t)d + 3) & ~(int64_t)3);          // floats per exchange record
  int64_t e = a.e_lo + (int64_t)blockIdx.x * GPB + tid / LPE;
  int2 pq = make_int2(0, 0);
  if (STAGE != STAGE_APPLY && e < a.e_hi) pq = make_int2(b.occ_ptr[e], b.occ_ptr[e + 1]);
  for (; e < a.e_hi; e += stride) {
    int beg = pq.x, end = pq.y;
    const int64_t en = e + stride;
    float2 gc = make_float2(0.f, 0.f);
    if constexpr (STAGE == STAGE_APPLY) {
      gc = *reinterpret_cast<const float2*>(b.acc + (size_t)e * xs);   // (sum of grow, occurrences) over ALL ranks
      beg = 0; end = 0;
    } else {
      if (en < a.e_hi) pq = make_int2(b.occ_ptr[en], b.occ_ptr[en + 1]);   // next entity's offsets, early
    }
    float* prow = const_cast<float*>(a.entity) + (size_t)e * (2 * (size_t)d);
    float* grow_e = (ADAM || STAGE == STAGE_ACC) ? nullptr : b.g_entity + (size_t)e * (2 * (size_t)d);
    const bool touched = (STAGE == STAGE_APPLY) ? gc.y > 0.f : beg != end;
    const float cntf = (STAGE == STAGE_APPLY) ? gc.y : (float)(end - beg);
    if (ADAM == 2 && !touched) continue;   // opt-in row-sparse Adam: rows not in the batch stay as they are

    // loads that do not depend on the index chain
    Chunk<VEC> mu[CPL], s[CPL], ep[CPL], mm[CPL], ms[CPL], vm[CPL], vs[CPL];
    float2 th = make_float2(0.f, 1.f), mb = make_float2(0.f, 0.f), vb = make_float2(0.f, 0.f);
    float io = 0.f, epw = 0.f;
    if (STAGE != STAGE_ACC && (ADAM || touched)) {
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int j = lig + i * LPE;
        if (j < C) {
          mu[i] = ld_chunk<VEC>(prow + (size_t)j * VEC);
          s[i] = ld_chunk<VEC>(prow + d + (size_t)j * VEC);
          if constexpr (ADAM) {
            const size_t o = (size_t)e * (2 * (size_t)d) + (size_t)j * VEC;
            mm[i] = ld_chunk_nt<VEC>(ad.m_entity + o); ms[i] = ld_chunk_nt<VEC>(ad.m_entity + o + d);
            vm[i] = ld_chunk_nt<VEC>(ad.v_entity + o); vs[i] = ld_chunk_nt<VEC>(ad.v_entity + o + d);
          }
          if constexpr (EPS == EPS_TABLE)
            if (touched) ep[i] = ld_chunk<VEC>(a.eps_entity + (size_t)e * d + (size_t)j * VEC);
        }
      }
      if (lig == 0) {
        th = *reinterpret_cast<const float2*>(a.bias + 2 * (size_t)e);
        if constexpr (ADAM) {
          mb = *reinterpret_cast<const float2*>(ad.m_bias + 2 * (size_t)e);
          vb = *reinterpret_cast<const float2*>(ad.v_bias + 2 * (size_t)e);
        }
      }
      if (touched) {
        io = a.inv_occ[e];
        if constexpr (EPS == EPS_TABLE) epw = a.eps_bias[e];
      }
    }

    // walk the inverted index: A = sum_r g_r * sumz_r, gs = sum_r g_r
    Chunk<VEC> A[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
      for (int t = 0; t < VEC; ++t) A[i].v[t] = 0.f;
    float gs = 0.f;
    int o = beg;
    if (STAGE != STAGE_APPLY && end - beg > VFM_HEAVY_LIST && b.n_heavy > 0) {
      const int slot = heavy_slot_of(b.heavy_ids, b.n_heavy, (int)e);
      if (slot >= 0) {      // pre-reduced by k_heavy: read the record, skip the walk
        const float* rec = b.heavy_acc + (size_t)slot * xs;
        gs = rec[0];
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          const int j = lig + i * LPE;
          if (j < C) A[i] = ld_chunk<VEC>(rec + 4 + (size_t)j * VEC);
        }
        o = end;
      }
    }
    for (; o + 1 < end; o += 2) {       // two occurrences in flight
      const int r0 = b.occ_rows[o], r1 = b.occ_rows[o + 1];
      const float g0 = b.grow[r0], g1 = b.grow[r1];
      gs += g0 + g1;
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int j = lig + i * LPE;
        if (j < C) {
          const Chunk<VEC> s0v = ld_chunk<VEC>(b.sumz + (size_t)r0 * d + (size_t)j * VEC);
          const Chunk<VEC> s1v = ld_chunk<VEC>(b.sumz + (size_t)r1 * d + (size_t)j * VEC);
#pragma unroll
          for (int t = 0; t < VEC; ++t) A[i].v[t] = fmaf(g1, s1v.v[t], fmaf(g0, s0v.v[t], A[i].v[t]));
        }
      }
    }
    if (o < end) {
      const int r0 = b.occ_rows[o];
      const float g0 = b.grow[r0];
      gs += g0;
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int j = lig + i * LPE;
        if (j < C) {
          const Chunk<VEC> s0v = ld_chunk<VEC>(b.sumz + (size_t)r0 * d + (size_t)j * VEC);
#pragma unroll
          for (int t = 0; t < VEC; ++t) A[i].v[t] = fmaf(g0, s0v.v[t], A[i].v[t]);
        }
      }
    }

    if constexpr (STAGE == STAGE_ACC) {   // store the statistics (dense: zeros for rows not in this shard)
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int j = lig + i * LPE;
        if (j < C) st_chunk<VEC>(b.acc + (size_t)e * xs + 4 + (size_t)j * VEC, A[i]);
      }
      if (lig == 0) *reinterpret_cast<float4*>(b.acc + (size_t)e * xs) = make_float4(gs, cntf, 0.f, 0.f);
      continue;
    }
    if constexpr (STAGE == STAGE_APPLY) {
      gs = gc.x;
      if (touched) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          const int j = lig + i * LPE;
          if (j < C) A[i] = ld_chunk<VEC>(b.acc + (size_t)e * xs + 4 + (size_t)j * VEC);
        }
      }
    }

    if (!touched && !ADAM) {   // entity not in the batch: dense zero row
      Chunk<VEC> zc;
#pragma unroll
      for (int t = 0; t < VEC; ++t) zc.v[t] = 0.f;
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int j = lig + i * LPE;
        if (j < C) {
          st_chunk_nt<VEC>(grow_e + (size_t)j * VEC, zc);
          st_chunk_nt<VEC>(grow_e + d + (size_t)j * VEC, zc);
        }
      }
      if (lig == 0) *reinterpret_cast<float2*>(b.g_bias + 2 * (size_t)e) = make_float2(0.f, 0.f);
      continue;
    }

    float c = 0.f;
    if (touched) {
      c = sh_cs[group_index(sh_hi, a.G, e)] * io * cntf;
    }
    float nb_eps = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int j = lig + i * LPE;
      if (j < C) {
        Chunk<VEC> gm, gv;
        if (touched) {
          Chunk<VEC> epc;
          if constexpr (EPS == EPS_TABLE) {
            epc = ep[i];
          } else if constexpr (EPS == EPS_ZERO) {
#pragma unroll
            for (int t = 0; t < VEC; ++t) epc.v[t] = 0.f;
          } else {
            float nb;
            eps_of_chunk<VEC>(a.key, (uint32_t)e, j, epc.v, nb);
            if (i == 0) nb_eps = nb;
          }
#pragma unroll
          for (int t = 0; t < VEC; ++t) {
            const float sg = fabsf(s[i].v[t]);
            const float z = fmaf(sg, epc.v[t], mu[i].v[t]);
            const float gz = A[i].v[t] - z * gs;  // sum_r g_r (sumz_rk - z_ek)
            gm.v[t] = gout * (gz + c * mu[i].v[t]);
            gv.v[t] = gout * signf(s[i].v[t]) * (gz * epc.v[t] + c * (sg - inv_sigma(sg)));
          }
        } else {
#pragma unroll
          for (int t = 0; t < VEC; ++t) { gm.v[t] = 0.f; gv.v[t] = 0.f; }
        }
        if constexpr (ADAM) {
          Chunk<VEC> pm, ps;
#pragma unroll
          for (int t = 0; t < VEC; ++t) {
            pm.v[t] = adam_update(mu[i].v[t], gm.v[t], mm[i].v[t], vm[i].v[t], ad);
            ps.v[t] = adam_update(s[i].v[t], gv.v[t], ms[i].v[t], vs[i].v[t], ad);
          }
          const size_t o2 = (size_t)e * (2 * (size_t)d) + (size_t)j * VEC;
          st_chunk<VEC>(prow + (size_t)j * VEC, pm);
          st_chunk<VEC>(prow + d + (size_t)j * VEC, ps);
          st_chunk_nt<VEC>(ad.m_entity + o2, mm[i]); st_chunk_nt<VEC>(ad.m_entity + o2 + d, ms[i]);
          st_chunk_nt<VEC>(ad.v_entity + o2, vm[i]); st_chunk_nt<VEC>(ad.v_entity + o2 + d, vs[i]);
        } else {
          st_chunk_nt<VEC>(grow_e + (size_t)j * VEC, gm);
          st_chunk_nt<VEC>(grow_e + d + (size_t)j * VEC, gv);
        }
      }
    }
    if (lig == 0) {
      float g0 = 0.f, g1 = 0.f;
      if (touched) {
        if constexpr (EPS == EPS_TABLE) nb_eps = epw;
        const float sg = fabsf(th.y);
        g0 = gout * (gs + c * th.x);
        g1 = gout * signf(th.y) * (gs * nb_eps + c * (sg - inv_sigma(sg)));
      }
      if constexpr (ADAM) {
        float2 pn;
        pn.x = adam_update(th.x, g0, mb.x, vb.x, ad);
        pn.y = adam_update(th.y, g1, mb.y, vb.y, ad);
        *reinterpret_cast<float2*>(const_cast<float*>(a.bias) + 2 * (size_t)e) = pn;
        *reinterpret_cast<float2*>(ad.m_bias + 2 * (size_t)e) = mb;
        *reinterpret_cast<float2*>(ad.v_bias + 2 * (size_t)e) = vb;
      } else {
        *reinterpret_cast<float2*>(b.g_bias + 2 * (size_t)e) = make_float2(g0, g1);
      }
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

// eps dump (tests)
__global__ void k_philox_dump(const KArgs a, float* eps_entity, float* eps_bias, float* eps_global) {
  const int64_t n8 = ((int64_t)a.d + 7) / 8;
  const int64_t total = a.T * n8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i / n8;
    const int p = (int)(i % n8);
    float n[8], nb;
    normal8b(a.key, (uint32_t)e, (uint32_t)p, n, nb);
    for (int t = 0; t < 8; ++t)
      if (p * 8 + t < a.d) eps_entity[e * a.d + p * 8 + t] = n[t];
    if (p == 0) eps_bias[e] = nb;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float n[8], nb;
    normal8b(a.key, 0xFFFFFFFFu, 0u, n, nb);
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
  if (p->e_lo < 0 || p->e_lo > p->T || (p->e_hi != 0 && p->e_hi < p->e_lo)) return fail(VFM_E_INVALID, "bad entity range");
  Shape s;
  if (!pick_shape(p->d, &s)) return fail(VFM_E_UNSUPPORTED, "embedding size d not supported (d%4==0: d<=1024, else d<=256)");
  return 0;
}

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

// eps source of a call: VFM_FLAG_EPS_ZERO > tables > Philox
int eps_mode(const vfm_problem_t* p, const float* ee, const float* eb, const float* eg, int* mode) {
  const int neps = (ee != nullptr) + (eb != nullptr) + (eg != nullptr);
  if (neps != 0 && neps != 3) return fail(VFM_E_INVALID, "give all three eps tables or none");
  *mode = (p->flags & VFM_FLAG_EPS_ZERO) ? EPS_ZERO : (neps == 3 ? EPS_TABLE : EPS_PHILOX);
  return 0;
}

KArgs make_args(const vfm_problem_t* p, const void* x, const float* y, const float* entity,
                const float* bias, const float* inv_occ, const float* scalars, const double* W,
                const float* ee, const float* eb, const float* eg) {
  KArgs a;
  memset(&a, 0, sizeof(a));
  a.B = p->B; a.T = p->T; a.F = p->F; a.d = p->d; a.lik = p->likelihood;
  a.e_lo = p->e_lo; a.e_hi = (p->e_hi > 0 && p->e_hi < p->T) ? p->e_hi : p->T;
  a.id64 = p->id_bits == 64; a.G = p->F; a.flags = p->flags;
  a.ll_scale_d = (double)p->nb_train / (double)(p->B_global > 0 ? p->B_global : 1);
  a.ll_scale = (float)a.ll_scale_d;
  a.key.seed_lo = (uint32_t)p->seed; a.key.seed_hi = (uint32_t)(p->seed >> 32);
  a.key.step_lo = (uint32_t)p->step; a.key.step_hi = (uint32_t)(p->step >> 32);
  a.x = x; a.y = y; a.entity = entity; a.bias = bias; a.inv_occ = inv_occ; a.scalars = scalars;
  a.W = W; a.eps_entity = ee; a.eps_bias = eb; a.eps_global = eg;
  for (int g = 0; g < p->F; ++g) { a.group_hi[g] = p->group_hi[g]; a.group_n[g] = p->group_n[g]; }
  return a;
}

// ---- forward dispatch: shape x eps source x mode x (F == 2 ?) ----
template <int LPE, int CPL, int VEC, int EPS, int MODE, int FF, bool ID64>
int launch_fwd_t(KArgs& a, const FwdOut& o, hipStream_t st) {
  constexpr int GPB = BLOCK / LPE;
  // persistent-ish grid: enough groups that each owns a few rows (pipelined), capped at
  // VFM_FWD_BLOCKS_PER_CU resident workgroups on each of the 256 CUs
  const int per_cu = env_int("VFM_FWD_BLOCKS_PER_CU", 4);
  int64_t nb = (a.B + GPB - 1) / GPB;
  int64_t cap = 256LL * per_cu;
  if (cap > VFM_MAX_FWD_BLOCKS) cap = VFM_MAX_FWD_BLOCKS;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL((k_fwd<LPE, CPL, VEC, EPS, MODE, FF, ID64>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, o);
  return 0;
}

template <int LPE, int CPL, int VEC>
int launch_fwd_s(int eps, int mode, int ff, KArgs& a, const FwdOut& o, hipStream_t st) {
#define FWD(E_, M_)                                                              \
  if (eps == E_ && mode == M_) {                                                 \
    if constexpr (VEC == 4) {                                                    \
      if constexpr (LPE >= 2) {                                                              \
        if (ff == 2 && a.id64) return launch_fwd_t<LPE, CPL, VEC, E_, M_, 2, true>(a, o, st);  \
        if (ff == 2) return launch_fwd_t<LPE, CPL, VEC, E_, M_, 2, false>(a, o, st);           \
      }                                                                                      \
    }                                                                            \
    return launch_fwd_t<LPE, CPL, VEC, E_, M_, 0, true>(a, o, st);               \
  }
  FWD(EPS_PHILOX, MODE_TRAIN) FWD(EPS_TABLE, MODE_TRAIN)
  FWD(EPS_PHILOX, MODE_PREDICT) FWD(EPS_TABLE, MODE_PREDICT) FWD(EPS_ZERO, MODE_PREDICT)
#undef FWD
  return fail(VFM_E_UNSUPPORTED, "forward: unsupported eps source / mode combination");
}

template <int LPE, int CPL, int VEC, int EPS, int ADAM, int STAGE = STAGE_FULL>
int launch_bwd_t(KArgs& a, const BwdArgs& b, const AdamArgs& ad, hipStream_t st) {
  constexpr int GPB = BLOCK / LPE;
  const int per_cu = env_int("VFM_BWD_BLOCKS_PER_CU", 8);
  int64_t nb = (a.e_hi - a.e_lo + GPB - 1) / GPB;
  const int64_t cap = 256LL * per_cu;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL((k_bwd<LPE, CPL, VEC, EPS, ADAM, STAGE>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, b, ad);
  return 0;
}

template <int LPE, int CPL, int VEC>
int launch_bwd_s(int eps, int adam, KArgs& a, const BwdArgs& b, const AdamArgs& ad, hipStream_t st) {
  if (eps == EPS_PHILOX && adam == 0) return launch_bwd_t<LPE, CPL, VEC, EPS_PHILOX, 0>(a, b, ad, st);
  if (eps == EPS_TABLE && adam == 0) return launch_bwd_t<LPE, CPL, VEC, EPS_TABLE, 0>(a, b, ad, st);
  if (eps == EPS_PHILOX && adam == 1) return launch_bwd_t<LPE, CPL, VEC, EPS_PHILOX, 1>(a, b, ad, st);
  if (eps == EPS_TABLE && adam == 1) return launch_bwd_t<LPE, CPL, VEC, EPS_TABLE, 1>(a, b, ad, st);
  if (eps == EPS_PHILOX && adam == 2) return launch_bwd_t<LPE, CPL, VEC, EPS_PHILOX, 2>(a, b, ad, st);
  if (eps == EPS_TABLE && adam == 2) return launch_bwd_t<LPE, CPL, VEC, EPS_TABLE, 2>(a, b, ad, st);
  if (adam == 10) return launch_bwd_t<LPE, CPL, VEC, EPS_ZERO, 0, STAGE_ACC>(a, b, ad, st);
  if (eps == EPS_PHILOX && adam == 11) return launch_bwd_t<LPE, CPL, VEC, EPS_PHILOX, 1, STAGE_APPLY>(a, b, ad, st);
  if (eps == EPS_TABLE && adam == 11) return launch_bwd_t<LPE, CPL, VEC, EPS_TABLE, 1, STAGE_APPLY>(a, b, ad, st);
  return fail(VFM_E_UNSUPPORTED, "backward: unsupported eps source");
}

#define FOR_SHAPES(X)                                                                          \
  X(1, 1, 4) X(2, 1, 4) X(4, 1, 4) X(8, 1, 4) X(16, 1, 4) X(32, 1, 4) X(64, 1, 4) X(64, 2, 4) \
  X(64, 4, 4) X(8, 1, 1) X(64, 1, 1) X(64, 4, 1)

int dispatch_fwd(const Shape& s, int eps, int mode, int ff, KArgs& a, const FwdOut& o, hipStream_t st) {
#define X(L_, C_, V_) \
  if (s.lpe == L_ && s.cpl == C_ && s.vec == V_) return launch_fwd_s<L_, C_, V_>(eps, mode, ff, a, o, st);
  FOR_SHAPES(X)
#undef X
  return fail(VFM_E_UNSUPPORTED, "no kernel instance for this embedding size");
}

int dispatch_bwd(const Shape& s, int eps, int adam, KArgs& a, const BwdArgs& b, const AdamArgs& ad,
                 hipStream_t st) {
#define X(L_, C_, V_) \
  if (s.lpe == L_ && s.cpl == C_ && s.vec == V_) return launch_bwd_s<L_, C_, V_>(eps, adam, a, b, ad, st);
  FOR_SHAPES(X)
#undef X
  return fail(VFM_E_UNSUPPORTED, "no kernel instance for this embedding size");
}

template <int LPE, int CPL, int VEC>
int launch_heavy_t(const vfm_index_t* idx, const float* sumz, const float* grow, int d, hipStream_t st) {
  constexpr int GPB = BLOCK / LPE;
  int64_t nb = ((int64_t)idx->n_items + GPB - 1) / GPB;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL((k_heavy<LPE, CPL, VEC>), dim3((unsigned)nb), dim3(BLOCK), 0, st, idx->heavy_items,
                     (int)idx->n_items, idx->occ_rows, sumz, grow, idx->heavy_acc, d);
  return 0;
}

// pre-reduce the long occurrence lists (if the index has any) and point the main kernel at the result
int run_heavy(const vfm_problem_t* p, const vfm_index_t* idx, const float* sumz, const float* grow,
              hipStream_t st, BwdArgs* b) {
  b->heavy_ids = nullptr; b->heavy_acc = nullptr; b->n_heavy = 0;
  if (idx->n_heavy <= 0 || idx->n_items <= 0) return 0;
  if (!idx->heavy_ids || !idx->heavy_items || !idx->heavy_acc)
    return fail(VFM_E_INVALID, "index: heavy_ids / heavy_items / heavy_acc missing");
  const size_t xs = 4 + (((size_t)p->d + 3) & ~(size_t)3);
  hipError_t e = hipMemsetAsync(idx->heavy_acc, 0, sizeof(float) * xs * (size_t)idx->n_heavy, st);
  if (e != hipSuccess) return fail_hip(e, "heavy_acc memset");
  Shape s;
  pick_shape(p->d, &s);
#define X(L_, C_, V_) \
  if (s.lpe == L_ && s.cpl == C_ && s.vec == V_) launch_heavy_t<L_, C_, V_>(idx, sumz, grow, p->d, st);
  FOR_SHAPES(X)
#undef X
  b->heavy_ids = idx->heavy_ids; b->heavy_acc = idx->heavy_acc; b->n_heavy = idx->n_heavy;
  return 0;
}

int check_index(const vfm_problem_t* p, const vfm_index_t* idx, const char* who) {
  if (!idx || !idx->occ_ptr || (p->B > 0 && !idx->occ_rows)) {
    snprintf(g_err, sizeof(g_err), "%s: inverted index missing", who);
    return VFM_E_INVALID;
  }
  return 0;
}

int after_launch(const char* where) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, where);
  return 0;
}

void adam_consts(float lr, float beta1, float beta2, int64_t step, float* step_size, float* bc2_sqrt) {
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  *step_size = (float)((double)lr / bc1);
  *bc2_sqrt = (float)sqrt(bc2);
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
  if (!W || (p->B > 0 && (!x || !inv_occ))) return fail(VFM_E_INVALID, "vfm_batch_norms: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_zero_f64, dim3(1), dim3(64), 0, st, W, (int)p->F);
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
  if (!partials) return fail(VFM_E_INVALID, "vfm_elbo_fwd_f32: partials is NULL");
  hipStream_t st0 = (hipStream_t)stream;
  if (p->B == 0) {  // empty shard (a rank without rows): zero sums, zero blocks; buffers may be NULL
    hipLaunchKernelGGL(k_zero_f64, dim3(1), dim3(64), 0, st0, partials, (int)VFM_N_PARTIALS);
    return after_launch("vfm_elbo_fwd_f32");
  }
  if (!x || !entity_params || !bias_params || !scalars || !pred)
    return fail(VFM_E_INVALID, "vfm_elbo_fwd_f32: NULL pointer");
  const bool train = y != nullptr;
  if (train && (!inv_occ || !W || !sumz || !grow))
    return fail(VFM_E_INVALID, "vfm_elbo_fwd_f32: y given, so inv_occ, W, sumz and grow are required");
  if (!train && (sumz || grow)) return fail(VFM_E_INVALID, "vfm_elbo_fwd_f32: sumz / grow need y");
  int eps;
  if (int rc = eps_mode(p, eps_entity, eps_bias, eps_global, &eps)) return rc;
  if (train && eps == EPS_ZERO) return fail(VFM_E_UNSUPPORTED, "vfm_elbo_fwd_f32: VFM_FLAG_EPS_ZERO is prediction-only");
  hipStream_t st = (hipStream_t)stream;
  KArgs a = make_args(p, x, y, entity_params, bias_params, inv_occ, scalars, W, eps_entity, eps_bias,
                      eps_global);
  FwdOut o{pred, partials, sumz, grow};
  Shape s;
  pick_shape(p->d, &s);
  if (int rc = dispatch_fwd(s, eps, train ? MODE_TRAIN : MODE_PREDICT, p->F == 2 ? 2 : 0, a, o, st)) return rc;
  return after_launch("vfm_elbo_fwd_f32");
}

int vfm_elbo_finalize_f32(const vfm_problem_t* p, double* partials, const float* scalars,
                          float* loss, void* stream) {
  if (int rc = check_problem(p)) return rc;
  if (!partials || !scalars || !loss) return fail(VFM_E_INVALID, "vfm_elbo_finalize_f32: NULL pointer");
  const double ll_scale = (double)p->nb_train / (double)(p->B_global > 0 ? p->B_global : 1);
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, (hipStream_t)stream, partials, scalars, ll_scale,
                     (int)p->flags, loss);
  return after_launch("vfm_elbo_finalize_f32");
}

int vfm_elbo_bwd_f32(const vfm_problem_t* p, const vfm_index_t* idx,
                     const float* entity_params, const float* bias_params,
                     const float* inv_occ, const float* scalars, const double* W,
                     const float* eps_entity, const float* eps_bias, const float* eps_global,
                     const float* sumz, const float* grow, const double* partials,
                     const float* grad_out, float* g_entity, float* g_bias, float* g_scalars,
                     void* stream) {
  if (int rc = check_problem(p)) return rc;
  if (int rc = check_index(p, idx, "vfm_elbo_bwd_f32")) return rc;
  if (!entity_params || !bias_params || !inv_occ || !scalars || !W || !partials || !grad_out ||
      !g_entity || !g_bias || !g_scalars || (p->B > 0 && (!sumz || !grow)))
    return fail(VFM_E_INVALID, "vfm_elbo_bwd_f32: NULL pointer");
  int eps;
  if (int rc = eps_mode(p, eps_entity, eps_bias, eps_global, &eps)) return rc;
  if (eps == EPS_ZERO) return fail(VFM_E_UNSUPPORTED, "vfm_elbo_bwd_f32: VFM_FLAG_EPS_ZERO is prediction-only");
  KArgs a = make_args(p, nullptr, nullptr, entity_params, bias_params, inv_occ, scalars, W, eps_entity,
                      eps_bias, eps_global);
  BwdArgs b{idx->occ_ptr, idx->occ_rows, sumz, grow, const_cast<double*>(partials), grad_out, g_entity, g_bias,
            g_scalars, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  if (int rc = run_heavy(p, idx, sumz, grow, (hipStream_t)stream, &b)) return rc;
  AdamArgs ad;
  memset(&ad, 0, sizeof(ad));
  Shape s;
  pick_shape(p->d, &s);
  if (int rc = dispatch_bwd(s, eps, 0, a, b, ad, (hipStream_t)stream)) return rc;
  return after_launch("vfm_elbo_bwd_f32");
}

int vfm_elbo_bwd_adam_f32(const vfm_problem_t* p, const vfm_index_t* idx,
                          float* entity_params, float* bias_params, float* scalars,
                          const float* inv_occ, const double* W,
                          const float* eps_entity, const float* eps_bias, const float* eps_global,
                          const float* sumz, const float* grow, double* partials,
                          float* m_entity, float* v_entity, float* m_bias, float* v_bias,
                          float* m_scalars, float* v_scalars,
                          float lr, float beta1, float beta2, float eps_adam, int64_t step, float* loss,
                          void* stream) {
  if (int rc = check_problem(p)) return rc;
  if (int rc = check_index(p, idx, "vfm_elbo_bwd_adam_f32")) return rc;
  if (!entity_params || !bias_params || !inv_occ || !scalars || !W || !partials || !m_entity ||
      !v_entity || !m_bias || !v_bias || !m_scalars || !v_scalars || step < 1 ||
      (p->B > 0 && (!sumz || !grow)))
    return fail(VFM_E_INVALID, "vfm_elbo_bwd_adam_f32: bad argument");
  if (p->flags & VFM_FLAG_NO_PRIOR_TERMS)
    return fail(VFM_E_INVALID, "vfm_elbo_bwd_adam_f32: single-rank only (gradients never leave the kernel)");
  int eps;
  if (int rc = eps_mode(p, eps_entity, eps_bias, eps_global, &eps)) return rc;
  if (eps == EPS_ZERO) return fail(VFM_E_UNSUPPORTED, "vfm_elbo_bwd_adam_f32: VFM_FLAG_EPS_ZERO is prediction-only");
  KArgs a = make_args(p, nullptr, nullptr, entity_params, bias_params, inv_occ, scalars, W, eps_entity,
                      eps_bias, eps_global);
  BwdArgs b{idx->occ_ptr, idx->occ_rows, sumz, grow, partials, nullptr, nullptr, nullptr, nullptr, loss, nullptr,
            nullptr, nullptr, nullptr, 0};
  if (int rc = run_heavy(p, idx, sumz, grow, (hipStream_t)stream, &b)) return rc;
  AdamArgs ad{m_entity, v_entity, m_bias, v_bias, m_scalars, v_scalars, beta1, beta2, eps_adam, 0.f, 0.f};
  adam_consts(lr, beta1, beta2, step, &ad.step_size, &ad.bc2_sqrt);
  Shape s;
  pick_shape(p->d, &s);
  if (int rc = dispatch_bwd(s, eps, (p->flags & VFM_FLAG_SPARSE_ADAM) ? 2 : 1, a, b, ad, (hipStream_t)stream))
    return rc;
  return after_launch("vfm_elbo_bwd_adam_f32");
}

int vfm_elbo_bwd_acc_f32(const vfm_problem_t* p, const vfm_index_t* idx,
                         const float* sumz, const float* grow, const double* partials, float* acc,
                         float* sums, void* stream) {
  if (int rc = check_problem(p)) return rc;
  if (int rc = check_index(p, idx, "vfm_elbo_bwd_acc_f32")) return rc;
  if (!partials || !acc || !sums || (p->B > 0 && (!sumz || !grow)))
    return fail(VFM_E_INVALID, "vfm_elbo_bwd_acc_f32: NULL pointer");
  KArgs a = make_args(p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  BwdArgs b{idx->occ_ptr, idx->occ_rows, sumz, grow, const_cast<double*>(partials), nullptr, nullptr, nullptr,
            nullptr, nullptr, acc, sums, nullptr, nullptr, 0};
  // (the pre-reduction covers whole lists, so with several entity chunks it runs with the first one)
  if (p->e_lo == 0)
    if (int rc = run_heavy(p, idx, sumz, grow, (hipStream_t)stream, &b)) return rc;
  if (p->e_lo != 0 && idx->n_heavy > 0) { b.heavy_ids = idx->heavy_ids; b.heavy_acc = idx->heavy_acc; b.n_heavy = idx->n_heavy; }
  AdamArgs ad;
  memset(&ad, 0, sizeof(ad));
  Shape s;
  pick_shape(p->d, &s);
  if (int rc = dispatch_bwd(s, EPS_ZERO, 10, a, b, ad, (hipStream_t)stream)) return rc;
  return after_launch("vfm_elbo_bwd_acc_f32");
}

int vfm_elbo_apply_adam_f32(const vfm_problem_t* p, const float* acc, const float* sums,
                            float* entity_params, float* bias_params, float* scalars, const float* inv_occ,
                            const double* W, const float* eps_entity, const float* eps_bias,
                            const float* eps_global, float* m_entity, float* v_entity, float* m_bias,
                            float* v_bias, float* m_scalars, float* v_scalars, float lr, float beta1,
                            float beta2, float eps_adam, int64_t step, void* stream) {
  if (int rc = check_problem(p)) return rc;
  if (!acc || !sums || !entity_params || !bias_params || !scalars || !inv_occ || !W || !m_entity ||
      !v_entity || !m_bias || !v_bias || !m_scalars || !v_scalars || step < 1)
    return fail(VFM_E_INVALID, "vfm_elbo_apply_adam_f32: bad argument");
  int eps;
  if (int rc = eps_mode(p, eps_entity, eps_bias, eps_global, &eps)) return rc;
  if (eps == EPS_ZERO) return fail(VFM_E_UNSUPPORTED, "vfm_elbo_apply_adam_f32: VFM_FLAG_EPS_ZERO is prediction-only");
  KArgs a = make_args(p, nullptr, nullptr, entity_params, bias_params, inv_occ, scalars, W, eps_entity, eps_bias,
                      eps_global);
  BwdArgs b{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
            const_cast<float*>(acc), const_cast<float*>(sums), nullptr, nullptr, 0};
  AdamArgs ad{m_entity, v_entity, m_bias, v_bias, m_scalars, v_scalars, beta1, beta2, eps_adam, 0.f, 0.f};
  adam_consts(lr, beta1, beta2, step, &ad.step_size, &ad.bc2_sqrt);
  Shape s;
  pick_shape(p->d, &s);
  if (int rc = dispatch_bwd(s, eps, 11, a, b, ad, (hipStream_t)stream)) return rc;
  return after_launch("vfm_elbo_apply_adam_f32");
}

int vfm_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                 float beta2, float eps, int64_t step, void* stream) {
  if (!p || !g || !m || !v || n < 0 || step < 1) return fail(VFM_E_INVALID, "vfm_adam_f32: bad argument");
  if (n == 0) return 0;
  if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) != 0)
    return fail(VFM_E_INVALID, "vfm_adam_f32: pointers must be 16-byte aligned");
  float step_size, bc2_sqrt;
  adam_consts(lr, beta1, beta2, step, &step_size, &bc2_sqrt);
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
