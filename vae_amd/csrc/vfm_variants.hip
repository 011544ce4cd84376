// vfm_variants.hip -- the ELBO variants of SURVEY 8(f)4 that the fused kernels do not cover, as ONE general
// (any number of fields, any d <= 1024) forward / backward pair:
//
//   objective  VFM_OBJ_SAMPLED      the sampled ELBO of vfm-torch.py:189-324,359 (z = mu + sigma*eps)
//              VFM_OBJ_CLOSED_FORM  the closed-form EXPECTED log-likelihood of vfm-tomasrch.py:369-451 -- no
//                                   sampling: per row  y_bar = m0 + sum_f mu_w + sum_{f<g} <mu_f, mu_g>,
//                                   T = s0^2 + sum_f s_w^2 + sum_{f<g} sum_k (mu_f^2 s_g^2 + mu_g^2 s_f^2 + s_f^2 s_g^2),
//                                   row term  1/2 log|alpha| - |alpha|/2 ((y - y_bar)^2 + T)   (:446-449);
//                                   loss = -nb_train/B * sum + KL terms (:569-588)
//   priors     NULL: N(0,1) (vfm-torch.py:162-164); else LEARNABLE GROUP PRIORS (vfm-tomasrch.py:206-290): one
//              Normal(mean, |scale|) per id group for the first-order weights and one per group and coordinate
//              for the embeddings (+ one for the global bias); their gradients come out of the backward
//   values     NULL: every feature value is 1 (entity ids); else x carries a VALUE per (row, field) --
//              sparse features with values != 1 (vfm.py:483-509): pred = w0 + sum_f v_f w_f +
//              1/2 sum_k [(sum_f v_f z_fk)^2 - sum_f v_f^2 z_fk^2]
//
// Same memory-access pattern as the fused kernels (row-parallel gather forward, entity-centric backward over
// the inverted index, dense gradient rows), a different epilogue.  Two implementations of the same arithmetic:
// vfm_variants8.hpp (d % 8 == 0: lane groups, 8 coordinates per lane, pipelined gathers -- what runs at the
// usual embedding sizes) and the scalar pair in this file (any d, eps tables; one wave per row / per entity,
// scalar loads, occurrence lists walked serially; also the cross-check of the first: VFM_VARIANT_SCALAR=1).
// gfx950 only, wave = 64.
#include <math.h>
#include <string.h>

#include "vfm_args.hpp"

namespace vfm {
namespace {

#include "vfm_rng.hpp"
#include "vfm_common.hpp"

constexpr int MAXKB = 16;      // d <= 64 * MAXKB

struct VarArgs {
  int64_t B, T;
  int32_t F, d, G, id64, lik, objective, eps_mode;   // eps_mode: EPS_PHILOX / EPS_TABLE (sampled objective)
  float ll_scale;                                    // nb_train / B_global
  RngKey key;
  const void* x;
  const float* xv;        // [B,F] feature values or NULL
  const float* y;
  const float* entity;
  const float* bias;
  const float* inv_occ;
  const float* scalars;
  const double* W;
  const float* priors;    // [2 | G | G | G*d | G*d] or NULL
  const float* eps_entity;
  const float* eps_bias;
  const float* eps_global;
  int64_t group_hi[VFM_MAX_FIELDS];
  double group_n[VFM_MAX_FIELDS];
  int32_t* status;        // vfm_index_t.status: the backward kernels count the index entries they had to clamp (may be NULL)
  int32_t n_occ;          // B * F: the bound of every list offset
};

// a corrupted index is clamped, never followed (list offsets to [0, n_occ], row numbers to [0, B)); `nclamp` counts
__device__ __forceinline__ void span_ok(const VarArgs& a, int& beg, int& end, int& nclamp) {
  if (beg < 0 || end < beg || end > a.n_occ) { beg = end = 0; ++nclamp; }
}
__device__ __forceinline__ int row_ok(const VarArgs& a, int r, int& nclamp) {
  if ((unsigned)r >= (unsigned)a.B) { r = 0; ++nclamp; }
  return r;
}

// prior of the global bias / a group's first-order weights / a group's embedding coordinate: (mean, sigma)
__device__ __forceinline__ float2 prior0(const VarArgs& a) {
  return a.priors ? make_float2(a.priors[0], fmaxf(fabsf(a.priors[1]), SIGMA_MIN)) : make_float2(0.f, 1.f);
}
__device__ __forceinline__ float2 prior_w(const VarArgs& a, int g) {
  return a.priors ? make_float2(a.priors[2 + g], fmaxf(fabsf(a.priors[2 + a.G + g]), SIGMA_MIN)) : make_float2(0.f, 1.f);
}
__device__ __forceinline__ float2 prior_v(const VarArgs& a, int g, int k) {
  if (!a.priors) return make_float2(0.f, 1.f);
  const float* pm = a.priors + 2 + 2 * a.G;
  return make_float2(pm[(size_t)g * a.d + k], fmaxf(fabsf(pm[(size_t)a.G * a.d + (size_t)g * a.d + k]), SIGMA_MIN));
}
// KL(N(mu, sg) || N(mp, sp)) = log(sp / sg) + (sg^2 + (mu - mp)^2) / (2 sp^2) - 1/2   (torch kl.py _kl_normal_normal)
__device__ __forceinline__ float kl_normal(float mu, float sg, float2 pr) {
  const float sgc = fmaxf(sg, SIGMA_MIN);
  const float dm = mu - pr.x;
  return logf(pr.y / sgc) + (sg * sg + dm * dm) / (2.0f * pr.y * pr.y) - 0.5f;
}

__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }

__device__ __forceinline__ int64_t load_id(const VarArgs& a, int64_t pos, float& bad) {
  const int64_t id = a.id64 ? ((const int64_t*)a.x)[pos] : (int64_t)((const int32_t*)a.x)[pos];
  const bool ok = id >= 0 && id < a.T;
  if (!ok) bad += 1.f;
  return ok ? id : 0;
}

// eps of (entity e, coordinate k) / of e's first-order weight, sampled objective
__device__ __forceinline__ float eps_v(const VarArgs& a, int64_t e, int k) {
  if (a.eps_mode == EPS_TABLE) return a.eps_entity[(size_t)e * a.d + k];
  float n[8], nb;
  normal8b(a.key, (uint32_t)e, ((uint32_t)k >> 3) + (a.key.chunk_off >> 1), n, nb);
  float v = n[0];
#pragma unroll
  for (int t = 1; t < 8; ++t) v = ((k & 7) == t) ? n[t] : v;
  return v;
}
__device__ __forceinline__ float eps_w(const VarArgs& a, int64_t e) {
  if (a.eps_mode == EPS_TABLE) return a.eps_bias[e];
  float n[8], nb;
  normal8b(a.key, (uint32_t)e, 0u, n, nb);
  return nb;
}
__device__ __forceinline__ float eps_0(const VarArgs& a) {
  if (a.eps_mode == EPS_TABLE) return a.eps_global[0];
  float n[8], nb;
  normal8b(a.key, 0xFFFFFFFFu, 0u, n, nb);
  return n[0];
}

// ---------------------------------------------------------------------------------------
// forward: one wave per row.  state [B, NS*d]: sampled: (sum_f v_f z_f);  closed form: (sum_f v_f mu_f |
// sum_f v_f^2 s_f^2 | sum_f v_f^2 (mu_f^2 + s_f^2)) -- what the backward needs of a row.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_var_fwd(const VarArgs a, float* __restrict__ pred, double* __restrict__ partials,
                                                   float* __restrict__ state, float* __restrict__ grow) {
  __shared__ float sh_red[6 * 4];
  const int lane = threadIdx.x & 63;
  const bool cf = a.objective == VFM_OBJ_CLOSED_FORM;
  const int NS = cf ? 3 : 1;
  const float alpha = a.scalars[0], m0 = a.scalars[1], s0 = a.scalars[2];
  const float aabs = fabsf(alpha), sg0 = fabsf(s0);
  const float w0 = cf ? m0 : fmaf(sg0, eps_0(a), m0);
  const float half_log_a = 0.5f * logf(aabs);
  const bool train = a.y != nullptr;
  float tot[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // ll, kl, g, alpha term, bad ids, -
  const int64_t nw = (int64_t)gridDim.x * (BLOCK / 64);
  for (int64_t r = (int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); r < a.B; r += nw) {
    float first = 0.f, tb = 0.f, klrow = 0.f;         // sum_f v w_f ; sum_f v^2 s_w^2 ; weighted KL (lane-partial)
    for (int f = 0; f < a.F; ++f) {                    // first-order weights
      const int64_t e = load_id(a, r * a.F + f, tot[4]);
      const float v = a.xv ? a.xv[r * a.F + f] : 1.f;
      const float2 th = *reinterpret_cast<const float2*>(a.bias + 2 * (size_t)e);
      const float sw = fabsf(th.y);
      first += v * (cf ? th.x : fmaf(sw, eps_w(a, e), th.x));
      tb = fmaf(v * v, th.y * th.y, tb);
      if (train && lane == 0) {
        const int g = group_index(a.group_hi, a.G, e);
        klrow = fmaf((float)(a.group_n[g] / a.W[g]) * a.inv_occ[e], kl_normal(th.x, sw, prior_w(a, g)), klrow);
      }
    }
    float y2 = 0.f, t2 = 0.f;
    for (int k = lane; k < a.d; k += 64) {
      float S = 0.f, M2 = 0.f, S2 = 0.f, R = 0.f;     // sum v*z (or v*mu); sum v^2 mu^2 (or v^2 z^2); sum v^2 s^2; sum (q^2 - v^4 mu^4)
      for (int f = 0; f < a.F; ++f) {
        float badk = 0.f;
        const int64_t e = load_id(a, r * a.F + f, badk);
        const float v = a.xv ? a.xv[r * a.F + f] : 1.f;
        const float* row = a.entity + (size_t)e * (2 * (size_t)a.d);
        const float mu = row[k], s = row[a.d + k], sg = fabsf(s);
        const float z = cf ? mu : fmaf(sg, eps_v(a, e, k), mu);
        S = fmaf(v, z, S);
        const float am = v * v * z * z, bs = v * v * s * s;
        M2 += am; S2 += bs;
        R += (am + bs) * (am + bs) - am * am;
        if (train) {
          const int g = group_index(a.group_hi, a.G, e);
          klrow = fmaf((float)(a.group_n[g] / a.W[g]) * a.inv_occ[e], kl_normal(mu, sg, prior_v(a, g, k)), klrow);
        }
      }
      const float Q = M2 + S2;
      y2 += S * S - M2;
      t2 += Q * Q - M2 * M2 - R;
      if (train) {
        float* st = state + (size_t)r * NS * a.d;
        st[k] = S;
        if (cf) { st[a.d + k] = S2; st[2 * a.d + k] = Q; }
      }
    }
    y2 = wave_sum(y2); t2 = wave_sum(t2);
    const float p = w0 + first + 0.5f * y2;
    if (lane == 0) pred[r] = p;
    if (train) {
      tot[1] += klrow;
      if (lane == 0) {
        const float yv = a.y[r];
        float ll, dll, at;
        if (cf) {            // closed-form expected log-likelihood (vfm-tomasrch.py:446-449; 'reg' only)
          const float Tn = s0 * s0 + tb + 0.5f * t2;
          const float diff = yv - p;
          ll = half_log_a - 0.5f * aabs * (diff * diff + Tn);
          dll = aabs * diff;
          at = 0.5f * (diff * diff + Tn);       // (positive half: the constant -n / (2 |alpha|) is taken off in fp64, k_var_finalize)
        } else {
          lik_terms(a.lik, yv, p, aabs, 0.5f * LN2 * __builtin_amdgcn_logf(aabs), ll, dll, at);
        }
        const float g = -a.ll_scale * dll;
        tot[0] += ll; tot[2] += g; tot[3] += at;
        grow[r] = g;
      }
    }
  }
  block_sum<6>(tot, sh_red);
  if (threadIdx.x == 0) {
    double* slot = partials + VFM_N_PARTIALS * (1 + (size_t)blockIdx.x);
#pragma unroll
    for (int i = 0; i < 6; ++i) slot[i] = (double)tot[i];
    slot[VFM_SLOT_NTERMS] = (blockIdx.x == 0 && train && (a.lik == VFM_LIK_NORMAL || a.objective == VFM_OBJ_CLOSED_FORM)) ? (double)a.B : 0.0;
    if (blockIdx.x == 0) { partials[7] = (double)gridDim.x; partials[VFM_P_REDUCED] = 0.0; }
  }
}

// loss triple with the prior-aware KL of the global bias
__global__ __launch_bounds__(BLOCK) void k_var_finalize(const VarArgs a, double* __restrict__ partials, float* __restrict__ loss) {
  __shared__ double sh[7][BLOCK / 64];
  const int nblk = (int)partials[7];
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int b = threadIdx.x; b < nblk; b += BLOCK) {
    const double* slot = partials + VFM_N_PARTIALS * (1 + (size_t)b);
#pragma unroll
    for (int i = 0; i < 7; ++i) acc[i] += slot[i];
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc[i] += __shfl_xor(acc[i], m, 64);
    if ((threadIdx.x & 63) == 0) sh[i][threadIdx.x >> 6] = acc[i];
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  double tot[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    tot[i] = 0;
    for (int w = 0; w < BLOCK / 64; ++w) tot[i] += sh[i][w];
  }
  {   // dloss/d|alpha|: sum of the rows' positive halves - n / (2 |alpha|), formed once in fp64 (VFM_P_ALPHA)
    double nterms = 0;
    for (int w = 0; w < BLOCK / 64; ++w) nterms += sh[VFM_SLOT_NTERMS][w];
    if (nterms > 0.0) tot[VFM_P_ALPHA] -= 0.5 * nterms / fabs((double)a.scalars[0]);
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) partials[i] = tot[i];
  partials[VFM_P_REDUCED] = 1.0;
  const float kl0 = kl_normal(a.scalars[1], fabsf(a.scalars[2]), prior0(a));
  const double nll = -(double)a.ll_scale * tot[VFM_P_LL];
  const double kl = (double)kl0 + tot[VFM_P_KL];
  const bool bad = tot[VFM_P_BADID] != 0.0;
  const float nanv = __builtin_nanf("");
  loss[0] = bad ? nanv : (float)(nll + kl);
  loss[1] = bad ? nanv : (float)nll;
  loss[2] = bad ? nanv : (float)kl;
}

// ---------------------------------------------------------------------------------------
// backward: one wave per table row e, dense gradient rows; prior gradients via float atomics (one
// flush per wave and group)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_var_bwd(const VarArgs a, const int32_t* __restrict__ occ_ptr,
                                                   const int32_t* __restrict__ occ_pos, const float* __restrict__ state,
                                                   const float* __restrict__ grow, const double* __restrict__ partials,
                                                   const float* __restrict__ grad_out, float* __restrict__ g_entity,
                                                   float* __restrict__ g_bias, float* __restrict__ g_scalars,
                                                   float* __restrict__ g_priors) {
  const int lane = threadIdx.x & 63;
  const bool cf = a.objective == VFM_OBJ_CLOSED_FORM;
  const int NS = cf ? 3 : 1;
  const float gout = grad_out[0];
  const float alpha = a.scalars[0], m0 = a.scalars[1], s0 = a.scalars[2];
  const float aabs = fabsf(alpha), sg0 = fmaxf(fabsf(s0), SIGMA_MIN);
  const float h = cf ? 0.5f * a.ll_scale * aabs : 0.f;       // dloss/dT_n
  if (blockIdx.x == 0 && threadIdx.x == 0) {                 // the three scalars + the global prior
    const bool ok = partials[VFM_P_REDUCED] == 1.0;
    const float nanv = __builtin_nanf("");
    const float sum_g = ok ? (float)partials[VFM_P_G] : nanv, sum_a = (float)partials[VFM_P_ALPHA];
    const float2 p0 = prior0(a);
    const float dm = m0 - p0.x;
    g_scalars[0] = (a.lik == VFM_LIK_NORMAL) ? gout * signf(alpha) * a.ll_scale * sum_a : 0.f;
    g_scalars[1] = gout * (sum_g + dm / (p0.y * p0.y));
    const float e0 = cf ? 0.f : eps_0(a);
    g_scalars[2] = gout * signf(s0) * (e0 * sum_g + 2.f * h * sg0 * (float)a.B + sg0 / (p0.y * p0.y) - 1.f / sg0);
    if (g_priors) {
      g_priors[0] = gout * (-dm / (p0.y * p0.y));
      g_priors[1] = gout * signf(a.priors[1]) * (1.f / p0.y - (sg0 * sg0 + dm * dm) / (p0.y * p0.y * p0.y));
    }
  }
  float acc_mp[MAXKB], acc_sp[MAXKB], acc_mw = 0.f, acc_sw = 0.f;
#pragma unroll
  for (int i = 0; i < MAXKB; ++i) { acc_mp[i] = 0.f; acc_sp[i] = 0.f; }
  int gcur = -1;
  auto flush = [&]() {
    if (!g_priors || gcur < 0) return;
    float* gm = g_priors + 2 + 2 * a.G;
#pragma unroll
    for (int i = 0; i < MAXKB; ++i) {
      const int k = lane + 64 * i;
      if (k < a.d) {
        atomicAdd(gm + (size_t)gcur * a.d + k, acc_mp[i]);
        atomicAdd(gm + (size_t)a.G * a.d + (size_t)gcur * a.d + k, acc_sp[i]);
      }
      acc_mp[i] = 0.f; acc_sp[i] = 0.f;
    }
    if (lane == 0) { atomicAdd(g_priors + 2 + gcur, acc_mw); atomicAdd(g_priors + 2 + a.G + gcur, acc_sw); }
    acc_mw = 0.f; acc_sw = 0.f;
  };
  const int64_t nw = (int64_t)gridDim.x * (BLOCK / 64);
  int nclamp = 0;
  for (int64_t e = (int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); e < a.T; e += nw) {
    int beg = occ_ptr[e], end = occ_ptr[e + 1];
    span_ok(a, beg, end, nclamp);
    float* ge = g_entity + (size_t)e * (2 * (size_t)a.d);
    if (beg == end) {                                         // not in the batch: dense zero row
      for (int k = lane; k < 2 * a.d; k += 64) ge[k] = 0.f;
      if (lane == 0) *reinterpret_cast<float2*>(g_bias + 2 * (size_t)e) = make_float2(0.f, 0.f);
      continue;
    }
    const int g = group_index(a.group_hi, a.G, e);
    if (g != gcur) { flush(); gcur = g; }
    // sums over the occurrences: sum g_r v, sum g_r v^2, sum v^2 (every lane walks the list)
    float gv = 0.f, gv2 = 0.f, v2 = 0.f;
    for (int o = beg; o < end; ++o) {
      const int pos = occ_pos[o];
      const float v = a.xv ? a.xv[pos] : 1.f;
      const float gr = grow[pos / a.F];
      gv = fmaf(gr, v, gv); gv2 = fmaf(gr * v, v, gv2); v2 = fmaf(v, v, v2);
    }
    const float c = (float)(a.group_n[g] / a.W[g]) * a.inv_occ[e] * (float)(end - beg);   // KL weight of e
    const float* row = a.entity + (size_t)e * (2 * (size_t)a.d);
    int kb = 0;
    for (int k = lane; k < a.d; k += 64, ++kb) {
      const float mu = row[k], s = row[a.d + k], sg = fmaxf(fabsf(s), SIGMA_MIN);
      const float ep = cf ? 0.f : eps_v(a, e, k);
      const float z = cf ? mu : fmaf(fabsf(s), ep, mu);
      float A1 = 0.f, A2 = 0.f, A3 = 0.f, Av2s = 0.f, Av2q = 0.f;
      for (int o = beg; o < end; ++o) {
        const int pos = occ_pos[o];
        const float v = a.xv ? a.xv[pos] : 1.f;
        const float* st = state + (size_t)(pos / a.F) * NS * a.d;
        A1 = fmaf(grow[pos / a.F] * v, st[k], A1);
        if (cf) {
          A2 = fmaf(v * v, st[a.d + k], A2);                  // sum_r v^2 * (sum_f' v'^2 s'^2)
          A3 = fmaf(v * v, st[2 * a.d + k], A3);
          Av2s += v * v * v * v;                              // own share inside those sums: v^4
        }
      }
      (void)Av2q;
      const float2 pr = prior_v(a, g, k);
      const float dm = mu - pr.x;
      // dpred/dz_e = v (S - v z): A1 - z * sum g v^2
      float gmu = A1 - z * gv2, gs_;
      if (cf) {
        const float b = s * s, am = mu * mu;
        gmu += 2.f * h * mu * (A2 - Av2s * b);
        gs_ = 2.f * h * s * (A3 - Av2s * (am + b));
      } else {
        gs_ = signf(s) * (A1 - z * gv2) * ep;
      }
      gmu += c * dm / (pr.y * pr.y);
      gs_ += c * signf(s) * (sg / (pr.y * pr.y) - 1.f / sg);
      ge[k] = gout * gmu;
      ge[a.d + k] = gout * gs_;
      if (g_priors) {
        acc_mp[kb] += gout * c * (-dm / (pr.y * pr.y));
        acc_sp[kb] += gout * c * signf(a.priors[2 + 2 * a.G + (size_t)a.G * a.d + (size_t)g * a.d + k]) *
                      (1.f / pr.y - (sg * sg + dm * dm) / (pr.y * pr.y * pr.y));
      }
    }
    if (lane == 0) {
      const float2 th = *reinterpret_cast<const float2*>(a.bias + 2 * (size_t)e);
      const float sw = fmaxf(fabsf(th.y), SIGMA_MIN);
      const float2 pr = prior_w(a, g);
      const float dm = th.x - pr.x;
      const float ew = cf ? 0.f : eps_w(a, e);
      const float g0 = gv + c * dm / (pr.y * pr.y);
      const float g1 = (cf ? 2.f * h * th.y * v2 : signf(th.y) * gv * ew) + c * signf(th.y) * (sw / (pr.y * pr.y) - 1.f / sw);
      *reinterpret_cast<float2*>(g_bias + 2 * (size_t)e) = make_float2(gout * g0, gout * g1);
      if (g_priors) {
        acc_mw += gout * c * (-dm / (pr.y * pr.y));
        acc_sw += gout * c * signf(a.priors[2 + a.G + g]) * (1.f / pr.y - (sw * sw + dm * dm) / (pr.y * pr.y * pr.y));
      }
    }
  }
  flush();
  if (nclamp != 0 && a.status) atomicAdd(a.status, nclamp);
}

// the inverted index stores ROW numbers; the variants need the position r*F + f of every occurrence (its value)
__global__ void k_positions(const int32_t* __restrict__ occ_ptr, const int32_t* __restrict__ occ_rows, const void* x,
                            int id64, int F, int64_t T, int32_t* __restrict__ occ_pos, int64_t B, int32_t* __restrict__ status) {
  // entity e's occurrences in row r: the fields f of r with x[r,f] == e, in field order (the index is stable)
  const int n_occ = (int)(B * F);
  int nclamp = 0;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < T; e += (int64_t)gridDim.x * blockDim.x) {
    int beg = occ_ptr[e], end = occ_ptr[e + 1];
    if (beg < 0 || end < beg || end > n_occ) { beg = end = 0; ++nclamp; }      // (a corrupted index is clamped, never followed)
    int last_r = -1, f = 0;
    for (int o = beg; o < end; ++o) {
      int r = occ_rows[o];
      if ((unsigned)r >= (unsigned)B) { r = 0; ++nclamp; }
      if (r != last_r) { last_r = r; f = 0; }
      for (; f < F; ++f) {
        const int64_t id = id64 ? ((const int64_t*)x)[(int64_t)r * F + f] : (int64_t)((const int32_t*)x)[(int64_t)r * F + f];
        if (id == e || (e == 0 && (id < 0 || id >= T))) break;
      }
      occ_pos[o] = r * F + (f < F ? f : F - 1);
      ++f;
    }
  }
  if (nclamp != 0 && status) atomicAdd(status, nclamp);
}

#include "vfm_variants8.hpp"

// lane-group shape of the d % 8 == 0 kernels: LPE lanes x CPL blocks of 8 coordinates cover the row
#define VFM_FOR_VAR_SHAPES(X) X(1, 1) X(2, 1) X(4, 1) X(8, 1) X(16, 1) X(32, 1) X(64, 1) X(64, 2)
void var_shape(int d, int* lpe, int* cpl) {
  const int D8 = d >> 3;
  int l = 1;
  while (l < D8 && l < 64) l <<= 1;
  *lpe = l;
  *cpl = (D8 + l - 1) / l;
}

template <int LPE, int CPL, bool CF, bool HASV>
void launch_var_fwd8(const VarArgs& a, float* pred, double* partials, float* state, float* grow, hipStream_t st) {
  constexpr int GPB = BLOCK / LPE;
  int64_t nb = (a.B + GPB - 1) / GPB;
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  if (a.priors)
    hipLaunchKernelGGL((k_var_fwd8<LPE, CPL, CF, HASV, true>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, pred, partials, state, grow);
  else
    hipLaunchKernelGGL((k_var_fwd8<LPE, CPL, CF, HASV, false>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, pred, partials, state, grow);
}

template <int LPE, int CPL, bool CF, bool HASV>
void launch_var_bwd8(const VarArgs& a, const vfm_index_t* idx, const int32_t* occ_pos, const float* state, const float* grow,
                     const double* partials, const float* grad_out, float* g_entity, float* g_bias, float* g_scalars,
                     float* g_priors, float* prows, hipStream_t st) {
  constexpr int GPB = BLOCK / LPE;
  int64_t nb = (a.T + GPB - 1) / GPB;
  if (nb > VAR_BWD_BLOCKS) nb = VAR_BWD_BLOCKS;
  if (a.priors) {
    hipLaunchKernelGGL((k_var_bwd8<LPE, CPL, CF, HASV, true>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, idx->occ_ptr,
                       idx->occ_rows, occ_pos, state, grow, partials, grad_out, g_entity, g_bias, g_scalars, g_priors, prows);
    int64_t epb = (a.T + nb - 1) / nb;
    epb = (epb + GPB - 1) / GPB * GPB;
    float* parts = prows + (size_t)(VAR_BWD_BLOCKS + a.G) * (2 * (size_t)a.d + 8);
    const unsigned ky = (unsigned)((2 * a.d + 2 + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(k_var_priors_part, dim3((unsigned)a.G, ky, VAR_PSUM_CH), dim3(BLOCK), 0, st, a, prows, epb, (int)nb, parts);
    hipLaunchKernelGGL(k_var_priors_sum, dim3((unsigned)a.G, ky), dim3(BLOCK), 0, st, a, parts, g_priors);
  } else {
    hipLaunchKernelGGL((k_var_bwd8<LPE, CPL, CF, HASV, false>), dim3((unsigned)nb), dim3(BLOCK), 0, st, a, idx->occ_ptr,
                       idx->occ_rows, occ_pos, state, grow, partials, grad_out, g_entity, g_bias, g_scalars, g_priors, prows);
  }
}

// the kernels of vfm_variants8.hpp serve d % 8 == 0 (env VFM_VARIANT_SCALAR=1 forces the scalar pair: A/B runs, tests)
// and draw eps in the kernel; eps tables go through the scalar pair
bool use_var8(const vfm_problem_t* p, const float* eps_entity) {
  return (p->d & 7) == 0 && p->d <= 1024 && eps_entity == nullptr && env_int("VFM_VARIANT_SCALAR", 0) == 0;
}

int check_var(const vfm_problem_t* p, int objective) {
  if (!p) return fail(VFM_E_INVALID, "problem is NULL");
  if (p->struct_size != (uint32_t)sizeof(vfm_problem_t) || p->abi_version != (uint32_t)VFM_ABI_VERSION)
    return fail(VFM_E_INVALID, "vfm_problem_t: struct_size / abi_version differ from this library's (VFM_STRUCT_INIT)");
  if (p->B < 0 || p->T <= 0 || p->T > 0xFFFFFFFELL || p->F < 1 || p->F > VFM_MAX_FIELDS || p->d < 1 ||
      (p->id_bits != 32 && p->id_bits != 64) || p->B * (int64_t)p->F > 0x7FFFFFFFLL)
    return fail(VFM_E_INVALID, "variant: bad problem");
  if (p->d > 64 * MAXKB) return fail(VFM_E_UNSUPPORTED, "variant: d above 1024");
  if (p->n_samples != 1) return fail(VFM_E_UNSUPPORTED, "variant: one variational sample");
  if (p->flags != 0) return fail(VFM_E_UNSUPPORTED, "variant: no flags (|.| link, single rank)");
  if (p->dev_step || p->wrec) return fail(VFM_E_UNSUPPORTED, "variant: no device-side step state / packed first-order records");
  if (objective != VFM_OBJ_SAMPLED && objective != VFM_OBJ_CLOSED_FORM) return fail(VFM_E_INVALID, "variant: unknown objective");
  if (objective == VFM_OBJ_CLOSED_FORM && p->likelihood != VFM_LIK_NORMAL)
    return fail(VFM_E_UNSUPPORTED, "variant: the closed-form expected log-likelihood is the Normal one (vfm-tomasrch.py:446-449)");
  return 0;
}

VarArgs make_var(const vfm_problem_t* p, int objective, const void* x, const float* xv, const float* y, const float* entity,
                 const float* bias, const float* inv_occ, const float* scalars, const double* W, const float* priors,
                 const float* ee, const float* eb, const float* eg) {
  VarArgs a;
  memset(&a, 0, sizeof(a));
  a.B = p->B; a.T = p->T; a.F = p->F; a.d = p->d; a.G = p->F; a.id64 = p->id_bits == 64; a.lik = p->likelihood;
  a.n_occ = (int32_t)(p->B * (int64_t)p->F);
  a.objective = objective; a.eps_mode = ee ? EPS_TABLE : EPS_PHILOX;
  a.ll_scale = (float)((double)p->nb_train / (double)(p->B_global > 0 ? p->B_global : 1));
  a.key.seed_lo = (uint32_t)p->seed; a.key.seed_hi = (uint32_t)(p->seed >> 32);
  a.key.step_lo = (uint32_t)p->step; a.key.step_hi = (uint32_t)(p->step >> 32);
  a.x = x; a.xv = xv; a.y = y; a.entity = entity; a.bias = bias; a.inv_occ = inv_occ; a.scalars = scalars; a.W = W;
  a.priors = priors; a.eps_entity = ee; a.eps_bias = eb; a.eps_global = eg;
  for (int g = 0; g < p->F; ++g) { a.group_hi[g] = p->group_hi[g]; a.group_n[g] = p->group_n[g]; }
  return a;
}

}  // namespace
}  // namespace vfm

using namespace vfm;

extern "C" {

int vfm_variant_fwd_f32(const vfm_problem_t* p, int32_t objective, const void* x, const float* values, const float* y,
                        const float* entity_params, const float* bias_params, const float* inv_occ,
                        const float* scalars, const double* W, const float* priors, const float* eps_entity,
                        const float* eps_bias, const float* eps_global, float* pred, double* partials, float* state,
                        float* grow, float* loss, void* stream) {
  if (int rc = check_var(p, objective)) return rc;
  const bool train = y != nullptr;
  if (!x || !entity_params || !bias_params || !scalars || !pred || !partials ||
      (train && (!inv_occ || !W || !state || !grow)))
    return fail(VFM_E_INVALID, "vfm_variant_fwd_f32: NULL pointer");
  const int neps = (eps_entity != nullptr) + (eps_bias != nullptr) + (eps_global != nullptr);
  if (neps != 0 && neps != 3) return fail(VFM_E_INVALID, "vfm_variant_fwd_f32: give all three eps tables or none");
  VarArgs a = make_var(p, objective, x, values, y, entity_params, bias_params, inv_occ, scalars, W, priors, eps_entity,
                       eps_bias, eps_global);
  hipStream_t st = (hipStream_t)stream;
  int64_t nb = (p->B + BLOCK / 64 - 1) / (BLOCK / 64);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  if (use_var8(p, eps_entity)) {
    int lpe, cpl;
    var_shape(p->d, &lpe, &cpl);
    const bool cf = objective == VFM_OBJ_CLOSED_FORM, hv = values != nullptr;
#define X(L_, C_)                                                                                         \
    if (lpe == L_ && cpl == C_) {                                                                         \
      if (cf && hv) launch_var_fwd8<L_, C_, true, true>(a, pred, partials, state, grow, st);              \
      else if (cf) launch_var_fwd8<L_, C_, true, false>(a, pred, partials, state, grow, st);              \
      else if (hv) launch_var_fwd8<L_, C_, false, true>(a, pred, partials, state, grow, st);              \
      else launch_var_fwd8<L_, C_, false, false>(a, pred, partials, state, grow, st);                     \
    }
    VFM_FOR_VAR_SHAPES(X)
#undef X
  } else {
    hipLaunchKernelGGL(k_var_fwd, dim3((unsigned)nb), dim3(BLOCK), 0, st, a, pred, partials, state, grow);
  }
  if (train && loss) hipLaunchKernelGGL(k_var_finalize, dim3(1), dim3(BLOCK), 0, st, a, partials, loss);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail_hip(e, "vfm_variant_fwd_f32");
}

int64_t vfm_variant_workspace_elems(int64_t B, int32_t F, int32_t d) {
  return B * (int64_t)F + (int64_t)(VAR_BWD_BLOCKS + F + (int64_t)F * VAR_PSUM_CH) * (2 * (int64_t)d + 8);
}

int vfm_variant_bwd_f32(const vfm_problem_t* p, int32_t objective, const vfm_index_t* idx, int32_t* occ_pos_ws,
                        const void* x, const float* values, const float* entity_params, const float* bias_params,
                        const float* inv_occ, const float* scalars, const double* W, const float* priors,
                        const float* eps_entity, const float* eps_bias, const float* eps_global, const float* state,
                        const float* grow, const double* partials, const float* grad_out, float* g_entity,
                        float* g_bias, float* g_scalars, float* g_priors, void* stream) {
  if (int rc = check_var(p, objective)) return rc;
  if (idx && (idx->struct_size != (uint32_t)sizeof(vfm_index_t) || idx->abi_version != (uint32_t)VFM_ABI_VERSION))
    return fail(VFM_E_INVALID, "vfm_index_t: struct_size / abi_version differ from this library's (VFM_STRUCT_INIT)");
  if (!idx || !idx->occ_ptr || (p->B > 0 && !idx->occ_rows) || !occ_pos_ws || !x || !entity_params || !bias_params ||
      !inv_occ || !scalars || !W || !partials || !grad_out || !g_entity || !g_bias || !g_scalars ||
      (p->B > 0 && (!state || !grow)) || ((priors != nullptr) != (g_priors != nullptr)))
    return fail(VFM_E_INVALID, "vfm_variant_bwd_f32: NULL pointer (g_priors goes with priors)");
  VarArgs a = make_var(p, objective, x, values, nullptr, entity_params, bias_params, inv_occ, scalars, W, priors,
                       eps_entity, eps_bias, eps_global);
  a.status = idx->status;
  hipStream_t st = (hipStream_t)stream;
  if (use_var8(p, eps_entity)) {
    // ws = [B*F positions (only used with values) | (VAR_BWD_BLOCKS + F) partial rows of the prior gradients | F * VAR_PSUM_CH chunk sums]
    float* prows = reinterpret_cast<float*>(occ_pos_ws + (size_t)p->B * p->F);
    if (g_priors) {       // row headers = -1 (no group)
      const hipError_t e = hipMemsetAsync(prows, 0xFF, sizeof(float) * (size_t)(VAR_BWD_BLOCKS + p->F) * (2 * (size_t)p->d + 8), st);
      if (e != hipSuccess) return fail_hip(e, "vfm_variant_bwd_f32: memset");
    }
    if (values) {
      int64_t nbp = (p->T + 255) / 256;
      if (nbp > 4096) nbp = 4096;
      hipLaunchKernelGGL(k_positions, dim3((unsigned)nbp), dim3(256), 0, st, idx->occ_ptr, idx->occ_rows, x,
                         (int)(p->id_bits == 64), (int)p->F, p->T, occ_pos_ws, p->B, idx->status);
    }
    int lpe, cpl;
    var_shape(p->d, &lpe, &cpl);
    const bool cf = objective == VFM_OBJ_CLOSED_FORM, hv = values != nullptr;
#define X(L_, C_)                                                                                                        \
    if (lpe == L_ && cpl == C_) {                                                                                        \
      if (cf && hv) launch_var_bwd8<L_, C_, true, true>(a, idx, occ_pos_ws, state, grow, partials, grad_out, g_entity, g_bias, g_scalars, g_priors, prows, st);   \
      else if (cf) launch_var_bwd8<L_, C_, true, false>(a, idx, occ_pos_ws, state, grow, partials, grad_out, g_entity, g_bias, g_scalars, g_priors, prows, st);   \
      else if (hv) launch_var_bwd8<L_, C_, false, true>(a, idx, occ_pos_ws, state, grow, partials, grad_out, g_entity, g_bias, g_scalars, g_priors, prows, st);   \
      else launch_var_bwd8<L_, C_, false, false>(a, idx, occ_pos_ws, state, grow, partials, grad_out, g_entity, g_bias, g_scalars, g_priors, prows, st);          \
    }
    VFM_FOR_VAR_SHAPES(X)
#undef X
    const hipError_t e8 = hipGetLastError();
    return e8 == hipSuccess ? 0 : fail_hip(e8, "vfm_variant_bwd_f32");
  }
  if (g_priors) {
    const hipError_t e = hipMemsetAsync(g_priors, 0, sizeof(float) * (size_t)(2 + 2 * p->F + 2 * (size_t)p->F * p->d), st);
    if (e != hipSuccess) return fail_hip(e, "vfm_variant_bwd_f32: memset");
  }
  int64_t nbp = (p->T + 255) / 256;
  if (nbp > 4096) nbp = 4096;
  hipLaunchKernelGGL(k_positions, dim3((unsigned)nbp), dim3(256), 0, st, idx->occ_ptr, idx->occ_rows, x,
                     (int)(p->id_bits == 64), (int)p->F, p->T, occ_pos_ws, p->B, idx->status);
  int64_t nb = (p->T + BLOCK / 64 - 1) / (BLOCK / 64);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(k_var_bwd, dim3((unsigned)nb), dim3(BLOCK), 0, st, a, idx->occ_ptr, occ_pos_ws, state, grow, partials,
                     grad_out, g_entity, g_bias, g_scalars, g_priors);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail_hip(e, "vfm_variant_bwd_f32");
}

}  // extern "C"
