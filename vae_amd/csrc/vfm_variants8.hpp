// vfm_variants8.hpp -- the ELBO-variant kernels for embedding sizes that are a multiple of 8 (vfm_variants.hip holds
// the general scalar pair; same arithmetic, same state layout, either forward goes with either backward).
// Included inside `namespace vfm { namespace {` of vfm_variants.hip, after VarArgs and the prior helpers.
//
// Layout of the work: a lane group of LPE lanes owns a batch row (forward) or a table row (backward); lane p owns
// the coordinate blocks kb = p + i*LPE (i < CPL) of 8 coordinates each = two float4 loads per table and exactly the
// 8 normals of ONE Philox call, so no normal is drawn twice and none is thrown away (the scalar kernels draw 8 per
// coordinate and keep one).  Forward: the (row, field) occurrences of a lane group form one stream with the ids
// two and the table rows one occurrence ahead of the arithmetic; no load sits under a branch (hipcc drains every
// load in flight at such a merge).  Backward: two occurrences of the inverted list in flight; the gradients of the
// learnable group priors are summed per workgroup in LDS and written as one partial row per (workgroup, group) --
// workgroups own CONTIGUOUS entity ranges, so a workgroup meets its groups one after the other -- and added up by
// k_var_priors_sum in row order: no atomics, the result does not depend on the scheduling.
#pragma once

constexpr int VAR_BWD_BLOCKS = 2048;           // workgroups of k_var_bwd8 (rows of the prior-gradient scratch: + G)

__device__ __forceinline__ void ld8(const float* __restrict__ p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void st8(float* __restrict__ p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// id of occurrence `pos`, either width, two dword loads and a mask (no branch); range check by the caller
struct VarId { uint32_t lo, hi; };
__device__ __forceinline__ VarId var_raw_id(const VarArgs& a, int64_t pos, int shift, int hi_off, uint32_t m64) {
  const char* pa = reinterpret_cast<const char*>(a.x) + ((size_t)pos << shift);
  VarId v;
  v.lo = *reinterpret_cast<const uint32_t*>(pa);
  const uint32_t h = *reinterpret_cast<const uint32_t*>(pa + hi_off);
  v.hi = (h & m64) | ((uint32_t)((int32_t)v.lo >> 31) & ~m64);
  return v;
}

// KL(N(mu, sg) || prior): N(0,1), or the group's learnable Normal(pm, ps) -- kl_normal of vfm_variants.hip with the
// hardware log2 / rcp
template <bool PRI>
__device__ __forceinline__ float kl_var(float mu, float sg, float pm, float ps) {
  if constexpr (!PRI) {
    return kl_std_normal(mu, sg);
  } else {
    const float dm = mu - pm, ip = __builtin_amdgcn_rcpf(ps);
    return LN2 * (__builtin_amdgcn_logf(ps) - __builtin_amdgcn_logf(fmaxf(sg, SIGMA_MIN))) +
           0.5f * (sg * sg + dm * dm) * (ip * ip) - 0.5f;
  }
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
template <int CPL, bool PRI>
struct VarFld {                 // one (row, field) occurrence in registers
  uint32_t e;
  int g;                        // its id group
  float v, io;
  float2 th;
  float mu[CPL][8], s[CPL][8];
  float pm[PRI ? CPL : 1][8], ps[PRI ? CPL : 1][8], pwm, pws;      // the group's priors (PRI)
};

template <int LPE, int CPL, bool CF, bool HASV, bool PRI>
__global__ __launch_bounds__(BLOCK) void k_var_fwd8(const VarArgs a, float* __restrict__ pred, double* __restrict__ partials,
                                                    float* __restrict__ state, float* __restrict__ grow) {
  constexpr int GPB = BLOCK / LPE;
  constexpr int NS = CF ? 3 : 1;
  __shared__ float sh_red[6 * 4];
  __shared__ float sh_cw[VFM_MAX_FIELDS];
  __shared__ int64_t sh_hi[VFM_MAX_FIELDS];
  const int tid = threadIdx.x, lig = tid % LPE, grp = tid / LPE;
  const bool train = a.y != nullptr;
  const int d = a.d, D8 = d >> 3, F = a.F;
  if (tid < a.G) {
    sh_cw[tid] = train ? (float)(a.group_n[tid] / a.W[tid]) : 0.f;
    sh_hi[tid] = a.group_hi[tid];
  }
  __syncthreads();
  const float alpha = a.scalars[0], m0 = a.scalars[1], s0 = a.scalars[2];
  const float aabs = fabsf(alpha), sg0 = fabsf(s0);
  const float w0 = CF ? m0 : fmaf(sg0, eps_0(a), m0);
  const float half_log_a = 0.5f * LN2 * __builtin_amdgcn_logf(aabs);
  const int id_shift = a.id64 ? 3 : 2, hi_off = a.id64 ? 4 : 0;
  const uint32_t m64 = 0u - (uint32_t)(a.id64 != 0);
  const float* pri_m = PRI ? a.priors + 2 + 2 * a.G : nullptr;        // [G, d] means, then [G, d] scales
  const float* pri_s = PRI ? pri_m + (size_t)a.G * d : nullptr;

  float tot[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // ll, kl, g, alpha term, bad ids, -
  // this lane group's rows r0, r0 + stride, ...; its occurrence stream is their (row, field) pairs in order
  const int64_t stride = (int64_t)gridDim.x * GPB;
  const int64_t r0 = (int64_t)blockIdx.x * GPB + grp;
  const int64_t nrows = r0 < a.B ? (a.B - 1 - r0) / stride + 1 : 0;
  const int64_t nocc = nrows * F;
  if (nocc > 0) {
    const int64_t last = (r0 + (nrows - 1) * stride) * F + (F - 1);          // clamp target of the prefetches
    int64_t ri = r0, pi = 0; int fi = 0;                                     // id cursor
    VarId nid; float nv = 1.f; bool nlive;
    auto load_id = [&]() {
      nlive = pi < nocc;
      const int64_t pos = nlive ? ri * F + fi : last;
      nid = var_raw_id(a, pos, id_shift, hi_off, m64);
      if constexpr (HASV) nv = a.xv[pos];
      if (nlive) { ++pi; if (++fi == F) { fi = 0; ri += stride; } }
    };
    auto issue = [&](VarFld<CPL, PRI>& q) {                                       // table rows of the occurrence whose id is in nid
      const bool ok = nid.hi == 0u && (int64_t)nid.lo < a.T;
      if (nlive && !ok) tot[4] += 1.f;
      q.e = ok ? nid.lo : 0u;
      q.v = nv;
      const float* row = a.entity + (size_t)q.e * (2 * (size_t)d);
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        int kb = lig + i * LPE;
        kb = kb < D8 ? kb : D8 - 1;                                          // (lanes past the end re-load the last block)
        ld8(row + 8 * kb, q.mu[i]);
        ld8(row + d + 8 * kb, q.s[i]);
      }
      q.th = *reinterpret_cast<const float2*>(a.bias + 2 * (size_t)q.e);
      q.io = train ? a.inv_occ[q.e] : 0.f;
      q.g = group_index(sh_hi, a.G, (int64_t)q.e);
      q.pwm = 0.f; q.pws = 1.f;
      if constexpr (PRI) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          int kb = lig + i * LPE;
          kb = kb < D8 ? kb : D8 - 1;
          ld8(pri_m + (size_t)q.g * d + 8 * kb, q.pm[i]);
          ld8(pri_s + (size_t)q.g * d + 8 * kb, q.ps[i]);
        }
        q.pwm = a.priors[2 + q.g];
        q.pws = a.priors[2 + a.G + q.g];
      }
    };
    float S[CPL][8], M2[CPL][8], S2[CPL][8], R[CPL][8];
    float first = 0.f, tb = 0.f, klrow = 0.f;
    auto reset = [&]() {
#pragma unroll
      for (int i = 0; i < CPL; ++i)
#pragma unroll
        for (int t = 0; t < 8; ++t) { S[i][t] = 0.f; M2[i][t] = 0.f; S2[i][t] = 0.f; R[i][t] = 0.f; }
      first = 0.f; tb = 0.f; klrow = 0.f;
    };
    int64_t rc = r0, left = nocc; int fc = 0;                                // consume cursor
    auto consume = [&](const VarFld<CPL, PRI>& q) {
      const float cw = sh_cw[q.g] * q.io;                                    // weight of this occurrence's KL terms
      const float v = q.v, v2 = v * v;
      float nb = 0.f;
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int kb = lig + i * LPE;
        const bool valid = kb < D8;
        float n[8], nbi = 0.f;
        if constexpr (!CF) normal8b(a.key, q.e, (uint32_t)(valid ? kb : 0) + (a.key.chunk_off >> 1), n, nbi);
        if (i == 0) nb = nbi;                                                // (kb == 0 sits in lane 0, i == 0)
        float klb = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float mu = q.mu[i][t], s = q.s[i][t], sg = fabsf(s);
          float z = mu;
          if constexpr (!CF) z = fmaf(sg, n[t], mu);
          const float am = v2 * z * z, bs = v2 * s * s;
          if (valid) {
            S[i][t] = fmaf(v, z, S[i][t]);
            M2[i][t] += am; S2[i][t] += bs;
            R[i][t] += (am + bs) * (am + bs) - am * am;
          }
          klb += kl_var<PRI>(mu, sg, PRI ? q.pm[i][t] : 0.f, PRI ? fmaxf(fabsf(q.ps[i][t]), SIGMA_MIN) : 1.f);
        }
        if (valid) klrow = fmaf(cw, klb, klrow);
      }
      if (lig == 0) {                                                        // first-order weight
        const float sw = fabsf(q.th.y);
        first = fmaf(v, CF ? q.th.x : fmaf(sw, nb, q.th.x), first);
        tb = fmaf(v2, q.th.y * q.th.y, tb);
        klrow = fmaf(cw, kl_var<PRI>(q.th.x, sw, q.pwm, fmaxf(fabsf(q.pws), SIGMA_MIN)), klrow);
      }
      if (fc == F - 1) {                                                     // the row is complete
        float y2 = 0.f, t2 = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          const int kb = lig + i * LPE;
          float Q[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            Q[t] = M2[i][t] + S2[i][t];
            y2 += S[i][t] * S[i][t] - M2[i][t];
            t2 += Q[t] * Q[t] - M2[i][t] * M2[i][t] - R[i][t];
          }
          if (train && kb < D8) {
            float* st = state + (size_t)rc * NS * d + 8 * kb;
            st8(st, S[i]);
            if constexpr (CF) { st8(st + d, S2[i]); st8(st + 2 * d, Q); }
          }
        }
        y2 = group_sum<LPE>(y2); t2 = group_sum<LPE>(t2);
        const float fo = group_sum<LPE>(first), tbs = group_sum<LPE>(tb);     // (only lane 0 carries them)
        const float p = w0 + fo + 0.5f * y2;
        if (lig == 0) pred[rc] = p;
        if (train) {
          tot[1] += klrow;
          if (lig == 0) {
            const float yv = a.y[rc];
            float ll, dll, at;
            if constexpr (CF) {     // closed-form expected log-likelihood (vfm-tomasrch.py:446-449; 'reg' only)
              const float Tn = s0 * s0 + tbs + 0.5f * t2;
              const float diff = yv - p;
              ll = half_log_a - 0.5f * aabs * (diff * diff + Tn);
              dll = aabs * diff;
              at = 0.5f * (diff * diff + Tn);       // (positive half: the constant -n / (2 |alpha|) is taken off in fp64, k_var_finalize)
            } else {
              lik_terms(a.lik, yv, p, aabs, half_log_a, ll, dll, at);
            }
            const float gq = -a.ll_scale * dll;
            tot[0] += ll; tot[2] += gq; tot[3] += at;
            grow[rc] = gq;
          }
        }
        reset();
      }
      --left;
      if (++fc == F) { fc = 0; rc += stride; }
    };
    VarFld<CPL, PRI> A, Bq;
    reset();
    load_id();
    issue(A);
    load_id();
    while (true) {
      issue(Bq);
      load_id();
      consume(A);
      if (left <= 0) break;
      issue(A);
      load_id();
      consume(Bq);
      if (left <= 0) break;
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

// ---------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------
// prior-gradient scratch: VAR_BWD_BLOCKS + G rows of [group (int), 0, 0, 0 | d mean grads | d scale grads | w mean, w scale, 0, 0]
__device__ __forceinline__ size_t var_prow_len(int d) { return 4 + 2 * (size_t)d + 4; }

template <int LPE, int CPL, bool CF, bool HASV, bool PRI>
__global__ __launch_bounds__(BLOCK) void k_var_bwd8(const VarArgs a, const int32_t* __restrict__ occ_ptr,
                                                    const int32_t* __restrict__ occ_rows, const int32_t* __restrict__ occ_pos,
                                                    const float* __restrict__ state, const float* __restrict__ grow,
                                                    const double* __restrict__ partials, const float* __restrict__ grad_out,
                                                    float* __restrict__ g_entity, float* __restrict__ g_bias,
                                                    float* __restrict__ g_scalars, float* __restrict__ g_priors,
                                                    float* __restrict__ prows) {
  constexpr int GPB = BLOCK / LPE;
  constexpr int NS = CF ? 3 : 1;
  __shared__ float sh_acc[PRI ? GPB * 16 * LPE * CPL : 1];      // [GPB][2 d'] partial prior gradients, d' = 8 LPE CPL
  __shared__ float sh_w[PRI ? 2 * GPB : 1];
  __shared__ float sh_cw[VFM_MAX_FIELDS];
  __shared__ int64_t sh_hi[VFM_MAX_FIELDS];
  const int tid = threadIdx.x, lig = tid % LPE, grp = tid / LPE;
  const int d = a.d, D8 = d >> 3;
  if (tid < a.G) { sh_cw[tid] = (float)(a.group_n[tid] / a.W[tid]); sh_hi[tid] = a.group_hi[tid]; }
  __syncthreads();
  const float gout = grad_out[0];
  const float alpha = a.scalars[0], m0 = a.scalars[1], s0 = a.scalars[2];
  const float aabs = fabsf(alpha), sg0 = fmaxf(fabsf(s0), SIGMA_MIN);
  const float h = CF ? 0.5f * a.ll_scale * aabs : 0.f;       // dloss/dT_n
  if (blockIdx.x == 0 && tid == 0) {                         // the three scalars + the global prior
    const bool ok = partials[VFM_P_REDUCED] == 1.0;
    const float nanv = __builtin_nanf("");
    const float sum_g = ok ? (float)partials[VFM_P_G] : nanv, sum_a = (float)partials[VFM_P_ALPHA];
    const float2 p0 = prior0(a);
    const float dm = m0 - p0.x;
    g_scalars[0] = (a.lik == VFM_LIK_NORMAL) ? gout * signf(alpha) * a.ll_scale * sum_a : 0.f;
    g_scalars[1] = gout * (sum_g + dm / (p0.y * p0.y));
    const float e0 = CF ? 0.f : eps_0(a);
    g_scalars[2] = gout * signf(s0) * (e0 * sum_g + 2.f * h * sg0 * (float)a.B + sg0 / (p0.y * p0.y) - 1.f / sg0);
    if (PRI) {
      g_priors[0] = gout * (-dm / (p0.y * p0.y));
      g_priors[1] = gout * signf(a.priors[1]) * (1.f / p0.y - (sg0 * sg0 + dm * dm) / (p0.y * p0.y * p0.y));
    }
  }
  const float* pri_m = PRI ? a.priors + 2 + 2 * a.G : nullptr;
  const float* pri_s = PRI ? pri_m + (size_t)a.G * d : nullptr;
  int nclamp = 0;                                            // index entries clamped (vfm_index_t.status)
  // this workgroup's contiguous entity range, walked one id group at a time (uniform loop)
  int64_t epb = (a.T + gridDim.x - 1) / gridDim.x;
  epb = (epb + GPB - 1) / GPB * GPB;
  int64_t e_lo = (int64_t)blockIdx.x * epb;
  const int64_t e_end = e_lo + epb < a.T ? e_lo + epb : a.T;
  int g = e_lo < a.T ? group_index(sh_hi, a.G, e_lo) : 0;
  while (e_lo < e_end) {
    const int64_t seg_hi = (sh_hi[g] < e_end && g + 1 < a.G) ? sh_hi[g] : e_end;
    float acc_mp[CPL][8], acc_sp[CPL][8], acc_mw = 0.f, acc_sw = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
      for (int t = 0; t < 8; ++t) { acc_mp[i][t] = 0.f; acc_sp[i][t] = 0.f; }
    float pwm = 0.f, pws = 1.f, pws_raw = 1.f;
    if constexpr (PRI) { pwm = a.priors[2 + g]; pws_raw = a.priors[2 + a.G + g]; pws = fmaxf(fabsf(pws_raw), SIGMA_MIN); }
    for (int64_t e = e_lo + grp; e < seg_hi; e += GPB) {
      int beg = occ_ptr[e], end = occ_ptr[e + 1];
      span_ok(a, beg, end, nclamp);
      float* ge = g_entity + (size_t)e * (2 * (size_t)d);
      if (beg == end) {                                       // not in the batch: dense zero row
        const float z8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          const int kb = lig + i * LPE;
          if (kb < D8) { st8(ge + 8 * kb, z8); st8(ge + d + 8 * kb, z8); }
        }
        if (lig == 0) *reinterpret_cast<float2*>(g_bias + 2 * (size_t)e) = make_float2(0.f, 0.f);
        continue;
      }
      // the entity's own row (independent of the walk)
      const float* row = a.entity + (size_t)e * (2 * (size_t)d);
      float mu[CPL][8], s[CPL][8];
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        int kb = lig + i * LPE;
        kb = kb < D8 ? kb : D8 - 1;
        ld8(row + 8 * kb, mu[i]);
        ld8(row + d + 8 * kb, s[i]);
      }
      const float2 th = *reinterpret_cast<const float2*>(a.bias + 2 * (size_t)e);
      const float c = sh_cw[g] * a.inv_occ[e] * (float)(end - beg);          // KL weight of e
      // walk: A1 = sum g_r v S_r, A2 = sum v^2 S2_r, A3 = sum v^2 Q_r; gv = sum g_r v, gv2 = sum g_r v^2, ...
      float A1[CPL][8], A2[CF ? CPL : 1][8], A3[CF ? CPL : 1][8];
#pragma unroll
      for (int i = 0; i < CPL; ++i)
#pragma unroll
        for (int t = 0; t < 8; ++t) { A1[i][t] = 0.f; if constexpr (CF) { A2[i][t] = 0.f; A3[i][t] = 0.f; } }
      float gv = 0.f, gv2 = 0.f, vv2 = 0.f, v4 = 0.f;
      auto one = [&](int r, float v, float gr) {
        const float gw = gr * v, w2 = v * v;
        gv += gw; gv2 = fmaf(gw, v, gv2); vv2 += w2; v4 = fmaf(w2, w2, v4);
        const float* st = state + (size_t)r * NS * d;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          int kb = lig + i * LPE;
          kb = kb < D8 ? kb : D8 - 1;
          float x1[8];
          ld8(st + 8 * kb, x1);
#pragma unroll
          for (int t = 0; t < 8; ++t) A1[i][t] = fmaf(gw, x1[t], A1[i][t]);
          if constexpr (CF) {
            float x2[8], x3[8];
            ld8(st + d + 8 * kb, x2);
            ld8(st + 2 * d + 8 * kb, x3);
#pragma unroll
            for (int t = 0; t < 8; ++t) { A2[i][t] = fmaf(w2, x2[t], A2[i][t]); A3[i][t] = fmaf(w2, x3[t], A3[i][t]); }
          }
        }
      };
      int o = beg;
      for (; o + 1 < end; o += 2) {                           // two occurrences in flight
        const int ra = row_ok(a, occ_rows[o], nclamp), rb = row_ok(a, occ_rows[o + 1], nclamp);
        float va = 1.f, vb = 1.f;
        if constexpr (HASV) { va = a.xv[occ_pos[o]]; vb = a.xv[occ_pos[o + 1]]; }
        const float ga = grow[ra], gb = grow[rb];
        one(ra, va, ga);
        one(rb, vb, gb);
      }
      if (o < end) {
        const int ra = row_ok(a, occ_rows[o], nclamp);
        float va = 1.f;
        if constexpr (HASV) va = a.xv[occ_pos[o]];
        one(ra, va, grow[ra]);
      }
      // epilogue (the formulas of k_var_bwd)
      float nb = 0.f;
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int kb = lig + i * LPE;
        const bool valid = kb < D8;
        const int kc = valid ? kb : 0;
        float n[8], nbi = 0.f;
        if constexpr (!CF) normal8b(a.key, (uint32_t)e, (uint32_t)kc + (a.key.chunk_off >> 1), n, nbi);
        if (i == 0) nb = nbi;
        float pm[8], ps[8];
        if constexpr (PRI) { ld8(pri_m + (size_t)g * d + 8 * kc, pm); ld8(pri_s + (size_t)g * d + 8 * kc, ps); }
        float gm8[8], gs8[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float m_ = mu[i][t], s_ = s[i][t], sg = fmaxf(fabsf(s_), SIGMA_MIN);
          const float ep = CF ? 0.f : n[t];
          const float z = CF ? m_ : fmaf(fabsf(s_), ep, m_);
          const float prm = PRI ? pm[t] : 0.f, prs = PRI ? fmaxf(fabsf(ps[t]), SIGMA_MIN) : 1.f;
          const float dm = m_ - prm, ip2 = 1.f / (prs * prs);
          float gmu = A1[i][t] - z * gv2, gs_;
          if constexpr (CF) {
            const float b2 = s_ * s_, am = m_ * m_;
            gmu += 2.f * h * m_ * (A2[i][t] - v4 * b2);
            gs_ = 2.f * h * s_ * (A3[i][t] - v4 * (am + b2));
          } else {
            gs_ = signf(s_) * (A1[i][t] - z * gv2) * ep;
          }
          gmu += c * dm * ip2;
          gs_ += c * signf(s_) * (sg * ip2 - 1.f / sg);
          gm8[t] = gout * gmu;
          gs8[t] = gout * gs_;
          if constexpr (PRI) {
            if (valid) {
              acc_mp[i][t] += gout * c * (-dm * ip2);
              acc_sp[i][t] += gout * c * signf(ps[t]) * (1.f / prs - (sg * sg + dm * dm) * ip2 / prs);
            }
          }
        }
        if (valid) { st8(ge + 8 * kb, gm8); st8(ge + d + 8 * kb, gs8); }
      }
      if (lig == 0) {
        const float sw = fmaxf(fabsf(th.y), SIGMA_MIN);
        const float dm = th.x - pwm, ip2 = 1.f / (pws * pws);
        const float g0 = gv + c * dm * ip2;
        const float g1 = (CF ? 2.f * h * th.y * vv2 : signf(th.y) * gv * nb) + c * signf(th.y) * (sw * ip2 - 1.f / sw);
        *reinterpret_cast<float2*>(g_bias + 2 * (size_t)e) = make_float2(gout * g0, gout * g1);
        if constexpr (PRI) {
          acc_mw += gout * c * (-dm * ip2);
          acc_sw += gout * c * signf(pws_raw) * (1.f / pws - (sw * sw + dm * dm) * ip2 / pws);
        }
      }
    }
    if constexpr (PRI) {           // this workgroup's share of group g's prior gradients -> row blockIdx.x + g
      constexpr int DP = 8 * LPE * CPL;
      __syncthreads();
#pragma unroll
      for (int i = 0; i < CPL; ++i)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int k = 8 * (lig + i * LPE) + t;
          sh_acc[(size_t)grp * 2 * DP + k] = acc_mp[i][t];
          sh_acc[(size_t)grp * 2 * DP + DP + k] = acc_sp[i][t];
        }
      if (lig == 0) { sh_w[2 * grp] = acc_mw; sh_w[2 * grp + 1] = acc_sw; }
      __syncthreads();
      float* prow = prows + ((size_t)blockIdx.x + (size_t)g) * var_prow_len(d);
      for (int k = tid; k < 2 * d; k += BLOCK) {
        const int kk = k < d ? k : DP + (k - d);
        float t = 0.f;
        for (int q = 0; q < GPB; ++q) t += sh_acc[(size_t)q * 2 * DP + kk];
        prow[4 + k] = t;
      }
      if (tid < 2) {
        float t = 0.f;
        for (int q = 0; q < GPB; ++q) t += sh_w[2 * q + tid];
        prow[4 + 2 * d + tid] = t;
      }
      if (tid == 0) reinterpret_cast<int*>(prow)[0] = g;
    }
    e_lo = seg_hi;
    ++g;
    if (g >= a.G) g = a.G - 1;
  }
  if (nclamp != 0 && a.status) atomicAdd(a.status, nclamp);
}

// g_priors[2 ..] = sum over the partial rows of each group, in a fixed order.  The workgroups whose entity range
// meets group g are consecutive, so are their rows (workgroup + g).  Two launches: VAR_PSUM_CH chunks of the row range
// each summed by its own workgroup (eight rows in flight), then the chunks (a single pass over ~10^3 rows with four
// loads in flight took longer than the backward kernel itself).
constexpr int VAR_PSUM_CH = 32;

__global__ __launch_bounds__(BLOCK) void k_var_priors_part(const VarArgs a, const float* __restrict__ prows, int64_t epb,
                                                           int nblk, float* __restrict__ parts) {
  const int g = blockIdx.x, d = a.d, c = blockIdx.z;
  const size_t len = 4 + 2 * (size_t)d + 4;
  const int k = blockIdx.y * BLOCK + threadIdx.x;
  if (k >= 2 * d + 2) return;
  const int64_t lo = g > 0 ? a.group_hi[g - 1] : 0;
  int64_t hi = (g + 1 < a.G) ? a.group_hi[g] : a.T;
  if (hi > a.T) hi = a.T;
  float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (hi > lo) {
    const int64_t b0 = lo / epb;
    int64_t b1 = (hi - 1) / epb;
    if (b1 > nblk - 1) b1 = nblk - 1;
    const int64_t per = (b1 - b0 + VAR_PSUM_CH) / VAR_PSUM_CH;          // rows per chunk
    int64_t b = b0 + c * per;
    int64_t be = b + per - 1 < b1 ? b + per - 1 : b1;
    const float* base = prows + (size_t)g * len + 4 + k;
    for (; b + 7 <= be; b += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] += base[(size_t)(b + u) * len];
    }
    for (; b <= be; ++b) t[0] += base[(size_t)b * len];
  }
  parts[((size_t)g * VAR_PSUM_CH + c) * (2 * (size_t)d + 2) + k] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
}

__global__ __launch_bounds__(BLOCK) void k_var_priors_sum(const VarArgs a, const float* __restrict__ parts,
                                                          float* __restrict__ g_priors) {
  const int g = blockIdx.x, G = a.G, d = a.d;
  const int k = blockIdx.y * BLOCK + threadIdx.x;
  if (k >= 2 * d + 2) return;
  float t = 0.f;
#pragma unroll 8
  for (int c = 0; c < VAR_PSUM_CH; ++c) t += parts[((size_t)g * VAR_PSUM_CH + c) * (2 * (size_t)d + 2) + k];
  float* gm = g_priors + 2 + 2 * G;
  if (k < d) gm[(size_t)g * d + k] = t;
  else if (k < 2 * d) gm[(size_t)G * d + (size_t)g * d + (k - d)] = t;
  else if (k == 2 * d) g_priors[2 + g] = t;
  else g_priors[2 + G + g] = t;
}
