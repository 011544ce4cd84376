// vfm_fwd2m.hpp -- k_fwd2m: the task-stream forward (vfm_fwd2.hpp) with S = 2..4 variational samples handled INSIDE
// the kernel.  Included behind vfm_fwd2.hpp inside `namespace vfm { namespace {` of vfm_fwd2m.hip.
#pragma once

// The reference's global N_VARIATIONAL_SAMPLES (vfm-torch.py:19,238-245,265,359): per sample its own eps, the entity
// terms averaged over the samples BEFORE the likelihood while w0 is not:
//     pred[s,r] = w0^s + 1/S sum_s' (w_u^s' + w_i^s' + <z_u^s', z_i^s'>),
// the likelihood averaged over S*B, grow[r] = sum_s dloss/dpred[s,r], the KL term once.  k_fwd runs one launch per
// sample and re-gathers every table row S times; here a task gathers its row ONCE and loops over the samples in
// registers (the cached item holds S samples), with the run reuse of k_fwd2 on top.  Same outputs as the S launches
// (pred [S,B], sumz [S,B,d], grow, partial slots incl. VFM_P_GE0).
constexpr int FWD2M_MAXS = 4;

// weighted KL share of this lane for the entity in R (sample-independent)
template <bool FULL, int EPS, int LINK>
__device__ __forceinline__ float kl_ent(const EntRegs<EPS>& R, bool v0, bool v1, bool owns_bias, float cs) {
  float klv = 0.f;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const bool valid = FULL || (c == 0 ? v0 : v1);
    v2f kq = {0.f, 0.f};
    float lg = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const v2f m2 = {R.mu[c].v[2 * h], R.mu[c].v[2 * h + 1]};
      const v2f g2 = {link_f<LINK>(R.s[c].v[2 * h]), link_f<LINK>(R.s[c].v[2 * h + 1])};
      kq = g2 * g2 + kq;
      kq = m2 * m2 + kq;
      lg += __builtin_amdgcn_logf(fmaxf(g2.x, SIGMA_MIN) * fmaxf(g2.y, SIGMA_MIN));
    }
    klv += valid ? fmaf(0.5f, kq.x + kq.y, fmaf(-LN2, lg, -2.0f)) : 0.f;
  }
  klv += owns_bias ? kl_std_normal(R.th.x, link_f<LINK>(R.th.y)) : 0.f;
  return klv * (cs * R.io);
}

// sample s of the entity in R: z (this lane's 8 coordinates), sampled first-order weight
template <bool FULL, int EPS, int LINK>
__device__ __forceinline__ void sample_s(const KArgs& a, const EntRegs<EPS>& R, int s, uint32_t pg, int off0, int off1,
                                         bool v0, bool v1, bool owns_bias, float (&z)[8], float& w) {
  float ep[8], epw = 0.f;
  if constexpr (EPS == EPS_TABLE) {       // tables of sample s: the s-th [T,d] / [T] blocks (test path: loaded here)
    const float* er = a.eps_entity + ((size_t)s * (size_t)a.T + (size_t)R.e) * (size_t)a.d;
    const Chunk<4> e0c = ld_chunk<4>(er + off0), e1c = ld_chunk<4>(er + off1);
#pragma unroll
    for (int t = 0; t < 4; ++t) { ep[t] = e0c.v[t]; ep[4 + t] = e1c.v[t]; }
    epw = a.eps_bias[(size_t)s * (size_t)a.T + (size_t)R.e];
  } else if constexpr (EPS == EPS_PHILOX) {
    normal8b(key_of_sample(a.key, s), R.e, pg, ep, epw);
  } else {
#pragma unroll
    for (int t = 0; t < 8; ++t) ep[t] = 0.f;
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const bool valid = FULL || (c == 0 ? v0 : v1);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const v2f m2 = {R.mu[c].v[2 * h], R.mu[c].v[2 * h + 1]};
      const v2f g2 = {link_f<LINK>(R.s[c].v[2 * h]), link_f<LINK>(R.s[c].v[2 * h + 1])};
      const v2f e2 = {ep[4 * c + 2 * h], ep[4 * c + 2 * h + 1]};
      const v2f z2 = g2 * e2 + m2;
      z[4 * c + 2 * h] = valid ? z2.x : 0.f;
      z[4 * c + 2 * h + 1] = valid ? z2.y : 0.f;
    }
  }
  w = owns_bias ? fmaf(link_f<LINK>(R.th.y), epw, R.th.x) : 0.f;
}

template <int LPE, bool FULL, int EPS, int MODE, bool ID64, int LINK>
__global__ __launch_bounds__(BLOCK) void k_fwd2m(const KArgs a, const FwdOut out) {
  constexpr int GPB = BLOCK / LPE;
  __shared__ float sh_cs[2];
  __shared__ int64_t sh_hi[2];
  __shared__ float sh_red[6 * 4];
  __shared__ float sh_e0[FWD2M_MAXS];

  const int tid = threadIdx.x;
  const int lig = tid % LPE;
  const int C = a.d >> 2;
  const int S = a.S;                            // 2 .. FWD2M_MAXS (checked by the launcher)

  if (MODE == MODE_TRAIN && tid < 2) {
    sh_cs[tid] = (float)(a.group_n[tid] / a.W[tid]);
    sh_hi[tid] = a.group_hi[tid];
  }
  if constexpr (EPS == EPS_PHILOX) {            // the global-bias eps of every sample: one wave draws them
    if (tid < 64) {
      for (int s = 0; s < S; ++s) {
        float n[8], nb;
        normal8b(key_of_sample(a.key, s), 0xFFFFFFFFu, 0u, n, nb);
        if (tid == 0) sh_e0[s] = n[0];
      }
    }
  } else if (tid < S) {
    sh_e0[tid] = (EPS == EPS_TABLE) ? a.eps_global[tid] : 0.f;
  }
  __syncthreads();
  const float alpha = a.scalars[0], m0 = a.scalars[1], s0 = a.scalars[2];
  const float aabs = link_f<LINK>(alpha), sg0 = link_f<LINK>(s0);
  const float half_log_a = 0.5f * LN2 * __builtin_amdgcn_logf(aabs);
  const bool owns_bias = lig == 0;
  float cs0 = 0.f, cs1 = 0.f;
  uint32_t hi0 = 0u;
  if constexpr (MODE == MODE_TRAIN) {
    cs0 = sh_cs[0]; cs1 = sh_cs[1];
    hi0 = sh_hi[0] > 0xFFFFFFFFLL ? 0xFFFFFFFFu : (uint32_t)sh_hi[0];
  }
  const uint32_t T32 = (uint32_t)a.T;
  const int j0 = 2 * lig, j1 = 2 * lig + 1;
  const bool v0 = j0 < C, v1 = j1 < C;
  const int off0 = 4 * (v0 ? j0 : C - 1), off1 = 4 * (v1 ? j1 : C - 1);
  const uint32_t pg = (uint32_t)lig + (a.key.chunk_off >> 1);

  const int NG = (int)gridDim.x * GPB;
  const int gid = (int)blockIdx.x * GPB + tid / LPE;
  const int q = (int)(a.B / NG), rem = (int)(a.B % NG);
  const int gbeg = gid * q + (gid < rem ? gid : rem);
  const int gend = gbeg + q + (gid < rem ? 1 : 0);
  const int glast = gend - 1;

  float tot[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // ll, kl, g, alpha-term, bad ids, sum_s eps0^s * g_s
  if (gbeg < gend) {
    int r0 = gbeg;
    Ids id0 = fold_ids<ID64>(load_row_ids<ID64>(a, r0), T32);
    Ids id1 = fold_ids<ID64>(load_row_ids<ID64>(a, r0 + 1 < gend ? r0 + 1 : glast), T32);
    RawIds<ID64> id2 = load_row_ids<ID64>(a, r0 + 2 < gend ? r0 + 2 : glast);
    bool cur_item = true;
    EntRegs<EPS> A, Bq;
    load_ent<EPS, MODE>(a, entity_of(id0.i, true, tot[4]), off0, off1, A);
    float ycur = 0.f;
    if constexpr (MODE == MODE_TRAIN) ycur = a.y[r0];
    float zi[FWD2M_MAXS][8], wi[FWD2M_MAXS], klwi = 0.f;      // the cached item: S samples
#pragma unroll
    for (int s = 0; s < FWD2M_MAXS; ++s) {
      wi[s] = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) zi[s][t] = 0.f;
    }

    auto step = [&](const EntRegs<EPS>& cu, EntRegs<EPS>& nx) -> bool {
      const bool item_now = cur_item;
      const int r_now = r0;
      const float y_now = ycur;
      const bool adv = !item_now;
      const uint32_t n_u = adv ? id1.u : id0.u, n_i = adv ? id1.i : id0.i;
      const int r_next = r0 + (adv ? 1 : 0);
      const bool live_next = r_next < gend;
      const bool item_next = adv && (n_i != id0.i);
      const uint32_t e_next = entity_of(item_next ? n_i : n_u, live_next, tot[4]);
      const Ids f2 = fold_ids<ID64>(id2, T32);
      id2 = load_row_ids<ID64>(a, r_next + 2 < gend ? r_next + 2 : glast);
      load_ent<EPS, MODE>(a, e_next, off0, off1, nx);
      if constexpr (MODE == MODE_TRAIN) ycur = a.y[live_next ? r_next : glast];
      id0.u = n_u; id0.i = n_i;
      id1.u = adv ? f2.u : id1.u;
      id1.i = adv ? f2.i : id1.i;
      r0 = r_next;
      cur_item = item_next;
      // ---- current task: the S samples of its entity ----
      float klw = 0.f;
      if constexpr (MODE == MODE_TRAIN) klw = kl_ent<FULL, EPS, LINK>(cu, v0, v1, owns_bias, (cu.e < hi0) ? cs0 : cs1);
      if (item_now) klwi = klw;
      float vsum = 0.f;                              // sum over samples of the row value (user task)
#pragma unroll
      for (int s = 0; s < FWD2M_MAXS; ++s) {
        if (s < S) {                                 // (uniform)
          float z[8], w;
          sample_s<FULL, EPS, LINK>(a, cu, s, pg, off0, off1, v0, v1, owns_bias, z, w);
          if (item_now) {
#pragma unroll
            for (int t = 0; t < 8; ++t) zi[s][t] = z[t];
            wi[s] = w;
          } else {
            v2f qv = {0.f, 0.f};
#pragma unroll
            for (int h = 0; h < 4; ++h) {
              const v2f zu = {z[2 * h], z[2 * h + 1]};
              const v2f zv = {zi[s][2 * h], zi[s][2 * h + 1]};
              qv = zu * zv + qv;
            }
            vsum += qv.x + qv.y + w + wi[s];
            if constexpr (MODE == MODE_TRAIN) {
              float* srow = out.sumz + ((size_t)s * (size_t)a.B + (size_t)r_now) * a.d;
              Chunk<4> s0c, s1c;
#pragma unroll
              for (int t = 0; t < 4; ++t) { s0c.v[t] = z[t] + zi[s][t]; s1c.v[t] = z[4 + t] + zi[s][4 + t]; }
              if (FULL || v0) st_chunk<4>(srow + off0, s0c);
              if (FULL || v1) st_chunk<4>(srow + off1, s1c);
            }
          }
        }
      }
      if (!item_now) {
        const float m = group_sum<LPE>(vsum) * a.inv_S;
        if constexpr (MODE == MODE_TRAIN) tot[1] += klw + klwi;
        if (lig == 0) {
          float gsum = 0.f;
          for (int s = 0; s < S; ++s) {
            const float e0 = sh_e0[s];
            const float pred = fmaf(sg0, e0, m0) + m;
            out.pred[(size_t)s * (size_t)a.B + (size_t)r_now] = pred;
            if constexpr (MODE == MODE_TRAIN) {
              float ll, dll, at;
              lik_terms(a.lik, y_now, pred, aabs, half_log_a, ll, dll, at);
              const float g = -a.ll_scale * dll;
              tot[0] += ll; tot[2] += g; tot[3] += at;
              tot[5] = fmaf(e0, g, tot[5]);
              gsum += g;
            }
          }
          if constexpr (MODE == MODE_TRAIN) out.grow[r_now] = gsum;
        }
      }
      return live_next;
    };

    while (true) {
      if (!step(A, Bq)) break;
      if (!step(Bq, A)) break;
    }
  }
  block_sum<6>(tot, sh_red);
  if (tid == 0) {
    double* slot = out.partials + VFM_N_PARTIALS * (1 + (size_t)blockIdx.x);
#pragma unroll
    for (int i = 0; i < 6; ++i) slot[i] = (double)tot[i];
    slot[VFM_SLOT_NTERMS] = slot_nterms(a, MODE == MODE_TRAIN);
    if (blockIdx.x == 0) {
      out.partials[7] = (double)gridDim.x;
      out.partials[VFM_P_REDUCED] = 0.0;
    }
  }
}
