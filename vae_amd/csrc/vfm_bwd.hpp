// vfm_bwd.hpp -- k_bwd: entity-centric gradients / fused Adam / multi-rank stages.
// Included inside `namespace vfm { namespace {` of vfm_bwd.hip (one object per link function).
#pragma once

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
//
// Variational samples S > 1 (STAGE_FULL only): sumz holds one [B,d] block per sample and grow[r] the
// sum over samples of dloss/dpred[s,r]; with A^s = sum_r grow_r sumz^s_r and gs = sum_r grow_r,
//   dloss/dmu_e = 1/S sum_s (A^s - z^s gs),   dloss/ds_e = link'(s_e) 1/S sum_s eps^s (A^s - z^s gs)
// (+ the KL part, which does not depend on the sample): the list of e is walked once per sample.
// Scaled moments (VFM_FLAG_SCALED_MOMENTS): with ms = m / b1^k and vs = v / b2^k stored instead of m and
// v (k = steps since the last period boundary), the decay of a row WITHOUT gradient is implicit -- its ms
// and vs do not change, so they are read but not written back: 16 instead of 24 bytes per parameter for
// the rows a batch does not touch.  Same dense-Adam mathematics (every row still moves every step):
//   ms += (1-b1) g / b1^k,  vs += (1-b2) g^2 / b2^k,  m = ms b1^k,  v = vs b2^k,  p -= lr_t m / (sqrt(v)/.. + eps)
// At the end of a period of VFM_MOMENT_PERIOD steps the true m, v are written for every row (k restarts),
// which bounds 1 / b1^k (0.9^-128 = 7e5).
__device__ __forceinline__ float adam_update(float p, float g, float& m, float& v, const AdamArgs& ad) {
  if (ad.scaled) {                     // uniform
    // (this form is not bitwise torch's anyway: hardware sqrt / rcp, 1 ulp each, instead of the IEEE
    // sqrt and the two IEEE divisions of the plain form below -- ~10 instead of ~45 VALU instructions
    // per coordinate, which the kernel would otherwise not hide behind its memory traffic)
    // p -= step_size m_t / (sqrt(v_t) / sqrt(bc2) + eps) with m_t = ms b1^k, v_t = vs b2^k, the per-step factors folded
    // into two scalars (a1, q2): sqrt is taken of the STORED second moment, so a row without gradient needs it once
    // however many steps are replayed (k_adam_catchup runs exactly these operations)
    m = fmaf(ad.c1, g, m);
    v = fmaf(ad.c2 * g, g, v);
    const float denom = fmaf(__builtin_amdgcn_sqrtf(v), ad.q2, ad.eps);
    const float pn = fmaf(-ad.a1 * m, __builtin_amdgcn_rcpf(denom), p);
    if (ad.store_true) { m = m * ad.s1; v = v * ad.s2; }
    return pn;
  }
  m = m + (g - m) * (1.0f - ad.b1);
  v = v * ad.b2 + ((1.0f - ad.b2) * g) * g;
  const float denom = __fsqrt_rn(v) / ad.bc2_sqrt + ad.eps;
  return p + (-ad.step_size * m) / denom;
}

__device__ __forceinline__ int heavy_slot_of(const int32_t* ids, int n, int e) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (ids[mid] < e) lo = mid + 1; else hi = mid;
  }
  return (n > 0 && ids[lo] == e) ? lo : -1;
}

// one replayed zero-gradient step of the scaled-moment form, given r = sqrt(stored second moment) and the step's
// (a1, q2): exactly adam_update's scaled branch with g = 0 (k_adam_catchup, vfm_adam.hpp, runs the same operations)
__device__ __forceinline__ float replay_one(float p, float m, float r, float2 c, float eps) {
  return fmaf(-c.x * m, __builtin_amdgcn_rcpf(fmaf(r, c.y, eps)), p);
}

template <int LPE, int CPL, int VEC, int EPS, int ADAM, int STAGE, int LINK, bool MULTI, bool PIPE = false, bool LA = false>
__global__ __launch_bounds__(BLOCK, (PIPE && CPL == 1) ? 4 : 1) void k_bwd(const KArgs a, const BwdArgs b, const AdamArgs ad_in) {
  constexpr int GPB = BLOCK / LPE;
  // step-dependent values: from the kernel arguments, or (replayable step, a.dev) from device memory
  AdamArgs ad = ad_in;
  RngKey key = a.key, next_key = b.next_key;
  int32_t la_step = b.la_step, la_k = b.la_k;
  if constexpr (ADAM == 1 && STAGE == STAGE_FULL)
    load_dev_step(a, key, ad, la_step, la_k, next_key,
                  blockIdx.x == 0 && threadIdx.x == 0 && a.row_filter != 3 && a.e_hi == a.T);
  static_assert(!PIPE || (ADAM == 1 && STAGE == STAGE_FULL && !MULTI), "the pipelined step is the fused single-sample one");
  static_assert(!LA || (ADAM == 1 && STAGE == STAGE_FULL && !MULTI), "look-ahead lazy Adam is a form of the fused dense step");
  __shared__ float sh_cs[VFM_MAX_FIELDS];
  __shared__ float sh_cs_next[PIPE ? VFM_MAX_FIELDS : 1];
  __shared__ int64_t sh_hi[VFM_MAX_FIELDS];
  __shared__ double sh_fin[7][BLOCK / 64];
  __shared__ float2 sh_tab[LA ? VFM_MOMENT_PERIOD + 1 : 1];      // LA: (a1, q2) of the period's earlier steps, for replays
  const int tid = threadIdx.x;
  const int lig = tid % LPE;
  const int d = a.d;
  const int C = (d + VEC - 1) / VEC;
  if (STAGE != STAGE_ACC && tid < a.G) {
    sh_cs[tid] = (float)(a.group_n[tid] / a.W[tid]);
    sh_hi[tid] = a.group_hi[tid];
    if constexpr (PIPE) sh_cs_next[tid] = b.zrec_next ? (float)(a.group_n[tid] / b.next_W[tid]) : 0.f;
  }
  if constexpr (LA) {
    for (int k = tid; k < la_k; k += BLOCK) sh_tab[k] = b.step_tab[k];
  }
  __syncthreads();
  if constexpr (LA) {
    if (blockIdx.x == 0 && tid == 0) b.step_tab[la_k] = make_float2(ad.a1, ad.q2);     // for later replays of this step
  }
  const float gout = (ADAM || STAGE == STAGE_ACC) ? 1.0f : b.grad_out[0];

  double fin[6] = {0, 0, 0, 0, 0, 0};
  // a.row_filter (fused Adam, STAGE_FULL): 0 = every row; 2 = only the rows of the batch (+ the scalars and the loss);
  // 3 / 4 (the long-list pre-reduction overlapped with this kernel, vfm_abi.hip): 4 = every row but the heavy
  // entities (+ the scalars and the loss), 3 = the heavy entities only, listed
  const bool duty = a.row_filter != 3;      // this launch forms the loss and moves the scalars
  const bool fold = STAGE == STAGE_FULL && b.loss != nullptr && duty;   // uniform: fold vfm_elbo_finalize_f32 in
  if (blockIdx.x == 0 && fold)
    reduce_slots_and_loss(b.partials, a.scalars, a.ll_scale_d, a.flags, b.loss, sh_fin, fin);
  if (STAGE == STAGE_ACC && blockIdx.x == 0 && tid == 0 && a.e_lo == 0) {
    // (a caller that skipped vfm_elbo_finalize_f32 gets NaN, not stale sums)
    const bool reduced = b.partials[VFM_P_REDUCED] == 1.0;
    b.sums[0] = reduced ? (float)b.partials[VFM_P_G] : __builtin_nanf("");   // this rank's row sums, to be summed over ranks
    b.sums[1] = reduced ? (float)b.partials[VFM_P_ALPHA] : __builtin_nanf("");
  }
  if (STAGE != STAGE_ACC && blockIdx.x == 0 && tid == 0 && a.e_hi == a.T && duty) {   // (last chunk of a chunked run)
    const float alpha = a.scalars[0], m0 = a.scalars[1], s0 = a.scalars[2];
    // (no fold and the forward's slots never reduced -- vfm_elbo_finalize_f32 skipped --: NaN, not stale sums)
    const bool stale = STAGE == STAGE_FULL && !fold && b.partials[VFM_P_REDUCED] != 1.0;
    const float sum_g = stale ? __builtin_nanf("")
                              : (STAGE == STAGE_APPLY) ? b.sums[0] : (float)(fold ? fin[VFM_P_G] : b.partials[VFM_P_G]);
    const float sum_a = (STAGE == STAGE_APPLY) ? b.sums[1]
                                               : (float)(fold ? fin[VFM_P_ALPHA] : b.partials[VFM_P_ALPHA]);
    float e0 = 0.f;
    if constexpr (EPS == EPS_TABLE) e0 = a.eps_global[0];
    if constexpr (EPS == EPS_PHILOX) {
      float n[8], nb;
      normal8b(key, 0xFFFFFFFFu, 0u, n, nb);
      e0 = n[0];
    }
    const float as0 = link_f<LINK>(s0);
    const float prior = (a.flags & VFM_FLAG_NO_PRIOR_TERMS) ? 0.f : 1.f;
    const float ga = (a.lik == VFM_LIK_NORMAL)
                         ? gout * dlink_f<LINK>(alpha) * a.ll_scale * sum_a : 0.f;
    const float gm = gout * (sum_g + prior * m0);
    // S > 1: every sample has its own eps0 -- the forward accumulated sum_s eps0^s sum_r g_sr
    const float ge0 = MULTI ? (float)(fold ? fin[VFM_P_GE0] : b.partials[VFM_P_GE0]) : e0 * sum_g;
    const float gs = gout * dlink_f<LINK>(s0) * (ge0 + prior * (as0 - inv_sigma(as0)));
    if constexpr (ADAM) {
      float* sc = const_cast<float*>(a.scalars);
      auto upd = [&](int i, float g) {
        float m = ad.m_scal[i], v = ad.v_scal[i];
        sc[i] = adam_update(sc[i], g, m, v, ad);
        ad.m_scal[i] = m; ad.v_scal[i] = v;
      };
      // alpha has no gradient under the Bernoulli likelihood (reference: grad None, Adam skips it)
      if (a.lik == VFM_LIK_NORMAL) upd(0, ga);
      upd(1, gm);
      upd(2, gs);
    } else {
      b.g_scalars[0] = ga; b.g_scalars[1] = gm; b.g_scalars[2] = gs;
    }
  }

  const int64_t stride = (int64_t)gridDim.x * GPB;
  const int64_t xs = 4 + (((int64_t)d + 3) & ~(int64_t)3);          // floats per exchange record
  // fused Adam with a row list (the lazy exact-Adam step): li runs over the list (the batch's entities), not over
  // the table.  The SAME instance serves the dense step, so the two agree bit for bit on the rows they share
  // (different template instances are compiled with different fma contractions).
  // (STAGE_APPLY with a list -- vfm_elbo_apply_adam_rows_f32: the multi-rank step's lazy exact form -- visits the
  // listed rows only; their records sit in the DENSE statistics table, at the entity's own index)
  // (the multi-rank stages take a list too -- vfm_elbo_bwd_acc_rows_f32 / vfm_elbo_apply_adam_rows_f32: the rows some
  // rank's shard contains; their records sit in the DENSE statistics table at the entity's own index, or, with
  // b.rec_by_slot, in a COMPACT buffer at the row's position in the list)
  const bool listed = b.row_ids != nullptr && (STAGE == STAGE_FULL ? ADAM != 0 : true);
  // a corrupted index is clamped, never followed: list offsets to [0, n_occ], row numbers to [0, B), listed entities to
  // [0, T); every clamp that fires is counted in b.status (vfm_index_t.status) for the caller's next look
  int nclamp = 0;
  const int n_occ = b.n_occ;
  const int Bm1 = a.B > 0 ? (int)a.B - 1 : 0;
  auto ent_ok = [&](int64_t v) -> int64_t {
    const bool ok = v >= 0 && v < a.T;
    nclamp += ok ? 0 : 1;
    return ok ? v : 0;
  };
  auto span_ok = [&](int2 v) -> int2 {
    const bool ok = v.x >= 0 && v.y >= v.x && v.y <= n_occ;
    nclamp += ok ? 0 : 1;
    return ok ? v : make_int2(0, 0);
  };
  auto row_ok = [&](int v) -> int {
    const bool ok = (unsigned)v <= (unsigned)Bm1;
    nclamp += ok ? 0 : 1;
    return ok ? v : 0;
  };
  const int64_t li_end = listed ? b.n_rows : a.e_hi;
  int64_t li = (listed ? 0 : a.e_lo) + (int64_t)blockIdx.x * GPB + tid / LPE;
  int64_t e_cur = li;
  if (listed && li < li_end) e_cur = ent_ok(b.row_ids[li]);
  int2 pq = make_int2(0, 0);
  if (STAGE != STAGE_APPLY && li < li_end) pq = span_ok(make_int2(b.occ_ptr[e_cur], b.occ_ptr[e_cur + 1]));
  for (; li < li_end; li += stride) {
    const int64_t e = listed ? e_cur : li;
    const int64_t rec = (listed && !b.rec_by_slot) ? e : li;      // where this row's statistics record sits (multi-rank stages)
    int beg = pq.x, end = pq.y;
    const int64_t en = li + stride;
    float2 gc = make_float2(0.f, 0.f);
    if constexpr (STAGE == STAGE_APPLY) {
      gc = *reinterpret_cast<const float2*>(b.acc + (size_t)rec * xs);   // (sum of grow, occurrences) over ALL ranks
      if (listed && en < li_end) e_cur = ent_ok(b.row_ids[en]);
      beg = 0; end = 0;
    } else {
      if (en < li_end) {                                                    // next entity's offsets, early
        const int64_t e_next = listed ? ent_ok(b.row_ids[en]) : en;
        e_cur = e_next;
        pq = span_ok(make_int2(b.occ_ptr[e_next], b.occ_ptr[e_next + 1]));
      }
    }
    bool in_next = false;          // PIPE: e is in the next batch -> its next-step record is written below
    if constexpr (PIPE) {
      if (b.zrec_next) in_next = b.next_occ_ptr[e + 1] != b.next_occ_ptr[e];
    }
    float* prow = const_cast<float*>(a.entity) + (size_t)e * (2 * (size_t)d);
    float* grow_e = (ADAM || STAGE == STAGE_ACC) ? nullptr : b.g_entity + (size_t)e * (2 * (size_t)d);
    const bool touched = (STAGE == STAGE_APPLY) ? gc.y > 0.f : beg != end;
    const float cntf = (STAGE == STAGE_APPLY) ? gc.y : (float)(end - beg);
    int la_gap = 0;                        // LA: skipped zero-gradient steps this row applies before this step's update
    if constexpr (LA) {
      if (!touched && b.next_occ_ptr[e + 1] == b.next_occ_ptr[e]) continue;     // in neither batch: the row waits
      la_gap = (la_step - 1) - b.last_step[e];
    }
    if (ADAM == 2 && !touched) continue;   // opt-in row-sparse Adam: rows not in the batch stay as they are
    if (ADAM == 1 && STAGE == STAGE_FULL && a.row_filter == 2 && !touched) continue;

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
      if (touched || (PIPE && in_next)) {
        io = a.inv_occ[e];
        if constexpr (EPS == EPS_TABLE) epw = a.eps_bias[e];
      }
    }

    // walk the inverted index: A = sum_r g_r * sumz_r, gs = sum_r g_r  (sz / hacc: the sample's blocks)
    int hslot = -1;
    if (STAGE != STAGE_APPLY && b.n_heavy > 0 && end - beg > VFM_HEAVY_MIN)
      hslot = heavy_slot_of(b.heavy_ids, b.n_heavy, (int)e);
    if (ADAM == 1 && STAGE == STAGE_FULL && a.row_filter == 4 && hslot >= 0) continue;   // (the heavy-only launch takes it)
    auto walk = [&](const float* __restrict__ sz, const float* __restrict__ hacc, Chunk<VEC>(&A)[CPL], float& gs) {
#pragma unroll
      for (int i = 0; i < CPL; ++i)
#pragma unroll
        for (int t = 0; t < VEC; ++t) A[i].v[t] = 0.f;
      gs = 0.f;
      int o = beg;
      if (hslot >= 0) {      // pre-reduced by k_heavy: read the record(s), skip the walk
        const float* rec = hacc + (size_t)hslot * xs;
        int4 hd = *reinterpret_cast<const int4*>(rec);      // (sum grow, count, first item, last item + 1)
        if (hd.z < 0 || hd.w < hd.z || hd.w > b.heavy_stride - b.n_heavy) { hd.z = hd.w = 0; ++nclamp; }
        if (hd.w - hd.z <= VFM_HEAVY_DIRECT) {      // few work items: add their records here, in item order
          for (int it = hd.z; it < hd.w; ++it) {
            const float* ir = hacc + ((size_t)b.n_heavy + (size_t)it) * xs;
            gs += ir[0];
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
              const int j = lig + i * LPE;
              if (j < C) {
                const Chunk<VEC> t4 = ld_chunk<VEC>(ir + 4 + (size_t)j * VEC);
#pragma unroll
                for (int t = 0; t < VEC; ++t) A[i].v[t] += t4.v[t];
              }
            }
          }
        } else {                                    // k_heavy_sum added them into the entity's record
          gs = __int_as_float(hd.x);
#pragma unroll
          for (int i = 0; i < CPL; ++i) {
            const int j = lig + i * LPE;
            if (j < C) A[i] = ld_chunk<VEC>(rec + 4 + (size_t)j * VEC);
          }
        }
        o = end;
      }
      // PIPE: the sample of the OTHER entity of the row (this step's records) stands in for the sumz row
      auto src = [&](int oo, int r) -> const float* {
        if constexpr (PIPE) return b.zrec + (size_t)ent_ok(b.occ_other[oo]) * xs + 4;
        return sz + (size_t)r * d;
      };
      for (; o + 1 < end; o += 2) {       // two occurrences in flight
        const int r0 = row_ok(b.occ_rows[o]), r1 = row_ok(b.occ_rows[o + 1]);
        const float g0 = b.grow[r0], g1 = b.grow[r1];
        const float* p0 = src(o, r0);
        const float* p1 = src(o + 1, r1);
        gs += g0 + g1;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          const int j = lig + i * LPE;
          if (j < C) {
            const Chunk<VEC> s0v = ld_chunk<VEC>(p0 + (size_t)j * VEC);
            const Chunk<VEC> s1v = ld_chunk<VEC>(p1 + (size_t)j * VEC);
#pragma unroll
            for (int t = 0; t < VEC; ++t) A[i].v[t] = fmaf(g1, s1v.v[t], fmaf(g0, s0v.v[t], A[i].v[t]));
          }
        }
      }
      if (o < end) {
        const int r0 = row_ok(b.occ_rows[o]);
        const float g0 = b.grow[r0];
        const float* p0 = src(o, r0);
        gs += g0;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          const int j = lig + i * LPE;
          if (j < C) {
            const Chunk<VEC> s0v = ld_chunk<VEC>(p0 + (size_t)j * VEC);
#pragma unroll
            for (int t = 0; t < VEC; ++t) A[i].v[t] = fmaf(g0, s0v.v[t], A[i].v[t]);
          }
        }
      }
    };
    Chunk<VEC> A[CPL];
    float gs;
    walk(b.sumz, b.heavy_acc, A, gs);

    if constexpr (STAGE == STAGE_ACC) {   // store the statistics (dense: zeros for rows not in this shard)
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int j = lig + i * LPE;
        if (j < C) st_chunk<VEC>(b.acc + (size_t)rec * xs + 4 + (size_t)j * VEC, A[i]);
      }
      if (lig == 0) *reinterpret_cast<float4*>(b.acc + (size_t)rec * xs) = make_float4(gs, cntf, 0.f, 0.f);
      continue;
    }
    if constexpr (STAGE == STAGE_APPLY) {
      gs = gc.x;
      if (touched) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          const int j = lig + i * LPE;
          if (j < C) A[i] = ld_chunk<VEC>(b.acc + (size_t)rec * xs + 4 + (size_t)j * VEC);
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
    float nb_next = 0.f, kl_next = 0.f, w_next = 0.f;      // PIPE: the next step's first-order eps / KL / sampled weight
    // S > 1 (uniform): g1s = 1/S sum_s (A^s - z^s gs), g2s = 1/S sum_s eps^s (A^s - z^s gs), nb_eps = mean_s eps_w^s
    // (MULTI is a template parameter so that the S = 1 instances carry none of this: registers, occupancy)
    constexpr bool multi = MULTI && STAGE == STAGE_FULL;
    Chunk<VEC> g1s[CPL], g2s[CPL];
    if (multi && touched) {
#pragma unroll
      for (int i = 0; i < CPL; ++i)
#pragma unroll
        for (int t = 0; t < VEC; ++t) { g1s[i].v[t] = 0.f; g2s[i].v[t] = 0.f; }
      for (int sm = 0; sm < a.S; ++sm) {
        if (sm > 0)
          walk(b.sumz + (size_t)sm * (size_t)a.B * d, b.heavy_acc + (size_t)sm * (size_t)b.heavy_stride * xs, A, gs);
        const RngKey ks = key_of_sample(key, sm);
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          const int j = lig + i * LPE;
          if (j < C) {
            Chunk<VEC> epc;
            if constexpr (EPS == EPS_TABLE) {
              epc = ld_chunk<VEC>(a.eps_entity + ((size_t)sm * (size_t)a.T + (size_t)e) * d + (size_t)j * VEC);
            } else if constexpr (EPS == EPS_ZERO) {
#pragma unroll
              for (int t = 0; t < VEC; ++t) epc.v[t] = 0.f;
            } else {
              float nb;
              eps_of_chunk<VEC>(ks, (uint32_t)e, j, epc.v, nb);
              if (i == 0) nb_eps += nb;
            }
#pragma unroll
            for (int t = 0; t < VEC; ++t) {
              const float z = fmaf(link_f<LINK>(s[i].v[t]), epc.v[t], mu[i].v[t]);
              const float gz = A[i].v[t] - z * gs;
              g1s[i].v[t] += gz;
              g2s[i].v[t] = fmaf(gz, epc.v[t], g2s[i].v[t]);
            }
          }
        }
        if constexpr (EPS == EPS_TABLE) nb_eps += a.eps_bias[(size_t)sm * (size_t)a.T + (size_t)e];
      }
      nb_eps *= a.inv_S;
#pragma unroll
      for (int i = 0; i < CPL; ++i)
#pragma unroll
        for (int t = 0; t < VEC; ++t) { g1s[i].v[t] *= a.inv_S; g2s[i].v[t] *= a.inv_S; }
    }
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int j = lig + i * LPE;
      if (j < C) {
        Chunk<VEC> gm, gv;
        if (touched && multi) {
#pragma unroll
          for (int t = 0; t < VEC; ++t) {
            const float sg = link_f<LINK>(s[i].v[t]);
            gm.v[t] = gout * (g1s[i].v[t] + c * mu[i].v[t]);
            gv.v[t] = gout * dlink_f<LINK>(s[i].v[t]) * (g2s[i].v[t] + c * (sg - inv_sigma(sg)));
          }
        } else if (touched) {
          Chunk<VEC> epc;
          if constexpr (EPS == EPS_TABLE) {
            epc = ep[i];
          } else if constexpr (EPS == EPS_ZERO) {
#pragma unroll
            for (int t = 0; t < VEC; ++t) epc.v[t] = 0.f;
          } else {
            float nb;
            eps_of_chunk<VEC>(key, (uint32_t)e, j, epc.v, nb);
            if (i == 0) nb_eps = nb;
          }
#pragma unroll
          for (int t = 0; t < VEC; ++t) {
            const float sg = link_f<LINK>(s[i].v[t]);
            const float z = fmaf(sg, epc.v[t], mu[i].v[t]);
            // sum_r g_r (sumz_rk - z_ek); PIPE: A already sums the other entity's z alone
            const float gz = PIPE ? A[i].v[t] : A[i].v[t] - z * gs;
            gm.v[t] = gout * (gz + c * mu[i].v[t]);
            gv.v[t] = gout * dlink_f<LINK>(s[i].v[t]) * (gz * epc.v[t] + c * (sg - inv_sigma(sg)));
          }
        } else {
#pragma unroll
          for (int t = 0; t < VEC; ++t) { gm.v[t] = 0.f; gv.v[t] = 0.f; }
        }
        if constexpr (LA) {
          if (la_gap > 0) {            // (uniform over the lane group) bring the row up to the step before this one
            float rm[VEC], rs[VEC];
#pragma unroll
            for (int t = 0; t < VEC; ++t) { rm[t] = __builtin_amdgcn_sqrtf(vm[i].v[t]); rs[t] = __builtin_amdgcn_sqrtf(vs[i].v[t]); }
            for (int k = la_k - la_gap; k < la_k; ++k) {
              const float2 c = sh_tab[k];
#pragma unroll
              for (int t = 0; t < VEC; ++t) {
                mu[i].v[t] = replay_one(mu[i].v[t], mm[i].v[t], rm[t], c, ad.eps);
                s[i].v[t] = replay_one(s[i].v[t], ms[i].v[t], rs[t], c, ad.eps);
              }
            }
          }
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
          if (!ad.scaled || touched || ad.store_true) {     // (scaled: rows without gradient keep ms, vs)
            st_chunk_nt<VEC>(ad.m_entity + o2, mm[i]); st_chunk_nt<VEC>(ad.m_entity + o2 + d, ms[i]);
            st_chunk_nt<VEC>(ad.v_entity + o2, vm[i]); st_chunk_nt<VEC>(ad.v_entity + o2 + d, vs[i]);
          }
          if constexpr (PIPE) {
            if (in_next) {     // the updated row is in registers: sample it for the next step right here
              Chunk<VEC> ep2, zn;
              float nb2;
              eps_of_chunk<VEC>(next_key, (uint32_t)e, j, ep2.v, nb2);
              if (i == 0) nb_next = nb2;
#pragma unroll
              for (int t = 0; t < VEC; ++t) {
                const float sg2 = link_f<LINK>(ps.v[t]);
                zn.v[t] = fmaf(sg2, ep2.v[t], pm.v[t]);
                kl_next += kl_std_normal(pm.v[t], sg2);
              }
              st_chunk<VEC>(b.zrec_next + (size_t)e * xs + 4 + (size_t)j * VEC, zn);
            }
          }

        } else {
          st_chunk_nt<VEC>(grow_e + (size_t)j * VEC, gm);
          st_chunk_nt<VEC>(grow_e + d + (size_t)j * VEC, gv);
        }
      }
    }
    if (lig == 0) {
      float g0 = 0.f, g1 = 0.f;
      if (touched) {
        if constexpr (EPS == EPS_TABLE)
          if (!multi) nb_eps = epw;
        const float sg = link_f<LINK>(th.y);
        g0 = gout * (gs + c * th.x);
        g1 = gout * dlink_f<LINK>(th.y) * (gs * nb_eps + c * (sg - inv_sigma(sg)));
      }
      if constexpr (LA) {
        if (la_gap > 0) {
          const float r0 = __builtin_amdgcn_sqrtf(vb.x), r1 = __builtin_amdgcn_sqrtf(vb.y);
          for (int k = la_k - la_gap; k < la_k; ++k) {
            const float2 c = sh_tab[k];
            th.x = replay_one(th.x, mb.x, r0, c, ad.eps);
            th.y = replay_one(th.y, mb.y, r1, c, ad.eps);
          }
        }
        b.last_step[e] = la_step;
      }
      if constexpr (ADAM) {
        float2 pn;
        pn.x = adam_update(th.x, g0, mb.x, vb.x, ad);
        pn.y = adam_update(th.y, g1, mb.y, vb.y, ad);
        *reinterpret_cast<float2*>(const_cast<float*>(a.bias) + 2 * (size_t)e) = pn;
        if (a.wrec) *reinterpret_cast<float2*>(a.wrec + 4 * (size_t)e) = pn;      // packed first-order record: (mu_w, s_w | 1/occ, 0)
        if constexpr (PIPE) {
          if (in_next) {
            const float sgw2 = link_f<LINK>(pn.y);
            w_next = fmaf(sgw2, nb_next, pn.x);
            kl_next += kl_std_normal(pn.x, sgw2);
          }
        }
        if (!ad.scaled || touched || ad.store_true) {
          *reinterpret_cast<float2*>(ad.m_bias + 2 * (size_t)e) = mb;
          *reinterpret_cast<float2*>(ad.v_bias + 2 * (size_t)e) = vb;
        }
      } else {
        *reinterpret_cast<float2*>(b.g_bias + 2 * (size_t)e) = make_float2(g0, g1);
      }
    }
    if constexpr (PIPE) {
      if (in_next) {           // (uniform over the lane group) header of the next-step record: (w, weighted KL, 0, 0)
        kl_next = group_sum<LPE>(kl_next);
        if (lig == 0)
          *reinterpret_cast<float4*>(b.zrec_next + (size_t)e * xs) =
              make_float4(w_next, kl_next * (sh_cs_next[group_index(sh_hi, a.G, e)] * io), 0.f, 0.f);
      }
    }
  }
  if (nclamp != 0 && b.status) atomicAdd(b.status, nclamp);       // (integer: the total does not depend on the order)
}
