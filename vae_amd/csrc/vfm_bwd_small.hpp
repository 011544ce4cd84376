// vfm_bwd_small.hpp -- k_bwd_small: the fused backward + dense Adam for SMALL TABLES, one wave per table row.
// Included inside `namespace vfm { namespace {` of vfm_bwd.hip (the fused-Adam part).
#pragma once

// Why.  With a small table under a large batch (BASELINE configs[1], the ML-100K shape: 2,625 entities under 80,000 rows,
// vfm-torch.py:31-57,77 -- ONE batch per epoch) every entity sits in ~60 rows.  k_bwd gives a table row to one lane group:
// 2,625 lane groups = 330 waves on a chip of 1,024 SIMDs, each walking its list serially; the long lists therefore go
// through k_heavy first (work items of `heavy_list` occurrences, one lane group each) and the step is three dependent
// launches of under-filled grids: 31 us of backward for 4.6 MB of traffic (profiles/r04_cfg2_pmc_summary.txt).
// Here ONE launch does all of it: a WAVE owns a table row; its 64 / LPE lane groups reduce the row's work items side by
// side (item j of the list = occurrences [beg + j L, beg + (j+1) L): exactly the cut vfm_build_index writes into
// heavy_items, recomputed here from the list itself), the item records meet in LDS, and the wave's first lane group adds
// them up, finishes the gradient and applies Adam.  2,625 waves, no pre-reduction kernel, no scratch table.
//
// BITWISE the trajectory of k_heavy + k_heavy_sum + k_bwd<ADAM = 1> on the same index: every sum is formed in the order
// those kernels use --
//   an item:          A = fma(g_r, sumz_r, A) over its occurrences in list order, gs += g_r              (k_heavy)
//   a heavy entity:   P_g = items g, g + GPB, g + 2 GPB ... added in that order, GPB = 256 / LPE (k_heavy_sum's lane
//                     groups), then P_0 + P_1 + ... in order; with at most VFM_HEAVY_DIRECT items k_bwd adds the items in
//                     order itself -- here: VFM_HEAVY_DIRECT partial sums of one item each (0 + x = x)
//   a short list:     k_bwd's own walk: A = fma(g1, s1, fma(g0, s0, A)) two occurrences at a time, gs += g0 + g1
// -- and the epilogue (eps regeneration, KL part, link, Adam) is k_bwd's, expression for expression (the unit is compiled
// with -ffp-contract=on: fusion follows the source expression, so equal source gives equal bits).  tests/test_gpu_model.py
// runs both paths side by side (VFM_BWD_SMALL=0 forces the three-launch path).
// Serves: STAGE_FULL, fused dense Adam (scaled or plain moments), Philox eps, one sample, every row, d % 4 == 0, d <= 256.
template <int LPE, int LINK>
__global__ __launch_bounds__(BLOCK) void k_bwd_small(const KArgs a, const BwdArgs b, const AdamArgs ad_in, int L, int THR) {
  constexpr int VEC = 4;
  constexpr int LGW = 64 / LPE;                      // lane groups of a wave
  constexpr int GPBH = BLOCK / LPE;                  // lane groups of a k_heavy_sum workgroup: the partial sums P_g
  constexpr int WPB = BLOCK / 64;                    // waves (= table rows) per workgroup
  constexpr int U = 8;                               // occurrences of a work item in flight
  AdamArgs ad = ad_in;
  RngKey key = a.key, next_key = b.next_key;
  int32_t la_step = 0, la_k = 0;
  load_dev_step(a, key, ad, la_step, la_k, next_key, blockIdx.x == 0 && threadIdx.x == 0);
  __shared__ float sh_cs[VFM_MAX_FIELDS];
  __shared__ int64_t sh_hi[VFM_MAX_FIELDS];
  __shared__ double sh_fin[7][BLOCK / 64];
  constexpr int NPMAX = GPBH > VFM_HEAVY_DIRECT ? GPBH : VFM_HEAVY_DIRECT;
  __shared__ __attribute__((aligned(16))) float sh_P[WPB][NPMAX][4 * LPE + 4];     // per wave: P_g = (A[0 .. 4 LPE) | gs, -, -, -)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lig = lane % LPE, lg = lane / LPE;
  const int d = a.d;
  const int C = (d + VEC - 1) / VEC;
  const int ligc = lig < C ? lig : C - 1;            // lanes past the last chunk re-load the last one (their sums are never used)
  if (tid < a.G) {
    sh_cs[tid] = (float)(a.group_n[tid] / a.W[tid]);
    sh_hi[tid] = a.group_hi[tid];
  }
  __syncthreads();
  const float gout = 1.0f;

  // ---- loss + the three scalars: workgroup 0, as in k_bwd
  double fin[6] = {0, 0, 0, 0, 0, 0};
  const bool fold = b.loss != nullptr;
  if (blockIdx.x == 0 && fold)
    reduce_slots_and_loss(b.partials, a.scalars, a.ll_scale_d, a.flags, b.loss, sh_fin, fin);
  if (blockIdx.x == 0 && tid == 0) {
    const float alpha = a.scalars[0], m0 = a.scalars[1], s0 = a.scalars[2];
    const bool stale = !fold && b.partials[VFM_P_REDUCED] != 1.0;
    const float sum_g = stale ? __builtin_nanf("") : (float)(fold ? fin[VFM_P_G] : b.partials[VFM_P_G]);
    const float sum_a = (float)(fold ? fin[VFM_P_ALPHA] : b.partials[VFM_P_ALPHA]);
    float e0 = 0.f;
    {
      float n[8], nb;
      normal8b(key, 0xFFFFFFFFu, 0u, n, nb);
      e0 = n[0];
    }
    const float as0 = link_f<LINK>(s0);
    const float prior = (a.flags & VFM_FLAG_NO_PRIOR_TERMS) ? 0.f : 1.f;
    const float ga = (a.lik == VFM_LIK_NORMAL)
                         ? gout * dlink_f<LINK>(alpha) * a.ll_scale * sum_a : 0.f;
    const float gm = gout * (sum_g + prior * m0);
    const float ge0 = e0 * sum_g;
    const float gs = gout * dlink_f<LINK>(s0) * (ge0 + prior * (as0 - inv_sigma(as0)));
    float* sc = const_cast<float*>(a.scalars);
    auto upd = [&](int i, float g) {
      float m = ad.m_scal[i], v = ad.v_scal[i];
      sc[i] = adam_update(sc[i], g, m, v, ad);
      ad.m_scal[i] = m; ad.v_scal[i] = v;
    };
    if (a.lik == VFM_LIK_NORMAL) upd(0, ga);
    upd(1, gm);
    upd(2, gs);
  }

  // ---- the table rows: one wave each
  int nclamp = 0;
  const int n_occ = b.n_occ;
  const int Bm1 = a.B > 0 ? (int)a.B - 1 : 0;
  auto row_ok = [&](int v, bool live = true) -> int {
    const bool ok = (unsigned)v <= (unsigned)Bm1;
    nclamp += (ok || !live) ? 0 : 1;
    return ok ? v : 0;
  };
  float(*P)[4 * LPE + 4] = sh_P[wave];
  for (int64_t e = (int64_t)blockIdx.x * WPB + wave; e < a.T; e += (int64_t)gridDim.x * WPB) {
    int beg = b.occ_ptr[e], end = b.occ_ptr[e + 1];
    if (beg < 0 || end < beg || end > n_occ) { beg = end = 0; ++nclamp; }
    const bool touched = beg != end;
    const float cntf = (float)(end - beg);
    float* prow = const_cast<float*>(a.entity) + (size_t)e * (2 * (size_t)d);
    // the row's own parameters and moments: issued before the list is walked (first lane group of the wave)
    Chunk<VEC> mu, s, mm, ms, vm, vs;
    float2 th = make_float2(0.f, 1.f), mb = make_float2(0.f, 0.f), vb = make_float2(0.f, 0.f);
    float io = 0.f;
    const bool mine = lg == 0 && lig < C;
    if (mine) {
      mu = ld_chunk<VEC>(prow + (size_t)lig * VEC);
      s = ld_chunk<VEC>(prow + d + (size_t)lig * VEC);
      const size_t o = (size_t)e * (2 * (size_t)d) + (size_t)lig * VEC;
      mm = ld_chunk_nt<VEC>(ad.m_entity + o); ms = ld_chunk_nt<VEC>(ad.m_entity + o + d);
      vm = ld_chunk_nt<VEC>(ad.v_entity + o); vs = ld_chunk_nt<VEC>(ad.v_entity + o + d);
    }
    if (lane == 0) {
      th = *reinterpret_cast<const float2*>(a.bias + 2 * (size_t)e);
      mb = *reinterpret_cast<const float2*>(ad.m_bias + 2 * (size_t)e);
      vb = *reinterpret_cast<const float2*>(ad.v_bias + 2 * (size_t)e);
    }
    if (touched && lg == 0) io = a.inv_occ[e];

    Chunk<VEC> A;
#pragma unroll
    for (int t = 0; t < VEC; ++t) A.v[t] = 0.f;
    float gs = 0.f;
    const int cnt = end - beg;
    if (cnt > THR) {
      // heavy: the items of the list, L occurrences each, in list order.  Lane group lg forms the partial sums P_g of
      // g = lg, lg + LGW, ... (g < GPBH): items g, g + GPBH, ... -- each item a sequential fma chain as k_heavy forms it
      const int n_items = (cnt + L - 1) / L;
      // partial sums: k_heavy_sum's lane groups (items g, g + GPBH, ...) -- or, with at most VFM_HEAVY_DIRECT items, one per
      // item, which is k_bwd's own in-order sum of the item records (with 64 lanes per entity GPBH is 4: fewer than that)
      const int NP = n_items <= VFM_HEAVY_DIRECT ? VFM_HEAVY_DIRECT : GPBH;
      for (int g = lg; g < NP; g += LGW) {
        Chunk<VEC> Pg;
#pragma unroll
        for (int t = 0; t < VEC; ++t) Pg.v[t] = 0.f;
        float Pgs = 0.f;
        for (int it = g; it < n_items; it += NP) {
          const int ib = beg + it * L, ie = (ib + L < end) ? ib + L : end;
          Chunk<VEC> Ai;
#pragma unroll
          for (int t = 0; t < VEC; ++t) Ai.v[t] = 0.f;
          float gi = 0.f;
          // U occurrences in flight: every load of a batch is issued before the first fma needs one; the sums themselves
          // are formed one occurrence at a time in list order, as k_heavy forms them.  The row ids of the NEXT batch are
          // fetched under the current one.
          int rn[U];
#pragma unroll
          for (int u = 0; u < U; ++u) rn[u] = b.occ_rows[ib + u < ie ? ib + u : ie - 1];
          for (int o = ib; o < ie; o += U) {
            int r[U];
            float g[U];
            Chunk<VEC> sv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) r[u] = row_ok(rn[u], o + u < ie);
#pragma unroll
            for (int u = 0; u < U; ++u) {
              g[u] = b.grow[r[u]];
              sv[u] = ld_chunk<VEC>(b.sumz + (size_t)r[u] * d + (size_t)ligc * VEC);      // (no load under a branch)
            }
#pragma unroll
            for (int u = 0; u < U; ++u) rn[u] = b.occ_rows[o + U + u < ie ? o + U + u : ie - 1];
#pragma unroll
            for (int u = 0; u < U; ++u) {
              if (o + u < ie) {
                gi += g[u];
#pragma unroll
                for (int t = 0; t < VEC; ++t) Ai.v[t] = fmaf(g[u], sv[u].v[t], Ai.v[t]);
              }
            }
          }
          Pgs += gi;
#pragma unroll
          for (int t = 0; t < VEC; ++t) Pg.v[t] += Ai.v[t];
        }
        *reinterpret_cast<float4*>(&P[g][4 * lig]) = make_float4(Pg.v[0], Pg.v[1], Pg.v[2], Pg.v[3]);
        if (lig == 0) P[g][4 * LPE] = Pgs;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // (one wave: its LDS writes are ordered before the reads below)
      __builtin_amdgcn_wave_barrier();
      if (lg == 0) {                                 // P_0 + P_1 + ... in order
        for (int g = 0; g < NP; ++g) {
          const float4 t4 = *reinterpret_cast<const float4*>(&P[g][4 * lig]);
          A.v[0] += t4.x; A.v[1] += t4.y; A.v[2] += t4.z; A.v[3] += t4.w;
          gs += P[g][4 * LPE];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // (... and the reads before the next row's writes)
      __builtin_amdgcn_wave_barrier();
    } else if (lg == 0) {
      // a short list: k_bwd's own walk, two occurrences in flight
      int o = beg;
      for (; o + 1 < end; o += 2) {
        const int r0 = row_ok(b.occ_rows[o]), r1 = row_ok(b.occ_rows[o + 1]);
        const float g0 = b.grow[r0], g1 = b.grow[r1];
        gs += g0 + g1;
        if (lig < C) {
          const Chunk<VEC> s0v = ld_chunk<VEC>(b.sumz + (size_t)r0 * d + (size_t)lig * VEC);
          const Chunk<VEC> s1v = ld_chunk<VEC>(b.sumz + (size_t)r1 * d + (size_t)lig * VEC);
#pragma unroll
          for (int t = 0; t < VEC; ++t) A.v[t] = fmaf(g1, s1v.v[t], fmaf(g0, s0v.v[t], A.v[t]));
        }
      }
      if (o < end) {
        const int r0 = row_ok(b.occ_rows[o]);
        const float g0 = b.grow[r0];
        gs += g0;
        if (lig < C) {
          const Chunk<VEC> s0v = ld_chunk<VEC>(b.sumz + (size_t)r0 * d + (size_t)lig * VEC);
#pragma unroll
          for (int t = 0; t < VEC; ++t) A.v[t] = fmaf(g0, s0v.v[t], A.v[t]);
        }
      }
    }
    if (lg != 0) continue;                           // (the row's epilogue: the wave's first lane group)

    float c = 0.f;
    if (touched) {
      c = sh_cs[group_index(sh_hi, a.G, e)] * io * cntf;
    }
    float nb_eps = 0.f;
    if (lig < C) {
      Chunk<VEC> gm, gv;
      if (touched) {
        Chunk<VEC> epc;
        float nb;
        eps_of_chunk<VEC>(key, (uint32_t)e, lig, epc.v, nb);
        nb_eps = nb;
#pragma unroll
        for (int t = 0; t < VEC; ++t) {
          const float sg = link_f<LINK>(s.v[t]);
          const float z = fmaf(sg, epc.v[t], mu.v[t]);
          const float gz = A.v[t] - z * gs;
          gm.v[t] = gout * (gz + c * mu.v[t]);
          gv.v[t] = gout * dlink_f<LINK>(s.v[t]) * (gz * epc.v[t] + c * (sg - inv_sigma(sg)));
        }
      } else {
#pragma unroll
        for (int t = 0; t < VEC; ++t) { gm.v[t] = 0.f; gv.v[t] = 0.f; }
      }
      Chunk<VEC> pm, ps;
#pragma unroll
      for (int t = 0; t < VEC; ++t) {
        pm.v[t] = adam_update(mu.v[t], gm.v[t], mm.v[t], vm.v[t], ad);
        ps.v[t] = adam_update(s.v[t], gv.v[t], ms.v[t], vs.v[t], ad);
      }
      const size_t o2 = (size_t)e * (2 * (size_t)d) + (size_t)lig * VEC;
      st_chunk<VEC>(prow + (size_t)lig * VEC, pm);
      st_chunk<VEC>(prow + d + (size_t)lig * VEC, ps);
      if (!ad.scaled || touched || ad.store_true) {     // (scaled: rows without gradient keep ms, vs)
        st_chunk_nt<VEC>(ad.m_entity + o2, mm); st_chunk_nt<VEC>(ad.m_entity + o2 + d, ms);
        st_chunk_nt<VEC>(ad.v_entity + o2, vm); st_chunk_nt<VEC>(ad.v_entity + o2 + d, vs);
      }
    }
    if (lig == 0) {
      float g0 = 0.f, g1 = 0.f;
      if (touched) {
        const float sg = link_f<LINK>(th.y);
        g0 = gout * (gs + c * th.x);
        g1 = gout * dlink_f<LINK>(th.y) * (gs * nb_eps + c * (sg - inv_sigma(sg)));
      }
      float2 pn;
      pn.x = adam_update(th.x, g0, mb.x, vb.x, ad);
      pn.y = adam_update(th.y, g1, mb.y, vb.y, ad);
      *reinterpret_cast<float2*>(const_cast<float*>(a.bias) + 2 * (size_t)e) = pn;
      if (a.wrec) *reinterpret_cast<float2*>(a.wrec + 4 * (size_t)e) = pn;      // packed first-order record: (mu_w, s_w | 1/occ, 0)
      if (!ad.scaled || touched || ad.store_true) {
        *reinterpret_cast<float2*>(ad.m_bias + 2 * (size_t)e) = mb;
        *reinterpret_cast<float2*>(ad.v_bias + 2 * (size_t)e) = vb;
      }
    }
  }
  if (nclamp != 0 && b.status) atomicAdd(b.status, nclamp);
}
