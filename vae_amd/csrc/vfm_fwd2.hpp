// vfm_fwd2.hpp -- k_fwd2: the two-field forward as a stream of SAMPLING TASKS.
// Included inside `namespace vfm { namespace {` of vfm_fwd2.hip (one object per link function).
#pragma once

// ---------------------------------------------------------------------------------------
// Why a second forward kernel.  k_fwd (vfm_fwd.hpp) samples every (row, field) OCCURRENCE: 2 B d normals
// per batch, where the reference draws one set per UNIQUE entity (vfm-torch.py:207-208,238-245), and it
// is bound by exactly that arithmetic (Philox + Box-Muller on the quarter-rate units; profiles/).
// With the rows of a batch ordered by the id of one column (VFM.fit does that; at ML-20M shape an item
// then owns a run of ~3.8 consecutive rows) the repeated draws of that column are pure waste.
//
// Structure.  A lane group of LPE lanes owns a CONTIGUOUS range of rows; lane p owns the 8 coordinates
// [8p, 8p+8) of an entity row = exactly the output of ONE Philox4x32-10 call (vfm_rng.hpp: counter
// (p, e, step)).  The group works through a stream of sampling tasks, one per loop iteration:
//     row r:  [ item task of x[r,1]  -- only if x[r,1] differs from the item cached in registers ]
//             user task of x[r,0]    -- samples the user, finishes the row against the cached item
// A task = gather the entity's table row -> eps -> z = mu + sigma*eps, first-order weight, KL.  An item
// task parks (z, w, weighted KL) of the item in registers; a user task takes the dot product with them,
// the likelihood and the row's outputs.  So a run of R rows sharing an item costs R + 1 tasks instead of
// 2R, and every task uses all 8 + 1 normals of its Philox call.  The four groups of a wave advance
// independently (different task kinds in the same iteration share the sampling code; only the short
// epilogues diverge).  Correctness does not depend on the row order: an unsorted batch just has no runs.
// The table row of task t+1 is gathered while task t computes (ids two rows ahead), as in k_fwd.
//
// Same outputs as k_fwd for F = 2, n_samples = 1 (pred, grow, sumz, per-workgroup partial slots), same
// eps stream (the backward regenerates eps from the same counters), FM term as the reference writes it
// for two fields: sum_k z_u z_i (vfm-torch.py:245).
// ---------------------------------------------------------------------------------------

template <int EPS>
struct EntRegs {              // table row of one sampling task, this lane's two chunks
  uint32_t e;
  Chunk<4> mu[2], s[2], ep[2];
  float2 th;                  // (mu_w, s_w)
  float io;                   // 1 / occ
  float epw;                  // first-order eps (table mode)
};

// raw ids of one row.  Nothing here looks at a loaded value (that would make the compiler wait for every
// load in flight right behind the load): ids are folded to 32 bits and range-checked when the id window
// slides, a pipeline stage later.
template <bool ID64>
struct RawIds {
  uint32_t ulo, uhi, ilo, ihi;       // (ID64 == false: uhi / ihi unused)
};
struct Ids {                          // folded: entity id, or BAD_ID when outside [0, T)
  uint32_t u, i;
};
constexpr uint32_t BAD_ID = 0xFFFFFFFFu;      // (T <= 0xFFFFFFFE: check_problem)

template <bool ID64>
__device__ __forceinline__ RawIds<ID64> load_row_ids(const KArgs& a, int r) {
  RawIds<ID64> v;
  if constexpr (ID64) {
    const uint4 t = reinterpret_cast<const uint4*>(a.x)[r];
    v.ulo = t.x; v.uhi = t.y; v.ilo = t.z; v.ihi = t.w;
  } else {
    const uint2 t = reinterpret_cast<const uint2*>(a.x)[r];
    v.ulo = t.x; v.uhi = 0u; v.ilo = t.y; v.ihi = 0u;
  }
  return v;
}

template <bool ID64>
__device__ __forceinline__ Ids fold_ids(const RawIds<ID64>& v, uint32_t T32) {
  Ids o;
  o.u = ((!ID64 || v.uhi == 0u) && v.ulo < T32) ? v.ulo : BAD_ID;
  o.i = ((!ID64 || v.ihi == 0u) && v.ilo < T32) ? v.ilo : BAD_ID;
  return o;
}

// entity of a folded id: out-of-range ids are clamped to 0 and counted
__device__ __forceinline__ uint32_t entity_of(uint32_t id, bool live, float& bad) {
  const bool ok = id != BAD_ID;
  bad += (live && !ok) ? 1.f : 0.f;
  return ok ? id : 0u;
}

template <int EPS, int MODE, bool WREC = false>
__device__ __forceinline__ void load_ent(const KArgs& a, uint32_t e, int off0, int off1, EntRegs<EPS>& R) {
  const size_t d = (size_t)a.d;
  R.e = e;
  if constexpr (EPS == EPS_ZREC) {
    // the "table" holds this step's SAMPLES, one record (w, weighted KL, 0, 0 | z[0..d-1]) per entity id, written
    // by the previous step's backward (or vfm_sample_records_f32): nothing to sample, half the bytes of (mu | s)
    const float* rec = a.entity + (size_t)e * (4 + d);
    R.mu[0] = ld_chunk<4>(rec + 4 + off0);
    R.mu[1] = ld_chunk<4>(rec + 4 + off1);
    const float4 h = *reinterpret_cast<const float4*>(rec);
    R.th = make_float2(h.x, 0.f);
    R.io = h.y;
    return;
  }
  const float* row = a.entity + (size_t)e * (2 * d);
  R.mu[0] = ld_chunk<4>(row + off0);
  R.mu[1] = ld_chunk<4>(row + off1);
  R.s[0] = ld_chunk<4>(row + d + off0);
  R.s[1] = ld_chunk<4>(row + d + off1);
  if constexpr (EPS == EPS_TABLE) {
    const float* er = a.eps_entity + (size_t)e * d;
    R.ep[0] = ld_chunk<4>(er + off0);
    R.ep[1] = ld_chunk<4>(er + off1);
    R.epw = a.eps_bias[e];
  }
  if constexpr (WREC) {
    // packed first-order record (mu_w, s_w, 1/occ, 0): ONE 16-byte load = one cache line per task, instead of an
    // 8-byte row of bias_params and a 4-byte entry of inv_occ in two different lines (vfm_problem_t.wrec)
    const float4 h = *reinterpret_cast<const float4*>(a.wrec + 4 * (size_t)e);
    R.th = make_float2(h.x, h.y);
    R.io = h.z;
  } else {
    R.th = *reinterpret_cast<const float2*>(a.bias + 2 * (size_t)e);
    if constexpr (MODE == MODE_TRAIN) R.io = a.inv_occ[e];
  }
}

// the arithmetic of one task: z (8 coordinates of this lane), sampled first-order weight (lane 0 of the
// group), this lane's share of the entity's weighted KL
template <bool FULL, int EPS, int MODE, int LINK>
__device__ __forceinline__ void sample_ent(const RngKey& key, const EntRegs<EPS>& R, uint32_t pg, bool v0, bool v1,
                                           bool owns_bias, float cs, float (&z)[8], float& w, float& klw) {
  if constexpr (EPS == EPS_ZREC) {          // precomputed sample: z, w and the entity's weighted KL come from the record
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      z[t] = (FULL || v0) ? R.mu[0].v[t] : 0.f;
      z[4 + t] = (FULL || v1) ? R.mu[1].v[t] : 0.f;
    }
    w = owns_bias ? R.th.x : 0.f;
    klw = (MODE == MODE_TRAIN && owns_bias) ? R.io : 0.f;
    return;
  }
  float ep[8], epw = 0.f;
  if constexpr (EPS == EPS_TABLE) {
#pragma unroll
    for (int t = 0; t < 4; ++t) { ep[t] = R.ep[0].v[t]; ep[4 + t] = R.ep[1].v[t]; }
    epw = R.epw;
  } else if constexpr (EPS == EPS_PHILOX) {
    normal8b(key, R.e, pg, ep, epw);
  } else {
#pragma unroll
    for (int t = 0; t < 8; ++t) ep[t] = 0.f;
  }
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
      const v2f e2 = {ep[4 * c + 2 * h], ep[4 * c + 2 * h + 1]};
      const v2f z2 = g2 * e2 + m2;
      z[4 * c + 2 * h] = valid ? z2.x : 0.f;
      z[4 * c + 2 * h + 1] = valid ? z2.y : 0.f;
      if constexpr (MODE == MODE_TRAIN) {
        kq = g2 * g2 + kq;
        kq = m2 * m2 + kq;
        // log s0 + log s1 = log(s0 * s1): one v_log_f32 per pair (the clamped product stays >= 1e-24)
        lg += __builtin_amdgcn_logf(fmaxf(g2.x, SIGMA_MIN) * fmaxf(g2.y, SIGMA_MIN));
      }
    }
    if constexpr (MODE == MODE_TRAIN) klv += valid ? fmaf(0.5f, kq.x + kq.y, fmaf(-LN2, lg, -2.0f)) : 0.f;
  }
  const float sgw = link_f<LINK>(R.th.y);
  w = owns_bias ? fmaf(sgw, epw, R.th.x) : 0.f;
  klw = 0.f;
  if constexpr (MODE == MODE_TRAIN) {
    klv += owns_bias ? kl_std_normal(R.th.x, sgw) : 0.f;
    klw = klv * (cs * R.io);
  }
}

template <int LPE, bool FULL, int EPS, int MODE, bool ID64, int LINK, bool WREC = false>
__global__ __launch_bounds__(BLOCK, 4) void k_fwd2(const KArgs a, const FwdOut out) {
  constexpr int GPB = BLOCK / LPE;
  __shared__ float sh_cs[2];
  __shared__ int64_t sh_hi[2];
  __shared__ float sh_red[6 * 4];
  __shared__ float sh_e0;

  const int tid = threadIdx.x;
  const int lig = tid % LPE;
  const int C = a.d >> 2;                       // chunks of 4 coordinates (d % 4 == 0 here)
  // rows of this group: the batch is dealt in contiguous ranges over all groups of the grid (B * 2 < 2^31).  The ids of
  // its first three rows are requested before anything else: at ~1 row per group (the ML-100K shape on a full grid) the
  // wave's life is one chain  ids -> table rows -> sample -> outputs,  and the prologue's barrier would otherwise sit
  // in front of that chain instead of under it
  const int NG = (int)gridDim.x * GPB;
  const int gid = (int)blockIdx.x * GPB + tid / LPE;
  const int q = (int)(a.B / NG), rem = (int)(a.B % NG);
  const int gbeg = gid * q + (gid < rem ? gid : rem);
  const int gend = gbeg + q + (gid < rem ? 1 : 0);
  const int glast = gend - 1;
  RawIds<ID64> raw0, raw1, raw2;
  float y_first = 0.f;
  if (gbeg < gend) {
    raw0 = load_row_ids<ID64>(a, gbeg);
    raw1 = load_row_ids<ID64>(a, gbeg + 1 < gend ? gbeg + 1 : glast);
    raw2 = load_row_ids<ID64>(a, gbeg + 2 < gend ? gbeg + 2 : glast);
    if constexpr (MODE == MODE_TRAIN) y_first = a.y[gbeg];
  }
  // eps stream step: the kernel argument, or (replayable step) device memory -- thread 0 of block 0 hands it on
  const RngKey key = key_of_step(a, blockIdx.x == 0 && tid == 0);

  if (MODE == MODE_TRAIN && EPS != EPS_ZREC && tid < 2) {
    sh_cs[tid] = (float)(a.group_n[tid] / a.W[tid]);
    sh_hi[tid] = a.group_hi[tid];
  }
  // the global-bias eps: ONE wave draws it (a Philox call in every wave's prologue is a tenth of a
  // workgroup's arithmetic at B = 100K)
  if constexpr (EPS == EPS_PHILOX || EPS == EPS_ZREC) {
    if (tid < 64) {
      float n[8], nb;
      normal8b(key, 0xFFFFFFFFu, 0u, n, nb);
      if (tid == 0) sh_e0 = n[0];
    }
  } else if (tid == 0) {
    sh_e0 = (EPS == EPS_TABLE) ? a.eps_global[0] : 0.f;
  }
  __syncthreads();
  const float alpha = a.scalars[0], m0 = a.scalars[1], s0 = a.scalars[2];
  const float aabs = link_f<LINK>(alpha);
  const float w0 = fmaf(link_f<LINK>(s0), sh_e0, m0);
  const float half_log_a = 0.5f * LN2 * __builtin_amdgcn_logf(aabs);
  const bool owns_bias = lig == 0;
  float cs0 = 0.f, cs1 = 0.f;
  uint32_t hi0 = 0u;                 // ids below hi0 belong to group 0 (ids are below 2^32)
  if constexpr (MODE == MODE_TRAIN && EPS != EPS_ZREC) {
    cs0 = sh_cs[0]; cs1 = sh_cs[1];
    hi0 = sh_hi[0] > 0xFFFFFFFFLL ? 0xFFFFFFFFu : (uint32_t)sh_hi[0];
  }
  const uint32_t T32 = (uint32_t)a.T;

  // this lane's two chunks (lanes past the last chunk re-load the last one; their values are masked)
  const int j0 = 2 * lig, j1 = 2 * lig + 1;
  const bool v0 = j0 < C, v1 = j1 < C;
  const int off0 = 4 * (v0 ? j0 : C - 1), off1 = 4 * (v1 ? j1 : C - 1);
  const uint32_t pg = (uint32_t)lig + (key.chunk_off >> 1);      // Philox counter word 0 of this lane

  float tot[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // ll, kl, g, alpha-term, bad ids, (unused)
  if (gbeg < gend) {
    // ---- pipeline state: folded ids of the current row and the next one, raw ids of the row after ----
    int r0 = gbeg;                           // row of the current task
    Ids id0 = fold_ids<ID64>(raw0, T32);
    Ids id1 = fold_ids<ID64>(raw1, T32);
    RawIds<ID64> id2 = raw2;
    bool cur_item = true;                    // the first task of a range is always an item task
    EntRegs<EPS> A, Bq;
    load_ent<EPS, MODE, WREC>(a, entity_of(id0.i, true, tot[4]), off0, off1, A);
    float ycur = y_first;                    // target of the row the current task belongs to
    // the cached item
    float zi[8], wi = 0.f, klwi = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) zi[t] = 0.f;

    // one task: decide + gather the next task into `nx`, then compute the current one from `cu`.
    // Returns false when the range is finished.  No load sits under a branch (a load in a conditional
    // block makes the compiler drain every load in flight at the merge point): the id window is slid
    // with selects and its newest row is (re)loaded every iteration.
    auto step = [&](const EntRegs<EPS>& cu, EntRegs<EPS>& nx) -> bool {
      const bool item_now = cur_item;
      const int r_now = r0;
      const float y_now = ycur;
      // ---- next task ----
      const bool adv = !item_now;                               // a user task finishes its row
      const uint32_t n_u = adv ? id1.u : id0.u, n_i = adv ? id1.i : id0.i;
      const int r_next = r0 + (adv ? 1 : 0);
      const bool live_next = r_next < gend;
      const bool item_next = adv && (n_i != id0.i);
      const uint32_t e_next = entity_of(item_next ? n_i : n_u, live_next, tot[4]);
      // slide the id window: rows (r0, r0+1, r0+2) -> (r_next, r_next+1, r_next+2).  The id load is the FIRST
      // load of an iteration: when the next iteration folds it, the 8 loads issued behind it are known to be
      // younger, so the wait is a counted one (as the last load it would be a drain of everything in flight)
      const Ids f2 = fold_ids<ID64>(id2, T32);
      id2 = load_row_ids<ID64>(a, r_next + 2 < gend ? r_next + 2 : glast);
      load_ent<EPS, MODE, WREC>(a, e_next, off0, off1, nx);
      if constexpr (MODE == MODE_TRAIN) ycur = a.y[live_next ? r_next : glast];
      id0.u = n_u; id0.i = n_i;
      id1.u = adv ? f2.u : id1.u;
      id1.i = adv ? f2.i : id1.i;
      r0 = r_next;
      cur_item = item_next;
      // ---- current task ----
      float z[8], w, klw;
      const float cs = (cu.e < hi0) ? cs0 : cs1;
      sample_ent<FULL, EPS, MODE, LINK>(key, cu, pg, v0, v1, owns_bias, cs, z, w, klw);
      if (item_now) {
#pragma unroll
        for (int t = 0; t < 8; ++t) zi[t] = z[t];
        wi = w; klwi = klw;
      } else {
        v2f qv = {0.f, 0.f};
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const v2f zu = {z[2 * h], z[2 * h + 1]};
          const v2f zv = {zi[2 * h], zi[2 * h + 1]};
          qv = zu * zv + qv;
        }
        const float val = group_sum<LPE>(qv.x + qv.y + w + wi);
        if constexpr (MODE == MODE_TRAIN) tot[1] += klw + klwi;
        if constexpr (MODE == MODE_TRAIN && EPS != EPS_ZREC) {     // (ZREC: the backward gathers the samples themselves)
          float* srow = out.sumz + (size_t)r_now * a.d;
          Chunk<4> s0c, s1c;
#pragma unroll
          for (int t = 0; t < 4; ++t) { s0c.v[t] = z[t] + zi[t]; s1c.v[t] = z[4 + t] + zi[4 + t]; }
          if (FULL || v0) st_chunk<4>(srow + off0, s0c);
          if (FULL || v1) st_chunk<4>(srow + off1, s1c);
        }
        const float pred = w0 + val;
        if (lig == 0) {
          out.pred[r_now] = pred;
          if constexpr (MODE == MODE_TRAIN) {
            float ll, dll, at;
            lik_terms(a.lik, y_now, pred, aabs, half_log_a, ll, dll, at);
            const float g = -a.ll_scale * dll;
            tot[0] += ll; tot[2] += g; tot[3] += at;
            out.grow[r_now] = g;
          }
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
      out.partials[VFM_P_REDUCED] = 0.0;       // the sums [0..5] are stale until the slots are reduced
    }
  }
}
