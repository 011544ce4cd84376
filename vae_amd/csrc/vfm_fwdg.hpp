// vfm_fwdg.hpp -- k_fwdg: the general-F forward with the rows' fields SPLIT over lane groups.
// Included inside `namespace vfm { namespace {` of vfm_fwdg.hip (one object per link function), after vfm_fwd2.hpp
// (EntRegs / load_ent / sample_ent: the per-task gather and sampling arithmetic of k_fwd2 are reused as they are).
#pragma once

// ---------------------------------------------------------------------------------------
// Why a third forward kernel.  k_fwd (vfm_fwd.hpp) gives a whole batch row to ONE lane group, which walks the row's F
// fields one after the other with two table rows in flight, each lane owning 4 coordinates.  At the Criteo shape (cfg5:
// 2,048 rows per GPU, F = 32, d = 256, a 2 GB table) that is 2,048 waves -- two per SIMD -- each a serial chain of 32
// dependent gathers, and every lane draws 8 normals per Philox call to keep 4: 57-65 us for 134 MB of table rows,
// 2.3 TB/s.  tools/gather_bench.hip shows what the memory system gives the same 65,536 random 2-KiB rows of a 2 GB
// table: 26 us (5.1-5.3 TB/s) from 8 waves per CU on, 118 us with 2 waves per CU and one row in flight -- the kernel's
// shape was the limit, not the table's size (profiles/r03_gather_bench.md).
//
// Structure.  Lane layout of k_fwd2: a lane group of LPE lanes, lane p owns the 8 coordinates [8p, 8p+8) of an entity
// row = exactly ONE Philox4x32-10 call, nothing drawn twice or thrown away.  The F fields of a row are cut in NS
// segments of SEG consecutive fields; a lane group streams the occurrences of ONE segment (table row of occurrence
// t+1 and the id of t+2 in flight while t computes) and keeps the segment's partial sums  sum_f z, sum_f z^2,
// sum_f w, sum_f KL  in registers.  The NS groups of a row sit in the same workgroup: partial sums meet in LDS (48
// bytes per lane, conflict-free b128), the first group of the row adds them IN SEGMENT ORDER and finishes the row
// (FM value 1/2[(sum z)^2 - sum z^2], likelihood, sumz / grow / pred).  NS and SEG depend on (F, d) only -- never on
// the batch size -- so the arithmetic of a row is the same in every launch that contains it (a rank's shard of a batch
// gives bitwise the rows of the whole batch).  cfg5: NS = 4 -> 8,192 lane groups = 4,096 waves, 16 per CU.
// ---------------------------------------------------------------------------------------
template <bool ID64>
__device__ __forceinline__ uint32_t raw_id_at(const KArgs& a, int64_t pos, uint32_t& hi) {
  if constexpr (ID64) {
    const uint2 t = reinterpret_cast<const uint2*>(a.x)[pos];
    hi = t.y;
    return t.x;
  } else {
    hi = 0u;
    return reinterpret_cast<const uint32_t*>(a.x)[pos];
  }
}

template <int LPE, bool FULL, int EPS, int MODE, bool ID64, int LINK, bool WREC>
__global__ __launch_bounds__(BLOCK, 4) void k_fwdg(const KArgs a, const FwdOut out, const int NS, const int SEG) {
  constexpr int GPB = BLOCK / LPE;
  __shared__ float sh_cs[VFM_MAX_FIELDS];
  __shared__ int64_t sh_hi[VFM_MAX_FIELDS];
  __shared__ float sh_red[6 * 4];
  __shared__ float sh_e0;
  __shared__ float4 sh_part[BLOCK * 3];          // per lane: (sum z [8] | sum z^2, sum w, sum KL, -)

  const int tid = threadIdx.x;
  const int lig = tid % LPE, grp = tid / LPE;
  const int F = a.F;
  const int C = a.d >> 2;                        // chunks of 4 coordinates (d % 4 == 0 here)
  const RngKey key = key_of_step(a, blockIdx.x == 0 && tid == 0);

  if (MODE == MODE_TRAIN && tid < a.G) {
    sh_cs[tid] = (float)(a.group_n[tid] / a.W[tid]);
    sh_hi[tid] = a.group_hi[tid];
  }
  if constexpr (EPS == EPS_PHILOX) {
    if (tid < 64) {
      float n[8], nb;
      normal8b(key, 0xFFFFFFFFu, 0u, n, nb);
      if (tid == 0) sh_e0 = n[0];
    }
  } else if (tid == 0) {
    sh_e0 = 0.f;
  }
  __syncthreads();
  const float alpha = a.scalars[0], m0 = a.scalars[1], s0 = a.scalars[2];
  const float aabs = link_f<LINK>(alpha);
  const float w0 = fmaf(link_f<LINK>(s0), sh_e0, m0);
  const float half_log_a = 0.5f * LN2 * __builtin_amdgcn_logf(aabs);
  const bool owns_bias = lig == 0;
  const uint32_t T32 = (uint32_t)a.T;

  const int j0 = 2 * lig, j1 = 2 * lig + 1;
  const bool v0 = j0 < C, v1 = j1 < C;
  const int off0 = 4 * (v0 ? j0 : C - 1), off1 = 4 * (v1 ? j1 : C - 1);
  const uint32_t pg = (uint32_t)lig + (key.chunk_off >> 1);

  // weight of an entity's KL: n_g / W_g of its id group (usually column f <-> group f)
  auto cs_of = [&](uint32_t e, int fcol) -> float {
    if constexpr (MODE != MODE_TRAIN) return 0.f;
    const int64_t id = (int64_t)e;
    const int64_t lo = fcol > 0 ? sh_hi[fcol - 1] : 0;
    if (id >= lo && id < sh_hi[fcol]) return sh_cs[fcol];
    return sh_cs[group_index(sh_hi, a.G, id)];
  };
  auto fold = [&](uint32_t lo, uint32_t hi, bool live, float& bad) -> uint32_t {
    const bool ok = hi == 0u && lo < T32;
    bad += (live && !ok) ? 1.f : 0.f;
    return ok ? lo : 0u;
  };

  const int RPB = GPB / NS;                      // rows per workgroup and iteration (NS <= GPB, both powers of two)
  const int seg = grp % NS, rloc = grp / NS;
  const int f0 = seg * SEG;
  const int nf_seg = (F - f0) < SEG ? (F - f0) : SEG;      // fields of this group's segment (<= 0: none)

  float tot[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // ll, kl, g, alpha-term, bad ids, (unused)
  for (int64_t rb = (int64_t)blockIdx.x * RPB; rb < a.B; rb += (int64_t)gridDim.x * RPB) {     // uniform trip count
    const int64_t r = rb + rloc;
    const bool live = r < a.B;
    const int nf = live ? nf_seg : 0;
    float sz[8], zz = 0.f, part = 0.f, kl = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) sz[t] = 0.f;
    if (nf > 0) {
      const int64_t base = r * F + f0;
      EntRegs<EPS> A, Bq;
      uint32_t h0, h1;
      const uint32_t l0 = raw_id_at<ID64>(a, base, h0);
      uint32_t l1 = raw_id_at<ID64>(a, base + (nf > 1 ? 1 : 0), h1);
      load_ent<EPS, MODE, WREC>(a, fold(l0, h0, true, tot[4]), off0, off1, A);
      auto consume = [&](const EntRegs<EPS>& cu, int fcol) {
        float z[8], w, klw;
        sample_ent<FULL, EPS, MODE, LINK>(key, cu, pg, v0, v1, owns_bias, cs_of(cu.e, fcol), z, w, klw);
#pragma unroll
        for (int t = 0; t < 8; ++t) { sz[t] += z[t]; zz = fmaf(z[t], z[t], zz); }
        part += w;
        kl += klw;
      };
      for (int i = 0; i < nf; i += 2) {
        // occurrence i+1 into Bq (its id arrived a stage ago), id of i+2; then the arithmetic of i
        {
          const uint32_t e1 = fold(l1, h1, i + 1 < nf, tot[4]);
          l1 = raw_id_at<ID64>(a, base + (i + 2 < nf ? i + 2 : nf - 1), h1);
          load_ent<EPS, MODE, WREC>(a, e1, off0, off1, Bq);
        }
        consume(A, f0 + i);
        if (i + 1 >= nf) break;
        {
          const uint32_t e2 = fold(l1, h1, i + 2 < nf, tot[4]);
          l1 = raw_id_at<ID64>(a, base + (i + 3 < nf ? i + 3 : nf - 1), h1);
          load_ent<EPS, MODE, WREC>(a, e2, off0, off1, A);
        }
        consume(Bq, f0 + i + 1);
      }
    }
    if (NS > 1) {                                // (uniform) the segments of a row meet in LDS, added in segment order
      sh_part[tid * 3 + 0] = make_float4(sz[0], sz[1], sz[2], sz[3]);
      sh_part[tid * 3 + 1] = make_float4(sz[4], sz[5], sz[6], sz[7]);
      sh_part[tid * 3 + 2] = make_float4(zz, part, kl, 0.f);
      __syncthreads();
      if (seg == 0) {
        for (int s2 = 1; s2 < NS; ++s2) {
          const int o = (tid + s2 * LPE) * 3;
          const float4 p0 = sh_part[o], p1 = sh_part[o + 1], p2 = sh_part[o + 2];
          sz[0] += p0.x; sz[1] += p0.y; sz[2] += p0.z; sz[3] += p0.w;
          sz[4] += p1.x; sz[5] += p1.y; sz[6] += p1.z; sz[7] += p1.w;
          zz += p2.x; part += p2.y; kl += p2.z;
        }
      }
      __syncthreads();                           // (the next iteration overwrites the partials)
    }
    if (seg == 0 && live) {                      // finish the row: FM value, likelihood, outputs
      float q = -zz;
#pragma unroll
      for (int t = 0; t < 8; ++t) q = fmaf(sz[t], sz[t], q);
      const float val = group_sum<LPE>(fmaf(0.5f, q, part));
      if constexpr (MODE == MODE_TRAIN) {
        tot[1] += kl;
        float* srow = out.sumz + (size_t)r * a.d;
        Chunk<4> c0, c1;
#pragma unroll
        for (int t = 0; t < 4; ++t) { c0.v[t] = sz[t]; c1.v[t] = sz[4 + t]; }
        if (FULL || v0) st_chunk<4>(srow + off0, c0);
        if (FULL || v1) st_chunk<4>(srow + off1, c1);
      }
      const float pred = w0 + val;
      if (lig == 0) {
        out.pred[r] = pred;
        if constexpr (MODE == MODE_TRAIN) {
          float ll, dll, at;
          lik_terms(a.lik, a.y[r], pred, aabs, half_log_a, ll, dll, at);
          const float g = -a.ll_scale * dll;
          tot[0] += ll; tot[2] += g; tot[3] += at;
          out.grow[r] = g;
        }
      }
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
