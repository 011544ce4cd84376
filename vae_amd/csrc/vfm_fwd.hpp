// vfm_fwd.hpp -- k_fwd: gather -> reparameterised sample -> FM -> ELBO.
// Included inside `namespace vfm { namespace {` of vfm_fwd.hip (one object per link function).
#pragma once

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
// Variational samples S > 1 (the reference's global N_VARIATIONAL_SAMPLES, vfm-torch.py:19,238-245,265):
// one launch per sample s.  Launch s draws its own eps (RngKey::step_hi carries s), stores sumz[s] and
// adds its row value  b_r^s + q_r^s  to a running sum kept in pred[0..B); the LAST launch turns the mean
// over samples into the S predictions  pred[s,r] = w0^s + mean_s'(b_r^s' + q_r^s')  (the reference
// averages the entity terms over samples BEFORE the likelihood but not w0, :244-245,265), the
// likelihood terms, grow[r] = sum_s dloss/dpred[s,r] and the KL term.

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
template <int VEC, int MODE, int LINK>
__device__ __forceinline__ void chunk_math(const Chunk<VEC>& mu, const Chunk<VEC>& s, const float (&ep)[VEC],
                                           bool valid, Chunk<VEC>& sz, float& zz, float& klv) {
  if constexpr (VEC == 4) {
    v2f zq = {0.f, 0.f}, kq = {0.f, 0.f};
    float lg = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const v2f m2 = {mu.v[2 * h], mu.v[2 * h + 1]};
      const v2f g2 = {link_f<LINK>(s.v[2 * h]), link_f<LINK>(s.v[2 * h + 1])};
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
    const float sg = link_f<LINK>(s.v[0]);
    const float z = valid ? fmaf(sg, ep[0], mu.v[0]) : 0.f;
    sz.v[0] += z;
    zz = fmaf(z, z, zz);
    if constexpr (MODE == MODE_TRAIN) klv += valid ? kl_std_normal(mu.v[0], sg) : 0.f;
  }
}

// first-order weight of one occurrence (the lane that owns it): sample + KL
template <int MODE, int LINK>
__device__ __forceinline__ void bias_math(const float2 th, float epw, bool owner, float& part, float& klv) {
  const float sgw = link_f<LINK>(th.y);
  part += owner ? fmaf(sgw, epw, th.x) : 0.f;
  if constexpr (MODE == MODE_TRAIN) klv += owner ? kl_std_normal(th.x, sgw) : 0.f;
}

// arithmetic of one occurrence (generic path): every lane draws its own chunk's eps
template <int LPE, int CPL, int VEC, int EPS, int MODE, int LINK>
__device__ __forceinline__ void consume_field(const RngKey& key, const FieldRegs<CPL, VEC, EPS>& R,
                                              int lig, int C, float cs, RowAcc<CPL, VEC>& acc) {
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
      eps_of_chunk<VEC>(key, R.e, j, ep, nb);
      if (i == 0) epw = nb;   // only the lane that owns coordinate 0 (lig == 0) uses it
    }
    chunk_math<VEC, MODE, LINK>(R.mu[i], R.s[i], ep, valid, acc.sz[i], acc.zz, klv);
  }
  if constexpr (EPS == EPS_TABLE) epw = R.epw;
  bias_math<MODE, LINK>(R.th, epw, lig == 0, acc.part, klv);
  if constexpr (MODE == MODE_TRAIN) acc.kl = fmaf(cs * R.io, klv, acc.kl);
}

// arithmetic of a two-field row (VEC == 4): ONE Philox call per lane serves both fields.  Lanes
// pair up (2m, 2m+1): the even lane draws the 8 normals of chunks (2m, 2m+1) of field 0's entity,
// the odd lane those of field 1's entity, and they exchange one half over DPP (quad_perm
// [1,0,3,2]).  Lane 0 / lane 1 own the first-order weights of field 0 / field 1 (their calls have
// p == 0 and carry the bias normal).
template <int LPE, int CPL, int EPS, int MODE, int LINK>
__device__ __forceinline__ void consume_row2(const RngKey& key, const FieldRegs<CPL, 4, EPS>& R0,
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
      normal8b(key, odd ? R1.e : R0.e, ((uint32_t)j + key.chunk_off) >> 1, n, nb);
      if (i == 0) epw = nb;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float send = odd ? n[t] : n[4 + t];     // what the partner lane needs from me
        const float recv = dpp_f<0xB1>(send);
        ep0[t] = odd ? recv : n[t];                   // field 0, my chunk
        ep1[t] = odd ? n[4 + t] : recv;               // field 1, my chunk
      }
    }
    chunk_math<4, MODE, LINK>(R0.mu[i], R0.s[i], ep0, valid, acc.sz[i], acc.zz, kl0);
    chunk_math<4, MODE, LINK>(R1.mu[i], R1.s[i], ep1, valid, acc.sz[i], acc.zz, kl1);
  }
  if constexpr (EPS == EPS_TABLE) epw = odd ? R1.epw : R0.epw;
  float klb = 0.f;
  bias_math<MODE, LINK>(odd ? R1.th : R0.th, epw, lig < 2, acc.part, klb);
  if constexpr (MODE == MODE_TRAIN) {
    const float c0 = cs0 * R0.io, c1 = cs1 * R1.io;
    acc.kl = fmaf(c0, kl0, fmaf(c1, kl1, fmaf(odd ? c1 : c0, klb, acc.kl)));
  }
}

// per-block constants of the row loop
struct RowConsts {
  float w0, aabs, half_log_a;
  const float* sh_w0;   // S > 1: w0 of every sample
  const float* sh_e0;   //        and the global-bias eps behind it
};

// finish a row: FM reduction over the group, likelihood, outputs
template <int LPE, int CPL, int VEC, int MODE>
__device__ __forceinline__ void finish_row(const KArgs& a, const FwdOut& out, int64_t r, int lig, int C,
                                           const RowConsts& rc, float y, RowAcc<CPL, VEC>& acc, float (&tot)[6]) {
  float q = -acc.zz;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    float qi = 0.f;
#pragma unroll
    for (int t = 0; t < VEC; ++t) qi = fmaf(acc.sz[i].v[t], acc.sz[i].v[t], qi);
    q += (lig + i * LPE < C) ? qi : 0.f;
  }
  const float val = group_sum<LPE>(fmaf(0.5f, q, acc.part));
  if constexpr (MODE == MODE_TRAIN) {
    tot[1] += acc.kl;
    float* srow = out.sumz + ((size_t)a.sample * (size_t)a.B + (size_t)r) * a.d;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int j = lig + i * LPE;
      if (j < C) st_chunk<VEC>(srow + (size_t)j * VEC, acc.sz[i]);
    }
  }
  if (a.S > 1) {                      // uniform: one launch per sample, see the header comment
    if (lig == 0) {
      float m = val;
      if (a.sample > 0) m += out.pred[r];
      if (a.sample + 1 < a.S) {
        out.pred[r] = m;
      } else {
        m *= a.inv_S;
        float gsum = 0.f;
        for (int s = 0; s < a.S; ++s) {
          const float pred = rc.sh_w0[s] + m;
          out.pred[(size_t)s * (size_t)a.B + (size_t)r] = pred;
          if constexpr (MODE == MODE_TRAIN) {
            float ll, dll, at;
            lik_terms(a.lik, y, pred, rc.aabs, rc.half_log_a, ll, dll, at);
            const float g = -a.ll_scale * dll;
            tot[0] += ll; tot[2] += g; tot[3] += at;
            tot[5] = fmaf(rc.sh_e0[s], g, tot[5]);
            gsum += g;
          }
        }
        if constexpr (MODE == MODE_TRAIN) out.grow[r] = gsum;
      }
    }
    return;
  }
  const float pred = rc.w0 + val;
  if (lig == 0) {
    out.pred[r] = pred;
    if constexpr (MODE == MODE_TRAIN) {
      float ll, dll, at;
      lik_terms(a.lik, y, pred, rc.aabs, rc.half_log_a, ll, dll, at);
      const float g = -a.ll_scale * dll;
      tot[0] += ll;
      tot[2] += g;
      tot[3] += at;
      out.grow[r] = g;
    }
  }
}

template <int LPE, int CPL, int VEC, int EPS, int MODE, int FF, bool ID64, int LINK>
__global__ __launch_bounds__(BLOCK) void k_fwd(const KArgs a, const FwdOut out) {
  constexpr int GPB = BLOCK / LPE;
  __shared__ float sh_cs[VFM_MAX_FIELDS];
  __shared__ int64_t sh_hi[VFM_MAX_FIELDS];
  __shared__ float sh_red[6 * 4];
  __shared__ float sh_w0[MAX_SAMPLES], sh_e0[MAX_SAMPLES];

  const int tid = threadIdx.x;
  const int lig = tid % LPE;
  const int F = (FF > 0) ? FF : a.F;
  const int C = (a.d + VEC - 1) / VEC;

  if (MODE == MODE_TRAIN && tid < a.G) {
    // (S > 1: the KL term is formed by the last sample's launch only)
    sh_cs[tid] = (a.sample + 1 < a.S) ? 0.f : (float)(a.group_n[tid] / a.W[tid]);
    sh_hi[tid] = a.group_hi[tid];
  }
  const float alpha = a.scalars[0], m0 = a.scalars[1], s0 = a.scalars[2];
  const float aabs = link_f<LINK>(alpha);
  const RngKey key0 = key_of_step(a, blockIdx.x == 0 && tid == 0);     // (replayable step: the step lives in device memory)
  const RngKey key = key_of_sample(key0, a.sample);   // this launch's sample (key0: sample 0)
  float e0 = 0.f;
  if constexpr (EPS == EPS_TABLE) e0 = a.eps_global[0];
  if constexpr (EPS == EPS_PHILOX) {
    float n[8], nb;
    normal8b(key, 0xFFFFFFFFu, 0u, n, nb);
    e0 = n[0];
  }
  const float w0 = fmaf(link_f<LINK>(s0), e0, m0);
  const float half_log_a = 0.5f * LN2 * __builtin_amdgcn_logf(aabs);
  if (a.S > 1 && a.sample + 1 == a.S && tid < a.S) {   // uniform in a.S: w0 of every sample for the final pass
    float es = 0.f;
    if constexpr (EPS == EPS_TABLE) es = a.eps_global[tid];
    if constexpr (EPS == EPS_PHILOX) {
      float n[8], nb;
      normal8b(key_of_sample(key0, tid), 0xFFFFFFFFu, 0u, n, nb);
      es = n[0];
    }
    sh_e0[tid] = es;
    sh_w0[tid] = fmaf(link_f<LINK>(s0), es, m0);
  }
  if (MODE == MODE_TRAIN || a.S > 1) __syncthreads();
  const RowConsts rc{w0, aabs, half_log_a, sh_w0, sh_e0};

  float tot[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // ll, kl, g, alpha-term, bad ids, sum_s eps0^s * g (S > 1)
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
        consume_row2<LPE, CPL, EPS, MODE, LINK>(key, A0, A1, lig, C, ((int64_t)A0.e < hi0) ? cs0 : cs1,
                                          ((int64_t)A1.e < hi0) ? cs0 : cs1, acc);
        finish_row<LPE, CPL, VEC, MODE>(a, out, r, lig, C, rc, yA, acc, tot);
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
        consume_row2<LPE, CPL, EPS, MODE, LINK>(key, B0, B1, lig, C, ((int64_t)B0.e < hi0) ? cs0 : cs1,
                                          ((int64_t)B1.e < hi0) ? cs0 : cs1, acc);
        finish_row<LPE, CPL, VEC, MODE>(a, out, r, lig, C, rc, yB, acc, tot);
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
        consume_field<LPE, CPL, VEC, EPS, MODE, LINK>(key, A, lig, C, cs_of(A.e, f), acc);
        if (f == F - 1) {
          float y = 0.f;
          if constexpr (MODE == MODE_TRAIN) y = a.y[r];
          finish_row<LPE, CPL, VEC, MODE>(a, out, r, lig, C, rc, y, acc, tot);
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
        consume_field<LPE, CPL, VEC, EPS, MODE, LINK>(key, Bq, lig, C, cs_of(Bq.e, f), acc);
        if (f == F - 1) {
          float y = 0.f;
          if constexpr (MODE == MODE_TRAIN) y = a.y[r];
          finish_row<LPE, CPL, VEC, MODE>(a, out, r, lig, C, rc, y, acc, tot);
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
  block_sum<6>(tot, sh_red);
  if (tid == 0) {
    double* slot = out.partials + VFM_N_PARTIALS * (1 + (size_t)blockIdx.x);
#pragma unroll
    for (int i = 0; i < 6; ++i) slot[i] = (double)tot[i];
    slot[VFM_SLOT_NTERMS] = slot_nterms(a, MODE == MODE_TRAIN && a.sample + 1 == a.S);
    if (blockIdx.x == 0) {
      out.partials[7] = (double)gridDim.x;
      out.partials[VFM_P_REDUCED] = 0.0;       // the sums [0..5] are stale until the slots are reduced
    }
  }
}
