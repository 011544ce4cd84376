// vfm_heavy.hpp -- k_heavy: parallel pre-reduction of long occurrence lists (skewed data).
// Included inside `namespace vfm { namespace {` of vfm_abi.hip.
#pragma once

// Long occurrence lists (skewed batches: a popular item can own 10^4 of the 10^5 rows; small tables: every
// entity of an ML-100K-shape table sits in ~60 rows of a batch): a list longer than the index's heavy-list
// length is cut in work items of at most that length (built with the index, vfm_index.hip).  k_heavy walks
// every work item with its own lane group and STORES the partial record (sum grow, count, 0, 0 | A) of the
// item, and into the header of the entity's record the range [first, last) of its items (words 2, 3, as integers).
// An entity of at most VFM_HEAVY_DIRECT items is finished by the main kernel itself (k_bwd adds the item records
// in item order); for the others k_heavy_sum adds the items IN A FIXED ORDER into the entity's record, which the
// main kernel reads instead of walking the list.  No atomics: the sums have a fixed order, so the step stays
// bitwise reproducible on skewed data too.  Without the split one lane group serialises the whole list
// (Zipf(1.1) items: 3.1 ms instead of 0.2 ms).
// Layout of the scratch table per sample: [n_heavy entity records | n_items item records].
template <int LPE, int CPL, int VEC>
__global__ __launch_bounds__(BLOCK) void k_heavy(const int32_t* __restrict__ items, int n_items,
                                                 const int32_t* __restrict__ occ_rows,
                                                 const float* __restrict__ sumz, const float* __restrict__ grow,
                                                 float* __restrict__ item_acc, int d,
                                                 const int32_t* __restrict__ occ_other, const float* __restrict__ zrec,
                                                 float* __restrict__ heavy_acc, int n_heavy, int n_occ, int B, int64_t T,
                                                 int32_t* __restrict__ status) {
  constexpr int GPB = BLOCK / LPE;
  const int lig = threadIdx.x % LPE;
  const int C = (d + VEC - 1) / VEC;
  const int64_t xs = 4 + (((int64_t)d + 3) & ~(int64_t)3);
  // (a corrupted index is clamped, never followed: vfm_index_t.status counts the clamps)
  int nclamp = 0;
  for (int it = blockIdx.x * GPB + threadIdx.x / LPE; it < n_items; it += gridDim.x * GPB) {
    int slot = items[4 * it], beg = items[4 * it + 1], end = items[4 * it + 2];
    if (slot < 0 || slot >= n_heavy || beg < 0 || end < beg || end > n_occ) { slot = 0; beg = end = 0; ++nclamp; }
    Chunk<VEC> A[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
      for (int t = 0; t < VEC; ++t) A[i].v[t] = 0.f;
    float gs = 0.f;
    constexpr int W = 8 / CPL > 2 ? 8 / CPL : 2;      // occurrences in flight (added in list order: the sum's order is fixed)
    // The row numbers (and, pipelined step, the other entities) of batch k+1 are fetched while the gathers of batch k are in
    // flight: an item is then one memory round trip per batch instead of two (index -> row).  Same sums, same order.
    int rn[W], on[W];
    auto load_idx = [&](int o0) {
#pragma unroll
      for (int u = 0; u < W; ++u) {
        const int oo = o0 + u < end ? o0 + u : beg;         // (no load under a branch: a dead slot re-reads the first one)
        rn[u] = occ_rows[oo];
        on[u] = zrec ? occ_other[oo] : 0;
      }
    };
    if (beg < end) load_idx(beg);
    for (int o = beg; o < end; o += W) {
      int r[W]; float g[W];
      const float* src[W];           // the sumz row, or (pipelined step) the other entity's sample record
#pragma unroll
      for (int u = 0; u < W; ++u) {
        const bool ok = o + u < end;
        r[u] = rn[u];
        if ((unsigned)r[u] >= (unsigned)B) { r[u] = 0; ++nclamp; }
        g[u] = ok ? grow[r[u]] : 0.f;
        gs += g[u];
        int64_t oe = on[u];
        if (oe < 0 || oe >= T) { oe = 0; ++nclamp; }
        src[u] = zrec ? zrec + (size_t)oe * xs + 4 : sumz + (size_t)r[u] * d;
      }
      Chunk<VEC> sv0[W];             // the first chunk's gathers, issued before the next batch's index loads
      const bool has0 = lig < C;
      if (has0) {
#pragma unroll
        for (int u = 0; u < W; ++u) sv0[u] = ld_chunk<VEC>(src[u] + (size_t)lig * VEC);
      }
      load_idx(o + W);               // (past the end: re-reads the first occurrence, never used)
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int j = lig + i * LPE;
        if (j < C) {
          Chunk<VEC> sv[W];
#pragma unroll
          for (int u = 0; u < W; ++u) sv[u] = i == 0 ? sv0[u] : ld_chunk<VEC>(src[u] + (size_t)j * VEC);
#pragma unroll
          for (int u = 0; u < W; ++u)
#pragma unroll
            for (int t = 0; t < VEC; ++t) A[i].v[t] = fmaf(g[u], sv[u].v[t], A[i].v[t]);
        }
      }
    }
    if (lig == 0) {                // the entity's item range, for whoever adds the items up
      int* hdr = reinterpret_cast<int*>(heavy_acc + (size_t)slot * xs);
      if (it == 0 || items[4 * (it - 1)] != slot) hdr[2] = it;
      if (it + 1 == n_items || items[4 * (it + 1)] != slot) hdr[3] = it + 1;
    }
    float* rec = item_acc + (size_t)it * xs;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int j = lig + i * LPE;
      if (j < C) st_chunk<VEC>(rec + 4 + (size_t)j * VEC, A[i]);
    }
    if (lig == 0) *reinterpret_cast<float4*>(rec) = make_float4(gs, (float)(end - beg), 0.f, 0.f);
  }
  if (nclamp != 0 && status) atomicAdd(status, nclamp);
}

// heavy entity `slot`: sum of its work items' records (the items of a slot are consecutive and in list order).
// One WORKGROUP per slot: lane group g adds the items lo+g, lo+g+GPB, ... in that order, then the GPB partial
// records are added in group order through LDS -- a fixed summation tree, whatever the scheduling.
template <int LPE, int CPL, int VEC>
__global__ __launch_bounds__(BLOCK) void k_heavy_sum(const int32_t* __restrict__ items, int n_items, int n_heavy,
                                                     const float* __restrict__ item_acc,
                                                     float* __restrict__ heavy_acc, int d) {
  (void)items;
  constexpr int GPB = BLOCK / LPE;
  __shared__ float sh[BLOCK * CPL * VEC + 2 * GPB];
  const int lig = threadIdx.x % LPE, grp = threadIdx.x / LPE;
  const int C = (d + VEC - 1) / VEC;
  const int64_t xs = 4 + (((int64_t)d + 3) & ~(int64_t)3);
  for (int slot = blockIdx.x; slot < n_heavy; slot += gridDim.x) {
    const int* hdr = reinterpret_cast<const int*>(heavy_acc + (size_t)slot * xs);
    int lo = hdr[2], end = hdr[3];            // the slot's work items (written by k_heavy)
    if (lo < 0 || end < lo || end > n_items) lo = end = 0;      // (never follow a corrupted range: k_heavy / k_bwd report it)
    if (end - lo <= VFM_HEAVY_DIRECT) continue;      // (uniform) few items: the main kernel adds them itself
    Chunk<VEC> A[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
      for (int t = 0; t < VEC; ++t) A[i].v[t] = 0.f;
    float gs = 0.f, cnt = 0.f;
    for (int it = lo + grp; it < end; it += GPB) {
      const float* rec = item_acc + (size_t)it * xs;
      gs += rec[0]; cnt += rec[1];
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int j = lig + i * LPE;
        if (j < C) {
          const Chunk<VEC> t4 = ld_chunk<VEC>(rec + 4 + (size_t)j * VEC);
#pragma unroll
          for (int t = 0; t < VEC; ++t) A[i].v[t] += t4.v[t];
        }
      }
    }
    __syncthreads();                          // (LDS of the previous slot has been read)
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
      for (int t = 0; t < VEC; ++t) sh[((grp * CPL + i) * VEC + t) * LPE + lig] = A[i].v[t];
    if (lig == 0) { sh[BLOCK * CPL * VEC + 2 * grp] = gs; sh[BLOCK * CPL * VEC + 2 * grp + 1] = cnt; }
    __syncthreads();
    if (grp == 0) {
      float* out = heavy_acc + (size_t)slot * xs;
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int j = lig + i * LPE;
        Chunk<VEC> tot;
#pragma unroll
        for (int t = 0; t < VEC; ++t) {
          float acc = 0.f;
          for (int g2 = 0; g2 < GPB; ++g2) acc += sh[((g2 * CPL + i) * VEC + t) * LPE + lig];
          tot.v[t] = acc;
        }
        if (j < C) st_chunk<VEC>(out + 4 + (size_t)j * VEC, tot);
      }
      if (lig == 0) {
        float tg = 0.f, tc = 0.f;
        for (int g2 = 0; g2 < GPB; ++g2) { tg += sh[BLOCK * CPL * VEC + 2 * g2]; tc += sh[BLOCK * CPL * VEC + 2 * g2 + 1]; }
        *reinterpret_cast<float2*>(out) = make_float2(tg, tc);      // (words 2, 3 keep the item range)
      }
    }
  }
}
