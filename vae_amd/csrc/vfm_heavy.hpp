// vfm_heavy.hpp -- k_heavy: parallel pre-reduction of long occurrence lists (skewed data).
// Included inside `namespace vfm { namespace {` of vfm_abi.hip.
#pragma once

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

