// vfm_index.hip -- vfm_build_index: the inverted index of one batch (entity -> batch rows), built on the GPU.
//
// Replaces what the reference does with torch.unique(x, return_inverse, return_counts) three times per
// step (vfm-torch.py:190-192): the kernels never need `unique`, only, per entity, the list of the batch
// rows that contain it (the backward walks it) -- plus the batch normalisers of :305-306, which read the same ids.
//
// Method: a STABLE least-significant-digit radix sort of the B*F (entity id, position) pairs by entity id, up to 9 bits
// per pass, ceil(log2(T) / 9) passes (two at the ML-20M shape); stable means an entity's rows come out in row order, so
// the backward's sums have a fixed order and the whole step stays bitwise reproducible.  The whole build is latency-bound
// (200 K keys: 3 MB per pass), so it is organised around the NUMBER of dependent launches and the dependent memory round
// trips inside each (seven launches at the ML-20M shape; sixteen before round 4):
//   memset           the per-tile-group digit counts of every pass (16 KB)
//   k_index_keys     ids -> uint32 keys (range-checked; out-of-range ids are counted and clamped to 0 like the forward
//                    does), values = positions r*F + f; the first pass's per-tile digit counts; the workgroup's share of
//                    the batch normalisers W_f = sum_r inv_occ[x_rf] (fp64, fixed order)
//   k_radix_scatter  one per pass: stable rank inside the tile (per-wave digit matching with ballots + a scan over the
//                    tile's 32 sub-tiles in LDS) + the tile's base, formed by the workgroup itself from TWO levels of
//                    digit counts (per tile, per group of G tiles: <= NB/G + G - 1 rows to add, not NB); the LAST pass
//                    writes occ_rows / occ_other directly and leaves the first position of every leading digit
//   k_radix_hist     between two passes: the next pass's digit counts of every tile of the scattered keys.  (Counting
//                    them inside the scatter with one atomic per key was tried: 400 K scattered device atomics took 45 us,
//                    each is a 64-byte request at the memory side; this kernel reads the keys coalesced: 4 us.)
//   k_index_count    per chunk of 1024 entities: occ_ptr = lower bound of e among the sorted keys -- the chunk's key range
//                    (between the first positions of its leading digits) is staged in LDS and searched there -- and the
//                    chunk's counts of heavy entities, work items, entities present, longest item list
//   k_index_write    every chunk adds up the chunks before it itself, then the heavy lists and the touched list; the last
//                    chunk leaves the totals in `counts` and finishes W
// Integer atomics only, and only where the total is what matters: two builds of one batch give identical buffers.
// gfx950 only, wave = 64.
#include <string.h>

#include "vfm_args.hpp"

namespace vfm {
namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;       // keys per workgroup per pass
constexpr int RS_SUB = RS_ITEMS * (RS_THREADS / 64); // 64-key sub-tiles of a tile, in key order
constexpr int RS_MAXBITS = 9;
constexpr int RS_MAXD = 1 << RS_MAXBITS;             // digits per pass, at most
constexpr int RS_MAXPASS = 4;                        // ceil(32 / 9)
// Entities per workgroup of the compaction launches = its threads.  256, not 1024: a build on a side stream runs BESIDE the
// step's persistent kernels (k_bwd: 4 waves of 128 VGPRs per SIMD; under VFM_FLAG_SHARE_GPU half the CUs keep one
// workgroup slot free = one wave of <= 128 VGPRs per SIMD).  A 1024-thread workgroup (4 waves per SIMD, 160-224 VGPRs) fits
// nowhere until the step's kernel ENDS: traced at 86 us for 8 us of work (tools/stream_timeline.py).
constexpr int HV_CHUNK = 256;
constexpr int HV_SELF = 1024;                        // chunks up to which a compaction workgroup adds up its predecessors itself
constexpr int NW_MAXF = 4;                           // fields up to which W is formed inside k_index_keys

// ---- geometry of one build: shared by vfm_index_workspace_bytes and vfm_build_index ----
struct Geo {
  int64_t n, n4, NB, G, NS, NBH;
  int passes, rb[RS_MAXPASS], shift[RS_MAXPASS];
  // offsets into the workspace, in 4-byte words
  int64_t k0, v0, k1, v1, zero_lo, L1[RS_MAXPASS], L2[RS_MAXPASS], zero_hi, excl, blk, bad, wpart, total;
};

int key_bits(int64_t T) {
  int bits = 1;
  while (bits < 32 && ((int64_t)1 << bits) < T) ++bits;
  return bits;
}

Geo geometry(int64_t n, int64_t T) {
  Geo g;
  memset(&g, 0, sizeof(g));
  g.n = n; g.n4 = (n + 3) & ~(int64_t)3;
  g.NB = (n + RS_TILE - 1) / RS_TILE;
  g.G = 32;
  while (g.G * g.G < g.NB) g.G *= 2;                 // rows a scatter workgroup adds up: NB/G + G - 1 <= ~2 sqrt(NB)
  g.NS = (g.NB + g.G - 1) / g.G;
  g.NBH = (T + HV_CHUNK - 1) / HV_CHUNK;
  const int bits = key_bits(T);
  g.passes = (bits + RS_MAXBITS - 1) / RS_MAXBITS;
  int sh = 0;
  for (int p = 0; p < g.passes; ++p) {
    g.rb[p] = bits / g.passes + (p < bits % g.passes ? 1 : 0);
    if (g.rb[p] < 2) g.rb[p] = 2;                     // (T <= 2: the extra bit is zero in every key)
    g.shift[p] = sh;
    sh += g.rb[p];
  }
  int64_t o = 0;
  g.k0 = o; o += g.n4; g.v0 = o; o += g.n4; g.k1 = o; o += g.n4; g.v1 = o; o += g.n4;
  for (int p = 0; p < g.passes; ++p) { g.L1[p] = o; o += g.NB * ((int64_t)1 << g.rb[p]); }
  g.zero_lo = o;
  for (int p = 0; p < g.passes; ++p) { g.L2[p] = o; o += g.NS * ((int64_t)1 << g.rb[p]); }
  g.zero_hi = o;
  g.excl = o; o += RS_MAXD + 4;
  g.blk = o; o += 5 * g.NBH + 8;
  g.bad = o; o += (g.NB + 4) & ~(int64_t)3;
  o = (o + 1) & ~(int64_t)1;                          // (doubles: 8-byte aligned)
  g.wpart = o; o += 2 * NW_MAXF * (g.NB + 1);
  g.total = o + 16;
  return g;
}

__device__ __forceinline__ void load_id(const void* __restrict__ x, int id64, int64_t i, uint32_t& lo, uint32_t& hi) {
  if (id64) {
    const uint2 t = reinterpret_cast<const uint2*>(x)[i];
    lo = t.x; hi = t.y;
  } else {
    lo = reinterpret_cast<const uint32_t*>(x)[i];
    hi = (lo >> 31) ? 0xFFFFFFFFu : 0u;
  }
}

// digit counts of one tile into its row of L1 (plain stores) and its group's row of L2 (integer atomics on consecutive
// addresses).  One LDS add per group of lanes of a wave with the same digit: a run of one popular id is one add.
__device__ __forceinline__ void tile_digit_counts(const uint32_t (&dg)[RS_ITEMS], int base, int n, int rb, unsigned int* sh_hist,
                                                  uint32_t* __restrict__ L1, uint32_t* __restrict__ L2, int G) {
  const int tid = threadIdx.x, ND = 1 << rb;
#pragma unroll
  for (int k = 0; k < RS_ITEMS; ++k) {
    const bool valid = base + k * RS_THREADS + tid < n;
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RS_MAXBITS; ++b) {
      if (b < rb) {
        const bool bit = (dg[k] >> b) & 1u;
        const unsigned long long bb = __ballot(bit);
        m &= bit ? bb : ~bb;
      }
    }
    if (valid && (m & ((1ull << (tid & 63)) - 1ull)) == 0ull) atomicAdd(&sh_hist[dg[k]], (unsigned int)__popcll(m));
  }
  __syncthreads();
  uint32_t* row1 = L1 + (size_t)blockIdx.x * ND;
  uint32_t* row2 = L2 + (size_t)(blockIdx.x / G) * ND;
  for (int i = tid; i < ND; i += RS_THREADS) {
    const unsigned int c = sh_hist[i];
    row1[i] = c;
    if (c) atomicAdd(&row2[i], c);
  }
}

// ids -> keys / positions, the first pass's digit counts of the tile, the tile's share of W and of the bad-id count
__global__ __launch_bounds__(RS_THREADS) void k_index_keys(const void* __restrict__ x, int id64, int n, uint32_t T32, int F,
                                                           uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                           int rb0, uint32_t* __restrict__ L1, uint32_t* __restrict__ L2, int G,
                                                           const float* __restrict__ inv_occ, double* __restrict__ wpart,
                                                           uint32_t* __restrict__ badpart) {
  __shared__ unsigned int sh_hist[RS_MAXD];
  __shared__ double sh_w[NW_MAXF][RS_THREADS / 64];
  __shared__ unsigned int sh_bad[RS_THREADS / 64];
  const int tid = threadIdx.x, ND = 1 << rb0;
  for (int i = tid; i < ND; i += RS_THREADS) sh_hist[i] = 0;
  __syncthreads();
  const int base = blockIdx.x * RS_TILE;
  unsigned int bad = 0;
  const bool want_w = inv_occ != nullptr;
  uint32_t lo[RS_ITEMS], hi[RS_ITEMS], kd[RS_ITEMS];
  float io[RS_ITEMS];
  // all ids first, then all 1/occ gathers: two dependent round trips for the tile, not sixteen (no load under a branch:
  // a slot past the end re-reads the last id and is masked)
#pragma unroll
  for (int k = 0; k < RS_ITEMS; ++k) {
    const int i = base + k * RS_THREADS + tid;
    load_id(x, id64, i < n ? i : n - 1, lo[k], hi[k]);
  }
#pragma unroll
  for (int k = 0; k < RS_ITEMS; ++k) {
    const bool ok = hi[k] == 0u && lo[k] < T32;
    io[k] = want_w ? inv_occ[ok ? lo[k] : 0u] : 0.f;
  }
  double w[NW_MAXF] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < RS_ITEMS; ++k) {
    const int i = base + k * RS_THREADS + tid;
    const bool valid = i < n;
    const bool ok = hi[k] == 0u && lo[k] < T32;
    const uint32_t key = ok ? lo[k] : 0u;
    kd[k] = key & (uint32_t)(ND - 1);
    if (valid) {
      bad += ok ? 0u : 1u;
      keys[i] = key;
      vals[i] = (uint32_t)i;
      const double v = ok ? (double)io[k] : 0.0;        // (ids out of range add nothing, as in vfm_batch_norms)
      const int f = i % F;
#pragma unroll
      for (int q = 0; q < NW_MAXF; ++q) w[q] += (q == f) ? v : 0.0;
    }
  }
  // bad ids + W: wave sums in a fixed tree, then the waves in order
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) bad += __shfl_xor(bad, m, 64);
  if ((tid & 63) == 0) sh_bad[tid >> 6] = bad;
  if (want_w) {
#pragma unroll
    for (int q = 0; q < NW_MAXF; ++q) {
      if (q < F) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) w[q] += __shfl_xor(w[q], m, 64);
      }
      if ((tid & 63) == 0) sh_w[q][tid >> 6] = w[q];
    }
  }
  tile_digit_counts(kd, base, n, rb0, sh_hist, L1, L2, G);      // (its barrier also covers sh_bad / sh_w)
  if (tid == 0) {
    unsigned int t = 0;
    for (int v = 0; v < RS_THREADS / 64; ++v) t += sh_bad[v];
    badpart[blockIdx.x] = t;
  }
  if (want_w && tid < NW_MAXF) {
    double t = 0.0;
    for (int v = 0; v < RS_THREADS / 64; ++v) t += sh_w[tid][v];
    wpart[(size_t)blockIdx.x * NW_MAXF + tid] = t;
  }
}

// between two passes: the digit counts of every tile of the freshly scattered keys
__global__ __launch_bounds__(RS_THREADS) void k_radix_hist(const uint32_t* __restrict__ keys, int n, int shift, int rb,
                                                           uint32_t* __restrict__ L1, uint32_t* __restrict__ L2, int G) {
  __shared__ unsigned int sh_hist[RS_MAXD];
  const int tid = threadIdx.x, ND = 1 << rb;
  for (int i = tid; i < ND; i += RS_THREADS) sh_hist[i] = 0;
  __syncthreads();
  const int base = blockIdx.x * RS_TILE;
  uint32_t dg[RS_ITEMS];
#pragma unroll
  for (int k = 0; k < RS_ITEMS; ++k) {
    const int i = base + k * RS_THREADS + tid;
    dg[k] = (keys[i < n ? i : n - 1] >> shift) & (uint32_t)(ND - 1);
  }
  tile_digit_counts(dg, base, n, rb, sh_hist, L1, L2, G);
}

// One radix pass.  L1 [NB][ND] / L2 [NS][ND]: this pass's digit counts per tile / per group of G tiles.
// LAST: the sorted positions go out as occ_rows / occ_other (+ the sorted keys, for the lower-bound searches of
// k_index_count) and workgroup 0 leaves excl[dg] = first position of leading digit dg (excl[ND] = n).
template <bool LAST>
__global__ __launch_bounds__(RS_THREADS) void k_radix_scatter(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                              int n, int shift, int rb, const uint32_t* __restrict__ L1,
                                                              const uint32_t* __restrict__ L2, int G, int NS,
                                                              uint32_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                              uint32_t* __restrict__ excl,
                                                              int32_t* __restrict__ occ_rows, int32_t* __restrict__ occ_other,
                                                              const void* __restrict__ x, int id64, int F, int64_t T) {
  __shared__ uint16_t sub[RS_SUB][RS_MAXD];          // [sub-tile][digit]: count, then start inside the tile
  __shared__ uint32_t sh_bef[RS_MAXD];               // digit: keys of it in the tiles before this one
  __shared__ uint32_t sh_tot[RS_MAXD];               // digit: keys of it anywhere, then (scanned) keys of smaller digits
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ND = 1 << rb;
  const uint32_t dmask = (uint32_t)(ND - 1);
  {
    uint4* z = reinterpret_cast<uint4*>(&sub[0][0]);
    for (int i = tid; i < RS_SUB * RS_MAXD * 2 / 16; i += RS_THREADS) z[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < RS_MAXD; i += RS_THREADS) { sh_bef[i] = 0u; sh_tot[i] = 0u; }
  }
  __syncthreads();
  const int me = blockIdx.x;
  const int tbase = me * RS_TILE;
  uint32_t key[RS_ITEMS], val[RS_ITEMS];
  int lower[RS_ITEMS];
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int k = 0; k < RS_ITEMS; ++k) {
    const int i = tbase + k * RS_THREADS + tid;     // = tbase + (k * 4 + wave) * 64 + lane: sub-tiles are in key order
    const bool valid = i < n;
    key[k] = valid ? kin[i] : 0u;
    val[k] = valid ? vin[i] : 0u;
  }
  // ---- this tile's base per digit: the keys of smaller digits anywhere + the keys of the digit in the tiles before.
  // Rows to add: every group row of L2 (total; the groups before mine also count as `before`) and the L1 rows of the
  // tiles of my own group before me -- NB/G + G - 1 rows at most.  Four digits per load, rows dealt over the row groups
  // of the workgroup, eight loads in flight per thread; the row groups meet in LDS (integer adds: order-free).
  {
    const int ms = me / G;                            // my group
    const int nrow = NS + (me - ms * G);
    const int lanes = ND >> 2;                        // threads covering one row (4 digits each): 1 .. 128
    const int RG = RS_THREADS / lanes;                // row groups: >= 2
    const int q = tid % lanes, rg = tid / lanes;
    uint4 bef = make_uint4(0u, 0u, 0u, 0u), tot = make_uint4(0u, 0u, 0u, 0u);
    for (int r0 = rg; r0 < nrow; r0 += RG * 8) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {                   // (no load under a branch: a dead slot re-reads row 0 and is masked)
        const int r = r0 + u * RG;
        const bool live = r < nrow, grp = r < NS;
        const uint32_t* row = !live ? L2 : (grp ? L2 + (size_t)r * ND : L1 + (size_t)(ms * G + (r - NS)) * ND);
        v[u] = reinterpret_cast<const uint4*>(row)[q];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = r0 + u * RG;
        const bool live = r < nrow, grp = r < NS;
        const uint32_t mt = (live && grp) ? 0xFFFFFFFFu : 0u, mb = (live && (!grp || r < ms)) ? 0xFFFFFFFFu : 0u;
        tot.x += v[u].x & mt; tot.y += v[u].y & mt; tot.z += v[u].z & mt; tot.w += v[u].w & mt;
        bef.x += v[u].x & mb; bef.y += v[u].y & mb; bef.z += v[u].z & mb; bef.w += v[u].w & mb;
      }
    }
    atomicAdd(&sh_bef[4 * q + 0], bef.x); atomicAdd(&sh_bef[4 * q + 1], bef.y);
    atomicAdd(&sh_bef[4 * q + 2], bef.z); atomicAdd(&sh_bef[4 * q + 3], bef.w);
    atomicAdd(&sh_tot[4 * q + 0], tot.x); atomicAdd(&sh_tot[4 * q + 1], tot.y);
    atomicAdd(&sh_tot[4 * q + 2], tot.z); atomicAdd(&sh_tot[4 * q + 3], tot.w);
  }
  // ---- rank inside the tile: lanes of a wave holding the same digit (ballots), counted per 64-key sub-tile ----
#pragma unroll
  for (int k = 0; k < RS_ITEMS; ++k) {
    const int i = tbase + k * RS_THREADS + tid;
    const bool valid = i < n;
    const uint32_t dg = (key[k] >> shift) & dmask;
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RS_MAXBITS; ++b) {
      if (b < rb) {                                  // (uniform)
        const bool bit = (dg >> b) & 1u;
        const unsigned long long bb = __ballot(bit);
        m &= bit ? bb : ~bb;
      }
    }
    m = valid ? m : 0ull;
    lower[k] = __popcll(m & lt);
    if (valid && lower[k] == 0) sub[k * (RS_THREADS / 64) + wave][dg] = (uint16_t)__popcll(m);
  }
  __syncthreads();
  if (tid < 64) {                                     // exclusive scan of the ND digit totals: 8 per lane + wave scan
    uint32_t v8[8], t8 = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int dg = 8 * tid + i; v8[i] = dg < ND ? sh_tot[dg] : 0u; t8 += v8[i]; }
    uint32_t inc = t8;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o, 64);
      if (tid >= o) inc += t;
    }
    uint32_t run = inc - t8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int dg = 8 * tid + i;
      if (dg < ND) {
        sh_tot[dg] = run;
        if (LAST && me == 0) excl[dg] = run;
      }
      run += v8[i];
    }
    if (LAST && me == 0 && tid == 63) excl[ND] = (uint32_t)n;
  }
  __syncthreads();
  // digit dg (two per thread): global base of this tile, then the sub-tiles' starts inside the tile
  for (int dg = tid; dg < ND; dg += RS_THREADS) {
    sh_bef[dg] += sh_tot[dg];
    uint32_t run = 0;
#pragma unroll 8
    for (int s = 0; s < RS_SUB; ++s) { const uint32_t c = sub[s][dg]; sub[s][dg] = (uint16_t)run; run += c; }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < RS_ITEMS; ++k) {
    const int i = tbase + k * RS_THREADS + tid;
    if (i < n) {
      const uint32_t dg = (key[k] >> shift) & dmask;
      const uint32_t pos = sh_bef[dg] + (uint32_t)sub[k * (RS_THREADS / 64) + wave][dg] + (uint32_t)lower[k];
      kout[pos] = key[k];
      if constexpr (LAST) {
        occ_rows[pos] = (int32_t)(val[k] / (uint32_t)F);
        if (occ_other) {       // two fields: the entity in the OTHER column of the occurrence's row (clamped like the keys)
          uint32_t lo, hi;
          load_id(x, id64, (int64_t)(val[k] ^ 1u), lo, hi);
          occ_other[pos] = (hi == 0u && (int64_t)lo < T) ? (int32_t)lo : 0;
        }
      } else {
        vout[pos] = val[k];
      }
    }
  }
}

// ---- compaction over the entities, in id order: occ_ptr, the heavy lists with their work items, the touched list ----
// blk rows: 0 occurrences, 1 heavy entities, 2 work items, 3 entities present, 4 most work items of one entity (a max)
__device__ __forceinline__ void ent_counts(uint32_t c, int L, int THR, uint32_t& hv, uint32_t& it, uint32_t& any) {
  any = c > 0 ? 1u : 0u;
  hv = c > (uint32_t)THR ? 1u : 0u;
  it = hv ? (c + (uint32_t)L - 1) / (uint32_t)L : 0u;
}

// SEARCH: occ_ptr[e] = lower bound of e among the sorted keys, written here.  The chunk's entities [e0, e0 + 1024] can
// only be found between the first position of e0's leading digit and the first position past (e0 + 1024)'s (excl, from
// the last scatter): that range of keys -- about 1024 n / T of them -- is staged in LDS with one coalesced round of loads
// and every thread searches there (a global binary search is ~10 dependent memory round trips); a longer range (skewed
// ids) is searched in global memory.  Else occ_ptr is given (vfm_rebuild_heavy).
constexpr int IC_STAGE = 6144;                       // keys staged per chunk, at most (24 KB)
template <bool SEARCH>
__global__ __launch_bounds__(HV_CHUNK) void k_index_count(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ excl,
                                                          int n, int shift_top, int ND_top, int32_t* __restrict__ occ_ptr,
                                                          int64_t T, int L, int THR, uint32_t* __restrict__ blk, int NBH) {
  __shared__ uint32_t sh[5][HV_CHUNK / 64];
  __shared__ int sh_lo[SEARCH ? HV_CHUNK + 1 : 1];
  __shared__ uint32_t sh_keys[SEARCH ? IC_STAGE : 1];
  const int tid = threadIdx.x;
  const int64_t e0 = blockIdx.x * (int64_t)HV_CHUNK;
  const int64_t e = e0 + tid;
  uint32_t c = 0;
  if constexpr (SEARCH) {
    // key range of the chunk: [first position of digit(e0), first position of digit(e0 + 1024) + 1)
    auto first_of = [&](int64_t ee, bool past) -> int {
      if (n == 0) return 0;
      const int64_t b = (int64_t)((uint64_t)ee >> shift_top) + (past ? 1 : 0);
      return b >= ND_top ? n : (int)excl[b];
    };
    const int64_t e_hi = e0 + HV_CHUNK < T ? e0 + HV_CHUNK : T;
    const int r_lo = first_of(e0, false), r_hi = first_of(e_hi, true);
    const bool staged = r_hi - r_lo <= IC_STAGE;          // (uniform)
    if (staged)
      for (int i = tid; i < r_hi - r_lo; i += HV_CHUNK) sh_keys[i] = keys[r_lo + i];
    __syncthreads();
    auto find = [&](int64_t ee) -> int {
      if (ee >= T || n == 0) return ee > 0 ? n : 0;
      if (staged) {
        int lo = 0, hi = r_hi - r_lo;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (sh_keys[mid] < (uint32_t)ee) lo = mid + 1; else hi = mid;
        }
        return r_lo + lo;
      }
      const int b = (int)((uint64_t)ee >> shift_top);
      if (b >= ND_top) return n;
      int lo = (int)excl[b], hi = (int)excl[b + 1];
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < (uint32_t)ee) lo = mid + 1; else hi = mid;
      }
      return lo;
    };
    sh_lo[tid] = find(e);
    if (tid == 0) sh_lo[HV_CHUNK] = find(e + HV_CHUNK);
    __syncthreads();
    if (e <= T) occ_ptr[e] = sh_lo[tid];
    if (tid == 0 && e + HV_CHUNK == T) occ_ptr[T] = sh_lo[HV_CHUNK];      // (T a multiple of the chunk: no thread has e == T)
    if (e < T) c = (uint32_t)(sh_lo[tid + 1] - sh_lo[tid]);
  } else {
    if (e < T) c = (uint32_t)(occ_ptr[e + 1] - occ_ptr[e]);
  }
  uint32_t v[5];
  v[0] = c;
  ent_counts(c, L, THR, v[1], v[2], v[3]);
  v[4] = v[2];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += __shfl_xor(v[r], m, 64);
    const uint32_t o = __shfl_xor(v[4], m, 64);
    v[4] = o > v[4] ? o : v[4];
  }
  if ((tid & 63) == 0)
    for (int r = 0; r < 5; ++r) sh[r][tid >> 6] = v[r];
  __syncthreads();
  if (tid < 5) {
    uint32_t t = 0;
    for (int w = 0; w < HV_CHUNK / 64; ++w) t = tid == 4 ? (sh[4][w] > t ? sh[4][w] : t) : t + sh[tid][w];
    blk[(size_t)tid * NBH + blockIdx.x] = t;
  }
}

// many chunks (NBH > HV_SELF): exclusive scans of the four count rows in place (one wave each), their totals and the
// maximum of row 4 behind the table (blk[5 * NBH + 0..4])
__global__ __launch_bounds__(320) void k_index_scan(uint32_t* __restrict__ blk, int NBH) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t* row = blk + (size_t)w * NBH;
  uint32_t carry = 0;
  for (int i0 = 0; i0 < NBH; i0 += 64) {
    const int i = i0 + lane;
    const uint32_t v = i < NBH ? row[i] : 0u;
    if (w == 4) {
      uint32_t mx = v;
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) { const uint32_t o = __shfl_xor(mx, m, 64); mx = o > mx ? o : mx; }
      carry = mx > carry ? mx : carry;
      continue;
    }
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (i < NBH) row[i] = carry + inc - v;
    carry += __shfl(inc, 63, 64);
  }
  if (lane == 0) blk[(size_t)5 * NBH + w] = carry;
}

template <bool SCANNED>      // SCANNED: k_index_scan ran; else every workgroup adds up the chunks before it itself
__global__ __launch_bounds__(HV_CHUNK) void k_index_write(const int32_t* __restrict__ occ_ptr, int64_t T, int L, int THR,
                                                          const uint32_t* __restrict__ blk, int NBH,
                                                          int32_t* __restrict__ heavy_ids, int32_t* __restrict__ items,
                                                          int cap_h, int cap_i, int32_t* __restrict__ touched_ids,
                                                          int32_t* __restrict__ counts, const uint32_t* __restrict__ badpart,
                                                          int NB, const double* __restrict__ wpart, double* __restrict__ W,
                                                          int F) {
  __shared__ uint32_t sh[3][HV_CHUNK / 64];
  __shared__ uint32_t sh_bt[7][HV_CHUNK / 64];
  __shared__ uint32_t sh_base[7];                    // 0..2: heavy / items / present in the chunks before me; 3..5: totals; 6: max items
  const int tid = threadIdx.x, me = blockIdx.x;
  const bool last = me == (int)gridDim.x - 1;
  if constexpr (SCANNED) {
    if (tid < 3) { sh_base[tid] = blk[(size_t)(1 + tid) * NBH + me]; sh_base[3 + tid] = blk[(size_t)5 * NBH + 1 + tid]; }
    if (tid == 3) sh_base[6] = blk[(size_t)5 * NBH + 4];
  } else {
    uint32_t b[7] = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
    for (int j = tid; j < NBH; j += HV_CHUNK) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const uint32_t v = blk[(size_t)(1 + r) * NBH + j];
        b[r] += j < me ? v : 0u;
        b[3 + r] += v;
      }
      const uint32_t v4 = blk[(size_t)4 * NBH + j];
      b[6] = v4 > b[6] ? v4 : b[6];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
#pragma unroll
      for (int r = 0; r < 6; ++r) b[r] += __shfl_xor(b[r], m, 64);
      const uint32_t o = __shfl_xor(b[6], m, 64);
      b[6] = o > b[6] ? o : b[6];
    }
    if ((tid & 63) == 0)
      for (int r = 0; r < 7; ++r) sh_bt[r][tid >> 6] = b[r];
    __syncthreads();
    if (tid < 7) {
      uint32_t t = 0;
      for (int w = 0; w < HV_CHUNK / 64; ++w) t = tid == 6 ? (sh_bt[6][w] > t ? sh_bt[6][w] : t) : t + sh_bt[tid][w];
      sh_base[tid] = t;
    }
  }
  const int64_t e = me * (int64_t)HV_CHUNK + tid;
  int beg = 0;
  uint32_t c = 0;
  if (e < T) { beg = occ_ptr[e]; c = (uint32_t)(occ_ptr[e + 1] - beg); }
  uint32_t v[3];
  ent_counts(c, L, THR, v[0], v[1], v[2]);
  uint32_t inc[3] = {v[0], v[1], v[2]};              // inclusive scans inside the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const uint32_t t = __shfl_up(inc[r], o, 64);
      if ((tid & 63) >= o) inc[r] += t;
    }
  }
  if ((tid & 63) == 63)
    for (int r = 0; r < 3; ++r) sh[r][tid >> 6] = inc[r];
  __syncthreads();
  if (tid < 3) {
    uint32_t r = 0;
    for (int w = 0; w < HV_CHUNK / 64; ++w) { const uint32_t t = sh[tid][w]; sh[tid][w] = r; r += t; }
  }
  __syncthreads();
  uint32_t ex[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) ex[r] = sh_base[r] + sh[r][tid >> 6] + inc[r] - v[r];
  if (v[2] && touched_ids) touched_ids[ex[2]] = (int32_t)e;
  if (v[0]) {
    const uint32_t slot = ex[0];
    uint32_t it = ex[1];
    if (slot < (uint32_t)cap_h) heavy_ids[slot] = (int32_t)e;
    const int end = beg + (int)c;
    for (int o = beg; o < end; o += L, ++it) {
      if (it < (uint32_t)cap_i) {
        const int oe = (o + L < end) ? o + L : end;
        *reinterpret_cast<int4*>(items + 4 * (size_t)it) = make_int4((int)slot, o, oe, 0);
      }
    }
  }
  if (last) {
    // the build's one readback: (bad ids, heavy entities, work items, entities present, most items of one entity, 0, 0, 0)
    if (tid >= 1 && tid < 4) counts[tid] = (int32_t)sh_base[2 + tid];
    if (tid == 4) counts[4] = (int32_t)sh_base[6];
    if (tid >= 5 && tid < 8) counts[tid] = 0;
    if (tid < 64) {             // ids out of range, summed over the key tiles
      uint32_t b = 0;
      if (badpart)
        for (int j = tid; j < NB; j += 64) b += badpart[j];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) b += __shfl_xor(b, m, 64);
      if (tid == 0) counts[0] = (int32_t)b;
    }
    if (W) {                    // W_f: a wave adds the tiles' shares of field f -- every lane its tiles in order, then the
      const int lane = tid & 63; //  lanes in a fixed tree: reproducible
      for (int f = tid >> 6; f < F && f < NW_MAXF; f += HV_CHUNK / 64) {
        double t = 0.0;
        if (wpart)                  // (an empty batch has no tiles: W = 0)
          for (int j = lane; j < NB; j += 64) t += wpart[(size_t)j * NW_MAXF + f];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);
        if (lane == 0) W[f] = t;
      }
    }
  }
}

// ---- rows of the look-ahead step: the entities of batch A or of batch B, in id order ----
__device__ __forceinline__ bool in_either(const int32_t* a, const int32_t* b, int64_t e, int64_t T) {
  return e < T && (a[e + 1] != a[e] || b[e + 1] != b[e]);
}
__global__ __launch_bounds__(HV_CHUNK) void k_union_count(const int32_t* __restrict__ occ_a, const int32_t* __restrict__ occ_b,
                                                          int64_t T, uint32_t* __restrict__ blk) {
  __shared__ uint32_t sh[HV_CHUNK / 64];
  const int64_t e = blockIdx.x * (int64_t)HV_CHUNK + threadIdx.x;
  const unsigned long long m = __ballot(in_either(occ_a, occ_b, e, T));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < HV_CHUNK / 64; ++w) t += sh[w];
    blk[blockIdx.x] = t;
  }
}
__global__ __launch_bounds__(64) void k_union_scan(uint32_t* __restrict__ blk, int NBH, int32_t* __restrict__ count) {
  const int lane = threadIdx.x;
  uint32_t carry = 0;
  for (int i0 = 0; i0 < NBH; i0 += 64) {
    const int i = i0 + lane;
    const uint32_t v = i < NBH ? blk[i] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (i < NBH) blk[i] = carry + inc - v;
    carry += __shfl(inc, 63, 64);
  }
  if (lane == 0) count[0] = (int32_t)carry;
}
template <bool SCANNED>      // SCANNED: k_union_scan ran; else every workgroup adds up the chunks before it itself
__global__ __launch_bounds__(HV_CHUNK) void k_union_write(const int32_t* __restrict__ occ_a, const int32_t* __restrict__ occ_b,
                                                          int64_t T, const uint32_t* __restrict__ blk, int NBH,
                                                          int32_t* __restrict__ rows, int32_t* __restrict__ count) {
  __shared__ uint32_t sh[HV_CHUNK / 64];
  __shared__ uint32_t sh_bt[2][HV_CHUNK / 64];
  __shared__ uint32_t sh_base[2];
  const int tid = threadIdx.x, lane = tid & 63, me = blockIdx.x;
  if constexpr (SCANNED) {
    if (tid == 0) sh_base[0] = blk[me];
  } else {
    uint32_t b0 = 0, b1 = 0;
    for (int j = tid; j < NBH; j += HV_CHUNK) { const uint32_t v = blk[j]; b0 += j < me ? v : 0u; b1 += v; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { b0 += __shfl_xor(b0, m, 64); b1 += __shfl_xor(b1, m, 64); }
    if (lane == 0) { sh_bt[0][tid >> 6] = b0; sh_bt[1][tid >> 6] = b1; }
    __syncthreads();
    if (tid < 2) {
      uint32_t t = 0;
      for (int w = 0; w < HV_CHUNK / 64; ++w) t += sh_bt[tid][w];
      sh_base[tid] = t;
      if (tid == 1 && me == 0) count[0] = (int32_t)t;
    }
  }
  const int64_t e = me * (int64_t)HV_CHUNK + tid;
  const bool in = in_either(occ_a, occ_b, e, T);
  const unsigned long long m = __ballot(in);
  if (lane == 0) sh[tid >> 6] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t wbase = 0;
  for (int w = 0; w < (tid >> 6); ++w) wbase += sh[w];
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  if (in) rows[sh_base[0] + wbase + (uint32_t)__popcll(m & lt)] = (int32_t)e;
}

// the two compaction launches over an occ_ptr that is already there (the second half of a build; vfm_rebuild_heavy)
int compaction_write(int64_t NBH64, const int32_t* occ_ptr, int64_t T, int L, int THR, uint32_t* blk, int32_t* heavy_ids,
                     int32_t* items, int64_t cap_heavy, int64_t cap_items, int32_t* touched_ids, int32_t* counts,
                     const uint32_t* badpart, int NB, const double* wpart, double* W, int F, hipStream_t st) {
  const int NBH = (int)NBH64;
  const int ch = (int)(cap_heavy > 0x7FFFFFFF ? 0x7FFFFFFF : cap_heavy), ci = (int)(cap_items > 0x7FFFFFFF ? 0x7FFFFFFF : cap_items);
  if (NBH > HV_SELF) {
    hipLaunchKernelGGL(k_index_scan, dim3(1), dim3(320), 0, st, blk, NBH);
    hipLaunchKernelGGL(k_index_write<true>, dim3(NBH), dim3(HV_CHUNK), 0, st, occ_ptr, T, L, THR, blk, NBH, heavy_ids, items,
                       ch, ci, touched_ids, counts, badpart, NB, wpart, W, F);
  } else {
    hipLaunchKernelGGL(k_index_write<false>, dim3(NBH), dim3(HV_CHUNK), 0, st, occ_ptr, T, L, THR, blk, NBH, heavy_ids, items,
                       ch, ci, touched_ids, counts, badpart, NB, wpart, W, F);
  }
  return 0;
}

}  // namespace
}  // namespace vfm

using namespace vfm;

extern "C" {

int32_t vfm_heavy_list_for(int64_t n_occ, int64_t T) {
  const int forced = env_int("VFM_HEAVY_LIST", 0);
  if (forced > 0) return forced < VFM_HEAVY_MIN ? VFM_HEAVY_MIN : forced;
  if (T >= VFM_HEAVY_UNITS) return VFM_HEAVY_LIST;
  const int64_t l = n_occ / VFM_HEAVY_UNITS;
  return (int32_t)(l < VFM_HEAVY_MIN ? VFM_HEAVY_MIN : (l > VFM_HEAVY_LIST ? VFM_HEAVY_LIST : l));
}

// The lower threshold vfm_rebuild_heavy is meant to be called with: a quarter of the work-item length (VFM_HEAVY_THR: A/B runs)
int32_t vfm_heavy_threshold(int32_t heavy_list) {
  int thr = env_int("VFM_HEAVY_THR", heavy_list / 4);
  if (thr > heavy_list) thr = heavy_list;
  return thr < VFM_HEAVY_MIN ? VFM_HEAVY_MIN : thr;
}

int64_t vfm_index_workspace_bytes(int64_t B, int32_t F, int64_t T) {
  if (B < 0 || F < 1 || T < 1 || T > 0xFFFFFFFELL || B * (int64_t)F > 0x7FFFFFFFLL) return -1;
  return geometry(B * F, T).total * 4;
}

int64_t vfm_union_workspace_bytes(int64_t T) { return T < 1 ? -1 : 4 * ((T + HV_CHUNK - 1) / HV_CHUNK + 4); }

int vfm_union_rows(int64_t T, const int32_t* occ_ptr_a, const int32_t* occ_ptr_b, void* ws, int32_t* rows, int32_t* count,
                   int32_t* count_host, void* stream) {
  if (T < 1 || T > 0x7FFFFFFELL || !occ_ptr_a || !occ_ptr_b || !ws || !rows || !count)
    return fail(VFM_E_INVALID, "vfm_union_rows: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int NBH = (int)((T + HV_CHUNK - 1) / HV_CHUNK);
  uint32_t* blk = reinterpret_cast<uint32_t*>(ws);
  hipLaunchKernelGGL(k_union_count, dim3(NBH), dim3(HV_CHUNK), 0, st, occ_ptr_a, occ_ptr_b, T, blk);
  if (NBH > HV_SELF) {
    hipLaunchKernelGGL(k_union_scan, dim3(1), dim3(64), 0, st, blk, NBH, count);
    hipLaunchKernelGGL(k_union_write<true>, dim3(NBH), dim3(HV_CHUNK), 0, st, occ_ptr_a, occ_ptr_b, T, blk, NBH, rows, count);
  } else {
    hipLaunchKernelGGL(k_union_write<false>, dim3(NBH), dim3(HV_CHUNK), 0, st, occ_ptr_a, occ_ptr_b, T, blk, NBH, rows, count);
  }
  hipError_t e = hipSuccess;
  if (count_host) e = hipMemcpyAsync(count_host, count, sizeof(int32_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipGetLastError();
  return e == hipSuccess ? 0 : fail_hip(e, "vfm_union_rows");
}

int vfm_build_index(int64_t B, int32_t F, int64_t T, int32_t id_bits, const void* x, void* ws, int32_t* occ_ptr,
                    int32_t* occ_rows, int32_t heavy_list, int32_t* heavy_ids, int64_t cap_heavy,
                    int32_t* heavy_items, int64_t cap_items, int32_t* touched_ids, int32_t* occ_other,
                    const float* inv_occ, double* W, int32_t* counts, int32_t* counts_host, void* stream) {
  if (B < 0 || F < 1 || F > VFM_MAX_FIELDS || T < 1 || T > 0xFFFFFFFELL || B * (int64_t)F > 0x7FFFFFFFLL ||
      (id_bits != 32 && id_bits != 64) || heavy_list < VFM_HEAVY_MIN)
    return fail(VFM_E_INVALID, "vfm_build_index: bad B, F, T, id_bits or heavy_list");
  if (occ_other && F != 2) return fail(VFM_E_INVALID, "vfm_build_index: occ_other is defined for two fields");
  if (!ws || !occ_ptr || !counts || (B > 0 && (!x || !occ_rows)) || cap_heavy < 0 || cap_items < 0 ||
      (cap_heavy > 0 && !heavy_ids) || (cap_items > 0 && !heavy_items))
    return fail(VFM_E_INVALID, "vfm_build_index: NULL pointer");
  if ((inv_occ == nullptr) != (W == nullptr)) return fail(VFM_E_INVALID, "vfm_build_index: give inv_occ and W together, or neither");
  if ((((uintptr_t)ws) & 15) != 0) return fail(VFM_E_INVALID, "vfm_build_index: ws must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const Geo g = geometry(B * F, T);
  const int n = (int)g.n, NB = (int)g.NB;
  uint32_t* w32 = reinterpret_cast<uint32_t*>(ws);
  const bool fused_w = W != nullptr && F <= NW_MAXF;
  if (W != nullptr && !fused_w)                       // many fields: the stand-alone normaliser kernels (vfm_batch_norms)
    if (int rc = launch_norms(B, F, T, id_bits, x, inv_occ, W, st)) return rc;
  hipError_t e = hipSuccess;
  if (B > 0) e = hipMemsetAsync(w32 + g.zero_lo, 0, (size_t)(g.zero_hi - g.zero_lo) * 4, st);
  if (e != hipSuccess) return fail_hip(e, "vfm_build_index: memset");
  double* wpart = reinterpret_cast<double*>(w32 + g.wpart);
  if (n > 0) {
    hipLaunchKernelGGL(k_index_keys, dim3(NB), dim3(RS_THREADS), 0, st, x, (int)(id_bits == 64), n, (uint32_t)T, (int)F,
                       w32 + g.k0, w32 + g.v0, g.rb[0], w32 + g.L1[0], w32 + g.L2[0], (int)g.G,
                       fused_w ? inv_occ : (const float*)nullptr, wpart, w32 + g.bad);
    uint32_t *kin = w32 + g.k0, *vin = w32 + g.v0, *kout = w32 + g.k1, *vout = w32 + g.v1;
    for (int p = 0; p < g.passes; ++p) {
      if (p > 0)        // (pass 0's digit counts came with the keys)
        hipLaunchKernelGGL(k_radix_hist, dim3(NB), dim3(RS_THREADS), 0, st, kin, n, g.shift[p], g.rb[p], w32 + g.L1[p],
                           w32 + g.L2[p], (int)g.G);
      if (p + 1 == g.passes) {
        hipLaunchKernelGGL(k_radix_scatter<true>, dim3(NB), dim3(RS_THREADS), 0, st, kin, vin, n, g.shift[p], g.rb[p],
                           w32 + g.L1[p], w32 + g.L2[p], (int)g.G, (int)g.NS, kout, vout, w32 + g.excl, occ_rows, occ_other, x,
                           (int)(id_bits == 64), (int)F, T);
      } else {
        hipLaunchKernelGGL(k_radix_scatter<false>, dim3(NB), dim3(RS_THREADS), 0, st, kin, vin, n, g.shift[p], g.rb[p],
                           w32 + g.L1[p], w32 + g.L2[p], (int)g.G, (int)g.NS, kout, vout, (uint32_t*)nullptr, occ_rows, occ_other,
                           x, (int)(id_bits == 64), (int)F, T);
      }
      uint32_t* t = kin; kin = kout; kout = t;
      t = vin; vin = vout; vout = t;
    }
    // (kin now names the sorted keys)
    hipLaunchKernelGGL(k_index_count<true>, dim3((unsigned)g.NBH), dim3(HV_CHUNK), 0, st, kin, w32 + g.excl, n,
                       g.shift[g.passes - 1], 1 << g.rb[g.passes - 1], occ_ptr, T, (int)heavy_list, (int)heavy_list, w32 + g.blk,
                       (int)g.NBH);
  } else {                                            // empty shard: every list is empty
    e = hipMemsetAsync(occ_ptr, 0, sizeof(int32_t) * (size_t)(T + 1), st);
    if (e != hipSuccess) return fail_hip(e, "vfm_build_index: memset of occ_ptr");
    hipLaunchKernelGGL(k_index_count<false>, dim3((unsigned)g.NBH), dim3(HV_CHUNK), 0, st, (const uint32_t*)nullptr,
                       (const uint32_t*)nullptr, 0, 0, 1, occ_ptr, T, (int)heavy_list, (int)heavy_list, w32 + g.blk, (int)g.NBH);
  }
  compaction_write(g.NBH, occ_ptr, T, (int)heavy_list, (int)heavy_list, w32 + g.blk, heavy_ids, heavy_items, cap_heavy, cap_items,
                   touched_ids, counts, n > 0 ? w32 + g.bad : (const uint32_t*)nullptr, NB, fused_w && n > 0 ? wpart : nullptr,
                   fused_w ? W : nullptr, (int)F, st);
  if (counts_host) {                                  // the build's one readback, enqueued here (pinned host memory)
    e = hipMemcpyAsync(counts_host, counts, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return fail_hip(e, "vfm_build_index: readback");
  }
  e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "vfm_build_index");
  return 0;
}

int vfm_rebuild_heavy(int64_t T, const int32_t* occ_ptr, void* ws, int32_t heavy_list, int32_t threshold, int32_t* heavy_ids,
                      int64_t cap_heavy, int32_t* heavy_items, int64_t cap_items, int32_t* counts, void* stream) {
  if (T < 1 || T > 0xFFFFFFFELL || !occ_ptr || !ws || !counts || heavy_list < VFM_HEAVY_MIN || threshold < VFM_HEAVY_MIN ||
      threshold > heavy_list || cap_heavy < 0 || cap_items < 0 || (cap_heavy > 0 && !heavy_ids) || (cap_items > 0 && !heavy_items))
    return fail(VFM_E_INVALID, "vfm_rebuild_heavy: bad argument (VFM_HEAVY_MIN <= threshold <= heavy_list)");
  hipStream_t st = (hipStream_t)stream;
  const int64_t NBH = (T + HV_CHUNK - 1) / HV_CHUNK;
  uint32_t* blk = reinterpret_cast<uint32_t*>(ws);       // (5 * NBH + 8 words: inside any workspace of vfm_index_workspace_bytes)
  hipLaunchKernelGGL(k_index_count<false>, dim3((unsigned)NBH), dim3(HV_CHUNK), 0, st, (const uint32_t*)nullptr,
                     (const uint32_t*)nullptr, 0, 0, 1, const_cast<int32_t*>(occ_ptr), T, (int)heavy_list, (int)threshold, blk,
                     (int)NBH);
  compaction_write(NBH, occ_ptr, T, (int)heavy_list, (int)threshold, blk, heavy_ids, heavy_items, cap_heavy, cap_items, nullptr,
                   counts, nullptr, 0, nullptr, nullptr, 0, st);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "vfm_rebuild_heavy");
  return 0;
}

}  // extern "C"
