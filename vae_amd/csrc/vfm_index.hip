// vfm_index.hip -- vfm_build_index: the inverted index of one batch (entity -> batch rows), built on the GPU.
//
// Replaces what the reference does with torch.unique(x, return_inverse, return_counts) three times per
// step (vfm-torch.py:190-192): the kernels never need `unique`, only, per entity, the list of the batch
// rows that contain it (the backward walks it).  The loader does not shuffle (vfm-torch.py:121-122), so
// an index is built once per batch and reused every epoch.
//
// Method: a STABLE least-significant-digit radix sort of the B*F (entity id, position) pairs by entity id,
// 8 bits per pass, ceil(log2(T) / 8) passes; stable means an entity's rows come out in row order, so the
// backward's sums have a fixed order and the whole step stays bitwise reproducible.  No atomics between
// workgroups:
//   k_index_keys     ids -> uint32 keys (range-checked; out-of-range ids are counted and clamped to 0
//                    like the forward does), values = positions r*F + f
//   per pass:  k_radix_hist     per-tile digit histograms                  hist[digit][tile]
//              k_radix_scan     per digit: exclusive scan of its row over the tiles + the digit's total (one
//                               workgroup per digit); only with more than RS_FUSE_NB tiles -- below, every
//                               scatter workgroup adds up the counts it needs itself
//              k_radix_scatter  stable rank inside the tile (per-wave digit matching with ballots + a
//                               scan over the tile's 32 sub-tiles in LDS) + base -> scatter
//   k_index_finish   occ_rows[i] = position / F;  occ_ptr[e] = lower bound of e in the sorted keys
//   k_heavy_count / k_heavy_scan / k_heavy_write   entities with more than `heavy_list` occurrences and
//                    their work items (vfm_index_t), compacted in id order
// gfx950 only, wave = 64.
#include <string.h>

#include "vfm_args.hpp"

namespace vfm {
namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;       // keys per workgroup per pass
constexpr int RS_SUB = RS_ITEMS * (RS_THREADS / 64); // 64-key sub-tiles of a tile, in key order
constexpr int HV_CHUNK = 1024;                       // entities per workgroup of the heavy-list compaction

__global__ __launch_bounds__(RS_THREADS) void k_index_keys(const void* __restrict__ x, int id64, int n, uint32_t T32,
                                                           uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                           unsigned int* __restrict__ counters) {
  unsigned int bad = 0;
  for (int i = blockIdx.x * RS_THREADS + threadIdx.x; i < n; i += gridDim.x * RS_THREADS) {
    uint32_t lo, hi;
    if (id64) {
      const uint2 t = reinterpret_cast<const uint2*>(x)[i];
      lo = t.x; hi = t.y;
    } else {
      lo = reinterpret_cast<const uint32_t*>(x)[i];
      hi = (lo >> 31) ? 0xFFFFFFFFu : 0u;
    }
    const bool ok = hi == 0u && lo < T32;
    bad += ok ? 0u : 1u;
    keys[i] = ok ? lo : 0u;
    vals[i] = (uint32_t)i;
  }
  if (bad) atomicAdd(&counters[0], bad);      // integer: the total does not depend on the order
}

__global__ __launch_bounds__(RS_THREADS) void k_radix_hist(const uint32_t* __restrict__ keys, int n, int shift,
                                                           uint32_t* __restrict__ hist, int NB) {
  __shared__ unsigned int sh[256];
  sh[threadIdx.x] = 0;
  __syncthreads();
  const int base = blockIdx.x * RS_TILE;
#pragma unroll
  for (int k = 0; k < RS_ITEMS; ++k) {
    const int i = base + k * RS_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&sh[(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * NB + blockIdx.x] = sh[threadIdx.x];
}

// Many tiles (NB > RS_FUSE_NB): one workgroup PER DIGIT scans its row hist[digit][0..NB) in place (exclusive) and
// leaves the digit's total in tot[digit]; the scatter workgroups add the totals of the smaller digits themselves.
// (A single workgroup scanning all 256 * NB counts took 383 us per pass at B = 1,048,576: 1.15 of the 1.3 ms build.)
__global__ __launch_bounds__(RS_THREADS) void k_radix_scan(uint32_t* __restrict__ hist, int NB, uint32_t* __restrict__ tot) {
  __shared__ uint32_t sh_w[RS_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t* row = hist + (size_t)blockIdx.x * NB;
  uint32_t carry = 0;
  for (int c0 = 0; c0 < NB; c0 += RS_THREADS * 4) {          // 4 consecutive tiles per thread and round
    const int i0 = c0 + tid * 4;
    uint32_t v[4], s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = (i0 + i < NB) ? row[i0 + i] : 0u; s += v[i]; }
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) sh_w[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, all = 0;
#pragma unroll
    for (int w = 0; w < RS_THREADS / 64; ++w) { const uint32_t t = sh_w[w]; if (w < wave) wbase += t; all += t; }
    uint32_t run = carry + wbase + inc - s;
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (i0 + i < NB) row[i0 + i] = run; run += v[i]; }
    carry += all;
    __syncthreads();
  }
  if (tid == 0) tot[blockIdx.x] = carry;
}

// FUSED (few tiles: NB <= RS_FUSE_NB): `base` holds the RAW per-tile digit counts of k_radix_hist and every
// workgroup forms its own offsets -- digit `tid`: the counts of the tiles before this one plus the totals of the
// smaller digits -- instead of waiting for a one-workgroup scan kernel (26 us per pass at cfg3, most of it latency).
constexpr int RS_FUSE_NB = 256;
template <bool FUSED>
__global__ __launch_bounds__(RS_THREADS) void k_radix_scatter(const uint32_t* __restrict__ kin,
                                                              const uint32_t* __restrict__ vin, int n, int shift,
                                                              const uint32_t* __restrict__ base, int NB,
                                                              uint32_t* __restrict__ kout, uint32_t* __restrict__ vout) {
  __shared__ uint32_t sub[RS_SUB][256];          // [sub-tile][digit]: count, then start position
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < RS_SUB * 256; i += RS_THREADS) (&sub[0][0])[i] = 0;
  __syncthreads();
  const int tbase = blockIdx.x * RS_TILE;
  uint32_t key[RS_ITEMS], val[RS_ITEMS];
  int lower[RS_ITEMS];
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int k = 0; k < RS_ITEMS; ++k) {
    const int i = tbase + k * RS_THREADS + tid;     // = tbase + (k * 4 + wave) * 64 + lane: sub-tiles are in key order
    const bool valid = i < n;
    key[k] = valid ? kin[i] : 0u;
    val[k] = valid ? vin[i] : 0u;
    const uint32_t dg = (key[k] >> shift) & 255u;
    // lanes of this wave holding the same digit
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (dg >> b) & 1u;
      const unsigned long long bb = __ballot(bit);
      m &= bit ? bb : ~bb;
    }
    m = valid ? m : 0ull;
    lower[k] = __popcll(m & lt);
    if (valid && lower[k] == 0) sub[k * (RS_THREADS / 64) + wave][dg] = (uint32_t)__popcll(m);
  }
  __syncthreads();
  uint32_t tile_base;
  if constexpr (FUSED) {
    __shared__ uint32_t sh_tot[256];
    const uint32_t* row = base + (size_t)tid * NB;
    uint32_t before = 0, total = 0;
    const int me = blockIdx.x;
    int t0 = 0;
    for (; t0 + 8 <= NB; t0 += 8) {
      uint32_t c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = row[t0 + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) { total += c[u]; before += (t0 + u < me) ? c[u] : 0u; }
    }
    for (; t0 < NB; ++t0) { const uint32_t c = row[t0]; total += c; before += (t0 < me) ? c : 0u; }
    sh_tot[tid] = total;
    __syncthreads();
    if (tid < 64) {                              // exclusive scan of the 256 digit totals: 4 per lane + wave scan
      uint32_t v[4], tot = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = sh_tot[4 * tid + i]; tot += v[i]; }
      uint32_t inc = tot;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (tid >= o) inc += t;
      }
      uint32_t run = inc - tot;
#pragma unroll
      for (int i = 0; i < 4; ++i) { sh_tot[4 * tid + i] = run; run += v[i]; }
    }
    __syncthreads();
    tile_base = sh_tot[tid] + before;
  } else {       // rows scanned by k_radix_scan, digit totals behind the histogram
    __shared__ uint32_t sh_tot[256];
    sh_tot[tid] = base[(size_t)256 * NB + tid];
    __syncthreads();
    if (tid < 64) {
      uint32_t v[4], tot = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = sh_tot[4 * tid + i]; tot += v[i]; }
      uint32_t inc = tot;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (tid >= o) inc += t;
      }
      uint32_t run = inc - tot;
#pragma unroll
      for (int i = 0; i < 4; ++i) { sh_tot[4 * tid + i] = run; run += v[i]; }
    }
    __syncthreads();
    tile_base = sh_tot[tid] + base[(size_t)tid * NB + blockIdx.x];
  }
  {   // digit `tid`: global base of this tile, then the sub-tiles in order
    uint32_t run = tile_base;
#pragma unroll 4
    for (int s = 0; s < RS_SUB; ++s) { const uint32_t c = sub[s][tid]; sub[s][tid] = run; run += c; }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < RS_ITEMS; ++k) {
    const int i = tbase + k * RS_THREADS + tid;
    if (i < n) {
      const uint32_t dg = (key[k] >> shift) & 255u;
      const uint32_t pos = sub[k * (RS_THREADS / 64) + wave][dg] + (uint32_t)lower[k];
      kout[pos] = key[k];
      vout[pos] = val[k];
    }
  }
}

__global__ __launch_bounds__(RS_THREADS) void k_index_finish(const uint32_t* __restrict__ keys,
                                                             const uint32_t* __restrict__ vals, int n, int F, int64_t T,
                                                             int32_t* __restrict__ occ_ptr, int32_t* __restrict__ occ_rows,
                                                             const void* __restrict__ x, int id64,
                                                             int32_t* __restrict__ occ_other) {
  const int64_t total = (int64_t)n > T + 1 ? (int64_t)n : T + 1;
  for (int64_t i = blockIdx.x * (int64_t)RS_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * RS_THREADS) {
    if (i < n) occ_rows[i] = (int32_t)(vals[i] / (uint32_t)F);
    if (occ_other && i < n) {      // two fields: the entity in the OTHER column of the occurrence's row (clamped like the keys)
      const uint32_t op = vals[i] ^ 1u;
      uint32_t lo, hi;
      if (id64) { const uint2 t = reinterpret_cast<const uint2*>(x)[op]; lo = t.x; hi = t.y; }
      else { lo = reinterpret_cast<const uint32_t*>(x)[op]; hi = (lo >> 31) ? 0xFFFFFFFFu : 0u; }
      occ_other[i] = (hi == 0u && (int64_t)lo < T) ? (int32_t)lo : 0;
    }
    if (i <= T) {            // first sorted position whose key is >= i
      int lo = 0, hi = n;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)keys[mid] < i) lo = mid + 1; else hi = mid;
      }
      occ_ptr[i] = lo;
    }
  }
}

// ---- compaction passes over the entities, in id order: (a) entities with more than L occurrences and their work
// items, (b) the entities the batch contains at all (`touched`: the row list of the lazy Adam step) ----
__device__ __forceinline__ void heavy_of(const int32_t* occ_ptr, int64_t e, int64_t T, int THR, int& cnt, int& beg, bool& any) {
  cnt = 0; beg = 0;
  if (e < T) { beg = occ_ptr[e]; cnt = occ_ptr[e + 1] - beg; }
  any = cnt > 0;
  if (cnt <= THR) cnt = 0;
}

__global__ __launch_bounds__(HV_CHUNK) void k_heavy_count(const int32_t* __restrict__ occ_ptr, int64_t T, int L, int THR,
                                                          uint32_t* __restrict__ blk /*[3][NBH]*/, int NBH) {
  __shared__ uint32_t sh[3][HV_CHUNK / 64];
  const int64_t e = blockIdx.x * (int64_t)HV_CHUNK + threadIdx.x;
  int cnt, beg; bool any;
  heavy_of(occ_ptr, e, T, THR, cnt, beg, any);
  uint32_t a = cnt > 0 ? 1u : 0u, b = cnt > 0 ? (uint32_t)((cnt + L - 1) / L) : 0u, c = any ? 1u : 0u;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); c += __shfl_xor(c, m, 64); }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = b; sh[2][threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x < 3) {
    uint32_t t = 0;
    for (int w = 0; w < HV_CHUNK / 64; ++w) t += sh[threadIdx.x][w];
    blk[(size_t)threadIdx.x * NBH + blockIdx.x] = t;
  }
}

// exclusive scans of the three block-count rows (one wave each); totals -> counters[1..3]
__global__ __launch_bounds__(192) void k_heavy_scan(uint32_t* __restrict__ blk, int NBH, unsigned int* __restrict__ counters) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t* row = blk + (size_t)w * NBH;
  uint32_t carry = 0;
  for (int i0 = 0; i0 < NBH; i0 += 64) {
    const int i = i0 + lane;
    const uint32_t v = i < NBH ? row[i] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (i < NBH) row[i] = carry + inc - v;
    carry += __shfl(inc, 63, 64);
  }
  if (lane == 0) counters[1 + w] = carry;
}

__global__ __launch_bounds__(HV_CHUNK) void k_heavy_write(const int32_t* __restrict__ occ_ptr, int64_t T, int L, int THR,
                                                          const uint32_t* __restrict__ blk, int NBH,
                                                          int32_t* __restrict__ heavy_ids, int32_t* __restrict__ items,
                                                          int cap_h, int cap_i, int32_t* __restrict__ touched_ids) {
  __shared__ uint32_t sh[3][HV_CHUNK / 64];
  const int tid = threadIdx.x;
  const int64_t e = blockIdx.x * (int64_t)HV_CHUNK + tid;
  int cnt, beg; bool any;
  heavy_of(occ_ptr, e, T, THR, cnt, beg, any);
  const uint32_t a = cnt > 0 ? 1u : 0u, b = cnt > 0 ? (uint32_t)((cnt + L - 1) / L) : 0u, c = any ? 1u : 0u;
  uint32_t ia = a, ib = b, ic = c;                // inclusive scans inside the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t ta = __shfl_up(ia, o, 64), tb = __shfl_up(ib, o, 64), tc = __shfl_up(ic, o, 64);
    if ((tid & 63) >= o) { ia += ta; ib += tb; ic += tc; }
  }
  if ((tid & 63) == 63) { sh[0][tid >> 6] = ia; sh[1][tid >> 6] = ib; sh[2][tid >> 6] = ic; }
  __syncthreads();
  if (tid < 3) {
    uint32_t r = 0;
    for (int w = 0; w < HV_CHUNK / 64; ++w) { const uint32_t t = sh[tid][w]; sh[tid][w] = r; r += t; }
  }
  __syncthreads();
  if (any && touched_ids) touched_ids[blk[2 * (size_t)NBH + blockIdx.x] + sh[2][tid >> 6] + ic - c] = (int32_t)e;
  if (cnt > 0) {
    const uint32_t slot = blk[blockIdx.x] + sh[0][tid >> 6] + ia - a;
    uint32_t it = blk[NBH + blockIdx.x] + sh[1][tid >> 6] + ib - b;
    if (slot < (uint32_t)cap_h) heavy_ids[slot] = (int32_t)e;
    for (int o = beg; o < beg + cnt; o += L, ++it) {
      if (it < (uint32_t)cap_i) {
        const int oe = (o + L < beg + cnt) ? o + L : beg + cnt;
        *reinterpret_cast<int4*>(items + 4 * (size_t)it) = make_int4((int)slot, o, oe, 0);
      }
    }
  }
}

// ---- rows of the look-ahead step: the entities of batch A or of batch B, in id order (same three-kernel compaction) ----
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
__global__ __launch_bounds__(HV_CHUNK) void k_union_write(const int32_t* __restrict__ occ_a, const int32_t* __restrict__ occ_b,
                                                          int64_t T, const uint32_t* __restrict__ blk, int32_t* __restrict__ rows) {
  __shared__ uint32_t sh[HV_CHUNK / 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t e = blockIdx.x * (int64_t)HV_CHUNK + tid;
  const bool in = in_either(occ_a, occ_b, e, T);
  const unsigned long long m = __ballot(in);
  if (lane == 0) sh[tid >> 6] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t wbase = 0;
  for (int w = 0; w < (tid >> 6); ++w) wbase += sh[w];
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  if (in) rows[blk[blockIdx.x] + wbase + (uint32_t)__popcll(m & lt)] = (int32_t)e;
}

int key_bits(int64_t T) {
  int bits = 1;
  while (bits < 32 && ((int64_t)1 << bits) < T) ++bits;
  return bits;
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
  if (B < 0 || F < 1 || T < 1 || B * (int64_t)F > 0x7FFFFFFFLL) return -1;
  const int64_t n = B * F;
  const int64_t NB = (n + RS_TILE - 1) / RS_TILE;
  const int64_t NBH = (T + HV_CHUNK - 1) / HV_CHUNK;
  // 4 key / value buffers, the radix histogram, the heavy block counts, a few counters
  return 4 * ((n + 3) & ~(int64_t)3) * 4 + (256 * NB + 256) * 4 + (3 * NBH + 4) * 4 + 64;
}

int64_t vfm_union_workspace_bytes(int64_t T) { return T < 1 ? -1 : 4 * ((T + HV_CHUNK - 1) / HV_CHUNK + 4); }

int vfm_union_rows(int64_t T, const int32_t* occ_ptr_a, const int32_t* occ_ptr_b, void* ws, int32_t* rows, int32_t* count,
                   void* stream) {
  if (T < 1 || T > 0x7FFFFFFELL || !occ_ptr_a || !occ_ptr_b || !ws || !rows || !count)
    return fail(VFM_E_INVALID, "vfm_union_rows: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int NBH = (int)((T + HV_CHUNK - 1) / HV_CHUNK);
  uint32_t* blk = reinterpret_cast<uint32_t*>(ws);
  hipLaunchKernelGGL(k_union_count, dim3(NBH), dim3(HV_CHUNK), 0, st, occ_ptr_a, occ_ptr_b, T, blk);
  hipLaunchKernelGGL(k_union_scan, dim3(1), dim3(64), 0, st, blk, NBH, count);
  hipLaunchKernelGGL(k_union_write, dim3(NBH), dim3(HV_CHUNK), 0, st, occ_ptr_a, occ_ptr_b, T, blk, rows);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail_hip(e, "vfm_union_rows");
}

int vfm_build_index(int64_t B, int32_t F, int64_t T, int32_t id_bits, const void* x, void* ws, int32_t* occ_ptr,
                    int32_t* occ_rows, int32_t heavy_list, int32_t* heavy_ids, int64_t cap_heavy,
                    int32_t* heavy_items, int64_t cap_items, int32_t* touched_ids, int32_t* occ_other, int32_t* counts,
                    void* stream) {
  if (B < 0 || F < 1 || F > VFM_MAX_FIELDS || T < 1 || T > 0xFFFFFFFELL || B * (int64_t)F > 0x7FFFFFFFLL ||
      (id_bits != 32 && id_bits != 64) || heavy_list < VFM_HEAVY_MIN)
    return fail(VFM_E_INVALID, "vfm_build_index: bad B, F, T, id_bits or heavy_list");
  if (occ_other && F != 2) return fail(VFM_E_INVALID, "vfm_build_index: occ_other is defined for two fields");
  if (!ws || !occ_ptr || !counts || (B > 0 && (!x || !occ_rows)) || cap_heavy < 0 || cap_items < 0 ||
      (cap_heavy > 0 && !heavy_ids) || (cap_items > 0 && !heavy_items))
    return fail(VFM_E_INVALID, "vfm_build_index: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  const int n = (int)(B * F);
  const int NB = (n + RS_TILE - 1) / RS_TILE;
  const int NBH = (int)((T + HV_CHUNK - 1) / HV_CHUNK);
  const size_t n4 = ((size_t)n + 3) & ~(size_t)3;
  uint32_t* k0 = reinterpret_cast<uint32_t*>(ws);
  uint32_t* v0 = k0 + n4;
  uint32_t* k1 = v0 + n4;
  uint32_t* v1 = k1 + n4;
  uint32_t* hist = v1 + n4;
  uint32_t* blk = hist + (size_t)256 * NB + 256;       // (256 digit totals behind the histogram)
  unsigned int* counters = reinterpret_cast<unsigned int*>(blk + (size_t)3 * NBH + 4);
  hipError_t e = hipMemsetAsync(counters, 0, 16, st);
  if (e != hipSuccess) return fail_hip(e, "vfm_build_index: memset");
  if (n > 0) {
    int g = (n + RS_THREADS - 1) / RS_THREADS;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_index_keys, dim3(g), dim3(RS_THREADS), 0, st, x, (int)(id_bits == 64), n, (uint32_t)T, k0, v0,
                       counters);
    const int passes = (key_bits(T) + 7) / 8;
    for (int p = 0; p < passes; ++p) {
      hipLaunchKernelGGL(k_radix_hist, dim3(NB), dim3(RS_THREADS), 0, st, k0, n, 8 * p, hist, NB);
      if (NB <= RS_FUSE_NB) {
        hipLaunchKernelGGL(k_radix_scatter<true>, dim3(NB), dim3(RS_THREADS), 0, st, k0, v0, n, 8 * p, hist, NB, k1, v1);
      } else {
        hipLaunchKernelGGL(k_radix_scan, dim3(256), dim3(RS_THREADS), 0, st, hist, NB, hist + (size_t)256 * NB);
        hipLaunchKernelGGL(k_radix_scatter<false>, dim3(NB), dim3(RS_THREADS), 0, st, k0, v0, n, 8 * p, hist, NB, k1, v1);
      }
      uint32_t* t = k0; k0 = k1; k1 = t;
      t = v0; v0 = v1; v1 = t;
    }
  }
  {
    const int64_t total = (int64_t)n > T + 1 ? (int64_t)n : T + 1;
    int64_t g = (total + RS_THREADS - 1) / RS_THREADS;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_index_finish, dim3((unsigned)g), dim3(RS_THREADS), 0, st, k0, v0, n, (int)F, T, occ_ptr, occ_rows,
                       x, (int)(id_bits == 64), occ_other);
  }
  const int thr = (int)heavy_list;
  hipLaunchKernelGGL(k_heavy_count, dim3(NBH), dim3(HV_CHUNK), 0, st, occ_ptr, T, (int)heavy_list, thr, blk, NBH);
  hipLaunchKernelGGL(k_heavy_scan, dim3(1), dim3(192), 0, st, blk, NBH, counters);
  hipLaunchKernelGGL(k_heavy_write, dim3(NBH), dim3(HV_CHUNK), 0, st, occ_ptr, T, (int)heavy_list, thr, blk, NBH, heavy_ids,
                     heavy_items, (int)(cap_heavy > 0x7FFFFFFF ? 0x7FFFFFFF : cap_heavy),
                     (int)(cap_items > 0x7FFFFFFF ? 0x7FFFFFFF : cap_items), touched_ids);
  e = hipMemcpyAsync(counts, counters, 16, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return fail_hip(e, "vfm_build_index: copy of the counters");
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
  const int NBH = (int)((T + HV_CHUNK - 1) / HV_CHUNK);
  uint32_t* blk = reinterpret_cast<uint32_t*>(ws);
  unsigned int* counters = reinterpret_cast<unsigned int*>(blk + (size_t)3 * NBH + 4);
  hipError_t e = hipMemsetAsync(counters, 0, 16, st);
  if (e != hipSuccess) return fail_hip(e, "vfm_rebuild_heavy: memset");
  hipLaunchKernelGGL(k_heavy_count, dim3(NBH), dim3(HV_CHUNK), 0, st, occ_ptr, T, (int)heavy_list, (int)threshold, blk, NBH);
  hipLaunchKernelGGL(k_heavy_scan, dim3(1), dim3(192), 0, st, blk, NBH, counters);
  hipLaunchKernelGGL(k_heavy_write, dim3(NBH), dim3(HV_CHUNK), 0, st, occ_ptr, T, (int)heavy_list, (int)threshold, blk, NBH, heavy_ids,
                     heavy_items, (int)(cap_heavy > 0x7FFFFFFF ? 0x7FFFFFFF : cap_heavy),
                     (int)(cap_items > 0x7FFFFFFF ? 0x7FFFFFFF : cap_items), (int32_t*)nullptr);
  e = hipMemcpyAsync(counts, counters, 16, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return fail_hip(e, "vfm_rebuild_heavy: copy of the counters");
  e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "vfm_rebuild_heavy");
  return 0;
}

}  // extern "C"
