// gather_bench.hip -- what does the memory system of one MI355X give a kernel that gathers random whole rows of a table?
//
// Built for one question: the general-F forward at the Criteo shape (cfg5: 65,536 gathers of 2-KiB rows
// from a 2 GB table per launch) moves 152 MB in 65 us = 2.3 TB/s.  Is that the rate of random row gathers over a
// multi-GB table (address translation), or the kernel's own shape (2,048 waves, three rows in flight each)?
//
//   hipcc -O3 --offload-arch=gfx950 tools/gather_bench.hip -o gpurun_out/gather_bench && gpurun_out/gather_bench > out.jsonl
//
// Every wave owns a contiguous slice of a list of row numbers (a random sample WITHOUT repeats of the table's rows, so
// no row is served by a cache because an earlier gather brought it in), keeps R rows in flight (R x ROWB/1024
// global_load_dwordx4 per lane) and adds what it loaded into registers; one float per lane is stored at the end.
// Sweeps: table size x rows per launch x waves per CU x rows in flight per wave.  One JSON line per point.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

// ROWV = float4 per lane per row (row = ROWV KiB), R = rows in flight per wave
template <int ROWV, int R>
__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ table, const int32_t* __restrict__ rows, int n_rows,
                                                float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  const int n_waves = (int)(gridDim.x * (blockDim.x >> 6));
  // contiguous slice of the list per wave (like a wave that owns a batch row and walks its fields)
  const int q = n_rows / n_waves, rem = n_rows % n_waves;
  const int beg = wave * q + (wave < rem ? wave : rem);
  const int end = beg + q + (wave < rem ? 1 : 0);
  v4f acc[ROWV];
#pragma unroll
  for (int v = 0; v < ROWV; ++v) acc[v] = (v4f){0.f, 0.f, 0.f, 0.f};
  constexpr size_t ROWF = (size_t)ROWV * 256;          // floats per row
  int i = beg;
  for (; i + R <= end; i += R) {
    int32_t e[R];
#pragma unroll
    for (int r = 0; r < R; ++r) e[r] = __builtin_amdgcn_readfirstlane(rows[i + r]);
    v4f t[R][ROWV];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int v = 0; v < ROWV; ++v)
        t[r][v] = *reinterpret_cast<const v4f*>(table + (size_t)e[r] * ROWF + (size_t)v * 256 + (size_t)lane * 4);
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int v = 0; v < ROWV; ++v) acc[v] += t[r][v];
  }
  for (; i < end; ++i) {
    const int32_t e = __builtin_amdgcn_readfirstlane(rows[i]);
#pragma unroll
    for (int v = 0; v < ROWV; ++v)
      acc[v] += *reinterpret_cast<const v4f*>(table + (size_t)e * ROWF + (size_t)v * 256 + (size_t)lane * 4);
  }
  v4f s = acc[0];
#pragma unroll
  for (int v = 1; v < ROWV; ++v) s += acc[v];
  out[(size_t)wave * 64 + lane] = s.x + s.y + s.z + s.w;
}

template <int ROWV, int R>
float run(const float* table, const int32_t* rows, int n_rows, float* out, int blocks, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_gather<ROWV, R>), dim3(blocks), dim3(256), 0, 0, table, rows, n_rows, out);
  CK(hipEventRecord(e0, 0));
  for (int it = 0; it < iters; ++it)
    hipLaunchKernelGGL((k_gather<ROWV, R>), dim3(blocks), dim3(256), 0, 0, table, rows + (size_t)(it % 4) * n_rows, n_rows, out);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms * 1e3f / iters;     // us per launch
}

template <int ROWV>
float run_r(int R, const float* table, const int32_t* rows, int n_rows, float* out, int blocks, int iters) {
  switch (R) {
    case 1: return run<ROWV, 1>(table, rows, n_rows, out, blocks, iters);
    case 2: return run<ROWV, 2>(table, rows, n_rows, out, blocks, iters);
    case 4: return run<ROWV, 4>(table, rows, n_rows, out, blocks, iters);
    default: return run<ROWV, 8>(table, rows, n_rows, out, blocks, iters);
  }
}

int main(int argc, char** argv) {
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  const bool quick = argc > 1 && atoi(argv[1]) == 1;
  const size_t table_bytes_list[] = {150ull << 20, 2048ull << 20, 8192ull << 20};
  const int row_kib_list[] = {2, 1};
  const int n_list[] = {65536, 1 << 20};
  const int wpc_list[] = {2, 4, 8, 16, 32};
  const int r_list[] = {1, 2, 4, 8};
  float* out;
  CK(hipMalloc(&out, (size_t)cus * 32 * 64 * sizeof(float)));
  std::mt19937_64 gen(12345);
  for (size_t tb : table_bytes_list) {
    if (quick && tb > (2048ull << 20)) continue;
    float* table;
    CK(hipMalloc(&table, tb));
    CK(hipMemset(table, 0, tb));
    for (int row_kib : row_kib_list) {
      const size_t n_table_rows = tb / ((size_t)row_kib << 10);
      for (int n : n_list) {
        if ((size_t)n * 4 > n_table_rows && n > 65536) continue;     // four disjoint lists must fit without repeats
        // four lists of n distinct rows each (launches cycle through them: a launch never re-reads what its predecessor did)
        const size_t need = std::min(n_table_rows, (size_t)n * 4);
        std::vector<int32_t> perm(n_table_rows);
        for (size_t i = 0; i < n_table_rows; ++i) perm[i] = (int32_t)i;
        for (size_t i = 0; i < need; ++i) {        // partial Fisher-Yates
          std::uniform_int_distribution<size_t> pick(i, n_table_rows - 1);
          std::swap(perm[i], perm[pick(gen)]);
        }
        std::vector<int32_t> lists((size_t)n * 4);
        for (size_t i = 0; i < (size_t)n * 4; ++i) lists[i] = perm[i % need];
        int32_t* rows;
        CK(hipMalloc(&rows, lists.size() * sizeof(int32_t)));
        CK(hipMemcpy(rows, lists.data(), lists.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        for (int wpc : wpc_list) {
          for (int R : r_list) {
            const int blocks = cus * wpc / 4;
            const int iters = n >= (1 << 20) ? 20 : 100;
            const float us = row_kib == 2 ? run_r<2>(R, table, rows, n, out, blocks, iters)
                                          : run_r<1>(R, table, rows, n, out, blocks, iters);
            const double gbs = (double)n * row_kib * 1024.0 / (us * 1e-6) / 1e9;
            printf("{\"table_MB\": %zu, \"row_bytes\": %d, \"rows_per_launch\": %d, \"waves_per_cu\": %d, \"rows_in_flight_per_wave\": %d, "
                   "\"us\": %.2f, \"GBs\": %.1f, \"cus\": %d}\n", tb >> 20, row_kib * 1024, n, wpc, R, us, gbs, cus);
            fflush(stdout);
          }
        }
        CK(hipFree(rows));
      }
    }
    CK(hipFree(table));
  }
  return 0;
}
