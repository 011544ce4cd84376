// c_abi_step.cpp -- the hot path driven from native code through the C ABI alone (include/vfm_hip.h +
// the HIP runtime; no Python, no PyTorch): what a maintainer binding libvfm_hip.so from C/C++ writes.
//
//   c_abi_step DIR n_steps lr
// reads DIR/{meta.txt, x.i64, y.f32, nb_occ.i64, entity.f32, bias.f32, scalars.f32} (written by
// tests/test_gpu_c_abi.py), runs n_steps of  forward -> fused backward + dense Adam  on that one batch with the
// in-kernel Philox eps, and writes DIR/losses.f32 (loss, nll, kl per step) and DIR/entity_out.f32.
//
// build: hipcc -O2 --offload-arch=gfx950 -Iinclude examples/c_abi_step.cpp -Lvae_amd -lvfm_hip -Wl,-rpath,$PWD/vae_amd
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "vfm_hip.h"

#define HIP_OK(e)                                                                         \
  do {                                                                                    \
    hipError_t err_ = (e);                                                                \
    if (err_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(err_)); exit(2); } \
  } while (0)
#define VFM_OK(e)                                                                    \
  do {                                                                               \
    int rc_ = (e);                                                                   \
    if (rc_ != 0) { fprintf(stderr, "%s: %d %s\n", #e, rc_, vfm_last_error()); exit(3); } \
  } while (0)

template <class T>
static std::vector<T> read_file(const std::string& path, size_t n) {
  std::vector<T> v(n);
  FILE* f = fopen(path.c_str(), "rb");
  if (!f || fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "cannot read %s\n", path.c_str()); exit(1); }
  fclose(f);
  return v;
}
template <class T>
static void write_file(const std::string& path, const std::vector<T>& v) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f || fwrite(v.data(), sizeof(T), v.size(), f) != v.size()) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(1); }
  fclose(f);
}
template <class T>
static T* to_device(const std::vector<T>& v) {
  T* d = nullptr;
  HIP_OK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
  HIP_OK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}
template <class T>
static T* device_zeros(size_t n) {
  T* d = nullptr;
  HIP_OK(hipMalloc(&d, std::max<size_t>(n, 1) * sizeof(T)));
  HIP_OK(hipMemset(d, 0, std::max<size_t>(n, 1) * sizeof(T)));
  return d;
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s DIR n_steps lr\n", argv[0]); return 1; }
  const std::string dir = argv[1];
  const int n_steps = atoi(argv[2]);
  const float lr = (float)atof(argv[3]);
  long long B, N, M, d, nb_train, likelihood;
  unsigned long long seed;
  {
    FILE* f = fopen((dir + "/meta.txt").c_str(), "r");
    if (!f || fscanf(f, "%lld %lld %lld %lld %lld %lld %llu", &B, &N, &M, &d, &nb_train, &likelihood, &seed) != 7) {
      fprintf(stderr, "bad meta.txt\n");
      return 1;
    }
    fclose(f);
  }
  const long long T = N + M, F = 2;
  if (vfm_abi_version() != VFM_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }

  auto x = read_file<int64_t>(dir + "/x.i64", (size_t)(B * F));
  auto y = read_file<float>(dir + "/y.f32", (size_t)B);
  auto occ = read_file<int64_t>(dir + "/nb_occ.i64", (size_t)T);
  auto ent = read_file<float>(dir + "/entity.f32", (size_t)(T * 2 * d));
  auto bias = read_file<float>(dir + "/bias.f32", (size_t)(T * 2));
  auto scal = read_file<float>(dir + "/scalars.f32", 3);

  vfm_problem_t p;
  VFM_STRUCT_INIT(p);                                 // zeroes it, sets (struct_size, abi_version): checked by every entry point
  p.B = B; p.B_global = B; p.T = T; p.nb_train = nb_train; p.F = (int32_t)F; p.d = (int32_t)d;
  p.likelihood = (int32_t)likelihood; p.id_bits = 64; p.n_samples = 1; p.flags = VFM_FLAG_SCALED_MOMENTS;
  p.group_hi[0] = N + 1; p.group_hi[1] = T;          // the `<= N` test of vfm-torch.py:316
  p.group_n[0] = (double)N; p.group_n[1] = (double)M;
  p.seed = seed;

  int64_t* d_x = to_device(x);
  float* d_y = to_device(y);
  int64_t* d_occ = to_device(occ);
  float *d_ent = to_device(ent), *d_bias = to_device(bias), *d_scal = to_device(scal);
  // inverted index of the batch (entity -> rows), built on the GPU by the library: workspace + outputs are
  // caller-owned; one small readback tells how many heavy lists / work items it made
  const int32_t L = vfm_heavy_list_for(B * F, T);
  const int64_t cap_h = B * F / L + 1, cap_i = 2 * B * F / L + 2;
  int32_t* d_ptr = device_zeros<int32_t>((size_t)T + 1);
  int32_t* d_rows = device_zeros<int32_t>((size_t)(B * F));
  int32_t* d_hid = device_zeros<int32_t>((size_t)cap_h);
  int32_t* d_items = device_zeros<int32_t>((size_t)(4 * cap_i));
  int32_t* d_counts = device_zeros<int32_t>(8);
  char* d_ws = device_zeros<char>((size_t)vfm_index_workspace_bytes(B, (int32_t)F, T));
  float* d_inv = device_zeros<float>((size_t)T);
  double* d_W = device_zeros<double>((size_t)F);
  float* d_pred = device_zeros<float>((size_t)B);
  float* d_grow = device_zeros<float>((size_t)B);
  float* d_sumz = device_zeros<float>((size_t)(B * d));
  double* d_part = device_zeros<double>(VFM_PARTIALS_LEN);
  float *d_me = device_zeros<float>(ent.size()), *d_ve = device_zeros<float>(ent.size());
  float *d_mb = device_zeros<float>(bias.size()), *d_vb = device_zeros<float>(bias.size());
  float *d_ms = device_zeros<float>(4), *d_vs = device_zeros<float>(4);
  float* d_loss = device_zeros<float>((size_t)(3 * n_steps));
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));

  VFM_OK(vfm_inv_occ_f32(d_occ, d_inv, T, st));                       // once per training set
  // once per batch: the index AND the batch normalisers W (vfm-torch.py:305-306), in the same launches
  VFM_OK(vfm_build_index(B, (int32_t)F, T, 64, d_x, d_ws, d_ptr, d_rows, L, d_hid, cap_h, d_items, cap_i, nullptr, nullptr,
                         d_inv, d_W, d_counts, nullptr, st));
  int32_t counts[8];
  HIP_OK(hipMemcpyAsync(counts, d_counts, sizeof(counts), hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  if (counts[0] != 0) { fprintf(stderr, "%d ids out of range\n", counts[0]); return 1; }
  vfm_index_t idx;
  VFM_STRUCT_INIT(idx);
  idx.occ_ptr = d_ptr; idx.occ_rows = d_rows;
  idx.status = d_counts + 5;                                          // the kernels count clamped index entries here (zeroed by the build)
  idx.heavy_list = L; idx.heavy_threshold = L;                       // what the lists were cut with (small tables: the one-launch backward re-derives the items)
  if (counts[1] > 0) {                                                // lists longer than L: work items + scratch records
    idx.heavy_ids = d_hid; idx.heavy_items = d_items; idx.n_heavy = counts[1]; idx.n_items = counts[2];
    idx.max_items = counts[4];
    idx.heavy_acc = device_zeros<float>((size_t)(counts[1] + counts[2]) * (size_t)(4 + (d + 3) / 4 * 4));
  }
  for (int s = 0; s < n_steps; ++s) {
    p.step = (uint64_t)s;
    VFM_OK(vfm_elbo_fwd_f32(&p, d_x, d_y, d_ent, d_bias, d_inv, d_scal, d_W, nullptr, nullptr, nullptr, d_pred,
                            d_part, d_sumz, d_grow, st));
    VFM_OK(vfm_elbo_bwd_adam_f32(&p, &idx, d_ent, d_bias, d_scal, d_inv, d_W, nullptr, nullptr, nullptr, d_sumz,
                                 d_grow, d_part, d_me, d_ve, d_mb, d_vb, d_ms, d_vs, lr, 0.9f, 0.999f, 1e-8f,
                                 (int64_t)s + 1, d_loss + 3 * s, st));
  }
  HIP_OK(hipStreamSynchronize(st));
  HIP_OK(hipMemcpy(counts, d_counts, sizeof(counts), hipMemcpyDeviceToHost));
  if (counts[5] != 0) { fprintf(stderr, "corrupted inverted index: %d entries clamped (vfm_index_t.status)\n", counts[5]); return 1; }
  std::vector<float> losses((size_t)(3 * n_steps));
  HIP_OK(hipMemcpy(losses.data(), d_loss, losses.size() * sizeof(float), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(ent.data(), d_ent, ent.size() * sizeof(float), hipMemcpyDeviceToHost));
  write_file(dir + "/losses.f32", losses);
  write_file(dir + "/entity_out.f32", ent);
  for (int s = 0; s < n_steps; ++s) printf("step %d loss %.6g nll %.6g kl %.6g\n", s, losses[3 * s], losses[3 * s + 1], losses[3 * s + 2]);
  return 0;
}
