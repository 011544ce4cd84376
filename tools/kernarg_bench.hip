#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
template <int N> struct Big { long v[N]; };
template <int N> __global__ void k(Big<N> a, long* out) { if (a.v[0] == 12345 && threadIdx.x == 0) out[0] = a.v[N - 1]; }
template <int N> double run(long* out, int iters) {
  Big<N> a; for (int i = 0; i < N; ++i) a.v[i] = i;
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, a, out);
  hipDeviceSynchronize();
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < iters; ++i) { hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, a, out); if (i % 200 == 199) hipDeviceSynchronize(); }
  auto t1 = std::chrono::high_resolution_clock::now();
  hipDeviceSynchronize();
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / iters;
}
int main() {
  long* out; hipMalloc(&out, 64);
  printf("kernarg    64 B: %.2f us per launch\n", run<8>(out, 4000));
  printf("kernarg   256 B: %.2f us per launch\n", run<32>(out, 4000));
  printf("kernarg   512 B: %.2f us per launch\n", run<64>(out, 4000));
  printf("kernarg  1024 B: %.2f us per launch\n", run<128>(out, 4000));
  printf("kernarg  1600 B: %.2f us per launch\n", run<200>(out, 4000));
  printf("kernarg  2048 B: %.2f us per launch\n", run<256>(out, 4000));
  printf("kernarg  4000 B: %.2f us per launch\n", run<500>(out, 4000));
  return 0;
}
