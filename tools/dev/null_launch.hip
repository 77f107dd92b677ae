// What does ONE launch cost on this machine?  An empty kernel (one wavefront), a 1 024-workgroup empty kernel and a kernel whose only
// wavefront spins for ~ 5 us: back-to-back launches in one stream, time per launch (host clock around 2 000 launches) and the
// kernel's own duration between two events.   hipcc --offload-arch=gfx950 -O2 -o build_exp/null_launch tools/dev/null_launch.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty() {}
__global__ void k_spin(long long cycles) {
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
}
template <typename F>
static void run(const char *name, F launch) {
  hipStream_t s;
  (void)hipStreamCreate(&s);
  for (int i = 0; i < 200; i++) launch(s);
  (void)hipStreamSynchronize(s);
  const int n = 2000;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; i++) launch(s);
  (void)hipStreamSynchronize(s);
  const double per = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
  hipEvent_t a, b;
  (void)hipEventCreate(&a), (void)hipEventCreate(&b);
  float ms = 0, acc = 0;
  for (int i = 0; i < 50; i++) {
    (void)hipEventRecord(a, s);
    launch(s);
    (void)hipEventRecord(b, s);
    (void)hipEventSynchronize(b);
    (void)hipEventElapsedTime(&ms, a, b);
    acc += ms;
  }
  printf("%-34s %6.2f us per launch back to back, %6.2f us between two events around one launch\n", name, per, acc / 50 * 1e3);
}
int main() {
  run("empty, 1 workgroup x 64", [](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); });
  run("empty, 1024 workgroups x 64", [](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(64), 0, s); });
  run("empty, 256 workgroups x 512", [](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, s); });
  run("spin 10 000 cycles, 1024 x 64", [](hipStream_t s) { hipLaunchKernelGGL(k_spin, dim3(1024), dim3(64), 0, s, 10000LL); });
  return 0;
}
