// one-off: how long do hipMalloc / hipFree / a first touch of several GB take on this box?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now().time_since_epoch()).count(); }
int main()
{
  hipStream_t s; hipStreamCreate(&s);
  for (int rep = 0; rep < 3; ++rep)
    for (size_t gb : { 1, 8, 17 }) {
      void* p = nullptr;
      double t0 = now(); hipError_t e = hipMalloc(&p, gb << 30); double t1 = now();
      hipMemsetAsync(p, 0, gb << 30, s); double t2 = now(); hipStreamSynchronize(s); double t3 = now();
      hipFree(p); double t4 = now();
      printf("rep %d %2zu GiB: hipMalloc %.2f ms (%d), memset enqueue %.2f ms, memset run %.2f ms, hipFree %.2f ms\n", rep, gb, t1 - t0, (int)e, t2 - t1, t3 - t2, t4 - t3);
    }
  // does a big hipMalloc on one thread block a kernel-launch loop on another?
  void* small = nullptr; hipMalloc(&small, 1 << 20);
  std::thread th([] { hipSetDevice(0); void* q = nullptr; double t0 = now(); hipMalloc(&q, (size_t)20 << 30); printf("thread: hipMalloc 20 GiB %.2f ms\n", now() - t0); hipFree(q); });
  double worst = 0; double t_end = now() + 1500;
  while (now() < t_end) { double t0 = now(); hipMemsetAsync(small, 0, 1 << 20, s); hipStreamSynchronize(s); worst = std::max(worst, now() - t0); }
  th.join();
  printf("main: worst 1 MiB memset + sync while the other thread allocated: %.2f ms\n", worst);
  return 0;
}
