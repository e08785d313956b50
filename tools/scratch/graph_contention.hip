// development aid: T host threads, a stream each; per round either K direct launches of a small kernel or ONE replay of a
// captured graph with the same K kernel nodes, then a stream wait.  Prints rounds per second for T = 1, 2, 4, 8.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void k_small(float* p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}
static double g_enq_us[64];
static double run(int T, int K, bool graph, int rounds) {
  std::vector<std::thread> th;
  auto t0 = std::chrono::steady_clock::now();
  for (int t = 0; t < T; t++)
    th.emplace_back([=] {
      hipSetDevice(0);
      hipStream_t s;
      hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
      float* d;
      hipMalloc(&d, 1 << 20);
      hipGraphExec_t ge = nullptr;
      if (graph) {
        hipGraph_t g;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int k = 0; k < K; k++) hipLaunchKernelGGL(k_small, dim3(40), dim3(256), 0, s, d, 10000);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      }
      double enq = 0;
      for (int r = 0; r < rounds; r++) {
        auto a = std::chrono::steady_clock::now();
        if (graph) hipGraphLaunch(ge, s);
        else for (int k = 0; k < K; k++) hipLaunchKernelGGL(k_small, dim3(40), dim3(256), 0, s, d, 10000);
        enq += std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
        hipStreamSynchronize(s);
      }
      g_enq_us[t] = 1e6 * enq / rounds;
      hipFree(d);
    });
  for (auto& x : th) x.join();
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return T * rounds / dt;
}
int main() {
  run(1, 20, false, 50);
  for (int K : {20, 40})
    for (int T : {1, 2, 4, 8}) {
      double a = run(T, K, false, 1500), ea = g_enq_us[0], b = run(T, K, true, 1500), eb = g_enq_us[0];
      printf("K=%d T=%d direct %.0f rounds/s (%.1f us/round/thread, enqueue %.1f us)  graph %.0f rounds/s (%.1f us, enqueue %.1f us)\n", K, T, a, 1e6 * T / a, ea, b, 1e6 * T / b, eb);
    }
  return 0;
}
