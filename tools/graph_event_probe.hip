// Probe: do the start/stop events of hipExtLaunchKernelGGL survive stream capture, i.e. can a kernel's own duration be
// read back after a hipGraph replay?  Also tries plain hipEventRecord nodes around the kernel.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s (%d) at line %d\n", hipGetErrorString(e), (int)e, __LINE__); return 1; } } while (0)
#define TRY(x) do { hipError_t e = (x); printf("  %-60s -> %s\n", #x, hipGetErrorString(e)); } while (0)

__global__ void spin(float* p, int n) {
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += __sinf(s + i);
  if (s == 123.f) p[0] = s;
}

int main() {
  float* buf; CK(hipMalloc(&buf, 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t a, b, c, d; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c)); CK(hipEventCreate(&d));
  // eager reference
  for (int i = 0; i < 3; ++i) hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, a, b, 0, buf, 20000);
  CK(hipStreamSynchronize(st));
  float ms = 0; CK(hipEventElapsedTime(&ms, a, b)); printf("eager hipExtLaunchKernelGGL events: %.2f us\n", ms * 1e3);
  // captured
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, buf, 1000);
  hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, a, b, 0, buf, 20000);
  printf("  launch with events during capture -> %s\n", hipGetErrorString(hipGetLastError()));
  TRY(hipEventRecord(c, st));
  hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, buf, 20000);
  TRY(hipEventRecord(d, st));
  hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, buf, 1000);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn)); printf("graph nodes: %zu\n", nn);
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipError_t e1 = hipEventElapsedTime(&ms, a, b);
    printf("replay %d: ext-launch events: %s %.2f us;", rep, hipGetErrorString(e1), ms * 1e3);
    hipError_t e2 = hipEventElapsedTime(&ms, c, d);
    printf("  record nodes around kernel: %s %.2f us\n", hipGetErrorString(e2), ms * 1e3);
  }
  return 0;
}
