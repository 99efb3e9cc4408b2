// Host cost of issuing kernels on this runtime (round 5, review item 5): what one update_kv pays for its four launches, and
// whether a pre-built HIP graph with per-call parameter updates would be cheaper.   hipcc -O2 --offload-arch=gfx950 launch_cost.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

struct Big { const void* a; const void* b; void* c; void* d; long s[12]; int i[16]; float f[4]; };   // ~ LogitsParams
__global__ void k_small(int* p) { if (p && threadIdx.x == 1024) *p = 1; }
__global__ void k_big(Big b) { if (b.c && threadIdx.x == 1024) *(int*)b.c = (int)b.s[3]; }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <typename F> double per_iter_us(F&& f, hipStream_t st, int n = 4000, int sync_every = 200) {
  for (int i = 0; i < 50; ++i) f();
  (void)hipStreamSynchronize(st);
  double t = 0;
  for (int done = 0; done < n; done += sync_every) {
    const double t0 = now();
    for (int i = 0; i < sync_every; ++i) f();
    t += now() - t0;
    (void)hipStreamSynchronize(st);
  }
  return t / n * 1e6;
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  int* buf;
  CK(hipMalloc(&buf, 4096));
  Big b{};
  b.c = nullptr;
  printf("{\n");
  printf(" \"launch_small_arg_us\": %.2f,\n", per_iter_us([&] { hipLaunchKernelGGL(k_small, dim3(32), dim3(256), 0, st, (int*)nullptr); }, st));
  printf(" \"launch_200B_arg_us\": %.2f,\n", per_iter_us([&] { hipLaunchKernelGGL(k_big, dim3(32), dim3(256), 0, st, b); }, st));
  printf(" \"launch_1024_blocks_us\": %.2f,\n", per_iter_us([&] { hipLaunchKernelGGL(k_big, dim3(1024), dim3(256), 0, st, b); }, st));
  printf(" \"ext_launch_no_events_us\": %.2f,\n", per_iter_us([&] { hipExtLaunchKernelGGL(k_big, dim3(32), dim3(256), 0, st, nullptr, nullptr, 0, b); }, st));
  printf(" \"four_launches_us\": %.2f,\n", per_iter_us([&] { for (int j = 0; j < 4; ++j) hipLaunchKernelGGL(k_big, dim3(32), dim3(256), 0, st, b); }, st, 2000, 100));
  {
    void* args[] = {&b};
    printf(" \"hipLaunchKernel_args_array_us\": %.2f,\n", per_iter_us([&] { (void)hipLaunchKernel((const void*)k_big, dim3(32), dim3(256), args, 0, st); }, st));
  }
  {
    hipFunction_t fn;
    hipError_t e = hipGetFuncBySymbol(&fn, (const void*)k_big);
    if (e == hipSuccess) {
      void* args[] = {&b};
      printf(" \"hipModuleLaunchKernel_us\": %.2f,\n", per_iter_us([&] { (void)hipModuleLaunchKernel(fn, 32, 1, 1, 256, 1, 1, 0, st, args, nullptr); }, st));
    } else {
      printf(" \"hipModuleLaunchKernel_us\": null,\n");
    }
  }
  // a graph of four dependent kernel nodes, instantiated once; per call: new parameters for two nodes + one launch
  {
    hipGraph_t g;
    CK(hipGraphCreate(&g, 0));
    std::vector<hipGraphNode_t> nodes(4);
    Big pb[4] = {b, b, b, b};
    void* args[4][1] = {{&pb[0]}, {&pb[1]}, {&pb[2]}, {&pb[3]}};
    hipKernelNodeParams kp[4];
    for (int j = 0; j < 4; ++j) {
      kp[j] = hipKernelNodeParams{};
      kp[j].func = (void*)k_big; kp[j].gridDim = dim3(32); kp[j].blockDim = dim3(256); kp[j].sharedMemBytes = 0;
      kp[j].kernelParams = args[j]; kp[j].extra = nullptr;
      CK(hipGraphAddKernelNode(&nodes[j], g, j ? &nodes[j - 1] : nullptr, j ? 1 : 0, &kp[j]));
    }
    hipGraphExec_t ex;
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    printf(" \"graph4_launch_only_us\": %.2f,\n", per_iter_us([&] { (void)hipGraphLaunch(ex, st); }, st, 2000, 100));
    long tick = 0;
    printf(" \"graph4_two_param_updates_plus_launch_us\": %.2f,\n", per_iter_us([&] {
      pb[0].s[3] = ++tick; pb[3].s[3] = tick;
      (void)hipGraphExecKernelNodeSetParams(ex, nodes[0], &kp[0]);
      (void)hipGraphExecKernelNodeSetParams(ex, nodes[3], &kp[3]);
      (void)hipGraphLaunch(ex, st);
    }, st, 2000, 100));
    // device time of the graph against four plain launches (events around 200 iterations)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 200; ++i) (void)hipGraphLaunch(ex, st);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf(" \"graph4_wall_us_per_iteration\": %.2f,\n", ms * 1e3 / 200);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 200; ++i) for (int j = 0; j < 4; ++j) hipLaunchKernelGGL(k_big, dim3(32), dim3(256), 0, st, b);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf(" \"four_launches_wall_us_per_iteration\": %.2f\n", ms * 1e3 / 200);
  }
  printf("}\n");
  return 0;
}
