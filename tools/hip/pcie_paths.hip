// How a vec step's bytes can cross PCIe, timed at the host batcher's own sizes (N = 4096 H1 environments:
// 1.15 MB of state rows up, 0.33 MB of controls down).  Prints one JSON object.
//   pull    : a kernel reads pinned host memory through its device alias (grid sweep) and stores to HBM
//   copy    : hipMemcpyAsync pinned -> device, whole and in 4 / 8 pieces on one stream
//   signal  : how long the host waits to learn that a kernel has finished: hipStreamSynchronize against a flag
//             the kernel stores into mapped host memory after a system-scope fence (host polls it)
// build: hipcc -O3 --offload-arch=gfx950 tools/hip/pcie_paths.hip -o tools/hip/pcie_paths.bin
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void pull_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}

// four independent 16-byte loads per lane in flight before the first store
__global__ void pull4_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (i + k * stride < n) v[k] = src[i + k * stride];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (i + k * stride < n) dst[i + k * stride] = v[k];
  }
}

__global__ void push_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long n, volatile int* flag, int* counter,
                            int gen) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
  if (flag) {
    __threadfence_system();
    __shared__ int last;
    if (threadIdx.x == 0) last = (atomicAdd(counter, 1) == (int)gridDim.x - 1);   // counter: device memory
    __syncthreads();
    if (last && threadIdx.x == 0) { *counter = 0; flag[0] = gen; }
  }
}

int main() {
  const size_t up = 4096 * (18 + 17) * 8, down = 4096 * 10 * 8;   // H1: nq + nv doubles up, nu doubles down (sizes only)
  const size_t big = 8u << 20;
  float4 *h, *hm, *d;
  CK(hipHostMalloc((void**)&h, big, hipHostMallocDefault));
  CK(hipHostGetDevicePointer((void**)&hm, h, 0));
  CK(hipMalloc((void**)&d, big));
  memset(h, 1, big);
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 200;
  printf("{");
  // ---- pull ----
  printf("\"pull_us\": {");
  bool first = true;
  for (size_t bytes : {up, up / 4, (size_t)(4u << 20)}) {
    for (int grid : {32, 128, 256, 512, 1024, 2048}) {
      for (int four = 0; four < 2; ++four) {
        const long n = (long)(bytes / 16);
        for (int w = 0; w < 20; ++w) {
          if (four) pull4_kernel<<<grid, 256, 0, s>>>(hm, d, n); else pull_kernel<<<grid, 256, 0, s>>>(hm, d, n);
        }
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) {
          if (four) pull4_kernel<<<grid, 256, 0, s>>>(hm, d, n); else pull_kernel<<<grid, 256, 0, s>>>(hm, d, n);
        }
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s\"%zu_B_grid%d%s\": %.2f", first ? "" : ", ", bytes, grid, four ? "_x4" : "", ms * 1000 / reps);
        first = false;
      }
    }
  }
  printf("}, ");
  // ---- copies ----
  printf("\"copy_us\": {");
  first = true;
  for (int pieces : {1, 2, 4, 8}) {
    const size_t piece = (up / pieces) & ~(size_t)15;
    for (int w = 0; w < 20; ++w) CK(hipMemcpyAsync(d, h, up, hipMemcpyHostToDevice, s));
    CK(hipStreamSynchronize(s));
    const double t0 = now();
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r)
      for (int p = 0; p < pieces; ++p)
        CK(hipMemcpyAsync((char*)d + p * piece, (char*)h + p * piece, piece, hipMemcpyHostToDevice, s));
    const double t_sub = now();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s\"h2d_%d_pieces\": {\"device\": %.2f, \"host_submit\": %.2f}", first ? "" : ", ", pieces, ms * 1000 / reps,
           (t_sub - t0) * 1e6 / reps);
    first = false;
  }
  {
    for (int w = 0; w < 20; ++w) CK(hipMemcpyAsync(h, d, down, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) CK(hipMemcpyAsync(h, d, down, hipMemcpyDeviceToHost, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf(", \"d2h_controls\": %.2f", ms * 1000 / reps);
    // one copy, host-timed from the call to the end of the synchronise (what a step pays)
    double best = 1e9;
    for (int r = 0; r < 50; ++r) {
      const double a = now();
      CK(hipMemcpyAsync(h, d, down, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      const double b = now();
      if (b - a < best) best = b - a;
    }
    printf(", \"d2h_controls_call_to_sync_best\": %.2f", best * 1e6);
    best = 1e9;
    for (int r = 0; r < 50; ++r) {
      const double a = now();
      CK(hipMemcpyAsync(d, h, up, hipMemcpyHostToDevice, s));
      CK(hipStreamSynchronize(s));
      const double b = now();
      if (b - a < best) best = b - a;
    }
    printf(", \"h2d_state_call_to_sync_best\": %.2f", best * 1e6);
  }
  printf("}, ");
  // ---- completion signal ----
  {
    int *hf, *hfm;
    CK(hipHostMalloc((void**)&hf, 64, hipHostMallocDefault));
    CK(hipHostGetDevicePointer((void**)&hfm, hf, 0));
    hf[0] = 0;
    int* cnt;
    CK(hipMalloc((void**)&cnt, 4));
    CK(hipMemset(cnt, 0, 4));
    const long n = (long)(down / 16);
    double sync_sum = 0, flag_sum = 0, sync_best = 1e9, flag_best = 1e9;
    int gen = 0;
    for (int r = 0; r < 300; ++r) {
      const double a = now();
      push_kernel<<<40, 256, 0, s>>>(d, hm, n, nullptr, nullptr, 0);
      CK(hipStreamSynchronize(s));
      const double b = now();
      if (r >= 100) { sync_sum += b - a; if (b - a < sync_best) sync_best = b - a; }
    }
    for (int r = 0; r < 300; ++r) {
      ++gen;
      const double a = now();
      push_kernel<<<40, 256, 0, s>>>(d, hm, n, hfm, cnt, gen);
      while (*(volatile int*)hf != gen) {
        __builtin_ia32_pause();
        if (now() - a > 2.0) { fprintf(stderr, "mapped flag never arrived\n"); return 1; }
      }
      const double b = now();
      if (r >= 100) { flag_sum += b - a; if (b - a < flag_best) flag_best = b - a; }
    }
    CK(hipStreamSynchronize(s));
    printf("\"signal_us\": {\"launch_to_stream_synchronize\": {\"mean\": %.2f, \"best\": %.2f}, \"launch_to_mapped_flag\": {\"mean\": %.2f, \"best\": %.2f}}",
           sync_sum * 1e6 / 200, sync_best * 1e6, flag_sum * 1e6 / 200, flag_best * 1e6);
  }
  printf("}\n");
  return 0;
}
