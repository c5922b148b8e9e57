// Do a bus-bound copy kernel on one stream and a chain of short kernels on another overlap on this stack?  (round 6: the
// weights copy of an IMPALA train would hide under the next train if they did.)  Device-side wall-clock stamps per kernel.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/overlap_probe tools/experiments/overlap_probe.hip && /tmp/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <sys/mman.h>
#include <fcntl.h>
#include <unistd.h>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// THROTTLE > 0: every wave waits for its stores to be acknowledged after THROTTLE of them -- the bus-bound copy then never has
// more than gridDim x 4 KB x THROTTLE in flight, instead of dumping 4 MB into the write path within a few microseconds
template <int THROTTLE>
__global__ void copy_to_host(float4* dst, const float4* src, long n4, unsigned long long* stamp) {
  if (blockIdx.x == 0 && threadIdx.x == 0) stamp[0] = wall_clock64();
  int k = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    dst[i] = src[i];
    if (THROTTLE > 0 && ++k == THROTTLE) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); k = 0; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(stamp + 1, (unsigned long long)wall_clock64());     // the LAST workgroup's end
}
__global__ void busy(float* x, int iters, unsigned long long* stamp) {
  if (blockIdx.x == 0 && threadIdx.x == 0) stamp[0] = wall_clock64();
  float v = x[blockIdx.x * blockDim.x + threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  x[blockIdx.x * blockDim.x + threadIdx.x] = v;
  if (blockIdx.x == 0 && threadIdx.x == 0) stamp[1] = wall_clock64();
}

__global__ void stamp_kernel(unsigned long long* p) { if (threadIdx.x == 0) *p = wall_clock64(); }

int run(const char* name, hipStream_t a, hipStream_t b, bool graph, int copy_blocks = 32, int throttle = 0, int memcpy_mode = 0) {
  const long n4 = (4 << 20) / 16;
  float4 *src, *dsth; float* x; unsigned long long *st_d, *st_h;
  CK(hipMalloc(&src, n4 * 16)); CK(hipHostMalloc(&dsth, n4 * 16, hipHostMallocMapped)); CK(hipMalloc(&x, 256 * 256 * 4));
  CK(hipMalloc(&st_d, 64 * 8)); CK(hipHostMalloc(&st_h, 64 * 8, 0));
  CK(hipMemset(st_d, 0, 64 * 8)); CK(hipMemset(x, 0, 256 * 256 * 4));
  void* dd; CK(hipHostGetDevicePointer(&dd, dsth, 0));
  void* reg = nullptr;
  if (posix_memalign(&reg, 4096, n4 * 16)) return 1;
  CK(hipHostRegister(reg, n4 * 16, hipHostRegisterDefault));
  // a slot of a shared-memory ring: MAP_SHARED pages of a /dev/shm file, page-locked with hipHostRegister, written at an offset
  static char* shm = nullptr;
  if (!shm) {
    int fd = open("/dev/shm/xt_overlap_probe", O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, n4 * 16 + (1 << 20))) return 1;
    shm = (char*)mmap(nullptr, n4 * 16 + (1 << 20), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd); unlink("/dev/shm/xt_overlap_probe");
    if (shm == MAP_FAILED) return 1;
    CK(hipHostRegister(shm, n4 * 16 + (1 << 20), hipHostRegisterDefault));
  }
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipGraphExec_t exec = nullptr;
  if (graph) {
    hipStream_t cs; CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    hipGraph_t g; CK(hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed));
    for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, cs, x, 600, st_d + 4 + 2 * k);
    CK(hipStreamEndCapture(cs, &g)); CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
  }
  for (int rep = 0; rep < 3; ++rep) {
    // "train k" on a, then the copy on b behind an event, then "train k+1" on a
    for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, a, x, 600, st_d + 40 + 2 * k);
    CK(hipEventRecord(ev, a)); CK(hipStreamWaitEvent(b, ev, 0));
    CK(hipMemsetAsync(st_d, 0, 16, b));
    if (memcpy_mode) {          // the runtime's own D2H (SDMA engine or blit kernel) bracketed by two stamp kernels
      hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, b, st_d + 0);
      CK(hipMemcpyAsync(memcpy_mode == 1 ? (void*)dsth : memcpy_mode == 2 ? reg : (void*)(shm + 4096 + 328), src, n4 * 16, hipMemcpyDeviceToHost, b));
      hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, b, st_d + 1);
    } else
    if (throttle == 0) hipLaunchKernelGGL(copy_to_host<0>, dim3(copy_blocks), dim3(256), 0, b, (float4*)dd, src, n4, st_d + 0);
    else if (throttle == 1) hipLaunchKernelGGL(copy_to_host<1>, dim3(copy_blocks), dim3(256), 0, b, (float4*)dd, src, n4, st_d + 0);
    else hipLaunchKernelGGL(copy_to_host<4>, dim3(copy_blocks), dim3(256), 0, b, (float4*)dd, src, n4, st_d + 0);
    { const auto t0 = std::chrono::steady_clock::now();        // the learner's book-keeping between two trains
      while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(40)) {} }
    if (graph) CK(hipGraphLaunch(exec, a));
    else for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, a, x, 600, st_d + 4 + 2 * k);
    CK(hipDeviceSynchronize());
  }
  CK(hipMemcpy(st_h, st_d, 64 * 8, hipMemcpyDeviceToHost));
  const double t0 = (double)st_h[0], us = 0.01;     // wall_clock64: 100 MHz
  printf("%-44s copy %6.1f..%6.1f us | next-train kernels start at:", name, 0.0, (st_h[1] - t0) * us);
  for (int k = 0; k < 8; ++k) printf(" %6.1f", ((double)st_h[4 + 2 * k] - t0) * us);
  printf("\n");
  return 0;
}

int main() {
  hipStream_t s1, s2, lo; int least, greatest;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  CK(hipStreamCreateWithPriority(&lo, hipStreamNonBlocking, least));
  printf("priority range least %d greatest %d\n", least, greatest);
  if (run("eager: a = created stream, b = created", s1, s2, false)) return 1;
  if (run("eager: a = NULL stream,   b = created", nullptr, s2, false)) return 1;
  if (run("eager: a = created,        b = low priority", s1, lo, false)) return 1;
  if (run("graph: a = created stream, b = created", s1, s2, true)) return 1;
  if (run("graph: a = NULL stream,   b = created", nullptr, s2, true)) return 1;
  if (run("graph: a = NULL stream,   b = low priority", nullptr, lo, true)) return 1;
  if (run("graph: a = NULL, b = created, 256 blocks", nullptr, s2, true, 256)) return 1;
  if (run("graph: a = NULL, b = created, 8 blocks", nullptr, s2, true, 8)) return 1;
  if (run("same stream (a = b = NULL), graph", nullptr, nullptr, true)) return 1;
  if (run("graph, NULL/created, hipMemcpyAsync -> hipHostMalloc", nullptr, s2, true, 32, 0, 1)) return 1;
  if (run("graph, NULL/created, hipMemcpyAsync -> hipHostRegister", nullptr, s2, true, 32, 0, 2)) return 1;
  if (run("eager, created/created, hipMemcpyAsync -> hipHostMalloc", s1, s2, false, 32, 0, 1)) return 1;
  if (run("graph, NULL/created, hipMemcpyAsync -> registered /dev/shm + 4424", nullptr, s2, true, 32, 0, 3)) return 1;
  for (int blocks : {32})
    for (int thr : {1, 4}) {
      char nm[96]; snprintf(nm, sizeof(nm), "graph, NULL/created, %d blocks, throttle %d", blocks, thr);
      if (run(nm, nullptr, s2, true, blocks, thr)) return 1;
    }
  return 0;
}
