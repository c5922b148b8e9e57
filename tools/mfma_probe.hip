// Micro-probe (diagnostic, not part of the library): cycles per v_mfma_f32_32x32x2_f32 from a LONE wave per SIMD under
// different instruction mixes.  hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, long long* cyc, int iters) {
  __shared__ float lds[8192];
  const int t = threadIdx.x, lane = t & 63;
  for (int i = t; i < 8192; i += 256) lds[i] = 0.001f * (i & 63);
  __syncthreads();
  f32x16 acc[4];
  for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
  float a[4] = {1.f, 2.f, 3.f, 4.f}, b[4] = {0.5f, 0.25f, 0.125f, 1.f};
  const float* p = lds + lane;
  int vx[10];
  for (int q = 0; q < 10; ++q) vx[q] = lane + q;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {          // one chain, bare
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[0], 0, 0, 0);
    } else if (MODE == 1) {   // four chains, bare
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[u], 0, 0, 0);
    } else if (MODE == 2) {   // four chains, 2 ds_read_b32 between MFMAs (operands of the NEXT iteration)
      float an[4], bn[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { an[u] = p[((i * 4 + u) & 63) * 64]; bn[u] = p[((i * 4 + u) & 63) * 64 + 4096]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[u], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) { a[u] = an[u]; b[u] = bn[u]; }
    } else if (MODE == 3) {   // one chain, 2 ds_read_b32 between MFMAs
      float an[4], bn[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { an[u] = p[((i * 4 + u) & 63) * 64]; bn[u] = p[((i * 4 + u) & 63) * 64 + 4096]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[0], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) { a[u] = an[u]; b[u] = bn[u]; }
    } else if (MODE == 5 || MODE == 7) {   // one chain, operands by ds_read_b128: 2 reads feed 4 MFMAs (k-contiguous LDS layout)
      typedef float f4 __attribute__((ext_vector_type(4)));
      const f4 an = *reinterpret_cast<const f4*>(lds + ((i & 15) * 256 + lane * 4));
      const f4 bn = *reinterpret_cast<const f4*>(lds + 4096 + ((i & 15) * 256 + lane * 4));
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[0], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) { a[u] = an[u]; b[u] = bn[u]; }
      if (MODE == 7 && (i & 3) == 3) __syncthreads();      // a block barrier every 16 MFMAs
    } else if (MODE == 6) {   // mode 3 + a block barrier every 16 MFMAs
      float an[4], bn[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { an[u] = p[((i * 4 + u) & 63) * 64]; bn[u] = p[((i * 4 + u) & 63) * 64 + 4096]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[0], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) { a[u] = an[u]; b[u] = bn[u]; }
      if ((i & 3) == 3) __syncthreads();
    } else if (MODE == 8 || MODE == 9) {   // one chain + 10 independent VALU ops per MFMA (+ 2 ds_read_b32 in mode 9)
      float an[4], bn[4];
      if (MODE == 9) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { an[u] = p[((i * 4 + u) & 63) * 64]; bn[u] = p[((i * 4 + u) & 63) * 64 + 4096]; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[0], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 10; ++q) asm volatile("v_add_u32 %0, %0, %1" : "+v"(vx[q]) : "v"(lane));
      }
      if (MODE == 9) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = an[u]; b[u] = bn[u]; }
      }
    } else if (MODE == 4) {   // 16x16x4 f32, four chains, bare (32-cycle issue per the guide)
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      f32x4* c4 = reinterpret_cast<f32x4*>(&acc[0]);
#pragma unroll
      for (int u = 0; u < 8; ++u) c4[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u & 3], b[u & 3], c4[u & 3], 0, 0, 0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
  for (int q = 0; q < 10; ++q) s += (float)vx[q];
  out[blockIdx.x * 256 + t] = s + a[0] + b[0];
  if (t == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
  const int iters = 2000;
  long long h[1024];
  const char* names[10] = {"32x32x2 one chain bare", "32x32x2 four chains bare", "32x32x2 four chains + 2 ds_read/MFMA",
                          "32x32x2 one chain + 2 ds_read/MFMA", "16x16x4 four chains bare (8 per iter)", "32x32x2 one chain + ds_read_b128 (2 per 4 MFMA)",
                          "32x32x2 one chain + 2 ds_read/MFMA + barrier/16", "32x32x2 one chain + ds_read_b128 + barrier/16",
                          "32x32x2 one chain + 10 VALU/MFMA", "32x32x2 one chain + 10 VALU + 2 ds_read/MFMA"};
  for (int blocks : {256, 512, 768}) {
    for (int mode = 0; mode < 10; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        if (mode == 4) hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        if (mode == 5) hipLaunchKernelGGL(probe<5>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        if (mode == 6) hipLaunchKernelGGL(probe<6>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        if (mode == 7) hipLaunchKernelGGL(probe<7>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        if (mode == 8) hipLaunchKernelGGL(probe<8>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        if (mode == 9) hipLaunchKernelGGL(probe<9>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
      }
      hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
      double sum = 0; for (int i = 0; i < blocks; ++i) sum += h[i];
      const int per_iter = mode == 4 ? 8 : 4;
      printf("%4d blocks (%d waves/SIMD)  %-40s %7.1f cycles per MFMA per wave\n", blocks, blocks / 256, names[mode],
             sum / blocks / iters / per_iter);
    }
  }
  return 0;
}
