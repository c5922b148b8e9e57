// Empirical semantics of ds_read_b64_tr_b8 / ds_read_b64_tr_b16 on gfx950: lane l supplies the address of an 8-byte
// chunk; the output shows, for every lane and element, which (source lane, element) it received.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v2i __attribute__((ext_vector_type(2)));
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds16[2048];
  __shared__ __attribute__((aligned(16))) uint8_t lds8[1024];
  const int l = threadIdx.x;
  for (int i = l; i < 512; i += 64) lds8[i] = (uint8_t)i;            // low 8 bits of the byte index
  for (int i = l; i < 512; i += 64) lds8[512 + i] = (uint8_t)(i >> 8);
  for (int i = l; i < 256; i += 64) lds16[i] = (uint16_t)i;
  __syncthreads();
  v2i a = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(lds8 + l * 8));
  v2i b = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(lds8 + 512 + l * 8));
  v4s q = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds16 + l * 4));
  out[l * 8 + 0] = a.x; out[l * 8 + 1] = a.y; out[l * 8 + 2] = b.x; out[l * 8 + 3] = b.y;
  out[l * 8 + 4] = (uint16_t)q.x | ((uint32_t)(uint16_t)q.y << 16);
  out[l * 8 + 5] = (uint16_t)q.z | ((uint32_t)(uint16_t)q.w << 16);
}
int main() {
  uint32_t* d; hipMalloc(&d, 64 * 8 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  uint32_t h[64 * 8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("tr_b8: lane -> 8 x (src lane:src byte)\n");
  for (int l = 0; l < 64; ++l) {
    printf("%2d:", l);
    for (int j = 0; j < 8; ++j) {
      const int lo = (h[l * 8 + (j >> 2)] >> (8 * (j & 3))) & 0xff, hi = (h[l * 8 + 2 + (j >> 2)] >> (8 * (j & 3))) & 0xff;
      const int idx = lo | (hi << 8);
      printf(" %2d:%d", idx / 8, idx % 8);
    }
    printf("\n");
  }
  printf("tr_b16: lane -> 4 x (src lane:src elem)\n");
  for (int l = 0; l < 64; ++l) {
    printf("%2d:", l);
    for (int j = 0; j < 4; ++j) {
      const int idx = (h[l * 8 + 4 + (j >> 1)] >> (16 * (j & 1))) & 0xffff;
      printf(" %2d:%d", idx / 4, idx % 4);
    }
    printf("\n");
  }
  return 0;
}
