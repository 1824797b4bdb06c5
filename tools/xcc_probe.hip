// Which XCD does workgroup b of a 1-D grid run on?  (HW_REG_XCC_ID, gfx942 / gfx950.)  Prints blockIdx % 8 against the XCC id for two grids.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned *out) {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main() {
  unsigned *d, h[1024];
  hipMalloc(&d, sizeof h);
  for (int grid : {256, 512}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(d, 0xff, sizeof h);
      probe<<<grid, 1024, 100 * 1024>>>(d);
      hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
      int mism = 0;
      for (int b = 0; b < grid; ++b) mism += ((h[b] & 0xf) != (unsigned)(b & 7));
      printf("grid %d rep %d: first 16 xcc ids:", grid, rep);
      for (int b = 0; b < 16; ++b) printf(" %u", h[b] & 0xf);
      printf("  | blocks whose xcc != blockIdx %% 8: %d (raw reg of block 0: 0x%x)\n", mism, h[0]);
    }
  }
  return 0;
}
