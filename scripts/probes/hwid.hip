// Where do the waves of a workgroup land?  Prints, for a few workgroups of 256 threads, the SIMD / CU / SE of each wave (HW_REG_HW_ID).
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/hwid scripts/probes/hwid.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 4) void k(unsigned* out, int spin)
{
  __shared__ unsigned pad[6000];
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  pad[threadIdx.x] = id;
  for (int i = 0; i < spin; ++i) asm volatile("s_nop 15");
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = id + (pad[threadIdx.x] & 0u);
}
int main()
{
  const int B = 2048;
  unsigned* d; hipMalloc(&d, B * 4 * sizeof(unsigned));
  hipLaunchKernelGGL(k, dim3(B), dim3(256), 0, 0, d, 20000);
  std::vector<unsigned> h(B * 4);
  hipMemcpy(h.data(), d, B * 4 * sizeof(unsigned), hipMemcpyDeviceToHost);
  int same = 0, distinct = 0;
  for (int b = 0; b < B; ++b) {
    unsigned m = 0;
    for (int w = 0; w < 4; ++w) m |= 1u << ((h[b * 4 + w] >> 4) & 3);
    if (__builtin_popcount(m) == 4) ++distinct; else if (__builtin_popcount(m) == 1) ++same;
    if (b < 6 || (b > 1024 && b < 1028)) { printf("block %d:", b); for (int w = 0; w < 4; ++w) { unsigned x = h[b * 4 + w]; printf("  wave%d simd %u cu %u sh %u se %u waveslot %u", w, (x >> 4) & 3, (x >> 8) & 15, (x >> 12) & 1, (x >> 13) & 7, x & 15); } printf("\n"); }
  }
  printf("blocks whose 4 waves sit on 4 different SIMDs: %d, on one SIMD: %d, of %d\n", distinct, same, B);
  return 0;
}
