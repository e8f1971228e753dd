// feasibility: VOPC SDWA compare into an SGPR pair + scalar stores of the lane masks
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define NMASK 256   // masks per wave (8 stages x 32)
__global__ __launch_bounds__(512) void k_sstore(const unsigned *in, unsigned long long *out, int reps) {
  unsigned a = in[blockIdx.x * 512 + threadIdx.x], b = in[(blockIdx.x * 512 + threadIdx.x + 12345) % (512 * 512)];
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned long long *dst = out + ((size_t)blockIdx.x * 8 + wave) * NMASK;   // wave-uniform
  for (int r = 0; r < reps; r++) {
#pragma unroll 8
    for (int k = 0; k < NMASK; k += 2) {
      unsigned long long m0, m1;
      unsigned x = a + k, y = b ^ (k * 77);
      asm volatile("v_cmp_lt_u16_sdwa %0, %2, %3 src0_sel:WORD_0 src1_sel:WORD_0\n\t"
                   "v_cmp_lt_u16_sdwa %1, %2, %3 src0_sel:WORD_1 src1_sel:WORD_1\n\t"
                   : "=s"(m0), "=s"(m1) : "v"(x), "v"(y));
      asm volatile("s_store_dwordx4 %0, %1, %2" :: "s"(__uint128_t(m0) | (__uint128_t(m1) << 64)), "s"(dst), "i"(0) : "memory");
      dst += 2;
    }
    dst -= NMASK;
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}
int main() {
  const int nb = 512;
  std::vector<unsigned> h(512 * 512);
  for (auto &v : h) v = (unsigned)rand() * 2654435761u;
  unsigned *din; unsigned long long *dout;
  hipMalloc(&din, h.size() * 4); hipMalloc(&dout, (size_t)nb * 8 * NMASK * 8);
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemset(dout, 0, (size_t)nb * 8 * NMASK * 8);
  k_sstore<<<nb, 512>>>(din, dout, 1);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  std::vector<unsigned long long> o((size_t)nb * 8 * NMASK);
  hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (int blk = 0; blk < nb; blk++) for (int w = 0; w < 8; w++) for (int k = 0; k < NMASK; k += 2) {
    unsigned long long e0 = 0, e1 = 0;
    for (int l = 0; l < 64; l++) {
      int t = w * 64 + l;
      unsigned a = h[blk * 512 + t], b = h[(blk * 512 + t + 12345) % (512 * 512)];
      unsigned x = a + k, y = b ^ (k * 77);
      if ((x & 0xffff) < (y & 0xffff)) e0 |= 1ull << l;
      if ((x >> 16) < (y >> 16)) e1 |= 1ull << l;
    }
    size_t idx = ((size_t)blk * 8 + w) * NMASK + k;
    if (o[idx] != e0 || o[idx + 1] != e1) bad++;
  }
  printf("mask pairs wrong: %zu of %zu\n", bad, o.size() / 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int reps : {1, 4}) {
    hipEventRecord(e0); for (int i = 0; i < 20; i++) k_sstore<<<nb, 512>>>(din, dout, reps); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("reps %d: %.2f us per launch (%d MiB of scalar stores, %d cmp pairs per wave)\n", reps, ms * 1e3 / 20, reps * 8, reps * NMASK / 2);
  }
  return 0;
}
