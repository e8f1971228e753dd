// back-to-back launches of k_acs_lds15 with ablation masks (see the kernel: 1 arithmetic, 4 decision stores,
// 8 metric stores, 16 minimum tracking) and the alternating minimum tracking the library uses
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DL15_DECPERM=0] -o scratch/l15_bench scratch/l15_bench.hip
#include "../isee3-decoder_amd/csrc/v224_hip.hip"

template <int ABL, bool ALT>
static double run15(int nlaunch, uint16_t *m0, uint16_t *m1, uint32_t *rows, int nrows, uint8_t *syms, V224Dev *ds,
                    uint32_t *rowmeta, hipStream_t st) {
  hipFuncSetAttribute((const void *)k_acs_lds15<ABL, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, L15_LDS_BYTES);
  hipFuncSetAttribute((const void *)k_acs_lds15<ABL, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, L15_LDS_BYTES);
  hipFuncSetAttribute((const void *)k_acs_lds15<ABL, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, L15_LDS_BYTES);
  k_init<<<V224_NSTATES / 8 / 256, 256, 0, st>>>(m0, 0, ds, rowmeta, nrows);
  k_init_start<<<1, 1, 0, st>>>(m0, 0);
  uint16_t *m[2] = { m0, m1 };
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto go = [&](int n, unsigned pass0) {
    for (int i = 0; i < n; i++) {
      unsigned pass = pass0 + i; int row0 = (int)((pass * 15) % (unsigned)nrows);
      const uint8_t *sy = syms + 2 * ((pass * 15) % 4000);
      if (!ALT) k_acs_lds15<ABL, true, true><<<256, 1024, L15_LDS_BYTES, st>>>(m[pass & 1], m[(pass & 1) ^ 1], rows, row0, nrows, sy, ds, pass, rowmeta);
      else if (pass & 1) k_acs_lds15<ABL, false, true><<<256, 1024, L15_LDS_BYTES, st>>>(m[pass & 1], m[(pass & 1) ^ 1], rows, row0, nrows, sy, ds, pass, rowmeta);
      else k_acs_lds15<ABL, true, false><<<256, 1024, L15_LDS_BYTES, st>>>(m[pass & 1], m[(pass & 1) ^ 1], rows, row0, nrows, sy, ds, pass, rowmeta);
    }
  };
  go(100, 0); hipStreamSynchronize(st);
  hipEventRecord(a, st); go(nlaunch, 100); hipEventRecord(b, st); hipStreamSynchronize(st);
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3 / nlaunch;
}

// two independent decoders on two streams: with L15_ALIAS their workgroups share the CUs
static double run15_dual(int nlaunch, uint16_t *m0, uint16_t *m1, uint16_t *n0, uint16_t *n1, uint32_t *rows, uint32_t *rows2, int nrows,
                         uint8_t *syms, V224Dev *ds, V224Dev *ds2, uint32_t *rowmeta, uint32_t *rowmeta2, hipStream_t st, hipStream_t st2) {
  k_init<<<V224_NSTATES / 8 / 256, 256, 0, st>>>(m0, 0, ds, rowmeta, nrows);  k_init_start<<<1, 1, 0, st>>>(m0, 0);
  k_init<<<V224_NSTATES / 8 / 256, 256, 0, st2>>>(n0, 0, ds2, rowmeta2, nrows); k_init_start<<<1, 1, 0, st2>>>(n0, 0);
  uint16_t *m[2] = { m0, m1 }, *n[2] = { n0, n1 };
  hipEvent_t a, b, c; hipEventCreate(&a); hipEventCreate(&b); hipEventCreate(&c);
  auto go = [&](int cnt, unsigned pass0) {
    for (int i = 0; i < cnt; i++) {
      unsigned pass = pass0 + i; int row0 = (int)((pass * 15) % (unsigned)nrows);
      const uint8_t *sy = syms + 2 * ((pass * 15) % 4000);
      if (pass & 1) {
        k_acs_lds15<0, false, true><<<256, 1024, L15_LDS_BYTES, st>>>(m[pass & 1], m[(pass & 1) ^ 1], rows, row0, nrows, sy, ds, pass, rowmeta);
        k_acs_lds15<0, false, true><<<256, 1024, L15_LDS_BYTES, st2>>>(n[pass & 1], n[(pass & 1) ^ 1], rows2, row0, nrows, sy, ds2, pass, rowmeta2);
      } else {
        k_acs_lds15<0, true, false><<<256, 1024, L15_LDS_BYTES, st>>>(m[pass & 1], m[(pass & 1) ^ 1], rows, row0, nrows, sy, ds, pass, rowmeta);
        k_acs_lds15<0, true, false><<<256, 1024, L15_LDS_BYTES, st2>>>(n[pass & 1], n[(pass & 1) ^ 1], rows2, row0, nrows, sy, ds2, pass, rowmeta2);
      }
    }
  };
  go(100, 0); hipStreamSynchronize(st); hipStreamSynchronize(st2);
  hipEventRecord(a, st); hipStreamWaitEvent(st2, a, 0);
  go(nlaunch, 100);
  hipEventRecord(c, st2); hipStreamWaitEvent(st, c, 0); hipEventRecord(b, st);
  hipStreamSynchronize(st); hipStreamSynchronize(st2);
  float ms = 0; hipEventElapsedTime(&ms, a, b); return ms * 1e3 / nlaunch;
}

int main(int argc, char **argv) {
  int nlaunch = argc > 1 ? atoi(argv[1]) : 2000;
  hipStream_t st;
  if (getenv("L15_PRIO")) { int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi); hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi); printf("high-priority stream (%d)\n", hi); }
  else hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  uint16_t *m0, *m1; uint32_t *rows, *rowmeta; uint8_t *syms; V224Dev *ds;
  int nrows = 1500;
  hipMalloc(&m0, V224_NSTATES * 2); hipMalloc(&m1, V224_NSTATES * 2);
  hipMalloc(&rows, (size_t)nrows * V224_ROWWORDS * 4); hipMalloc(&rowmeta, nrows * 4);
  hipMalloc(&syms, 8192 + 64); hipMalloc(&ds, sizeof(V224Dev));
  std::vector<uint8_t> h(8192 + 64);
  for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)(rand() & 0xff);
  hipMemcpy(syms, h.data(), h.size(), hipMemcpyHostToDevice);
  printf("m0 %p m1 %p rows %p rowmeta %p syms %p ds %p\n", (void *)m0, (void *)m1, (void *)rows, (void *)rowmeta, (void *)syms, (void *)ds);
  for (int rep = 0; rep < 1; rep++) {
#define L(ABL, ALT, what) printf("LDS15 DIRECTOUT=%d L0DIRECT=%d ABL=%2d alt=%d %-34s: %7.2f us/launch\n", L15_DIRECTOUT, L15_L0DIRECT, ABL, ALT, what, run15<ABL, ALT>(nlaunch, m0, m1, rows, nrows, syms, ds, rowmeta, st));
    L(0, false, "full, min in and out")
    L(0, true, "full, alternating min (library)")
    L(1, true, "no arithmetic (memory + LDS)")
    L(12, true, "no stores (loads + LDS + arith)")
    L(13, true, "loads + LDS only")
    L(4, true, "no decision stores")
    L(8, true, "no metric stores")
    L(16, false, "no min tracking")
    L(29, false, "loads + LDS, no min")
    {
      uint16_t *n0, *n1; uint32_t *rows2, *rowmeta2; V224Dev *ds2; hipStream_t st2;
      hipMalloc(&n0, V224_NSTATES * 2); hipMalloc(&n1, V224_NSTATES * 2); hipMalloc(&rows2, (size_t)nrows * V224_ROWWORDS * 4);
      hipMalloc(&rowmeta2, nrows * 4); hipMalloc(&ds2, sizeof(V224Dev)); hipStreamCreateWithFlags(&st2, hipStreamNonBlocking);
      double t = run15_dual(nlaunch, m0, m1, n0, n1, rows, rows2, nrows, syms, ds, ds2, rowmeta, rowmeta2, st, st2);
      printf("LDS15 ALIAS=%d two decoders on two streams: %7.2f us per launch PAIR (%.2f us per launch-equivalent)\n", L15_ALIAS, t, t / 2);
    }
  }
  return 0;
}
