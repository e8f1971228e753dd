// does the speed of the LDS15 pass depend on WHICH two allocations hold the metrics?  Six separate 16 MiB
// hipMallocs, every ordered pair; prints the memory-only (ABL=1) time per pair and the full time for a few.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DL15_DSTORE=1 -o scratch/l15_pairs scratch/l15_pairs.hip
#include "../isee3-decoder_amd/csrc/v224_hip.hip"

template <int ABL>
static double run15(int nlaunch, uint16_t *m0, uint16_t *m1, uint32_t *rows, int nrows, uint8_t *syms, V224Dev *ds,
                    uint32_t *rowmeta, hipStream_t st) {
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
      if (pass & 1) k_acs_lds15<ABL, false, true><<<256, 1024, L15_LDS_BYTES, st>>>(m[pass & 1], m[(pass & 1) ^ 1], rows, row0, nrows, sy, ds, pass, rowmeta);
      else k_acs_lds15<ABL, true, false><<<256, 1024, L15_LDS_BYTES, st>>>(m[pass & 1], m[(pass & 1) ^ 1], rows, row0, nrows, sy, ds, pass, rowmeta);
    }
  };
  go(50, 0); hipStreamSynchronize(st);
  hipEventRecord(a, st); go(nlaunch, 50); hipEventRecord(b, st); hipStreamSynchronize(st);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms * 1e3 / nlaunch;
}

int main(int argc, char **argv) {
  int nlaunch = argc > 1 ? atoi(argv[1]) : 400;
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  uint32_t *rows, *rowmeta; uint8_t *syms; V224Dev *ds;
  int nrows = 600;
  const size_t MiB = 1 << 20;
  hipMalloc(&rows, (size_t)nrows * V224_ROWWORDS * 4); hipMalloc(&rowmeta, nrows * 4);
  hipMalloc(&syms, 8192 + 64); hipMalloc(&ds, sizeof(V224Dev));
  std::vector<uint8_t> h(8192 + 64);
  for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)(rand() & 0xff);
  hipMemcpy(syms, h.data(), h.size(), hipMemcpyHostToDevice);
  uint16_t *buf[6];
  for (int i = 0; i < 6; i++) { hipMalloc(&buf[i], 16 * MiB); printf("buf%d %p\n", i, (void *)buf[i]); }
  printf("memory-only us (rows: m0 = buf i, cols: m1 = buf j)\n");
  double best = 1e9, worst = 0; int bi = 0, bj = 1, wi = 0, wj = 1;
  for (int i = 0; i < 6; i++) {
    printf("%4d: ", i);
    for (int j = 0; j < 6; j++) {
      if (i == j) { printf("     -"); continue; }
      double t = run15<1>(nlaunch, buf[i], buf[j], rows, nrows, syms, ds, rowmeta, st);
      printf("%6.2f", t);
      if (t < best) { best = t; bi = i; bj = j; }
      if (t > worst) { worst = t; wi = i; wj = j; }
    }
    printf("\n");
  }
  printf("in place (m1 == m0; timing only, racy by construction): memory-only %6.2f us, full %6.2f us\n",
         run15<1>(nlaunch, buf[0], buf[0], rows, nrows, syms, ds, rowmeta, st), run15<0>(nlaunch, buf[0], buf[0], rows, nrows, syms, ds, rowmeta, st));
  printf("full launch: best pair (%d,%d) %6.2f us, worst pair (%d,%d) %6.2f us\n", bi, bj,
         run15<0>(nlaunch, buf[bi], buf[bj], rows, nrows, syms, ds, rowmeta, st), wi, wj,
         run15<0>(nlaunch, buf[wi], buf[wj], rows, nrows, syms, ds, rowmeta, st));
  return 0;
}
