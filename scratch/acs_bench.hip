// micro-benchmark of ACS kernel variants: back-to-back dependent launches, wall time per launch
#include "../isee3-decoder_amd/csrc/v224_hip.hip"
#include <chrono>

__global__ __launch_bounds__(256) void k_empty(unsigned *p) { if (threadIdx.x == 9999) p[0] = 1; }
__global__ __launch_bounds__(256) void k_minonly(V224Dev *ds, unsigned pass, unsigned *p) {
  unsigned a = input_min(ds, pass);
  output_min(ds, pass, a + threadIdx.x, 0);
}
__global__ __launch_bounds__(256) void k_touch(const uint32_t *in, uint32_t *out, int per) {
  unsigned t = blockIdx.x * 256 + threadIdx.x; unsigned acc = 0;
  for (int i = 0; i < per; i++) acc += in[(size_t)i * (1u << 17) + t];
  out[t] = acc;
}
// store-pattern test: 512 blocks x 256 threads, each thread writes 128 B (total 16 MiB)
template <int PAT>
__global__ __launch_bounds__(256) void k_storepat(uint32_t *out, unsigned seed) {
  unsigned t = blockIdx.x * 256 + threadIdx.x;
  uint4 v = make_uint4(t, seed, t ^ seed, 7);
  uint4 *o = reinterpret_cast<uint4 *>(out);
  if (PAT == 0) {            // row per lane: thread t owns bytes [128t, 128t+128)
#pragma unroll
    for (int j = 0; j < 8; j++) o[t * 8 + j] = v;
  } else {                   // lane-contiguous: wave w owns 8 KiB, instruction j writes 1 KiB contiguous
    unsigned w = t >> 6, l = t & 63;
#pragma unroll
    for (int j = 0; j < 8; j++) o[w * 512 + j * 64 + l] = v;
  }
}
template <int PAT>
static double run_store(int nlaunch, uint32_t *a, uint32_t *b, hipStream_t st) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto go = [&](int n) { for (int i = 0; i < n; i++) k_storepat<PAT><<<512, 256, 0, st>>>((i & 1) ? a : b, (unsigned)i); };
  go(100); hipStreamSynchronize(st); hipEventRecord(e0, st); go(nlaunch); hipEventRecord(e1, st); hipStreamSynchronize(st);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3 / nlaunch;
}
template <int ABL, bool ALT = false, int TILES = 1>
static double run_lds8(int nlaunch, uint16_t *m0, uint16_t *m1, uint32_t *rows, int nrows, uint8_t *syms, V224Dev *ds,
                       uint32_t *rowmeta, hipStream_t st) {
  k_init<<<V224_NSTATES / 8 / 256, 256, 0, st>>>(m0, 0, ds, rowmeta, nrows);
  k_init_start<<<1, 1, 0, st>>>(m0, 0);
  uint16_t *m[2] = { m0, m1 };
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto go = [&](int n, unsigned pass0) {
    for (int i = 0; i < n; i++) {
      unsigned pass = pass0 + i;
      int row0 = (int)((pass * 8) % (unsigned)(nrows - 8));
      if (!ALT) k_acs_lds8<ABL, true, true, TILES><<<512 / TILES, 512, 0, st>>>(m[pass & 1], m[(pass & 1) ^ 1], rows, row0, syms + 2 * ((pass * 8) % 4096), ds, pass, rowmeta);
      else if (pass & 1) k_acs_lds8<ABL, false, true, TILES><<<512 / TILES, 512, 0, st>>>(m[pass & 1], m[(pass & 1) ^ 1], rows, row0, syms + 2 * ((pass * 8) % 4096), ds, pass, rowmeta);
      else k_acs_lds8<ABL, true, false, TILES><<<512 / TILES, 512, 0, st>>>(m[pass & 1], m[(pass & 1) ^ 1], rows, row0, syms + 2 * ((pass * 8) % 4096), ds, pass, rowmeta);
    }
  };
  go(200, 0); hipStreamSynchronize(st); hipEventRecord(a, st); go(nlaunch, 200); hipEventRecord(b, st); hipStreamSynchronize(st);
  float ms = 0; hipEventElapsedTime(&ms, a, b); return ms * 1e3 / nlaunch;
}
static double run_lds8_graph(int nlaunch, uint16_t *m0, uint16_t *m1, uint32_t *rows, int nrows, uint8_t *syms, V224Dev *ds,
                             uint32_t *rowmeta, hipStream_t st) {
  k_init<<<V224_NSTATES / 8 / 256, 256, 0, st>>>(m0, 0, ds, rowmeta, nrows);
  k_init_start<<<1, 1, 0, st>>>(m0, 0);
  hipStreamSynchronize(st);
  uint16_t *m[2] = { m0, m1 };
  const int G = 128;                       // launches per graph (even: buffers and minima parity repeat)
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < G; i++) {
    unsigned pass = i; int row0 = (i * 8) % (nrows - 8);
    k_acs_lds8<0><<<512, 512, 0, st>>>(m[pass & 1], m[(pass & 1) ^ 1], rows, row0, syms + 2 * ((pass * 8) % 4096), ds, pass, rowmeta);
  }
  hipStreamEndCapture(st, &g);
  if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("graph instantiate failed\n"); return -1; }
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  int reps = nlaunch / G;
  hipEventRecord(a, st);
  for (int r = 0; r < reps; r++) hipGraphLaunch(ge, st);
  hipEventRecord(b, st); hipStreamSynchronize(st);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3 / (reps * G);
}
// two independent decoders on two streams: do their launches overlap on the device?
static double run_lds8_dual(int nlaunch, uint16_t *m0, uint16_t *m1, uint16_t *n0, uint16_t *n1, uint32_t *rows, uint32_t *rows2, int nrows,
                            uint8_t *syms, V224Dev *ds, V224Dev *ds2, uint32_t *rowmeta, uint32_t *rowmeta2, hipStream_t st, hipStream_t st2) {
  k_init<<<V224_NSTATES / 8 / 256, 256, 0, st>>>(m0, 0, ds, rowmeta, nrows);  k_init_start<<<1, 1, 0, st>>>(m0, 0);
  k_init<<<V224_NSTATES / 8 / 256, 256, 0, st2>>>(n0, 0, ds2, rowmeta2, nrows); k_init_start<<<1, 1, 0, st2>>>(n0, 0);
  uint16_t *m[2] = { m0, m1 }, *n[2] = { n0, n1 };
  hipEvent_t a, b, c; hipEventCreate(&a); hipEventCreate(&b); hipEventCreate(&c);
  auto go = [&](int cnt, unsigned pass0) {
    for (int i = 0; i < cnt; i++) {
      unsigned pass = pass0 + i; int row0 = (int)((pass * 8) % (unsigned)(nrows - 8));
      k_acs_lds8<0><<<512, 512, 0, st>>>(m[pass & 1], m[(pass & 1) ^ 1], rows, row0, syms + 2 * ((pass * 8) % 4096), ds, pass, rowmeta);
      k_acs_lds8<0><<<512, 512, 0, st2>>>(n[pass & 1], n[(pass & 1) ^ 1], rows2, row0, syms + 2 * ((pass * 8) % 4096), ds2, pass, rowmeta2);
    }
  };
  go(100, 0); hipStreamSynchronize(st); hipStreamSynchronize(st2);
  hipEventRecord(a, st); hipStreamWaitEvent(st2, a, 0);
  go(nlaunch, 100);
  hipEventRecord(c, st2); hipStreamWaitEvent(st, c, 0); hipEventRecord(b, st);
  hipStreamSynchronize(st); hipStreamSynchronize(st2);
  float ms = 0; hipEventElapsedTime(&ms, a, b); return ms * 1e3 / nlaunch;
}
template <int MODE>
static double run_misc(int nlaunch, int nblk, V224Dev *ds, uint32_t *buf, uint32_t *buf2, hipStream_t st) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto go = [&](int n) { for (int i = 0; i < n; i++) {
      if (MODE == 0) k_empty<<<nblk, 256, 0, st>>>(buf);
      else if (MODE == 1) k_minonly<<<nblk, 256, 0, st>>>(ds, (unsigned)i, buf);
      else k_touch<<<nblk, 256, 0, st>>>((i & 1) ? buf : buf2, (i & 1) ? buf2 : buf, 32); } };
  go(100); hipStreamSynchronize(st); hipEventRecord(a, st); go(nlaunch); hipEventRecord(b, st); hipStreamSynchronize(st);
  float ms = 0; hipEventElapsedTime(&ms, a, b); return ms * 1e3 / nlaunch;
}
template <int K, int CP, int CPD, int ABL = 0>
static double run(int nlaunch, uint16_t *m0, uint16_t *m1, uint32_t *rows, int nrows, uint8_t *syms, V224Dev *ds,
                  uint32_t *rowmeta, hipStream_t st) {
  k_init<<<V224_NSTATES / 8 / 256, 256, 0, st>>>(m0, 0, ds, rowmeta, nrows);
  k_init_start<<<1, 1, 0, st>>>(m0, 0);
  uint16_t *m[2] = { m0, m1 };
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  auto go = [&](int n, unsigned pass0) {
    for (int i = 0; i < n; i++) {
      unsigned pass = pass0 + i;
      int row0 = (int)((pass * K) % (unsigned)(nrows - K));
      k_acs_fused<K, CP, CPD, ABL><<<(1u << (V224_SBITS - K - 1)) / 256, 256, 0, st>>>(
          (const uint32_t *)m[pass & 1], (uint32_t *)m[(pass & 1) ^ 1], rows, row0, syms + 2 * ((pass * K) % 4096), ds, pass, rowmeta);
    }
  };
  go(200, 0);
  hipStreamSynchronize(st);
  hipEventRecord(a, st);
  go(nlaunch, 200);
  hipEventRecord(b, st);
  hipStreamSynchronize(st);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms * 1e3 / nlaunch;
}

int main(int argc, char **argv) {
  int nlaunch = argc > 1 ? atoi(argv[1]) : 3000;
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  uint16_t *m0, *m1; uint32_t *rows, *rowmeta; uint8_t *syms; V224Dev *ds;
  int nrows = 2048;
  hipMalloc(&m0, V224_NSTATES * 2); hipMalloc(&m1, V224_NSTATES * 2);
  hipMalloc(&rows, (size_t)nrows * V224_ROWWORDS * 4); hipMalloc(&rowmeta, nrows * 4);
  hipMalloc(&syms, 8192 + 64); hipMalloc(&ds, sizeof(V224Dev));
  std::vector<uint8_t> h(8192 + 64);
  for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)(rand() & 0xff);
  hipMemcpy(syms, h.data(), h.size(), hipMemcpyHostToDevice);
#define RUN(K, ABL) { double t = run<K, 0, 0, ABL>(nlaunch, m0, m1, rows, nrows, syms, ds, rowmeta, st); printf("K=%d ABL=%d (%s) : %7.2f us/launch  %6.3f us/bit\n", K, ABL, ABL == 0 ? "full" : ABL == 1 ? "memory only" : "compute only", t, t / K); }
  k_init<<<V224_NSTATES / 8 / 256, 256, 0, st>>>(m0, 0, ds, rowmeta, nrows);
  for (int nb : {256, 512, 1024, 2048}) {
    printf("empty kernel %4d blocks: %6.2f us | min-only: %6.2f us | touch(32 loads/thread, 512 blk only): %6.2f us\n", nb,
           run_misc<0>(nlaunch, nb, ds, (uint32_t *)m0, (uint32_t *)m1, st), run_misc<1>(nlaunch, nb, ds, (uint32_t *)m0, (uint32_t *)m1, st),
           run_misc<2>(nlaunch, 512, ds, (uint32_t *)m0, (uint32_t *)m1, st));
  }
  printf("store 16 MiB row-per-lane: %6.2f us | lane-contiguous: %6.2f us\n", run_store<0>(nlaunch, (uint32_t *)m0, (uint32_t *)m1, st), run_store<1>(nlaunch, (uint32_t *)m0, (uint32_t *)m1, st));
  for (int rep = 0; rep < 2; rep++) {
#define L(ABL, what) printf("LDS8 PK=%d ABL=%2d %-30s: %7.2f us/launch\n", LDS8_PK, ABL, what, run_lds8<ABL>(nlaunch, m0, m1, rows, nrows, syms, ds, rowmeta, st));
    L(0, "full") L(14, "arith + LDS only") L(30, "arith + LDS, no min")
    {
      uint16_t *n0, *n1; uint32_t *rows2, *rowmeta2; V224Dev *ds2; hipStream_t st2;
      hipMalloc(&n0, V224_NSTATES * 2); hipMalloc(&n1, V224_NSTATES * 2); hipMalloc(&rows2, (size_t)nrows * V224_ROWWORDS * 4);
      hipMalloc(&rowmeta2, nrows * 4); hipMalloc(&ds2, sizeof(V224Dev)); hipStreamCreateWithFlags(&st2, hipStreamNonBlocking);
      double t = run_lds8_dual(nlaunch, m0, m1, n0, n1, rows, rows2, nrows, syms, ds, ds2, rowmeta, rowmeta2, st, st2);
      printf("LDS8 ALIAS=%d two decoders on two streams: %7.2f us per launch PAIR (%.2f us per launch-equivalent)\n", LDS8_ALIAS, t, t / 2);
      hipFree(n0); hipFree(n1); hipFree(rows2); hipFree(rowmeta2); hipFree(ds2);
    }
    printf("LDS8 via hipGraph (128 launches per graph): %7.2f us/launch\n", run_lds8_graph(nlaunch, m0, m1, rows, nrows, syms, ds, rowmeta, st));
    printf("LDS8 2 tiles per workgroup: %7.2f us/launch, alternating min: %7.2f\n", run_lds8<0, false, 2>(nlaunch, m0, m1, rows, nrows, syms, ds, rowmeta, st), run_lds8<0, true, 2>(nlaunch, m0, m1, rows, nrows, syms, ds, rowmeta, st));
    printf("LDS8 alternating min tracking: %7.2f us/launch\n", run_lds8<0, true>(nlaunch, m0, m1, rows, nrows, syms, ds, rowmeta, st));
    double t5 = run<5, 0, 0, 0>(nlaunch, m0, m1, rows, nrows, syms, ds, rowmeta, st);
    printf("K=5  : %7.2f us/launch  %6.3f us/bit\n", t5, t5 / 5);
  }
  for (int rep = 0; rep < 0; rep++) {
    RUN(5, 0) RUN(5, 1) RUN(5, 2) RUN(6, 0) RUN(6, 1) RUN(6, 2) RUN(4, 0) RUN(4, 1) RUN(4, 2) RUN(3, 0) RUN(3, 1) RUN(3, 2)
  }
  return 0;
}
