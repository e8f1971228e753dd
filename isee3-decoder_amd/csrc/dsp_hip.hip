// dsp_hip.hip -- pmdemod / symdemod kernels for MI355X (gfx950) behind include/isee3_dsp_hip.h.
//
// Everything here is HBM-bandwidth work on small per-element arithmetic: no MFMA, no LDS tiling
// beyond block reductions.  Built with -ffp-contract=off: the reference's double arithmetic is
// non-fused (x86-64 baseline), and symdemod's output must be byte-identical.
//
// symdemod (symdemod.c:202-335): the window's samples get an exact int64 prefix sum, after which
//   every integrate-and-dump value is a difference of prefix entries.  timesearch = one thread per
//   timing offset, walking the symbols in order so the double energy accumulation rounds exactly
//   as the reference's loop does; adjacent threads read adjacent prefix entries (coalesced).
// pmdemod (pmdemod.c:204-368): int16 IQ -> double2, Stockham radix-2 double-precision FFT
//   (ping-pong in HBM, unit-stride reads), |X|^2 arg-max with the reference's "last maximum wins"
//   rule, closed-form carrier spin-down (see carrier_params.c), two tree reductions, quantise.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>

#pragma GCC visibility push(default)
#include "../../include/isee3_dsp_hip.h"
extern "C" void pmd_carrier_params(double cstep, uint64_t *u_hi, uint64_t *u_lo, double *logrho);
#pragma GCC visibility pop

static thread_local char g_err[512] = "";
static std::atomic<int> g_device{-1};    // process-wide default device of create / alloc calls (any thread)
extern "C" const char *isee3dsp_last_error(void) { return g_err; }
extern "C" int isee3dsp_set_device(int dev) {
  if (hipSetDevice(dev) != hipSuccess) return -1;
  g_device.store(dev, std::memory_order_release);
  return 0;
}
#define CHK(expr)                                                                                \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      snprintf(g_err, sizeof g_err, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      goto fail;                                                                                 \
    }                                                                                            \
  } while (0)

extern "C" void *isee3dsp_dev_alloc(size_t bytes) {
  void *d = nullptr;
  { int gd = g_device.load(std::memory_order_acquire); if (gd >= 0 && hipSetDevice(gd) != hipSuccess) return nullptr; }
  if (hipMalloc(&d, bytes ? bytes : 1) != hipSuccess) return nullptr;
  return d;
}
extern "C" void isee3dsp_dev_free(void *d) { if (d) (void)hipFree(d); }
extern "C" int isee3dsp_h2d(void *d, const void *h, size_t n) { return hipMemcpy(d, h, n, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1; }
extern "C" int isee3dsp_d2h(void *h, const void *d, size_t n) { return hipMemcpy(h, d, n, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; }

// Stream of a front-end handle, chosen by isee3dsp_share_stream(mode) in the creating thread: 0 its own (default); 1 ONE
// stream per device shared by all such handles (never destroyed); 2 the null stream.  Why: MI355X feeds compute queues
// through four hardware pipes; HIP's hardware queues land on them in creation order (queue 0 = the null stream's), and
// only queues 0..3 run side by side without one starving the other (v224_hip.hip / DESIGN.md "One stream per decoder").
// The in-process chain has four things to run at once -- two Viterbi decoders, pmdemod, symdemod: decoders on their own
// streams, pmdemod on the shared front-end stream (mode 1), symdemod on the null stream (mode 2), whose queue exists in
// every process and is otherwise idle in this library.  All other streams here are hipStreamNonBlocking, so nothing
// synchronises with the null stream implicitly; a host application that keeps the null stream busy itself shares it with
// symdemod's short kernels (correct, slower): ISEE3_CHAIN_SY_NULL=0 then puts both stages on the mode-1 stream.
// (ISEE3DSP_HIGH_PRIORITY=1 asks for the highest stream priority for created streams; measured: no effect.)
static thread_local int t_share_stream = 0;
static hipStream_t g_shared_stream[64];
extern "C" void isee3dsp_share_stream(int on) { t_share_stream = on; }
static hipError_t dsp_stream_create_one(hipStream_t *st) {
  int least = 0, greatest = 0;
  const char *e = getenv("ISEE3DSP_HIGH_PRIORITY");
  if (e && atoi(e) && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest < least &&
      hipStreamCreateWithPriority(st, hipStreamNonBlocking, greatest) == hipSuccess) return hipSuccess;
  (void)hipGetLastError();
  return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}
static hipError_t dsp_stream_create(hipStream_t *st, int *owned) {
  int dev = 0;
  *owned = 1;
  if (t_share_stream == 2) { *st = nullptr; *owned = 0; return hipSuccess; }       // the null stream
  if (t_share_stream && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (!g_shared_stream[dev] && dsp_stream_create_one(&g_shared_stream[dev]) != hipSuccess) g_shared_stream[dev] = nullptr;
    if (g_shared_stream[dev]) { *st = g_shared_stream[dev]; *owned = 0; return hipSuccess; }
  }
  return dsp_stream_create_one(st);
}

static int grow(void **p, size_t *cap, size_t need) {
  if (*cap >= need) return 0;
  if (*p) (void)hipFree(*p);
  *p = nullptr; *cap = 0;
  if (hipMalloc(p, need) != hipSuccess) return -1;
  *cap = need;
  return 0;
}

// Small results and index tables travel through pinned, device-mapped host memory: kernels store scalars / short arrays
// straight into it (no copy command at all), and what has to be copied (index tables up, a flag down) is a truly
// asynchronous DMA instead of the staged, blocking copy a pageable buffer gets.  One stream synchronise per engine call.
struct Pin { void *h, *d; size_t cap; };
static int pin_grow(Pin *p, size_t need) {
  if (p->cap >= need) return 0;
  if (p->h) (void)hipHostFree(p->h);
  p->h = p->d = nullptr; p->cap = 0;
  if (need < 4096) need = 4096;
  if (hipHostMalloc(&p->h, need, hipHostMallocMapped) != hipSuccess) { p->h = nullptr; return -1; }
  if (hipHostGetDevicePointer(&p->d, p->h, 0) != hipSuccess) { (void)hipHostFree(p->h); p->h = nullptr; return -1; }
  p->cap = need;
  return 0;
}
static void pin_free(Pin *p) { if (p->h) (void)hipHostFree(p->h); p->h = p->d = nullptr; p->cap = 0; }

// ===========================================================================================
// symdemod
// ===========================================================================================
#define SCAN_BLOCK 4096                      // samples per workgroup (two rounds of 256 threads x 8)
#define SCAN_ROW 264                         // LDS row stride (8-byte words) of the transposed prefix image

struct Symd {
  int dev; hipStream_t st; int own_st;
  int cap, n;
  int16_t *d_s, *d_s2;                       // the window buffer and its ping-pong partner (store_slide)
  long long *d_P; long long *d_blk; int nblk_cap;
  void *d_idx; size_t idx_cap;
  void *d_e; size_t e_cap;
  void *d_out; size_t out_cap;
  void *d_sym; size_t sym_cap;
  void *d_part; size_t part_cap;
  void *d_terms; size_t terms_cap;
  unsigned *d_flag;
  int inexact_last;                          // the previous window's sums left the exactly-representable range
  Pin pin_idx, pin_e, pin_out, pin_hdr;      // index table staging; energies, symbols, {flags, energy sum} written by kernels
};

__device__ __forceinline__ long long wave_incl_scan(long long v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    long long u = __shfl_up(v, o, 64);
    if ((int)(threadIdx.x & 63) >= o) v += u;
  }
  return v;
}

// 8 consecutive samples of a 16-byte aligned group (d_s comes from hipMalloc, groups start at multiples of 8)
__device__ __forceinline__ void load8(const int16_t *__restrict__ s, long long base, int n, int (&x)[8]) {
  if (base + 8 <= n) {
    const uint4 w = *reinterpret_cast<const uint4 *>(s + base);
    const unsigned u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int k = 0; k < 4; k++) { x[2 * k] = (int)(short)(u[k] & 0xffffu); x[2 * k + 1] = (int)(short)(u[k] >> 16); }
  } else {
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = (base + k < n) ? (int)s[base + k] : 0;
  }
}

// block-local sums of 4096 samples (16-byte loads, lane-contiguous)
__global__ __launch_bounds__(256) void k_scan_partial(const int16_t *__restrict__ s, int n,
                                                      long long *__restrict__ blk) {
  long long acc = 0;
#pragma unroll
  for (int r = 0; r < 2; r++) {
    int x[8];
    load8(s, (long long)blockIdx.x * SCAN_BLOCK + r * 2048 + threadIdx.x * 8, n, x);
#pragma unroll
    for (int k = 0; k < 8; k++) acc += x[k];
  }
  acc = wave_incl_scan(acc);
  __shared__ long long ws[4];
  if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) blk[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
// P[0] = 0, P[i+1] = sum(s[0..i]).  Two rounds of 2048 samples: thread t scans its 8 consecutive samples (one 16-byte
// load), a wave scan and four wave totals give the offsets, and the 2048 prefix values go through LDS (k-major,
// padded rows) so that every global store instruction of a wave writes 512 contiguous bytes.
__global__ __launch_bounds__(256) void k_scan_final(const int16_t *__restrict__ s, int n,
                                                    const long long *__restrict__ blk,
                                                    long long *__restrict__ P) {
  __shared__ long long sh[8 * SCAN_ROW];
  __shared__ long long ws[2][4];
  __shared__ long long wc[4];
  const int t = threadIdx.x;
  // the workgroup's carry = the sum of all block sums before it, formed here from k_scan_partial's raw sums (exact integers:
  // any order) instead of by a one-workgroup scan kernel in between (22 us at 10 MS/s, one launch less per window)
  long long carry;
  {
    long long acc = 0;
    for (int i = t; i < (int)blockIdx.x; i += 256) acc += blk[i];
    acc = wave_incl_scan(acc);
    if ((t & 63) == 63) wc[t >> 6] = acc;
    __syncthreads();
    carry = wc[0] + wc[1] + wc[2] + wc[3];
  }
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const long long base = (long long)blockIdx.x * SCAN_BLOCK + r * 2048;
    int x[8];
    load8(s, base + t * 8, n, x);
    long long v[8], acc = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { acc += x[k]; v[k] = acc; }
    const long long inc = wave_incl_scan(acc);
    if ((t & 63) == 63) ws[r][t >> 6] = inc;
    __syncthreads();                                   // also: the previous round's image has been read
    long long off = carry + inc - acc;
    for (int w = 0; w < (t >> 6); w++) off += ws[r][w];
#pragma unroll
    for (int k = 0; k < 8; k++) sh[k * SCAN_ROW + t] = off + v[k];
    carry += ws[r][0] + ws[r][1] + ws[r][2] + ws[r][3];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int e = j * 256 + t;                       // element e of the round = thread e >> 3, slot e & 7
      if (base + e < n) P[base + e + 1] = sh[(e & 7) * SCAN_ROW + (e >> 3)];
    }
  }
  if (blockIdx.x == 0 && t == 0) P[0] = 0;
}

// the window buffer after memmove(samples, samples + slide, keep) (symdemod.c:101-112): everything beyond `keep` stays
// what it was.  Ping-pong: out becomes the new buffer.
// keep <= slide: source [slide, slide + keep) and destination [0, keep) do not overlap -- copied in place, 16 bytes per lane
// where both are aligned; the rest of the buffer is not touched at all (memmove's own semantics)
__global__ __launch_bounds__(256) void k_slide_inplace(int16_t *__restrict__ buf, int slide, int keep) {
  const long long stride = (long long)gridDim.x * 256;
  if ((slide & 7) == 0) {
    const int n8 = keep >> 3;
    const uint4 *src = reinterpret_cast<const uint4 *>(buf + slide);
    uint4 *dst = reinterpret_cast<uint4 *>(buf);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) dst[i] = src[i];
    for (long long i = ((long long)n8 << 3) + (long long)blockIdx.x * 256 + threadIdx.x; i < keep; i += stride) buf[i] = buf[i + slide];
  } else
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < keep; i += stride) buf[i] = buf[i + slide];
}
__global__ __launch_bounds__(256) void k_slide(const int16_t *__restrict__ in, int16_t *__restrict__ out, int slide, int keep, int cap) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < cap; i += (long long)gridDim.x * 256)
    out[i] = i < keep ? in[i + slide] : in[i];
}

// symdemod.c:260-335: thread t = timing offset; the energy is accumulated symbol by symbol IN ORDER in double, as the
// reference's loop does (this is the form that stays exact when sums leave the 2^53 range).  The integrator sums of
// TS_BATCH symbols are formed first -- their prefix reads are independent, so their latencies overlap -- and then added
// in order: 460 -> ~100 us per window at 10 MS/s (9 760 offsets x 1 024 symbols) with batches of 16; the kernel is pure load
// latency (the 1 024 dependent double additions of a thread are ~3 us), so the batch is 64: a quarter of the round trips.
#define TS_BATCH 64
__global__ __launch_bounds__(64) void k_timesearch(const long long *__restrict__ P, int lo,
                                                   const int *__restrict__ sw, int symbolclocks,
                                                   int nsymbols, int noff, double *__restrict__ energies) {
  int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= noff) return;
  const long long *Pb = P + lo + t;
  double energy = 0;
  int i = 0;
  if (symbolclocks == 1) {
    for (; i + TS_BATCH <= nsymbols; i += TS_BATCH) {
      // the 2 TS_BATCH + 1 prefix values of the batch: ALL loads are issued before the first use.  Left to itself the
      // compiler sinks each load next to its subtraction (two loads in flight: 1 024 dependent round trips per thread,
      // 150 us per window); the empty asm statements pin the order: loads above the memory barrier, values through "+v".
      long long pv[2 * TS_BATCH + 1];
#pragma unroll
      for (int u = 0; u <= 2 * TS_BATCH; u++) pv[u] = Pb[sw[2 * i + u]];
      asm volatile("" ::: "memory");
#pragma unroll
      for (int u = 0; u <= 2 * TS_BATCH; u++) asm volatile("" : "+v"(pv[u]));
#pragma unroll
      for (int u = 0; u < TS_BATCH; u++) {
        const long long a = pv[2 * u], b = pv[2 * u + 1], c = pv[2 * u + 2];
        const long long sym = -(b - a) + (c - b);
        energy += (double)(sym * sym);
      }
    }
  }
  int k = 2 * i * symbolclocks;
  for (; i < nsymbols; i++) {
    long long sym = 0;
    for (int j = 0; j < symbolclocks; j++, k += 2) {
      long long a = Pb[sw[k]], b = Pb[sw[k + 1]], c = Pb[sw[k + 2]];
      sym += -(b - a) + (c - b);
    }
    energy += (double)(sym * sym);
  }
  energies[t] = energy;
}

// The same ordered sum with four waves per 64 offsets: the kernel above is pure load latency -- it reads the window's WHOLE
// prefix array (9 761 offsets x 2 049 boundaries x 8 B = 160 MB at 10 MS/s) with 153 waves and at most 63 loads each in
// flight.  Here wave w of a workgroup forms the terms (double)(sym * sym) of symbols [128 r + 32 w, + 32) for the
// workgroup's 64 offsets -- the same integer products, the same conversions -- and parks them in LDS; wave 0 then adds
// the round's 128 terms IN ORDER (the reference's order of additions, bit for bit) while nothing else changes.  Four times
// the loads in flight.  symbolclocks == 1 only.
#define TS4_SYMS 32                      /* symbols per wave and round (2 x 32 + 1 loads) */
__global__ __launch_bounds__(256) void k_timesearch4(const long long *__restrict__ P, int lo, const int *__restrict__ sw,
                                                     int nsymbols, int noff, double *__restrict__ energies) {
  extern __shared__ double ts4_terms[];                    // [4 * TS4_SYMS][64]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + lane;
  const long long *Pb = P + lo + (t < noff ? t : noff - 1);       // lanes beyond the last offset repeat it (their result is dropped)
  double energy = 0;
  for (int r0 = 0; r0 < nsymbols; r0 += 4 * TS4_SYMS) {
    const int i0 = r0 + w * TS4_SYMS;
    if (i0 + TS4_SYMS <= nsymbols) {
      long long pv[2 * TS4_SYMS + 1];
#pragma unroll
      for (int u = 0; u <= 2 * TS4_SYMS; u++) pv[u] = Pb[sw[2 * i0 + u]];
      asm volatile("" ::: "memory");
#pragma unroll
      for (int u = 0; u <= 2 * TS4_SYMS; u++) asm volatile("" : "+v"(pv[u]));
#pragma unroll
      for (int u = 0; u < TS4_SYMS; u++) {
        const long long a = pv[2 * u], b = pv[2 * u + 1], c = pv[2 * u + 2];
        const long long sym = -(b - a) + (c - b);
        ts4_terms[(w * TS4_SYMS + u) * 64 + lane] = (double)(sym * sym);
      }
    } else {
      for (int u = 0; u < TS4_SYMS && i0 + u < nsymbols; u++) {
        const long long a = Pb[sw[2 * (i0 + u)]], b = Pb[sw[2 * (i0 + u) + 1]], c = Pb[sw[2 * (i0 + u) + 2]];
        const long long sym = -(b - a) + (c - b);
        ts4_terms[(w * TS4_SYMS + u) * 64 + lane] = (double)(sym * sym);
      }
    }
    __syncthreads();
    if (w == 0) {
      const int n = nsymbols - r0 < 4 * TS4_SYMS ? nsymbols - r0 : 4 * TS4_SYMS;
      for (int u = 0; u < n; u++) energy += ts4_terms[u * 64 + lane];
    }
    __syncthreads();                                       // the terms have been consumed: the next round may overwrite them
  }
  if (w == 0 && t < noff) energies[t] = energy;
}

// Parallel form of the same sum.  The reference adds (double)(sym*sym) symbol by symbol; as long as
// every term and the running total stay below 2^53 each partial sum is an exactly representable integer
// and the order of addition cannot matter, so the sum may be formed in u64 by many threads.  Thread =
// offset (coalesced prefix reads), block = a slice of TS_SLICE symbols.  Any term or total >= 2^53 raises
// `inexact`, and the host then runs the sequential kernel above for that window instead.
#define TS_SLICE 16
#define TS_LIMIT (1ull << 53)
__global__ __launch_bounds__(256) void k_timesearch_part(const long long *__restrict__ P, int lo,
                                                         const int *__restrict__ sw, int symbolclocks, int nsymbols,
                                                         int noff, unsigned long long *__restrict__ part,
                                                         unsigned *__restrict__ inexact) {
  int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= noff) return;
  const long long *Pb = P + lo + t;
  int i0 = blockIdx.y * TS_SLICE, i1 = i0 + TS_SLICE < nsymbols ? i0 + TS_SLICE : nsymbols;
  unsigned long long acc = 0; unsigned bad = 0;
  int k = 2 * i0 * symbolclocks;
  if (symbolclocks == 1 && i1 - i0 == TS_SLICE) {      // a whole slice: its 2 TS_SLICE + 1 prefix values in flight at once (see k_timesearch)
    long long pv[2 * TS_SLICE + 1];
#pragma unroll
    for (int u = 0; u <= 2 * TS_SLICE; u++) pv[u] = Pb[sw[k + u]];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int u = 0; u <= 2 * TS_SLICE; u++) asm volatile("" : "+v"(pv[u]));
#pragma unroll
    for (int u = 0; u < TS_SLICE; u++) {
      const long long a = pv[2 * u], b = pv[2 * u + 1], c = pv[2 * u + 2];
      const long long sym = -(b - a) + (c - b);
      const unsigned long long m = (unsigned long long)(sym < 0 ? -sym : sym);
      if (m >= (1ull << 26)) bad = 1;
      acc += m * m;
      if (acc >= TS_LIMIT) bad = 1;
    }
    i0 = i1;
  }
  for (int i = i0; i < i1; i++) {
    long long sym = 0;
    for (int j = 0; j < symbolclocks; j++, k += 2) {
      long long a = Pb[sw[k]], b = Pb[sw[k + 1]], c = Pb[sw[k + 2]];
      sym += -(b - a) + (c - b);
    }
    unsigned long long m = (unsigned long long)(sym < 0 ? -sym : sym);
    if (m >= (1ull << 26)) bad = 1;               // sym^2 >= 2^52: be conservative
    unsigned long long sq = m * m;
    acc += sq;
    if (acc >= TS_LIMIT) bad = 1;
  }
  part[(size_t)blockIdx.y * noff + t] = acc;
  if (bad) atomicOr(inexact, 1u);
}
__global__ __launch_bounds__(256) void k_timesearch_fin(const unsigned long long *__restrict__ part, int nslices, int noff,
                                                        double *__restrict__ energies, unsigned *__restrict__ inexact) {
  int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= noff) return;
  unsigned long long acc = 0; unsigned bad = 0;
  int s = 0;
  for (; s + 8 <= nslices; s += 8) {                  // eight independent loads in flight, then the ordered overflow checks
    unsigned long long v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = part[(size_t)(s + u) * noff + t];
#pragma unroll
    for (int u = 0; u < 8; u++) { acc += v[u]; if (acc >= TS_LIMIT) bad = 1; }
  }
  for (; s < nslices; s++) { acc += part[(size_t)s * noff + t]; if (acc >= TS_LIMIT) bad = 1; }
  energies[t] = (double)acc;
  if (bad) atomicOr(inexact, 1u);
}

// symdemod.c:202-256: one thread per symbol; a single lane then sums sym^2 in order
__global__ __launch_bounds__(256) void k_demod(const long long *__restrict__ P, const int *__restrict__ edges,
                                               int symbolclocks, int nsymbols, double gain,
                                               uint8_t *__restrict__ out, long long *__restrict__ symv) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nsymbols) return;
  long long integ = 0;
  int k = 2 * i * symbolclocks;
  for (int j = 0; j < symbolclocks; j++, k += 2) {
    long long a = P[edges[k]], b = P[edges[k + 1]], c = P[edges[k + 2]];
    integ += -(b - a) + (c - b);
  }
  symv[i] = integ;
  if (gain != 0 && out) {
    double scaled = gain * (double)integ + 128;
    if (scaled > 255) scaled = 255; else if (scaled < 0) scaled = 0;
    out[i] = (unsigned char)scaled;
  }
}
// the reference's own order of additions (symdemod.c:297,325: energy += (double)(sym * sym), symbol by symbol): the squares
// are formed by the whole block (independent loads), one thread then adds them in order out of LDS -- the same values in
// the same order as the one-thread loop this replaces (90 us of dependent global loads at 1 024 symbols)
__global__ __launch_bounds__(256) void k_seq_energy(const long long *__restrict__ symv, int nsymbols, double *energy) {
  __shared__ double sq[1024];
  double e = 0;
  for (int base = 0; base < nsymbols; base += 1024) {
    for (int i = threadIdx.x; i < 1024 && base + i < nsymbols; i += 256) sq[i] = (double)(symv[base + i] * symv[base + i]);
    __syncthreads();
    if (threadIdx.x == 0) for (int i = 0; i < 1024 && base + i < nsymbols; i++) e += sq[i];
    __syncthreads();
  }
  if (threadIdx.x == 0) *energy = e;
}
// exact-integer form of the same sum by one block (see k_timesearch_part); *inexact != 0 => rerun k_seq_energy
__global__ __launch_bounds__(256) void k_par_energy(const long long *__restrict__ symv, int nsymbols, double *energy,
                                                    unsigned *__restrict__ inexact) {
  __shared__ unsigned long long ws[256];
  __shared__ unsigned wbad;
  if (threadIdx.x == 0) wbad = 0;
  __syncthreads();
  unsigned long long acc = 0; unsigned bad = 0;
  for (int i = threadIdx.x; i < nsymbols; i += 256) {
    long long v = symv[i];
    unsigned long long m = (unsigned long long)(v < 0 ? -v : v);
    if (m >= (1ull << 26)) bad = 1;
    acc += m * m;
    if (acc >= TS_LIMIT) bad = 1;
  }
  ws[threadIdx.x] = acc;
  if (bad) atomicOr(&wbad, 1u);
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { ws[threadIdx.x] += ws[threadIdx.x + o]; if (ws[threadIdx.x] >= TS_LIMIT) atomicOr(&wbad, 1u); }
    __syncthreads();
  }
  if (threadIdx.x == 0) { *energy = (double)ws[0]; *inexact = wbad; }
}

extern "C" void *symd_create(int max_samples) {
  Symd *h = (Symd *)calloc(1, sizeof(Symd));
  if (!h) return nullptr;
  { int gd = g_device.load(std::memory_order_acquire); h->dev = gd >= 0 ? gd : 0; }
  h->cap = max_samples > 0 ? max_samples : 1;
  CHK(hipSetDevice(h->dev));
  CHK(dsp_stream_create(&h->st, &h->own_st));
  CHK(hipMalloc(&h->d_s, sizeof(int16_t) * ((size_t)h->cap + 8)));
  CHK(hipMemsetAsync(h->d_s, 0, sizeof(int16_t) * ((size_t)h->cap + 8), h->st));     // the reference's buffer comes from calloc-like use: see store ops
  CHK(hipMalloc(&h->d_P, sizeof(long long) * ((size_t)h->cap + 1)));
  h->nblk_cap = (h->cap + SCAN_BLOCK - 1) / SCAN_BLOCK;
  CHK(hipMalloc(&h->d_blk, sizeof(long long) * (size_t)h->nblk_cap));
  CHK(hipMalloc(&h->d_flag, 2 * sizeof(unsigned)));
  return h;
fail:
  symd_destroy(h);
  return nullptr;
}
extern "C" void symd_destroy(void *p) {
  Symd *h = (Symd *)p;
  if (!h) return;
  (void)hipSetDevice(h->dev);
  (void)hipStreamSynchronize(h->st); if (h->st && h->own_st) (void)hipStreamDestroy(h->st);
  (void)hipFree(h->d_s); (void)hipFree(h->d_s2); (void)hipFree(h->d_P); (void)hipFree(h->d_blk); (void)hipFree(h->d_idx);
  (void)hipFree(h->d_e); (void)hipFree(h->d_out); (void)hipFree(h->d_sym); (void)hipFree(h->d_part); (void)hipFree(h->d_terms);
  (void)hipFree(h->d_flag);
  pin_free(&h->pin_idx); pin_free(&h->pin_e); pin_free(&h->pin_out); pin_free(&h->pin_hdr);
  free(h);
}
// ---- the window buffer kept in HBM (symdemod.c:96-125 without the host copy) ----
extern "C" int symd_store_reset(void *p) {
  Symd *h = (Symd *)p;
  if (!h) return -1;
  CHK(hipSetDevice(h->dev));
  CHK(hipMemsetAsync(h->d_s, 0, sizeof(int16_t) * ((size_t)h->cap + 8), h->st));
  h->n = 0; h->inexact_last = 0;
  return 0;
fail:
  return -1;
}
extern "C" int symd_store_put(void *p, int at, const int16_t *src, int n, int src_is_dev) {
  Symd *h = (Symd *)p;
  if (!h || at < 0 || n < 0 || (long long)at + n > h->cap) {
    snprintf(g_err, sizeof g_err, "symd_store_put: [%d, %d) outside the buffer of %d samples", at, at + n, h ? h->cap : 0);
    return -1;
  }
  if (n == 0) return 0;
  CHK(hipSetDevice(h->dev));
  CHK(hipMemcpyAsync(h->d_s + at, src, sizeof(int16_t) * (size_t)n, src_is_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->st));
  if (src_is_dev) CHK(hipStreamSynchronize(h->st));         // the producer may recycle its block as soon as this returns
  return 0;
fail:
  return -1;
}
extern "C" int symd_store_slide(void *p, int slide, int nsamples) {
  Symd *h = (Symd *)p;
  if (!h || slide < 0 || nsamples < slide || nsamples > h->cap) { snprintf(g_err, sizeof g_err, "symd_store_slide: bad arguments"); return -1; }
  if (slide == 0) return 0;
  CHK(hipSetDevice(h->dev));
  if (nsamples - slide <= slide) {                          // the usual case: a window's worth goes, less than that stays
    int nb = ((nsamples - slide) / 8 + 255) / 256; if (nb > 4096) nb = 4096; if (nb < 1) nb = 1;
    k_slide_inplace<<<nb, 256, 0, h->st>>>(h->d_s, slide, nsamples - slide);
    CHK(hipGetLastError());
    return 0;
  }
  if (!h->d_s2) CHK(hipMalloc(&h->d_s2, sizeof(int16_t) * ((size_t)h->cap + 8)));
  {
    int nb = (h->cap + 255) / 256; if (nb > 4096) nb = 4096;
    k_slide<<<nb, 256, 0, h->st>>>(h->d_s, h->d_s2, slide, nsamples - slide, h->cap);
    CHK(hipGetLastError());
    int16_t *t = h->d_s; h->d_s = h->d_s2; h->d_s2 = t;
  }
  return 0;
fail:
  return -1;
}
extern "C" int symd_store_scan(void *p, int n) {
  Symd *h = (Symd *)p;
  if (!h || n < 0 || n > h->cap) { snprintf(g_err, sizeof g_err, "symd_store_scan: bad size %d (cap %d)", n, h ? h->cap : 0); return -1; }
  CHK(hipSetDevice(h->dev));
  h->n = n;
  if (n > 0) {
    int nblk = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    k_scan_partial<<<nblk, 256, 0, h->st>>>(h->d_s, n, h->d_blk);
    k_scan_final<<<nblk, 256, 0, h->st>>>(h->d_s, n, h->d_blk, h->d_P);
    CHK(hipGetLastError());
  }
  return 0;
fail:
  return -1;
}
extern "C" int symd_load(void *p, const int16_t *samples, int n, int is_dev) {
  Symd *h = (Symd *)p;
  if (!h || n < 0 || n > h->cap) { snprintf(g_err, sizeof g_err, "symd_load: bad size %d (cap %d)", n, h ? h->cap : 0); return -1; }
  if (symd_store_put(p, 0, samples, n, is_dev) != 0) return -1;
  return symd_store_scan(p, n);
}
// the ordered timing search: four waves per 64 offsets where it applies (ISEE3DSP_TS1=1: the one-wave kernel)
static int launch_ordered_search(Symd *h, int lo, const int *d_sw, int symbolclocks, int nsymbols, int noff, double *d_en) {
  static const bool one_wave = getenv("ISEE3DSP_TS1") && atoi(getenv("ISEE3DSP_TS1")) != 0;
  if (symbolclocks == 1 && !one_wave) {
    const size_t lds = sizeof(double) * 4 * TS4_SYMS * 64;
    static std::atomic<unsigned long long> attr_set{0};
    const unsigned long long bit = 1ull << (h->dev & 63);
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
      if (hipFuncSetAttribute((const void *)k_timesearch4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
      attr_set.fetch_or(bit, std::memory_order_release);
    }
    k_timesearch4<<<(noff + 63) / 64, 256, lds, h->st>>>(h->d_P, lo, d_sw, nsymbols, noff, d_en);
  } else
    k_timesearch<<<(noff + 63) / 64, 64, 0, h->st>>>(h->d_P, lo, d_sw, symbolclocks, nsymbols, noff, d_en);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int symd_timesearch(void *p, int lo, const int *sw, int symbolclocks, int nsymbols, int noff,
                               double *energies) {
  Symd *h = (Symd *)p;
  if (!h) return -1;
  {
    int nsw = 2 * symbolclocks * nsymbols + 1;
    if (lo < 0 || noff < 1 || lo + noff - 1 + sw[nsw - 1] > h->n) {
      snprintf(g_err, sizeof g_err, "symd_timesearch: window [%d, %d) outside the %d loaded samples", lo,
               lo + noff - 1 + sw[nsw - 1], h->n);
      return -1;
    }
    CHK(hipSetDevice(h->dev));
    if (grow(&h->d_idx, &h->idx_cap, sizeof(int) * (size_t)nsw) || pin_grow(&h->pin_idx, sizeof(int) * (size_t)nsw) ||
        pin_grow(&h->pin_e, sizeof(double) * (size_t)noff) || pin_grow(&h->pin_hdr, 64)) {
      snprintf(g_err, sizeof g_err, "symd_timesearch: allocation failed");
      return -1;
    }
    memcpy(h->pin_idx.h, sw, sizeof(int) * (size_t)nsw);
    CHK(hipMemcpyAsync(h->d_idx, h->pin_idx.h, sizeof(int) * (size_t)nsw, hipMemcpyHostToDevice, h->st));
    const int nslices = (nsymbols + TS_SLICE - 1) / TS_SLICE;
    volatile unsigned *hflag = (volatile unsigned *)h->pin_hdr.h;
    unsigned flag = 1;
    // the exact-integer parallel form first -- unless the previous window already left its range (it then will again:
    // at 10 MS/s one symbol spans 9 760 samples and the window total passes 2^53)
    if (!getenv("ISEE3DSP_SEQUENTIAL") && !h->inexact_last && grow(&h->d_part, &h->part_cap, sizeof(unsigned long long) * (size_t)nslices * (size_t)noff) == 0) {
      CHK(hipMemsetAsync(h->d_flag, 0, sizeof(unsigned), h->st));
      k_timesearch_part<<<dim3((noff + 255) / 256, nslices), 256, 0, h->st>>>(h->d_P, lo, (const int *)h->d_idx, symbolclocks,
                                                                            nsymbols, noff, (unsigned long long *)h->d_part, h->d_flag);
      k_timesearch_fin<<<(noff + 255) / 256, 256, 0, h->st>>>((const unsigned long long *)h->d_part, nslices, noff,
                                                            (double *)h->pin_e.d, h->d_flag);
      CHK(hipMemcpyAsync(h->pin_hdr.h, h->d_flag, sizeof(unsigned), hipMemcpyDeviceToHost, h->st));
      CHK(hipStreamSynchronize(h->st));
      flag = *hflag;
    }
    // remember it for the next windows, and look again every 32nd (a capture may change level)
    h->inexact_last = flag ? (h->inexact_last % 32) + 1 : 0;
    if (h->inexact_last == 32 || getenv("ISEE3DSP_RETRY_EXACT")) h->inexact_last = 0;
    if (flag) {     // some sum left the exactly-representable range: the reference's own order of additions
      if (launch_ordered_search(h, lo, (const int *)h->d_idx, symbolclocks, nsymbols, noff, (double *)h->pin_e.d) != 0) goto fail;
      CHK(hipGetLastError());
      CHK(hipStreamSynchronize(h->st));
    }
    memcpy(energies, h->pin_e.h, sizeof(double) * (size_t)noff);
  }
  return 0;
fail:
  return -1;
}
extern "C" int symd_demod(void *p, const int *edges, int symbolclocks, int nsymbols, double gain,
                          uint8_t *out, int out_is_dev, double *energy_sum) {
  Symd *h = (Symd *)p;
  if (!h) return -1;
  {
    int ne = 2 * symbolclocks * nsymbols + 1;
    if (edges[0] < 0 || edges[ne - 1] > h->n) {
      snprintf(g_err, sizeof g_err, "symd_demod: edges [%d, %d] outside the %d loaded samples", edges[0], edges[ne - 1], h->n);
      return -1;
    }
    CHK(hipSetDevice(h->dev));
    if (grow(&h->d_idx, &h->idx_cap, sizeof(int) * (size_t)ne) || grow(&h->d_sym, &h->sym_cap, sizeof(long long) * (size_t)nsymbols) ||
        pin_grow(&h->pin_idx, sizeof(int) * (size_t)ne) || pin_grow(&h->pin_out, (size_t)nsymbols) || pin_grow(&h->pin_hdr, 64)) {
      snprintf(g_err, sizeof g_err, "symd_demod: allocation failed");
      return -1;
    }
    memcpy(h->pin_idx.h, edges, sizeof(int) * (size_t)ne);
    CHK(hipMemcpyAsync(h->d_idx, h->pin_idx.h, sizeof(int) * (size_t)ne, hipMemcpyHostToDevice, h->st));
    // symbols, energy sum and its flag are stored by the kernels straight into mapped host memory
    uint8_t *dst = (out && out_is_dev) ? out : (uint8_t *)h->pin_out.d;
    double *d_esum = (double *)((char *)h->pin_hdr.d + 16);
    unsigned *d_eflag = (unsigned *)((char *)h->pin_hdr.d + 8);
    k_demod<<<(nsymbols + 255) / 256, 256, 0, h->st>>>(h->d_P, (const int *)h->d_idx, symbolclocks, nsymbols, gain,
                                                       (gain != 0 && out) ? dst : nullptr, (long long *)h->d_sym);
    if (energy_sum) k_par_energy<<<1, 256, 0, h->st>>>((const long long *)h->d_sym, nsymbols, d_esum, d_eflag);
    CHK(hipGetLastError());
    CHK(hipStreamSynchronize(h->st));
    if (energy_sum) {
      if (*(volatile unsigned *)((char *)h->pin_hdr.h + 8) || getenv("ISEE3DSP_SEQUENTIAL")) {
        k_seq_energy<<<1, 256, 0, h->st>>>((const long long *)h->d_sym, nsymbols, d_esum);
        CHK(hipStreamSynchronize(h->st));
      }
      *energy_sum = *(volatile double *)((char *)h->pin_hdr.h + 16);
    }
    if (gain != 0 && out && !out_is_dev) memcpy(out, h->pin_out.h, (size_t)nsymbols);
  }
  return 0;
fail:
  return -1;
}

// ---- one whole window in one go (symd_window) ------------------------------------------------------------------------
// symdemod.c:127-131,190-192 without -t: timing search, its first maximum, final demodulation with gain 100 / sqrt(maxenergy).
// Taken step by step (symd_timesearch, the host's arg-max and boundary recurrence, symd_demod) a window costs two host round
// trips; at 250 kS/s those, not the kernels, are its 160 us, and the Viterbi stage waits for the symbols.  Here the host
// SPECULATES: it hands over the boundary tables of trial_demod for a few timing adjustments around zero (each table is the
// reference's own FP recurrence started at firstsample + adjustment, computed while the GPU scans), the arg-max stays on
// the device, the demodulation picks the table of the adjustment that won -- one synchronisation.  An adjustment
// outside the tables, a gain the device rounds differently from the host, or sums outside the exact range make the call
// return 1 and change nothing: the caller then takes the step-by-step calls for that window.
struct WinHdr { int bi; int miss; double best; double gain; unsigned inexact; unsigned pad; };
// the first maximum of the offsets' energies (symdemod.c:326-331: strict '>'); every thread of a 256-thread workgroup calls it
// and gets the result.  k_window_demod's few workgroups each do it for themselves: as a kernel of its own (k_ts_argmax, one
// workgroup) it was 18 us per window at 10 MS/s, most of it the launch.
__device__ __forceinline__ void ts_argmax(const double *__restrict__ en, int noff, int *out_bi, double *out_best) {
  __shared__ double wv[256]; __shared__ int wi[256];
  double be = 0; int bi = -1;
  for (int t = threadIdx.x; t < noff; t += 256) {                    // ascending t per thread: strict '>' keeps the first maximum
    const double e = en[t];
    if (bi < 0 || e > be) { be = e; bi = t; }
  }
  wv[threadIdx.x] = be; wi[threadIdx.x] = bi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      const double oe = wv[threadIdx.x + o]; const int oi = wi[threadIdx.x + o];
      const double me = wv[threadIdx.x]; const int mi = wi[threadIdx.x];
      if (oi >= 0 && (mi < 0 || oe > me || (oe == me && oi < mi))) { wv[threadIdx.x] = oe; wi[threadIdx.x] = oi; }
    }
    __syncthreads();
  }
  *out_bi = wi[0]; *out_best = wv[0];
}
__global__ __launch_bounds__(256) void k_window_demod(const long long *__restrict__ P, const int *__restrict__ tables, int ne,
                                                      int first_off, int spec_lo, int nspec, const unsigned char *__restrict__ table_ok,
                                                      int symbolclocks, int nsymbols, const double *__restrict__ en, int noff,
                                                      const unsigned *__restrict__ inexact, uint8_t *__restrict__ out,
                                                      WinHdr *__restrict__ host_hdr) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  int bi; double best;
  ts_argmax(en, noff, &bi, &best);
  const int j = bi + first_off - spec_lo;
  const double maxenergy = best / (double)nsymbols;                  // symdemod.c:333
  const double gain = 100. / sqrt(maxenergy);                        // :190
  const bool miss = bi < 0 || j < 0 || j >= nspec || !table_ok[j < 0 || j >= nspec ? 0 : j];
  if (i == 0) { host_hdr->bi = bi; host_hdr->best = best; host_hdr->gain = gain; host_hdr->miss = miss ? 1 : 0;
                host_hdr->inexact = inexact ? *inexact : 0u; }
  if (miss || i >= nsymbols) return;
  const int *__restrict__ edges = tables + (size_t)j * ne;
  long long integ = 0;
  int k = 2 * i * symbolclocks;
  for (int c = 0; c < symbolclocks; c++, k += 2) {
    const long long a = P[edges[k]], b = P[edges[k + 1]], d = P[edges[k + 2]];
    integ += -(b - a) + (d - b);
  }
  double scaled = gain * (double)integ + 128;                        // :240-251 (not fused: -ffp-contract=off)
  if (scaled > 255) scaled = 255; else if (scaled < 0) scaled = 0;
  out[i] = (unsigned char)scaled;
}
extern "C" int symd_window(void *p, int firstsample, const int *sw, int symbolclocks, int nsymbols, int first_off, int noff,
                           const int *edges, int spec_lo, int nspec, uint8_t *out, int *symphase, double *maxenergy) {
  Symd *h = (Symd *)p;
  if (!h || !sw || !edges || !out || nspec < 1 || nspec > 16 || noff < 1 || nsymbols < 1) {
    snprintf(g_err, sizeof g_err, "symd_window: bad argument");
    return -1;
  }
  {
    const int nsw = 2 * symbolclocks * nsymbols + 1, lo = firstsample + first_off;
    if (lo < 0 || lo + noff - 1 + sw[nsw - 1] > h->n) return 1;     // the step-by-step call reports it
    CHK(hipSetDevice(h->dev));
    const size_t tab_ints = (size_t)nsw * (size_t)(1 + nspec);
    if (grow(&h->d_idx, &h->idx_cap, sizeof(int) * tab_ints + 64) || pin_grow(&h->pin_idx, sizeof(int) * tab_ints + 64) ||
        grow(&h->d_e, &h->e_cap, sizeof(double) * (size_t)noff + sizeof(WinHdr)) || pin_grow(&h->pin_out, (size_t)nsymbols) ||
        pin_grow(&h->pin_hdr, 256)) {
      snprintf(g_err, sizeof g_err, "symd_window: allocation failed");
      return -1;
    }
    // staging: [sw | nspec edge tables | nspec validity bytes]
    int *st = (int *)h->pin_idx.h;
    memcpy(st, sw, sizeof(int) * (size_t)nsw);
    memcpy(st + nsw, edges, sizeof(int) * (size_t)nsw * (size_t)nspec);
    unsigned char *ok = (unsigned char *)(st + tab_ints);
    int any = 0;
    for (int j = 0; j < nspec; j++) { const int *e = edges + (size_t)j * nsw; ok[j] = e[0] >= 0 && e[nsw - 1] <= h->n; any |= ok[j]; }
    if (!any) return 1;
    CHK(hipMemcpyAsync(h->d_idx, st, sizeof(int) * tab_ints + 64, hipMemcpyHostToDevice, h->st));
    const int *d_sw = (const int *)h->d_idx, *d_tab = d_sw + nsw;
    const unsigned char *d_ok = (const unsigned char *)(d_sw + tab_ints);
    double *d_en = (double *)h->d_e;
    WinHdr *host_hdr_d = (WinHdr *)((char *)h->pin_hdr.d + 64);
    volatile WinHdr *host_hdr = (volatile WinHdr *)((char *)h->pin_hdr.h + 64);
    const int nslices = (nsymbols + TS_SLICE - 1) / TS_SLICE;
    const bool exact_form = !getenv("ISEE3DSP_SEQUENTIAL") && !h->inexact_last &&
                            grow(&h->d_part, &h->part_cap, sizeof(unsigned long long) * (size_t)nslices * (size_t)noff) == 0;
    if (exact_form) {
      CHK(hipMemsetAsync(h->d_flag, 0, sizeof(unsigned), h->st));
      k_timesearch_part<<<dim3((noff + 255) / 256, nslices), 256, 0, h->st>>>(h->d_P, lo, d_sw, symbolclocks, nsymbols, noff,
                                                                            (unsigned long long *)h->d_part, h->d_flag);
      k_timesearch_fin<<<(noff + 255) / 256, 256, 0, h->st>>>((const unsigned long long *)h->d_part, nslices, noff, d_en, h->d_flag);
    } else if (launch_ordered_search(h, lo, d_sw, symbolclocks, nsymbols, noff, d_en) != 0) goto fail;
    k_window_demod<<<(nsymbols + 255) / 256, 256, 0, h->st>>>(h->d_P, d_tab, nsw, first_off, spec_lo, nspec, d_ok, symbolclocks, nsymbols,
                                                              d_en, noff, exact_form ? h->d_flag : nullptr, (uint8_t *)h->pin_out.d, host_hdr_d);
    CHK(hipGetLastError());
    CHK(hipStreamSynchronize(h->st));
    const unsigned flag = exact_form ? host_hdr->inexact : 1u;
    // the same bookkeeping as symd_timesearch: remember a window that left the exact range, look again every 32nd
    h->inexact_last = flag ? (h->inexact_last % 32) + 1 : 0;
    if (h->inexact_last == 32 || getenv("ISEE3DSP_RETRY_EXACT")) h->inexact_last = 0;
    if (exact_form && flag) return 1;                               // the ordered kernel has to redo the search
    if (host_hdr->miss || host_hdr->bi < 0) return 1;
    const double best = host_hdr->best, me = best / nsymbols, gain = 100. / sqrt(me);
    if (memcmp(&gain, (const void *)&host_hdr->gain, sizeof gain) != 0) return 1;   // the device rounded the gain differently: do not trust the bytes
    *symphase = first_off + host_hdr->bi;
    *maxenergy = me;
    memcpy(out, h->pin_out.h, (size_t)nsymbols);
  }
  return 0;
fail:
  return -1;
}

// ===========================================================================================
// pmdemod
// ===========================================================================================
#define PIN_DFT 8192       /* 3 x 768 double2 */
#define PIN_ROT 49152      /* RED_BLOCKS double2 */
#define PIN_BYTES 65536
struct Pmd {
  int dev; hipStream_t st; int own_st;
  int N, logN;
  double2 *buf, *spec, *tmp, *tw, *lo;   // buf / tw: only for N < 2^12 (register-radix path); lo = optional de-chirp table
  double2 *twA, *twB, *twR;              // LDS-pass path: W_N^(4096 h), W_N^l (l < 4096), W_256^m
  int16_t *d_iq;                         // staging for host blocks
  const int16_t *cur_iq; int cur_flip;   // the block pmd_load announced (device memory): read by the FFT's first pass,
                                         // by the spin-down sum and by the output kernel -- no double-precision copy of it exists
  void *d_red; size_t red_cap;           // reduction scratch
  hipEvent_t ev_peak, ev_mix;            // behind the transform + peak kernels / behind the two spin-down passes (the *_begin / *_end calls)
  void *d_peakpart; size_t peakpart_cap; // one peak record per workgroup of the last FFT pass
  int dft_nb, mix_nb;                    // workgroups of the k_dft_bins / k_rotate4 launch in flight (their partial sums are in pin_hdr)
  int ask_first, ask_last;               // the bin range of the peak search in flight (a fall-back repeats it)
  int spec_valid;                        // spec holds the double transform of the current block (pmd_get_spectrum)
  int last_path;                         // pmd_last_peak_path
  int16_t *d_out16; double *d_pre;
  int have_lo;
  Pin pin_hdr;                           // written by the kernels: peak record @0, spin-down sum @128, variance sum @144 (per-sample
                                         // path), DftStatus @256, k_dft_bins' partial sums @PIN_DFT, k_rotate4's @PIN_ROT
};

__global__ __launch_bounds__(256) void k_twiddles(double2 *tw, int N) {
  int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= N / 2) return;
  double s, c;
  sincospi(-2.0 * (double)k / (double)N, &s, &c);
  tw[k] = make_double2(c, s);
}

// pmdemod.c:209-229 (+ :237-243 when a de-chirp table is set)
__global__ __launch_bounds__(256) void k_pmd_load(const short2 *__restrict__ iq, double2 *__restrict__ buf,
                                                  const double2 *__restrict__ lo, int N, int flip) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  short2 v = iq[i];
  double x = flip ? (double)v.y : (double)v.x, y = flip ? (double)v.x : (double)v.y;
  if (lo) {                                // buffer[i] *= conj(lophase)
    double pr = lo[i].x, pi = -lo[i].y;
    double nx = x * pr - y * pi, ny = x * pi + y * pr;
    x = nx; y = ny;
  }
  buf[i] = make_double2(x, y);
}

// one Stockham radix-2 stage: reads unit-stride, writes runs of s
__global__ __launch_bounds__(256) void k_fft_stage(const double2 *__restrict__ x, double2 *__restrict__ y,
                                                   const double2 *__restrict__ tw, int N, int s) {
  int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= N / 2) return;
  int q = t & (s - 1), ps = t - q;          // p * s
  double2 a = x[t], b = x[t + N / 2], w = tw[ps];
  double dr = a.x - b.x, di = a.y - b.y;
  y[q + 2 * ps] = make_double2(a.x + b.x, a.y + b.y);
  y[q + 2 * ps + s] = make_double2(dr * w.x - di * w.y, dr * w.y + di * w.x);
}

// Radix-R Stockham stage, R = 4, 8, 16: thread t loads x[t + j*N/R] (unit stride across lanes), does the
// R-point DFT in registers (radix-2 DIF network with constant twiddles, bit-reversed read-out), applies the
// stage twiddle W_N^(p*s*k) from the half-circle table and stores y[q + R*p*s + k*s].  A 2^23-point
// transform takes 6 launches instead of 23, a 2^18-point one 5 instead of 18.
template <int R> struct FftConst;
template <> struct FftConst<2>  { static constexpr int LG = 1; };
template <> struct FftConst<4>  { static constexpr int LG = 2; };
template <> struct FftConst<8>  { static constexpr int LG = 3; };
template <> struct FftConst<16> { static constexpr int LG = 4; };
// cos / sin of m*pi/8, m = 0..7 (omega_16^m = c - j s)
__device__ __constant__ double k_c16[8] = { 1.0, 0.92387953251128675613, 0.70710678118654752440, 0.38268343236508977173,
                                            0.0, -0.38268343236508977173, -0.70710678118654752440, -0.92387953251128675613 };
__device__ __constant__ double k_s16[8] = { 0.0, 0.38268343236508977173, 0.70710678118654752440, 0.92387953251128675613,
                                            1.0, 0.92387953251128675613, 0.70710678118654752440, 0.38268343236508977173 };
__host__ __device__ constexpr int brev(int v, int bits) { int r = 0; for (int i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i); return r; }

template <int R>
__global__ __launch_bounds__(256) void k_fft_radix(const double2 *__restrict__ x, double2 *__restrict__ y,
                                                   const double2 *__restrict__ tw, int N, int s) {
  constexpr int LG = FftConst<R>::LG;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= N / R) return;
  const int q = t & (s - 1), ps = t - q;
  double2 a[R];
#pragma unroll
  for (int j = 0; j < R; j++) a[j] = x[t + j * (N / R)];
  // radix-2 DIF: span R/2, R/4, ... 1; twiddle omega_R^(m * R/(2*span)) for position m inside a span
#pragma unroll
  for (int span = R / 2; span >= 1; span >>= 1) {
#pragma unroll
    for (int base = 0; base < R; base += 2 * span) {
#pragma unroll
      for (int m = 0; m < span; m++) {
        const double2 u = a[base + m], v = a[base + m + span];
        a[base + m] = make_double2(u.x + v.x, u.y + v.y);
        const double dr = u.x - v.x, di = u.y - v.y;
        const int e = m * (8 / span);                    // exponent of omega_16 (multiples of 16/(2*span))
        if (e == 0) a[base + m + span] = make_double2(dr, di);
        else if (e == 4) a[base + m + span] = make_double2(di, -dr);          // * (-j)
        else { const double c = k_c16[e], sn = k_s16[e];
               a[base + m + span] = make_double2(dr * c + di * sn, di * c - dr * sn); }   // * (c - j sn)
      }
    }
  }
  // a[brev(k)] = b_k
#pragma unroll
  for (int k = 0; k < R; k++) {
    const double2 b = a[brev(k, LG)];
    double2 w;
    if (k == 0) w = make_double2(1.0, 0.0);
    else {
      const int idx = ps * k;                              // < N
      if (idx < N / 2) w = tw[idx];
      else { const double2 h = tw[idx - N / 2]; w = make_double2(-h.x, -h.y); }
    }
    y[q + R * ps + k * s] = make_double2(b.x * w.x - b.y * w.y, b.x * w.y + b.y * w.x);
  }
}

// ---- LDS-staged passes (N >= 2^12) ---------------------------------------------------------------------------------
// One launch = one Stockham stage of radix R = R1 * R2 (32 .. 256), i.e. 5 .. 8 of the log2 N butterfly levels per trip
// through HBM: 2^23 points take 3 launches (8 + 8 + 7 levels, 3 x 256 MiB of traffic) instead of 6, 2^18 points 3 (6 + 6 + 6),
// 2^16 points 2.  A workgroup owns a tile of FT = 16 consecutive columns t (thread index = column + 16 * row part), so
// every global access is a run of 16 consecutive double2 = 256 bytes:
//   step 1  thread (c, a): loads x[t + (a + R2 b) N/R], b < R1, does the R1-point DFT over b in registers, multiplies by
//           W_R^(a k1) and parks the results in LDS as Z[a][k1][c];
//   step 2  thread (c, k1): pulls Z[.][k1][c], does the R2-point DFT over a: X[k1 + R1 k2]; multiplies by the stage
//           twiddle W_N^(p s k) (two-level table: W_N^(4096 h) * W_N^l) and stores y[q + R p s + k s].
// In the first stage (s = 1) a tile's output is ONE contiguous block of 16 R elements; later stages write 256-byte runs.
// The first stage reads the int16 (I, Q) pairs themselves (pmdemod.c:209-229, de-chirp :237-243 included): the block is
// never expanded to doubles in memory.
#define FT 16
// the passes exist in two element types: double2 (every transform whose values are used) and float2 (pmdemod's SEARCH
// transform, which only has to find the bins worth evaluating exactly: pmd_fft_peak_begin)
template <typename V> struct VecOf;
template <> struct VecOf<double2> { using T = double; };
template <> struct VecOf<float2>  { using T = float; };
template <typename V> __device__ __forceinline__ V mkv(typename VecOf<V>::T x, typename VecOf<V>::T y) { V r; r.x = x; r.y = y; return r; }
template <typename V> __device__ __forceinline__ V vec_as(const double2 w) { return mkv<V>((typename VecOf<V>::T)w.x, (typename VecOf<V>::T)w.y); }
template <typename V> __device__ __forceinline__ double2 as_d2(const V w) { return make_double2((double)w.x, (double)w.y); }
template <int R, typename V> __device__ __forceinline__ void dft_regs(V (&a)[R]) {     // radix-2 DIF network; a[brev(k)] = X[k]
  using T = typename VecOf<V>::T;
#pragma unroll
  for (int span = R / 2; span >= 1; span >>= 1) {
#pragma unroll
    for (int base = 0; base < R; base += 2 * span) {
#pragma unroll
      for (int m = 0; m < span; m++) {
        const V u = a[base + m], v = a[base + m + span];
        a[base + m] = mkv<V>(u.x + v.x, u.y + v.y);
        const T dr = u.x - v.x, di = u.y - v.y;
        const int e = m * (8 / span);                    // exponent of omega_16
        if (e == 0) a[base + m + span] = mkv<V>(dr, di);
        else if (e == 4) a[base + m + span] = mkv<V>(di, -dr);          // * (-j)
        else { const T c = (T)k_c16[e], sn = (T)k_s16[e];
               a[base + m + span] = mkv<V>(dr * c + di * sn, di * c - dr * sn); }   // * (c - j sn)
      }
    }
  }
}
__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// sample i of the block as pmdemod.c:209-229 forms it (+ :237-243 with a de-chirp table)
__device__ __forceinline__ double2 iq_sample(const short2 *__restrict__ iq, const double2 *__restrict__ lo, int i, int flip) {
  const short2 v = iq[i];
  double x = flip ? (double)v.y : (double)v.x, y = flip ? (double)v.x : (double)v.y;
  if (lo) {                                // buffer[i] *= conj(lophase)
    const double pr = lo[i].x, pi = -lo[i].y;
    const double nx = x * pr - y * pi, ny = x * pi + y * pr;
    x = nx; y = ny;
  }
  return make_double2(x, y);
}
template <int LG> struct PassShape;
template <> struct PassShape<5> { static constexpr int R1 = 8,  R2 = 4; };
template <> struct PassShape<6> { static constexpr int R1 = 8,  R2 = 8; };
template <> struct PassShape<7> { static constexpr int R1 = 16, R2 = 8; };
template <> struct PassShape<8> { static constexpr int R1 = 16, R2 = 16; };
constexpr int lg2c(int v) { int r = 0; while ((1 << r) < v) r++; return r; }

// SRC: where the first stage's input comes from -- 0 a double2 array, 1 int16 (I, Q) pairs (pmdemod), 2 int16 REAL samples
// zero-padded beyond nvalid (icesync.c:151-163), 3 conj(x[i] * v[i]) (the inverse transform of a product: icesync.c:166-170)
#define SRC_C2 0
#define SRC_IQ 1
#define SRC_REAL16 2
#define SRC_CONJPROD 3
struct PeakRec { double e; int idx; int near; };    // near: only written by the search transform's PEAK pass
template <typename T> __device__ __forceinline__ bool peak_better(T e, int i, T be, int bi) {
  return e > be || (e == be && i > bi);     // ">=" while scanning upward == last maximum wins
}
// PEAK (the last pass of pmdemod's transform): the |X|^2 arg-max of pmdemod.c:255-279 over bins [pfirst, plast) rides on
// the pass that produces the bins -- one PeakRec per workgroup -- instead of reading the 16 N bytes of spectrum again
// FTC: columns per workgroup (FT = 16 everywhere except where the first pass reads 2- or 4-byte samples: there 32 columns
// make the read runs 128 bytes -- a whole cache line per row and tile -- instead of 64)
// V = float2 (the search transform): a PEAK pass stores NO spectrum -- its per-workgroup record also says how many of the
// workgroup's bins lie within PEAK_NEAR of the workgroup's maximum (PeakRec::near), which is what search_peak needs to know
// that the records hold EVERY bin near the global maximum.
#define PEAK_NEAR 9.765625e-4      // 2^-10: relative distance in |X|^2 inside which single precision may have swapped two bins
                                   // (its error against the largest bin is ~1e-6: three orders of margin)
template <int LG, int SRC, bool FIRST, bool PEAK = false, int FTC = FT, typename V = double2>
__global__ __launch_bounds__(FTC * (PassShape<LG>::R1 > PassShape<LG>::R2 ? PassShape<LG>::R1 : PassShape<LG>::R2))
void k_fft_pass(const V *__restrict__ x, const short2 *__restrict__ iq, const double2 *__restrict__ lo, int flip,
                V *__restrict__ y, const double2 *__restrict__ twA, const double2 *__restrict__ twB,
                const double2 *__restrict__ twR, int N, int s, int pfirst = 0, int plast = 0, PeakRec *__restrict__ ppart = nullptr) {
  using T = typename VecOf<V>::T;
  constexpr bool SEARCH = std::is_same<V, float2>::value;
  constexpr int R1 = PassShape<LG>::R1, R2 = PassShape<LG>::R2, R = R1 * R2;
  extern __shared__ double2 Zraw[];                       // [R2][R1][FTC]
  V *Z = reinterpret_cast<V *>(Zraw);
  const int c = threadIdx.x & (FTC - 1), r = threadIdx.x / FTC;
  const int t = blockIdx.x * FTC + c, stride = N / R;
  if (r < R2) {                                           // ---- step 1, thread (c, a = r)
    V v[R1];
    if constexpr (SRC == SRC_IQ) {
      // all R1 loads first, then the conversions: written per sample (iq_sample: load, test `lo`, convert) the compiler
      // waited for every load before issuing the next -- sixteen dependent memory round trips, 45 of the pass's 74 us
      short2 raw[R1];
#pragma unroll
      for (int b = 0; b < R1; b++) raw[b] = iq[t + (r + R2 * b) * stride];
      if (lo) {                                             // buffer[i] *= conj(lophase), pmdemod.c:237-243
        double2 l[R1];
#pragma unroll
        for (int b = 0; b < R1; b++) l[b] = lo[t + (r + R2 * b) * stride];
#pragma unroll
        for (int b = 0; b < R1; b++) {
          const double x = flip ? (double)raw[b].y : (double)raw[b].x, y = flip ? (double)raw[b].x : (double)raw[b].y;
          const double pr = l[b].x, pi = -l[b].y;
          v[b] = mkv<V>((T)(x * pr - y * pi), (T)(x * pi + y * pr));
        }
      } else {
#pragma unroll
        for (int b = 0; b < R1; b++)
          v[b] = mkv<V>(flip ? (T)raw[b].y : (T)raw[b].x, flip ? (T)raw[b].x : (T)raw[b].y);
      }
    } else if constexpr (SRC == SRC_CONJPROD) {             // conj(x[i] * v[i]); both operand sets loaded before the first product
      double2 xa[R1], va[R1];
#pragma unroll
      for (int b = 0; b < R1; b++) { xa[b] = as_d2(x[t + (r + R2 * b) * stride]); va[b] = lo[t + (r + R2 * b) * stride]; }
#pragma unroll
      for (int b = 0; b < R1; b++) { const double2 p = cmul(xa[b], va[b]); v[b] = mkv<V>((T)p.x, (T)-p.y); }
    } else if constexpr (SRC == SRC_REAL16) {               // int16 real samples, zero beyond flip = nvalid (icesync.c:151-163)
      int16_t ra[R1];
#pragma unroll
      for (int b = 0; b < R1; b++) {
        const int i = t + (r + R2 * b) * stride;
        ra[b] = reinterpret_cast<const int16_t *>(iq)[i < flip ? i : 0];       // always a load (no branch), masked below
      }
#pragma unroll
      for (int b = 0; b < R1; b++) v[b] = mkv<V>(t + (r + R2 * b) * stride < flip ? (T)ra[b] : (T)0, (T)0);
    } else
#pragma unroll
    for (int b = 0; b < R1; b++) {
      const int i = t + (r + R2 * b) * stride;
      if constexpr (SRC == SRC_IQ) v[b] = vec_as<V>(iq_sample(iq, lo, i, flip));
      else if constexpr (SRC == SRC_REAL16) v[b] = mkv<V>(i < flip ? (T)reinterpret_cast<const int16_t *>(iq)[i] : (T)0, (T)0);   // flip = nvalid
      else if constexpr (SRC == SRC_CONJPROD) { const double2 p = cmul(as_d2(x[i]), lo[i]); v[b] = mkv<V>((T)p.x, (T)-p.y); }
      else v[b] = x[i];
    }
    dft_regs<R1>(v);
#pragma unroll
    for (int k1 = 0; k1 < R1; k1++) {
      V val = v[brev(k1, lg2c(R1))];
      if (k1 > 0) val = cmul(val, vec_as<V>(twR[(r * k1) * (256 / R)]));      // W_R^(a k1); a = 0 reads W^0 = 1 exactly
      Z[(r * R1 + k1) * FTC + c] = val;
    }
  }
  __syncthreads();
  T be = (T)-1; int bi = -1;
  T ef[(PEAK && SEARCH) ? R2 : 1];                        // search transform: the energies of this thread's bins (-1: outside [pfirst, plast))
  if constexpr (PEAK && SEARCH) {
#pragma unroll
    for (int k2 = 0; k2 < R2; k2++) ef[k2] = (T)-1;
  }
  V u_first[FIRST ? R2 : 1];                              // a first stage's results (see the transposed store below)
  if (r < R1) {                                           // ---- step 2, thread (c, k1 = r)
    V u[R2];
#pragma unroll
    for (int a = 0; a < R2; a++) u[a] = Z[(a * R1 + r) * FTC + c];
    dft_regs<R2>(u);
    const int q = t & (s - 1), ps = t - q;
    V *__restrict__ out = y + q + (size_t)R * ps;
    // stage twiddle W_N^(ps k), k = k1 + R1 k2 (ps k < N), from the two-level table W_N^(4096 h) * W_N^l.  In the first
    // stage ps = t differs from lane to lane and every lookup is a 64-address gather: there the thread looks up only
    // W^(ps k1) and the step W^(ps R1) and walks k2 by multiplication (<= 15 products: ~1e-15 relative); later stages
    // have ONE ps per tile, their lookups are broadcasts and stay direct.
    // W_N^idx = twB[idx & 4095] (* twA[idx >> 12] when that is not W^0).  In the first stage both table words of both
    // lookups are loaded before the first use (four loads in flight instead of four dependent round trips).
    auto tw2_sel = [&](unsigned idx, double2 wb, double2 wa) { return vec_as<V>((idx >> 12) ? cmul(wa, wb) : wb); };
    if constexpr (FIRST) {
      V wk = mkv<V>((T)1, (T)0), wstep = wk;                // first stage <=> s == 1
      {
        const unsigned i1 = (unsigned)ps * (unsigned)r, i2 = (unsigned)ps * (unsigned)R1;
        const double2 b1 = twB[i1 & 4095u], a1 = twA[i1 >> 12], b2 = twB[i2 & 4095u], a2 = twA[i2 >> 12];
        if (ps != 0) { wk = tw2_sel(i1, b1, a1); wstep = tw2_sel(i2, b2, a2); }
      }
#pragma unroll
      for (int k2 = 0; k2 < R2; k2++) {
        const int k = r + R1 * k2;
        V val = u[brev(k2, lg2c(R2))];
        if (ps != 0 && k != 0) val = cmul(val, wk);
        wk = cmul(wk, wstep);
        u_first[k2] = val;                                  // parked: leaves through the LDS transpose below
      }
    } else {
      // later stages: ONE ps per tile.  W^(ps k), k = k1 + R1 k2, is walked like the first stage's: two table lookups
      // (W^(ps k1), W^(ps R1)) and R2 products instead of 2 R2 dependent lookups (the compiler waited for each one: up to
      // 32 L2 round trips per thread); <= 15 products deep, ~1e-15 relative (measured against the value-by-value lookups:
      // profiles/r03x_*).
      V wk = mkv<V>((T)1, (T)0), wstep = wk;
      {
        const unsigned i1 = (unsigned)ps * (unsigned)r, i2 = (unsigned)ps * (unsigned)R1;
        const double2 b1 = twB[i1 & 4095u], a1 = twA[i1 >> 12], b2 = twB[i2 & 4095u], a2 = twA[i2 >> 12];
        if (ps != 0) { wk = tw2_sel(i1, b1, a1); wstep = tw2_sel(i2, b2, a2); }
      }
#pragma unroll
      for (int k2 = 0; k2 < R2; k2++) {
        const int k = r + R1 * k2;
        V val = u[brev(k2, lg2c(R2))];
        if (ps != 0 && k != 0) val = cmul(val, wk);
        wk = cmul(wk, wstep);
        if constexpr (!(PEAK && SEARCH)) out[(size_t)k * s] = val;
        if constexpr (PEAK) {
          const int i = q + R * ps + k * s;                 // the bin this value is (the last pass: N fits an int)
          const T e = val.x * val.x + val.y * val.y;
          if (i >= pfirst && i < plast) {
            if constexpr (SEARCH) ef[k2] = e;
            if (peak_better(e, i, be, bi)) { be = e; bi = i; }
          }
        }
      }
    }
  }
  if constexpr (FIRST) {
    // First stage (s = 1): the tile's output is ONE contiguous block of FTC * R elements, y[R t + k] -- but thread (c, k1)
    // holds k = k1 + R1 k2 of column c, so a store instruction straight from registers would scatter 32- or 64-byte pieces
    // over FTC columns R * 16 bytes apart (measured: the pass ran at 1.5 TB/s, bound by exactly that).  The values go
    // through LDS once more (column-major rows of R + 1 elements: no bank conflict either way) and leave as 1 KiB per wave
    // instruction.  (PEAK never rides on a first pass: fft_forward.)
    constexpr int TH = FTC * (R1 > R2 ? R1 : R2);
    __syncthreads();                                        // every thread has pulled its part of Z
    if (r < R1) {
#pragma unroll
      for (int k2 = 0; k2 < R2; k2++) Z[c * (R + 1) + r + R1 * k2] = u_first[k2];
    }
    __syncthreads();
    V *__restrict__ blk = y + (size_t)R * ((size_t)blockIdx.x * FTC);
#pragma unroll 4
    for (int idx = threadIdx.x; idx < FTC * R; idx += TH) {
      const V o = Z[(idx / R) * (R + 1) + (idx % R)];
      blk[idx] = o;
    }
  }
  if constexpr (PEAK) {
    constexpr int TH = FTC * (R1 > R2 ? R1 : R2), NW = (TH + 63) / 64;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const T oe = __shfl_xor(be, o, 64); const int oi = __shfl_xor(bi, o, 64);
      if (peak_better(oe, oi, be, bi)) { be = oe; bi = oi; }
    }
    __syncthreads();                                        // every thread has read its part of Z
    PeakRec *ws = reinterpret_cast<PeakRec *>(Zraw);
    if ((threadIdx.x & 63) == 0) { ws[threadIdx.x >> 6].e = (double)be; ws[threadIdx.x >> 6].idx = bi; }
    __syncthreads();
    int near = 0;
    if constexpr (SEARCH) {
      // how many of the workgroup's bins lie within PEAK_NEAR of ITS maximum (the maximum itself included)
      double wm = ws[0].e;
      for (int w = 1; w < NW; w++) wm = ws[w].e > wm ? ws[w].e : wm;
      const T lim = (T)(wm * (1.0 - PEAK_NEAR));
#pragma unroll
      for (int k2 = 0; k2 < R2; k2++) near += (ef[k2] >= (T)0 && ef[k2] >= lim) ? 1 : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) near += __shfl_xor(near, o, 64);
      int *wn = reinterpret_cast<int *>(ws + NW);
      if ((threadIdx.x & 63) == 0) wn[threadIdx.x >> 6] = near;
      __syncthreads();
      near = 0;
      for (int w = 0; w < NW; w++) near += wn[w];
    }
    if (threadIdx.x == 0) {
      double fe = ws[0].e; int fi = ws[0].idx;
      for (int w = 1; w < NW; w++) if (peak_better(ws[w].e, ws[w].idx, fe, fi)) { fe = ws[w].e; fi = ws[w].idx; }
      ppart[blockIdx.x].e = fe; ppart[blockIdx.x].idx = fi; ppart[blockIdx.x].near = near;
    }
  }
}
// W_N^(4096 h) for h < N / 4096, W_N^l for l < min(N, 4096), W_256^m
__global__ __launch_bounds__(256) void k_twiddles2(double2 *twA, double2 *twB, double2 *twR, int N) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  double sn, cs;
  if (i < N / 4096) { sincospi(-2.0 * (double)i * 4096.0 / (double)N, &sn, &cs); twA[i] = make_double2(cs, sn); }
  if (i < 4096 && i < N) { sincospi(-2.0 * (double)i / (double)N, &sn, &cs); twB[i] = make_double2(cs, sn); }
  if (i < 256) { sincospi(-2.0 * (double)i / 256.0, &sn, &cs); twR[i] = make_double2(cs, sn); }
}

__global__ __launch_bounds__(256) void k_peak_partial(const double2 *__restrict__ spec, int first, int last,
                                                      PeakRec *__restrict__ part) {
  double be = -1.0; int bi = -1;
  for (int i = first + blockIdx.x * 256 + threadIdx.x; i < last; i += gridDim.x * 256) {
    double2 v = spec[i];
    double e = v.x * v.x + v.y * v.y;
    if (peak_better(e, i, be, bi)) { be = e; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    double oe = __shfl_xor(be, o, 64); int oi = __shfl_xor(bi, o, 64);
    if (peak_better(oe, oi, be, bi)) { be = oe; bi = oi; }
  }
  __shared__ PeakRec ws[4];
  if ((threadIdx.x & 63) == 0) { ws[threadIdx.x >> 6].e = be; ws[threadIdx.x >> 6].idx = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) if (peak_better(ws[w].e, ws[w].idx, be, bi)) { be = ws[w].e; bi = ws[w].idx; }
    part[blockIdx.x].e = be; part[blockIdx.x].idx = bi;
  }
}
__global__ __launch_bounds__(256) void k_peak_final(const PeakRec *__restrict__ part, int nparts,
                                                    const double2 *__restrict__ spec, int N, pmd_peak *out) {
  __shared__ PeakRec ws[256];
  double be = -1.0; int bi = -1;
  for (int p = threadIdx.x; p < nparts; p += 256) if (peak_better(part[p].e, part[p].idx, be, bi)) { be = part[p].e; bi = part[p].idx; }
  ws[threadIdx.x].e = be; ws[threadIdx.x].idx = bi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o && peak_better(ws[threadIdx.x + o].e, ws[threadIdx.x + o].idx, ws[threadIdx.x].e, ws[threadIdx.x].idx))
      ws[threadIdx.x] = ws[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  be = ws[0].e; bi = ws[0].idx;
  out->peak = bi; out->maxenergy = be;
  if (bi >= 0) {
    int next = (bi + 1) % N, prev = (N + bi - 1) % N;
    out->peak_re = spec[bi].x; out->peak_im = spec[bi].y;
    out->next_re = spec[next].x; out->next_im = spec[next].y;
    out->prev_re = spec[prev].x; out->prev_im = spec[prev].y;
  }
}

// carrier_i of pmdemod.c:332-335 in closed form (carrier_params.c)
__device__ __forceinline__ double2 carrier_at(unsigned long long i, unsigned long long u_hi,
                                              unsigned long long u_lo, double logrho) {
  unsigned long long f = i * u_hi + __umul64hi(i, u_lo);     // frac(i * u) in 0.64 fixed point
  double phi = (double)f * 5.42101086242752217e-20;           // 2^-64
  double s, c;
  sincospi(2.0 * phi, &s, &c);
  double mag = 1.0 + (double)i * logrho;
  return make_double2(mag * c, -(mag * s));
}

#define RED_BLOCKS 1024
// pass 1 (pmdemod.c:328-336): sum of sample_i * carrier_i.  The samples are re-formed from the int16 block and the products
// are not stored: pass 2 forms them again (same operations, same bits) -- 4 B instead of 2 x 16 B of traffic per sample.
__global__ __launch_bounds__(256) void k_mix(const short2 *__restrict__ iq, const double2 *__restrict__ lo, int flip, int N,
                                             unsigned long long u_hi, unsigned long long u_lo, double logrho,
                                             double2 *__restrict__ part) {
  double sr = 0, si = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
    const double2 v = iq_sample(iq, lo, i, flip), c = carrier_at((unsigned long long)i, u_hi, u_lo, logrho);
    const double nx = v.x * c.x - v.y * c.y, ny = v.x * c.y + v.y * c.x;
    sr += nx; si += ny;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sr += __shfl_xor(sr, o, 64); si += __shfl_xor(si, o, 64); }
  __shared__ double2 ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = make_double2(sr, si);
  __syncthreads();
  if (threadIdx.x == 0)
    part[blockIdx.x] = make_double2(ws[0].x + ws[1].x + ws[2].x + ws[3].x, ws[0].y + ws[1].y + ws[2].y + ws[3].y);
}
__global__ void k_sum2(const double2 *__restrict__ part, int n, double2 *out, double2 *out2 = nullptr) {
  __shared__ double2 ws[256];
  double sr = 0, si = 0;
  for (int i = threadIdx.x; i < n; i += 256) { sr += part[i].x; si += part[i].y; }
  ws[threadIdx.x] = make_double2(sr, si);
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { ws[threadIdx.x].x += ws[threadIdx.x + o].x; ws[threadIdx.x].y += ws[threadIdx.x + o].y; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { *out = ws[0]; if (out2) *out2 = ws[0]; }
}
// pass 2 (pmdemod.c:341-348) + quantise (:360-368)
__global__ __launch_bounds__(256) void k_rotate(const short2 *__restrict__ iq, const double2 *__restrict__ lo, int flip, int N,
                                                unsigned long long u_hi, unsigned long long u_lo, double logrho,
                                                double ur, double ui, double amp,
                                                int16_t *__restrict__ out16, double *__restrict__ pre,
                                                double2 *__restrict__ part) {
  double acc = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
    const double2 s0 = iq_sample(iq, lo, i, flip), c = carrier_at((unsigned long long)i, u_hi, u_lo, logrho);
    const double2 v = make_double2(s0.x * c.x - s0.y * c.y, s0.x * c.y + s0.y * c.x);       // what pass 1 summed
    double nx = v.x * ur - v.y * ui, ny = v.x * ui + v.y * ur;
    double d = nx - amp;
    acc += d * d;
    double q = ny * 0.70710678118654752440;       // M_SQRT1_2
    if (pre) pre[i] = q;
    out16[i] = (short)q;                          // truncation toward zero, as the C cast
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  __shared__ double ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = make_double2(ws[0] + ws[1] + ws[2] + ws[3], 0.0);
}

// ---- the same two passes with the carrier STEPPED instead of evaluated per sample -----------------------------------
// carrier_at() costs a 128-bit product and a double sincospi per sample, twice per sample and block (both passes form
// sample * carrier): the two kernels were compute-bound at 0.75 TB/s.  The carrier is a geometric sequence, so a thread
// seeds ONE closed-form value and multiplies: thread g takes the four consecutive samples 4g .. 4g+3 (one 16-byte load,
// carriers c, c w1, c w2, c w3 with w_k = carrier_k) and moves on by T = gridDim.x * 256 groups with c *= carrier_{4T}.  Every value
// is at most CARRIER_RESEED + 1 roundings away from a closed-form one (~2e-16 each; N = 2^23 needs 8 steps per thread),
// far inside the 1e-9 the spin-down is held to; after CARRIER_RESEED steps the thread seeds again.  Both kernels run the
// SAME arithmetic on the same grid, so pass 2 re-forms exactly the products pass 1 summed.
#define CARRIER_RESEED 64
// carrier_1, carrier_2, carrier_3 and carrier_stride: every workgroup forms them itself (four closed-form values; a kernel of
// their own was one more 5 us launch boundary per block)
__device__ __forceinline__ void carrier_steps(unsigned long long u_hi, unsigned long long u_lo, double logrho, unsigned long long stride,
                                              double2 (&cs)[4]) {
  __shared__ double2 s_cs[4];
  const unsigned t = threadIdx.x;
  if (t < 4) s_cs[t] = carrier_at(t < 3 ? (unsigned long long)(t + 1) : stride, u_hi, u_lo, logrho);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; k++) cs[k] = s_cs[k];
}
// k_sum2's sum, formed by every workgroup for itself (all 256 threads get it): the same additions in the same order
__device__ __forceinline__ double2 sum2_block(const double2 *__restrict__ part, int n) {
  __shared__ double2 ws[256];
  double sr = 0, si = 0;
  for (int i = threadIdx.x; i < n; i += 256) { sr += part[i].x; si += part[i].y; }
  ws[threadIdx.x] = make_double2(sr, si);
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { ws[threadIdx.x].x += ws[threadIdx.x + o].x; ws[threadIdx.x].y += ws[threadIdx.x + o].y; }
    __syncthreads();
  }
  return ws[0];
}
__device__ __forceinline__ void iq_group4(const short2 *__restrict__ iq, const double2 *__restrict__ lo, int g, int flip,
                                          double2 (&v)[4]) {
  const uint4 raw = reinterpret_cast<const uint4 *>(iq)[g];
  const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const short a = (short)(w[k] & 0xffffu), b = (short)(w[k] >> 16);            // short2 {x, y} in memory order
    double x = flip ? (double)b : (double)a, y = flip ? (double)a : (double)b;
    if (lo) {                                // buffer[i] *= conj(lophase), as iq_sample
      const double2 l = lo[4 * g + k];
      const double pr = l.x, pi = -l.y;
      const double nx = x * pr - y * pi, ny = x * pi + y * pr;
      x = nx; y = ny;
    }
    v[k] = make_double2(x, y);
  }
}
__global__ __launch_bounds__(256) void k_mix4(const short2 *__restrict__ iq, const double2 *__restrict__ lo, int flip, int N,
                                              unsigned long long u_hi, unsigned long long u_lo, double logrho,
                                              double2 *__restrict__ part) {
  const int T = gridDim.x * 256, ng = N >> 2;
  double2 cs[4];
  carrier_steps(u_hi, u_lo, logrho, 4ull * (unsigned long long)T, cs);
  const double2 w1 = cs[0], w2 = cs[1], w3 = cs[2], wT = cs[3];
  double sr = 0, si = 0;
  double2 c0 = make_double2(1.0, 0.0);
  int it = 0;
  for (int g = blockIdx.x * 256 + threadIdx.x; g < ng; g += T, it++) {
    c0 = (it & (CARRIER_RESEED - 1)) == 0 ? carrier_at(4ull * (unsigned long long)g, u_hi, u_lo, logrho) : cmul(c0, wT);
    double2 v[4];
    iq_group4(iq, lo, g, flip, v);
    const double2 c[4] = {c0, cmul(c0, w1), cmul(c0, w2), cmul(c0, w3)};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      sr += v[k].x * c[k].x - v[k].y * c[k].y;
      si += v[k].x * c[k].y + v[k].y * c[k].x;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sr += __shfl_xor(sr, o, 64); si += __shfl_xor(si, o, 64); }
  __shared__ double2 ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = make_double2(sr, si);
  __syncthreads();
  if (threadIdx.x == 0)
    part[blockIdx.x] = make_double2(ws[0].x + ws[1].x + ws[2].x + ws[3].x, ws[0].y + ws[1].y + ws[2].y + ws[3].y);
}
// pass 2; every workgroup adds pass 1's partial sums itself (sum2_block: no kernel and no host round trip between the
// passes) and turns the total into conj(dc) / |dc| (pmdemod.c:337-338); workgroup 0 also leaves it in the mailbox (dc_out).
// Its own partial sums (the variance terms) go to pinned host memory: pmd_mix_end adds them (sum2_host)
__global__ __launch_bounds__(256) void k_rotate4(const short2 *__restrict__ iq, const double2 *__restrict__ lo, int flip, int N,
                                                 unsigned long long u_hi, unsigned long long u_lo, double logrho,
                                                 const double2 *__restrict__ mixpart, double2 *__restrict__ dc_out,
                                                 int16_t *__restrict__ out16, double *__restrict__ pre,
                                                 double2 *__restrict__ part) {
  double2 cs[4];
  carrier_steps(u_hi, u_lo, logrho, 4ull * (unsigned long long)gridDim.x * 256ull, cs);
  const double2 w1 = cs[0], w2 = cs[1], w3 = cs[2], wT = cs[3];
  const double2 dcs = sum2_block(mixpart, (int)gridDim.x);      // (k_mix4 ran on the same grid)
  if (blockIdx.x == 0 && threadIdx.x == 0) *dc_out = dcs;
  const double dcr = dcs.x / N, dci = dcs.y / N;
  const double amp = hypot(dcr, dci);
  const double ur = dcr / amp, ui = -dci / amp;
  const int T = gridDim.x * 256, ng = N >> 2;
  double acc = 0;
  double2 c0 = make_double2(1.0, 0.0);
  int it = 0;
  for (int g = blockIdx.x * 256 + threadIdx.x; g < ng; g += T, it++) {
    c0 = (it & (CARRIER_RESEED - 1)) == 0 ? carrier_at(4ull * (unsigned long long)g, u_hi, u_lo, logrho) : cmul(c0, wT);
    double2 s0[4];
    iq_group4(iq, lo, g, flip, s0);
    const double2 c[4] = {c0, cmul(c0, w1), cmul(c0, w2), cmul(c0, w3)};
    double q[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const double2 v = make_double2(s0[k].x * c[k].x - s0[k].y * c[k].y, s0[k].x * c[k].y + s0[k].y * c[k].x);   // what pass 1 summed
      const double nx = v.x * ur - v.y * ui, ny = v.x * ui + v.y * ur;
      const double d = nx - amp;
      acc += d * d;
      q[k] = ny * 0.70710678118654752440;       // M_SQRT1_2
    }
    if (pre) {
      reinterpret_cast<double2 *>(pre)[2 * g] = make_double2(q[0], q[1]);
      reinterpret_cast<double2 *>(pre)[2 * g + 1] = make_double2(q[2], q[3]);
    }
    const unsigned lo16 = (unsigned)(unsigned short)(short)q[0] | ((unsigned)(unsigned short)(short)q[1] << 16);   // truncation toward zero, as the C cast
    const unsigned hi16 = (unsigned)(unsigned short)(short)q[2] | ((unsigned)(unsigned short)(short)q[3] << 16);
    reinterpret_cast<uint2 *>(out16)[g] = make_uint2(lo16, hi16);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  __shared__ double ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = make_double2(ws[0] + ws[1] + ws[2] + ws[3], 0.0);
}

// ---- pmdemod's peak search through a SINGLE-precision transform + exact bins ------------------------------------------
// pmdemod.c:253-318 needs three numbers of the 2^23-point spectrum: the largest bin (last one on ties) and its two
// neighbours.  The transform that finds it need not be the one that evaluates it: k_fft_pass<..., float2> moves half the
// bytes per pass and its last pass stores nothing (302 MB instead of 704 MB of traffic at 2^23), leaving one record per
// workgroup.  k_dft_bins -- every workgroup for itself, the records are 32 KiB -- finds the largest record and checks that it
// is the ONLY bin whose single-precision energy lies within PEAK_NEAR of the maximum, then evaluates X[k-1], X[k], X[k+1]
// from the int16 block in DOUBLE precision (stepped twiddles as k_mix4's carriers, seeded from the transform's own table of
// W_N^m) and leaves one partial sum per workgroup in pinned host memory; pmd_fft_peak_end adds them in a fixed order
// (sum2_host = k_sum2's order).  What leaves is at least as accurate as the double transform's bins (a direct sum: ~1e-16
// sqrt(N) relative).  The check is COMPLETE by construction: a bin within PEAK_NEAR of the global maximum is within
// PEAK_NEAR of its own workgroup's maximum, and a workgroup that holds two such bins says so (PeakRec::near > 1).  Then --
// or with a second candidate anywhere (1.6 % of carrier-less noise blocks), or nothing positive (an all-zero block: every
// bin ties) -- status = 1 and pmd_fft_peak_end runs the double transform.  (First form: a candidate-list kernel, up to four
// candidates and a final-sum kernel -- three launches of 6-7 us each behind the passes instead of one.)
struct DftStatus { int status, peak; };                    // in pinned host memory: 0 = the partial sums are those of bin `peak`
// the largest record under pmdemod.c's rule, and whether it stands alone; every thread of a 256-thread workgroup calls it.
// One 16-byte load per record, the thread's records stay in registers for the second look, wave reductions by shuffles: the
// first form (a load per field and pass -- 40 of the kernel's 49 loads per wave, profiles/r03ao_* -- and a tree through LDS):
// k_dft_bins 30.6 -> 28.9 us at 2^23.
#define SP_MAXREC 16                   /* records per thread held in registers: up to 4 096 records (2^24 points) */
__device__ __forceinline__ bool search_peak(const PeakRec *__restrict__ part, int nparts, int *peak) {
  __shared__ PeakRec ws[4];
  __shared__ int wn[4];
  const int tid = (int)threadIdx.x;
  uint4 raw[SP_MAXREC];
#pragma unroll
  for (int k = 0; k < SP_MAXREC; k++) {
    const int p = tid + 256 * k;
    raw[k] = p < nparts ? reinterpret_cast<const uint4 *>(part)[p] : make_uint4(0u, 0xbff00000u, 0xffffffffu, 0u);      // e = -1.0, idx = -1
  }
  double be = -1.0; int bi = -1;
#pragma unroll
  for (int k = 0; k < SP_MAXREC; k++) {
    const double e = __hiloint2double((int)raw[k].y, (int)raw[k].x); const int i = (int)raw[k].z;
    if (i >= 0 && peak_better(e, i, be, bi)) { be = e; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double oe = __shfl_xor(be, o, 64); const int oi = __shfl_xor(bi, o, 64);
    if (peak_better(oe, oi, be, bi)) { be = oe; bi = oi; }
  }
  if ((tid & 63) == 0) { ws[tid >> 6].e = be; ws[tid >> 6].idx = bi; }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 4; w++) if (peak_better(ws[w].e, ws[w].idx, be, bi)) { be = ws[w].e; bi = ws[w].idx; }
  const double lim = be * (1.0 - PEAK_NEAR);
  int mine = 0;
  if (bi >= 0 && be > 0.0) {
#pragma unroll
    for (int k = 0; k < SP_MAXREC; k++) {
      const double e = __hiloint2double((int)raw[k].y, (int)raw[k].x); const int i = (int)raw[k].z;
      if (i >= 0 && e >= lim) mine += (int)raw[k].w > 1 ? 2 : 1;          // (an ambiguous workgroup counts twice)
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if ((tid & 63) == 0) wn[tid >> 6] = mine;
  __syncthreads();
  *peak = bi;
  return bi >= 0 && be > 0.0 && nparts <= 256 * SP_MAXREC && wn[0] + wn[1] + wn[2] + wn[3] == 1;
}
// (the twiddles W_N^m come from the transform's own two-level table -- W_N^(4096 h) * W_N^l, both entries correctly rounded
// -- instead of a sincospi per seed)
__device__ __forceinline__ double2 tw_at(const double2 *__restrict__ twA, const double2 *__restrict__ twB, unsigned m) {
  const double2 wb = twB[m & 4095u], wa = twA[m >> 12];
  return (m >> 12) ? cmul(wa, wb) : wb;
}
// X[bin] = sum_n sample_n e^{-2 pi j bin n / N} for bin = peak - 1, peak, peak + 1 (mod N); one partial sum per workgroup
// and bin: part[b * gridDim.x + blockIdx.x] (pinned host memory)
__global__ __launch_bounds__(256) void k_dft_bins(const short2 *__restrict__ iq, const double2 *__restrict__ lo, int flip, int N,
                                                  const double2 *__restrict__ twA, const double2 *__restrict__ twB,
                                                  const PeakRec *__restrict__ rec, int nrec, int force_fallback,
                                                  DftStatus *__restrict__ status, double2 *__restrict__ part) {
  int peak;
  const bool ok = search_peak(rec, nrec, &peak) && !force_fallback;
  if (blockIdx.x == 0 && threadIdx.x == 0) { status->status = ok ? 0 : 1; status->peak = peak; }
  if (!ok) return;
  __shared__ double2 sw[3][4];
  __shared__ double2 wsum[3][4];
  const int T = gridDim.x * 256, ng = N >> 2;
  const unsigned long long nm = (unsigned long long)N - 1ull;
  unsigned long long u[3] = {(unsigned long long)((N + peak - 1) % N), (unsigned long long)peak, (unsigned long long)((peak + 1) % N)};
  if (threadIdx.x < 12) {
    const int b = threadIdx.x >> 2, j = threadIdx.x & 3;
    const unsigned long long bin = b == 0 ? u[0] : b == 1 ? u[1] : u[2];
    sw[b][j] = tw_at(twA, twB, (unsigned)((bin * (j < 3 ? (unsigned long long)(j + 1) : 4ull * (unsigned long long)T)) & nm));
  }
  __syncthreads();
  double2 w1[3], w2[3], w3[3], wT[3], c0[3];
  double sr[3] = {0, 0, 0}, si[3] = {0, 0, 0};
#pragma unroll
  for (int b = 0; b < 3; b++) {
    w1[b] = sw[b][0]; w2[b] = sw[b][1]; w3[b] = sw[b][2]; wT[b] = sw[b][3];
    c0[b] = make_double2(1.0, 0.0);
  }
  int it = 0;
  for (int g = blockIdx.x * 256 + threadIdx.x; g < ng; g += T, it++) {
    double2 v[4];
    iq_group4(iq, lo, g, flip, v);
#pragma unroll
    for (int b = 0; b < 3; b++) {
      c0[b] = (it & (CARRIER_RESEED - 1)) == 0 ? tw_at(twA, twB, (unsigned)((u[b] * 4ull * (unsigned long long)g) & nm)) : cmul(c0[b], wT[b]);
      const double2 c[4] = {c0[b], cmul(c0[b], w1[b]), cmul(c0[b], w2[b]), cmul(c0[b], w3[b])};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        sr[b] += v[k].x * c[k].x - v[k].y * c[k].y;
        si[b] += v[k].x * c[k].y + v[k].y * c[k].x;
      }
    }
  }
#pragma unroll
  for (int b = 0; b < 3; b++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sr[b] += __shfl_xor(sr[b], o, 64); si[b] += __shfl_xor(si[b], o, 64); }
    if ((threadIdx.x & 63) == 0) wsum[b][threadIdx.x >> 6] = make_double2(sr[b], si[b]);
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int b = threadIdx.x;
    part[(size_t)b * gridDim.x + blockIdx.x] =
      make_double2(wsum[b][0].x + wsum[b][1].x + wsum[b][2].x + wsum[b][3].x, wsum[b][0].y + wsum[b][1].y + wsum[b][2].y + wsum[b][3].y);
  }
}
// k_sum2's sum on the host: thread t adds elements t, t + 256, ...; then the tree 128, 64, ... 1 -- the same additions in
// the same order, so a total formed here equals the one k_sum2 (or sum2_block, on the device) forms
static double2 sum2_host(const volatile double2 *part, int n) {
  double2 ws[256];
  for (int t = 0; t < 256; t++) {
    double sr = 0, si = 0;
    for (int i = t; i < n; i += 256) { sr += part[i].x; si += part[i].y; }
    ws[t] = make_double2(sr, si);
  }
  for (int o = 128; o > 0; o >>= 1)
    for (int t = 0; t < o; t++) { ws[t].x += ws[t + o].x; ws[t].y += ws[t + o].y; }
  return ws[0];
}

// what a transform needs besides its data: the twiddle tables of its size and a stream
struct FftCtx { int N, logN; const double2 *twA, *twB, *twR; hipStream_t st; };
struct FftSrc { const void *x; const void *i16; const double2 *aux; int iparam; };   // see SRC_*: (x) | (iq, lo, flip) | (samples, -, nvalid) | (x, v); x: elements of the transform's type

struct PeakAsk { int first, last; PeakRec *part; int nparts; };     // nparts: set by the launch (one per workgroup)
template <int LG, int SRC, bool FIRST, bool PEAK = false, int FTC = FT, typename V = double2>
static int launch_pass(const FftCtx &c, const FftSrc &in, V *dst, int s, PeakAsk *pk = nullptr) {
  constexpr int R1 = PassShape<LG>::R1, R2 = PassShape<LG>::R2, R = R1 * R2, TH = FTC * (R1 > R2 ? R1 : R2);
  size_t lds = sizeof(V) * (FIRST ? R + 1 : R) * FTC;                  // a first stage re-uses the tile as its padded output image
  if (PEAK && lds < 512) lds = 512;                                    // (the peak records of the waves)
  // ISEE3DSP_FFT_LDS_KB=n: every pass ASKS for n KiB of LDS (it uses what it needs).  With 88 exactly one FFT workgroup fits a
  // CU and 72 KiB stay free: a Viterbi workgroup (68.6 KiB) always finds room beside it, whereas two 64 KiB FFT workgroups
  // on a CU make the 256-workgroup ACS launch wait for one of them to finish
  static const size_t lds_floor = getenv("ISEE3DSP_FFT_LDS_KB") ? (size_t)atoi(getenv("ISEE3DSP_FFT_LDS_KB")) * 1024u : 0u;
  if (lds < lds_floor && lds_floor <= 160u * 1024u) lds = lds_floor;
  // 64 KiB of dynamic LDS at R = 256: raised once per device and instantiation (bit d = done on device d; transforms run
  // from several host threads, a doubled call is harmless, the flag itself is atomic)
  static std::atomic<unsigned long long> attr_set{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_set.load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute((const void *)k_fft_pass<LG, SRC, FIRST, PEAK, FTC, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    attr_set.fetch_or(bit, std::memory_order_release);
  }
  if constexpr (PEAK) {
    pk->nparts = c.N / R / FTC;
    k_fft_pass<LG, SRC, FIRST, true, FTC, V><<<c.N / R / FTC, TH, lds, c.st>>>((const V *)in.x, (const short2 *)in.i16, in.aux, in.iparam, dst, c.twA, c.twB, c.twR, c.N, s,
                                                                               pk->first, pk->last, pk->part);
  } else
    k_fft_pass<LG, SRC, FIRST, false, FTC, V><<<c.N / R / FTC, TH, lds, c.st>>>((const V *)in.x, (const short2 *)in.i16, in.aux, in.iparam, dst, c.twA, c.twB, c.twR, c.N, s);
  return 0;
}
template <int SRC, bool FIRST, bool PEAK = false, typename V = double2>
static int launch_pass_lg(const FftCtx &c, int lg, const FftSrc &in, V *dst, int s, PeakAsk *pk = nullptr) {
  // the search transform (8-byte elements): 32 columns everywhere -- 256-byte runs, 64 KiB tiles at radix 256
  if constexpr (std::is_same<V, float2>::value) {
    // (16 columns -- 32 KiB tiles, four or five workgroups per CU -- measured the same: 29 + 26 + 24 us against 28 + 27 + 26 at
    // 2^23, and twice the records for k_dft_bins; profiles/r03ak_*)
    switch (lg) {
    case 5: return launch_pass<5, SRC, FIRST, PEAK, 32, V>(c, in, dst, s, pk);
    case 6: return launch_pass<6, SRC, FIRST, PEAK, 32, V>(c, in, dst, s, pk);
    case 7: return launch_pass<7, SRC, FIRST, PEAK, 32, V>(c, in, dst, s, pk);
    case 8: return launch_pass<8, SRC, FIRST, PEAK, 32, V>(c, in, dst, s, pk);
    }
    return -1;
  }
  // a first pass over int16 samples (4 or 2 bytes each) takes 32 columns per workgroup where its tile still fits 64 KiB of LDS
  constexpr bool NARROW = FIRST && (SRC == SRC_IQ || SRC == SRC_REAL16);
  // (tiles of at most 32 KiB -- 8 columns for the later radix-256 passes, radix 128 x 16 columns first -- so that workgroups are
  // short-lived and several fit beside a Viterbi workgroup were measured in the 10 MS/s chain: 36.0-37.2 ms against 34.4-35.8,
  // the decoder no further along when the front end ends; profiles/r03p_chain10M_fft_small_tiles.txt)
  switch (lg) {
  case 5: return launch_pass<5, SRC, FIRST, PEAK, NARROW ? 32 : FT, V>(c, in, dst, s, pk);
  case 6: return launch_pass<6, SRC, FIRST, PEAK, NARROW ? 32 : FT, V>(c, in, dst, s, pk);
  case 7: return launch_pass<7, SRC, FIRST, PEAK, NARROW ? 32 : FT, V>(c, in, dst, s, pk);
  case 8: return launch_pass<8, SRC, FIRST, PEAK, FT, V>(c, in, dst, s, pk);
  }
  return -1;
}
// forward unnormalised transform of N = 2^logN >= 2^12 points: ceil(logN / 8) LDS-staged passes of 5..8 levels each, the
// first one reading `in` in the form SRC; ping-pong between out and tmp so that the last pass lands in `out`
// (pk: the last pass -- never the first: N >= 2^12 takes two -- also leaves per-workgroup peak records)
template <int SRC, typename V = double2>
static int fft_forward(const FftCtx &c, const FftSrc &in, V *out, V *tmp, PeakAsk *pk = nullptr) {
  const int npass = (c.logN + 7) / 8, base = c.logN / npass, extra = c.logN % npass;
  FftSrc cur = in;
  int s = 1;
  if (pk) pk->nparts = 0;
  for (int i = 0; i < npass; i++) {
    // the larger radices first (2^23 = 8 + 8 + 7).  ISEE3DSP_FFT_ORDER=asc puts the smaller ones first (7 + 8 + 8: the first
    // pass then takes 32 columns = 128-byte read runs): measured at 2^23, 102 + 72 + 67 us against 98 + 73 + 60
    static const bool asc = getenv("ISEE3DSP_FFT_ORDER") && getenv("ISEE3DSP_FFT_ORDER")[0] == 'a';
    const int lg = base + ((asc ? i >= npass - extra : i < extra) ? 1 : 0);
    V *dst = ((npass - 1 - i) & 1) == 0 ? out : tmp;
    if (pk && i == npass - 1 && i > 0) { if (launch_pass_lg<SRC_C2, false, true, V>(c, lg, cur, dst, s, pk) != 0) return -1; }
    else if ((i == 0 ? launch_pass_lg<SRC, true, false, V>(c, lg, cur, dst, s) : launch_pass_lg<SRC_C2, false, false, V>(c, lg, cur, dst, s)) != 0) return -1;
    cur = FftSrc{dst, nullptr, nullptr, 0}; s <<= lg;
  }
  return 0;
}
static int fft_tables(FftCtx *c, int N, double2 **twA, double2 **twB, double2 **twR, hipStream_t st) {
  int lg = 0; while ((1 << lg) < N) lg++;
  const int nA = N / 4096 > 0 ? N / 4096 : 1;
  if (hipMalloc(twA, sizeof(double2) * (size_t)nA) != hipSuccess || hipMalloc(twB, sizeof(double2) * 4096) != hipSuccess ||
      hipMalloc(twR, sizeof(double2) * 256) != hipSuccess) return -1;
  k_twiddles2<<<(nA > 4096 ? nA : 4096) / 256, 256, 0, st>>>(*twA, *twB, *twR, N);
  *c = FftCtx{N, lg, *twA, *twB, *twR, st};
  return 0;
}

extern "C" void *pmd_create(int fftsize) {
  Pmd *h = nullptr;
  int lg = 0;
  while ((1 << lg) < fftsize) lg++;
  if (fftsize < 16 || (1 << lg) != fftsize || lg > 24) { snprintf(g_err, sizeof g_err, "pmd_create: fftsize %d not a power of two in [16, 2^24]", fftsize); return nullptr; }
  h = (Pmd *)calloc(1, sizeof(Pmd));
  if (!h) return nullptr;
  { int gd = g_device.load(std::memory_order_acquire); h->dev = gd >= 0 ? gd : 0; }
  h->N = fftsize; h->logN = lg;
  CHK(hipSetDevice(h->dev));
  CHK(dsp_stream_create(&h->st, &h->own_st));
  CHK(hipMalloc(&h->spec, sizeof(double2) * (size_t)fftsize));
  CHK(hipMalloc(&h->tmp, sizeof(double2) * (size_t)fftsize));
  CHK(hipMalloc(&h->d_iq, sizeof(int16_t) * 2 * (size_t)fftsize));
  CHK(hipMalloc(&h->d_out16, sizeof(int16_t) * (size_t)fftsize));
  CHK(hipMalloc(&h->d_pre, sizeof(double) * (size_t)fftsize));
  h->red_cap = sizeof(double2) * (RED_BLOCKS + 64) + sizeof(pmd_peak) + sizeof(PeakRec) * RED_BLOCKS;
  CHK(hipMalloc(&h->d_red, h->red_cap));
  if (pin_grow(&h->pin_hdr, PIN_BYTES) != 0) { snprintf(g_err, sizeof g_err, "pmd_create: pinned mailbox"); goto fail; }
  CHK(hipEventCreateWithFlags(&h->ev_peak, hipEventDisableTiming));
  CHK(hipEventCreateWithFlags(&h->ev_mix, hipEventDisableTiming));
  if (lg >= 12 && !getenv("ISEE3DSP_FFT_REGISTER_RADIX")) {
    FftCtx c;
    if (fft_tables(&c, fftsize, &h->twA, &h->twB, &h->twR, h->st) != 0) { snprintf(g_err, sizeof g_err, "pmd_create: twiddle tables"); goto fail; }
  } else {                                           // small transforms: register-radix stages on a double copy of the block
    CHK(hipMalloc(&h->buf, sizeof(double2) * (size_t)fftsize));
    CHK(hipMalloc(&h->tw, sizeof(double2) * (size_t)(fftsize / 2)));
    k_twiddles<<<(fftsize / 2 + 255) / 256, 256, 0, h->st>>>(h->tw, fftsize);
  }
  CHK(hipGetLastError());
  CHK(hipStreamSynchronize(h->st));
  return h;
fail:
  pmd_destroy(h);
  return nullptr;
}
extern "C" void pmd_destroy(void *p) {
  Pmd *h = (Pmd *)p;
  if (!h) return;
  (void)hipSetDevice(h->dev);
  (void)hipStreamSynchronize(h->st); if (h->st && h->own_st) (void)hipStreamDestroy(h->st);
  (void)hipFree(h->buf); (void)hipFree(h->spec); (void)hipFree(h->tmp); (void)hipFree(h->tw); (void)hipFree(h->lo);
  (void)hipFree(h->twA); (void)hipFree(h->twB); (void)hipFree(h->twR);
  (void)hipFree(h->d_iq); (void)hipFree(h->d_out16); (void)hipFree(h->d_pre); (void)hipFree(h->d_red); (void)hipFree(h->d_peakpart);
  pin_free(&h->pin_hdr);
  if (h->ev_peak) (void)hipEventDestroy(h->ev_peak);
  if (h->ev_mix) (void)hipEventDestroy(h->ev_mix);
  free(h);
}
// de-chirp LO table (N complex doubles = the lophase sequence of pmdemod.c:237-243, computed by the
// host with the reference's own sequential recurrence); NULL switches de-chirp off
extern "C" int pmd_set_dechirp(void *p, const double *lophase_ri) {
  Pmd *h = (Pmd *)p;
  if (!h) return -1;
  CHK(hipSetDevice(h->dev));
  if (!lophase_ri) { h->have_lo = 0; return 0; }
  if (!h->lo) CHK(hipMalloc(&h->lo, sizeof(double2) * (size_t)h->N));
  CHK(hipMemcpy(h->lo, lophase_ri, sizeof(double2) * (size_t)h->N, hipMemcpyHostToDevice));
  h->have_lo = 1;
  return 0;
fail:
  return -1;
}
// The block is only announced here (a host block is copied to the handle's staging buffer): the FFT's first pass, the
// spin-down sum and the output kernel each read the int16 pairs themselves.  A DEVICE block must therefore stay valid
// and unchanged until pmd_mix_quantise has returned.
extern "C" int pmd_load(void *p, const int16_t *iq, int is_dev, int flip) {
  Pmd *h = (Pmd *)p;
  if (!h) return -1;
  CHK(hipSetDevice(h->dev));
  if (!is_dev) {
    CHK(hipMemcpyAsync(h->d_iq, iq, sizeof(int16_t) * 2 * (size_t)h->N, hipMemcpyHostToDevice, h->st));
    h->cur_iq = h->d_iq;
  } else h->cur_iq = iq;
  h->cur_flip = flip;
  if (h->buf) {
    k_pmd_load<<<(h->N + 255) / 256, 256, 0, h->st>>>((const short2 *)h->cur_iq, h->buf, h->have_lo ? h->lo : nullptr, h->N, flip);
    CHK(hipGetLastError());
  }
  return 0;
fail:
  return -1;
}
// The two engine calls of a block in an asynchronous form: *_begin enqueues and records an event, *_end waits for exactly
// that event and reads the results out of the pinned mailbox.  A caller that alternates two handles on one stream can then
// have block k+1's transform in the stream before it waits for block k's (cli/pmdemod_core.c), instead of leaving the GPU
// idle while the host forms Quinn's estimate and the quad-precision carrier parameters.
static int peak_begin(Pmd *h, int firstbin, int lastbin, bool search);
extern "C" int pmd_fft_peak_end(void *p, pmd_peak *out) {
  Pmd *h = (Pmd *)p;
  if (!h || !out) return -1;
  CHK(hipSetDevice(h->dev));
  CHK(hipEventSynchronize(h->ev_peak));
  if (h->last_path == 1) {               // search transform: the exact bins' partial sums are in the mailbox
    const volatile DftStatus *st = (const volatile DftStatus *)((char *)h->pin_hdr.h + 256);
    if (st->status == 0) {
      const volatile double2 *part = (const volatile double2 *)((char *)h->pin_hdr.h + PIN_DFT);
      const double2 xp = sum2_host(part, h->dft_nb), xk = sum2_host(part + h->dft_nb, h->dft_nb), xn = sum2_host(part + 2 * h->dft_nb, h->dft_nb);
      out->peak = st->peak; out->maxenergy = xk.x * xk.x + xk.y * xk.y;
      out->peak_re = xk.x; out->peak_im = xk.y; out->next_re = xn.x; out->next_im = xn.y; out->prev_re = xp.x; out->prev_im = xp.y;
      return 0;
    }
    // single precision could not name the peak for certain: the double transform decides
    if (peak_begin(h, h->ask_first, h->ask_last, false) != 0) return -1;
    h->last_path = 2;
    CHK(hipEventSynchronize(h->ev_peak));
  }
  memcpy(out, h->pin_hdr.h, sizeof(pmd_peak));
  return 0;
fail:
  return -1;
}
extern "C" int pmd_fft_peak(void *p, int firstbin, int lastbin, pmd_peak *out) {
  if (pmd_fft_peak_begin(p, firstbin, lastbin) != 0) return -1;
  return pmd_fft_peak_end(p, out);
}
extern "C" int pmd_fft_peak_begin(void *p, int firstbin, int lastbin) {
  Pmd *h = (Pmd *)p;
  if (!h) return -1;
  if (firstbin < 0 || lastbin > h->N || firstbin > lastbin) { snprintf(g_err, sizeof g_err, "pmd_fft_peak: bad bin range"); return -1; }
  if (!h->cur_iq) { snprintf(g_err, sizeof g_err, "pmd_fft_peak: no block loaded"); return -1; }
  // ISEE3DSP_FFT_F64=1: the double transform always (its spectrum is then what pmd_get_spectrum copies, without a second run)
  static const bool f64_only = getenv("ISEE3DSP_FFT_F64") && atoi(getenv("ISEE3DSP_FFT_F64")) != 0;
  return peak_begin(h, firstbin, lastbin, !h->buf && !f64_only && ((uintptr_t)h->cur_iq & 15u) == 0);
}
static int peak_begin(Pmd *h, int firstbin, int lastbin, bool search) {
  CHK(hipSetDevice(h->dev));
  h->ask_first = firstbin; h->ask_last = lastbin;
  if (search) {
    // single-precision search transform (spec / tmp as its ping-pong buffers), candidates, exact bins
    static const bool force_fb = getenv("ISEE3DSP_FFT_F32_FORCE_FALLBACK") != nullptr;     // test hook
    const FftCtx c{h->N, h->logN, h->twA, h->twB, h->twR, h->st};
    const FftSrc in{nullptr, h->cur_iq, h->have_lo ? h->lo : nullptr, h->cur_flip};
    // the exact bins: every workgroup resident at once (146 VGPRs: three workgroups per CU)
    int nbd = (h->N / 4 + 255) / 256; if (nbd > 768) nbd = 768;         // (256 .. 768 workgroups measured the same; PIN_DFT holds 768)
    if (grow(&h->d_peakpart, &h->peakpart_cap, sizeof(PeakRec) * (size_t)(h->N / 32 / FT)) != 0) { snprintf(g_err, sizeof g_err, "pmd_fft_peak: scratch"); return -1; }
    PeakAsk ask{firstbin, lastbin, (PeakRec *)h->d_peakpart, 0};
    h->spec_valid = 0; h->last_path = 1;
    if (fft_forward<SRC_IQ, float2>(c, in, (float2 *)h->spec, (float2 *)h->tmp, &ask) != 0 || ask.nparts <= 0) { snprintf(g_err, sizeof g_err, "pmd_fft_peak: FFT launch failed"); return -1; }
    h->dft_nb = nbd;
    k_dft_bins<<<nbd, 256, 0, h->st>>>((const short2 *)h->cur_iq, h->have_lo ? h->lo : nullptr, h->cur_flip, h->N, h->twA, h->twB, ask.part, ask.nparts,
                                       force_fb ? 1 : 0, (DftStatus *)((char *)h->pin_hdr.d + 256), (double2 *)((char *)h->pin_hdr.d + PIN_DFT));
    CHK(hipGetLastError());
    CHK(hipEventRecord(h->ev_peak, h->st));
    return 0;
  }
  {
    PeakAsk ask{firstbin, lastbin, nullptr, 0};
    h->spec_valid = 1; h->last_path = 0;
    if (!h->buf) {
      // LDS-staged passes, the first one straight from the int16 block, the last one leaving the peak records
      const FftCtx c{h->N, h->logN, h->twA, h->twB, h->twR, h->st};
      const FftSrc in{nullptr, h->cur_iq, h->have_lo ? h->lo : nullptr, h->cur_flip};
      const bool ride = !getenv("ISEE3DSP_PEAK_SEPARATE") && grow(&h->d_peakpart, &h->peakpart_cap, sizeof(PeakRec) * (size_t)(h->N / 32 / FT)) == 0;
      ask.part = (PeakRec *)h->d_peakpart;
      if (fft_forward<SRC_IQ>(c, in, h->spec, h->tmp, ride ? &ask : nullptr) != 0) { snprintf(g_err, sizeof g_err, "pmd_fft_peak: FFT launch failed"); return -1; }
    } else {
      // stages ping-pong so that the last one lands in spec; buf is never written.  Radix plan: the
      // small remainder radix first (its short output runs matter least while s is tiny), then radix 16.
      int radices[12], nst = 0, rem = h->logN;
      if (getenv("ISEE3DSP_FFT_RADIX2")) { while (rem > 0) { radices[nst++] = 2; rem--; } }
      else {
        if (rem % 4) { radices[nst++] = 1 << (rem % 4); rem -= rem % 4; }
        while (rem > 0) { radices[nst++] = 16; rem -= 4; }
      }
      const double2 *src = h->buf;
      int s = 1;
      for (int st = 0; st < nst; st++) {
        bool to_spec = ((nst - 1 - st) & 1) == 0;
        double2 *dst = to_spec ? h->spec : h->tmp;
        const int R = radices[st], nthr = h->N / R;
        if (R == 2) k_fft_stage<<<(nthr + 255) / 256, 256, 0, h->st>>>(src, dst, h->tw, h->N, s);
        else if (R == 4) k_fft_radix<4><<<(nthr + 255) / 256, 256, 0, h->st>>>(src, dst, h->tw, h->N, s);
        else if (R == 8) k_fft_radix<8><<<(nthr + 255) / 256, 256, 0, h->st>>>(src, dst, h->tw, h->N, s);
        else k_fft_radix<16><<<(nthr + 255) / 256, 256, 0, h->st>>>(src, dst, h->tw, h->N, s);
        src = dst; s *= R;
      }
    }
    PeakRec *part = (PeakRec *)((char *)h->d_red + sizeof(double2) * (RED_BLOCKS + 64) + sizeof(pmd_peak));
    int nb = (lastbin - firstbin + 255) / 256;
    if (nb > RED_BLOCKS) nb = RED_BLOCKS;
    if (nb < 1) nb = 1;
    if (ask.nparts > 0) { part = ask.part; nb = ask.nparts; }                                  // the last FFT pass has done it
    else k_peak_partial<<<nb, 256, 0, h->st>>>(h->spec, firstbin, lastbin, part);
    k_peak_final<<<1, 256, 0, h->st>>>(part, nb, h->spec, h->N, (pmd_peak *)h->pin_hdr.d);      // straight into mapped host memory
    CHK(hipGetLastError());
    CHK(hipEventRecord(h->ev_peak, h->st));
  }
  return 0;
fail:
  return -1;
}
// 0 enqueued (pmd_mix_end collects); 1 this block takes the synchronous call (small or unaligned, ISEE3DSP_CARRIER_CLOSED=1)
extern "C" int pmd_mix_begin(void *p, double cstep, int16_t *out16, double *pre, int out_is_dev) {
  Pmd *h = (Pmd *)p;
  if (!h) return -1;
  if (!h->cur_iq) { snprintf(g_err, sizeof g_err, "pmd_mix_quantise: no block loaded"); return -1; }
  CHK(hipSetDevice(h->dev));
  {
    double2 *part = (double2 *)h->d_red;
    const short2 *iq = (const short2 *)h->cur_iq;
    const double2 *lo = h->have_lo ? h->lo : nullptr;
    int16_t *o16v = (out16 && out_is_dev) ? out16 : h->d_out16;
    double *oprev = pre ? (out_is_dev ? pre : h->d_pre) : nullptr;
    static const bool closed_form = getenv("ISEE3DSP_CARRIER_CLOSED") && atoi(getenv("ISEE3DSP_CARRIER_CLOSED")) != 0;
    if (closed_form || h->N < 1024 || ((uintptr_t)iq & 15u) != 0 || ((uintptr_t)o16v & 7u) != 0 || ((uintptr_t)oprev & 15u) != 0) return 1;
    uint64_t u_hi, u_lo; double logrho;
    pmd_carrier_params(cstep, &u_hi, &u_lo, &logrho);
    // stepped carrier, four samples per thread and step, both passes and their sums without a host round trip in between
    int nb4 = (h->N / 4 + 255) / 256; if (nb4 > RED_BLOCKS) nb4 = RED_BLOCKS;
    h->mix_nb = nb4;
    k_mix4<<<nb4, 256, 0, h->st>>>(iq, lo, h->cur_flip, h->N, u_hi, u_lo, logrho, part);
    k_rotate4<<<nb4, 256, 0, h->st>>>(iq, lo, h->cur_flip, h->N, u_hi, u_lo, logrho, part, (double2 *)((char *)h->pin_hdr.d + 128), o16v, oprev,
                                      (double2 *)((char *)h->pin_hdr.d + PIN_ROT));
    CHK(hipGetLastError());
    if (out16 && !out_is_dev) CHK(hipMemcpyAsync(out16, h->d_out16, sizeof(int16_t) * (size_t)h->N, hipMemcpyDeviceToHost, h->st));
    if (pre && !out_is_dev) CHK(hipMemcpyAsync(pre, h->d_pre, sizeof(double) * (size_t)h->N, hipMemcpyDeviceToHost, h->st));
    CHK(hipEventRecord(h->ev_mix, h->st));
  }
  return 0;
fail:
  return -1;
}
extern "C" int pmd_mix_end(void *p, pmd_mix *res) {
  Pmd *h = (Pmd *)p;
  if (!h || !res) return -1;
  CHK(hipSetDevice(h->dev));
  CHK(hipEventSynchronize(h->ev_mix));
  {
    const volatile double *m = (const volatile double *)((char *)h->pin_hdr.h + 128);
    const double dcr = m[0] / h->N, dci = m[1] / h->N;
    const double2 var = sum2_host((const volatile double2 *)((char *)h->pin_hdr.h + PIN_ROT), h->mix_nb);
    res->dc_re = dcr; res->dc_im = dci; res->amplitude = hypot(dcr, dci); res->diffsumsq = var.x / h->N;
  }
  return 0;
fail:
  return -1;
}
extern "C" int pmd_mix_quantise(void *p, double cstep, pmd_mix *res, int16_t *out16, double *pre, int out_is_dev) {
  Pmd *h = (Pmd *)p;
  if (!h) return -1;
  {
    const int b = pmd_mix_begin(p, cstep, out16, pre, out_is_dev);
    if (b < 0) return -1;
    if (b == 0) return pmd_mix_end(p, res);
  }
  CHK(hipSetDevice(h->dev));
  {
    uint64_t u_hi, u_lo; double logrho;
    pmd_carrier_params(cstep, &u_hi, &u_lo, &logrho);
    double2 *part = (double2 *)h->d_red;
    const short2 *iq = (const short2 *)h->cur_iq;
    const double2 *lo = h->have_lo ? h->lo : nullptr;
    int nb = (h->N + 255) / 256; if (nb > RED_BLOCKS) nb = RED_BLOCKS;
    // closed-form carrier per sample (small or unaligned blocks; ISEE3DSP_CARRIER_CLOSED=1)
    k_mix<<<nb, 256, 0, h->st>>>(iq, lo, h->cur_flip, h->N, u_hi, u_lo, logrho, part);
    k_sum2<<<1, 256, 0, h->st>>>(part, nb, (double2 *)((char *)h->pin_hdr.d + 128));
    double2 dc;
    CHK(hipGetLastError());
    CHK(hipStreamSynchronize(h->st));
    { const volatile double *m = (const volatile double *)((char *)h->pin_hdr.h + 128); dc.x = m[0]; dc.y = m[1]; }
    double dcr = dc.x / h->N, dci = dc.y / h->N;
    double amp = hypot(dcr, dci);                 // cabs, pmdemod.c:337
    double ur = dcr / amp, ui = -dci / amp;       // conj(dc) / amp, :338
    int16_t *o16 = (out16 && out_is_dev) ? out16 : h->d_out16;
    double *opre = pre ? (out_is_dev ? pre : h->d_pre) : nullptr;
    k_rotate<<<nb, 256, 0, h->st>>>(iq, lo, h->cur_flip, h->N, u_hi, u_lo, logrho, ur, ui, amp, o16, opre, part);
    k_sum2<<<1, 256, 0, h->st>>>(part, nb, (double2 *)((char *)h->pin_hdr.d + 144));
    if (out16 && !out_is_dev) CHK(hipMemcpyAsync(out16, h->d_out16, sizeof(int16_t) * (size_t)h->N, hipMemcpyDeviceToHost, h->st));
    if (pre && !out_is_dev) CHK(hipMemcpyAsync(pre, h->d_pre, sizeof(double) * (size_t)h->N, hipMemcpyDeviceToHost, h->st));
    CHK(hipStreamSynchronize(h->st));
    { const volatile double *m = (const volatile double *)((char *)h->pin_hdr.h + 144); dc.x = m[0]; dc.y = m[1]; }
    res->dc_re = dcr; res->dc_im = dci; res->amplitude = amp; res->diffsumsq = dc.x / h->N;
  }
  return 0;
fail:
  return -1;
}
// ===========================================================================================
// icesync.c:55-208 -- FFT sync-vector correlator
// ===========================================================================================
// Corr_result[n] = sum_m samples[m] * vec[m - n] through three transforms of corr_size points: the vector's (once per
// vector, conjugated: icesync.c:122-135), the zero-padded frame's (:151-163), and the inverse of their product (:166-170;
// FFTW's unnormalised c2r = Re(FFT(conj(Y)))).  Then the first maximum > 0 in [low, high), folded above size/2 (:188-206).
struct Isync {
  int dev; hipStream_t st; int own_st;
  int N, logN; FftCtx c;
  double2 *twA, *twB, *twR;
  double2 *V, *D, *R, *tmp;        // conj(FFT(vec)), FFT(frame), result, ping-pong
  int16_t *d_s; int have_vec;
  void *d_red;
};
#define ISYNC_FAIL (-1234567890)     /* icesync.c:31 SYNC_FAIL */
__device__ __forceinline__ bool first_better(double e, int i, double be, int bi) { return e > be || (e == be && bi >= 0 && i < bi); }
__global__ __launch_bounds__(256) void k_conj_inplace(double2 *v, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) v[i].y = -v[i].y;
}
// all samples zero?  (icesync.c:151-158) -> *flag stays 0
__global__ __launch_bounds__(256) void k_any_nonzero(const int16_t *__restrict__ s, int n, unsigned *flag) {
  int nz = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) nz |= s[i] != 0;
  if (__syncthreads_or(nz) && threadIdx.x == 0) atomicOr(flag, 1u);
}
__global__ __launch_bounds__(256) void k_isync_peak_partial(const double2 *__restrict__ r, int first, int last, PeakRec *__restrict__ part) {
  double be = 0.0; int bi = -1;                       // maxpeak starts at 0: only positive values count (icesync.c:189-198)
  for (int i = first + blockIdx.x * 256 + threadIdx.x; i < last; i += gridDim.x * 256) {
    const double e = r[i].x;
    if (first_better(e, i, be, bi)) { be = e; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    double oe = __shfl_xor(be, o, 64); int oi = __shfl_xor(bi, o, 64);
    if (oi >= 0 && first_better(oe, oi, be, bi)) { be = oe; bi = oi; }
  }
  __shared__ PeakRec ws[4];
  if ((threadIdx.x & 63) == 0) { ws[threadIdx.x >> 6].e = be; ws[threadIdx.x >> 6].idx = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) if (ws[w].idx >= 0 && first_better(ws[w].e, ws[w].idx, be, bi)) { be = ws[w].e; bi = ws[w].idx; }
    part[blockIdx.x].e = be; part[blockIdx.x].idx = bi;
  }
}
__global__ __launch_bounds__(256) void k_isync_peak_final(const PeakRec *__restrict__ part, int nparts, PeakRec *out) {
  __shared__ PeakRec ws[256];
  double be = 0.0; int bi = -1;
  for (int p = threadIdx.x; p < nparts; p += 256) if (part[p].idx >= 0 && first_better(part[p].e, part[p].idx, be, bi)) { be = part[p].e; bi = part[p].idx; }
  ws[threadIdx.x].e = be; ws[threadIdx.x].idx = bi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o && ws[threadIdx.x + o].idx >= 0 &&
        first_better(ws[threadIdx.x + o].e, ws[threadIdx.x + o].idx, ws[threadIdx.x].e, ws[threadIdx.x].idx)) ws[threadIdx.x] = ws[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = ws[0];
}

extern "C" void *isync_create(int corr_size) {
  Isync *h = nullptr;
  int lg = 0;
  while ((1 << lg) < corr_size) lg++;
  if (corr_size < 4096 || (1 << lg) != corr_size || lg > 24) { snprintf(g_err, sizeof g_err, "isync_create: size %d not a power of two in [2^12, 2^24]", corr_size); return nullptr; }
  h = (Isync *)calloc(1, sizeof(Isync));
  if (!h) return nullptr;
  { int gd = g_device.load(std::memory_order_acquire); h->dev = gd >= 0 ? gd : 0; }
  h->N = corr_size; h->logN = lg;
  CHK(hipSetDevice(h->dev));
  CHK(dsp_stream_create(&h->st, &h->own_st));
  if (fft_tables(&h->c, corr_size, &h->twA, &h->twB, &h->twR, h->st) != 0) { snprintf(g_err, sizeof g_err, "isync_create: twiddle tables"); goto fail; }
  CHK(hipMalloc(&h->V, sizeof(double2) * (size_t)corr_size));
  CHK(hipMalloc(&h->D, sizeof(double2) * (size_t)corr_size));
  CHK(hipMalloc(&h->R, sizeof(double2) * (size_t)corr_size));
  CHK(hipMalloc(&h->tmp, sizeof(double2) * (size_t)corr_size));
  CHK(hipMalloc(&h->d_s, sizeof(int16_t) * (size_t)corr_size));
  CHK(hipMalloc(&h->d_red, sizeof(PeakRec) * (RED_BLOCKS + 2) + 16));
  CHK(hipStreamSynchronize(h->st));
  return h;
fail:
  isync_destroy(h);
  return nullptr;
}
extern "C" void isync_destroy(void *p) {
  Isync *h = (Isync *)p;
  if (!h) return;
  (void)hipSetDevice(h->dev);
  (void)hipStreamSynchronize(h->st); if (h->st && h->own_st) (void)hipStreamDestroy(h->st);
  (void)hipFree(h->twA); (void)hipFree(h->twB); (void)hipFree(h->twR);
  (void)hipFree(h->V); (void)hipFree(h->D); (void)hipFree(h->R); (void)hipFree(h->tmp); (void)hipFree(h->d_s); (void)hipFree(h->d_red);
  free(h);
}
// icesync.c:122-135: vec[0..synclen) zero-padded to the transform size, transformed, conjugated
extern "C" int isync_set_vector(void *p, const double *vec, int synclen) {
  Isync *h = (Isync *)p;
  double *host = nullptr;
  if (!h || !vec || synclen < 1 || synclen > h->N) { snprintf(g_err, sizeof g_err, "isync_set_vector: bad arguments"); return -1; }
  CHK(hipSetDevice(h->dev));
  host = (double *)calloc((size_t)h->N, 2 * sizeof(double));
  if (!host) return -1;
  for (int i = 0; i < synclen; i++) host[2 * i] = vec[i];
  CHK(hipMemcpy(h->D, host, sizeof(double2) * (size_t)h->N, hipMemcpyHostToDevice));
  free(host); host = nullptr;
  if (fft_forward<SRC_C2>(h->c, FftSrc{h->D, nullptr, nullptr, 0}, h->V, h->tmp) != 0) { snprintf(g_err, sizeof g_err, "isync_set_vector: FFT launch failed"); return -1; }
  k_conj_inplace<<<(h->N + 255) / 256, 256, 0, h->st>>>(h->V, h->N);
  CHK(hipGetLastError());
  CHK(hipStreamSynchronize(h->st));
  h->have_vec = 1;
  return 0;
fail:
  free(host);
  return -1;
}
// icesync.c:139-208.  samples: nsamples (= (int)ceil of the reference's Framesamples loop bound) int16 values, host or
// device memory.  *peakindex = the folded index, or ISYNC_FAIL (all samples zero, or no positive correlation in the
// window); *maxpeak = Corr_result at the peak (0 on failure).  result_ri (optional, host): Corr_result as N doubles.
extern "C" int isync_search(void *p, const int16_t *samples, int nsamples, int is_dev, int low, int high,
                            int *peakindex, double *maxpeak, double *result) {
  Isync *h = (Isync *)p;
  if (!h || !samples || nsamples < 0 || nsamples > h->N || low < 0 || high < 0 || !peakindex) { snprintf(g_err, sizeof g_err, "isync_search: bad arguments"); return -1; }
  if (!h->have_vec) { snprintf(g_err, sizeof g_err, "isync_search: no sync vector set"); return -1; }
  CHK(hipSetDevice(h->dev));
  {
    const int16_t *src = samples;
    if (!is_dev) { CHK(hipMemcpyAsync(h->d_s, samples, sizeof(int16_t) * (size_t)nsamples, hipMemcpyHostToDevice, h->st)); src = h->d_s; }
    unsigned *flag = (unsigned *)((char *)h->d_red + sizeof(PeakRec) * (RED_BLOCKS + 2));
    PeakRec *part = (PeakRec *)h->d_red, *res = part + RED_BLOCKS;
    unsigned hflag = 0; PeakRec hres;
    CHK(hipMemsetAsync(flag, 0, sizeof(unsigned), h->st));
    int nb = (nsamples + 255) / 256; if (nb > RED_BLOCKS) nb = RED_BLOCKS; if (nb < 1) nb = 1;
    k_any_nonzero<<<nb, 256, 0, h->st>>>(src, nsamples, flag);
    if (fft_forward<SRC_REAL16>(h->c, FftSrc{nullptr, src, nullptr, nsamples}, h->D, h->tmp) != 0 ||
        fft_forward<SRC_CONJPROD>(h->c, FftSrc{h->D, nullptr, h->V, 0}, h->R, h->tmp) != 0) {
      snprintf(g_err, sizeof g_err, "isync_search: FFT launch failed");
      return -1;
    }
    if (high > h->N) high = h->N;                     // icesync.c:192-193
    int np = (high - low + 255) / 256; if (np > RED_BLOCKS) np = RED_BLOCKS; if (np < 1) np = 1;
    k_isync_peak_partial<<<np, 256, 0, h->st>>>(h->R, low, high, part);
    k_isync_peak_final<<<1, 256, 0, h->st>>>(part, np, res);
    CHK(hipMemcpyAsync(&hflag, flag, sizeof hflag, hipMemcpyDeviceToHost, h->st));
    CHK(hipMemcpyAsync(&hres, res, sizeof hres, hipMemcpyDeviceToHost, h->st));
    if (result) {
      CHK(hipStreamSynchronize(h->st));
      double2 *tmp = (double2 *)malloc(sizeof(double2) * (size_t)h->N);
      if (!tmp) return -1;
      if (hipMemcpy(tmp, h->R, sizeof(double2) * (size_t)h->N, hipMemcpyDeviceToHost) != hipSuccess) { free(tmp); return -1; }
      for (int i = 0; i < h->N; i++) result[i] = tmp[i].x;
      free(tmp);
    }
    CHK(hipStreamSynchronize(h->st));
    if (maxpeak) *maxpeak = 0;
    if (!hflag || hres.idx < 0 || hres.e == 0) { *peakindex = ISYNC_FAIL; return 0; }     // :157-158, :200-203
    if (maxpeak) *maxpeak = hres.e;
    *peakindex = hres.idx > h->N / 2 ? h->N - hres.idx : hres.idx;                        // :204-205
  }
  return 0;
fail:
  return -1;
}

extern "C" int pmd_last_peak_path(void *p) { Pmd *h = (Pmd *)p; return h ? h->last_path : -1; }
extern "C" int pmd_get_spectrum(void *p, double *out_ri) {
  Pmd *h = (Pmd *)p;
  if (!h) return -1;
  CHK(hipSetDevice(h->dev));
  if (!h->spec_valid) {                  // the peak search went through the single-precision transform: run the double one now
    if (!h->cur_iq) { snprintf(g_err, sizeof g_err, "pmd_get_spectrum: no block loaded"); return -1; }
    if (peak_begin(h, h->ask_first, h->ask_last, false) != 0) return -1;
    CHK(hipEventSynchronize(h->ev_peak));
  }
  CHK(hipMemcpy(out_ri, h->spec, sizeof(double2) * (size_t)h->N, hipMemcpyDeviceToHost));
  return 0;
fail:
  return -1;
}
