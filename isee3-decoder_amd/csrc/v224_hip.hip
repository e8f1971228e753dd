// v224_hip.hip -- K=24 r=1/2 Viterbi decoder for MI355X (gfx950): C-ABI + HIP kernels.
//
// Replaces viterbi224_port.c / viterbi224_sse2.c behind the reference's own API
// (reference viterbi224.h:8-16; include/viterbi224.h here).  Semantics are the PORT's:
//   ACS       port.c:159-195   tie -> predecessor with MSB set, metrics never saturate
//   init      port.c:34-48     1000 everywhere, 0 at the start state
//   chainback port.c:72-101    rows n % len, bytes MSB-first
//   decodebit port.c:104-143   walk back from the newest row with ring wrap
//
// Data layout in HBM (per decoder):
//   metrics   2 x 2^23 x u16 (16 MiB each, ping-pong).  16 bits suffice: the port's u32 metrics
//             are only ever compared pairwise and their spread is < 12 731 + 510, so a common
//             offset can be subtracted at will (struct V224Dev::off keeps the running total).
//   decisions ring of `len` rows x 1 MiB; rowmeta[row] says in which bit order the row was
//             written (port order by the simple engine, permuted by the multi-step engines).
//   V224Dev   per-workgroup minima + running offset, updated by the kernels themselves.
//
// Engines:
//   SIMPLE  one trellis step per launch.  Thread t owns butterflies 16t..16t+15: two 32-byte
//           coalesced reads (old[i], old[i+2^22]), one 64-byte write (new[2i..2i+31]) and one
//           decision dword in port order.  34.6 MB of HBM traffic per step.
//   FUSED   see v224_fused.hip.inc: K steps per launch with the path metrics of 2^K states held
//           in packed-u16 VGPRs, so metric traffic drops to 32 MiB / K per step.
//   LDS     see v224_lds.hip.inc: 8 steps per launch, 32 KiB tiles staged in LDS, two register
//           levels of 4 steps each, lane-contiguous stores.
//   LDS15   see v224_lds15.hip.inc (default): 15 steps per launch, one 64 KiB tile per workgroup, four
//           register levels; metrics kept in a tile-major order between launches (layout = 1), converted
//           back for anything that needs the natural order.
//
// Beyond the reference's nine functions (include/viterbi224_hip.h): block streaming decode with the tracebacks
// on a second stream, a batch of independent frames over several decoders, and ONE stream decoded by several
// decoders at once whose seams are verified (equal path metrics up to a constant => identical continuation).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <atomic>
#include <mutex>
#include <vector>

#include "v224_common.h"
#pragma GCC visibility push(default)
#include "../../include/viterbi224_hip.h"
#pragma GCC visibility pop

// ------------------------------------------------------------------------------------------
// error reporting
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void v224_set_error(const char *what, hipError_t e, const char *file, int line) {
  snprintf(g_err, sizeof g_err, "%s:%d: %s -> %s", file, line, what, hipGetErrorString(e));
}
extern "C" const char *v224hip_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------
// host object
// ------------------------------------------------------------------------------------------
struct EvPair { hipEvent_t a, b; unsigned steps, launches; };

struct V224 {
  int len, engine, K, dev;
  hipStream_t st;           // ACS stream
  hipStream_t st2;          // traceback stream: == st by default, an own stream with V224HIP_TB_STREAM=1 (tb_own_stream())
  uint16_t *m[2];
  int cur;                  // index of the "old" metric buffer
  uint32_t *rows;           // len x 2^18 dwords
  uint32_t *rowmeta;        // len
  V224Dev *ds;
  uint8_t *dsyms; size_t dsyms_cap;
  uint8_t *dout;  size_t dout_cap;
  uint8_t *dmisc;           // small device scratch (chainback output etc.)
  size_t dmisc_cap;
  int dp;                   // next row to write
  unsigned long long nsteps;// trellis steps since init
  unsigned pass;            // ACS launches since init (minima ping-pong index)
  bool min_valid;           // blkmin[pass & 1] describes the current metric buffer
  int layout;               // 0: m[cur] is in natural state order; 1: in the L15 tile order (v224_lds15.hip.inc)
  bool fresh;               // nothing has run since init: m[cur] is uniform except at `start`
  unsigned start;
  int chunk;                // stream chunk (bits)
  int profile;              // time every profile-th run of ACS launches (0 = off)
  unsigned long long launches_seen;
  std::vector<EvPair> ev_busy, ev_free;
  unsigned long long prof_launches, prof_steps; double prof_ms;
  hipEvent_t ev_acs[2], ev_tb[2];
  // kept between calls (grow-only, freed by delete): seam snapshots of stream_decode_split ([0] as the earlier decoder
  // "A" of a seam, [1] as the later one "B"), their events, the warm-up output nobody reads
  uint16_t *snap[2]; hipEvent_t ev_snap[2];
  uint8_t *warmout; size_t warmout_cap;
  size_t dsyms_off;         // update_viterbi224_blk: next free byte of dsyms (symbols of queued launches stay put)
};

// the PROCESS-WIDE default device of create calls (one process per GPU; the chain library creates its decoders in worker
// threads, which must see what the host selected): atomic, set by v224hip_set_device from any thread
static std::atomic<int> g_device{-1};

// Where a decoder's tracebacks run.  MI355X feeds its compute queues through FOUR hardware pipes, and HIP's hardware
// queues land on them round robin in the order they are created.  Two busy queues on one pipe do not run side by side, and a
// queue that waits for a signal of another queue ON THE SAME PIPE stalls it for ~1 ms per wait (measured,
// scratch/pipe_scan.py: a decoder whose traceback stream is the 4th queue after its pass stream: 1.66 -> 0.62 Msymbols/s;
// two decoders with own traceback streams and one foreign queue between them: 2.44 -> 1.67; the chain 35 -> 57 ms --
// all depending on how many streams the process had created and destroyed before).  A decoder with a pass stream AND a
// traceback stream needs two pipes, two of them all four, and nobody controls which queue gets which.  So by default a
// decoder has ONE stream: the tracebacks of a chunk run in it, behind the chunk's passes, with the wave-per-bit
// kernel k_decodebits_spec (~35 us per chunk of any size; another decoder's passes fill the GPU meanwhile).  Two
// decoders then need two pipes, the whole chain four (two decoders, pmdemod, symdemod).
// V224HIP_TB_STREAM=1 gives every decoder its own traceback stream again (tracebacks under the next chunk's passes).
static bool tb_own_stream(void) { const char *e = getenv("V224HIP_TB_STREAM"); return e && atoi(e) != 0; }

extern "C" int v224hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return -1;
  return n;
}
extern "C" int v224hip_set_device(int dev) {
  if (hipSetDevice(dev) != hipSuccess) return -1;
  g_device.store(dev, std::memory_order_release);
  return 0;
}

// ------------------------------------------------------------------------------------------
// kernels: init, simple ACS
// ------------------------------------------------------------------------------------------
// all metrics 1000 above the start state's (port.c:41-44); `start` = where that state sits in the buffer's order
__global__ __launch_bounds__(256) void k_init(uint16_t *m, unsigned start, V224Dev *ds,
                                              uint32_t *rowmeta, int len) {
  unsigned t = blockIdx.x * 256 + threadIdx.x;            // 2^20 threads x 8 states
  uint4 v;
  const unsigned fill = (V224_BASE + 1000u) * 0x10001u;
  v.x = v.y = v.z = v.w = fill;
  if (t == (start >> 3)) {
    const unsigned one = (start & 1u) ? (V224_BASE << 16) | (V224_BASE + 1000u) : ((V224_BASE + 1000u) << 16) | V224_BASE;
    switch ((start >> 1) & 3u) { case 0: v.x = one; break; case 1: v.y = one; break; case 2: v.z = one; break; default: v.w = one; }
  }
  reinterpret_cast<uint4 *>(m)[t] = v;
  if (t == 0) {
    ds->blkmin[0][0] = V224_BASE; ds->nmin[0] = 1; ds->nmin[1] = 0;
    ds->off = -(long long)V224_BASE;
  }
  (void)rowmeta; (void)len;
}

// wave64 minimum with DPP row operations (VALU only; __shfl_xor would be six ds_bpermute round trips)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#define V224_DPPMIN(ctrl, rowmask)                                                        \
  { unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rowmask, 0xf, false); v = o < v ? o : v; }
  V224_DPPMIN(0xB1, 0xf)    // quad_perm [1,0,3,2]
  V224_DPPMIN(0x4E, 0xf)    // quad_perm [2,3,0,1]
  V224_DPPMIN(0x141, 0xf)   // row_half_mirror
  V224_DPPMIN(0x140, 0xf)   // row_mirror: every lane of a 16-lane row now holds the row minimum
  V224_DPPMIN(0x142, 0xa)   // row_bcast15 into rows 1 and 3
  V224_DPPMIN(0x143, 0xc)   // row_bcast31 into rows 2 and 3: lane 63 holds the wave minimum
#undef V224_DPPMIN
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// minimum of the metrics this launch reads = min over the previous launch's per-workgroup minima.
// Called by every thread of a 256-thread workgroup (contains a barrier).
__device__ __forceinline__ unsigned input_min(const V224Dev *ds, unsigned pass) {
  const unsigned *a = ds->blkmin[pass & 1];
  const unsigned n = ds->nmin[pass & 1];
  unsigned m = 0xffffffffu;
  for (unsigned i = threadIdx.x; i < n; i += 256) { unsigned v = a[i]; m = v < m ? v : m; }
  m = wave_min_u32(m);
  __shared__ unsigned s_in[4];
  if ((threadIdx.x & 63) == 0) s_in[threadIdx.x >> 6] = m;
  __syncthreads();
  unsigned x = s_in[0] < s_in[1] ? s_in[0] : s_in[1];
  unsigned y = s_in[2] < s_in[3] ? s_in[2] : s_in[3];
  return x < y ? x : y;
}
// publish this workgroup's output minimum (mn already wave-reduced); thread 0 of block 0 also keeps
// the running offset and the entry count
__device__ __forceinline__ void output_min(V224Dev *ds, unsigned pass, unsigned mn, long long off_add) {
  __shared__ unsigned s_out[4];
  if ((threadIdx.x & 63) == 0) s_out[threadIdx.x >> 6] = mn;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned a = s_out[0] < s_out[1] ? s_out[0] : s_out[1];
    unsigned b = s_out[2] < s_out[3] ? s_out[2] : s_out[3];
    ds->blkmin[(pass + 1) & 1][blockIdx.x] = a < b ? a : b;
    if (blockIdx.x == 0) { ds->nmin[(pass + 1) & 1] = gridDim.x; ds->off += off_add; }
  }
}

// parity of (2i & POLY1) for i = 0..15 as a 16-bit mask (bit n set => flip)
__host__ __device__ constexpr unsigned par_low16() {
  unsigned m = 0;
  for (unsigned n = 0; n < 16; n++) {
    unsigned x = (2 * n) & V224_POLY1, p = 0;
    while (x) { p ^= x & 1; x >>= 1; }
    m |= p << n;
  }
  return m;
}

// One trellis step, port bit order.  grid = 2^18 threads.
__global__ __launch_bounds__(256) void k_acs_simple(const uint16_t *__restrict__ oldm,
                                                    uint16_t *__restrict__ newm,
                                                    uint32_t *__restrict__ row,
                                                    const uint8_t *__restrict__ syms,
                                                    V224Dev *ds, unsigned pass,
                                                    uint32_t *rowmeta, int rowidx) {
  const unsigned t = blockIdx.x * 256 + threadIdx.x;
  const unsigned s0 = syms[0], s1 = syms[1];
  const unsigned c0 = s0 + 255 - s1, c1 = s0 + s1;

  const uint4 *pi = reinterpret_cast<const uint4 *>(oldm + 16 * t);
  const uint4 *pj = reinterpret_cast<const uint4 *>(oldm + V224_NBFLY + 16 * t);
  uint4 vi[2] = { pi[0], pi[1] }, vj[2] = { pj[0], pj[1] };
  const unsigned adj = input_min(ds, pass) - V224_BASE;
  const unsigned *wi = reinterpret_cast<const unsigned *>(vi);
  const unsigned *wj = reinterpret_cast<const unsigned *>(vj);

  // parity(2i & POLY1) for i = 16t+n splits into parity(32t & POLY1) ^ parity(2n & POLY1)
  constexpr unsigned PLOW = par_low16();
  const unsigned pbase = __popc((32u * t) & V224_POLY1) & 1u;
  const unsigned pmask = pbase ? (~PLOW & 0xffffu) : PLOW;

  unsigned outw[16];
  unsigned dec = 0, mn = 0xffffffffu;
#pragma unroll
  for (int n = 0; n < 16; n++) {
    unsigned oi = ((wi[n >> 1] >> (16 * (n & 1))) & 0xffffu) - adj;
    unsigned oj = ((wj[n >> 1] >> (16 * (n & 1))) & 0xffffu) - adj;
    unsigned c = (n & 1) ? c1 : c0;
    unsigned bm = ((pmask >> n) & 1u) ? 510u - c : c;
    unsigned a0 = oi + bm, a1 = oj + 510u - bm;      // into new state 2i
    unsigned b0 = oi + 510u - bm, b1 = oj + bm;      // into new state 2i+1
    unsigned d0 = a0 >= a1, d1 = b0 >= b1;           // tie -> MSB-set predecessor (port.c:178)
    unsigned n0 = d0 ? a1 : a0, n1 = d1 ? b1 : b0;
    dec |= (d0 | (d1 << 1)) << (2 * n);
    outw[n] = n0 | (n1 << 16);
    unsigned m2 = n0 < n1 ? n0 : n1;
    mn = m2 < mn ? m2 : mn;
  }
  uint4 *po = reinterpret_cast<uint4 *>(newm + 32 * t);
#pragma unroll
  for (int q = 0; q < 4; q++) po[q] = make_uint4(outw[4 * q], outw[4 * q + 1], outw[4 * q + 2], outw[4 * q + 3]);
  row[t] = dec;

  mn = wave_min_u32(mn);
  output_min(ds, pass, mn, (long long)(int)adj);      // adj can be negative after a fused pass
  if (blockIdx.x == 0 && threadIdx.x == 0) rowmeta[rowidx] = V224_META_PORT;
}

// ------------------------------------------------------------------------------------------
// decision-bit fetch (all layouts) and traceback kernels
// ------------------------------------------------------------------------------------------
#include "v224_fused.hip.inc"
#include "v224_lds.hip.inc"
#include "v224_lds15.hip.inc"

__device__ __forceinline__ unsigned get_decision(const uint32_t *__restrict__ rows,
                                                 const uint32_t *__restrict__ rowmeta,
                                                 int row, unsigned state) {
  const uint32_t *r = rows + (size_t)row * V224_ROWWORDS;
  unsigned meta = rowmeta[row];
  if (meta == V224_META_PORT) return (r[state >> 5] >> (state & 31)) & 1u;
  if (meta & (1u << 16)) return lds8_get_decision(r, meta & 0xffu, state);
  if (meta & (1u << 17)) return lds15_get_decision(r, meta & 0xffu, state);
  return fused_get_decision(r, meta, state);
}

// out[j] = decodebit(delay, endstate) as it would read right after trellis step (first + j);
// dp_first = ring index of the row following that step's row.  0xff while history < delay.
__global__ __launch_bounds__(64) void k_decodebits(const uint32_t *__restrict__ rows,
                                                   const uint32_t *__restrict__ rowmeta, int len,
                                                   int dp_first, unsigned long long steps_first,
                                                   int n, int delay, unsigned endstate,
                                                   uint8_t *__restrict__ out) {
  int j = blockIdx.x * 64 + threadIdx.x;
  if (j >= n) return;
  if (steps_first + (unsigned long long)j < (unsigned long long)delay) { out[j] = 0xff; return; }
  int row = (int)(((long long)dp_first + j) % len);
  unsigned st = endstate & V224_SMASK, bit = 0;
  for (int d = 0; d < delay; d++) {
    if (--row < 0) row = len - 1;
    bit = get_decision(rows, rowmeta, row, st);
    st = (bit << (V224_K - 2)) | (st >> 1);
  }
  out[j] = (uint8_t)bit;
}

// The same outputs with one WAVE per output bit and six trellis steps per memory round trip (the speculation of
// k_chainback_spec below: after j steps only 2^j states are possible, so 63 lanes fetch the decisions of all candidates
// of the next six steps at once): 34 dependent round trips instead of 200.  This is what lets the tracebacks of a chunk
// run IN the decoder's own stream (~35 us per chunk, whatever its size) instead of on a second one -- see tb_mode().
__global__ __launch_bounds__(256) void k_decodebits_spec(const uint32_t *__restrict__ rows, const uint32_t *__restrict__ rowmeta, int len,
                                                         int dp_first, unsigned long long steps_first, int n, int delay,
                                                         unsigned endstate, uint8_t *__restrict__ out) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);                  // four output bits per workgroup
  if (j >= n) return;                                                 // (whole waves leave: no wave-level op below is split)
  const unsigned lane = threadIdx.x & 63u;
  if (steps_first + (unsigned long long)j < (unsigned long long)delay) { if (lane == 0) out[j] = 0xff; return; }
  const unsigned lvl = 31u - (unsigned)__clz((int)(lane + 1u));
  const unsigned cand = lane + 1u - (1u << lvl);
  unsigned st = endstate & V224_SMASK, last = 0;
  int row = (int)(((long long)dp_first + j) % len), remaining = delay;
  while (remaining > 0) {
    const int steps = remaining >= 6 ? 6 : remaining;
    unsigned d = 0;
    if (lane < 63u && (int)lvl < steps) {
      int r = (row - 1 - (int)lvl) % len;
      if (r < 0) r += len;
      const unsigned cs = ((cand << (V224_SBITS - lvl)) | (st >> lvl)) & V224_SMASK;
      d = get_decision(rows, rowmeta, r, cs);
    }
    unsigned c = 0;
    for (int k = 0; k < steps; k++) {
      last = (unsigned)__builtin_amdgcn_readlane((int)d, (int)((1u << k) - 1u + c));
      c |= last << k;
    }
    st = ((c << (V224_SBITS - steps)) | (st >> steps)) & V224_SMASK;
    row = (row - steps) % len;
    if (row < 0) row += len;
    remaining -= steps;
  }
  if (lane == 0) out[j] = (uint8_t)last;
}

// framed chainback, port.c:86-98 (row = n % len).  One lane; the walk is a dependent chain.
__global__ void k_chainback(const uint32_t *__restrict__ rows, const uint32_t *__restrict__ rowmeta,
                            int len, unsigned nbits, unsigned endstate, uint8_t *__restrict__ data) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned st = endstate & V224_SMASK, acc = 0;
  for (unsigned n = nbits; n-- > 0;) {
    acc = ((st & 1u) << 7) | (acc >> 1);
    if ((n & 7u) == 0) data[n >> 3] = (uint8_t)acc;
    unsigned b = get_decision(rows, rowmeta, (int)(n % (unsigned)len), st);
    st = (b << (V224_K - 2)) | (st >> 1);
  }
}

// Same walk, six steps per memory round trip.  After j steps the state is
// (c << (23-j)) | (s >> j) with c = the j decision bits read so far: only 2^j candidates, so one wave
// fetches the decision of EVERY candidate of the next six steps at once (1+2+4+8+16+32 = 63 lanes,
// one dependent-load latency), then resolves the six bits with readlane.  The bits leaving on the
// right (what the port packs into data[]) are bits 0..5 of s: known before any load returns.
__global__ __launch_bounds__(64) void k_chainback_spec(const uint32_t *__restrict__ rows,
                                                       const uint32_t *__restrict__ rowmeta, int len,
                                                       unsigned nbits, unsigned endstate,
                                                       uint8_t *__restrict__ data, unsigned row0 = 0) {
  const unsigned lane = threadIdx.x;
  const unsigned lvl = 31u - (unsigned)__clz((int)(lane + 1u));      // lane 0 -> 0, 1-2 -> 1, 3-6 -> 2, ...
  const unsigned cand = lane + 1u - (1u << lvl);
  unsigned st = endstate & V224_SMASK, acc = 0;
  long long n = (long long)nbits - 1;                                // step index, as the port's nbits--
  while (n >= 0) {
    const int steps = n >= 5 ? 6 : (int)(n + 1);
    unsigned d = 0;
    if (lane < 63u && (int)lvl < steps) {
      const unsigned cs = ((cand << (V224_SBITS - lvl)) | (st >> lvl)) & V224_SMASK;
      d = get_decision(rows, rowmeta, (int)((unsigned long long)(row0 + n - lvl) % (unsigned)len), cs);
    }
    unsigned c = 0;
    for (int j = 0; j < steps; j++) {
      const unsigned nn = (unsigned)(n - j);
      acc = (((st >> j) & 1u) << 7) | (acc >> 1);
      if ((nn & 7u) == 0 && lane == 0) data[nn >> 3] = (uint8_t)acc;
      const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)d, (int)((1u << j) - 1u + c));
      c |= b << j;
    }
    st = ((c << (V224_SBITS - steps)) | (st >> steps)) & V224_SMASK;
    n -= steps;
  }
}

// Framed chainback in 16 pieces at once (one workgroup of 16 waves), results identical to the serial walk by
// construction.  Wave w owns the steps of output bytes [w*nbytes/16, (w+1)*nbytes/16).  The top wave starts from the
// caller's end state; every other wave starts CB_WARM steps above its piece from state 0: survivor paths merge going
// back in time (after a few constraint lengths on a decodable frame), so by the time it reaches its piece it is
// normally on the true path.  "Normally" is then checked: going down from the top, the state a piece was entered with
// must equal the state the piece above (already known to be true) left with; a piece that fails is walked again from
// the true state by wave 0 before the check goes on.  So a frame of merged paths costs (piece + warm-up)/6 memory
// round trips instead of nbits/6 (1 024 bits: 32 instead of 171), and a frame of pure noise at worst ~1.3x the serial walk.
#define CB_WAVES 16
#define CB_WARM 128                          /* measured (scratch/cb_warm.py): 0-1 of 180 pieces walked again at Eb/N0 >= 2 dB, 13 % at 1 dB */
#define CB_MAXBYTES 10240                     /* 81 920 bits: vtest224.c:30 caps frames at 80 000 */
#define CB_MINBITS 512

// steps n_from .. n_to (downwards, inclusive) of the port's loop from state st; bytes go to obuf (LDS) unless null
__device__ __forceinline__ unsigned cb_walk(const uint32_t *__restrict__ rows, const uint32_t *__restrict__ rowmeta, int len,
                                            unsigned row0, long long n_from, long long n_to, unsigned st, uint8_t *obuf,
                                            unsigned lane, unsigned lvl, unsigned cand) {
  unsigned acc = 0;
  long long n = n_from;
  while (n >= n_to) {
    const int steps = n - n_to >= 5 ? 6 : (int)(n - n_to + 1);
    unsigned d = 0;
    if (lane < 63u && (int)lvl < steps) {
      const unsigned cs = ((cand << (V224_SBITS - lvl)) | (st >> lvl)) & V224_SMASK;
      d = get_decision(rows, rowmeta, (int)((unsigned long long)(row0 + n - lvl) % (unsigned)len), cs);
    }
    unsigned c = 0;
    for (int j = 0; j < steps; j++) {
      const unsigned nn = (unsigned)(n - j);
      acc = (((st >> j) & 1u) << 7) | (acc >> 1);
      if (obuf && (nn & 7u) == 0 && lane == 0) obuf[nn >> 3] = (uint8_t)acc;
      const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)d, (int)((1u << j) - 1u + c));
      c |= b << j;
    }
    st = ((c << (V224_SBITS - steps)) | (st >> steps)) & V224_SMASK;
    n -= steps;
  }
  return st;
}

__global__ __launch_bounds__(CB_WAVES * 64) void k_chainback_par(const uint32_t *__restrict__ rows, const uint32_t *__restrict__ rowmeta,
                                                                 int len, unsigned nbits, unsigned endstate,
                                                                 uint8_t *__restrict__ data, unsigned row0, unsigned *__restrict__ redone, int cbwarm) {
  __shared__ unsigned s_in[CB_WAVES], s_out[CB_WAVES];
  __shared__ uint8_t s_data[CB_MAXBYTES];
  const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  const unsigned lvl = 31u - (unsigned)__clz((int)(lane + 1u));
  const unsigned cand = lane + 1u - (1u << lvl);
  const unsigned nbytes = (nbits + 7u) >> 3;
  auto piece_lo = [&](unsigned k) { return (long long)(8u * (k * nbytes / CB_WAVES)); };          // first step of piece k
  auto piece_hi = [&](unsigned k) { return k + 1 == CB_WAVES ? (long long)nbits : piece_lo(k + 1); };   // one past its last
  {
    const long long lo = piece_lo(w), hi = piece_hi(w);
    long long top = hi + cbwarm;
    unsigned st = 0;
    if (top >= (long long)nbits) { top = nbits; st = endstate & V224_SMASK; }                    // nothing to guess
    if (top > hi) st = cb_walk(rows, rowmeta, len, row0, top - 1, hi, st, nullptr, lane, lvl, cand);
    if (lane == 0) s_in[w] = st;
    st = cb_walk(rows, rowmeta, len, row0, hi - 1, lo, st, s_data, lane, lvl, cand);
    if (lane == 0) s_out[w] = st;
  }
  __syncthreads();
  unsigned cur = s_out[CB_WAVES - 1], nredo = 0;             // the top piece started from the caller's end state: true
  for (int k = CB_WAVES - 2; k >= 0; k--) {
    if (s_in[k] != cur) {                                    // (same value in every thread: the barriers below are uniform)
      if (w == 0) {
        const unsigned st = cb_walk(rows, rowmeta, len, row0, piece_hi(k) - 1, piece_lo(k), cur, s_data, lane, lvl, cand);
        if (lane == 0) s_out[k] = st;
      }
      nredo++;
      __syncthreads();
    }
    cur = s_out[k];
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < nbytes; i += CB_WAVES * 64) data[i] = s_data[i];
  if (redone && threadIdx.x == 0 && nredo) atomicAdd(redone, nredo);
}

// decodebit (port.c:124-141) with the same six-steps-per-round-trip speculation: walk `delay` rows back from dp with
// ring wrap, return the last decision read.  (One call of the reference's per-bit pattern = one of these.)
__global__ __launch_bounds__(64) void k_decodebit_spec(const uint32_t *__restrict__ rows,
                                                       const uint32_t *__restrict__ rowmeta, int len, int dp,
                                                       int delay, unsigned endstate, uint8_t *__restrict__ out) {
  const unsigned lane = threadIdx.x;
  const unsigned lvl = 31u - (unsigned)__clz((int)(lane + 1u));
  const unsigned cand = lane + 1u - (1u << lvl);
  unsigned st = endstate & V224_SMASK, last = 0;
  int row = dp, remaining = delay;
  while (remaining > 0) {
    const int steps = remaining >= 6 ? 6 : remaining;
    unsigned d = 0;
    if (lane < 63u && (int)lvl < steps) {
      int r = (row - 1 - (int)lvl) % len;
      if (r < 0) r += len;
      const unsigned cs = ((cand << (V224_SBITS - lvl)) | (st >> lvl)) & V224_SMASK;
      d = get_decision(rows, rowmeta, r, cs);
    }
    unsigned c = 0;
    for (int j = 0; j < steps; j++) {
      last = (unsigned)__builtin_amdgcn_readlane((int)d, (int)((1u << j) - 1u + c));
      c |= last << j;
    }
    st = ((c << (V224_SBITS - steps)) | (st >> steps)) & V224_SMASK;
    row = (row - steps) % len;
    if (row < 0) row += len;
    remaining -= steps;
  }
  if (lane == 0) *out = (uint8_t)last;
}

// decodeword, sse2.c:206-243: result = bit<<63 | result>>1 per step
__global__ void k_decodeword(const uint32_t *__restrict__ rows, const uint32_t *__restrict__ rowmeta,
                             int len, int dp, int delay, unsigned endstate,
                             unsigned long long *out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned st = endstate & V224_SMASK;
  unsigned long long r = 0;
  int row = dp;
  for (int d = 0; d < delay; d++) {
    if (--row < 0) row = len - 1;
    unsigned bit = get_decision(rows, rowmeta, row, st);
    st = (bit << (V224_K - 2)) | (st >> 1);
    r = ((unsigned long long)bit << 63) | (r >> 1);
  }
  *out = r;
}

// first index holding the minimum (port.c:113-122: strict <, lowest index wins).
// minval = min of the current buffer, already known from the last ACS launch's slot.
__global__ __launch_bounds__(256) void k_argmin(const uint16_t *__restrict__ m, const V224Dev *ds,
                                                unsigned pass, unsigned *out) {
  unsigned t = blockIdx.x * 256 + threadIdx.x;       // 2^20 threads x 8 states
  unsigned minval = input_min(ds, pass);
  uint4 v = reinterpret_cast<const uint4 *>(m)[t];
  const unsigned *w = reinterpret_cast<const unsigned *>(&v);
  unsigned best = 0xffffffffu;
#pragma unroll
  for (int n = 7; n >= 0; n--) {
    unsigned x = (w[n >> 1] >> (16 * (n & 1))) & 0xffffu;
    if (x == minval) best = 8 * t + n;
  }
  best = wave_min_u32(best);
  if ((threadIdx.x & 63) == 0 && best != 0xffffffffu) atomicMin(out, best);
}

__global__ __launch_bounds__(256) void k_max(const uint16_t *__restrict__ m, unsigned *out) {
  unsigned t = blockIdx.x * 256 + threadIdx.x;
  uint4 v = reinterpret_cast<const uint4 *>(m)[t];
  const unsigned *w = reinterpret_cast<const unsigned *>(&v);
  unsigned mx = 0;
#pragma unroll
  for (int n = 0; n < 8; n++) {
    unsigned x = (w[n >> 1] >> (16 * (n & 1))) & 0xffffu;
    mx = x > mx ? x : mx;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { unsigned y = __shfl_xor(mx, o, 64); mx = y > mx ? y : mx; }
  if ((threadIdx.x & 63) == 0) atomicMax(out, mx);
}

// test export: any layout -> port bit order
__global__ __launch_bounds__(256) void k_export_row(const uint32_t *__restrict__ rows,
                                                    const uint32_t *__restrict__ rowmeta, int row,
                                                    uint32_t *__restrict__ out) {
  unsigned t = blockIdx.x * 256 + threadIdx.x;       // 2^18 words
  unsigned w = 0;
  for (unsigned b = 0; b < 32; b++) w |= get_decision(rows, rowmeta, row, 32 * t + b) << b;
  out[t] = w;
}

__global__ __launch_bounds__(256) void k_export_metrics(const uint16_t *__restrict__ m,
                                                        const V224Dev *ds, unsigned pass,
                                                        uint32_t *__restrict__ out) {
  unsigned t = blockIdx.x * 256 + threadIdx.x;       // 2^23 threads
  out[t] = (unsigned)m[t] - input_min(ds, pass);
}
// recompute the per-workgroup minima of the current metric buffer (only needed when a launch that
// skipped publishing is followed by something that wants them: a ring-wrap remainder pass)
__global__ __launch_bounds__(256) void k_publish_min(const uint16_t *__restrict__ m, V224Dev *ds, unsigned pass) {
  unsigned mn = 0xffffffffu;
  for (unsigned t = blockIdx.x * 256 + threadIdx.x; t < V224_NSTATES / 8; t += gridDim.x * 256) {
    uint4 v = reinterpret_cast<const uint4 *>(m)[t];
    const unsigned *w = reinterpret_cast<const unsigned *>(&v);
#pragma unroll
    for (int n = 0; n < 8; n++) { unsigned x = (w[n >> 1] >> (16 * (n & 1))) & 0xffffu; mn = x < mn ? x : mn; }
  }
  mn = wave_min_u32(mn);
  __shared__ unsigned s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = mn;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned a = s[0] < s[1] ? s[0] : s[1], b = s[2] < s[3] ? s[2] : s[3];
    ds->blkmin[pass & 1][blockIdx.x] = a < b ? a : b;
    if (blockIdx.x == 0) ds->nmin[pass & 1] = gridDim.x;
  }
}
__global__ __launch_bounds__(256) void k_cur_min(const V224Dev *ds, unsigned pass, unsigned *out) {
  unsigned v = input_min(ds, pass);
  if (threadIdx.x == 0) *out = v;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return (e && *e) ? atoi(e) : dflt;
}

// Where the two metric buffers sit relative to each other in physical memory decides how fast the 16 MiB
// write burst at the end of an LDS15 pass drains: allocations fall into two classes, a pair from the same class
// runs the memory phases of a pass in ~8.8 us, a mixed pair in ~11.5 us (scratch/l15_pairs.hip; which class an
// allocation lands in varies from process to process).  So: time the pass with its arithmetic compiled out
// (template ABL = 1: same loads, LDS traffic and stores) on a few candidate partners for m[0], keep the fastest.
static int place_metrics(V224 *v) {
  const int NCAND = 6, NWARM = 4, NPROBE = 12;
  uint16_t *cand[NCAND] = {v->m[1], nullptr, nullptr, nullptr, nullptr, nullptr};
  float t[NCAND] = {0, 0, 0, 0, 0, 0};
  hipEvent_t a = nullptr, b = nullptr;
  int best = 0, n = 0;
  HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
  k_init<<<V224_NSTATES / 8 / 256, 256, 0, v->st>>>(v->m[0], 0, v->ds, v->rowmeta, v->len);
  for (int c = 0; c < NCAND; c++) {
    if (c > 0 && hipMalloc(&cand[c], sizeof(uint16_t) * V224_NSTATES) != hipSuccess) { (void)hipGetLastError(); cand[c] = nullptr; break; }
    n = c + 1;
    uint16_t *m[2] = {v->m[0], cand[c]};
    for (int i = -NWARM; i < NPROBE; i++) {
      if (i == 0) HIPCHK(hipEventRecord(a, v->st));
      if (i & 1) k_acs_lds15<1, false, true><<<256, 1024, L15_LDS_BYTES, v->st>>>(m[1], m[0], v->rows, 0, v->len, v->dmisc, v->ds, (unsigned)i, v->rowmeta);
      else       k_acs_lds15<1, true, false><<<256, 1024, L15_LDS_BYTES, v->st>>>(m[0], m[1], v->rows, 0, v->len, v->dmisc, v->ds, (unsigned)i, v->rowmeta);
    }
    HIPCHK(hipEventRecord(b, v->st));
    HIPCHK(hipEventSynchronize(b));
    HIPCHK(hipEventElapsedTime(&t[c], a, b));
    if (t[c] < t[best]) best = c;
    // two classes only: as soon as one candidate is clearly faster than another, it is in the right one
    float worst = t[0];
    for (int k = 1; k <= c; k++) if (t[k] > worst) worst = t[k];
    if (worst > 1.12f * t[best]) break;
  }
  v->m[1] = cand[best];
  for (int c = 0; c < n; c++) if (c != best) (void)hipFree(cand[c]);
  HIPCHK(hipMemsetAsync(v->rowmeta, 0, (size_t)v->len * sizeof(uint32_t), v->st));
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  if (getenv("V224HIP_VERBOSE")) {
    fprintf(stderr, "v224hip: metric buffer placement, memory phases of a pass in us:");
    for (int c = 0; c < n; c++) fprintf(stderr, " %.2f%s", t[c] * 1e3f / NPROBE, c == best ? "*" : "");
    fprintf(stderr, "\n");
  }
  return 0;
fail:
  if (a) (void)hipEventDestroy(a);
  if (b) (void)hipEventDestroy(b);
  for (int c = 1; c < NCAND; c++) if (cand[c] && cand[c] != v->m[1]) (void)hipFree(cand[c]);
  return -1;
}

extern "C" void *v224hip_create(int len, int engine, int k) {
  V224 *v = nullptr;
  if (len <= 0) { snprintf(g_err, sizeof g_err, "create: len must be > 0"); return nullptr; }
  {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
      snprintf(g_err, sizeof g_err, "create: no HIP device visible");
      return nullptr;
    }
  }
  v = new V224();
  v->len = len;
  v->engine = engine >= 0 ? engine : env_int("V224HIP_ENGINE", V224HIP_ENGINE_LDS15);
  v->K = k > 0 ? k : env_int("V224HIP_K", FUSED_DEFAULT_K);
  if (v->K < 1) v->K = 1;
  if (v->K > FUSED_MAX_K) v->K = FUSED_MAX_K;
  { int gd = g_device.load(std::memory_order_acquire); v->dev = gd >= 0 ? gd : env_int("V224HIP_DEVICE", 0); }
  // whole passes per chunk (1020 = 68 x 15).  With the tracebacks in the pass stream a chunk costs ~35 us of traceback
  // latency whatever its size, and the ring only has to hold delay + ONE chunk: the default doubles at equal memory.
  v->chunk = env_int("V224HIP_CHUNK", v->engine == V224HIP_ENGINE_LDS ? 1024 : (tb_own_stream() ? 1020 : 2040));
  HIPCHK(hipSetDevice(v->dev));
  if (v->engine == V224HIP_ENGINE_LDS15) {       // 133 KiB of dynamic LDS per workgroup: above the default cap
    // the per-thread parity table of the 15-step kernel, once per device; decoders are created from several threads at
    // once (chain stages, segment workers): the flag and the upload are under one lock
    static std::mutex tab_mu;
    static bool tab_done[64] = {false};
    {
      std::lock_guard<std::mutex> lk(tab_mu);
      if (v->dev >= 0 && v->dev < 64 && !tab_done[v->dev]) {
        std::vector<unsigned> tab(4 * 1024);
        l15_build_rot_tab(tab.data());
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(l15_rot_tab), tab.data(), tab.size() * sizeof(unsigned)));
        tab_done[v->dev] = true;
      }
    }
    HIPCHK(hipFuncSetAttribute((const void *)k_acs_lds15<0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, L15_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void *)k_acs_lds15<0, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, L15_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void *)k_acs_lds15<0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, L15_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void *)k_acs_lds15<1, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, L15_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void *)k_acs_lds15<1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, L15_LDS_BYTES));
  }
  {
    // V224HIP_STREAM_PRIORITY=low|high: the decoder's stream at the lowest / highest priority the device offers (default:
    // normal).  An experiment knob: measured in the chain (decoders low and / or front end high), no effect on any stage.
    const char *pr = getenv("V224HIP_STREAM_PRIORITY");
    int least = 0, greatest = 0;
    if (pr && (pr[0] == 'l' || pr[0] == 'h') && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest < least)
      HIPCHK(hipStreamCreateWithPriority(&v->st, hipStreamNonBlocking, pr[0] == 'l' ? least : greatest));
    else HIPCHK(hipStreamCreateWithFlags(&v->st, hipStreamNonBlocking));
  }
  if (tb_own_stream()) {
    if (getenv("V224HIP_TEST_GAP")) {      // experiment (scratch/pipe_scan.py): GAP other hardware queues created in between
      static hipStream_t dummies[64]; static int nd = 0;
      for (int g = 0; g < atoi(getenv("V224HIP_TEST_GAP")) && nd < 64; g++) HIPCHK(hipStreamCreateWithFlags(&dummies[nd++], hipStreamNonBlocking));
    }
    HIPCHK(hipStreamCreateWithFlags(&v->st2, hipStreamNonBlocking));
  } else v->st2 = v->st;                   // tracebacks in the pass stream
  HIPCHK(hipMalloc(&v->m[0], sizeof(uint16_t) * V224_NSTATES));
  HIPCHK(hipMalloc(&v->m[1], sizeof(uint16_t) * V224_NSTATES));
  HIPCHK(hipMalloc(&v->rows, (size_t)len * V224_ROWWORDS * sizeof(uint32_t)));
  // (the reference mallocs its ring, port.c:58: rows never written are undefined there; here they read as zero)
  HIPCHK(hipMemsetAsync(v->rows, 0, (size_t)len * V224_ROWWORDS * sizeof(uint32_t), v->st));
  HIPCHK(hipMalloc(&v->rowmeta, (size_t)len * sizeof(uint32_t)));
  HIPCHK(hipMemsetAsync(v->rowmeta, 0, (size_t)len * sizeof(uint32_t), v->st));
  HIPCHK(hipMalloc(&v->ds, sizeof(V224Dev)));
  v->dmisc_cap = 1 << 20;
  HIPCHK(hipMalloc(&v->dmisc, v->dmisc_cap));
  HIPCHK(hipMemsetAsync(v->dmisc, 0, 4096, v->st));      // (counters live here: chainback pieces redone at +96)
  for (int i = 0; i < 2; i++) {
    HIPCHK(hipEventCreateWithFlags(&v->ev_acs[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&v->ev_tb[i], hipEventDisableTiming));
  }
  if (v->engine == V224HIP_ENGINE_LDS15 && len >= 15 && env_int("V224HIP_PLACE", 1) && place_metrics(v) != 0) goto fail;
  if (getenv("V224HIP_VERBOSE"))
    fprintf(stderr, "v224hip: create len %d: m0 %p m1 %p rows %p rowmeta %p ds %p dmisc %p\n", len, (void *)v->m[0], (void *)v->m[1],
            (void *)v->rows, (void *)v->rowmeta, (void *)v->ds, (void *)v->dmisc);
  if (init_viterbi224(v, 0) != 0) goto fail;
  return v;
fail:
  delete_viterbi224(v);
  return nullptr;
}

extern "C" void *create_viterbi224(int len) { return v224hip_create(len, -1, 0); }

extern "C" void delete_viterbi224(void *p) {
  V224 *v = (V224 *)p;
  if (!v) return;
  (void)hipSetDevice(v->dev);
  if (v->st) (void)hipStreamSynchronize(v->st);
  if (v->st2 && v->st2 != v->st) (void)hipStreamSynchronize(v->st2);
  for (auto &e : v->ev_busy) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (auto &e : v->ev_free) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (int i = 0; i < 2; i++) {
    if (v->ev_acs[i]) (void)hipEventDestroy(v->ev_acs[i]);
    if (v->ev_tb[i]) (void)hipEventDestroy(v->ev_tb[i]);
  }
  (void)hipFree(v->m[0]); (void)hipFree(v->m[1]); (void)hipFree(v->rows); (void)hipFree(v->rowmeta);
  (void)hipFree(v->ds); (void)hipFree(v->dsyms); (void)hipFree(v->dout); (void)hipFree(v->dmisc);
  (void)hipFree(v->snap[0]); (void)hipFree(v->snap[1]); (void)hipFree(v->warmout);
  for (int i = 0; i < 2; i++) if (v->ev_snap[i]) (void)hipEventDestroy(v->ev_snap[i]);
  if (v->st2 && v->st2 != v->st) (void)hipStreamDestroy(v->st2);
  if (v->st) (void)hipStreamDestroy(v->st);
  delete v;
}

static int init_enqueue(V224 *v, int starting_state, bool wait_tracebacks) {
  HIPCHK(hipSetDevice(v->dev));
  if (wait_tracebacks && v->st2 != v->st) HIPCHK(hipStreamSynchronize(v->st2));        // no traceback may still be reading
  v->cur = 0; v->dp = 0; v->nsteps = 0; v->pass = 0; v->min_valid = true;
  v->fresh = true; v->start = (unsigned)starting_state & V224_SMASK;
  // one kernel, straight in the metric order the engine's first pass wants (a fresh buffer in the other order is a
  // two-element patch away: ensure_layout)
  v->layout = (v->engine == V224HIP_ENGINE_LDS15 && v->len >= 15) ? 1 : 0;
  k_init<<<V224_NSTATES / 8 / 256, 256, 0, v->st>>>(v->m[0], v->layout ? l15_phys(v->start) : v->start, v->ds, v->rowmeta, v->len);
  HIPCHK(hipGetLastError());
  return 0;
fail:
  return -1;
}
extern "C" int init_viterbi224(void *p, int starting_state) {
  V224 *v = (V224 *)p;
  if (!v) return -1;
  return init_enqueue(v, starting_state, true);
}

// ---- profiling helpers -------------------------------------------------------------------
static void prof_harvest(V224 *v, bool all) {
  size_t keep = 0;
  for (size_t i = 0; i < v->ev_busy.size(); i++) {
    EvPair &e = v->ev_busy[i];
    bool done = all ? (hipEventSynchronize(e.b) == hipSuccess) : (hipEventQuery(e.b) == hipSuccess);
    if (done) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
        v->prof_ms += ms; v->prof_launches += e.launches; v->prof_steps += e.steps;
      }
      v->ev_free.push_back(e);
    } else v->ev_busy[keep++] = e;
  }
  v->ev_busy.resize(keep);
}
static bool prof_begin(V224 *v, EvPair *e) {
  if (!v->profile) return false;
  if ((v->launches_seen++ % (unsigned)v->profile) != 0) return false;
  if (v->ev_free.empty()) {
    if (v->ev_busy.size() >= 2048) prof_harvest(v, false);
    if (v->ev_free.empty()) {
      if (v->ev_busy.size() >= 8192) prof_harvest(v, true);
      else { EvPair n; n.steps = 0;
             if (hipEventCreate(&n.a) != hipSuccess || hipEventCreate(&n.b) != hipSuccess) return false;
             v->ev_free.push_back(n); }
    }
  }
  *e = v->ev_free.back(); v->ev_free.pop_back();
  (void)hipEventRecord(e->a, v->st);
  return true;
}
static void prof_end(V224 *v, EvPair *e, unsigned steps, unsigned launches) {
  e->steps = steps; e->launches = launches;
  (void)hipEventRecord(e->b, v->st);
  v->ev_busy.push_back(*e);
}

// ---- ACS dispatch --------------------------------------------------------------------------
// Enqueue nbits trellis steps reading symbols from device memory (2 per step).
static void ensure_min_valid(V224 *v) {
  if (v->min_valid) return;
  k_publish_min<<<512, 256, 0, v->st>>>(v->m[v->cur], v->ds, v->pass);
  v->min_valid = true;
}

// Bring m[cur] into the wanted order (0 natural, 1 L15).  Right after init only the start state differs
// from the fill, so the switch is a two-element patch; otherwise one transposing copy into the other buffer
// (the per-workgroup minima describe the same values before and after).
__global__ void k_move_start(uint16_t *m, unsigned from, unsigned to) {
  m[from] = (uint16_t)(V224_BASE + 1000u); m[to] = (uint16_t)V224_BASE;
}
static void ensure_layout(V224 *v, int want) {
  if (v->layout == want) return;
  if (v->fresh) {
    const unsigned nat = v->start, l15 = l15_phys(v->start);
    k_move_start<<<1, 1, 0, v->st>>>(v->m[v->cur], want ? nat : l15, want ? l15 : nat);
  } else {
    if (want) k_l15_from_nat<<<512, 512, 0, v->st>>>(v->m[v->cur], v->m[v->cur ^ 1]);
    else      k_l15_to_nat<<<512, 512, 0, v->st>>>(v->m[v->cur], v->m[v->cur ^ 1]);
    v->cur ^= 1;
  }
  v->layout = want;
}

static int enqueue_acs(V224 *v, const uint8_t *d_syms, int nbits) {
  int done = 0;
  unsigned nlaunch = 0;
  // profiling brackets the whole run of back-to-back launches with ONE event pair (an event
  // pair around every single launch inflates it by ~3 us of signal handling)
  EvPair ev; const bool timed = prof_begin(v, &ev);
  while (done < nbits) {
    int k = 1;
    if (v->engine == V224HIP_ENGINE_LDS15 && nbits - done >= 15 && v->len >= 15) {
      // 15 steps per launch on the L15 tile order; the pass may wrap the ring.  Minimum tracking alternates
      // (adjust, no publish) / (no adjust, publish) as below; the last full pass of a run publishes.
      ensure_layout(v, 1);
      const bool last = (nbits - done - 15) < 15;
      const bool tin = v->min_valid, tout = !tin || last;
#define LDS15_GO(TIN, TOUT) k_acs_lds15<0, TIN, TOUT><<<256, 1024, L15_LDS_BYTES, v->st>>>(v->m[v->cur], v->m[v->cur ^ 1], \
                               v->rows, v->dp, v->len, d_syms + 2 * done, v->ds, v->pass, v->rowmeta)
      if (tin && tout) LDS15_GO(true, true); else if (tin) LDS15_GO(true, false); else LDS15_GO(false, true);
#undef LDS15_GO
      v->min_valid = tout;
      v->fresh = false;
      nlaunch++;
      v->cur ^= 1; v->pass++;
      v->dp = (v->dp + 15) % v->len;
      v->nsteps += 15u;
      done += 15;
      continue;
    }
    ensure_layout(v, 0);
    v->fresh = false;
    const bool lds8 = v->engine == V224HIP_ENGINE_LDS || v->engine == V224HIP_ENGINE_LDS15;
    if (v->engine != V224HIP_ENGINE_SIMPLE) {
      k = lds8 ? 8 : v->K;
      if (k > nbits - done) k = nbits - done;
      if (k > v->len - v->dp) k = v->len - v->dp;      // a pass never wraps the ring
    }
    if (lds8 && k == 8) {
      // minimum tracking is split over two launches where possible: (adjust, no publish) then
      // (no adjust, publish).  The last launch of a run always publishes, so every other kernel
      // (remainder passes, traceback helpers, the next call) finds valid minima.
      const bool last = (nbits - done - 8) < 8;      // next pass is a remainder pass or none
      const bool tin = v->min_valid, tout = !tin || last;
#define LDS8_GO(TIN, TOUT) k_acs_lds8<0, TIN, TOUT><<<512, 512, 0, v->st>>>(v->m[v->cur], v->m[v->cur ^ 1], v->rows, v->dp, \
                                                   d_syms + 2 * done, v->ds, v->pass, v->rowmeta)
      if (tin && tout) LDS8_GO(true, true); else if (tin) LDS8_GO(true, false); else LDS8_GO(false, true);
#undef LDS8_GO
      v->min_valid = tout;
    } else if (v->engine != V224HIP_ENGINE_SIMPLE) {
      ensure_min_valid(v);
      if (fused_launch(k, v->m[v->cur], v->m[v->cur ^ 1], v->rows, v->dp, d_syms + 2 * done, v->ds,
                       v->pass, v->rowmeta, v->st) != 0) {
        snprintf(g_err, sizeof g_err, "fused_launch(k=%d) failed", k);
        return -1;
      }
    } else {
      ensure_min_valid(v);
      k_acs_simple<<<V224_NSTATES / 32 / 256, 256, 0, v->st>>>(
          v->m[v->cur], v->m[v->cur ^ 1], v->rows + (size_t)v->dp * V224_ROWWORDS,
          d_syms + 2 * done, v->ds, v->pass, v->rowmeta, v->dp);
    }
    nlaunch++;
    v->cur ^= 1; v->pass++;
    v->dp += k; if (v->dp >= v->len) v->dp = 0;
    v->nsteps += (unsigned)k;
    done += k;
  }
  if (timed) prof_end(v, &ev, (unsigned)nbits, nlaunch);
  if (hipGetLastError() != hipSuccess) { snprintf(g_err, sizeof g_err, "ACS launch failed"); return -1; }
  return 0;
}

static int ensure_cap(uint8_t **buf, size_t *cap, size_t need) {
  if (*cap >= need) return 0;
  if (*buf) (void)hipFree(*buf);
  *buf = nullptr; *cap = 0;
  size_t n = need < 4096 ? 4096 : need;
  if (hipMalloc(buf, n) != hipSuccess) return -1;
  *cap = n;
  return 0;
}

extern "C" int update_viterbi224_blk(void *p, const unsigned char *syms, int nbits) {
  V224 *v = (V224 *)p;
  if (!v) return -1;
  if (nbits <= 0) return 0;
  HIPCHK(hipSetDevice(v->dev));
  {
    // Symbols of launches that are still queued must stay where they are.  Calls bump-allocate from a staging
    // buffer (at least 64 KiB: 16 384 of the reference's one-bit calls, vdecode.c:145) and the stream is only
    // drained when the buffer wraps or has to grow -- not on every call.
    const size_t need = (2 * (size_t)nbits + 3) & ~(size_t)3;
    if (v->dsyms_off + need > v->dsyms_cap) {
      HIPCHK(hipStreamSynchronize(v->st));
      if (ensure_cap(&v->dsyms, &v->dsyms_cap, need < (64u << 10) ? (64u << 10) : need) != 0) {
        snprintf(g_err, sizeof g_err, "update: device symbol buffer allocation failed");
        return -1;
      }
      v->dsyms_off = 0;
    }
    uint8_t *dst = v->dsyms + v->dsyms_off;
    v->dsyms_off += need;
    HIPCHK(hipMemcpyAsync(dst, syms, 2 * (size_t)nbits, hipMemcpyHostToDevice, v->st));
    if (enqueue_acs(v, dst, nbits) != 0) return -1;
  }
  return 0;                                          // port convention (port.c:194)
fail:
  return -1;
}

extern "C" int v224hip_update_dev(void *p, const uint8_t *d_syms, int nbits) {
  V224 *v = (V224 *)p;
  if (!v) return -1;
  if (nbits <= 0) return 0;
  if (hipSetDevice(v->dev) != hipSuccess) return -1;
  return enqueue_acs(v, d_syms, nbits);
}

static int best_state(V224 *v, unsigned *state) {
  unsigned *d = (unsigned *)v->dmisc;
  ensure_layout(v, 0);
  ensure_min_valid(v);
  HIPCHK(hipMemsetAsync(d, 0xff, sizeof(unsigned), v->st));
  k_argmin<<<V224_NSTATES / 8 / 256, 256, 0, v->st>>>(v->m[v->cur], v->ds, v->pass, d);
  HIPCHK(hipMemcpyAsync(state, d, sizeof(unsigned), hipMemcpyDeviceToHost, v->st));
  HIPCHK(hipStreamSynchronize(v->st));
  return 0;
fail:
  return -1;
}

extern "C" int decodebit_viterbi224(void *p, int delay, int endstate) {
  V224 *v = (V224 *)p;
  if (!v) return -1;
  if (delay <= 0) return -1;                           // port.c:128-129: bit stays -1
  uint8_t bit = 0;
  unsigned st = (unsigned)endstate;
  HIPCHK(hipSetDevice(v->dev));
  if (endstate < 0 && best_state(v, &st) != 0) return -1;
  // like the port, read whatever the ring holds
  if (getenv("V224HIP_SERIAL_CHAINBACK")) k_decodebits<<<1, 64, 0, v->st>>>(v->rows, v->rowmeta, v->len, v->dp, 1ull << 40, 1, delay, st, v->dmisc + 64);
  else k_decodebit_spec<<<1, 64, 0, v->st>>>(v->rows, v->rowmeta, v->len, v->dp, delay, st, v->dmisc + 64);
  HIPCHK(hipMemcpyAsync(&bit, v->dmisc + 64, 1, hipMemcpyDeviceToHost, v->st));
  HIPCHK(hipStreamSynchronize(v->st));
  return bit;
fail:
  return -1;
}

extern "C" unsigned long long decodeword_viterbi224(void *p, int delay, int endstate) {
  V224 *v = (V224 *)p;
  unsigned long long r = 0;
  unsigned st = (unsigned)endstate & 0xffffffu;        // sse2.c:228
  if (!v) return 0;
  HIPCHK(hipSetDevice(v->dev));
  if (endstate < 0 && best_state(v, &st) != 0) return 0;
  k_decodeword<<<1, 1, 0, v->st>>>(v->rows, v->rowmeta, v->len, v->dp, delay, st,
                                   (unsigned long long *)(v->dmisc + 128));
  HIPCHK(hipMemcpyAsync(&r, v->dmisc + 128, sizeof r, hipMemcpyDeviceToHost, v->st));
  HIPCHK(hipStreamSynchronize(v->st));
  return r;
fail:
  return 0;
}

// framed chainback on stream st: 16 verified pieces at once where the frame is long enough, else the serial walk
static void launch_chainback(V224 *v, hipStream_t st, unsigned nbits, unsigned endstate, uint8_t *d_out, unsigned row0) {
  const int serial = env_int("V224HIP_SERIAL_CHAINBACK", 0);   // 1: one lane, 2: one wave (read per call: tests switch it)
  if (serial == 1 && row0 == 0) k_chainback<<<1, 64, 0, st>>>(v->rows, v->rowmeta, v->len, nbits, endstate, d_out);
  else if (serial || nbits < CB_MINBITS || nbits > 8u * CB_MAXBYTES)
    k_chainback_spec<<<1, 64, 0, st>>>(v->rows, v->rowmeta, v->len, nbits, endstate, d_out, row0);
  else
    k_chainback_par<<<1, CB_WAVES * 64, 0, st>>>(v->rows, v->rowmeta, v->len, nbits, endstate, d_out, row0, (unsigned *)(v->dmisc + 96),
                                                 env_int("V224HIP_CB_WARM", CB_WARM));
}

extern "C" int chainback_viterbi224(void *p, unsigned char *data, unsigned int nbits,
                                    unsigned int endstate) {
  V224 *v = (V224 *)p;
  if (!v) return -1;
  if (nbits == 0) return 0;
  {
    size_t nbytes = (nbits + 7) / 8;
    HIPCHK(hipSetDevice(v->dev));
    if (ensure_cap(&v->dout, &v->dout_cap, nbytes) != 0) return -1;
    launch_chainback(v, v->st, nbits, endstate, v->dout, 0);
    // the port writes data[n>>3] only where (n & 7) == 0, i.e. nbits/8 bytes (+1 if ragged)
    HIPCHK(hipMemcpyAsync(data, v->dout, nbytes, hipMemcpyDeviceToHost, v->st));
    HIPCHK(hipStreamSynchronize(v->st));
  }
  return 0;
fail:
  return -1;
}

// ---- batch of independent frames ------------------------------------------------------------
// Frame f goes to decoder f % ndec; every decoder works on its own stream, so with two decoders one frame's
// passes run under the other's (two LDS15 workgroups fit a CU) and a frame's traceback runs under the next
// frame's ACS.  Nothing synchronises with the host until all frames are enqueued.
extern "C" int v224hip_decode_frames(void *const *decoders, int ndec, const uint8_t *syms, int nframes,
                                     int framebits, int startstate, unsigned int endstate, uint8_t *out) {
  uint8_t *d_syms = nullptr, *d_out = nullptr;       // = decoder 0's staging buffers
  int rc = -1;
  if (!decoders || ndec <= 0 || nframes < 0 || framebits <= 0 || !syms || !out) {
    snprintf(g_err, sizeof g_err, "decode_frames: bad argument");
    return -1;
  }
  if (nframes == 0) return 0;
  {
    const size_t symbytes = 2 * (size_t)framebits, outbytes = ((size_t)framebits + 7) / 8;
    V224 *v0 = (V224 *)decoders[0];
    for (int i = 0; i < ndec; i++) {
      V224 *v = (V224 *)decoders[i];
      if (!v || v->len < framebits || v->dev != v0->dev) {
        snprintf(g_err, sizeof g_err, "decode_frames: decoder %d is NULL, shorter than a frame or on another device", i);
        return -1;
      }
    }
    HIPCHK(hipSetDevice(v0->dev));
    // Two things make a batch cheaper when the rings are long enough (len >= 2 * padded frame):
    //  * a frame is padded with erasures to whole 15-step passes -- steps after the frame's last one cannot change the
    //    decisions before it, and the traceback starts at the frame's own last row -- so no remainder pass and no
    //    switch of the metric order;
    //  * frames of one decoder alternate between the two halves of its ring, and a frame's traceback runs on the
    //    decoder's second stream under the next frame's passes.
    bool dual = v0->st2 != v0->st;        // ring halves only pay with an own traceback stream
    const int padbits = v0->engine == V224HIP_ENGINE_LDS15 ? (framebits + 14) / 15 * 15 : framebits;
    for (int i = 0; i < ndec; i++) {
      V224 *v = (V224 *)decoders[i];
      if (v->len < 2 * padbits || v->engine != v0->engine || v->st2 == v->st) dual = false;
    }
    bool pad = padbits != framebits;
    for (int i = 0; i < ndec; i++) if (((V224 *)decoders[i])->len < padbits || ((V224 *)decoders[i])->engine != v0->engine) pad = false;
    const int runbits = (dual || pad) ? padbits : framebits;
    const size_t stride = 2 * (size_t)runbits;
    // staging lives in decoder 0 (grow-only, kept between calls): no hipMalloc / hipFree per batch
    for (int i = 0; i < ndec; i++) { HIPCHK(hipStreamSynchronize(((V224 *)decoders[i])->st)); HIPCHK(hipStreamSynchronize(((V224 *)decoders[i])->st2)); }
    if (ensure_cap(&v0->dsyms, &v0->dsyms_cap, stride * nframes) != 0 || ensure_cap(&v0->dout, &v0->dout_cap, outbytes * nframes) != 0) {
      snprintf(g_err, sizeof g_err, "decode_frames: device staging allocation failed");
      return -1;
    }
    v0->dsyms_off = v0->dsyms_cap;
    d_syms = v0->dsyms; d_out = v0->dout;
    if (stride != symbytes) HIPCHK(hipMemsetAsync(d_syms, 128, stride * nframes, v0->st));       // erasures behind every frame
    HIPCHK(hipMemcpy2DAsync(d_syms, stride, syms, symbytes, symbytes, (size_t)nframes, hipMemcpyHostToDevice, v0->st));
    HIPCHK(hipEventRecord(v0->ev_acs[0], v0->st));
    for (int i = 1; i < ndec; i++) HIPCHK(hipStreamWaitEvent(((V224 *)decoders[i])->st, v0->ev_acs[0], 0));
    for (int f = 0; f < nframes; f++) {
      V224 *v = (V224 *)decoders[f % ndec];
      const int h = dual ? (f / ndec) & 1 : 0;
      if (dual && f / ndec >= 2) HIPCHK(hipStreamWaitEvent(v->st, v->ev_tb[h], 0));   // this half's previous traceback is done
      if (init_enqueue(v, startstate, !dual) != 0) goto fail;
      v->dp = h * padbits;
      if (enqueue_acs(v, d_syms + stride * f, runbits) != 0) goto fail;
      if (dual) {
        HIPCHK(hipEventRecord(v->ev_acs[h], v->st));
        HIPCHK(hipStreamWaitEvent(v->st2, v->ev_acs[h], 0));
        launch_chainback(v, v->st2, (unsigned)framebits, endstate, d_out + outbytes * f, (unsigned)(h * padbits));
        HIPCHK(hipEventRecord(v->ev_tb[h], v->st2));
      } else if (!getenv("V224HIP_FRAMES_NO_TB"))          // (measurement hook: passes only)
        launch_chainback(v, v->st, (unsigned)framebits, endstate, d_out + outbytes * f, 0);
    }
    HIPCHK(hipGetLastError());
    for (int i = 0; i < ndec; i++) {
      V224 *v = (V224 *)decoders[i];
      if (i > 0) HIPCHK(hipStreamSynchronize(v->st));
      HIPCHK(hipStreamSynchronize(v->st2));
      v->dp = 0;                          // the object is left as after a plain init + update: callers re-init anyway
    }
    HIPCHK(hipMemcpyAsync(out, d_out, outbytes * nframes, hipMemcpyDeviceToHost, v0->st));
    HIPCHK(hipStreamSynchronize(v0->st));
  }
  rc = 0;
fail:
  if (rc != 0) for (int i = 0; i < ndec; i++) if (decoders[i]) { (void)hipStreamSynchronize(((V224 *)decoders[i])->st); (void)hipStreamSynchronize(((V224 *)decoders[i])->st2); }
  return rc;
}

static int metric_extreme(V224 *v, int want_max, long long *out) {
  unsigned *d = (unsigned *)(v->dmisc + 256);
  unsigned h[2]; long long off;
  HIPCHK(hipSetDevice(v->dev));
  ensure_min_valid(v);
  HIPCHK(hipMemsetAsync(d, 0, sizeof(unsigned), v->st));
  if (want_max) k_max<<<V224_NSTATES / 8 / 256, 256, 0, v->st>>>(v->m[v->cur], d);
  k_cur_min<<<1, 256, 0, v->st>>>(v->ds, v->pass, d + 1);
  HIPCHK(hipMemcpyAsync(&h[0], d, sizeof(unsigned), hipMemcpyDeviceToHost, v->st));
  HIPCHK(hipMemcpyAsync(&h[1], d + 1, sizeof(unsigned), hipMemcpyDeviceToHost, v->st));
  HIPCHK(hipMemcpyAsync(&off, &v->ds->off, sizeof off, hipMemcpyDeviceToHost, v->st));
  HIPCHK(hipStreamSynchronize(v->st));
  *out = (long long)(want_max ? h[0] : h[1]) + off;
  return 0;
fail:
  return -1;
}
extern "C" int max_metric_viterbi224(void *p) {
  long long r;
  if (!p || metric_extreme((V224 *)p, 1, &r) != 0) return -1;
  return (int)r;
}
extern "C" int min_metric_viterbi224(void *p) {
  long long r;
  if (!p || metric_extreme((V224 *)p, 0, &r) != 0) return -1;
  return (int)r;
}

// ---- streaming block decode -------------------------------------------------------------
extern "C" int v224hip_stream_chunk(void *p) { return p ? ((V224 *)p)->chunk : -1; }

// ACS of chunk c runs on st; its tracebacks run on st2 while chunk c+1's ACS proceeds.
// Ring safety: chunk c+2's ACS (which may overwrite rows chunk c's traceback still reads once
// len < delay + 3*chunk) waits for chunk c's traceback event.
extern "C" int v224hip_stream_decode_dev(void *p, const uint8_t *d_syms, int nbits, int delay,
                                         uint8_t *d_out) {
  V224 *v = (V224 *)p;
  if (!v) return -1;
  if (nbits <= 0) return 0;
  if (delay <= 0 || v->len < delay + (v->st2 == v->st ? 1 : 2) * v->chunk) {
    snprintf(g_err, sizeof g_err, "stream_decode: need len >= delay + %d*chunk (len=%d delay=%d chunk=%d)",
             v->st2 == v->st ? 1 : 2, v->len, delay, v->chunk);
    return -1;
  }
  HIPCHK(hipSetDevice(v->dev));
  if (v->st2 == v->st) {
    // one stream: the chunk's passes, then its tracebacks (wave per bit), then the next chunk
    for (int done = 0; done < nbits;) {
      int n = nbits - done < v->chunk ? nbits - done : v->chunk;
      int dp_first = (v->dp + 1) % v->len;            // ring index after the chunk's first step
      unsigned long long steps_first = v->nsteps + 1;
      if (enqueue_acs(v, d_syms + 2 * (size_t)done, n) != 0) return -1;
      k_decodebits_spec<<<(n + 3) / 4, 256, 0, v->st>>>(v->rows, v->rowmeta, v->len, dp_first, steps_first, n, delay, 0u, d_out + done);
      done += n;
    }
    HIPCHK(hipGetLastError());
    return 0;
  }
  {
    int c = 0;
    for (int done = 0; done < nbits; c++) {
      int n = nbits - done < v->chunk ? nbits - done : v->chunk;
      if (c >= 2) HIPCHK(hipStreamWaitEvent(v->st, v->ev_tb[c & 1], 0));
      int dp_first = (v->dp + 1) % v->len;            // ring index after the chunk's first step
      unsigned long long steps_first = v->nsteps + 1;
      if (enqueue_acs(v, d_syms + 2 * (size_t)done, n) != 0) return -1;
      HIPCHK(hipEventRecord(v->ev_acs[c & 1], v->st));
      HIPCHK(hipStreamWaitEvent(v->st2, v->ev_acs[c & 1], 0));
      k_decodebits<<<(n + 63) / 64, 64, 0, v->st2>>>(v->rows, v->rowmeta, v->len, dp_first, steps_first,
                                                    n, delay, 0u, d_out + done);
      HIPCHK(hipEventRecord(v->ev_tb[c & 1], v->st2));
      done += n;
    }
    HIPCHK(hipGetLastError());
    // leave st ordered after the last tracebacks so that later API calls see a quiet ring
    HIPCHK(hipStreamWaitEvent(v->st, v->ev_tb[(c - 1) & 1], 0));
    if (c >= 2) HIPCHK(hipStreamWaitEvent(v->st, v->ev_tb[c & 1], 0));
  }
  return 0;
fail:
  return -1;
}

// ---- one stream over several decoders, verified -------------------------------------------------------
// dst[i] = m[i] - (minimum of m): what two decoders must agree on for their futures to be identical
__global__ __launch_bounds__(256) void k_snapshot_rel(const uint16_t *__restrict__ m, const V224Dev *ds, unsigned pass,
                                                      uint16_t *__restrict__ dst) {
  const unsigned mn = input_min(ds, pass);
  const unsigned t = blockIdx.x * 256 + threadIdx.x;           // 2^20 threads x 8 states
  uint4 v = reinterpret_cast<const uint4 *>(m)[t];
  const unsigned sub = (mn & 0xffffu) * 0x10001u;
  v.x = as_u32(as_v2(v.x) - as_v2(sub)); v.y = as_u32(as_v2(v.y) - as_v2(sub));
  v.z = as_u32(as_v2(v.z) - as_v2(sub)); v.w = as_u32(as_v2(v.w) - as_v2(sub));
  reinterpret_cast<uint4 *>(dst)[t] = v;
}
__global__ __launch_bounds__(256) void k_count_diff(const uint4 *__restrict__ a, const uint4 *__restrict__ b, unsigned *count) {
  const unsigned t = blockIdx.x * 256 + threadIdx.x;
  const uint4 x = a[t], y = b[t];
  const bool d = (x.x != y.x) | (x.y != y.y) | (x.z != y.z) | (x.w != y.w);
  if (__syncthreads_or(d) && threadIdx.x == 0) atomicAdd(count, 1u);
}

struct SplitItem { long long bit0; long long nbits; uint8_t *out; int snap; };   // snap: 0 none, else seam index (+ = as A, - = as B)

// cont: decoder 0 is in the middle of a stream (no init; the block continues it); the other decoders start fresh inside the
// block.  *holder = the decoder that stands at the end of the block with the stream's state afterwards.
static int split_core(void *const *decoders, int ndec, const uint8_t *d_syms, int nbits, int delay,
                      uint8_t *d_out, int warm_bits, bool cont, int *nfallback, int *holder) {
  unsigned *d_cnt = nullptr, h_cnt[8] = {0};
  int rc = -1, fb = 0;
  if (nfallback) *nfallback = 0;
  if (holder) *holder = 0;
  if (!decoders || ndec < 1 || ndec > 8 || !d_syms || !d_out || nbits < 0 || delay <= 0) {
    snprintf(g_err, sizeof g_err, "stream_decode_split: bad argument");
    return -1;
  }
  V224 *v0 = (V224 *)decoders[0];
  for (int j = 0; j < ndec; j++) {
    V224 *v = (V224 *)decoders[j];
    if (!v || v->dev != v0->dev || v->chunk != v0->chunk || v->len < delay + (v->st2 == v->st ? 1 : 2) * v->chunk) {
      snprintf(g_err, sizeof g_err, "stream_decode_split: decoder %d is NULL, on another device, has another chunk size or a short ring", j);
      return -1;
    }
  }
  // The seam window.  Decoder j's metrics are compared with decoder j-1's after trellis step start[j] - check; equal
  // (up to a constant) metrics make every decision row written AFTER that step identical in both.  The first
  // traceback of part j walks `delay` rows back from step start[j] + 1, so all rows it reads are such rows iff
  // check >= delay - 1: check = delay rounded up to whole chunks.  (The reference allows any -d, vdecode.c:86-91.)
  const long long chunk = v0->chunk, check = ((long long)delay + chunk - 1) / chunk * chunk;
  // warm = check + the bits given to forgetting the fresh start (at least two chunks); `warm_bits` counts both, as the
  // header says, and is raised when a long delay leaves too little of it for forgetting
  long long warm = ((long long)warm_bits + chunk - 1) / chunk * chunk;
  if (warm < check + 2 * chunk) warm = check + 2 * chunk;
  // parts: equal finishing times => part 0 is `warm` longer than the others' own share
  int P = ndec;
  while (P > 1 && ((long long)nbits + (P - 1) * warm) / P / chunk * chunk < warm + chunk) P--;     // too short to split that far
  long long start[9];
  {
    const long long per = P > 1 ? ((long long)nbits + (P - 1) * warm) / P / chunk * chunk : nbits;   // bits each decoder processes
    start[0] = 0;
    for (int j = 1; j < P; j++) start[j] = per + (long long)(j - 1) * (per - warm);
    start[P] = nbits;
  }
  HIPCHK(hipSetDevice(v0->dev));
  // seam resources live in the decoder objects (allocated on first use, kept until delete): decoder j holds the
  // snapshot it takes as the LATER side ("B") of seam j in snap[1] and as the EARLIER side ("A") of seam j+1 in snap[0]
  for (int j = 0; j < P && P > 1; j++) {
    V224 *v = (V224 *)decoders[j];
    for (int w = 0; w < 2; w++) {
      if ((w == 0 && j == P - 1) || (w == 1 && j == 0)) continue;
      if (!v->snap[w]) HIPCHK(hipMalloc(&v->snap[w], sizeof(uint16_t) * V224_NSTATES));
      if (!v->ev_snap[w]) HIPCHK(hipEventCreateWithFlags(&v->ev_snap[w], hipEventDisableTiming));
    }
    if (j > 0 && ensure_cap(&v->warmout, &v->warmout_cap, (size_t)warm) != 0) {
      snprintf(g_err, sizeof g_err, "stream_decode_split: warm-up buffer allocation failed");
      return -1;
    }
  }
  if (P > 1) {
    d_cnt = (unsigned *)(v0->dmisc + 1024);
    HIPCHK(hipMemsetAsync(d_cnt, 0, 8 * sizeof(unsigned), v0->st));
  }
  {
    // work lists: decoder j warms up from start[j] - warm, snapshots at the checkpoint start[j] - check (as B of seam j),
    // decodes its part, and snapshots at start[j+1] - check on the way (as A of seam j+1)
    std::vector<SplitItem> work[8];
    for (int j = 0; j < P; j++) {
      V224 *v = (V224 *)decoders[j];
      if (j > 0) {
        work[j].push_back({start[j] - warm, warm - check, v->warmout, -j});
        work[j].push_back({start[j] - check, check, v->warmout + (warm - check), 0});
      }
      if (j < P - 1) {
        work[j].push_back({start[j], start[j + 1] - check - start[j], d_out + start[j], j + 1});
        work[j].push_back({start[j + 1] - check, check, d_out + start[j + 1] - check, 0});
      } else work[j].push_back({start[j], start[j + 1] - start[j], d_out + start[j], 0});
    }
    size_t it[8] = {0}; long long done[8] = {0};
    const long long slab = 8 * chunk;
    for (int j = cont ? 1 : 0; j < P; j++) if (init_viterbi224(decoders[j], 0) != 0) goto fail;
    for (bool any = true; any;) {                       // slab by slab, round robin, so that the decoders' launches interleave
      any = false;
      for (int j = 0; j < P; j++) {
        if (it[j] >= work[j].size()) continue;
        any = true;
        V224 *v = (V224 *)decoders[j];
        SplitItem &w = work[j][it[j]];
        const long long n = w.nbits - done[j] < slab ? w.nbits - done[j] : slab;
        if (n > 0 && v224hip_stream_decode_dev(v, d_syms + 2 * (w.bit0 + done[j]), (int)n, delay, w.out + done[j]) != 0) goto fail;
        done[j] += n;
        if (done[j] >= w.nbits) {
          if (w.snap != 0) {
            ensure_layout(v, 1);                       // both sides of a seam are compared in the same (tile) order
            ensure_min_valid(v);
            const int side = w.snap > 0 ? 0 : 1;
            k_snapshot_rel<<<V224_NSTATES / 8 / 256, 256, 0, v->st>>>(v->m[v->cur], v->ds, v->pass, v->snap[side]);
            HIPCHK(hipEventRecord(v->ev_snap[side], v->st));
          }
          it[j]++; done[j] = 0;
        }
      }
    }
    for (int j = 1; j < P; j++) {                       // all snapshots are enqueued: compare
      V224 *va = (V224 *)decoders[j - 1], *vb = (V224 *)decoders[j];
      HIPCHK(hipStreamWaitEvent(v0->st, va->ev_snap[0], 0));
      HIPCHK(hipStreamWaitEvent(v0->st, vb->ev_snap[1], 0));
      k_count_diff<<<V224_NSTATES / 8 / 256, 256, 0, v0->st>>>(reinterpret_cast<const uint4 *>(va->snap[0]),
                                                                reinterpret_cast<const uint4 *>(vb->snap[1]), d_cnt + j);
    }
    HIPCHK(hipGetLastError());
    if (P > 1) HIPCHK(hipMemcpyAsync(h_cnt, d_cnt, sizeof h_cnt, hipMemcpyDeviceToHost, v0->st));
    for (int j = 0; j < P; j++) HIPCHK(hipStreamSynchronize(((V224 *)decoders[j])->st));
    if (P > 1 && getenv("V224HIP_SPLIT_FORCE_FALLBACK")) h_cnt[P - 1] = 1;         // test hook: exercise the redo path
    for (int j = 1; j < P; j++) {
      if (h_cnt[j] == 0) continue;
      // the warm-up of part j had not forgotten its start: everything from start[j] on is decoded again by the
      // decoder of part j-1, which stands exactly there with the true path metrics
      fb = P - j;
      if (v224hip_stream_decode_dev(decoders[j - 1], d_syms + 2 * start[j], (int)(nbits - start[j]), delay, d_out + start[j]) != 0) goto fail;
      HIPCHK(hipStreamSynchronize(((V224 *)decoders[j - 1])->st));
      break;
    }
    if (holder) *holder = fb ? P - fb - 1 : P - 1;
  }
  if (nfallback) *nfallback = fb;
  rc = 0;
fail:
  if (rc != 0) for (int j = 0; j < ndec; j++) if (decoders[j]) (void)hipStreamSynchronize(((V224 *)decoders[j])->st);
  return rc;
}

extern "C" int v224hip_stream_decode_split(void *const *decoders, int ndec, const uint8_t *d_syms, int nbits, int delay,
                                           uint8_t *d_out, int warm_bits, int *nfallback) {
  return split_core(decoders, ndec, d_syms, nbits, delay, d_out, warm_bits, false, nfallback, nullptr);
}

// The next block of a stream that decoders[*holder] is in the middle of (host buffers).  A long block is shared with the
// other decoders exactly as v224hip_stream_decode_split shares a whole stream -- they start fresh inside the block, the
// seams are verified -- and *holder moves to the decoder that stands at the block's end; a short block simply continues on
// the holder.  out[] is what v224hip_stream_decode() of ONE decoder fed with the same blocks would write.
extern "C" int v224hip_stream_decode_shared(void *const *decoders, int ndec, int *holder, const uint8_t *syms, int nbits,
                                            int delay, uint8_t *out, int warm_bits) {
  if (!decoders || ndec < 1 || ndec > 8 || !holder || *holder < 0 || *holder >= ndec || !syms || !out || nbits < 0) {
    snprintf(g_err, sizeof g_err, "stream_decode_shared: bad argument");
    return -1;
  }
  if (nbits == 0) return 0;
  V224 *v0 = (V224 *)decoders[*holder];
  if (!v0) return -1;
  const long long chunk = v0->chunk;
  long long warm = ((long long)warm_bits + chunk - 1) / chunk * chunk;
  const long long check = ((long long)delay + chunk - 1) / chunk * chunk;
  if (warm < check + 2 * chunk) warm = check + 2 * chunk;
  // worth sharing?  one decoder: n steps at 18 us per 15; two: (n + warm) / 2 steps each at the pair rate (23 us per 15) + the
  // seam (two 16 MiB snapshots, one compare, ~0.15 ms): break-even near n = 2 warm
  if (ndec < 2 || 5LL * nbits < 11 * warm) return v224hip_stream_decode(v0, syms, nbits, delay, out);
  {
    void *order[8];
    int k = 0, redone = 0, h = 0;
    order[k++] = decoders[*holder];
    for (int i = 0; i < ndec && k < 2; i++) if (i != *holder) order[k++] = decoders[i];      // two decoders: all a CU can hold
    HIPCHK(hipSetDevice(v0->dev));
    HIPCHK(hipStreamSynchronize(v0->st));
    if (ensure_cap(&v0->dsyms, &v0->dsyms_cap, 2 * (size_t)nbits) != 0 || ensure_cap(&v0->dout, &v0->dout_cap, (size_t)nbits) != 0) return -1;
    v0->dsyms_off = v0->dsyms_cap;
    HIPCHK(hipMemcpyAsync(v0->dsyms, syms, 2 * (size_t)nbits, hipMemcpyHostToDevice, v0->st));
    HIPCHK(hipStreamSynchronize(v0->st));                   // the other decoder's stream reads the symbols too
    if (split_core(order, k, v0->dsyms, nbits, delay, v0->dout, (int)warm, true, &redone, &h) != 0) return -1;
    HIPCHK(hipMemcpy(out, v0->dout, (size_t)nbits, hipMemcpyDeviceToHost));
    for (int i = 0; i < ndec; i++) if (decoders[i] == order[h]) *holder = i;
  }
  return 0;
fail:
  return -1;
}

// ---- a stream that is still arriving, on two decoders, exactly ---------------------------------------------------
// The verified split (split_core) needs the whole stream up front; v224hip_stream_decode_shared pays one warm-up per
// block.  Here the cut is PLANNED for an expected length when the stream begins: decoder 0 decodes [0, start1) as the
// symbols come in, decoder 1 starts from a fresh init at start1 - warm as soon as the symbols reach that far and runs to
// the end -- one warm-up for the whole stream, both decoders busy from then on.  The seam is verified exactly as in
// split_core (relative metrics of all 2^23 states equal at start1 - check), with the same fall-back (decoder 0 simply
// goes on past start1).  The expected length only places the cut: a shorter stream may never reach decoder 1's part, a
// longer one makes its part longer.
struct V224Prog {
  V224 *d[2];
  int ndec, delay;
  long long chunk, check, warm, expected;
  long long start1;                        // the cut; < 0 until it is fixed (and for good when can_split is false)
  bool can_split;
  uint8_t *d_syms, *d_out;
  long long cap;                           // bits the device buffers hold
  long long avail, a_pos, b_pos;           // bits uploaded; next bit of decoder 0 / of decoder 1 (b_pos starts at start1 - warm)
  bool b_started, a_snapped, b_snapped;
  // how far each decoder has really got (not just been told to go): events recorded behind its calls, polled oldest first
  enum { NEV = 32 };
  struct Track { hipEvent_t ev[NEV]; long long pos[NEV]; int head, count; long long done; } tr[2];
  double t_begin, t_cut, a_done_at_cut, avail_at_cut;      // V224HIP_VERBOSE report
};
static double prog_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
#define PROG_LOOK_PRE 6    /* chunks decoder 0 is told ahead of what it has finished while the cut is still open */
#define PROG_LOOK 24       /* ... and either decoder afterwards: feed() hands out work for some tens of ms, never the whole backlog */
#define PROG_SLAB 8        /* chunks per call when both decoders have a backlog: their launches must interleave in the queues */

// share of decoder 0's progress that decoder 1 makes while the symbols' producer still runs on the GPU ($V224HIP_PROG_GAMMA)
static double prog_gamma(void) {
  static const double gm = getenv("V224HIP_PROG_GAMMA") ? atof(getenv("V224HIP_PROG_GAMMA")) : 0.33;
  return gm < -3 ? -3 : gm > 1 ? 1 : gm;
}
static void prog_poll(V224Prog *g, int j) {
  V224Prog::Track &t = g->tr[j];
  while (t.count > 0 && hipEventQuery(t.ev[t.head]) == hipSuccess) {
    t.done = t.pos[t.head];
    t.head = (t.head + 1) % V224Prog::NEV; t.count--;
  }
  (void)hipGetLastError();                 // hipErrorNotReady is not an error here
}
static void prog_mark(V224Prog *g, int j, long long pos) {
  V224Prog::Track &t = g->tr[j];
  if (t.count >= V224Prog::NEV) return;                       // all in flight: the next poll will see an older one first
  const int i = (t.head + t.count) % V224Prog::NEV;
  if (!t.ev[i] && hipEventCreateWithFlags(&t.ev[i], hipEventDisableTiming) != hipSuccess) { t.ev[i] = nullptr; return; }
  if (hipEventRecord(t.ev[i], g->d[j]->st) != hipSuccess) return;
  t.pos[i] = pos; t.count++;
}

static int prog_grow(V224Prog *g, long long need) {
  if (need <= g->cap) return 0;
  long long ncap = g->cap * 2 > need ? g->cap * 2 : need + 65536;
  uint8_t *ns = nullptr, *no = nullptr;
  for (int j = 0; j < g->ndec; j++) if (hipStreamSynchronize(g->d[j]->st) != hipSuccess) return -1;
  if (hipMalloc(&ns, 2 * (size_t)ncap) != hipSuccess || hipMalloc(&no, (size_t)ncap) != hipSuccess) { (void)hipFree(ns); (void)hipGetLastError(); return -1; }
  if (hipMemcpy(ns, g->d_syms, 2 * (size_t)g->avail, hipMemcpyDeviceToDevice) != hipSuccess ||
      hipMemcpy(no, g->d_out, (size_t)g->cap, hipMemcpyDeviceToDevice) != hipSuccess) { (void)hipFree(ns); (void)hipFree(no); return -1; }
  (void)hipFree(g->d_syms); (void)hipFree(g->d_out);
  g->d_syms = ns; g->d_out = no; g->cap = ncap;
  return 0;
}

static int prog_snapshot(V224 *v, int side) {
  ensure_layout(v, 1);
  ensure_min_valid(v);
  k_snapshot_rel<<<V224_NSTATES / 8 / 256, 256, 0, v->st>>>(v->m[v->cur], v->ds, v->pass, v->snap[side]);
  HIPCHK(hipEventRecord(v->ev_snap[side], v->st));
  return 0;
fail:
  return -1;
}

// enqueue what the symbols known so far allow.  Unless `final`, a decoder advances in whole chunks only (a ragged call
// would cost remainder passes and a switch of the metric order in the middle of the stream), and is told at most PROG_LOOK
// chunks ahead of what it has finished: feed() returns at once, and the launches of the two decoders sit next to each
// other in the device queues (a decoder handed its whole backlog in one go keeps the host inside that call until the
// backlog has drained, and the other decoder idle meanwhile).
//
// Where the cut goes is decided when decoder 1 can start, not before: if decoder 0 has finished a_done bits by then
// and both run at the same (pair) rate from then on, they finish together when x - a_done = expected - x + warm, i.e.
// x = (expected + warm + a_done) / 2 -- later than the middle by half of what decoder 0 managed alone.  Symbols that
// arrive no faster than one decoder decodes push x to the end (no second part: nothing to gain); symbols that are all
// there at once give the even split of split_core.  Until the cut is fixed decoder 0 is only told PROG_LOOK_PRE chunks
// ahead of what it has finished, so that its snapshot at x - check can still be placed.

// one step of decoder j: at most PROG_SLAB chunks.  > 0: something was enqueued (or a snapshot taken), 0: nothing to do now
static int prog_step(V224Prog *g, int j, bool final) {
  const long long q = g->chunk;
  V224 *v = g->d[j];
  long long &pos = j == 0 ? g->a_pos : g->b_pos;
  bool &snapped = j == 0 ? g->a_snapped : g->b_snapped;
  long long lim = g->avail;
  uint8_t *out = g->d_out + pos;
  if (j == 0) {
    if (g->start1 >= 0) {
      if (lim > g->start1) lim = g->start1;
      if (!snapped && lim > g->start1 - g->check) lim = g->start1 - g->check;
    }
  } else {
    if (g->start1 < 0 || (final && g->avail <= g->start1) || g->avail <= pos) return 0;     // no second part (planned, or reached)
    if (!g->b_started) {
      if (g->avail - pos < q && !final) return 0;
      if (init_viterbi224(v, 0) != 0) return -1;
      g->b_started = true;
    }
    if (!snapped) { if (lim > g->start1 - g->check) lim = g->start1 - g->check; }
    else if (pos < g->start1 && lim > g->start1) lim = g->start1;
    if (pos < g->start1) out = v->warmout + (pos - (g->start1 - g->warm));                    // warm-up and seam window: not output
  }
  if (!final) {
    const long long look = (j == 0 && g->start1 < 0 && g->can_split) ? PROG_LOOK_PRE : PROG_LOOK;
    if (lim > (g->tr[j].done + look * q) / q * q) lim = (g->tr[j].done + look * q) / q * q;
  }
  long long n = lim - pos;
  if (n > PROG_SLAB * q) n = PROG_SLAB * q;
  if (!(final && pos + n == g->avail)) n = n / q * q;          // ragged only at the very end of the stream
  if (n > 0) {
    if (v224hip_stream_decode_dev(v, g->d_syms + 2 * pos, (int)n, g->delay, out) != 0) return -1;
    pos += n;
    prog_mark(g, j, pos);
  }
  if (g->start1 >= 0 && !snapped && pos == g->start1 - g->check && (j == 0 || g->b_started)) {
    if (prog_snapshot(v, j) != 0) return -1;
    snapped = true;
    return 1;
  }
  return n > 0 ? 1 : 0;
}

static int prog_advance(V224Prog *g, bool final) {
  const long long q = g->chunk;
  for (int j = 0; j < g->ndec; j++) prog_poll(g, j);
  if (g->can_split && g->start1 < 0) {
    if (final) g->can_split = false;                           // everything is known and decoder 1 never got going
    else {
      const long long a_done = g->tr[0].done;
      // Both decoders run at the same rate once all symbols are there.  While symbols still arrive, whoever produces them
      // shares the GPU, and decoder 1 -- the second 1 024-thread workgroup on every CU -- gets less of it than decoder 0
      // (measured in the chain, 250 kS/s and 10 MS/s: about a third of decoder 0's progress).  Balance
      //   x - a_done - a_rate t_rem  =  expected - x + warm - gamma a_rate t_rem ,    t_rem = (expected - avail) / arrival rate,
      // i.e. the cut moves later by (1 - gamma) a_rate t_rem / 2.  All symbols there at once: t_rem = 0, the even split.
      long long bias = 0;
      {
        const double dt = prog_now() - g->t_begin;
        if (dt > 0.05 && g->avail > 0 && g->expected > g->avail && a_done > 0) {
          const double arrival = (double)g->avail / dt, a_rate = (double)a_done / dt;
          const double t_rem = (double)(g->expected - g->avail) / arrival;
          bias = (long long)((1.0 - prog_gamma()) * a_rate * t_rem / 2.0);
        }
      }
      const long long x0 = (g->expected + g->warm + a_done) / 2 / q * q;          // the even split of what is left
      long long x = ((g->expected + g->warm + a_done) / 2 + bias) / q * q;
      if (x - g->check < g->a_pos) x = (g->a_pos + g->check + q - 1) / q * q;      // (cannot happen with PROG_LOOK_PRE < warm / chunk)
      if (x0 + 2 * q > g->expected) g->can_split = false;      // what would be left for decoder 1 is not worth a warm-up
      else if (x + 2 * q > g->expected) { /* only the (estimated) bias says so: look again at the next feed */ }
      else if (g->avail >= x - g->warm + q && x - g->warm >= 0) {
        g->start1 = x; g->b_pos = x - g->warm; g->tr[1].done = g->b_pos;
        g->t_cut = prog_now(); g->a_done_at_cut = (double)a_done; g->avail_at_cut = (double)g->avail;
      }
    }
  }
  for (;;) {                                                   // slab by slab, the two decoders in turn
    const int ra = prog_step(g, 0, final);
    if (ra < 0) return -1;
    const int rb = g->ndec > 1 ? prog_step(g, 1, final) : 0;
    if (rb < 0) return -1;
    if (ra == 0 && rb == 0) break;
  }
  return 0;
}

extern "C" void v224hip_progressive_abort(void *h) {
  V224Prog *g = (V224Prog *)h;
  if (!g) return;
  for (int j = 0; j < g->ndec; j++) if (g->d[j]) (void)hipStreamSynchronize(g->d[j]->st);
  for (int j = 0; j < 2; j++) for (int i = 0; i < V224Prog::NEV; i++) if (g->tr[j].ev[i]) (void)hipEventDestroy(g->tr[j].ev[i]);
  (void)hipFree(g->d_syms); (void)hipFree(g->d_out);
  delete g;
}

extern "C" void *v224hip_progressive_begin(void *const *decoders, int ndec, long long expected_bits, int delay, int warm_bits) {
  V224Prog *g = nullptr;
  if (!decoders || ndec < 1 || !decoders[0] || delay <= 0 || expected_bits < 0) {
    snprintf(g_err, sizeof g_err, "progressive_begin: bad argument");
    return nullptr;
  }
  if (ndec > 2) ndec = 2;                                      // two decoders are all a CU can hold
  V224 *v0 = (V224 *)decoders[0];
  for (int j = 0; j < ndec; j++) {
    V224 *v = (V224 *)decoders[j];
    if (!v || v->dev != v0->dev || v->chunk != v0->chunk || v->len < delay + (v->st2 == v->st ? 1 : 2) * v->chunk) {
      snprintf(g_err, sizeof g_err, "progressive_begin: decoder %d is NULL, on another device, has another chunk size or a short ring", j);
      return nullptr;
    }
  }
  g = new V224Prog();
  g->ndec = ndec; g->delay = delay;
  g->d[0] = v0; g->d[1] = ndec > 1 ? (V224 *)decoders[1] : nullptr;
  g->chunk = v0->chunk;
  g->check = ((long long)delay + g->chunk - 1) / g->chunk * g->chunk;
  g->warm = ((long long)warm_bits + g->chunk - 1) / g->chunk * g->chunk;
  // at least one chunk for the fresh start to be forgotten in front of the seam window (measured: <= 765 steps down to
  // Eb/N0 0 dB, profiles/r02h_metric_convergence.txt); a caller that expects worse asks for more, and a seam that has not
  // converged is caught by the check either way
  if (g->warm < g->check + g->chunk) g->warm = g->check + g->chunk;
  g->expected = expected_bits;
  g->t_begin = prog_now();
  g->start1 = -1; g->b_pos = 0;
  g->can_split = ndec > 1 && (expected_bits + g->warm) / 2 / g->chunk * g->chunk >= g->warm + g->chunk;     // as split_core: worth two parts at all?
  HIPCHK(hipSetDevice(v0->dev));
  g->cap = expected_bits + expected_bits / 8 + 65536;
  HIPCHK(hipMalloc(&g->d_syms, 2 * (size_t)g->cap));
  HIPCHK(hipMalloc(&g->d_out, (size_t)g->cap));
  if (g->can_split) {
    V224 *a = g->d[0], *b = g->d[1];
    if (!a->snap[0]) HIPCHK(hipMalloc(&a->snap[0], sizeof(uint16_t) * V224_NSTATES));
    if (!a->ev_snap[0]) HIPCHK(hipEventCreateWithFlags(&a->ev_snap[0], hipEventDisableTiming));
    if (!b->snap[1]) HIPCHK(hipMalloc(&b->snap[1], sizeof(uint16_t) * V224_NSTATES));
    if (!b->ev_snap[1]) HIPCHK(hipEventCreateWithFlags(&b->ev_snap[1], hipEventDisableTiming));
    if (ensure_cap(&b->warmout, &b->warmout_cap, (size_t)g->warm) != 0) { snprintf(g_err, sizeof g_err, "progressive_begin: warm-up buffer allocation failed"); goto fail; }
  }
  if (init_viterbi224(v0, 0) != 0) goto fail;
  return g;
fail:
  v224hip_progressive_abort(g);
  return nullptr;
}

extern "C" int v224hip_progressive_feed(void *h, const uint8_t *syms, int nbits) {
  V224Prog *g = (V224Prog *)h;
  if (!g || nbits < 0 || (nbits > 0 && !syms)) { snprintf(g_err, sizeof g_err, "progressive_feed: bad argument"); return -1; }
  if (nbits == 0) return 0;
  HIPCHK(hipSetDevice(g->d[0]->dev));
  if (prog_grow(g, g->avail + nbits) != 0) { snprintf(g_err, sizeof g_err, "progressive_feed: cannot grow the symbol buffer"); return -1; }
  HIPCHK(hipMemcpy(g->d_syms + 2 * g->avail, syms, 2 * (size_t)nbits, hipMemcpyHostToDevice));   // done when it returns: both streams may read it
  g->avail += nbits;
  return prog_advance(g, false);
fail:
  return -1;
}

// All symbols are in: finish both decoders, check the seam, hand out all avail bits (out[] = what one decoder's
// v224hip_stream_decode of the whole stream writes).  *redone = 1 when the seam check failed and decoder 0 decoded the
// second part again.  The handle is gone afterwards, whatever the result.
extern "C" int v224hip_progressive_end(void *h, uint8_t *out, long long cap, long long *nbits_out, int *redone) {
  V224Prog *g = (V224Prog *)h;
  int rc = -1, fb = 0;
  unsigned cnt = 0;
  if (redone) *redone = 0;
  if (nbits_out) *nbits_out = 0;
  if (!g) return -1;
  if (!out || cap < g->avail) { snprintf(g_err, sizeof g_err, "progressive_end: output buffer too small"); goto done; }
  HIPCHK(hipSetDevice(g->d[0]->dev));
  {
    const double t_end = prog_now();
    prog_poll(g, 0);
    const long long a_done_end = g->tr[0].done;
    if (prog_advance(g, true) != 0) goto done;
    if (getenv("V224HIP_VERBOSE")) {
      double t_done[2] = {0, 0};
      for (int j = 0; j < g->ndec; j++) { (void)hipStreamSynchronize(g->d[j]->st); t_done[j] = prog_now() - g->t_begin; }
      fprintf(stderr, "v224hip progressive: at end() decoder 0 stood at %lld (told up to %lld), decoder 1 at %lld (told up to %lld, from %lld); decoder 0 done at %.2f ms, decoder 1 at >= %.2f ms\n",
              a_done_end, g->a_pos, g->tr[1].done, g->b_pos, g->start1 >= 0 ? g->start1 - g->warm : -1, t_done[0], t_done[1]);
      fprintf(stderr, "v224hip progressive: %lld bits (expected %lld); cut at %lld fixed %.2f ms after begin (decoder 0 had finished %.0f, %.0f known); "
                      "end() called at %.2f ms with decoder 0 at %lld; all decoded at %.2f ms\n", g->avail, g->expected, g->start1,
              g->start1 >= 0 ? g->t_cut - g->t_begin : -1.0, g->a_done_at_cut, g->avail_at_cut, t_end - g->t_begin, a_done_end, prog_now() - g->t_begin);
    }
  }
  if (g->start1 >= 0 && g->avail > g->start1) {
    // (decoder 0 stands at start1, both snapshots are enqueued: avail > start1 > start1 - check)
    V224 *a = g->d[0], *b = g->d[1];
    unsigned *d_cnt = (unsigned *)(a->dmisc + 1024);
    HIPCHK(hipMemsetAsync(d_cnt, 0, sizeof(unsigned), a->st));
    HIPCHK(hipStreamWaitEvent(a->st, b->ev_snap[1], 0));
    k_count_diff<<<V224_NSTATES / 8 / 256, 256, 0, a->st>>>(reinterpret_cast<const uint4 *>(a->snap[0]), reinterpret_cast<const uint4 *>(b->snap[1]), d_cnt);
    HIPCHK(hipMemcpyAsync(&cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost, a->st));
    HIPCHK(hipStreamSynchronize(a->st));
    if (getenv("V224HIP_SPLIT_FORCE_FALLBACK")) cnt = 1;                               // test hook, as in split_core
    if (cnt != 0) {
      fb = 1;
      HIPCHK(hipStreamSynchronize(b->st));                                             // its writes into d_out must not land later
      if (v224hip_stream_decode_dev(a, g->d_syms + 2 * g->start1, (int)(g->avail - g->start1), g->delay, g->d_out + g->start1) != 0) goto done;
    }
  }
  for (int j = 0; j < g->ndec; j++) HIPCHK(hipStreamSynchronize(g->d[j]->st));
  if (g->avail) HIPCHK(hipMemcpy(out, g->d_out, (size_t)g->avail, hipMemcpyDeviceToHost));
  if (nbits_out) *nbits_out = g->avail;
  if (redone) *redone = fb;
  rc = 0;
done:
fail:
  v224hip_progressive_abort(g);
  return rc;
}

extern "C" int v224hip_stream_decode(void *p, const uint8_t *syms, int nbits, int delay, uint8_t *out) {
  V224 *v = (V224 *)p;
  if (!v) return -1;
  if (nbits <= 0) return 0;
  HIPCHK(hipSetDevice(v->dev));
  HIPCHK(hipStreamSynchronize(v->st));
  if (ensure_cap(&v->dsyms, &v->dsyms_cap, 2 * (size_t)nbits) != 0) return -1;
  v->dsyms_off = v->dsyms_cap;                      // the whole buffer is in use until the next drain
  if (ensure_cap(&v->dout, &v->dout_cap, (size_t)nbits) != 0) return -1;
  HIPCHK(hipMemcpyAsync(v->dsyms, syms, 2 * (size_t)nbits, hipMemcpyHostToDevice, v->st));
  if (v224hip_stream_decode_dev(p, v->dsyms, nbits, delay, v->dout) != 0) return -1;
  HIPCHK(hipMemcpyAsync(out, v->dout, (size_t)nbits, hipMemcpyDeviceToHost, v->st));
  HIPCHK(hipStreamSynchronize(v->st));
  return 0;
fail:
  return -1;
}

extern "C" int v224hip_set_option(void *p, const char *key, long value) {
  V224 *v = (V224 *)p;
  if (!v || !key) return -1;
  if (!strcmp(key, "chunk")) { if (value < 1) return -1; v->chunk = (int)value; return 0; }
  if (!strcmp(key, "profile")) { if (value < 0) return -1; v->profile = (int)value; return 0; }
  return -1;
}

// "chainback_redone": pieces of parallel framed chainbacks that failed their seam check and were walked again (read and reset)
extern "C" long v224hip_get_counter(void *p, const char *key) {
  V224 *v = (V224 *)p;
  unsigned c = 0;
  if (!v || !key) return -1;
  if (!strcmp(key, "dp")) return (long)v->dp;                           // host-side bookkeeping, nothing to wait for
  if (!strcmp(key, "steps")) return (long)v->nsteps;
  if (strcmp(key, "chainback_redone")) return -1;
  HIPCHK(hipSetDevice(v->dev));
  if (v->st2 != v->st) HIPCHK(hipStreamSynchronize(v->st2));
  HIPCHK(hipMemcpyAsync(&c, v->dmisc + 96, sizeof c, hipMemcpyDeviceToHost, v->st));
  HIPCHK(hipMemsetAsync(v->dmisc + 96, 0, sizeof c, v->st));
  HIPCHK(hipStreamSynchronize(v->st));
  return (long)c;
fail:
  return -1;
}

extern "C" int v224hip_sync(void *p) {
  V224 *v = (V224 *)p;
  if (!v) return -1;
  if (hipSetDevice(v->dev) != hipSuccess) return -1;
  if (hipStreamSynchronize(v->st) != hipSuccess) return -1;
  if (v->st2 != v->st && hipStreamSynchronize(v->st2) != hipSuccess) return -1;
  return 0;
}

extern "C" int v224hip_acs_stats(void *p, unsigned long long *launches, double *total_ms,
                                 unsigned long long *steps, int reset) {
  V224 *v = (V224 *)p;
  if (!v) return -1;
  prof_harvest(v, true);
  if (launches) *launches = v->prof_launches;
  if (total_ms) *total_ms = v->prof_ms;
  if (steps) *steps = v->prof_steps;
  if (reset) { v->prof_launches = 0; v->prof_ms = 0; v->prof_steps = 0; }
  return 0;
}

extern "C" int v224hip_export_row(void *p, int row, uint8_t *out) {
  V224 *v = (V224 *)p;
  uint32_t *d = nullptr;
  if (!v || row < 0 || row >= v->len) return -1;
  HIPCHK(hipSetDevice(v->dev));
  HIPCHK(hipMalloc(&d, V224_ROWWORDS * sizeof(uint32_t)));
  k_export_row<<<V224_ROWWORDS / 256, 256, 0, v->st>>>(v->rows, v->rowmeta, row, d);
  HIPCHK(hipMemcpyAsync(out, d, V224_ROWWORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, v->st));
  HIPCHK(hipStreamSynchronize(v->st));
  (void)hipFree(d);
  return 0;
fail:
  if (d) (void)hipFree(d);
  return -1;
}

extern "C" int v224hip_export_metrics(void *p, uint32_t *out) {
  V224 *v = (V224 *)p;
  uint32_t *d = nullptr;
  if (!v) return -1;
  HIPCHK(hipSetDevice(v->dev));
  HIPCHK(hipMalloc(&d, (size_t)V224_NSTATES * sizeof(uint32_t)));
  ensure_layout(v, 0);
  ensure_min_valid(v);
  k_export_metrics<<<V224_NSTATES / 256, 256, 0, v->st>>>(v->m[v->cur], v->ds, v->pass, d);
  HIPCHK(hipMemcpyAsync(out, d, (size_t)V224_NSTATES * sizeof(uint32_t), hipMemcpyDeviceToHost, v->st));
  HIPCHK(hipStreamSynchronize(v->st));
  (void)hipFree(d);
  return 0;
fail:
  if (d) (void)hipFree(d);
  return -1;
}

extern "C" void *v224hip_dev_alloc(size_t bytes) {
  void *d = nullptr;
  if (hipMalloc(&d, bytes ? bytes : 1) != hipSuccess) return nullptr;
  return d;
}
extern "C" void v224hip_dev_free(void *d) { if (d) (void)hipFree(d); }
extern "C" int v224hip_h2d(void *d, const void *h, size_t n) {
  return hipMemcpy(d, h, n, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}
extern "C" int v224hip_d2h(void *h, const void *d, size_t n) {
  return hipMemcpy(h, d, n, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
