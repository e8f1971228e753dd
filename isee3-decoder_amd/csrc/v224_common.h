// v224_common.h -- shared definitions for the HIP Viterbi library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define V224_K        24
#define V224_SBITS    23                       // state bits
#define V224_NSTATES  (1u << V224_SBITS)       // 2^23
#define V224_NBFLY    (1u << (V224_SBITS - 1)) // butterflies per step
#define V224_ROWWORDS (V224_NSTATES / 32)      // 2^18 dwords = 1 MiB per decision row
#define V224_POLY1    073665667u               // code.h:59 (MCQLI24)
#define V224_POLY2    073665665u               // code.h:60 ; POLY1 ^ POLY2 == 2
#define V224_SMASK    (V224_NSTATES - 1)

// Stored metric = true port metric - off (mod 2^16).  Every ACS launch first subtracts
// (min of its input - V224_BASE), so the smallest stored value is V224_BASE on entry; the fused
// kernel additionally drops a common +255 per stage, so values may dip by 255 per stage below
// BASE and rise by (spread + 255 per stage) above it.  spread <= 1000 + 23*510 = 12 730.
#define V224_BASE     4096u

// row layout codes (rowmeta[row]): 0 = port bit order; else (K << 8) | stage for fused passes
#define V224_META_PORT 0u

// Minimum tracking without atomics: launch n writes one minimum per workgroup into
// blkmin[(n+1)&1][blockIdx] and the count into nmin[(n+1)&1]; every workgroup of launch n+1 reduces
// that small array itself (L2-resident, hidden behind its metric loads).  A single atomicMin slot
// costs ~12 ns per workgroup (same-address atomics serialise): 6 us of a 17 us launch.
#define V224_MAXBLK 4096
struct V224Dev {
  long long off;           // true metric = stored + off
  unsigned nmin[2];
  unsigned scratch[4];     // argmin / max / misc reductions
  unsigned blkmin[2][V224_MAXBLK];
};

#define HIPCHK(expr)                                                                   \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) { v224_set_error(#expr, _e, __FILE__, __LINE__); goto fail; } \
  } while (0)

void v224_set_error(const char *what, hipError_t e, const char *file, int line);
