// VALU issue-rate microbenchmark with inline asm (nothing for the compiler to fold)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITER 1000
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define KERN(NAME, INSN)                                                                         \
  __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned seed) {                    \
    unsigned a[8], b = seed | 3, c = seed * 7 + 1;                                               \
    for (int k = 0; k < 8; k++) a[k] = threadIdx.x * (k + 3) + seed;                            \
    for (int i = 0; i < ITER; i++) {                                                             \
      asm volatile(INSN("%0") INSN("%1") INSN("%2") INSN("%3") INSN("%4") INSN("%5") INSN("%6") INSN("%7") \
                   INSN("%0") INSN("%1") INSN("%2") INSN("%3") INSN("%4") INSN("%5") INSN("%6") INSN("%7") \
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                   : "v"(b), "v"(c));   /* %9 is clobbered by the mixed kernels: harmless, it is only an input value */                                                            \
    }                                                                                            \
    out[blockIdx.x * 256 + threadIdx.x] = a[0] ^ a[1] ^ a[2] ^ a[3] ^ a[4] ^ a[5] ^ a[6] ^ a[7]; \
  }
#define I_ADD(r) "v_add_u32 " r ", " r ", %8\n"
#define I_SUB(r) "v_sub_u32 " r ", " r ", %8\n"
#define I_PKADD(r) "v_pk_add_u16 " r ", " r ", %8\n"
#define I_PKSUB(r) "v_pk_sub_i16 " r ", " r ", %8\n"
#define I_PKMIN(r) "v_pk_min_u16 " r ", " r ", %8\n"
#define I_PKSHR(r) "v_pk_lshrrev_b16 " r ", 1, " r "\n"
#define I_PKMAD(r) "v_pk_mad_u16 " r ", " r ", %8, %9\n"
#define I_SHR(r) "v_lshrrev_b32 " r ", 1, " r "\n"
#define I_AND(r) "v_and_b32 " r ", " r ", %8\n"
#define I_BFI(r) "v_bfi_b32 " r ", %8, %9, " r "\n"
#define I_ANDOR(r) "v_and_or_b32 " r ", " r ", %8, %9\n"
#define I_LSHLOR(r) "v_lshl_or_b32 " r ", " r ", 1, %9\n"
#define I_PERM(r) "v_perm_b32 " r ", " r ", %8, %9\n"
#define I_ALIGN(r) "v_alignbit_b32 " r ", " r ", %8, 1\n"
#define I_MIN(r) "v_min_u32 " r ", " r ", %8\n"
#define I_ADD3(r) "v_add3_u32 " r ", " r ", %8, %9\n"
#define I_FMA(r) "v_fma_f32 " r ", " r ", %8, %9\n"
#define I_OR3(r) "v_or3_b32 " r ", " r ", %8, %9\n"
#define I_CNDM(r) "v_cndmask_b32 " r ", " r ", %8, vcc\n"
#define I_MINU16(r) "v_min_u16 " r ", " r ", %8\n"
#define I_MIX_PKADD_AND(r) "v_pk_add_u16 " r ", " r ", %8\n" "v_and_b32 %9, %9, %8\n"
#define I_MIX_PKADD_ADD(r) "v_pk_add_u16 " r ", " r ", %8\n" "v_add_u32 %9, %9, %8\n"
#define I_MIX_PKMIN_SHR(r) "v_pk_min_u16 " r ", " r ", %8\n" "v_lshrrev_b32 %9, 1, %9\n"
#define I_MIX_BFI_AND(r) "v_bfi_b32 " r ", %8, %8, " r "\n" "v_and_b32 %9, %9, %8\n"
#define I_MIX_PKADD_PKMIN(r) "v_pk_add_u16 " r ", " r ", %8\n" "v_pk_min_u16 %9, %9, %8\n"
#define I_MIX_ADD_AND(r) "v_add_u32 " r ", " r ", %8\n" "v_and_b32 %9, %9, %8\n"
// grouped: 8 packed ops then 8 VOP2 ops per iteration (same totals as k_mix_pkadd_and)
__global__ __launch_bounds__(256) void k_grp_pkadd_and(unsigned *out, unsigned seed) {
  unsigned a[8], c[8], b = seed | 3;
  for (int k = 0; k < 8; k++) { a[k] = threadIdx.x * (k + 3) + seed; c[k] = a[k] ^ 0x55; }
  for (int i = 0; i < ITER; i++) {
    asm volatile("v_pk_add_u16 %0, %0, %16\n v_pk_add_u16 %1, %1, %16\n v_pk_add_u16 %2, %2, %16\n v_pk_add_u16 %3, %3, %16\n"
                 "v_pk_add_u16 %4, %4, %16\n v_pk_add_u16 %5, %5, %16\n v_pk_add_u16 %6, %6, %16\n v_pk_add_u16 %7, %7, %16\n"
                 "v_and_b32 %8, %8, %16\n v_and_b32 %9, %9, %16\n v_and_b32 %10, %10, %16\n v_and_b32 %11, %11, %16\n"
                 "v_and_b32 %12, %12, %16\n v_and_b32 %13, %13, %16\n v_and_b32 %14, %14, %16\n v_and_b32 %15, %15, %16\n"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                   "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : "v"(b));
  }
  unsigned r = 0; for (int k = 0; k < 8; k++) r ^= a[k] ^ c[k];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
// alternating with the same 16 registers
__global__ __launch_bounds__(256) void k_alt_pkadd_and(unsigned *out, unsigned seed) {
  unsigned a[8], c[8], b = seed | 3;
  for (int k = 0; k < 8; k++) { a[k] = threadIdx.x * (k + 3) + seed; c[k] = a[k] ^ 0x55; }
  for (int i = 0; i < ITER; i++) {
    asm volatile("v_pk_add_u16 %0, %0, %16\n v_and_b32 %8, %8, %16\n v_pk_add_u16 %1, %1, %16\n v_and_b32 %9, %9, %16\n"
                 "v_pk_add_u16 %2, %2, %16\n v_and_b32 %10, %10, %16\n v_pk_add_u16 %3, %3, %16\n v_and_b32 %11, %11, %16\n"
                 "v_pk_add_u16 %4, %4, %16\n v_and_b32 %12, %12, %16\n v_pk_add_u16 %5, %5, %16\n v_and_b32 %13, %13, %16\n"
                 "v_pk_add_u16 %6, %6, %16\n v_and_b32 %14, %14, %16\n v_pk_add_u16 %7, %7, %16\n v_and_b32 %15, %15, %16\n"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                   "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : "v"(b));
  }
  unsigned r = 0; for (int k = 0; k < 8; k++) r ^= a[k] ^ c[k];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
// a whole packed butterfly written with VOP2 / SDWA instructions only (no VOP3P): 4 add, 2 sub, 4 half-word min,
// then and / shift / or for the decision bits = 16 instructions
__global__ __launch_bounds__(256) void k_bfly_vop2(unsigned *out, unsigned seed) {
  unsigned a[8], b = seed | 3, c = seed * 7 + 1, acc = 0;
  for (int k = 0; k < 8; k++) a[k] = threadIdx.x * (k + 3) + seed;
  for (int i = 0; i < ITER; i++) {
    asm volatile("v_add_u32 %2, %0, %9\n v_add_u32 %3, %1, %10\n v_add_u32 %4, %0, %10\n v_add_u32 %5, %1, %9\n"
                 "v_sub_u32 %6, %2, %3\n v_sub_u32 %7, %4, %5\n"
                 "v_min_u16_sdwa %0, %2, %3 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n"
                 "v_min_u16_sdwa %0, %2, %3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n"
                 "v_min_u16_sdwa %1, %4, %5 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n"
                 "v_min_u16_sdwa %1, %4, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n"
                 "v_and_b32 %6, %6, %9\n v_and_b32 %7, %7, %9\n v_lshrrev_b32 %6, 3, %6\n v_lshrrev_b32 %7, 2, %7\n"
                 "v_or_b32 %8, %8, %6\n v_or_b32 %8, %8, %7\n"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(acc) : "v"(b), "v"(c));
  }
  out[blockIdx.x * 256 + threadIdx.x] = a[0] ^ a[1] ^ a[2] ^ a[3] ^ a[4] ^ a[5] ^ a[6] ^ a[7] ^ acc;
}
// the same butterfly as the kernels have it: VOP3P arithmetic, perm / shift / bfi decisions = 13 instructions (+3 pad adds)
__global__ __launch_bounds__(256) void k_bfly_pk(unsigned *out, unsigned seed) {
  unsigned a[8], b = seed | 3, c = seed * 7 + 1, acc = 0;
  for (int k = 0; k < 8; k++) a[k] = threadIdx.x * (k + 3) + seed;
  for (int i = 0; i < ITER; i++) {
    asm volatile("v_pk_add_u16 %2, %0, %9\n v_pk_add_u16 %3, %1, %10\n v_pk_add_u16 %4, %0, %10\n v_pk_add_u16 %5, %1, %9\n"
                 "v_pk_sub_i16 %6, %2, %3\n v_pk_sub_i16 %7, %4, %5\n"
                 "v_pk_min_u16 %0, %2, %3\n v_pk_min_u16 %1, %4, %5\n"
                 "v_perm_b32 %6, %7, %6, %9\n v_lshrrev_b32 %8, 1, %8\n v_bfi_b32 %8, %10, %6, %8\n"
                 "v_pk_add_u16 %2, %0, %9\n v_pk_add_u16 %3, %1, %10\n v_pk_add_u16 %4, %0, %10\n v_pk_add_u16 %5, %1, %9\n v_pk_sub_i16 %6, %2, %3\n"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(acc) : "v"(b), "v"(c));
  }
  out[blockIdx.x * 256 + threadIdx.x] = a[0] ^ a[1] ^ a[2] ^ a[3] ^ a[4] ^ a[5] ^ a[6] ^ a[7] ^ acc;
}
// VOP2 adds / subs with VOP3P min (the LDS8_PK = 0 mix)
__global__ __launch_bounds__(256) void k_bfly_mixed(unsigned *out, unsigned seed) {
  unsigned a[8], b = seed | 3, c = seed * 7 + 1, acc = 0;
  for (int k = 0; k < 8; k++) a[k] = threadIdx.x * (k + 3) + seed;
  for (int i = 0; i < ITER; i++) {
    asm volatile("v_add_u32 %2, %0, %9\n v_add_u32 %3, %1, %10\n v_add_u32 %4, %0, %10\n v_add_u32 %5, %1, %9\n"
                 "v_sub_u32 %6, %2, %3\n v_sub_u32 %7, %4, %5\n"
                 "v_pk_min_u16 %0, %2, %3\n v_pk_min_u16 %1, %4, %5\n"
                 "v_and_b32 %6, %6, %9\n v_and_b32 %7, %7, %9\n v_lshrrev_b32 %6, 3, %6\n v_lshrrev_b32 %7, 2, %7\n"
                 "v_or_b32 %8, %8, %6\n v_or_b32 %8, %8, %7\n v_add_u32 %2, %0, %9\n v_add_u32 %3, %1, %10\n"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(acc) : "v"(b), "v"(c));
  }
  out[blockIdx.x * 256 + threadIdx.x] = a[0] ^ a[1] ^ a[2] ^ a[3] ^ a[4] ^ a[5] ^ a[6] ^ a[7] ^ acc;
}
KERN(k_mix_pkadd_and, I_MIX_PKADD_AND) KERN(k_mix_pkadd_add, I_MIX_PKADD_ADD) KERN(k_mix_pkmin_shr, I_MIX_PKMIN_SHR)
KERN(k_mix_bfi_and, I_MIX_BFI_AND) KERN(k_mix_pkadd_pkmin, I_MIX_PKADD_PKMIN) KERN(k_mix_add_and, I_MIX_ADD_AND)
KERN(k_add, I_ADD) KERN(k_sub, I_SUB) KERN(k_pkadd, I_PKADD) KERN(k_pksub, I_PKSUB) KERN(k_pkmin, I_PKMIN)
KERN(k_pkshr, I_PKSHR) KERN(k_pkmad, I_PKMAD) KERN(k_shr, I_SHR) KERN(k_and, I_AND) KERN(k_bfi, I_BFI)
KERN(k_andor, I_ANDOR) KERN(k_lshlor, I_LSHLOR) KERN(k_perm, I_PERM) KERN(k_align, I_ALIGN) KERN(k_min, I_MIN)
KERN(k_add3, I_ADD3) KERN(k_fma, I_FMA) KERN(k_or3, I_OR3) KERN(k_minu16, I_MINU16)
template <typename F> static void run(const char *name, F kern, unsigned *out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  printf("%-14s", name);
  for (int waves : {1, 2, 4, 8}) {
    int blocks = 256 * waves;
    kern<<<blocks, 256>>>(out, 1); hipDeviceSynchronize();
    hipEventRecord(a); kern<<<blocks, 256>>>(out, 2); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double n = (double)waves * ITER * 16;
    printf("  w%d: %5.2f cyc", waves, ms * 1e6 / n * 2.4);
  }
  printf("   (cycles per asm slot; mixed kernels have 2 instructions per slot)\n");
}
int main() {
  unsigned *out; hipMalloc(&out, 256 * 8 * 256 * 4 * 4);
#define R(n) run(#n, n, out);
  R(k_bfly_vop2) R(k_bfly_pk) R(k_bfly_mixed)
  R(k_grp_pkadd_and) R(k_alt_pkadd_and)
  R(k_mix_pkadd_and) R(k_mix_pkadd_add) R(k_mix_pkmin_shr) R(k_mix_bfi_and) R(k_mix_pkadd_pkmin) R(k_mix_add_and)
  R(k_add) R(k_sub) R(k_pkadd) R(k_pksub) R(k_pkmin) R(k_pkshr) R(k_pkmad) R(k_shr) R(k_and) R(k_bfi) R(k_andor)
  R(k_lshlor) R(k_perm) R(k_align) R(k_min) R(k_add3) R(k_fma) R(k_or3) R(k_minu16)
  return 0;
}
