/* isee3_code.h -- code constants of the ISEE-3/ICE downlink (reference code.h:54-63, MCQLI24)
 * and small helpers shared by the C pipe stages. */
#ifndef ISEE3_CODE_H
#define ISEE3_CODE_H
#include <stdint.h>

#define ISEE3_K       24
#define ISEE3_POLY1   073665667u
#define ISEE3_POLY2   073665665u
#define ISEE3_G1FLIP  0
#define ISEE3_G2FLIP  1          /* second symbol of every pair is inverted */

#define ISEE3_FRAMEBITS    1024  /* vdecode.c:14 */
#define ISEE3_FRAMESYMBOLS 2048
#define ISEE3_SYNCWORD     0x12fc819fbeULL   /* framer.c:18, decode.c:24 */

static inline int isee3_parity(unsigned long long x) { return __builtin_parityll(x); }

#endif
