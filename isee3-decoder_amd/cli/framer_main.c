/* framer -- minor-frame synchroniser for the decoded ISEE-3/ICE bit stream.
 * Drop-in for reference framer.c:38-98: ASCII '0'/'1' on stdin, a 1024-bit shift register, and a hex
 * dump of the register every time its last 40 bits equal the sync word 0x12fc819fbe.  Pure host code:
 * one compare per decoded bit, nothing here for a GPU to do; it completes `... | vdecode | framer`. */
#include <locale.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "isee3_code.h"
#include "timefmt.h"

int main(int argc, char **argv) {
  unsigned long long reg[16], frames = 1, bits = 0;
  int bitrate = 512, c;
  const char *lang = getenv("LANG");
  setlocale(LC_ALL, lang ? lang : "en_US.utf8");
  memset(reg, 0, sizeof reg);
  while ((c = getopt(argc, argv, "r:")) != -1)
    if (c == 'r') bitrate = atoi(optarg);
  while ((c = getchar()) != EOF) {
    unsigned long long in = (c == '1');
    for (int k = 0; k < 15; k++) reg[k] = (reg[k] << 1) | (reg[k + 1] >> 63);   /* 1024-bit shift left */
    reg[15] = (reg[15] << 1) | in;
    if ((reg[15] & 0xffffffffffULL) == ISEE3_SYNCWORD) {
      printf("Frame %'llu at bit %'llu (%s)\n", frames, bits, isee3_format_hms((double)bits / bitrate));
      for (int k = 0; k < 16; k++)
        for (int n = 56; n >= 0; n -= 8) {
          printf("%02llx", (reg[k] >> n) & 0xff);
          putchar(n == 0 && (k % 2) == 1 ? '\n' : ' ');
        }
      frames++;
      putchar('\n');
      fflush(stdout);
    }
    bits++;
  }
  return 0;
}
