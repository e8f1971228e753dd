/* timefmt.h -- dd:hh:mm:ss.sss for the stderr status lines (role of reference timeformat.c:26-62) */
#ifndef ISEE3_TIMEFMT_H
#define ISEE3_TIMEFMT_H
#include <stdio.h>
static inline const char *isee3_format_hms(double t) {
  static char buf[64];
  int days = (int)(t / 86400.); t -= days * 86400.;
  int hours = (int)(t / 3600.); t -= hours * 3600.;
  int minutes = (int)(t / 60.); t -= minutes * 60.;
  int n = 0;
  if (days > 0) n += snprintf(buf + n, sizeof buf - (size_t)n, "%d:", days);
  if (days > 0 || hours > 0) n += snprintf(buf + n, sizeof buf - (size_t)n, "%02d:", hours);
  n += snprintf(buf + n, sizeof buf - (size_t)n, "%02d:", minutes);
  snprintf(buf + n, sizeof buf - (size_t)n, "%s%.3lf", t < 10.0 ? "0" : "", t);
  return buf;
}
#endif
