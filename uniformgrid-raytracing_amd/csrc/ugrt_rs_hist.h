// ugrt_rs_hist.h -- layout of the radix sort's state (ugrt_sort.hip)
#ifndef UGRT_RS_HIST_H
#define UGRT_RS_HIST_H

#include "ugrt_ctx.h"

#define RS_BINS 256
#define RS_MAXPASS 4
// A pass's digit histogram is kept in RS_COPIES rows that are summed when it is read: the workgroups that add to it (the
// histogram kernel for the first pass, the tiles of pass p for pass p + 1) pick a row by their index, so that an
// address sees an eighth of the adds (same-address atomics take ~12 ns each, whoever issues them).
#define RS_COPIES 8

#endif
