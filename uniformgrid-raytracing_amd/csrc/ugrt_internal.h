// ugrt_internal.h -- shared by the host and device translation units of libugrt.so
#ifndef UGRT_INTERNAL_H
#define UGRT_INTERNAL_H

#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "ugrt.h"
#include "ugrt_fmath.h"

typedef uint32_t u32;
typedef uint64_t u64;

// sets ugrt_last_error() and returns `code`
int ugrt_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#endif
