// host_util.h — error reporting shared by the host-side translation units.
// The reference prints and __debugbreak()s (reference src/common.h:29-39) or
// exit(1)s from the loader; this library returns codes and keeps the message.
#pragma once

#if defined(__GNUC__)
#define GPT_PRINTF_LIKE __attribute__((format(printf, 1, 2)))
#else
#define GPT_PRINTF_LIKE
#endif

// stores a printf-formatted message retrievable through gpt_last_error()
void gpt_set_error(const char *fmt, ...) GPT_PRINTF_LIKE;

#include <cstdio>
// fopen(path, "rb"); if that fails, the same with the file name matched case-insensitively inside its directory: the
// reference's scenes were written on a case-insensitive file system ("geometry/Right.obj" names right.obj)
FILE *gpt_fopen_read(const char *path);
