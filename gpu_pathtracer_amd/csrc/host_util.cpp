#include "host_util.h"

#include <cstdarg>
#include <cstdio>

#include "../../include/gpt.h"

static thread_local char g_error[1024] = "";

void gpt_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

extern "C" const char *gpt_last_error(void) { return g_error; }
