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

#include <dirent.h>
#include <strings.h>
#include <string>

FILE *gpt_fopen_read(const char *path)
{
    FILE *f = std::fopen(path, "rb");
    if (f) return f;
    const std::string full = path;
    const size_t slash = full.find_last_of('/');
    const std::string dir = slash == std::string::npos ? "." : full.substr(0, slash == 0 ? 1 : slash);
    const std::string name = slash == std::string::npos ? full : full.substr(slash + 1);
    DIR *d = opendir(dir.c_str());
    if (!d) return nullptr;
    std::string match;
    while (const dirent *e = readdir(d))
        if (strcasecmp(e->d_name, name.c_str()) == 0) {
            match = e->d_name;
            break;
        }
    closedir(d);
    if (match.empty()) return nullptr;
    return std::fopen((dir + "/" + match).c_str(), "rb");
}
