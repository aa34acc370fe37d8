// The environment-switch table (env.h): the library's only getenv.
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "kernels.h"
#include "env.h"

namespace mpu {
namespace {
struct Entry { const char* name; int kind; long dflt; const char* doc; };
const Entry g_tab[ENV_COUNT] = {
#define MPU_ENV_ROW(id, name, kind, dflt, doc) {name, kind, dflt, doc},
    MPU_ENV_TABLE(MPU_ENV_ROW)
#undef MPU_ENV_ROW
};
long g_val[ENV_COUNT];
bool g_read = false;
void read_all() {
    for (int i = 0; i < ENV_COUNT; ++i) {
        const char* e = getenv(g_tab[i].name);
        switch (g_tab[i].kind) {
            case ENV_ON: g_val[i] = (e && e[0] == '0') ? 0 : 1; break;
            case ENV_OFF: g_val[i] = (e && e[0] == '1') ? 1 : 0; break;
            case ENV_IMPL: g_val[i] = (e && strcmp(e, "regs") == 0) ? 0 : 1; break;
            default: g_val[i] = e ? atol(e) : g_tab[i].dflt; break;
        }
    }
    g_read = true;
}
}  // namespace

long env(EnvId id) {
    if (!g_read) read_all();
    return g_val[id];
}
}  // namespace mpu

using namespace mpu;

extern "C" {

// One line per switch: "NAME<TAB>value in force<TAB>default<TAB>what it does". NUL-terminated, truncated to cap; returns the
// full length in bytes.
int64_t mpu_env_describe(char* buf, int64_t cap) {
    if (!g_read) read_all();
    int64_t n = 0;
    for (int i = 0; i < ENV_COUNT; ++i) {
        char line[512];
        const int k = snprintf(line, sizeof(line), "%s\t%ld\t%ld\t%s\n", g_tab[i].name, g_val[i], g_tab[i].dflt, g_tab[i].doc);
        if (buf && n + k < cap) memcpy(buf + n, line, (size_t)k);
        n += k;
    }
    if (buf && cap > 0) buf[n < cap ? n : cap - 1] = 0;
    return n;
}

}  // extern "C"
