// Library identification entry points (+ the switch table of the development build).
#include "common.hpp"

extern "C" const char* agp_version(void) { return "agplace_hip 0.1.0"; }
extern "C" const char* agp_arch(void) { return "gfx950"; }

#if defined(AGP_TUNING)
// Development build only (`make tuning`): the experiment switches that AGP_TUNE(key, default) reads, set by the A/B harnesses
// under tools/ through this extra export.  The release library has neither the table nor the export: its AGP_TUNE is the
// default itself and it reads no process environment.
#include <string.h>
namespace {
struct Switch { char key[32]; int value; };
Switch g_switches[64];
int g_nswitches = 0;
}  // namespace
int agp_tune_lookup(const char* key, int def) {
    for (int i = 0; i < g_nswitches; ++i)
        if (!strcmp(g_switches[i].key, key)) return g_switches[i].value;
    return def;
}
extern "C" int agp_debug_set(const char* key, int value) {
    if (!key || strlen(key) >= sizeof(g_switches[0].key)) return AGP_E_BADARG;
    for (int i = 0; i < g_nswitches; ++i)
        if (!strcmp(g_switches[i].key, key)) { g_switches[i].value = value; return AGP_OK; }
    if (g_nswitches >= 64) return AGP_E_BADARG;
    strcpy(g_switches[g_nswitches].key, key);
    g_switches[g_nswitches++].value = value;
    return AGP_OK;
}
#endif
