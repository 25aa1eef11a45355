// Library identification entry points.
#include "common.hpp"

extern "C" const char* agp_version(void) { return "agplace_hip 0.1.0"; }
extern "C" const char* agp_arch(void) { return "gfx950"; }
