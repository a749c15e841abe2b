// Library-wide state: version string and the per-thread error message.
#include "common.hpp"

namespace dmh {
thread_local char g_err[512] = "";
}

extern "C" {
const char* dmh_version(void) { return "dmh_hip 0.1 (gfx950)"; }
const char* dmh_last_error(void) { return dmh::g_err; }
}
