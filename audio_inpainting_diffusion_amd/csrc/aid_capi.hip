// ABI version + last-error plumbing of libaid_hip.so.
#include <string.h>
#include "aid_common.h"

static char g_last_error[256] = "";

void aid_set_error(const char* msg) {
    strncpy(g_last_error, msg ? msg : "", sizeof(g_last_error) - 1);
    g_last_error[sizeof(g_last_error) - 1] = 0;
}

// Name of the device kernel the most recent aid_conv2d call dispatched to (measurement aid: bench.py groups conv launch
// times per kernel family with it).  Host-side bookkeeping only.
static const char* g_last_kernel = "";
void aid_note_kernel(const char* name) { g_last_kernel = name ? name : ""; }
extern "C" const char* aid_last_kernel(void) { return g_last_kernel; }

extern "C" int aid_abi_version(void) { return AID_ABI_VERSION; }
extern "C" const char* aid_last_error(void) { return g_last_error; }
