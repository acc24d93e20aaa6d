// ABI version + last-error plumbing of libaid_hip.so.
#include <string.h>
#include "aid_common.h"

static char g_last_error[256] = "";

void aid_set_error(const char* msg) {
    strncpy(g_last_error, msg ? msg : "", sizeof(g_last_error) - 1);
    g_last_error[sizeof(g_last_error) - 1] = 0;
}

extern "C" int aid_abi_version(void) { return AID_ABI_VERSION; }
extern "C" const char* aid_last_error(void) { return g_last_error; }
