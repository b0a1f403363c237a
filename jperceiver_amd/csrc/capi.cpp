// Error-string plumbing for the C-ABI (include/jperceiver_hip.h).
#include <string.h>
static thread_local char g_err[512] = "";
extern "C" void jp_set_last_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* jp_last_error_string(void) { return g_err; }
extern "C" int jp_abi_version(void) { return 1; }
