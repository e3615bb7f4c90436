// ABI bookkeeping entry points of libvt355.so (see include/vt355.h).
#include "common.h"
extern "C" int vt_version(void) { return 2; }
extern "C" const char* vt_arch(void) { return "gfx950"; }
extern "C" const char* vt_error_string(int code) {
    switch (code) {
        case VT_OK: return "ok";
        case VT_ERR_BAD_SHAPE: return "bad shape / stride / leading dimension";
        case VT_ERR_BAD_ALIGN: return "pointer not sufficiently aligned";
        case VT_ERR_LAUNCH: return "kernel launch failed";
        case VT_ERR_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown error";
    }
}
