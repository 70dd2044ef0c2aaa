// libfgcn: version, error text, device check.
#include <cstring>
#include <new>

#include "fgcn_common.hpp"

// Settings the launchers read: the math mode, the product form inside FGCN_MATH_BF16X3, the kernel-variant table.  They live in a
// context; every thread has a CURRENT context (fgcn_ctx_set_current), and a thread that never set one reads the process-wide
// defaults.  A launcher reads its thread's current context once, on the host, at call time -- two threads with two contexts (two
// models in two math modes on two streams) do not see each other's settings.
struct fgcn_ctx {
    int math_mode;
    int products;
    int tuning[32];
};

namespace fgcn {
char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
// process-wide defaults: FGCN_MATH_BF16X3 (the benchmarked arithmetic: float32-accurate products on the bf16 matrix pipe), bf16x3
// products, tuning key 0 = 1 and every other key 0
static fgcn_ctx g_default = {FGCN_MATH_BF16X3, FGCN_PRODUCTS_BF16X3, {1}};
static thread_local fgcn_ctx* t_current = nullptr;
static inline fgcn_ctx& cur() { return t_current ? *t_current : g_default; }
int tuning(int key) { return (key >= 0 && key < 32) ? cur().tuning[key] : 0; }
int math_mode() { return cur().math_mode; }
int products() { return cur().products; }
}  // namespace fgcn

extern "C" int fgcn_ctx_create(fgcn_ctx** out) {
    if (!out) return fgcn::fail(FGCN_E_BADARG, "ctx_create: null pointer");
    *out = new (std::nothrow) fgcn_ctx(fgcn::cur());   // starts as a copy of the calling thread's current settings
    if (!*out) return fgcn::fail(FGCN_E_BADARG, "ctx_create: out of memory");
    return FGCN_OK;
}

extern "C" int fgcn_ctx_destroy(fgcn_ctx* ctx) {
    if (ctx && ctx == fgcn::t_current) return fgcn::fail(FGCN_E_BADARG, "ctx_destroy: the context is current on this thread");
    delete ctx;
    return FGCN_OK;
}

extern "C" int fgcn_ctx_set_current(fgcn_ctx* ctx) {
    fgcn::t_current = ctx;
    return FGCN_OK;
}

extern "C" fgcn_ctx* fgcn_ctx_get_current(void) { return fgcn::t_current; }

extern "C" int fgcn_set_math_mode(int mode) {
    if (mode != FGCN_MATH_F32 && mode != FGCN_MATH_BF16 && mode != FGCN_MATH_BF16X3)
        return fgcn::fail(FGCN_E_BADARG, "set_math_mode: %d is not one of FGCN_MATH_F32 / BF16 / BF16X3", mode);
    fgcn::cur().math_mode = mode;
    return FGCN_OK;
}

extern "C" int fgcn_get_math_mode(void) { return fgcn::cur().math_mode; }

extern "C" int fgcn_set_products(int products) {
    if (products != FGCN_PRODUCTS_BF16X3 && products != FGCN_PRODUCTS_F16X2)
        return fgcn::fail(FGCN_E_BADARG, "set_products: %d is not FGCN_PRODUCTS_BF16X3 / _F16X2", products);
    fgcn::cur().products = products;
    return FGCN_OK;
}

extern "C" int fgcn_get_products(void) { return fgcn::cur().products; }

extern "C" int fgcn_set_tuning(int key, int value) {
    if (key < 0 || key >= 32) return fgcn::fail(FGCN_E_BADARG, "set_tuning: key %d out of range", key);
    fgcn::cur().tuning[key] = value;
    return FGCN_OK;
}

extern "C" int fgcn_get_tuning(int key) { return fgcn::tuning(key); }

extern "C" int fgcn_version(void) { return 100; }  // 0.1.0

extern "C" const char* fgcn_last_error(void) { return fgcn::error_buffer(); }

extern "C" int fgcn_check_device(void) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fgcn::fail(FGCN_E_ARCH, "hipGetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return fgcn::fail(FGCN_E_ARCH, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fgcn::fail(FGCN_E_ARCH, "libfgcn is built for gfx950 (MI355X) only; device %d is %s", dev,
                          prop.gcnArchName);
    return FGCN_OK;
}
