// libfgcn: version, error text, device check.
#include <cstring>

#include "fgcn_common.hpp"

namespace fgcn {
char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
static int g_tuning[32] = {1};            // key 0 defaults to 1, every other key to 0
int tuning(int key) { return (key >= 0 && key < 32) ? g_tuning[key] : 0; }
static int g_math_mode = FGCN_MATH_F32;
int math_mode() { return g_math_mode; }
static int g_products = FGCN_PRODUCTS_BF16X3;
int products() { return g_products; }
}  // namespace fgcn

extern "C" int fgcn_set_math_mode(int mode) {
    if (mode != FGCN_MATH_F32 && mode != FGCN_MATH_BF16 && mode != FGCN_MATH_BF16X3)
        return fgcn::fail(FGCN_E_BADARG, "set_math_mode: %d is not one of FGCN_MATH_F32 / BF16 / BF16X3", mode);
    fgcn::g_math_mode = mode;
    return FGCN_OK;
}

extern "C" int fgcn_get_math_mode(void) { return fgcn::g_math_mode; }

extern "C" int fgcn_set_products(int products) {
    if (products != FGCN_PRODUCTS_BF16X3 && products != FGCN_PRODUCTS_F16X2)
        return fgcn::fail(FGCN_E_BADARG, "set_products: %d is not FGCN_PRODUCTS_BF16X3 / _F16X2", products);
    fgcn::g_products = products;
    return FGCN_OK;
}

extern "C" int fgcn_get_products(void) { return fgcn::g_products; }

extern "C" int fgcn_set_tuning(int key, int value) {
    if (key < 0 || key >= 32) return fgcn::fail(FGCN_E_BADARG, "set_tuning: key %d out of range", key);
    fgcn::g_tuning[key] = value;
    return FGCN_OK;
}

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
