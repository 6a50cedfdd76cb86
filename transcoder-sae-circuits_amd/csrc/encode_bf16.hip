// bf16 encode path (v_mfma_f32_32x32x16_bf16): placeholder until the kernel lands; the C-ABI
// reports TSC_ERR_UNSUPPORTED for TSC_BF16 dictionaries instead of silently computing in fp32.
#include "tsc_internal.h"

int tsc_pack_w_bf16(const float *, int, int, void *, hipStream_t) {
    return tsc_fail(TSC_ERR_UNSUPPORTED, "the bf16 encode path is not built into this library");
}
int tsc_launch_encode_bf16(int, const EncodeArgs &, const void *, hipStream_t) {
    return tsc_fail(TSC_ERR_UNSUPPORTED, "the bf16 encode path is not built into this library");
}
