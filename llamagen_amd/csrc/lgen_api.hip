#include "../../include/lgen.h"
extern "C" int lgen_abi_version(void) { return LGEN_ABI_VERSION; }

// Lane streams with a CU mask (see lgen.h): thin wrappers so that the host code stays on ONE HIP runtime
// (the one this library and torch share) instead of dlopen-ing another copy through ctypes.
#include <hip/hip_runtime.h>
extern "C" int lgen_stream_create_cu_mask(const unsigned int* mask_words, int n_words, void** stream_out) {
    if (!mask_words || n_words < 1 || !stream_out) return LGEN_ERR_BAD_ARG;
    hipStream_t s = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask_words);
    if (e != hipSuccess) return (int)e;
    *stream_out = (void*)s;
    return 0;
}
extern "C" int lgen_stream_destroy(void* stream) {
    if (!stream) return LGEN_ERR_BAD_ARG;
    return (int)hipStreamDestroy((hipStream_t)stream);
}
