#include "../../include/lgen.h"
extern "C" int lgen_abi_version(void) { return LGEN_ABI_VERSION; }
