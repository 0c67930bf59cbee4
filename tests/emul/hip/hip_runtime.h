// stand-in for <hip/hip_runtime.h> in the sanitizer build of tests/emul (TEST INFRASTRUCTURE ONLY)
#pragma once
#include "../hip_emul.h"
