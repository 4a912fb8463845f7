// TEST INFRASTRUCTURE: lets `#include <hip/hip_runtime.h>` resolve to the CPU SIMT emulator
// when the kernel sources are compiled with g++ for logic tests (see ../hipemu.h).
#pragma once
#include "../hipemu.h"
