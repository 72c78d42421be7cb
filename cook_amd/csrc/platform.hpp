// platform.hpp — the ONE header that knows the library is also compiled for the SIMT emulator of the test suite: it picks the
// implementation of the primitives (wave rendezvous, scoped accesses, DPP reductions, launch shapes, ...) the kernels are written
// against.  Every other source file is free of build-target conditionals.
#pragma once
#ifdef __HIP_EMU__
#include "emu_prims.hpp"  // tests/simt_emu/ (on the include path of the emulated build only): test infrastructure
#else
#include "gpu_prims.hpp"
#endif
