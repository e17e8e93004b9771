// ps_env.cpp -- the one piece of process environment the library depends on, settled when the library is loaded.
//
// Every launch chain of the library (the chains of a PsBatchQueue, the lanes + upload + download streams of the pipelined
// stream) wants a hardware queue of its own.  The HIP runtime hands a process GPU_MAX_HW_QUEUES of them (default 4) and lets
// further streams SHARE a queue, which serialises them: two chains on one queue read 415 k frame-pairs/s instead of 560 k, the
// streamed form 130 k instead of 400 k (profiles/r05f/hw_queues.txt).  The runtime reads the variable once, when it initialises
// at the process' first HIP call, and offers no API for it -- so the library sets its default (16) from a constructor that runs
// when the shared object is loaded: before main() for a program that links it, before the first HIP call of any program that
// has not touched the GPU yet.  A value the host set itself is never overridden.  A process that initialised the HIP runtime
// BEFORE loading the library (Python: `import torch; torch.cuda.init()` first) has to set the variable itself;
// putslam_amd/_lib.py does so at import.  What the library found is readable as option "hw_queues_seen".
#include <cstdlib>

#include "ps_internal.h"

namespace {
int g_seen = 0;
int g_defaulted = 0;

__attribute__((constructor(101))) void ps_env_init()
{
    const char *v = std::getenv("GPU_MAX_HW_QUEUES");
    if (!v || !*v) {
        g_defaulted = setenv("GPU_MAX_HW_QUEUES", "16", 0) == 0 ? 1 : 0;
        v = std::getenv("GPU_MAX_HW_QUEUES");
    }
    g_seen = v ? std::atoi(v) : 0;
}
} // namespace

extern "C" {
int psi_hw_queues_seen(void) { return g_seen; }
int psi_hw_queues_defaulted(void) { return g_defaulted; }
}
