// ps_internal.h -- what the translation units of libputslam_hip.so share beyond the public C ABI (include/putslam_hip.h).
// Not installed, not part of the ABI: the names start with psi_ and may change between versions.
#pragma once
#include "putslam_hip.h"

extern "C" {
// Error text of a context (what ps_last_error returns), set from another translation unit.
void psi_set_error(PsContext *ctx, const char *what);
// Every settable option of `src` (kernel variants, staged-scoring knobs) copied to `dst`: lanes of the pipelined stream and
// chains of a batch queue run what their parent context would.
void psi_copy_options(PsContext *dst, const PsContext *src);
// GPU_MAX_HW_QUEUES as the process environment held it when the library was loaded (after the library's own default, ps_env.cpp):
// the number of hardware queues the HIP runtime gives the process' streams; 0 = the variable held no number.
int psi_hw_queues_seen(void);
// 1 if the library set the variable itself (it was unset when the library was loaded).
int psi_hw_queues_defaulted(void);
}
