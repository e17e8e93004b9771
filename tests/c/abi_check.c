/* The public headers must be plain C (C99): this file is compiled with gcc -std=c99 -pedantic by tests/test_abi.py. */
#include "putslam_hip.h"
#include "putslam_shard.h"

int abi_check(void)
{
    PsDMatch m = {0, 1, 0, 2.0f};
    PsRansacParams p = {0, PS_REPROJECTION_ERROR, 0, 0, 0.04, 2.0, 0.0002, 0.2, 15, 3, 0};
    PsRansacConfig c = {PS_EST_RANSAC, 487, 42u, 0};
    PsRansacStats s;
    PsFrameSet f = {0, 0, 0, 0, 0};
    PsPairResults r = {0, 0, 0, 0, 0};
    PsHostPairResults h;
    PsShardRunParams sp;
    PsStreamResults mode = PS_RESULTS_INLIERS;
    int64_t lo, hi;
    (void)m; (void)p; (void)c; (void)s; (void)f; (void)r; (void)h; (void)sp; (void)mode; (void)lo; (void)hi;
    return (int)sizeof(PsDMatch) + PS_ABI_VERSION + PS_DESC_BYTES + PS_SHARD_RECORD_FLOATS + (int)PS_ERR_BUSY;
}
