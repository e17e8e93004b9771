/* The public headers must be plain C (C99): this file is compiled with gcc -std=c99 -pedantic by tests/test_abi.py. */
#include "putslam_hip.h"
#include "putslam_shard.h"

int abi_check(void)
{
    PsDMatch m = {0, 1, 0, 2.0f};
    PsRansacParams p = {0, PS_REPROJECTION_ERROR, 0, 0, 0.04, 2.0, 0.0002, 0.2, 15, 3, 0};
    PsRansacConfig c = {PS_EST_RANSAC, 487, 42u, 0};
    PsRansacStats s;
    PsFrameSet f = {0, 0, 0, 0, 0, 0, 0};
    PsPairResults r = {0, 0, 0, 0, 0};
    PsHostPairResults h;
    PsShardRunParams sp;
    PsShardJob job = {0, 0, 0, 0, 0, 0, 0};
    PsBatchQueue *q = 0;
    int64_t ticket = -1;
    int (*submit)(PsBatchQueue *, const PsRansacParams *, const PsRansacConfig *, const float *, const PsFrameSet *, const int32_t *, int,
                  const PsPairResults *, int64_t *) = ps_batch_queue_submit;
    int (*gather)(PsShardGroup *, int, int, int64_t *) = ps_shard_gather_records_async;
    PsStreamResults mode = PS_RESULTS_INLIERS;
    int64_t lo, hi;
    (void)m; (void)p; (void)c; (void)s; (void)f; (void)r; (void)h; (void)sp; (void)job; (void)q; (void)ticket; (void)submit; (void)gather; (void)mode; (void)lo; (void)hi;
    return (int)sizeof(PsDMatch) + PS_ABI_VERSION + PS_DESC_BYTES + PS_SHARD_RECORD_FLOATS + PS_SHARD_GATHERS_IN_FLIGHT + (int)PS_ERR_BUSY;
}
