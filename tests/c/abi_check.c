/* The public header must be plain C (C99): this file is compiled with gcc -std=c99 -pedantic by tests/test_abi.py. */
#include "putslam_hip.h"

int abi_check(void)
{
    PsDMatch m = {0, 1, 0, 2.0f};
    PsRansacParams p = {0, PS_REPROJECTION_ERROR, 0, 0, 0.04, 2.0, 0.0002, 0.2, 15, 3, 0};
    PsRansacConfig c = {PS_EST_RANSAC, 487, 42u, 0};
    PsRansacStats s;
    PsFrameSet f = {0, 0, 0, 0, 0};
    PsPairResults r = {0, 0, 0, 0, 0};
    (void)m; (void)p; (void)c; (void)s; (void)f; (void)r;
    return (int)sizeof(PsDMatch) + PS_ABI_VERSION + PS_DESC_BYTES;
}
