// Stand-alone instantiations of the per-hypothesis prologue (sample -> float Umeyama + Jacobi SVD -> general inverse) for
// profiles/isa_mix.py: compiled to assembly only (never run), one kernel per part, so that the parts can be counted.
#include "../../putslam_amd/csrc/ps_kernels.h"
using namespace psdev;
extern "C" __global__ void k_sample(uint64_t seed, uint32_t M, uint32_t *out)
{
    uint32_t idx[3];
    sample_triplet(seed, nullptr, threadIdx.x + blockIdx.x * 256, M, idx);
    out[threadIdx.x * 3] = idx[0]; out[threadIdx.x * 3 + 1] = idx[1]; out[threadIdx.x * 3 + 2] = idx[2];
}
extern "C" __global__ void k_umeyama(const float *in, float *out)
{
    float s[3][3], d[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { s[i][j] = in[threadIdx.x * 18 + i * 3 + j]; d[i][j] = in[threadIdx.x * 18 + 9 + i * 3 + j]; }
    Rigid m;
    bool ok = umeyama3(s, d, m);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) out[threadIdx.x * 13 + i * 3 + j] = m.R[i][j]; out[threadIdx.x * 13 + 9 + i] = m.t[i]; }
    out[threadIdx.x * 13 + 12] = ok;
}
extern "C" __global__ void k_inverse(const float *in, float *out)
{
    Rigid m, iv;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) m.R[i][j] = in[threadIdx.x * 12 + i * 3 + j]; m.t[i] = in[threadIdx.x * 12 + 9 + i]; }
    inverse_rigid_general(m, iv);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) out[threadIdx.x * 12 + i * 3 + j] = iv.R[i][j]; out[threadIdx.x * 12 + 9 + i] = iv.t[i]; }
}
