// ps_shard.hip -- include/putslam_shard.h: sharding over the GPUs of one node for C / C++ hosts, RCCL underneath.
//
// The exchanges are those of bench.py / putslam_amd/sharding.py: one broadcast of the run's parameter
// block, one gather of 72-byte per-pair records to the rank that composes the trajectories (the reference's only sequential
// step, src/PUTSLAM/PUTSLAM.cpp:735-740).  Collectives are queued on the members' own context streams, behind the kernels that
// produce what they send.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "putslam_shard.h"

static_assert(PS_SHARD_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");

namespace {

struct Member {
    int device = 0, rank = 0;
    PsContext *ctx = nullptr;
    ncclComm_t comm = nullptr;
    float *rec = nullptr;      // device: this member's packed records
    size_t recCap = 0;
    float *gathered = nullptr; // device, root only: [world][pairs][18]
    size_t gatheredCap = 0;
    uint8_t *blob = nullptr;   // device: parameter block
};

__global__ void ps_pack_records(const float *__restrict__ pose, const PsRansacStats *__restrict__ stats, int valid, int pairs,
                                float *__restrict__ rec)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pairs) return;
    float *r = rec + (size_t)p * PS_SHARD_RECORD_FLOATS;
    if (p < valid) {
        for (int i = 0; i < 16; ++i) r[i] = pose[(size_t)p * 16 + i];
        r[16] = (float)stats[p].numInliers;
        r[17] = (float)stats[p].numMatchesIn;
    } else {
        for (int i = 0; i < PS_SHARD_RECORD_FLOATS; ++i) r[i] = 0.0f;
    }
}

} // namespace

struct PsShardGroup {
    int world = 0;
    std::vector<Member> m;
    std::string err;
};

namespace {

int sfail(PsShardGroup *g, int code, const char *what, const char *detail = nullptr)
{
    if (g) {
        g->err = what;
        if (detail) {
            g->err += ": ";
            g->err += detail;
        }
    }
    return code;
}

#define SH_HIP(call)                                                                  \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) return sfail(g, PS_ERR_HIP, #call, hipGetErrorString(e_)); \
    } while (0)
#define SH_NCCL(call)                                                                   \
    do {                                                                                \
        ncclResult_t r_ = (call);                                                       \
        if (r_ != ncclSuccess) return sfail(g, PS_ERR_HIP, #call, ncclGetErrorString(r_)); \
    } while (0)

int member_buffers(PsShardGroup *g, Member &mb)
{
    SH_HIP(hipSetDevice(mb.device));
    SH_HIP(hipMalloc((void **)&mb.blob, sizeof(PsShardRunParams)));
    return PS_OK;
}

hipStream_t mstream(const Member &mb) { return (hipStream_t)ps_context_stream(mb.ctx); }

int find_root(const PsShardGroup *g, int root)
{
    for (size_t i = 0; i < g->m.size(); ++i)
        if (g->m[i].rank == root) return (int)i;
    return -1;
}

} // namespace

extern "C" {

int ps_shard_group_create(const int *devices, int numDevices, PsShardGroup **out)
{
    if (!out) return PS_ERR_BAD_ARG;
    *out = nullptr;
    int avail = 0;
    if (hipGetDeviceCount(&avail) != hipSuccess || avail <= 0) return PS_ERR_NO_DEVICE; // no CPU fallback
    if (numDevices < 1 || numDevices > 64) return PS_ERR_BAD_ARG;
    std::vector<int> dev((size_t)numDevices);
    for (int i = 0; i < numDevices; ++i) {
        dev[(size_t)i] = devices ? devices[i] : i;
        if (dev[(size_t)i] < 0 || dev[(size_t)i] >= avail) return PS_ERR_BAD_ARG;
    }
    PsShardGroup *g = new PsShardGroup();
    g->world = numDevices;
    g->m.resize((size_t)numDevices);
    std::vector<ncclComm_t> comms((size_t)numDevices, nullptr);
    ncclResult_t r = ncclCommInitAll(comms.data(), numDevices, dev.data());
    if (r != ncclSuccess) {
        fprintf(stderr, "ps_shard_group_create: ncclCommInitAll: %s\n", ncclGetErrorString(r));
        delete g;
        return PS_ERR_HIP;
    }
    for (int i = 0; i < numDevices; ++i) {
        Member &mb = g->m[(size_t)i];
        mb.device = dev[(size_t)i];
        mb.rank = i;
        mb.comm = comms[(size_t)i];
        int rc = ps_context_create(mb.device, &mb.ctx);
        if (rc == PS_OK) rc = member_buffers(g, mb);
        if (rc != PS_OK) {
            ps_shard_group_destroy(g);
            return rc;
        }
    }
    *out = g;
    return PS_OK;
}

int ps_shard_unique_id(uint8_t id[PS_SHARD_ID_BYTES])
{
    if (!id) return PS_ERR_BAD_ARG;
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return PS_ERR_HIP;
    memcpy(id, u.internal, PS_SHARD_ID_BYTES);
    return PS_OK;
}

int ps_shard_group_create_rank(int device, int rank, int worldSize, const uint8_t id[PS_SHARD_ID_BYTES], PsShardGroup **out)
{
    if (!out || !id || worldSize < 1 || rank < 0 || rank >= worldSize) return PS_ERR_BAD_ARG;
    *out = nullptr;
    int avail = 0;
    if (hipGetDeviceCount(&avail) != hipSuccess || avail <= 0) return PS_ERR_NO_DEVICE;
    if (device < 0 || device >= avail) return PS_ERR_BAD_ARG;
    if (hipSetDevice(device) != hipSuccess) return PS_ERR_HIP;
    PsShardGroup *g = new PsShardGroup();
    g->world = worldSize;
    g->m.resize(1);
    Member &mb = g->m[0];
    mb.device = device;
    mb.rank = rank;
    ncclUniqueId u;
    memcpy(u.internal, id, PS_SHARD_ID_BYTES);
    ncclResult_t r = ncclCommInitRank(&mb.comm, worldSize, u, rank);
    if (r != ncclSuccess) {
        fprintf(stderr, "ps_shard_group_create_rank: ncclCommInitRank: %s\n", ncclGetErrorString(r));
        mb.comm = nullptr;
        delete g;
        return PS_ERR_HIP;
    }
    int rc = ps_context_create(device, &mb.ctx);
    if (rc == PS_OK) rc = member_buffers(g, mb);
    if (rc != PS_OK) {
        ps_shard_group_destroy(g);
        return rc;
    }
    *out = g;
    return PS_OK;
}

void ps_shard_group_destroy(PsShardGroup *g)
{
    if (!g) return;
    for (Member &mb : g->m) {
        (void)hipSetDevice(mb.device);
        if (mb.ctx) (void)ps_context_synchronize(mb.ctx);
        if (mb.comm) (void)ncclCommDestroy(mb.comm);
        if (mb.rec) (void)hipFree(mb.rec);
        if (mb.gathered) (void)hipFree(mb.gathered);
        if (mb.blob) (void)hipFree(mb.blob);
        if (mb.ctx) ps_context_destroy(mb.ctx);
    }
    delete g;
}

const char *ps_shard_last_error(const PsShardGroup *g) { return g ? g->err.c_str() : "null group"; }
int ps_shard_world_size(const PsShardGroup *g) { return g ? g->world : (int)PS_ERR_BAD_ARG; }
int ps_shard_local_count(const PsShardGroup *g) { return g ? (int)g->m.size() : (int)PS_ERR_BAD_ARG; }
int ps_shard_rank(const PsShardGroup *g, int local)
{
    return (g && local >= 0 && local < (int)g->m.size()) ? g->m[(size_t)local].rank : (int)PS_ERR_BAD_ARG;
}
int ps_shard_device(const PsShardGroup *g, int local)
{
    return (g && local >= 0 && local < (int)g->m.size()) ? g->m[(size_t)local].device : (int)PS_ERR_BAD_ARG;
}
PsContext *ps_shard_context(PsShardGroup *g, int local)
{
    return (g && local >= 0 && local < (int)g->m.size()) ? g->m[(size_t)local].ctx : nullptr;
}

void ps_shard_range(int64_t total, int world, int rank, int64_t *lo, int64_t *hi)
{
    if (world < 1) world = 1;
    const int64_t base = total / world, rem = total % world;
    const int64_t l = rank * base + (rank < rem ? rank : rem);
    if (lo) *lo = l;
    if (hi) *hi = l + base + (rank < rem ? 1 : 0);
}

int ps_shard_broadcast_params(PsShardGroup *g, PsShardRunParams *perLocal, int root)
{
    if (!g || !perLocal || root < 0 || root >= g->world) return sfail(g, PS_ERR_BAD_ARG, "ps_shard_broadcast_params: bad argument");
    const int rl = find_root(g, root);
    if (rl >= 0) {
        Member &mb = g->m[(size_t)rl];
        SH_HIP(hipSetDevice(mb.device));
        SH_HIP(hipMemcpyAsync(mb.blob, &perLocal[rl], sizeof(PsShardRunParams), hipMemcpyHostToDevice, mstream(mb)));
    }
    SH_NCCL(ncclGroupStart());
    for (Member &mb : g->m) {
        ncclResult_t r = ncclBroadcast(mb.blob, mb.blob, sizeof(PsShardRunParams), ncclUint8, root, mb.comm, mstream(mb));
        if (r != ncclSuccess) {
            (void)ncclGroupEnd();
            return sfail(g, PS_ERR_HIP, "ncclBroadcast", ncclGetErrorString(r));
        }
    }
    SH_NCCL(ncclGroupEnd());
    for (size_t i = 0; i < g->m.size(); ++i) {
        Member &mb = g->m[i];
        SH_HIP(hipSetDevice(mb.device));
        SH_HIP(hipMemcpyAsync(&perLocal[i], mb.blob, sizeof(PsShardRunParams), hipMemcpyDeviceToHost, mstream(mb)));
        SH_HIP(hipStreamSynchronize(mstream(mb)));
    }
    return PS_OK;
}

int ps_shard_gather_records(PsShardGroup *g, const PsPairResults *results, const int32_t *validPairs, int pairsPerRank,
                            float *hostRecords, int root)
{
    if (!g || !results || pairsPerRank < 0 || root < 0 || root >= g->world)
        return sfail(g, PS_ERR_BAD_ARG, "ps_shard_gather_records: bad argument");
    if (pairsPerRank == 0) return PS_OK;
    const int rl = find_root(g, root);
    if (rl >= 0 && !hostRecords) return sfail(g, PS_ERR_BAD_ARG, "ps_shard_gather_records: this process drives the root and needs hostRecords");
    const size_t recFloats = (size_t)pairsPerRank * PS_SHARD_RECORD_FLOATS;
    for (size_t i = 0; i < g->m.size(); ++i) {
        Member &mb = g->m[i];
        SH_HIP(hipSetDevice(mb.device));
        if (mb.recCap < recFloats) {
            SH_HIP(hipStreamSynchronize(mstream(mb)));
            if (mb.rec) (void)hipFree(mb.rec);
            mb.rec = nullptr;
            SH_HIP(hipMalloc((void **)&mb.rec, recFloats * sizeof(float)));
            mb.recCap = recFloats;
        }
        if ((int)i == rl && mb.gatheredCap < recFloats * (size_t)g->world) {
            SH_HIP(hipStreamSynchronize(mstream(mb)));
            if (mb.gathered) (void)hipFree(mb.gathered);
            mb.gathered = nullptr;
            SH_HIP(hipMalloc((void **)&mb.gathered, recFloats * (size_t)g->world * sizeof(float)));
            mb.gatheredCap = recFloats * (size_t)g->world;
        }
        const int valid = validPairs ? validPairs[i] : pairsPerRank;
        if (valid < 0 || valid > pairsPerRank || !results[i].pose || !results[i].stats)
            return sfail(g, PS_ERR_BAD_ARG, "ps_shard_gather_records: bad results / validPairs");
        hipLaunchKernelGGL(ps_pack_records, dim3((unsigned)((pairsPerRank + 255) / 256)), dim3(256), 0, mstream(mb), results[i].pose,
                           results[i].stats, valid, pairsPerRank, mb.rec);
        SH_HIP(hipGetLastError());
    }
    SH_NCCL(ncclGroupStart());
    for (size_t i = 0; i < g->m.size(); ++i) {
        Member &mb = g->m[i];
        ncclResult_t r = ncclGather(mb.rec, (int)i == rl ? mb.gathered : nullptr, recFloats, ncclFloat, root, mb.comm, mstream(mb));
        if (r != ncclSuccess) {
            (void)ncclGroupEnd();
            return sfail(g, PS_ERR_HIP, "ncclGather", ncclGetErrorString(r));
        }
    }
    SH_NCCL(ncclGroupEnd());
    if (rl >= 0) {
        Member &mb = g->m[(size_t)rl];
        SH_HIP(hipSetDevice(mb.device));
        SH_HIP(hipMemcpyAsync(hostRecords, mb.gathered, recFloats * (size_t)g->world * sizeof(float), hipMemcpyDeviceToHost, mstream(mb)));
    }
    for (Member &mb : g->m) {
        SH_HIP(hipSetDevice(mb.device));
        SH_HIP(hipStreamSynchronize(mstream(mb)));
    }
    return PS_OK;
}

int ps_shard_synchronize(PsShardGroup *g)
{
    if (!g) return PS_ERR_BAD_ARG;
    for (Member &mb : g->m) {
        SH_HIP(hipSetDevice(mb.device));
        SH_HIP(hipStreamSynchronize(mstream(mb)));
    }
    return PS_OK;
}

} // extern "C"
