// pigs_comm.cpp -- thin RCCL binding: one ncclAllReduce(sum, fp64) per block over the
// concatenated estimator vector (a few KB: latency-bound over xGMI, ring/tree irrelevant).
#include "pigs_comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

struct pigs_comm {
    ncclComm_t comm;
};

namespace {

struct Api {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Api g_api;
std::once_flag g_once;
thread_local char g_msg[256];

const char *load()
{
    std::call_once(g_once, [] {
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) {
            g_api.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (g_api.h) break;
        }
        if (!g_api.h) return;
        g_api.GetUniqueId    = (decltype(g_api.GetUniqueId))dlsym(g_api.h, "ncclGetUniqueId");
        g_api.CommInitRank   = (decltype(g_api.CommInitRank))dlsym(g_api.h, "ncclCommInitRank");
        g_api.CommInitAll    = (decltype(g_api.CommInitAll))dlsym(g_api.h, "ncclCommInitAll");
        g_api.AllReduce      = (decltype(g_api.AllReduce))dlsym(g_api.h, "ncclAllReduce");
        g_api.CommDestroy    = (decltype(g_api.CommDestroy))dlsym(g_api.h, "ncclCommDestroy");
        g_api.GetErrorString = (decltype(g_api.GetErrorString))dlsym(g_api.h, "ncclGetErrorString");
    });
    if (!g_api.h) return "librccl.so could not be loaded";
    if (!g_api.GetUniqueId || !g_api.CommInitRank || !g_api.CommInitAll || !g_api.AllReduce || !g_api.CommDestroy)
        return "librccl.so lacks a required symbol";
    return nullptr;
}

const char *err(const char *what, ncclResult_t r)
{
    snprintf(g_msg, sizeof g_msg, "%s: %s", what, g_api.GetErrorString ? g_api.GetErrorString(r) : "rccl error");
    return g_msg;
}

} // namespace

const char *pigs_comm_get_unique_id(char id[128])
{
    if (const char *e = load()) return e;
    ncclUniqueId u;
    ncclResult_t r = g_api.GetUniqueId(&u);
    if (r != ncclSuccess) return err("ncclGetUniqueId", r);
    static_assert(sizeof(u) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id, &u, 128);
    return nullptr;
}

const char *pigs_comm_create_rank(pigs_comm **out, int nranks, int rank, const char id[128])
{
    if (const char *e = load()) return e;
    ncclUniqueId u;
    memcpy(&u, id, 128);
    ncclComm_t c;
    ncclResult_t r = g_api.CommInitRank(&c, nranks, u, rank);
    if (r != ncclSuccess) return err("ncclCommInitRank", r);
    *out = new pigs_comm{c};
    return nullptr;
}

const char *pigs_comm_create_all(pigs_comm **out, int nranks, const int *devices)
{
    if (const char *e = load()) return e;
    std::vector<ncclComm_t> cs(nranks);
    ncclResult_t r = g_api.CommInitAll(cs.data(), nranks, devices);
    if (r != ncclSuccess) return err("ncclCommInitAll", r);
    for (int i = 0; i < nranks; ++i) out[i] = new pigs_comm{cs[i]};
    return nullptr;
}

const char *pigs_comm_allreduce_sum_f64(pigs_comm *c, double *d_buf, int n, hipStream_t s)
{
    if (const char *e = load()) return e;
    ncclResult_t r = g_api.AllReduce(d_buf, d_buf, (size_t)n, ncclDouble, ncclSum, c->comm, s);
    if (r != ncclSuccess) return err("ncclAllReduce", r);
    return nullptr;
}

void pigs_comm_destroy(pigs_comm *c)
{
    if (!c) return;
    if (g_api.CommDestroy) (void)g_api.CommDestroy(c->comm);
    delete c;
}
