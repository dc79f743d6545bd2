// Built-in collective for sharded pricing: a sum-all-reduce of a handful of doubles over RCCL
// (xGMI inside a node).  librccl is dlopen'ed on first use so that single-GPU users and the
// CPU-only symbol check never need it.  Payloads are 3 doubles (payoff sums) or 3p+2 doubles per
// exercise date (LSM moments): latency-bound, so the call is issued on the ctx's compute stream and
// nothing on the host waits for it.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "mcg_internal.hpp"

namespace {

struct NcclId {
    char internal[128];
};
using comm_t = void*;

struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(comm_t*, int, NcclId, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*CommCount)(comm_t, int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

Rccl g_rccl;
std::once_flag g_once;
std::string g_load_err;

void load_rccl() {
    // MCG_RCCL_LIB (tests: a path that does not exist exercises the failure branch) replaces the search list
    const char* forced = std::getenv("MCG_RCCL_LIB");
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    std::string why = "?";
    for (const char* n : names) {
        g_rccl.lib = dlopen(forced ? forced : n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.lib) break;
        const char* e = dlerror();  // one call: dlerror() clears the message it returns
        if (e) why = e;
        if (forced) break;
    }
    if (!g_rccl.lib) {
        g_load_err = "cannot dlopen librccl: " + why;
        return;
    }
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(g_rccl.lib, "ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(g_rccl.lib, "ncclCommCount");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(g_rccl.lib, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce) {
        g_load_err = "librccl is missing ncclGetUniqueId/ncclCommInitRank/ncclAllReduce";
        g_rccl.lib = nullptr;
    }
}

bool have_rccl() {
    std::call_once(g_once, load_rccl);
    return g_rccl.lib != nullptr;
}

const char* nccl_err(int rc) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "rccl error"; }

constexpr int kNcclFloat64 = 8;  // ncclDouble
constexpr int kNcclSum = 0;

int rccl_allreduce(void* user, double* buf, int count, void* stream) {
    mcg_ctx* ctx = (mcg_ctx*)user;
    int rc = g_rccl.AllReduce(buf, buf, (size_t)count, kNcclFloat64, kNcclSum, (comm_t)ctx->rccl_comm,
                              (hipStream_t)stream);
    if (rc != 0) {
        mcg::set_error("ncclAllReduce failed: %s", nccl_err(rc));
        return 1;
    }
    return 0;
}

}  // namespace

namespace mcg {

// mcg_finalize: give the communicator back (after the stream has drained).
void comm_release(mcg_ctx* ctx) {
    if (!ctx->rccl_comm) return;
    if (g_rccl.lib && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((comm_t)ctx->rccl_comm);
    ctx->rccl_comm = nullptr;
    if (ctx->allreduce == rccl_allreduce) {
        ctx->allreduce = nullptr;
        ctx->allreduce_user = nullptr;
    }
}

int comm_rccl_count(mcg_ctx* ctx) {
    int n = 0;
    if (!ctx->rccl_comm || !g_rccl.CommCount || g_rccl.CommCount((comm_t)ctx->rccl_comm, &n) != 0) return 0;
    return n;
}

}  // namespace mcg

extern "C" {

int mcg_comm_unique_id(unsigned char id[128]) {
    if (!id) return mcg::fail(MCG_ERR_INVALID, "id is NULL");
    if (!have_rccl()) return mcg::fail(MCG_ERR_COMM, "%s", g_load_err.c_str());
    NcclId nid;
    int rc = g_rccl.GetUniqueId(&nid);
    if (rc != 0) return mcg::fail(MCG_ERR_COMM, "ncclGetUniqueId failed: %s", nccl_err(rc));
    std::memcpy(id, nid.internal, 128);
    return MCG_OK;
}

int mcg_comm_init_rank(mcg_ctx* ctx, const unsigned char id[128], int n_ranks, int rank) {
    if (!ctx || !id) return mcg::fail(MCG_ERR_INVALID, "ctx/id is NULL");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return mcg::fail(MCG_ERR_INVALID, "bad rank %d of %d", rank, n_ranks);
    if (!have_rccl()) return mcg::fail(MCG_ERR_COMM, "%s", g_load_err.c_str());
    MCG_HIP(hipSetDevice(ctx->device));
    NcclId nid;
    std::memcpy(nid.internal, id, 128);
    mcg::comm_release(ctx);  // a second init on the same ctx replaces the communicator
    mcg::shm_release(ctx);   // ... and the node-local one, mailbox included
    comm_t comm = nullptr;
    int rc = g_rccl.CommInitRank(&comm, n_ranks, nid, rank);
    if (rc != 0) return mcg::fail(MCG_ERR_COMM, "ncclCommInitRank failed: %s", nccl_err(rc));
    ctx->rccl_comm = comm;
    ctx->n_ranks = n_ranks;
    ctx->rank = rank;
    ctx->allreduce = rccl_allreduce;
    ctx->allreduce_user = ctx;
    return MCG_OK;
}

}  // extern "C"
