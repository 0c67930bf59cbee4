// fake_rccl.cpp -- TEST INFRASTRUCTURE ONLY.  An in-process stand-in for the five RCCL calls libfuzzyhip
// opens with dlopen (FUZZYHIP_RCCL_LIB names this file): the "ranks" are THREADS of one process that share
// one GPU, so the sharded entry point (local top-k -> all-gather -> device merge over n_ranks > 1 lists) can
// be run on a one-GPU box, where RCCL itself refuses two ranks on one device.  Never loaded by the product
// unless that variable points here.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

struct ncclUniqueId { char internal[128]; };

struct Group {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    std::vector<const void *> send;
};
struct ncclComm { Group *g; int rank; };

static std::mutex g_mu;
static std::map<std::string, Group *> g_groups;
static uint64_t g_next = 1;

static void barrier(Group *g)
{
    std::unique_lock<std::mutex> lk(g->mu);
    const uint64_t gen = g->gen;
    if (++g->arrived == g->n) { g->arrived = 0; ++g->gen; g->cv.notify_all(); }
    else g->cv.wait(lk, [&] { return g->gen != gen; });
}

extern "C" int ncclGetUniqueId(ncclUniqueId *id)
{
    std::lock_guard<std::mutex> lk(g_mu);
    memset(id->internal, 0, sizeof id->internal);
    const uint64_t v = g_next++;
    memcpy(id->internal, &v, 8);
    memcpy(id->internal + 8, "fakerccl", 8);
    return 0;
}

extern "C" int ncclCommInitRank(ncclComm **out, int n, ncclUniqueId id, int rank)
{
    Group *g;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        Group *&slot = g_groups[std::string(id.internal, sizeof id.internal)];
        if (!slot) { slot = new Group(); slot->n = n; slot->send.assign((size_t)n, nullptr); }
        g = slot;
    }
    if (g->n != n || rank < 0 || rank >= n) return 4;          // ncclInvalidArgument
    *out = new ncclComm{g, rank};
    barrier(g);                                                // collective, as the real call
    return 0;
}

extern "C" int ncclCommDestroy(ncclComm *c) { delete c; return 0; }

extern "C" int ncclAllGather(const void *send, void *recv, size_t count, int dtype, ncclComm *c, hipStream_t st)
{
    if (dtype != 5) return 4;                                  // only ncclUint64 is used
    const size_t bytes = count * 8;
    Group *g = c->g;
    if (hipStreamSynchronize(st) != hipSuccess) return 1;      // the send buffer is complete
    { std::lock_guard<std::mutex> lk(g->mu); g->send[(size_t)c->rank] = send; }
    barrier(g);
    for (int r = 0; r < g->n; ++r)
        if (hipMemcpyAsync(static_cast<char *>(recv) + (size_t)r * bytes, g->send[(size_t)r], bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return 1;
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
    barrier(g);                                                // nobody reuses its send buffer before all have read it
    return 0;
}

extern "C" const char *ncclGetErrorString(int r) { return r == 0 ? "no error" : (r == 4 ? "invalid argument (fake rccl)" : "unhandled device error (fake rccl)"); }
