// dist.cpp — the ONE collective of the multi-GPU hot path, reachable from the C ABI / the C++ host: frames shard across
// the GPUs of a node with no steady-state exchange; every rank needs the same weight blob once, at start-up, and gets it by an RCCL
// broadcast over xGMI from the rank that loaded / generated it.  The reference has no multi-GPU inference path; this
// replaces "every process reads the model file" of a hand-rolled multi-process deployment.
// RCCL is bound at run time (dlopen of librccl.so on first use): single-GPU users of libhp_hip.so do not need it installed.
#include "hp_common.hpp"

#include <dlfcn.h>

#include <memory>
#include <mutex>
#include <string>

namespace {

// the handful of RCCL entry points used, with their ABI types restated (rccl.h: ncclUniqueId is 128 opaque bytes, ncclComm_t an
// opaque pointer, ncclFloat = 7, results are ints with 0 = success)
struct nccl_uid {
    char internal[128];
};
using get_uid_t = int (*)(nccl_uid*);
using init_rank_t = int (*)(void**, int, nccl_uid, int);
using bcast_t = int (*)(const void*, void*, size_t, int, int, void*, hipStream_t);
using destroy_t = int (*)(void*);
using errstr_t = const char* (*)(int);
constexpr int NCCL_FLOAT = 7;

struct rccl_api {
    void* lib = nullptr;
    get_uid_t get_uid = nullptr;
    init_rank_t init_rank = nullptr;
    bcast_t bcast = nullptr;
    destroy_t destroy = nullptr;
    errstr_t errstr = nullptr;
};

int load_rccl(rccl_api** out)
{
    static rccl_api api;
    static std::once_flag once;
    static bool ok = false;
    static std::string why = "missing symbols"; // dlerror() is consumed by its first call: captured once, here
    std::call_once(once, [] {
        for (const char* name : { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so" }) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.lib)
                break;
            const char* e = dlerror();
            why = e ? e : "dlopen failed";
        }
        if (!api.lib)
            return;
        why = "missing symbols";
        api.get_uid = (get_uid_t)dlsym(api.lib, "ncclGetUniqueId");
        api.init_rank = (init_rank_t)dlsym(api.lib, "ncclCommInitRank");
        api.bcast = (bcast_t)dlsym(api.lib, "ncclBroadcast");
        api.destroy = (destroy_t)dlsym(api.lib, "ncclCommDestroy");
        api.errstr = (errstr_t)dlsym(api.lib, "ncclGetErrorString");
        ok = api.get_uid && api.init_rank && api.bcast && api.destroy;
    });
    HP_REQUIRE(ok, HP_ERR_HIP, "hp_dist: librccl.so could not be loaded (%s)", why.c_str());
    *out = &api;
    return HP_OK;
}

#define HP_RCCL_TRY(api, expr)                                                                                    \
    do {                                                                                                          \
        const int _r = (expr);                                                                                    \
        if (_r != 0) {                                                                                            \
            hp::set_error("%s failed: %s", #expr, (api)->errstr ? (api)->errstr(_r) : "rccl error");              \
            return HP_ERR_HIP;                                                                                    \
        }                                                                                                         \
    } while (0)

} // namespace

struct hp_comm {
    void* comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;
    rccl_api* api = nullptr;
};

extern "C" {

int hp_dist_unique_id(char id[HP_DIST_ID_BYTES])
{
    HP_REQUIRE(id, HP_ERR_INVALID, "hp_dist_unique_id: null buffer");
    rccl_api* api = nullptr;
    HP_TRY(load_rccl(&api));
    nccl_uid u;
    HP_RCCL_TRY(api, api->get_uid(&u));
    memcpy(id, u.internal, sizeof(u.internal));
    return HP_OK;
}

int hp_dist_init(hp_comm** out, int rank, int world, const char id[HP_DIST_ID_BYTES])
{
    HP_REQUIRE(out && id && world >= 1 && rank >= 0 && rank < world, HP_ERR_INVALID, "hp_dist_init: bad rank %d / world %d", rank, world);
    rccl_api* api = nullptr;
    HP_TRY(load_rccl(&api));
    std::unique_ptr<hp_comm> c(new hp_comm());
    c->rank = rank, c->world = world, c->api = api;
    nccl_uid u;
    memcpy(u.internal, id, sizeof(u.internal));
    HP_RCCL_TRY(api, api->init_rank(&c->comm, world, u, rank)); // on the device hp_init() selected for this process
    HP_HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    *out = c.release();
    return HP_OK;
}

void hp_dist_destroy(hp_comm* c)
{
    if (!c)
        return;
    if (c->stream) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamDestroy(c->stream);
    }
    if (c->comm && c->api)
        (void)c->api->destroy(c->comm);
    delete c;
}

int hp_dist_broadcast_weights(hp_comm* c, float* host_weights, size_t n, int root)
{
    HP_REQUIRE(c && host_weights && n > 0 && root >= 0 && root < c->world, HP_ERR_INVALID, "hp_dist_broadcast_weights: bad argument");
    if (c->world == 1)
        return HP_OK;
    hp::dev_buf buf;
    HP_TRY(buf.alloc(n * sizeof(float)));
    if (c->rank == root)
        HP_HIP_TRY(hipMemcpyAsync(buf.p, host_weights, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HP_RCCL_TRY(c->api, c->api->bcast(buf.p, buf.p, n, NCCL_FLOAT, root, c->comm, c->stream));
    if (c->rank != root)
        HP_HIP_TRY(hipMemcpyAsync(host_weights, buf.p, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HP_HIP_TRY(hipStreamSynchronize(c->stream));
    return HP_OK;
}

void hp_dist_shard(int total_frames, int rank, int world, int* start, int* count)
{
    if (world < 1)
        world = 1;
    const int base = total_frames / world, rem = total_frames % world;
    if (start)
        *start = rank * base + (rank < rem ? rank : rem);
    if (count)
        *count = base + (rank < rem ? 1 : 0);
}

} // extern "C"
