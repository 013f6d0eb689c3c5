// nb_comm.hip -- see nb_comm.h.
#include "nb_comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

namespace nb {

namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    std::string why;  // why loading failed
};

Rccl g_rccl;
std::once_flag g_once;

void load_rccl()
{
    Rccl &r = g_rccl;
    // by soname first: an RCCL this process already holds (PyTorch's) is shared, not doubled
    for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.lib)
            break;
    }
    if (!r.lib) {
        r.why = std::string("librccl.so.1 could not be loaded: ") + dlerror();
        return;
    }
    auto sym = [&](const char *n) {
        void *p = dlsym(r.lib, n);
        if (!p && r.why.empty())
            r.why = std::string("librccl lacks ") + n;
        return p;
    };
    r.get_unique_id = (decltype(r.get_unique_id))sym("ncclGetUniqueId");
    r.comm_init_rank = (decltype(r.comm_init_rank))sym("ncclCommInitRank");
    r.comm_destroy = (decltype(r.comm_destroy))sym("ncclCommDestroy");
    r.all_gather = (decltype(r.all_gather))sym("ncclAllGather");
    r.all_reduce = (decltype(r.all_reduce))sym("ncclAllReduce");
    r.error_string = (decltype(r.error_string))sym("ncclGetErrorString");
}

int rccl(const Rccl *&out)
{
    std::call_once(g_once, load_rccl);
    if (!g_rccl.why.empty())
        return fail(NB_ERR_STATE, "RCCL: " + g_rccl.why);
    out = &g_rccl;
    return NB_OK;
}

int check(const Rccl *r, ncclResult_t e, const char *what)
{
    if (e == ncclSuccess)
        return NB_OK;
    return fail(NB_ERR_HIP, std::string(what) + ": " + r->error_string(e));
}

}  // namespace

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

int comm_unique_id(void *id)
{
    static_assert(sizeof(ncclUniqueId) == COMM_ID_BYTES, "ncclUniqueId size");
    if (!id)
        return fail(NB_ERR_ARG, "comm_unique_id: id is NULL");
    const Rccl *r = nullptr;
    NB_TRY(rccl(r));
    ncclUniqueId u;
    NB_TRY(check(r, r->get_unique_id(&u), "ncclGetUniqueId"));
    std::memcpy(id, &u, sizeof(u));
    return NB_OK;
}

int comm_create(Comm **out, int rank, int world, const void *id)
{
    if (!out || !id || world < 1 || rank < 0 || rank >= world)
        return fail(NB_ERR_ARG, "comm_create: need an id and 0 <= rank < world");
    const Rccl *r = nullptr;
    NB_TRY(rccl(r));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    Comm *c = new Comm();
    c->rank = rank;
    c->world = world;
    const int rc = check(r, r->comm_init_rank(&c->comm, world, u, rank), "ncclCommInitRank");
    if (rc != NB_OK) {
        delete c;
        return rc;
    }
    *out = c;
    return NB_OK;
}

void comm_destroy(Comm *c)
{
    if (!c)
        return;
    const Rccl *r = nullptr;
    if (c->comm && rccl(r) == NB_OK)
        (void)r->comm_destroy(c->comm);
    delete c;
}

int comm_rank(const Comm *c) { return c->rank; }
int comm_world(const Comm *c) { return c->world; }

int comm_all_gather_f64(Comm *c, double *buf, size_t count, hipStream_t stream)
{
    const Rccl *r = nullptr;
    NB_TRY(rccl(r));
    // in place: the send buffer is this rank's slice of the receive buffer
    return check(r, r->all_gather(buf + (size_t)c->rank * count, buf, count, ncclDouble, c->comm, stream), "ncclAllGather");
}

int comm_all_reduce_max_f64(Comm *c, double *buf, size_t count, hipStream_t stream)
{
    const Rccl *r = nullptr;
    NB_TRY(rccl(r));
    return check(r, r->all_reduce(buf, buf, count, ncclDouble, ncclMax, c->comm, stream), "ncclAllReduce");
}

int comm_all_reduce_sum_f64(Comm *c, double *buf, size_t count, hipStream_t stream)
{
    const Rccl *r = nullptr;
    NB_TRY(rccl(r));
    return check(r, r->all_reduce(buf, buf, count, ncclDouble, ncclSum, c->comm, stream), "ncclAllReduce");
}

}  // namespace nb
