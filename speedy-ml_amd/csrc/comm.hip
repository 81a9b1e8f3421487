// RCCL from the C-ABI, for hosts that are not Python (a multi-rank Fortran driver under MPI): the one data-path collective of
// the hybrid step -- every rank needs every region's outvec (the MPI gather-to-root + root-side tiling of
// src/mpires.f90:347-454 becomes one all-gather of the bank's contiguous outvec slab over xGMI).
//
// librccl is resolved with dlopen at the first call, not at link time: inside a Python process torch has already loaded its
// own copy and a second, link-time copy of the library would give the process two RCCL runtimes.
#include <dlfcn.h>

#include "bank.h"

namespace {

constexpr int ID_BYTES = 128;                 // NCCL_UNIQUE_ID_BYTES
struct UniqueId { char internal[ID_BYTES]; };
typedef int (*fn_get_id)(UniqueId *);
typedef int (*fn_init_rank)(void **, int, UniqueId, int);
typedef int (*fn_all_gather)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*fn_destroy)(void *);
typedef const char *(*fn_err)(int);
constexpr int NCCL_FLOAT64 = 8;               // ncclDouble in rccl.h's ncclDataType_t

struct Rccl {
    void *lib = nullptr;
    fn_get_id get_id = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_destroy destroy = nullptr;
    fn_err err = nullptr;
};

int load(Rccl **out)
{
    static Rccl r;
    if (!r.lib) {
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) return sml::fail(SML_ERR_STATE, "librccl.so cannot be loaded: %s", dlerror());
        r.get_id = (fn_get_id)dlsym(r.lib, "ncclGetUniqueId");
        r.init_rank = (fn_init_rank)dlsym(r.lib, "ncclCommInitRank");
        r.all_gather = (fn_all_gather)dlsym(r.lib, "ncclAllGather");
        r.destroy = (fn_destroy)dlsym(r.lib, "ncclCommDestroy");
        r.err = (fn_err)dlsym(r.lib, "ncclGetErrorString");
        if (!r.get_id || !r.init_rank || !r.all_gather || !r.destroy) {
            r.lib = nullptr;
            return sml::fail(SML_ERR_STATE, "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy");
        }
    }
    *out = &r;
    return SML_OK;
}

int check(Rccl *r, int rc, const char *what)
{
    if (rc == 0) return SML_OK;
    return sml::fail(SML_ERR_HIP, "%s failed: %s", what, r->err ? r->err(rc) : "RCCL error");
}

}  // namespace

struct sml_comm {
    void *comm = nullptr;
    int nranks = 0, rank = 0;
};

extern "C" {

int sml_comm_unique_id(char *id128)
{
    SML_REQUIRE(id128, "sml_comm_unique_id: null pointer");
    Rccl *r;
    int rc = load(&r);
    if (rc) return rc;
    UniqueId id;
    if ((rc = check(r, r->get_id(&id), "ncclGetUniqueId"))) return rc;
    memcpy(id128, id.internal, ID_BYTES);
    return SML_OK;
}

int sml_comm_create(int nranks, int rank, const char *id128, sml_comm **out)
{
    SML_REQUIRE(out && id128 && nranks >= 1 && rank >= 0 && rank < nranks, "sml_comm_create: bad arguments");
    Rccl *r;
    int rc = load(&r);
    if (rc) return rc;
    UniqueId id;
    memcpy(id.internal, id128, ID_BYTES);
    sml_comm *c = new sml_comm;
    c->nranks = nranks; c->rank = rank;
    if ((rc = check(r, r->init_rank(&c->comm, nranks, id, rank), "ncclCommInitRank"))) { delete c; return rc; }
    *out = c;
    return SML_OK;
}

int sml_comm_destroy(sml_comm *c)
{
    if (!c) return SML_OK;
    Rccl *r;
    if (load(&r) == SML_OK && c->comm) (void)r->destroy(c->comm);
    delete c;
    return SML_OK;
}

int sml_comm_allgather_outvec(sml_comm *c, sml_bank *bank, double *all_outvec_dev, void *stream)
{
    SML_REQUIRE(c && bank && all_outvec_dev, "sml_comm_allgather_outvec: bad arguments");
    Rccl *r;
    int rc = load(&r);
    if (rc) return rc;
    // every rank contributes its whole [capacity][max_n_out] slab: with processor_decomposition's equal blocks (the region count
    // divides by the rank count) the result IS the region-ordered slab sml_exchange_scatter wants
    const size_t count = (size_t)bank->capacity * bank->max_n_out;
    return check(r, r->all_gather(bank->d_outvec, all_outvec_dev, count, NCCL_FLOAT64, c->comm, sml::as_stream(stream)), "ncclAllGather");
}

}  // extern "C"
