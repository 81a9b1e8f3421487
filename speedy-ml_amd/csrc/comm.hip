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
    double *send = nullptr, *stage = nullptr;          // padded contribution / gathered slabs of the ragged split
    size_t send_count = 0, stage_count = 0;
};

namespace {

// all_out[r][:] = stage[owner(r)][slot(r)][:]  (sml_domain_region_owner's rule)
__global__ void k_unpack_regions(const double *__restrict__ stage, int nranks, int slots, int nreg, int width, double *__restrict__ all_out)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)nreg * width) return;
    const int r = (int)(t / width), o = (int)(t % width);
    const int per = nreg / nranks, left = nreg % nranks;
    int rank, slot;
    if (r < per * nranks) { rank = r / per; slot = r % per; }
    else { rank = r - (nreg - left) + 1; slot = per; }
    all_out[t] = stage[((long)rank * slots + slot) * width + o];
}

}  // namespace

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
    if (c->send) (void)hipFree(c->send);
    if (c->stage) (void)hipFree(c->stage);
    delete c;
    return SML_OK;
}

int sml_comm_unpack_regions(const double *stage_dev, int nranks, int slots_per_rank, int number_of_regions, int max_n_out,
                            double *all_outvec_dev, void *stream)
{
    SML_REQUIRE(stage_dev && all_outvec_dev && nranks > 0 && number_of_regions > 0 && max_n_out > 0, "sml_comm_unpack_regions: bad arguments");
    const int per = number_of_regions / nranks, left = number_of_regions % nranks;
    SML_REQUIRE(slots_per_rank >= per + (left ? 1 : 0), "sml_comm_unpack_regions: %d slots per rank cannot hold %d regions on %d ranks", slots_per_rank,
                number_of_regions, nranks);
    const long total = (long)number_of_regions * max_n_out;
    hipLaunchKernelGGL(k_unpack_regions, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sml::as_stream(stream), stage_dev, nranks, slots_per_rank,
                       number_of_regions, max_n_out, all_outvec_dev);
    SML_HIP(hipGetLastError());
    return SML_OK;
}

int sml_comm_allgather_outvec(sml_comm *c, sml_bank *bank, int number_of_regions, double *all_outvec_dev, void *stream)
{
    SML_REQUIRE(c && bank && all_outvec_dev && number_of_regions > 0, "sml_comm_allgather_outvec: bad arguments");
    Rccl *r;
    int rc = load(&r);
    if (rc) return rc;
    hipStream_t st = sml::as_stream(stream);
    const int per = number_of_regions / c->nranks, left = number_of_regions % c->nranks;
    const int mine = per + ((c->rank >= 1 && c->rank <= left) ? 1 : 0);
    SML_REQUIRE(bank->capacity >= mine, "sml_comm_allgather_outvec: the bank holds %d slots, rank %d of %d owns %d of %d regions", bank->capacity, c->rank,
                c->nranks, mine, number_of_regions);
    const size_t width = (size_t)bank->max_n_out;
    if (left == 0 && bank->capacity == per)       // equal blocks: the gathered slab IS the region-ordered slab
        return check(r, r->all_gather(bank->d_outvec, all_outvec_dev, (size_t)per * width, NCCL_FLOAT64, c->comm, st), "ncclAllGather");
    // ragged (or an over-sized bank): every rank contributes per + 1 slots; its own first `mine` are real
    const int slots = per + 1;
    const size_t send_count = (size_t)slots * width, stage_count = send_count * c->nranks;
    if (c->send_count < send_count) {
        if (c->send) (void)hipFree(c->send);
        SML_HIP(hipMalloc((void **)&c->send, send_count * sizeof(double)));
        SML_HIP(hipMemset(c->send, 0, send_count * sizeof(double)));
        c->send_count = send_count;
    }
    if (c->stage_count < stage_count) {
        if (c->stage) (void)hipFree(c->stage);
        SML_HIP(hipMalloc((void **)&c->stage, stage_count * sizeof(double)));
        c->stage_count = stage_count;
    }
    SML_HIP(hipMemcpyAsync(c->send, bank->d_outvec, (size_t)mine * width * sizeof(double), hipMemcpyDeviceToDevice, st));
    if ((rc = check(r, r->all_gather(c->send, c->stage, send_count, NCCL_FLOAT64, c->comm, st), "ncclAllGather"))) return rc;
    return sml_comm_unpack_regions(c->stage, c->nranks, slots, number_of_regions, bank->max_n_out, all_outvec_dev, stream);
}

}  // extern "C"
