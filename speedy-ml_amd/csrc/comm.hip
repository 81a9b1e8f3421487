// RCCL from the C-ABI, for hosts that are not Python (a multi-rank Fortran driver under MPI): the one data-path collective of
// the hybrid step -- every rank needs every region's outvec (the MPI gather-to-root + root-side tiling of
// src/mpires.f90:347-454 becomes one all-gather of the bank's contiguous outvec slab over xGMI).
//
// librccl is resolved with dlopen at the first call, not at link time: inside a Python process torch has already loaded its
// own copy and a second, link-time copy of the library would give the process two RCCL runtimes.
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

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

// Host-staged transport (rehearsal only): the ranks of one node meet in a POSIX shared-memory segment.  RCCL refuses two ranks on
// one device, so a box with ONE GPU can run the multi-rank path only this way (as bench.py's SML_DIST_BACKEND=gloo does for the
// Python host); numbers are never taken with it.
struct ShmHeader {
    std::atomic<int> magic, arrive, generation, id_ready;
    char id[128];
    unsigned long long doubles_per_rank;
    unsigned long long nonce;                     // the launch this segment belongs to (see launch_nonce)
};
constexpr int SHM_MAGIC = 0x534d4c43;

struct sml_comm {
    void *comm = nullptr;
    int nranks = 0, rank = 0;
    double *send = nullptr, *stage = nullptr;          // padded contribution / gathered slabs of the ragged split
    size_t send_count = 0, stage_count = 0;
    // shm transport
    ShmHeader *hdr = nullptr;
    double *shm_data = nullptr;
    size_t shm_bytes = 0;
    std::string shm_name;
};

namespace {

// A rendezvous name is reused from run to run (the Fortran host's default is a constant), so what is found under it may be the
// leftover of an earlier -- possibly crashed -- run.  Every published id file / segment therefore carries a per-launch token and a
// peer ignores one whose token is not its own: SML_COMM_NONCE when the launcher exports one (a job id), else the parent process id,
// which the ranks of one launch share (children of one mpirun / torchrun / shell) and two launches do not.
unsigned long long launch_nonce()
{
    const char *e = getenv("SML_COMM_NONCE");
    if (e && *e) return strtoull(e, nullptr, 0) ^ 0x9e3779b97f4a7c15ull;
    return (unsigned long long)getppid();
}

int rendezvous_timeout_s()
{
    const char *e = getenv("SML_COMM_TIMEOUT_S");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 120;
}

// names this process has published and not yet withdrawn: removed at exit, so that a host which never reaches sml_comm_destroy (the
// reference's program main ends with mpi_finalize, not killmpi) or dies between publishing and the collective init leaves nothing behind
std::mutex g_pending_mu;
std::vector<std::string> g_pending_files, g_pending_shm;
void withdraw_all()
{
    std::lock_guard<std::mutex> lk(g_pending_mu);
    for (const std::string &f : g_pending_files) (void)unlink(f.c_str());
    for (const std::string &n : g_pending_shm) (void)shm_unlink(n.c_str());
    g_pending_files.clear(); g_pending_shm.clear();
}
void publish(std::vector<std::string> &list, const std::string &name)
{
    static std::once_flag once;
    std::call_once(once, [] { atexit(withdraw_all); });
    std::lock_guard<std::mutex> lk(g_pending_mu);
    list.push_back(name);
}
void withdraw(std::vector<std::string> &list, const std::string &name, bool is_shm)
{
    std::lock_guard<std::mutex> lk(g_pending_mu);
    for (size_t i = 0; i < list.size(); ++i)
        if (list[i] == name) {
            if (is_shm) (void)shm_unlink(name.c_str()); else (void)unlink(name.c_str());
            list.erase(list.begin() + i);
            return;
        }
}

int shm_barrier(sml_comm *c)
{
    ShmHeader *h = c->hdr;
    const int g = h->generation.load();
    if (h->arrive.fetch_add(1) + 1 == c->nranks) {
        h->arrive.store(0);
        h->generation.fetch_add(1);
        return SML_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (h->generation.load() == g) {
        sched_yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(rendezvous_timeout_s()))
            return sml::fail(SML_ERR_STATE, "sml_comm (shm): rank %d waited %d s for its peers at a barrier", c->rank, rendezvous_timeout_s());
    }
    return SML_OK;
}

// every rank's `count` doubles -> [nranks][count] on every rank, through the host
int shm_all_gather(sml_comm *c, const double *send_dev, double *recv_dev, size_t count, hipStream_t st)
{
    if (count > c->hdr->doubles_per_rank) return sml::fail(SML_ERR_ARG, "sml_comm (shm): %zu doubles per rank exceed the segment's %llu", count, c->hdr->doubles_per_rank);
    SML_HIP(hipMemcpyAsync(c->shm_data + (size_t)c->rank * count, send_dev, count * sizeof(double), hipMemcpyDeviceToHost, st));
    SML_HIP(hipStreamSynchronize(st));
    int rc = shm_barrier(c);
    if (rc) return rc;
    SML_HIP(hipMemcpyAsync(recv_dev, c->shm_data, count * c->nranks * sizeof(double), hipMemcpyHostToDevice, st));
    SML_HIP(hipStreamSynchronize(st));
    return shm_barrier(c);                      // nobody overwrites its block before everyone has read it
}

int comm_all_gather(sml_comm *c, const double *send_dev, double *recv_dev, size_t count, hipStream_t st)
{
    if (c->hdr) return shm_all_gather(c, send_dev, recv_dev, count, st);
    Rccl *r;
    int rc = load(&r);
    if (rc) return rc;
    return check(r, r->all_gather(send_dev, recv_dev, count, NCCL_FLOAT64, c->comm, st), "ncclAllGather");
}

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

/* For hosts without MPI (the image has no Fortran MPI module): all ranks of ONE node call this with the same `name`.
 * Transport from the environment: SML_COMM_TRANSPORT=rccl (default) -- rank 0 draws the RCCL id and leaves it in /dev/shm/<name>.id
 * for the others (what MPI_Bcast does in an MPI build), then sml_comm_create; =shm -- the host-staged rehearsal transport above
 * (several ranks on one GPU).  max_doubles_per_rank bounds one rank's contribution to a collective (shm only). */
int sml_comm_bootstrap(int nranks, int rank, const char *name, uint64_t max_doubles_per_rank, sml_comm **out)
{
    SML_REQUIRE(out && name && *name && nranks >= 1 && rank >= 0 && rank < nranks, "sml_comm_bootstrap: bad arguments");
    const char *tr = getenv("SML_COMM_TRANSPORT");
    const bool shm = tr && !strcmp(tr, "shm");
    const unsigned long long nonce = launch_nonce();
    const int limit = rendezvous_timeout_s();
    const auto t0 = std::chrono::steady_clock::now();
    auto timed_out = [&] { return std::chrono::steady_clock::now() - t0 > std::chrono::seconds(limit); };
    if (!shm) {
        // id file = [nonce (8 bytes) | RCCL unique id (128 bytes)], written under a temporary name and renamed into place
        const std::string path = std::string("/dev/shm/") + name + ".id";
        char rec[8 + ID_BYTES];
        if (rank == 0) {
            int rc = sml_comm_unique_id(rec + 8);
            if (rc) return rc;
            memcpy(rec, &nonce, 8);
            (void)unlink(path.c_str());                                  // a leftover of an earlier run goes first
            const std::string tmp = path + ".tmp";
            FILE *f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(rec, 1, sizeof rec, f) != sizeof rec) { if (f) fclose(f); return sml::fail(SML_ERR_STATE, "sml_comm_bootstrap: cannot write %s", tmp.c_str()); }
            fclose(f);
            publish(g_pending_files, path);
            if (rename(tmp.c_str(), path.c_str())) return sml::fail(SML_ERR_STATE, "sml_comm_bootstrap: cannot publish %s", path.c_str());
        } else {
            for (;;) {
                FILE *f = fopen(path.c_str(), "rb");
                if (f) {
                    const size_t got = fread(rec, 1, sizeof rec, f);
                    fclose(f);
                    unsigned long long theirs = 0;
                    memcpy(&theirs, rec, 8);
                    if (got == sizeof rec && theirs == nonce) break;         // (anything else: another launch's file, or one being replaced)
                }
                if (timed_out())
                    return sml::fail(SML_ERR_STATE, "sml_comm_bootstrap: rank %d waited %d s for an id file %s of this launch (nonce %llu)", rank, limit,
                                     path.c_str(), nonce);
                usleep(2000);
            }
        }
        int rc = sml_comm_create(nranks, rank, rec + 8, out);
        // ncclCommInitRank is collective: once it has returned here every peer has read the file
        if (rank == 0) withdraw(g_pending_files, path, false);
        return rc;
    }
    const std::string shm_name = std::string("/") + name;
    const size_t per = max_doubles_per_rank ? (size_t)max_doubles_per_rank : (size_t)1 << 19;
    const size_t bytes = sizeof(ShmHeader) + 64 + per * nranks * sizeof(double);
    void *mem = MAP_FAILED;
    if (rank == 0) {
        (void)shm_unlink(shm_name.c_str());
        const int fd = shm_open(shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes)) { if (fd >= 0) close(fd); return sml::fail(SML_ERR_STATE, "sml_comm_bootstrap: cannot create shared memory %s", shm_name.c_str()); }
        publish(g_pending_shm, shm_name);
        mem = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (mem == MAP_FAILED) return sml::fail(SML_ERR_STATE, "sml_comm_bootstrap: mmap of %s failed", shm_name.c_str());
        ShmHeader *h = (ShmHeader *)mem;
        h->arrive.store(0); h->generation.store(0); h->id_ready.store(0);
        h->doubles_per_rank = per;
        h->nonce = nonce;
        h->magic.store(SHM_MAGIC);
    } else {
        // attach to the segment of THIS launch: one that is too small, not initialised yet, or carries another launch's nonce (a
        // leftover rank 0 is about to replace) is let go and looked up again
        for (;;) {
            const int fd = shm_open(shm_name.c_str(), O_RDWR, 0600);
            struct stat sb;
            if (fd >= 0 && !fstat(fd, &sb) && (size_t)sb.st_size >= bytes) {
                mem = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                if (mem != MAP_FAILED) {
                    ShmHeader *h = (ShmHeader *)mem;
                    if (h->magic.load() == SHM_MAGIC && h->nonce == nonce && h->doubles_per_rank == per) { close(fd); break; }
                    munmap(mem, bytes);
                    mem = MAP_FAILED;
                }
            }
            if (fd >= 0) close(fd);
            if (timed_out())
                return sml::fail(SML_ERR_STATE, "sml_comm_bootstrap: rank %d waited %d s for a shared-memory segment %s of this launch (nonce %llu)", rank, limit,
                                 shm_name.c_str(), nonce);
            usleep(2000);
        }
    }
    sml_comm *c = new sml_comm;
    c->nranks = nranks; c->rank = rank; c->hdr = (ShmHeader *)mem; c->shm_bytes = bytes; c->shm_name = shm_name;
    c->shm_data = (double *)((char *)mem + ((sizeof(ShmHeader) + 63) & ~(size_t)63));
    int rc = shm_barrier(c);
    // everybody is attached: the name is not needed any more (the mappings stay), and a later run cannot find this segment
    if (rank == 0) withdraw(g_pending_shm, shm_name, true);
    if (rc) { munmap(mem, bytes); delete c; return rc; }
    *out = c;
    return SML_OK;
}

int sml_comm_destroy(sml_comm *c)
{
    if (!c) return SML_OK;
    if (c->hdr) {
        (void)shm_barrier(c);                   // (nobody is still reading; the name went right after the bootstrap)
        munmap((void *)c->hdr, c->shm_bytes);
        c->hdr = nullptr;
    }
    Rccl *r;
    if (c->comm && load(&r) == SML_OK) (void)r->destroy(c->comm);
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
    int rc;
    hipStream_t st = sml::as_stream(stream);
    const int per = number_of_regions / c->nranks, left = number_of_regions % c->nranks;
    const int mine = per + ((c->rank >= 1 && c->rank <= left) ? 1 : 0);
    SML_REQUIRE(bank->capacity >= mine, "sml_comm_allgather_outvec: the bank holds %d slots, rank %d of %d owns %d of %d regions", bank->capacity, c->rank,
                c->nranks, mine, number_of_regions);
    const size_t width = (size_t)bank->max_n_out;
    if (left == 0 && bank->capacity == per)       // equal blocks: the gathered slab IS the region-ordered slab
        return comm_all_gather(c, bank->d_outvec, all_outvec_dev, (size_t)per * width, st);
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
    if ((rc = comm_all_gather(c, c->send, c->stage, send_count, st))) return rc;
    return sml_comm_unpack_regions(c->stage, c->nranks, slots, number_of_regions, bank->max_n_out, all_outvec_dev, stream);
}

}  // extern "C"

namespace sml {
// the minimum over the ranks of one small integer (a collective: every rank of the communicator calls it) -- how the ranks of a hybrid
// engine agree on the storage mode of their banks
int comm_agree_min(sml_comm *c, int mine, int *agreed)
{
    SML_REQUIRE(c && agreed, "comm_agree_min: bad arguments");
    double *buf = nullptr;
    SML_HIP(hipMalloc((void **)&buf, sizeof(double) * (size_t)(c->nranks + 1)));
    const double v = (double)mine;
    int rc = SML_OK;
    if (hipMemcpy(buf, &v, sizeof v, hipMemcpyHostToDevice) != hipSuccess) rc = sml::fail(SML_ERR_HIP, "comm_agree_min: upload failed");
    if (!rc) rc = comm_all_gather(c, buf, buf + 1, 1, nullptr);
    std::vector<double> all((size_t)c->nranks, 0.0);
    if (!rc && (hipDeviceSynchronize() != hipSuccess || hipMemcpy(all.data(), buf + 1, sizeof(double) * all.size(), hipMemcpyDeviceToHost) != hipSuccess))
        rc = sml::fail(SML_ERR_HIP, "comm_agree_min: download failed");
    (void)hipFree(buf);
    if (rc) return rc;
    int m = mine;
    for (double a : all) m = std::min(m, (int)a);
    *agreed = m;
    return SML_OK;
}
}  // namespace sml
