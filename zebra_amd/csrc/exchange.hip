// One-node multi-GPU exchange of the memory rows a batch rewrote (the reference is one process on one
// device: train.py:145-146), INSIDE the native step loop: pack -> all-gather -> scatter -> projected-row refresh are
// enqueued from C++ on the pipeline's main stream, behind the GRU update of the batch and in front of the next batch's
// aggregation, which reads the exchanged rows.  No Python, no torch.distributed between two steps.
//
// Transport 1, ZT_XCHG_RCCL (the one a real node runs): RCCL's ncclAllGather over xGMI, one rank per GPU.  The
//   communicator is made here from a unique id the caller distributes (zt_exchange_unique_id on rank 0, any side channel --
//   torch.distributed's store in zebra_amd/distributed.py).  RCCL is looked up in the process at run time (dlsym: the copy
//   torch loaded, else librccl.so.1): the library itself does not link against it, so it loads where there is no RCCL.
// Transport 2, ZT_XCHG_SHM (tests / rehearsal only): ranks that SHARE one GPU cannot form an RCCL communicator (it
//   refuses two ranks on one device), so the payload goes through a POSIX shared-memory segment with two host
//   synchronisations per step.  Same pack / scatter kernels, same fixed-size payload: what it proves is that the
//   sharded native loop equals the single-GPU run; it says nothing about speed.
//
// Payload per rank and batch: cap = ceil(2 max_B / world) rows of [id | memory row (D) | last_update] floats, and -- only
// when the caller asks for them -- [message row | message time]: an eval step consumes a batch's messages inside the step
// that stored them (model/tgn_model.py:159-172), so no later step reads another rank's messages; the tests that compare
// that table switch them on.
#include "common.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>

#include <rccl/rccl.h>          // types only: every call goes through a pointer found with dlsym

using namespace zt;

namespace {

struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    const char *(*GetErrorString)(ncclResult_t);
    bool ok;
};

const RcclApi *rccl()
{
    static RcclApi api = [] {
        RcclApi a;
        memset(&a, 0, sizeof(a));
        void *h = RTLD_DEFAULT;                       // the RCCL torch brought along, if the process has one
        if (dlsym(h, "ncclAllGather") == nullptr) {
            h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (h == nullptr) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (h == nullptr) return a;
        }
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather && a.GetErrorString;
        return a;
    }();
    return api.ok ? &api : nullptr;
}

#define ZT_NCCL(expr)                                                                                   \
    do {                                                                                                \
        ncclResult_t r__ = (expr);                                                                      \
        if (r__ != ncclSuccess) {                                                                       \
            set_error("%s failed: %s (%s:%d)", #expr, rccl()->GetErrorString(r__), __FILE__, __LINE__); \
            return ZT_ERR_HIP;                                                                          \
        }                                                                                               \
    } while (0)

// the shared-memory segment of transport 2: two halves (steps alternate), a pair of counters per rank
struct ShmHeader {
    std::atomic<long long> written[64];      // step whose rows rank r has put into its slot
    std::atomic<long long> read_done[64];    // step whose rows rank r has copied out
};

}  // namespace

struct zt_exchange {
    int rank, world, kind, with_messages;
    zt_row_tables tables;
    int64_t cap;               // rows per rank and step
    int row_floats;
    float *send, *recv;        // [cap][row_floats], [world * cap][row_floats]
    int32_t *ids;              // [world * cap] ids of the received rows (-1: padding): what the projected table refreshes
    ncclComm_t comm;
    // transport 2
    char shm_name[80];
    void *shm;
    size_t shm_bytes, half_bytes;
    long long step;
};

namespace {

// buffers, communicator, mapping and the handle itself (zt_exchange_destroy; every failing path of zt_exchange_create)
void release_exchange(zt_exchange *x);

// [id | row of every table] of the rows a rank's GRU update rewrote (k_pack_rows of memory_update.hip, with the ids in a
// compact array beside the payload on the receiving side)
__global__ __launch_bounds__(256) void k_xchg_pack(zt_row_tables T, int row_floats, const int *__restrict__ ids,
                                                   const int *__restrict__ n_valid, long long cap, float *__restrict__ out)
{
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= cap) return;
    const int id = r < (long long)*n_valid ? ids[r] : -1;
    float *o = out + r * row_floats;
    if (lane == 0) o[0] = __int_as_float(id);
    int col = 1;
    for (int t = 0; t < T.n; ++t) {
        const int w = T.width[t];
        const float *src = T.ptr[t] + (size_t)(id < 0 ? 0 : id) * w;
        for (int c = lane; c < w; c += 64) o[col + c] = id < 0 ? 0.f : src[c];
        col += w;
    }
}

// every received row with id >= 0 overwrites the local tables; ids_out[r] = id (or -1)
__global__ __launch_bounds__(256) void k_xchg_scatter(zt_row_tables T, int row_floats, const float *__restrict__ recv, long long rows,
                                                      int *__restrict__ ids_out)
{
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float *in = recv + r * row_floats;
    const int id = __float_as_int(in[0]);
    if (lane == 0) ids_out[r] = id;
    if (id < 0) return;
    int col = 1;
    for (int t = 0; t < T.n; ++t) {
        const int w = T.width[t];
        float *dst = T.ptr[t] + (size_t)id * w;
        for (int c = lane; c < w; c += 64) dst[c] = in[col + c];
        col += w;
    }
}

bool spin_until(const std::atomic<long long> *c, int world, long long want)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        bool all = true;
        for (int r = 0; r < world; ++r)
            if (c[r].load(std::memory_order_acquire) < want) { all = false; break; }
        if (all) return true;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;
        std::this_thread::yield();
    }
}

}  // namespace

extern "C" int zt_exchange_unique_id(void *id_out, int64_t bytes)
{
    if (!id_out || bytes != NCCL_UNIQUE_ID_BYTES) { set_error("zt_exchange_unique_id: the id is %d bytes", NCCL_UNIQUE_ID_BYTES); return ZT_ERR_ARG; }
    if (rccl() == nullptr) { set_error("zt_exchange_unique_id: no RCCL in this process (librccl.so.1 not found)"); return ZT_ERR_UNSUPPORTED; }
    ncclUniqueId id;
    ZT_NCCL(rccl()->GetUniqueId(&id));
    memcpy(id_out, &id, NCCL_UNIQUE_ID_BYTES);
    return ZT_OK;
}

extern "C" int zt_exchange_create(zt_exchange **out, const zt_exchange_desc *d)
{
    if (!out || !d || d->world < 1 || d->world > 64 || d->rank < 0 || d->rank >= d->world || d->cap_rows <= 0 || !d->memory ||
        !d->last_update || d->D <= 0 || (d->with_messages && (!d->messages || !d->msg_ts || d->msg_dim <= 0)) ||
        (d->transport != ZT_XCHG_RCCL && d->transport != ZT_XCHG_SHM) || (d->transport == ZT_XCHG_RCCL && !d->unique_id) ||
        (d->transport == ZT_XCHG_SHM && (!d->shm_name || strlen(d->shm_name) == 0 || strlen(d->shm_name) > 60))) {
        set_error("zt_exchange_create: bad argument");
        return ZT_ERR_ARG;
    }
    zt_exchange *x = new zt_exchange();
    memset(x, 0, sizeof(*x));
    // every way out of this function but the last releases what has been made so far (round-5 advisor: the error returns
    // behind the hipMallocs leaked the buffers and the handle)
    struct Guard {
        zt_exchange *x;
        ~Guard() { if (x) release_exchange(x); }
    } guard{x};
    x->rank = d->rank; x->world = d->world; x->kind = d->transport; x->with_messages = d->with_messages ? 1 : 0;
    x->cap = d->cap_rows;
    x->tables.n = 0;
    auto add = [&](float *p, int w) { x->tables.ptr[x->tables.n] = p; x->tables.width[x->tables.n] = w; x->tables.n++; };
    add(d->memory, d->D);
    add(d->last_update, 1);
    if (x->with_messages) { add(d->messages, d->msg_dim); add(d->msg_ts, 1); }
    x->row_floats = 1;
    for (int q = 0; q < x->tables.n; ++q) x->row_floats += x->tables.width[q];
    const size_t rowb = (size_t)x->row_floats * 4;
    ZT_HIP(hipMalloc(&x->send, (size_t)x->cap * rowb));
    ZT_HIP(hipMalloc(&x->recv, (size_t)x->world * x->cap * rowb));
    ZT_HIP(hipMalloc(&x->ids, (size_t)x->world * x->cap * 4));
    if (x->kind == ZT_XCHG_RCCL) {
        if (rccl() == nullptr) { set_error("zt_exchange_create: no RCCL in this process (librccl.so.1 not found)"); return ZT_ERR_UNSUPPORTED; }
        ncclUniqueId id;
        memcpy(&id, d->unique_id, NCCL_UNIQUE_ID_BYTES);
        ZT_NCCL(rccl()->CommInitRank(&x->comm, x->world, id, x->rank));         // (collective: every rank calls it)
    } else {
        // Rank 0 MAKES the segment -- an old one of that name (a killed run: only rank 0's destroy unlinks) is removed first,
        // the new one is created exclusively and starts as zeros (ftruncate of a fresh object), so no counter of an earlier
        // run can be mistaken for this run's steps -- and the other ranks only ever OPEN it: the caller orders them behind
        // rank 0's create (TGN.enable_exchange: a barrier); a rank that comes early waits for the name to appear.
        snprintf(x->shm_name, sizeof(x->shm_name), "/%s", d->shm_name[0] == '/' ? d->shm_name + 1 : d->shm_name);
        x->half_bytes = (size_t)x->world * x->cap * rowb;
        x->shm_bytes = sizeof(ShmHeader) + 2 * x->half_bytes;
        int fd = -1;
        if (x->rank == 0) {
            (void)shm_unlink(x->shm_name);
            fd = shm_open(x->shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd >= 0 && ftruncate(fd, (off_t)x->shm_bytes) != 0) { close(fd); fd = -1; (void)shm_unlink(x->shm_name); }
        } else {
            const auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                fd = shm_open(x->shm_name, O_RDWR, 0600);
                if (fd >= 0) {
                    struct stat sb;
                    if (fstat(fd, &sb) == 0 && (size_t)sb.st_size >= x->shm_bytes) break;      // (rank 0 has sized it)
                    close(fd);
                    fd = -1;
                } else if (errno != ENOENT) break;
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) { errno = ETIMEDOUT; break; }
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
        }
        if (fd < 0) {
            set_error("zt_exchange_create: shared-memory segment %s: %s", x->shm_name, strerror(errno));
            return ZT_ERR_ARG;
        }
        x->shm = mmap(nullptr, x->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (x->shm == MAP_FAILED) { set_error("zt_exchange_create: mmap failed"); x->shm = nullptr; return ZT_ERR_ARG; }
    }
    guard.x = nullptr;
    *out = x;
    return ZT_OK;
}

// new table pointers (Memory tensors replaced: restore_memory, __init_memory__), same widths; the communicator stays
extern "C" int zt_exchange_set_tables(zt_exchange *x, float *memory, float *last_update, float *messages, float *msg_ts)
{
    if (!x || !memory || !last_update || (x->with_messages && (!messages || !msg_ts))) { set_error("zt_exchange_set_tables: bad argument"); return ZT_ERR_ARG; }
    x->tables.ptr[0] = memory; x->tables.ptr[1] = last_update;
    if (x->with_messages) { x->tables.ptr[2] = messages; x->tables.ptr[3] = msg_ts; }
    return ZT_OK;
}

namespace {
void release_exchange(zt_exchange *x)
{
    if (x->comm != nullptr && rccl() != nullptr) (void)rccl()->CommDestroy(x->comm);
    if (x->shm != nullptr) {
        (void)munmap(x->shm, x->shm_bytes);
        if (x->rank == 0) (void)shm_unlink(x->shm_name);
    }
    if (x->send) (void)hipFree(x->send);
    if (x->recv) (void)hipFree(x->recv);
    if (x->ids) (void)hipFree(x->ids);
    delete x;
}
}  // namespace

extern "C" int zt_exchange_destroy(zt_exchange *x)
{
    if (!x) return ZT_OK;
    (void)hipDeviceSynchronize();
    release_exchange(x);
    return ZT_OK;
}

// The exchange of one step on stream s: rows_dev[0 .. *count_dev) are the ids this rank's GRU update rewrote (at most cap
// of them: a rank owns at most cap batch positions).  On return (stream order) the tables hold every rank's rows and
// *ids_out / *n_ids_out name them (ids < 0: padding) for the caller's projected-row refresh.
int zt::exchange_step(zt_exchange *x, const int32_t *rows_dev, const int32_t *count_dev, void *stream, const int32_t **ids_out,
                      int64_t *n_ids_out)
{
    if (!x || !rows_dev || !count_dev) { set_error("zt_exchange_step: bad argument"); return ZT_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    const size_t n_send = (size_t)x->cap * x->row_floats;
    ZT_PROF_BEGIN(s, P_EXCHANGE);
    k_xchg_pack<<<(unsigned)((x->cap + 3) / 4), 256, 0, s>>>(x->tables, x->row_floats, rows_dev, count_dev, x->cap, x->send);
    if (x->kind == ZT_XCHG_RCCL) {
        ZT_NCCL(rccl()->AllGather(x->send, x->recv, n_send, ncclFloat32, x->comm, s));
    } else {
        ShmHeader *h = reinterpret_cast<ShmHeader *>(x->shm);
        const long long st = ++x->step;
        char *half = reinterpret_cast<char *>(x->shm) + sizeof(ShmHeader) + (size_t)(st & 1) * x->half_bytes;
        // the half was last used by step st - 2: every rank must have copied that out
        if (!spin_until(h->read_done, x->world, st - 2)) { set_error("zt_exchange_step: a rank did not finish step %lld", st - 2); return ZT_ERR_TIMEOUT; }
        ZT_HIP(hipMemcpyAsync(half + (size_t)x->rank * n_send * 4, x->send, n_send * 4, hipMemcpyDeviceToHost, s));
        ZT_HIP(hipStreamSynchronize(s));
        h->written[x->rank].store(st, std::memory_order_release);
        if (!spin_until(h->written, x->world, st)) { set_error("zt_exchange_step: a rank did not reach step %lld", st); return ZT_ERR_TIMEOUT; }
        ZT_HIP(hipMemcpyAsync(x->recv, half, x->half_bytes, hipMemcpyHostToDevice, s));
        ZT_HIP(hipStreamSynchronize(s));
        h->read_done[x->rank].store(st, std::memory_order_release);
    }
    const long long rows = (long long)x->world * x->cap;
    k_xchg_scatter<<<(unsigned)((rows + 3) / 4), 256, 0, s>>>(x->tables, x->row_floats, x->recv, rows, x->ids);
    ZT_PROF_END(s, P_EXCHANGE);
    ZT_LAUNCH_CHECK();
    if (ids_out) *ids_out = x->ids;
    if (n_ids_out) *n_ids_out = rows;
    return ZT_OK;
}

void zt::exchange_shape(const zt_exchange *x, int *rank, int *world) { *rank = x->rank; *world = x->world; }
