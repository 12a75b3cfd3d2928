// mock_rccl.cpp -- an in-process stand-in for the eight RCCL entry points the sharded engine uses (csrc/engine.cpp: RcclApi),
// for tests on ONE GPU: the "ranks" are threads of one process, a grouped ncclSend / ncclRecv becomes event-ordered
// device-to-device copies between the ranks' buffers.  Loaded instead of librccl.so.1 through EKF_RCCL_LIBRARY.  It lets the suite
// drive the engine's IN-STREAM exchange (exchange_rows' RCCL branch: grouped send / recv between every pair of ranks, on the
// engine's stream, no host callback, no stream drain) with more than one rank, which a one-GPU box cannot do with the real library --
// and it checks what the real library would: every send has a matching receive of the same size.  Test infrastructure only.
//   hipcc -shared -fPIC tests/cpp/mock_rccl.cpp -o <dir>/libmock_rccl.so
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclChar = 0 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct MockComm *ncclComm_t;
}

namespace {
struct Post { // one send, posted by its sender for its receiver
    const void *src = nullptr;
    size_t bytes = 0;
    hipEvent_t ready = nullptr; // the sender's stream has produced the data
    hipEvent_t done = nullptr;  // the receiver's copy has finished
    bool live = false;
};
struct World {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long gen = 0;
    std::vector<Post> box; // [src * world + dst]
    // statistics (per rank)
    std::vector<long long> sent, recvd, n_send, n_recv, n_groups;
    int mismatches = 0;
    void barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        const long g = gen;
        if (++arrived == world) {
            arrived = 0;
            ++gen;
            cv.notify_all();
        } else cv.wait(lk, [&] { return gen != g; });
    }
};
struct Op { bool send; void *buf; size_t bytes; int peer; hipStream_t stream; };
std::mutex g_mu;
std::map<long long, World *> g_worlds;
long long g_next_id = 1;
thread_local std::vector<Op> t_ops;
thread_local int t_depth = 0;
thread_local ncclComm_t t_comm = nullptr;
} // namespace

struct MockComm { World *w; int rank; };

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::lock_guard<std::mutex> lk(g_mu);
    std::memset(id, 0, sizeof(*id));
    const long long k = g_next_id++;
    std::memcpy(id->internal, &k, sizeof(k));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int world, ncclUniqueId id, int rank)
{
    long long k;
    std::memcpy(&k, id.internal, sizeof(k));
    std::lock_guard<std::mutex> lk(g_mu);
    World *&w = g_worlds[k];
    if (!w) {
        w = new World();
        w->world = world;
        w->box.resize((size_t)world * world);
        w->sent.assign(world, 0); w->recvd.assign(world, 0); w->n_send.assign(world, 0); w->n_recv.assign(world, 0); w->n_groups.assign(world, 0);
    }
    if (w->world != world || rank < 0 || rank >= world) return ncclInvalidArgument;
    *comm = new MockComm{w, rank};
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    delete comm; // (the world and its statistics stay for mock_rccl_stats)
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    if (t_depth++ == 0) { t_ops.clear(); t_comm = nullptr; }
    return ncclSuccess;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (t_depth == 0 || (t_comm && t_comm != comm)) return ncclInvalidArgument; // (the engine always groups; one communicator per group)
    t_comm = comm;
    t_ops.push_back(Op{true, const_cast<void *>(buf), count, peer, stream});
    return ncclSuccess;
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (t_depth == 0 || (t_comm && t_comm != comm)) return ncclInvalidArgument;
    t_comm = comm;
    t_ops.push_back(Op{false, buf, count, peer, stream});
    return ncclSuccess;
}

// The exchange itself: (1) every sender records "data ready" on its stream and posts the send; (2) every receiver makes its stream
// wait for that event and enqueues the copy, then records "copied"; (3) every sender makes its stream wait for "copied" -- its
// later kernels may overwrite the source.  Three rendezvous of the ranks' threads; nothing waits on the host for the GPU.
ncclResult_t ncclGroupEnd()
{
    if (--t_depth > 0) return ncclSuccess;
    if (!t_comm) return ncclSuccess; // an empty group
    World *w = t_comm->w;
    const int me = t_comm->rank, W = w->world;
    ncclResult_t rc = ncclSuccess;
    for (const Op &o : t_ops)
        if (o.send) {
            Post &p = w->box[(size_t)me * W + o.peer];
            p.src = o.buf; p.bytes = o.bytes; p.live = true;
            if (!p.ready && hipEventCreateWithFlags(&p.ready, hipEventDisableTiming) != hipSuccess) rc = ncclUnhandledCudaError;
            if (!p.done && hipEventCreateWithFlags(&p.done, hipEventDisableTiming) != hipSuccess) rc = ncclUnhandledCudaError;
            if (p.ready && hipEventRecord(p.ready, o.stream) != hipSuccess) rc = ncclUnhandledCudaError;
            w->sent[me] += (long long)o.bytes;
            ++w->n_send[me];
        }
    w->barrier();
    for (const Op &o : t_ops)
        if (!o.send) {
            Post &p = w->box[(size_t)o.peer * W + me];
            if (!p.live || p.bytes != o.bytes) { // a receive without its send, or of another size: what hangs or corrupts with the real library
                std::lock_guard<std::mutex> lk(w->mu);
                ++w->mismatches;
                rc = ncclInvalidArgument;
                continue;
            }
            if (hipStreamWaitEvent(o.stream, p.ready, 0) != hipSuccess) rc = ncclUnhandledCudaError;
            if (hipMemcpyAsync(o.buf, p.src, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess) rc = ncclUnhandledCudaError;
            if (hipEventRecord(p.done, o.stream) != hipSuccess) rc = ncclUnhandledCudaError;
            w->recvd[me] += (long long)o.bytes;
            ++w->n_recv[me];
        }
    w->barrier();
    for (const Op &o : t_ops)
        if (o.send) {
            Post &p = w->box[(size_t)me * W + o.peer];
            if (hipStreamWaitEvent(o.stream, p.done, 0) != hipSuccess) rc = ncclUnhandledCudaError;
        }
    w->barrier(); // (everyone has consumed the posts of this group)
    for (const Op &o : t_ops)
        if (o.send) w->box[(size_t)me * W + o.peer].live = false;
    ++w->n_groups[me];
    t_ops.clear();
    t_comm = nullptr;
    return rc;
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock RCCL error"; }

// statistics of the most recently created world: bytes sent / received, sends, receives and groups of `rank`; returns the number of
// mismatched receives seen so far (0 in a correct run), -1 without a world
int mock_rccl_stats(int rank, long long *sent, long long *recvd, long long *n_send, long long *n_recv, long long *n_groups)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_worlds.empty()) return -1;
    World *w = g_worlds.rbegin()->second;
    if (rank < 0 || rank >= w->world) return -1;
    if (sent) *sent = w->sent[rank];
    if (recvd) *recvd = w->recvd[rank];
    if (n_send) *n_send = w->n_send[rank];
    if (n_recv) *n_recv = w->n_recv[rank];
    if (n_groups) *n_groups = w->n_groups[rank];
    return w->mismatches;
}
}
