// fake_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl with the entry points libfluctus_hip.so binds (api.hip: rccl_load),
// selected with FLX_RCCL_LIB=<this library>.  RCCL refuses communicators with duplicate devices, so on the 1-GPU test box the
// ncclSend / ncclRecv branches of flx_gather / flx_gather_local could never run with more than one rank.  Here the "ranks" are host
// THREADS of one process (or several comms driven by one thread, ncclCommInitAll style) whose contexts may all sit on device 0:
//   * ncclCommInitRank is a rendezvous: it returns when all nranks ranks carrying the same unique id have arrived;
//   * ncclSend / ncclRecv are queued until ncclGroupEnd (outside a group they are posted at once), then posted to a process-wide
//     mailbox keyed by (communicator id, source rank, destination rank).  The RECEIVER moves the bytes: it waits for the matching
//     send (bounded: FAKE_RCCL_TIMEOUT_MS, default 20 s -> ncclSystemError instead of a hang), makes its stream wait for an event
//     the sender recorded behind everything previously enqueued on the sender's stream, and enqueues hipMemcpyAsync(device to
//     device) on its own stream; the sender's stream then waits for the copy, so the send buffer may be reused in stream order --
//     the ordering contract of the real library, minus the xGMI transport.
// FAKE_RCCL_BREAK=1 corrupts the first float of every received message (a transport that delivers wrong bytes); =2 drops every
// send (the receiver times out); =3 makes ncclRecv fail inside the group (the error path of api.hip's always-closed groups); =4 makes ncclCommInitRank fail.
// It contains no product code and the product never loads it unless FLX_RCCL_LIB says so.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace {

struct SendPost { const void *ptr; size_t bytes; hipEvent_t ready; hipEvent_t *done; bool *taken; };
struct Op { bool send; void *ptr; size_t bytes; int peer; struct ncclComm *comm; hipStream_t stream; };

std::mutex g_mu;
std::condition_variable g_cv;
std::map<std::tuple<uint64_t, int, int>, std::deque<SendPost>> g_mail;      // (comm id, src, dst) -> posted sends
std::map<uint64_t, int> g_arrived;                                          // rendezvous of ncclCommInitRank
std::atomic<uint64_t> g_nextId{1};
std::atomic<uint64_t> g_calls[8];                                           // 0 send 1 recv 2 groupStart 3 groupEnd 4 initRank 5 initAll 6 abort
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

int envInt(const char *n, int d) { const char *v = getenv(n); return v && *v ? atoi(v) : d; }
int brk() { return envInt("FAKE_RCCL_BREAK", 0); }
std::chrono::milliseconds timeout() { return std::chrono::milliseconds(envInt("FAKE_RCCL_TIMEOUT_MS", 20000)); }

} // namespace

struct ncclComm { uint64_t id; int rank, nranks, device; bool aborted; };

namespace {

ncclResult_t postRecv(const Op &o)
{
    if (brk() == 3) return ncclInternalError;
    SendPost sp;
    {
        std::unique_lock<std::mutex> lk(g_mu);
        auto key = std::make_tuple(o.comm->id, o.peer, o.comm->rank);
        if (!g_cv.wait_for(lk, timeout(), [&] { auto it = g_mail.find(key); return it != g_mail.end() && !it->second.empty(); })) {
            fprintf(stderr, "[fake_rccl] recv %d <- %d: no matching send within the timeout\n", o.comm->rank, o.peer);
            return ncclSystemError;
        }
        sp = g_mail[key].front(); g_mail[key].pop_front();
    }
    ncclResult_t res = ncclSuccess;
    if (sp.bytes != o.bytes) { fprintf(stderr, "[fake_rccl] size mismatch %zu vs %zu\n", sp.bytes, o.bytes); res = ncclInvalidArgument; }
    if (hipSetDevice(o.comm->device) != hipSuccess) res = ncclUnhandledCudaError;
    if (res == ncclSuccess && hipStreamWaitEvent(o.stream, sp.ready, 0) != hipSuccess) res = ncclUnhandledCudaError;
    if (res == ncclSuccess && o.bytes && hipMemcpyAsync(o.ptr, sp.ptr, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess) res = ncclUnhandledCudaError;
    if (res == ncclSuccess && brk() == 1 && o.bytes >= 4) { const float bad = -12345.0f; (void)hipMemcpyAsync(o.ptr, &bad, 4, hipMemcpyHostToDevice, o.stream); }
    (void)hipEventRecord(*sp.done, o.stream);
    { std::lock_guard<std::mutex> lk(g_mu); *sp.taken = true; }
    g_cv.notify_all();
    return res;
}

// One host thread may drive BOTH ends of a transfer inside one group (flx_gather_local: ncclCommInitAll style).  Post every send of the
// group without waiting for its receive, run the receives, then finish the sends.
ncclResult_t flush()
{
    std::vector<Op> ops; ops.swap(t_ops);
    ncclResult_t first = ncclSuccess;
    struct Pending { Op o; hipEvent_t *done; bool *taken; };
    std::vector<Pending> sends;
    for (const Op &o : ops) {
        if (!o.send) continue;
        if (brk() == 2) continue;
        hipEvent_t ready, *done = new hipEvent_t; bool *taken = new bool(false);
        if (hipSetDevice(o.comm->device) != hipSuccess || hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(done, hipEventDisableTiming) != hipSuccess || hipEventRecord(ready, o.stream) != hipSuccess) { if (first == ncclSuccess) first = ncclUnhandledCudaError; continue; }
        { std::lock_guard<std::mutex> lk(g_mu); g_mail[std::make_tuple(o.comm->id, o.comm->rank, o.peer)].push_back(SendPost{o.ptr, o.bytes, ready, done, taken}); }
        g_cv.notify_all();
        sends.push_back(Pending{o, done, taken});
    }
    for (const Op &o : ops) if (!o.send) { ncclResult_t r = postRecv(o); if (r != ncclSuccess && first == ncclSuccess) first = r; }
    for (const Pending &p : sends) {
        std::unique_lock<std::mutex> lk(g_mu);
        if (!g_cv.wait_for(lk, timeout(), [&] { return *p.taken; })) { fprintf(stderr, "[fake_rccl] send %d -> %d: no matching receive\n", p.o.comm->rank, p.o.peer); if (first == ncclSuccess) first = ncclSystemError; continue; }
        lk.unlock();
        if (hipSetDevice(p.o.comm->device) != hipSuccess || hipStreamWaitEvent(p.o.stream, *p.done, 0) != hipSuccess) { if (first == ncclSuccess) first = ncclUnhandledCudaError; }
    }
    return first;
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof(*id));
    const uint64_t v = g_nextId.fetch_add(1);
    memcpy(id->internal, "FAKERCCL", 8);
    memcpy(id->internal + 8, &v, 8);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    g_calls[4]++;
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks || memcmp(id.internal, "FAKERCCL", 8) != 0) return ncclInvalidArgument;
    if (brk() == 4) return ncclSystemError;                                  // no communicator can be formed
    uint64_t v; memcpy(&v, id.internal + 8, 8);
    int dev = 0; (void)hipGetDevice(&dev);
    {
        std::unique_lock<std::mutex> lk(g_mu);
        g_arrived[v]++;
        g_cv.notify_all();
        if (!g_cv.wait_for(lk, timeout(), [&] { return g_arrived[v] >= nranks; })) { fprintf(stderr, "[fake_rccl] ncclCommInitRank: %d of %d ranks arrived\n", g_arrived[v], nranks); return ncclSystemError; }
    }
    *comm = new ncclComm{v, rank, nranks, dev, false};
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist)
{
    g_calls[5]++;
    if (!comms || ndev < 1) return ncclInvalidArgument;
    const uint64_t v = g_nextId.fetch_add(1);
    for (int i = 0; i < ndev; i++) comms[i] = new ncclComm{v, i, ndev, devlist ? devlist[i] : i, false};
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return ncclSuccess; }
ncclResult_t ncclCommAbort(ncclComm_t comm) { g_calls[6]++; delete comm; return ncclSuccess; }
ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) { if (!comm || !count) return ncclInvalidArgument; *count = comm->nranks; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank) { if (!comm || !rank) return ncclInvalidArgument; *rank = comm->rank; return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled cuda error";
    case ncclSystemError: return "unhandled system error (fake_rccl: peer did not show up)";
    case ncclInternalError: return "internal error";
    case ncclInvalidArgument: return "invalid argument";
    default: return "error";
    }
}

ncclResult_t ncclGroupStart() { g_calls[2]++; t_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd()
{
    g_calls[3]++;
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    return flush();
}

static size_t typeBytes(ncclDataType_t t) { return t == ncclFloat32 || t == ncclInt32 || t == ncclUint32 ? 4 : t == ncclFloat64 || t == ncclInt64 || t == ncclUint64 ? 8 : t == ncclFloat16 || t == ncclBfloat16 ? 2 : 1; }

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    g_calls[0]++;
    if (!comm || peer < 0 || peer >= comm->nranks) return ncclInvalidArgument;
    t_ops.push_back(Op{true, const_cast<void *>(buf), count * typeBytes(type), peer, comm, stream});
    return t_depth > 0 ? ncclSuccess : flush();
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    g_calls[1]++;
    if (!comm || peer < 0 || peer >= comm->nranks) return ncclInvalidArgument;
    if (brk() == 3) return ncclInternalError;
    t_ops.push_back(Op{false, buf, count * typeBytes(type), peer, comm, stream});
    return t_depth > 0 ? ncclSuccess : flush();
}

// test hook: how often each entry point ran, and the calling thread's group depth (must be 0 after every flx_* call)
void fake_rccl_counters(uint64_t *out8) { for (int i = 0; i < 8; i++) out8[i] = g_calls[i].load(); }
int fake_rccl_group_depth() { return t_depth; }

} // extern "C"
