// Test infrastructure (tests/test_gpu_multi.py): a stand-in for librccl with the four entry points csrc/tgp_multi.hip binds
// (ncclCommInitAll, ncclCommDestroy, ncclAllGather, ncclGetErrorString), so that the RCCL branch of the in-library multi-GPU handle --
// W worker threads, each calling ncclAllGather on its own communicator and stream, no group call -- runs with W > 1 on a box with ONE GPU
// (TGP_MULTI_RCCL_LIB=<this library> TGP_MULTI_TRANSPORT=rccl; ranks may share a device).  It keeps the collective's contract as the
// handle uses it: every rank calls with the same count; rank q's `count` values land at recv + q * count on every rank; the call is
// ordered on the caller's stream behind the work that filled `send`.  It makes no claim about RCCL's performance or its thread rules.
// Build: hipcc -shared -fPIC -o libstub_rccl.so stub_rccl.cpp
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <vector>

namespace {
std::atomic<int> g_total_calls{0};
struct Group {
    int n = 0;
    std::mutex m;
    std::condition_variable cv;
    int waiting = 0;
    long generation = 0;
    std::vector<const void*> send;
    std::vector<size_t> count;
    std::vector<hipEvent_t> ready;
    int calls = 0, live = 0;
    void barrier() {
        std::unique_lock<std::mutex> lk(m);
        const long g = generation;
        if (++waiting == n) {
            waiting = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != g; });
        }
    }
};
struct Comm {
    Group* g;
    int rank, dev;
};
}  // namespace

extern "C" {

int ncclCommInitAll(void** comms, int n, const int* devs) {
    if (!comms || n < 1) return 4;      // ncclInvalidArgument
    Group* g = new Group();
    g->n = n;
    g->live = n;
    g->send.assign(n, nullptr);
    g->count.assign(n, 0);
    g->ready.assign(n, nullptr);
    for (int r = 0; r < n; ++r) {
        const int dev = devs ? devs[r] : r;
        if (hipSetDevice(dev) != hipSuccess || hipEventCreateWithFlags(&g->ready[r], hipEventDisableTiming) != hipSuccess) return 1;      // ncclUnhandledCudaError
        comms[r] = new Comm{g, r, dev};
    }
    return 0;
}

int ncclCommDestroy(void* comm) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return 4;
    Group* g = c->g;
    bool last;
    {
        std::lock_guard<std::mutex> lk(g->m);
        (void)hipEventDestroy(g->ready[c->rank]);
        last = --g->live == 0;
    }
    delete c;
    if (last) delete g;
    return 0;
}

// sendcount elements of `datatype` (the handle passes 8 = ncclFloat64) from every rank, in rank order, into every rank's recvbuff
int ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, void* comm, hipStream_t stream) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c || datatype != 8) return 4;
    Group* g = c->g;
    if (hipSetDevice(c->dev) != hipSuccess) return 1;
    // my contribution is complete once the stream has passed this point
    if (hipEventRecord(g->ready[c->rank], stream) != hipSuccess) return 1;
    {
        std::lock_guard<std::mutex> lk(g->m);
        g->send[c->rank] = sendbuff;
        g->count[c->rank] = sendcount;
        ++g->calls;
        ++g_total_calls;
    }
    g->barrier();      // every rank has called and recorded
    int rc = 0;
    for (int q = 0; q < g->n && rc == 0; ++q) {
        if (g->count[q] != sendcount) rc = 4;
        else if (hipStreamWaitEvent(stream, g->ready[q], 0) != hipSuccess) rc = 1;
        else if (hipMemcpyAsync(static_cast<char*>(recvbuff) + (size_t)q * sendcount * 8, g->send[q], sendcount * 8, hipMemcpyDeviceToDevice, stream) != hipSuccess) rc = 1;
    }
    // (a stub may be slow: nobody returns before every rank's copies are done, so no rank reuses a send buffer another still reads)
    if (rc == 0 && hipStreamSynchronize(stream) != hipSuccess) rc = 1;
    g->barrier();
    return rc;
}

const char* ncclGetErrorString(int code) { return code == 0 ? "no error" : (code == 4 ? "stub: invalid argument" : "stub: HIP error"); }

// (for the test: ncclAllGather calls of the process so far, over all ranks)
int stub_rccl_total_calls() { return g_total_calls.load(); }
}
