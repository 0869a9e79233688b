// In-library multi-GPU handle (SURVEY.md 8b "Threading", 8e): ONE process owns one tgp_handle, one HIP stream and one RCCL
// communicator per device (ncclCommInitAll) and runs the time-sharded protocol of the tgp_shard_* entry points itself -- a caller
// that is not a torch.distributed job (the Julia glue: one ccall) reaches all the GPUs of a node.  The reference has no counterpart:
// src/util/scan.jl:15-28 is the sequential loop this replaces.
//
//   rank r (= one host worker thread + device r)          exchange
//   tgp_shard_reduce      segment -> ONE filter element
//                                                         all-gather of W elements (<= 1.9 KB each at d = 8)
//   tgp_shard_fold        carry-in state of the segment
//   tgp_shard_logpdf      -> 4 doubles                    summed on the host (one process: no collective needed)
//  or
//   tgp_shard_smoother_forward    -> smoother element | final filtered state
//                                                         all-gather of W elements
//   tgp_shard_smoother_backward   smoothed state at the segment end, local smoother; the rank's ONE stream synchronisation
//
// LTI models of the stationary-gain engine (tgp_steady.hip) take its two-half shard calls instead (tgp_shard_steady_begin / _finish: ONE
// all-gather of a (1 + 2 d + 3 d^2)-double element per rank, segments aligned to its 512-step tiles); if any rank reports that the engine
// does not apply, the call is repeated on the protocol above and the handle remembers it for the bound model.
//
// Transports of the all-gather: RCCL (ncclAllGather on the rank's stream; librccl is opened at run time, so that a process that already
// holds an RCCL -- torch -- shares it) when the devices are distinct; peer copies ordered by HIP events ("copy") when a device is
// listed more than once (ranks sharing a GPU: the single-GPU tests), when RCCL cannot be opened, or with TGP_MULTI_TRANSPORT=copy.
// (tests/stub_rccl.cpp through TGP_MULTI_RCCL_LIB + TGP_MULTI_TRANSPORT=rccl: the RCCL branch with W > 1 on a box with one GPU.)
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tgp_hip.h"

namespace {

// ---- the four RCCL entry points the exchange needs (rccl.h: ncclResult_t = int, ncclComm_t = opaque pointer) ----------------------
using comm_t = void*;
constexpr int kNcclFloat64 = 8;      // ncclDataType_t::ncclFloat64 / ncclDouble
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool open(std::string& why) {
        // TGP_MULTI_RCCL_LIB: another library with the four entry points (tests/stub_rccl.cpp: W > 1 ranks through the RCCL branch on a
        // box with one GPU)
        if (const char* other = std::getenv("TGP_MULTI_RCCL_LIB")) lib = dlopen(other, RTLD_NOW | RTLD_LOCAL);
        else
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
                lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (lib) break;
            }
        if (!lib) {
            why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?");
            return false;
        }
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !AllGather) {
            why = "librccl lacks ncclCommInitAll / ncclCommDestroy / ncclAllGather";
            return false;
        }
        return true;
    }
};

// ---- host barrier of the W worker threads (also carries "some rank failed" across a phase boundary) ---------------------------------
struct Barrier {
    std::mutex m;
    std::condition_variable cv;
    int n = 1, waiting = 0;
    bool acc = false, result = false;
    uint64_t gen = 0;
    // every participant learns whether ANY of them arrived with `bad` (one answer per generation, the same for all)
    bool arrive(bool bad) {
        if (n == 1) return bad;
        std::unique_lock<std::mutex> lk(m);
        const uint64_t g = gen;
        acc = acc || bad;
        if (++waiting == n) {
            waiting = 0;
            result = acc;
            acc = false;
            ++gen;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return gen != g; });
        }
        return result;      // (stable until the NEXT generation completes, which needs this thread too)
    }
};

}  // namespace

struct tgp_multi {
    int W = 0;
    std::vector<int> dev;
    std::vector<tgp_handle*> h;
    std::vector<hipStream_t> st;
    bool use_rccl = false;
    Rccl rccl;
    std::vector<comm_t> comm;
    std::string transport_note;
    // model
    int64_t T = 0;
    int d = 0, p = 0;
    bool have_model = false;
    // exchange buffers, per rank (on the rank's device)
    std::vector<double*> slot[3], gath[3], stats;      // phase 0 / 1: the general engine's elements; 2: the stationary-gain engine's
    size_t slot_n[3] = {0, 0, 0};
    bool try_steady = false;
    bool try_one = false;                  // the one-launch path's time segments (tgp_segment_*): tried first
    std::vector<double*> ybuf;             // per rank, on its device: the neighbours' halo observations (+ the segment itself for host inputs)
    std::vector<size_t> ybuf_n;
    std::vector<int> served;
    std::vector<hipEvent_t> ev_slot, ev_done;
    double* host_stats = nullptr;      // pinned, [W][4]
    std::vector<double> lml;           // per-rank share of the last combined call
    // workers
    std::vector<std::thread> thr;
    std::mutex jm;
    std::condition_variable jcv, dcv;
    std::function<int(int)> job;
    uint64_t job_gen = 0;
    int job_left = 0;
    bool quit = false;
    std::vector<int> rc;
    Barrier bar;
    std::string err;

    int fail(int code, const std::string& msg) {
        err = msg;
        return code;
    }
};

namespace {

// Balanced contiguous segments. Long series: interior boundaries on multiples of 512 steps -- a segment of the stationary-gain engine
// that hands its end state to the next one must be a whole number of that engine's tiles (the general engine does not care).
constexpr int64_t kAlign = 512;
int64_t seg_lo(int64_t T, int W, int r) {
    if (r <= 0) return 0;
    if (r >= W) return T;
    const int64_t base = T / W, rem = T % W;
    const int64_t lo = r * base + (r < rem ? r : rem);      // = parallel.segment_bounds
    return base >= 8 * kAlign ? lo / kAlign * kAlign : lo;
}
void segment(int64_t T, int W, int r, int64_t& lo, int64_t& hi) {
    lo = seg_lo(T, W, r);
    hi = seg_lo(T, W, r + 1);
}

void worker(tgp_multi* m, int r) {
    (void)hipSetDevice(m->dev[r]);
    (void)tgp_bind_host_thread(m->dev[r]);      // (a rank's host thread beside its GPU: the hand-overs through pinned memory are per call)
    uint64_t seen = 0;
    for (;;) {
        std::function<int(int)> fn;
        {
            std::unique_lock<std::mutex> lk(m->jm);
            m->jcv.wait(lk, [&] { return m->quit || m->job_gen != seen; });
            if (m->quit) return;
            seen = m->job_gen;
            fn = m->job;
        }
        const int rc = fn(r);
        {
            std::lock_guard<std::mutex> lk(m->jm);
            m->rc[r] = rc;
            if (--m->job_left == 0) m->dcv.notify_all();
        }
    }
}

// run fn(rank) on every rank's worker (W == 1: inline); returns the first non-zero code, the message of that rank in m->err
int run_all(tgp_multi* m, const std::function<int(int)>& fn) {
    if (m->W == 1) {
        (void)hipSetDevice(m->dev[0]);
        m->rc[0] = fn(0);
    } else {
        std::unique_lock<std::mutex> lk(m->jm);
        m->job = fn;
        m->job_left = m->W;
        ++m->job_gen;
        m->jcv.notify_all();
        m->dcv.wait(lk, [&] { return m->job_left == 0; });
    }
    // the rank that failed FIRST in protocol order explains the call; ranks that stopped because another failed report kAborted
    for (int r = 0; r < m->W; ++r)
        if (m->rc[r] != TGP_OK && m->rc[r] != -1) {
            if (m->err.empty()) m->err = std::string("rank ") + std::to_string(r) + ": " + tgp_last_error(m->h[r]);
            return m->rc[r];
        }
    for (int r = 0; r < m->W; ++r)
        if (m->rc[r] == -1) return m->fail(TGP_EHIP, "a rank stopped without a code");
    return TGP_OK;
}

// phase boundary: every rank arrives; if any rank has failed, all stop (no rank is left waiting in a collective)
bool sync_ok(tgp_multi* m, int my_rc) { return !m->bar.arrive(my_rc != TGP_OK); }

// all-gather of the phase's slot of every rank into every rank's gathered buffer, ordered on the rank's stream
int all_gather(tgp_multi* m, int r, int phase) {
    const size_t n = m->slot_n[phase];
    if (m->use_rccl)
        return m->rccl.AllGather(m->slot[phase][r], m->gath[phase][r], n, kNcclFloat64, m->comm[r], m->st[r]) == 0 ? TGP_OK : TGP_EHIP;
    // copy transport: my slot is complete at ev_slot[r]; once every rank has recorded its event, pull the W slots.  (Every rank passes
    // the same two barriers whatever fails: an error travels through the barrier, never by leaving early.)
    bool bad = m->bar.arrive(hipEventRecord(m->ev_slot[r], m->st[r]) != hipSuccess);
    for (int q = 0; !bad && q < m->W; ++q) {
        if (q != r && hipStreamWaitEvent(m->st[r], m->ev_slot[q], 0) != hipSuccess) bad = true;
        hipError_t e = m->dev[q] == m->dev[r]
                           ? hipMemcpyAsync(m->gath[phase][r] + q * n, m->slot[phase][q], n * sizeof(double), hipMemcpyDeviceToDevice, m->st[r])
                           : hipMemcpyPeerAsync(m->gath[phase][r] + q * n, m->dev[r], m->slot[phase][q], m->dev[q], n * sizeof(double), m->st[r]);
        if (e != hipSuccess) bad = true;
    }
    // nobody overwrites its slot (next phase / next call) before every reader has taken it
    bad = m->bar.arrive(bad || hipEventRecord(m->ev_done[r], m->st[r]) != hipSuccess);
    if (bad) return TGP_EHIP;
    for (int q = 0; q < m->W; ++q)
        if (q != r && hipStreamWaitEvent(m->st[r], m->ev_done[q], 0) != hipSuccess) return TGP_EHIP;
    return TGP_OK;
}

// forward half shared by every call: pass 1, exchange, carry-in.  `stopped`: every rank has left together at a phase boundary (the
// caller returns at once); otherwise the code of the fold is the rank's own and the caller carries it to ITS next boundary.
int forward(tgp_multi* m, int r, const double* y, const uint8_t* miss, uint32_t flags, bool& stopped) {
    stopped = true;
    int rc = tgp_shard_reduce(m->h[r], y, miss, flags & TGP_IN_DEVICE, m->slot[0][r]);
    if (!sync_ok(m, rc)) return rc != TGP_OK ? rc : -1;
    rc = all_gather(m, r, 0);
    if (!sync_ok(m, rc)) return rc != TGP_OK ? rc : -1;
    stopped = false;
    return tgp_shard_fold(m->h[r], m->gath[0][r], m->W, r);
}

// One call on the stationary-gain engine's shards. TGP_OK with *all_served: done. TGP_OK without: some rank's segment is not the
// engine's (every rank knows: the gathered elements carry it) -- run the general protocol. TGP_EUNSUPPORTED: not one of its models.
int steady_call(tgp_multi* m, const double* const* y, const double* const* Rnew, uint32_t flags, bool post, double* const* mean_out,
                double* const* var_out, bool* all_served) {
    *all_served = false;
    const int rc = run_all(m, [&](int r) {
        int c = tgp_shard_steady_begin(m->h[r], y[r], flags & TGP_IN_DEVICE, r == 0, r == m->W - 1, post ? 1 : 0, m->slot[2][r]);
        if (!sync_ok(m, c)) return c != TGP_OK ? c : -1;
        c = all_gather(m, r, 2);
        if (!sync_ok(m, c)) return c != TGP_OK ? c : -1;
        return tgp_shard_steady_finish(m->h[r], m->gath[2][r], m->W, r, post ? Rnew[r] : nullptr, flags, post ? mean_out[r] : nullptr,
                                       post ? var_out[r] : nullptr, &m->lml[r], &m->served[r]);
    });
    if (rc != TGP_OK) return rc;
    bool all = true;
    for (int r = 0; r < m->W; ++r) all = all && m->served[r] != 0;
    *all_served = all;
    return TGP_OK;
}

// One call on the one-launch path's time segments (tgp_segment_*, round 4): a rank needs its neighbours' `halo` observations next to the
// boundary and nothing else -- no exchange of filter elements, no phase boundary between the ranks; the shares of the log marginal
// likelihood are added on the host.  TGP_OK with *served: done.  TGP_OK without: the plan declines the model / the segments (the same verdict
// whatever the data), or the data holds a NaN -- the caller goes on to the older protocols.
int one_launch_call(tgp_multi* m, const double* const* y, const double* const* Rnew, uint32_t flags, bool post, double* const* mean_out,
                    double* const* var_out, double* lml_out, bool* served) {
    *served = false;
    std::vector<int64_t> b((size_t)m->W + 1);
    for (int r = 0; r <= m->W; ++r) b[r] = seg_lo(m->T, m->W, r);
    int32_t ok = 0, halo = 0;
    (void)hipSetDevice(m->dev[0]);
    int rc = tgp_segment_plan(m->h[0], m->T, m->W, b.data(), &ok, &halo);
    if (rc != TGP_OK) return m->fail(rc, std::string("rank 0: ") + tgp_last_error(m->h[0]));
    if (!ok) {
        m->try_one = false;          // (a property of the bound model and the series' length)
        return TGP_OK;
    }
    const bool idev = (flags & TGP_IN_DEVICE) != 0;
    rc = run_all(m, [&](int r) {
        const int64_t lo = b[r], hi = b[r + 1], Ts = hi - lo;
        const size_t need = (size_t)2 * halo + (idev ? 0 : (size_t)Ts);
        if (m->ybuf_n[r] < need) {
            if (m->ybuf[r]) (void)hipFree(m->ybuf[r]);
            m->ybuf[r] = nullptr;
            m->ybuf_n[r] = 0;
            if (hipMalloc(reinterpret_cast<void**>(&m->ybuf[r]), need * sizeof(double)) != hipSuccess) return (int)TGP_EHIP;
            m->ybuf_n[r] = need;
        }
        double* buf = m->ybuf[r];
        const double *yseg = nullptr, *yl = nullptr, *yr = nullptr;
        hipError_t e = hipSuccess;
        const size_t hb = (size_t)halo * sizeof(double);
        if (!idev) {         // [left halo | segment | right halo] from the ranks' host arrays
            if (r > 0) e = hipMemcpyAsync(buf, y[r - 1] + (b[r] - b[r - 1] - halo), hb, hipMemcpyHostToDevice, m->st[r]);
            if (e == hipSuccess) e = hipMemcpyAsync(buf + halo, y[r], (size_t)Ts * sizeof(double), hipMemcpyHostToDevice, m->st[r]);
            if (e == hipSuccess && r + 1 < m->W) e = hipMemcpyAsync(buf + halo + Ts, y[r + 1], hb, hipMemcpyHostToDevice, m->st[r]);
            yseg = buf + halo;
            yl = r > 0 ? buf : nullptr;
            yr = r + 1 < m->W ? buf + halo + Ts : nullptr;
        } else {             // the neighbours' edges come over the fabric (or from the same device)
            auto pull = [&](double* dst, const double* src, int q) {
                return m->dev[q] == m->dev[r] ? hipMemcpyAsync(dst, src, hb, hipMemcpyDeviceToDevice, m->st[r])
                                              : hipMemcpyPeerAsync(dst, m->dev[r], src, m->dev[q], hb, m->st[r]);
            };
            if (r > 0) e = pull(buf, y[r - 1] + (b[r] - b[r - 1] - halo), r - 1);
            if (e == hipSuccess && r + 1 < m->W) e = pull(buf + halo, y[r + 1], r + 1);
            yseg = y[r];
            yl = r > 0 ? buf : nullptr;
            yr = r + 1 < m->W ? buf + halo : nullptr;
        }
        if (e != hipSuccess) return (int)TGP_EHIP;
        return tgp_segment_logpdf_and_posterior_marginals(m->h[r], m->T, lo, hi, yseg, yl, yr, post ? Rnew[r] : nullptr, flags, post ? mean_out[r] : nullptr,
                                                          post ? var_out[r] : nullptr, &m->lml[r]);
    });
    if (rc != TGP_OK) return rc;
    double sum = 0.0;
    for (int r = 0; r < m->W; ++r) sum += m->lml[r];
    if (!(sum == sum) || sum - sum != 0.0) return TGP_OK;          // a NaN / infinite total (an observation that is not a number): the general protocol
    if (lml_out) *lml_out = sum;
    *served = true;
    return TGP_OK;
}

int check_call(tgp_multi* m, const void* y) {
    if (!m) return TGP_EINVAL;
    m->err.clear();
    if (!m->have_model) return m->fail(TGP_EINVAL, "no model set (call tgp_multi_model_set first)");
    if (!y) return m->fail(TGP_EINVAL, "y is NULL (expected an array of one pointer per rank)");
    return TGP_OK;
}

}  // namespace

extern "C" {

int tgp_create_multi(tgp_multi** out, int ndev, const int* devices) {
    if (!out) return TGP_EINVAL;
    *out = nullptr;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return TGP_EHIP;
    if (ndev == 0 && devices == nullptr) ndev = have;          // every visible device
    if (ndev < 1 || ndev > 64) return TGP_EINVAL;
    tgp_multi* m = new tgp_multi();
    m->W = ndev;
    m->dev.resize(ndev);
    for (int r = 0; r < ndev; ++r) {
        m->dev[r] = devices ? devices[r] : r;
        if (m->dev[r] < 0 || m->dev[r] >= have) {
            delete m;
            return TGP_EINVAL;
        }
    }
    m->h.assign(ndev, nullptr);
    m->st.assign(ndev, nullptr);
    m->rc.assign(ndev, 0);
    m->lml.assign(ndev, 0.0);
    m->bar.n = ndev;
    auto cleanup = [&](int code) {
        tgp_destroy_multi(m);
        return code;
    };
    for (int r = 0; r < ndev; ++r) {
        if (tgp_create(&m->h[r], m->dev[r]) != TGP_OK) return cleanup(TGP_EHIP);
        // a stream of the rank's OWN (single handles share a per-device pool since round 5: ranks that meet on one device must not -- the
        // exchange orders the ranks' streams against each other with events)
        if (hipSetDevice(m->dev[r]) != hipSuccess || hipStreamCreateWithFlags(&m->st[r], hipStreamNonBlocking) != hipSuccess) return cleanup(TGP_EHIP);
        if (tgp_set_stream(m->h[r], m->st[r]) != TGP_OK) return cleanup(TGP_EHIP);
    }
    m->ev_slot.assign(ndev, nullptr);
    m->ev_done.assign(ndev, nullptr);
    for (int r = 0; r < ndev; ++r) {
        if (hipSetDevice(m->dev[r]) != hipSuccess || hipEventCreateWithFlags(&m->ev_slot[r], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&m->ev_done[r], hipEventDisableTiming) != hipSuccess)
            return cleanup(TGP_EHIP);
    }
    if (hipHostMalloc(reinterpret_cast<void**>(&m->host_stats), (size_t)ndev * 4 * sizeof(double), hipHostMallocDefault) != hipSuccess)
        return cleanup(TGP_EHIP);
    // transport
    const char* want = std::getenv("TGP_MULTI_TRANSPORT");
    const bool distinct = std::set<int>(m->dev.begin(), m->dev.end()).size() == (size_t)ndev;
    const bool force_rccl = want != nullptr && std::strcmp(want, "rccl") == 0;      // (ranks that share a device: only a stub library accepts them)
    if (want != nullptr && std::strcmp(want, "copy") == 0) m->transport_note = "copy (TGP_MULTI_TRANSPORT)";
    else if (!distinct && !force_rccl) m->transport_note = "copy (a device is listed more than once)";
    else {
        std::string why;
        if (m->rccl.open(why)) {
            m->comm.assign(ndev, nullptr);
            const int rc = m->rccl.CommInitAll(m->comm.data(), ndev, m->dev.data());
            if (rc == 0) {
                m->use_rccl = true;
                m->transport_note = "rccl";
            } else {
                m->comm.clear();
                m->transport_note = std::string("copy (ncclCommInitAll: ") + (m->rccl.GetErrorString ? m->rccl.GetErrorString(rc) : "error") + ")";
            }
        } else {
            m->transport_note = "copy (" + why + ")";
        }
    }
    if (!m->use_rccl) {      // peer copies between distinct devices need peer access
        for (int r = 0; r < ndev; ++r)
            for (int q = 0; q < ndev; ++q)
                if (m->dev[q] != m->dev[r]) {
                    (void)hipSetDevice(m->dev[r]);
                    (void)hipDeviceEnablePeerAccess(m->dev[q], 0);      // (already enabled: an error we do not care about)
                    (void)hipGetLastError();
                }
    }
    if (ndev > 1)
        for (int r = 0; r < ndev; ++r) m->thr.emplace_back(worker, m, r);
    *out = m;
    return TGP_OK;
}

int tgp_destroy_multi(tgp_multi* m) {
    if (!m) return TGP_OK;
    {
        std::lock_guard<std::mutex> lk(m->jm);
        m->quit = true;
        m->jcv.notify_all();
    }
    for (auto& t : m->thr) t.join();
    for (int r = 0; r < m->W; ++r) {
        (void)hipSetDevice(m->dev[r]);
        if (m->st[r]) (void)hipStreamSynchronize(m->st[r]);
        if (m->use_rccl && r < (int)m->comm.size() && m->comm[r]) (void)m->rccl.CommDestroy(m->comm[r]);
        for (int ph = 0; ph < 3; ++ph) {
            if (r < (int)m->slot[ph].size() && m->slot[ph][r]) (void)hipFree(m->slot[ph][r]);
            if (r < (int)m->gath[ph].size() && m->gath[ph][r]) (void)hipFree(m->gath[ph][r]);
        }
        if (r < (int)m->stats.size() && m->stats[r]) (void)hipFree(m->stats[r]);
        if (r < (int)m->ybuf.size() && m->ybuf[r]) (void)hipFree(m->ybuf[r]);
        if (r < (int)m->ev_slot.size() && m->ev_slot[r]) (void)hipEventDestroy(m->ev_slot[r]);
        if (r < (int)m->ev_done.size() && m->ev_done[r]) (void)hipEventDestroy(m->ev_done[r]);
        if (m->h[r]) (void)tgp_destroy(m->h[r]);
        if (m->st[r]) (void)hipStreamDestroy(m->st[r]);
    }
    if (m->host_stats) (void)hipHostFree(m->host_stats);
    delete m;
    return TGP_OK;
}

const char* tgp_multi_last_error(const tgp_multi* m) { return m ? m->err.c_str() : "null handle"; }
int tgp_multi_ndev(const tgp_multi* m) { return m ? m->W : 0; }
const char* tgp_multi_transport(const tgp_multi* m) { return m ? m->transport_note.c_str() : ""; }
tgp_handle* tgp_multi_handle(tgp_multi* m, int rank) { return (m && rank >= 0 && rank < m->W) ? m->h[rank] : nullptr; }

int tgp_multi_segment(int64_t T, int ndev, int rank, int64_t* t0, int64_t* t1) {
    if (T <= 0 || ndev < 1 || rank < 0 || rank >= ndev || !t0 || !t1) return TGP_EINVAL;
    segment(T, ndev, rank, *t0, *t1);
    return TGP_OK;
}

int tgp_multi_set_option(tgp_multi* m, int option, int64_t value) {
    if (!m) return TGP_EINVAL;
    for (int r = 0; r < m->W; ++r) {
        const int rc = tgp_set_option(m->h[r], option, value);
        if (rc != TGP_OK) return m->fail(rc, tgp_last_error(m->h[r]));
    }
    return TGP_OK;
}

int tgp_multi_model_set(tgp_multi* m, int64_t T, int d, int p, int ordering, uint32_t flags, const double* A, const double* a, const double* Q,
                        const double* H, const double* hh, const double* R, const double* x0m, const double* x0P) {
    if (!m) return TGP_EINVAL;
    m->err.clear();
    m->have_model = false;
    if (flags & TGP_DEVICE_PTRS) return m->fail(TGP_EINVAL, "tgp_multi_model_set takes host pointers (the blocks are sliced per segment and uploaded)");
    if (d < 1 || d > 16) return m->fail(TGP_EUNSUPPORTED, "time sharding serves the scan engine (d <= 16); the dense path runs replicas");
    if (ordering != 0) return m->fail(TGP_EUNSUPPORTED, "time sharding of a Reverse-ordered model is not implemented");
    if (T < (int64_t)m->W) return m->fail(TGP_EINVAL, "fewer time steps than ranks");
    // per-step arrays: the segment of rank r starts lo_r steps in; shared (Fill) blocks are the same everywhere
    auto at = [&](const double* base, bool shared, int64_t lo, int64_t stride) { return (base == nullptr || shared) ? base : base + lo * stride; };
    auto bind = [&](int r) {
        int64_t lo, hi;
        segment(T, m->W, r, lo, hi);
        const bool sR = (flags & TGP_SHARED_R) != 0;
        // R per step: [T][p] (diagonal) for p > 1 and p == 1 alike; dense noise is whitened by the host before it reaches the ABI
        return tgp_model_set(m->h[r], hi - lo, d, p, ordering, flags, at(A, flags & TGP_SHARED_A, lo, (int64_t)d * d), at(a, flags & TGP_SHARED_a, lo, d),
                             at(Q, flags & TGP_SHARED_Q, lo, (int64_t)d * d), at(H, flags & TGP_SHARED_H, lo, (int64_t)p * d), at(hh, flags & TGP_SHARED_h, lo, p),
                             at(R, sR, lo, p), x0m, x0P);
    };
    // rank 0 first: a first bind may run the kernel-variant self-test, whose verdict the other ranks then find in the on-disk cache
    (void)hipSetDevice(m->dev[0]);
    int rc = bind(0);
    if (rc != TGP_OK) return m->fail(rc, std::string("rank 0: ") + tgp_last_error(m->h[0]));
    if (m->W > 1) {
        rc = run_all(m, [&](int r) { return r == 0 ? TGP_OK : bind(r); });
        if (rc != TGP_OK) return rc;
    }
    // exchange buffers
    for (int ph = 0; ph < 3; ++ph) {
        m->slot[ph].resize(m->W, nullptr);
        m->gath[ph].resize(m->W, nullptr);
    }
    m->stats.resize(m->W, nullptr);
    m->served.assign(m->W, 0);
    m->try_steady = tgp_shard_steady_slot_size(d) > 0 && p == 1;      // (tgp_shard_steady_begin decides per call whether the model is one of the engine's)
    m->try_one = m->try_steady;                                       // (tgp_segment_plan decides)
    m->ybuf.resize(m->W, nullptr);
    m->ybuf_n.resize(m->W, 0);
    if (d != m->d) {
        m->slot_n[0] = (size_t)tgp_shard_slot_size(0, d);
        m->slot_n[1] = (size_t)tgp_shard_slot_size(1, d);
        m->slot_n[2] = (size_t)(tgp_shard_steady_slot_size(d) > 0 ? tgp_shard_steady_slot_size(d) : 1);
        for (int r = 0; r < m->W; ++r) {
            if (hipSetDevice(m->dev[r]) != hipSuccess) return m->fail(TGP_EHIP, "hipSetDevice");
            for (int ph = 0; ph < 3; ++ph) {
                const size_t n = m->slot_n[ph] * sizeof(double);
                if (m->slot[ph][r]) (void)hipFree(m->slot[ph][r]);
                if (m->gath[ph][r]) (void)hipFree(m->gath[ph][r]);
                m->slot[ph][r] = m->gath[ph][r] = nullptr;
                if (hipMalloc(reinterpret_cast<void**>(&m->slot[ph][r]), n) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&m->gath[ph][r]), n * m->W) != hipSuccess)
                    return m->fail(TGP_EHIP, "hipMalloc of the exchange buffers");
            }
            if (!m->stats[r] && hipMalloc(reinterpret_cast<void**>(&m->stats[r]), 4 * sizeof(double)) != hipSuccess) return m->fail(TGP_EHIP, "hipMalloc");
        }
    }
    m->T = T;
    m->d = d;
    m->p = p;
    m->have_model = true;
    return TGP_OK;
}

int tgp_multi_logpdf(tgp_multi* m, const double* const* y, const uint8_t* const* missing, uint32_t flags, double* out) {
    int rc = check_call(m, y);
    if (rc != TGP_OK) return rc;
    if (!out) return m->fail(TGP_EINVAL, "out is NULL");
    if (m->try_one && !missing) {
        bool served = false;
        rc = one_launch_call(m, y, nullptr, flags, false, nullptr, nullptr, out, &served);
        if (rc != TGP_OK || served) return rc;
        m->err.clear();
    }
    if (m->try_steady && !missing) {
        bool served = false;
        rc = steady_call(m, y, nullptr, flags, false, nullptr, nullptr, &served);
        if (rc == TGP_OK && served) {
            double sum = 0.0;
            for (int r = 0; r < m->W; ++r) sum += m->lml[r];
            *out = sum;
            return TGP_OK;
        }
        if (rc != TGP_OK && rc != TGP_EUNSUPPORTED) return rc;
        m->err.clear();
        m->try_steady = false;      // (remembered for the bound model)
    }
    rc = run_all(m, [&](int r) {
        bool stopped = false;
        int c = forward(m, r, y[r], missing ? missing[r] : nullptr, flags, stopped);
        if (c != TGP_OK) return c;      // (no boundary behind the fold in this call)
        c = tgp_shard_logpdf(m->h[r], m->stats[r]);
        if (c != TGP_OK) return c;
        // one 32-byte copy + the rank's one synchronisation
        if (hipMemcpyAsync(m->host_stats + 4 * r, m->stats[r], 4 * sizeof(double), hipMemcpyDeviceToHost, m->st[r]) != hipSuccess ||
            hipStreamSynchronize(m->st[r]) != hipSuccess)
            return (int)TGP_EHIP;
        return (int)TGP_OK;
    });
    if (rc != TGP_OK) return rc;
    double s[4] = {0, 0, 0, 0};
    for (int r = 0; r < m->W; ++r)
        for (int k = 0; k < 4; ++k) s[k] += m->host_stats[4 * r + k];
    if (s[2] != 0.0 || s[3] != 0.0) return m->fail(TGP_ENOTPD, "innovation variance / predicted covariance not positive definite (some segment)");
    *out = s[0];
    return TGP_OK;
}

static int multi_posterior(tgp_multi* m, const double* const* y, const uint8_t* const* missing, const double* const* Rnew, uint32_t flags,
                           double* const* mean_out, double* const* var_out, double* lml_out) {
    int rc = check_call(m, y);
    if (rc != TGP_OK) return rc;
    if (!Rnew || !mean_out || !var_out) return m->fail(TGP_EINVAL, "Rnew / mean_out / var_out is NULL (expected arrays of one pointer per rank)");
    if (m->try_one && !missing) {
        bool served = false;
        rc = one_launch_call(m, y, Rnew, flags, true, mean_out, var_out, lml_out, &served);
        if (rc != TGP_OK || served) return rc;
        m->err.clear();
    }
    if (m->try_steady && !missing) {
        bool served = false;
        rc = steady_call(m, y, Rnew, flags, true, mean_out, var_out, &served);
        if (rc == TGP_OK && served) {
            if (lml_out) {
                double sum = 0.0;
                for (int r = 0; r < m->W; ++r) sum += m->lml[r];
                *lml_out = sum;
            }
            return TGP_OK;
        }
        if (rc != TGP_OK && rc != TGP_EUNSUPPORTED) return rc;
        m->err.clear();
        m->try_steady = false;
    }
    rc = run_all(m, [&](int r) {
        bool stopped = false;
        int c = forward(m, r, y[r], missing ? missing[r] : nullptr, flags, stopped);
        if (stopped) return c;
        if (c == TGP_OK) c = tgp_shard_smoother_forward(m->h[r], m->slot[1][r]);
        if (!sync_ok(m, c)) return c != TGP_OK ? c : -1;
        c = all_gather(m, r, 1);
        if (!sync_ok(m, c)) return c != TGP_OK ? c : -1;
        return tgp_shard_smoother_backward(m->h[r], m->gath[1][r], m->W, r, Rnew[r], flags, mean_out[r], var_out[r], &m->lml[r]);
    });
    if (rc != TGP_OK) return rc;
    if (lml_out) {
        double s = 0.0;
        for (int r = 0; r < m->W; ++r) s += m->lml[r];
        *lml_out = s;
    }
    return TGP_OK;
}

int tgp_multi_posterior_marginals(tgp_multi* m, const double* const* y, const uint8_t* const* missing, const double* const* Rnew, uint32_t flags,
                                  double* const* mean_out, double* const* var_out) {
    return multi_posterior(m, y, missing, Rnew, flags, mean_out, var_out, nullptr);
}

int tgp_multi_logpdf_and_posterior_marginals(tgp_multi* m, const double* const* y, const uint8_t* const* missing, const double* const* Rnew,
                                             uint32_t flags, double* lml_out, double* const* mean_out, double* const* var_out) {
    if (m && !lml_out) return m->fail(TGP_EINVAL, "lml_out is NULL");
    return multi_posterior(m, y, missing, Rnew, flags, mean_out, var_out, lml_out);
}

}  // extern "C"
