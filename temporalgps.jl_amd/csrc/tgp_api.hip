// C ABI of libtgp_hip.so (see include/tgp_hip.h): handle, device buffers, launch orchestration.
// Host orchestration of one call (single stream, kernel boundaries are the only synchronisation):
//   forward:  k_reduce_filter -> k_scan_reduce^* -> k_scan_apply(top) -> k_scan_apply^* -> k_apply_filter -> k_finalize
//   smoother: (forward with MODE 2) -> k_scan_reduce^* -> k_scan_apply(top) -> k_scan_apply^* -> k_smooth
#include "../../include/tgp_hip.h"

#include <sched.h>
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <sys/stat.h>

#include <algorithm>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "tgp_kernels.hpp"
#include "tgp_dense.hpp"
#include <chrono>
#include "tgp_steady.hpp"
#include "tgp_modal.hpp"
#include "tgp_sweep.hpp"
#include "tgp_wide.hpp"
#include "tgp_adjoint_host.hpp"
#include "tgp_alloc.hpp"
#include <map>
#include <unordered_map>

#include "tgp_api_alloc.inc"      // the caching allocator (tgp_alloc.hpp)

namespace tgp {
const KernelTable *kernel_table_d1(), *kernel_table_d2(), *kernel_table_d3(), *kernel_table_d4(), *kernel_table_d5(),
    *kernel_table_d6(), *kernel_table_d7(), *kernel_table_d8(), *kernel_table_d9(), *kernel_table_d10(), *kernel_table_d11(),
    *kernel_table_d12(), *kernel_table_d13(), *kernel_table_d14(), *kernel_table_d15(), *kernel_table_d16();
const KernelTable* kernel_table(int d) {
    switch (d) {
        case 1: return kernel_table_d1();
        case 2: return kernel_table_d2();
        case 3: return kernel_table_d3();
        case 4: return kernel_table_d4();
        case 5: return kernel_table_d5();
        case 6: return kernel_table_d6();
        case 7: return kernel_table_d7();
        case 8: return kernel_table_d8();
        case 9: return kernel_table_d9();
        case 10: return kernel_table_d10();
        case 11: return kernel_table_d11();
        case 12: return kernel_table_d12();
        case 13: return kernel_table_d13();
        case 14: return kernel_table_d14();
        case 15: return kernel_table_d15();
        case 16: return kernel_table_d16();
        default: return nullptr;
    }
}
}  // namespace tgp

// fully-inlined builds of d = 5..8 (tgp_inst_d5i.hip, tgp_inst_d6i.hip; d = 7, 8 as 13 parts each, tgp_inst_part.hip):
// same struct layouts, other namespace.
namespace tgp_i {
struct KernelTable;
const KernelTable* kernel_table_d5_i();
const KernelTable* kernel_table_d6_i();
const KernelTable* kernel_table_d7_i();
const KernelTable* kernel_table_d8_i();
}  // namespace tgp_i

// the per-step passes built for closed-form SDE transitions (tgp_inst_sde.hip; d <= kSdeBuildMaxD): same struct layouts, other namespace
namespace tgp_s {
struct KernelTable;
const KernelTable* sde_kernel_table(int d);
}  // namespace tgp_s

using namespace tgp;

static const KernelTable* fast_kernel_table(int d) {
    if (d == 5) return reinterpret_cast<const KernelTable*>(tgp_i::kernel_table_d5_i());
    if (d == 6) return reinterpret_cast<const KernelTable*>(tgp_i::kernel_table_d6_i());
    if (d == 7) return reinterpret_cast<const KernelTable*>(tgp_i::kernel_table_d7_i());
    if (d == 8) return reinterpret_cast<const KernelTable*>(tgp_i::kernel_table_d8_i());
    return nullptr;
}
// Per (state dimension, LTI layout family?): which operations of the inlined (fast) build reproduced the out-of-line (safe)
// build in the run-time known-answer check. Bit kOpDecided = the check has run.
enum VariantOp { kOpM0 = 0, kOpM1, kOpM2, kOpM3, kOpAffine, kOpGrad, kOpCount, kOpGroupM1 = 25, kOpGroupM3 = 26, kOpGroupMarg = 27, kOpGroupAff = 28, kOpGroup = 29, kOpDecided = 30 };
// Keyed by DEVICE as well (the check runs on the device the model is bound to; a mixed-device process does not inherit another
// GPU's verdict) and guarded by a mutex: two host threads binding their first model race neither the check nor the table.
constexpr int kMaxVariantDevices = 16;
static unsigned g_variant[kMaxVariantDevices][17][2] = {{{0u}}};
static std::mutex g_variant_mutex;
static const unsigned kAllOps = (1u << kOpCount) - 1u;

// table = safe build with the entries whose operations passed replaced by the fast build's
static void merge_tables(const KernelTable* safe, const KernelTable* fast, unsigned ok, KernelTable& out) {
    out = *safe;
    const bool any_fwd = (ok & ((1u << kOpM0) | (1u << kOpM1) | (1u << kOpM2) | (1u << kOpM3))) != 0;
    if (any_fwd) {     // pass 1 and the filter-monoid scans are shared by every forward operation: validated by any that passed
        out.reduce_filter = fast->reduce_filter;
        if (fast->reduce_filter_tab != nullptr) {       // the table pass of the same build (compared with the safe build's by the check)
            out.filter_table_size = fast->filter_table_size;
            out.filter_table = fast->filter_table;
            out.reduce_filter_tab = fast->reduce_filter_tab;
        }
        out.scan_reduce_c[kScanFilter] = fast->scan_reduce_c[kScanFilter];
        out.scan_apply_c[kScanFilter] = fast->scan_apply_c[kScanFilter];
    }
    for (int m = 0; m < 4; ++m)
        if (ok & (1u << (kOpM0 + m))) out.apply_filter_m[m] = fast->apply_filter_m[m];
    if (ok & (1u << kOpM2)) { out.smooth = fast->smooth; out.apply_filter_m[4] = fast->apply_filter_m[4]; out.compose_smoother = fast->compose_smoother; }
    if (ok & (1u << kOpAffine)) { out.reduce_affine = fast->reduce_affine; out.apply_affine = fast->apply_affine; }
    if (ok & ((1u << kOpAffine) | (1u << kOpM2))) {
        out.scan_reduce_c[kScanAffine] = fast->scan_reduce_c[kScanAffine];
        out.scan_apply_c[kScanAffine] = fast->scan_apply_c[kScanAffine];
    }
    if (ok & (1u << kOpGrad)) {
        out.reduce_filter_ad = fast->reduce_filter_ad;
        out.apply_filter_ad = fast->apply_filter_ad;
        out.scan_reduce_c[kScanAD] = fast->scan_reduce_c[kScanAD];
        out.scan_apply_c[kScanAD] = fast->scan_apply_c[kScanAD];
    }
}
static void select_table(tgp_handle* h, int d, bool lti, int variant);

// result[0] = sum lml + nmiss * log(2 pi 1e15)/2 (missings.jl:45-53); result[1] = nmiss; result[2] = bad.
// Fixed-order summation: the result is bit-reproducible from run to run.
__global__ __launch_bounds__(256) void k_finalize(const double* __restrict__ partial, int64_t nblocks, double* __restrict__ result,
                                                  const double* __restrict__ steady_rec = nullptr) {
    __shared__ double sh[12];
    double a = 0.0, b = 0.0;
    int c = 0;
    for (int64_t i = threadIdx.x; i < nblocks; i += 256) {
        a += partial[3 * i];
        b += partial[3 * i + 1];
        c |= partial[3 * i + 2] != 0.0;
    }
    block_sum3(a, b, c, sh);
    if (threadIdx.x == 0) {
        result[0] = a + b * 0.5 * (kLog2Pi + log(kLargeVar));
        result[1] = b;
        result[2] = (double)c;
        if (steady_rec != nullptr) result[5] = steady_rec[0];     // first mean-only step of chunk 0 (TGP_OPT_STEADY policy, finish_wait)
    }
}

// time-sharded logpdf: (lml, n missing, not-PD count, Cholesky flag) as four doubles the ranks can all-reduce(sum)
__global__ void k_pack_stats(const double* __restrict__ result, double* __restrict__ stats) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    stats[0] = result[0];
    stats[1] = result[1];
    stats[2] = result[2];
    stats[3] = (double)(*reinterpret_cast<const int*>(result + 4) != 0);
}

// model re-layout (once per model / chunk size)
// Lane = chunk, so the (slow, strided) reads happen once here and every later pass reads coalesced rows.
__global__ __launch_bounds__(256) void k_tile_model(ModelView raw, int d, uint32_t mask, int nc_t, int nc_e, int L0, int64_t n0,
                                                    double* __restrict__ tile_t, double* __restrict__ tile_e) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n0) return;
    const int Lt = L0 / raw.p;
    for (int tl = 0; tl < Lt; ++tl) tile_transition(raw, d, mask, nc_t, Lt, c, tl, tile_t);
    for (int i = 0; i < L0; ++i) tile_emission(raw, d, mask, nc_e, L0, c, i, tile_e);
}

// closed-form SDE transitions (ModelView::sde): the transition record is the step's tau alone, in processing order
// (tau < 0 marks the first transition; the skipped predict of a Reverse model's first processing step gets 0)
__global__ __launch_bounds__(256) void k_tile_dt(const double* __restrict__ times, int64_t Tt, int ordering, int Lt, int64_t n0, double* __restrict__ tile_t) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n0) return;
    for (int tl = 0; tl < Lt; ++tl) {
        const int64_t tproc = c * (int64_t)Lt + tl;
        if (tproc >= Tt) break;
        const int64_t tt = ordering == 0 ? tproc : Tt - tproc;
        double tau = 0.0;
        if (!(ordering != 0 && tproc == 0)) tau = (tt == 0) ? -1.0 : times[tt] - times[tt - 1];
        tile_t[fs_index(c, tl, 0, Lt, 1)] = tau;
    }
}

// gradient pass: partial[4b + 0..3] -> result[0] lml (+ missing compensation), [1] n missing, [2] bad, [3] d lml / d theta
__global__ __launch_bounds__(256) void k_finalize_ad(const double* __restrict__ partial, int64_t nblocks, double* __restrict__ result) {
    __shared__ double sh[12];
    __shared__ double sh2[4];
    double a = 0.0, b = 0.0, dl = 0.0;
    int c = 0;
    for (int64_t i = threadIdx.x; i < nblocks; i += 256) {
        a += partial[4 * i];
        b += partial[4 * i + 1];
        c |= partial[4 * i + 2] != 0.0;
        dl += partial[4 * i + 3];
    }
    block_sum3(a, b, c, sh);
    block_sum_d(dl, sh2);
    if (threadIdx.x == 0) {
        result[0] = a + b * 0.5 * (kLog2Pi + log(kLargeVar));
        result[1] = b;
        result[2] = (double)c;
        result[3] = dl;
    }
}

namespace {

// TGP_POISON=1 (environment, debugging): every fresh device allocation is filled with 0xFF bytes (a NaN pattern for doubles, a set
// flag for masks), so that a kernel reading memory nobody wrote produces NaN deterministically instead of depending on what the
// allocator handed back. The -m gpu suite is run once with it per round.
inline bool poison_allocations() {
    static const bool on = [] { const char* e = std::getenv("TGP_POISON"); return e != nullptr && e[0] == '1'; }();
    return on;
}

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)tgp_alloc::dev_free(p);
        p = nullptr;
        cap = 0;
        hipError_t e = tgp_alloc::dev_malloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        if (e == hipSuccess && poison_allocations()) {
            e = hipMemset(p, 0xFF, bytes);
            (void)hipDeviceSynchronize();       // (the fill runs on the null stream: keep it ahead of the handle's own stream)
        }
        return e;
    }
    void release() {
        if (p) (void)tgp_alloc::dev_free(p);
        p = nullptr;
        cap = 0;
    }
    double* d() const { return static_cast<double*>(p); }
};

struct ScanCtx {
    int monoid = 0, NC = 0, NS = 0;
    std::vector<int64_t> n;
    std::vector<double*> E, S;
    double* fin = nullptr;
    DevBuf slab;
};

struct ProfEntry {
    std::string name;
    double ms = 0.0;
    int64_t calls = 0;
};
struct PendingEvt {
    int idx;
    hipEvent_t a, b;
};

constexpr int kTopBS = 512;

}  // namespace

struct tgp_handle {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::string err;
    // dense large-state engine (d > 16, tgp_dense.hip): sequential in time, fp64 MFMA
    tgp_dense::Engine* dense = nullptr;
    bool is_dense = false;
    int dense_structure = 1;     // TGP_OPT_DENSE_STRUCTURE
    int dense_fused = 1;         // TGP_OPT_DENSE_FUSED
    // model
    bool have_model = false, lti = false;
    int64_t T = 0;
    int d = 0, p = 1, ordering = 0;
    ModelView mv{};       // what the kernels see (tile pointer valid after ensure_tiled)
    ModelView raw{};      // the arrays as handed over (reference layout): source of the tiling
    DevBuf tile_t, tile_e;
    int tile_L0 = 0;
    // SDE-described transitions (tgp_model_set_sde): A_k, Q_k are built on the device from the time stamps
    bool sde = false;
    DevBuf bF, bPinf, btimes, bAQ1, bsde;
    bool have_AQ1 = false;
    bool binding_sde = false;    // inside tgp_model_set_sde's call of tgp_model_set
    bool sde_closed = false;     // the drift matrix has the closed-form exponential of ModelView::sde (bsde holds the coefficients)
    bool tile_is_dt = false;     // the transition record currently holds tau alone (value passes); false: A_k, Q_k (gradient passes, d > kSdeInKernelMaxD)
    const double* times_dev = nullptr;
    double normF = 0.0;
    const KernelTable* kt = nullptr;   // -> ktm when the inlined and out-of-line builds are mixed entry by entry
    KernelTable ktm{};
    int variant_code = 1;        // 1 out-of-line build, 2 every operation inlined, 3 mixed
    DevBuf bA, ba, bQ, bH, bh, bR;
    std::vector<double> x0m, x0P;
    DevBuf bx0, bx0r, bx0fold;
    bool fold_valid = false;
    // per-call staging
    DevBuf by, bmiss, bRnew, beps_t, beps_e, bo1, bo2, bo3;
    DevBuf bflip_y, bflip_m, bflip_P;      // a Reverse-ordered LTI model served as the Forward model on the flipped series (reverse_by_flip)
    std::vector<double> flip_host;
    // scans and scratch
    ScanCtx F, Rv, Fad;
    DevBuf btan, bx0ad, tile_tan;
    DevBuf fs, partial, result, segtmp;
    double* host_result = nullptr;  // pinned, 8 doubles: [0] lml [1] nmiss [2] filter-bad ; int flags at [4]
    int64_t opt_chunk = 0;
    int profile = 0;
    int timing = 0;              // TGP_OPT_TIMING
    int variant_opt = 0;   // TGP_OPT_VARIANT: 0 auto (run-time check), 1 safe, 2 fast
    int L0 = 0;
    int64_t n0 = 0;
    bool reduce_valid = false, smoother_valid = false;
    bool fused = false;          // the current forward elements were produced with the fused level-0 reduce
    bool group_active = false;   // ... by the group-per-chunk pass 1 (32 chunks per block, its own chunking)
    bool use_group = false;      // group-per-chunk logpdf kernels validated for this model (tgp_group.hpp)
    bool use_group_aff = false;  // ... and the group-layout scans over the smoother's affine elements
    bool use_group_sm = false;   // ... and the group-per-chunk smoother passes (tgp_group_smooth.hpp)
    bool use_group_marg = false; // ... and the group-per-chunk prior-marginals passes
    bool use_group_m1 = false, use_group_m3 = false;   // ... and the MODE 1 / MODE 3 output variants
    const double* alt_H = nullptr;   // alternative emission block of the current tgp_posterior_marginals_at call (device)
    const double* alt_h = nullptr;
    int alt_p = 0;
    int force_group_post = 0;
    DevBuf balt;
    int opt_split = 1;           // TGP_OPT_SPLIT_SMOOTHER
    int opt_table = 1;           // TGP_OPT_SHARED_PARTS: pass 1 with the chunks' shared matrix parts from a table
    int opt_steady = 1;          // TGP_OPT_STEADY (bit 0): mean-only steps of passes 2 / 3 once a chunk's covariance repeats with period 2
    // TGP_OPT_STEADY = 2 (default): the stationary-gain scan engine (tgp_steady.hip) serves tgp_logpdf / tgp_[logpdf_and_]posterior_marginals of
    // Forward LTI models with one noise variance, scalar observations and no missing data.  Whether it applies (the covariance settles
    // within the head tables, the series is longer than head + tail) is decided on the device inside every call; a call that finds it
    // does not is re-run on the general path and the bound model is remembered as such (steady2_state = -1).
    int opt_steady2 = 1;
    bool opt_sde_closed = true;
    tgp_steady::Engine* steady2 = nullptr;
    int steady2_state = 0;       // 0 untried for the bound model, 1 served the last call, -1 does not apply
    bool steady2_last = false;   // the last logpdf / posterior-marginals call was served by it
    // TGP_OPT_STEADY = 3 (default since round 4): the ONE-LAUNCH form of that engine (tgp_modal.hip) is tried first -- host plan + one kernel
    // over y + the host's sum of the workgroups' partial sums; a model / series it does not serve (tgp_plan::Info::why) goes on as with 2.
    int opt_modal = 1;
    tgp_modal::Engine* modal = nullptr;
    int modal_state = 0;         // 0 untried for the bound model, 1 served the last call, -1 does not apply
    bool has_R_over = false;     // tgp_logpdf_noise: the host plans of THIS call read R_over instead of the bound model's noise variance
    double R_over = 0.0;
    int smooth_state = 0;        // the dense-powers one-launch smoother (smooth_lti_call): 0 untried / applies, -1 does not apply
    bool modal_last = false;
    int64_t dense_last_n0 = -1;   // >= 0: the last call ran on the dense-power one-launch kernels behind a head of that many steps with gains of their own
    // TGP_OPT_SWEEP (default 1): the sweep engine (tgp_sweep.hip) serves tgp_logpdf / tgp_[logpdf_and_]posterior_marginals of Forward
    // models with scalar observations, d <= 4, shared A / a / Q / H (or closed-form SDE transitions) whose gains vary in time: a mask, a noise
    // variance or an emission offset per step, irregular spacing.  One launch; a call whose warm-ups prove too short is repeated with longer
    // ones (remembered for the bound model), a model it does not serve goes on to the general engine.
    int opt_sweep = 1;
    long long opt_stream_min_T = -1;      // TGP_OPT_STREAM_MIN_T
    tgp_sweep::Engine* sweep = nullptr;
    int sweep_state = 0;          // 0 untried for the bound model, 1 served the last call, -1 does not apply
    bool sweep_last = false;
    int sweep_W = 0, sweep_Wb = 0;     // warm-ups the bound model's last served call needed (0: estimate)
    int sweep_fC = 0, sweep_fW = 0, sweep_fWb = 0;   // TGP_OPT_SWEEP_CHUNK / _WARMUP / _WARMUP_BACK (tests; 0 automatic)
    int64_t sweep_info[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // tgp_sweep_info
    double sweep_dist[2] = {0.0, 0.0};
    double sweep_Rrep = 1.0;      // a representative noise variance of the bound model (the warm-up estimate)
    double sweep_tau = 0.0;       // SDE: median gap
    std::vector<double> sde_coef_host, sde_A1Q1_host;   // SDE: closed-form coefficients and the first transition, on the host
    DevBuf btau;                  // SDE: tau_k = t_k - t_(k-1), [T] (tau_0 unused: the first transition is explicit)
    int num_cu = 0;
    std::vector<double> hostm;   // host copy of the shared blocks of an LTI model: A | a | Q | H | hh | R (what the host plan reads)
    std::vector<double> widem;   // the same for a wide LTI model (8 < d <= 63, scalar observations): what tgp_wide's plan reads
    tgp_wide::Engine* wide = nullptr;
    int wide_state = 0;          // 0 untried for the bound model, 1 served the last call, -1 does not apply
    int wide_post_state = 0;     // ... its posterior half
    const double* wide_ht = nullptr;      // device: the emission offset per step of such a model (a mean function at the inputs), else null
    int opt_wide = 1;            // TGP_WIDE=0: such models on the dense engine's one-CU passes as before (A/B runs)
    std::vector<double> sweepm;  // the same for every model with shared A, a, Q, H and scalar observations (hh, R: the first step's where they are per step)
    void* steady2_scope = nullptr;
    bool table_pending = false;  // the kernel-variant choice (and its run-time check) of the general engine is deferred to its first use
    int shard2_first = 1, shard2_last = 1, shard2_post = 0;      // the open two-half call of a stationary-gain time shard
    bool shard2_open = false;
    double* adj_host = nullptr;  // pinned: the record + the head's observations of an adjoint call
    double* flt_host = nullptr;  // pinned: head observations, head outputs and the workgroups' partial sums of an LTI filter call
    size_t flt_cap = 0;
    double* sm_sync = nullptr;   // pinned, 32 doubles: [0..7] the head's end state, [8..15] the kernel's xi at the head's end, [16], [17] their flags (smooth_lti_call)
    long long smooth_seq = 0;
    DevBuf steady_rec;           // ... the chunks' records (ModelView::steady)
    int steady_calls = 0;        // 1: the last posterior-path forward pass (mode 2) wrote the records
    // Policy of the posterior path: the build with these steps has slightly longer full steps, and a pass takes as long as its slowest
    // wave -- the one holding chunk 0, which starts from x0 and settles last. The first call on a bound model / chunk length returns
    // that chunk's first mean-only step with its results (k_finalize -> result[5]); if it lies beyond half of the chunk (a filter
    // that takes hundreds of steps to converge), later calls run the plain build. A choice of kernel, not of results: both are
    // bit-identical.
    bool steady_known = false, steady_pays = true, steady_result_pending = false;
    int steady_known_L0 = 0;
    DevBuf ftab;                 // ... the table (k_filter_table), valid for (tab_L0, tab_nlast) of the bound model
    int tab_L0 = 0, tab_nlast = 0;
    // The table costs one lane ~150 sequential steps (1.4 ms at d = 3: more than the whole call), so it is never built on the
    // caller's critical path: the SECOND eligible call on a bound model launches k_filter_table on a side stream and still runs the
    // general pass; calls use the table once its event has completed. A model that is evaluated once (a hyper-parameter search
    // binds a new model per evaluation) never builds one.
    hipStream_t side_stream = nullptr;
    hipEvent_t tab_ev = nullptr, tab_dep = nullptr;
    int tab_state = 0;           // 0 none, 1 being built on side_stream, 2 ready
    int tab_calls = 0;           // eligible calls seen with the current (model, chunk length)
    bool capturing = false;      // inside graph_call's stream capture: no cross-stream work may be started
    // hipGraph replay of the launch chain of repeated calls (TGP_OPT_GRAPH): slot 0 tgp_logpdf, 1 tgp_posterior_marginals
    struct GraphSlot {
        uint64_t key[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        bool seen = false;           // the key was used by the previous call (buffers are sized): the next one captures
        bool failed = false;         // capture was refused for this chain
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        int64_t replays = 0;
        uint64_t last_seq = 0;
    } gslot[2];
    int opt_graph = 0;           // TGP_OPT_GRAPH: 0 off (default: measured, no gain -- see DESIGN 9), 1 on, -1 on for T <= kGraphAutoT
    int64_t graph_replays = 0;
    uint64_t call_seq = 0;       // counts the compute entry points (check_ready): a graph is replayed only by the call right after its own
    int opt_group = 1;           // TGP_OPT_GROUP
    int opt_group_scan = 1;      // TGP_OPT_GROUP bit 2 (value & 4) switches the group-layout block scans off
    int opt_fuse = 1;            // TGP_OPT_FUSE_SCAN
    // timing
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double kernel_ms = 0.0, h2d_ms = 0.0, d2h_ms = 0.0;
    std::vector<ProfEntry> prof;
    std::vector<PendingEvt> pending;
    std::vector<hipEvent_t> evpool;

    int fail(int code, const std::string& msg) {
        err = msg;
        // a runtime error may leave a kernel of this call in flight: drain the stream before the caller can drop the handle, so that no block the
        // caching allocator parks (tgp_alloc.hpp) is still in use by the device (round-5 advice)
        if (code == TGP_EHIP && stream != nullptr) {
            (void)hipStreamSynchronize(stream);
            (void)hipGetLastError();
        }
        return code;
    }
};

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return h->fail(TGP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define TRY(expr)             \
    do {                      \
        int rc_ = (expr);     \
        if (rc_ != TGP_OK) return rc_; \
    } while (0)

namespace {

// ------------------------------------------------------------------------------------------ profiling
struct LaunchScope {
    tgp_handle* h;
    int idx = -1;
    hipEvent_t a = nullptr, b = nullptr;
    LaunchScope(tgp_handle* h_, const char* name) : h(h_) {
        if (!h->profile) return;
        for (size_t i = 0; i < h->prof.size(); ++i)
            if (h->prof[i].name == name) idx = (int)i;
        if (idx < 0) {
            h->prof.push_back(ProfEntry{name, 0.0, 0});
            idx = (int)h->prof.size() - 1;
        }
        auto get = [&]() {
            hipEvent_t e = nullptr;
            if (!h->evpool.empty()) {
                e = h->evpool.back();
                h->evpool.pop_back();
            } else {
                (void)hipEventCreate(&e);
            }
            return e;
        };
        a = get();
        b = get();
        (void)hipEventRecord(a, h->stream);
    }
    ~LaunchScope() {
        if (idx < 0) return;
        (void)hipEventRecord(b, h->stream);
        h->pending.push_back(PendingEvt{idx, a, b});
    }
};

void resolve_profile(tgp_handle* h) {
    for (auto& pe : h->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess) {
            h->prof[pe.idx].ms += ms;
            h->prof[pe.idx].calls += 1;
        }
        h->evpool.push_back(pe.a);
        h->evpool.push_back(pe.b);
    }
    h->pending.clear();
}

// ------------------------------------------------------------------------------------------ helpers
int bind_device(tgp_handle* h) {
    HIPCHK(hipSetDevice(h->device));
    return TGP_OK;
}

int stage_in(tgp_handle* h, DevBuf& buf, const void* src, size_t bytes, bool is_dev, const void** out) {
    if (src == nullptr) {
        *out = nullptr;
        return TGP_OK;
    }
    if (is_dev) {
        *out = src;
        return TGP_OK;
    }
    HIPCHK(buf.ensure(bytes));
    HIPCHK(hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, h->stream));
    *out = buf.p;
    return TGP_OK;
}

// device destination for an output array: the user's pointer if it is a device pointer, else staging
int stage_out(tgp_handle* h, DevBuf& buf, double* user, size_t bytes, bool is_dev, double** out) {
    if (user == nullptr) {
        *out = nullptr;
        return TGP_OK;
    }
    if (is_dev) {
        *out = user;
        return TGP_OK;
    }
    HIPCHK(buf.ensure(bytes));
    *out = buf.d();
    return TGP_OK;
}

int copy_back(tgp_handle* h, double* user, const double* dev, size_t bytes, bool is_dev) {
    if (user == nullptr || is_dev) return TGP_OK;
    HIPCHK(hipMemcpyAsync(user, dev, bytes, hipMemcpyDeviceToHost, h->stream));
    return TGP_OK;
}

void pack_state(int d, const double* m, const double* P, std::vector<double>& out) {
    out.clear();
    for (int i = 0; i < d; ++i) out.push_back(m[i]);
    for (int j = 0; j < d; ++j)
        for (int i = 0; i <= j; ++i) out.push_back(P[i + j * d]);
}
void unpack_state(int d, const double* pk, double* m, double* P) {
    for (int i = 0; i < d; ++i) m[i] = pk[i];
    int n = d;
    for (int j = 0; j < d; ++j)
        for (int i = 0; i <= j; ++i) {
            P[i + j * d] = pk[n];
            P[j + i * d] = pk[n];
            ++n;
        }
}

int upload_x0(tgp_handle* h, DevBuf& buf, const double* m, const double* P) {
    std::vector<double> pk;
    pack_state(h->d, m, P, pk);
    HIPCHK(buf.ensure(pk.size() * sizeof(double)));
    HIPCHK(hipMemcpyAsync(buf.p, pk.data(), pk.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));  // pk is a temporary
    return TGP_OK;
}

int state_size(int d) { return d + d * (d + 1) / 2; }
int felem_size(int d) { return d * d + 2 * d + d * (d + 1); }
int aelem_size(int d) { return d * d + d + d * (d + 1) / 2; }

void choose_chunk(tgp_handle* h, int for_mode = -1) {
    int64_t L0 = h->opt_chunk;
    if (L0 <= 0) {
        // One lane per chunk, 256-lane workgroups, 256 CUs: kernel time goes with ceil(workgroups / 256), so
        // size the chunk to land just under k full rounds of 256 workgroups. Measured at T = 1e7 (ms per bench step,
        // k = 1 (L0 = 153) against k = 2 (L0 = 77)):  d = 2: 0.78 / 0.66;  d = 3: 1.07 / 1.10;  d = 4: 1.71 / 2.05;
        // d = 5: 3.02 / 3.78;  d = 6: 6.9 / 9.7;  d = 8: 150 / 180.  From d = 4 on the kernels spill: one wave per
        // SIMD keeps the scratch working set cache-resident, and the block scans see half the elements.
        const int64_t round = 256LL * 256, Tm0 = h->T * h->p;
        // (round 2, T = 1e7, d = 3: logpdf alone 0.270 ms at k = 1 against 0.245 ms at k = 2 -- its two passes run two workgroups per CU;
        //  the combined call is the same either way. LTI layout only: a per-step model would be re-tiled at every change of L0.)
        // (closed-form SDE transitions at d = 3: passes capped at 256 registers, two workgroups per CU -- T = 1e7 combined call 1.49 -> 1.34 ms)
        const bool sde_cf3 = h->d == 3 && h->sde && h->sde_closed && h->opt_sde_closed;
        const int64_t kmin = (h->d <= 2 || (h->d == 3 && for_mode == 0 && h->lti) || sde_cf3) ? 2 : 1;
        int64_t k = (Tm0 + round * 160 - 1) / (round * 160);
        if (k < kmin) k = kmin;
        L0 = (Tm0 + round * k - 1) / (round * k);
        if (h->d >= 9) {
            // The out-of-line d >= 9 kernels keep their matrices in private memory (~225 d^2 bytes per lane: 18 KB at d = 9,
            // 55 KB at d = 16). The runtime sizes the scratch arena by the waves of a dispatch and ABORTS the queue
            // (HSA_STATUS_ERROR_OUT_OF_RESOURCES) somewhere between 280 and 480 MB -- summed over the handles (HIP streams)
            // alive in the process: keep a dispatch under ~115 MB.
            int64_t cap = (int64_t)(0.5e6 / ((double)h->d * h->d));
            cap = cap / 64 * 64;
            if (cap < 256) cap = 256;
            const int64_t Lmin = (Tm0 + cap - 1) / cap;
            if (L0 < Lmin) L0 = Lmin;
        }
        if (L0 < 8) L0 = 8;
    }
    const int64_t Tm = h->T * h->p;                       // processing steps (one scalar observation each)
    L0 = ((L0 + h->p - 1) / h->p) * h->p;                 // whole time steps per chunk
    if (L0 > Tm) L0 = Tm;
    h->L0 = (int)L0;
    h->n0 = (Tm + L0 - 1) / L0;
}

int scan_prepare(tgp_handle* h, ScanCtx& c, int monoid, int64_t n0) {
    c.monoid = monoid;
    c.NC = (monoid == kFilter) ? felem_size(h->d) : (monoid == kFilterAD) ? 2 * felem_size(h->d) : aelem_size(h->d);
    c.NS = (monoid == kFilterAD) ? 2 * state_size(h->d) : state_size(h->d);
    // the dual-number top-level scan runs in ONE 256-lane block (a 512-lane block caps it at 256 VGPRs and spills)
    // likewise the d >= 5 elements (>= 60 doubles each): keep the top block at 256 lanes
    const int64_t top_cap = (monoid == kFilterAD || h->d >= 5) ? 256 : (int64_t)kTopBS * kScanE;
    c.n.clear();
    c.n.push_back(n0);
    while (c.n.back() > top_cap) c.n.push_back((c.n.back() + 256 * kScanE - 1) / (256 * kScanE));
    size_t total = (size_t)c.NS;
    for (int64_t n : c.n) total += (size_t)(c.NC + c.NS) * (size_t)n;
    HIPCHK(c.slab.ensure(total * sizeof(double)));
    double* p = c.slab.d();
    c.E.clear();
    c.S.clear();
    for (int64_t n : c.n) {
        c.E.push_back(p);
        p += (size_t)c.NC * n;
        c.S.push_back(p);
        p += (size_t)c.NS * n;
    }
    c.fin = p;
    return TGP_OK;
}

static bool group_scan_ok(const tgp_handle* h, const ScanCtx& c) {
    if (!h->opt_group_scan || h->kt->group_scan_reduce == nullptr) return false;
    // under the lane-per-chunk passes the group-layout scans pay from d = 5 on (d = 6, T = 1e7: filter top scan 269 -> ~110 us)
    const bool pays = h->d >= 5 || h->opt_group == 2;
    if (&c == &h->F && c.monoid == kFilter) return h->group_active || (h->use_group && h->opt_group && pays);
    if (&c == &h->Rv && c.monoid == kAffineCov) return h->group_active || (h->use_group_aff && h->opt_group && pays);
    return false;
}

// first = 1: level 0 has been reduced by the producing chunk kernel itself (fused), start one level up
void scan_up(tgp_handle* h, ScanCtx& c, size_t first = 0) {
    // group-layout block scans (filter elements; affine elements with covariance, i.e. the smoother's reverse scan): always
    // with the group-per-chunk passes, and for d >= 7 also under the lane-per-chunk passes (same element formats; the
    // lane-per-element scans cost ~1 ms per launch there)
    const bool grp = group_scan_ok(h, c);
    for (size_t l = first; grp && l + 1 < c.n.size(); ++l) {
        LaunchScope ls(h, c.monoid == kFilter ? "k_group_scan_reduce<filter>" : "k_group_scan_reduce<affine>");
        h->kt->group_scan_reduce(c.monoid, c.E[l], c.n[l], c.E[l + 1], c.n[l + 1], h->stream);
    }
    if (grp) return;
    for (size_t l = first; l + 1 < c.n.size(); ++l) {
        LaunchScope ls(h, c.monoid == kFilter ? "k_scan_reduce<filter>" : c.monoid == kFilterAD ? "k_scan_reduce<filter,grad>" : "k_scan_reduce<affine>");
        h->kt->scan_reduce(c.monoid, c.E[l], c.n[l], c.E[l + 1], c.n[l + 1], h->stream);
    }
}

// last = 1: stop above level 0 (the consuming chunk kernel scans its own block against S[1], fused)
void scan_down(tgp_handle* h, ScanCtx& c, const double* x0dev, int last = 0) {
    const int top = (int)c.n.size() - 1;
    if (group_scan_ok(h, c)) {
        const bool flt = c.monoid == kFilter;
        {
            LaunchScope ls(h, flt ? "k_group_scan_apply<filter,top>" : "k_group_scan_apply<affine,top>");
            h->kt->group_scan_apply(c.monoid, c.E[top], c.n[top], x0dev, 1, c.S[top], c.fin, h->stream);
        }
        for (int l = top - 1; l >= last; --l) {
            LaunchScope ls(h, flt ? "k_group_scan_apply<filter>" : "k_group_scan_apply<affine>");
            h->kt->group_scan_apply(c.monoid, c.E[l], c.n[l], c.S[l + 1], c.n[l + 1], c.S[l], nullptr, h->stream);
        }
        return;
    }
    {
        LaunchScope ls(h, c.monoid == kFilter ? "k_scan_apply<filter,top>" : c.monoid == kFilterAD ? "k_scan_apply<filter,grad,top>" : "k_scan_apply<affine,top>");
        h->kt->scan_apply(c.monoid, c.n[top] <= 256 * kScanE ? 256 : kTopBS, c.E[top], c.n[top], x0dev, 1, c.S[top], c.fin, h->stream);
    }
    for (int l = top - 1; l >= last; --l) {
        LaunchScope ls(h, c.monoid == kFilter ? "k_scan_apply<filter>" : c.monoid == kFilterAD ? "k_scan_apply<filter,grad>" : "k_scan_apply<affine>");
        h->kt->scan_apply(c.monoid, 256, c.E[l], c.n[l], c.S[l + 1], c.n[l + 1], c.S[l], nullptr, h->stream);
    }
}

// reduce the top level of `c` to a single element (device resident; *elem_dev points into h->segtmp)
int scan_total_dev(tgp_handle* h, ScanCtx& c, const double** elem_dev) {
    const int top = (int)c.n.size() - 1;
    const double* src = c.E[top];
    int64_t n = c.n[top];
    HIPCHK(h->segtmp.ensure((size_t)c.NC * 4 * sizeof(double)));
    double* t0 = h->segtmp.d();
    double* t1 = t0 + (size_t)c.NC * 2;
    while (true) {
        int64_t nhi = (n + 256 * kScanE - 1) / (256 * kScanE);
        {
            LaunchScope ls(h, "k_scan_reduce<segment>");
            h->kt->scan_reduce(c.monoid, src, n, t0, nhi, h->stream);
        }
        if (nhi == 1) break;
        src = t0;
        n = nhi;
        std::swap(t0, t1);
    }
    *elem_dev = t0;
    return TGP_OK;
}

// ... and copy it to the host
int scan_total_to_host(tgp_handle* h, ScanCtx& c, double* elem_out) {
    const double* t0 = nullptr;
    TRY(scan_total_dev(h, c, &t0));
    HIPCHK(hipMemcpyAsync(elem_out, t0, (size_t)c.NC * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return TGP_OK;
}

__global__ void k_zero8(double* r) {
    if (threadIdx.x < 8) r[threadIdx.x] = 0.0;
}

struct CallTimer {
    tgp_handle* h;
    // clear == false: a later phase of a multi-phase (time-sharded) call, the flags of the earlier phases are kept.
    // The four hipEvents of a call (h2d / kernels / d2h split of tgp_last_timing) are recorded only with TGP_OPT_TIMING:
    // records and elapsed-time queries cost the host ~30 us per call, a few percent of a 0.4 ms logpdf.
    explicit CallTimer(tgp_handle* h_, bool clear = true) : h(h_) {
        h->steady_result_pending = false;      // (left set by a device-resident shard call that never reads its result record back)
        if (h->timing) (void)hipEventRecord(h->ev[0], h->stream);
        // lml / flags of this call (a kernel, not a memset node: the chain is also recorded into hipGraphs, kernel nodes only)
        if (clear) hipLaunchKernelGGL(k_zero8, dim3(1), dim3(64), 0, h->stream, h->result.d());
    }
    void inputs_done() { if (h->timing) (void)hipEventRecord(h->ev[1], h->stream); }
    void kernels_done() { if (h->timing) (void)hipEventRecord(h->ev[2], h->stream); }
    // one 64-byte D2H into pinned memory + ONE stream sync per call; then decode lml and the error flags
    int finish(double* lml_out = nullptr) {
        TRY(finish_enqueue());
        return finish_wait(h, lml_out);
    }
    int finish_enqueue() {
        HIPCHK(hipMemcpyAsync(h->host_result, h->result.p, 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (h->timing) (void)hipEventRecord(h->ev[3], h->stream);
        return TGP_OK;
    }
    static int finish_wait(tgp_handle* h, double* lml_out) {
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->timing) {
            float a = 0.f, b = 0.f, c = 0.f;
            (void)hipEventElapsedTime(&a, h->ev[0], h->ev[1]);
            (void)hipEventElapsedTime(&b, h->ev[1], h->ev[2]);
            (void)hipEventElapsedTime(&c, h->ev[2], h->ev[3]);
            h->h2d_ms = a;
            h->kernel_ms = b;
            h->d2h_ms = c;
        }
        resolve_profile(h);
        if (h->steady_result_pending) {
            h->steady_result_pending = false;
            h->steady_known = true;
            h->steady_pays = h->host_result[5] < 0.5 * (double)h->L0;
        }
        if (lml_out) *lml_out = h->host_result[0];
        int flags = 0;
        std::memcpy(&flags, &h->host_result[4], sizeof flags);
        if (h->host_result[2] != 0.0) return h->fail(TGP_ENOTPD, "innovation variance / predicted covariance not positive definite");
        if (flags) return h->fail(TGP_ENOTPD, "matrix not positive definite (Cholesky failed)");
        return TGP_OK;
    }
};

// The general (chunked-scan) engine's kernel table of a d = 5..8 model is chosen by a run-time known-answer check (variant_selftest:
// seconds on a machine that has not cached its verdict). A model the stationary-gain engine serves may never need it: tgp_model_set
// leaves the choice pending and the first entry point that really runs the general engine resolves it.
void resolve_table(tgp_handle* h) {
    if (!h->table_pending) return;
    h->table_pending = false;
    select_table(h, h->d, h->lti, h->variant_opt);
    h->reduce_valid = false;
    h->smoother_valid = false;
}
// general == false: the caller tries the stationary-gain engine first and calls resolve_table itself before the general path
int check_ready(tgp_handle* h, bool general = true) {
    if (!h) return TGP_EINVAL;
    if (!h->have_model) return h->fail(TGP_EINVAL, "no model set (call tgp_model_set first)");
    ++h->call_seq;
    TRY(bind_device(h));
    if (general) resolve_table(h);
    return TGP_OK;
}
// entry points the dense large-state engine (d > 16) does not serve
int scan_only(tgp_handle* h, const char* what) {
    if (h && h->is_dense) return h->fail(TGP_EUNSUPPORTED, std::string(what) + ": not available for state dimension d > 16 (dense path)");
    return TGP_OK;
}
int dense_fail(tgp_handle* h, int rc) { return rc == TGP_OK ? TGP_OK : h->fail(rc, tgp_dense::last_error(h->dense)); }

// ---- hipGraph replay (TGP_OPT_GRAPH) --------------------------------------------------------------------------------------
// A logpdf / posterior-marginals call on a short series is a chain of ~10-15 dependent launches of a few microseconds each:
// the host's enqueue cost and the per-launch dispatch dominate (BASELINE config 1, T = 1e4). The second call with the same
// device pointers records the chain into a hipGraph (stream capture: the same host code path, nothing is launched twice), later
// calls replay it with one hipGraphLaunch. Any change of model, option, stream or argument drops the graph.
constexpr int64_t kGraphAutoT = 1 << 20;
void drop_graphs(tgp_handle* h) {
    for (auto& g : h->gslot) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
        g = tgp_handle::GraphSlot{};
    }
}
bool graph_eligible(const tgp_handle* h, uint32_t flags, bool outputs) {
    if (h->is_dense || h->profile || h->timing || h->opt_graph == 0) return false;
    if (h->opt_graph < 0 && h->T > kGraphAutoT) return false;
    if (!(flags & TGP_IN_DEVICE) || (flags & TGP_REUSE_REDUCE)) return false;      // host buffers are staged with sizes the host decides per call
    return !outputs || (flags & TGP_OUT_DEVICE) != 0;
}
// body(): enqueues the kernels of the whole call on h->stream, no host synchronisation; the 64-byte result copy follows the graph
template <class Body>
int graph_call(tgp_handle* h, int slot, const uint64_t (&key)[8], Body&& body, double* lml_out) {
    auto& g = h->gslot[slot];
    // valid only for the call that directly follows its own previous use: any other entry point in between may have re-tiled the
    // model, re-sized a buffer or changed the chunking the recorded launches were made for
    const bool same = g.seen && std::memcmp(g.key, key, sizeof key) == 0 && g.last_seq + 1 == h->call_seq;
    g.last_seq = h->call_seq;
    if (same && g.exec) {
        HIPCHK(hipGraphLaunch(g.exec, h->stream));
        ++g.replays;
        ++h->graph_replays;
        HIPCHK(hipMemcpyAsync(h->host_result, h->result.p, 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        return CallTimer::finish_wait(h, lml_out);
    }
    if (same && !g.failed) {
        if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeRelaxed) == hipSuccess) {
            h->capturing = true;
            const int rc = body();
            h->capturing = false;
            hipGraph_t graph = nullptr;
            const hipError_t e2 = hipStreamEndCapture(h->stream, &graph);
            if (rc == TGP_OK && e2 == hipSuccess && graph) {
                hipGraphExec_t ex = nullptr;
                if (hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0) == hipSuccess) {
                    g.graph = graph;
                    g.exec = ex;
                    HIPCHK(hipGraphLaunch(ex, h->stream));
                    HIPCHK(hipMemcpyAsync(h->host_result, h->result.p, 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
                    return CallTimer::finish_wait(h, lml_out);
                }
            }
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipGetLastError();
            if (rc != TGP_OK && e2 == hipSuccess) return rc;     // an argument error found by the host code: report it
        }
        g.failed = true;     // this chain cannot be captured: plain launches from now on
    }
    if (!same) {
        const bool failed = false;
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
        g = tgp_handle::GraphSlot{};
        g.failed = failed;
        std::memcpy(g.key, key, sizeof key);
        g.seen = true;
        g.last_seq = h->call_seq;
    }
    TRY(body());
    HIPCHK(hipMemcpyAsync(h->host_result, h->result.p, 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return CallTimer::finish_wait(h, lml_out);
}

// General (per-step) layout: (re)build the time-tiled copy of the per-step arrays for the current chunk size.
// full: the transition record must hold A_k, Q_k themselves (the gradient passes read them beside their tangents)
int ensure_tiled(tgp_handle* h, bool full = false) {
    if (h->lti) return TGP_OK;
    const bool dt_mode = h->sde && h->sde_closed && !full && h->opt_sde_closed;
    if (h->tile_L0 == h->L0 && h->mv.tile_mask != 0u && h->tile_is_dt == dt_mode) return TGP_OK;
    uint32_t mask = tile_mask_of(h->raw);
    if (h->sde) mask |= dt_mode ? kTileDt : (kTileA | kTileQ);          // A, Q come from the time stamps, not from raw arrays
    const int nc_t = tile_offset_t(mask, 0u, h->d), nc_e = tile_offset_e(mask, 0u, h->d);
    const size_t nblk = (size_t)((h->n0 + 63) / 64) * 64;
    HIPCHK(h->tile_t.ensure((nblk * (size_t)(h->L0 / h->p) * (size_t)nc_t + 1) * sizeof(double)));
    HIPCHK(h->tile_e.ensure((nblk * (size_t)h->L0 * (size_t)nc_e + 1) * sizeof(double)));
    if (dt_mode) {
        LaunchScope ls(h, "k_tile_dt");
        hipLaunchKernelGGL(k_tile_dt, dim3((unsigned)((h->n0 + 255) / 256)), dim3(256), 0, h->stream, h->times_dev, h->T, h->ordering, h->L0 / h->p, h->n0,
                           h->tile_t.d());
    } else if (h->sde) {
        LaunchScope ls(h, "k_tile_sde");
        h->kt->tile_sde(h->bF.d(), h->bPinf.d(), h->times_dev, h->have_AQ1 ? h->bAQ1.d() : nullptr, h->T, h->ordering, h->L0 / h->p, h->n0, h->normF, h->tile_t.d(), h->stream);
    }
    {
        LaunchScope ls(h, "k_tile_model");
        // in SDE mode the transition record is already written: tile only the (optional) per-step emission arrays
        hipLaunchKernelGGL(k_tile_model, dim3((unsigned)((h->n0 + 255) / 256)), dim3(256), 0, h->stream, h->raw, h->d,
                           h->sde ? (mask & ~(kTileA | kTilea | kTileQ | kTileDt)) : mask, h->sde ? 0 : nc_t, nc_e, h->L0, h->n0, h->tile_t.d(),
                           h->tile_e.d());
    }
    h->tile_is_dt = dt_mode;
    if (h->sde && h->d <= kSdeBuildMaxD) {     // d <= 4: the passes that evaluate the transitions are a build of their own
        const KernelTable* base = kernel_table(h->d);
        const KernelTable* cf = dt_mode ? reinterpret_cast<const KernelTable*>(tgp_s::sde_kernel_table(h->d)) : nullptr;
        if (cf) {
            h->ktm = *base;
            h->ktm.reduce_filter = cf->reduce_filter;
            for (int m = 0; m < 4; ++m) h->ktm.apply_filter_m[m] = cf->apply_filter_m[m];
            h->ktm.smooth = cf->smooth;
            h->ktm.reduce_affine = cf->reduce_affine;
            h->ktm.apply_affine = cf->apply_affine;
            h->kt = &h->ktm;
        } else {
            h->kt = base;
        }
    }
    h->mv.sde = dt_mode ? h->bsde.d() : nullptr;
    h->mv.sde_first = h->have_AQ1 ? 1 : 0;
    h->mv.tile_t = h->tile_t.d();
    h->mv.tile_e = h->tile_e.d();
    h->mv.nc_t = nc_t;
    h->mv.nc_e = nc_e;
    h->mv.tile_mask = mask;
    h->tile_L0 = h->L0;
    return TGP_OK;
}

// forward pass 1 + upward scans (skipped when the caller vouches for reuse)
// for_mode: the pass-2 mode the caller will run next (0 logpdf ...), -1 unknown (time-sharded protocol)
int forward_reduce(tgp_handle* h, uint32_t flags, int for_mode = -1) {
    if ((flags & TGP_REUSE_REDUCE) && h->reduce_valid) return TGP_OK;
    // Group-per-chunk logpdf kernels (tgp_group.hpp). Measured at T = 1e7 (pass 1 + pass 2, ms; lane-per-chunk inlined
    // build in brackets): d = 5 1.9 (0.70), d = 6 2.2 (1.55), d = 7 2.9 (4.3), d = 8 3.4 (12.1) -- their time hardly
    // depends on d (LDS exchanges and shuffles, not flops), so they pay from d = 7 on (TGP_OPT_GROUP = 2 forces them).
    // General (per-step) layout, d = 5..16 (tgp_group.hpp GroupStep): the lane-per-chunk pass 1 holds the element AND the step's
    // own A, Q per lane and is bound by its own spill traffic from d = 6 (7.9 ms at T = 1e7 against a 0.9 ms HBM floor); the group
    // layout needs column j only. logpdf, filtering distributions and (pass 2 MODE 2 + pass 3) the posterior marginals.
    // (posterior path in the group layout, tgp_group_smooth.hpp: pass 2 + pass 3 take 7.7 + 7.4 ms at T = 1e7 for d = 7 and 8
    // alike -- 498 / 310 VGPRs, one wave per SIMD, bound by the D + 10 LDS exchanges of a step; the lane-per-chunk kernels
    // need 6.3 + 1.8 (+ 1.8 for their own pass 1) at d = 7 and 15.4 + 5.3 (+ 6.7) at d = 8: group from d = 8 on. Per-step layout:
    // measured at T = 1e7 -- d = 6: 7.2 + 5.6 ms in the group layout against ~11 ms lane-per-chunk, d = 7: 8.5 + 9.9 against ~30 ms;
    // from d = 9 the lane-per-chunk passes are out-of-line private-memory code (d = 14, T = 2e5: 1150 -> 13.8 ms) -- group from d = 7)
    const bool ps_layout = !h->lti && !h->sde && h->d >= 5 && h->d <= 16;
    const bool grp_post = for_mode == 2 && h->use_group_sm && h->ordering == 0 && h->kt->group_apply_posterior != nullptr && (h->lti || ps_layout) &&
                          (h->d >= 8 || (ps_layout && h->d >= 7) || h->opt_group == 2 || h->force_group_post);
    const bool ps_group = ps_layout && (for_mode == 0 || for_mode == 1 || grp_post);
    const bool group_pays = h->d >= 7 || h->opt_group == 2 || h->force_group_post || (ps_group && h->d >= 6);
    // filtering distributions (MODE 1) and the materialised posterior (MODE 3): group layout where the alternative is the
    // out-of-line build (d >= 9); MODE 3 shares the smoother's validation, Forward models only
    const bool grp_out = ((for_mode == 1 && h->use_group_m1) || (for_mode == 3 && h->use_group_m3 && h->ordering == 0)) &&
                         (h->d >= 9 || h->opt_group == 2);
    if ((for_mode == 0 || grp_post || grp_out || (ps_group && for_mode == 1 && h->use_group_m1)) && h->use_group && h->opt_group && group_pays &&
        h->kt->group_reduce_filter != nullptr && (h->lti || ps_group)) {
        // 8 chunks per wave: 16384 chunks are two waves per SIMD; longer chunks also mean fewer scan elements, and the
        // d >= 7 block scans (spill-bound, ~1.5 ms per launch) are what is left of the call
        int64_t L0 = h->opt_chunk;
        if (L0 <= 0) {
            // (sixteen lanes per chunk: half as many chunks for the same number of waves)
            const int64_t nch = h->kt->group_chunks_per_block == 32 ? 16384 : 8192;
            L0 = (h->T * h->p + nch - 1) / nch;
            if (L0 < 8) L0 = 8;
        }
        const int64_t Tm = h->T * h->p;                        // processing steps (one scalar observation each)
        L0 = ((L0 + h->p - 1) / h->p) * h->p;                  // whole time steps per chunk
        if (L0 > Tm) L0 = Tm;
        h->L0 = (int)L0;
        h->n0 = (Tm + L0 - 1) / L0;
        TRY(scan_prepare(h, h->F, kFilter, h->n0));
        {
            LaunchScope ls(h, h->lti ? "k_group_reduce_filter<lti>" : "k_group_reduce_filter<per-step>");
            h->kt->group_reduce_filter(h->mv, h->L0, h->n0, h->F.E[0], h->stream);
        }
        scan_up(h, h->F, 0);
        h->fused = false;
        h->group_active = true;
        h->reduce_valid = true;
        h->smoother_valid = false;
        return TGP_OK;
    }
    h->group_active = false;
    choose_chunk(h, for_mode);
    TRY(ensure_tiled(h));
    TRY(scan_prepare(h, h->F, kFilter, h->n0));
    // two or more scan levels: the level-0 reduce / apply live inside the chunk kernels (256 chunks per block == the
    // scan blocking), saving two launches and two passes over the element array per forward scan
    const bool fused = h->F.n.size() >= 2 && h->opt_fuse;
    // LTI model, one noise variance, no missing data, Forward, p = 1: the matrix parts of a chunk's element do not depend on the
    // observations -- every chunk of the same length has the same (Abar, C, J) and the same per-step (w, Cv, 1/s). They are
    // computed once (k_filter_table, one lane, L0 steps) and pass 1 only runs the vector half of the recursion per chunk
    // (d^2 + 3 d multiply-adds per step instead of ~6 d^3): T = 1e7, d = 3: 131 -> ~30 us.
    const bool shared_parts = h->opt_table && h->lti && !h->sde && h->p == 1 && h->ordering == 0 && h->mv.sR == 0 && h->mv.missing == nullptr &&
                              h->kt->reduce_filter_tab != nullptr && (int64_t)h->L0 * (2 * h->d + 1) <= kFilterTableLds;
    bool use_tab = false;
    if (shared_parts) {
        const int64_t Tm = h->T * h->p;
        const int nlast = (int)(Tm - (h->n0 - 1) * (int64_t)h->L0);
        if (h->tab_L0 != h->L0 || h->tab_nlast != nlast) {          // another chunking: start over
            if (h->tab_state == 1) (void)hipStreamSynchronize(h->side_stream);
            h->tab_state = 0;
            h->tab_calls = 0;
            h->tab_L0 = h->L0;
            h->tab_nlast = nlast;
        }
        if (h->tab_state == 1 && hipEventQuery(h->tab_ev) == hipSuccess) h->tab_state = 2;
        if (h->tab_state == 0 && h->opt_table == 2 && !h->capturing) {   // (tests: table built in line, on the handle's own stream)
            HIPCHK(h->ftab.ensure((size_t)h->kt->filter_table_size(h->L0) * sizeof(double)));
            LaunchScope ls(h, "k_filter_table");
            h->kt->filter_table(h->mv, h->L0, nlast, h->ftab.d(), h->stream);
            h->tab_state = 2;
        }
        if (h->tab_state == 2) {
            use_tab = true;
        } else if (h->tab_state == 0 && !h->capturing && ++h->tab_calls >= 2) {
            if (!h->side_stream) {
                HIPCHK(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
                HIPCHK(hipEventCreateWithFlags(&h->tab_ev, hipEventDisableTiming));
                HIPCHK(hipEventCreateWithFlags(&h->tab_dep, hipEventDisableTiming));
            }
            HIPCHK(h->ftab.ensure((size_t)h->kt->filter_table_size(h->L0) * sizeof(double)));
            HIPCHK(hipEventRecord(h->tab_dep, h->stream));           // the model blocks were uploaded on the handle's stream
            HIPCHK(hipStreamWaitEvent(h->side_stream, h->tab_dep, 0));
            h->kt->filter_table(h->mv, h->L0, nlast, h->ftab.d(), h->side_stream);
            HIPCHK(hipEventRecord(h->tab_ev, h->side_stream));
            h->tab_state = 1;
        }
    }
    if (use_tab) {
        LaunchScope ls(h, "k_reduce_filter<lti,shared parts>");
        h->kt->reduce_filter_tab(h->mv, h->L0, h->n0, h->ftab.d(), h->F.E[0], fused ? h->F.E[1] : nullptr, fused ? h->F.n[1] : 0, h->stream);
    } else {
        LaunchScope ls(h, h->lti ? "k_reduce_filter<lti>" : "k_reduce_filter<per-step>");
        h->kt->reduce_filter(h->lti, h->mv, h->L0, h->n0, h->F.E[0], fused ? h->F.E[1] : nullptr, fused ? h->F.n[1] : 0, h->stream);
    }
    scan_up(h, h->F, fused ? 1 : 0);
    h->fused = fused;
    h->reduce_valid = true;
    h->smoother_valid = false;
    return TGP_OK;
}

int set_obs(tgp_handle* h, const double* y, const uint8_t* missing, uint32_t flags) {
    if (y == nullptr) return h->fail(TGP_EINVAL, "y is NULL");
    const bool dev = (flags & TGP_IN_DEVICE) != 0;
    const void* p = nullptr;
    if ((flags & TGP_REUSE_REDUCE) && h->reduce_valid) return TGP_OK;  // caller vouches: same y as the previous call
    TRY(stage_in(h, h->by, y, (size_t)h->T * h->p * sizeof(double), dev, &p));
    h->mv.y = static_cast<const double*>(p);
    TRY(stage_in(h, h->bmiss, missing, (size_t)h->T * h->p, dev, &p));
    h->mv.missing = static_cast<const uint8_t*>(p);
    return TGP_OK;
}

// forward filter to the end. mode 0/1/2 as in chunk_apply_filter. Fills h->result (lml, nmiss, bad).
int forward_apply(tgp_handle* h, int mode, const FilterOut& fo, const double* x0dev = nullptr) {
    scan_down(h, h->F, x0dev ? x0dev : h->bx0.d(), h->fused ? 1 : 0);
    // stationary-covariance steps (tgp_chunk_body.inc): lane-per-chunk passes of a shared-layout model, one shared R, scalar
    // observations, no missing data
    h->mv.steady = nullptr;
    if (mode == 2) h->steady_calls = 0;
    if (h->steady_known_L0 != h->L0) {
        h->steady_known = false;
        h->steady_known_L0 = h->L0;
    }
    if (h->opt_steady && !(mode == 2 && h->steady_known && !h->steady_pays) && !h->group_active && h->lti && h->d <= kSteadyMaxD && h->p == 1 && h->mv.sR == 0 && h->mv.missing == nullptr &&
        (mode == 0 || mode == 1 || mode == 2)) {
        HIPCHK(h->steady_rec.ensure((size_t)(1 + h->d * (h->d + 1)) * (size_t)h->n0 * sizeof(double)));
        h->mv.steady = h->steady_rec.d();
        if (mode == 2) h->steady_calls = 1;
    }
    if (h->group_active) {
        const int64_t cpb = h->kt->group_chunks_per_block;
        const int64_t nb = (h->n0 + cpb - 1) / cpb;
        HIPCHK(h->partial.ensure((size_t)nb * 3 * sizeof(double)));
        if (mode == 2) {
            TRY(scan_prepare(h, h->Rv, kAffineCov, h->n0));
            LaunchScope ls(h, h->lti ? "k_group_apply_filter<lti,posterior>" : "k_group_apply_filter<per-step,posterior>");
            h->kt->group_apply_posterior(h->mv, h->L0, h->n0, h->F.S[0], fo.fs, h->Rv.E[0], h->partial.d(), nullptr, nullptr, nullptr, h->stream);
        } else if (mode == 3) {
            if (h->ordering != 0 || !h->use_group_m3) return h->fail(TGP_EINVAL, "internal: group-per-chunk elements with an unsupported materialise pass");
            LaunchScope ls(h, "k_group_apply_filter<lti,materialise>");
            h->kt->group_apply_posterior(h->mv, h->L0, h->n0, h->F.S[0], nullptr, nullptr, h->partial.d(), fo.G_out, fo.g_out, fo.L_out, h->stream);
        } else {
            LaunchScope ls(h, h->lti ? (mode == 1 ? "k_group_apply_filter<lti,filter>" : "k_group_apply_filter<lti,logpdf>")
                                     : (mode == 1 ? "k_group_apply_filter<per-step,filter>" : "k_group_apply_filter<per-step,logpdf>"));
            h->kt->group_apply_logpdf(h->mv, h->L0, h->n0, h->F.S[0], h->partial.d(), mode == 1 ? fo.m_out : nullptr, mode == 1 ? fo.P_out : nullptr,
                                      h->stream);
        }
        {
            LaunchScope ls(h, "k_finalize");
            hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, h->stream, h->partial.d(), nb, h->result.d(), (const double*)nullptr);
        }
        return TGP_OK;
    }
    const int64_t nblocks = (h->n0 + 255) / 256;
    HIPCHK(h->partial.ensure((size_t)nblocks * 3 * sizeof(double)));
    double* R0 = nullptr;
    if (mode == 2 || mode == 4) {
        TRY(scan_prepare(h, h->Rv, kAffineCov, h->n0));
        R0 = h->Rv.E[0];
    }
    {
        const char* nm = mode == 0 ? (h->lti ? "k_apply_filter<lti,logpdf>" : "k_apply_filter<per-step,logpdf>")
                         : mode == 1 ? (h->lti ? "k_apply_filter<lti,filter>" : "k_apply_filter<per-step,filter>")
                         : mode == 2 ? (h->lti ? "k_apply_filter<lti,posterior>" : "k_apply_filter<per-step,posterior>")
                         : mode == 4 ? (h->lti ? "k_apply_filter<lti,scratch>" : "k_apply_filter<per-step,scratch>")
                                     : (h->lti ? "k_apply_filter<lti,materialise>" : "k_apply_filter<per-step,materialise>");
        LaunchScope ls(h, nm);
        h->kt->apply_filter(h->lti, mode, h->mv, h->L0, h->n0, h->F.S[0], h->fused ? h->F.E[0] : nullptr, h->fused ? h->F.S[1] : nullptr,
                            h->fused ? h->F.n[1] : 0, fo, R0, h->partial.d(), h->stream);
    }
    {
        LaunchScope ls(h, "k_finalize");
        const double* rec = (mode == 2 && h->mv.steady != nullptr && !h->steady_known) ? h->mv.steady : nullptr;
        if (rec != nullptr) h->steady_result_pending = true;
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, h->stream, h->partial.d(), nblocks, h->result.d(), rec);
    }
    if (mode == 2 && h->mv.steady != nullptr && getenv("TGP_STEADY_DEBUG") != nullptr) {
        // debug aid: where in their chunks the waves switched to the mean-only steps (histogram over the chunks, by 8 steps)
        std::vector<double> ks((size_t)h->n0);
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(ks.data(), h->mv.steady, ks.size() * sizeof(double), hipMemcpyDeviceToHost));
        std::vector<int64_t> hist((size_t)h->L0 / 8 + 2, 0);
        for (double k : ks) ++hist[(size_t)std::min<double>(k, (double)h->L0) / 8];
        fprintf(stderr, "[tgp steady] L0 = %d, n0 = %lld; first mean-only step of the chunks (bins of 8; last bin = never):", h->L0, (long long)h->n0);
        for (size_t i = 0; i < hist.size(); ++i) if (hist[i]) fprintf(stderr, " [%zu..]:%lld", i * 8, (long long)hist[i]);
        fprintf(stderr, "\n");
    }
    return TGP_OK;
}

int* flag_ptr(tgp_handle* h) { return reinterpret_cast<int*>(h->result.d() + 4); }

#include "tgp_api_engines.inc"      // how a call reaches the stationary-gain scan engine (tgp_steady.hip), the one-launch form (tgp_modal.hip) and the sweep engine (tgp_sweep.hip)
}  // namespace

// ------------------------------------------------------------------------------------------ run-time variant check
// d = 5, 6 exist in two builds: out-of-line building blocks with the matrices in private memory (safe, slow) and fully
// inlined (fast, but the spill-heavy form hipcc has miscompiled for us). The first model of such a d on a process runs a
// known-answer comparison of the two builds over every entry point and both layouts; the fast build is used only if it
// reproduces the safe one to 1e-9.
static unsigned variant_selftest(int device, int d, bool lti);

// =========================================================================================== C ABI
extern "C" {

const char* tgp_version(void) { return "tgp_hip 0.1 (gfx950)"; }

}  // extern "C"

// ---- the handles' streams (round 5): a small per-device POOL, handed out round robin, never destroyed ------------------------------------------
// A HIP stream is an HSA queue, and queues are a scarce resource: a process that keeps thousands of models alive (a hyper-parameter search that
// binds a model per evaluation and leaves the old ones to the garbage collector -- examples/exact_time_learning.jl is that loop) used to abort
// in the runtime with HSA_STATUS_ERROR_OUT_OF_RESOURCES at queue creation (profiles/r04_sweeps.txt), each queue bringing its scratch arena
// along.  Handles now share kStreamPool streams per device.  Semantics are unchanged: a call enqueues on its handle's stream and returns
// after synchronising it; two handles that share a stream serialise on the device (their calls were independent anyway), and a
// synchronisation may wait for the other handle's call as well.  tgp_set_stream still substitutes the caller's own stream.
namespace {
// Two classes: kernels with next to no scratch (d <= 4, the dense MFMA engine) share kStreamPool streams; the scan engine's d = 5 .. 16 kernels --
// inlined builds with kilobytes, out-of-line builds with up to 55 KB of private memory per lane -- share kHeavyPool: the runtime sizes a queue's
// scratch arena for its hungriest kernel times every wave slot of the device and KEEPS it, and more than two or three such arenas alive abort the
// process from the runtime's queue-event thread (measured: the variant self-test of d = 9 with pools of 4 and 8 streams; 1 and 2 pass).
constexpr int kStreamPool = 8, kHeavyPool = 2;
constexpr int kStreamPoolDevices = 64;
std::mutex g_stream_mutex;
hipStream_t g_streams[kStreamPoolDevices][kStreamPool + kHeavyPool] = {};
unsigned g_stream_next[kStreamPoolDevices][2] = {};
hipError_t pool_stream(int device, hipStream_t* out, bool heavy = false) {
    if (device < 0 || device >= kStreamPoolDevices) return hipStreamCreateWithFlags(out, hipStreamNonBlocking);      // (never pooled: destroyed with the handle)
    std::lock_guard<std::mutex> lock(g_stream_mutex);
    static const unsigned pool = [] {
        const char* v = std::getenv("TGP_STREAM_POOL");      // (1 .. kStreamPool; measurements)
        const int n = v ? std::atoi(v) : kStreamPool;
        return (unsigned)(n < 1 ? 1 : (n > kStreamPool ? kStreamPool : n));
    }();
    const unsigned k = heavy ? kStreamPool + g_stream_next[device][1]++ % kHeavyPool : g_stream_next[device][0]++ % pool;
    if (g_streams[device][k] == nullptr) {
        hipError_t rc = hipStreamCreateWithFlags(&g_streams[device][k], hipStreamNonBlocking);
        if (rc != hipSuccess) return rc;
    }
    *out = g_streams[device][k];
    return hipSuccess;
}
bool heavy_stream(int device, hipStream_t s) {
    if (device < 0 || device >= kStreamPoolDevices) return false;
    for (int k = kStreamPool; k < kStreamPool + kHeavyPool; ++k)
        if (g_streams[device][k] == s && s != nullptr) return true;
    return false;
}
bool pooled_stream(int device) { return device >= 0 && device < kStreamPoolDevices; }
// Handles that share a pooled stream also share its LOCK for the length of a call (round-5 advice): the one-launch kernels and their host halves
// talk through pinned flags and watch the stream with hipStreamQuery, graph capture owns the stream while it lasts, and the caching allocator
// assumes that a parked block is not in use -- none of which holds if a second thread enqueues on the same stream mid-call.  Calls of handles on
// one stream serialise on the device anyway; with the lock they serialise on the host as well.  A stream the caller substituted (tgp_set_stream)
// is the caller's to keep to one thread at a time.
std::recursive_mutex g_stream_locks[kStreamPoolDevices][kStreamPool + kHeavyPool];
std::recursive_mutex* stream_lock_of(int device, hipStream_t s) {
    if (device < 0 || device >= kStreamPoolDevices || s == nullptr) return nullptr;
    for (int k = 0; k < kStreamPool + kHeavyPool; ++k)
        if (g_streams[device][k] == s) return &g_stream_locks[device][k];
    return nullptr;
}
struct StreamGuard {
    std::recursive_mutex* m = nullptr;
    explicit StreamGuard(const tgp_handle* h) {
        if (h && h->stream != nullptr && h->stream == h->own_stream) m = stream_lock_of(h->device, h->stream);
        if (m) m->lock();
    }
    ~StreamGuard() {
        if (m) m->unlock();
    }
    StreamGuard(const StreamGuard&) = delete;
    StreamGuard& operator=(const StreamGuard&) = delete;
};
}  // namespace

extern "C" {

int tgp_create(tgp_handle** out, int device) {
    if (!out) return TGP_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return TGP_EHIP;
    if (device < 0 || device >= ndev) return TGP_EINVAL;
    tgp_handle* h = new tgp_handle();
    h->device = device;
    if (hipSetDevice(device) != hipSuccess || pool_stream(device, &h->own_stream) != hipSuccess) {
        delete h;
        return TGP_EHIP;
    }
    h->stream = h->own_stream;
    for (auto& e : h->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            delete h;
            return TGP_EHIP;
        }
    if (tgp_alloc::host_malloc(reinterpret_cast<void**>(&h->host_result), 8 * sizeof(double), hipHostMallocDefault) != hipSuccess ||
        h->result.ensure(8 * sizeof(double)) != hipSuccess) {
        delete h;
        return TGP_EHIP;
    }
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ncu <= 0) ncu = 256;
        h->num_cu = ncu;
    }
    *out = h;
    return TGP_OK;
}

int tgp_destroy(tgp_handle* h) {
    if (!h) return TGP_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->side_stream) {
        (void)hipStreamSynchronize(h->side_stream);
        (void)hipEventDestroy(h->tab_ev);
        (void)hipEventDestroy(h->tab_dep);
        (void)hipStreamDestroy(h->side_stream);
    }
    if (h->wide) tgp_wide::destroy(h->wide);
    h->wide = nullptr;
    drop_graphs(h);
    for (DevBuf* b : {&h->bA, &h->ba, &h->bQ, &h->bH, &h->bh, &h->bR, &h->bx0, &h->bx0r, &h->bx0fold, &h->by, &h->bmiss, &h->bRnew, &h->beps_t,
                      &h->beps_e, &h->bo1, &h->bo2, &h->bo3, &h->bflip_y, &h->bflip_m, &h->bflip_P, &h->F.slab, &h->Rv.slab, &h->fs, &h->partial, &h->result, &h->segtmp, &h->tile_t, &h->tile_e, &h->Fad.slab, &h->btan, &h->bx0ad, &h->tile_tan, &h->balt, &h->bF, &h->bPinf, &h->btimes, &h->bAQ1, &h->bsde, &h->ftab, &h->btau})
        b->release();
    for (auto& e : h->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& pe : h->pending) {
        (void)hipEventDestroy(pe.a);
        (void)hipEventDestroy(pe.b);
    }
    for (auto& e : h->evpool) (void)hipEventDestroy(e);
    if (h->dense) tgp_dense::destroy(h->dense);
    if (h->steady2) tgp_steady::destroy(h->steady2);
    if (h->modal) tgp_modal::destroy(h->modal);
    if (h->sweep) tgp_sweep::destroy(h->sweep);
    if (h->host_result) (void)tgp_alloc::host_free(h->host_result);
    if (h->adj_host) (void)tgp_alloc::host_free(h->adj_host);
    if (h->flt_host) (void)tgp_alloc::host_free(h->flt_host);
    if (h->sm_sync) (void)tgp_alloc::host_free(h->sm_sync);
    if (h->own_stream && !pooled_stream(h->device)) (void)hipStreamDestroy(h->own_stream);      // (pool streams live as long as the process)
    delete h;
    return TGP_OK;
}

const char* tgp_last_error(const tgp_handle* h) { return h ? h->err.c_str() : "null handle"; }

int tgp_set_option(tgp_handle* h, int option, int64_t value) {
    if (!h) return TGP_EINVAL;
    drop_graphs(h);
    if (option == TGP_OPT_GRAPH) {
        if (value < -1 || value > 1) return h->fail(TGP_EINVAL, "TGP_OPT_GRAPH must be -1 (auto), 0 or 1");
        h->opt_graph = (int)value;
        return TGP_OK;
    }
    if (option == TGP_OPT_CHUNK) {
        if (value < 0 || value > 4096) return h->fail(TGP_EINVAL, "TGP_OPT_CHUNK out of range");
        h->opt_chunk = value;
        h->reduce_valid = false;
        h->smoother_valid = false;
        return TGP_OK;
    }
    if (option == TGP_OPT_PROFILE) {
        h->profile = value != 0;
        return TGP_OK;
    }
    if (option == TGP_OPT_VARIANT) {
        if (value < 0 || value > 3) return h->fail(TGP_EINVAL, "TGP_OPT_VARIANT must be 0, 1, 2 or 3");
        h->variant_opt = (int)value;
        if (h->have_model) {
            h->table_pending = false;
            select_table(h, h->d, h->lti, (int)value);
            h->reduce_valid = false;
            h->smoother_valid = false;
        }
        return TGP_OK;
    }
    if (option == TGP_OPT_SPLIT_SMOOTHER) {
        if (value < 0 || value > 2) return h->fail(TGP_EINVAL, "TGP_OPT_SPLIT_SMOOTHER must be 0, 1 or 2");
        h->opt_split = (int)value;
        h->reduce_valid = false;
        h->smoother_valid = false;
        return TGP_OK;
    }
    if (option == TGP_OPT_SHARED_PARTS) {
        h->opt_table = value != 0 ? (int)value : 0;      // 1 default policy; 2 build the table synchronously on the first call (tests)
        h->reduce_valid = false;
        h->smoother_valid = false;
        return TGP_OK;
    }
    if (option == TGP_OPT_STEADY) {
        h->opt_steady = value != 0;
        h->opt_steady2 = value >= 2;
        h->opt_modal = value >= 3;
        h->steady2_state = 0;
        h->modal_state = 0;
        h->smooth_state = 0;
        h->steady_known = false;
        h->smoother_valid = false;
        return TGP_OK;
    }
    if (option == TGP_OPT_SWEEP) {
        h->opt_sweep = value != 0;
        h->sweep_state = 0;
        return TGP_OK;
    }
    if (option == TGP_OPT_WIDE) {
        h->opt_wide = value != 0;
        h->wide_state = 0;
        h->wide_post_state = 0;
        return TGP_OK;
    }
    if (option == TGP_OPT_STREAM_MIN_T) {
        h->opt_stream_min_T = value < 0 ? -1 : (long long)value;
        return TGP_OK;
    }
    if (option == TGP_OPT_SWEEP_CHUNK || option == TGP_OPT_SWEEP_WARMUP || option == TGP_OPT_SWEEP_WARMUP_BACK) {
        if (value < 0 || value > (1 << 20)) return h->fail(TGP_EINVAL, "sweep geometry out of range");
        (option == TGP_OPT_SWEEP_CHUNK ? h->sweep_fC : option == TGP_OPT_SWEEP_WARMUP ? h->sweep_fW : h->sweep_fWb) = (int)value;
        h->sweep_state = 0;
        h->sweep_W = h->sweep_Wb = 0;
        return TGP_OK;
    }
    if (option == TGP_OPT_SDE_CLOSED_FORM) {
        h->opt_sde_closed = value != 0;
        h->reduce_valid = false;
        h->smoother_valid = false;
        return TGP_OK;
    }
    if (option == TGP_OPT_DENSE_FUSED) {
        h->dense_fused = value < 0 ? 0 : (value > 2 ? 2 : (int)value);       // 0 / 1 / 2; takes effect at the next tgp_model_set
        return TGP_OK;
    }
    if (option == TGP_OPT_DENSE_STRUCTURE) {
        h->dense_structure = value != 0;   // takes effect at the next tgp_model_set
        return TGP_OK;
    }
    if (option == TGP_OPT_TIMING) {
        h->timing = value != 0;
        return TGP_OK;
    }
    if (option == TGP_OPT_GROUP) {
        if (value < 0 || value > 6 || (value & 3) == 3) return h->fail(TGP_EINVAL, "TGP_OPT_GROUP must be 0, 1 or 2 (+ 4)");
        h->opt_group = (int)(value & 3);
        h->opt_group_scan = (value & 4) ? 0 : 1;
        h->reduce_valid = false;
        h->smoother_valid = false;
        return TGP_OK;
    }
    if (option == TGP_OPT_FUSE_SCAN) {
        h->opt_fuse = value != 0;
        h->reduce_valid = false;
        h->smoother_valid = false;
        return TGP_OK;
    }
    return h->fail(TGP_EINVAL, "unknown option");
}

int tgp_kernel_variant(const tgp_handle* h) {
    if (!h || !h->have_model) return 0;
    if (h->is_dense) return 16 + tgp_dense::structure(h->dense);   // dense path: 16 | (A sparse) | 2 (H sparse)
    if (h->table_pending && bind_device(const_cast<tgp_handle*>(h)) == TGP_OK) resolve_table(const_cast<tgp_handle*>(h));   // a diagnostic of the general engine
    return h->variant_code;
}

int64_t tgp_graph_replays(const tgp_handle* h) { return h ? h->graph_replays : 0; }

int tgp_steady_steps(tgp_handle* h, int64_t* mean_only, int64_t* total) {
    if (!h || !mean_only || !total) return TGP_EINVAL;
    *mean_only = 0;
    *total = h->T * h->p;
    if (h->is_dense) return TGP_OK;
    if (h->dense_last_n0 >= 0) {              // k_filter_one / k_adjoint_one: the same, on dense powers
        *mean_only = h->T - h->dense_last_n0;
        return TGP_OK;
    }
    if (h->modal_last && h->modal) {          // one-launch path: every step beyond the head's n0 ran with the stationary gains
        *mean_only = h->T - tgp_modal::last_plan(h->modal).n0;
        return TGP_OK;
    }
    if (h->steady2_last && h->steady2) {      // stationary-gain engine: every step beyond the head's n0 ran with the stationary gains
        int64_t info[4] = {0, 0, 0, 0};
        HIPCHK(hipSetDevice(h->device));
        if (tgp_steady::last_info(h->steady2, h->stream, info) != 0) return h->fail(TGP_EHIP, "tgp_steady::last_info");
        if (info[3] == 1) *mean_only = h->T - info[0];
        return TGP_OK;
    }
    if (h->steady_calls == 0 || h->mv.steady == nullptr) return TGP_OK;     // the last forward pass of a posterior path ran full steps only
    HIPCHK(hipSetDevice(h->device));
    std::vector<double> ks((size_t)h->n0);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(ks.data(), h->mv.steady, ks.size() * sizeof(double), hipMemcpyDeviceToHost));
    const int64_t Tm = h->T * h->p;
    for (int64_t c = 0; c < h->n0; ++c) {
        const int64_t len = std::min<int64_t>(h->L0, Tm - c * (int64_t)h->L0), k = (int64_t)ks[(size_t)c];
        if (k < len) *mean_only += len - k;
    }
    return TGP_OK;
}

int tgp_sweep_info(tgp_handle* h, int64_t* info, double* dist) {
    if (!h) return TGP_EINVAL;
    if (info) {
        for (int i = 0; i < 8; ++i) info[i] = h->sweep_info[i];
        info[0] = h->sweep_last ? 1 : 0;
        info[7] = h->sweep_state;
    }
    if (dist) { dist[0] = h->sweep_dist[0]; dist[1] = h->sweep_dist[1]; }
    return TGP_OK;
}

int tgp_set_stream(tgp_handle* h, void* hip_stream) {
    if (!h) return TGP_EINVAL;
    drop_graphs(h);
    (void)hipStreamSynchronize(h->stream);
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
    return TGP_OK;
}

int tgp_get_stream(tgp_handle* h, void** hip_stream) {
    if (!h || !hip_stream) return TGP_EINVAL;
    *hip_stream = static_cast<void*>(h->stream);
    return TGP_OK;
}

// The CPUs next to a device: /sys/bus/pci/devices/<bus id>/local_cpulist ("0-63,128-191").  A host thread on the other socket pays a second hop for
// every flag in pinned memory, every kernel-argument line and every doorbell: measured on the two-socket MI355X boxes of this pool as ~13 us per
// headline step (0.107 against 0.121 ms), the whole difference "from box to box" of rounds 5 and 6.
int tgp_bind_host_thread(int device) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) {
        (void)hipGetLastError();
        return TGP_EHIP;
    }
    for (char* c = bus; *c; ++c) *c = (char)tolower((unsigned char)*c);
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/local_cpulist";
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return TGP_EUNSUPPORTED;
    char line[4096] = {0};
    const bool got = fgets(line, (int)sizeof line, f) != nullptr;
    fclose(f);
    if (!got) return TGP_EUNSUPPORTED;
    cpu_set_t allowed, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return TGP_EUNSUPPORTED;
    int n = 0;
    for (const char* p = line; *p;) {
        char* end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') {
            b = strtol(p + 1, &end, 10);
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
            if (c >= 0 && CPU_ISSET((int)c, &allowed)) {
                CPU_SET((int)c, &want);
                ++n;
            }
        while (*p == ',' || *p == ' ' || *p == '\n') ++p;
    }
    if (n == 0) return TGP_EUNSUPPORTED;      // (nothing of the device's node is allowed to this process: leave the thread where it is)
    return sched_setaffinity(0, sizeof want, &want) == 0 ? TGP_OK : TGP_EUNSUPPORTED;
}

int tgp_stream_synchronize(void* hip_stream) {
    return hipStreamSynchronize(static_cast<hipStream_t>(hip_stream)) == hipSuccess ? TGP_OK : TGP_EHIP;
}

int tgp_model_set(tgp_handle* h, int64_t T, int d, int p, int ordering, uint32_t flags, const double* A, const double* a,
                  const double* Q, const double* H, const double* hh, const double* R, const double* x0m, const double* x0P) {
    if (h) drop_graphs(h);
    if (!h) return TGP_EINVAL;
    TRY(bind_device(h));
    h->have_model = false;
    h->steady_known = false;
    h->steady_calls = 0;
    h->mv.steady = nullptr;
    h->steady2_state = 0;
    h->steady2_last = false;
    h->modal_state = 0;
    h->smooth_state = 0;
    h->modal_last = false;
    h->dense_last_n0 = -1;
    h->hostm.clear();
    h->sweep_state = 0;
    h->sweep_last = false;
    h->sweep_W = h->sweep_Wb = 0;
    if (!h->binding_sde) { h->sde_coef_host.clear(); h->sde_A1Q1_host.clear(); }
    h->fold_valid = false;
    h->reduce_valid = false;
    h->smoother_valid = false;
    h->sde = false;
    if (h->tab_state == 1) (void)hipStreamSynchronize(h->side_stream);       // a table of the previous model may still be in flight
    h->tab_state = 0;
    h->tab_calls = 0;
    h->tab_L0 = 0;
    if (T <= 0) return h->fail(TGP_EINVAL, "T must be positive");
    if (ordering != 0 && ordering != 1) return h->fail(TGP_EINVAL, "ordering must be 0 (Forward) or 1 (Reverse)");
    if (pooled_stream(h->device)) {      // the stream class of the model's kernels (see pool_stream): calls are blocking, nothing is in flight here
        const bool heavy = d >= 5 && d <= 16;
        if (heavy != heavy_stream(h->device, h->own_stream)) {
            hipStream_t ns = nullptr;
            HIPCHK(pool_stream(h->device, &ns, heavy));
            if (h->stream == h->own_stream) h->stream = ns;
            h->own_stream = ns;
        }
    }
    h->is_dense = false;
    if (d > 16) {
        // dense large-state path (tgp_dense.hip): the arrays are re-packed into the padded MFMA layout, nothing is borrowed
        if (p < 1) return h->fail(TGP_EINVAL, "p must be positive");
        if (!A || !a || !Q || !H || !hh || !R || !x0m || !x0P) return h->fail(TGP_EINVAL, "null model array");
        if (!h->dense) h->dense = tgp_dense::create(h->device);
        const bool dev = (flags & TGP_DEVICE_PTRS) != 0;
        auto cnt = [&](uint32_t bit, int64_t per) { return (size_t)((flags & bit) ? per : per * T) * sizeof(double); };
        const void *pA, *pa, *pQ, *pH, *ph, *pR;
        TRY(stage_in(h, h->bA, A, cnt(TGP_SHARED_A, (int64_t)d * d), dev, &pA));
        TRY(stage_in(h, h->ba, a, cnt(TGP_SHARED_a, d), dev, &pa));
        TRY(stage_in(h, h->bQ, Q, cnt(TGP_SHARED_Q, (int64_t)d * d), dev, &pQ));
        TRY(stage_in(h, h->bH, H, cnt(TGP_SHARED_H, (int64_t)p * d), dev, &pH));
        TRY(stage_in(h, h->bh, hh, cnt(TGP_SHARED_h, p), dev, &ph));
        TRY(stage_in(h, h->bR, R, cnt(TGP_SHARED_R, p), dev, &pR));
        tgp_dense::ModelDesc md{};
        md.T = T; md.d = d; md.p = p; md.ordering = ordering;
        md.A = (const double*)pA; md.a = (const double*)pa; md.Q = (const double*)pQ;
        md.H = (const double*)pH; md.h = (const double*)ph; md.R = (const double*)pR;
        md.sA = (flags & TGP_SHARED_A) ? 0 : (int64_t)d * d;
        md.sa = (flags & TGP_SHARED_a) ? 0 : d;
        md.sQ = (flags & TGP_SHARED_Q) ? 0 : (int64_t)d * d;
        md.sH = (flags & TGP_SHARED_H) ? 0 : (int64_t)p * d;
        md.sh = (flags & TGP_SHARED_h) ? 0 : p;
        md.sR = (flags & TGP_SHARED_R) ? 0 : p;
        md.x0m = x0m; md.x0P = x0P;
        tgp_dense::set_profile(h->dense, h->profile);
        tgp_dense::set_structure(h->dense, h->dense_structure);
        tgp_dense::set_fused(h->dense, h->dense_fused);
        TRY(dense_fail(h, tgp_dense::model_set(h->dense, md, h->stream)));
        HIPCHK(hipStreamSynchronize(h->stream));
        for (DevBuf* b : {&h->bA, &h->bQ, &h->bH}) b->release();   // the packed copy is what the kernels read
        h->T = T; h->d = d; h->p = p; h->ordering = ordering;
        h->x0m.assign(x0m, x0m + d);                        // tgp_rand draws x0 on the host
        h->x0P.assign(x0P, x0P + (size_t)d * d);
        h->mv.small_out = (flags & TGP_SMALL_OUTPUT) != 0 || p > 1;
        h->lti = false;
        h->widem.clear();
        h->wide_state = 0;
        h->wide_post_state = 0;
        {
            // (every block shared; the emission offset may be per step -- a mean function at the inputs: the gains do not see it)
            const uint32_t need_shared = TGP_SHARED_A | TGP_SHARED_a | TGP_SHARED_Q | TGP_SHARED_H | TGP_SHARED_R;
            h->wide_ht = nullptr;
            if ((flags & need_shared) == need_shared && p == 1 && ordering == 0 && tgp_wide::supports(d)) {
                const size_t dd = (size_t)d * d;
                const bool h_per_step = !(flags & TGP_SHARED_h);
                h->widem.assign(2 * dd + 2 * (size_t)d + 2, 0.0);
                double* q = h->widem.data();
                const struct { const double* src; size_t n; } parts[6] = {{A, dd}, {a, (size_t)d}, {Q, dd}, {H, (size_t)d}, {hh, 1}, {R, 1}};
                size_t off = 0;
                for (int k = 0; k < 6; ++k) {
                    const auto& pt = parts[k];
                    if (k == 4 && h_per_step) { off += 1; continue; }      // (hh stays 0: the kernels subtract h_t)
                    if (dev) HIPCHK(hipMemcpy(q + off, pt.src, pt.n * sizeof(double), hipMemcpyDeviceToHost));
                    else std::memcpy(q + off, pt.src, pt.n * sizeof(double));
                    off += pt.n;
                }
                if (h_per_step) h->wide_ht = static_cast<const double*>(ph);
            }
        }
        h->is_dense = true;
        h->have_model = true;
        return TGP_OK;
    }
    if (p < 1 || p > 64) return h->fail(TGP_EUNSUPPORTED, "observation dimension p must be in 1..64 (diagonal noise) on the scan path");
    const KernelTable* kt = kernel_table(d);
    if (!kt) return h->fail(TGP_EUNSUPPORTED, "state dimension d must be positive");
    {
        const uint32_t lb = TGP_SHARED_A | TGP_SHARED_a | TGP_SHARED_Q | TGP_SHARED_H | TGP_SHARED_h;
        const bool lti_model = (flags & lb) == lb && !h->binding_sde;     // (tgp_model_set_sde: transitions are per-step whatever the flags of its placeholder blocks say)
        // (what steady2_eligible will ask of the model: such a model's logpdf / posterior-marginals / adjoint calls never touch the table)
        h->table_pending = h->variant_opt == 0 && h->opt_steady2 && lti_model && p == 1 && (flags & TGP_SHARED_R) && ordering == 0 &&
                           (tgp_steady::supports(d) || (tgp_wide::supports(d) && h->opt_wide));      // (... or the wide-state engine's, tgp_wide.hip)
        select_table(h, d, lti_model, h->table_pending ? 1 : h->variant_opt);
        kt = h->kt;
    }
    if (!A || !a || !Q || !H || !hh || !R || !x0m || !x0P) return h->fail(TGP_EINVAL, "null model array");
    h->kt = kt;
    h->T = T;
    h->d = d;
    h->p = p;
    h->ordering = ordering;
    const bool dev = (flags & TGP_DEVICE_PTRS) != 0;
    auto cnt = [&](uint32_t bit, int64_t per) { return (size_t)((flags & bit) ? per : per * T) * sizeof(double); };
    const void *pA, *pa, *pQ, *pH, *ph, *pR;
    TRY(stage_in(h, h->bA, A, cnt(TGP_SHARED_A, d * d), dev, &pA));
    TRY(stage_in(h, h->ba, a, cnt(TGP_SHARED_a, d), dev, &pa));
    TRY(stage_in(h, h->bQ, Q, cnt(TGP_SHARED_Q, d * d), dev, &pQ));
    TRY(stage_in(h, h->bH, H, cnt(TGP_SHARED_H, (int64_t)p * d), dev, &pH));
    TRY(stage_in(h, h->bh, hh, cnt(TGP_SHARED_h, p), dev, &ph));
    TRY(stage_in(h, h->bR, R, cnt(TGP_SHARED_R, p), dev, &pR));
    ModelView& mv = h->mv;
    mv = ModelView{};
    mv.T = T * p;       // processing steps
    mv.Tt = T;
    mv.p = p;
    mv.small_out = (flags & TGP_SMALL_OUTPUT) != 0 || p > 1;
    mv.ordering = ordering;
    mv.A = (const double*)pA;
    mv.a = (const double*)pa;
    mv.Q = (const double*)pQ;
    mv.H = (const double*)pH;
    mv.h = (const double*)ph;
    mv.R = (const double*)pR;
    mv.sA = (flags & TGP_SHARED_A) ? 0 : d * d;
    mv.sa = (flags & TGP_SHARED_a) ? 0 : d;
    mv.sQ = (flags & TGP_SHARED_Q) ? 0 : d * d;
    mv.sH = (flags & TGP_SHARED_H) ? 0 : (int64_t)p * d;
    mv.sh = (flags & TGP_SHARED_h) ? 0 : p;
    mv.sR = (flags & TGP_SHARED_R) ? 0 : p;
    const uint32_t lti_bits = TGP_SHARED_A | TGP_SHARED_a | TGP_SHARED_Q | TGP_SHARED_H | TGP_SHARED_h;
    h->lti = (flags & lti_bits) == lti_bits;
    h->raw = mv;
    h->tile_L0 = 0;
    h->x0m.assign(x0m, x0m + d);
    h->x0P.assign(x0P, x0P + d * d);
    TRY(upload_x0(h, h->bx0, x0m, x0P));
    HIPCHK(hipStreamSynchronize(h->stream));
    // (the shared blocks of a model the host plans read -- tgp_steady_plan.hpp: every block shared; tgp_sweep_plan.hpp: noise / offset may be per step,
    //  SDE-bound models bring their A1, Q1 through tgp_model_set_sde)
    const uint32_t host_bits = TGP_SHARED_A | TGP_SHARED_a | TGP_SHARED_Q | TGP_SHARED_H;
    const bool one_launch_model = h->lti && !h->binding_sde && p == 1 && (flags & TGP_SHARED_R) && tgp_steady::supports(d);      // what `hostm` stands for
    h->sweepm.clear();
    if ((flags & host_bits) == host_bits && p == 1 && tgp_steady::supports(d)) {
        const size_t dd = (size_t)d * d;
        h->sweepm.assign(2 * dd + 2 * (size_t)d + 2, 0.0);
        double* q = h->sweepm.data();
        const struct { const double* src; size_t n; } parts[6] = {{A, dd}, {a, (size_t)d}, {Q, dd}, {H, (size_t)d}, {hh, 1}, {R, 1}};
        size_t off = 0;
        for (const auto& pt : parts) {
            if (dev) HIPCHK(hipMemcpy(q + off, pt.src, pt.n * sizeof(double), hipMemcpyDeviceToHost));
            else std::memcpy(q + off, pt.src, pt.n * sizeof(double));
            off += pt.n;
        }
        // a representative noise variance (the sweep engine's warm-up estimate): the median of a sample of the per-step values below the
        // "missing" level of missings.jl:43
        h->sweep_Rrep = q[2 * dd + 2 * (size_t)d + 1];
        if (!(flags & TGP_SHARED_R)) {
            const size_t ns = (size_t)std::min<int64_t>(T, 4096);
            std::vector<double> smp(ns);
            if (dev) HIPCHK(hipMemcpy(smp.data(), R, ns * sizeof(double), hipMemcpyDeviceToHost));
            else std::memcpy(smp.data(), R, ns * sizeof(double));
            std::vector<double> okv;
            for (double v : smp)
                if (v > 0.0 && v < 1e14) okv.push_back(v);
            if (!okv.empty()) {
                std::nth_element(okv.begin(), okv.begin() + okv.size() / 2, okv.end());
                h->sweep_Rrep = okv[okv.size() / 2];
            } else {
                h->sweep_Rrep = 1.0;
            }
        }
        if (one_launch_model) h->hostm = h->sweepm;
    }
    // 9 <= d <= 16 with every block shared: what the wide-state engine's plan reads (tgp_wide.hip; beyond 16 the dense branch above keeps it)
    h->widem.clear();
    h->wide_state = 0;
    h->wide_post_state = 0;
    {
        const uint32_t need_shared = TGP_SHARED_A | TGP_SHARED_a | TGP_SHARED_Q | TGP_SHARED_H | TGP_SHARED_R;
        h->wide_ht = nullptr;
        if ((flags & need_shared) == need_shared && p == 1 && ordering == 0 && !h->binding_sde && tgp_wide::supports(d)) {
            const size_t dd = (size_t)d * d;
            const bool h_per_step = !(flags & TGP_SHARED_h);      // (a mean function at the inputs: see the dense branch)
            h->widem.assign(2 * dd + 2 * (size_t)d + 2, 0.0);
            double* q = h->widem.data();
            const struct { const double* src; size_t n; } parts[6] = {{A, dd}, {a, (size_t)d}, {Q, dd}, {H, (size_t)d}, {hh, 1}, {R, 1}};
            size_t off = 0;
            for (int k = 0; k < 6; ++k) {
                const auto& pt = parts[k];
                if (k == 4 && h_per_step) { off += 1; continue; }
                if (dev) HIPCHK(hipMemcpy(q + off, pt.src, pt.n * sizeof(double), hipMemcpyDeviceToHost));
                else std::memcpy(q + off, pt.src, pt.n * sizeof(double));
                off += pt.n;
            }
            if (h_per_step) h->wide_ht = h->mv.h;
        }
    }
    h->have_model = true;
    return TGP_OK;
}

// Does exp(F tau) have the closed form of ModelView::sde? The indices split into the connected components of F's sparsity pattern; a
// component S qualifies when F_SS = -lambda I + N with N^min(|S|, 3) = 0 (one eigenvalue, nilpotency <= 3: what Matern-1/2, -3/2 and
// -5/2 terms and their scaled / stretched sums produce; lambda = 0 covers integrated white noise). Rows of one component must be
// contiguous (the loader shares one exponential between neighbouring rows of equal lambda, so equal lambdas of DIFFERENT components
// are fine too: the exponential is the same number). coef <- [lambda per row | N | N^2 / 2 | Pinf (symmetrised) | A1 | Q1].
static bool sde_closed_form(int d, const double* F, const double* Pinf, const double* A1, const double* Q1, std::vector<double>& coef) {
    std::vector<int> comp(d);
    for (int i = 0; i < d; ++i) comp[i] = i;
    auto find = [&](int i) { while (comp[i] != i) i = comp[i] = comp[comp[i]]; return i; };
    for (int j = 0; j < d; ++j)
        for (int i = 0; i < d; ++i)
            if (i != j && F[i + j * d] != 0.0) comp[find(i)] = find(j);
    for (int i = 0; i < d * d; ++i)
        if (!std::isfinite(F[i])) return false;
    std::vector<double> lam(d, 0.0), N((size_t)d * d, 0.0), N2((size_t)d * d, 0.0);
    std::vector<char> seen(d, 0);
    for (int r = 0; r < d; ++r) {
        const int root = find(r);
        if (seen[root]) continue;
        seen[root] = 1;
        std::vector<int> S;
        for (int i = 0; i < d; ++i)
            if (find(i) == root) S.push_back(i);
        const int n = (int)S.size();
        if (n > 3 || S.back() - S.front() != n - 1) return false;
        double tr = 0.0;
        for (int i : S) tr += F[i + i * d];
        const double l = -tr / n;
        std::vector<double> Nb((size_t)n * n), Nb2((size_t)n * n, 0.0), Nb3((size_t)n * n, 0.0), Ab((size_t)n * n), Ab2((size_t)n * n, 0.0), Ab3((size_t)n * n, 0.0);
        for (int jj = 0; jj < n; ++jj)
            for (int ii = 0; ii < n; ++ii) {
                Nb[ii + jj * n] = F[S[ii] + S[jj] * d] + (ii == jj ? l : 0.0);
                Ab[ii + jj * n] = std::fabs(F[S[ii] + S[jj] * d]) + (ii == jj ? std::fabs(l) : 0.0);
            }
        for (int jj = 0; jj < n; ++jj)
            for (int ii = 0; ii < n; ++ii)
                for (int k = 0; k < n; ++k) {
                    Nb2[ii + jj * n] += Nb[ii + k * n] * Nb[k + jj * n];
                    Ab2[ii + jj * n] += Ab[ii + k * n] * Ab[k + jj * n];
                }
        for (int jj = 0; jj < n; ++jj)
            for (int ii = 0; ii < n; ++ii)
                for (int k = 0; k < n; ++k) {
                    Nb3[ii + jj * n] += Nb2[ii + k * n] * Nb[k + jj * n];
                    Ab3[ii + jj * n] += Ab2[ii + k * n] * Ab[k + jj * n];
                }
        // N^n must vanish: to rounding, measured against the size of the terms that cancel in it (the same product of absolute values)
        const std::vector<double>& top = n == 1 ? Nb : (n == 2 ? Nb2 : Nb3);
        const std::vector<double>& mag = n == 1 ? Ab : (n == 2 ? Ab2 : Ab3);
        double mmax = 0.0;
        for (double v : mag) mmax = std::max(mmax, v);
        for (double v : top)
            if (!(std::fabs(v) <= 1e-13 * mmax)) return false;
        for (int ii = 0; ii < n; ++ii) lam[S[ii]] = l;
        for (int jj = 0; jj < n; ++jj)
            for (int ii = 0; ii < n; ++ii) {
                N[S[ii] + S[jj] * d] = n >= 2 ? Nb[ii + jj * n] : 0.0;
                N2[S[ii] + S[jj] * d] = n >= 3 ? 0.5 * Nb2[ii + jj * n] : 0.0;
            }
    }
    coef.assign((size_t)d + 5 * (size_t)d * d, 0.0);
    double* q = coef.data();
    for (int i = 0; i < d; ++i) q[i] = lam[i];
    for (int i = 0; i < d * d; ++i) { q[d + i] = N[i]; q[d + d * d + i] = N2[i]; }
    for (int j = 0; j < d; ++j)
        for (int i = 0; i < d; ++i) q[d + 2 * d * d + i + j * d] = 0.5 * (Pinf[i + j * d] + Pinf[j + i * d]);
    if (A1 && Q1)
        for (int i = 0; i < d * d; ++i) { q[d + 3 * d * d + i] = A1[i]; q[d + 4 * d * d + i] = Q1[i]; }
    return true;
}

int tgp_model_set_sde(tgp_handle* h, int64_t T, int d, int ordering, uint32_t flags, const double* F, const double* a, const double* H,
                      const double* hh, const double* R, const double* times, const double* A1, const double* Q1, const double* x0m,
                      const double* x0P) {
    if (h) drop_graphs(h);
    if (!h) return TGP_EINVAL;
    if (!F || !times) return h->fail(TGP_EINVAL, "null F / times");
    if (h && h->tab_state == 1) (void)hipStreamSynchronize(h->side_stream);
    if (h) { h->tab_state = 0; h->tab_calls = 0; h->tab_L0 = 0; }
    if (d > 8) return h->fail(TGP_EUNSUPPORTED, "tgp_model_set_sde: d <= 8 (build the per-step blocks on the host for larger d)");
    // (time stamps in non-decreasing order: a negative gap would make exp(F dt) expansive and Q indefinite in the tiled record, and is the
    //  marker of the first transition in the closed-form record)
    for (int64_t k = 1; k < T; ++k)
        if (!(times[k] >= times[k - 1])) return h->fail(TGP_EINVAL, "tgp_model_set_sde: the time stamps must be non-decreasing (and finite)");
    // shared placeholder blocks for A and Q (never read: the tiled record supplies them); a must be shared
    if (!(flags & TGP_SHARED_a)) return h->fail(TGP_EUNSUPPORTED, "tgp_model_set_sde: the transition offset a must be shared");
    std::vector<double> zero((size_t)d * d, 0.0);
    const bool dev = (flags & TGP_DEVICE_PTRS) != 0;
    if (dev) return h->fail(TGP_EUNSUPPORTED, "tgp_model_set_sde: model blocks are host pointers (times may be a device pointer via TGP_IN_DEVICE semantics is not offered)");
    // (the kernel table must be chosen by the verdict of the PER-STEP family of the known-answer check: until round 3 the placeholder flags made
    //  tgp_model_set pick the LTI family's, and a d = 8 model with per-step noise ran inlined kernels that had failed the per-step check)
    h->binding_sde = true;
    int rc = tgp_model_set(h, T, d, 1, ordering, flags | TGP_SHARED_A | TGP_SHARED_Q, zero.data(), a, zero.data(), H, hh, R, x0m, x0P);
    h->binding_sde = false;
    if (rc != TGP_OK) return rc;
    HIPCHK(h->bF.ensure((size_t)d * d * sizeof(double)));
    HIPCHK(h->bPinf.ensure((size_t)d * d * sizeof(double)));
    HIPCHK(h->btimes.ensure((size_t)T * sizeof(double)));
    HIPCHK(hipMemcpyAsync(h->bF.p, F, (size_t)d * d * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->bPinf.p, x0P, (size_t)d * d * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->btimes.p, times, (size_t)T * sizeof(double), hipMemcpyHostToDevice, h->stream));
    h->have_AQ1 = (A1 != nullptr && Q1 != nullptr);
    if (h->have_AQ1) {
        HIPCHK(h->bAQ1.ensure((size_t)2 * d * d * sizeof(double)));
        HIPCHK(hipMemcpyAsync(h->bAQ1.p, A1, (size_t)d * d * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->bAQ1.d() + (size_t)d * d, Q1, (size_t)d * d * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    h->times_dev = h->btimes.d();
    double nrm = 0.0;     // 1-norm of F
    for (int j = 0; j < d; ++j) {
        double cs = 0.0;
        for (int i = 0; i < d; ++i) cs += std::fabs(F[i + j * d]);
        nrm = cs > nrm ? cs : nrm;
    }
    h->normF = nrm;
    h->sde = true;
    h->sde_closed = false;
    h->tile_is_dt = false;
    if (d <= kSdeInKernelMaxD) {
        std::vector<double> coef;
        if (sde_closed_form(d, F, x0P, h->have_AQ1 ? A1 : nullptr, h->have_AQ1 ? Q1 : nullptr, coef)) {
            HIPCHK(h->bsde.ensure(coef.size() * sizeof(double)));
            HIPCHK(hipMemcpyAsync(h->bsde.p, coef.data(), coef.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            h->sde_closed = true;
        }
    }
    h->lti = false;        // transitions are per-step (tiled), whatever the emission flags say
    h->tile_L0 = 0;
    // what the sweep engine reads (tgp_sweep.hpp): the gaps as one plain array, the closed-form coefficients and the first transition on the host
    h->sde_coef_host.clear();
    h->sde_A1Q1_host.clear();
    if (h->sde_closed && d <= tgp_sweep::kMaxD) {
        std::vector<double> coef;
        (void)sde_closed_form(d, F, x0P, nullptr, nullptr, coef);
        h->sde_coef_host = coef;
        std::vector<double> tau((size_t)T);
        tau[0] = -1.0;
        for (int64_t k = 1; k < T; ++k) tau[(size_t)k] = times[k] - times[k - 1];
        HIPCHK(h->btau.ensure((size_t)T * sizeof(double)));
        HIPCHK(hipMemcpyAsync(h->btau.p, tau.data(), (size_t)T * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        {
            const size_t ns = (size_t)std::min<int64_t>(T - 1, 4096);
            std::vector<double> smp(tau.begin() + 1, tau.begin() + 1 + ns);
            h->sweep_tau = 1.0;
            if (!smp.empty()) {
                std::nth_element(smp.begin(), smp.begin() + smp.size() / 2, smp.end());
                h->sweep_tau = smp[smp.size() / 2];
            }
        }
        // the first transition: as given, else the reference's dt_1 := 1 (lti_sde.jl:139) by the same closed form
        h->sde_A1Q1_host.assign((size_t)2 * d * d, 0.0);
        double* A1h = h->sde_A1Q1_host.data();
        double* Q1h = A1h + (size_t)d * d;
        if (h->have_AQ1) {
            std::memcpy(A1h, A1, (size_t)d * d * sizeof(double));
            std::memcpy(Q1h, Q1, (size_t)d * d * sizeof(double));
        } else {
            const double* q = coef.data();
            for (int j = 0; j < d; ++j)
                for (int i = 0; i < d; ++i) A1h[i + j * d] = std::exp(-q[i]) * (q[d + d * d + i + j * d] + q[d + i + j * d] + (i == j ? 1.0 : 0.0));
            for (int j = 0; j < d; ++j)
                for (int i = 0; i < d; ++i) {
                    double acc = 0.0;
                    for (int k = 0; k < d; ++k)
                        for (int l = 0; l < d; ++l) acc += A1h[i + k * d] * 0.5 * (x0P[k + l * d] + x0P[l + k * d]) * A1h[j + l * d];
                    Q1h[i + j * d] = 0.5 * (x0P[i + j * d] + x0P[j + i * d]) - acc;
                }
        }
    }
    return TGP_OK;
}

int tgp_model_set_x0(tgp_handle* h, const double* x0m, const double* x0P) {
    if (h) drop_graphs(h);
    StreamGuard stream_guard_(h);
    TRY(check_ready(h));
    if (!x0m || !x0P) return h->fail(TGP_EINVAL, "null x0");
    h->x0m.assign(x0m, x0m + h->d);
    h->x0P.assign(x0P, x0P + (size_t)h->d * h->d);
    h->steady_known = false;
    h->modal_state = 0;
    h->smooth_state = 0;
    if (h->is_dense) return dense_fail(h, tgp_dense::set_x0(h->dense, x0m, x0P, h->stream));
    h->fold_valid = false;
    h->smoother_valid = false;
    return upload_x0(h, h->bx0, x0m, x0P);
}

#include "tgp_api_lti_front.inc"      // Reverse-ordered LTI priors by flipping the series, wide LTI models (tgp_wide.hip), logpdf of an LTI model on the one-launch kernels
int tgp_logpdf(tgp_handle* h, const double* y, const uint8_t* missing, uint32_t flags, double* out) {
    StreamGuard stream_guard_(h);
    TRY(check_ready(h, /*general=*/false));
    if (!out) return h->fail(TGP_EINVAL, "out is NULL");
    h->steady2_last = false;
    h->modal_last = false;
    h->dense_last_n0 = -1;
    if (reverse_by_flip(h, y, missing, flags)) {
        const double* yf = nullptr;
        uint32_t ff = flags;
        TRY(flip_series(h, y, flags, &yf, &ff));
        bool served = false;
        {
            AsForward as_forward(h);
            TRY(logpdf_lti_one_launch(h, yf, ff, out, &served));
        }
        if (served) return TGP_OK;
    }
    if (missing == nullptr) {
        bool served = false;
        TRY(logpdf_lti_one_launch(h, y, flags, out, &served));
        if (served) return TGP_OK;
    }
    if (steady2_eligible(h, missing, flags)) {
        CallTimer tm(h, /*clear=*/false);      // (the engine's set-up kernel clears the result record itself: one launch less)
        TRY(set_obs(h, y, missing, flags));
        tm.inputs_done();
        TRY(steady2_enqueue(h, nullptr, false, nullptr, nullptr));
        tm.kernels_done();
        const int rc = tm.finish(out);
        if (rc != TGP_OK || steady2_served(h)) return rc;
        // not applicable to this model (decided on the device): the general path below serves this and every later call
    }
    if (lti_but_offset(h) && missing == nullptr && y != nullptr && !(flags & TGP_REUSE_REDUCE)) {      // a mean function on a regular grid: stationary gains
        bool served = false;
        TRY(smooth_lti_call(h, y, flags, nullptr, nullptr, nullptr, out, &served));
        if (served) return TGP_OK;
    }
    h->sweep_last = false;
    if (sweep_eligible(h, flags)) {
        bool served = false;
        TRY(sweep_call(h, y, missing, flags, nullptr, nullptr, nullptr, out, &served));
        if (served) return TGP_OK;
    }
    if (!h->widem.empty() && missing == nullptr && h->opt_chunk == 0 && h->variant_opt == 0 && h->opt_group != 2) {      // wide LTI models (8 < d <= 63): the stationary closed loop across the chip (tgp_wide.hip)
        bool served = false;
        TRY(wide_call(h, y, flags, nullptr, nullptr, nullptr, out, &served));
        if (served) return TGP_OK;
    }
    resolve_table(h);
    if (graph_eligible(h, flags, false)) {
        const uint64_t key[8] = {1, (uint64_t)(uintptr_t)y, (uint64_t)(uintptr_t)missing, flags, 0, 0, 0, 0};
        return graph_call(h, 0, key, [&]() -> int {
            CallTimer tm(h);
            TRY(set_obs(h, y, missing, flags));
            TRY(forward_reduce(h, flags, 0));
            FilterOut fo{};
            TRY(forward_apply(h, 0, fo));
            return TGP_OK;
        }, out);
    }
    CallTimer tm(h);
    TRY(set_obs(h, y, missing, flags));
    tm.inputs_done();
    if (h->is_dense) {
        tgp_dense::set_profile(h->dense, h->profile);
        TRY(dense_fail(h, tgp_dense::filter(h->dense, h->mv.y, h->mv.missing, nullptr, nullptr, h->result.d(), h->stream)));
        tm.kernels_done();
        return tm.finish(out);
    }
    TRY(forward_reduce(h, flags, 0));
    FilterOut fo{};
    TRY(forward_apply(h, 0, fo));
    tm.kernels_done();
    return tm.finish(out);
}

// logpdf of the bound model with ANOTHER noise variance (everything else as bound): what the joint model of DESIGN 3.18 is -- the prior's blocks
// with Rbar -- without binding a second model.  The one-launch paths only (their plans are host functions of the blocks, rebuilt per call):
// Forward LTI models, scalar observations, no missing data; TGP_EUNSUPPORTED otherwise (bind the model with the new variance instead).
int tgp_logpdf_noise(tgp_handle* h, const double* y, uint32_t flags, double R, double* out) {
    StreamGuard stream_guard_(h);
    TRY(check_ready(h, /*general=*/false));
    if (!out || !y) return h->fail(TGP_EINVAL, "tgp_logpdf_noise: null argument");
    if (!(R > 0.0) || !std::isfinite(R)) return h->fail(TGP_EINVAL, "tgp_logpdf_noise: the noise variance must be positive");
    h->steady2_last = false;
    h->modal_last = false;
    h->dense_last_n0 = -1;
    if (!steady2_eligible(h, nullptr, flags) || !h->opt_modal || h->hostm.empty())
        return h->fail(TGP_EUNSUPPORTED, "tgp_logpdf_noise: Forward LTI models with scalar observations on the one-launch paths only");
    const int modal_state = h->modal_state, smooth_state = h->smooth_state;       // (the verdicts are the BOUND model's: kept for it)
    h->modal_state = 0;
    h->smooth_state = 0;
    h->R_over = R;
    h->has_R_over = true;
    bool served = false;
    int rc = modal_call(h, y, flags, nullptr, nullptr, nullptr, out, &served);
    if (rc == TGP_OK && !served) rc = smooth_lti_call(h, y, flags, nullptr, nullptr, nullptr, out, &served);
    if (rc == TGP_OK && !served) rc = filter_lti_call(h, y, flags, nullptr, nullptr, out, &served);
    h->has_R_over = false;
    h->modal_state = modal_state;
    h->smooth_state = smooth_state;
    if (rc != TGP_OK) return rc;
    if (!served) return h->fail(TGP_EUNSUPPORTED, "tgp_logpdf_noise: no one-launch path takes this model with this noise variance");
    return TGP_OK;
}

#include "tgp_api_segments.inc"      // time segments on the one-launch path, the plan and adjoint entry points without a handle
#include "tgp_api_lti_calls.inc"      // the LTI one-launch calls behind the entry points: adjoint, filter, smoother / posterior draw, pair statistic
int tgp_filter(tgp_handle* h, const double* y, const uint8_t* missing, uint32_t flags, double* m_out, double* P_out, double* lml_out) {
    StreamGuard stream_guard_(h);
    TRY(check_ready(h, /*general=*/false));
    h->dense_last_n0 = -1;
    h->modal_last = false;
    h->steady2_last = false;
    if (reverse_by_flip(h, y, missing, flags) && m_out && P_out) {
        // the Forward twin on the flipped series into device scratch, rows flipped back into the caller's arrays
        const bool odev_r = (flags & TGP_OUT_DEVICE) != 0;
        const size_t nm_r = (size_t)h->T * h->d * sizeof(double), nP_r = nm_r * h->d;
        const double* yf = nullptr;
        uint32_t ff = flags;
        TRY(flip_series(h, y, flags, &yf, &ff));
        HIPCHK(h->bflip_m.ensure(nm_r));
        HIPCHK(h->bflip_P.ensure(nP_r));
        bool served = false;
        {
            AsForward as_forward(h);
            TRY(filter_lti_call(h, yf, ff | TGP_OUT_DEVICE, h->bflip_m.d(), h->bflip_P.d(), lml_out, &served));
        }
        if (served) {
            double *dm = nullptr, *dP = nullptr;
            TRY(stage_out(h, h->bo1, m_out, nm_r, odev_r, &dm));
            TRY(stage_out(h, h->bo2, P_out, nP_r, odev_r, &dP));
            launch_flip(h, h->bflip_m.d(), dm, h->T, h->d);
            launch_flip(h, h->bflip_P.d(), dP, h->T, h->d * h->d);
            TRY(copy_back(h, m_out, dm, nm_r, odev_r));
            TRY(copy_back(h, P_out, dP, nP_r, odev_r));
            HIPCHK(hipStreamSynchronize(h->stream));
            resolve_profile(h);
            return TGP_OK;
        }
    }
    if (missing == nullptr && y != nullptr) {
        bool served = false;
        TRY(filter_lti_call(h, y, flags, m_out, P_out, lml_out, &served));
        if (served) return TGP_OK;
    }
    if (!h->widem.empty() && missing == nullptr && h->p == 1 && h->opt_chunk == 0 && h->variant_opt == 0 && h->opt_group != 2) {      // wide LTI models: tgp_wide.hip
        bool served = false;
        TRY(wide_filter_call(h, y, flags, m_out, P_out, lml_out, &served));
        if (served) return TGP_OK;
    }
    resolve_table(h);
    const bool odev = (flags & TGP_OUT_DEVICE) != 0;
    const size_t nm = (size_t)h->T * h->d * sizeof(double), nP = nm * h->d;
    CallTimer tm(h);
    TRY(set_obs(h, y, missing, flags));
    tm.inputs_done();
    if (h->is_dense) {
        double *dm = nullptr, *dP = nullptr;
        TRY(stage_out(h, h->bo1, m_out, nm, odev, &dm));
        TRY(stage_out(h, h->bo2, P_out, nP, odev, &dP));
        tgp_dense::set_profile(h->dense, h->profile);
        TRY(dense_fail(h, tgp_dense::filter(h->dense, h->mv.y, h->mv.missing, dm, dP, h->result.d(), h->stream)));
        tm.kernels_done();
        TRY(copy_back(h, m_out, dm, nm, odev));
        TRY(copy_back(h, P_out, dP, nP, odev));
        return tm.finish(lml_out);
    }
    TRY(forward_reduce(h, flags, 1));
    FilterOut fo{};
    TRY(stage_out(h, h->bo1, m_out, nm, odev, &fo.m_out));
    TRY(stage_out(h, h->bo2, P_out, nP, odev, &fo.P_out));
    TRY(forward_apply(h, 1, fo));
    tm.kernels_done();
    TRY(copy_back(h, m_out, fo.m_out, nm, odev));
    TRY(copy_back(h, P_out, fo.P_out, nP, odev));
    return tm.finish(lml_out);
}

// posterior(model, y) of a Forward LTI model (lgssm.jl:193-221; scalar observations, one noise variance, no missing data, d <= 6): the
// reverse-time transitions of the head on the host, everything behind it -- two constant fills and g_(t+1) = m_t - G mu_(t+1) -- by the
// filter's ONE kernel (tgp_modal.hpp: PosteriorOut).
static int posterior_lti_call(tgp_handle* h, const double* y, uint32_t flags, double* G, double* g, double* L, double* xfm, double* xfP, bool* served) {
    *served = false;
    tgp_plan::ModelHost mh;
    if (!h->opt_modal || chunk_engine_requested(h) || h->is_dense || !h->lti || h->p != 1 || h->ordering != 0 || h->sde || h->d > tgp_plan::kRandMaxD || !modal_host_model(h, mh)) return TGP_OK;
    tgp_plan::FilterPlan fp;
    tgp_modal::plan_filter(mh, h->T, fp);
    if (fp.why != tgp_plan::kOk) return TGP_OK;
    const int d = h->d;
    const size_t dd = (size_t)d * d, nhs = (size_t)fp.nhs;
    const long long nwg = tgp_modal::filter_workgroups(fp, h->T);
    const size_t need = nhs * (1 + 2 * dd) + (nhs + 1) * d + 2 * dd + (size_t)nwg + d + 8;
    TRY(ensure_pinned(h, need));
    double *yh = h->flt_host, *Gh = yh + nhs, *Lh = Gh + nhs * dd, *gh = Lh + nhs * dd, *Gss = gh + (nhs + 1) * d, *Lss = Gss + dd, *fin = Lss + dd,
           *part = fin + d;
    const bool odev = (flags & TGP_OUT_DEVICE) != 0;
    const size_t ng = (size_t)h->T * d * sizeof(double), nG = ng * d;
    CallTimer tm(h, /*clear=*/false);
    TRY(set_obs(h, y, nullptr, flags));
    tm.inputs_done();
    HIPCHK(hipMemcpyAsync(yh, h->mv.y, nhs * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    double mu_end[tgp_plan::kRandMaxD], quad_head = 0.0;
    if (!tgp_modal::plan_posterior_head(mh, fp, yh, Gh, gh, Lh, Gss, Lss, mu_end, &quad_head)) return TGP_OK;      // (the general engine reports it)
    double *dG = nullptr, *dg = nullptr, *dL = nullptr;
    TRY(stage_out(h, h->bo1, G, nG, odev, &dG));
    TRY(stage_out(h, h->bo2, g, ng, odev, &dg));
    TRY(stage_out(h, h->bo3, L, nG, odev, &dL));
    HIPCHK(hipMemcpyAsync(dG, Gh, nhs * dd * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dL, Lh, nhs * dd * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dg, gh, (nhs + 1) * d * sizeof(double), hipMemcpyHostToDevice, h->stream));
    {
        LaunchScope ls(h, "k_filter_one");
        const tgp_modal::PosteriorOut po{Gss, Lss, dG, dg, dL, fin};
        const int rc = tgp_modal::filter_lti(h->stream, fp, mu_end, h->mv.y, h->T, nullptr, nullptr, part, &po);
        if (rc != 0) return h->fail(TGP_EHIP, std::string("tgp_posterior: launch: ") + hipGetErrorString((hipError_t)rc));
    }
    tm.kernels_done();
    TRY(copy_back(h, G, dG, nG, odev));
    TRY(copy_back(h, g, dg, ng, odev));
    TRY(copy_back(h, L, dL, nG, odev));
    if (h->timing) (void)hipEventRecord(h->ev[3], h->stream);
    HIPCHK(hipStreamSynchronize(h->stream));
    resolve_profile(h);
    double ssq = 0.0;
    for (long long w = 0; w < nwg; ++w) ssq += part[w];
    const double lml = -0.5 * ((double)h->T * 1.8378770664093454835606594728112 + fp.LS + (double)(h->T - fp.n0) * fp.logS + quad_head + fp.iS * ssq);
    for (int i = 0; i < 8; ++i) h->host_result[i] = 0.0;
    h->host_result[0] = lml;
    if (xfm)
        for (int i = 0; i < d; ++i) xfm[i] = fin[i];
    if (xfP)
        for (size_t e = 0; e < dd; ++e) xfP[e] = fp.Pss[e];      // (symmetric: either storage order)
    h->reduce_valid = false;
    h->smoother_valid = false;
    h->modal_last = false;
    h->steady2_last = false;
    h->dense_last_n0 = fp.n0;
    *served = true;
    return TGP_OK;
}

int tgp_posterior(tgp_handle* h, const double* y, const uint8_t* missing, uint32_t flags, double* G, double* g, double* L,
                  double* xfm, double* xfP) {
    StreamGuard stream_guard_(h);
    TRY(check_ready(h, /*general=*/false));
    if ((G || g || L) && !(G && g && L)) return h->fail(TGP_EINVAL, "G, g, L must be given together");
    h->dense_last_n0 = -1;
    h->modal_last = false;
    h->steady2_last = false;
    if (G && missing == nullptr && y != nullptr) {
        bool served = false;
        TRY(posterior_lti_call(h, y, flags, G, g, L, xfm, xfP, &served));
        if (served) return TGP_OK;
    }
    resolve_table(h);
    if (h->is_dense) {
        const bool odev_d = (flags & TGP_OUT_DEVICE) != 0;
        const size_t ng_d = (size_t)h->T * h->d * sizeof(double), nG_d = ng_d * h->d;
        CallTimer tm(h);
        TRY(set_obs(h, y, missing, flags));
        tm.inputs_done();
        double *dG = nullptr, *dg = nullptr, *dL = nullptr;
        TRY(stage_out(h, h->bo1, G, nG_d, odev_d, &dG));
        TRY(stage_out(h, h->bo2, g, ng_d, odev_d, &dg));
        TRY(stage_out(h, h->bo3, L, nG_d, odev_d, &dL));
        tgp_dense::set_profile(h->dense, h->profile);
        TRY(dense_fail(h, tgp_dense::posterior(h->dense, h->mv.y, h->mv.missing, dG, dg, dL, xfm, xfP, h->result.d(), h->stream)));
        tm.kernels_done();
        TRY(copy_back(h, G, dG, nG_d, odev_d));
        TRY(copy_back(h, g, dg, ng_d, odev_d));
        TRY(copy_back(h, L, dL, nG_d, odev_d));
        return tm.finish();
    }
    // Reverse-ordered prior (step_posterior(::Reverse), lgssm.jl:223-228): the lane-per-chunk materialise pass only
    if (h->ordering != 0 && !G) return h->fail(TGP_EUNSUPPORTED, "posterior of a Reverse-ordered model: G, g, L must be requested");
    const bool odev = (flags & TGP_OUT_DEVICE) != 0;
    const size_t ng = (size_t)h->T * h->d * sizeof(double), nG = ng * h->d;
    CallTimer tm(h);
    TRY(set_obs(h, y, missing, flags));
    tm.inputs_done();
    TRY(forward_reduce(h, flags, (G != nullptr) ? 3 : 0));
    FilterOut fo{};
    TRY(stage_out(h, h->bo1, G, nG, odev, &fo.G_out));
    TRY(stage_out(h, h->bo2, g, ng, odev, &fo.g_out));
    TRY(stage_out(h, h->bo3, L, nG, odev, &fo.L_out));
    if (h->ordering != 0) {
        HIPCHK(h->bx0r.ensure((size_t)state_size(h->d) * sizeof(double)));
        fo.xfin = h->bx0r.d();
    }
    TRY(forward_apply(h, (G != nullptr) ? 3 : 0, fo));
    tm.kernels_done();
    TRY(copy_back(h, G, fo.G_out, nG, odev));
    TRY(copy_back(h, g, fo.g_out, ng, odev));
    TRY(copy_back(h, L, fo.L_out, nG, odev));
    if (xfm && xfP) {
        std::vector<double> pk(state_size(h->d));
        // Forward prior: the final filtering state; Reverse prior: the state after the last step's predict (written by pass 2)
        HIPCHK(hipMemcpyAsync(pk.data(), h->ordering != 0 ? fo.xfin : h->F.fin, pk.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        unpack_state(h->d, pk.data(), xfm, xfP);
    }
    return tm.finish();
}

static int smoother_forward_impl(tgp_handle* h, uint32_t flags, const double* x0dev = nullptr, bool allow_group = false) {
    TRY(forward_reduce(h, flags, allow_group ? 2 : -1));
    const size_t fsz = (size_t)((h->n0 + 63) / 64) * 64 * h->L0 * state_size(h->d) * sizeof(double);
    HIPCHK(h->fs.ensure(fsz));
    FilterOut fo{};
    fo.fs = h->fs.d();
    // d >= 5, lane-per-chunk passes: MODE 2 split into filter + scratch (MODE 4) and a kernel that composes the chunk smoother
    // elements from the scratch -- each half spills less than the fused kernel (d = 6, T = 1e7: 2.87 ms fused)
    // (measured, T = 1e7: d = 5 0.92 fused / 0.50 + 0.53 split; d = 6 2.87 / 1.01 + 0.85; d = 7 6.3 / 1.77 + 1.87: split from d = 6;
    // TGP_OPT_SPLIT_SMOOTHER = 2 forces it for d = 5 too; d >= 8 runs the group smoother, and the d = 8 compose kernel
    // faults (memory aperture violation in the start-up self-test), so it is never launched)
    const bool split = !h->group_active && h->opt_split && ((h->d >= 6 && h->d <= 7) || (h->opt_split == 2 && h->d == 5)) && h->kt->apply_filter_m[4] != nullptr &&
                       h->kt->compose_smoother != nullptr;
    TRY(forward_apply(h, split ? 4 : 2, fo, x0dev));
    if (split) {
        LaunchScope ls(h, h->lti ? "k_compose_smoother<lti>" : "k_compose_smoother<per-step>");
        h->kt->compose_smoother(h->lti, h->mv, h->L0, h->n0, h->F.S[0], h->fs.d(), h->Rv.E[0], flag_ptr(h), h->stream);
    }
    scan_up(h, h->Rv);
    h->smoother_valid = true;
    return TGP_OK;
}

static int smoother_backward_impl(tgp_handle* h, const double* xs_dev, const double* Rnew_dev, int64_t sRn, double* mean_dev, double* var_dev) {
    scan_down(h, h->Rv, xs_dev);
    if (h->group_active) {
        LaunchScope ls(h, h->lti ? "k_group_smooth<lti>" : "k_group_smooth<per-step>");
        h->kt->group_smooth(h->mv, h->L0, h->n0, h->F.S[0], h->Rv.S[0], h->fs.d(), Rnew_dev, sRn, mean_dev, var_dev, flag_ptr(h), h->alt_H, h->alt_h,
                            h->alt_p, h->stream);
        return TGP_OK;
    }
    {
        LaunchScope ls(h, h->lti ? "k_smooth<lti>" : "k_smooth<per-step>");
        h->kt->smooth(h->lti, h->mv, h->L0, h->n0, h->F.S[0], h->Rv.S[0], h->fs.d(), Rnew_dev, sRn, mean_dev, var_dev,
                      flag_ptr(h), h->stream);
    }
    return TGP_OK;
}

int tgp_posterior_marginals(tgp_handle* h, const double* y, const uint8_t* missing, const double* Rnew, uint32_t flags,
                            double* mean_out, double* var_out, double* lml_out) {
    StreamGuard stream_guard_(h);
    TRY(check_ready(h, /*general=*/false));
    if (!Rnew || !mean_out || !var_out) return h->fail(TGP_EINVAL, "null Rnew / output");
    if (h->ordering != 0) return h->fail(TGP_EUNSUPPORTED, "posterior of a Reverse-ordered model is not implemented on the device");
    const bool idev = (flags & TGP_IN_DEVICE) != 0, odev = (flags & TGP_OUT_DEVICE) != 0;
    const bool rshared = (flags & TGP_SHARED_R) != 0;
    const size_t nT = (size_t)h->T * h->p * sizeof(double);   // one value per (time step, observation)
    h->steady2_last = false;
    h->modal_last = false;
    h->dense_last_n0 = -1;
    if (steady2_eligible(h, missing, flags)) {
        bool served = false;
        TRY(modal_call(h, y, flags, Rnew, mean_out, var_out, lml_out, &served));
        if (served) return TGP_OK;
        // no well-conditioned modal form: both recursions on dense powers, still ONE kernel
        if (missing == nullptr && y != nullptr) {
            TRY(smooth_lti_call(h, y, flags, Rnew, mean_out, var_out, lml_out, &served));
            if (served) return TGP_OK;
        }
    }
    if (steady2_eligible(h, missing, flags)) {
        CallTimer tm(h, /*clear=*/false);
        const void* pR = nullptr;
        TRY(stage_in(h, h->bRnew, Rnew, rshared ? sizeof(double) : nT, idev, &pR));
        TRY(set_obs(h, y, missing, flags));
        tm.inputs_done();
        double *dm = nullptr, *dv = nullptr;
        TRY(stage_out(h, h->bo1, mean_out, nT, odev, &dm));
        TRY(stage_out(h, h->bo2, var_out, nT, odev, &dv));
        TRY(steady2_enqueue(h, (const double*)pR, !rshared, dm, dv));
        tm.kernels_done();
        TRY(copy_back(h, mean_out, dm, nT, odev));
        TRY(copy_back(h, var_out, dv, nT, odev));
        const int rc = tm.finish(lml_out);
        if (rc != TGP_OK || steady2_served(h)) {
            h->smoother_valid = false;
            return rc;
        }
    }
    if (lti_but_offset(h) && missing == nullptr && y != nullptr && !(flags & TGP_REUSE_REDUCE)) {      // a mean function on a regular grid: stationary gains
        bool served = false;
        TRY(smooth_lti_call(h, y, flags, Rnew, mean_out, var_out, lml_out, &served));
        if (served) return TGP_OK;
    }
    h->sweep_last = false;
    if (sweep_eligible(h, flags)) {
        bool served = false;
        TRY(sweep_call(h, y, missing, flags, Rnew, mean_out, var_out, lml_out, &served));
        if (served) return TGP_OK;
    }
    if (!h->widem.empty() && missing == nullptr && h->p == 1 && h->opt_chunk == 0 && h->variant_opt == 0 && h->opt_group != 2) {      // wide LTI models: tgp_wide.hip
        bool served = false;
        TRY(wide_call(h, y, flags, Rnew, mean_out, var_out, lml_out, &served));
        if (served) return TGP_OK;
    }
    resolve_table(h);
    if (graph_eligible(h, flags, true)) {
        const uint64_t key[8] = {2, (uint64_t)(uintptr_t)y, (uint64_t)(uintptr_t)missing, flags, (uint64_t)(uintptr_t)Rnew, (uint64_t)(uintptr_t)mean_out,
                                 (uint64_t)(uintptr_t)var_out, 0};
        return graph_call(h, 1, key, [&]() -> int {
            CallTimer tm(h);
            TRY(set_obs(h, y, missing, flags));
            TRY(smoother_forward_impl(h, flags, nullptr, /*allow_group=*/true));
            TRY(smoother_backward_impl(h, h->F.fin, Rnew, rshared ? 0 : 1, mean_out, var_out));
            return TGP_OK;
        }, lml_out);
    }
    CallTimer tm(h);
    const void* pR = nullptr;
    TRY(stage_in(h, h->bRnew, Rnew, rshared ? (size_t)h->p * sizeof(double) : nT, idev, &pR));
    TRY(set_obs(h, y, missing, flags));
    tm.inputs_done();
    if (h->is_dense) {
        double *dm2 = nullptr, *dv2 = nullptr;
        TRY(stage_out(h, h->bo1, mean_out, nT, odev, &dm2));
        TRY(stage_out(h, h->bo2, var_out, nT, odev, &dv2));
        tgp_dense::set_profile(h->dense, h->profile);
        tgp_dense::set_segment(h->dense, h->opt_chunk);     // TGP_OPT_CHUNK: segment length of the checkpointed smoother
        TRY(dense_fail(h, tgp_dense::posterior_marginals(h->dense, h->mv.y, h->mv.missing, (const double*)pR, rshared ? 0 : h->p, dm2, dv2,
                                                         h->result.d(), h->stream)));
        tm.kernels_done();
        TRY(copy_back(h, mean_out, dm2, nT, odev));
        TRY(copy_back(h, var_out, dv2, nT, odev));
        return tm.finish(lml_out);
    }
    TRY(smoother_forward_impl(h, flags, nullptr, /*allow_group=*/true));
    double *dm = nullptr, *dv = nullptr;
    TRY(stage_out(h, h->bo1, mean_out, nT, odev, &dm));
    TRY(stage_out(h, h->bo2, var_out, nT, odev, &dv));
    TRY(smoother_backward_impl(h, h->F.fin, (const double*)pR, rshared ? 0 : 1, dm, dv));
    tm.kernels_done();
    TRY(copy_back(h, mean_out, dm, nT, odev));
    TRY(copy_back(h, var_out, dv, nT, odev));
    return tm.finish(lml_out);
}

int tgp_logpdf_and_posterior_marginals(tgp_handle* h, const double* y, const uint8_t* missing, const double* Rnew, uint32_t flags,
                                       double* lml_out, double* mean_out, double* var_out) {
    if (h && !lml_out) return h->fail(TGP_EINVAL, "lml_out is NULL");
    return tgp_posterior_marginals(h, y, missing, Rnew, flags, mean_out, var_out, lml_out);
}

int tgp_posterior_marginals_at(tgp_handle* h, const double* y, const uint8_t* missing, int pn, const double* Hn, const double* hn,
                               const double* Rn, uint32_t flags, double* mean_out, double* var_out, double* lml_out) {
    StreamGuard stream_guard_(h);
    TRY(check_ready(h));
    TRY(scan_only(h, "tgp_posterior_marginals_at"));
    if (pn < 1 || pn > 4096 || !Hn || !hn || !Rn || !mean_out || !var_out) return h->fail(TGP_EINVAL, "bad alternative emission block / output");
    if (h->ordering != 0 || !h->lti || !h->use_group_sm || !h->opt_group || h->kt->group_smooth == nullptr)
        return h->fail(TGP_EUNSUPPORTED, "tgp_posterior_marginals_at: Forward LTI models with the group-per-chunk smoother (d = 5..16) only");
    const bool idev = (flags & TGP_IN_DEVICE) != 0, odev = (flags & TGP_OUT_DEVICE) != 0;
    const bool rshared = (flags & TGP_SHARED_R) != 0;
    const size_t nOut = (size_t)h->T * pn * sizeof(double);
    CallTimer tm(h);
    // Hn | hn always from the host (small); Rn as the flags say
    const size_t nH = (size_t)pn * h->d, nAll = nH + pn;
    std::vector<double> blk(nAll);
    std::memcpy(blk.data(), Hn, nH * sizeof(double));
    std::memcpy(blk.data() + nH, hn, (size_t)pn * sizeof(double));
    HIPCHK(h->balt.ensure(nAll * sizeof(double)));
    HIPCHK(hipMemcpyAsync(h->balt.p, blk.data(), nAll * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const void* pR = nullptr;
    TRY(stage_in(h, h->bRnew, Rn, rshared ? (size_t)pn * sizeof(double) : nOut, idev, &pR));
    TRY(set_obs(h, y, missing, flags));
    tm.inputs_done();
    h->force_group_post = 1;
    int rc = smoother_forward_impl(h, flags, nullptr, /*allow_group=*/true);
    h->force_group_post = 0;
    if (rc != TGP_OK) return rc;
    if (!h->group_active) return h->fail(TGP_EUNSUPPORTED, "tgp_posterior_marginals_at: the group-per-chunk smoother is not available for this model");
    double *dm = nullptr, *dv = nullptr;
    TRY(stage_out(h, h->bo1, mean_out, nOut, odev, &dm));
    TRY(stage_out(h, h->bo2, var_out, nOut, odev, &dv));
    h->alt_H = h->balt.d();
    h->alt_h = h->balt.d() + nH;
    h->alt_p = pn;
    rc = smoother_backward_impl(h, h->F.fin, (const double*)pR, rshared ? 0 : 1, dm, dv);
    h->alt_H = h->alt_h = nullptr;
    h->alt_p = 0;
    if (rc != TGP_OK) return rc;
    tm.kernels_done();
    TRY(copy_back(h, mean_out, dm, nOut, odev));
    TRY(copy_back(h, var_out, dv, nOut, odev));
    return tm.finish(lml_out);
}

int tgp_smoother_forward(tgp_handle* h, const double* y, const uint8_t* missing, uint32_t flags, double* rev_elem_out, double* xfm,
                         double* xfP, double* lml_out) {
    StreamGuard stream_guard_(h);
    TRY(check_ready(h));
    TRY(scan_only(h, "tgp_smoother_forward"));
    if (h->ordering != 0) return h->fail(TGP_EUNSUPPORTED, "posterior of a Reverse-ordered model is not implemented on the device");
    CallTimer tm(h);
    TRY(set_obs(h, y, missing, flags));
    tm.inputs_done();
    TRY(smoother_forward_impl(h, flags));
    if (rev_elem_out) TRY(scan_total_to_host(h, h->Rv, rev_elem_out));
    tm.kernels_done();
    if (xfm && xfP) {
        std::vector<double> pk(state_size(h->d));
        HIPCHK(hipMemcpyAsync(pk.data(), h->F.fin, pk.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        unpack_state(h->d, pk.data(), xfm, xfP);
    }
    return tm.finish(lml_out);
}

int tgp_smoother_backward(tgp_handle* h, const double* xs_m, const double* xs_P, const double* Rnew, uint32_t flags, double* mean_out,
                          double* var_out) {
    StreamGuard stream_guard_(h);
    TRY(check_ready(h));
    TRY(scan_only(h, "tgp_smoother_backward"));
    if (!h->smoother_valid) return h->fail(TGP_EINVAL, "tgp_smoother_backward needs a preceding tgp_smoother_forward");
    if (!Rnew || !mean_out || !var_out) return h->fail(TGP_EINVAL, "null Rnew / output");
    const bool idev = (flags & TGP_IN_DEVICE) != 0, odev = (flags & TGP_OUT_DEVICE) != 0;
    const bool rshared = (flags & TGP_SHARED_R) != 0;
    const size_t nT = (size_t)h->T * h->p * sizeof(double);   // one value per (time step, observation)
    CallTimer tm(h);
    const void* pR = nullptr;
    TRY(stage_in(h, h->bRnew, Rnew, rshared ? (size_t)h->p * sizeof(double) : nT, idev, &pR));
    const double* xs_dev = h->F.fin;
    if (xs_m && xs_P) {
        TRY(upload_x0(h, h->bx0r, xs_m, xs_P));
        xs_dev = h->bx0r.d();
    }
    tm.inputs_done();
    double *dm = nullptr, *dv = nullptr;
    TRY(stage_out(h, h->bo1, mean_out, nT, odev, &dm));
    TRY(stage_out(h, h->bo2, var_out, nT, odev, &dv));
    TRY(smoother_backward_impl(h, xs_dev, (const double*)pR, rshared ? 0 : 1, dm, dv));
    tm.kernels_done();
    TRY(copy_back(h, mean_out, dm, nT, odev));
    TRY(copy_back(h, var_out, dv, nT, odev));
    return tm.finish();
}

static int affine_impl(tgp_handle* h, bool rnd, const double* x0dev, const double* eps_t, const double* eps_e, double* mean_dev,
                       double* var_dev) {
    h->reduce_valid = false;
    h->smoother_valid = false;
    h->group_active = false;
    // group layout (tgp_group_smooth.hpp): prior marginals and rand, both orderings, LTI and per-step layouts. LTI Forward marginals
    // from d = 7 (measured in round 1); everything else from d = 9, where the lane-per-chunk alternative is out-of-line
    // private-memory code (d = 14, T = 2e5: rand 60-150 ms, Reverse / per-step marginals 176-270 ms)
    const bool grp_affine = h->use_group_marg && h->opt_group && h->kt->group_reduce_marginals != nullptr && !h->sde &&
                            (h->opt_group == 2 || h->d >= 9 || (!rnd && h->lti && h->ordering == 0 && h->d >= 7));
    if (grp_affine) {
        const int64_t Tm = h->T * h->p;
        int64_t L0 = h->opt_chunk;
        if (L0 <= 0) {
            const int64_t nch = h->kt->group_chunks_per_block == 32 ? 16384 : 8192;
            L0 = (Tm + nch - 1) / nch;
            if (L0 < 8) L0 = 8;
        }
        L0 = ((L0 + h->p - 1) / h->p) * h->p;
        if (L0 > Tm) L0 = Tm;
        h->L0 = (int)L0;
        h->n0 = (Tm + L0 - 1) / L0;
        TRY(scan_prepare(h, h->Rv, kAffineCov, h->n0));      // (rand: the covariance part of the elements stays zero)
        h->group_active = true;                  // the scans over Rv run in the group layout too
        int* badg = flag_ptr(h);
        {
            LaunchScope ls(h, rnd ? "k_group_reduce_affine<rand>" : "k_group_reduce_affine<marginals>");
            h->kt->group_reduce_marginals(rnd, h->mv, h->L0, h->n0, eps_t, h->Rv.E[0], badg, h->stream);
        }
        scan_up(h, h->Rv);
        scan_down(h, h->Rv, x0dev);
        {
            LaunchScope ls(h, rnd ? "k_group_apply_affine<rand>" : "k_group_apply_affine<marginals>");
            h->kt->group_apply_marginals(rnd, h->mv, h->L0, h->n0, h->Rv.S[0], eps_t, eps_e, mean_dev, var_dev, badg, h->stream);
        }
        h->group_active = false;
        return TGP_OK;
    }
    choose_chunk(h);
    TRY(ensure_tiled(h));
    const int monoid = rnd ? kAffineMean : kAffineCov;
    TRY(scan_prepare(h, h->Rv, monoid, h->n0));
    int* bad = flag_ptr(h);
    {
        LaunchScope ls(h, rnd ? "k_reduce_affine<rand>" : "k_reduce_affine<marginals>");
        h->kt->reduce_affine(h->lti, rnd, h->mv, h->L0, h->n0, eps_t, h->Rv.E[0], bad, h->stream);
    }
    scan_up(h, h->Rv);
    scan_down(h, h->Rv, x0dev);
    {
        LaunchScope ls(h, rnd ? "k_apply_affine<rand>" : "k_apply_affine<marginals>");
        h->kt->apply_affine(h->lti, rnd, h->mv, h->L0, h->n0, h->Rv.S[0], eps_t, eps_e, mean_dev, var_dev, bad, h->stream);
    }
    return TGP_OK;
}

// Prior marginals of an LTI model (every block shared, scalar observations): lgssm.jl:99-109 never sees data, and with constant blocks the
// predicted state (m_t, P_t) runs into its fixed point -- at once for a GP prior, whose x0 IS the stationary distribution.  The host runs
// the recursion until it no longer changes (2 ulp, or the emitted values alternate in their last bits), the device writes the head and the constant: the
// call is bound by its 16 B/step of output.
__global__ __launch_bounds__(256) void k_fill_marginals(double* __restrict__ mean, double* __restrict__ var, long long T, const double* __restrict__ tab, int n,
                                                        int reverse) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += stride) {
        const long long k = reverse ? T - 1 - t : t;
        const int i = k < n ? (int)k : n - 1;
        mean[t] = tab[i];
        var[t] = tab[n + i];
    }
}

static int lti_marginals(tgp_handle* h, double* dm, double* dv, bool* served) {
    *served = false;
    // (the shared blocks on the host: `hostm` up to d = 8, `widem` -- the same layout -- for the wide LTI models of 8 < d <= 63, tgp_wide.hip)
    const std::vector<double>& blocks = !h->hostm.empty() ? h->hostm : h->widem;
    if (!h->opt_modal || chunk_engine_requested(h) || blocks.empty() || h->p != 1 || h->sde || (h->hostm.empty() && (!h->opt_wide || h->wide_ht != nullptr))) return TGP_OK;
    const int d = h->d;
    const size_t dd = (size_t)d * d;
    const double* q = blocks.data();
    const double *A = q, *a = q + dd, *Q = q + dd + d, *H = q + 2 * dd + d, hh = q[2 * dd + 2 * d], R = q[2 * dd + 2 * d + 1];
    constexpr int kMax = 4096;
    std::vector<double> m(h->x0m), P(dd), Pn(dd), t1(dd), mn(d), tab_m, tab_v;
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) P[i * d + j] = h->x0P[(i < j ? i : j) + (size_t)(i < j ? j : i) * d];      // Symmetric(x0.P), row-major
    double pm2 = 0.0, pv2 = 0.0, prev_delta = 0.0, rate = 0.0;
    bool settled = false;
    for (int t = 0; t < kMax && (int64_t)t < h->T; ++t) {
        for (int i = 0; i < d; ++i) {
            double v = a[i];
            for (int k = 0; k < d; ++k) v += A[i + k * d] * m[k];
            mn[i] = v;
            for (int j = 0; j < d; ++j) {
                double w = 0.0;
                for (int k = 0; k < d; ++k) w += A[i + k * d] * P[(k <= j ? k * d + j : j * d + k)];
                t1[i * d + j] = w;
            }
        }
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                double w = 0.0;
                for (int k = 0; k < d; ++k) w += t1[i * d + k] * A[j + k * d];
                Pn[i * d + j] = w + Q[(i < j ? i : j) + (size_t)(i < j ? j : i) * d];
            }
        // Forward: the emission of the PREDICTED state (lgssm.jl:99-103); Reverse: of the state before the transition (:111-115)
        const std::vector<double>&mE = h->ordering == 0 ? mn : m, &PE = h->ordering == 0 ? Pn : P;
        double mu = hh, va = R;
        for (int i = 0; i < d; ++i) {
            mu += H[i] * mE[i];
            double w = 0.0;
            for (int k = 0; k < d; ++k) w += H[k] * PE[(k <= i ? k * d + i : i * d + k)];
            va += w * H[i];
        }
        tab_m.push_back(mu);
        tab_v.push_back(va);
        bool moved = false;
        double delta = 0.0;      // the step's largest change, in units of the scale the criterion uses
        for (int i = 0; i < d; ++i) {
            const double sm = std::fabs(mn[i]) + std::sqrt(std::fabs(Pn[i * d + i]));
            if (sm > 0.0) delta = std::max(delta, std::fabs(mn[i] - m[i]) / sm);
            moved = moved || std::fabs(mn[i] - m[i]) > 4.5e-16 * sm;
            for (int j = 0; j < d; ++j) {
                const double sp = 0.5 * (std::fabs(Pn[i * d + i]) + std::fabs(Pn[j * d + j]));
                if (sp > 0.0) delta = std::max(delta, std::fabs(Pn[i * d + j] - P[i * d + j]) / sp);
                moved = moved || std::fabs(Pn[i * d + j] - P[i * d + j]) > 4.5e-16 * sp;
            }
        }
        // the contraction rate, while the changes are still well above rounding: "no longer moves" bounds the DISTANCE to the fixed point by
        // (last change) / (1 - rate) only -- a recursion that creeps (rate -> 1) is not settled when its steps fall below 2 ulp (round-4 advice)
        if (delta > 1e-11 && prev_delta > 1e-11 && delta < prev_delta) rate = delta / prev_delta;
        prev_delta = delta;
        // a recursion whose EMITTED values have come back to those of two steps ago alternates in its last bits: as settled as it gets
        const bool cyc = t >= 2 && mu == pm2 && va == pv2;
        pm2 = tab_m.size() >= 2 ? tab_m[tab_m.size() - 2] : 0.0;
        pv2 = tab_v.size() >= 2 ? tab_v[tab_v.size() - 2] : 0.0;
        m = mn;
        P = Pn;
        if (!moved || cyc) {
            settled = true;
            break;
        }
    }
    if (!settled && (int64_t)tab_m.size() < h->T) return TGP_OK;      // (does not settle within the table: the general engine)
    if (settled && (int64_t)tab_m.size() < h->T && rate > 0.0 && !(4.5e-16 / (1.0 - rate) <= 1e-10)) return TGP_OK;      // (creeps: not within 1e-10 of its fixed point)
    const int n = (int)tab_m.size();
    std::vector<double> tab(2 * (size_t)n);
    std::memcpy(tab.data(), tab_m.data(), sizeof(double) * n);
    std::memcpy(tab.data() + n, tab_v.data(), sizeof(double) * n);
    HIPCHK(h->balt.ensure(tab.size() * sizeof(double)));
    HIPCHK(hipMemcpyAsync(h->balt.p, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));      // (tab is a temporary)
    {
        LaunchScope ls(h, "k_fill_marginals<lti>");
        const long long want = (h->T + 255) / 256;
        const unsigned blocks = (unsigned)(want < 4096 ? want : 4096);
        hipLaunchKernelGGL(k_fill_marginals, dim3(blocks), dim3(256), 0, h->stream, dm, dv, (long long)h->T, static_cast<const double*>(h->balt.p), n, h->ordering);
    }
    *served = true;
    return TGP_OK;
}

int tgp_marginals(tgp_handle* h, uint32_t flags, double* mean_out, double* var_out) {
    StreamGuard stream_guard_(h);
    TRY(check_ready(h, /*general=*/false));
    if (!mean_out || !var_out) return h->fail(TGP_EINVAL, "null output");
    const bool odev = (flags & TGP_OUT_DEVICE) != 0;
    const size_t nT = (size_t)h->T * h->p * sizeof(double);   // one value per (time step, observation)
    if (!h->is_dense || !h->widem.empty()) {      // (LTI models: the recursion on the host until it settles, one fill kernel)
        double *dm0 = nullptr, *dv0 = nullptr;
        TRY(stage_out(h, h->bo1, mean_out, nT, odev, &dm0));
        TRY(stage_out(h, h->bo2, var_out, nT, odev, &dv0));
        bool served = false;
        TRY(lti_marginals(h, dm0, dv0, &served));
        if (served) {
            TRY(copy_back(h, mean_out, dm0, nT, odev));
            TRY(copy_back(h, var_out, dv0, nT, odev));
            HIPCHK(hipStreamSynchronize(h->stream));
            resolve_profile(h);
            return TGP_OK;
        }
        if (!h->is_dense) resolve_table(h);
    }
    CallTimer tm(h);
    tm.inputs_done();
    double *dm = nullptr, *dv = nullptr;
    TRY(stage_out(h, h->bo1, mean_out, nT, odev, &dm));
    TRY(stage_out(h, h->bo2, var_out, nT, odev, &dv));
    if (h->is_dense) {
        TRY(dense_fail(h, tgp_dense::marginals(h->dense, dm, dv, h->result.d(), h->stream)));
        tm.kernels_done();
        TRY(copy_back(h, mean_out, dm, nT, odev));
        TRY(copy_back(h, var_out, dv, nT, odev));
        return tm.finish();
    }
    TRY(affine_impl(h, false, h->bx0.d(), nullptr, nullptr, dm, dv));
    tm.kernels_done();
    TRY(copy_back(h, mean_out, dm, nT, odev));
    TRY(copy_back(h, var_out, dv, nT, odev));
    return tm.finish();
}

int tgp_rand(tgp_handle* h, const double* eps_t, const double* eps_e, const double* eps_0, uint32_t flags, double* y_out) {
    StreamGuard stream_guard_(h);
    TRY(check_ready(h, /*general=*/false));
    if (!eps_t || !eps_e || !eps_0 || !y_out) return h->fail(TGP_EINVAL, "null eps / output");
    const bool idev = (flags & TGP_IN_DEVICE) != 0, odev = (flags & TGP_OUT_DEVICE) != 0;
    const size_t nT = (size_t)h->T * h->p * sizeof(double);   // one value per (time step, observation)
    const int d = h->d;
    // x0 = m + cholesky(Symmetric(P + 1e-12 I)).U' eps_0   (gaussian.jl:35-43) -- d x d, on the host
    std::vector<double> U((size_t)d * d, 0.0), x0(d), zeroP((size_t)d * d, 0.0);
    for (int j = 0; j < d; ++j) {
        for (int i = 0; i <= j; ++i) {
            double acc = h->x0P[i + j * d] + (i == j ? 1e-12 : 0.0);
            for (int k = 0; k < i; ++k) acc -= U[k + i * d] * U[k + j * d];
            if (i == j) {
                if (!(acc > 0.0)) return h->fail(TGP_ENOTPD, "x0.P + 1e-12 I is not positive definite");
                U[j + j * d] = std::sqrt(acc);
            } else {
                U[i + j * d] = acc / U[i + i * d];
            }
        }
    }
    for (int i = 0; i < d; ++i) {
        double acc = 0.0;
        for (int k = 0; k <= i; ++k) acc += U[k + i * d] * eps_0[k];
        x0[i] = h->x0m[i] + acc;
    }
    // LTI, Forward, scalar observations, d <= 6: ONE kernel over the draws on the dense powers of the transition (tgp_modal::rand_lti)
    const bool rand_one = !chunk_engine_requested(h) && !h->is_dense && h->opt_modal && h->lti && h->p == 1 && h->ordering == 0 && !h->sde && d <= tgp_plan::kRandMaxD && !h->hostm.empty();
    // ... wide LTI models (8 < d <= 63, tgp_wide.hip): ONE kernel on the open loop, chunks warmed up on the same draws
    const bool rand_wide = !rand_one && !chunk_engine_requested(h) && h->opt_group != 2 && h->opt_modal && h->opt_wide && !h->widem.empty() && h->wide_ht == nullptr &&
                           h->p == 1 && h->ordering == 0;
    if (!rand_one && !rand_wide) resolve_table(h);      // (the general engine's kernel choice -- seconds on a machine without a cached verdict -- is not part of the call's time)
    CallTimer tm(h);
    if (rand_wide) {
        if (!h->wide) h->wide = tgp_wide::create();
        const size_t dd = (size_t)d * d;
        const double* q = h->widem.data();
        tgp_wide::ModelHost mh;
        mh.d = d;
        mh.A = q; mh.a = q + dd; mh.Q = q + dd + d; mh.H = q + 2 * dd + d; mh.hh = q[2 * dd + 2 * d]; mh.R = q[2 * dd + 2 * d + 1];
        mh.x0m = h->x0m.data();
        mh.x0P = h->x0P.data();
        const void *pet = nullptr, *pee = nullptr;
        TRY(stage_in(h, h->beps_t, eps_t, (size_t)h->T * d * sizeof(double), idev, &pet));
        TRY(stage_in(h, h->beps_e, eps_e, nT, idev, &pee));
        tm.inputs_done();
        double* dy = nullptr;
        TRY(stage_out(h, h->bo1, y_out, nT, odev, &dy));
        bool declined = true;
        std::string err;
        {
            LaunchScope ls(h, "k_wide_rand");
            if (tgp_wide::rand(h->wide, h->stream, mh, h->T, x0.data(), (const double*)pet, (const double*)pee, dy, &declined, &err) != 0) return h->fail(TGP_EHIP, err);
        }
        if (!declined) {
            tm.kernels_done();
            TRY(copy_back(h, y_out, dy, nT, odev));
            return tm.finish();
        }
    }
    if (rand_one) {
        tgp_plan::ModelHost mh;
        tgp_plan::RandPlan rp;
        if (modal_host_model(h, mh)) {
            tgp_modal::plan_rand(mh, rp);
            if (rp.why == tgp_plan::kOk && h->T >= 2) {
                const void *pet = nullptr, *pee = nullptr;
                TRY(stage_in(h, h->beps_t, eps_t, (size_t)h->T * d * sizeof(double), idev, &pet));
                TRY(stage_in(h, h->beps_e, eps_e, nT, idev, &pee));
                tm.inputs_done();
                double* dy = nullptr;
                TRY(stage_out(h, h->bo1, y_out, nT, odev, &dy));
                {
                    LaunchScope ls(h, "k_rand_one");
                    const int rc = tgp_modal::rand_lti(h->stream, rp, x0.data(), (const double*)pet, (const double*)pee, h->T, dy, nullptr);
                    if (rc != 0) return h->fail(TGP_EHIP, std::string("tgp_rand: launch: ") + hipGetErrorString((hipError_t)rc));
                }
                tm.kernels_done();
                TRY(copy_back(h, y_out, dy, nT, odev));
                return tm.finish();
            }
        }
    }
    resolve_table(h);
    if (h->is_dense) {
        const void *pet_d = nullptr, *pee_d = nullptr;
        TRY(stage_in(h, h->beps_t, eps_t, (size_t)h->T * d * sizeof(double), idev, &pet_d));
        TRY(stage_in(h, h->beps_e, eps_e, nT, idev, &pee_d));
        tm.inputs_done();
        double* dy_d = nullptr;
        TRY(stage_out(h, h->bo1, y_out, nT, odev, &dy_d));
        TRY(dense_fail(h, tgp_dense::rand(h->dense, x0.data(), (const double*)pet_d, (const double*)pee_d, h->mv.small_out, dy_d, h->stream)));
        tm.kernels_done();
        TRY(copy_back(h, y_out, dy_d, nT, odev));
        return tm.finish();
    }
    TRY(upload_x0(h, h->bx0r, x0.data(), zeroP.data()));
    const void *pet = nullptr, *pee = nullptr;
    TRY(stage_in(h, h->beps_t, eps_t, (size_t)h->T * d * sizeof(double), idev, &pet));
    TRY(stage_in(h, h->beps_e, eps_e, nT, idev, &pee));
    tm.inputs_done();
    double* dy = nullptr;
    TRY(stage_out(h, h->bo1, y_out, nT, odev, &dy));
    TRY(affine_impl(h, true, h->bx0r.d(), (const double*)pet, (const double*)pee, dy, nullptr));
    tm.kernels_done();
    TRY(copy_back(h, y_out, dy, nT, odev));
    return tm.finish();
}

#include "tgp_api_grad.inc"      // tgp_logpdf_grad / tgp_logpdf_grad_sde: tangent scans and the dual-number kernels
#include "tgp_api_cache.inc"      // the verdict cache on disk
#include "tgp_api_shard.inc"      // time shards: the scan-element exchange of the general engine and of the stationary-gain engine
int tgp_elem_apply(int kind, int d, const double* elem, const double* m, const double* P, double* m_out, double* P_out) {
    if (!elem || !m || !P || !m_out || !P_out) return TGP_EINVAL;
    const KernelTable* kt = kernel_table(d);
    return kt ? kt->host_apply(kind, elem, m, P, m_out, P_out) : TGP_EUNSUPPORTED;
}

int tgp_elem_combine(int kind, int d, const double* earlier, const double* later, double* out) {
    if (!earlier || !later || !out) return TGP_EINVAL;
    const KernelTable* kt = kernel_table(d);
    return kt ? kt->host_combine(kind, earlier, later, out) : TGP_EUNSUPPORTED;
}

int tgp_last_timing(const tgp_handle* h, double* kernel_ms, double* h2d_ms, double* d2h_ms) {
    if (!h) return TGP_EINVAL;
    if (kernel_ms) *kernel_ms = h->kernel_ms;
    if (h2d_ms) *h2d_ms = h->h2d_ms;
    if (d2h_ms) *d2h_ms = h->d2h_ms;
    return TGP_OK;
}

namespace {
__global__ void k_empty() {}
}  // namespace
int tgp_profile_empty_launch(tgp_handle* h) {
    if (!h) return TGP_EINVAL;
    StreamGuard stream_guard_(h);
    TRY(bind_device(h));
    {
        LaunchScope ls(h, "k_empty");
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, h->stream);
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    resolve_profile(h);
    return TGP_OK;
}

int tgp_profile_reset(tgp_handle* h) {
    if (!h) return TGP_EINVAL;
    resolve_profile(h);
    h->prof.clear();
    if (h->dense) tgp_dense::profile_reset(h->dense);
    return TGP_OK;
}
int tgp_profile_count(tgp_handle* h) {
    if (!h) return 0;
    if (!h->pending.empty()) {                     // the tgp_shard_* phases enqueue without a closing synchronisation
        (void)hipStreamSynchronize(h->stream);
        resolve_profile(h);
    }
    return (int)h->prof.size() + (h->dense ? tgp_dense::profile_count(h->dense) : 0);
}
int tgp_profile_get(tgp_handle* h, int idx, char* name, int name_cap, double* total_ms, int64_t* calls) {
    if (!h || idx < 0) return TGP_EINVAL;
    if (idx >= (int)h->prof.size()) {
        const int di = idx - (int)h->prof.size();
        if (!h->dense || di >= tgp_dense::profile_count(h->dense)) return TGP_EINVAL;
        const tgp_dense::KernelTime kt = tgp_dense::profile_get(h->dense, di);
        if (name && name_cap > 0) {
            std::strncpy(name, kt.name, (size_t)name_cap - 1);
            name[name_cap - 1] = 0;
        }
        if (total_ms) *total_ms = kt.ms;
        if (calls) *calls = kt.calls;
        return TGP_OK;
    }
    if (name && name_cap > 0) {
        std::strncpy(name, h->prof[idx].name.c_str(), (size_t)name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (total_ms) *total_ms = h->prof[idx].ms;
    if (calls) *calls = h->prof[idx].calls;
    return TGP_OK;
}

}  // extern "C"
