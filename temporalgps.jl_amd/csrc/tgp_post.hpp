// Stationary-gain engine, STREAMING posterior kernel (round 6): logpdf + posterior marginals of an LTI model with one noise variance,
// scalar observations and no missing data behind its head -- what k_steady_one computes (tgp_modal.hip: lgssm.jl:99-238, lgc.jl:46-52,247-257 in the
// modal coordinates of the stationary closed loop), with the workgroups made persistent:
//   * 2048 waves, one RUN of consecutive 1024-step tiles each (64 lanes x 16 steps); a wave streams its run through ONE 8 KB slice of LDS, the next
//     tile's observations travelling while the current one is in work; the filter state passes from tile to tile exactly;
//   * per tile the zero-start sweeps + one DPP scan + the WJ / WG corrections of k_steady_one -- at sixteen steps per lane the scans cost half of what
//     they cost there per step, and nothing is chained through LDS or barriers;
//   * the backward recursion needs the NEXT tile's innovations: a tile's outputs are finished one tile late, from the left-edge state of the tile behind
//     it (tiles are longer than the halo: whatever lies further right has decayed by 2^-64); a run's last tile takes it from a `ghost` pass over the
//     first halo steps of the next run, a run's first tile its entry state from a state-only pass over the halo steps in front of it.
// Same host protocol as k_steady_one with the head on the host (tgp_modal.hip `complete`): head_in / z0p / zeta_out / head_out and their flags, the
// tail variances out of the pinned tables behind their stage flag, sum r^2 per workgroup into `part`.
#pragma once
#include <hip/hip_runtime.h>

#include "tgp_steady_plan.hpp"

namespace tgp_post {

constexpr int kNW = 8, kN = 16, kTile = 64 * kN, kMaxWG = 256;
constexpr int kMaxD = 8;

struct Geometry {
    long long G = 0, R = 0, C = 1;      // tiles behind the head, runs (one wave each), tiles per run in the middle of the series
    int nwg = 0;
};
// the path serves the call: a halo inside one tile
bool applies(const tgp_plan::Modal& md, long long T);
Geometry choose_geometry(const tgp_plan::Modal& md, long long T);

struct Call {
    long long T = 0;
    const double* y = nullptr;          // device
    const double* Rnew = nullptr;       // device: one value, or T values when rnew_per_step
    int rnew_per_step = 0;
    double *mean = nullptr, *var = nullptr;      // device
    // pinned host memory (the protocol of tgp_modal.hip's host head)
    const double* htab = nullptr;       // packed tables: [0] n1, tail variances at tvb_off (behind stage flag 2)
    int tvb_off = 0;
    const long long* flag = nullptr;    // the stages' flags
    double* part = nullptr;             // [nwg] sum r^2 per workgroup
    double* head_in = nullptr;
    const double* z0p = nullptr;
    double* zeta_out = nullptr;
    const double* head_out = nullptr;
    long long* hflag = nullptr;
    long long seq = 0;
    void* xch = nullptr;                // device memory, xch_bytes(): the runs' exchange records (zeroed once; sequence numbers only grow)
};
inline size_t xch_bytes() { return (size_t)kMaxWG * kNW * kMaxD * 16; }
int enqueue(hipStream_t stream, const tgp_plan::Modal& md, const Geometry& g, const Call& c, const char** kname);

}  // namespace tgp_post
