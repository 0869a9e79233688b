// Stationary-gain engine, STREAMING logpdf kernel (round 6): the log marginal likelihood of an LTI model with one noise
// variance, scalar observations and no missing data (lgssm.jl:147-165 on the reference's `Fill` layout) behind its head, as ONE kernel that
// reads y exactly once -- no halo, no pass before it, no wait for the host inside it.
//
// The steps behind the head are cut into R runs, one per wave; a wave streams its run tile by tile (64 lanes x N consecutive steps) through
// its own LDS slice and carries the filter state from tile to tile exactly.  Inside a tile a lane's zero-start end state is a table product
// (the recursion from zero), the lanes' start states come from one DPP scan, and the second sweep is the reference's recursion itself in the modal
// coordinates of the stationary closed loop: r = u - fw . z, z' = M z + fb u + fa.  A run does not know the state it starts from: it starts
// from zero and also accumulates V = sum_t w_t r_t over its first tile (w_t = fw' M^t: how a start state moves the innovations), so that
//     sum r^2 (start state s) = Q - 2 s . V + s' Wt s,   Wt = sum_t w_t' w_t  (data-free, from the host),
// to rounding once the tile is longer than the `halo` of the plan.  The waves of a workgroup apply that to each other after the one barrier
// at the kernel's end (s = the end state of the run before); the first run of every workgroup is closed by the HOST from the workgroups'
// (Q, V, E) triples in pinned memory -- and the head's end state z0 enters the same way, so the kernel never waits for the host.
#pragma once
#include <hip/hip_runtime.h>

#include "tgp_steady_plan.hpp"

namespace tgp_lml {

constexpr int kNW = 8;            // waves per workgroup (one workgroup per CU: its LDS holds the eight tiles)
constexpr int kMaxWG = 512;       // workgroups of a launch at most (n = 16: two per CU; n = 32: one per CU, 256)

struct Geometry {
    int n = 32;                    // steps per lane (16 or 32)
    long long G = 0, R = 0, C = 1; // tiles of 64 n steps behind the head (the last one may be partial), runs (one wave each) that hold a tile, C = Chi
    long long W = 0, Chi = 1, Clo = 1;      // whole tiles; tiles per run of a workgroup's waves 0-3 / 4-7 (waves w and w + 4 share a SIMD)
    int nwg = 0;
    long long first_tile = 0;      // steps of a run's first tile (what Wt sums over)
};

// R runs over the G tiles behind the head; every run starts with a whole tile (>= halo) unless the series is shorter than that (then ONE run)
Geometry choose_geometry(const tgp_plan::Modal& md, long long T);

// pinned host memory the launch writes: part [nwg][1 + 2 d] (value, check) PAIRS (Q, V, E per workgroup; check = the value's bits ^ record_key(seq): a pair
// of this call is known when seen), head_in [nhs], flags [0]: head_in is there
struct Buffers {
    double* part = nullptr;
    double* head_in = nullptr;
    long long* flags = nullptr;
};
constexpr size_t kStampOff = 2 * (size_t)kMaxWG * (1 + 2 * tgp_plan::kMaxD);      // development stamps (TGP_LML_DBG) behind the largest record table
inline size_t part_doubles(int) { return kStampOff + 8 * (size_t)kMaxWG; }
__host__ __device__ inline unsigned long long record_key(long long seq) { return (unsigned long long)seq * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull; }
// Have all records of call `seq` arrived?  *next: the first pair not yet seen (0 at the first call; the scan resumes there).  With every record there
// the kernel has read all of y and written all it writes: the host may go on without synchronising the stream (which stays ordered).
bool records_there(const Geometry& g, int d, const double* part, long long seq, size_t* next);

// Wt = sum_{t < n} w_t' w_t (d x d, row-major), n = Geometry::first_tile -- data-free, O(d^2 log n) on the host
void quad_table(const tgp_plan::Modal& md, long long n, double* Wt);
// Enqueues the kernel on `stream` (no synchronisation).  0 or a hipError_t.
int enqueue(hipStream_t stream, const tgp_plan::Modal& md, const Geometry& g, long long T, const double* y, const Buffers& b, long long seq, const double* Wt,
            const char** kname);
// After flags[1] (or the stream) says the kernel is through: sum r^2 over the steps [nhs, T) given the head's end state z0 (modal coordinates)
double finish(const tgp_plan::Modal& md, const Geometry& g, const double* part, const double* z0, const double* Wt);

}  // namespace tgp_lml
