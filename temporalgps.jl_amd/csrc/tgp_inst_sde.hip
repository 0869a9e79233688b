// The per-step lane-per-chunk passes for models whose transitions are evaluated in closed form from the time stamps (ModelView::sde,
// TGP_OPT_SDE_CLOSED_FORM; d <= kSdeBuildMaxD): the same kernels as tgp_inst_dN.hip's <D, false, ...> instantiations, compiled a second
// time (namespace tgp_s, TGP_SDE_BUILD) with a loader that computes A_k, Q_k instead of reading them. The API swaps these launchers into
// a private copy of the kernel table while a model's transition record holds tau alone (ensure_tiled).
#define TGP_NS tgp_s
#define TGP_SDE_BUILD 1
#include "tgp_kernels.hpp"

namespace tgp_s {
namespace {
inline dim3 grid_for(int64_t n, int bs) { return dim3((unsigned)((n + bs - 1) / bs)); }
template <int D> struct Ops {
    static constexpr size_t kLdsFilter = WaveIO<true, true, false, false, (D <= kPrefetchMaxD)>::lds_bytes();
    static constexpr size_t kLdsSmooth = WaveIO<false, true, true, true, (D <= kPrefetchMaxD)>::lds_bytes();
    static constexpr size_t kLdsAffine = WaveIO<true, true, true, true, (D <= kPrefetchMaxD)>::lds_bytes();
    static void reduce_filter(bool, const ModelView& mv, int L0, int64_t n0, double* E0, double* E1, int64_t n1, hipStream_t s) {
        hipLaunchKernelGGL((k_reduce_filter<D, false>), grid_for(n0, 256), dim3(256), kLdsFilter, s, mv, L0, n0, E0, E1, n1);
    }
    template <int MODE>
    static void apply_filter(bool, const ModelView& mv, int L0, int64_t n0, double* S0, const double* E0, const double* S1, int64_t n1,
                             const FilterOut& out, double* R0, double* partial, hipStream_t s) {
        hipLaunchKernelGGL((k_apply_filter<D, false, MODE>), grid_for(n0, 256), dim3(256), kLdsFilter, s, mv, L0, n0, S0, E0, S1, n1, out, R0, partial);
    }
    static void smooth(bool, const ModelView& mv, int L0, int64_t n0, const double* S0, const double* S0r, const double* fs, const double* Rnew,
                       int64_t sRn, double* mean_out, double* var_out, int* bad, hipStream_t s) {
        const dim3 g = grid_for(n0, 256), b(256);
        if (sRn != 0) hipLaunchKernelGGL((k_smooth<D, false, true>), g, b, kLdsSmooth, s, mv, L0, n0, S0, S0r, fs, Rnew, sRn, mean_out, var_out, bad);
        else hipLaunchKernelGGL((k_smooth<D, false, false>), g, b, kLdsSmooth, s, mv, L0, n0, S0, S0r, fs, Rnew, sRn, mean_out, var_out, bad);
    }
    static void reduce_affine(bool, bool rnd, const ModelView& mv, int L0, int64_t n0, const double* eps_t, double* E0, int* bad, hipStream_t s) {
        const dim3 g = grid_for(n0, 256), b(256);
        if (rnd) hipLaunchKernelGGL((k_reduce_affine<D, false, true>), g, b, 0, s, mv, L0, n0, eps_t, E0, bad);
        else hipLaunchKernelGGL((k_reduce_affine<D, false, false>), g, b, 0, s, mv, L0, n0, eps_t, E0, bad);
    }
    static void apply_affine(bool, bool rnd, const ModelView& mv, int L0, int64_t n0, const double* S0, const double* eps_t, const double* eps_e,
                             double* mean_out, double* var_out, int* bad, hipStream_t s) {
        const dim3 g = grid_for(n0, 256), b(256);
        if (rnd) hipLaunchKernelGGL((k_apply_affine<D, false, true>), g, b, kLdsAffine, s, mv, L0, n0, S0, eps_t, eps_e, mean_out, var_out, bad);
        else hipLaunchKernelGGL((k_apply_affine<D, false, false>), g, b, kLdsAffine, s, mv, L0, n0, S0, eps_t, eps_e, mean_out, var_out, bad);
    }
    static const KernelTable* table() {
        static const KernelTable t = [] {
            KernelTable k{};
            k.d = D;
            k.reduce_filter = reduce_filter;
            k.apply_filter_m[0] = apply_filter<0>;
            k.apply_filter_m[1] = apply_filter<1>;
            k.apply_filter_m[2] = apply_filter<2>;
            k.apply_filter_m[3] = apply_filter<3>;
            k.smooth = smooth;
            k.reduce_affine = reduce_affine;
            k.apply_affine = apply_affine;
            return k;
        }();
        return &t;
    }
};
}  // namespace

// the entries above (everything else null), or null for a d without this build
const KernelTable* sde_kernel_table(int d) {
    static_assert(kSdeBuildMaxD == 4, "one Ops<D> per dimension of the build");
    switch (d) {
        case 1: return Ops<1>::table();
        case 2: return Ops<2>::table();
        case 3: return Ops<3>::table();
        case 4: return Ops<4>::table();
        default: return nullptr;
    }
}
}  // namespace tgp_s
