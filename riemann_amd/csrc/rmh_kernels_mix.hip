// rmh_kernels_mix.hip -- instantiations and launchers of the MFMA kernels over the mixture of
// Gaussians (models/benchmarks.py:31-62 MixtureDist)
#include "rmh_run.hpp"
#include "rmh_launch.hpp"

namespace rmh {

// ------------------------------------------------------ mixture of Gaussians
hipError_t launch_mix_eval(int NT, const MixArgs &m, const ChainArgs &c, int want_grad, double *lp_out,
                           double *grad_out, double *lprior_out, double *llik_out, hipStream_t st)
{
    RMH_NT_DISPATCH(NT, {
        using M = MixModel<kNT>;
        if (want_grad)
            return launch(k_eval<M, true, 4>, c.nblocks, 4, 0, st, m, c, lp_out, grad_out, lprior_out, llik_out);
        return launch(k_eval<M, false, 4>, c.nblocks, 4, 0, st, m, c, lp_out, grad_out, lprior_out, llik_out);
    })
}

hipError_t launch_mix_hmc(int NT, const MixArgs &m, const HmcArgs &p, const ChainArgs &c,
                          const RunArgs &r, bool adapt, hipStream_t st)
{
    RMH_NT_DISPATCH(NT, {
        using M = MixModel<kNT>;
        const size_t lds = sizeof(double) * (M::lds_doubles(true) + icdf_doubles(c.nmode));
        if (r.pt_K > 0) { // parallel tempering: HMC / MALA (with or without a mass matrix) within the rungs
            if (p.chM) return launch(k_run_hmc<M, 4, false, true, true>, c.nblocks, 4, lds, st, m, p, c, r);
            return launch(k_run_hmc<M, 4, false, false, true>, c.nblocks, 4, lds, st, m, p, c, r);
        }
        if (p.chM) {
            if (adapt) return launch(k_run_hmc<M, 4, true, true>, c.nblocks, 4, lds, st, m, p, c, r);
            return launch(k_run_hmc<M, 4, false, true>, c.nblocks, 4, lds, st, m, p, c, r);
        }
        if (adapt) return launch_hmc_ks<M, 4, true, false, false>(m, p, c, r, lds, st);
        return launch_hmc_ks<M, 4, false, false, false>(m, p, c, r, lds, st);
    })
}

hipError_t launch_mix_propose_hmc(int NT, const MixArgs &m, const HmcArgs &p, const ChainArgs &c,
                                  double *theta_p, double *logq, hipStream_t st)
{
    RMH_NT_DISPATCH(NT, {
        using M = MixModel<kNT>;
        const size_t lds = sizeof(double) * (M::lds_doubles(true) + icdf_doubles(c.nmode));
        if (p.chM) return launch(k_propose_hmc<M, 4, true>, c.nblocks, 4, lds, st, m, p, c, theta_p, logq);
        return launch(k_propose_hmc<M, 4, false>, c.nblocks, 4, lds, st, m, p, c, theta_p, logq);
    })
}

hipError_t launch_mix_rw(int NT, const MixArgs &m, const RwArgs &p, const ChainArgs &c,
                         const RunArgs &r, bool adapt, hipStream_t st)
{
    RMH_NT_DISPATCH(NT, {
        using M = MixModel<kNT>;
        const size_t lds = sizeof(double) * (tri_tiles(kNT) * 64 + icdf_doubles(c.nmode));
        if (r.pt_K > 0)
            return launch(k_run_rw<M, 4, false, false, true>, c.nblocks, 4, lds, st, m, p, c, r);
        // (compile-time k-step count at d = 16 (NT - 1) + 1 .. 4: launch_rw_ks, rmh_launch.hpp; not at d = 16 NT, see launch_hmc_ks)
        if (adapt) return launch_rw_ks<M, 4, true, false, false, false>(m, p, c, r, lds, st);
        return launch_rw_ks<M, 4, false, false, false, false>(m, p, c, r, lds, st);
    })
}

hipError_t launch_mix_pcn(int NT, const MixArgs &m, const RwArgs &p, const ChainArgs &c,
                          const RunArgs &r, bool adapt, hipStream_t st)
{
    RMH_NT_DISPATCH(NT, {
        using M = MixModel<kNT>;
        const size_t lds = sizeof(double) * (tri_tiles(kNT) * 64 + icdf_doubles(c.nmode));
        if (r.pt_K > 0) return launch(k_run_rw<M, 4, false, true, true>, c.nblocks, 4, lds, st, m, p, c, r);
        if (adapt) return launch(k_run_rw<M, 4, true, true, false>, c.nblocks, 4, lds, st, m, p, c, r);
        return launch(k_run_rw<M, 4, false, true, false>, c.nblocks, 4, lds, st, m, p, c, r);
    })
}

} // namespace rmh
