// rmh_kernels_logreg.hip -- instantiations and launchers of the MFMA kernels over the logistic
// regression model (benchmarks.py LogisticRegression; BASELINE cfg 3 and 4)
#include "rmh_run.hpp"
#include "rmh_launch.hpp"


namespace rmh {

// ------------------------------------------------------- logistic regression
hipError_t launch_logreg_eval(int NT, const LogRegArgs &m, const ChainArgs &c, int want_grad,
                              double *lp_out, double *grad_out, double *lprior_out,
                              double *llik_out, hipStream_t st)
{
    RMH_NT_DISPATCH(NT, {
        using M = LogRegModel<kNT>;
        const size_t lds = sizeof(double) * M::lds_doubles(true);
        if (want_grad)
            return launch(k_eval<M, true, 4>, c.nblocks, 4, lds, st, m, c, lp_out, grad_out,
                          lprior_out, llik_out);
        return launch(k_eval<M, false, 4>, c.nblocks, 4, lds, st, m, c, lp_out, grad_out,
                      lprior_out, llik_out);
    })
}

// random walk (plain, AdaptScale, or parallel tempering) on the logistic model: the
// proposal's tile image stays in global memory, the LDS belongs to the X stages
hipError_t launch_logreg_rw(int NT, const LogRegArgs &m, const RwArgs &p, const ChainArgs &c,
                            const RunArgs &r, bool adapt, hipStream_t st)
{
    RMH_NT_DISPATCH(NT, {
        using M = LogRegModel<kNT>;
        const size_t lds = sizeof(double) * (M::lds_doubles(false) + kIcdfDoubles);
        if (r.pt_K > 0)
            return launch(k_run_rw<M, 4, false, false, true, true>, c.nblocks, 4, lds, st, m, p, c, r);
        if (adapt)
            return launch(k_run_rw<M, 4, true, false, false, true>, c.nblocks, 4, lds, st, m, p, c, r);
        return launch(k_run_rw<M, 4, false, false, false, true>, c.nblocks, 4, lds, st, m, p, c, r);
    })
}

hipError_t launch_logreg_pcn(int NT, const LogRegArgs &m, const RwArgs &p, const ChainArgs &c,
                             const RunArgs &r, bool adapt, hipStream_t st)
{
    RMH_NT_DISPATCH(NT, {
        using M = LogRegModel<kNT>;
        const size_t lds = sizeof(double) * (M::lds_doubles(false) + kIcdfDoubles);
        if (r.pt_K > 0) return launch(k_run_rw<M, 4, false, true, true, true>, c.nblocks, 4, lds, st, m, p, c, r);
        if (adapt) return launch(k_run_rw<M, 4, true, true, false, true>, c.nblocks, 4, lds, st, m, p, c, r);
        return launch(k_run_rw<M, 4, false, true, false, true>, c.nblocks, 4, lds, st, m, p, c, r);
    })
}

hipError_t launch_logreg_hmc(int NT, const LogRegArgs &m, const HmcArgs &p, const ChainArgs &c,
                             const RunArgs &r, bool adapt, hipStream_t st)
{
    RMH_NT_DISPATCH(NT, {
        using M = LogRegModel<kNT>;
        const size_t lds = sizeof(double) * (M::lds_doubles(true) + kIcdfDoubles);
        // d > 64: eight waves per workgroup (two per SIMD, <= 256 registers each).  The
        // second wave cannot overlap VALU with MFMA on this chip, but it fills the
        // s_waitcnt / barrier gaps of the stage loop: 0.52 -> 0.61 of the FP64 roof,
        // although the 256-register cap costs ~300 spilled registers outside the loop.
        constexpr int NW = (kNT >= 5) ? 8 : 4;
        if constexpr (kNT == 7) { // d = 97..100: columns 96..99 of the gradient on the VALU (rmh_models.hpp)
            if (m.ks == 25 && r.pt_K == 0 && !p.chM && !policy(RMH_POLICY_NO_TAIL)) {
                using MT = LogRegModel<7, true>;
                if (adapt) return launch(k_run_hmc<MT, NW, true, false>, c.nblocks, NW, lds, st, m, p, c, r);
                return launch(k_run_hmc<MT, NW, false, false>, c.nblocks, NW, lds, st, m, p, c, r);
            }
        }
        if (r.pt_K > 0) { // parallel tempering: HMC / MALA (with or without a mass matrix) within the rungs
            if (p.chM) return launch(k_run_hmc<M, 4, false, true, true>, c.nblocks, 4, lds, st, m, p, c, r);
            return launch(k_run_hmc<M, 4, false, false, true>, c.nblocks, 4, lds, st, m, p, c, r);
        }
        if (p.chM) { // mass-matrix variant: one wave per SIMD (register budget)
            if (adapt) return launch(k_run_hmc<M, 4, true, true>, c.nblocks, 4, lds, st, m, p, c, r);
            return launch(k_run_hmc<M, 4, false, true>, c.nblocks, 4, lds, st, m, p, c, r);
        }
        if (adapt) return launch(k_run_hmc<M, NW, true, false>, c.nblocks, NW, lds, st, m, p, c, r);
        return launch(k_run_hmc<M, NW, false, false>, c.nblocks, NW, lds, st, m, p, c, r);
    })
}

hipError_t launch_logreg_propose_hmc(int NT, const LogRegArgs &m, const HmcArgs &p,
                                     const ChainArgs &c, double *theta_p, double *logq,
                                     hipStream_t st)
{
    RMH_NT_DISPATCH(NT, {
        using M = LogRegModel<kNT>;
        const size_t lds = sizeof(double) * (M::lds_doubles(true) + kIcdfDoubles);
        if (p.chM) return launch(k_propose_hmc<M, 4, true>, c.nblocks, 4, lds, st, m, p, c, theta_p, logq);
        return launch(k_propose_hmc<M, 4, false>, c.nblocks, 4, lds, st, m, p, c, theta_p, logq);
    })
}

} // namespace rmh
