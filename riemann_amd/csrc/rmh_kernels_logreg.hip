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
        const size_t lds = sizeof(double) * (M::lds_doubles(false) + icdf_doubles(c.nmode));
        if (r.pt_K > 0)
            return launch(k_run_rw<M, 8, false, false, true, true>, c.nblocks, 8, lds, st, m, p, c, r);
        // eight waves per workgroup, as the HMC kernels (one workgroup per CU whatever the dimension: a second wave per
        // SIMD fills the barrier / s_waitcnt gaps of the stage loop; what the 256-register cap spills sits outside it)
        // (compile-time k-step counts at d = 16 NT and d = 16 (NT - 1) + 1 .. 4: launch_rw_ks, rmh_launch.hpp)
        if (adapt) return launch_rw_ks<M, 8, true, false, true>(m, p, c, r, lds, st);
        return launch_rw_ks<M, 8, false, false, true>(m, p, c, r, lds, st);
    })
}

hipError_t launch_logreg_pcn(int NT, const LogRegArgs &m, const RwArgs &p, const ChainArgs &c,
                             const RunArgs &r, bool adapt, hipStream_t st)
{
    RMH_NT_DISPATCH(NT, {
        using M = LogRegModel<kNT>;
        const size_t lds = sizeof(double) * (M::lds_doubles(false) + icdf_doubles(c.nmode));
        if (r.pt_K > 0) return launch(k_run_rw<M, 8, false, true, true, true>, c.nblocks, 8, lds, st, m, p, c, r);
        if (adapt) return launch_rw_ks<M, 8, true, true, true>(m, p, c, r, lds, st);
        return launch_rw_ks<M, 8, false, true, true>(m, p, c, r, lds, st);
    })
}

hipError_t launch_logreg_hmc(int NT, const LogRegArgs &m, const HmcArgs &p, const ChainArgs &c,
                             const RunArgs &r, bool adapt, hipStream_t st)
{
    RMH_NT_DISPATCH(NT, {
        using M = LogRegModel<kNT>;
        const size_t lds = sizeof(double) * (M::lds_doubles(true) + icdf_doubles(c.nmode));
        // d > 64: eight waves per workgroup (two per SIMD, <= 256 registers each).  The
        // second wave cannot overlap VALU with MFMA on this chip, but it fills the
        // s_waitcnt / barrier gaps of the stage loop: 0.52 -> 0.61 of the FP64 roof,
        // although the 256-register cap costs ~300 spilled registers outside the loop.
        // (round 3: the link-function table and the inverse-CDF table alone are 78 KB of LDS, so there is ONE
        // workgroup per CU whatever the dimension: with four waves the small dimensions ran one wave per
        // SIMD -- d = 64 took 4.25 ms per iteration where d = 68, the first eight-wave size, took 3.53.
        // Eight waves for every dimension; sixteen at d <= 16 spill 100 registers at the 128 cap.)
        constexpr int NW = 8;
        if constexpr (kNT >= 2) { // d = 16 NF + 1 .. 16 NF + 4: the last column tile of the gradient on the VALU (rmh_models.hpp)
            if (m.ks == 4 * (kNT - 1) + 1 && r.pt_K == 0 && !(c.nmode && p.chM) && !policy(RMH_POLICY_NO_TAIL)) {
                using MT = LogRegModel<(kNT >= 2 ? kNT : 2), true>;   // (kNT = 1 never gets here; the launcher is not a template)
                if (c.nmode) { // the 64-bit normal mode (its draw as a compile-time variant; with a mass matrix: the generic kernel)
                    if (adapt) return launch(k_run_hmc<MT, NW, true, false, false, true>, c.nblocks, NW, lds, st, m, p, c, r);
                    return launch(k_run_hmc<MT, NW, false, false, false, true>, c.nblocks, NW, lds, st, m, p, c, r);
                }
                if (p.chM) { // with a mass matrix: the same eight-wave evaluation, the triangular products around it
                    if (adapt) return launch(k_run_hmc<MT, NW, true, true>, c.nblocks, NW, lds, st, m, p, c, r);
                    return launch(k_run_hmc<MT, NW, false, true>, c.nblocks, NW, lds, st, m, p, c, r);
                }
                if (adapt) return launch(k_run_hmc<MT, NW, true, false>, c.nblocks, NW, lds, st, m, p, c, r);
                return launch(k_run_hmc<MT, NW, false, false>, c.nblocks, NW, lds, st, m, p, c, r);
            }
        }
        if (m.ks == 4 * kNT && r.pt_K == 0 && !policy(RMH_POLICY_NO_TAIL)) { // d = 16 NT: no k-step test inside the eta chains
            using MF = LogRegModel<kNT, false, true>;
            if (p.chM) {
                if (adapt) return launch(k_run_hmc<MF, NW, true, true>, c.nblocks, NW, lds, st, m, p, c, r);
                return launch(k_run_hmc<MF, NW, false, true>, c.nblocks, NW, lds, st, m, p, c, r);
            }
            if (adapt) return launch(k_run_hmc<MF, NW, true, false>, c.nblocks, NW, lds, st, m, p, c, r);
            return launch(k_run_hmc<MF, NW, false, false>, c.nblocks, NW, lds, st, m, p, c, r);
        }
        if (r.pt_K > 0) { // parallel tempering: HMC / MALA (with or without a mass matrix) within the rungs
            if (p.chM) return launch(k_run_hmc<M, 8, false, true, true>, c.nblocks, 8, lds, st, m, p, c, r);
            return launch(k_run_hmc<M, 8, false, false, true>, c.nblocks, 8, lds, st, m, p, c, r);
        }
        if (p.chM) { // mass-matrix variant
            if (adapt) return launch(k_run_hmc<M, NW, true, true>, c.nblocks, NW, lds, st, m, p, c, r);
            return launch(k_run_hmc<M, NW, false, true>, c.nblocks, NW, lds, st, m, p, c, r);
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
        const size_t lds = sizeof(double) * (M::lds_doubles(true) + icdf_doubles(c.nmode));
        if (p.chM) return launch(k_propose_hmc<M, 4, true>, c.nblocks, 4, lds, st, m, p, c, theta_p, logq);
        return launch(k_propose_hmc<M, 4, false>, c.nblocks, 4, lds, st, m, p, c, theta_p, logq);
    })
}

} // namespace rmh
