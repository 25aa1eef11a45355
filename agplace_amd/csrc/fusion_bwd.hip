// Backward of the FCODE integrator (discretise-then-optimise, reference ffns.py:84 uses plain
// `odeint`, i.e. autograd through the unrolled solver).
//
// STATUS: not implemented yet -- the entry point exists so that the C ABI is complete and the
// Python autograd.Function fails loudly (AGP_E_UNSUPPORTED) instead of silently detaching.
#include "common.hpp"

extern "C" int agp_fcode_bwd(const float* traj, const float* gy, const void* w_hi, const void* w_lo,
                             const void* wt_hi, const void* wt_lo, const float* bias, int b, int act,
                             int method, const float* dt, int nsteps, float* gx, float* gw, float* gb,
                             void* stream) {
    (void)traj; (void)gy; (void)w_hi; (void)w_lo; (void)wt_hi; (void)wt_lo; (void)bias; (void)b;
    (void)act; (void)method; (void)dt; (void)nsteps; (void)gx; (void)gw; (void)gb; (void)stream;
    return AGP_E_UNSUPPORTED;
}
