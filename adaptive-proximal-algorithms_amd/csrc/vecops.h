// vecops.h — launchers of the fused vector kernels (vecops.hip).
#pragma once
#include <functional>

#include "apx_internal.h"

namespace apx {

// second stage of gemv_n + f epilogue; partial f sums land in fsc[0..1]; finalize => Ctrl.f_x
int32_t vo_finish_rows(apx_ctx* ctx, int32_t dtype, int fk, int splits, int64_t ld, int64_t m, const void* b,
                       const void* x, int64_t n, void* res, Ctrl* c, double* fsc, bool finalize);
int32_t vo_f_final(apx_ctx* ctx, int fk, const double* fsc, Ctrl* c);
int32_t vo_logistic_grad(apx_ctx* ctx, int32_t dtype, void* grad, int64_t n, const Ctrl* c);
// K7+K8 then the rule (and, without a linear operator, the end of the iteration)
int32_t vo_reduce_rule(apx_ctx* ctx, int32_t dtype, const void* x, const void* xprev, const void* grad,
                       const void* gprev, const void* v, const void* aty, int64_t n, const ProxDev& g, Ctrl* c,
                       bool has_lin, apx_record* recs, int64_t rec_cap);
// K9 then the end of the iteration
int32_t vo_dual_post(apx_ctx* ctx, int32_t dtype, void* y, const void* axnew, int64_t m, const ProxDev& h, Ctrl* c, apx_record* recs,
                     int64_t rec_cap);
int32_t vo_l2_scale(apx_ctx* ctx, int32_t dtype, void* aty, int64_t n, const double* sums, const ProxDev& h, Ctrl* c);
int32_t vo_dual_post_sums(apx_ctx* ctx, int32_t dtype, void* y, const void* axnew, int64_t m, const ProxDev& h, Ctrl* c, double* out2);
int32_t vo_final_sums(apx_ctx* ctx, const double* sums, const ProxDev& h, Ctrl* c, apx_record* recs, int64_t rec_cap);
int32_t vo_dual(apx_ctx* ctx, int32_t dtype, void* y, const void* ax, void* axprev, int64_t m, const ProxDev& h,
                Ctrl* c, apx_record* recs, int64_t rec_cap);
// K9 with the rows of A sharded over ranks: `reduce(buf, count)` all-reduces device doubles
int32_t vo_dual_sharded(apx_ctx* ctx, int32_t dtype, void* y, const void* ax, void* axprev, int64_t m,
                        const ProxDev& h, Ctrl* c, apx_record* recs, int64_t rec_cap,
                        const std::function<int32_t(double*, int64_t)>& reduce);
// K5+K6
// the step of a primal-dual iteration with f = Zero that also leaves k_reduce_n's block partials of the NEXT iteration in `red`
// ([kRedQ][kMaxRedBlocks]) ; vo_rule_only: k_rule on them
int32_t vo_step_red(apx_ctx* ctx, int32_t dtype, void* x, void* xprev, const void* grad, void* gprev, void* v, const void* aty,
                    int64_t n, const ProxDev& g, const Ctrl* c, double* red);
int32_t vo_rule_only(apx_ctx* ctx, const double* red, int64_t n, const ProxDev& g, Ctrl* c, bool has_lin, apx_record* recs, int64_t rec_cap);
// the totals of the one-pass closing kernel's partials + the rule + the end of the iteration + the step (csrc/slab.hip SlabTail)
int32_t vo_step_rule(apx_ctx* ctx, int32_t dtype, void* x, void* xprev, const void* grad, void* gprev, void* v, int64_t n,
                     const ProxDev& g, Ctrl* c, const Ctrl* snap, const double* wred, int32_t ldw, apx_record* recs, int64_t rec_cap);
int32_t vo_step(apx_ctx* ctx, int32_t dtype, void* x, void* xprev, const void* grad, void* gprev, void* v,
                const void* aty, int64_t n, const ProxDev& g, const Ctrl* c);
// K10: out3 (device) = { <grad, z-x>, sum (z-x)^2, raw g(z) accumulator }
int32_t vo_bt_trial(apx_ctx* ctx, int32_t dtype, const void* x, const void* grad, void* z, int64_t n,
                    const ProxDev& g, double gamma, double* out3);
// op-level prox; out_val_sum (device) = raw accumulator of g(y)
int32_t vo_prox(apx_ctx* ctx, int32_t dtype, const void* v, void* y, int64_t n, const ProxDev& g, double gamma,
                bool conj, double* out_val_sum);
// host-driven loops (AdaPDM+): raw sums to device scalars, read back by the host
int32_t vo_reduce_sums(apx_ctx* ctx, int32_t dtype, const void* x, const void* xprev, const void* grad,
                       const void* gprev, const void* v, const void* aty, int64_t n, const ProxDev& g, const Ctrl* c,
                       double* out5);
int32_t vo_dual_trial(apx_ctx* ctx, int32_t dtype, const void* y, const void* ax, const void* axprev, void* ynext,
                      int64_t m, const ProxDev& h, double sigma, double rho, double* out3);
int32_t vo_diff_sumsq(apx_ctx* ctx, int32_t dtype, const void* a, const void* b, int64_t n, double* out1);
int32_t vo_add(apx_ctx* ctx, int32_t dtype, void* out, const void* a, const void* b, int64_t n, const int* done);
// comparison baselines
int32_t vo_extrap(apx_ctx* ctx, int32_t dtype, void* out, const void* a, const void* b, double c, int64_t n);
int32_t vo_xbar(apx_ctx* ctx, int32_t dtype, void* xbar, const void* x, double phi, int64_t n);
int32_t vo_mp_trial(apx_ctx* ctx, int32_t dtype, const void* xprev, const void* aty, const void* atyprev,
                    const void* gradprev, void* v, void* x, int64_t n, const ProxDev& g, double theta, double gamma,
                    double* out3);
int32_t vo_mp_dualres(apx_ctx* ctx, int32_t dtype, const void* yold, const void* axold, const void* ynew,
                      const void* axnew, int64_t m, double sigma_prev, const ProxDev& h, double* out2);
int32_t vo_primal_res(apx_ctx* ctx, int32_t dtype, const void* v, const void* x, const void* grad, const void* aty,
                      int64_t n, double gamma, const ProxDev& g, double* out2);
// g(x) raw accumulator -> out_val_sum (device)
int32_t vo_gval(apx_ctx* ctx, int32_t dtype, const void* x, int64_t n, const ProxDev& g, double* out_val_sum);
double vo_gval_final_host(const ProxDev& g, double s);

}  // namespace apx
