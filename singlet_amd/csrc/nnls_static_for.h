// compile-time loop shared by the NNLS kernels
#pragma once
#include <utility>
#include <type_traits>

// compile-time loop: guarantees that b[] / x[] are only ever indexed by constants
// (so they live in VGPRs) regardless of the optimiser's unroll thresholds.
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}


// x / y, correctly rounded, for finite operands in the normal range (no over- / underflow on the way): the
// instruction sequence hipcc emits for an FP64 division (v_rcp_f64, two Newton steps, quotient, remainder,
// correction) WITHOUT its v_div_scale / v_div_fixup frame, which only acts on denormal or extreme-exponent
// operands and on inf / nan / zero divisors.  In the NNLS the divisors are x + 1e-15 >= 1e-15 and Gram
// diagonals >= 1e-15, the dividends finite steps: bit-identical to `x / y` there, 8 instructions instead of 11
// and no special-case control flow (hipcc had wrapped the division of the tol term in exec-mask branches).
__device__ __forceinline__ double sgl_div_normal(double x, double y) {
    double r = __builtin_amdgcn_rcp(y);
    double e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = x * r;
    const double rem = __builtin_fma(-y, q, x);
    return __builtin_fma(rem, r, q);
}

// One coordinate step of nnls (src/singlet.cpp:233-247) without the row update, branch-free.  In: diff0 = b_i / a_ii
// (before the penalties), x_i >= 0, running tol, go (false: the column has stopped -- its step is forced to zero, which
// leaves x, b and tol as they are).  Out: x_i, tol updated; returns nd = -delta, the factor of the row update
// b += a[:, i] * nd.
//   clamp (-diff > x_i): x_i -> 0, delta = -x_i, tol = 1 unless x_i was 0 (then nothing changes);
//   otherwise            x_i += diff, delta = diff, tol += |diff / (x_i + 1e-15)|  (diff == 0 adds exact zeros).
// Both branches are ONE expression: nd = min(-diff, x_i) is x_i exactly when the reference clamps and -diff otherwise,
// x_i - nd is 0 / x_i + diff, and |nd / (x_i - nd + 1e-15)| is the tol term of the second branch and an exact 0 in the
// "clamped at zero already" case -- which leaves a single select, the tol = 1 of a coordinate that was clamped from a
// positive value.  22 VALU instructions per step where the select-per-quantity form needed 38 (5 selects of 2 v_cndmask,
// the negation); same values bit for bit (the sign of a zero step differs, which no later operation can see).
// The two halves of the step, for the kernels that test whether the coordinate moves at all before paying for the rest
// (four columns per wave: a coordinate at rest in all four skips sgl_nnls_apply and the row update -- with nd = 0,
// sgl_nnls_apply leaves x_i and tol as they are, bit for bit: x_i - 0, tol + |0 / (x_i + 1e-15)|, no reset).
__device__ __forceinline__ double sgl_nnls_nd(double diff0, double xi, bool go, double L1, double L2, double& diff) {
    diff = diff0 - L1;                             // exact no-op when L1 == 0
    diff = __builtin_fma(L2, xi, diff);            // exact no-op when L2 == 0 (x >= 0)
    diff *= go ? 1.0 : 0.0;                        // one multiply instead of two v_cndmask (finite operands)
    double nd;
    asm("v_min_f64 %0, -%1, %2" : "=v"(nd) : "v"(diff), "v"(xi));
    return nd;
}
__device__ __forceinline__ void sgl_nnls_apply(double diff, double nd, double& xi, double& tol) {
    const double xn = xi - nd;
    const double tadd = __builtin_fabs(sgl_div_normal(nd, xn + 1e-15));
    const bool reset = (-diff > xi) & (xi != 0.0);
    tol = reset ? 1.0 : tol + tadd;
    xi = xn;
}
// The same first half for the solves against PER-COLUMN Grams (predict_mask, src/singlet.cpp:458-463), where a diagonal entry
// a_ii - asub_ii can be exactly zero -- a factor whose row of the other factor matrix is all zero, or lies entirely inside
// the column's drawn rows -- and the reference's `b(i) / a(i, i)` is +-inf or NaN (l.233): non-finite steps must come out as
// the reference's do.  With nd = clamp ? x_i : -diff (instead of v_min_f64, which returns the OTHER operand for a NaN):
//   diff NaN   l.237 false, l.243 true: x_i += NaN, b -= a.col(i) * NaN, tol NaN     = nd NaN:  x_i - nd, b += a.col(i) * nd, tol + |nd / ..|
//   diff +inf  the same branch: x_i = inf, b -= a.col(i) * inf, tol += |inf / inf|   = nd -inf: the same expressions
//   diff -inf  l.237 true: the clamp of a finite step (x_i -> 0, tol = 1; nothing if x_i == 0)     = nd x_i
// and a stopped column is gated by a select (0 * inf would be NaN).  Finite steps: the same bits as sgl_nnls_nd.
__device__ __forceinline__ double sgl_nnls_nd_strict(double diff0, double xi, bool go, double L1, double L2, double& diff) {
    diff = diff0 - L1;
    diff = __builtin_fma(L2, xi, diff);
    diff = go ? diff : 0.0;
    return (-diff > xi) ? xi : -diff;
}
// b_i / g_ii for a per-column Gram: the Markstein form (correctly rounded quotient from the correctly rounded reciprocal r_ii,
// see nnls_lane.h) wherever r_ii is a normal number; elsewhere (g_ii zero, denormal, huge, non-finite) the IEEE division the
// reference performs.  `any_irregular` is wave-uniform and false for all but degenerate columns: one scalar branch per coordinate.
__device__ __forceinline__ double sgl_nnls_quotient(double bi, double gii, double rii, bool any_irregular) {
    const double q0 = bi * rii;
    double diff0 = __builtin_fma(__builtin_fma(-q0, gii, bi), rii, q0);
    if (any_irregular) {
        const double q = bi / gii;
        diff0 = __builtin_isnormal(rii) ? diff0 : q;
    }
    return diff0;
}

__device__ __forceinline__ double sgl_nnls_step(double diff0, double& xi, double& tol, bool go, double L1, double L2) {
    double diff;
    const double nd = sgl_nnls_nd(diff0, xi, go, L1, L2, diff);
    sgl_nnls_apply(diff, nd, xi, tol);
    return nd;
}
