/*
 * fp64_lean.h — correctly rounded fp64 divide / sqrt / reciprocal length for gfx950 without the
 * compiler's range scaffolding, for operands a cheap test has shown to be in range.
 *
 * hipcc expands `a / b` on doubles into 11 VALU instructions:
 *     s0 = v_div_scale(b, b, a)          s1 = v_div_scale(a, b, a) -> vcc
 *     r  = v_rcp_f64(s0)                 (quarter rate: 16.4 cycles against 4.4 for an fp64 FMA,
 *     e  = fma(-s0, r, 1); r = fma(r, e, r)       scripts/ubench/fp64_issue.hip)
 *     e  = fma(-s0, r, 1); r = fma(r, e, r)
 *     q  = s1 * r;  rem = fma(-s0, q, s1)
 *     v_div_fmas(rem, r, q)  [fma, then * 2^+-128 when vcc]      v_div_fixup(.., b, a)  [inf / nan / 0 cases]
 * and `sqrt(x)` into ~18 (scale select + ldexp, v_rsq_f64, 2 mul, 7 fma, ldexp back, class test + 2 selects).
 * When neither operand needs scaling v_div_scale returns its operand unchanged with vcc = 0, v_div_fmas
 * is a plain fma and v_div_fixup passes its first operand through, so
 *
 *     div_with(a, b, rcp_refined(b))  ==  a / b          bit for bit, by construction,
 *
 * and the five instructions of rcp_refined depend on the DENOMINATOR ONLY: every further division by the
 * same b costs three (mul, fma, fma).  That is where the time goes on this path: a ray direction is the
 * denominator of every cube-face and plane test along it, |d|^2 of every sphere root.
 *
 * Ranges (v_div_scale_f64 scales when the denominator is subnormal or above 2^1022, when the exponents
 * differ by >= 768, when the quotient would be subnormal, or when the numerator's exponent field is <= 53;
 * v_div_fixup steps in for zeros, infinities, NaNs and exponent differences beyond +-1024): with
 *     2^-120 <= |b| < 2^120     and     2^-640 <= |a| < 2^640
 * none of that applies (exponent difference <= 760 < 768, quotient >= 2^-760) and the equality holds by
 * construction.  That covers everything the tracer divides: it admits numerators in [2^-240, 2^240) only
 * (c2rt_trace.inc, Oob).  num_ok() below admits the wider 2^-900 <= |a| < 2^700 for the device check: out there
 * v_div_scale does scale (by 2^+-128, exactly: a power of two) and v_div_fmas scales back, so the results still
 * agree — by the sweep (tests/fp64_lean_check.hip: 0 mismatches in 5.5e11 operand pairs over that whole
 * rectangle, profiles/r03_fp64_lean_sweep.json), not by the argument above.  The window tests are two 32-bit integer instructions on the high dword
 * (in_window: v_lshl_add_u32 + v_cmp_lt_u32; zero, subnormal, infinite and NaN operands fall outside any
 * window).  Callers branch wave-uniformly (`__all`) to the compiler's expansion when a lane is outside —
 * tests/fp64_lean_check.hip and near_one's callers; the tracer itself folds one window into an accumulator and
 * redoes the tile instead (c2rt_trace.inc: Oob).
 *
 * sqrt_lean / inv_len: the compiler's own rsq + Goldschmidt sequence minus the 2^-767 scale test, the two
 * ldexp and the zero / infinity select — identical bits for 2^-700 <= x < 2^700 (sqrt_ok accepts 2^+-240).  inv_len also returns
 * RN(1 / RN(sqrt x)), what `1.0 / sqrt(x)` yields, but seeds the reciprocal with the 0.5/sqrt(x) the
 * square root has already refined instead of a second quarter-rate v_rcp_f64 and two Newton steps.
 * Verified against the compiler's expansions on the device and against the host's IEEE operations:
 * tests/fp64_lean_check.hip (bounded sample inside `-m gpu`, scripts/fp64_lean_sweep.sh for the long run).
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace c2rt {

#define LEAN_DEV __device__ __forceinline__

/* true iff x is finite and 2^lo <= |x| < 2^hi (lo > -1022: subnormals and zeros are outside) */
template <int LO, int HI>
LEAN_DEV bool in_window(double x)
{
    static_assert(LO > -1022 && HI <= 1023 && LO < HI, "normal numbers only");
    const uint32_t t = ((uint32_t)__double2hiint(x) << 1) - ((uint32_t)(LO + 1023) << 21);
    return t < ((uint32_t)(HI - LO) << 21);
}

/* windows in which the sequences below are the compiler's expansions bit for bit (the tracer tests ONE
 * window, sqrt_ok's, for numerators and radicands alike: c2rt_trace.inc, Oob) */
LEAN_DEV bool den_ok(double b) { return in_window<-120, 120>(b); }
LEAN_DEV bool num_ok(double a) { return in_window<-900, 700>(a); } /* (wider than the by-construction window: header) */
/* (the square of den_ok's window: a length taken from such an x can be divided by) */
LEAN_DEV bool sqrt_ok(double x) { return in_window<-240, 240>(x); }

/* the refined reciprocal of the compiler's division: v_rcp_f64 + two Newton steps */
LEAN_DEV double rcp_refined(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    return __builtin_fma(r, e, r);
}

/* a / b given r = rcp_refined(b); den_ok(b) && num_ok(a) */
LEAN_DEV double div_with(double a, double b, double r)
{
    const double q = a * r;
    const double rem = __builtin_fma(-b, q, a);
    return __builtin_fma(rem, r, q);
}

/* sqrt(x) for 2^-700 <= x < 2^700; *half_inv = the refined 0.5 / sqrt(x) of the same iteration */
LEAN_DEV double sqrt_lean(double x, double *half_inv = nullptr)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    double d = __builtin_fma(-g, g, x);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    if (half_inv) *half_inv = h;
    return g;
}

/* len = sqrt(x) and inv = 1.0 / len, both correctly rounded, for sqrt_ok(x).  The reciprocal starts from
 * 2h ~ 1/sqrt(x) (relative error ~2^-51 after the square root's own refinement): one Newton step brings it
 * within the last bit, then the division's correction step (q = 1 * r; rem = 1 - len * q; q + rem * r) rounds
 * it correctly — Markstein's theorem — EXCEPT for a significand of all ones, len = 2^e (2 - 2^-52) (the
 * double below a power of two: |p - c| on a sphere of radius 1, 2, 4, ... is one a third of the time):
 * 1 / len = 2^-(e+1) (1 + 2^-53 + 2^-106 + ..) sits 2^-106 above a rounding midpoint, the iteration arrives
 * at the lower neighbour 2^-(e+1) with rem = 2^-53 exactly and the correction step lands on the tie, which
 * rounds to even: back to 2^-(e+1).  The correct result is its upper neighbour, whose low dword is 1: four
 * 32-bit instructions patch it in. */
LEAN_DEV void inv_len(double x, double &len, double &inv)
{
    double h;
    len = sqrt_lean(x, &h);
    double r = h + h;
    const double e = __builtin_fma(-len, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double rem = __builtin_fma(-len, r, 1.0);
    r = __builtin_fma(rem, r, r);
    const uint32_t lhi = (uint32_t)__double2hiint(len), llo = (uint32_t)__double2loint(len);
    const bool ones = ((lhi | 0xFFF00000u) & llo) == 0xFFFFFFFFu;
    inv = __hiloint2double(__double2hiint(r), ones ? 1 : __double2loint(r));
}

/*
 * Re-normalising a vector that is already unit up to rounding: s = |d|^2 lies within a few ulp of 1, and
 * then len = RN(sqrt s) and inv = RN(1 / len) are decided by integer arithmetic on s's distance from 1 in
 * ulps (the reference normalises the screen ray in Camera.getScreenRay, rt/camera.d:147, and again in
 * Node.intersect, rt/node.d:34-36; same for every shadow ray, rt/scene.d:66 -> rt/node.d:34-36).
 *
 * Let u = 2^-52.  Doubles just above 1 are 1 + k u, just below 1 they are 1 - j u/2.
 *   s = 1 + k u (k >= 0):  sqrt s = 1 + k u/2 - k^2 u^2/8 + ..   lies strictly between 1 + (k-1) u/2 and
 *     1 + k u/2 for k >= 1, hence RN = 1 + (k/2) u for even k; for odd k the value is just below the midpoint
 *     of 1 + ((k-1)/2) u and 1 + ((k+1)/2) u, so RN = 1 + ((k-1)/2) u.  Both: len = 1 + (k >> 1) u.
 *   s = 1 - j u/2 (j >= 1):  sqrt s = 1 - j u/4 - j^2 u^2/32 - ..  in units of u/2 below 1 that is j/2 plus a
 *     tiny excess: even j -> just beyond 1 - (j/2) u/2, RN = 1 - (j/2)(u/2); odd j -> just beyond the
 *     midpoint of (j-1)/2 and (j+1)/2 steps, towards the larger step count, RN = 1 - ((j+1)/2)(u/2).
 *     Both: len = 1 - ((j + 1) >> 1) u/2.
 *   inv of len = 1 + m u (m >= 0): 1/len = 1 - m u + m^2 u^2 - .. = 1 - 2m (u/2) + tiny: RN = 1 - 2m (u/2).
 *   inv of len = 1 - n u/2 (n >= 1): 1/len = 1 + n u/2 + n^2 u^2/4 + ..: even n -> 1 + (n/2) u + tiny,
 *     RN = 1 + (n/2) u; odd n -> just above the midpoint, RN = 1 + ((n+1)/2) u.  Both: 1 + ((n + 1) >> 1) u.
 * (the "tiny" terms are below 2^-98 for the |k|, j <= 64 accepted here, far from any rounding boundary).
 * As bit patterns (ONE = 0x3FF0000000000000): s = ONE + k for a signed k (k = -j below 1), and the four
 * cases collapse to   len = ONE + (k >> 1)  [arithmetic shift = floor(k / 2)],   and with dl = k >> 1:
 * inv = ONE - 2 dl  for dl >= 0,   inv = ONE + ((1 - dl) >> 1)  for dl < 0.
 */
LEAN_DEV bool near_one(double s)
{
    const uint64_t bits = (uint64_t)__double_as_longlong(s);
    return bits - (0x3FF0000000000000ull - 64ull) <= 128ull;
}
/* for near_one(s): len = sqrt(s), inv = 1.0 / len, as IEEE would round them */
LEAN_DEV void unit_len(double s, double &len, double &inv)
{
    const int k = __double2loint(s);   /* bits(s) - ONE, which fits the low dword's sign-extension */
    const int dl = k >> 1;
    const int di = dl >= 0 ? -2 * dl : (1 - dl) >> 1;
    len = __longlong_as_double(0x3FF0000000000000ll + (long long)dl);
    inv = __longlong_as_double(0x3FF0000000000000ll + (long long)di);
}

/*
 * fp32: `Color / float` (rt/color.d:128-132) divides three channels by one denominator.  hipcc's correctly rounded
 * f32 division is v_div_scale x2, v_rcp_f32, fma, fma | mul, fma, fma, fma, v_div_fmas, v_div_fixup; without
 * scaling (finite operands with 2^-60 <= |x| < 2^60, or a numerator of +0 over a positive denominator) that is
 * the two functions below, and the first depends on the denominator only.
 */
LEAN_DEV bool f32_ok(float x)
{
    const uint32_t t = (__float_as_uint(x) << 1) - ((uint32_t)(127 - 60) << 24);
    return t < ((uint32_t)120 << 24);
}
LEAN_DEV float rcp_refined_f32(float f)
{
    const float r = __builtin_amdgcn_rcpf(f);
    const float e = __builtin_fmaf(-f, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
LEAN_DEV float div_with_f32(float a, float f, float r)
{
    float q = a * r;
    float rem = __builtin_fmaf(-f, q, a);
    q = __builtin_fmaf(rem, r, q);
    rem = __builtin_fmaf(-f, q, a);
    return __builtin_fmaf(rem, r, q);
}

#undef LEAN_DEV

} // namespace c2rt
