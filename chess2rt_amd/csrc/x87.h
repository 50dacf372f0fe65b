/*
 * x87.h — the two expressions of Sphere.intersect that the reference evaluates in x87 extended
 * precision, reproduced bit for bit with 64-bit integer arithmetic (the device has no 80-bit type).
 *
 *     info.u = (PI + angle) / (2 * PI);                                  — rt/geometry.d:119
 *     info.v = 1.0 - (PI / 2 + asin((p.y - center.y) / R)) / PI;         — rt/geometry.d:120
 *
 * `PI` is std.math.PI, an 80-bit `real`: `angle` (a double) and the asin result (a double: D's
 * asin(double) returns double) are promoted, every +, -, / rounds to the 64-bit significand of the x87
 * (round to nearest even, the precision-control default on x86-64 Linux / Windows-LDC alike), and the
 * assignment to the double field rounds a second time.  Evaluating the same expression in doubles
 * gives a result that differs in the last place in about one case in a few thousand (double rounding)
 * and wherever the constants' extra 11 bits matter.
 *
 * X87 values here: (neg, e, m), value = (-1)^neg * m * 2^(e - 63), m in [2^63, 2^64) or m == 0.
 * Only what the two expressions need: finite operands, results far from overflow / underflow
 * (|angle| <= pi, |asin| <= pi/2: every intermediate lies in [0, 8) and is 0 or >= 2^-70).
 * Plain C, no floating-point operation inside: any compiler reproduces it.  Built for the device
 * (c2rt_kernels.hip) and, by tests/x87_check.c, for the host, where it is compared with the
 * compiler's own `long double` arithmetic.
 */
#ifndef C2RT_X87_H
#define C2RT_X87_H

#include <stdint.h>

#if defined(__HIPCC__)
#define X87_FN static __host__ __device__ __forceinline__
#else
#define X87_FN static inline
#endif
#ifdef __HIP_DEVICE_COMPILE__
#define X87_MULHI(a, b) __umul64hi((a), (b))
#define X87_CLZ(x) __clzll((long long)(x))
#define X87_BITS(d) ((uint64_t)__double_as_longlong(d))
#define X87_DOUBLE(u) __longlong_as_double((long long)(u))
#else
#include <string.h>
#define X87_MULHI(a, b) ((uint64_t)(((unsigned __int128)(a) * (b)) >> 64))
#define X87_CLZ(x) __builtin_clzll(x)
static inline uint64_t x87_bits_(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double x87_double_(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
#define X87_BITS(d) x87_bits_(d)
#define X87_DOUBLE(u) x87_double_(u)
#endif

typedef struct { uint64_t m; int32_t e; int32_t neg; } x87_t;

/* std.math.PI = 0xC90FDAA22168C235 * 2^-62 (the nearest 64-bit-significand value of pi) */
#define X87_PI_M 0xC90FDAA22168C235ull

X87_FN x87_t x87_make(uint64_t m, int e, int neg) { x87_t r; r.m = m; r.e = e; r.neg = neg; return r; }

/* finite double -> x87 (exact) */
X87_FN x87_t x87_from_double(double d)
{
    const uint64_t u = X87_BITS(d);
    const int neg = (int)(u >> 63);
    int ex = (int)((u >> 52) & 0x7ff);
    uint64_t f = u & 0xfffffffffffffull;
    if (ex == 0) {
        if (f == 0) return x87_make(0, 0, neg);
        const int sh = X87_CLZ(f);           /* subnormal: normalise */
        return x87_make(f << sh, -1022 - 52 + 63 - sh, neg);
    }
    return x87_make((1ull << 63) | (f << 11), ex - 1023, neg);
}

/* round (hi, lo) — a 128-bit significand whose top bit is set, value hi*2^64+lo scaled so that hi is
 * the 64-bit result significand — to nearest even; `sticky`: non-zero bits below lo */
X87_FN x87_t x87_round(uint64_t hi, uint64_t lo, int sticky, int e, int neg)
{
    const uint64_t half = 1ull << 63;
    const int up = (lo > half) | ((lo == half) & (sticky | (int)(hi & 1)));
    if (up) {
        hi += 1;
        if (hi == 0) { hi = half; e += 1; }
    }
    return x87_make(hi, e, neg);
}

/* a + b, one rounding to the 64-bit significand */
X87_FN x87_t x87_add(x87_t a, x87_t b)
{
    if (a.m == 0) return b;
    if (b.m == 0) return a;
    if (a.e < b.e || (a.e == b.e && a.m < b.m)) { const x87_t t = a; a = b; b = t; } /* |a| >= |b| */
    const int d = a.e - b.e;
    /* b's significand as (bh, bl) aligned under a's (ah = a.m, al = 0), plus a sticky bit */
    uint64_t bh, bl;
    int sticky = 0;
    if (d == 0) { bh = b.m; bl = 0; }
    else if (d < 64) { bh = b.m >> d; bl = b.m << (64 - d); }
    else if (d == 64) { bh = 0; bl = b.m; }
    else if (d < 128) { bh = 0; bl = b.m >> (d - 64); sticky = (b.m << (128 - d)) != 0; }
    else { bh = 0; bl = 0; sticky = 1; }
    uint64_t hi, lo;
    int e = a.e;
    if (a.neg == b.neg) {
        lo = bl;
        hi = a.m + bh;
        if (hi < a.m) { /* carry out: shift right by one */
            sticky |= (int)(lo & 1);
            lo = (lo >> 1) | (hi << 63);
            hi = (hi >> 1) | (1ull << 63);
            e += 1;
        }
    } else {
        /* a - b with a borrow from the sticky part */
        uint64_t l = 0 - bl, h = a.m - bh - (bl != 0);
        if (sticky) { /* true value is a hair below (h, l): subtract one unit of the last kept place, keep sticky */
            if (l == 0) h -= 1;
            l -= 1;
        }
        hi = h; lo = l;
        if (hi == 0 && lo == 0) return x87_make(0, 0, 0); /* exact cancellation: +0 in round-to-nearest */
        if (hi == 0) { hi = lo; lo = 0; e -= 64; }
        const int sh = X87_CLZ(hi);
        if (sh) { hi = (hi << sh) | (lo >> (64 - sh)); lo <<= sh; e -= sh; }
    }
    return x87_round(hi, lo, sticky, e, a.neg);
}

/* a / PI (std.math.PI), one rounding; a != 0 */
X87_FN x87_t x87_div_pi(x87_t a)
{
    if (a.m == 0) return a;
    const uint64_t b = X87_PI_M;
    /* numerator N = a.m * 2^64 (a.m < b: quotient in [2^63, 2^64)) or a.m * 2^63 (a.m >= b), Q = floor(N / b) */
    const int big = a.m >= b;
    const uint64_t nh = big ? a.m >> 1 : a.m, nl = big ? a.m << 63 : 0;
    /* estimate with R = floor(2^127 / b) = 0xA2F9836E4E441529: Q0 = floor(nh * R / 2^63) underestimates Q by a
     * few units at most; the remainder loop makes it exact */
    const uint64_t R = 0xA2F9836E4E441529ull;
    const uint64_t ph = X87_MULHI(nh, R), pl = nh * R;
    uint64_t q = (ph << 1) | (pl >> 63);
    /* r = N - q * b (fits 64 bits + a little: q <= Q) */
    uint64_t th = X87_MULHI(q, b), tl = q * b;
    uint64_t rl = nl - tl, rh = nh - th - (nl < tl);
    /* (q never overestimates: R and the dropped low word both round down; it is short by at most 3.
     * The count is bounded anyway: a device loop must not be able to spin.) */
    for (int it = 0; it < 8 && (rh != 0 || rl >= b); ++it) {
        const uint64_t o = rl;
        rl -= b;
        rh -= (o < b);
        q += 1;
    }
    /* q: 64-bit quotient (top bit set), remainder rl < b: round to nearest even on (rl / b) vs 1/2 */
    const uint64_t twice = rl << 1;
    const int over = (int)(rl >> 63);                 /* 2*rl overflowed 64 bits: certainly > b */
    const int gt = over | (twice > b), eq = !over & (twice == b);
    int e = a.e - 1 - (big ? 0 : 1);                  /* PI = m * 2^(1 - 63): exponent 1 */
    uint64_t m = q;
    if (gt | (eq & (int)(q & 1))) {
        m += 1;
        if (m == 0) { m = 1ull << 63; e += 1; }
    }
    return x87_make(m, e, a.neg);
}

/* x87 -> double, round to nearest even (normal range or zero only) */
X87_FN double x87_to_double(x87_t a)
{
    if (a.m == 0) return X87_DOUBLE((uint64_t)a.neg << 63);
    uint64_t f = a.m >> 11;
    const uint64_t rem = a.m & 0x7ff;
    int e = a.e;
    if (rem > 0x400 || (rem == 0x400 && (f & 1))) {
        f += 1;
        if (f >> 53) { f >>= 1; e += 1; }
    }
    return X87_DOUBLE(((uint64_t)a.neg << 63) | ((uint64_t)(e + 1023) << 52) | (f & 0xfffffffffffffull));
}

/* info.u = (PI + angle) / (2 * PI) */
X87_FN double x87_sphere_u(double angle)
{
    x87_t s = x87_add(x87_make(X87_PI_M, 1, 0), x87_from_double(angle));
    x87_t q = x87_div_pi(s);     /* (PI + angle) / PI ... */
    if (q.m) q.e -= 1;           /* ... / 2: exact, and division by 2*PI rounds the same way */
    return x87_to_double(q);
}

/* info.v = 1.0 - (PI / 2 + asin_value) / PI */
X87_FN double x87_sphere_v(double asin_value)
{
    x87_t s = x87_add(x87_make(X87_PI_M, 0, 0), x87_from_double(asin_value)); /* PI / 2: exponent 0 */
    x87_t q = x87_div_pi(s);
    q.neg ^= 1;
    return x87_to_double(x87_add(x87_make(1ull << 63, 0, 0), q));
}

#endif /* C2RT_X87_H */
