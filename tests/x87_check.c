/* x87_check.c — chess2rt_amd/csrc/x87.h (integer emulation of the reference's x87 `real` arithmetic in
 * Sphere.intersect's u, v) against the compiler's own long double arithmetic on this x86-64 host.
 * usage: x87_check [n_random]   -> prints "mismatches 0" and exits 0 when every case agrees bit for bit */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "../chess2rt_amd/csrc/x87.h"

#define PIL 3.141592653589793238462643383279502884L

static double ref_u(double angle) { volatile long double s = PIL + (long double)angle; volatile long double q = s / (2 * PIL); return (double)q; }
static double ref_v(double as)
{
    volatile long double s = PIL / 2 + (long double)as;
    volatile long double q = s / PIL;
    volatile long double r = 1.0L - q;
    return (double)r;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd(void) { uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

static unsigned long long bad = 0, total = 0;
static void check(double a)
{
    if (fabs(a) <= 3.14159265358979323846) { /* atan2 range */
        double e = ref_u(a), g = x87_sphere_u(a);
        total++;
        if (x87_bits_(e) != x87_bits_(g)) { if (bad < 10) printf("u(%a): x87 %a emulated %a\n", a, e, g); bad++; }
    }
    if (fabs(a) <= 1.57079632679489661923) { /* asin range */
        double e = ref_v(a), g = x87_sphere_v(a);
        total++;
        if (x87_bits_(e) != x87_bits_(g)) { if (bad < 10) printf("v(%a): x87 %a emulated %a\n", a, e, g); bad++; }
    }
}

int main(int argc, char **argv)
{
    _Static_assert(__LDBL_MANT_DIG__ == 64, "needs x87 long double");
    const long n = argc > 1 ? atol(argv[1]) : 2000000;
    const double pi = 3.14159265358979323846;
    /* special values: zero, +-pi, +-pi/2, their neighbours, powers of two across the exponent range
     * (ties of the 64-bit rounding sit where the angle's last bit is half an x87 ulp), subnormals */
    const double sp[] = {0.0, -0.0, pi, -pi, pi / 2, -pi / 2, nextafter(pi, 0), nextafter(-pi, 0), nextafter(pi / 2, 0), nextafter(-pi / 2, 0),
                         1.0, -1.0, 0.5, -0.5, 3.0, -3.0, 1.5, -1.5, 5e-324, -5e-324, 2.2250738585072014e-308, 1e-300, -1e-300};
    for (unsigned i = 0; i < sizeof sp / sizeof *sp; ++i) check(sp[i]);
    for (int e = -1074; e <= 1; ++e)
        for (int k = 0; k < 64; ++k) {
            double base = ldexp(1.0, e);
            uint64_t bits = x87_bits_(base) + (rnd() & 0xfffffffffffffull) * (k != 0);
            if (k & 1) bits |= 1; /* odd last bit: the tie patterns */
            double a = x87_double_(bits);
            check(a); check(-a);
        }
    for (long i = 0; i < n; ++i) {
        /* uniform in value over (-pi, pi), and log-uniform in magnitude */
        double a = ((double)(rnd() >> 11) * 0x1p-53 * 2 - 1) * pi;
        check(a);
        double m = ldexp((double)(rnd() >> 11) * 0x1p-53 + 1.0, -(int)(rnd() % 80));
        check((rnd() & 1) ? m : -m);
    }
    printf("cases %llu mismatches %llu\n", total, bad);
    return bad != 0;
}
