// v_cndmask_b32 issue cost against where its mask lives and how recently it was written (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define KERNEL(name, ninstr, body)                                                                    \
    __global__ void __launch_bounds__(1024) name(unsigned long long *out, int iters, double seed)     \
    {                                                                                                 \
        double a = seed + threadIdx.x, b = seed * 3 + 1;                                              \
        double x0, x1; int i0 = threadIdx.x, i1 = 3, j0 = 0, j1 = 0, j2 = 0, j3 = 0, j4 = 0, j5 = 0, j6 = 0;         \
        for (int i = 0; i < iters; ++i) {                                                             \
            REP8(body)                                                                                \
        }                                                                                             \
        if (j0 + j1 + j2 + j3 + j4 + j5 + j6 == 123456789) out[1000] = 1;                             \
    }                                                                                                 \
    static const int name##_n = ninstr;
#define OUTS : "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3), "+v"(j4), "+v"(j5), "+v"(j6) : "v"(a), "v"(b), "v"(i0), "v"(i1)
KERNEL(k_b, 4, asm volatile("v_cmp_lt_f64 vcc, %7, %8\nv_cndmask_b32 %0, %9, %10, vcc\nv_cndmask_b32 %1, %9, %10, vcc\nv_cndmask_b32 %2, %9, %10, vcc" OUTS : "vcc");)
KERNEL(k_c, 8, asm volatile("v_cmp_lt_f64 vcc, %7, %8\nv_cndmask_b32 %0, %9, %10, vcc\nv_cndmask_b32 %1, %9, %10, vcc\nv_cndmask_b32 %2, %9, %10, vcc\nv_cndmask_b32 %3, %9, %10, vcc\nv_cndmask_b32 %4, %9, %10, vcc\nv_cndmask_b32 %5, %9, %10, vcc\nv_cndmask_b32 %6, %9, %10, vcc" OUTS : "vcc");)
KERNEL(k_d, 4, asm volatile("v_cmp_lt_f64 s[20:21], %7, %8\nv_cndmask_b32 %0, %9, %10, s[20:21]\nv_cndmask_b32 %1, %9, %10, s[20:21]\nv_cndmask_b32 %2, %9, %10, s[20:21]" OUTS : "s20", "s21");)
KERNEL(k_d7, 8, asm volatile("v_cmp_lt_f64 s[20:21], %7, %8\nv_cndmask_b32 %0, %9, %10, s[20:21]\nv_cndmask_b32 %1, %9, %10, s[20:21]\nv_cndmask_b32 %2, %9, %10, s[20:21]\nv_cndmask_b32 %3, %9, %10, s[20:21]\nv_cndmask_b32 %4, %9, %10, s[20:21]\nv_cndmask_b32 %5, %9, %10, s[20:21]\nv_cndmask_b32 %6, %9, %10, s[20:21]" OUTS : "s20", "s21");)
KERNEL(k_e, 4, asm volatile("s_mov_b64 vcc, s[20:21]\nv_cndmask_b32 %0, %9, %10, vcc\nv_cndmask_b32 %1, %9, %10, vcc\nv_cndmask_b32 %2, %9, %10, vcc" OUTS : "vcc", "s20", "s21");)
KERNEL(k_f, 4, asm volatile("v_cndmask_b32 %0, %9, %10, vcc\nv_cndmask_b32 %1, %9, %10, vcc\nv_cndmask_b32 %2, %9, %10, vcc\nv_cndmask_b32 %3, %9, %10, vcc" OUTS : "vcc");)
KERNEL(k_g, 4, asm volatile("v_cmp_lt_f64 vcc, %7, %8\nv_add_f64 %7, %7, %8\nv_add_f64 %8, %7, %8\nv_cndmask_b32 %0, %9, %10, vcc" : "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3), "+v"(j4), "+v"(j5), "+v"(j6), "+v"(a), "+v"(b) : "v"(i0), "v"(i1) : "vcc");)
// a select of a double done as v_cndmask pairs vs arithmetic blends
KERNEL(k_h, 3, asm volatile("v_cmp_lt_f64 vcc, %7, %8\nv_cndmask_b32 %0, %9, %10, vcc\nv_cndmask_b32 %1, %10, %9, vcc" OUTS : "vcc");)
// VOP3 64-bit ops with abs/neg modifiers, v_cmp with exec write (v_cmpx)
KERNEL(k_x, 2, asm volatile("v_cmpx_lt_f64 vcc, %7, %8\ns_mov_b64 exec, -1" OUTS : "vcc");)
KERNEL(k_y, 4, asm volatile("v_cmp_lt_f64 vcc, %7, %8\ns_and_saveexec_b64 s[20:21], vcc\nv_mov_b32 %0, %9\ns_mov_b64 exec, s[20:21]" OUTS : "vcc", "s20", "s21", "scc");)
KERNEL(k_z, 4, asm volatile("v_cmp_lt_f64 vcc, %7, %8\ns_and_saveexec_b64 s[20:21], vcc\ns_cbranch_execz 1f\nv_mov_b32 %0, %9\n1:\ns_mov_b64 exec, s[20:21]" OUTS : "vcc", "s20", "s21", "scc");)

KERNEL(k_m1, 4, asm volatile("v_cmp_lt_f64 vcc, %7, %8\nv_cndmask_b32_e64 %0, %9, %10, vcc\nv_cndmask_b32_e64 %1, %9, %10, vcc\nv_cndmask_b32_e64 %2, %9, %10, vcc" OUTS : "vcc");)
KERNEL(k_m2, 6, asm volatile("v_cmp_lt_f64 vcc, %[a], %[b]\nv_cndmask_b32 %[o0], %[p], %[q], vcc\nv_add_f64 %[x], %[a], %[b]\nv_cndmask_b32 %[o1], %[p], %[q], vcc\nv_add_f64 %[y], %[a], %[b]\nv_cndmask_b32 %[o2], %[p], %[q], vcc" : [o0] "+v"(j0), [o1] "+v"(j1), [o2] "+v"(j2), [x] "=&v"(x0), [y] "=&v"(x1) : [a] "v"(a), [b] "v"(b), [p] "v"(i0), [q] "v"(i1) : "vcc");)
KERNEL(k_m3, 4, asm volatile("v_cmp_lt_f64 vcc, %7, %8\nv_cndmask_b32 %0, 0, %10, vcc\nv_cndmask_b32 %1, 0, %10, vcc\nv_cndmask_b32 %2, 0, %10, vcc" OUTS : "vcc");)
KERNEL(k_m4, 4, asm volatile("v_cmp_lt_f64 vcc, %7, %8\nv_cndmask_b32 %0, %9, %10, vcc\ns_nop 0\nv_cndmask_b32 %1, %9, %10, vcc" OUTS : "vcc");)
KERNEL(k_m5, 6, asm volatile("v_cmp_lt_f64 vcc, %7, %8\nv_cndmask_b32 %0, %9, %10, vcc\nv_xor_b32 %3, %9, %10\nv_cndmask_b32 %1, %9, %10, vcc\nv_xor_b32 %4, %9, %10\nv_cndmask_b32 %2, %9, %10, vcc" OUTS : "vcc");)
KERNEL(k_m6, 4, asm volatile("v_cmp_lt_f64 vcc, %7, %8\nv_cndmask_b32 %0, %9, %9, vcc\nv_cndmask_b32 %1, %10, %10, vcc\nv_cndmask_b32 %2, %9, %9, vcc" OUTS : "vcc");)
KERNEL(k_m7, 4, asm volatile("v_cmp_lt_f64 vcc, %7, %8\nv_cndmask_b32 %0, %0, %10, vcc\nv_cndmask_b32 %1, %1, %10, vcc\nv_cndmask_b32 %2, %2, %10, vcc" OUTS : "vcc");)
KERNEL(k_m8, 4, asm volatile("v_cmp_lt_u32 vcc, %9, %10\nv_addc_co_u32 %0, vcc, %9, %10, vcc\nv_cmp_lt_u32 vcc, %9, %10\nv_addc_co_u32 %1, vcc, %9, %10, vcc" OUTS : "vcc");)
typedef void (*kern_t)(unsigned long long *, int, double);
struct Entry { const char *name; kern_t k; int n; };
int main()
{
    Entry es[] = {{"cmp->vcc, 3 cnd(vcc)", k_b, 4}, {"cmp->vcc, 7 cnd(vcc)", k_c, 8}, {"cmp->sgpr, 3 cnd(sgpr)", k_d, 4}, {"cmp->sgpr, 7 cnd(sgpr)", k_d7, 8},
                  {"s_mov vcc, 3 cnd(vcc)", k_e, 4}, {"4 cnd(vcc), vcc never written", k_f, 4}, {"cmp->vcc, 2 add_f64, cnd(vcc)", k_g, 4},
                  {"cmp->vcc, 2 cnd (f64 select)", k_h, 3}, {"v_cmpx + s_mov exec", k_x, 2}, {"cmp, saveexec, v_mov, restore", k_y, 4}, {"same + s_cbranch_execz", k_z, 5}, {"cmp->vcc, 3 cnd_e64(vcc)", k_m1, 4}, {"cmp, (cnd(vcc), add_f64) x3", k_m2, 6}, {"cmp, 3 cnd(vcc) src0=0", k_m3, 4}, {"cmp, cnd, s_nop, cnd", k_m4, 4}, {"cmp, (cnd(vcc), v_xor) x3", k_m5, 6}, {"cmp, 3 cnd(vcc) same srcs", k_m6, 4}, {"cmp, 3 cnd(vcc) src0=dst", k_m7, 4}, {"2x (cmp_u32, addc vcc)", k_m8, 4}};
    unsigned long long *out;
    hipMalloc(&out, 8192 * 8);
    const int iters = 100000;
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_b, dim3(256), dim3(1024), 0, 0, out, iters / 10, 1.0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%-34s %9s %9s   ns per GROUP (and per instruction) at 1 / 4 waves per SIMD; cycles = ns x 2.4\n", "pattern", "1 wave", "4 waves");
    for (auto &e : es) {
        double res[2];
        int wv[2] = {1, 4};
        for (int w = 0; w < 2; ++w) {
            const int threads = 64 * 4 * wv[w];
            hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, out, iters / 20, 1.0);
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, out, iters, 1.0);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            res[w] = (double)ms * 1e6 / ((double)iters * 8.0 * wv[w]);
        }
        printf("%-34s %9.3f %9.3f   cycles/group: %6.2f %6.2f   per instr (%d): %5.2f %5.2f\n", e.name, res[0], res[1], res[0] * 2.4, res[1] * 2.4, e.n, res[0] * 2.4 / e.n, res[1] * 2.4 / e.n);
    }
    return 0;
}
