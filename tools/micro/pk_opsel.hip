// Check of the v_pk_add_f32 operand-select / negate forms the Winograd transforms use (development tool): prints each form's
// two results beside the expected values.   hipcc --offload-arch=gfx950 -O3 -o var/pk_opsel tools/micro/pk_opsel.hip && var/pk_opsel
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define PK(NAME, MODS)                                                                               \
    __device__ __forceinline__ f32x2 NAME(const f32x2 a, const f32x2 b) {                            \
        f32x2 r;                                                                                     \
        asm("v_pk_add_f32 %0, %1, %2 " MODS : "=v"(r) : "v"(a), "v"(b));                             \
        return r;                                                                                    \
    }
PK(pk_add, "")
PK(pk_sub, "neg_lo:[0,1] neg_hi:[0,1]")
PK(pk_bfly, "op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]")                  // (a.lo + b.hi, a.lo - b.hi)
PK(pk_bfly_neg, "op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[1,1] neg_hi:[1,0]")  // (-a.lo - b.hi, -a.lo + b.hi)
PK(pk_col01, "op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1]")                  // (a.lo - b.lo, a.hi + b.lo)
PK(pk_col23, "op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]")     // (-a.hi + b.lo, a.hi - b.hi)
#define PKMOV(NAME, MODS)                                                                            \
    __device__ __forceinline__ f32x2 NAME(const f32x2 a, const f32x2 b) {                            \
        f32x2 r;                                                                                     \
        asm("v_pk_mov_b32 %0, %1, %2 " MODS : "=v"(r) : "v"(a), "v"(b));                             \
        return r;                                                                                    \
    }
PKMOV(pk_lo_lo, "op_sel:[0,0]")     // (a.lo, b.lo)
PKMOV(pk_hi_hi, "op_sel:[1,1]")     // (a.hi, b.hi)
PKMOV(pk_lo_hi, "op_sel:[0,1]")     // (a.lo, b.hi)
__global__ void k2(float* o) {
    const f32x2 a = {1.f + threadIdx.x, 10.f}, b = {100.f, 1000.f};
    f32x2 r[3] = {pk_lo_lo(a, b), pk_hi_hi(a, b), pk_lo_hi(a, b)};
    for (int i = 0; i < 3; ++i) { o[2 * i] = r[i].x; o[2 * i + 1] = r[i].y; }
}
__global__ void k(float* o) {
    const f32x2 a = {1.f + threadIdx.x, 10.f}, b = {100.f, 1000.f};
    f32x2 r[6] = {pk_add(a, b), pk_sub(a, b), pk_bfly(a, a), pk_bfly_neg(a, a), pk_col01(a, b), pk_col23(a, b)};
    for (int i = 0; i < 6; ++i) { o[2 * i] = r[i].x; o[2 * i + 1] = r[i].y; }
}
int main() {
    float* d; hipMalloc(&d, 64); k<<<1, 1>>>(d); float h[12]; hipMemcpy(h, d, 48, hipMemcpyDeviceToHost);
    const float a0 = 1, a1 = 10, b0 = 100, b1 = 1000;
    const float e[12] = {a0 + b0, a1 + b1, a0 - b0, a1 - b1, a0 + a1, a0 - a1, -a0 - a1, -a0 + a1, a0 - b0, a1 + b0, -a1 + b0, a1 - b1};
    const char* n[6] = {"add", "sub", "bfly", "bfly_neg", "col01", "col23"};
    int bad = 0;
    for (int i = 0; i < 6; ++i) {
        printf("%-9s got (%g, %g) expected (%g, %g)\n", n[i], h[2 * i], h[2 * i + 1], e[2 * i], e[2 * i + 1]);
        bad += h[2 * i] != e[2 * i] || h[2 * i + 1] != e[2 * i + 1];
    }
    k2<<<1, 1>>>(d); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    const float e2[6] = {a0, b0, a1, b1, a0, b1};
    const char* n2[3] = {"mov lo,lo", "mov hi,hi", "mov lo,hi"};
    for (int i = 0; i < 3; ++i) {
        printf("%-9s got (%g, %g) expected (%g, %g)\n", n2[i], h[2 * i], h[2 * i + 1], e2[2 * i], e2[2 * i + 1]);
        bad += h[2 * i] != e2[2 * i] || h[2 * i + 1] != e2[2 * i + 1];
    }
    printf(bad ? "MISMATCH\n" : "all forms as expected\n");
    return bad;
}
