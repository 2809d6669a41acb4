/* TEST INFRASTRUCTURE ONLY (oracle/_ref build).
 *
 * Interposes libm's sin, cos, pow and log in the reference harness with the correctly
 * rounded oracle_math.h versions, for the arguments the reference's path passes them:
 * phi = 2*pi*r1 in [0, 2*pi) (src/core/vec3.h:153-155), the phase of a noise texture and
 * the angle of a rotate_y (src/materials/textures/texture.h:81, src/objects/hittable.h:79-81;
 * |x| <= 2^30), an exponent of 5 (src/materials/material.h:129) and log of a positive number
 * (src/materials/volumes/constant_medium.h:40).  Any other argument is forwarded to glibc.
 *
 * Why: glibc 2.35's sin/cos return a value 1 ulp away from the correctly rounded one
 * in ~0.14 % of calls (and which calls depends on whether the CPU selects its FMA
 * variant), so "the reference's output" is only defined up to that noise.  With this
 * file linked in, the unmodified reference source is evaluated with a libm that has a
 * machine-independent definition, which the restatement and the GPU kernels share.
 * oracle/_ref/ref_harness_glibc is the same harness WITHOUT this file.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <math.h>
#include "../oracle_math.h"

typedef double (*fn1)(double);
typedef double (*fn2)(double, double);

double sin(double x) {
    if (x >= -0x1p30 && x <= 0x1p30) { double s, c; oracle_sincos(x, &s, &c); return s; }
    static fn1 next; if (!next) next = (fn1)dlsym(RTLD_NEXT, "sin");
    return next(x);
}
double cos(double x) {
    if (x >= -0x1p30 && x <= 0x1p30) { double s, c; oracle_sincos(x, &s, &c); return c; }
    static fn1 next; if (!next) next = (fn1)dlsym(RTLD_NEXT, "cos");
    return next(x);
}
double pow(double x, double y) {
    if (y == 5.0 && x == x && fabs(x) <= 1e60) return oracle_pow5(x);
    static fn2 next; if (!next) next = (fn2)dlsym(RTLD_NEXT, "pow");
    return next(x, y);
}
double log(double x) {
    if (x > 0.0 && x < INFINITY) return oracle_log(x);
    static fn1 next; if (!next) next = (fn1)dlsym(RTLD_NEXT, "log");
    return next(x);
}
