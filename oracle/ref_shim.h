/*
 * TEST INFRASTRUCTURE ONLY — never linked into the product library.
 *
 * Force-included (gcc -include) in front of the reference's OpenCL C kernel
 * file so that gcc can compile it as gnu99 C on the host.  It only supplies
 * what the OpenCL C *language* provides implicitly:
 *   - the `ulong` type, the address-space / kernel qualifiers (no-ops on a CPU),
 *   - get_global_id(), which returns a work-item index the driver sets,
 *   - single-precision overloads of the math builtins (OpenCL's sqrt/fmin/...
 *     on float arguments are float functions; C's are double).
 * No reference source text lives here.
 */
#ifndef WBREF_SHIM_H
#define WBREF_SHIM_H

#include <math.h>
#include <stdbool.h>
#include <stdint.h>

typedef uint64_t ulong;

#define __kernel
#define __global
#define __constant

/* current work-item id, set by the driver before each call of the kernel */
extern __thread int wbref_gid;
static inline int get_global_id(int dim) { (void)dim; return wbref_gid; }

#define sqrt  sqrtf
#define fabs  fabsf
#define fmin  fminf
#define fmax  fmaxf
#define pow   powf
#define atan2 atan2f
#define asin  asinf

#endif
