/* TEST INFRASTRUCTURE -- CPU oracle for the DCNv2 kernels; see dcn_ref.inc.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load the library built from this file.  Exports dcnf_* (float, the
 * reference's type) and dcnd_* (double, gradient checks). */
#include <math.h>
#include <stddef.h>

#define REAL float
#define FN(name) dcnf_##name
#include "dcn_ref.inc"
#undef REAL
#undef FN

#define REAL double
#define FN(name) dcnd_##name
#include "dcn_ref.inc"
#undef REAL
#undef FN
