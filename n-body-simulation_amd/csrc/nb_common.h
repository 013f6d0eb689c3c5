// nb_common.h -- internal declarations shared by the HIP sources of libnbody_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>

#include "nbody.h"

#include "nb_error.h"

namespace nb {

#define NB_HIP(call)                                                                            \
    do {                                                                                        \
        hipError_t nb_e_ = (call);                                                              \
        if (nb_e_ != hipSuccess)                                                                \
            return ::nb::fail(NB_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(nb_e_) + \
                                              " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

#define NB_TRY(call)            \
    do {                        \
        int nb_rc_ = (call);    \
        if (nb_rc_ != NB_OK)    \
            return nb_rc_;      \
    } while (0)

// Device array of `count` elements (plus a little slack, so that a kernel may read one vector
// past the end of the last element it needs).
template <typename T>
inline int dev_alloc(T *&p, size_t count)
{
    void *v = nullptr;
    hipError_t e = hipMalloc(&v, count * sizeof(T) + 256);
    if (e != hipSuccess)
        return fail(NB_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    p = (T *)v;
    return NB_OK;
}

// ---- internal units ----
// Lengths are divided by len_unit and masses by mass_unit, both powers of two, before they
// are rounded to float: scaling by a power of two is exact, and it keeps r^3 and m/r^3 of
// astronomical SI values (1e36, 1e-12) far from the float range limits.  Positions are taken
// relative to `origin` (the middle of the bodies' bounding box when they were set) first, so
// that a system far from the coordinate origin keeps all 24 bits for its own extent; forces
// depend on differences only.
struct Units {
    double origin[3];   // metres; subtracted in FP64 before scaling
    double len_unit;    // metres per internal length unit (2^k)
    double mass_unit;   // kilograms per internal mass unit (2^k)
    double acc_factor;  // multiply sum(m' d' / r'^3) by this to get m/s^2: G mass_unit / len_unit^2
};

// Scalars every force kernel needs, in internal units.
struct ForceConsts {
    float eps2;      // eps^2 / len_unit^2 (inside the square root)
    float eps3;      // eps^2 / len_unit^3 (the reference adds eps^2 to r^3)
    float cth;       // collision_th / len_unit
    int exact_denom; // 0: eps3 cannot change r^3 + eps3 in float for any pair outside the cutoff
    int eps2_negligible;  // 1: eps2 cannot change d^2 + eps2 in float for any pair outside the cutoff
    int precise_near;     // 1: the cutoff is so small against the system's extent that the float rounding of
                          // the coordinates (2^-24 of the extent) shows in the distance of the closest pairs
                          // that still contribute: the direct sum then adds the coordinates' low parts for
                          // runs near the cutoff (about a quarter more arithmetic there)
};

struct IntegrateArgs {
    double acc_factor;
    double dt;       // drift: x += v dt
    double dt_kick;  // kick: v += a dt_kick (dt, or dt/2 on the first leapfrog step)
    double inv_len_unit;
    double origin[3];  // Units::origin
    int do_integrate;  // 0: only store accelerations
    int static_fixed;  // 1: bodies that are not dynamic keep their position and velocity
};

// ---- direct sum (nb_direct.hip; kernels in nb_direct_impl.h) ----
// posm: {x, y, m, radius + collision_th} per body in internal units (float4), n records.
// Integrates targets [t0, t0+nt): reads posm_cur and pos_cur, writes vel/acc and the targets'
// entries of pos_next and posm_next (nothing the step reads is written by it).  part: scratch of jsplit*nt double2 when jsplit > 1.
// meta / meta_next: direct_meta_bytes(n) each, the bounding boxes of the source runs of posm_cur /
// posm_next; *meta_valid: on entry whether meta holds the boxes of posm_cur (computed first
// otherwise), on return whether the step left in meta_next the boxes of posm_next (a step that
// advances all bodies does) or, for a force evaluation, whether meta still holds those of posm_cur.
int direct_step(hipStream_t s, const float4 *posm_cur, float4 *posm_next, const float *rad, int n, int t0, int nt,
                const ForceConsts &fc, const IntegrateArgs &ia, const uint8_t *dyn, const double2 *pos_cur,
                double2 *pos_next, const double *radius, double collision_th, double2 *vel, double2 *acc,
                double2 *part, int jsplit, void *meta, void *meta_next, bool *meta_valid);
int direct_choose_jsplit(int n, int nt);
// Record arrays handed to direct_step hold direct_record_capacity(n) float4, the ones past n
// written by pad_posm (the kernel stages whole tiles).
size_t direct_record_capacity(int n);
int pad_posm(hipStream_t s, float4 *posm, int n);
// bytes of the per-run metadata scratch direct_step needs for n bodies
size_t direct_meta_bytes(int n);

// pos/mass/radius (double) -> posm records + scaled radius, bodies [i0, i1)
// n: bodies in all (the low parts of the coordinates live behind the records: DirectTraits<2>::lo_of).
int pack_posm(hipStream_t s, const double2 *pos, const double *mass, const double *radius, int n, int i0, int i1,
              const Units &units, double collision_th, float4 *posm, float *rad);
// A record array holds direct_record_floats4(n) float4: direct_record_capacity(n) records and,
// behind them, as many float2 {x, y} "low parts": what rounding the scaled FP64 coordinate to the
// record's float left over.  Runs near the cutoff add them so that the difference of two close
// bodies carries float precision of the difference, not of the coordinates.
size_t direct_record_floats4(int n);

// ---- diagnostics (nb_diag.hip) ----
// potential energy of the pairs outside the cutoff, FP64, O(n^2); *out is a device double
int potential_energy(hipStream_t s, const double2 *pos, const double *mass, const double *radius, int n, double gravity,
                     double eps, double collision_th, double *out);

int potential_energy3(hipStream_t s, const double *pos, const double *mass, const double *radius, int n, double gravity,
                      double eps, double collision_th, double *out);

}  // namespace nb
