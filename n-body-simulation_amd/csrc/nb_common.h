// nb_common.h -- internal declarations shared by the HIP sources of libnbody_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>

#include "nbody.h"

#include "nb_error.h"

namespace nb {

// ---- positions along a Hilbert curve (host and device) ----
// Bodies are ordered along Hilbert curves -- the tree's keys level by level (nb_bh.hip, nb_bh3.hip) and
// the slots of the exchange order (nb_handle.h, nb_refresh.hip): the cells of every level are contiguous
// ranges of the order, as with Z order, and consecutive cells are always neighbours in space (in Z order
// a quarter of them are not), so runs of consecutive bodies fill tight boxes.
//
// 2-D: through the four quadrants of a cell {TL, TR, BL, BR} = {0, 1, 2, 3} (2 * lower half + right
// half): entry 4 * orientation + quadrant of HILBERT_DIGIT is the quadrant's position along the curve,
// of HILBERT_NEXT the orientation inside that quadrant (orientation = swap x and y | invert both << 1, 0 at the root).
constexpr uint32_t HILBERT_DIGIT = 0x1e36b49cu, HILBERT_NEXT = 0xb3da640du;

// Index along the curve of the cell (x, y) of a 2^bits x 2^bits grid (y counted downwards or upwards: either way a Hilbert curve).
__host__ __device__ inline uint64_t hilbert2_index(uint32_t x, uint32_t y, int bits)
{
    uint64_t k = 0;
    uint32_t curve = 0u;
    for (int l = bits - 1; l >= 0; --l) {
        const uint32_t at = 2u * (4u * curve + ((((y >> l) & 1u) << 1) | ((x >> l) & 1u)));
        k = (k << 2) | (uint64_t)((HILBERT_DIGIT >> at) & 3u);
        curve = (HILBERT_NEXT >> at) & 3u;
    }
    return k;
}

// 3-D: index along the curve of the cell (c0, c1, c2) of a grid of 2^bits cells per axis, three bits per level,
// the top level first.  Skilling's transform of the three bit strings (Programming the Hilbert curve, AIP
// Conf. Proc. 707, 2004): a bit of the result depends on the bits of its own level and above only, so
// cutting the index at any level gives the curve through that level's cells.
__host__ __device__ inline uint64_t hilbert3_index(uint32_t c0, uint32_t c1, uint32_t c2, int bits)
{
    if (bits <= 0)
        return 0ull;
    uint32_t c[3] = {c0, c1, c2};
    const uint32_t top = 1u << (bits - 1);
    for (uint32_t q = top; q > 1u; q >>= 1) {
        const uint32_t low = q - 1u;
        for (int d = 0; d < 3; ++d) {
            if (c[d] & q) {
                c[0] ^= low;
            } else {
                const uint32_t t = (c[0] ^ c[d]) & low;
                c[0] ^= t;
                c[d] ^= t;
            }
        }
    }
    c[1] ^= c[0];
    c[2] ^= c[1];
    uint32_t t = 0u;
    for (uint32_t q = top; q > 1u; q >>= 1)
        if (c[2] & q)
            t ^= q - 1u;
    c[0] ^= t;
    c[1] ^= t;
    c[2] ^= t;
    uint64_t k = 0;
    for (int l = bits - 1; l >= 0; --l)
        k = (k << 3) | (uint64_t)((((c[0] >> l) & 1u) << 2) | (((c[1] >> l) & 1u) << 1) | ((c[2] >> l) & 1u));
    return k;
}

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
