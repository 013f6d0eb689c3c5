// nb_direct3.hip -- the three-dimensional all-pairs step: the traits of nb_direct_impl.h for records
// {x, y, z, m} with the cutoff radius in an array of its own, and the host entry points declared in
// nb_dim3.h.  The reference is two-dimensional (src/cpu/vector.h holds x and y); this is its force
// law with dz added to the distance: DirectSum::calculateAcceleration / isCollide
// (src/cpu/directSum.cpp:24-72) and the update order of DirectSum::update (:74-79).
#include "nb_common.h"
#include "nb_dim3.h"

#ifndef NB_DS3_SUB
#define NB_DS3_SUB 256  // records per run with its own bounding box (128: 4.35 -> 4.33 T interactions/s at N = 1 M)
#endif

#include "nb_direct_impl.h"

namespace nb {

template <>
struct DirectTraits<3> {
    static constexpr int TPT = 2, TILE = 1024, SUB = NB_DS3_SUB, FLUSH = 32, NEAR_UNROLL = 8;
    static constexpr int SPLIT = TILE;  // a slice of the source range is a whole number of tiles
    static constexpr bool W_IN_RECORD = false;
    typedef float4 Lo;
    static __device__ __forceinline__ void coords(float4 r, float (&c)[3])
    {
        c[0] = r.x;
        c[1] = r.y;
        c[2] = r.z;
    }
    static __device__ __forceinline__ float mass(float4 r) { return r.w; }
    static __device__ __forceinline__ float cutoff(float4, const float *w, int j) { return w[j]; }
    static __device__ __forceinline__ void lo_parts(float4 l, float (&c)[3])
    {
        c[0] = l.x;
        c[1] = l.y;
        c[2] = l.z;
    }
    static __device__ __forceinline__ float4 with_coords(float4 r, const float (&c)[3]) { return make_float4(c[0], c[1], c[2], r.w); }
    static __device__ __forceinline__ float4 make_record(const float (&c)[3], float m, float) { return make_float4(c[0], c[1], c[2], m); }
    static __device__ __forceinline__ float4 make_lo(const float (&l)[3]) { return make_float4(l[0], l[1], l[2], 0.f); }
    static __device__ __forceinline__ float4 pad_record() { return make_float4(1.0e15f, 1.0e15f, 1.0e15f, 0.f); }
    // the low parts live behind the cap records of an array
    static __host__ __device__ float4 *lo_of(const float4 *posm, size_t cap) { return (float4 *)(posm + cap); }
};

size_t direct3_record_capacity(int n) { return ds_record_capacity<3>(n); }
size_t direct3_meta_bytes(int n) { return ds_meta_bytes<3>(n); }
int direct3_choose_jsplit(int n, int nt) { return ds_choose_jsplit<3>(n, nt); }

int pack_posm3(hipStream_t s, const double *pos, const double *mass, const double *radius, int n, const Units &units,
               double collision_th, float4 *posm, float *wcut, float *rad)
{
    NB_TRY(ds_pack<3>(s, pos, mass, radius, n, 0, n, units, collision_th, posm, wcut, rad));
    return ds_pad<3>(s, posm, wcut, n);
}

int pack_posm3_range(hipStream_t s, const double *pos, const double *mass, int n, int i0, int i1, const Units &units,
                     float4 *posm)
{
    return ds_pack<3>(s, pos, mass, nullptr, n, i0, i1, units, 0.0, posm, nullptr, nullptr);
}

int direct3_step(hipStream_t s, Direct3State &st, int n, int t0, int nt, const ForceConsts &fc, const IntegrateArgs &ia,
                 double collision_th)
{
    DirectArgs<3> a;
    const size_t cap = direct3_record_capacity(n);
    a.posm_cur = st.posm[st.cur];
    a.posm_next = st.posm[st.cur ^ 1];
    a.posl_cur = DirectTraits<3>::lo_of(a.posm_cur, cap);
    a.posl_next = DirectTraits<3>::lo_of(a.posm_next, cap);
    a.wcut = st.wcut;
    a.rad = st.rad;
    a.dyn = st.dyn;
    a.pos_cur = st.pos_cur;
    a.pos_next = st.pos_next;
    a.radius = st.radius;
    a.collision_th = collision_th;
    a.vel = st.vel;
    a.acc = st.acc;
    a.part = st.part;
    a.n = n;
    a.t0 = t0;
    a.nt = nt;
    a.fc = fc;
    a.ia = ia;
    const size_t mb = direct3_meta_bytes(n);
    return ds_launch<3>(s, a, st.jsplit, (uint8_t *)st.meta + st.cur * mb, (uint8_t *)st.meta + (st.cur ^ 1) * mb, &st.meta_valid,
                        "direct3_step");
}

}  // namespace nb
