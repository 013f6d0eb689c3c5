// nb_direct.hip -- the two-dimensional all-pairs step: the traits of nb_direct_impl.h for records
// {x, y, m, radius + collision_th} and the host entry points declared in nb_common.h.
// Reference: DirectSum::update (src/cpu/directSum.cpp:24-79), DirectSumTiledKernel
// (src/gpu/DS/directSum_kernel.cu:61-106).
#include "nb_common.h"

#ifndef NB_DS_TPT
#define NB_DS_TPT 2  // targets per thread
#endif
#ifndef NB_DS_UNROLL
#define NB_DS_UNROLL 16  // records per unrolled body of the near pair loop
#endif
#ifndef NB_DS_TILE
#define NB_DS_TILE 512  // records per LDS tile (8 KiB), two buffers
#endif
// (256 -> 128: the boxes are tighter and fewer pairs of a wave and a run count as near, 4.71 -> 4.84
// T interactions/s at N = 65 536, 5.62 -> 5.74 at 262 144, 5.89 -> 5.95 at 1 M; 64: 4.87 / 5.64 /
// 5.82, the per-run scalar work begins to show; 32: 4.64 / 4.98 / 4.98)
#ifndef NB_DS_SUB
#define NB_DS_SUB 128  // records per run with its own bounding box
#endif
#ifndef NB_DS_FLUSH
#define NB_DS_FLUSH 32  // pairs between FP32 -> FP64 folds
#endif

#include "nb_direct_impl.h"

namespace nb {

template <>
struct DirectTraits<2> {
    static constexpr int TPT = NB_DS_TPT, TILE = NB_DS_TILE, SUB = NB_DS_SUB, FLUSH = NB_DS_FLUSH, NEAR_UNROLL = NB_DS_UNROLL;
    static constexpr int SPLIT = SUB;  // a slice of the source range is a whole number of runs
    static constexpr bool W_IN_RECORD = true;
    typedef float2 Lo;
    static __device__ __forceinline__ void coords(float4 r, float (&c)[2])
    {
        c[0] = r.x;
        c[1] = r.y;
    }
    static __device__ __forceinline__ float mass(float4 r) { return r.z; }
    static __device__ __forceinline__ float cutoff(float4 r, const float *, int) { return r.w; }
    static __device__ __forceinline__ void lo_parts(float2 l, float (&c)[2])
    {
        c[0] = l.x;
        c[1] = l.y;
    }
    static __device__ __forceinline__ float4 with_coords(float4 r, const float (&c)[2]) { return make_float4(c[0], c[1], r.z, r.w); }
    static __device__ __forceinline__ float4 make_record(const float (&c)[2], float m, float w) { return make_float4(c[0], c[1], m, w); }
    static __device__ __forceinline__ float2 make_lo(const float (&l)[2]) { return make_float2(l[0], l[1]); }
    static __device__ __forceinline__ float4 pad_record() { return make_float4(1.0e15f, 1.0e15f, 0.f, __builtin_inff()); }
    // the low parts live behind the cap records of an array
    static __host__ __device__ float2 *lo_of(const float4 *posm, size_t cap) { return (float2 *)(posm + cap); }
};

size_t direct_record_capacity(int n) { return ds_record_capacity<2>(n); }
size_t direct_record_floats4(int n) { return direct_record_capacity(n) + direct_record_capacity(n) / 2; }
size_t direct_meta_bytes(int n) { return ds_meta_bytes<2>(n); }
int direct_choose_jsplit(int n, int nt) { return ds_choose_jsplit<2>(n, nt); }

int pad_posm(hipStream_t s, float4 *posm, int n) { return ds_pad<2>(s, posm, nullptr, n); }

int pack_posm(hipStream_t s, const double2 *pos, const double *mass, const double *radius, int n, int i0, int i1,
              const Units &units, double collision_th, float4 *posm, float *rad)
{
    return ds_pack<2>(s, (const double *)pos, mass, radius, n, i0, i1, units, collision_th, posm, nullptr, rad);
}

int direct_step(hipStream_t s, const float4 *posm_cur, float4 *posm_next, const float *rad, int n, int t0, int nt,
                const ForceConsts &fc, const IntegrateArgs &ia, const uint8_t *dyn, const double2 *pos_cur,
                double2 *pos_next, const double *radius, double collision_th, double2 *vel, double2 *acc,
                double2 *part, int jsplit, void *meta, void *meta_next, bool *meta_valid)
{
    DirectArgs<2> a;
    const size_t cap = direct_record_capacity(n);
    a.posm_cur = posm_cur;
    a.posm_next = posm_next;
    a.posl_cur = DirectTraits<2>::lo_of(posm_cur, cap);
    a.posl_next = DirectTraits<2>::lo_of(posm_next, cap);
    a.wcut = nullptr;
    a.rad = rad;
    a.dyn = dyn;
    a.pos_cur = (const double *)pos_cur;
    a.pos_next = (double *)pos_next;
    a.radius = radius;
    a.collision_th = collision_th;
    a.vel = (double *)vel;
    a.acc = (double *)acc;
    a.part = (double *)part;
    a.n = n;
    a.t0 = t0;
    a.nt = nt;
    a.fc = fc;
    a.ia = ia;
    return ds_launch<2>(s, a, jsplit, meta, meta_next, meta_valid, "direct_step");
}

}  // namespace nb
