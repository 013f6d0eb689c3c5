// microbench.hip -- measures the FP32 vector-ALU rates the direct-sum kernel is priced
// against on gfx950: v_fma_f32, v_pk_fma_f32, v_rsq_f32 and a mix shaped like one pair
// interaction, at 1/2/4/8 waves per SIMD.  Stand-alone diagnostic (not part of the library):
//   hipcc -O3 --offload-arch=gfx950 microbench.hip -o microbench && ./microbench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                           \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

constexpr int ITERS = 4096;

// 8 independent v_fma_f32 per iteration
__global__ void k_fma(float *out, float b, float c)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n\t"
                     "v_fma_f32 %1, %1, %8, %9\n\t"
                     "v_fma_f32 %2, %2, %8, %9\n\t"
                     "v_fma_f32 %3, %3, %8, %9\n\t"
                     "v_fma_f32 %4, %4, %8, %9\n\t"
                     "v_fma_f32 %5, %5, %8, %9\n\t"
                     "v_fma_f32 %6, %6, %8, %9\n\t"
                     "v_fma_f32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// 8 independent v_pk_fma_f32 per iteration (two FMAs each)
__global__ void k_pkfma(float *out, float b, float c)
{
    float2v bb = {b, b}, cc = {c, c};
    float t = threadIdx.x;
    float2v a0 = {t, t + 1}, a1 = {t + 2, t + 3}, a2 = {t + 4, t + 5}, a3 = {t + 6, t + 7};
    float2v a4 = {t + 8, t + 9}, a5 = {t + 10, t + 11}, a6 = {t + 12, t + 13}, a7 = {t + 14, t + 15};
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n\t"
                     "v_pk_fma_f32 %1, %1, %8, %9\n\t"
                     "v_pk_fma_f32 %2, %2, %8, %9\n\t"
                     "v_pk_fma_f32 %3, %3, %8, %9\n\t"
                     "v_pk_fma_f32 %4, %4, %8, %9\n\t"
                     "v_pk_fma_f32 %5, %5, %8, %9\n\t"
                     "v_pk_fma_f32 %6, %6, %8, %9\n\t"
                     "v_pk_fma_f32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(bb), "v"(cc));
    }
    float2v s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

// 8 independent v_rsq_f32 per iteration
__global__ void k_rsq(float *out, float b, float c)
{
    float a0 = threadIdx.x + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_rsq_f32 %0, %0\n\t"
                     "v_rsq_f32 %1, %1\n\t"
                     "v_rsq_f32 %2, %2\n\t"
                     "v_rsq_f32 %3, %3\n\t"
                     "v_rsq_f32 %4, %4\n\t"
                     "v_rsq_f32 %5, %5\n\t"
                     "v_rsq_f32 %6, %6\n\t"
                     "v_rsq_f32 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b + c;
}

// 1 v_rsq_f32 + 6 v_fma_f32 interleaved, 2 chains (does the transcendental overlap the FMAs?)
__global__ void k_mix(float *out, float b, float c)
{
    float a0 = threadIdx.x + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_rsq_f32 %0, %0\n\t"
                     "v_fma_f32 %1, %1, %8, %9\n\t"
                     "v_fma_f32 %2, %2, %8, %9\n\t"
                     "v_fma_f32 %3, %3, %8, %9\n\t"
                     "v_rsq_f32 %4, %4\n\t"
                     "v_fma_f32 %5, %5, %8, %9\n\t"
                     "v_fma_f32 %6, %6, %8, %9\n\t"
                     "v_fma_f32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// v_cmp + v_cndmask pairs (cost of the cutoff select), 4 pairs per iteration
__global__ void k_cmpsel(float *out, float b, float c)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_cmp_gt_f32 vcc, %0, %4\n\t"
                     "v_cndmask_b32 %0, %5, %0, vcc\n\t"
                     "v_cmp_gt_f32 vcc, %1, %4\n\t"
                     "v_cndmask_b32 %1, %5, %1, vcc\n\t"
                     "v_cmp_gt_f32 vcc, %2, %4\n\t"
                     "v_cndmask_b32 %2, %5, %2, vcc\n\t"
                     "v_cmp_gt_f32 vcc, %3, %4\n\t"
                     "v_cndmask_b32 %3, %5, %3, vcc"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                     : "v"(b), "v"(c)
                     : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}


// ---- issue costs of the instruction kinds the tree traversal is made of ----

// 8 independent vector compares that write a scalar register pair
__global__ void k_cmp_sgpr(float *out, float b, float c)
{
    float a0 = threadIdx.x;
    unsigned long long acc = 0;
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_cmp_gt_f32_e64 s[20:21], %1, %2\n\t"
                     "v_cmp_gt_f32_e64 s[22:23], %1, %3\n\t"
                     "v_cmp_gt_f32_e64 s[24:25], %2, %1\n\t"
                     "v_cmp_gt_f32_e64 s[26:27], %3, %1\n\t"
                     "v_cmp_gt_f32_e64 s[28:29], %1, %2\n\t"
                     "v_cmp_gt_f32_e64 s[30:31], %1, %3\n\t"
                     "v_cmp_gt_f32_e64 s[36:37], %2, %1\n\t"
                     "v_cmp_gt_f32_e64 s[38:39], %3, %1\n\t"
                     "s_mov_b64 %0, s[38:39]"
                     : "=s"(acc)
                     : "v"(a0), "v"(b), "v"(c)
                     : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s36", "s37",
                       "s38", "s39");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + (float)acc;
}

// 8 independent v_readfirstlane_b32
__global__ void k_rfl(float *out, float b, float c)
{
    float a0 = threadIdx.x;
    unsigned acc = 0;
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_readfirstlane_b32 s20, %1\n\t"
                     "v_readfirstlane_b32 s21, %2\n\t"
                     "v_readfirstlane_b32 s22, %3\n\t"
                     "v_readfirstlane_b32 s23, %1\n\t"
                     "v_readfirstlane_b32 s24, %2\n\t"
                     "v_readfirstlane_b32 s25, %3\n\t"
                     "v_readfirstlane_b32 s26, %1\n\t"
                     "v_readfirstlane_b32 s27, %2\n\t"
                     "s_mov_b32 %0, s27"
                     : "=s"(acc)
                     : "v"(a0), "v"(b), "v"(c)
                     : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + (float)acc;
}

// 8 independent scalar 64-bit logic operations
__global__ void k_salu(float *out, float b, float c)
{
    unsigned long long acc = 0;
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("s_and_b64 s[20:21], exec, exec\n\t"
                     "s_or_b64 s[22:23], exec, exec\n\t"
                     "s_and_b64 s[24:25], exec, exec\n\t"
                     "s_or_b64 s[26:27], exec, exec\n\t"
                     "s_and_b64 s[28:29], exec, exec\n\t"
                     "s_or_b64 s[30:31], exec, exec\n\t"
                     "s_and_b64 s[36:37], exec, exec\n\t"
                     "s_or_b64 %0, exec, s[36:37]"
                     : "=s"(acc)
                     :
                     : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s36", "s37",
                       "scc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc + c;
}

// 4 v_fma_f32 and 4 scalar operations interleaved, all independent: do they share an issue slot?
__global__ void k_valu_salu(float *out, float b, float c)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    unsigned long long acc = 0;
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f32 %0, %0, %5, %6\n\t"
                     "s_and_b64 s[20:21], exec, exec\n\t"
                     "v_fma_f32 %1, %1, %5, %6\n\t"
                     "s_or_b64 s[22:23], exec, exec\n\t"
                     "v_fma_f32 %2, %2, %5, %6\n\t"
                     "s_and_b64 s[24:25], exec, exec\n\t"
                     "v_fma_f32 %3, %3, %5, %6\n\t"
                     "s_or_b64 %4, exec, s[24:25]"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=s"(acc)
                     : "v"(b), "v"(c)
                     : "s20", "s21", "s22", "s23", "s24", "s25", "scc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (float)acc;
}

// a dependent chain crossing between the units, as one tree-node visit does:
// compare -> scalar logic -> scalar compare -> branch (never taken) -> select, 2 per iteration
__global__ void k_cross(float *out, float b, float c)
{
    float a0 = threadIdx.x, a1 = a0 + 1;
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %2\n\t"
                     "s_and_b64 s[20:21], s[20:21], exec\n\t"
                     "s_cmp_eq_u64 s[20:21], 0\n\t"
                     "s_cbranch_scc1 1f\n\t"
                     "v_cndmask_b32_e64 %0, %3, %0, s[20:21]\n\t"
                     "1:\n\t"
                     "v_cmp_gt_f32_e64 s[22:23], %1, %2\n\t"
                     "s_and_b64 s[22:23], s[22:23], exec\n\t"
                     "s_cmp_eq_u64 s[22:23], 0\n\t"
                     "s_cbranch_scc1 2f\n\t"
                     "v_cndmask_b32_e64 %1, %3, %1, s[22:23]\n\t"
                     "2:"
                     : "+v"(a0), "+v"(a1)
                     : "v"(b), "v"(c)
                     : "s20", "s21", "s22", "s23", "scc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1;
}

// 8 taken branches per iteration
__global__ void k_branch(float *out, float b, float c)
{
    float a0 = threadIdx.x;
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("s_branch 1f\n\t s_nop 0\n\t 1: s_branch 2f\n\t s_nop 0\n\t 2: s_branch 3f\n\t s_nop 0\n\t 3: s_branch 4f\n\t"
                     "s_nop 0\n\t 4: s_branch 5f\n\t s_nop 0\n\t 5: s_branch 6f\n\t s_nop 0\n\t 6: s_branch 7f\n\t s_nop 0\n\t"
                     "7: s_branch 8f\n\t s_nop 0\n\t 8: v_add_f32 %0, %0, %1"
                     : "+v"(a0)
                     : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + c;
}

// 8 LDS broadcast reads of 16 B per iteration, each waited for (the visit loop's access)
__global__ void k_ldsb(float *out, float b, float c)
{
    __shared__ float4 buf[256];
    buf[threadIdx.x] = make_float4(b, c, b, c);
    __syncthreads();
    float acc = 0.f;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 v;
            const unsigned addr = (unsigned)(size_t)&buf[(i + k) & 255];
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
            acc += v.x;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// 8 LDS broadcast reads per iteration issued back to back, one wait: throughput by access width
template <int WIDTH>
__global__ void k_ldsw(float *out, float b, float c)
{
    __shared__ float4 buf[256];
    buf[threadIdx.x] = make_float4(b, c, b, c);
    __syncthreads();
    float acc = 0.f;
    const unsigned base = (unsigned)(size_t)&buf[0];
    for (int i = 0; i < ITERS; ++i) {
        const unsigned addr = base + ((i & 15) << 7);
        typedef float f4 __attribute__((ext_vector_type(4)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        if (WIDTH == 16) {
            f4 v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\t"
                         "ds_read_b128 %3, %8 offset:48\n\tds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\t"
                         "ds_read_b128 %6, %8 offset:96\n\tds_read_b128 %7, %8 offset:112\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7)
                         : "v"(addr));
            acc += v0.x + v1.x + v2.x + v3.x + v4.x + v5.x + v6.x + v7.x;
        } else if (WIDTH == 8) {
            f2 v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:16\n\tds_read_b64 %2, %8 offset:32\n\t"
                         "ds_read_b64 %3, %8 offset:48\n\tds_read_b64 %4, %8 offset:64\n\tds_read_b64 %5, %8 offset:80\n\t"
                         "ds_read_b64 %6, %8 offset:96\n\tds_read_b64 %7, %8 offset:112\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7)
                         : "v"(addr));
            acc += v0.x + v1.x + v2.x + v3.x + v4.x + v5.x + v6.x + v7.x;
        } else {
            float v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:16\n\tds_read_b32 %2, %8 offset:32\n\t"
                         "ds_read_b32 %3, %8 offset:48\n\tds_read_b32 %4, %8 offset:64\n\tds_read_b32 %5, %8 offset:80\n\t"
                         "ds_read_b32 %6, %8 offset:96\n\tds_read_b32 %7, %8 offset:112\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7)
                         : "v"(addr));
            acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// ---- all-pairs inner loop with the squared distances from the matrix cores ----
// 32 targets x 32 sources per step: r^2 = |a|^2 + |b|^2 - 2 a.b as a rank-4 product on
// v_mfma_f32_32x32x2_f32 (two issues), then per pair rsq, cube and three accumulations on the
// vector ALU: 7 vector issue slots per pair instead of 10.
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_pairs_mfma(float *out, float b, float c)
{
    __shared__ float4 tile[1024];
    for (int k = threadIdx.x; k < 1024; k += 256)
        tile[k] = make_float4(b * (float)(k % 37) + 3.f, c * (float)(k % 41) + 5.f, 1.f, 0.f);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const bool hi = lane >= 32;
    const float ax = 0.01f * (lane & 31), ay = 0.02f * (lane & 31);
    const float A1 = hi ? 1.f : ax * ax + ay * ay, A2 = hi ? -2.f * ay : -2.f * ax;
    float S[16], FX[16], FY[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        S[r] = FX[r] = FY[r] = 0.f;
    for (int it = 0; it < ITERS / 8; ++it) {
#pragma unroll 2
        for (int t = 0; t < 32; ++t) {
            const float4 src = tile[t * 32 + (lane & 31)];
            const float bx = src.x - 1.5f, by = src.y - 2.5f;
            const float n2 = __builtin_fmaf(by, by, __builtin_fmaf(bx, bx, 1e-9f));
            const float B1 = hi ? n2 : 1.f, B2 = hi ? by : bx;
            f16v d = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            d = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B1, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x2f32(A2, B2, d, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float rinv = __builtin_amdgcn_rsqf(d[r]);
                const float sc = (rinv * rinv) * rinv;
                S[r] += sc;
                FX[r] = __builtin_fmaf(sc, bx, FX[r]);
                FY[r] = __builtin_fmaf(sc, by, FY[r]);
            }
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        acc += S[r] + FX[r] + FY[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <typename K>
static double run(K kernel, int blocks, float *out, int reps)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e-3 / reps;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    double clk = prop.clockRate * 1e3;
    printf("device %s CUs %d clock %.0f MHz\n", prop.name, cus, clk / 1e6);
    float *out;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * cus * 16));
    const int reps = 20;
    for (int wps = 1; wps <= 8; wps *= 2) {
        int blocks = cus * wps;  // 256-thread blocks: one wave per SIMD each
        double lanes = (double)blocks * 256;
        double t;
        t = run(k_fma, blocks, out, reps);
        double fma_rate = lanes * ITERS * 8 / t;
        t = run(k_pkfma, blocks, out, reps);
        double pk_rate = lanes * ITERS * 8 * 2 / t;
        t = run(k_rsq, blocks, out, reps);
        double rsq_rate = lanes * ITERS * 8 / t;
        t = run(k_mix, blocks, out, reps);
        double mix_rate = lanes * ITERS * 8 / t;
        t = run(k_cmpsel, blocks, out, reps);
        double sel_rate = lanes * ITERS * 8 / t;
        {
            const double simd_cycles = 1.0;  // per-SIMD instruction slots: time * clock / (ITERS * 8 * waves)
            (void)simd_cycles;
            struct { const char *name; double t; double per_iter; } rows[] = {
                {"v_cmp->sgpr", run(k_cmp_sgpr, blocks, out, reps), 8}, {"v_readfirstlane", run(k_rfl, blocks, out, reps), 8},
                {"salu b64", run(k_salu, blocks, out, reps), 8},        {"4 fma + 4 salu", run(k_valu_salu, blocks, out, reps), 8},
                {"cross chain x2", run(k_cross, blocks, out, reps), 2}, {"taken branch", run(k_branch, blocks, out, reps), 8},
                {"lds b128 bcast", run(k_ldsb, blocks, out, reps), 8},  {"v_fma", run(k_fma, blocks, out, reps), 8},
                {"lds b128 x8", run(k_ldsw<16>, blocks, out, reps), 8}, {"lds b64 x8", run(k_ldsw<8>, blocks, out, reps), 8},
                {"lds b32 x8", run(k_ldsw<4>, blocks, out, reps), 8}};
            printf("waves/SIMD %d, cycles per item per SIMD (at the nominal clock):", wps);
            for (auto &r : rows)
                printf("  %s %.2f", r.name, r.t * clk / (ITERS * r.per_iter * wps));
            printf("\n");
        }
        {
            const double tm = run(k_pairs_mfma, blocks, out, reps);
            printf("waves/SIMD %d: pairs with MFMA distances: %.2f T pairs/s\n", wps,
                   (double)blocks * 4 * (ITERS / 8) * 32 * 1024 / tm / 1e12);
        }
        printf("waves/SIMD %d: fma %.2f Tlane-op/s (%.1f TFLOP/s, %.1f lane-op/clk/CU) | pk_fma %.2f Tfma/s (%.1f TFLOP/s) | "
               "rsq %.2f T/s | mix(2rsq+6fma) %.2f Tinst/s | cmp+cndmask %.2f Tinst/s\n",
               wps, fma_rate / 1e12, 2 * fma_rate / 1e12, fma_rate / clk / cus, pk_rate / 1e12, 2 * pk_rate / 1e12,
               rsq_rate / 1e12, mix_rate / 1e12, sel_rate / 1e12);
    }
    CHECK(hipFree(out));
    return 0;
}
