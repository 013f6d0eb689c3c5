// nb_dev.h -- device helpers shared by the Barnes-Hut sources (2-D and 3-D).
#pragma once

#include <hip/hip_runtime.h>

namespace nb {

static __device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Exclusive scan of one int per thread over a 256-thread block; total = block sum.
static __device__ __forceinline__ int block_excl_scan_256(int v, int &total)
{
    __shared__ int wsum[4];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o);
        if (lane >= o)
            incl += t;
    }
    __syncthreads();  // protects wsum against the previous call's readers
    if (lane == 63)
        wsum[w] = incl;
    __syncthreads();
    int base = 0;
    int tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int s = wsum[k];
        if (k < w)
            base += s;
        tot += s;
    }
    total = tot;
    return base + incl - v;
}

static __device__ __forceinline__ int block_sum_256(int v)
{
    __shared__ int wsum2[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o);
    __syncthreads();
    if (lane_id() == 0)
        wsum2[threadIdx.x >> 6] = v;
    __syncthreads();
    return wsum2[0] + wsum2[1] + wsum2[2] + wsum2[3];
}

// A double moved between the lanes of a row of 16 by the VALU's data-parallel primitives (no trip
// through the LDS crossbar as __shfl makes): CTRL 0x110 | k takes the value k lanes below in the
// row; lanes without a source in their row get 0.
template <int CTRL>
static __device__ __forceinline__ double row_move(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// The value one lane holds, for all lanes (the lane is the same for the whole wave).
static __device__ __forceinline__ double lane_value(double v, int l)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

static __device__ __forceinline__ uint64_t lane_value(uint64_t v, int l)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}

// Inclusive scan over the runs of a wave: on return lane l holds the sums of s[] and the maximum of
// m over the lanes start .. l, where start (<= l) is the first lane of l's run.  All 64 lanes must be
// active.  The scan runs inside the rows of 16 lanes first (4 steps of row_move), then hands the
// end of each row to the next (3 steps).  The order of the additions depends on the lane numbers
// alone.
template <int NSUM>
static __device__ __forceinline__ void wave_run_scan(double (&s)[NSUM], double &m, int lane, int start)
{
    const int dist = lane - start;
    const int in_row = dist < (lane & 15) ? dist : (lane & 15);
#define NB_ROW_STEP(O)                             \
    {                                              \
        double t[NSUM];                            \
        _Pragma("unroll") for (int v = 0; v < NSUM; ++v) t[v] = row_move<0x110 | O>(s[v]); \
        const double tm = row_move<0x110 | O>(m);  \
        if (O <= in_row) {                         \
            _Pragma("unroll") for (int v = 0; v < NSUM; ++v) s[v] += t[v]; \
            m = fmax(m, tm);                       \
        }                                          \
    }
    NB_ROW_STEP(1)
    NB_ROW_STEP(2)
    NB_ROW_STEP(4)
    NB_ROW_STEP(8)
#undef NB_ROW_STEP
#pragma unroll
    for (int r = 1; r < 4; ++r) {
        // the end of row r - 1 is complete by now; the lanes of row r whose run began before the row take it
        double e[NSUM];
#pragma unroll
        for (int v = 0; v < NSUM; ++v)
            e[v] = lane_value(s[v], 16 * r - 1);
        const double em = lane_value(m, 16 * r - 1);
        if ((lane >> 4) == r && start < 16 * r) {
#pragma unroll
            for (int v = 0; v < NSUM; ++v)
                s[v] = e[v] + s[v];
            m = fmax(em, m);
        }
    }
}

// chunk_info word of the leaf pass (per 64 sorted bodies): first lane that starts a run of equal
// keys (64: none) | lane where the run that goes on into the next chunk starts << 8, or:
constexpr unsigned LEAF_NO_OWNER = 0xffu;  // no run of this chunk goes on into the next

// The radix sort (sort_pairs) orders key << 2 by eight 8-bit digits, lowest first.  Keys whose
// digits fill only the top `used_bits` bits of that word are zero below, and the passes over
// those zero digits would move nothing: this is the first pass that has work.
static __host__ __device__ __forceinline__ int sort_first_pass(int used_bits)
{
    const int p = (64 - used_bits) / 8;
    return p < 7 ? p : 7;
}

// Workgroups are dealt to the eight XCDs round-robin (blockIdx % 8 labels the blocks that share an
// XCD and its L2).  This bijection hands every XCD runs of NB_XCD_CHUNK consecutive logical blocks,
// so that blocks working on neighbouring data share an L2, while every XCD still gets its share of
// every part of the range (a contiguous eighth per XCD leaves the XCD that owns a dense region behind
// the others).  Placement is a speed matter only.
#ifndef NB_XCD_CHUNK
#define NB_XCD_CHUNK 32
#endif
static __device__ __forceinline__ uint32_t xcd_chunked_block(uint32_t orig, uint32_t nblocks)
{
    constexpr uint32_t C = NB_XCD_CHUNK;
    if (C <= 1u || orig >= nblocks - nblocks % (8u * C))
        return orig;  // the last, partial round of chunks stays as dealt
    const uint32_t x = orig & 7u, q = orig >> 3;
    return ((q / C) * 8u + x) * C + q % C;
}

}  // namespace nb
