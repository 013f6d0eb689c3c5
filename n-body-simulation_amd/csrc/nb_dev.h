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
