// nb_bh_sort.hip -- the radix sort of the Barnes-Hut pipeline (2-D and 3-D keys, and the curve
// indices of the order renewal) and the compaction that lists a rank's targets.  The reference sorts
// nothing: its GPU program regroups the bodies level by level with atomics
// (src/gpu/BH/barnesHut_kernel.cu:245-288), its CPU program inserts them one by one
// (src/cpu/quadtree.cpp:73-144).
#include "nb_bh_internal.h"

namespace nb {

// ------------------------------------------------------------------ 3 radix sort

#ifndef NB_SORT_ITEMS
#define NB_SORT_ITEMS 8
#endif
constexpr int SORT_ITEMS = NB_SORT_ITEMS;
constexpr int SORT_TILE = 256 * SORT_ITEMS;
constexpr int SORT_BITS = 8;
static_assert(SORT_BINS == 1 << SORT_BITS, "nb_bh_internal.h");
constexpr int KEY_BITS = 2 * BH_KEY_LEVELS;
constexpr int SORT_PASSES = (KEY_BITS + SORT_BITS - 1) / SORT_BITS;

// The 62 key bits are sorted as the top of a 64-bit word (key << 2), eight bits a pass, so that the
// passes line up with the TOP of the key: keys computed to `key_levels` digits are zero in the low
// 64 - 2 key_levels bits of that word, and a pass whose digit lies wholly in there would move
// nothing and returns at once (8 levels -> two passes with work, 9..12 -> three); the first pass
// with work is left in cut[2] by the kernel that decides the key levels.  Pass p reads
// buffer p & 1 and writes the other, so the last pass always leaves the result in buffer 0 and
// the keys are written where the first pass with work reads them: the host launches the same
// kernels every step and never needs to know how many did anything.
static_assert(SORT_BITS == 8 && SORT_PASSES == 8, "sort_first_pass (nb_dev.h) assumes eight 8-bit passes");

__device__ __forceinline__ uint32_t sort_digit(uint64_t key, int pass)
{
    return (uint32_t)((key << 2) >> (pass * SORT_BITS)) & (SORT_BINS - 1);
}

__global__ __launch_bounds__(256) void k_sort_count(const uint64_t *keys, int n, int pass, const uint32_t *cut,
                                                    uint32_t *hist, int nblk)
{
    if (pass < (int)cut[2])
        return;
    __shared__ uint32_t h[SORT_BINS];
    h[threadIdx.x] = 0;
    __syncthreads();
    // Every thread takes SORT_ITEMS consecutive keys.  Keys that follow each other mostly share
    // the digit (the bodies arrive in a spatial order, and the upper passes see nearly sorted
    // keys), so a thread adds a whole run of equal digits with one LDS atomic, and a wave whose
    // keys all have one digit adds them with a single one: one atomic per key would serialise
    // on a handful of counters.
    const int i0 = blockIdx.x * SORT_TILE + threadIdx.x * SORT_ITEMS;
    uint32_t d[SORT_ITEMS];
    int cnt = n - i0;  // keys of this thread
    cnt = cnt < 0 ? 0 : (cnt > SORT_ITEMS ? SORT_ITEMS : cnt);
    if (cnt == SORT_ITEMS) {
        static_assert(SORT_ITEMS % 2 == 0, "two keys per 16-byte load");
        const ulonglong2 *src = (const ulonglong2 *)(keys + i0);  // i0 is a multiple of SORT_ITEMS
#pragma unroll
        for (int it = 0; it < SORT_ITEMS / 2; ++it) {
            const ulonglong2 v = src[it];
            d[2 * it] = sort_digit(v.x, pass);
            d[2 * it + 1] = sort_digit(v.y, pass);
        }
    } else {
#pragma unroll
        for (int it = 0; it < SORT_ITEMS; ++it)
            d[it] = it < cnt ? sort_digit(keys[i0 + it], pass) : 0u;
    }
    bool one_run = true;
#pragma unroll
    for (int it = 1; it < SORT_ITEMS; ++it)
        one_run = one_run && (it >= cnt || d[it] == d[0]);
    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d[0]);
    if (__builtin_amdgcn_ballot_w64(!(one_run && (cnt == 0 || d[0] == d0))) == 0ull) {
        // the whole wave holds one digit
        int c = cnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            c += __shfl_xor(c, o);
        if ((threadIdx.x & 63) == 0 && c > 0)
            atomicAdd(&h[d0], (uint32_t)c);
    } else if (cnt > 0) {
        uint32_t run_d = d[0], run = 1;
#pragma unroll
        for (int it = 1; it < SORT_ITEMS; ++it) {
            if (it >= cnt)
                break;
            if (d[it] == run_d) {
                run += 1;
            } else {
                atomicAdd(&h[run_d], run);
                run_d = d[it];
                run = 1;
            }
        }
        atomicAdd(&h[run_d], run);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

// Sort pass, step 2: the per-block counts of every digit turned into running offsets.  A digit's
// row (hist[d][0..nblk)) is cut into SORT_SEGS segments, one block each: exclusive scan of the
// segment in place, the segment's total into totals[d][segment]; the scatter adds the totals of
// the segments before its own.  (One block per whole row is a chain of dependent loads that more
// threads do not shorten: 62 us a pass at N = 50 M, whose rows hold 24 415 counts; 8 segments 35 us,
// 16 segments 21 us and 5 us more in the scatter, which reads a digit's segment totals.)
__global__ __launch_bounds__(256) void k_scan_rows(uint32_t *hist, int nblk, uint32_t *totals, int pass,
                                                   const uint32_t *cut)
{
    if (pass < (int)cut[2])
        return;
    const int seglen = (nblk + SORT_SEGS - 1) / SORT_SEGS;
    const int s0 = (int)blockIdx.y * seglen;
    const int s1 = s0 + seglen < nblk ? s0 + seglen : nblk;
    uint32_t *row = hist + (size_t)blockIdx.x * nblk;
    // every thread sums a stretch of consecutive counts, one block scan over the 256 sums, then
    // the stretch is rewritten with its running offsets
    const int per = (seglen + 255) / 256;
    const int beg = s0 + (int)threadIdx.x * per;
    const int end = beg + per < s1 ? beg + per : s1;
    int sum = 0;
    for (int i = beg; i < end; ++i)
        sum += (int)row[i];
    int total;
    uint32_t run = (uint32_t)block_excl_scan_256(sum, total);
    for (int i = beg; i < end; ++i) {
        const uint32_t v = row[i];
        row[i] = run;
        run += v;
    }
    if (threadIdx.x == 0)
        totals[blockIdx.x * SORT_SEGS + blockIdx.y] = (uint32_t)total;
}

// Stable scatter of one 8-bit digit.  Wave w of a block owns a contiguous chunk of
// 64*SORT_ITEMS keys and ranks them sub-round by sub-round against its own LDS counters
// (no block barrier inside), so the order inside a digit is the input order.
__global__ __launch_bounds__(256) void k_sort_scatter(const uint64_t *keys_in, const uint32_t *idx_in,
                                                      uint64_t *keys_out, uint32_t *idx_out, int n, int pass,
                                                      const uint32_t *cut, const uint32_t *hist, int nblk,
                                                      const uint32_t *totals)
{
    if (pass < (int)cut[2])
        return;
    __shared__ uint32_t wcnt[4][SORT_BINS];
    __shared__ uint32_t dbase[SORT_BINS];
    __shared__ uint32_t lstart[SORT_BINS];
    __shared__ uint64_t lkey[SORT_TILE];
    __shared__ uint32_t lval[SORT_TILE];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // packed: the values are the low 30 bits of the keys (no pass with work reads those bits); they
    // are moved with the keys, and the last pass takes the pairs apart again
    const bool packed = cut[5] != 0u;
    const bool unpack = packed && pass == SORT_PASSES - 1;
    // thread = digit: the digit's keys in all blocks, in the segments of the row before this block's own, and
    // in the blocks of its segment before this one -- asked for here, ahead of the keys, and used after the ranking
    const int myseg = (int)blockIdx.x / ((nblk + SORT_SEGS - 1) / SORT_SEGS);
    uint32_t digit_total = 0, segs_before = 0;
#pragma unroll
    for (int g = 0; g < SORT_SEGS; ++g) {
        const uint32_t v = totals[tid * SORT_SEGS + g];
        digit_total += v;
        segs_before += g < myseg ? v : 0u;
    }
    const uint32_t in_segment_before = hist[(size_t)tid * nblk + blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        wcnt[k][tid] = 0;
    __syncthreads();

    const int chunk = blockIdx.x * SORT_TILE + w * (64 * SORT_ITEMS);
    uint64_t key[SORT_ITEMS];
    uint32_t val[SORT_ITEMS];
    uint32_t rank[SORT_ITEMS];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    volatile uint32_t *mycnt = wcnt[w];
    // every load of the thread is asked for before the first is used: inside the ranking loop, whose wave
    // barriers order memory operations, each round would wait for its own trip to memory (eight in a row)
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const int i = chunk + it * 64 + lane;
        const bool valid = i < n;
        key[it] = valid ? keys_in[i] : ~0ull;
        val[it] = (valid && !packed) ? idx_in[i] : 0u;
    }
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const int i = chunk + it * 64 + lane;
        const bool valid = i < n;
        const uint32_t d = sort_digit(key[it], pass);
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < SORT_BITS; ++bit) {
            const bool one = (d >> bit) & 1u;
            const uint64_t bal = __ballot(one);
            peers &= one ? bal : ~bal;
        }
        const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
        const uint32_t cnt = (uint32_t)__popcll(peers);
        uint32_t prev = 0;
        if (valid)
            prev = mycnt[d];
        __builtin_amdgcn_wave_barrier();
        if (valid && before == 0)
            mycnt[d] = prev + cnt;
        __builtin_amdgcn_wave_barrier();
        rank[it] = prev + before;
    }
    __syncthreads();
    {
        // thread = digit: exclusive prefix over the four waves, the digit's place in the block's
        // own sorted order, and where the block's run of that digit starts in the output
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t c = wcnt[k][tid];
            wcnt[k][tid] = run;
            run += c;
        }
        int all;
        const int local_start = block_excl_scan_256((int)run, all);
        // start of digit `tid` in the output = keys with a smaller digit (scan of the 256 totals)
        const int before = block_excl_scan_256((int)digit_total, all);
        lstart[tid] = (uint32_t)local_start;
        dbase[tid] = (uint32_t)before + segs_before + in_segment_before - (uint32_t)local_start;
    }
    __syncthreads();
    // Keys go through LDS in the block's sorted order, so that consecutive threads write
    // consecutive output positions of a digit's run instead of 64 separate places per wave.
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const int i = chunk + it * 64 + lane;
        if (i < n) {
            const uint32_t d = sort_digit(key[it], pass);
            const uint32_t p = lstart[d] + wcnt[w][d] + rank[it];
            lkey[p] = key[it];
            if (!packed)
                lval[p] = val[it];
        }
    }
    __syncthreads();
    const int first = blockIdx.x * SORT_TILE;
    const int count = n - first < SORT_TILE ? n - first : SORT_TILE;
    for (int t = tid; t < count; t += 256) {
        const uint64_t k = lkey[t];
        const uint32_t d = sort_digit(k, pass);
        const uint32_t dst = dbase[d] + (uint32_t)t;
        keys_out[dst] = unpack ? k & ~(uint64_t)SORT_PACK_MASK : k;
        if (!packed)
            idx_out[dst] = lval[t];
        else if (unpack)
            idx_out[dst] = (uint32_t)k & SORT_PACK_MASK;
    }
}

int sort_blocks(int n) { return (n + SORT_TILE - 1) / SORT_TILE; }
size_t sort_hist_words(int n) { return (size_t)SORT_BINS * sort_blocks(n); }
size_t sort_totals_words() { return (size_t)SORT_BINS * SORT_SEGS; }

int sort_pairs(hipStream_t s, uint64_t *const keys[2], uint32_t *const idx[2], int n, const uint32_t *cut, uint32_t *hist,
               uint32_t *totals, int first_launch)
{
    const int nblk = sort_blocks(n);
    // first_launch > 0: cut[2] >= first_launch holds on the device -- the caller chose the key width
    // itself, or had k_bbox_final raise cut[2] to it and repairs the order afterwards (k_sort_fix);
    // the passes below it would return at once, at a launch each
    for (int pass = first_launch; pass < SORT_PASSES; ++pass) {
        const int in = pass & 1, out = in ^ 1;
        hipLaunchKernelGGL(k_sort_count, dim3(nblk), dim3(256), 0, s, keys[in], n, pass, cut, hist, nblk);
        hipLaunchKernelGGL(k_scan_rows, dim3(SORT_BINS, SORT_SEGS), dim3(256), 0, s, hist, nblk, totals, pass, cut);
        hipLaunchKernelGGL(k_sort_scatter, dim3(nblk), dim3(256), 0, s, keys[in], idx[in], keys[out], idx[out], n, pass, cut,
                           hist, nblk, totals);
    }
    static_assert(SORT_PASSES % 2 == 0, "the last pass must write buffer 0");
    NB_HIP(hipGetLastError());
    return NB_OK;
}

// When the keys of a step need sort passes below the ones the host launched for it (it only knew
// the keys of an earlier step: cut[7] < cut[2]), the pairs come out ordered by the key bits the
// launched passes cover -- runs of equal upper bits, inside which the input order is kept.  The
// thread of a run's first pair then sorts the run by the whole key, stably, by insertion: together
// that is the order the full sort gives.  Runs are bodies that share a cell at least four levels
// below the deepest the host expected, i.e. short; this is the rare step in which the bounding box
// has outgrown that by a factor of 16 since the host last looked, and the next steps are launched
// for the new depth.
__global__ __launch_bounds__(256) void k_sort_fix(uint64_t *keys, uint32_t *idx, int n, const uint32_t *cut)
{
    if (cut[7] >= cut[2])
        return;  // every pass the keys need was launched
    const int shift = (int)cut[2] * SORT_BITS - 2;  // the launched passes ordered key bits [shift, 62)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint64_t pre = keys[i] >> shift;
        // (a neighbouring run's thread may be moving its pairs meanwhile: they keep their upper bits)
        if (i > 0 && (keys[i - 1] >> shift) == pre)
            continue;  // not the first pair of a run
        int end = i + 1;
        while (end < n && (keys[end] >> shift) == pre)
            ++end;
        for (int a = i + 1; a < end; ++a) {
            const uint64_t ka = keys[a];
            const uint32_t va = idx[a];
            int at = a;
            while (at > i && keys[at - 1] > ka) {
                keys[at] = keys[at - 1];
                idx[at] = idx[at - 1];
                --at;
            }
            keys[at] = ka;
            idx[at] = va;
        }
    }
}

// The repair is almost never needed: a small grid keeps the launch that finds that out short.
constexpr int SORT_FIX_BLOCKS = 2048;

int bh_sort(hipStream_t s, BhBuffers &b, const BhStepArgs &a)
{
    b.sorted = 0;
    NB_TRY(sort_pairs(s, b.keys, b.idx, a.n, b.cut, b.hist, b.hist_totals, b.sort_first));
    if (b.sort_first > 0) {
        const int blocks = (a.n + 255) / 256;
        hipLaunchKernelGGL(k_sort_fix, dim3(blocks < SORT_FIX_BLOCKS ? blocks : SORT_FIX_BLOCKS), dim3(256), 0, s, b.keys[0],
                           b.idx[0], a.n, b.cut);
        NB_HIP(hipGetLastError());
    }
    return NB_OK;
}

// ---- targets of this rank (multi-GPU): sorted positions k whose body lies in [t0, t1) ----
// The list is cut into the traversal's groups of 64 entries, and every tile of TL_TILE sorted positions starts a
// group of its own (the entries left over in a tile's last group are TL_PAD): a group's targets then lie within
// one tile of the sorted order.  A rank's slots are a piece of the curve the bodies were ordered along when the
// slots were dealt, so nearly all of its targets are one run of the sorted order -- but the few that have crossed
// one of the curve's jumps since sit anywhere, and 64 of those in one group made a wave whose box was the whole
// system: it walked the tree for every lane by itself and the launch waited for it (236 of 240 us for a
// sixteenth of 5 M bodies).  One group per stray body costs a wave slot for the time of an ordinary wave.

constexpr int TL_ITEMS = 8;
constexpr int TL_TILE = 256 * TL_ITEMS;

// One launch: a block per tile counts its targets, takes room for its groups at the end of the list by one atomic
// add, and writes them.  The tiles' groups therefore come in any order -- every group is walked by a wave of its
// own, so the order is a matter of placement only.  `count` must be zero when the launch starts: the launch
// leaves `count_next`, the word the launch after it will use, at zero (the traversal that read it has finished).
__global__ __launch_bounds__(256) void k_tl_build(const uint32_t *idx, int n, uint32_t t0, uint32_t t1, uint32_t *tlist,
                                                  uint32_t *count, uint32_t *count_next)
{
    __shared__ uint32_t s_first;
    const int base = blockIdx.x * TL_TILE + threadIdx.x * TL_ITEMS;
    bool own[TL_ITEMS];
    int c = 0;
#pragma unroll
    for (int it = 0; it < TL_ITEMS; ++it) {
        const int k = base + it;
        own[it] = false;
        if (k < n) {
            const uint32_t o = idx[k];
            own[it] = o >= t0 && o < t1;
        }
        c += own[it] ? 1 : 0;
    }
    int tot;
    const int before = block_excl_scan_256(c, tot);
    const int padded = (tot + 63) & ~63;
    if (threadIdx.x == 0) {
        s_first = padded ? atomicAdd(count, (uint32_t)padded) : 0u;
        if (blockIdx.x == 0)
            *count_next = 0u;
    }
    __syncthreads();
    const int first = (int)s_first;
    int pos = first + before;
#pragma unroll
    for (int it = 0; it < TL_ITEMS; ++it)
        if (own[it])
            tlist[pos++] = (uint32_t)(base + it);
    if ((int)threadIdx.x < padded - tot)  // at most 63 entries
        tlist[first + tot + (int)threadIdx.x] = TL_PAD;
}

size_t target_list_blocks(int n) { return (size_t)(n + TL_TILE - 1) / TL_TILE + 1; }
size_t target_list_capacity(int n) { return (size_t)n + 64 * target_list_blocks(n); }
int target_list_groups(int n, int ntarget) { return (ntarget + 63) / 64 + (int)target_list_blocks(n); }

int bh_target_list(hipStream_t s, const uint32_t *sorted_idx, int n, int t0, int t1, uint32_t *tlist, uint32_t *tcount,
                   unsigned &calls, const uint32_t **count)
{
    const int nblk = (n + TL_TILE - 1) / TL_TILE;
    uint32_t *mine = tcount + (calls & 1u), *next = tcount + ((calls + 1u) & 1u);
    hipLaunchKernelGGL(k_tl_build, dim3(nblk), dim3(256), 0, s, sorted_idx, n, (uint32_t)t0, (uint32_t)t1, tlist, mine, next);
    NB_HIP(hipGetLastError());
    calls += 1u;
    *count = mine;
    return NB_OK;
}

}  // namespace nb
