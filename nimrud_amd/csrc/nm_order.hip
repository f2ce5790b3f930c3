// nm_order.hip - the spatial order of a cloud: compact Z-order keys of the finest lattice, a radix sort of our
// own, and the coordinates carried into sorted order by the sort's last pass.
//
// reference code replaced: none directly - the reference has no spatial order; what it has at this place is
// np.unique over the voxel addresses (nimrud/utils/geometry.py:150), whose only role for the feature path is to
// make the search set a SET.  here the occupancy index does that (nm_index.hip) and the order exists for the
// device's sake: 64 consecutive queries must be neighbours in space.
//
// why not the library sort (rocPRIM onesweep, rounds 1-2): three passes at 1.95 TB/s plus a histogram pass, a
// key/row pair stream written by a key kernel and a separate gather of the coordinates - 0.53 ms of a 2.69 ms
// step at 10 M points.  what the order needs is weaker than a sort and that is what this file exploits:
//   * results do not depend on the order (integer moments of voxel sets), so nothing breaks if a pass is not
//     perfectly stable: a pair's rank inside its (tile, wave, digit) group comes from ONE LDS atomic on the
//     wave's own counter - no ballots, no match loops.  tiles, waves and rounds are ordered by construction;
//     lanes that meet on a counter inside one instruction are served in lane order on this hardware, which
//     makes the result an exact, deterministic sort - but that last step is observed, not promised, and only the
//     compactness of a wave's 64 queries would suffer without it;
//   * the first pass needs no row numbers in memory (row = position), the last pass writes no keys;
//   * no decoupled look-back, no spinning: per-tile digit counts (written by the pass before - the key
//     kernel for the first pass, a small counting kernel for the others), one flat scan of the digit-major
//     count matrix, then a scatter kernel with no inter-block dependency at all.  a step can be captured in a
//     hipGraph and replayed back to back (the library's look-back state was the suspect when that stalled).
//
// per pass: tiles of 8192 pairs (2048 for clouds of up to 2^20 points), 512 threads, 68 KB of LDS (two blocks per
// CU); a tile's pairs are ranked, placed in digit order in LDS and leave as runs of consecutive addresses.
// blocks are mapped to tiles so that each XCD owns one contiguous range of tiles: the short runs of neighbouring
// tiles meet in one L2.  small clouds: up to 64 tiles the scatter blocks sum the count matrix themselves (no scan
// launches), and a key of at most 20 bits is sorted in two passes (the third pass's launches leave at once - the
// host cannot know the key width, the lattice is built on the device).

#include "nm_common.h"
#include "nm_index.h"

constexpr int SORT_THREADS = 512;
constexpr int SORT_ITEMS_BIG = 16;        // tiles of 8192 pairs: clouds above SORT_SMALL_N points
constexpr int SORT_ITEMS_SMALL = 4;       // tiles of 2048 pairs: small clouds, where 13 blocks of 8192 pairs leave
                                          // the chip empty (100 k points: 49 blocks instead of 13)
constexpr int64_t SORT_SMALL_N = 1 << 20;
constexpr int SORT_DIRECT_TILES = 64;     // up to this many tiles a scatter block sums the count matrix itself:
                                          // no scan launches (six of a step's launches, ~5 us each).  (measured
                                          // at 153 tiles of 8192: every block reading 0.6 MB of counts costs
                                          // 75 us more than the scans it saves)
constexpr int SORT_BINS = 1 << NM_ORDER_PASS_BITS;        // 1024
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_CHUNK = SCAN_THREADS * SCAN_ITEMS;     // 4096 counts per block
static_assert(SORT_BINS == 2 * SORT_THREADS, "the bin scan gives every thread two bins");
static_assert(3 * NM_ORDER_PASS_BITS >= NM_ORDER_KEY_BITS, "three passes must cover the key");

// 3-D Morton (Z-order) spread of up to 21 bits: a Z-order run of points is compact in all three axes at EVERY
// coarser scale, which keeps the boxes the search kernel stages small (a column-major order scatters vertical
// structures - poles, facades - over the whole slab)
__device__ __forceinline__ uint64_t nm_spread3(uint32_t v)
{
    uint64_t x = v & 0x1FFFFFu;
    x = (x | (x << 32)) & 0x001F00000000FFFFull;
    x = (x | (x << 16)) & 0x001F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

__device__ __forceinline__ uint64_t nm_spread2(uint32_t v)
{
    uint64_t x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}

// the sort key of a point: its cell of the ordering lattice, Z-order with the always-zero bits squeezed out
// (bit b of every axis that HAS a bit b, lowest bits first - the order of the plain 3-way interleave with only
// wx+wy+wz key bits), cut to NM_ORDER_KEY_BITS from the top.  the cell comes from a multiplication by
// fl(1/e): a point within rounding of a cell face may get its neighbour's key, which the order does not mind
// (every kernel that NEEDS the cell - index, search - computes it exactly)
__device__ __forceinline__ uint32_t nm_order_key(const double* __restrict__ p, const OrderDev& O)
{
    if (!O.valid) return 0u;
    const LatticeDev& L = O.L;
    const ZLayout& Z = O.Z;
    int32_t cx = nm_clamp_cell(floor((p[0] - L.min_x) * L.inv_edge));
    int32_t cy = nm_clamp_cell(floor((p[1] - L.min_y) * L.inv_edge));
    int32_t cz = nm_clamp_cell(floor((p[2] - L.min_z) * L.inv_edge));
    cx = min(max(cx, 0), (int32_t)((1u << L.wx) - 1u));
    cy = min(max(cy, 0), (int32_t)((1u << L.wy) - 1u));
    cz = min(max(cz, 0), (int32_t)((1u << L.wz) - 1u));
    uint64_t k = 0ull;
    if (O.morton) {
        // with w1 <= w2 the two smaller widths: bits below w1 are interleaved 3-way, bits in [w1, w2) 2-way
        // among the axes that still have bits, the rest belongs to the widest axis alone
        const uint32_t c[3] = {(uint32_t)cx, (uint32_t)cy, (uint32_t)cz};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const uint32_t lo = c[a] & ((1u << Z.w1) - 1u);
            k |= nm_spread3(lo) << a;
            if (Z.off2[a] >= 0) {
                const uint32_t mid = (c[a] >> Z.w1) & ((1u << (Z.w2 - Z.w1)) - 1u);
                k |= nm_spread2(mid) << (3 * Z.w1 + Z.off2[a]);
                k |= (uint64_t)(c[a] >> Z.w2) << (3 * Z.w1 + 2 * (Z.w2 - Z.w1));
            }
        }
    } else {
        k = nm_cell_key((uint32_t)cx, (uint32_t)cy, (uint32_t)cz, L);
    }
    return (uint32_t)(k >> O.shift);
}

// ---- pass 0, first half: keys, and the digit counts of every tile of the key stream ---------------------------
// counts are kept digit-major, H[digit * hstride + tile] (hstride = tiles rounded up to four; the padding stays
// zero): a flat exclusive scan of H then IS the table of global offsets (all of digit 0's tiles, then digit 1's, ...)
template <int ITEMS>
__global__ __launch_bounds__(SORT_THREADS) void k_order_keys_hist(const double* __restrict__ xyz, int64_t n,
                                                                  int64_t stride,
                                                                  const OrderDev* __restrict__ od,
                                                                  uint32_t* __restrict__ keys,
                                                                  uint32_t* __restrict__ H, int32_t hstride,
                                                                  int32_t tiles, const ScaleDev* __restrict__ ladder,
                                                                  int32_t n_scales, int32_t parts)
{
    constexpr int TILE = SORT_THREADS * ITEMS;
    // blocks appended behind the tiles' reset a third of the ladder's indexes: one launch less per step (3-6 us: what a
    // small cloud's step notices).  the 0.3 GB of a 10 M-point ladder do NOT hide in here - the counting kernels
    // lengthen by what the clear kernel took, leading blocks instead of trailing ones make it worse (measured) - the
    // memory system is busy with the kernels' own reads
    if ((int32_t)blockIdx.x >= tiles) {
        nm_index_clear_part(ladder, n_scales, (uint64_t)(blockIdx.x - tiles) * SORT_THREADS + threadIdx.x,
                            (uint64_t)(gridDim.x - tiles) * SORT_THREADS, 0, parts);
        return;
    }
    __shared__ uint32_t lh[SORT_BINS];
    const int tid = threadIdx.x;
    const int32_t tile = blockIdx.x;
    lh[tid] = 0u;
    lh[tid + SORT_THREADS] = 0u;
    __syncthreads();
    const OrderDev& O = *od;
    const uint32_t mask = (1u << O.bpp) - 1u;
    const int64_t base = (int64_t)tile * TILE;
#pragma unroll 4
    for (int i = 0; i < ITEMS; ++i) {
        const int64_t idx = base + i * SORT_THREADS + tid;
        if (idx < n) {
            const uint32_t k = nm_order_key(xyz + idx * stride, O);
            keys[idx] = k;
            atomicAdd(&lh[k & mask], 1u);
        }
    }
    __syncthreads();
    H[(size_t)tid * hstride + tile] = lh[tid];
    H[(size_t)(tid + SORT_THREADS) * hstride + tile] = lh[tid + SORT_THREADS];
    if (tile == tiles - 1)       // the padding columns of every row (the scan runs in place)
        for (int32_t pad = tiles; pad < hstride; ++pad) {
            H[(size_t)tid * hstride + pad] = 0u;
            H[(size_t)(tid + SORT_THREADS) * hstride + pad] = 0u;
        }
}

// ---- passes 1, 2, first half: digit counts of every tile of a pair stream ------------------------------------------
template <int ITEMS>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_hist(const uint2* __restrict__ pairs, int64_t n,
                                                            int pass, const OrderDev* __restrict__ od,
                                                            uint32_t* __restrict__ H, int32_t hstride,
                                                            int32_t tiles, const ScaleDev* __restrict__ ladder,
                                                            int32_t n_scales, int32_t parts)
{
    constexpr int TILE = SORT_THREADS * ITEMS;
    if ((int32_t)blockIdx.x >= tiles) {       // appended blocks: this pass's third of the index reset
        nm_index_clear_part(ladder, n_scales, (uint64_t)(blockIdx.x - tiles) * SORT_THREADS + threadIdx.x,
                            (uint64_t)(gridDim.x - tiles) * SORT_THREADS, pass, parts);
        return;
    }
    if (pass >= od->passes) return;       // a short key is sorted in two passes: the third one's launches leave
    __shared__ uint32_t lh[SORT_BINS];
    const int tid = threadIdx.x;
    const int32_t tile = blockIdx.x;
    lh[tid] = 0u;
    lh[tid + SORT_THREADS] = 0u;
    __syncthreads();
    const int bpp = od->bpp;
    const uint32_t shift = (uint32_t)(pass * bpp), mask = (1u << bpp) - 1u;
    const int64_t base = (int64_t)tile * TILE;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int64_t idx = base + i * SORT_THREADS + tid;
        if (idx < n) atomicAdd(&lh[(pairs[idx].x >> shift) & mask], 1u);
    }
    __syncthreads();
    H[(size_t)tid * hstride + tile] = lh[tid];
    H[(size_t)(tid + SORT_THREADS) * hstride + tile] = lh[tid + SORT_THREADS];
    if (tile == tiles - 1)       // the padding columns of every row (the scan runs in place)
        for (int32_t pad = tiles; pad < hstride; ++pad) {
            H[(size_t)tid * hstride + pad] = 0u;
            H[(size_t)(tid + SORT_THREADS) * hstride + pad] = 0u;
        }
}

// ---- flat exclusive scan of the count matrix, two launches ------------------------------------------------------------
__device__ __forceinline__ uint32_t nm_block_sum(uint32_t v, uint32_t* wsum)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += wsum[w];
    return t;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const uint32_t* __restrict__ H, int64_t len,
                                                              uint32_t* __restrict__ partial, int pass,
                                                              const OrderDev* __restrict__ od)
{
    if (pass >= od->passes) return;
    __shared__ uint32_t wsum[SCAN_THREADS / 64];
    const int64_t at = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
    if (at < len) {      // len is a multiple of SORT_BINS, hence of SCAN_ITEMS
        const uint4* p = (const uint4*)(H + at);
#pragma unroll
        for (int q = 0; q < SCAN_ITEMS / 4; ++q) {
            const uint4 v = p[q];
            s += v.x + v.y + v.z + v.w;
        }
    }
    const uint32_t t = nm_block_sum(s, wsum);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(uint32_t* __restrict__ H, int64_t len,
                                                             const uint32_t* __restrict__ partial, int pass,
                                                             const OrderDev* __restrict__ od)
{
    if (pass >= od->passes) return;
    __shared__ uint32_t wsum[SCAN_THREADS / 64];
    __shared__ uint32_t wpre[SCAN_THREADS / 64];
    // everything before this block's chunk
    uint32_t before = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += SCAN_THREADS) before += partial[b];
    before = nm_block_sum(before, wsum);
    const int64_t at = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
    if (at < len) {
        const uint4* p = (const uint4*)(H + at);
#pragma unroll
        for (int q = 0; q < SCAN_ITEMS / 4; ++q) {
            const uint4 x = p[q];
            v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
            s += x.x + x.y + x.z + x.w;
        }
    }
    // exclusive scan of the threads' sums: inside the wave, then across the block's waves
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    __syncthreads();       // nm_block_sum's readers are done with wsum
    if (lane == 63) wpre[w] = incl;
    __syncthreads();
    uint32_t run = before + incl - s;
    for (int ww = 0; ww < w; ++ww) run += wpre[ww];
    if (at < len) {
        uint4* p = (uint4*)(H + at);
#pragma unroll
        for (int q = 0; q < SCAN_ITEMS / 4; ++q) {
            uint4 x;
            x.x = run; run += v[4 * q];
            x.y = run; run += v[4 * q + 1];
            x.z = run; run += v[4 * q + 2];
            x.w = run; run += v[4 * q + 3];
            p[q] = x;
        }
    }
}

// ---- a pass, second half: rank, place in digit order in LDS, leave as runs ------------------------------------------
//   PASS 0: keys in, rows = positions; pairs out
//   PASS 1: pairs in, pairs out
//   PASS 2: pairs in; the permutation out (no keys: nobody reads them)
struct SortIO {
    const uint32_t* keys_in;
    const uint2* pairs_in;
    uint2* pairs_out;
    uint32_t* order_out;
};

template <int PASS, int ITEMS, bool DIRECT>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter(SortIO io, int64_t n, int32_t tiles,
                                                               int32_t hstride,
                                                               const uint32_t* __restrict__ offsets,
                                                               const OrderDev* __restrict__ od)
{
    constexpr int TILE = SORT_THREADS * ITEMS;
    constexpr int WAVES = SORT_THREADS / 64;
    constexpr int CNT_STRIDE = SORT_BINS + 4;
    constexpr int BUF_BYTES = TILE * 8 > WAVES * CNT_STRIDE * 4 ? TILE * 8 : WAVES * CNT_STRIDE * 4;
    if (PASS >= od->passes) return;       // a short key is sorted in two passes: the third one's launches leave
    const bool last = PASS + 1 == od->passes;      // this pass writes the permutation, not pairs
    // up to 68 KB of LDS: more than a kernel may declare statically (the code object then fails to load), so it
    // is dynamic and the launch asks for it (scatter_lds_bytes)
    extern __shared__ __attribute__((aligned(16))) unsigned char sort_lds[];
    uint2* buf = (uint2*)sort_lds;                                  // TILE pairs
    uint32_t* cnt = (uint32_t*)sort_lds;                            // before that: every wave's digit counters,
                                                                    // CNT_STRIDE words each ([SORT_BINS]: the bin
                                                                    // of the slots beyond the end of a partial tile)
    uint32_t* delta = (uint32_t*)(sort_lds + BUF_BYTES);            // global offset of a digit's run minus its
                                                                    // first LDS slot
    uint32_t* wsum = delta + SORT_BINS;       // SORT_THREADS / 64 wave totals, twice
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int32_t tile = (int32_t)nm_xcd_batch(blockIdx.x, gridDim.x);
    const int bpp = od->bpp;
    const uint32_t shift = (uint32_t)(PASS * bpp), mask = (1u << bpp) - 1u;
    const int64_t base = (int64_t)tile * TILE;
    const int32_t count = (int32_t)(n - base < TILE ? n - base : TILE);
    for (int t = tid; t < WAVES * CNT_STRIDE; t += SORT_THREADS) cnt[t] = 0u;
    // a wave owns ITEMS * 64 consecutive pairs of the tile and takes them 64 at a time, lane = position:
    // the order of a tile's pairs is (wave, round, lane).
    // (everything per item is computed unconditionally - slots beyond the end read the tile's last pair and go
    // to a bin of their own; conditional definitions of these arrays cost the compiler 250 registers and spills)
    uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
    const int32_t p0 = w * (ITEMS * 64) + lane;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int32_t p = p0 + i * 64;
        const int64_t at = base + (p < count ? p : count - 1);
        if (PASS == 0) {
            key[i] = io.keys_in[at];
            val[i] = (uint32_t)at;
        } else {
            const uint2 kv = io.pairs_in[at];
            key[i] = kv.x;
            val[i] = kv.y;
        }
    }
    // global offsets of this tile's runs.  from the scanned matrix: one entry per digit.  DIRECT (few tiles, no
    // scan launches): the digit's row of raw counts - its total, and the part before this tile; the totals are
    // scanned over the digits below, together with the block's own counts
    uint32_t g0 = 0, g1 = 0, rowsum = 0;
    if (DIRECT) {
        // (rows are 16-byte aligned and at most SORT_DIRECT_TILES long: a handful of independent wide loads)
        const uint4* r0 = (const uint4*)(offsets + (size_t)(2 * tid) * hstride);
        const uint4* r1 = (const uint4*)(offsets + (size_t)(2 * tid + 1) * hstride);
        uint32_t t0 = 0, t1 = 0;
#pragma unroll
        for (int q = 0; q < SORT_DIRECT_TILES / 4; ++q) {
            if (4 * q >= tiles) break;
            const uint4 a = r0[q], b = r1[q];
            const int32_t left = tile - 4 * q;        // entries of this quad that belong to tiles before this one
            g0 += (left > 0 ? a.x : 0u) + (left > 1 ? a.y : 0u) + (left > 2 ? a.z : 0u) + (left > 3 ? a.w : 0u);
            g1 += (left > 0 ? b.x : 0u) + (left > 1 ? b.y : 0u) + (left > 2 ? b.z : 0u) + (left > 3 ? b.w : 0u);
            t0 += a.x + a.y + a.z + a.w;
            t1 += b.x + b.y + b.z + b.w;
        }
        g1 += t0;            // digit 2t + 1 starts behind all of digit 2t
        rowsum = t0 + t1;
    } else {
        g0 = offsets[(size_t)(2 * tid) * hstride + tile];
        g1 = offsets[(size_t)(2 * tid + 1) * hstride + tile];
    }
    __syncthreads();
    // rank inside the (wave, digit) group: one LDS atomic per pair on the wave's own counter.  a wave's LDS
    // operations execute in program order, so rounds are ranked in order; inside one instruction the lanes
    // that meet on a counter are served in lane order on this hardware (observed, not promised: tests measure
    // the inversions of the finished order) - then the pass is stable and the order exactly sorted.  nothing
    // but the compactness of a wave's queries depends on it.
    uint32_t* mine = cnt + w * CNT_STRIDE;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int32_t p = p0 + i * 64;
        const uint32_t d = p < count ? (key[i] >> shift) & mask : (uint32_t)SORT_BINS;
        rank[i] = atomicAdd(&mine[d], 1u);
    }
    __syncthreads();
    // first LDS slot of every (digit, wave): digits in order, inside a digit the waves in order.  thread t
    // owns bins 2t and 2t + 1
    uint32_t c0 = 0, c1 = 0;
    uint32_t pre0[WAVES], pre1[WAVES];
#pragma unroll
    for (int ww = 0; ww < WAVES; ++ww) {
        pre0[ww] = c0;
        pre1[ww] = c1;
        c0 += cnt[ww * CNT_STRIDE + 2 * tid];
        c1 += cnt[ww * CNT_STRIDE + 2 * tid + 1];
    }
    const uint32_t s = c0 + c1;
    uint32_t incl = s, gincl = rowsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off);
        const uint32_t go = DIRECT ? __shfl_up(gincl, off) : 0u;
        if (lane >= off) {
            incl += o;
            gincl += go;
        }
    }
    if (lane == 63) {
        wsum[w] = incl;
        wsum[WAVES + w] = gincl;
    }
    __syncthreads();
    uint32_t excl = incl - s, gbase = gincl - rowsum;
    for (int ww = 0; ww < w; ++ww) {
        excl += wsum[ww];
        gbase += wsum[WAVES + ww];
    }
    if (DIRECT) {
        g0 += gbase;
        g1 += gbase;
    }
#pragma unroll
    for (int ww = 0; ww < WAVES; ++ww) {
        cnt[ww * CNT_STRIDE + 2 * tid] = excl + pre0[ww];
        cnt[ww * CNT_STRIDE + 2 * tid + 1] = excl + c0 + pre1[ww];
    }
    delta[2 * tid] = g0 - excl;
    delta[2 * tid + 1] = g1 - (excl + c0);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) rank[i] += mine[(key[i] >> shift) & mask];     // now the LDS slot
    __syncthreads();       // the counters have been read: the buffer is free
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int32_t p = p0 + i * 64;
        if (p < count) buf[rank[i]] = make_uint2(key[i], val[i]);
    }
    __syncthreads();
    if (!last) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int32_t p = i * SORT_THREADS + tid;
            if (p < count) {
                const uint2 kv = buf[p];
                io.pairs_out[delta[(kv.x >> shift) & mask] + (uint32_t)p] = kv;
            }
        }
    } else {
        // the last pass writes the permutation alone (nobody reads the keys again).  (carrying the
        // coordinates here was measured: 204 us against 37 + 98 for this pass plus a gather in sorted order -
        // a tile of this pass is a run of keys with equal LOW bits, i.e. rows from all over the cloud, so its
        // gather has no locality, while the gather in sorted order reads neighbours in space, which in a
        // scanner's or tiler's output are mostly neighbours in memory)
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int32_t p = i * SORT_THREADS + tid;
            if (p < count) {
                const uint2 kv = buf[p];
                io.order_out[delta[(kv.x >> shift) & mask] + (uint32_t)p] = kv.y;
            }
        }
    }
}

// the coordinates in sorted order: (n,3) contiguous
__global__ __launch_bounds__(256) void k_gather_xyz(const double* __restrict__ xyz, int64_t n, int64_t stride,
                                                    const uint32_t* __restrict__ order,
                                                    double* __restrict__ out)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* p = xyz + (int64_t)order[i] * stride;
    const double x = p[0], y = p[1], z = p[2];
    out[i * 3 + 0] = x;
    out[i * 3 + 1] = y;
    out[i * 3 + 2] = z;
}

template <int ITEMS>
constexpr size_t scatter_lds_bytes()
{
    constexpr size_t tile = (size_t)SORT_THREADS * ITEMS * 8, counters = (size_t)(SORT_THREADS / 64) * (SORT_BINS + 4) * 4;
    return (tile > counters ? tile : counters) + SORT_BINS * 4 + 128;
}

// ---- host side -----------------------------------------------------------------------------------------------------------

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct OrderScratch {
    size_t keys, pairs_a, pairs_b, hist, partial, total;
    int32_t items;           // pairs per thread of a tile (SORT_ITEMS_BIG or SORT_ITEMS_SMALL)
    int32_t tiles;
    int32_t hstride;         // row stride of the count matrix: tiles rounded up to a multiple of four
    int64_t hist_len;
    int32_t scan_blocks;
};

static void order_scratch(int64_t n, OrderScratch* S)
{
    if (n < 1) n = 1;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += align_up(bytes);
        return at;
    };
    S->items = n <= SORT_SMALL_N ? SORT_ITEMS_SMALL : SORT_ITEMS_BIG;
    const int64_t tile = (int64_t)SORT_THREADS * S->items;
    S->tiles = (int32_t)((n + tile - 1) / tile);
    S->hstride = (S->tiles + 3) & ~3;
    S->hist_len = (int64_t)SORT_BINS * S->hstride;
    S->scan_blocks = (int32_t)((S->hist_len + SCAN_CHUNK - 1) / SCAN_CHUNK);
    S->keys = take((size_t)n * 4);
    S->pairs_a = take((size_t)n * 8);
    S->pairs_b = take((size_t)n * 8);
    S->hist = take((size_t)S->hist_len * 4);
    S->partial = take((size_t)S->scan_blocks * 4);
    S->total = off;
}

size_t nm_order_scratch_bytes(int64_t n)
{
    OrderScratch S;
    order_scratch(n, &S);
    return S.total;
}

template <int ITEMS, bool DIRECT>
static int order_build(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride, const OrderDev* od,
                       const OrderScratch& S, char* w, uint32_t* order, double* sorted_xyz,
                       const ScaleDev* clear_ladder, int clear_scales, hipStream_t s)
{
    uint32_t* keys = (uint32_t*)(w + S.keys);
    uint2* pa = (uint2*)(w + S.pairs_a);
    uint2* pb = (uint2*)(w + S.pairs_b);
    uint32_t* H = (uint32_t*)(w + S.hist);
    uint32_t* partial = (uint32_t*)(w + S.partial);
    const int tiles = S.tiles;
    constexpr size_t LDS = scatter_lds_bytes<ITEMS>();
    // (the attribute belongs to the function on the context's device - a process may own several - and is set
    // once per context and kernel family: three runtime calls per step are host time a 0.1 ms step can see)
    constexpr uint32_t family = 1u << ((ITEMS == SORT_ITEMS_BIG ? 0 : 2) + (DIRECT ? 1 : 0));
    if (!(ctx->order_attr_set & family)) {
        NM_HIP(ctx, hipFuncSetAttribute((const void*)k_sort_scatter<0, ITEMS, DIRECT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        NM_HIP(ctx, hipFuncSetAttribute((const void*)k_sort_scatter<1, ITEMS, DIRECT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        NM_HIP(ctx, hipFuncSetAttribute((const void*)k_sort_scatter<2, ITEMS, DIRECT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        ctx->order_attr_set |= family;
    }
    auto scan = [&](int pass) {
        if (DIRECT) return;        // the scatter blocks sum the raw counts themselves
        k_scan_reduce<<<S.scan_blocks, SCAN_THREADS, 0, s>>>(H, S.hist_len, partial, pass, od);
        k_scan_apply<<<S.scan_blocks, SCAN_THREADS, 0, s>>>(H, S.hist_len, partial, pass, od);
    };
    // (the host's half of nm_order_plan's rule: a cloud this small is sorted in two passes - the plan keeps 20 key
    // bits - so the third pass's launches are not made; above, the device decides and a short key's third pass leaves)
    const bool two = ctx->order_points > 0 && ctx->order_points <= NM_ORDER_TWO_PASS_N;
    const int parts = two ? 2 : 3;          // counting kernels that share the index reset
    SortIO io{};
    io.order_out = order;          // whichever pass is the key's last writes the permutation
    // blocks appended to the counting kernels reset a share each of the ladder's indexes (when asked to)
    int extra = 0;
    if (clear_ladder) {
        extra = tiles / 2;
        if (extra < 8) extra = 8;
        if (extra > 512) extra = 512;
    }
    k_order_keys_hist<ITEMS><<<tiles + extra, SORT_THREADS, 0, s>>>(d_xyz, n, stride, od, keys, H, S.hstride, tiles,
                                                                     clear_ladder, clear_scales, parts);
    scan(0);
    io.keys_in = keys;
    io.pairs_out = pa;
    k_sort_scatter<0, ITEMS, DIRECT><<<tiles, SORT_THREADS, LDS, s>>>(io, n, tiles, S.hstride, H, od);
    k_sort_hist<ITEMS><<<tiles + extra, SORT_THREADS, 0, s>>>(pa, n, 1, od, H, S.hstride, tiles, clear_ladder,
                                                               clear_scales, parts);
    scan(1);
    io.pairs_in = pa;
    io.pairs_out = pb;
    k_sort_scatter<1, ITEMS, DIRECT><<<tiles, SORT_THREADS, LDS, s>>>(io, n, tiles, S.hstride, H, od);
    if (!two) {
        k_sort_hist<ITEMS><<<tiles + extra, SORT_THREADS, 0, s>>>(pb, n, 2, od, H, S.hstride, tiles, clear_ladder,
                                                                   clear_scales, parts);
        scan(2);
        io.pairs_in = pb;
        io.pairs_out = nullptr;
        k_sort_scatter<2, ITEMS, DIRECT><<<tiles, SORT_THREADS, LDS, s>>>(io, n, tiles, S.hstride, H, od);
    }
    if (sorted_xyz) k_gather_xyz<<<(int)((n + 255) / 256), 256, 0, s>>>(d_xyz, n, stride, order, sorted_xyz);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}

int nm_order_build(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride, const OrderDev* d_order_dev,
                   void* scratch, size_t scratch_bytes, uint32_t* order, double* sorted_xyz, hipStream_t s,
                   const ScaleDev* clear_ladder, int clear_scales)
{
    OrderScratch S;
    order_scratch(n, &S);
    if (scratch_bytes < S.total) NM_FAIL(ctx, NM_ERR_WORKSPACE, "spatial order: scratch %zu < %zu", scratch_bytes, S.total);
    char* w = (char*)scratch;
    if (S.items == SORT_ITEMS_BIG) {
        if (S.tiles <= SORT_DIRECT_TILES)
            return order_build<SORT_ITEMS_BIG, true>(ctx, d_xyz, n, stride, d_order_dev, S, w, order, sorted_xyz, clear_ladder,
                                                     clear_scales, s);
        return order_build<SORT_ITEMS_BIG, false>(ctx, d_xyz, n, stride, d_order_dev, S, w, order, sorted_xyz, clear_ladder,
                                                     clear_scales, s);
    }
    if (S.tiles <= SORT_DIRECT_TILES)
        return order_build<SORT_ITEMS_SMALL, true>(ctx, d_xyz, n, stride, d_order_dev, S, w, order, sorted_xyz, clear_ladder,
                                                     clear_scales, s);
    return order_build<SORT_ITEMS_SMALL, false>(ctx, d_xyz, n, stride, d_order_dev, S, w, order, sorted_xyz, clear_ladder,
                                                     clear_scales, s);
}

// ---- the order as an entry point of its own (inspection, tests) ------------------------------------------------------
__global__ void k_order_plan(LatticeDev L, OrderDev* out, int64_t points)
{
    if (threadIdx.x == 0) {
        OrderDev O;
        nm_order_plan(L, &O, points);
        *out = O;
    }
}

__global__ __launch_bounds__(256) void k_order_keys_of(const double* __restrict__ xyz3, int64_t n,
                                                       const OrderDev* __restrict__ od,
                                                       uint32_t* __restrict__ keys)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) keys[i] = nm_order_key(xyz3 + i * 3, *od);
}

extern "C" size_t nm_spatial_order_workspace_bytes(int64_t n)
{
    return align_up(sizeof(OrderDev)) + nm_order_scratch_bytes(n);
}

extern "C" int nm_spatial_order(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride,
                                const nm_lattice* lat, uint32_t* d_order, double* d_sorted_xyz,
                                uint32_t* d_keys_sorted, void* d_work, size_t work_bytes, void* stream)
{
    NM_ENTER_STREAM(ctx, stream);
    if (!d_xyz || !d_order || !d_sorted_xyz || !d_work || n < 1 || stride < 3 || n >= ((int64_t)1 << 31))
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_spatial_order: bad arguments");
    int rc = validate_lattice(ctx, lat);
    if (rc) return rc;
    if (work_bytes < nm_spatial_order_workspace_bytes(n))
        NM_FAIL(ctx, NM_ERR_WORKSPACE, "nm_spatial_order: workspace %zu < required %zu", work_bytes,
                nm_spatial_order_workspace_bytes(n));
    hipStream_t s = (hipStream_t)stream;
    OrderDev* od = (OrderDev*)d_work;
    char* scratch = (char*)d_work + align_up(sizeof(OrderDev));
    ctx->order_points = n;
    k_order_plan<<<1, 64, 0, s>>>(make_lattice_dev(lat), od, n);
    rc = nm_order_build(ctx, d_xyz, n, stride, od, scratch, work_bytes - align_up(sizeof(OrderDev)), d_order,
                        d_sorted_xyz, s);
    if (rc) return rc;
    if (d_keys_sorted)
        k_order_keys_of<<<(int)((n + 255) / 256), 256, 0, s>>>(d_sorted_xyz, n, od, d_keys_sorted);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}
