// nm_index.hip - lattice kernels: cloud bounds, voxel addresses, the per-scale occupancy index.
//
// reference code replaced: VoxelFilter (nimrud/utils/geometry.py:23-154).
// all lattice arithmetic is fp64 and compiled with -ffp-contract=off so that cells and centres are
// bit-identical to numpy's.

#include "nm_common.h"
#include "nm_index.h"

#include <rocprim/device/device_radix_sort.hpp>

// ---------------------------------------------------------------------------------------------------
// bounds: per-axis min/max.  doubles are mapped to order-preserving u64 so that the hardware's
// integer atomic min/max can be used; a second tiny kernel maps them back.
// ---------------------------------------------------------------------------------------------------

__device__ __forceinline__ uint64_t nm_order_encode(double v)
{
    uint64_t b = (uint64_t)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double nm_order_decode(uint64_t k)
{
    uint64_t b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}

#ifndef NM_MAP_FORM
#define NM_MAP_FORM 1       // the ladder path may index a scale with a word per superblock (IndexDev::map)
#endif
#ifndef NM_BOUNDS_BLOCKS
#define NM_BOUNDS_BLOCKS 1024
#endif
static_assert((size_t)NM_BOUNDS_BLOCKS * 6 * 8 <= NM_BOUNDS_SCRATCH_BYTES, "scratch of the atomic-free bounds pass");
__global__ void k_bounds_init(uint64_t* mm)
{
    int t = threadIdx.x;
    if (t < 3) mm[t] = ~0ull;          // running min
    else if (t < 6) mm[t] = 0ull;      // running max
}

// `partial` (NM_BOUNDS_BLOCKS * 6 words of scratch) set: every block stores its own extrema there and
// k_bounds_finish folds them - no atomics.  (1024 blocks x 6 atomics on six addresses serialise into
// ~20 us, a third of the pass at 10 M points and most of it at 1 M.)  null: atomics on mm itself.
__global__ __launch_bounds__(256) void k_bounds(const double* __restrict__ xyz, int64_t n,
                                                int64_t stride, uint64_t* mm, uint64_t* partial)
{
    double lo[3] = {INFINITY, INFINITY, INFINITY};
    double hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    // four rows in flight per thread (the loop is bound by the latency of its loads, not by their number)
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    for (; i + 3 * step < n; i += 4 * step) {
        double v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double* p = xyz + (i + u * step) * stride;
#pragma unroll
            for (int a = 0; a < 3; ++a) v[u][a] = p[a];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                lo[a] = fmin(lo[a], v[u][a]);
                hi[a] = fmax(hi[a], v[u][a]);
            }
    }
    for (; i < n; i += step) {
        const double* p = xyz + i * stride;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double v = p[a];
            lo[a] = fmin(lo[a], v);
            hi[a] = fmax(hi[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fmin(lo[a], __shfl_xor(lo[a], off));
            hi[a] = fmax(hi[a], __shfl_xor(hi[a], off));
        }
    }
    // one atomic set per block: same-address atomics serialise
    __shared__ double slo[4][3], shi[4][3];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            slo[w][a] = lo[a];
            shi[w][a] = hi[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        const double l = fmin(fmin(slo[0][a], slo[1][a]), fmin(slo[2][a], slo[3][a]));
        const double h = fmax(fmax(shi[0][a], shi[1][a]), fmax(shi[2][a], shi[3][a]));
        if (partial) {
            partial[blockIdx.x * 6 + a] = nm_order_encode(l);
            partial[blockIdx.x * 6 + 3 + a] = nm_order_encode(h);
        } else {
            atomicMin((unsigned long long*)&mm[a], (unsigned long long)nm_order_encode(l));
            atomicMax((unsigned long long*)&mm[3 + a], (unsigned long long)nm_order_encode(h));
        }
    }
}

__global__ __launch_bounds__(64) void k_bounds_finish(uint64_t* mm, const uint64_t* __restrict__ partial,
                                                      int blocks)
{
    const int t = threadIdx.x;
    if (partial) {
        // one wave: lane t folds blocks t, t + 64, ...; then across the lanes
        uint64_t v[6];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            v[a] = ~0ull;
            v[3 + a] = 0ull;
        }
        for (int b = t; b < blocks; b += 64) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const uint64_t l = partial[b * 6 + a], h = partial[b * 6 + 3 + a];
                v[a] = l < v[a] ? l : v[a];
                v[3 + a] = h > v[3 + a] ? h : v[3 + a];
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const uint64_t l = (uint64_t)__shfl_xor((unsigned long long)v[a], off);
                const uint64_t h = (uint64_t)__shfl_xor((unsigned long long)v[3 + a], off);
                v[a] = l < v[a] ? l : v[a];
                v[3 + a] = h > v[3 + a] ? h : v[3 + a];
            }
        }
        if (t == 0) {
#pragma unroll
            for (int a = 0; a < 6; ++a) ((double*)mm)[a] = nm_order_decode(v[a]);
        }
        return;
    }
    if (t < 6) {
        double v = nm_order_decode(mm[t]);
        ((double*)mm)[t] = v;
    }
}

// the extrema of a cloud into d_minmax (6 doubles).  d_partial: NM_BOUNDS_SCRATCH_BYTES of scratch for the
// atomic-free form, or null.
int nm_bounds_scratch(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride, double* d_minmax,
                      void* d_partial, hipStream_t s)
{
    uint64_t* mm = (uint64_t*)d_minmax;
    uint64_t* partial = (uint64_t*)d_partial;
    int64_t blocks = (n + 255) / 256;
    if (blocks > NM_BOUNDS_BLOCKS) blocks = NM_BOUNDS_BLOCKS;
    if (!partial) k_bounds_init<<<1, 64, 0, s>>>(mm);
    k_bounds<<<(int)blocks, 256, 0, s>>>(d_xyz, n, stride, mm, partial);
    k_bounds_finish<<<1, 64, 0, s>>>(mm, partial, (int)blocks);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}

// the bounds pass alone: per-block extrema into `d_partial` (NM_BOUNDS_SCRATCH_BYTES); returns the number of
// blocks that wrote one.  whoever reads them folds them (k_make_ladder)
int nm_bounds_partial(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride, void* d_partial,
                      int* blocks_out, hipStream_t s)
{
    int64_t blocks = (n + 255) / 256;
    if (blocks > NM_BOUNDS_BLOCKS) blocks = NM_BOUNDS_BLOCKS;
    k_bounds<<<(int)blocks, 256, 0, s>>>(d_xyz, n, stride, nullptr, (uint64_t*)d_partial);
    NM_HIP(ctx, hipGetLastError());
    *blocks_out = (int)blocks;
    return NM_OK;
}

extern "C" int nm_bounds(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride,
                         double* d_minmax, void* stream)
{
    NM_ENTER_STREAM(ctx, stream);
    if (!d_xyz || !d_minmax || n < 1 || stride < 3)
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_bounds: bad arguments");
    return nm_bounds_scratch(ctx, d_xyz, n, stride, d_minmax, nullptr, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------
// reference-order voxel addresses (geometry.py:103-116) -> sort -> unique (geometry.py:150)
// ---------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_addresses(const double* __restrict__ xyz, int64_t n,
                                                   int64_t stride, LatticeDev L,
                                                   uint64_t* __restrict__ addr, int64_t* oob_count)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    bool oob = false;
    if (i < n) {
        const double* p = xyz + i * stride;
        double fx = nm_cell_f(p[0], L.min_x, L.edge);
        double fy = nm_cell_f(p[1], L.min_y, L.edge);
        double fz = nm_cell_f(p[2], L.min_z, L.edge);
        // _check_in_bounds (geometry.py:95-97): min_corner <= p <= max_corner.  in cell terms a
        // point of the cloud the lattice was built from always lands in [0, 2^w).
        oob = fx < 0.0 || fy < 0.0 || fz < 0.0 || fx >= (double)(1u << L.wx) ||
              fy >= (double)(1u << L.wy) || fz >= (double)(1u << L.wz);
        int64_t cx = (int64_t)nm_clamp_cell(fx), cy = (int64_t)nm_clamp_cell(fy),
                cz = (int64_t)nm_clamp_cell(fz);
        // out-of-bounds points are counted (the host raises, as _check_in_bounds does) and clamped
        cx = min(max(cx, (int64_t)0), (int64_t)((1u << L.wx) - 1u));
        cy = min(max(cy, (int64_t)0), (int64_t)((1u << L.wy) - 1u));
        cz = min(max(cz, (int64_t)0), (int64_t)((1u << L.wz) - 1u));
        addr[i] = (uint64_t)(cx + (cy << L.s0) + (cz << L.s1));
    }
    unsigned long long m = __ballot(oob);
    if (m && oob_count && (threadIdx.x & 63) == 0)
        atomicAdd((unsigned long long*)oob_count, (unsigned long long)__popcll(m));
}

extern "C" int nm_coordinate_to_address(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride,
                                        const nm_lattice* lat, int64_t* d_addr_out, int64_t* d_oob,
                                        void* stream)
{
    NM_ENTER_STREAM(ctx, stream);
    if (n < 0 || stride < 3 || (n > 0 && (!d_xyz || !d_addr_out)))
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_coordinate_to_address: bad arguments");
    int rc = validate_lattice(ctx, lat);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (d_oob) NM_HIP(ctx, hipMemsetAsync(d_oob, 0, sizeof(int64_t), s));
    if (n == 0) return NM_OK;
    LatticeDev L = make_lattice_dev(lat);
    k_addresses<<<(int)((n + 255) / 256), 256, 0, s>>>(d_xyz, n, stride, L, (uint64_t*)d_addr_out,
                                                      d_oob);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}

// ordered compaction of run heads of a sorted array: 2048 keys per block, decoupled by one atomic
// per block is NOT ordered, so voxelize uses a two-pass count / prefix / write over blocks.
constexpr int UNIQ_BLOCK = 256;
constexpr int UNIQ_ITEMS = 8;
constexpr int UNIQ_TILE = UNIQ_BLOCK * UNIQ_ITEMS;

__global__ __launch_bounds__(UNIQ_BLOCK) void k_unique_count(const uint64_t* __restrict__ sorted,
                                                             int64_t n,
                                                             uint32_t* __restrict__ tile_count)
{
    __shared__ uint32_t wsum[UNIQ_BLOCK / 64];
    int64_t base = (int64_t)blockIdx.x * UNIQ_TILE;
    uint32_t c = 0;
#pragma unroll
    for (int it = 0; it < UNIQ_ITEMS; ++it) {
        int64_t i = base + it * UNIQ_BLOCK + threadIdx.x;
        if (i < n) {
            uint64_t k = sorted[i];
            bool head = (i == 0 || sorted[i - 1] != k);
            c += head ? 1u : 0u;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < UNIQ_BLOCK / 64; ++w) t += wsum[w];
        tile_count[blockIdx.x] = t;
    }
}

// exclusive prefix over the tile counts, one block (tiles <= a few 10^4 even at 5e7 points)
__global__ __launch_bounds__(1024) void k_tile_prefix(uint32_t* tile_count, int64_t tiles,
                                                      int64_t* counters)
{
    __shared__ uint64_t part[1024];
    int t = threadIdx.x;
    int64_t per = (tiles + 1023) / 1024;
    int64_t lo = t * per, hi = lo + per;
    if (hi > tiles) hi = tiles;
    uint64_t s = 0;
    for (int64_t i = lo; i < hi; ++i) s += tile_count[i];
    part[t] = s;
    __syncthreads();
    // Hillis-Steele over 1024 partials
    for (int off = 1; off < 1024; off <<= 1) {
        uint64_t v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint64_t run = (t == 0) ? 0 : part[t - 1];
    for (int64_t i = lo; i < hi; ++i) {
        uint32_t c = tile_count[i];
        tile_count[i] = (uint32_t)run;   // M < 2^32 is guaranteed by n < 2^32 check on the host
        run += c;
    }
    if (t == 1023) counters[0] = (int64_t)part[1023];
}

__global__ __launch_bounds__(UNIQ_BLOCK) void k_unique_write(const uint64_t* __restrict__ sorted,
                                                             int64_t n,
                                                             const uint32_t* __restrict__ tile_base,
                                                             int64_t* __restrict__ out)
{
    __shared__ uint32_t wbase[UNIQ_BLOCK / 64];
    __shared__ uint32_t running;
    int64_t base = (int64_t)blockIdx.x * UNIQ_TILE;
    if (threadIdx.x == 0) running = tile_base[blockIdx.x];
    __syncthreads();
    for (int it = 0; it < UNIQ_ITEMS; ++it) {
        int64_t i = base + it * UNIQ_BLOCK + threadIdx.x;
        bool head = false;
        uint64_t k = 0;
        if (i < n) {
            k = sorted[i];
            head = (i == 0 || sorted[i - 1] != k);
        }
        unsigned long long m = __ballot(head);
        int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 0) wbase[w] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t off = running;
        for (int ww = 0; ww < w; ++ww) off += wbase[ww];
        if (head) out[off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (int64_t)k;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (int ww = 0; ww < UNIQ_BLOCK / 64; ++ww) t += wbase[ww];
            running += t;
        }
        __syncthreads();
    }
}

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// rocPRIM's radix sort hands anything up to 2^20 items to a merge sort (block sort + ten merge passes at
// 1 M items: 21 launches, 164 us); its onesweep radix passes are faster from a few hundred thousand items
// (measured on MI355X, order stage of a step: 1 M points 0.208 -> 0.188 ms; at 100 k points onesweep's fixed
// cost makes it 0.143 against 0.073 ms, so small clouds stay on the merge sort)
#ifndef NM_MERGE_SORT_LIMIT
#define NM_MERGE_SORT_LIMIT 262144
#endif
using NmSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                rocprim::default_config, NM_MERGE_SORT_LIMIT>;
// the spatial order's (key, row) pairs, both 32 bits: keys of NM_ORDER_KEY_BITS = 30 bits in THREE onesweep
// passes of 10 bits.  (rocPRIM's tuned gfx950 entry for this type pair is 8 bits x 1024 threads x 16 items:
// four passes, 0.655 ms for the order stage of the 10 M-point benchmark step; 10 bits x 1024 x 14: 0.53 ms
// (8 items 0.575, 12 0.545, 13-15 0.53, 16 0.60).  11 bits does not fit the rank kernel's LDS; blocks of 512 or
// 256 threads are slower at every item count; the histogram kernel is best left at 1024 x 16.)
#ifndef NM_ONESWEEP_BITS
#define NM_ONESWEEP_BITS 10
#define NM_ONESWEEP_BLOCK 1024
#define NM_ONESWEEP_ITEMS 14
#endif
#ifndef NM_HIST_BLOCK
#define NM_HIST_BLOCK 1024
#define NM_HIST_ITEMS 16
#endif
using NmOnesweep32 = rocprim::radix_sort_onesweep_config<rocprim::kernel_config<NM_HIST_BLOCK, NM_HIST_ITEMS>,
                                                         rocprim::kernel_config<NM_ONESWEEP_BLOCK, NM_ONESWEEP_ITEMS>,
                                                         NM_ONESWEEP_BITS,
                                                         rocprim::block_radix_rank_algorithm::match>;
using NmSortConfig32 = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                  NmOnesweep32, NM_MERGE_SORT_LIMIT>;

static size_t sort_keys_temp_bytes(int64_t n)
{
    size_t temp = 0;
    (void)rocprim::radix_sort_keys(nullptr, temp, (uint64_t*)nullptr, (uint64_t*)nullptr,
                                   (size_t)n, 0, 64, (hipStream_t)0);
    return temp;
}

size_t nm_sort_pairs_temp_bytes(int64_t n)
{
    size_t temp = 0;
    (void)rocprim::radix_sort_pairs<NmSortConfig>(nullptr, temp, (uint64_t*)nullptr, (uint64_t*)nullptr,
                                    (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0, 64,
                                    (hipStream_t)0);
    size_t temp32 = 0;      // nm_order_build sorts 32-bit keys when they fit
    (void)rocprim::radix_sort_pairs<NmSortConfig32>(nullptr, temp32, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                    (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0, 32,
                                    (hipStream_t)0);
    return temp > temp32 ? temp : temp32;
}

extern "C" size_t nm_voxelize_workspace_bytes(int64_t n)
{
    if (n < 1) n = 1;
    int64_t tiles = (n + UNIQ_TILE - 1) / UNIQ_TILE;
    return align_up((size_t)n * 8) * 2 + align_up(sort_keys_temp_bytes(n)) +
           align_up((size_t)tiles * 4) + 256;
}

extern "C" int nm_voxelize(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride,
                           const nm_lattice* lat, int64_t* d_addr_out, int64_t* d_count,
                           void* d_work, size_t work_bytes, void* stream)
{
    NM_ENTER_STREAM(ctx, stream);
    if (!d_xyz || !d_addr_out || !d_count || !d_work || n < 1 || stride < 3 ||
        n >= (int64_t)1 << 31)
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_voxelize: bad arguments");
    int rc = validate_lattice(ctx, lat);
    if (rc) return rc;
    if (work_bytes < nm_voxelize_workspace_bytes(n))
        NM_FAIL(ctx, NM_ERR_WORKSPACE, "nm_voxelize: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    LatticeDev L = make_lattice_dev(lat);
    char* w = (char*)d_work;
    uint64_t* addr = (uint64_t*)w;           w += align_up((size_t)n * 8);
    uint64_t* sorted = (uint64_t*)w;         w += align_up((size_t)n * 8);
    size_t temp_bytes = sort_keys_temp_bytes(n);
    void* temp = w;                          w += align_up(temp_bytes);
    int64_t tiles = (n + UNIQ_TILE - 1) / UNIQ_TILE;
    uint32_t* tile_count = (uint32_t*)w;

    NM_HIP(ctx, hipMemsetAsync(d_count, 0, 2 * sizeof(int64_t), s));
    k_addresses<<<(int)((n + 255) / 256), 256, 0, s>>>(d_xyz, n, stride, L, addr, d_count + 1);
    unsigned end_bit = (unsigned)(L.wx + L.wy + L.wz);
    NM_HIP(ctx, rocprim::radix_sort_keys(temp, temp_bytes, addr, sorted, (size_t)n, 0, end_bit, s));
    k_unique_count<<<(int)tiles, UNIQ_BLOCK, 0, s>>>(sorted, n, tile_count);
    k_tile_prefix<<<1, 1024, 0, s>>>(tile_count, tiles, d_count);
    k_unique_write<<<(int)tiles, UNIQ_BLOCK, 0, s>>>(sorted, n, tile_count, d_addr_out);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}

__global__ __launch_bounds__(256) void k_addr_to_coord(const int64_t* __restrict__ addr, int64_t m,
                                                       LatticeDev L, double* __restrict__ out)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= m) return;
    uint64_t a = (uint64_t)addr[i];
    // masks of geometry.py:74-79, shifts back of :131-132
    int32_t cx = (int32_t)(a & ((1ull << L.wx) - 1ull));
    int32_t cy = (int32_t)((a >> L.s0) & ((1ull << L.wy) - 1ull));
    int32_t cz = (int32_t)((a >> L.s1) & ((1ull << L.wz) - 1ull));
    out[i * 3 + 0] = nm_centre(cx, L.min_x, L.edge, L.half_edge);
    out[i * 3 + 1] = nm_centre(cy, L.min_y, L.edge, L.half_edge);
    out[i * 3 + 2] = nm_centre(cz, L.min_z, L.edge, L.half_edge);
}

extern "C" int nm_address_to_coordinate(nm_ctx* ctx, const int64_t* d_addr, int64_t m,
                                        const nm_lattice* lat, double* d_xyz_out, void* stream)
{
    NM_ENTER_STREAM(ctx, stream);
    if (m < 0 || (m > 0 && (!d_addr || !d_xyz_out)))
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_address_to_coordinate: bad arguments");
    int rc = validate_lattice(ctx, lat);
    if (rc) return rc;
    if (m == 0) return NM_OK;
    LatticeDev L = make_lattice_dev(lat);
    k_addr_to_coord<<<(int)((m + 255) / 256), 256, 0, (hipStream_t)stream>>>(d_addr, m, L,
                                                                             d_xyz_out);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}

// ---------------------------------------------------------------------------------------------------
// the per-scale occupancy index
// ---------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_cell_keys(const double* __restrict__ xyz, int64_t n,
                                                   int64_t stride, LatticeDev L,
                                                   uint64_t* __restrict__ key,
                                                   uint32_t* __restrict__ val)
{
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* p = xyz + i * stride;
    int32_t cx = nm_clamp_cell(nm_cell_f(p[0], L.min_x, L.edge));
    int32_t cy = nm_clamp_cell(nm_cell_f(p[1], L.min_y, L.edge));
    int32_t cz = nm_clamp_cell(nm_cell_f(p[2], L.min_z, L.edge));
    // points outside the lattice (a query cloud that is not the search cloud) are clamped for
    // ordering only; the search kernel recomputes their true cell.
    cx = min(max(cx, 0), (int32_t)((1u << L.wx) - 1u));
    cy = min(max(cy, 0), (int32_t)((1u << L.wy) - 1u));
    cz = min(max(cz, 0), (int32_t)((1u << L.wz) - 1u));
    key[i] = nm_cell_key((uint32_t)cx, (uint32_t)cy, (uint32_t)cz, L);
    val[i] = (uint32_t)i;
}

// both index kernels walk the sorted keys in contiguous chunks: a 256-thread block owns
// INDEX_CHUNK keys, each of its 4 waves a contiguous quarter, 64 keys per iteration.  counters are
// bumped once per block (same-address atomics serialise at ~10 ns each on this chip).
constexpr int INDEX_ITERS = 16;
constexpr int INDEX_WAVE_KEYS = 64 * INDEX_ITERS;        // 1024 keys per wave
constexpr int INDEX_CHUNK = 4 * INDEX_WAVE_KEYS;         // 4096 keys per block

// pass A: the first key of every superblock allocates a leaf, zeroes it and publishes key -> leaf in
// the hash table.
__global__ __launch_bounds__(256) void k_index_leaves(const uint64_t* __restrict__ skey, int64_t n,
                                                      IndexDev I)
{
    __shared__ uint32_t wtotal[4];
    __shared__ uint32_t wbase[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t wave_lo = (int64_t)blockIdx.x * INDEX_CHUNK + (int64_t)w * INDEX_WAVE_KEYS;
    // count the superblock heads of this wave's keys
    uint32_t total = 0;
    for (int it = 0; it < INDEX_ITERS; ++it) {
        const int64_t i = wave_lo + it * 64 + lane;
        bool head = false;
        if (i < n) {
            const uint64_t sb = skey[i] >> NM_LOCAL_BITS;
            head = (i == 0) || ((skey[i - 1] >> NM_LOCAL_BITS) != sb);
        }
        total += (uint32_t)__popcll(__ballot(head));
    }
    if (lane == 0) wtotal[w] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = wtotal[0] + wtotal[1] + wtotal[2] + wtotal[3];
        uint32_t base = t ? atomicAdd(&I.counters[0], t) : 0u;
        for (int ww = 0; ww < 4; ++ww) {
            wbase[ww] = base;
            base += wtotal[ww];
        }
    }
    __syncthreads();
    if (total == 0) return;
    uint32_t running = wbase[w];
    for (int it = 0; it < INDEX_ITERS; ++it) {
        const int64_t i = wave_lo + it * 64 + lane;
        bool head = false;
        uint64_t sb = 0;
        if (i < n) {
            sb = skey[i] >> NM_LOCAL_BITS;
            head = (i == 0) || ((skey[i - 1] >> NM_LOCAL_BITS) != sb);
        }
        const unsigned long long m = __ballot(head);
        if (head) {
            const uint32_t idx = running + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (idx >= I.leaf_capacity) {
                I.counters[2] = 1u;   // cannot happen when the workspace was sized by the library
                I.status[NM_ST_LEAF_OVERFLOW] = 1u;
            } else {
                uint4* leaf = (uint4*)(I.leaf + (size_t)idx * NM_LEAF_WORDS);
#pragma unroll
                for (int q = 0; q < NM_LEAF_WORDS / 4; ++q) leaf[q] = make_uint4(0u, 0u, 0u, 0u);
                uint32_t slot = nm_hash64(sb) & I.hash_mask;
                for (;;) {
                    const unsigned long long prev =
                        atomicCAS((unsigned long long*)&I.hash[slot].key,
                                  (unsigned long long)NM_HASH_EMPTY, (unsigned long long)sb);
                    if (prev == NM_HASH_EMPTY) {
                        I.hash[slot].val = idx;
                        break;
                    }
                    slot = (slot + 1) & I.hash_mask;
                }
            }
        }
        running += (uint32_t)__popcll(m);
    }
}

// pass B: set the occupancy bit of every key's cell.  per 64 keys: one hash lookup per run of equal
// superblock (broadcast to the run), a segmented OR over runs of equal row word, and one atomicOr per
// run (runs may continue in the next 64 keys, hence atomic).  distinct cells (= M) are counted on the
// way, one counter update per block.
__global__ __launch_bounds__(256) void k_index_bits(const uint64_t* __restrict__ skey, int64_t n,
                                                    IndexDev I)
{
    __shared__ uint32_t wcells[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t wave_lo = (int64_t)blockIdx.x * INDEX_CHUNK + (int64_t)w * INDEX_WAVE_KEYS;
    uint32_t cells = 0;
    for (int it = 0; it < INDEX_ITERS; ++it) {
        const int64_t i = wave_lo + it * 64 + lane;
        if (wave_lo + it * 64 >= n) break;
        const bool valid = i < n;
        const uint64_t k = valid ? skey[i] : 0ull;
        const uint64_t prev = (valid && i > 0) ? skey[i - 1] : ~k;
        const bool cell_head = valid && (k != prev);
        // run heads within this group of 64: lane 0 always starts a run
        const bool row_head = lane == 0 || (k >> NM_SBX_BITS) != (prev >> NM_SBX_BITS) || !valid;
        const bool sb_head = lane == 0 || (k >> NM_LOCAL_BITS) != (prev >> NM_LOCAL_BITS) || !valid;
        cells += (uint32_t)__popcll(__ballot(cell_head));
        const unsigned long long below = (2ull << lane) - 1ull;   // lanes <= this one
        // superblock -> leaf: looked up by the run's first lane, fetched by everyone in the run
        const unsigned long long sbm = __ballot(sb_head);
        int32_t leaf = -1;
        if (sb_head && valid) leaf = nm_hash_find(I, k >> NM_LOCAL_BITS);
        leaf = __shfl(leaf, 63 - __clzll((long long)(sbm & below)));
        // segmented OR of the bits of one row word
        const unsigned long long rowm = __ballot(row_head);
        const int seg_start = 63 - __clzll((long long)(rowm & below));
        uint32_t bits = valid ? (1u << ((uint32_t)k & 31u)) : 0u;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t other = __shfl_up(bits, off);
            if (lane - off >= seg_start) bits |= other;
        }
        // the last lane of a run holds the run's OR
        const bool tail = valid && (lane == 63 || ((rowm >> (lane + 1)) & 1ull) || i + 1 >= n);
        if (tail && leaf >= 0) {
            const uint32_t local = (uint32_t)k & ((1u << NM_LOCAL_BITS) - 1u);
            atomicOr(&I.leaf[(size_t)leaf * NM_LEAF_WORDS + (local >> NM_SBX_BITS)], bits);
        }
    }
    if (lane == 0) wcells[w] = cells;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = wcells[0] + wcells[1] + wcells[2] + wcells[3];
        if (t) atomicAdd(&I.counters[1], t);
    }
}

static uint64_t lattice_superblocks(const LatticeDev& L)
{
    // bx+by+bz <= 64 - 11, so this fits
    return 1ull << (L.bx + L.by + L.bz);
}

void nm_index_layout(const LatticeDev& L, int64_t n_search, IndexLayout* out)
{
    uint64_t cap = lattice_superblocks(L);
    if (cap > (uint64_t)n_search) cap = (uint64_t)n_search;
    if (cap < 1) cap = 1;
    uint64_t hcap = 64;
    while (hcap < cap * 2) hcap <<= 1;
    out->leaf_capacity = (uint32_t)cap;
    out->hash_capacity = (uint32_t)hcap;
    out->hash_bytes = align_up((size_t)hcap * sizeof(HashEntry));
    out->leaf_bytes = align_up((size_t)cap * NM_LEAF_WORDS * 4);
    out->counter_bytes = 256;
    out->total = out->hash_bytes + out->leaf_bytes + out->counter_bytes;
}

int nm_sort_cells(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride, const LatticeDev& L,
                  uint64_t* key_tmp, uint32_t* val_tmp, uint64_t* key_sorted, uint32_t* val_sorted,
                  void* sort_temp, size_t sort_temp_bytes, hipStream_t s)
{
    k_cell_keys<<<(int)((n + 255) / 256), 256, 0, s>>>(d_xyz, n, stride, L, key_tmp, val_tmp);
    NM_HIP(ctx, rocprim::radix_sort_pairs<NmSortConfig>(sort_temp, sort_temp_bytes, key_tmp, key_sorted, val_tmp,
                                          val_sorted, (size_t)n, 0, (unsigned)L.keybits, s));
    return NM_OK;
}

int nm_index_build(nm_ctx* ctx, const uint64_t* key_sorted, int64_t n, const IndexLayout& lay,
                   void* index_mem, IndexDev* out, hipStream_t s)
{
    char* w = (char*)index_mem;
    IndexDev I;
    I.hash = (HashEntry*)w;         w += lay.hash_bytes;
    I.leaf = (uint32_t*)w;          w += lay.leaf_bytes;
    I.counters = (uint32_t*)w;
    I.hash_mask = lay.hash_capacity - 1;
    I.leaf_capacity = lay.leaf_capacity;
    I.status = ctx->d_status;
    NM_HIP(ctx, hipMemsetAsync(I.hash, 0xFF, (size_t)lay.hash_capacity * sizeof(HashEntry), s));
    NM_HIP(ctx, hipMemsetAsync(I.counters, 0, 256, s));
    int blocks = (int)((n + INDEX_CHUNK - 1) / INDEX_CHUNK);
    k_index_leaves<<<blocks, 256, 0, s>>>(key_sorted, n, I);
    k_index_bits<<<blocks, 256, 0, s>>>(key_sorted, n, I);
    NM_HIP(ctx, hipGetLastError());
    *out = I;
    return NM_OK;
}


// ---------------------------------------------------------------------------------------------------
// index build from a spatially coherent (but not sorted-at-this-scale) key stream.
//
// the whole-ladder path sorts the cloud once, by its cell keys at the finest scale, and keeps a copy of
// the coordinates in that order.  at every other scale the keys of that stream are no longer sorted,
// but neighbours in the stream are still neighbours in space, so runs of equal superblock / equal row
// word are long.  duplicates are therefore squeezed out inside each wave (compare with the previous
// key), and what is left goes through the hash table with atomics:
//   k_index_fused        phase 1: cell keys of this scale; run heads CAS their superblock key into the
//                        table; a block's winners get leaf numbers from ONE counter bump per block, zero
//                        their leaf and publish it.  phase 2: run tails OR the run's bits into the
//                        leaf, through a per-block table in LDS (no return value awaited)
//   k_count_voxels_all   M = set bits of the allocated leaves, all scales of a ladder in one launch
// ---------------------------------------------------------------------------------------------------

__device__ __forceinline__ uint64_t nm_point_key(const double* __restrict__ p, const LatticeDev& L)
{
    const int32_t cx = nm_cell_index(p[0], L.min_x, L.edge, L.inv_edge, 0, (int32_t)((1u << L.wx) - 1u));
    const int32_t cy = nm_cell_index(p[1], L.min_y, L.edge, L.inv_edge, 0, (int32_t)((1u << L.wy) - 1u));
    const int32_t cz = nm_cell_index(p[2], L.min_z, L.edge, L.inv_edge, 0, (int32_t)((1u << L.wz) - 1u));
    return nm_cell_key((uint32_t)cx, (uint32_t)cy, (uint32_t)cz, L);
}

constexpr uint32_t BITS_EMPTY = 0xFFFFFFFFu;     // free slot of a block's row-word table

// inclusive OR over the lanes [lane - dist, lane] of a wave (dist = distance to the head of the lane's run),
// in the vector ALU: four row shifts inside the rows of 16 lanes, then row_bcast15 / row_bcast31 carry lane 15 /
// 47 into the next row and lane 31 into the upper half - for the lanes whose run reaches that far back.
// (as six __shfl_up it was six round trips through the LDS crossbar, each waited for)
__device__ __forceinline__ uint32_t nm_run_or(uint32_t b, int dist, int lane)
{
#define NM_RUN_STEP(ctrl, rowmask, cond)                                                                   \
    {                                                                                                      \
        const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, ctrl, rowmask, 0xF, true);     \
        b |= (cond) ? o : 0u;                                                                              \
    }
    NM_RUN_STEP(0x111, 0xF, dist >= 1)                 // row_shr:1
    NM_RUN_STEP(0x112, 0xF, dist >= 2)                 // row_shr:2
    NM_RUN_STEP(0x114, 0xF, dist >= 4)                 // row_shr:4
    NM_RUN_STEP(0x118, 0xF, dist >= 8)                 // row_shr:8
    NM_RUN_STEP(0x142, 0xA, dist > (lane & 15))        // row_bcast15 into rows 1 and 3
    NM_RUN_STEP(0x143, 0xC, dist > (lane & 31))        // row_bcast31 into rows 2 and 3
#undef NM_RUN_STEP
    return b;
}

// ---- the builder: both passes in one kernel -----------------------------------------------------------
// it is a chain of dependent memory round trips (coordinates -> key -> table probe -> CAS; leaf number
// -> atomic), so every wave works on FUSED_GROUPS independent groups of 64 points at a time, and the
// block is kept small enough in LDS for the register budget to bound the occupancy.
// (history: as two kernels - insert, then bits - the keys went through memory in between, 160 MB per
// scale at 10 M points, and the second kernel probed the table again for every run: 0.95 ms for the five
// scales of the benchmark against 0.69 ms now.)  the block keeps what the bit phase needs about its
// points in LDS - the table slot of the point's superblock and the cell's 11 local bits - so the second
// phase touches memory only for the leaf numbers and the bits themselves.
//
// a leaf number may belong to a superblock another block entered and has not published yet.  the
// reader then waits for it.  that cannot deadlock: whoever entered a key is a running block, and a
// block publishes its leaves before its own bit phase, i.e. before it ever waits for anybody.  the
// wait is bounded all the same (counters[3] flags a timeout; the library then reports M = -1).
#ifndef NM_SPIN_LIMIT
#define NM_SPIN_LIMIT (1 << 22)
#endif
constexpr uint32_t LEAF_PENDING = 0xFFFFFFFFu;     // what the 0xFF-filled table holds before publication
constexpr uint32_t LEAF_NONE = 0xFFFFFFFEu;        // published: no room (capacity overflow)

// a block owns FUSED_CHUNK points.  LDS per block: 4 B per point of stash + 4 B per point of scratch
// (list of created slots, then the table of row words): 16 KB at 2048 points, so the register
// budget (6 waves per SIMD), not LDS, bounds the occupancy; the kernel is a chain of dependent
// memory operations and lives on the number of waves in flight.
#ifndef NM_FUSED_ITERS
#define NM_FUSED_ITERS 4      // = groups: a wave reads its 256 points once for all densely indexed scales
#endif
#ifndef NM_FUSED_GROUPS
#define NM_FUSED_GROUPS 4
#endif
constexpr int FUSED_ITERS = NM_FUSED_ITERS;
constexpr int FUSED_GROUPS = NM_FUSED_GROUPS;   // groups of 64 points a wave has in flight
constexpr int FUSED_WAVE_KEYS = 64 * FUSED_ITERS;
constexpr int FUSED_CHUNK = 4 * FUSED_WAVE_KEYS;
constexpr int FUSED_TABLE = FUSED_CHUNK / 2;
constexpr int FUSED_TABLE_BITS = FUSED_ITERS == 16 ? 11 : (FUSED_ITERS == 8 ? 10 : 9);
static_assert(FUSED_ITERS % FUSED_GROUPS == 0 && (1 << FUSED_TABLE_BITS) == FUSED_TABLE, "fused build geometry");

// GATHER: the coordinates are not in sorted order yet - `xyz` is the cloud as the caller holds it (row stride
// `stride`), `order` the permutation the spatial sort has just written: the wave gathers its 256 points, writes
// them to `sorted` (the copy the search kernel reads, (n,3) contiguous) and keeps them in registers for its own
// work.  one pass over the cloud less than a gather kernel followed by a builder that reads its output
template <bool GATHER>
__global__ __launch_bounds__(256) void k_index_fused(const double* __restrict__ xyz_in, int64_t n,
                                                     int64_t stride, const uint32_t* __restrict__ order,
                                                     double* __restrict__ sorted,
                                                     const ScaleDev* __restrict__ ladder,
                                                     int32_t n_scales)
{
    const double* __restrict__ xyz = GATHER ? sorted : xyz_in;     // what the hash-form pass re-reads
    __shared__ uint32_t stash_slot[FUSED_CHUNK];     // table slot of the point's superblock
    __shared__ uint16_t stash_local[FUSED_CHUNK];    // the cell's 11 bits inside the superblock
    // phase 1 needs the list of created slots, phase 2 the bit table: same memory
    __shared__ uint32_t scratch[2 * FUSED_TABLE];
    __shared__ uint32_t won_count;
    __shared__ uint32_t leaf_base;
    static_assert(2 * FUSED_TABLE >= FUSED_CHUNK, "the list of created slots must fit the scratch");
    uint32_t* won_slot = scratch;
    uint32_t* t_word = scratch;
    uint32_t* t_bits = scratch + FUSED_TABLE;
    // one launch builds every index of the ladder: the block keeps its points and walks the scales; the
    // launch has one ramp and one tail instead of one per scale
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t wave_lo = (int64_t)blockIdx.x * FUSED_CHUNK + (int64_t)w * FUSED_WAVE_KEYS;
    const unsigned long long below = (2ull << lane) - 1ull;

    // ---- pass 1: the densely indexed scales (leaf = superblock key: no table, nothing to insert, nothing to
    //      wait for).  the wave's points are read ONCE into registers and serve every dense scale: cell key ->
    //      (row word of the superblock's own leaf, bit); runs of equal row word are OR-ed inside the wave, the
    //      block's distinct words meet in the LDS table and leave as one atomicOr each.
    bool any_dense = false;
    for (int32_t sc = 0; sc < n_scales; ++sc)
        any_dense = any_dense || (ladder[sc].valid && !ladder[sc].shared && ladder[sc].I.hash == nullptr);
    static_assert(FUSED_ITERS == FUSED_GROUPS, "a wave reads its points once");
    if (any_dense || GATHER) {
        for (int it = 0; it < FUSED_ITERS; it += FUSED_GROUPS) {      // (trip count is block-uniform: barriers)
            const int64_t base = wave_lo + (int64_t)it * 64;
            double px[FUSED_GROUPS], py[FUSED_GROUPS], pz[FUSED_GROUPS];
            bool valid[FUSED_GROUPS];
#pragma unroll
            for (int g = 0; g < FUSED_GROUPS; ++g) {
                const int64_t i = base + g * 64 + lane;
                valid[g] = i < n;
                px[g] = py[g] = pz[g] = 0.0;
                if (valid[g]) {
                    const double* p = GATHER ? xyz_in + (int64_t)order[i] * stride : xyz_in + i * 3;
                    px[g] = p[0];
                    py[g] = p[1];
                    pz[g] = p[2];
                }
            }
            if (GATHER) {
#pragma unroll
                for (int g = 0; g < FUSED_GROUPS; ++g) {
                    const int64_t i = base + g * 64 + lane;
                    if (valid[g]) {
                        sorted[i * 3 + 0] = px[g];
                        sorted[i * 3 + 1] = py[g];
                        sorted[i * 3 + 2] = pz[g];
                    }
                }
            }
#pragma nounroll
            for (int32_t sc = 0; sc < n_scales; ++sc) {
                if (!ladder[sc].valid || ladder[sc].shared || ladder[sc].I.hash != nullptr) continue;
                const LatticeDev L = ladder[sc].L;
                const IndexDev I = ladder[sc].I;
                __syncthreads();       // the previous flush has read the table
                for (int t = threadIdx.x; t < FUSED_TABLE; t += blockDim.x) {
                    t_word[t] = BITS_EMPTY;
                    t_bits[t] = 0u;
                }
                __syncthreads();
#ifdef NM_DIAG_FORCE_TIMEOUT
                // diagnostic build (tests only): behave as if the hash form's bounded wait had run out
                if (blockIdx.x == 0 && threadIdx.x == 0) {
                    I.counters[3] = 1u;
                    I.status[NM_ST_INDEX_TIMEOUT] = 1u;
                }
#endif
#pragma unroll
                for (int g = 0; g < FUSED_GROUPS; ++g) {
                    uint32_t word = BITS_EMPTY, b = 0u;
                    if (valid[g]) {
                        const double p[3] = {px[g], py[g], pz[g]};
                        const uint64_t k = nm_point_key(p, L);
                        word = (uint32_t)(k >> NM_SBX_BITS);      // superblock * 64 + row inside it
                        b = 1u << ((uint32_t)k & 31u);
                    }
                    // (lane 0 keeps its own word in `prev`: it is a head anyway)
                    const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)word, (int)word, 0x138, 0xF, 0xF,
                                                                                 false);      // wave_shr:1
                    const bool row_head = lane == 0 || word != prev;
                    const unsigned long long rowm = __ballot(row_head);
                    const int seg_start = 63 - __clzll((long long)(rowm & below));
                    b = nm_run_or(b, lane - seg_start, lane);
                    const bool tail = valid[g] && (lane == 63 || ((rowm >> (lane + 1)) & 1ull));
                    if (tail) {
                        uint32_t ts = (word * 0x9E3779B1u) >> (32 - FUSED_TABLE_BITS);
                        bool stored = false;
#pragma unroll 1
                        for (int probe = 0; probe < 8; ++probe) {
                            const uint32_t seen = atomicCAS(&t_word[ts], BITS_EMPTY, word);
                            if (seen == BITS_EMPTY || seen == word) {
                                atomicOr(&t_bits[ts], b);
                                stored = true;
                                break;
                            }
                            ts = (ts + 1) & (FUSED_TABLE - 1);
                        }
                        if (!stored) atomicOr(&I.leaf[word], b);
                    }
                }
                __syncthreads();
                for (int t = threadIdx.x; t < FUSED_TABLE; t += blockDim.x) {
                    const uint32_t wd = t_word[t];
                    if (wd != BITS_EMPTY) atomicOr(&I.leaf[wd], t_bits[t]);
                }
            }
        }
    }

    // ---- pass 2: the scales with a directory (hash form)
#pragma nounroll
    for (int32_t sc = 0; sc < n_scales; ++sc) {
        if (!ladder[sc].valid || ladder[sc].shared) continue;      // (block-uniform)
        const LatticeDev L = ladder[sc].L;
        const IndexDev I = ladder[sc].I;
        const bool dense = I.hash == nullptr;
        const bool mapf = I.map != nullptr;
        __syncthreads();       // the previous scale's flush has read the scratch
        if (dense) continue;       // done in the first pass
        if (threadIdx.x == 0) won_count = 0u;
        __syncthreads();

        // ---- phase 1: keys, run heads into the table, every point's (slot, local) into the stash
        uint64_t carry = ~0ull;          // superblock of the previous point of this wave; none at its start:
        uint32_t carry_slot = 0u;        // the first point of a wave always probes
        for (int it = 0; it < FUSED_ITERS; it += FUSED_GROUPS) {
            const int64_t base = wave_lo + (int64_t)it * 64;
            if (base >= n) break;
            uint64_t sb[FUSED_GROUPS];
            uint32_t local[FUSED_GROUPS];
            bool valid[FUSED_GROUPS], head[FUSED_GROUPS];
            uint32_t slot[FUSED_GROUPS];
            uint64_t peek[FUSED_GROUPS];
    #pragma unroll
            for (int g = 0; g < FUSED_GROUPS; ++g) {
                const int64_t i = base + g * 64 + lane;
                valid[g] = i < n;
                uint64_t k = ~0ull;
                if (valid[g]) k = nm_point_key(xyz + i * 3, L);
                sb[g] = k >> NM_LOCAL_BITS;
                local[g] = (uint32_t)k & ((1u << NM_LOCAL_BITS) - 1u);
                uint64_t prev = __shfl_up(sb[g], 1);
                if (lane == 0) prev = carry;
                carry = __shfl(sb[g], 63);
                head[g] = valid[g] && sb[g] != prev;
            }
    #pragma unroll
            for (int g = 0; g < FUSED_GROUPS; ++g) {
                // dense index: the "slot" of a superblock is its key, which also is its leaf; nothing to probe
                slot[g] = dense || mapf ? (uint32_t)sb[g] : nm_hash64(sb[g]) & I.hash_mask;
                peek[g] = (head[g] && !dense && !mapf) ? I.hash[slot[g]].key : sb[g];
            }
            if (mapf) {
                // map form: the superblock's own word.  whoever turns it from EMPTY to PENDING creates the leaf
#pragma unroll
                for (int g = 0; g < FUSED_GROUPS; ++g) {
                    if (!head[g]) continue;
                    uint32_t v = __hip_atomic_load(&I.map[slot[g]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v == NM_MAP_EMPTY) {
                        v = atomicCAS(&I.map[slot[g]], NM_MAP_EMPTY, NM_MAP_PENDING);
                        if (v == NM_MAP_EMPTY) won_slot[atomicAdd(&won_count, 1u)] = slot[g];
                    }
                }
            }
    #pragma unroll
            for (int g = 0; g < FUSED_GROUPS; ++g) {
                if (!head[g] || peek[g] == sb[g]) continue;      // already in the table, at slot[g]
                uint32_t sl = slot[g];
                uint64_t pk = peek[g];
                for (;;) {
                    if (pk == sb[g]) break;
                    if (pk != NM_HASH_EMPTY) {
                        sl = (sl + 1) & I.hash_mask;
                        pk = I.hash[sl].key;
                        continue;
                    }
                    const unsigned long long seen =
                        atomicCAS((unsigned long long*)&I.hash[sl].key,
                                  (unsigned long long)NM_HASH_EMPTY, (unsigned long long)sb[g]);
                    if (seen == NM_HASH_EMPTY) {
                        won_slot[atomicAdd(&won_count, 1u)] = sl;
                        break;
                    }
                    if (seen == sb[g]) break;
                    sl = (sl + 1) & I.hash_mask;
                    pk = I.hash[sl].key;
                }
                slot[g] = sl;
            }
            // the slot of a point that is not a head is its run head's
    #pragma unroll
            for (int g = 0; g < FUSED_GROUPS; ++g) {
                const unsigned long long hm = __ballot(head[g]) & below;
                const int src = hm ? 63 - __clzll((long long)hm) : 0;
                uint32_t sl = __shfl(slot[g], src);
                if (!hm) sl = carry_slot;
                carry_slot = __shfl(sl, 63);
                stash_slot[w * FUSED_WAVE_KEYS + (it + g) * 64 + lane] = sl;
                stash_local[w * FUSED_WAVE_KEYS + (it + g) * 64 + lane] = (uint16_t)local[g];
            }
        }
        __syncthreads();
        // ---- the block's new leaves: one counter bump, zeroed, then published
        const uint32_t total = won_count;
        if (total) {
            if (threadIdx.x == 0) leaf_base = atomicAdd(&I.counters[0], total);
            __syncthreads();
            {
                // the block's leaves are one contiguous run: zeroed by all threads side by side, every store
                // instruction of a wave a full 512 bytes (one thread per leaf wrote 8 bytes each to 64 different
                // leaves per instruction: 32 partial writes per leaf on the memory side).
                // device-scope stores: they go through to memory, where the other blocks' atomics on these
                // leaves will execute (a plain store would sit in this XCD's L2 until a release fence writes
                // the whole L2 back - measured: 8x slower kernel)
                const uint32_t usable = leaf_base < I.leaf_capacity
                                            ? (total < I.leaf_capacity - leaf_base ? total : I.leaf_capacity - leaf_base)
                                            : 0u;
                unsigned long long* z = (unsigned long long*)(I.leaf + (size_t)leaf_base * NM_LEAF_WORDS);
                for (uint32_t e = threadIdx.x; e < usable * (NM_LEAF_WORDS / 2); e += blockDim.x)
                    __hip_atomic_store(z + e, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (usable < total && threadIdx.x == 0) {
                    I.counters[2] = 1u;
                    I.status[NM_ST_LEAF_OVERFLOW] = 1u;
                }
            }
            // the zeroes must have arrived before anyone can learn the leaf number
            __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < total; t += blockDim.x) {
                const uint32_t idx = leaf_base + t;
                if (mapf)
                    __hip_atomic_store(&I.map[won_slot[t]], idx < I.leaf_capacity ? idx : NM_MAP_NONE,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else
                    __hip_atomic_store(&I.hash[won_slot[t]].val, idx < I.leaf_capacity ? idx : LEAF_NONE,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        // ---- phase 2: the bits.  (the scratch now is the table of row words)
        for (int t = threadIdx.x; t < FUSED_TABLE; t += blockDim.x) {
            t_word[t] = BITS_EMPTY;
            t_bits[t] = 0u;
        }
        __syncthreads();
        for (int it = 0; it < FUSED_ITERS; it += FUSED_GROUPS) {
            if (wave_lo + (int64_t)it * 64 >= n) break;
            uint32_t sl[FUSED_GROUPS], loc[FUSED_GROUPS];
            bool valid[FUSED_GROUPS], row_head[FUSED_GROUPS], sb_head[FUSED_GROUPS];
            uint32_t val[FUSED_GROUPS];
    #pragma unroll
            for (int g = 0; g < FUSED_GROUPS; ++g) {
                sl[g] = stash_slot[w * FUSED_WAVE_KEYS + (it + g) * 64 + lane];
                loc[g] = stash_local[w * FUSED_WAVE_KEYS + (it + g) * 64 + lane];
                valid[g] = wave_lo + (int64_t)(it + g) * 64 + lane < n;
                const uint32_t prev_sl = __shfl_up(sl[g], 1);
                const uint32_t prev_loc = __shfl_up(loc[g], 1);
                // runs are delimited inside one group of 64 only: lane 0 always starts one
                sb_head[g] = lane == 0 || sl[g] != prev_sl || !valid[g];
                row_head[g] = sb_head[g] || (loc[g] >> NM_SBX_BITS) != (prev_loc >> NM_SBX_BITS);
                val[g] = LEAF_NONE;
                if (sb_head[g] && valid[g])
                    val[g] = dense ? sl[g]
                             : mapf ? __hip_atomic_load(&I.map[sl[g]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                    : __hip_atomic_load(&I.hash[sl[g]].val, __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT);
            }
    #pragma unroll
            for (int g = 0; g < FUSED_GROUPS; ++g) {
                if (sb_head[g] && valid[g]) {
                    // not published yet: its creator is still in phase 1.  (a dense index has no creators)
                    const uint32_t pending = mapf ? NM_MAP_PENDING : LEAF_PENDING;
                    for (int spin = 0; !dense && val[g] == pending && spin < NM_SPIN_LIMIT; ++spin) {
                        __builtin_amdgcn_s_sleep(8);
                        val[g] = mapf ? __hip_atomic_load(&I.map[sl[g]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                      : __hip_atomic_load(&I.hash[sl[g]].val, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT);
                    }
    #ifdef NM_DIAG_FORCE_TIMEOUT
                    // diagnostic build (tests only): the first block behaves as if its first wait ran out
                    if (blockIdx.x == 0 && it == 0 && g == 0) val[g] = pending;
    #endif
                    if (val[g] == pending) {
                        I.counters[3] = 1u;
                        I.status[NM_ST_INDEX_TIMEOUT] = 1u;     // sticky: the next call into the library fails
                        val[g] = LEAF_NONE;
                    }
                }
                const unsigned long long sbm = __ballot(sb_head[g]);
                const int32_t leaf = (int32_t)__shfl(val[g], 63 - __clzll((long long)(sbm & below)));
                const unsigned long long rowm = __ballot(row_head[g]);
                const int seg_start = 63 - __clzll((long long)(rowm & below));
                uint32_t bits = valid[g] ? (1u << (loc[g] & 31u)) : 0u;
    #pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t other = __shfl_up(bits, off);
                    if (lane - off >= seg_start) bits |= other;
                }
                const bool tail = valid[g] && (lane == 63 || ((rowm >> (lane + 1)) & 1ull));
                if (tail && leaf >= 0) {
                    const uint32_t word = (uint32_t)leaf * NM_LEAF_WORDS + (loc[g] >> NM_SBX_BITS);
                    uint32_t ts = (word * 0x9E3779B1u) >> (32 - FUSED_TABLE_BITS);
                    bool stored = false;
    #pragma unroll 1
                    for (int probe = 0; probe < 8; ++probe) {
                        const uint32_t seen = atomicCAS(&t_word[ts], BITS_EMPTY, word);
                        if (seen == BITS_EMPTY || seen == word) {
                            atomicOr(&t_bits[ts], bits);
                            stored = true;
                            break;
                        }
                        ts = (ts + 1) & (FUSED_TABLE - 1);
                    }
                    if (!stored) atomicOr(&I.leaf[word], bits);
                }
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < FUSED_TABLE; t += blockDim.x) {
            const uint32_t word = t_word[t];
            if (word != BITS_EMPTY) atomicOr(&I.leaf[word], t_bits[t]);
        }
    }
}

// ---- the ladder in device memory ------------------------------------------------------------------------
// scalar lattice parameters of VoxelFilter.__init__ / _calculate_shift (geometry.py:37-64) on the device:
//   min_corner = min - e/2, max_corner = max + e/2                       geometry.py:37-38
//   widths = ceil(log2((max_corner - min_corner) / e))                   geometry.py:56
// numpy's log2 is an fp64 function: ceil(fl(log2 x)).  for x = 2^k (1 + f) that is k + 1 unless the sum
// k + f/ln 2 rounds back to k in fp64 - only for f below 2^-40 or so - which is reproduced here from the
// exponent and the mantissa of x, without a log2 in device code.
__device__ inline int32_t nm_ceil_log2(double x)
{
    if (!(x > 0.0) || isinf(x)) return -10000;
    int ex = 0;
    const double m = frexp(x, &ex);          // x = m * 2^ex, m in [0.5, 1)
    const int k = ex - 1;                    // x = (2m) * 2^k, 2m in [1, 2)
    const double f = 2.0 * m - 1.0;          // exact
    if (f == 0.0) return k;
    if (f < 0x1p-30) {
        const double l = (double)k + f * 1.4426950408889634;     // fl(log2 x) to first order in f
        return (int32_t)ceil(l);
    }
    return k + 1;
}

struct LadderSpec {
    int32_t n_scales;
    int32_t finest;                    // scale whose lattice orders the cloud
    double edge[NM_MAX_LADDER];
    double radius[NM_MAX_LADDER];
    // where each scale's index lives (host-computed from the point count alone)
    HashEntry* hash[NM_MAX_LADDER];
    uint32_t* leaf[NM_MAX_LADDER];
    uint32_t* counters[NM_MAX_LADDER];
    uint32_t hash_capacity;            // slots allocated per scale (power of two)
    uint32_t leaf_capacity;            // leaves allocated per scale
    int64_t order_points;              // nm_order_plan's point count
};

__device__ inline void nm_scale_finish(ScaleDev* S, double radius, uint32_t hash_capacity,
                                       uint32_t leaf_capacity, bool allow_dense)
{
    const LatticeDev& L = S->L;
    const int sb_bits = L.bx + L.by + L.bz;
    if (allow_dense && sb_bits <= NM_DENSE_LOG2 && (1ull << sb_bits) <= (uint64_t)leaf_capacity) {
        // few superblocks (the coarse scales): one leaf each, no directory
        S->I.hash = nullptr;
        S->I.hash_mask = (1u << sb_bits) - 1u;
        S->I.leaf_capacity = 1u << sb_bits;
        leaf_capacity = 0;          // skip the sizing below
    }
    S->I.map = nullptr;
#if NM_MAP_FORM
    // a word per superblock where the table's room holds that (4 B x superblocks <= 16 B x slots): MAP form
    if (allow_dense && S->I.hash && sb_bits <= 30 && (1ull << sb_bits) <= 4ull * hash_capacity) {
        uint64_t cap = 1ull << sb_bits;
        if (cap > leaf_capacity) cap = leaf_capacity;
        S->I.map = (uint32_t*)S->I.hash;
        S->I.hash_mask = (uint32_t)((1ull << sb_bits) - 1ull);
        S->I.leaf_capacity = (uint32_t)cap;
    } else
#endif
    // as many leaves as the lattice has superblocks or the cloud has points, whichever is smaller; the
    // table is kept at most half full
    if (S->I.hash) {
        uint64_t cap = 1ull << sb_bits;
        if (cap > leaf_capacity) cap = leaf_capacity;
        if (cap < 1) cap = 1;
        uint64_t hcap = 64;
        while (hcap < cap * 2 && hcap < hash_capacity && hcap < (1ull << 31)) hcap <<= 1;
        S->I.hash_mask = (uint32_t)(hcap - 1);
        S->I.leaf_capacity = (uint32_t)cap;
    }
    S->r2 = radius * radius;
    // static pruning of the candidate window is sound only while the rounding of cells and centres stays
    // far below the 1e-4-cell padding of the bounds: 16 ulp of the largest coordinate the lattice can
    // produce must be smaller than that
    double maxabs = 0.0;
    const double mins[3] = {L.min_x, L.min_y, L.min_z};
    const int32_t ws[3] = {L.wx, L.wy, L.wz};
    for (int a = 0; a < 3; ++a) {
        const double lo = mins[a];
        const double hi = lo + ldexp(L.edge, ws[a]);
        maxabs = fmax(maxabs, fmax(fabs(lo), fabs(hi)));
    }
    S->prune_ok = 16.0 * maxabs * 2.220446049250313e-16 < 1e-4 * L.edge ? 1 : 0;
}

// from the cloud's extrema (6 doubles on the device) to every scale's lattice.  one thread per scale.
// `partial` set: the extrema are still the per-block pieces of the bounds pass (k_bounds); this kernel folds
// them first (what k_bounds_finish does as a launch of its own) and leaves them in `minmax_out`.
constexpr int LADDER_THREADS = 256;     // the fold of up to 1024 per-block extrema is a chain of loads: four waves
                                        // take four rounds where one took sixteen (10 -> 6 us at 10 M points)
__global__ __launch_bounds__(LADDER_THREADS) void k_make_ladder(const double* __restrict__ minmax_in,
                                                    const uint64_t* __restrict__ partial, int blocks,
                                                    double* __restrict__ minmax_out, LadderSpec P,
                                                    ScaleDev* __restrict__ ladder, OrderDev* __restrict__ order_dev,
                                                    uint32_t* __restrict__ status)
{
    const int sc = threadIdx.x;
    double minmax[6];
    if (partial) {
        uint64_t v[6];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            v[a] = ~0ull;
            v[3 + a] = 0ull;
        }
        for (int b = sc; b < blocks; b += LADDER_THREADS) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const uint64_t l = partial[b * 6 + a], h = partial[b * 6 + 3 + a];
                v[a] = l < v[a] ? l : v[a];
                v[3 + a] = h > v[3 + a] ? h : v[3 + a];
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const uint64_t l = (uint64_t)__shfl_xor((unsigned long long)v[a], off);
                const uint64_t h = (uint64_t)__shfl_xor((unsigned long long)v[3 + a], off);
                v[a] = l < v[a] ? l : v[a];
                v[3 + a] = h > v[3 + a] ? h : v[3 + a];
            }
        }
        // the four waves' extrema meet in LDS; every thread folds them again (the scale threads need them all)
        __shared__ uint64_t wave_v[LADDER_THREADS / 64][6];
        if ((sc & 63) == 0) {
#pragma unroll
            for (int a = 0; a < 6; ++a) wave_v[sc >> 6][a] = v[a];
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            for (int w = 0; w < LADDER_THREADS / 64; ++w) {
                const uint64_t l = wave_v[w][a], h = wave_v[w][3 + a];
                v[a] = l < v[a] ? l : v[a];
                v[3 + a] = h > v[3 + a] ? h : v[3 + a];
            }
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) minmax[a] = nm_order_decode(v[a]);
        if (sc == 0) {
#pragma unroll
            for (int a = 0; a < 6; ++a) minmax_out[a] = minmax[a];
        }
    } else {
#pragma unroll
        for (int a = 0; a < 6; ++a) minmax[a] = minmax_in[a];
    }
    if (sc < P.n_scales) {
        ScaleDev S;
        const double e = P.edge[sc];
        const double half = e / 2;
        int32_t w[3];
        double mn[3];
        int32_t bad = 0;
        int sum = 0;
        for (int a = 0; a < 3; ++a) {
            const double lo = minmax[a], hi = minmax[3 + a];
            if (!(lo <= hi) || isinf(lo) || isinf(hi)) bad = NM_LAT_NOT_FINITE;
            mn[a] = lo - half;
            const double mx = hi + half;
            w[a] = nm_ceil_log2((mx - mn[a]) / e);
            sum += w[a];
        }
        if (!bad) {
            if (sum > 64) bad = NM_LAT_TOO_SMALL_EDGE;                       // geometry.py:59-60
            else if (w[0] < 1 || w[1] < 1 || w[2] < 1) bad = NM_LAT_NO_EXTENT;
            else if (w[0] > 30 || w[1] > 30 || w[2] > 30) bad = NM_LAT_DEVICE_LIMIT;
        }
        if (bad) {
            for (int a = 0; a < 3; ++a) w[a] = 1;                            // harmless placeholders
            status[NM_ST_LATTICE] = (uint32_t)bad;
        }
        LatticeDev& L = S.L;
        L.min_x = mn[0]; L.min_y = mn[1]; L.min_z = mn[2];
        L.edge = e;
        L.half_edge = e * 0.5;
        L.inv_edge = 1.0 / e;
        L.wx = w[0]; L.wy = w[1]; L.wz = w[2];
        L.s0 = w[0];
        L.s1 = w[0] + w[1];
        L.bx = L.wx > NM_SBX_BITS ? L.wx - NM_SBX_BITS : 0;
        L.by = L.wy > NM_SBY_BITS ? L.wy - NM_SBY_BITS : 0;
        L.bz = L.wz > NM_SBZ_BITS ? L.wz - NM_SBZ_BITS : 0;
        L.keybits = NM_LOCAL_BITS + L.bx + L.by + L.bz;
        // a scale with the edge length of an earlier one has that scale's lattice: it borrows its index
        int owner = sc;
        for (int j = sc - 1; j >= 0; --j)
            if (P.edge[j] == e) owner = j;
        S.I.hash = P.hash[owner];
        S.I.leaf = P.leaf[owner];
        S.I.counters = P.counters[owner];
        S.I.status = status;
        S.stats = P.counters[sc];
        S.shared = owner != sc;
        S.reserved = 0;
        S.valid = bad ? 0 : 1;
        nm_scale_finish(&S, P.radius[sc], P.hash_capacity, P.leaf_capacity, !bad);
        ladder[sc] = S;
        if (sc == P.finest) {
            OrderDev O;
            nm_order_plan(S.L, &O, P.order_points);
            O.valid = S.valid;
            *order_dev = O;
        }
    }
}

// the same array from lattices the host already has (nm_multiscale_features, nm_scale_features)
struct LadderPut {
    int32_t allow_dense;               // the ladder path may index coarse scales densely; nm_scale_features not
    int32_t n_scales;
    int32_t first;                     // index of P.scale[0] in the device array
    int32_t finest;                    // index (in the device array) of the ordering scale, -1: not here
    ScaleDev scale[8];                 // (stats / shared filled in by the host: nm_ladder_put)
    double radius[8];
    uint32_t hash_capacity[8], leaf_alloc[8];      // slots of the table, leaves the workspace has room for
    int64_t order_points;              // nm_order_plan's point count
};

__global__ void k_put_ladder(LadderPut P, ScaleDev* __restrict__ ladder, OrderDev* __restrict__ order_dev)
{
    const int t = threadIdx.x;
    if (t < P.n_scales) {
        ScaleDev S = P.scale[t];
        S.valid = 1;
        const uint32_t hmask = S.I.hash_mask, lcap = S.I.leaf_capacity;
        nm_scale_finish(&S, P.radius[t], P.hash_capacity[t], P.leaf_alloc[t], P.allow_dense != 0);
        S.reserved = 0;
        if (S.I.hash && !S.I.map) {
            // the host sized this index exactly: keep its numbers
            S.I.hash_mask = hmask;
            S.I.leaf_capacity = lcap;
        }
        ladder[P.first + t] = S;
        if (order_dev && P.first + t == P.finest) {
            OrderDev O;
            nm_order_plan(S.L, &O, P.order_points);
            *order_dev = O;
        }
    }
}

int nm_ladder_put(nm_ctx* ctx, const LatticeDev* L, const IndexDev* I, const double* radii, int n_scales,
                  int finest, uint32_t leaf_alloc, ScaleDev* d_ladder, OrderDev* d_order, hipStream_t s)
{
    for (int first = 0; first < n_scales; first += 8) {
        LadderPut P;
        P.allow_dense = leaf_alloc > 0;
        P.n_scales = n_scales - first < 8 ? n_scales - first : 8;
        P.first = first;
        P.finest = finest;
        P.order_points = ctx->order_points;
        for (int t = 0; t < P.n_scales; ++t) {
            // same lattice as an earlier scale (same edge, hence same corner and widths): borrow its index
            int owner = first + t;
            for (int j = first + t - 1; j >= 0; --j)
                if (L[j].edge == L[first + t].edge && L[j].min_x == L[first + t].min_x &&
                    L[j].min_y == L[first + t].min_y && L[j].min_z == L[first + t].min_z &&
                    L[j].wx == L[first + t].wx && L[j].wy == L[first + t].wy && L[j].wz == L[first + t].wz)
                    owner = j;
            P.scale[t].L = L[first + t];
            P.scale[t].I = I[owner];
            P.scale[t].stats = I[first + t].counters;
            P.scale[t].shared = owner != first + t;
            P.scale[t].r2 = 0.0;
            P.scale[t].valid = 1;
            P.scale[t].prune_ok = 0;
            P.radius[t] = radii[first + t];
            P.hash_capacity[t] = I[owner].hash_mask + 1u;
            P.leaf_alloc[t] = leaf_alloc ? leaf_alloc : I[owner].leaf_capacity;
        }
        k_put_ladder<<<1, 64, 0, s>>>(P, d_ladder, d_order);
    }
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}

int nm_ladder_make(nm_ctx* ctx, const double* d_minmax, const void* d_bounds_partial, int bounds_blocks,
                   double* d_minmax_out, const double* edges, const double* radii,
                   int n_scales, int finest, void* const* hash, void* const* leaf, void* const* counters,
                   uint32_t hash_capacity, uint32_t leaf_capacity, ScaleDev* d_ladder, OrderDev* d_order,
                   hipStream_t s)
{
    LadderSpec P;
    P.n_scales = n_scales;
    P.finest = finest;
    for (int i = 0; i < n_scales; ++i) {
        P.edge[i] = edges[i];
        P.radius[i] = radii[i];
        P.hash[i] = (HashEntry*)hash[i];
        P.leaf[i] = (uint32_t*)leaf[i];
        P.counters[i] = (uint32_t*)counters[i];
    }
    P.hash_capacity = hash_capacity;
    P.leaf_capacity = leaf_capacity;
    P.order_points = ctx->order_points;
    k_make_ladder<<<1, LADDER_THREADS, 0, s>>>(d_minmax, (const uint64_t*)d_bounds_partial, bounds_blocks, d_minmax_out, P,
                                   d_ladder, d_order, ctx->d_status);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}

IndexDev nm_index_at(nm_ctx* ctx, void* index_mem, const IndexLayout& lay)
{
    char* w = (char*)index_mem;
    IndexDev I;
    I.hash = (HashEntry*)w;         w += lay.hash_bytes;
    I.leaf = (uint32_t*)w;          w += lay.leaf_bytes;
    I.counters = (uint32_t*)w;
    I.hash_mask = lay.hash_capacity - 1;
    I.leaf_capacity = lay.leaf_capacity;
    I.status = ctx->d_status;
    return I;
}

// the ladder keeps one index per scale and prepares / finishes them together: a launch that clears a few
// megabytes or counts a few thousand leaves costs 5-8 us of which almost nothing is work, and there were
// four of them per scale.

// hash tables to all ones (free), counter blocks to zero
__global__ __launch_bounds__(256) void k_index_clear_all(const ScaleDev* __restrict__ ladder)
{
    // (a column of blocks per scale: the scales are cleared side by side)
    nm_index_clear_part(ladder + blockIdx.y, 1, blockIdx.x * (uint64_t)blockDim.x + threadIdx.x,
                        (uint64_t)gridDim.x * blockDim.x, 0, 1);
}

// M of every index (set bits of its allocated leaves)
__global__ __launch_bounds__(256) void k_count_voxels_all(const ScaleDev* __restrict__ ladder)
{
    __shared__ uint32_t wsum[4];
    if (ladder[blockIdx.y].shared) return;       // counted by the owner, into the counters both read
    const IndexDev I = ladder[blockIdx.y].I;
    const uint32_t n_leaves = min(I.counters[0], I.leaf_capacity);
    const uint64_t words = (uint64_t)n_leaves * NM_LEAF_WORDS / 4;     // as uint4
    uint32_t c = 0;
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < words;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = ((const uint4*)I.leaf)[t];
        c += (uint32_t)(__popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (t) atomicAdd(&I.counters[1], t);
    }
    if (!I.hash) {
        // dense index: every superblock has a leaf; the leaves that hold a voxel are counted for the info
        // block (counters[4]), so that "leaves" means the same thing in both forms
        uint32_t occupied = 0;
        for (uint32_t l = blockIdx.x * blockDim.x + threadIdx.x; l < n_leaves; l += gridDim.x * blockDim.x) {
            const uint4* q = (const uint4*)(I.leaf + (size_t)l * NM_LEAF_WORDS);
            uint32_t any = 0;
#pragma unroll
            for (int i = 0; i < NM_LEAF_WORDS / 4; ++i) any |= q[i].x | q[i].y | q[i].z | q[i].w;
            occupied += any ? 1u : 0u;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) occupied += __shfl_xor(occupied, off);
        if ((threadIdx.x & 63) == 0 && occupied) atomicAdd(&I.counters[4], occupied);
    }
}

int nm_index_clear_all(nm_ctx* ctx, const ScaleDev* d_ladder, int n, hipStream_t s)
{
#ifndef NM_CLEAR_BLOCKS
#define NM_CLEAR_BLOCKS 1024
#endif
    k_index_clear_all<<<dim3(NM_CLEAR_BLOCKS, n), 256, 0, s>>>(d_ladder);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}

int nm_index_count_all(nm_ctx* ctx, const ScaleDev* d_ladder, int n, hipStream_t s)
{
    k_count_voxels_all<<<dim3(128, n), 256, 0, s>>>(d_ladder);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}

// fills the indexes of scales [first, first + count) of a ladder that nm_index_clear_all has prepared
int nm_index_build_ladder(nm_ctx* ctx, const double* sorted_xyz, int64_t n, const ScaleDev* d_ladder,
                          int first, int count, hipStream_t s)
{
    k_index_fused<false><<<(int)((n + FUSED_CHUNK - 1) / FUSED_CHUNK), 256, 0, s>>>(
        sorted_xyz, n, 3, nullptr, nullptr, d_ladder + first, count);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}

// the same from the cloud as the caller holds it plus the spatial order: gathers the coordinates into
// `sorted_xyz` on the way (no gather kernel, one read of the sorted copy less)
int nm_index_build_ladder_gather(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride,
                                 const uint32_t* order, double* sorted_xyz, const ScaleDev* d_ladder, int first,
                                 int count, hipStream_t s)
{
    k_index_fused<true><<<(int)((n + FUSED_CHUNK - 1) / FUSED_CHUNK), 256, 0, s>>>(
        d_xyz, n, stride, order, sorted_xyz, d_ladder + first, count);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}
