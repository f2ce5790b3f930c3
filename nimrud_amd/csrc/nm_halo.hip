// nm_halo.hip - multi-GPU tiling: halo selection and the halo exchange over RCCL.
//
// the reference has no distributed code; its legacy partitioners pair every query tile with a search
// tile grown by the largest scale (prototypes/mso.py:892-927, utils/geometry.py:203-253
// nested_regions).  here a rank owns one spatial tile of the cloud - a Morton-contiguous run of it, or
// any other subset - and, before the scale loop, receives from every other rank the points that can
// matter to its queries: those within margin = max_s(r_s + sqrt(3)/2 e_s) of one of its own points.
//
// two ways to decide "can matter":
//   boxes      inside the destination tile's bounding box grown by the margin.  exact enough for convex
//              tiles (median bisection), far too generous for Morton runs, which are L-shaped.
//   cell sets  inside the destination tile's set of occupied COARSE cells dilated by the margin.  the
//              coarse grid is the same on every rank (a pure function of the global extrema and the
//              margin), one bit per cell, at most 2^21 cells: 256 KB per rank, all-gathered once per
//              exchange.  a tile of any shape then receives about its true surface layer.
//
// nm_halo_exchange runs the whole exchange through RCCL behind the C ABI: all-gather of the tile boxes
// (their extrema are the GLOBAL extrema every rank must build its lattices from, geometry.py:37),
// all-gather of the cell sets, all-gather of the pair counts, ONE host synchronisation to learn the
// sizes, then grouped ncclSend / ncclRecv (an all-to-all-v: each pair of ranks talks over its own xGMI
// link; nothing is reduced, so ring bandwidth never enters).

#include "nm_common.h"
#include "nm_index.h"

#include <string.h>

#include <rccl/rccl.h>

constexpr int NM_MAX_BOXES = 64;
constexpr int NM_CELLSET_LOG2 = 21;                                  // at most 2^21 coarse cells
constexpr int NM_CELLSET_WORDS = 1 << (NM_CELLSET_LOG2 - 5);         // 65 536 words = 256 KB per rank
static_assert(NM_CELLSET_WORDS == NM_HALO_CELLSET_WORDS, "header and kernels disagree");

// ---- the coarse grid ---------------------------------------------------------------------------------
// cubic cells of edge e_c over the global bounding box; e_c = margin / 4 unless that needs more than
// 2^21 cells, then as fine as fits.  a point within `margin` (per axis) of a point of a tile lies at most
// D = floor(margin / e_c) + 1 cells from that point's cell on every axis.
struct CoarseGrid {
    double lo[3];
    double inv_e;
    int32_t dim[3];
    int32_t D;
};

__host__ __device__ inline CoarseGrid nm_coarse_grid(const double* g, double margin)
{
    CoarseGrid G;
    double ext[3];
    for (int a = 0; a < 3; ++a) {
        G.lo[a] = g[a];
        ext[a] = g[3 + a] - g[a];
        if (!(ext[a] > 0.0)) ext[a] = 0.0;
    }
    double e = margin > 0.0 ? margin * 0.25 : 1.0;
    for (int it = 0; it < 400; ++it) {
        double cells = 1.0;
        for (int a = 0; a < 3; ++a) cells *= floor(ext[a] / e) + 1.0;
        if (cells <= (double)(1 << NM_CELLSET_LOG2)) break;
        e *= 1.25;
    }
    G.inv_e = 1.0 / e;
    for (int a = 0; a < 3; ++a) G.dim[a] = (int32_t)(floor(ext[a] / e) + 1.0);
    G.D = (int32_t)floor(margin * G.inv_e * (1.0 + 1e-12)) + 1;
    return G;
}

__device__ __forceinline__ int32_t nm_coarse_cell(const CoarseGrid& G, double x, double y, double z)
{
    int32_t c[3];
    const double p[3] = {x, y, z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double f = floor((p[a] - G.lo[a]) * G.inv_e);
        f = fmin(fmax(f, 0.0), (double)(G.dim[a] - 1));
        c[a] = (int32_t)f;
    }
    return (c[2] * G.dim[1] + c[1]) * G.dim[0] + c[0];
}

// where a point of this rank has to go: the destinations' boxes, or their cell sets
struct DestSet {
    const double* boxes;       // boxes mode: n_dest x 6 (lo xyz, hi xyz, already grown by the margin)
    const uint32_t* cellsets;  // cell mode: n_dest x NM_CELLSET_WORDS bits, already dilated
    const double* global;      // cell mode: the 6 global extrema (device)
    double margin;             // cell mode
    int32_t n_dest;
    int32_t skip;              // this rank (never selected)
};

__device__ __forceinline__ bool nm_in_box(const double* __restrict__ b, double x, double y, double z)
{
    return x >= b[0] && y >= b[1] && z >= b[2] && x <= b[3] && y <= b[4] && z <= b[5];
}

// ---- count / pack -------------------------------------------------------------------------------------
// one kernel body for both: PACK = false counts per destination, PACK = true writes the rows.
template <bool PACK>
__global__ __launch_bounds__(256) void k_halo_select(const double* __restrict__ xyz, int64_t n,
                                                     int64_t stride, DestSet B,
                                                     unsigned long long* __restrict__ counts,
                                                     const int64_t* __restrict__ offsets,
                                                     unsigned long long* __restrict__ cursor,
                                                     double* __restrict__ out)
{
    __shared__ double sbox[NM_MAX_BOXES * 6];
    __shared__ uint32_t scount[NM_MAX_BOXES];
    const bool cells = B.cellsets != nullptr;
    CoarseGrid G;
    if (cells) G = nm_coarse_grid(B.global, B.margin);
    else
        for (int t = threadIdx.x; t < B.n_dest * 6; t += blockDim.x) sbox[t] = B.boxes[t];
    for (int t = threadIdx.x; t < B.n_dest; t += blockDim.x) scount[t] = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // block-uniform trip count so that every __ballot sees whole waves
    for (int64_t base = blockIdx.x * (int64_t)blockDim.x; base < n;
         base += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = base + threadIdx.x;
        const bool valid = i < n;
        double x = 0, y = 0, z = 0;
        if (valid) {
            const double* p = xyz + i * stride;
            x = p[0];
            y = p[1];
            z = p[2];
        }
        int32_t cell = 0;
        if (cells && valid) cell = nm_coarse_cell(G, x, y, z);
        for (int b = 0; b < B.n_dest; ++b) {
            bool in = valid && b != B.skip;
            if (in) {
                if (cells)
                    in = (B.cellsets[(size_t)b * NM_CELLSET_WORDS + (cell >> 5)] >> (cell & 31)) & 1u;
                else
                    in = nm_in_box(sbox + b * 6, x, y, z);
            }
            const unsigned long long m = __ballot(in);
            if (!m) continue;
            if (!PACK) {
                if (lane == 0) atomicAdd(&scount[b], (uint32_t)__popcll(m));
            } else {
                // one cursor bump per wave and destination; order inside a segment is immaterial (the
                // search side only looks at which cells are occupied)
                unsigned long long at = 0;
                const int leader = __ffsll((long long)m) - 1;
                if (lane == leader) at = atomicAdd(&cursor[b], (unsigned long long)__popcll(m));
                at = __shfl(at, leader);
                if (in) {
                    const int64_t row = offsets[b] + (int64_t)at +
                                        (int64_t)__popcll(m & ((1ull << lane) - 1ull));
                    out[row * 3 + 0] = x;
                    out[row * 3 + 1] = y;
                    out[row * 3 + 2] = z;
                }
            }
        }
    }
    if (!PACK) {
        __syncthreads();
        for (int t = threadIdx.x; t < B.n_dest; t += blockDim.x)
            if (scount[t]) atomicAdd(&counts[t], (unsigned long long)scount[t]);
    }
}

static int check_select(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride, int32_t n_dest)
{
    if (n < 0 || stride < 3 || (n > 0 && !d_xyz) || n_dest < 1 || n_dest > NM_MAX_BOXES)
        NM_FAIL(ctx, NM_ERR_INVALID, "halo: bad arguments (at most %d destinations)", NM_MAX_BOXES);
    return NM_OK;
}

static int select_blocks(int64_t n)
{
    int64_t blocks = (n + 255) / 256;
    return (int)(blocks > 2048 ? 2048 : blocks);
}

static int halo_count(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride, const DestSet& B,
                      int64_t* d_counts, hipStream_t s)
{
    NM_HIP(ctx, hipMemsetAsync(d_counts, 0, sizeof(int64_t) * B.n_dest, s));
    if (n == 0) return NM_OK;
    k_halo_select<false><<<select_blocks(n), 256, 0, s>>>(d_xyz, n, stride, B,
                                                         (unsigned long long*)d_counts, nullptr,
                                                         nullptr, nullptr);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}

static int halo_pack(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride, const DestSet& B,
                     const int64_t* d_offsets, int64_t* d_cursor, double* d_out, hipStream_t s)
{
    NM_HIP(ctx, hipMemsetAsync(d_cursor, 0, sizeof(int64_t) * B.n_dest, s));
    if (n == 0) return NM_OK;
    k_halo_select<true><<<select_blocks(n), 256, 0, s>>>(d_xyz, n, stride, B, nullptr, d_offsets,
                                                        (unsigned long long*)d_cursor, d_out);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}

extern "C" int nm_halo_count(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride,
                             const double* d_boxes, int32_t n_boxes, int32_t skip, int64_t* d_counts,
                             void* stream)
{
    NM_ENTER_STREAM(ctx, stream);
    int rc = check_select(ctx, d_xyz, n, stride, n_boxes);
    if (rc) return rc;
    if (!d_boxes || !d_counts) NM_FAIL(ctx, NM_ERR_INVALID, "nm_halo_count: null arguments");
    const DestSet B{d_boxes, nullptr, nullptr, 0.0, n_boxes, skip};
    return halo_count(ctx, d_xyz, n, stride, B, d_counts, (hipStream_t)stream);
}

extern "C" int nm_halo_pack(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride,
                            const double* d_boxes, int32_t n_boxes, int32_t skip,
                            const int64_t* d_offsets, int64_t* d_cursor, double* d_out, void* stream)
{
    NM_ENTER_STREAM(ctx, stream);
    int rc = check_select(ctx, d_xyz, n, stride, n_boxes);
    if (rc) return rc;
    if (!d_boxes || !d_offsets || !d_cursor || !d_out)
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_halo_pack: null arguments");
    const DestSet B{d_boxes, nullptr, nullptr, 0.0, n_boxes, skip};
    return halo_pack(ctx, d_xyz, n, stride, B, d_offsets, d_cursor, d_out, (hipStream_t)stream);
}

extern "C" int nm_halo_count_cells(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride,
                                   const double* d_global_minmax, double margin,
                                   const uint32_t* d_cellsets, int32_t n_ranks, int32_t skip,
                                   int64_t* d_counts, void* stream)
{
    NM_ENTER_STREAM(ctx, stream);
    int rc = check_select(ctx, d_xyz, n, stride, n_ranks);
    if (rc) return rc;
    if (!d_global_minmax || !d_cellsets || !d_counts || !(margin > 0.0))
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_halo_count_cells: bad arguments");
    const DestSet B{nullptr, d_cellsets, d_global_minmax, margin, n_ranks, skip};
    return halo_count(ctx, d_xyz, n, stride, B, d_counts, (hipStream_t)stream);
}

extern "C" int nm_halo_pack_cells(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride,
                                  const double* d_global_minmax, double margin,
                                  const uint32_t* d_cellsets, int32_t n_ranks, int32_t skip,
                                  const int64_t* d_offsets, int64_t* d_cursor, double* d_out,
                                  void* stream)
{
    NM_ENTER_STREAM(ctx, stream);
    int rc = check_select(ctx, d_xyz, n, stride, n_ranks);
    if (rc) return rc;
    if (!d_global_minmax || !d_cellsets || !d_offsets || !d_cursor || !d_out || !(margin > 0.0))
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_halo_pack_cells: bad arguments");
    const DestSet B{nullptr, d_cellsets, d_global_minmax, margin, n_ranks, skip};
    return halo_pack(ctx, d_xyz, n, stride, B, d_offsets, d_cursor, d_out, (hipStream_t)stream);
}

// ---- the cell set of a tile ----------------------------------------------------------------------------
// byte map of the coarse cells the tile's points occupy -> dilated by D cells along x, y, z (separable:
// the result is the box dilation, a superset of the ball) -> packed to bits.

__global__ __launch_bounds__(256) void k_cells_mark(const double* __restrict__ xyz, int64_t n,
                                                    int64_t stride, const double* __restrict__ global,
                                                    double margin, uint8_t* __restrict__ map)
{
    const CoarseGrid G = nm_coarse_grid(global, margin);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const double* p = xyz + i * stride;
        map[nm_coarse_cell(G, p[0], p[1], p[2])] = 1;       // racing stores of the same value
    }
}

template <int AXIS>
__global__ __launch_bounds__(256) void k_cells_dilate(const double* __restrict__ global, double margin,
                                                      const uint8_t* __restrict__ in,
                                                      uint8_t* __restrict__ out)
{
    const CoarseGrid G = nm_coarse_grid(global, margin);
    const int32_t total = G.dim[0] * G.dim[1] * G.dim[2];
    const int32_t step = AXIS == 0 ? 1 : (AXIS == 1 ? G.dim[0] : G.dim[0] * G.dim[1]);
    for (int32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < total; c += gridDim.x * blockDim.x) {
        const int32_t x = c % G.dim[0], y = (c / G.dim[0]) % G.dim[1], z = c / (G.dim[0] * G.dim[1]);
        const int32_t pos = AXIS == 0 ? x : (AXIS == 1 ? y : z);
        const int32_t lo = max(pos - G.D, 0), hi = min(pos + G.D, G.dim[AXIS] - 1);
        uint8_t v = 0;
        for (int32_t q = lo; q <= hi && !v; ++q) v = in[c + (q - pos) * step];
        out[c] = v;
    }
}

__global__ __launch_bounds__(256) void k_cells_pack(const double* __restrict__ global, double margin,
                                                    const uint8_t* __restrict__ map,
                                                    uint32_t* __restrict__ bits)
{
    const CoarseGrid G = nm_coarse_grid(global, margin);
    const int32_t total = G.dim[0] * G.dim[1] * G.dim[2];
    // every word of the set is written (cells beyond the grid are clear): the set is sent whole
    const int32_t c = blockIdx.x * blockDim.x + threadIdx.x;          // grid covers 2^21 cells exactly
    const bool on = c < total && map[c] != 0;
    const unsigned long long m = __ballot(on);
    const int lane = threadIdx.x & 63;
    if (lane == 0) bits[c >> 5] = (uint32_t)m;
    if (lane == 32) bits[c >> 5] = (uint32_t)(m >> 32);
}

extern "C" size_t nm_halo_cellset_workspace_bytes(void) { return (size_t)2 << NM_CELLSET_LOG2; }

extern "C" int nm_halo_cellset(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride,
                               const double* d_global_minmax, double margin, uint32_t* d_cellset,
                               void* d_work, size_t work_bytes, void* stream)
{
    NM_ENTER_STREAM(ctx, stream);
    if (n < 0 || stride < 3 || (n > 0 && !d_xyz) || !d_global_minmax || !d_cellset || !d_work ||
        !(margin > 0.0))
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_halo_cellset: bad arguments");
    if (work_bytes < nm_halo_cellset_workspace_bytes())
        NM_FAIL(ctx, NM_ERR_WORKSPACE, "nm_halo_cellset: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    uint8_t* a = (uint8_t*)d_work;
    uint8_t* b = a + ((size_t)1 << NM_CELLSET_LOG2);
    NM_HIP(ctx, hipMemsetAsync(a, 0, (size_t)1 << NM_CELLSET_LOG2, s));
    if (n > 0) k_cells_mark<<<select_blocks(n), 256, 0, s>>>(d_xyz, n, stride, d_global_minmax, margin, a);
    k_cells_dilate<0><<<1024, 256, 0, s>>>(d_global_minmax, margin, a, b);
    k_cells_dilate<1><<<1024, 256, 0, s>>>(d_global_minmax, margin, b, a);
    k_cells_dilate<2><<<1024, 256, 0, s>>>(d_global_minmax, margin, a, b);
    k_cells_pack<<<(1 << NM_CELLSET_LOG2) / 256, 256, 0, s>>>(d_global_minmax, margin, b, d_cellset);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}

// (n,3) contiguous copy of the geometry columns of a strided cloud (the own-tile part of the search
// buffer that the halo rows are appended to)
__global__ __launch_bounds__(256) void k_copy_xyz(const double* __restrict__ xyz, int64_t n,
                                                  int64_t stride, double* __restrict__ out)
{
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= n * 3) return;
    const int64_t i = t / 3;
    out[t] = xyz[i * stride + (t - i * 3)];
}

extern "C" int nm_copy_xyz(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride, double* d_out,
                           void* stream)
{
    NM_ENTER_STREAM(ctx, stream);
    if (n < 0 || stride < 3 || (n > 0 && (!d_xyz || !d_out)))
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_copy_xyz: bad arguments");
    if (n == 0) return NM_OK;
    k_copy_xyz<<<(int)((n * 3 + 255) / 256), 256, 0, (hipStream_t)stream>>>(d_xyz, n, stride, d_out);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}

// ---------------------------------------------------------------------------------------------------
// RCCL
// ---------------------------------------------------------------------------------------------------

#define NM_NCCL(ctx, call)                                                                \
    do {                                                                                  \
        ncclResult_t _r = (call);                                                         \
        if (_r != ncclSuccess)                                                            \
            NM_FAIL(ctx, NM_ERR_COMM, "%s failed: %s", #call, ncclGetErrorString(_r));    \
    } while (0)

static_assert(NM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

extern "C" int nm_comm_unique_id(void* id_out)
{
    if (!id_out) return NM_ERR_INVALID;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return NM_ERR_COMM;
    memcpy(id_out, &id, sizeof(id));
    return NM_OK;
}

extern "C" int nm_comm_create(nm_ctx* ctx, int32_t n_ranks, int32_t rank, const void* id,
                              void** comm_out)
{
    NM_ENTER(ctx);
    if (!id || !comm_out || n_ranks < 1 || n_ranks > NM_MAX_BOXES || rank < 0 || rank >= n_ranks)
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_comm_create: bad arguments (1 <= ranks <= %d)", NM_MAX_BOXES);
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    NM_NCCL(ctx, ncclCommInitRank(&comm, n_ranks, uid, rank));
    *comm_out = (void*)comm;
    return NM_OK;
}

extern "C" int nm_comm_destroy(nm_ctx* ctx, void* comm)
{
    if (!ctx) return NM_ERR_INVALID;
    nm_device_guard guard(ctx->device);
    if (comm) NM_NCCL(ctx, ncclCommDestroy((ncclComm_t)comm));
    return NM_OK;
}

// boxes of all ranks -> global extrema (also left in the caller's buffer), boxes grown by the margin
__global__ void k_halo_boxes(double* __restrict__ boxes, int32_t n_ranks, double margin,
                             double* __restrict__ global, double* __restrict__ global_out)
{
    const int a = threadIdx.x;
    if (a < 3) {
        double lo = INFINITY, hi = -INFINITY;
        for (int r = 0; r < n_ranks; ++r) {
            lo = fmin(lo, boxes[r * 6 + a]);
            hi = fmax(hi, boxes[r * 6 + 3 + a]);
        }
        global[a] = lo;
        global[3 + a] = hi;
        if (global_out) {
            global_out[a] = lo;
            global_out[3 + a] = hi;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < n_ranks * 6; t += blockDim.x)
        boxes[t] += (t % 6) < 3 ? -margin : margin;
}

struct HaloLayout {
    size_t local, boxes, global, counts, matrix, offsets, cursor, cellset_work, own_cells, cellsets,
        send, total_fixed;
};

static inline size_t halo_align(size_t v) { return (v + 255) / 256 * 256; }

static void halo_layout(int32_t n_ranks, HaloLayout* H)
{
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t at = off;
        off += halo_align(bytes);
        return at;
    };
    H->local = take(6 * 8);
    H->boxes = take((size_t)n_ranks * 6 * 8);
    H->global = take(6 * 8);
    H->counts = take((size_t)(n_ranks + 3) * 8);
    H->matrix = take((size_t)n_ranks * (n_ranks + 3) * 8);
    H->offsets = take((size_t)n_ranks * 8);
    H->cursor = take((size_t)n_ranks * 8);
    H->cellset_work = take(nm_halo_cellset_workspace_bytes());
    H->own_cells = take((size_t)NM_CELLSET_WORDS * 4);
    H->cellsets = take((size_t)n_ranks * NM_CELLSET_WORDS * 4);
    H->send = off;
    H->total_fixed = off;
}

extern "C" size_t nm_halo_workspace_bytes(int64_t send_capacity_rows, int32_t n_ranks)
{
    if (n_ranks < 1 || n_ranks > NM_MAX_BOXES || send_capacity_rows < 0) return 0;
    HaloLayout H;
    halo_layout(n_ranks, &H);
    return H.total_fixed + halo_align((size_t)send_capacity_rows * 24);
}

struct OffsetList {
    int64_t v[NM_MAX_BOXES];
};
__global__ void k_put_offsets(OffsetList L, int32_t n, int64_t* __restrict__ out)
{
    if ((int)threadIdx.x < n) out[threadIdx.x] = L.v[threadIdx.x];
}
// what every rank tells every other: its pair counts, how much it can send and receive, and whether it is well
__global__ void k_halo_announce(int64_t* __restrict__ counts, int32_t n_ranks, int64_t send_capacity,
                                int64_t recv_capacity, int64_t status)
{
    if (threadIdx.x == 0) {
        counts[n_ranks] = send_capacity;
        counts[n_ranks + 1] = recv_capacity;
        counts[n_ranks + 2] = status;
    }
}

// the contribution of a rank that has nothing to contribute (an empty tile, or a rank that must fail): an empty
// box, an empty cell set, no rows for anybody
__global__ void k_halo_nothing(double* __restrict__ local, uint32_t* __restrict__ own_cells, int32_t words,
                               int64_t* __restrict__ counts, int32_t n_ranks)
{
    for (int t = threadIdx.x; t < words; t += blockDim.x) own_cells[t] = 0u;
    if (threadIdx.x < 6) local[threadIdx.x] = threadIdx.x < 3 ? INFINITY : -INFINITY;
    if ((int)threadIdx.x < n_ranks) counts[threadIdx.x] = 0;
}

// the host side of step 3, on a count matrix every rank holds identically (row r: rows r sends to each rank,
// then r's send capacity, receive capacity and status word).  fills this rank's offsets and totals and returns
// the verdict - NM_OK, NM_ERR_WORKSPACE with the first rank that is short of room, or the first unwell rank's
// status - which is therefore the SAME on every rank: either all go on to the exchange or none does.
// (a function of its own so that the CPU tests can drive it with synthetic matrices: tests/test_abi_and_host.py)
extern "C" int nm_halo_plan_from_matrix(const int64_t* matrix, int32_t n_ranks, int32_t rank,
                                        int64_t* send_off, int64_t* recv_off, int64_t* sent_rows,
                                        int64_t* recv_rows, int32_t* culprit)
{
    if (!matrix || n_ranks < 1 || rank < 0 || rank >= n_ranks || !send_off || !recv_off || !sent_rows ||
        !recv_rows)
        return NM_ERR_INVALID;
    const int row = n_ranks + 3;
    auto pair = [&](int from, int to) { return matrix[(size_t)from * row + to]; };
    int64_t sent = 0, received = 0;
    for (int j = 0; j < n_ranks; ++j) {
        send_off[j] = sent;
        recv_off[j] = received;
        sent += pair(rank, j);
        received += pair(j, rank);
    }
    *sent_rows = sent;
    *recv_rows = received;
    if (culprit) *culprit = -1;
    for (int r = 0; r < n_ranks; ++r) {
        const int64_t st = matrix[(size_t)r * row + n_ranks + 2];
        if (st != 0) {
            if (culprit) *culprit = r;
            return (int)st;
        }
    }
    for (int r = 0; r < n_ranks; ++r) {
        int64_t out_r = 0, in_r = 0;
        for (int j = 0; j < n_ranks; ++j) {
            out_r += pair(r, j);
            in_r += pair(j, r);
        }
        if (out_r > matrix[(size_t)r * row + n_ranks] || in_r > matrix[(size_t)r * row + n_ranks + 1]) {
            if (culprit) *culprit = r;
            return NM_ERR_WORKSPACE;
        }
    }
    return NM_OK;
}

extern "C" int nm_halo_exchange(nm_ctx* ctx, void* nccl_comm, int32_t n_ranks, int32_t rank,
                                const double* d_xyz, int64_t n, int64_t stride, double margin,
                                int32_t mode, double* d_recv, int64_t recv_capacity_rows,
                                int64_t* h_recv_rows, int64_t* h_sent_rows, double* d_global_minmax,
                                void* d_work, size_t work_bytes, void* stream)
{
    // this call holds collectives: whatever is wrong with THIS rank alone - a sticky failure of an earlier call,
    // bad cloud arguments - must not make it leave while the others wait in an all-gather.  such a rank goes
    // through the collectives with an empty contribution and a status word that makes every rank return that
    // status after the one host synchronisation.  only what makes the collectives themselves impossible (no
    // communicator, no workspace for the fixed part, a wrong rank count) returns at once.
    if (!ctx) return NM_ERR_INVALID;
    nm_device_guard _nm_guard(ctx->device);
    const bool include_self = (mode & NM_HALO_INCLUDE_SELF) != 0;
    const bool reuse = (mode & NM_HALO_REUSE_PLAN) != 0;
    const int32_t mode_key = mode & ~NM_HALO_REUSE_PLAN;
    mode &= ~(NM_HALO_INCLUDE_SELF | NM_HALO_REUSE_PLAN);
    if (!nccl_comm || n_ranks < 1 || n_ranks > NM_MAX_BOXES || rank < 0 || rank >= n_ranks || !h_recv_rows ||
        !h_sent_rows || !d_work)
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_halo_exchange: bad arguments");
    HaloLayout H;
    halo_layout(n_ranks, &H);
    if (work_bytes < H.total_fixed)
        NM_FAIL(ctx, NM_ERR_WORKSPACE, "nm_halo_exchange: workspace %zu < required %zu", work_bytes,
                H.total_fixed);
    int count = 0;
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    NM_NCCL(ctx, ncclCommCount(comm, &count));
    if (count != n_ranks)
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_halo_exchange: communicator has %d ranks, not %d", count,
                n_ranks);
    int local_status = nm_capturing((hipStream_t)stream) ? NM_OK : nm_status_poll(ctx, false);
    if (local_status == NM_OK &&
        (n < 0 || (n > 0 && (!d_xyz || stride < 3)) || !(margin > 0.0) ||
         (mode != NM_HALO_BOXES && mode != NM_HALO_CELLS) || recv_capacity_rows < 0 ||
         (recv_capacity_rows > 0 && !d_recv))) {
        ctx->error = "nm_halo_exchange: bad arguments";
        local_status = NM_ERR_INVALID;
    }
    bool contribute = local_status == NM_OK && n > 0;       // an empty tile is a legitimate input
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)d_work;
    double* local = (double*)(w + H.local);
    double* boxes = (double*)(w + H.boxes);
    double* global = (double*)(w + H.global);
    int64_t* counts = (int64_t*)(w + H.counts);
    int64_t* matrix = (int64_t*)(w + H.matrix);
    int64_t* offsets = (int64_t*)(w + H.offsets);
    int64_t* cursor = (int64_t*)(w + H.cursor);
    uint32_t* own_cells = (uint32_t*)(w + H.own_cells);
    uint32_t* cellsets = (uint32_t*)(w + H.cellsets);
    double* send = (double*)(w + H.send);
    const int64_t send_capacity = (int64_t)((work_bytes - H.total_fixed) / 24);
    const int32_t skip = include_self ? -1 : rank;
    const int row = n_ranks + 3;        // entries every rank announces

    int rc = NM_OK;
    OffsetList send_off, recv_off;
    std::vector<int64_t>& host = ctx->halo_matrix;
    auto pair = [&](int from, int to) { return host[(size_t)from * row + to]; };
    // ---- a step on unchanged tiles: the plan of the last full call stands (every rank passes the flag or none
    //      does: it is part of the collective's contract).  boxes / cell sets, the global extrema and the pair
    //      counts are still in the workspace and on the host: pack, exchange - no collective but the
    //      point-to-point group, no host synchronisation
    // (a rank that is unwell still takes its part in the planned exchange - there is no all-gather to carry its
    // status - and returns the status afterwards)
    const bool planned = reuse && ctx->halo_ranks == n_ranks && ctx->halo_rank == rank &&
                         ctx->halo_mode == mode_key && ctx->halo_n == n && ctx->halo_work == d_work &&
                         host.size() == (size_t)n_ranks * row &&
                         host[(size_t)rank * row + n_ranks] <= send_capacity &&
                         host[(size_t)rank * row + n_ranks + 1] <= recv_capacity_rows;
    if (reuse && !planned)
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_halo_exchange: NM_HALO_REUSE_PLAN without a matching plan (same "
                "communicator size, rank, mode, tile size, workspace and capacities as the last full call)");
    if (planned) contribute = n > 0 && d_xyz && stride >= 3;
    if (!planned) {
    // 1. every tile's box; their extrema are the global extrema (geometry.py:37 needs the GLOBAL minimum)
    if (contribute) {
        rc = nm_bounds_scratch(ctx, d_xyz, n, stride, local, nullptr, s);
        if (rc) return rc;
    } else {
        k_halo_nothing<<<1, 256, 0, s>>>(local, own_cells, NM_CELLSET_WORDS, counts, n_ranks);
    }
    NM_NCCL(ctx, ncclAllGather(local, boxes, 6, ncclDouble, comm, s));
    k_halo_boxes<<<1, 64, 0, s>>>(boxes, n_ranks, margin, global, d_global_minmax);
    // 2. who needs which of my points
    DestSet B0{boxes, nullptr, nullptr, 0.0, n_ranks, skip};
    if (mode == NM_HALO_CELLS) {
        if (contribute) {
            rc = nm_halo_cellset(ctx, d_xyz, n, stride, global, margin, own_cells, w + H.cellset_work,
                                 nm_halo_cellset_workspace_bytes(), stream);
            if (rc) return rc;
        }
        NM_NCCL(ctx, ncclAllGather(own_cells, cellsets, NM_CELLSET_WORDS, ncclUint32, comm, s));
        B0 = DestSet{nullptr, cellsets, global, margin, n_ranks, skip};
    }
    if (contribute) {
        rc = halo_count(ctx, d_xyz, n, stride, B0, counts, s);
        if (rc) return rc;
    }
    k_halo_announce<<<1, 64, 0, s>>>(counts, n_ranks, send_capacity, recv_capacity_rows, (int64_t)local_status);
    // 3. everybody learns every pair count, every capacity and every status; the one host synchronisation
    NM_NCCL(ctx, ncclAllGather(counts, matrix, (size_t)row, ncclInt64, comm, s));
    host.assign((size_t)n_ranks * row, 0);
    ctx->halo_ranks = 0;            // no plan until this call has one
    NM_HIP(ctx, hipMemcpyAsync(host.data(), matrix, host.size() * 8, hipMemcpyDeviceToHost, s));
    NM_HIP(ctx, hipStreamSynchronize(s));
    ctx->halo_host_syncs += 1;
    }   // !planned
    int32_t culprit = -1;
    // the matrix is the same on every rank, so either every rank returns here or none does: nobody is
    // left waiting in the exchange for a rank that has given up
    const int verdict = nm_halo_plan_from_matrix(host.data(), n_ranks, rank, send_off.v, recv_off.v, h_sent_rows,
                                                 h_recv_rows, &culprit);
    if (verdict == NM_ERR_WORKSPACE)
        NM_FAIL(ctx, NM_ERR_WORKSPACE,
                "nm_halo_exchange: rank %d is short of room (it can send %lld and receive %lld rows); this rank "
                "sends %lld and receives %lld", culprit, (long long)host[(size_t)culprit * row + n_ranks],
                (long long)host[(size_t)culprit * row + n_ranks + 1], (long long)*h_sent_rows,
                (long long)*h_recv_rows);
    if (verdict != NM_OK) {
        if (culprit != rank)
            NM_FAIL(ctx, verdict, "nm_halo_exchange: rank %d reported status %d; nothing was exchanged", culprit,
                    verdict);
        return verdict;       // this rank's own failure: ctx->error already says what
    }
    if (!planned) {
        ctx->halo_ranks = n_ranks; ctx->halo_rank = rank; ctx->halo_mode = mode_key; ctx->halo_n = n;
        ctx->halo_work = d_work;
    } else if (d_global_minmax) {
        NM_HIP(ctx, hipMemcpyAsync(d_global_minmax, global, 6 * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    DestSet B{boxes, nullptr, nullptr, 0.0, n_ranks, skip};
    if (mode == NM_HALO_CELLS) B = DestSet{nullptr, cellsets, global, margin, n_ranks, skip};
    ctx->halo_exchanges += 1;
    // 4. pack and exchange: grouped point-to-point, one send and one receive per neighbour
    if (contribute) {
        k_put_offsets<<<1, 64, 0, s>>>(send_off, n_ranks, offsets);
        rc = halo_pack(ctx, d_xyz, n, stride, B, offsets, cursor, send, s);
        if (rc) return rc;
    }
    NM_NCCL(ctx, ncclGroupStart());
    for (int j = 0; j < n_ranks; ++j) {
        const int64_t to = pair(rank, j), from = pair(j, rank);
        if (to > 0)
            NM_NCCL(ctx, ncclSend(send + send_off.v[j] * 3, (size_t)to * 3, ncclDouble, j, comm, s));
        if (from > 0)
            NM_NCCL(ctx, ncclRecv(d_recv + recv_off.v[j] * 3, (size_t)from * 3, ncclDouble, j, comm, s));
    }
    NM_NCCL(ctx, ncclGroupEnd());
    return planned ? local_status : NM_OK;
}

// how often nm_halo_exchange has synchronised the host and how often it has exchanged, since the context was made
extern "C" int nm_halo_stats(nm_ctx* ctx, int64_t* host_syncs, int64_t* exchanges)
{
    if (!ctx) return NM_ERR_INVALID;
    if (host_syncs) *host_syncs = ctx->halo_host_syncs;
    if (exchanges) *exchanges = ctx->halo_exchanges;
    return NM_OK;
}
