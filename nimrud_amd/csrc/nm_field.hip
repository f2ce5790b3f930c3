// nm_field.hip - vector-field multiscale operator.
//
// the legacy pycuda generation of the reference has an operator that averages arbitrary per-point
// attributes over spherical neighborhoods (prototypes/mso.py:12-173, V_MSO: voxelize the search space,
// carry the attribute field over to the voxels, mean over the voxels within each radius).  this is the
// same operator on the lattice of nimrud/minimal: the search cloud is voxel-filtered exactly as for the
// features (geometry.py:103-154), a voxel's attribute is the mean of the attributes of the search points
// that fall into it, and a query point receives the mean of the voxel attributes over the voxel centres
// within `radius` (the inclusive fp64 predicate of multiscale.py:87-103, same centres).
// the legacy code is fp32, strict-less-than, nearest-observation interpolation and depends on its
// partitioning: it cannot be reproduced number for number, and the reference's current path has no
// counterpart, so parity here is pinned by the build's own oracle (oracle.one_scale_field_mean).
//
// it reuses the sort and the occupancy index of the one-scale path; on top of them the sorted keys give
// every voxel a rank (run heads of equal keys), the voxel attributes are summed run by run (in sorted
// order: deterministic), and each row word of a leaf learns the rank of its first voxel, so that the
// query kernel turns an occupied candidate cell into a rank with one popcount.
#include "nm_common.h"
#include "nm_index.h"

#include <rocprim/device/device_scan.hpp>

constexpr int NM_FIELD_MAX_DIMS = 16;

static inline size_t field_align(size_t v) { return (v + 255) / 256 * 256; }

// flag[i] = 1 where sorted key i starts a new voxel
__global__ __launch_bounds__(256) void k_field_heads(const uint64_t* __restrict__ key, int64_t n,
                                                     uint32_t* __restrict__ flag)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || key[i] != key[i - 1]) ? 1u : 0u;
}

// rank[i] = inclusive scan of flag = voxel number + 1.  heads record where their voxel's run starts and,
// where they also start a row word of their leaf, the rank of that row's first voxel
__global__ __launch_bounds__(256) void k_field_starts(const uint64_t* __restrict__ key,
                                                      const uint32_t* __restrict__ rank, int64_t n,
                                                      IndexDev I, uint32_t* __restrict__ start,
                                                      uint32_t* __restrict__ rowbase,
                                                      uint32_t* __restrict__ n_voxels)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = key[i];
    const bool head = i == 0 || k != key[i - 1];
    if (i == n - 1) {
        *n_voxels = rank[i];
        start[rank[i]] = (uint32_t)n;          // end of the last run
    }
    if (!head) return;
    const uint32_t v = rank[i] - 1u;
    start[v] = (uint32_t)i;
    if (i == 0 || (k >> NM_SBX_BITS) != (key[i - 1] >> NM_SBX_BITS)) {
        const int32_t leaf = nm_hash_find(I, k >> NM_LOCAL_BITS);
        if (leaf >= 0)
            rowbase[(size_t)leaf * NM_LEAF_WORDS + (((uint32_t)k >> NM_SBX_BITS) & (NM_LEAF_WORDS - 1))] = v;
    }
}

// voxel attribute = mean of the attributes of its points, summed in sorted order
__global__ __launch_bounds__(256) void k_field_voxel_mean(const uint32_t* __restrict__ start,
                                                          const uint32_t* __restrict__ row_of,
                                                          const uint32_t* __restrict__ n_voxels,
                                                          const double* __restrict__ attr,
                                                          int64_t astride, int32_t dims,
                                                          double* __restrict__ vmean)
{
    const uint32_t m = *n_voxels;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < m; v += gridDim.x * blockDim.x) {
        const uint32_t lo = start[v], hi = start[v + 1];
        double acc[NM_FIELD_MAX_DIMS];
#pragma unroll
        for (int d = 0; d < NM_FIELD_MAX_DIMS; ++d) acc[d] = 0.0;
        for (uint32_t t = lo; t < hi; ++t) {
            const double* a = attr + (int64_t)row_of[t] * astride;
#pragma unroll
            for (int d = 0; d < NM_FIELD_MAX_DIMS; ++d)
                if (d < dims) acc[d] += a[d];
        }
        const double inv = 1.0 / (double)(hi - lo);
#pragma unroll
        for (int d = 0; d < NM_FIELD_MAX_DIMS; ++d)
            if (d < dims) vmean[(size_t)v * dims + d] = acc[d] * inv;
    }
}

// one lane per query: every occupied candidate cell inside the radius contributes its voxel's attribute
__global__ __launch_bounds__(64) void k_field_query(const double* __restrict__ query, int64_t nq,
                                                    int64_t qstride,
                                                    const uint32_t* __restrict__ order,   // nullable
                                                    LatticeDev L, IndexDev I,
                                                    const uint32_t* __restrict__ rowbase,
                                                    const double* __restrict__ vmean, int32_t dims,
                                                    double r2, int32_t dmin, int32_t W,
                                                    double* __restrict__ out, int64_t ostride)
{
    const int64_t slot = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (slot >= nq) return;
    // in the spatial order of the sort when the queries are the search points themselves: the lanes of
    // a wave then walk the same leaves
    const int64_t qi = order ? (int64_t)order[slot] : slot;
    const double* p = query + qi * qstride;
    const double qx = p[0], qy = p[1], qz = p[2];
    const int32_t hx = nm_clamp_cell(nm_cell_f(qx, L.min_x, L.edge));
    const int32_t hy = nm_clamp_cell(nm_cell_f(qy, L.min_y, L.edge));
    const int32_t hz = nm_clamp_cell(nm_cell_f(qz, L.min_z, L.edge));
    double acc[NM_FIELD_MAX_DIMS];
#pragma unroll
    for (int d = 0; d < NM_FIELD_MAX_DIMS; ++d) acc[d] = 0.0;
    double count = 0.0;
    for (int32_t k = 0; k < W; ++k) {
        const int32_t gz = hz + dmin + k;
        if (gz < 0 || gz >= (1 << L.wz)) continue;
        double d = qz - nm_centre(gz, L.min_z, L.edge, L.half_edge);
        const double dz2 = d * d;
        for (int32_t j = 0; j < W; ++j) {
            const int32_t gy = hy + dmin + j;
            if (gy < 0 || gy >= (1 << L.wy)) continue;
            d = qy - nm_centre(gy, L.min_y, L.edge, L.half_edge);
            const double dy2 = d * d;
            int32_t cached_sb = INT32_MIN;
            uint32_t word = 0, base = 0;
            for (int32_t i = 0; i < W; ++i) {
                const int32_t gx = hx + dmin + i;
                if (gx < 0 || gx >= (1 << L.wx)) continue;
                const int32_t sbx = gx >> NM_SBX_BITS;
                if (sbx != cached_sb) {
                    cached_sb = sbx;
                    const int32_t leaf = nm_hash_find(
                        I, nm_sb_key((uint32_t)sbx, (uint32_t)(gy >> NM_SBY_BITS),
                                     (uint32_t)(gz >> NM_SBZ_BITS), L));
                    word = 0u;
                    if (leaf >= 0) {
                        const size_t at = (size_t)leaf * NM_LEAF_WORDS + (gz & 7) * 8 + (gy & 7);
                        word = I.leaf[at];
                        base = rowbase[at];
                    }
                }
                const uint32_t bit = (uint32_t)gx & 31u;
                if (!((word >> bit) & 1u)) continue;
                d = qx - nm_centre(gx, L.min_x, L.edge, L.half_edge);
                const double s = (d * d + dy2) + dz2;
                if (!(s <= r2)) continue;
                const uint32_t v = base + (uint32_t)__popc(word & ((1u << bit) - 1u));
                const double* a = vmean + (size_t)v * dims;
                count += 1.0;
#pragma unroll
                for (int c = 0; c < NM_FIELD_MAX_DIMS; ++c)
                    if (c < dims) acc[c] += a[c];
            }
        }
    }
    double* o = out + qi * ostride;
    const double inv = count > 0.0 ? 1.0 / count : 0.0;
#pragma unroll
    for (int c = 0; c < NM_FIELD_MAX_DIMS; ++c)
        if (c < dims) o[c] = acc[c] * inv;
}

struct FieldLayout {
    size_t key_tmp, val_tmp, key_sorted, val_sorted, sort_temp, sort_temp_bytes;
    size_t index, flag, rank, start, rowbase, vmean, scan_temp, scan_temp_bytes, count, total;
    IndexLayout ilay;
};

static void field_layout(int64_t ns, const LatticeDev& L, int32_t dims, FieldLayout* S)
{
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t at = off;
        off += field_align(bytes);
        return at;
    };
    S->key_tmp = take((size_t)ns * 8);
    S->val_tmp = take((size_t)ns * 4);
    S->key_sorted = take((size_t)ns * 8);
    S->val_sorted = take((size_t)ns * 4);
    S->sort_temp_bytes = nm_sort_pairs_temp_bytes(ns);
    S->sort_temp = take(S->sort_temp_bytes);
    nm_index_layout(L, ns, &S->ilay);
    S->index = take(S->ilay.total);
    S->flag = take((size_t)ns * 4);
    S->rank = take((size_t)ns * 4);
    S->start = take((size_t)(ns + 1) * 4);
    S->rowbase = take((size_t)S->ilay.leaf_capacity * NM_LEAF_WORDS * 4);
    S->vmean = take((size_t)ns * dims * 8);
    size_t scan_bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, scan_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)ns,
                                  rocprim::plus<uint32_t>(), (hipStream_t)0);
    S->scan_temp_bytes = scan_bytes;
    S->scan_temp = take(scan_bytes);
    S->count = take(256);
    S->total = off;
}

extern "C" size_t nm_field_workspace_bytes(int64_t n_query, int64_t n_search, const nm_lattice* lat,
                                           int32_t dims)
{
    (void)n_query;
    if (!lat || n_search < 1 || dims < 1 || dims > NM_FIELD_MAX_DIMS) return 0;
    FieldLayout S;
    field_layout(n_search, make_lattice_dev(lat), dims, &S);
    return S.total;
}

extern "C" int nm_field_mean(nm_ctx* ctx, const double* d_query, int64_t n_query, int64_t query_stride,
                             const double* d_search, int64_t n_search, int64_t search_stride,
                             const double* d_attr, int64_t attr_stride, int32_t dims,
                             const nm_lattice* lat, double radius, double* d_out, int64_t out_stride,
                             void* d_work, size_t work_bytes, void* stream)
{
    NM_ENTER_STREAM(ctx, stream);
    if (!d_search || n_search < 2 || search_stride < 3 || n_query < 0 || !d_work || !d_attr ||
        dims < 1 || dims > NM_FIELD_MAX_DIMS || attr_stride < dims || out_stride < dims ||
        n_search >= ((int64_t)1 << 31) || (n_query > 0 && (!d_query || !d_out || query_stride < 3)))
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_field_mean: bad arguments (1 <= dims <= %d)", NM_FIELD_MAX_DIMS);
    int rc = validate_lattice(ctx, lat);
    if (rc) return rc;
    if (!(radius >= 0.0)) NM_FAIL(ctx, NM_ERR_RADIUS, "radius must be non-negative");
    const LatticeDev L = make_lattice_dev(lat);
    if (L.keybits > 64) NM_FAIL(ctx, NM_ERR_LATTICE, "lattice too large for the device sort key");
    const double m = floor(radius / lat->edge + 0.5 + 1e-9);
    if (!(m >= 0.0) || m > 1000.0)
        NM_FAIL(ctx, NM_ERR_RADIUS, "radius/edge ratio %g is outside the supported range",
                radius / lat->edge);
    const int32_t dmin = -(int32_t)m, W = 2 * (int32_t)m + 1;
    FieldLayout S;
    field_layout(n_search, L, dims, &S);
    if (work_bytes < S.total)
        NM_FAIL(ctx, NM_ERR_WORKSPACE, "nm_field_mean: workspace %zu < required %zu", work_bytes, S.total);
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)d_work;
    uint64_t* key_sorted = (uint64_t*)(w + S.key_sorted);
    uint32_t* row_of = (uint32_t*)(w + S.val_sorted);
    rc = nm_sort_cells(ctx, d_search, n_search, search_stride, L, (uint64_t*)(w + S.key_tmp),
                       (uint32_t*)(w + S.val_tmp), key_sorted, row_of, w + S.sort_temp, S.sort_temp_bytes, s);
    if (rc) return rc;
    IndexDev I;
    rc = nm_index_build(ctx, key_sorted, n_search, S.ilay, w + S.index, &I, s);
    if (rc) return rc;
    uint32_t* flag = (uint32_t*)(w + S.flag);
    uint32_t* rank = (uint32_t*)(w + S.rank);
    uint32_t* start = (uint32_t*)(w + S.start);
    uint32_t* rowbase = (uint32_t*)(w + S.rowbase);
    double* vmean = (double*)(w + S.vmean);
    uint32_t* n_voxels = (uint32_t*)(w + S.count);
    const int blocks = (int)((n_search + 255) / 256);
    k_field_heads<<<blocks, 256, 0, s>>>(key_sorted, n_search, flag);
    size_t scan_bytes = S.scan_temp_bytes;
    NM_HIP(ctx, rocprim::inclusive_scan(w + S.scan_temp, scan_bytes, flag, rank, (size_t)n_search,
                                        rocprim::plus<uint32_t>(), s));
    k_field_starts<<<blocks, 256, 0, s>>>(key_sorted, rank, n_search, I, start, rowbase, n_voxels);
    k_field_voxel_mean<<<blocks < 4096 ? blocks : 4096, 256, 0, s>>>(start, row_of, n_voxels, d_attr,
                                                                     attr_stride, dims, vmean);
    if (n_query > 0)
        k_field_query<<<(int)((n_query + 63) / 64), 64, 0, s>>>(d_query, n_query, query_stride,
                                                               (d_query == d_search && n_query == n_search &&
                                                                query_stride == search_stride)
                                                                   ? row_of : nullptr,
                                                               L, I,
                                                               rowbase, vmean, dims, radius * radius,
                                                               dmin, W, d_out, out_stride);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}
