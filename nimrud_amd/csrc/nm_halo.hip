// nm_halo.hip - halo selection for the multi-GPU tiling.
//
// the reference has no distributed code; its legacy partitioners pair every query tile with a search
// tile grown by the largest scale (prototypes/mso.py:892-927, utils/geometry.py:203-253
// nested_regions).  here a rank owns one spatial tile of the cloud and, before the scale loop, sends
// every other rank the points of its tile that fall inside that rank's bounding box grown by
// (largest radius + largest voxel diagonal); the exchange itself is RCCL's all-to-all
// (torch.distributed) on the packed rows these kernels produce.

#include "nm_common.h"

constexpr int NM_MAX_BOXES = 64;

struct BoxSet {
    const double* boxes;   // n_boxes x 6: lo xyz, hi xyz (already grown by the margin)
    int32_t n_boxes;
    int32_t skip;          // this rank's own box (never selected)
};

__device__ __forceinline__ bool nm_in_box(const double* __restrict__ b, double x, double y, double z)
{
    return x >= b[0] && y >= b[1] && z >= b[2] && x <= b[3] && y <= b[4] && z <= b[5];
}

__global__ __launch_bounds__(256) void k_halo_count(const double* __restrict__ xyz, int64_t n,
                                                    int64_t stride, BoxSet B,
                                                    unsigned long long* __restrict__ counts)
{
    __shared__ double sbox[NM_MAX_BOXES * 6];
    __shared__ uint32_t scount[NM_MAX_BOXES];
    for (int t = threadIdx.x; t < B.n_boxes * 6; t += blockDim.x) sbox[t] = B.boxes[t];
    for (int t = threadIdx.x; t < B.n_boxes; t += blockDim.x) scount[t] = 0u;
    __syncthreads();
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
        for (int b = 0; b < B.n_boxes; ++b) {
            const bool in = valid && b != B.skip && nm_in_box(sbox + b * 6, x, y, z);
            const unsigned long long m = __ballot(in);
            if (m && (threadIdx.x & 63) == 0) atomicAdd(&scount[b], (uint32_t)__popcll(m));
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < B.n_boxes; t += blockDim.x)
        if (scount[t]) atomicAdd(&counts[t], (unsigned long long)scount[t]);
}

__global__ __launch_bounds__(256) void k_halo_pack(const double* __restrict__ xyz, int64_t n,
                                                   int64_t stride, BoxSet B,
                                                   const int64_t* __restrict__ offsets,
                                                   unsigned long long* __restrict__ cursor,
                                                   double* __restrict__ out)
{
    __shared__ double sbox[NM_MAX_BOXES * 6];
    for (int t = threadIdx.x; t < B.n_boxes * 6; t += blockDim.x) sbox[t] = B.boxes[t];
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
        for (int b = 0; b < B.n_boxes; ++b) {
            const bool in = valid && b != B.skip && nm_in_box(sbox + b * 6, x, y, z);
            const unsigned long long m = __ballot(in);
            if (!m) continue;
            // one cursor bump per wave and destination; order inside a segment is immaterial (the
            // search side only looks at which cells are occupied)
            unsigned long long base = 0;
            const int leader = __ffsll((long long)m) - 1;
            if (lane == leader) base = atomicAdd(&cursor[b], (unsigned long long)__popcll(m));
            base = __shfl(base, leader);
            if (in) {
                const int64_t row = offsets[b] + (int64_t)base +
                                    (int64_t)__popcll(m & ((1ull << lane) - 1ull));
                out[row * 3 + 0] = x;
                out[row * 3 + 1] = y;
                out[row * 3 + 2] = z;
            }
        }
    }
}

static int check_boxes(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride,
                       const double* d_boxes, int32_t n_boxes)
{
    if (n < 0 || stride < 3 || (n > 0 && !d_xyz) || !d_boxes || n_boxes < 1 ||
        n_boxes > NM_MAX_BOXES)
        NM_FAIL(ctx, NM_ERR_INVALID, "halo: bad arguments (at most %d boxes)", NM_MAX_BOXES);
    return NM_OK;
}

extern "C" int nm_halo_count(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride,
                             const double* d_boxes, int32_t n_boxes, int32_t skip, int64_t* d_counts,
                             void* stream)
{
    if (!ctx) return NM_ERR_INVALID;
    int rc = check_boxes(ctx, d_xyz, n, stride, d_boxes, n_boxes);
    if (rc) return rc;
    if (!d_counts) NM_FAIL(ctx, NM_ERR_INVALID, "nm_halo_count: d_counts is null");
    hipStream_t s = (hipStream_t)stream;
    NM_HIP(ctx, hipMemsetAsync(d_counts, 0, sizeof(int64_t) * n_boxes, s));
    if (n == 0) return NM_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    BoxSet B{d_boxes, n_boxes, skip};
    k_halo_count<<<(int)blocks, 256, 0, s>>>(d_xyz, n, stride, B, (unsigned long long*)d_counts);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}

extern "C" int nm_halo_pack(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride,
                            const double* d_boxes, int32_t n_boxes, int32_t skip,
                            const int64_t* d_offsets, int64_t* d_cursor, double* d_out, void* stream)
{
    if (!ctx) return NM_ERR_INVALID;
    int rc = check_boxes(ctx, d_xyz, n, stride, d_boxes, n_boxes);
    if (rc) return rc;
    if (!d_offsets || !d_cursor || !d_out)
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_halo_pack: null output arguments");
    hipStream_t s = (hipStream_t)stream;
    NM_HIP(ctx, hipMemsetAsync(d_cursor, 0, sizeof(int64_t) * n_boxes, s));
    if (n == 0) return NM_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    BoxSet B{d_boxes, n_boxes, skip};
    k_halo_pack<<<(int)blocks, 256, 0, s>>>(d_xyz, n, stride, B, d_offsets,
                                           (unsigned long long*)d_cursor, d_out);
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
    if (!ctx) return NM_ERR_INVALID;
    if (n < 0 || stride < 3 || (n > 0 && (!d_xyz || !d_out)))
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_copy_xyz: bad arguments");
    if (n == 0) return NM_OK;
    k_copy_xyz<<<(int)((n * 3 + 255) / 256), 256, 0, (hipStream_t)stream>>>(d_xyz, n, stride, d_out);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}
