/*
 * c_abi_demo.c - the multiscale operator through the C ABI alone: no Python, no torch.
 *
 * what a compiled host would do to replace nimrud/minimal/multiscale.py:27-67 (process_single_core) with
 * libnimrud_hip.so: device buffers from hipMalloc, the cloud's extrema from nm_bounds, the lattices of
 * VoxelFilter.__init__ (nimrud/utils/geometry.py:37-38, 55-64) built on the host, one nm_multiscale_features
 * call for the whole ladder - or, with NM_DEMO_DEVICE_LATTICE=1 in the environment, no lattice code at all:
 * nm_ladder_features measures the cloud and builds the lattices on the device, and nm_check reports what
 * VoxelFilter would have raised.  tests/test_gpu_parity.py runs both forms and compares their output with the
 * oracle and with the Python host's.
 *
 *   c_abi_demo <points.f64> <n> <features_out.f64> <edge> <radius> [<edge> <radius> ...]
 *
 * points.f64: n rows of 3 little-endian doubles; features_out.f64: n rows of 4*S doubles.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "nimrud_hip.h"

#define CHECK_HIP(call)                                                              \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));               \
            return 2;                                                                \
        }                                                                            \
    } while (0)

#define CHECK_NM(call)                                                               \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != 0) {                                                              \
            fprintf(stderr, "%s: %d %s\n", #call, rc_, nm_last_error(ctx));          \
            return 3;                                                                \
        }                                                                            \
    } while (0)

/* geometry.py:37-38, 55-64: corner = min - e/2, widths = ceil(log2((max + e/2 - corner) / e)) */
static int make_lattice(const double* lo, const double* hi, double e, nm_lattice* lat)
{
    int sum = 0;
    lat->edge = e;
    for (int a = 0; a < 3; ++a) {
        const double half = e / 2;
        const double min_corner = lo[a] - half, max_corner = hi[a] + half;
        const double w = ceil(log2((max_corner - min_corner) / e));
        if (w < 1.0) return -1;
        lat->min_corner[a] = min_corner;
        lat->widths[a] = (int32_t)w;
        sum += lat->widths[a];
    }
    if (sum > 64) return -1;                       /* geometry.py:59-60 */
    lat->shifts[0] = lat->widths[0];
    lat->shifts[1] = lat->widths[0] + lat->widths[1];
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 6 || (argc - 4) % 2 != 0) {
        fprintf(stderr, "usage: %s points.f64 n features_out.f64 edge radius [edge radius ...]\n", argv[0]);
        return 1;
    }
    const int64_t n = atoll(argv[2]);
    const int n_scales = (argc - 4) / 2;
    double* h_xyz = (double*)malloc((size_t)n * 24);
    FILE* f = fopen(argv[1], "rb");
    if (!f || !h_xyz || fread(h_xyz, 24, (size_t)n, f) != (size_t)n) {
        fprintf(stderr, "cannot read %lld points from %s\n", (long long)n, argv[1]);
        return 1;
    }
    fclose(f);

    nm_ctx* ctx = NULL;
    if (nm_create(&ctx, 0) != 0) {
        fprintf(stderr, "nm_create failed\n");
        return 3;
    }
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    double *d_xyz, *d_minmax, *d_feat;
    CHECK_HIP(hipMalloc((void**)&d_xyz, (size_t)n * 24));
    CHECK_HIP(hipMalloc((void**)&d_minmax, 6 * sizeof(double)));
    CHECK_HIP(hipMalloc((void**)&d_feat, (size_t)n * 4 * n_scales * sizeof(double)));
    CHECK_HIP(hipMemcpyAsync(d_xyz, h_xyz, (size_t)n * 24, hipMemcpyHostToDevice, stream));

    nm_lattice* lats = (nm_lattice*)calloc((size_t)n_scales, sizeof(nm_lattice));
    double* radii = (double*)calloc((size_t)n_scales, sizeof(double));
    double* edges = (double*)calloc((size_t)n_scales, sizeof(double));
    for (int s = 0; s < n_scales; ++s) {
        edges[s] = atof(argv[4 + 2 * s]);
        radii[s] = atof(argv[5 + 2 * s]);
    }
    const char* mode = getenv("NM_DEMO_DEVICE_LATTICE");
    const int device_lattice = mode && mode[0] == '1';
    size_t work_bytes;
    void* d_work;
    if (device_lattice) {
        /* everything on the device: no extrema, no lattices on the host, nothing to wait for */
        work_bytes = nm_ladder_workspace_bytes(n, n, n_scales);
        CHECK_HIP(hipMalloc(&d_work, work_bytes));
        CHECK_NM(nm_ladder_features(ctx, d_xyz, n, 3, d_xyz, n, 3, edges, radii, n_scales, NULL, d_feat,
                                    4 * (int64_t)n_scales, NULL, d_work, work_bytes, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        CHECK_NM(nm_check(ctx, 1));      /* e.g. "edge length is too small to address this space" */
    } else {
        /* the cloud's extrema (points.min(0) / points.max(0), geometry.py:37-38) */
        double mm[6];
        CHECK_NM(nm_bounds(ctx, d_xyz, n, 3, d_minmax, stream));
        CHECK_HIP(hipMemcpyAsync(mm, d_minmax, sizeof(mm), hipMemcpyDeviceToHost, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        for (int s = 0; s < n_scales; ++s) {
            if (make_lattice(mm, mm + 3, edges[s], &lats[s]) != 0) {
                fprintf(stderr, "edge length %g cannot address this space\n", edges[s]);
                return 1;
            }
        }
        work_bytes = nm_multiscale_workspace_bytes(n, n, lats, n_scales);
        CHECK_HIP(hipMalloc(&d_work, work_bytes));
        /* query cloud = search cloud: same pointer, same stride */
        CHECK_NM(nm_multiscale_features(ctx, d_xyz, n, 3, d_xyz, n, 3, lats, radii, n_scales, d_feat,
                                        4 * (int64_t)n_scales, NULL, d_work, work_bytes, stream));
    }
    double* h_feat = (double*)malloc((size_t)n * 4 * n_scales * sizeof(double));
    CHECK_HIP(hipMemcpyAsync(h_feat, d_feat, (size_t)n * 4 * n_scales * sizeof(double),
                             hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    f = fopen(argv[3], "wb");
    if (!f || fwrite(h_feat, 4 * n_scales * sizeof(double), (size_t)n, f) != (size_t)n) {
        fprintf(stderr, "cannot write %s\n", argv[3]);
        return 1;
    }
    fclose(f);
    printf("%lld points, %d scales, workspace %.1f MB, abi %d, lattices built on the %s\n", (long long)n,
           n_scales, work_bytes / 1e6, nm_abi_version(), device_lattice ? "device" : "host");
    (void)hipFree(d_work);
    (void)hipFree(d_feat);
    (void)hipFree(d_minmax);
    (void)hipFree(d_xyz);
    (void)hipStreamDestroy(stream);
    nm_destroy(ctx);
    free(h_feat);
    free(radii);
    free(edges);
    free(lats);
    free(h_xyz);
    return 0;
}
