/*
 * lattice_oracle.c - CPU ORACLE, plain C restatement of one analysis scale of nimrud/minimal.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/nimrud_oracle.py for the rules).  It exists
 * so that ALL rows of the full-size configurations can be checked on the CPU in seconds: it is
 * independent of scipy's kd-tree, of numpy.cov and of LAPACK, and - unlike the GPU path - it does NOT use
 * integer moments: it gathers the actual voxel centres of every neighborhood in fp64 and computes mean,
 * ddof=1 covariance and eigenvalues from them, the way the reference does.
 *
 * restated from (paths relative to the reference checkout):
 *   nimrud/utils/geometry.py:103-116   cell = floor((p - min_corner)/e); address = x + (y<<s0) + (z<<s1)
 *   nimrud/utils/geometry.py:142-154   unique occupied voxels, centre = cell*e + min_corner + e*0.5
 *   nimrud/minimal/multiscale.py:87-103 all voxel centres with ((dx*dx + dy*dy) + dz*dz) <= r*r  (scipy
 *                                       ckdtree p=2 compares squared distances; inclusive)
 *   nimrud/minimal/features.py:21-57   population; || q - mean ||; eigenvalues of the ddof=1 covariance
 *                                       (ascending), / their sum, [largest, middle]; zeros when undefined
 *
 * the neighbor search enumerates the lattice sites around the query's home cell and looks each one up in a
 * hash set of occupied addresses - a search STRATEGY of this oracle; the inclusion PREDICATE and the voxel
 * centres are the reference's.  eigenvalues by cyclic Jacobi rotations (fp64, converged to machine
 * precision).  compiled with -ffp-contract=off.  pinned against the golden vectors in tests/test_oracle.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    uint64_t* slots;   /* address + 1, 0 = empty */
    uint64_t mask;
} addr_set;

static inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

static int set_insert(addr_set* s, uint64_t a)
{
    uint64_t i = mix64(a) & s->mask;
    for (;;) {
        if (s->slots[i] == 0) {
            s->slots[i] = a + 1;
            return 1;
        }
        if (s->slots[i] == a + 1) return 0;
        i = (i + 1) & s->mask;
    }
}

static inline int set_has(const addr_set* s, uint64_t a)
{
    uint64_t i = mix64(a) & s->mask;
    for (;;) {
        if (s->slots[i] == 0) return 0;
        if (s->slots[i] == a + 1) return 1;
        i = (i + 1) & s->mask;
    }
}

/* geometry.py:137, evaluated left to right */
static inline double centre(int64_t cell, double mn, double e) { return ((double)cell * e + mn) + e * 0.5; }

/* eigenvalues of a symmetric 3x3 by cyclic Jacobi; returns them sorted ascending in w[0..2] */
static void jacobi3(double a[3][3], double w[3])
{
    for (int sweep = 0; sweep < 50; ++sweep) {
        double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off <= 1e-300 || off <= 1e-22 * diag) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
            }
    }
    w[0] = a[0][0];
    w[1] = a[1][1];
    w[2] = a[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (w[j] > w[j + 1]) {
                double t = w[j];
                w[j] = w[j + 1];
                w[j + 1] = t;
            }
}

/*
 * out[i*4 .. i*4+3] = [population, centroid distance, l1/sum, l2/sum] of query i.
 * returns the number of occupied voxels M, or -1 on allocation failure.
 */
long nm_oracle_scale(const double* query, long nq, long qstride, const double* search, long ns,
                     long sstride, const double* min_corner, double e, const int* widths, double r,
                     double* out, int threads)
{
    /* geometry.py:62: shifts = cumsum(widths)[:-1] */
    const int shifts[2] = {widths[0], widths[0] + widths[1]};
    uint64_t cap = 64;
    while (cap < (uint64_t)ns * 2) cap <<= 1;
    addr_set set;
    set.slots = (uint64_t*)calloc(cap, sizeof(uint64_t));
    set.mask = cap - 1;
    if (!set.slots) return -1;
    long m = 0;
    for (long i = 0; i < ns; ++i) {
        const double* p = search + i * sstride;
        /* geometry.py:108: floor((p - min)/e) */
        int64_t cx = (int64_t)floor((p[0] - min_corner[0]) / e);
        int64_t cy = (int64_t)floor((p[1] - min_corner[1]) / e);
        int64_t cz = (int64_t)floor((p[2] - min_corner[2]) / e);
        uint64_t addr = (uint64_t)(cx + (cy << shifts[0]) + (cz << shifts[1]));
        m += set_insert(&set, addr);
    }
    const double r2 = r * r;
    const int64_t reach = (int64_t)floor(r / e + 0.5 + 1e-9);
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    (void)threads;
#pragma omp parallel
    {
        long cap_nb = (2 * reach + 1) * (2 * reach + 1) * (2 * reach + 1);
        double* nb = (double*)malloc((size_t)cap_nb * 3 * sizeof(double));
#pragma omp for schedule(dynamic, 1024)
        for (long i = 0; i < nq; ++i) {
            const double* q = query + i * qstride;
            double* o = out + i * 4;
            o[0] = o[1] = o[2] = o[3] = 0.0;
            double fx = floor((q[0] - min_corner[0]) / e), fy = floor((q[1] - min_corner[1]) / e),
                   fz = floor((q[2] - min_corner[2]) / e);
            if (!(fabs(fx) < 4e18 && fabs(fy) < 4e18 && fabs(fz) < 4e18)) continue;
            int64_t hx = (int64_t)fx, hy = (int64_t)fy, hz = (int64_t)fz;
            long k = 0;
            for (int64_t gz = hz - reach; gz <= hz + reach; ++gz) {
                if (gz < 0 || gz >= ((int64_t)1 << widths[2])) continue;
                double cz = centre(gz, min_corner[2], e), dz = q[2] - cz;
                for (int64_t gy = hy - reach; gy <= hy + reach; ++gy) {
                    if (gy < 0 || gy >= ((int64_t)1 << widths[1])) continue;
                    double cy = centre(gy, min_corner[1], e), dy = q[1] - cy;
                    for (int64_t gx = hx - reach; gx <= hx + reach; ++gx) {
                        if (gx < 0 || gx >= ((int64_t)1 << widths[0])) continue;
                        double cxv = centre(gx, min_corner[0], e), dx = q[0] - cxv;
                        double s = (dx * dx + dy * dy) + dz * dz;
                        if (!(s <= r2)) continue;
                        uint64_t addr = (uint64_t)(gx + (gy << shifts[0]) + (gz << shifts[1]));
                        if (!set_has(&set, addr)) continue;
                        nb[3 * k] = cxv;
                        nb[3 * k + 1] = cy;
                        nb[3 * k + 2] = cz;
                        ++k;
                    }
                }
            }
            o[0] = (double)k;
            if (k == 0) continue;
            /* features.py:21-29: distance from the query to the mean of the neighborhood */
            double mean[3] = {0, 0, 0};
            for (long t = 0; t < k; ++t)
                for (int c = 0; c < 3; ++c) mean[c] += nb[3 * t + c];
            for (int c = 0; c < 3; ++c) mean[c] /= (double)k;
            double d0 = q[0] - mean[0], d1 = q[1] - mean[1], d2 = q[2] - mean[2];
            o[1] = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
            if (k < 2) continue;
            /* features.py:43-57: ddof=1 covariance of mean-centred coordinates, eigenvalues / sum */
            double cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            for (long t = 0; t < k; ++t) {
                double v[3] = {nb[3 * t] - mean[0], nb[3 * t + 1] - mean[1], nb[3 * t + 2] - mean[2]};
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) cov[a][b] += v[a] * v[b];
            }
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) cov[a][b] /= (double)(k - 1);
            double w[3];
            jacobi3(cov, w);
            double sum = w[0] + w[1] + w[2];
            if (sum > 0.0) {
                o[2] = w[2] / sum;
                o[3] = w[1] / sum;
            }
        }
        free(nb);
    }
    free(set.slots);
    return m;
}


/*
 * the neighbor lists themselves (multiscale.py:103: chunk_tree.query_ball_tree(search_tree, radius)) as voxel
 * ADDRESSES: for every query the addresses of the occupied voxels whose centre lies within r (inclusive, fp64,
 * ((dx*dx + dy*dy) + dz*dz) <= r*r), in ascending address order (z outer, x inner).  the caller turns an address
 * into the reference's search-voxel index with a binary search in the sorted unique address array (np.unique,
 * geometry.py:150).  two calls: addr_out == NULL fills counts[nq]; then offsets[nq + 1] (exclusive prefix of the
 * counts) and addr_out[offsets[nq]].  returns the number of occupied voxels M, or -1 on allocation failure.
 */
long nm_oracle_neighbors(const double* query, long nq, long qstride, const double* search, long ns, long sstride,
                         const double* min_corner, double e, const int* widths, double r, int* counts,
                         const long long* offsets, unsigned long long* addr_out, int threads)
{
    const int shifts[2] = {widths[0], widths[0] + widths[1]};
    uint64_t cap = 64;
    while (cap < (uint64_t)ns * 2) cap <<= 1;
    addr_set set;
    set.slots = (uint64_t*)calloc(cap, sizeof(uint64_t));
    set.mask = cap - 1;
    if (!set.slots) return -1;
    long m = 0;
    for (long i = 0; i < ns; ++i) {
        const double* p = search + i * sstride;
        int64_t cx = (int64_t)floor((p[0] - min_corner[0]) / e);
        int64_t cy = (int64_t)floor((p[1] - min_corner[1]) / e);
        int64_t cz = (int64_t)floor((p[2] - min_corner[2]) / e);
        m += set_insert(&set, (uint64_t)(cx + (cy << shifts[0]) + (cz << shifts[1])));
    }
    const double r2 = r * r;
    const int64_t reach = (int64_t)floor(r / e + 0.5 + 1e-9);
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    (void)threads;
#pragma omp parallel for schedule(dynamic, 1024)
    for (long i = 0; i < nq; ++i) {
        const double* q = query + i * qstride;
        long k = 0;
        unsigned long long* out = addr_out ? addr_out + offsets[i] : NULL;
        double fx = floor((q[0] - min_corner[0]) / e), fy = floor((q[1] - min_corner[1]) / e),
               fz = floor((q[2] - min_corner[2]) / e);
        if (fabs(fx) < 4e18 && fabs(fy) < 4e18 && fabs(fz) < 4e18) {
            int64_t hx = (int64_t)fx, hy = (int64_t)fy, hz = (int64_t)fz;
            for (int64_t gz = hz - reach; gz <= hz + reach; ++gz) {
                if (gz < 0 || gz >= ((int64_t)1 << widths[2])) continue;
                double dz = q[2] - centre(gz, min_corner[2], e);
                for (int64_t gy = hy - reach; gy <= hy + reach; ++gy) {
                    if (gy < 0 || gy >= ((int64_t)1 << widths[1])) continue;
                    double dy = q[1] - centre(gy, min_corner[1], e);
                    for (int64_t gx = hx - reach; gx <= hx + reach; ++gx) {
                        if (gx < 0 || gx >= ((int64_t)1 << widths[0])) continue;
                        double dx = q[0] - centre(gx, min_corner[0], e);
                        double s = (dx * dx + dy * dy) + dz * dz;
                        if (!(s <= r2)) continue;
                        uint64_t addr = (uint64_t)(gx + (gy << shifts[0]) + (gz << shifts[1]));
                        if (!set_has(&set, addr)) continue;
                        if (out) out[k] = addr;
                        ++k;
                    }
                }
            }
        }
        if (counts) counts[i] = (int)k;
    }
    free(set.slots);
    return m;
}
