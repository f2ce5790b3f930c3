// nm_features.hip - the fused per-scale kernel: lattice ball search + exact integer moments +
// fp64 3x3 symmetric eigen-solve + feature assembly.
//
// reference code replaced:
//   nimrud/minimal/multiscale.py:87-123  (kd-tree radius search, per-neighborhood feature loops)
//   nimrud/minimal/features.py:14-57     (take / population / centroid / pca)
//
// why there is no tree: the search set is the set of occupied sites of a cubic lattice
// (geometry.py:108,137), so the candidates of a query are the W^3 sites around its home cell
// (W = 2*floor(r/e + 1/2) + 1) and the neighborhood's first and second moments are sums of small
// integers - exact and order-independent.  only the inclusion test is floating point, and it is
// evaluated with the reference's own operation order on bit-identical voxel centres:
//     ((dx*dx + dy*dy) + dz*dz) <= r*r        (scipy ckdtree, p = 2, fp64, no FMA)
//
// execution model (gfx950, wave64): one 64-thread workgroup = one wave = 64 consecutive queries of
// the cell-sorted order, one query per lane.  the wave stages the occupancy bits of the box that
// covers all its lanes' candidate windows into LDS as 64-bit x-rows (funnel-shifted out of the
// 32x8x8 leaves of the index), then every lane walks its W*W rows: one LDS read gives the W
// occupancy bits of a row, W fp64 add/sub pairs give the W inside/outside bits, and a packed LDS
// lookup turns the surviving bit mask into (count, sum i, sum i^2).  lanes whose windows do not fit
// the staged box are deferred to another pass of the same wave.

#include "nm_common.h"
#include <type_traits>
#include "nm_index.h"

constexpr int ROWS_CAP = 512;   // staged (y,z) rows per wave, 8 B each
constexpr int SBT_CAP = 192;    // superblock slots of the staged box (3 in x)
constexpr int ANCHOR_EYZ = 22;  // y/z extent of the fallback box around an anchor lane
#ifndef NM_CENTRE_TABLE
#define NM_CENTRE_TABLE 1
#endif
#ifndef NM_FUSE_GATHER
#define NM_FUSE_GATHER 1        // the index builder gathers the coordinates into sorted order itself
#endif
#ifndef NM_FUSE_CLEAR
#define NM_FUSE_CLEAR 1         // the spatial sort's counting kernels reset the ladder's indexes (no clear launch)
#endif
#ifndef NM_DIAG_SUMS
#define NM_DIAG_SUMS 0          // row walk: sum j*k*n from per-(j+k) sums (adds) instead of a multiply-add per row
#endif
#ifndef NM_SLAB_MIN_W
#define NM_SLAB_MIN_W 11        // windows this wide and wider take the slab-fused path of the search kernel
#endif
constexpr int NM_BOX_EX = 62;   // x extent of a staged box: 64-bit rows, kept shifted left by two
#ifndef NM_KEEP_QUERY
#define NM_KEEP_QUERY 1         // the query stays in registers across the scale loop: 0 never, 1 not with the forest
                                // epilogue (97 registers there: a wave of occupancy), 2 always
#endif
#ifndef NM_SPLIT_SCALES_BELOW
#define NM_SPLIT_SCALES_BELOW 1600000     // (slots) the search kernel is launched with a workgroup per (scale, batch) up to here:
                                          // 300 k points -31 %, 1.25 M -7 %, 2.5 M +8 % (the queries are read per scale again)
#endif
#ifndef NM_ADAPTIVE_ANCHOR
#define NM_ADAPTIVE_ANCHOR 1    // the fallback box around an anchor lane takes its shape from the pending lanes' box
#endif
#ifndef NM_XREFLECT
#define NM_XREFLECT 1           // r = 3e: mirror the window in x as well (a bit-reversed copy of the staged rows)
#endif
constexpr int NM_BOX_EX_REFL = 60;   // ... and by two at the other end as well where a reversed copy is read
// a wave that keeps its rows twice: 430 rows, 72 superblock slots (an extent of 61 x 7 rows touches 9 x 2 superblocks
// in y and z, 54 slots) and the moment table are 7680 bytes = six of gfx950's 1280-byte LDS granules, 20 waves to a
// CU.  (512 rows, 9472 bytes, allocate 10240: 16 waves, and the kernel was 4 % slower than without the mirror)
#ifndef NM_ROWS_CAP_REFL
#define NM_ROWS_CAP_REFL 430
#endif
constexpr int NM_SBT_CAP_REFL = 72;
constexpr int NM_ANCHOR_EYZ_REFL = 20;

// one launch of the search kernel: the scales [s_begin, s_end) of a ladder whose per-scale data (lattice,
// index, r^2) lives in device memory.  all scales of one launch share the candidate window (W, dmin) and
// the static row bounds.
struct ScaleArgs {
    const double* query;
    int64_t nq;              // rows of the query cloud
    int64_t n_slots;         // entries of `order` (= nq, or the search size when queries are a prefix)
    int64_t qstride;
    const uint32_t* order;   // order[slot] = query row processed in sorted slot `slot`
    int32_t direct;          // 1: `query` is already in slot order, (n_slots,3); rows go to order[slot]
    const ScaleDev* scales;  // device array, one entry per scale of the ladder
    int32_t s_begin, s_end;
    int32_t dmin;            // first candidate offset (= -(W-1)/2)
    int32_t W;               // candidates per axis
    double* feat;            // scale s writes columns 4s..4s+3 of every row
    int64_t fstride;
    // kNN fallback on: one bit per slot and scale (scale s at sparse + s * sparse_words), set where the
    // population came out below sparse_k (else null)
    unsigned long long* sparse;
    int64_t sparse_words;
    int32_t sparse_k;
    // covariance output on: upper triangle per row, scale s at columns 6s..6s+5, else null
    double* cov;
    int64_t cstride;
    // normal output on: scale s at columns 3s..3s+2, else null
    double* normal;
    int64_t nstride;
    // classifier behind the last scale (FOREST kernels): rows of n_features = 4 * scales of the ladder
    ForestDev F;
};

// ---- 3x3 symmetric eigenvalues, fp64, non-iterative ------------------------------------------------
// eigenvalues of [[a00,a01,a02],[a01,a11,a12],[a02,a12,a22]] in descending order.
// the trigonometric solution of the characteristic cubic is accurate only for the eigenvalue that is
// well separated from the other two (the other two lose sqrt(eps) when they nearly coincide, which
// is the normal case here: collinear and coplanar lattice neighborhoods have exact double roots).
// so: take the separated eigenvalue from the cubic, form its eigenvector from the cross products of
// the rows of (A - lambda*I), deflate A onto the orthogonal complement and solve the remaining
// symmetric 2x2 in closed form (hypot form, no cancellation).  all three eigenvalues then carry an
// absolute error of a few ulp of ||A||, like LAPACK's.
__device__ __forceinline__ void nm_eig3(double a00, double a01, double a02, double a11, double a12,
                                        double a22, double& l0, double& l1, double& l2)
{
    const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
    const double q = (a00 + a11 + a22) * (1.0 / 3.0);
    const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
    const double p2 = b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * p1;
    if (!(p2 > 0.0)) {
        l0 = l1 = l2 = q;
        return;
    }
    const double p = sqrt(p2 * (1.0 / 6.0));
    const double inv = 1.0 / p;
    const double c00 = b00 * inv, c11 = b11 * inv, c22 = b22 * inv;
    const double c01 = a01 * inv, c02 = a02 * inv, c12 = a12 * inv;
    const double det = c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) +
                       c02 * (c01 * c12 - c11 * c02);
    const double r = fmin(fmax(det * 0.5, -1.0), 1.0);
    // with x = (lambda - q)/p the characteristic cubic is x^3 - 3x - 2r = 0, roots 2cos(phi + 2k*pi/3).
    // r >= 0: the largest root (in [sqrt3, 2]) is the separated one; r < 0: the smallest (mirror
    // image).  Newton from x = 2 descends monotonically onto that root; f' >= 6 there, so it is
    // quadratic and well conditioned.  (this replaces acos/cos: the fp64 library versions cost ~100
    // registers, i.e. two waves of occupancy.)  the approximate reciprocal only perturbs the step.
    const bool top = r >= 0.0;
    const double ra = fabs(r);
    double x = 2.0;
#pragma unroll
    for (int it = 0; it < 7; ++it) {
        const double x2 = x * x;
        const double f = x * (x2 - 3.0) - 2.0 * ra;
        const double fp = 3.0 * x2 - 3.0;
        x = x - f * __builtin_amdgcn_rcp(fp);
    }
    const double lam = top ? q + p * x : q - p * x;

    // eigenvector of lam: the cross product of two rows of (A - lam*I) with the largest norm
    const double m00 = a00 - lam, m11 = a11 - lam, m22 = a22 - lam;
    double x0 = a01 * a12 - a02 * m11, y0 = a02 * a01 - m00 * a12, z0 = m00 * m11 - a01 * a01;  // r0 x r1
    double x1 = a01 * m22 - a02 * a12, y1 = a02 * a02 - m00 * m22, z1 = m00 * a12 - a01 * a02;  // r0 x r2
    double x2 = m11 * m22 - a12 * a12, y2 = a12 * a02 - a01 * m22, z2 = a01 * a12 - m11 * a02;  // r1 x r2
    double n0 = x0 * x0 + y0 * y0 + z0 * z0;
    double n1 = x1 * x1 + y1 * y1 + z1 * z1;
    double n2 = x2 * x2 + y2 * y2 + z2 * z2;
    double vx = x0, vy = y0, vz = z0, nn = n0;
    if (n1 > nn) { vx = x1; vy = y1; vz = z1; nn = n1; }
    if (n2 > nn) { vx = x2; vy = y2; vz = z2; nn = n2; }
    if (!(nn > 0.0)) {
        // (A - lam*I) has rank <= 1: the other two eigenvalues coincide; they share what is left of
        // the trace
        const double rest = 0.5 * (3.0 * q - lam);
        if (top) {
            l0 = lam; l1 = rest; l2 = rest;
        } else {
            l0 = rest; l1 = rest; l2 = lam;
        }
        return;
    }
    const double vn = 1.0 / sqrt(nn);
    vx *= vn; vy *= vn; vz *= vn;
    // orthonormal basis (u, w) of the complement of v
    double ux, uy, uz;
    if (fabs(vx) > fabs(vy)) {
        const double s = 1.0 / sqrt(vx * vx + vz * vz);
        ux = -vz * s; uy = 0.0; uz = vx * s;
    } else {
        const double s = 1.0 / sqrt(vy * vy + vz * vz);
        ux = 0.0; uy = vz * s; uz = -vy * s;
    }
    const double wx = vy * uz - vz * uy, wy = vz * ux - vx * uz, wz = vx * uy - vy * ux;
    // 2x2 block of A in that basis
    const double aux = a00 * ux + a01 * uy + a02 * uz, auy = a01 * ux + a11 * uy + a12 * uz,
                 auz = a02 * ux + a12 * uy + a22 * uz;
    const double awx = a00 * wx + a01 * wy + a02 * wz, awy = a01 * wx + a11 * wy + a12 * wz,
                 awz = a02 * wx + a12 * wy + a22 * wz;
    const double e00 = ux * aux + uy * auy + uz * auz;
    const double e01 = ux * awx + uy * awy + uz * awz;
    const double e11 = wx * awx + wy * awy + wz * awz;
    const double mid = 0.5 * (e00 + e11), hd = 0.5 * (e00 - e11);
    const double rad = sqrt(hd * hd + e01 * e01);
    const double hi = mid + rad, lo = mid - rad;
    if (top) {
        l0 = lam; l1 = hi; l2 = lo;
    } else {
        l0 = hi; l1 = lo; l2 = lam;
    }
}

// ---- the same solve for the lattice kernels, cheaper ------------------------------------------------------
// all fp64, but the eigenvector is formed on the matrix scaled to unit size, and reciprocals and
// square roots come from the hardware approximations (v_rcp_f64 / v_rsq_f64) plus two Newton steps
// (full fp64 accuracy, no special-case handling; the library forms cost 12-14 instructions each).
// (an fp32 eigenvector was tried: 5 % faster kernel, but the features then agree with LAPACK to 3e-7
// instead of 1e-15 and l1 + l2 can exceed 1 by 1e-7 - not worth it.)
// (the file is compiled with -ffp-contract=off because cells, centres and squared distances must round like
// numpy's; none of that applies to the eigen-solve below, so it spells its fused multiply-adds out - a third
// fewer instructions in the epilogue, and one rounding less per term)
__device__ __forceinline__ double nm_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

__device__ __forceinline__ double nm_rcp_fast(double x)
{
    double y = __builtin_amdgcn_rcp(x);     // v_rcp_f64: full exponent range, ~26 good bits
    y = y * nm_fma(-x, y, 2.0);
    return y * nm_fma(-x, y, 2.0);
}
__device__ __forceinline__ double nm_rsqrt_fast(double x)
{
    double y = __builtin_amdgcn_rsq(x);     // v_rsq_f64
    const double hx = -0.5 * x;
    y = y * nm_fma(hx * y, y, 1.5);
    return y * nm_fma(hx * y, y, 1.5);
}

#ifndef NM_CUBIC_NEWTON_STEPS
#define NM_CUBIC_NEWTON_STEPS 3   // from the quadratic start: error 6.7e-4, 3.8e-7, 1.3e-13, < 1e-21 (the chord start of
                                  // round 2 - 1.3e-2 - needed four: one v_rcp_f64 and four fp64 instructions more)
#endif
__device__ __forceinline__ void nm_eig3_fast(double a00, double a01, double a02, double a11,
                                             double a12, double a22, double& l0, double& l1,
                                             double& l2)
{
    const double p1 = nm_fma(a01, a01, nm_fma(a02, a02, a12 * a12));
    const double q = (a00 + a11 + a22) * (1.0 / 3.0);
    const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
    const double p2 = nm_fma(b00, b00, nm_fma(b11, b11, nm_fma(b22, b22, 2.0 * p1)));
    if (!(p2 > 0.0)) {
        l0 = l1 = l2 = q;
        return;
    }
    const double x6 = p2 * (1.0 / 6.0);
    const double inv = nm_rsqrt_fast(x6);
    const double p = x6 * inv;
    const double c00 = b00 * inv, c11 = b11 * inv, c22 = b22 * inv;
    const double c01 = a01 * inv, c02 = a02 * inv, c12 = a12 * inv;
    const double m0 = nm_fma(c11, c22, -(c12 * c12));
    const double m1 = nm_fma(c01, c22, -(c12 * c02));
    const double m2 = nm_fma(c01, c12, -(c11 * c02));
    const double det = nm_fma(c00, m0, nm_fma(-c01, m1, c02 * m2));
    const double r = fmin(fmax(det * 0.5, -1.0), 1.0);
    const bool top = r >= 0.0;
    const double ra = fabs(r);
    // with x = (lambda - q)/p the characteristic cubic is x^3 - 3x - 2r = 0; for r >= 0 its largest root, in
    // [sqrt 3, 2], is the one well separated from the other two (r < 0: mirror image).  Newton from a
    // quadratic fit of the root (within 6.7e-4 of it) is quadratic and well conditioned (f' >= 6 there);
    // the approximate reciprocal only perturbs the step.
    // start: least-squares quadratic through the root 2 cos(acos(ra) / 3) on Chebyshev nodes of [0, 1]
    double x = nm_fma(ra, nm_fma(ra, -0.05343134715, 0.32018433581), 1.73271398205);
#pragma unroll
    for (int it = 0; it < NM_CUBIC_NEWTON_STEPS; ++it) {
        const double x2 = x * x;
        const double f = nm_fma(x, x2 - 3.0, -2.0 * ra);
        const double fp = nm_fma(3.0, x2, -3.0);
        x = nm_fma(-f, __builtin_amdgcn_rcp(fp), x);
    }
    // the other two roots of x^3 - 3x - 2r are -x/2 +- sqrt(3 (1 - x^2/4)).  with t = 1 - x^2/4
    // formed as (2-x)(2+x)/4 (the difference is exact) an error d in x becomes d / (2 sqrt t) in the
    // pair, so whenever the pair is not nearly double (t >= 1e-6: error < 1e-13) the closed form is as
    // good as the deflation below and a third of its cost.  nearly all neighborhoods take it; a wave
    // runs the deflation only when one of its lanes has a (near-)double pair.
    const double t = (2.0 - x) * (2.0 + x) * 0.25;
    if (__builtin_expect(t >= 1e-6, 1)) {
        const double t3 = 3.0 * t;
        const double s3 = t3 * nm_rsqrt_fast(t3);
        const double h = 0.5 * x;
        const double f0 = top ? x : h + s3;
        const double f1 = top ? s3 - h : h - s3;
        l0 = nm_fma(p, f0, q);
        l1 = nm_fma(p, f1, q);
        l2 = nm_fma(3.0, q, -l0) - l1;
        return;
    }
    const double sx = top ? x : -x;
    const double lam = nm_fma(p, sx, q);

    // eigenvector of the scaled matrix C - sx*I: the cross product of two rows with the largest norm
    const double m00 = c00 - sx, m11 = c11 - sx, m22 = c22 - sx;
    double x0 = c01 * c12 - c02 * m11, y0 = c02 * c01 - m00 * c12, z0 = m00 * m11 - c01 * c01;
    double x1 = c01 * m22 - c02 * c12, y1 = c02 * c02 - m00 * m22, z1 = m00 * c12 - c01 * c02;
    double x2 = m11 * m22 - c12 * c12, y2 = c12 * c02 - c01 * m22, z2 = c01 * c12 - m11 * c02;
    double n0 = x0 * x0 + y0 * y0 + z0 * z0;
    double n1 = x1 * x1 + y1 * y1 + z1 * z1;
    double n2 = x2 * x2 + y2 * y2 + z2 * z2;
    double vx = x0, vy = y0, vz = z0, nn = n0;
    if (n1 > nn) { vx = x1; vy = y1; vz = z1; nn = n1; }
    if (n2 > nn) { vx = x2; vy = y2; vz = z2; nn = n2; }
    if (!(nn > 1e-200)) {
        const double rest = 0.5 * (3.0 * q - lam);
        if (top) {
            l0 = lam; l1 = rest; l2 = rest;
        } else {
            l0 = rest; l1 = rest; l2 = lam;
        }
        return;
    }
    const double vn = nm_rsqrt_fast(nn);
    vx *= vn; vy *= vn; vz *= vn;
    double ux, uy, uz;
    if (fabs(vx) > fabs(vy)) {
        const double sc = nm_rsqrt_fast(vx * vx + vz * vz);
        ux = -vz * sc; uy = 0.0; uz = vx * sc;
    } else {
        const double sc = nm_rsqrt_fast(vy * vy + vz * vz);
        ux = 0.0; uy = vz * sc; uz = -vy * sc;
    }
    const double wx = vy * uz - vz * uy, wy = vz * ux - vx * uz, wz = vx * uy - vy * ux;
    // 2x2 block of A in that basis, fp64
    const double aux = a00 * ux + a01 * uy + a02 * uz, auy = a01 * ux + a11 * uy + a12 * uz,
                 auz = a02 * ux + a12 * uy + a22 * uz;
    const double awx = a00 * wx + a01 * wy + a02 * wz, awy = a01 * wx + a11 * wy + a12 * wz,
                 awz = a02 * wx + a12 * wy + a22 * wz;
    const double e00 = ux * aux + uy * auy + uz * auz;
    const double e01 = ux * awx + uy * awy + uz * awz;
    const double e11 = wx * awx + wy * awy + wz * awz;
    const double mid = 0.5 * (e00 + e11), hd = 0.5 * (e00 - e11);
    const double h2 = hd * hd + e01 * e01;
    const double rad = h2 > 0.0 ? h2 * nm_rsqrt_fast(h2) : 0.0;
    const double hi = mid + rad, lo = mid - rad;
    if (top) {
        l0 = lam; l1 = hi; l2 = lo;
    } else {
        l0 = hi; l1 = lo; l2 = lam;
    }
}

// features from the integer moments of a neighborhood, in candidate-index space (offset d = i + dmin)
//   n, S1 = sum of (i,j,k), S2 = sum of outer products; (qx - cx_home ...) = query minus home centre
__device__ __forceinline__ void nm_features_from_moments(
    double n, double sx, double sy, double sz, double sxx, double sxy, double sxz, double syy,
    double syz, double szz, double ux, double uy, double uz, double dmin, double edge, double* out)
{
    out[0] = n;
    out[1] = 0.0;
    out[2] = 0.0;
    out[3] = 0.0;
    if (n < 1.0) return;
    // centroid (features.py:21-29): mean of the neighbor centres = home centre + e*(S1/n + dmin)
    const double inv_n = nm_rcp_fast(n);
    double mx = nm_fma(-nm_fma(sx, inv_n, dmin), edge, ux);
    double my = nm_fma(-nm_fma(sy, inv_n, dmin), edge, uy);
    double mz = nm_fma(-nm_fma(sz, inv_n, dmin), edge, uz);
    const double d2 = nm_fma(mx, mx, nm_fma(my, my, mz * mz));
    out[1] = d2 > 0.0 ? d2 * nm_rsqrt_fast(d2) : 0.0;
    if (n < 2.0) return;   // covariance undefined: zeros (multiscale.py:4-5)
    // n*(n-1)/e^2 times the ddof=1 covariance (features.py:43), exact in integers:
    //   n*S2 - S1*S1^T.  normalised eigenvalues are invariant to that scale.
    double a00 = nm_fma(n, sxx, -(sx * sx)), a01 = nm_fma(n, sxy, -(sx * sy)), a02 = nm_fma(n, sxz, -(sx * sz));
    double a11 = nm_fma(n, syy, -(sy * sy)), a12 = nm_fma(n, syz, -(sy * sz)), a22 = nm_fma(n, szz, -(sz * sz));
    double l0, l1, l2;
    nm_eig3_fast(a00, a01, a02, a11, a12, a22, l0, l1, l2);
    double tr = a00 + a11 + a22;      // = l0 + l1 + l2 (features.py:55)
    const double inv_tr = nm_rcp_fast(tr);
    out[2] = l0 * inv_tr;
    out[3] = l1 * inv_tr;
}

// upper triangle of the ddof=1 covariance (features.py:43) from the integer moments; gx, gy, gz = -1 where
// the moments were taken in a frame mirrored on that axis.  cov = e^2 (n S2 - S1 S1^T) / (n (n - 1))
__device__ __forceinline__ void nm_covariance_from_moments(
    double n, double sx, double sy, double sz, double sxx, double sxy, double sxz, double syy,
    double syz, double szz, double gx, double gy, double gz, double edge, double* __restrict__ c)
{
    if (n < 2.0) {
        c[0] = c[1] = c[2] = c[3] = c[4] = c[5] = 0.0;
        return;
    }
    const double f = edge * edge / (n * (n - 1.0));
    c[0] = (n * sxx - sx * sx) * f;
    c[1] = (n * sxy - sx * sy) * f * (gx * gy);
    c[2] = (n * sxz - sx * sz) * f * (gx * gz);
    c[3] = (n * syy - sy * sy) * f;
    c[4] = (n * syz - sy * sz) * f * (gy * gz);
    c[5] = (n * szz - sz * sz) * f;
}

// unit eigenvector of the smallest eigenvalue of n S2 - S1 S1^T (the plane normal of the neighborhood), in
// the true frame (gx, gy, gz = -1 where the moments were taken mirrored), last non-zero of (x, y, z) positive.
// the smallest eigenvalue comes from the robust solver; its eigenvector is the largest cross product of
// two rows of A - lambda I, on the matrix scaled to unit size.
__device__ __forceinline__ void nm_normal_from_moments(
    double n, double sx, double sy, double sz, double sxx, double sxy, double sxz, double syy,
    double syz, double szz, double gx, double gy, double gz, double* __restrict__ v)
{
    v[0] = v[1] = v[2] = 0.0;
    if (n < 3.0) return;
    double a00 = n * sxx - sx * sx, a01 = n * sxy - sx * sy, a02 = n * sxz - sx * sz;
    double a11 = n * syy - sy * sy, a12 = n * syz - sy * sz, a22 = n * szz - sz * sz;
    const double tr = a00 + a11 + a22;
    if (!(tr > 0.0)) return;
    const double inv = 1.0 / tr;
    a00 *= inv; a01 *= inv; a02 *= inv; a11 *= inv; a12 *= inv; a22 *= inv;
    double l0, l1, l2;
    nm_eig3(a00, a01, a02, a11, a12, a22, l0, l1, l2);
    const double m00 = a00 - l2, m11 = a11 - l2, m22 = a22 - l2;
    double x0 = a01 * a12 - a02 * m11, y0 = a02 * a01 - m00 * a12, z0 = m00 * m11 - a01 * a01;
    double x1 = a01 * m22 - a02 * a12, y1 = a02 * a02 - m00 * m22, z1 = m00 * a12 - a01 * a02;
    double x2 = m11 * m22 - a12 * a12, y2 = a12 * a02 - a01 * m22, z2 = a01 * a12 - m11 * a02;
    const double n0 = x0 * x0 + y0 * y0 + z0 * z0, n1 = x1 * x1 + y1 * y1 + z1 * z1,
                 n2 = x2 * x2 + y2 * y2 + z2 * z2;
    double vx = x0, vy = y0, vz = z0, nn = n0;
    if (n1 > nn) { vx = x1; vy = y1; vz = z1; nn = n1; }
    if (n2 > nn) { vx = x2; vy = y2; vz = z2; nn = n2; }
    if (!(nn > 0.0)) {
        // A - lambda I has rank <= 1: every direction orthogonal to its one row direction will do
        vx = 0.0; vy = 0.0; vz = 1.0;
        const double r0 = m00 * m00 + a01 * a01 + a02 * a02, r1 = a01 * a01 + m11 * m11 + a12 * a12,
                     r2 = a02 * a02 + a12 * a12 + m22 * m22;
        double rx = m00, ry = a01, rz = a02, rr = r0;
        if (r1 > rr) { rx = a01; ry = m11; rz = a12; rr = r1; }
        if (r2 > rr) { rx = a02; ry = a12; rz = m22; rr = r2; }
        if (rr > 0.0) {
            // a vector orthogonal to (rx, ry, rz)
            if (fabs(rx) > fabs(rz)) { vx = -ry; vy = rx; vz = 0.0; }
            else { vx = 0.0; vy = -rz; vz = ry; }
            if (vx == 0.0 && vy == 0.0 && vz == 0.0) { vx = 1.0; }
        }
        nn = vx * vx + vy * vy + vz * vz;
    }
    const double s = 1.0 / sqrt(nn);
    vx *= s * gx;
    vy *= s * gy;
    vz *= s * gz;
    const bool flip = vz < 0.0 || (vz == 0.0 && (vy < 0.0 || (vy == 0.0 && vx < 0.0)));
    v[0] = flip ? -vx : vx;
    v[1] = flip ? -vy : vy;
    v[2] = flip ? -vz : vz;
}

// wave-wide min / max of an int32 with DPP row operations (6 VALU instructions + a readlane) instead
// of 6 rounds through the LDS crossbar: quad swaps, half-row and row mirrors leave every row of 16
// lanes holding its own result, row_bcast15 / row_bcast31 fold the four rows into lane 63.
template <bool IS_MIN>
__device__ __forceinline__ int32_t wave_reduce_i32(int32_t v)
{
#define NM_STEP(ctrl, rowmask)                                                              \
    {                                                                                       \
        const int32_t o = __builtin_amdgcn_update_dpp(v, v, ctrl, rowmask, 0xF, false);     \
        v = IS_MIN ? min(v, o) : max(v, o);                                                 \
    }
    NM_STEP(0xB1, 0xF)    // quad_perm [1,0,3,2]
    NM_STEP(0x4E, 0xF)    // quad_perm [2,3,0,1]
    NM_STEP(0x141, 0xF)   // row_half_mirror
    NM_STEP(0x140, 0xF)   // row_mirror
    NM_STEP(0x142, 0xA)   // row_bcast15 into rows 1 and 3
    NM_STEP(0x143, 0xC)   // row_bcast31 into rows 2 and 3
#undef NM_STEP
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int32_t wave_min_i32(int32_t v) { return wave_reduce_i32<true>(v); }
__device__ __forceinline__ int32_t wave_max_i32(int32_t v) { return wave_reduce_i32<false>(v); }

// the bounding box of the pending lanes needs three minima and three maxima per pass.  written through
// update_dpp each step costs three instructions (two copies and the min); as DPP-modified v_min / v_max
// it is one.  the six chains are interleaved, so a value is next read five instructions after it was
// written (the DPP read-after-write hazard needs two wait states), and the leading s_nop covers the
// instruction that produced the inputs and any preceding write of EXEC.
__device__ __forceinline__ void wave_bbox(int32_t& lox, int32_t& hix, int32_t& loy, int32_t& hiy,
                                          int32_t& loz, int32_t& hiz)
{
#define NM_SIX(ctrl)                                        \
    "v_min_i32_dpp %0, %0, %0 " ctrl "\n"                   \
    "v_max_i32_dpp %1, %1, %1 " ctrl "\n"                   \
    "v_min_i32_dpp %2, %2, %2 " ctrl "\n"                   \
    "v_max_i32_dpp %3, %3, %3 " ctrl "\n"                   \
    "v_min_i32_dpp %4, %4, %4 " ctrl "\n"                   \
    "v_max_i32_dpp %5, %5, %5 " ctrl "\n"
    asm volatile("s_nop 4\n"
                 NM_SIX("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 NM_SIX("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 NM_SIX("row_half_mirror row_mask:0xf bank_mask:0xf")
                 NM_SIX("row_mirror row_mask:0xf bank_mask:0xf")
                 NM_SIX("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 NM_SIX("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1\n"
                 : "+v"(lox), "+v"(hix), "+v"(loy), "+v"(hiy), "+v"(loz), "+v"(hiz));
#undef NM_SIX
    lox = __builtin_amdgcn_readlane(lox, 63);
    hix = __builtin_amdgcn_readlane(hix, 63);
    loy = __builtin_amdgcn_readlane(loy, 63);
    hiy = __builtin_amdgcn_readlane(hiy, 63);
    loz = __builtin_amdgcn_readlane(loz, 63);
    hiz = __builtin_amdgcn_readlane(hiz, 63);
}

__device__ __forceinline__ void lds_fence()
{
    // one wave per workgroup: LDS operations of a wave complete in order, the compiler must not
    // move accesses across this point.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---- static geometry of the candidate window ---------------------------------------------------------
// whatever the position of a query inside its home cell, candidate (dx,dy,dz) (cell offsets) lies at a
// distance between sqrt(sum max(|d|-1/2,0)^2) and sqrt(sum (|d|+1/2)^2) cells.  so for the row
// (dy,dz) of the window the x-candidates split, symmetrically about the centre, into
//   |dx| <= a : always inside  (no test)      a < |dx| <= b : must be tested      |dx| > b : never
// at r = 3e only 194 of the 343 candidates need the fp64 test.  eta pads the cell by 1e-4 cells for
// the rounding of the home-cell assignment and of the voxel centres (the host only enables pruning
// when 16 ulp of the largest coordinate is below that, see nm_scale_features).
struct RowBound {
    int8_t a, b;
};

constexpr double nm_sq(double v) { return v * v; }
constexpr double nm_pos(double v) { return v > 0.0 ? v : 0.0; }

// |d - u| for a cell offset d and a query offset u in [-eta, 1/2 + eta] (mirrored axes) or in
// [-1/2 - eta, 1/2 + eta] (x): smallest and largest possible value
constexpr double nm_abs(double v) { return v < 0.0 ? -v : v; }
constexpr double nm_max(double a, double b) { return a > b ? a : b; }
constexpr double nm_near(double d, double ulo, double uhi)
{
    // distance from d to the interval [ulo, uhi]
    return d < ulo ? ulo - d : (d > uhi ? d - uhi : 0.0);
}
constexpr double nm_far(double d, double ulo, double uhi)
{
    return nm_max(nm_abs(d - ulo), nm_abs(d - uhi));
}

// the kernel mirrors every lane's window in y and z so that the query lies in the upper half of its
// home cell on those axes (u in [0, 1/2]); rows are indexed in that mirrored frame.  in this table x stays
// two-sided: mirroring x needs the occupancy rows bit-reversed, which only the r = 3e instance pays for (a
// second, reversed copy of the staged rows; its classes come from nm_mirrored_class below).
constexpr RowBound nm_row_bound(int W, double rho2, int j, int k)
{
    const int c = (W - 1) / 2;
    const double eta = 1e-4;
    const double dy = (double)(j - c), dz = (double)(k - c);
    const double ny = nm_near(dy, -eta, 0.5 + eta), fy = nm_far(dy, -eta, 0.5 + eta);
    const double nz = nm_near(dz, -eta, 0.5 + eta), fz = nm_far(dz, -eta, 0.5 + eta);
    int a = -1, b = -1;
    for (int ax = 0; ax <= c; ++ax) {
        const double fx = nm_far((double)ax, -0.5 - eta, 0.5 + eta);
        const double nx = nm_near((double)ax, -0.5 - eta, 0.5 + eta);
        const double far2 = nm_sq(fx) + nm_sq(fy) + nm_sq(fz);
        const double near2 = nm_sq(nx) + nm_sq(ny) + nm_sq(nz);
        if (far2 <= rho2 * (1.0 - 1e-9) && a == ax - 1) a = ax;
        if (near2 <= rho2 * (1.0 + 1e-9)) b = ax;
    }
    return RowBound{(int8_t)a, (int8_t)b};
}

constexpr int NM_MAX_W = 11;
struct RowBoundTable {
    RowBound rb[NM_MAX_W * NM_MAX_W];   // [j * W + k]
};

template <int W>
constexpr RowBoundTable nm_make_bounds(double rho2, bool prune)
{
    RowBoundTable t{};
    for (int j = 0; j < W; ++j)
        for (int k = 0; k < W; ++k)
            t.rb[j * W + k] = prune ? nm_row_bound(W, rho2, j, k)
                                    : RowBound{(int8_t)-1, (int8_t)((W - 1) / 2)};
    return t;
}

// the benchmark ratio r = 3e gets its table at compile time, so the unrolled loops contain only the
// tests that can matter
constexpr RowBoundTable NM_BOUNDS_RHO3 = nm_make_bounds<7>(9.0, true);
// r = 4e and r = 5e (the reference's example ladder is one voxel edge with radii of 3, 4 and 5 edges:
// nimrud/utils/point_clouds.py:29-35) get theirs at compile time too: with the table in a kernel parameter it
// lives in 40-60 scalar registers and the kernel spills a thousand of them
constexpr RowBoundTable NM_BOUNDS_RHO4 = nm_make_bounds<9>(16.0, true);
constexpr RowBoundTable NM_BOUNDS_RHO5 = nm_make_bounds<11>(25.0, true);
// the compile-time table of an integer ratio RHO (W = 2 RHO + 1)
template <int RHO>
constexpr const RowBoundTable& nm_static_bounds()
{
    if constexpr (RHO == 3) return NM_BOUNDS_RHO3;
    else if constexpr (RHO == 4) return NM_BOUNDS_RHO4;
    else return NM_BOUNDS_RHO5;
}

// ---- the inside/outside masks of a query's window, as the search kernel packs them -------------------------
// row (j, k) of the window has W bits, candidate c in bit 2 + (k % RPR) * W + c of register j * RPJ + k / RPR
// (RPR = 30 / W rows to a register, every y-slab j starts a register of its own).
// for r = 3e the tests of one register form a compile-time chain: going down from the highest tested bit, every
// test enters the register with ONE v_alignbit whose shift is the distance to the next tested bit below it
//     x = (x << d) | (hi(t) >> (32 - d))          sign of t = "outside", lands on bit d - 1
// so after the last test (d = its position + 1) every sign sits on its candidate's bit.  the bits in between hold
// exponent bits of some t: they belong to candidates that are always or never inside, and the closing
// (~x & tested) | always  (one v_bitop3) overwrites them.
constexpr int NM_RHO3_W = 7, NM_RHO3_RPR = 30 / NM_RHO3_W, NM_RHO3_RPJ = (NM_RHO3_W + NM_RHO3_RPR - 1) / NM_RHO3_RPR;
struct ChainTable {
    int8_t d[NM_RHO3_W][NM_RHO3_W][NM_RHO3_W];            // [j][k][i]: alignbit distance, 0 = not tested
    uint32_t tested[NM_RHO3_W * NM_RHO3_RPJ];
    uint32_t always[NM_RHO3_W * NM_RHO3_RPJ];
};
// class of candidate (i, j, k) of a window mirrored on ALL THREE axes (the query in the upper half of its home
// cell on each): 0 never inside, 1 always inside, 2 must be tested.  same padding as nm_row_bound.
constexpr int nm_mirrored_class(int W, double rho2, int i, int j, int k)
{
    const int c = (W - 1) / 2;
    const double eta = 1e-4;
    const double dx = (double)(i - c), dy = (double)(j - c), dz = (double)(k - c);
    const double far2 = nm_sq(nm_far(dx, -eta, 0.5 + eta)) + nm_sq(nm_far(dy, -eta, 0.5 + eta)) +
                        nm_sq(nm_far(dz, -eta, 0.5 + eta));
    const double near2 = nm_sq(nm_near(dx, -eta, 0.5 + eta)) + nm_sq(nm_near(dy, -eta, 0.5 + eta)) +
                         nm_sq(nm_near(dz, -eta, 0.5 + eta));
    if (far2 <= rho2 * (1.0 - 1e-9)) return 1;
    if (near2 <= rho2 * (1.0 + 1e-9)) return 2;
    return 0;
}

// xrefl: the window is mirrored in x too (classes per candidate, no longer symmetric about the centre)
constexpr ChainTable nm_make_chain(const RowBoundTable& B, bool xrefl)
{
    constexpr int W = NM_RHO3_W, C = (W - 1) / 2, RPR = NM_RHO3_RPR, RPJ = NM_RHO3_RPJ;
    ChainTable t{};
    for (int j = 0; j < W; ++j)
        for (int h = 0; h < RPJ; ++h) {
            int prev_k = -1, prev_i = -1, prev_p = -1;          // the tested bit above the current one
            for (int k = (h + 1) * RPR < W ? (h + 1) * RPR - 1 : W - 1; k >= h * RPR; --k) {
                const RowBound rb = B.rb[j * W + k];
                for (int i = W - 1; i >= 0; --i) {
                    const int ad = i > C ? i - C : C - i;
                    const int p = 2 + (k % RPR) * W + i;
                    int cls = (rb.b < 0 || ad > rb.b) ? 0 : (ad <= rb.a ? 1 : 2);
                    if (xrefl) cls = nm_mirrored_class(W, 9.0, i, j, k);
                    if (cls == 0) continue;
                    if (cls == 1) {
                        t.always[j * RPJ + h] |= 1u << p;
                        continue;
                    }
                    t.tested[j * RPJ + h] |= 1u << p;
                    if (prev_p >= 0) t.d[j][prev_k][prev_i] = (int8_t)(prev_p - p);
                    prev_k = k; prev_i = i; prev_p = p;
                }
            }
            if (prev_p >= 0) t.d[j][prev_k][prev_i] = (int8_t)(prev_p + 1);
        }
    return t;
}

constexpr ChainTable NM_CHAIN_RHO3 = nm_make_chain(NM_BOUNDS_RHO3, NM_XREFLECT != 0);
constexpr int nm_chain_tests(const ChainTable& t)
{
    int n = 0;
    for (int r = 0; r < NM_RHO3_W * NM_RHO3_RPJ; ++r)
        for (int b = 0; b < 32; ++b) n += (t.tested[r] >> b) & 1u;
    return n;
}
static_assert(nm_chain_tests(NM_CHAIN_RHO3) == (NM_XREFLECT ? 88 : 115), "tests per query at r = 3e");

// ceil(2^20 / n) for the divisors the staging loops use (n <= ROWS_CAP): (t * v[n]) >> 20 is t / n, exactly,
// for every operand the loops form (t < ROWS_CAP; the static_assert below goes through all of them)
struct Recip20 {
    uint32_t v[ROWS_CAP + 1];
};
constexpr Recip20 nm_make_recip()
{
    Recip20 t{};
    t.v[0] = 0u;
    for (uint32_t n = 1; n <= ROWS_CAP; ++n) t.v[n] = ((1u << 20) + n - 1u) / n;
    return t;
}
constexpr bool nm_recip_exact()
{
    const Recip20 t = nm_make_recip();
    for (uint32_t n = 1; n <= ROWS_CAP; ++n)
        for (uint32_t x = 0; x < (uint32_t)ROWS_CAP; ++x)
            if (((x * t.v[n]) >> 20) != x / n) return false;
    return true;
}
static_assert(nm_recip_exact(), "the 20-bit reciprocals must divide every operand below ROWS_CAP exactly");
__device__ const Recip20 NM_RECIP20 = nm_make_recip();

// packed LUT fields: bits [0,8) count, [8,20) sum of bit index, [20,32) sum of index^2
template <int W>
struct MomentLut {
    uint32_t v[1 << W];
};
template <int W>
constexpr MomentLut<W> nm_make_lut()
{
    MomentLut<W> t{};
    for (int m = 0; m < (1 << W); ++m) {
        uint32_t n = 0, s1 = 0, s2 = 0;
        for (int i = 0; i < W; ++i)
            if (m & (1 << i)) {
                n += 1;
                s1 += i;
                s2 += i * i;
            }
        t.v[m] = n | (s1 << 8) | (s2 << 20);
    }
    return t;
}
template <int W>
__device__ const MomentLut<W> NM_LUT = nm_make_lut<W>();

// ---- per-lane slow path: any W, direct index lookups (no staging) -------------------------------------------
// used by the generic kernel (unusual radius/edge ratios) and, inside the table kernels, for lattices whose
// coordinates are so large that the static pruning of the window is not sound.  same arithmetic.
__device__ __forceinline__ void nm_lane_generic(const ScaleArgs& A, const LatticeDev& L, const IndexDev& I,
                                                uint32_t* stats, double r2, int32_t s, uint32_t qi, double qx,
                                                double qy, double qz, bool* sparse_out)
{
    const int32_t hx = nm_clamp_cell(nm_cell_f(qx, L.min_x, L.edge));
    const int32_t hy = nm_clamp_cell(nm_cell_f(qy, L.min_y, L.edge));
    const int32_t hz = nm_clamp_cell(nm_cell_f(qz, L.min_z, L.edge));
    const int32_t W = A.W, dmin = A.dmin;
    double n = 0, sx = 0, sy = 0, sz = 0, sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
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
            uint32_t word = 0;
            for (int32_t i = 0; i < W; ++i) {
                const int32_t gx = hx + dmin + i;
                if (gx < 0 || gx >= (1 << L.wx)) continue;
                const int32_t sbx = gx >> NM_SBX_BITS;
                if (sbx != cached_sb) {
                    cached_sb = sbx;
                    int32_t leaf = nm_hash_find(
                        I, nm_sb_key((uint32_t)sbx, (uint32_t)(gy >> NM_SBY_BITS),
                                     (uint32_t)(gz >> NM_SBZ_BITS), L));
                    word = leaf >= 0 ? I.leaf[(size_t)leaf * NM_LEAF_WORDS + (gz & 7) * 8 + (gy & 7)]
                                     : 0u;
                }
                if (!((word >> (gx & 31)) & 1u)) continue;
                d = qx - nm_centre(gx, L.min_x, L.edge, L.half_edge);
                const double sq = (d * d + dy2) + dz2;
                if (sq <= r2) {
                    n += 1.0;
                    sx += i;
                    sy += j;
                    sz += k;
                    sxx += (double)i * i;
                    sxy += (double)i * j;
                    sxz += (double)i * k;
                    syy += (double)j * j;
                    syz += (double)j * k;
                    szz += (double)k * k;
                }
            }
        }
    }
    double out[4];
    const double ux = qx - nm_centre(hx, L.min_x, L.edge, L.half_edge);
    const double uy = qy - nm_centre(hy, L.min_y, L.edge, L.half_edge);
    const double uz = qz - nm_centre(hz, L.min_z, L.edge, L.half_edge);
    nm_features_from_moments(n, sx, sy, sz, sxx, sxy, sxz, syy, syz, szz, ux, uy, uz, (double)dmin,
                             L.edge, out);
    double* o = A.feat + (int64_t)qi * A.fstride + 4 * s;
    o[0] = out[0];
    o[1] = out[1];
    o[2] = out[2];
    o[3] = out[3];
    if (A.cov)
        nm_covariance_from_moments(n, sx, sy, sz, sxx, sxy, sxz, syy, syz, szz, 1.0, 1.0, 1.0, L.edge,
                                   A.cov + (int64_t)qi * A.cstride + 6 * s);
    if (A.normal)
        nm_normal_from_moments(n, sx, sy, sz, sxx, sxy, sxz, syy, syz, szz, 1.0, 1.0, 1.0,
                               A.normal + (int64_t)qi * A.nstride + 3 * s);
    if (n < 2.0) atomicAdd(&stats[8], 1u);
    *sparse_out = n < (double)A.sparse_k;
}

// ---- the classifier behind the last scale ---------------------------------------------------------------------
// sklearn's RandomForestClassifier.predict / predict_proba (prototypes/apc.py:1463,1022,1034) on the row the
// wave has just finished: the lane's 4S features are read back from the feature matrix, cast to fp32 as
// sklearn does and staged in LDS ([feature][lane], conflict-free) over the search phase's buffers, then
// every lane walks the trees, sixteen at a time.  (measured: most of that read-back comes from HBM, not L2 -
// the five scales' writes of other waves have pushed the lines out - 1.4 GB per 10 M rows.  keeping the
// features in a dedicated LDS stage across the scale loop avoids it and is SLOWER, 2.36 against 2.0 ms: its
// 5 KB per wave cost a wave of occupancy, and the tree walk lives on waves in flight.)  the 64 lanes of a wave are neighbours in space: their paths mostly coincide, so a node fetch
// touches few cache lines.  node = 8 bytes {fp32 threshold, packed}; x_f32 <= threshold_f64 is evaluated as
// x_f32 <= largest fp32 not above the threshold, which is the same predicate.
// node = {fp32 threshold, packed}.  internal: left child << 13 | feature << 8, so that
// (packed & 0x1F00) | lane * 4 already is the LDS byte address of this lane's value of the feature;
// leaf: bit 31 | row of its class distribution << 13 (bits 8..12 clear: a leaf "reads" feature 0, harmlessly).
constexpr int NM_FOREST_GROUP = 8;      // trees per group (more than 8 per array: the compiler gives up unrolling)
#ifndef NM_FOREST_GROUPS
#define NM_FOREST_GROUPS 2               // groups in flight per lane
#endif

__device__ __forceinline__ void nm_forest_epilogue(const ScaleArgs& A, const uint2* __restrict__ nodes,
                                                   float* xs, int lane, bool have, uint32_t qi)
{
    const ForestDev& F = A.F;
    bool undefined = false;      // a scale of this ladder had no addressable lattice: its columns are NaN
    if (have) {
        const double* row = A.feat + (int64_t)qi * A.fstride;
        for (int f = 0; f < F.n_features; ++f) {
            const double v = row[f];
            undefined = undefined || v != v;
            xs[f * 64 + lane] = (float)v;
        }
    }
    lds_fence();
    if (!have) return;
    if (undefined) {
        if (F.proba)
            for (int c = 0; c < F.n_classes; ++c) F.proba[(int64_t)qi * F.pstride + c] = __builtin_nan("");
        if (F.label) F.label[qi] = -1;
        return;
    }
    const uint32_t lane4 = (uint32_t)lane * 4u;
    const char* xsb = (const char*)xs;
    double acc[NM_FUSED_FOREST_CLASSES];
#pragma unroll
    for (int c = 0; c < NM_FUSED_FOREST_CLASSES; ++c) acc[c] = 0.0;
    // a descent is a chain of dependent steps - LDS read of the feature, compare, node fetch from L1/L2 - and
    // the walk waits on memory most of the time.  the steps of one level are issued for all trees of all
    // NM_FOREST_GROUPS groups before anything is waited for: that many x 8 LDS reads, then as many node
    // fetches in flight per lane.  (a branch-free form - finished trees re-fetch their leaf - measured slower:
    // depths differ, and it pays the deepest tree's levels for all of them.)
#define NM_FOREST_LOAD(R, OFF)                                                                  \
    _Pragma("unroll") for (int g = 0; g < NM_FOREST_GROUP; ++g) {                               \
        const int t = t0 + (OFF) + g < F.n_trees ? t0 + (OFF) + g : F.n_trees - 1;              \
        R[g] = nodes[F.roots[t]];                                                               \
    }
#define NM_FOREST_READ(R, V)                                                                    \
    _Pragma("unroll") for (int g = 0; g < NM_FOREST_GROUP; ++g)                                 \
        V[g] = *(const float*)(xsb + ((R[g].y & 0x1F00u) | lane4));
#define NM_FOREST_STEP(R, V)                                                                    \
    _Pragma("unroll") for (int g = 0; g < NM_FOREST_GROUP; ++g) {                               \
        if ((int32_t)R[g].y >= 0) {                                                             \
            R[g] = nodes[(R[g].y >> 13) + (V[g] <= __uint_as_float(R[g].x) ? 0u : 1u)];         \
            any = true;                                                                         \
        }                                                                                       \
    }
#define NM_FOREST_VOTE(R, OFF)                                                                  \
    _Pragma("unroll") for (int g = 0; g < NM_FOREST_GROUP; ++g) {                               \
        if (t0 + (OFF) + g >= F.n_trees) break;                                                 \
        nm_forest_vote<NM_FUSED_FOREST_CLASSES>(F.leaf_value, F.leaf_stride, F.n_classes,       \
                                                (R[g].y >> 13) & 0x3FFFFu, acc);                \
    }
    for (int t0 = 0; t0 < F.n_trees; t0 += NM_FOREST_GROUPS * NM_FOREST_GROUP) {
        uint2 ra[NM_FOREST_GROUP], rb[NM_FOREST_GROUP];
        NM_FOREST_LOAD(ra, 0)
        NM_FOREST_LOAD(rb, NM_FOREST_GROUP)
#if NM_FOREST_GROUPS >= 3
        uint2 rc[NM_FOREST_GROUP];
        NM_FOREST_LOAD(rc, 2 * NM_FOREST_GROUP)
#endif
#if NM_FOREST_GROUPS >= 4
        uint2 rd[NM_FOREST_GROUP];
        NM_FOREST_LOAD(rd, 3 * NM_FOREST_GROUP)
#endif
        for (;;) {
            float va[NM_FOREST_GROUP], vb[NM_FOREST_GROUP];
            NM_FOREST_READ(ra, va)
            NM_FOREST_READ(rb, vb)
#if NM_FOREST_GROUPS >= 3
            float vc[NM_FOREST_GROUP];
            NM_FOREST_READ(rc, vc)
#endif
#if NM_FOREST_GROUPS >= 4
            float vd[NM_FOREST_GROUP];
            NM_FOREST_READ(rd, vd)
#endif
            bool any = false;
            NM_FOREST_STEP(ra, va)
            NM_FOREST_STEP(rb, vb)
#if NM_FOREST_GROUPS >= 3
            NM_FOREST_STEP(rc, vc)
#endif
#if NM_FOREST_GROUPS >= 4
            NM_FOREST_STEP(rd, vd)
#endif
            if (!any) break;
        }
        // the votes are added in tree order, like sklearn's accumulate-then-divide
        NM_FOREST_VOTE(ra, 0)
        NM_FOREST_VOTE(rb, NM_FOREST_GROUP)
#if NM_FOREST_GROUPS >= 3
        NM_FOREST_VOTE(rc, 2 * NM_FOREST_GROUP)
#endif
#if NM_FOREST_GROUPS >= 4
        NM_FOREST_VOTE(rd, 3 * NM_FOREST_GROUP)
#endif
    }
#undef NM_FOREST_LOAD
#undef NM_FOREST_READ
#undef NM_FOREST_STEP
#undef NM_FOREST_VOTE
    int best = 0;
    double bestv = -1.0;
#pragma unroll
    for (int c = 0; c < NM_FUSED_FOREST_CLASSES; ++c) {
        if (c < F.n_classes) {
            const double pr = acc[c] / (double)F.n_trees;
            if (F.proba) F.proba[(int64_t)qi * F.pstride + c] = pr;
            if (pr > bestv) {   // first maximum wins, like numpy.argmax
                bestv = pr;
                best = c;
            }
        }
    }
    if (F.label) F.label[qi] = best;
}

// the same classifier as a kernel of its own, over the finished rows in the ladder's spatial order (the
// lanes of a wave are neighbours in space, as in the epilogue) with all the registers and LDS to itself.
// measured SLOWER than the epilogue (2.9 against 2.1 ms, 10 M rows x 32 trees): the tree walk is a chain of
// dependent L1/L2 fetches and waits most of the time; in the epilogue it waits while other waves of the same
// SIMD run their vector-ALU bound search.  used when the last kernel cannot carry the classifier (unusual
// radius/edge ratio, kNN fallback on) or when nm_set_forest_mode asks for it.
__global__ __launch_bounds__(64) void k_forest_ordered(ScaleArgs A, const uint2* __restrict__ nodes)
{
    __shared__ float xs[NM_FUSED_FOREST_FEATURES * 64];
    const int lane = threadIdx.x;
    const int64_t batch = nm_xcd_batch(blockIdx.x, gridDim.x);
    const int64_t slot = batch * 64 + lane;
    bool have = slot < A.n_slots;
    uint32_t qi = 0;
    if (have) {
        qi = A.order[slot];
        have = qi < A.nq;
    }
    nm_forest_epilogue(A, nodes, xs, lane, have, qi);
}

#ifndef NM_SEARCH_ATTR
#define NM_SEARCH_ATTR
#endif
// a * k + c through the full-rate 24-bit multiplier, for a constant k and a sum whose low bits are all that
// is read: written as C the compiler sees that only low bits are needed, drops the "24-bit" and picks
// v_mul_lo_u32 (quarter rate) for k = 3, 5, 6
__device__ __forceinline__ uint32_t nm_mad24(uint32_t a, uint32_t k, uint32_t c)
{
    if (k == 0u) return c;
    if (k == 1u) return a + c;
    if ((k & (k - 1u)) == 0u) return (a << __builtin_ctz(k)) + c;
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(k), "v"(c));
    return r;
}

// a * b + c for small signed numbers (the compiler turns __mul24(a, b) + c into v_mad_u64_u32, quarter rate)
__device__ __forceinline__ int32_t nm_mad24i(int32_t a, int32_t b, int32_t c)
{
    int32_t r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// base + row * stride + col with one 32 x 32 -> 64 multiply-add (rows and strides are below 2^31, checked
// on the host)
template <typename T>
__device__ __forceinline__ T* nm_row_ptr(T* base, uint32_t row, int64_t stride, int32_t col)
{
    return base + ((uint64_t)row * (uint64_t)(uint32_t)stride + (uint64_t)(uint32_t)col);
}

// RHO: 0 = the window's static bounds come in the kernel parameter RT; 3, 4, 5 = r is that many edges and the
// bounds are compile-time constants (RHO = 3, the benchmark's ratio, also has its tests chained at compile time)
typedef const __attribute__((address_space(1))) uint32_t* NmGlobalU32;
static_assert(NM_LEAF_WORDS == 64, "leaf words are addressed as (leaf << 6) + word");

template <int W, int RHO, bool FOREST, bool LOOP>
__global__ __launch_bounds__(64) NM_SEARCH_ATTR void k_scale_features(ScaleArgs A, RowBoundTable RT,
                                                       const ScaleDev* __restrict__ scales,
                                                       const uint2* __restrict__ forest_nodes)
{
    static_assert(W >= 3 && W <= 11 && (W & 1), "LUT kernel covers W = 3,5,7,9,11");
    static_assert(RHO == 0 || W == 2 * RHO + 1, "a compile-time table is for the window of its ratio");
    static_assert(RHO == 0 || (RHO >= 3 && RHO <= 5), "compile-time tables exist for r = 3e, 4e, 5e");
    constexpr bool RHO3 = RHO == 3;
    constexpr int C = (W - 1) / 2;
    constexpr int RPR = 30 / W;                   // rows of the window per mask register, from bit 2
    constexpr int RPJ = (W + RPR - 1) / RPR;      // mask registers per y-slab
    constexpr int MASK_REGS = W * RPJ;
    static_assert(!RHO3 || (RPR == NM_RHO3_RPR && RPJ == NM_RHO3_RPJ), "layout of the compile-time chains");
    // one block of LDS: the staged rows, the superblock table and the moment table; the classifier's
    // feature stage reuses all of it after the last scale
    // wide windows (W >= 11: 121 row masks would need 66 registers) test and walk one y-slab at a time INSIDE the
    // pass loop, behind the staging: their centre table cannot share the row buffer
    constexpr bool SLAB = W >= NM_SLAB_MIN_W;
    // r = 3e: the window is mirrored in x too, per lane (88 instead of 115 tests).  the occupancy cannot be
    // mirrored per lane at a bearable price (a bit reversal per row and lane), so the staged rows are kept twice,
    // as read and bit-reversed: a mirrored lane walks the reversed copy and everything else is as for y and z
    constexpr bool XREFL = NM_XREFLECT && RHO3 && !SLAB;
    constexpr int BOX_EX = XREFL ? NM_BOX_EX_REFL : NM_BOX_EX;
    constexpr int RCAP = XREFL ? NM_ROWS_CAP_REFL : ROWS_CAP;
    constexpr int SCAP = XREFL ? NM_SBT_CAP_REFL : SBT_CAP;
    constexpr int ANCHOR = XREFL ? NM_ANCHOR_EYZ_REFL : ANCHOR_EYZ;
    static_assert(RCAP <= ROWS_CAP && ANCHOR * ANCHOR <= RCAP && 3 * ((ANCHOR + 6) / 8 + 1) * ((ANCHOR + 6) / 8 + 1) <= SCAP,
                  "the anchor box must fit");
    constexpr int CTAB_OFS = RCAP * 8 + SCAP * 4 + (4 << W);
    constexpr int REV_OFS = CTAB_OFS + (SLAB ? 3 * 64 * 8 : 0);     // (skewing the copy by 8 or 128 bytes against
                                                                    //  bank conflicts changes nothing, measured)
    constexpr int SEARCH_BYTES = REV_OFS + (XREFL ? RCAP * 8 : 0);
    constexpr int STAGE_BYTES = FOREST ? NM_FUSED_FOREST_FEATURES * 64 * 4 : 0;   // = SEARCH_BYTES at W = 7
    constexpr int LDS_BYTES = SEARCH_BYTES > STAGE_BYTES ? SEARCH_BYTES : STAGE_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
    uint64_t* rows = (uint64_t*)lds_raw;
    int32_t* sbt = (int32_t*)(lds_raw + RCAP * 8);
    uint32_t* lut = (uint32_t*)(lds_raw + RCAP * 8 + SCAP * 4);

    const int lane = threadIdx.x;
    const int32_t dmin = A.dmin;
    const int32_t dmax = dmin + W - 1;

    // LOOP = false: one scale per workgroup.  a launch may then cover several scales, workgroups (scale, batch) with
    // the scale the slow index (small clouds: a 1.25 M-point tile's 19 531 waves of five scales each are 3.8 rounds
    // of the card's wave slots, and the last round's tail is a tenth of the kernel; 97 656 fifths are 19 rounds)
    const int64_t n_batches = (A.n_slots + 63) / 64;
    const int32_t s_first = LOOP ? A.s_begin : A.s_begin + (int32_t)((int64_t)blockIdx.x / n_batches);
    const int64_t batch = LOOP ? nm_xcd_batch(blockIdx.x, gridDim.x)
                               : nm_xcd_batch((int64_t)blockIdx.x % n_batches, n_batches);
    // the moment table comes from constant data (computing it cost every wave ~70 vector instructions)
    for (int m = lane; m < (1 << W); m += 64) lut[m] = NM_LUT<W>.v[m];

    // every scale of this launch in turn: the wave keeps its 64 queries.  nothing else per-lane is carried from
    // one scale to the next in registers
    // the wave keeps its 64 queries in registers through the scale loop (six registers: 94 of the 96 that five
    // waves allow; read again per scale - from HBM, the L2 has long moved on - the kernel was 3-4 % slower)
    constexpr bool KEEPQ = NM_KEEP_QUERY >= 2 || (NM_KEEP_QUERY == 1 && !FOREST);
    const int64_t slot = batch * 64 + lane;
    bool have0 = slot < A.n_slots;
    uint32_t qi = 0;
    double qx0 = 0.0, qy0 = 0.0, qz0 = 0.0;
    if (have0) {
        qi = A.order[slot];
        have0 = qi < A.nq;    // prefix mode: the sorted order also holds the non-query search rows
    }
    if (KEEPQ && have0) {
        const double* p = nm_row_ptr(A.query, A.direct ? (uint32_t)slot : qi, A.qstride, 0);
        qx0 = p[0];
        qy0 = p[1];
        qz0 = p[2];
    }
#pragma nounroll
    for (int32_t s = s_first; s < (LOOP ? A.s_end : s_first + 1); ++s) {
    bool have = have0;
    double qx = qx0, qy = qy0, qz = qz0;
    if (!KEEPQ && have) {
        const double* p = nm_row_ptr(A.query, A.direct ? (uint32_t)slot : qi, A.qstride, 0);
        qx = p[0];
        qy = p[1];
        qz = p[2];
    }
    // (the scale array is a kernel parameter of its own, restrict-qualified: the compiler can then prove
    // that nothing in the kernel writes it and reads it with scalar loads into SGPRs)
    const ScaleDev* __restrict__ S = scales + s;
    if (!S->valid) {
        // a lattice that cannot be addressed: reported through the context's status words at the caller's next
        // synchronisation point; until then its columns hold NaN, never stale memory that looks like features
        if (have) {
            double* o = nm_row_ptr(A.feat, qi, A.fstride, 4 * s);
            o[0] = o[1] = o[2] = o[3] = __builtin_nan("");
        }
        continue;
    }
    const LatticeDev L = S->L;
    const IndexDev I = S->I;
    const double r2 = S->r2;
    // coordinates too large for the static window bounds: this scale is left to k_scale_features_fallback
    // (keeping the per-lane walk out of this kernel keeps its registers at 80)
    if (!S->prune_ok) continue;
    const int32_t hx = nm_cell_index(qx, L.min_x, L.edge, L.inv_edge, -1073741824, 1073741823);
    const int32_t hy = nm_cell_index(qy, L.min_y, L.edge, L.inv_edge, -1073741824, 1073741823);
    const int32_t hz = nm_cell_index(qz, L.min_z, L.edge, L.inv_edge, -1073741824, 1073741823);
    // a query whose whole candidate window lies outside the lattice has no neighbors
    const bool far = hx + dmax < 0 || hx + dmin >= (1 << L.wx) || hy + dmax < 0 ||
                     hy + dmin >= (1 << L.wy) || hz + dmax < 0 || hz + dmin >= (1 << L.wz);
    bool done = !have || far;
    if (have && far) {
        double* o = nm_row_ptr(A.feat, qi, A.fstride, 4 * s);
        o[0] = 0.0;
        o[1] = 0.0;
        o[2] = 0.0;
        o[3] = 0.0;
        if (A.cov) {
            double* c = nm_row_ptr(A.cov, qi, A.cstride, 6 * s);
            c[0] = c[1] = c[2] = c[3] = c[4] = c[5] = 0.0;
        }
        if (A.normal) {
            double* v = nm_row_ptr(A.normal, qi, A.nstride, 3 * s);
            v[0] = v[1] = v[2] = 0.0;
        }
    }

    // the wave's box of home cells (it is also the first pass's box).  when it is at most 64 cells wide
    // on every axis - nearly always - each voxel centre the wave will need is computed ONCE, by one lane,
    // into a table in LDS (bit-identical: same expression), instead of 3 W times by every lane
    int32_t blox = done ? INT32_MAX : hx, bhix = done ? INT32_MIN : hx;
    int32_t bloy = done ? INT32_MAX : hy, bhiy = done ? INT32_MIN : hy;
    int32_t bloz = done ? INT32_MAX : hz, bhiz = done ? INT32_MIN : hz;
    wave_bbox(blox, bhix, bloy, bhiy, bloz, bhiz);
    const bool tab = NM_CENTRE_TABLE && blox <= bhix && (int64_t)bhix - blox + W <= 64 &&
                     (int64_t)bhiy - bloy + W <= 64 && (int64_t)bhiz - bloz + W <= 64;
    double* ctab = SLAB ? (double*)(lds_raw + CTAB_OFS)
                        : (double*)rows;      // 3 x 64 doubles, twice; the row buffer is not in use yet
    lds_fence();                       // the previous scale's last pass has read the row buffer
    if (tab) {
        const double ccx = nm_centre(blox + dmin + lane, L.min_x, L.edge, L.half_edge);
        const double ccy = nm_centre(bloy + dmin + lane, L.min_y, L.edge, L.half_edge);
        const double ccz = nm_centre(bloz + dmin + lane, L.min_z, L.edge, L.half_edge);
        ctab[lane] = ccx;
        ctab[64 + lane] = ccy;
        ctab[128 + lane] = ccz;
        if constexpr (!SLAB) {
            // ... and once more back to front (entry m of a reversed table is entry 63 - m): a mirrored lane
            // then reads its window with the same compile-time offsets as the others, from another base, instead
            // of forming an address per candidate
            ctab[192 + 63 - lane] = ccx;
            ctab[256 + 63 - lane] = ccy;
            ctab[320 + 63 - lane] = ccz;
        }
    }
    lds_fence();
    // table positions of this lane's window (lanes that are done read somewhere harmless)
    const int32_t tx = done ? 0 : hx - blox, ty = done ? C : hy - bloy + C, tz = done ? C : hz - bloz + C;

    // reflect the window in y and z so that the query is in the upper half of its home cell there
    // (population, centroid distance and eigenvalues are invariant under these reflections)
    const double uy_home = qy - (tab ? ctab[64 + ty] : nm_centre(hy, L.min_y, L.edge, L.half_edge));
    const double uz_home = qz - (tab ? ctab[128 + tz] : nm_centre(hz, L.min_z, L.edge, L.half_edge));
    const int32_t sgn_y = uy_home < 0.0 ? -1 : 1;
    const int32_t sgn_z = uz_home < 0.0 ? -1 : 1;
    int32_t sgn_x = 1;
    if constexpr (XREFL) {
        const double ux_home = qx - (tab ? ctab[tx + C] : nm_centre(hx, L.min_x, L.edge, L.half_edge));
        sgn_x = ux_home < 0.0 ? -1 : 1;
    }

    // ---- phase A (once per wave and scale): the inside/outside bit of every candidate that needs a test,
    //      as W-bit row masks packed RPR to a register.  independent of the occupancy.
    uint32_t inside[SLAB ? 1 : MASK_REGS];
    if constexpr (!SLAB) {
        // squared coordinate differences to the W candidate centres per axis (bit-identical centres)
        // index i of the y and z tables is in the lane's mirrored frame: cell = home + sgn*(i - C)
        double dx2[W], dy2[W], dz2[W];
        if (tab) {
            // (ctab[t - k] = reversed[63 - t + k]: window index i is offset i - C from the lane's base in either)
            const double* cx = XREFL && sgn_x < 0 ? ctab + 192 + 63 - (tx + C) : ctab + tx + C;
            const double* cy = sgn_y < 0 ? ctab + 256 + 63 - ty : ctab + 64 + ty;
            const double* cz = sgn_z < 0 ? ctab + 320 + 63 - tz : ctab + 128 + tz;
#pragma unroll
            for (int i = 0; i < W; ++i) {
                double d = qx - cx[i - C];
                dx2[i] = d * d;
                d = qy - cy[i - C];
                dy2[i] = d * d;
                d = qz - cz[i - C];
                dz2[i] = d * d;
            }
        } else {
#pragma unroll
            for (int i = 0; i < W; ++i) {
                double d = qx - nm_centre(XREFL ? hx + sgn_x * (i - C) : hx + dmin + i, L.min_x, L.edge, L.half_edge);
                dx2[i] = d * d;
                d = qy - nm_centre(hy + sgn_y * (i - C), L.min_y, L.edge, L.half_edge);
                dy2[i] = d * d;
                d = qz - nm_centre(hz + sgn_z * (i - C), L.min_z, L.edge, L.half_edge);
                dz2[i] = d * d;
            }
        }
#pragma unroll
        for (int r = 0; r < MASK_REGS; ++r) inside[r] = 0u;
#pragma unroll
        for (int j = 0; j < W; ++j) {
            double pxy[W];
#pragma unroll
            for (int i = 0; i < W; ++i) pxy[i] = dx2[i] + dy2[j];
            if constexpr (RHO3) {
                // compile-time chains (see ChainTable): one v_alignbit per test, one v_bitop3 per register
#pragma unroll
                for (int h = 0; h < RPJ; ++h) {
                    uint32_t x = 0u;
#pragma unroll
                    for (int k = ((h + 1) * RPR < W ? (h + 1) * RPR : W) - 1; k >= h * RPR; --k) {
#pragma unroll
                        for (int i = W - 1; i >= 0; --i) {
                            const int d = NM_CHAIN_RHO3.d[j][k][i];
                            if (d == 0) continue;
                            const double sq = pxy[i] + dz2[k];
                            const double t = r2 - sq;      // sign bit set  <=>  sq > r^2  (exact)
                            x = __builtin_amdgcn_alignbit(x, (uint32_t)__double2hiint(t), 32 - d);
                        }
                    }
                    inside[j * RPJ + h] = (~x & NM_CHAIN_RHO3.tested[j * RPJ + h]) | NM_CHAIN_RHO3.always[j * RPJ + h];
                }
            } else {
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    RowBound rb = RT.rb[j * W + k];
                    if constexpr (RHO > 0) rb = nm_static_bounds<RHO>().rb[j * W + k];
                    // "outside" bits of the candidates C-b .. C+b (the ones further out are never inside: their
                    // bits of the row mask simply stay clear), candidate C-b in bit 0
                    if (rb.b < 0) continue;
                    uint32_t outside = 0u;
#pragma unroll
                    for (int i = W - 1; i >= 0; --i) {
                        const int ad = i > C ? i - C : C - i;
                        if (ad > rb.b) continue;
                        if (ad <= rb.a) {
                            outside = outside << 1;
                        } else {
                            const double sq = pxy[i] + dz2[k];
                            const double t = r2 - sq;      // sign bit set  <=>  sq > r^2  (exact)
                            outside = __builtin_amdgcn_alignbit(outside, (uint32_t)__double2hiint(t), 31);
                        }
                    }
                    const uint32_t span = (1u << (2 * rb.b + 1)) - 1u;
                    inside[j * RPJ + k / RPR] |= ((~outside) & span) << ((k % RPR) * W + 2 + (C - rb.b));
                }
            }
            // keep the rows of different j apart: without this the scheduler interleaves all W*W
            // chains and the live set (W*W partial sums) costs two waves of occupancy
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    lds_fence();

    uint32_t m_n = 0, m_sx = 0, m_sy = 0, m_sz = 0, m_sxx = 0, m_sxy = 0, m_sxz = 0, m_syy = 0,
             m_syz = 0, m_szz = 0;
    uint32_t passes = 0;
    for (;;) {
        const unsigned long long todo = __ballot(!done);
        if (!todo) break;
        ++passes;
        // ---- choose the box: bounding box of the pending lanes if it fits, else a fixed box around
        //      the first pending lane
        int32_t lox = blox, hix = bhix, loy = bloy, hiy = bhiy, loz = bloz, hiz = bhiz;
        if (passes > 1) {      // the first pass's box is the wave's box, already known
            lox = done ? INT32_MAX : hx; hix = done ? INT32_MIN : hx;
            loy = done ? INT32_MAX : hy; hiy = done ? INT32_MIN : hy;
            loz = done ? INT32_MAX : hz; hiz = done ? INT32_MIN : hz;
            wave_bbox(lox, hix, loy, hiy, loz, hiz);
        }
        int64_t ex64 = (int64_t)hix - lox + W, ey64 = (int64_t)hiy - loy + W,
                ez64 = (int64_t)hiz - loz + W;
        int32_t ox = lox + dmin, oy = loy + dmin, oz = loz + dmin;
        int32_t ey = (int32_t)ey64, ez = (int32_t)ez64;
        int32_t ex = ex64 < BOX_EX ? (int32_t)ex64 : BOX_EX;      // cells of the rows anybody will look at
        bool fits = ex64 <= BOX_EX && ey64 <= RCAP && ez64 <= RCAP && ey64 * ez64 <= RCAP;
        if (fits) {
            int32_t nsy = ((oy + ey - 1) >> NM_SBY_BITS) - (oy >> NM_SBY_BITS) + 1;
            int32_t nsz = ((oz + ez - 1) >> NM_SBZ_BITS) - (oz >> NM_SBZ_BITS) + 1;
            fits = 3 * nsy * nsz <= SCAP;
        }
        if (!fits) {
            // (readlane, not a shuffle: the box stays in scalar registers and so does everything derived
            // from it)
            const int anchor = __ffsll((long long)todo) - 1;
            const int32_t ax = __builtin_amdgcn_readlane(hx, anchor), ay = __builtin_amdgcn_readlane(hy, anchor),
                          az = __builtin_amdgcn_readlane(hz, anchor);
#if NM_ADAPTIVE_ANCHOR
            // the box around the anchor takes its shape from the pending lanes' own box: the thinner of its (y, z)
            // extents is kept whole if it is below the default, and the rows that saves widen the other axis -
            // a scanner's far ground (one or two cell layers in z, lanes metres apart in x and y) or a pole (a few
            // cells in y, many in z) gets a flat or a tall box instead of 20 x 20 - and on every axis the window is
            // pushed inside the lanes' range, so that no row of it lies where no lane is.  (all of this is scalar)
            const bool thin_y = ey64 <= ez64;
            const int64_t thin64 = thin_y ? ey64 : ez64, wide64 = thin_y ? ez64 : ey64;
            int32_t thin = thin64 < ANCHOR ? (int32_t)thin64 : ANCHOR;
            int32_t wide = RCAP / thin;
            if (wide64 < wide) wide = (int32_t)wide64;
            if (wide > ROWS_CAP / W) wide = ROWS_CAP / W;
            if (3 * ((thin + 6) / 8 + 1) * ((wide + 6) / 8 + 1) > SCAP) {
                thin = ANCHOR;
                wide = ANCHOR;
            }
            ey = thin_y ? thin : wide;
            ez = thin_y ? wide : thin;
            ex = BOX_EX;
            const int32_t cx = ax + dmin - (BOX_EX - W) / 2, cy = ay + dmin - (ey - W) / 2,
                          cz = az + dmin - (ez - W) / 2;
            // (window [o, o + e) inside [lo + dmin, hi + dmax] where the range is at least as long)
            const int64_t hx_end = (int64_t)hix + dmax + 1, hy_end = (int64_t)hiy + dmax + 1,
                          hz_end = (int64_t)hiz + dmax + 1;
            ox = ex64 >= BOX_EX ? (int32_t)max((int64_t)lox + dmin, min((int64_t)cx, hx_end - BOX_EX)) : lox + dmin;
            oy = ey64 >= ey ? (int32_t)max((int64_t)loy + dmin, min((int64_t)cy, hy_end - ey)) : loy + dmin;
            oz = ez64 >= ez ? (int32_t)max((int64_t)loz + dmin, min((int64_t)cz, hz_end - ez)) : loz + dmin;
#else
            ox = ax + dmin - (BOX_EX - W) / 2;
            oy = ay + dmin - (ANCHOR - W) / 2;
            oz = az + dmin - (ANCHOR - W) / 2;
            ey = ANCHOR;
            ez = ANCHOR;
            ex = BOX_EX;
#endif
        }
        ox = __builtin_amdgcn_readfirstlane(ox);
        oy = __builtin_amdgcn_readfirstlane(oy);
        oz = __builtin_amdgcn_readfirstlane(oz);
        ey = __builtin_amdgcn_readfirstlane(ey);
        ez = __builtin_amdgcn_readfirstlane(ez);
        ex = __builtin_amdgcn_readfirstlane(ex);
        const bool sel = !done && hx + dmin >= ox && hx + dmax < ox + BOX_EX && hy + dmin >= oy &&
                         hy + dmax < oy + ey && hz + dmin >= oz && hz + dmax < oz + ez;

        // ---- stage: leaf numbers of the box's superblocks (integer divisions by wave-uniform small
        //      numbers are done with a 20-bit reciprocal: exact for operands below 2^10)
        // the staged 64-bit rows start two cells before the box: bit b of a row is cell ox - 2 + b, so a
        // lane's shifted row already is (occupancy << 2), a byte offset into the table
        const int32_t sbx0 = (ox - 2) >> NM_SBX_BITS, sby0 = oy >> NM_SBY_BITS, sbz0 = oz >> NM_SBZ_BITS;
        const int32_t nsy = ((oy + ey - 1) >> NM_SBY_BITS) - sby0 + 1;
        const int32_t nsz = ((oz + ez - 1) >> NM_SBZ_BITS) - sbz0 + 1;
        const int32_t nsb = __builtin_amdgcn_readfirstlane(3 * nsy * nsz);
        const uint32_t inv_nsy = NM_RECIP20.v[nsy];     // (two scalar loads; the divisions cost 50 vector
        const uint32_t inv_ey = NM_RECIP20.v[ey];       //  instructions per pass, eight of them quarter rate)
#pragma nounroll
        for (int32_t t = lane; t < nsb; t += 64) {
            const uint32_t t3 = __umul24((uint32_t)t, 0x5556u) >> 16;  // t / 3 for t < 2^15
            const int32_t ix = t - 3 * (int32_t)t3;
            const uint32_t iz = __umul24(t3, inv_nsy) >> 20;           // t3 / nsy
            const int32_t iy = (int32_t)t3 - __mul24((int32_t)iz, nsy);
            int32_t sx = sbx0 + ix, sy = sby0 + iy, sz = sbz0 + (int32_t)iz;
            bool ok = sx >= 0 && sy >= 0 && sz >= 0 && sx < (1 << L.bx) && sy < (1 << L.by) &&
                      sz < (1 << L.bz);
            sbt[t] = ok ? nm_hash_find(I, nm_sb_key((uint32_t)sx, (uint32_t)sy, (uint32_t)sz, L))
                        : -1;
        }
        lds_fence();
        // ---- stage: 64-bit x-rows of the box, funnel-shifted out of the 32-bit leaf words
        const int32_t nrows = ey * ez;
        const uint32_t sh = (uint32_t)((ox - 2) & 31);
        // bit b of a row is cell ox - 2 + b and the selected lanes read bits 2 .. ex + 1: where those end inside
        // the second leaf word - a wave's 64 queries rarely span more than 40 cells - the third is not fetched
        const bool third = sh + (uint32_t)ex > 62u;
        // (the index's pointers come out of the scale array in memory; told that they are global ones the compiler
        // fetches with global_load.  leaves within 4 GB of the array's start - all but the hash form of clouds beyond
        // 16 M points - are addressed as base + 32-bit offset: one instruction per address instead of five of 64-bit
        // arithmetic.  a missing leaf reads leaf 0 and is masked out: no branch around any of the loads)
        const NmGlobalU32 leaf_g = (NmGlobalU32)I.leaf;
        const bool near4g = I.leaf_capacity <= (1u << 24);
        auto stage_rows = [&](auto near_tag) {
            constexpr bool NEAR = decltype(near_tag)::value;
#pragma nounroll
            for (int32_t rr = lane; rr < nrows; rr += 64) {
                const int32_t rz = (int32_t)(__umul24((uint32_t)rr, inv_ey) >> 20);   // rr / ey (rr < 512)
                const int32_t y = oy + (rr - __mul24(rz, ey)), z = oz + rz;
                const int32_t c0 = nm_mad24i((z >> NM_SBZ_BITS) - sbz0, nsy, (y >> NM_SBY_BITS) - sby0);
                const int32_t t0 = __mul24(c0, 3);
                const uint32_t wofs = (uint32_t)((z & 7) * 8 + (y & 7));
                const int32_t l0 = sbt[t0], l1 = sbt[t0 + 1], l2 = third ? sbt[t0 + 2] : -1;
                uint32_t w0, w1, w2 = 0u;
                if constexpr (NEAR) {
                    const __attribute__((address_space(1))) char* lb = (const __attribute__((address_space(1))) char*)leaf_g;
                    w0 = *(NmGlobalU32)(lb + (((uint32_t)max(l0, 0) << 8) + wofs * 4u));
                    w1 = *(NmGlobalU32)(lb + (((uint32_t)max(l1, 0) << 8) + wofs * 4u));
                    if (third) w2 = *(NmGlobalU32)(lb + (((uint32_t)max(l2, 0) << 8) + wofs * 4u));
                } else {
                    w0 = leaf_g[((uint64_t)(uint32_t)max(l0, 0) << 6) + wofs];
                    w1 = leaf_g[((uint64_t)(uint32_t)max(l1, 0) << 6) + wofs];
                    if (third) w2 = leaf_g[((uint64_t)(uint32_t)max(l2, 0) << 6) + wofs];
                }
                w0 &= ~(uint32_t)(l0 >> 31);
                w1 &= ~(uint32_t)(l1 >> 31);
                w2 &= ~(uint32_t)(l2 >> 31);
                const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh);
                const uint32_t hi = __builtin_amdgcn_alignbit(w2, w1, sh);
                rows[rr] = (uint64_t)lo | ((uint64_t)hi << 32);
                if constexpr (XREFL)      // bit b of the reversed row is cell ox + 61 - b
                    ((uint64_t*)(lds_raw + REV_OFS))[rr] = (uint64_t)__brev(hi) | ((uint64_t)__brev(lo) << 32);
            }
        };
        if (near4g) stage_rows(std::true_type{});
        else stage_rows(std::false_type{});
        lds_fence();

        // ---- phase B: walk the W*W rows of every selected lane.  branch-free: W row reads, then W
        //      table reads, per j; rows nobody occupies add the table's zero entry.
        if constexpr (SLAB) {
        if (sel) {
            // ---- wide window: per y-slab j the W row masks (phase A) and at once the walk of its W rows
            //      (phase B); nothing but the per-k sums and the running moments lives from slab to slab.
            //      (what a lane brings in here is invariant in the pass loop, and what is derived from it
            //      would be hoisted out of the loop and kept in registers through all of it: opaque copies)
            double px = qx, py = qy, pz = qz;
            int32_t gx = hx, gy = hy, gz = hz, sy_ = sgn_y, sz_ = sgn_z, tx_ = tx, ty_ = ty, tz_ = tz;
            asm volatile("" : "+v"(px), "+v"(py), "+v"(pz), "+v"(gx), "+v"(gy), "+v"(gz), "+v"(sy_), "+v"(sz_),
                              "+v"(tx_), "+v"(ty_), "+v"(tz_));
            double dx2[W], dz2[W];
#pragma unroll
            for (int i = 0; i < W; ++i) {
                double d = px - (tab ? ctab[tx_ + i] : nm_centre(gx + dmin + i, L.min_x, L.edge, L.half_edge));
                dx2[i] = d * d;
                d = pz - (tab ? ctab[128 + tz_ + sz_ * (i - C)]
                              : nm_centre(gz + sz_ * (i - C), L.min_z, L.edge, L.half_edge));
                dz2[i] = d * d;
            }
            const int32_t rx = gx + dmin - ox;
            const int32_t rhome8 = (__mul24(gz - oz, ey) + (gy - oy)) << 3;
            const int32_t step_z8 = (sz_ < 0 ? -ey : ey) << 3, step_y8 = sy_ << 3;
            const unsigned char* rows8 = (const unsigned char*)rows;
            uint32_t bk[W];
#pragma unroll
            for (int i = 0; i < W; ++i) bk[i] = 0u;
            uint32_t n = 0, sx = 0, sxx = 0, sy = 0, syy = 0, sxy = 0, syz = 0;
#pragma unroll
            for (int j = 0; j < W; ++j) {
                double dyj = py - (tab ? ctab[64 + ty_ + sy_ * (j - C)]
                                       : nm_centre(gy + sy_ * (j - C), L.min_y, L.edge, L.half_edge));
                dyj = dyj * dyj;
                double pxy[W];
#pragma unroll
                for (int i = 0; i < W; ++i) pxy[i] = dx2[i] + dyj;
                uint32_t m[RPJ];
#pragma unroll
                for (int h = 0; h < RPJ; ++h) m[h] = 0u;
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    RowBound rb = RT.rb[j * W + k];
                    if constexpr (RHO > 0) rb = nm_static_bounds<RHO>().rb[j * W + k];
                    if (rb.b < 0) continue;
                    uint32_t outside = 0u;
#pragma unroll
                    for (int i = W - 1; i >= 0; --i) {
                        const int ad = i > C ? i - C : C - i;
                        if (ad > rb.b) continue;
                        if (ad <= rb.a) {
                            outside = outside << 1;
                        } else {
                            const double sq = pxy[i] + dz2[k];
                            const double t = r2 - sq;      // sign bit set  <=>  sq > r^2  (exact)
                            outside = __builtin_amdgcn_alignbit(outside, (uint32_t)__double2hiint(t), 31);
                        }
                    }
                    const uint32_t span = (1u << (2 * rb.b + 1)) - 1u;
                    m[k / RPR] |= ((~outside) & span) << ((k % RPR) * W + 2 + (C - rb.b));
                }
                uint32_t ajj = 0u, cjj = 0u;
                uint32_t valid[W];
                const int32_t base_j = rhome8 + __mul24(j - C, step_y8);
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    if constexpr (RHO > 0)
                        if (nm_static_bounds<RHO>().rb[j * W + k].b < 0) continue;
                    const uint64_t row = *(const uint64_t*)(rows8 + (base_j + __mul24(k - C, step_z8)));
                    const uint32_t in4 = (m[k / RPR] >> ((k % RPR) * W)) & (((1u << W) - 1u) << 2);
                    valid[k] = (uint32_t)(row >> rx) & in4;      // 4 * (occupied & inside)
                }
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    if constexpr (RHO > 0)
                        if (nm_static_bounds<RHO>().rb[j * W + k].b < 0) continue;
                    const uint32_t t = lut[valid[k] >> 2];
                    ajj += t;
                    bk[k] += t;
                    cjj += (t & 0xFFu) * (uint32_t)k;
                }
                const uint32_t na = ajj & 0xFFu, xa = (ajj >> 8) & 0xFFFu, xxa = ajj >> 20;
                n += na;
                sx += xa;
                sxx += xxa;
                sy += na * j;
                syy += na * (j * j);
                sxy += xa * j;
                syz += cjj * j;
                __builtin_amdgcn_sched_barrier(0);
            }
            uint32_t sz = 0, szz = 0, sxz = 0;
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const uint32_t nb = bk[i] & 0xFFu, xb = (bk[i] >> 8) & 0xFFFu;
                sz += nb * i;
                szz += nb * (i * i);
                sxz += xb * i;
            }
            m_n = n; m_sx = sx; m_sy = sy; m_sz = sz; m_sxx = sxx; m_sxy = sxy; m_sxz = sxz;
            m_syy = syy; m_syz = syz; m_szz = szz;
            done = true;
        }
        } else
        if (sel) {
            // (mirrored lane: window index i is cell hx - dmin - i, bit ox + 61 - hx + dmin + i of the reversed
            // row; it has to arrive on bit 2 + i like the others.  hx + dmax < ox + 60 keeps the shift >= 0)
            const int32_t rx = XREFL && sgn_x < 0 ? ox + 59 + dmin - hx : hx + dmin - ox;
            // row of the home cell, and the lane's signed strides through the mirrored window
            // (byte offsets; 24-bit multiplies are full rate, 32-bit ones a quarter)
            const int32_t rhome8 = ((__mul24(hz - oz, ey) + (hy - oy)) << 3) + (XREFL && sgn_x < 0 ? REV_OFS : 0);
            const int32_t step_z8 = (sgn_z < 0 ? -ey : ey) << 3, step_y8 = sgn_y << 3;
            const unsigned char* rows8 = (const unsigned char*)rows;
            uint32_t aj[W], bk[W], cj[NM_DIAG_SUMS ? 2 * W - 1 : W];
#pragma unroll
            for (int i = 0; i < W; ++i) aj[i] = bk[i] = 0u;
#pragma unroll
            for (int i = 0; i < (NM_DIAG_SUMS ? 2 * W - 1 : W); ++i) cj[i] = 0u;
            // the packed masks are loop-invariant; without this the compiler unpacks all W*W fields
            // ahead of the pass loop and keeps them in W*W registers
#pragma unroll
            for (int r = 0; r < MASK_REGS; ++r) asm volatile("" : "+v"(inside[r]));
#pragma unroll
            for (int j = 0; j < W; ++j) {
                // rows of the reflected window that can never hold a neighbor are skipped at compile time
                // for r = 3e (14 of the 49); with the run-time table they simply look up an empty mask
                uint32_t valid[W];
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    if constexpr (RHO > 0)
                        if (nm_static_bounds<RHO>().rb[j * W + k].b < 0) continue;
                    const uint64_t row = *(const uint64_t*)(
                        rows8 + (__mul24(k - C, step_z8) + (rhome8 + __mul24(j - C, step_y8))));
                    const uint32_t in4 = (inside[j * RPJ + k / RPR] >> ((k % RPR) * W)) &
                                         (((1u << W) - 1u) << 2);
                    valid[k] = (uint32_t)(row >> rx) & in4;      // 4 * (occupied & inside)
                }
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    if constexpr (RHO > 0)
                        if (nm_static_bounds<RHO>().rb[j * W + k].b < 0) continue;
                    const uint32_t t = lut[valid[k] >> 2];
                    aj[j] += t;
                    bk[k] += t;
                    if (NM_DIAG_SUMS)      // per-(j+k) sums: sum (j+k)^2 n = Syy + 2 Syz + Szz gives sum j*k n
                        cj[j + k] += t;    // with additions only (a multiply-add costs two issue slots more)
                    else if (W <= 7)  // low 8 bits = sum k*n (< 256 for W <= 7); a 24-bit multiply keeps them
                        cj[j] = nm_mad24(t, (uint32_t)k, cj[j]);
                    else
                        cj[j] += (t & 0xFFu) * (uint32_t)k;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            uint32_t n = 0, sx = 0, sxx = 0, sy = 0, syy = 0, sxy = 0, sz = 0, szz = 0, sxz = 0,
                     syz = 0;
#pragma unroll
            for (int i = 0; i < W; ++i) {
                uint32_t na = aj[i] & 0xFFu, xa = (aj[i] >> 8) & 0xFFFu, xxa = aj[i] >> 20;
                uint32_t nb = bk[i] & 0xFFu, xb = (bk[i] >> 8) & 0xFFFu;
                uint32_t cji = NM_DIAG_SUMS ? 0u : (W <= 7 ? (cj[i] & 0xFFu) : cj[i]);
                n += na;
                sx += xa;
                sxx += xxa;
                sy += na * i;
                syy += na * (i * i);
                sxy += xa * i;
                sz += nb * i;
                szz += nb * (i * i);
                sxz += xb * i;
                syz += cji * i;
            }
            if (NM_DIAG_SUMS) {
                uint32_t sdd = 0;
#pragma unroll
                for (int i = 1; i < 2 * W - 1; ++i) sdd += (cj[i] & 0xFFu) * (uint32_t)(i * i);
                syz = (sdd - syy - szz) >> 1;
            }
            // the ten integer moments wait in registers; the eigen-solve runs once, after the last
            // pass, for all lanes together (an extra pass costs staging + this row walk only)
            m_n = n; m_sx = sx; m_sy = sy; m_sz = sz; m_sxx = sxx; m_sxy = sxy; m_sxz = sxz;
            m_syy = syy; m_syz = syz; m_szz = szz;
            done = true;
        }
        lds_fence();
    }
    if (passes > 1 && lane == 0) atomicAdd(&S->stats[9], passes - 1);

    // ---- epilogue: features from the integer moments (features.py:21-57), one query per lane
    const bool emit = have && !far;
    if (emit) {
        double out[4];
        // the moments are in the mirrored frame; so must be the query's offset from its home centre
        double ux = qx - nm_centre(hx, L.min_x, L.edge, L.half_edge);
        if constexpr (XREFL) ux = fabs(ux);
        const double uy = fabs(uy_home);
        const double uz = fabs(uz_home);
        nm_features_from_moments((double)m_n, (double)m_sx, (double)m_sy, (double)m_sz,
                                 (double)m_sxx, (double)m_sxy, (double)m_sxz, (double)m_syy,
                                 (double)m_syz, (double)m_szz, ux, uy, uz, (double)dmin, L.edge, out);
        double* o = nm_row_ptr(A.feat, qi, A.fstride, 4 * s);
        o[0] = out[0];
        o[1] = out[1];
        o[2] = out[2];
        o[3] = out[3];
        if (A.cov)
            nm_covariance_from_moments((double)m_n, (double)m_sx, (double)m_sy, (double)m_sz,
                                       (double)m_sxx, (double)m_sxy, (double)m_sxz, (double)m_syy,
                                       (double)m_syz, (double)m_szz, (double)sgn_x, (double)sgn_y, (double)sgn_z,
                                       L.edge, nm_row_ptr(A.cov, qi, A.cstride, 6 * s));
        if (A.normal)
            nm_normal_from_moments((double)m_n, (double)m_sx, (double)m_sy, (double)m_sz, (double)m_sxx,
                                   (double)m_sxy, (double)m_sxz, (double)m_syy, (double)m_syz,
                                   (double)m_szz, (double)sgn_x, (double)sgn_y, (double)sgn_z,
                                   nm_row_ptr(A.normal, qi, A.nstride, 3 * s));
    }
    const unsigned long long degenerate = __ballot(emit && m_n < 2u);
    if (degenerate && lane == (__ffsll((long long)degenerate) - 1))
        atomicAdd(&S->stats[8], (uint32_t)__popcll(degenerate));
    if (A.sparse) {
        const unsigned long long sparse = __ballot(have && (far || m_n < (uint32_t)A.sparse_k));
        if (lane == 0) A.sparse[(int64_t)s * A.sparse_words + batch] = sparse;
    }
    }   // scale loop

    if (FOREST) {
        const int64_t slot = batch * 64 + lane;
        bool have = slot < A.n_slots;
        uint32_t qi = 0;
        if (have) {
            qi = A.order[slot];
            have = qi < A.nq;
        }
        lds_fence();                         // the last pass has read the row buffer
        nm_forest_epilogue(A, forest_nodes, (float*)lds_raw, lane, have, qi);
    }
}

// ---- generic kernel: any W, direct index lookups per lane (no staging).  slow path for unusual
//      radius/edge ratios; same arithmetic.  ONLY_UNPRUNED: the companion of the table kernels, a small
//      persistent grid that handles exactly the scales they skip (lattices whose coordinates are too large
//      for the static window bounds) and otherwise leaves at once.
template <bool ONLY_UNPRUNED>
__global__ __launch_bounds__(64) void k_scale_features_generic(ScaleArgs A)
{
    if (ONLY_UNPRUNED) {
        bool any = false;
        for (int32_t s = A.s_begin; s < A.s_end; ++s)
            any = any || (A.scales[s].valid && !A.scales[s].prune_ok);
        if (!any) return;
    }
    const int64_t n_batches = (A.n_slots + 63) / 64;
    for (int64_t batch = blockIdx.x; batch < n_batches; batch += gridDim.x) {
        const int64_t slot = batch * 64 + threadIdx.x;
        bool have = slot < A.n_slots;
        uint32_t qi = 0;
        if (have) {
            qi = A.order[slot];
            have = qi < A.nq;
        }
        double qx = 0.0, qy = 0.0, qz = 0.0;
        if (have) {
            const double* p = A.query + (A.direct ? slot : (int64_t)qi) * A.qstride;
            qx = p[0];
            qy = p[1];
            qz = p[2];
        }
        for (int32_t s = A.s_begin; s < A.s_end; ++s) {
            const ScaleDev* __restrict__ S = A.scales + s;
            if (!S->valid) {
                if (!ONLY_UNPRUNED && have) {      // (with a table kernel behind it, that kernel writes the NaNs)
                    double* o = A.feat + (int64_t)qi * A.fstride + 4 * s;
                    o[0] = o[1] = o[2] = o[3] = __builtin_nan("");
                }
                continue;
            }
            if (ONLY_UNPRUNED && S->prune_ok) continue;
            bool sp = false;
            if (have) nm_lane_generic(A, S->L, S->I, S->stats, S->r2, s, qi, qx, qy, qz, &sp);
            if (A.sparse) {
                const unsigned long long sparse = __ballot(have && sp);
                if (threadIdx.x == 0) A.sparse[(int64_t)s * A.sparse_words + batch] = sparse;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------

__global__ void k_publish_info_all(const ScaleDev* __restrict__ ladder, int32_t n, int64_t* info)
{
    const int i = threadIdx.x;
    if (i < n) {
        const uint32_t* counters = ladder[i].I.counters;      // of the index (a borrowed one: its owner's)
        const uint32_t* stats = ladder[i].stats;              // of this scale
        // M (-1: the index build timed out or overflowed, or the lattice is not addressable)
        const bool bad = !ladder[i].valid || counters[3] || counters[2];
        info[4 * i + 0] = bad ? -1 : (int64_t)counters[1];
        info[4 * i + 1] = stats[8];      // neighborhoods with population < 2
        info[4 * i + 2] = stats[9];      // extra passes of the search kernel
        info[4 * i + 3] = ladder[i].I.hash ? counters[0] : counters[4];   // leaves that hold a voxel
    }
}

// ---- k-nearest-voxel fallback -------------------------------------------------------------------------
// queries whose radius neighborhood has fewer than k voxels are re-evaluated on their k nearest occupied
// voxel centres within rk (rk2 = rk*rk).  one lane per query; the lane walks the index leaves that
// intersect the cube around its home cell that contains the ball of radius rk.  exact fp64 distances in
// the reference's operation order, ties broken by the smaller voxel address.

struct KnnArgs {
    ScaleArgs S;
    int32_t scale;             // which scale of S.scales
    int32_t k;
    double rk2;
    int32_t max_shell;
    const uint32_t* list;      // slots whose population is below k, compacted
    const uint32_t* count;
};

// bits -> list of slots.  one counter atomic per block of 256 words (16 k slots)
__global__ __launch_bounds__(256) void k_knn_compact(const unsigned long long* __restrict__ mask,
                                                     int64_t n_words, uint32_t* __restrict__ list,
                                                     uint32_t* __restrict__ count)
{
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t base;
    const int64_t wi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long m = wi < n_words ? mask[wi] : 0ull;
    const uint32_t c = (uint32_t)__popcll(m);
    // inclusive scan inside the wave, then across the block's four waves
    uint32_t incl = c;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < w) before += wsum[i];
        total += wsum[i];
    }
    if (threadIdx.x == 0) base = total ? atomicAdd(count, total) : 0u;
    __syncthreads();
    uint32_t at = base + before + incl - c;
    while (m) {
        const int bit = __ffsll((long long)m) - 1;
        m &= m - 1ull;
        list[at++] = (uint32_t)(wi * 64 + bit);
    }
}

// KMAX: slots of the per-lane best list (8 serves k <= 8: half the insertion network and half its registers).
// CodeT: the tie-break key - the candidate's cell offsets from the home cell, biased, z high / x low (the
// order of the reference's voxel address) - 32 bits while the cube stays within +-511 cells, else 64.
template <int KMAX, typename CodeT>
__global__ __launch_bounds__(64) void k_knn_fallback(KnnArgs K)
{
    constexpr int BITS = sizeof(CodeT) == 4 ? 10 : 11;
    constexpr int32_t BIAS = 1 << (BITS - 1);
    constexpr CodeT CODE_MAX = sizeof(CodeT) == 4 ? (CodeT)0x7FFFFFFF : (CodeT)INT64_MAX;
    const ScaleArgs& A = K.S;
    const ScaleDev* __restrict__ SD = A.scales + K.scale;
    if (!SD->valid) return;
    const LatticeDev L = SD->L;
    const IndexDev I = SD->I;
    // the launch covers every slot; the waves beyond the list leave at once
    const int64_t idx = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (idx >= (int64_t)*K.count) return;
    const int64_t slot = K.list[idx];
    const uint32_t qi = A.order[slot];
    double* o = A.feat + (int64_t)qi * A.fstride + 4 * K.scale;
    const double* p = A.query + (A.direct ? slot : (int64_t)qi) * A.qstride;
    const double qx = p[0], qy = p[1], qz = p[2];
    const int32_t hx = nm_clamp_cell(nm_cell_f(qx, L.min_x, L.edge));
    const int32_t hy = nm_clamp_cell(nm_cell_f(qy, L.min_y, L.edge));
    const int32_t hz = nm_clamp_cell(nm_cell_f(qz, L.min_z, L.edge));

    double bd[KMAX];
    CodeT bc[KMAX];              // address-like code of the voxel: tie break and offsets
    double kth_best = INFINITY;  // bd[k-1]: a candidate beyond it cannot be among the k nearest
    int32_t found = 0;
    // two stages: a small cube first - most sparse neighborhoods find their k voxels just outside the
    // radius - and the full cube only when the k-th best is not yet provably final (every cell outside
    // a cube of half-width S is at least S + 1/2 cells away).
    for (int stage = 0; stage < 2; ++stage) {
#pragma unroll
    for (int t = 0; t < KMAX; ++t) {
        bd[t] = INFINITY;
        bc[t] = CODE_MAX;
    }
    found = 0;
    kth_best = INFINITY;
    // walk the leaves that intersect the cube of half-width S around the home cell: one hash lookup per
    // leaf, then its 64 row words; only occupied cells cost a distance.  sparse neighborhoods (the only
    // ones that get here) touch few leaves.
    const int32_t S_small = (-A.dmin) + 2;
    const int32_t S = (stage == 0 && S_small < K.max_shell) ? S_small : K.max_shell;
    const int32_t x_lo = max(hx - S, 0), x_hi = min(hx + S, (1 << L.wx) - 1);
    const int32_t y_lo = max(hy - S, 0), y_hi = min(hy + S, (1 << L.wy) - 1);
    const int32_t z_lo = max(hz - S, 0), z_hi = min(hz + S, (1 << L.wz) - 1);
    for (int32_t sbz = z_lo >> NM_SBZ_BITS; sbz <= (z_hi >> NM_SBZ_BITS); ++sbz)
        for (int32_t sby = y_lo >> NM_SBY_BITS; sby <= (y_hi >> NM_SBY_BITS); ++sby)
            for (int32_t sbx = x_lo >> NM_SBX_BITS; sbx <= (x_hi >> NM_SBX_BITS); ++sbx) {
                const int32_t leaf = nm_hash_find(
                    I, nm_sb_key((uint32_t)sbx, (uint32_t)sby, (uint32_t)sbz, L));
                if (leaf < 0) continue;
                // bits of this leaf's 32 x-cells that lie inside [x_lo, x_hi]
                const int32_t bx0 = sbx << NM_SBX_BITS;
                const int32_t lo_bit = max(x_lo - bx0, 0), hi_bit = min(x_hi - bx0, 31);
                const uint32_t xmask = (hi_bit >= 31 ? ~0u : ((2u << hi_bit) - 1u)) & (~0u << lo_bit);
                for (int32_t lz = 0; lz < 8; ++lz) {
                    const int32_t gz = (sbz << NM_SBZ_BITS) + lz;
                    if (gz < z_lo || gz > z_hi) continue;
                    double d = qz - nm_centre(gz, L.min_z, L.edge, L.half_edge);
                    const double dz2 = d * d;
                    // the eight row words of this z layer in two 16-byte loads
                    const uint4 w0 = *(const uint4*)(I.leaf + (size_t)leaf * NM_LEAF_WORDS + lz * 8);
                    const uint4 w1 = *(const uint4*)(I.leaf + (size_t)leaf * NM_LEAF_WORDS + lz * 8 + 4);
                    const uint32_t words[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                    for (int32_t ly = 0; ly < 8; ++ly) {
                        const int32_t gy = (sby << NM_SBY_BITS) + ly;
                        uint32_t word = words[ly] & xmask;
                        if (gy < y_lo || gy > y_hi || !word) continue;
                        d = qy - nm_centre(gy, L.min_y, L.edge, L.half_edge);
                        const double dy2 = d * d;
                        while (word) {
                            const int32_t bit = __ffs((int)word) - 1;
                            word &= word - 1u;
                            const int32_t gx = bx0 + bit;
                            d = qx - nm_centre(gx, L.min_x, L.edge, L.half_edge);
                            double cd = (d * d + dy2) + dz2;
                            if (!(cd <= K.rk2)) continue;
                            ++found;
                            if (cd > kth_best) continue;      // cannot be among the k nearest
                            // offsets from the home cell, biased, z high / x low: the same order as
                            // the reference's voxel address, which is the tie break
                            CodeT cc = ((CodeT)(gz - hz + BIAS) << (2 * BITS)) |
                                       ((CodeT)(gy - hy + BIAS) << BITS) | (CodeT)(gx - hx + BIAS);
#pragma unroll
                            for (int t = 0; t < KMAX; ++t) {
                                const bool before = cd < bd[t] || (cd == bd[t] && cc < bc[t]);
                                const double td = before ? bd[t] : cd;
                                const CodeT tc = before ? bc[t] : cc;
                                bd[t] = before ? cd : bd[t];
                                bc[t] = before ? cc : bc[t];
                                cd = td;
                                cc = tc;
                            }
#pragma unroll
                            for (int t = 0; t < KMAX; ++t)
                                if (t == K.k - 1) kth_best = bd[t];
                        }
                    }
                }
            }
    double kth = INFINITY;
#pragma unroll
    for (int t = 0; t < KMAX; ++t)
        if (t == K.k - 1) kth = bd[t];
    const double reach = ((double)S + 0.5 - 1e-6) * L.edge;
    if (S >= K.max_shell || kth <= reach * reach) break;
    }   // stage
    const int32_t use = found < K.k ? found : K.k;
    double n = 0, sx = 0, sy = 0, sz = 0, sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
#pragma unroll
    for (int t = 0; t < KMAX; ++t) {
        if (t < use) {
            const double ox = (double)((int32_t)(bc[t] & ((1 << BITS) - 1)) - BIAS);
            const double oy = (double)((int32_t)((bc[t] >> BITS) & ((1 << BITS) - 1)) - BIAS);
            const double oz = (double)((int32_t)(bc[t] >> (2 * BITS)) - BIAS);
            n += 1.0;
            sx += ox; sy += oy; sz += oz;
            sxx += ox * ox; sxy += ox * oy; sxz += ox * oz;
            syy += oy * oy; syz += oy * oz; szz += oz * oz;
        }
    }
    double out[4];
    const double ux = qx - nm_centre(hx, L.min_x, L.edge, L.half_edge);
    const double uy = qy - nm_centre(hy, L.min_y, L.edge, L.half_edge);
    const double uz = qz - nm_centre(hz, L.min_z, L.edge, L.half_edge);
    nm_features_from_moments(n, sx, sy, sz, sxx, sxy, sxz, syy, syz, szz, ux, uy, uz, 0.0, L.edge, out);
    o[1] = out[1];
    o[2] = out[2];
    o[3] = out[3];
    if (A.cov)
        nm_covariance_from_moments(n, sx, sy, sz, sxx, sxy, sxz, syy, syz, szz, 1.0, 1.0, 1.0, L.edge,
                                   A.cov + (int64_t)qi * A.cstride + 6 * K.scale);
    if (A.normal)
        nm_normal_from_moments(n, sx, sy, sz, sxx, sxy, sxz, syy, syz, szz, 1.0, 1.0, 1.0,
                               A.normal + (int64_t)qi * A.nstride + 3 * K.scale);
}

// workspace of the fallback: the sparse bits (one mask per scale), the compacted slots, their count
struct KnnLayout {
    size_t mask, list, count;
    int64_t words;           // 64-bit words of one scale's mask
};

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

static void knn_layout(int64_t n_slots, int n_scales, size_t* off, KnnLayout* Kl)
{
    auto take = [&](size_t bytes) {
        size_t at = *off;
        *off += align_up(bytes);
        return at;
    };
    Kl->words = (n_slots + 63) / 64;
    Kl->mask = take((size_t)Kl->words * 8 * (size_t)n_scales);
    Kl->list = take((size_t)n_slots * 4);
    Kl->count = take(256);
}

// re-evaluates the sparse rows of scale `scale` (their bits were left by the search kernel)
static int launch_knn_fallback(nm_ctx* ctx, const ScaleArgs& A, int scale, double radius, double edge,
                               uint32_t* list, uint32_t* count, hipStream_t s)
{
    if (ctx->knn_k <= 0 || A.nq <= 0) return NM_OK;
    const int64_t n_words = (A.n_slots + 63) / 64;
    NM_HIP(ctx, hipMemsetAsync(count, 0, 4, s));
    k_knn_compact<<<(int)((n_words + 255) / 256), 256, 0, s>>>(A.sparse + (int64_t)scale * A.sparse_words,
                                                               n_words, list, count);
    KnnArgs K;
    K.S = A;
    K.scale = scale;
    K.list = list;
    K.count = count;
    K.k = ctx->knn_k;
    const double rk = radius * ctx->knn_radius_factor;
    K.rk2 = rk * rk;
    double shells = ceil(rk / edge + 0.5);
    if (shells > 1000.0) shells = 1000.0;
    K.max_shell = (int32_t)shells;
    const int blocks = (int)((A.n_slots + 63) / 64);
    const bool small_codes = K.max_shell <= 500;        // +-511 cells fit the 10-bit fields
    if (K.k <= 8) {
        if (small_codes) k_knn_fallback<8, int32_t><<<blocks, 64, 0, s>>>(K);
        else k_knn_fallback<8, int64_t><<<blocks, 64, 0, s>>>(K);
    } else {
        if (small_codes) k_knn_fallback<16, int32_t><<<blocks, 64, 0, s>>>(K);
        else k_knn_fallback<16, int64_t><<<blocks, 64, 0, s>>>(K);
    }
    return NM_OK;
}

// candidates per axis: all integer offsets d with |d + 1/2 - f| <= r/e for some f in [0,1), with a
// guard for rounding in r/e.  a superset is harmless: every candidate is tested exactly.
static int candidate_width(double radius, double edge, int32_t* dmin)
{
    double rho = radius / edge;
    double m = floor(rho + 0.5 + 1e-9);
    if (!(m >= 0.0) || m > 1000.0) return -1;
    *dmin = -(int32_t)m;
    return 2 * (int32_t)m + 1;
}

// picks the kernel instance for (W, r/e) and launches it for the scales [A.s_begin, A.s_end), which all
// share that window; `forest`: evaluate A.F behind the last of them (table kernels only)
template <bool FOREST, bool LOOP>
static void launch_table_kernel(const ScaleArgs& A, double rho, int W, int blocks, hipStream_t s)
{
    const double rho2 = rho * rho;
    // whether pruning is sound for a lattice is decided on the device (ScaleDev::prune_ok): the tables
    // here are the pruned ones, lattices that cannot use them take the per-lane path inside the kernel
    const bool whole = fabs(rho - floor(rho + 0.5)) < 1e-9;      // r is a whole number of edges
    switch (W) {
        case 3: k_scale_features<3, 0, FOREST, LOOP><<<blocks, 64, 0, s>>>(A, nm_make_bounds<3>(rho2, true), A.scales, A.F.nodes); break;
        case 5: k_scale_features<5, 0, FOREST, LOOP><<<blocks, 64, 0, s>>>(A, nm_make_bounds<5>(rho2, true), A.scales, A.F.nodes); break;
        case 7:
            if (whole) k_scale_features<7, 3, FOREST, LOOP><<<blocks, 64, 0, s>>>(A, NM_BOUNDS_RHO3, A.scales, A.F.nodes);
            else k_scale_features<7, 0, FOREST, LOOP><<<blocks, 64, 0, s>>>(A, nm_make_bounds<7>(rho2, true), A.scales, A.F.nodes);
            break;
        case 9:
            if (whole) k_scale_features<9, 4, FOREST, LOOP><<<blocks, 64, 0, s>>>(A, NM_BOUNDS_RHO4, A.scales, A.F.nodes);
            else k_scale_features<9, 0, FOREST, LOOP><<<blocks, 64, 0, s>>>(A, nm_make_bounds<9>(rho2, true), A.scales, A.F.nodes);
            break;
        default:
            if (whole) k_scale_features<11, 5, FOREST, LOOP><<<blocks, 64, 0, s>>>(A, NM_BOUNDS_RHO5, A.scales, A.F.nodes);
            else k_scale_features<11, 0, FOREST, LOOP><<<blocks, 64, 0, s>>>(A, nm_make_bounds<11>(rho2, true), A.scales, A.F.nodes);
            break;
    }
}

// returns true when the forest (if asked for) was evaluated by the launch
static bool launch_scale_kernel(const ScaleArgs& A, double radius, double edge, int W, bool forest,
                                hipStream_t s)
{
    // one workgroup per 64-query batch: the hardware dispatcher balances the uneven batches (waves
    // that need several passes) better than a persistent grid did (measured: persistent -17 %)
    const int blocks = (int)((A.n_slots + 63) / 64);
    const double rho = radius / edge;
    if (W == 3 || W == 5 || W == 7 || W == 9 || W == 11) {
        // the scales the table kernel has to skip (decided on the device) first: the classifier in the
        // table kernel's epilogue needs every column of the row
        k_scale_features_generic<true><<<blocks < 2048 ? blocks : 2048, 64, 0, s>>>(A);
        const bool loop = A.s_end - A.s_begin > 1;
        // small clouds: a workgroup per (scale, batch) instead of a wave walking the scales - shorter workgroups, a
        // shorter tail (NM_SPLIT_SCALES_BELOW slots; above, keeping the queries in registers across the loop wins)
        const bool split = loop && !forest && A.n_slots <= NM_SPLIT_SCALES_BELOW;
        if (forest) launch_table_kernel<true, true>(A, rho, W, blocks, s);
        else if (split) launch_table_kernel<false, false>(A, rho, W, blocks * (A.s_end - A.s_begin), s);
        else if (loop) launch_table_kernel<false, true>(A, rho, W, blocks, s);
        else launch_table_kernel<false, false>(A, rho, W, blocks, s);
        return forest;
    }
    k_scale_features_generic<false><<<blocks, 64, 0, s>>>(A);
    return false;
}

static int check_scale_args(nm_ctx* ctx, const char* who, const double* d_query, int64_t n_query,
                            int64_t query_stride, const double* d_search, int64_t n_search,
                            int64_t search_stride, const double* d_feat, int64_t feat_stride,
                            const void* d_work)
{
    if (!d_search || n_search < 2 || search_stride < 3 || n_query < 0 || feat_stride < 4 || !d_work ||
        n_search >= ((int64_t)1 << 31) || n_query >= ((int64_t)1 << 31))
        NM_FAIL(ctx, NM_ERR_INVALID, "%s: bad arguments", who);
    if (n_query > 0 && (!d_query || !d_feat || query_stride < 3))
        NM_FAIL(ctx, NM_ERR_INVALID, "%s: bad query/feature arguments", who);
    // row strides are multiplied as 32-bit numbers on the device (one multiply-add per address)
    const int64_t lim = (int64_t)1 << 31;
    if (query_stride >= lim || search_stride >= lim || feat_stride >= lim || ctx->cov_stride >= lim ||
        ctx->normal_stride >= lim)
        NM_FAIL(ctx, NM_ERR_INVALID, "%s: a row stride of 2^31 elements or more", who);
    return NM_OK;
}

static void fill_outputs(nm_ctx* ctx, ScaleArgs* A)
{
    A->cov = ctx->cov_out;
    A->cstride = ctx->cov_stride;
    A->normal = ctx->normal_out;
    A->nstride = ctx->normal_stride;
    A->sparse_k = ctx->knn_k;
    A->F = ForestDev{};
}

// ---- one scale, self-contained (sorts at this scale) ----------------------------------------------------

struct ScaleLayout {
    size_t key_tmp, val_tmp, key_sorted, val_sorted;   // search cloud (and query cloud when shared)
    size_t qkey_tmp, qval_tmp, qkey_sorted, qval_sorted;
    size_t sort_temp, sort_temp_bytes;
    size_t index;
    size_t scale_dev;                                  // the one ScaleDev the search kernel reads
    IndexLayout ilay;
    KnnLayout knn;
    size_t total;
};

static void scale_layout(int64_t nq, int64_t ns, const LatticeDev& L, bool shared, ScaleLayout* S)
{
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t at = off;
        off += align_up(bytes);
        return at;
    };
    S->key_tmp = take((size_t)ns * 8);
    S->val_tmp = take((size_t)ns * 4);
    S->key_sorted = take((size_t)ns * 8);
    S->val_sorted = take((size_t)ns * 4);
    if (!shared) {
        S->qkey_tmp = take((size_t)nq * 8);
        S->qval_tmp = take((size_t)nq * 4);
        S->qkey_sorted = take((size_t)nq * 8);
        S->qval_sorted = take((size_t)nq * 4);
    } else {
        S->qkey_tmp = S->qval_tmp = S->qkey_sorted = S->qval_sorted = 0;
    }
    S->sort_temp_bytes = nm_sort_pairs_temp_bytes(ns > nq ? ns : nq);
    S->sort_temp = take(S->sort_temp_bytes);
    nm_index_layout(L, ns, &S->ilay);
    S->index = take(S->ilay.total);
    S->scale_dev = take(sizeof(ScaleDev));
    knn_layout(ns > nq ? ns : nq, 1, &off, &S->knn);
    S->total = off;
}

extern "C" size_t nm_scale_workspace_bytes(int64_t n_query, int64_t n_search, const nm_lattice* lat)
{
    if (!lat || n_query < 0 || n_search < 1) return 0;
    LatticeDev L = make_lattice_dev(lat);
    ScaleLayout S;
    // sized for the separate-cloud case so one workspace serves both
    scale_layout(n_query > 0 ? n_query : 1, n_search, L, false, &S);
    return S.total;
}

extern "C" int nm_scale_features(nm_ctx* ctx, const double* d_query, int64_t n_query,
                                 int64_t query_stride, const double* d_search, int64_t n_search,
                                 int64_t search_stride, const nm_lattice* lat, double radius,
                                 double* d_feat, int64_t feat_stride, int64_t* d_info, void* d_work,
                                 size_t work_bytes, void* stream)
{
    NM_ENTER_STREAM(ctx, stream);
    int rc = check_scale_args(ctx, "nm_scale_features", d_query, n_query, query_stride, d_search,
                              n_search, search_stride, d_feat, feat_stride, d_work);
    if (rc) return rc;
    rc = validate_lattice(ctx, lat);
    if (rc) return rc;
    if (!(radius >= 0.0)) NM_FAIL(ctx, NM_ERR_RADIUS, "radius must be non-negative");
    LatticeDev L = make_lattice_dev(lat);
    if (L.keybits > 64) NM_FAIL(ctx, NM_ERR_LATTICE, "lattice too large for the device sort key");
    int32_t dmin = 0;
    int W = candidate_width(radius, lat->edge, &dmin);
    if (W < 0) NM_FAIL(ctx, NM_ERR_RADIUS, "radius/edge ratio %g is outside the supported range",
                       radius / lat->edge);

    // the query cloud is the search cloud, or its leading rows: sort and index once
    const bool shared = (d_query == d_search && n_query <= n_search && n_query > 0 &&
                         query_stride == search_stride);
    ScaleLayout S;
    scale_layout(n_query > 0 ? n_query : 1, n_search, L, shared, &S);
    if (work_bytes < S.total)
        NM_FAIL(ctx, NM_ERR_WORKSPACE, "nm_scale_features: workspace %zu < required %zu", work_bytes,
                S.total);
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)d_work;

    // profile stages are [keys+sort | index | search kernel]: four marks per call.  with a separate
    // query cloud both sorts and the index are booked together under the first stage.
    nm_profile_mark(ctx, s);
    rc = nm_sort_cells(ctx, d_search, n_search, search_stride, L, (uint64_t*)(w + S.key_tmp),
                       (uint32_t*)(w + S.val_tmp), (uint64_t*)(w + S.key_sorted),
                       (uint32_t*)(w + S.val_sorted), w + S.sort_temp, S.sort_temp_bytes, s);
    if (rc) return rc;
    if (shared) nm_profile_mark(ctx, s);
    IndexDev I;
    rc = nm_index_build(ctx, (const uint64_t*)(w + S.key_sorted), n_search, S.ilay, w + S.index, &I,
                        s);
    if (rc) return rc;
    ScaleDev* d_scale = (ScaleDev*)(w + S.scale_dev);
    rc = nm_ladder_put(ctx, &L, &I, &radius, 1, -1, 0u, d_scale, nullptr, s);
    if (rc) return rc;

    const uint32_t* order = (const uint32_t*)(w + S.val_sorted);
    if (!shared && n_query > 0) {
        rc = nm_sort_cells(ctx, d_query, n_query, query_stride, L, (uint64_t*)(w + S.qkey_tmp),
                           (uint32_t*)(w + S.qval_tmp), (uint64_t*)(w + S.qkey_sorted),
                           (uint32_t*)(w + S.qval_sorted), w + S.sort_temp, S.sort_temp_bytes, s);
        if (rc) return rc;
        order = (const uint32_t*)(w + S.qval_sorted);
    }
    if (!shared) nm_profile_mark(ctx, s);
    nm_profile_mark(ctx, s);

    if (n_query > 0) {
        ScaleArgs A;
        A.query = d_query;
        A.nq = n_query;
        A.n_slots = shared ? n_search : n_query;
        A.qstride = query_stride;
        A.order = order;
        A.direct = 0;
        A.scales = d_scale;
        A.s_begin = 0;
        A.s_end = 1;
        A.dmin = dmin;
        A.W = W;
        A.feat = d_feat;
        A.fstride = feat_stride;
        fill_outputs(ctx, &A);
        A.sparse = ctx->knn_k > 0 ? (unsigned long long*)(w + S.knn.mask) : nullptr;
        A.sparse_words = S.knn.words;
        launch_scale_kernel(A, radius, lat->edge, W, false, s);
        ctx->profile_launches += 1;
        rc = launch_knn_fallback(ctx, A, 0, radius, lat->edge, (uint32_t*)(w + S.knn.list),
                                 (uint32_t*)(w + S.knn.count), s);
        if (rc) return rc;
    }
    nm_profile_mark(ctx, s);
    NM_HIP(ctx, hipGetLastError());
    if (d_info) k_publish_info_all<<<1, 64, 0, s>>>(d_scale, 1, d_info);
    nm_status_snapshot(ctx, s);
    return NM_OK;
}

// ---- the whole ladder: one spatial order for every scale ----------------------------------------------
// the workspace depends on the point counts and the number of scales only - not on the lattices, which
// may not exist on the host at all (nm_ladder_features builds them on the device): every scale gets room
// for as many leaves as there are search points and a directory twice that, and uses what its lattice
// needs (min(superblocks of the lattice, points) leaves).

struct LadderLayout {
    // per cloud: the order and the coordinates in that order
    size_t s_order, s_xyz;
    size_t q_order, q_xyz;
    size_t order_scratch, order_scratch_bytes;     // the spatial sort's keys, pair buffers and count matrix
    size_t minmax;                   // 6 doubles: the search cloud's extrema
    size_t bounds_scratch;           // the bounds pass's per-block extrema
    size_t ladder;                   // ScaleDev[n_scales]
    size_t order_dev;                // OrderDev
    size_t hash[NM_MAX_LADDER], leaf[NM_MAX_LADDER], counters[NM_MAX_LADDER];
    uint32_t hash_capacity, leaf_capacity;
    KnnLayout knn;
    size_t total;
};

// edges: the scales' edge lengths, or null.  with them, a scale that has the edge length of an earlier one gets
// no index of its own (it borrows that scale's: ScaleDev::shared) - a ladder of one voxel edge and three radii
// needs a third of the index memory.  without them every scale is given room (the upper bound
// nm_ladder_workspace_bytes reports; any layout made with edges fits inside it).
static void ladder_layout(int64_t nq, int64_t ns, int n_scales, bool shared, bool knn, LadderLayout* S,
                          const double* edges = nullptr)
{
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t at = off;
        off += align_up(bytes);
        return at;
    };
    S->s_order = take((size_t)ns * 4);
    S->s_xyz = take((size_t)ns * 24);
    if (!shared) {
        S->q_order = take((size_t)nq * 4);
        S->q_xyz = take((size_t)nq * 24);
    } else {
        S->q_order = S->q_xyz = 0;
    }
    S->order_scratch_bytes = nm_order_scratch_bytes(ns > nq ? ns : nq);
    S->order_scratch = take(S->order_scratch_bytes);
    S->minmax = take(64);
    S->bounds_scratch = take(NM_BOUNDS_SCRATCH_BYTES);
    S->ladder = take(sizeof(ScaleDev) * (size_t)NM_MAX_LADDER);
    S->order_dev = take(sizeof(OrderDev));
    uint64_t cap = (uint64_t)(ns > 1 ? ns : 1);
    uint64_t hcap = 64;
    while (hcap < cap * 2 && hcap < (1ull << 31)) hcap <<= 1;      // the mask is 32 bits wide
    S->leaf_capacity = (uint32_t)cap;
    S->hash_capacity = (uint32_t)hcap;
    for (int i = 0; i < n_scales && i < NM_MAX_LADDER; ++i) {
        int owner = i;
        if (edges)
            for (int j = i - 1; j >= 0; --j)
                if (edges[j] == edges[i]) owner = j;
        if (owner != i) {
            S->hash[i] = S->hash[owner];
            S->leaf[i] = S->leaf[owner];
        } else {
            S->hash[i] = take((size_t)hcap * sizeof(HashEntry));
            S->leaf[i] = take((size_t)cap * NM_LEAF_WORDS * 4);
        }
        S->counters[i] = take(256);
    }
    if (knn) knn_layout(ns > nq ? ns : nq, n_scales, &off, &S->knn);
    else S->knn = KnnLayout{0, 0, 0, 0};
    S->total = off;
}

extern "C" size_t nm_ladder_workspace_bytes(int64_t n_query, int64_t n_search, int32_t n_scales)
{
    if (n_scales < 1 || n_scales > NM_MAX_LADDER || n_query < 0 || n_search < 1) return 0;
    LadderLayout S;
    // sized for separate clouds and the kNN fallback so one workspace serves every mode
    ladder_layout(n_query > 0 ? n_query : 1, n_search, n_scales, false, true, &S);
    return S.total;
}

// the same with the edge lengths known: scales of equal edge share one index
extern "C" size_t nm_ladder_workspace_bytes_for(int64_t n_query, int64_t n_search, const double* edges,
                                                int32_t n_scales)
{
    if (!edges || n_scales < 1 || n_scales > NM_MAX_LADDER || n_query < 0 || n_search < 1) return 0;
    LadderLayout S;
    ladder_layout(n_query > 0 ? n_query : 1, n_search, n_scales, false, true, &S, edges);
    return S.total;
}

extern "C" size_t nm_multiscale_workspace_bytes(int64_t n_query, int64_t n_search,
                                                const nm_lattice* lats, int32_t n_scales)
{
    if (!lats) return 0;
    return nm_ladder_workspace_bytes(n_query, n_search, n_scales);
}

// what both ladder entry points do once the scale array is in device memory
struct LadderCall {
    const double* d_query; int64_t n_query, query_stride;
    const double* d_search; int64_t n_search, search_stride;
    const double* edges; const double* radii; int32_t n_scales;
    double* d_feat; int64_t feat_stride; int64_t* d_info;
};

static int run_ladder(nm_ctx* ctx, const LadderCall& C, const LadderLayout& S, char* w, bool shared,
                      hipStream_t s)
{
    const ScaleDev* d_ladder = (const ScaleDev*)(w + S.ladder);
    const OrderDev* d_order = (const OrderDev*)(w + S.order_dev);
    // (the search cloud's coordinates are gathered into sorted order by the index builder below: no gather
    // kernel, one read of the sorted copy less.  NM_FUSE_GATHER = 0: a gather kernel behind the sort)
    // (the sort's counting kernels reset the indexes on the way: NM_FUSE_CLEAR)
    int rc = nm_order_build(ctx, C.d_search, C.n_search, C.search_stride, d_order, w + S.order_scratch,
                            S.order_scratch_bytes, (uint32_t*)(w + S.s_order),
                            NM_FUSE_GATHER ? nullptr : (double*)(w + S.s_xyz), s,
                            NM_FUSE_CLEAR ? d_ladder : nullptr, C.n_scales);
    if (rc) return rc;
    const uint32_t* q_order = (const uint32_t*)(w + S.s_order);
    const double* q_xyz = (const double*)(w + S.s_xyz);
    if (!shared && C.n_query > 0) {
        rc = nm_order_build(ctx, C.d_query, C.n_query, C.query_stride, d_order, w + S.order_scratch,
                            S.order_scratch_bytes, (uint32_t*)(w + S.q_order), (double*)(w + S.q_xyz), s);
        if (rc) return rc;
        q_order = (const uint32_t*)(w + S.q_order);
        q_xyz = (const double*)(w + S.q_xyz);
    }
    nm_profile_mark(ctx, s);           // end of the "order" stage
    // every scale has its own index; one launch clears them all, one builds them all (the block keeps its
    // points and walks the scales), one counts them all at the end
    if (!NM_FUSE_CLEAR) {
        rc = nm_index_clear_all(ctx, d_ladder, C.n_scales, s);
        if (rc) return rc;
    }
    if (NM_FUSE_GATHER)
        rc = nm_index_build_ladder_gather(ctx, C.d_search, C.n_search, C.search_stride,
                                          (const uint32_t*)(w + S.s_order), (double*)(w + S.s_xyz), d_ladder, 0,
                                          C.n_scales, s);
    else
        rc = nm_index_build_ladder(ctx, (const double*)(w + S.s_xyz), C.n_search, d_ladder, 0, C.n_scales, s);
    if (rc) return rc;
    nm_profile_mark(ctx, s);           // end of the "index" stage

    if (C.n_query > 0) {
        ScaleArgs A;
        A.query = q_xyz;
        A.nq = C.n_query;
        A.n_slots = shared ? C.n_search : C.n_query;
        A.qstride = 3;
        A.order = q_order;
        A.direct = 1;
        A.scales = d_ladder;
        A.feat = C.d_feat;
        A.fstride = C.feat_stride;
        fill_outputs(ctx, &A);
        A.sparse = ctx->knn_k > 0 ? (unsigned long long*)(w + S.knn.mask) : nullptr;
        A.sparse_words = S.knn.words;
        const bool want_forest = ctx->forest_on;
        // in the search kernel's epilogue only when asked to (nm_set_forest_mode) and when nothing rewrites
        // rows afterwards (the kNN fallback does)
        const bool in_kernel = want_forest && ctx->forest_epilogue && ctx->knn_k == 0;
        bool forest_done = false;
        if (want_forest) {
            A.F = ctx->forest;
            A.F.n_features = 4 * C.n_scales;
        }
        // consecutive scales with the same candidate window run in ONE launch (the wave keeps its queries
        // and walks the scales): the benchmark ladder, r = 3e throughout, is a single launch
        int i = 0;
        while (i < C.n_scales) {
            int32_t dmin = 0;
            const int W = candidate_width(C.radii[i], C.edges[i], &dmin);
            const double rho = C.radii[i] / C.edges[i];
            int j = i + 1;
            if (ctx->fuse_scales) {
                while (j < C.n_scales) {
                    int32_t dm2 = 0;
                    const int W2 = candidate_width(C.radii[j], C.edges[j], &dm2);
                    if (W2 != W || fabs(C.radii[j] / C.edges[j] - rho) > 1e-12 * rho) break;
                    ++j;
                }
            }
            A.s_begin = i;
            A.s_end = j;
            A.dmin = dmin;
            A.W = W;
            const bool last = j == C.n_scales;
            forest_done = launch_scale_kernel(A, C.radii[i], C.edges[i], W, in_kernel && last, s);
            ctx->profile_launches += 1;
            for (int k = i; k < j; ++k) {
                rc = launch_knn_fallback(ctx, A, k, C.radii[k], C.edges[k], (uint32_t*)(w + S.knn.list),
                                         (uint32_t*)(w + S.knn.count), s);
                if (rc) return rc;
            }
            i = j;
        }
        if (want_forest && !forest_done) {
            if (ctx->forest_mode == 2) {
                // the classifier as its own launch over the finished matrix, trees staged through LDS
                rc = nm_forest_rows(ctx, A.F, C.d_feat, C.feat_stride, C.n_query, A.F.n_features, s);
                if (rc) return rc;
            } else {
                // the classifier as its own launch behind the last scale, rows taken in the spatial order
                A.s_begin = 0;
                A.s_end = C.n_scales;
                k_forest_ordered<<<(int)((A.n_slots + 63) / 64), 64, 0, s>>>(A, A.F.nodes);
            }
        }
    }
    nm_profile_mark(ctx, s);           // end of the "search" stage
    NM_HIP(ctx, hipGetLastError());
    if (C.d_info) {
        rc = nm_index_count_all(ctx, d_ladder, C.n_scales, s);
        if (rc) return rc;
        k_publish_info_all<<<1, 64, 0, s>>>(d_ladder, C.n_scales, C.d_info);
        NM_HIP(ctx, hipGetLastError());
    }
    nm_status_snapshot(ctx, s);
    return NM_OK;
}

static int check_ladder_args(nm_ctx* ctx, const char* who, int32_t n_scales, const double* edges,
                             const double* radii, int64_t feat_stride, int* finest)
{
    if (n_scales > NM_MAX_LADDER)
        NM_FAIL(ctx, NM_ERR_INVALID, "%s: at most %d scales per call", who, NM_MAX_LADDER);
    if (feat_stride < 4 * (int64_t)n_scales)
        NM_FAIL(ctx, NM_ERR_INVALID, "%s: feat_stride < 4 * n_scales", who);
    if (ctx->cov_out && ctx->cov_stride < 6 * (int64_t)n_scales)
        NM_FAIL(ctx, NM_ERR_INVALID, "%s: covariance stride < 6 * n_scales", who);
    if (ctx->normal_out && ctx->normal_stride < 3 * (int64_t)n_scales)
        NM_FAIL(ctx, NM_ERR_INVALID, "%s: normal stride < 3 * n_scales", who);
    if (ctx->forest_on && ctx->forest_features != 4 * n_scales)
        NM_FAIL(ctx, NM_ERR_INVALID, "%s: the forest takes %d features, the ladder makes %d", who,
                ctx->forest_features, 4 * n_scales);
    *finest = 0;
    for (int i = 0; i < n_scales; ++i) {
        if (!(edges[i] > 0.0)) NM_FAIL(ctx, NM_ERR_LATTICE, "edge length must be positive");
        if (!(radii[i] >= 0.0)) NM_FAIL(ctx, NM_ERR_RADIUS, "radius must be non-negative");
        int32_t dm;
        if (candidate_width(radii[i], edges[i], &dm) < 0)
            NM_FAIL(ctx, NM_ERR_RADIUS, "radius/edge ratio %g is outside the supported range",
                    radii[i] / edges[i]);
        if (edges[i] < edges[*finest]) *finest = i;
    }
    return NM_OK;
}

extern "C" int nm_multiscale_features(nm_ctx* ctx, const double* d_query, int64_t n_query,
                                      int64_t query_stride, const double* d_search, int64_t n_search,
                                      int64_t search_stride, const nm_lattice* lats,
                                      const double* radii, int32_t n_scales, double* d_feat,
                                      int64_t feat_stride, int64_t* d_info, void* d_work,
                                      size_t work_bytes, void* stream)
{
    NM_ENTER_STREAM(ctx, stream);
    if (n_scales < 0 || (n_scales > 0 && (!lats || !radii)))
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_multiscale_features: bad scale arguments");
    if (n_scales == 0) return NM_OK;
    double edges[NM_MAX_LADDER];
    for (int i = 0; i < n_scales && i < NM_MAX_LADDER; ++i) edges[i] = lats[i].edge;
    int finest = 0;
    int rc = check_ladder_args(ctx, "nm_multiscale_features", n_scales, edges, radii, feat_stride, &finest);
    if (rc) return rc;
    rc = check_scale_args(ctx, "nm_multiscale_features", d_query, n_query, query_stride, d_search,
                          n_search, search_stride, d_feat, feat_stride, d_work);
    if (rc) return rc;
    for (int i = 0; i < n_scales; ++i) {
        rc = validate_lattice(ctx, &lats[i]);
        if (rc) return rc;
        if (make_lattice_dev(&lats[i]).keybits > 64)
            NM_FAIL(ctx, NM_ERR_LATTICE, "lattice too large for the device sort key");
    }
    const bool shared = (d_query == d_search && n_query <= n_search && n_query > 0 &&
                         query_stride == search_stride);
    LadderLayout S;
    // (every scale keeps room of its own here: the caller's lattices decide which scales share an index, not
    // their edge lengths alone - nm_ladder_put)
    ladder_layout(n_query > 0 ? n_query : 1, n_search, n_scales, shared, ctx->knn_k > 0, &S);
    if (work_bytes < S.total)
        NM_FAIL(ctx, NM_ERR_WORKSPACE, "nm_multiscale_features: workspace %zu < required %zu",
                work_bytes, S.total);
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)d_work;
    nm_profile_mark(ctx, s);
    // the host's lattices, with indexes sized for them, into the device array the kernels read
    LatticeDev L[NM_MAX_LADDER];
    IndexDev I[NM_MAX_LADDER];
    for (int i = 0; i < n_scales; ++i) {
        L[i] = make_lattice_dev(&lats[i]);
        IndexLayout lay;
        nm_index_layout(L[i], n_search, &lay);
        I[i].hash = (HashEntry*)(w + S.hash[i]);
        I[i].leaf = (uint32_t*)(w + S.leaf[i]);
        I[i].counters = (uint32_t*)(w + S.counters[i]);
        I[i].hash_mask = lay.hash_capacity - 1;
        I[i].leaf_capacity = lay.leaf_capacity;
        I[i].status = ctx->d_status;
    }
    ctx->order_points = n_query > n_search ? n_query : n_search;     // (the sort's pass count: nm_order_plan)
    rc = nm_ladder_put(ctx, L, I, radii, n_scales, finest, S.leaf_capacity, (ScaleDev*)(w + S.ladder),
                       (OrderDev*)(w + S.order_dev), s);
    if (rc) return rc;
    LadderCall C{d_query, n_query, query_stride, d_search, n_search, search_stride, edges, radii,
                 n_scales, d_feat, feat_stride, d_info};
    return run_ladder(ctx, C, S, w, shared, s);
}

// ---- the ladder with the lattices built on the device ----------------------------------------------------
extern "C" int nm_ladder_features(nm_ctx* ctx, const double* d_query, int64_t n_query,
                                  int64_t query_stride, const double* d_search, int64_t n_search,
                                  int64_t search_stride, const double* edges, const double* radii,
                                  int32_t n_scales, const double* d_minmax, double* d_feat,
                                  int64_t feat_stride, int64_t* d_info, void* d_work, size_t work_bytes,
                                  void* stream)
{
    NM_ENTER_STREAM(ctx, stream);
    if (n_scales < 0 || (n_scales > 0 && (!edges || !radii)))
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_ladder_features: bad scale arguments");
    if (n_scales == 0) return NM_OK;
    int finest = 0;
    int rc = check_ladder_args(ctx, "nm_ladder_features", n_scales, edges, radii, feat_stride, &finest);
    if (rc) return rc;
    rc = check_scale_args(ctx, "nm_ladder_features", d_query, n_query, query_stride, d_search, n_search,
                          search_stride, d_feat, feat_stride, d_work);
    if (rc) return rc;
    const bool shared = (d_query == d_search && n_query <= n_search && n_query > 0 &&
                         query_stride == search_stride);
    LadderLayout S;
    ladder_layout(n_query > 0 ? n_query : 1, n_search, n_scales, shared, ctx->knn_k > 0, &S, edges);
    if (work_bytes < S.total)
        NM_FAIL(ctx, NM_ERR_WORKSPACE, "nm_ladder_features: workspace %zu < required %zu", work_bytes,
                S.total);
    hipStream_t s = (hipStream_t)stream;
    char* w = (char*)d_work;
    nm_profile_mark(ctx, s);
    // the search cloud's extrema never leave the device: bounds pass (unless the caller has them, e.g. the
    // global extrema of a multi-GPU job), then one tiny kernel turns them into every scale's lattice
    const void* partial = nullptr;
    int bounds_blocks = 0;
    if (!d_minmax) {
        rc = nm_bounds_partial(ctx, d_search, n_search, search_stride, w + S.bounds_scratch, &bounds_blocks, s);
        if (rc) return rc;
        partial = w + S.bounds_scratch;
    }
    void* hash[NM_MAX_LADDER];
    void* leaf[NM_MAX_LADDER];
    void* counters[NM_MAX_LADDER];
    for (int i = 0; i < n_scales; ++i) {
        hash[i] = w + S.hash[i];
        leaf[i] = w + S.leaf[i];
        counters[i] = w + S.counters[i];
    }
    ctx->order_points = n_query > n_search ? n_query : n_search;     // (the sort's pass count: nm_order_plan)
    rc = nm_ladder_make(ctx, d_minmax, partial, bounds_blocks, (double*)(w + S.minmax), edges, radii, n_scales,
                        finest, hash, leaf, counters, S.hash_capacity, S.leaf_capacity,
                        (ScaleDev*)(w + S.ladder), (OrderDev*)(w + S.order_dev), s);
    if (rc) return rc;
    LadderCall C{d_query, n_query, query_stride, d_search, n_search, search_stride, edges, radii,
                 n_scales, d_feat, feat_stride, d_info};
    return run_ladder(ctx, C, S, w, shared, s);
}

// ---------------------------------------------------------------------------------------------------
// neighbor lists (parity / inspection): per candidate cell, binary-search the sorted unique address
// array; emitted index = position in that array = the reference's search-voxel index.
// ---------------------------------------------------------------------------------------------------

__device__ __forceinline__ int64_t nm_find_address(const int64_t* __restrict__ addr, int64_t m,
                                                   int64_t key)
{
    int64_t lo = 0, hi = m;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (addr[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return (lo < m && addr[lo] == key) ? lo : -1;
}

__global__ __launch_bounds__(64) void k_scale_neighbors(const double* __restrict__ query, int64_t nq,
                                                        int64_t qstride,
                                                        const int64_t* __restrict__ addr, int64_t m,
                                                        LatticeDev L, double r2, int32_t dmin,
                                                        int32_t W, int32_t* __restrict__ count,
                                                        const int64_t* __restrict__ offsets,
                                                        int64_t* __restrict__ index)
{
    const int64_t q = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (q >= nq) return;
    const double* p = query + q * qstride;
    const double qx = p[0], qy = p[1], qz = p[2];
    const int32_t hx = nm_clamp_cell(nm_cell_f(qx, L.min_x, L.edge));
    const int32_t hy = nm_clamp_cell(nm_cell_f(qy, L.min_y, L.edge));
    const int32_t hz = nm_clamp_cell(nm_cell_f(qz, L.min_z, L.edge));
    int32_t n = 0;
    int64_t* out = index ? index + offsets[q] : nullptr;
    // z outer, x inner: ascending address = ascending index
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
            for (int32_t i = 0; i < W; ++i) {
                const int32_t gx = hx + dmin + i;
                if (gx < 0 || gx >= (1 << L.wx)) continue;
                d = qx - nm_centre(gx, L.min_x, L.edge, L.half_edge);
                const double s = (d * d + dy2) + dz2;
                if (!(s <= r2)) continue;
                const int64_t a = (int64_t)gx + ((int64_t)gy << L.s0) + ((int64_t)gz << L.s1);
                const int64_t pos = nm_find_address(addr, m, a);
                if (pos < 0) continue;
                if (out) out[n] = pos;
                ++n;
            }
        }
    }
    if (count) count[q] = n;
}

extern "C" int nm_scale_neighbors(nm_ctx* ctx, const double* d_query, int64_t n_query,
                                  int64_t query_stride, const int64_t* d_addr, int64_t m,
                                  const nm_lattice* lat, double radius, int32_t* d_nbr_count,
                                  const int64_t* d_nbr_offsets, int64_t* d_nbr_index, void* stream)
{
    NM_ENTER_STREAM(ctx, stream);
    if (n_query < 0 || m < 0 || query_stride < 3 || (n_query > 0 && !d_query) || (m > 0 && !d_addr))
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_scale_neighbors: bad arguments");
    if (!d_nbr_count && !d_nbr_index)
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_scale_neighbors: nothing to write");
    if (d_nbr_index && !d_nbr_offsets)
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_scale_neighbors: index output needs offsets");
    int rc = validate_lattice(ctx, lat);
    if (rc) return rc;
    int32_t dmin = 0;
    int W = candidate_width(radius, lat->edge, &dmin);
    if (W < 0) NM_FAIL(ctx, NM_ERR_RADIUS, "radius/edge ratio outside the supported range");
    if (n_query == 0) return NM_OK;
    LatticeDev L = make_lattice_dev(lat);
    k_scale_neighbors<<<(int)((n_query + 63) / 64), 64, 0, (hipStream_t)stream>>>(
        d_query, n_query, query_stride, d_addr, m, L, radius * radius, dmin, W, d_nbr_count,
        d_nbr_offsets, d_nbr_index);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}

// ---------------------------------------------------------------------------------------------------
// explicit neighborhoods: features.population / centroid / pca on arbitrary point sets (CSR)
// ---------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(64) void k_neighborhood_features(const double* __restrict__ pts,
                                                              const int64_t* __restrict__ offsets,
                                                              const double* __restrict__ query,
                                                              int64_t nb, double* __restrict__ feat,
                                                              int64_t fstride)
{
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= nb) return;
    const int64_t lo = offsets[b], hi = offsets[b + 1];
    const double n = (double)(hi - lo);
    double* o = feat + b * fstride;
    o[0] = n;
    o[1] = 0.0;
    o[2] = 0.0;
    o[3] = 0.0;
    if (hi <= lo) return;
    // mean first, then deviations, like numpy.cov (features.py:43)
    double mx = 0, my = 0, mz = 0;
    for (int64_t i = lo; i < hi; ++i) {
        mx += pts[i * 3];
        my += pts[i * 3 + 1];
        mz += pts[i * 3 + 2];
    }
    mx /= n;
    my /= n;
    mz /= n;
    const double ux = query[b * 3] - mx, uy = query[b * 3 + 1] - my, uz = query[b * 3 + 2] - mz;
    o[1] = sqrt(ux * ux + uy * uy + uz * uz);
    if (hi - lo < 2) return;
    double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0;
    for (int64_t i = lo; i < hi; ++i) {
        double x = pts[i * 3] - mx, y = pts[i * 3 + 1] - my, z = pts[i * 3 + 2] - mz;
        a00 += x * x;
        a01 += x * y;
        a02 += x * z;
        a11 += y * y;
        a12 += y * z;
        a22 += z * z;
    }
    double l0, l1, l2;
    nm_eig3(a00, a01, a02, a11, a12, a22, l0, l1, l2);
    const double tr = a00 + a11 + a22;
    if (tr > 0.0) {
        o[2] = l0 / tr;
        o[3] = l1 / tr;
    }
}

extern "C" int nm_neighborhood_features(nm_ctx* ctx, const double* d_points, const int64_t* d_offsets,
                                        const double* d_query, int64_t n_neighborhoods,
                                        double* d_feat, int64_t feat_stride, void* stream)
{
    NM_ENTER_STREAM(ctx, stream);
    if (n_neighborhoods < 0 || feat_stride < 4 ||
        (n_neighborhoods > 0 && (!d_offsets || !d_query || !d_feat)))
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_neighborhood_features: bad arguments");
    if (n_neighborhoods == 0) return NM_OK;
    k_neighborhood_features<<<(int)((n_neighborhoods + 63) / 64), 64, 0, (hipStream_t)stream>>>(
        d_points, d_offsets, d_query, n_neighborhoods, d_feat, feat_stride);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}
