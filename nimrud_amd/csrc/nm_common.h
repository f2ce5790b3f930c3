// nm_common.h - shared host/device definitions for libnimrud_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/nimrud_hip.h"

// sticky status (nm_ctx::h_status / d_status, 64 words of device-mapped host memory).  kernels only ever set
// words; every feature call leaves an event behind, and the host turns a non-zero word into NM_ERR_HIP /
// NM_ERR_LATTICE on the NEXT call or in nm_check - never a silent result.
enum {
    NM_ST_INDEX_TIMEOUT = 0,   // the index builder's bounded wait for a leaf number ran out
    NM_ST_LEAF_OVERFLOW = 1,   // more occupied superblocks than the workspace has leaves for
    NM_ST_LATTICE = 2,         // a lattice built on the device is not addressable (code below)
    NM_ST_WORDS = 64
};
enum {
    NM_LAT_TOO_SMALL_EDGE = 1,   // sum of widths > 64 (geometry.py:59-60)
    NM_LAT_NO_EXTENT = 2,        // some width < 1 (the reference fails at geometry.py:74)
    NM_LAT_DEVICE_LIMIT = 3,     // some width > 30: outside the device path's range
    NM_LAT_NOT_FINITE = 4        // extrema are not finite numbers
};

constexpr int NM_FUSED_FOREST_FEATURES = 21;    // 21 * 64 * 4 B = the LDS the search phase leaves behind
constexpr int NM_FUSED_FOREST_CLASSES = 8;

// the classifier behind the last scale (nm_set_forest_output): 8-byte nodes, see nm_forest::d_packed8
struct ForestDev {
    const uint2* nodes;          // {fp32 threshold rounded down, packed}: bit 31 leaf (row of its class distribution
                                 // << 13); else left child << 13 | feature << 8 (ForestModel.pack_nodes8)
    const double* leaf_value;    // (n_leaves, leaf_stride), rows sum to 1
    int32_t leaf_stride;         // doubles per row; 8 and 64-byte aligned rows: fetched with 16-byte loads
    const int32_t* roots;        // root node of each tree
    int32_t n_trees, n_classes, n_features;
    int32_t n_nodes;             // nodes of all trees (tree t holds nodes roots[t] .. roots[t+1] - 1, breadth-first)
    double* proba;               // (Nq, pstride) or null
    int64_t pstride;
    int32_t* label;              // (Nq,) or null
};

struct nm_ctx {
    int device;
    std::string error;
    int num_cus;
    uint32_t* h_status = nullptr;       // pinned, device-mapped host memory, NM_ST_WORDS words, zero = healthy
    uint32_t* d_status = nullptr;       // its device address: what the kernels write through
    hipEvent_t status_event = nullptr;  // end of the last feature call
    bool status_pending = false;
    int sticky = 0;                     // a past asynchronous failure: every later call returns it
    // stage timing (nm_profile_begin/end): events[4*k .. 4*k+3] bracket the three stages of call k
    bool profiling = false;
    std::vector<hipEvent_t> events;
    size_t events_used = 0;
    // whole-ladder pipelining: the index of scale i+1 is built on `aux` while the fused kernel of
    // scale i runs on the caller's stream (nm_set_overlap)
    bool overlap = false;        // measured: the long kernel starves the build stream; no gain yet
    hipStream_t aux = nullptr;
    std::vector<hipEvent_t> sync_events;
    // k-nearest-voxel fallback for sparse neighborhoods (nm_set_knn_fallback); 0 = off
    double* cov_out = nullptr;   // nm_set_covariance_output
    int64_t cov_stride = 0;
    double* normal_out = nullptr;   // nm_set_normal_output
    int64_t normal_stride = 0;
    int knn_k = 0;
    double knn_radius_factor = 3.0;
    // classifier behind the last scale of the ladder (nm_set_forest_output)
    bool forest_on = false;
    bool forest_epilogue = true;     // true: inside the last search kernel; false: own launch behind it
    int forest_mode = 1;             // nm_set_forest_mode: 0 own launch (row walk from memory), 1 epilogue, 2 own
                                     // launch with the trees staged through LDS (k_forest_tiles)
    bool forest_tiles_attr = false;  // k_forest_tiles' dynamic-LDS attribute is set
    ForestDev forest{};
    int forest_features = 0;
    // consecutive scales with the same candidate window run in one launch (nm_set_fuse_scales)
    bool fuse_scales = true;
    int64_t profile_launches = 0;   // search-kernel launches since nm_profile_begin
    // the halo exchange's last plan (nm_halo_exchange): the gathered count matrix of a step, kept on the host
    // so that a step on UNCHANGED tiles (NM_HALO_REUSE_PLAN) needs no host synchronisation to learn the sizes
    std::vector<int64_t> halo_matrix;
    int32_t halo_ranks = 0, halo_rank = -1, halo_mode = -1;
    int64_t halo_n = -1;
    const void* halo_work = nullptr;
    int64_t halo_host_syncs = 0, halo_exchanges = 0;
    int64_t order_points = 0;       // points of the largest cloud of the call in progress: decides the sort's pass count
                                    // on the host (nm_order_plan's rule; set by the entry points)
    uint32_t order_attr_set = 0;    // spatial sort: kernel families whose dynamic-LDS attribute is set (nm_order.hip)
};

// classifier on the rows of a finished feature matrix, into F.proba / F.label (nm_api.hip)
int nm_forest_rows(nm_ctx* ctx, const ForestDev& F, const double* d_feat, int64_t feat_stride, int64_t n,
                   int32_t n_features, hipStream_t s);

// next profiling event recorded on `s`, or a no-op when profiling is off
static inline void nm_profile_mark(nm_ctx* ctx, hipStream_t s)
{
    if (!ctx->profiling) return;
    if (ctx->events_used == ctx->events.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        ctx->events.push_back(e);
    }
    (void)hipEventRecord(ctx->events[ctx->events_used++], s);
}

// RAII: make the context's device current for the duration of an entry point and restore the caller's
// (the process-wide current device is also what torch reads)
struct nm_device_guard {
    int prev = -1;
    bool switched = false;
    explicit nm_device_guard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~nm_device_guard()
    {
        if (switched) (void)hipSetDevice(prev);
    }
};

// defined in nm_api.hip: fold a completed status snapshot into ctx->sticky; returns the sticky status
int nm_status_poll(nm_ctx* ctx, bool wait);
// enqueue a snapshot of the device status words behind the work just enqueued on `s`
void nm_status_snapshot(nm_ctx* ctx, hipStream_t s);

// is `s` being captured into a hipGraph?  (a step has no host synchronisation and can be captured; while it
// is, the library must not query events - that would invalidate the capture)
static inline bool nm_capturing(hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

// first lines of every entry point that launches work or creates streams / events
#define NM_ENTER(ctx)                                   \
    if (!(ctx)) return NM_ERR_INVALID;                  \
    nm_device_guard _nm_guard((ctx)->device);           \
    {                                                   \
        const int _st = nm_status_poll((ctx), false);   \
        if (_st) return _st;                            \
    }
// the same for entry points that enqueue on a caller's stream: no event query while it is being captured
#define NM_ENTER_STREAM(ctx, stream)                                                 \
    if (!(ctx)) return NM_ERR_INVALID;                                               \
    nm_device_guard _nm_guard((ctx)->device);                                        \
    if (!nm_capturing((hipStream_t)(stream))) {                                      \
        const int _st = nm_status_poll((ctx), false);                                \
        if (_st) return _st;                                                         \
    }

#define NM_FAIL(ctx, code, ...)                                   \
    do {                                                          \
        char _buf[512];                                           \
        snprintf(_buf, sizeof(_buf), __VA_ARGS__);                \
        if (ctx) (ctx)->error = _buf;                             \
        return (code);                                            \
    } while (0)

#define NM_HIP(ctx, call)                                                                 \
    do {                                                                                  \
        hipError_t _e = (call);                                                           \
        if (_e != hipSuccess)                                                             \
            NM_FAIL(ctx, NM_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(_e));      \
    } while (0)

// ---- superblock ("sb") geometry of the occupancy index -------------------------------------------
// the occupied lattice cells of one scale are stored as a sparse set of dense leaves.  a leaf covers
// SBX x SBY x SBZ = 32 x 8 x 8 cells and is 64 row-words of 32 bits: word (lz*8 + ly), bit lx.
// rows run along x because the reference packs x in the low address bits (geometry.py:111-115).
constexpr int NM_SBX_BITS = 5;
constexpr int NM_SBY_BITS = 3;
constexpr int NM_SBZ_BITS = 3;
constexpr int NM_LOCAL_BITS = NM_SBX_BITS + NM_SBY_BITS + NM_SBZ_BITS;   // 11
constexpr int NM_LEAF_WORDS = 1 << (NM_SBY_BITS + NM_SBZ_BITS);          // 64 x u32 = 256 B
constexpr uint64_t NM_HASH_EMPTY = ~0ull;

// device view of a lattice plus derived constants
struct LatticeDev {
    double min_x, min_y, min_z;
    double edge;
    double half_edge;           // edge * 0.5 (exact)
    double inv_edge;            // fl(1 / edge): the fast path of nm_cell_index
    int32_t wx, wy, wz;         // address widths (geometry.py:56)
    int32_t s0, s1;             // address shifts (geometry.py:62)
    int32_t bx, by, bz;         // bits of the superblock coordinate per axis: max(w - local, 0)
    int32_t keybits;            // significant bits of the sort key
};

static inline LatticeDev make_lattice_dev(const nm_lattice* lat)
{
    LatticeDev d;
    d.min_x = lat->min_corner[0];
    d.min_y = lat->min_corner[1];
    d.min_z = lat->min_corner[2];
    d.edge = lat->edge;
    d.half_edge = lat->edge * 0.5;
    d.inv_edge = 1.0 / lat->edge;
    d.wx = lat->widths[0];
    d.wy = lat->widths[1];
    d.wz = lat->widths[2];
    d.s0 = lat->shifts[0];
    d.s1 = lat->shifts[1];
    d.bx = d.wx > NM_SBX_BITS ? d.wx - NM_SBX_BITS : 0;
    d.by = d.wy > NM_SBY_BITS ? d.wy - NM_SBY_BITS : 0;
    d.bz = d.wz > NM_SBZ_BITS ? d.wz - NM_SBZ_BITS : 0;
    d.keybits = NM_LOCAL_BITS + d.bx + d.by + d.bz;
    return d;
}

static inline int validate_lattice(nm_ctx* ctx, const nm_lattice* lat)
{
    if (!lat) NM_FAIL(ctx, NM_ERR_INVALID, "lattice is null");
    if (!(lat->edge > 0.0)) NM_FAIL(ctx, NM_ERR_LATTICE, "edge length must be positive");
    int sum = 0;
    for (int a = 0; a < 3; ++a) {
        if (lat->widths[a] < 1 || lat->widths[a] > 30)
            NM_FAIL(ctx, NM_ERR_LATTICE,
                    "axis %d address width %d outside the device path's range [1,30]", a,
                    lat->widths[a]);
        sum += lat->widths[a];
    }
    if (sum > 64) NM_FAIL(ctx, NM_ERR_LATTICE, "edge length is too small to address this space");
    if (lat->shifts[0] != lat->widths[0] || lat->shifts[1] != lat->widths[0] + lat->widths[1])
        NM_FAIL(ctx, NM_ERR_LATTICE, "shifts are not the cumulative widths");
    return NM_OK;
}

// one slot of the superblock directory.  key and value sit in one 16-byte entry so that a probe is a
// single load (the builders are latency-bound: every dependent memory round trip counts).
struct HashEntry {
    uint64_t key;            // superblock key, NM_HASH_EMPTY = free
    uint32_t val;            // leaf number
    uint32_t pad;
};

// the occupancy index of one scale (all pointers into the caller's workspace).
// DENSE form (hash == nullptr): the lattice has so few superblocks (at most 2^21 and no more than points) that
// every one of them gets a leaf, leaf number = superblock key - no table, no probing, nothing to insert or to
// wait for; hash_mask = number of superblocks - 1.
struct IndexDev {
    struct HashEntry* hash;  // open-addressing table: one 16-byte entry per slot, key + leaf number; or null
    uint32_t hash_mask;      // capacity - 1 (capacity is a power of two)
    uint32_t* leaf;          // leaves, NM_LEAF_WORDS words each
    uint32_t leaf_capacity;  // leaves the workspace has room for
    uint32_t* counters;      // [0] leaves allocated, [1] occupied cells M (low), [2] overflow flag
    uint32_t* status;        // the context's sticky status words (NM_ST_*)
    // MAP form (map != nullptr; it lives where the table would): one word per superblock of the lattice, the
    // leaf number or NM_MAP_EMPTY; hash_mask = number of superblocks - 1.  a lattice with too many superblocks
    // for a leaf each (dense form) but few enough for a word each - a 50 M-point scan's finest scale: 10 M
    // superblocks, 0.7 M of them occupied - is found with ONE load and no probing, and cleared in 42 MB instead
    // of a table sized for as many entries as there are points (0.5 GB)
    uint32_t* map = nullptr;
};
constexpr uint32_t NM_MAP_EMPTY = 0xFFFFFFFFu;      // (what the 0xFF fill leaves)
constexpr uint32_t NM_MAP_NONE = 0xFFFFFFFEu;       // no room for a leaf (capacity overflow, flagged)
constexpr uint32_t NM_MAP_PENDING = 0xFFFFFFFDu;    // entered, its leaf not published yet (the builder waits)

// ---- the scale ladder in device memory ------------------------------------------------------------------
// everything a kernel needs to know about one analysis scale.  the ladder path keeps an array of these in
// its workspace: written by k_put_ladder from host lattices (nm_multiscale_features) or built on the device
// from the cloud's extrema (nm_ladder_features), so that no lattice ever has to visit the host.
struct ScaleDev {
    LatticeDev L;
    IndexDev I;
    double r2;               // radius * radius (fp64 product, as scipy forms it)
    int32_t valid;           // 0: the lattice cannot be addressed; every kernel leaves at once
    int32_t prune_ok;        // static pruning of the candidate window is sound for this lattice
    uint32_t* stats;         // this scale's own counter block: [8] neighborhoods below 2 voxels, [9] extra passes
    int32_t shared;          // 1: same edge length as an earlier scale of the ladder - same lattice, and I is THAT
                             // scale's index (the reference's ladders look like this: one voxel edge, several
                             // radii - nimrud/utils/point_clouds.py:29-35); nobody clears, builds or counts it twice
    int32_t reserved;
};

// how the three axes share the compact Z-order key of the one-time spatial sort (k_order_keys)
struct ZLayout {
    int32_t w1, w2;          // smallest and middle width
    int32_t off2[3];         // slot of an axis in the 2-way zone, -1 for the axis with the smallest width
};

struct OrderDev {
    LatticeDev L;            // the finest lattice of the ladder
    ZLayout Z;
    int32_t morton;          // 1: compact Z-order key, 0: superblock key (some width above 21 bits)
    int32_t shift;           // low key bits dropped so that the key fits NM_ORDER_KEY_BITS
    int32_t valid;
    int32_t bpp;             // key bits per radix pass: ceil(key bits / passes), at most NM_ORDER_PASS_BITS
    int32_t passes;          // 2 for keys of at most 2 * NM_ORDER_PASS_BITS bits, else 3
};

// bits of the spatial order's sort key (a wider key loses its low bits): three radix passes of at most 10 bits.
// (the benchmark scene's finest lattice has 31: the one bit makes no measurable difference to the index
// build or the search; round 1 measured that six do)
#ifndef NM_ORDER_KEY_BITS
#define NM_ORDER_KEY_BITS 30
#endif
constexpr int NM_ORDER_PASS_BITS = 10;

// the spatial order of a lattice: which key, how its bits are laid out, what is dropped to fit the sort key
// `points`: the size of the largest cloud the order will be built for.  up to NM_ORDER_TWO_PASS_N points the key
// keeps 20 bits - a million key values, a few points to a value: which 64 points are a wave's is decided by
// the bits above - and is sorted in TWO passes, which the host then knows without asking the device: it launches
// two (the third pass's three launches were a tenth of a small cloud's step)
#ifndef NM_ORDER_TWO_PASS_LOG2
#define NM_ORDER_TWO_PASS_LOG2 22
#endif
constexpr int64_t NM_ORDER_TWO_PASS_N = (int64_t)1 << NM_ORDER_TWO_PASS_LOG2;
__host__ __device__ inline void nm_order_plan(const LatticeDev& L, OrderDev* O, int64_t points)
{
    O->L = L;
    int wmax = L.wx > L.wy ? L.wx : L.wy;
    if (L.wz > wmax) wmax = L.wz;
    O->morton = wmax <= 21;
    // axis roles in the compact Z-order key: the axis with the smallest width drops out of the 2-way
    // zone; with ties the later axis is treated as the smaller one (it then simply has no bits there)
    const int wd[3] = {L.wx, L.wy, L.wz};
    int smallest = 0;
    for (int a = 1; a < 3; ++a)
        if (wd[a] <= wd[smallest]) smallest = a;
    O->Z.w1 = wd[smallest];
    O->Z.w2 = 64;
    for (int a = 0, slot2 = 0; a < 3; ++a) {
        if (a == smallest) {
            O->Z.off2[a] = -1;
        } else {
            O->Z.off2[a] = slot2++;
            if (wd[a] < O->Z.w2) O->Z.w2 = wd[a];
        }
    }
    const int bits = O->morton ? L.wx + L.wy + L.wz : L.keybits;
    const int key_bits = points > 0 && points <= NM_ORDER_TWO_PASS_N ? 2 * NM_ORDER_PASS_BITS : NM_ORDER_KEY_BITS;
    O->shift = bits > key_bits ? bits - key_bits : 0;
    const int kept = bits - O->shift;
    O->passes = kept <= 2 * NM_ORDER_PASS_BITS ? 2 : 3;
    int bpp = (kept + O->passes - 1) / O->passes;
    if (bpp < 1) bpp = 1;
    if (bpp > NM_ORDER_PASS_BITS) bpp = NM_ORDER_PASS_BITS;
    O->bpp = bpp;
    O->valid = 1;
}

constexpr int NM_MAX_LADDER = 32;     // scales per ladder call
#ifndef NM_DENSE_LOG2_VALUE
#define NM_DENSE_LOG2_VALUE 21
#endif
constexpr int NM_DENSE_LOG2 = NM_DENSE_LOG2_VALUE;     // a scale with at most 2^21 superblocks (and no more than there are points) is indexed densely: up to 512 MB of leaves, zeroed per call at HBM speed - cheaper than the insert protocol (measured at 2^18 and 2^21)

#if defined(__HIPCC__)

// ---- exact lattice arithmetic (device) ---------------------------------------------------------------
// floor((p - min_corner) / e) exactly as geometry.py:108: fp64 subtract, IEEE divide, floor.
// compiled with -ffp-contract=off so nothing fuses.
__device__ __forceinline__ double nm_cell_f(double p, double mn, double e) { return floor((p - mn) / e); }

// the same cell with the division replaced by a multiplication wherever that provably gives the same floor,
// as an int32 clamped to [lo, hi] (|lo|, |hi| <= 2^30).
// with x = p - mn (the reference's own subtraction), t = x / e, the reference floors fl(t), |fl(t) - t| <= u|t|
// (u = 2^-53); q = fl(x * fl(1/e)) has |q - t| <= (2u + u^2)|t|.  so q and fl(t) lie within 3.1 u |q| of each
// other, and their floors can differ only if an integer lies that close to q.
//   * "far enough from every integer" is decided on the high word of q's fraction: at least 2^-19 from the nearest
//     integer covers 3.1 u |q| for every |q| < 2^31; beyond that both floors lie outside [lo, hi] on the same side
//     and the clamp makes them equal.  otherwise (also for NaN and infinite q, whose fraction is NaN and fails the
//     test) the reference's division decides.  lattice-aligned clouds take the slow path for every point; clouds
//     in general position 4 times in 10^6 points and axes.  (rounds 1-2 tested |q - rint(q)| > |q| 2^-51: one
//     instruction more.)
//   * v_cvt_i32_f64 saturates: one conversion and one v_med3_i32 replace two fp64 min/max, a canonicalisation and
//     two integer min/max.  (inline assembly: in C++ the out-of-range cast is undefined.)
__device__ __forceinline__ int32_t nm_cell_index(double p, double mn, double e, double inv_e, int32_t lo,
                                                 int32_t hi)
{
    const double x = p - mn;
    const double q = x * inv_e;
    double c = floor(q);
    const double fr = __builtin_amdgcn_fract(q);
    const uint32_t fh = (uint32_t)__double2hiint(fr);
    // 2^-19 <= fr < 1 - 2^-19   (high words 0x3EC00000 and 0x3FEFFFFC, low words zero)
    if (!(fh - 0x3EC00000u < 0x3FEFFFFCu - 0x3EC00000u)) {
        c = floor(x / e);
        if (!(c == c)) c = -1073741824.0;       // NaN: what nm_clamp_cell's fmax/fmin chain made of it
    }
    int32_t ci;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(ci) : "v"(c));
    return min(max(ci, lo), hi);       // (one v_med3_i32 where the compiler can see lo <= hi)
}

// voxel centre exactly as geometry.py:137: (cell * e + min_corner) + e*0.5, left to right
__device__ __forceinline__ double nm_centre(int32_t cell, double mn, double e, double he)
{
    double t = (double)cell * e;
    t = t + mn;
    return t + he;
}

__device__ __forceinline__ int32_t nm_clamp_cell(double c)
{
    // cells of points far outside the lattice are clamped; the callers treat anything outside
    // [0, 2^w) as "no such cell"
    c = fmin(fmax(c, -1073741824.0), 1073741823.0);
    return (int32_t)c;
}

// sort key: [sbz | sbx | sby | lz:3 | ly:3 | lx:5].  consecutive keys run along x inside a leaf; the
// next leaf in key order is the y neighbour, so a run of queries that straddles two leaves stays
// narrow in x.
__device__ __forceinline__ uint64_t nm_sb_key(uint32_t sbx, uint32_t sby, uint32_t sbz,
                                              const LatticeDev& L)
{
    return ((uint64_t)sbz << (L.bx + L.by)) | ((uint64_t)sbx << L.by) | (uint64_t)sby;
}

__device__ __forceinline__ uint64_t nm_cell_key(uint32_t cx, uint32_t cy, uint32_t cz,
                                                const LatticeDev& L)
{
    uint64_t sb = nm_sb_key(cx >> NM_SBX_BITS, cy >> NM_SBY_BITS, cz >> NM_SBZ_BITS, L);
    uint32_t local = ((cz & 7u) << 8) | ((cy & 7u) << 5) | (cx & 31u);
    return (sb << NM_LOCAL_BITS) | local;
}

// acc[c] += the class distribution of leaf `row` (ForestDev::leaf_value).  rows of 8 doubles are 64-byte aligned
// and zero behind n_classes: 16-byte loads, as many as the classes need, one cache line per lane
template <int MAXC>
__device__ __forceinline__ void nm_forest_vote(const double* __restrict__ leaf_value, int32_t stride, int32_t nc,
                                               uint32_t row, double* acc)
{
    static_assert(MAXC >= 8, "the padded form adds up to eight classes");
    if (stride == 8) {          // (the entry points check: at most 8 classes, rows 64-byte aligned)
        const double2* v2 = (const double2*)__builtin_assume_aligned(leaf_value + (int64_t)row * 8, 64);
        const int pairs = (nc + 1) >> 1;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (p < pairs) {
                const double2 v = v2[p];
                acc[2 * p] += v.x;
                acc[2 * p + 1] += v.y;
            }
        return;
    }
    const double* val = leaf_value + (int64_t)row * stride;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (c < nc) acc[c] += val[c];
}

// part `part` of `parts` of the per-call reset of every index of a ladder (dense: leaves zeroed, all of them count as
// allocated; map and hash forms: the directory to "empty"; counter blocks to zero), done by `workers` threads of
// which this is number `worker`.  one kernel of its own (k_index_clear_all), or thirds of it ridden along by the
// blocks appended to the spatial sort's three counting kernels, whose bandwidth is idle (nm_order.hip)
__device__ __forceinline__ void nm_index_clear_part(const ScaleDev* __restrict__ ladder, int32_t n_scales,
                                                    uint64_t worker, uint64_t workers, int part, int parts)
{
    for (int32_t sc = 0; sc < n_scales; ++sc) {
        if (ladder[sc].shared) {
            // a borrowed index is cleared by its owner; the borrower's own counter block holds its statistics
            if (part == 0 && worker < 64) ladder[sc].stats[worker] = 0u;
            continue;
        }
        const IndexDev I = ladder[sc].I;
        uint4* base;
        uint64_t quads;
        uint4 fill;
        if (!I.hash) {
            base = (uint4*)I.leaf;
            quads = ((uint64_t)I.hash_mask + 1ull) * (NM_LEAF_WORDS / 4);
            fill = make_uint4(0u, 0u, 0u, 0u);
        } else {
            // (map form: a word per superblock - a quarter of a quad each)
            base = (uint4*)I.hash;
            quads = I.map ? ((uint64_t)I.hash_mask + 4ull) / 4ull : (uint64_t)I.hash_mask + 1ull;
            fill = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        }
        const uint64_t lo = quads * (uint64_t)part / (uint64_t)parts, hi = quads * (uint64_t)(part + 1) / (uint64_t)parts;
        for (uint64_t i = lo + worker; i < hi; i += workers) base[i] = fill;
        if (part == 0 && worker < 64) I.counters[worker] = (!I.hash && worker == 0) ? I.hash_mask + 1u : 0u;
    }
}

// XCD-aware block -> batch mapping: workgroups are dealt round-robin over the 8 XCDs, so give every
// XCD one contiguous eighth of the batches; neighbouring batches then share that XCD's L2.  bijective for
// any grid size.
__device__ __forceinline__ int64_t nm_xcd_batch(int64_t b, int64_t nb)
{
    int64_t xcd = b & 7, q = nb >> 3, r = nb & 7;
    int64_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (b >> 3);
}

__device__ __forceinline__ uint32_t nm_hash64(uint64_t k)
{
    uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
    uint32_t h = lo * 0x9E3779B1u ^ (hi + 0x7F4A7C15u) * 0x85EBCA6Bu;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 13;
    return h;
}

// leaf number of a superblock, or -1
__device__ __forceinline__ int32_t nm_hash_find(const IndexDev& I, uint64_t key)
{
    if (!I.hash) return key <= (uint64_t)I.hash_mask ? (int32_t)key : -1;      // dense: leaf = superblock
    if (I.map) {                                                               // map: a word per superblock
        if (key > (uint64_t)I.hash_mask) return -1;
        const uint32_t v = I.map[key];
        return (int32_t)v >= 0 ? (int32_t)v : -1;
    }
    uint32_t slot = nm_hash64(key) & I.hash_mask;
    for (;;) {
        const uint4 e = *(const uint4*)&I.hash[slot];          // one 16-byte load: key and value
        const uint64_t k = (uint64_t)e.x | ((uint64_t)e.y << 32);
        if (k == key) return (int32_t)e.z;
        if (k == NM_HASH_EMPTY) return -1;
        slot = (slot + 1) & I.hash_mask;
    }
}

#endif  // __HIPCC__
