// nm_api.hip - context management and the classifier slot (random-forest evaluation).
//
// reference code replaced by nm_forest_eval: sklearn RandomForestClassifier.predict / predict_proba
// at nimrud/prototypes/apc.py:1008,1022,1034,1745,1752 (nimrud/minimal/classification.py is a stub).

#include "nm_common.h"

extern "C" int nm_abi_version(void) { return 6; }

// nm_create has no context to leave a message in yet: what failed goes to stderr
#define NM_CREATE_TRY(call)                                                                       \
    do {                                                                                          \
        const hipError_t _e = (call);                                                             \
        if (_e != hipSuccess) {                                                                   \
            fprintf(stderr, "nm_create: %s failed: %s\n", #call, hipGetErrorString(_e));          \
            if (ctx) nm_destroy(ctx);                                                             \
            return NM_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

extern "C" int nm_create(nm_ctx** out, int device)
{
    if (!out) return NM_ERR_INVALID;
    *out = nullptr;
    nm_ctx* ctx = nullptr;
    int count = 0;
    NM_CREATE_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) {
        fprintf(stderr, "nm_create: device %d of %d\n", device, count);
        return NM_ERR_HIP;
    }
    // the caller's current device is left as it was (torch reads the same process-wide setting)
    nm_device_guard guard(device);
    int now = -1;
    NM_CREATE_TRY(hipGetDevice(&now));
    if (now != device) return NM_ERR_HIP;
    hipDeviceProp_t prop;
    NM_CREATE_TRY(hipGetDeviceProperties(&prop, device));
    ctx = new nm_ctx();
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount;
    // the sticky status words: the only memory the library owns - 256 B of pinned host memory that the kernels
    // see through its device address.  a kernel writes there only when something is wrong, so nothing has to
    // be copied back after a call (the copy of a device-side block was a 5 us kernel behind every step)
    NM_CREATE_TRY(hipHostMalloc((void**)&ctx->h_status, NM_ST_WORDS * 4, hipHostMallocMapped));
    for (int i = 0; i < NM_ST_WORDS; ++i) ctx->h_status[i] = 0u;
    NM_CREATE_TRY(hipHostGetDevicePointer((void**)&ctx->d_status, ctx->h_status, 0));
    NM_CREATE_TRY(hipEventCreateWithFlags(&ctx->status_event, hipEventDisableTiming));
    *out = ctx;
    return NM_OK;
}

// ---- asynchronous failures ---------------------------------------------------------------------------
// kernels report what the host cannot know when it enqueues them (an index build that timed out, a
// lattice built on the device that cannot be addressed) by setting a word of ctx->d_status.  every
// feature call ends with a snapshot of those words into pinned memory; whoever enters the library next
// (or calls nm_check) sees a completed snapshot and gets the failure as a status code.  d_status is
// never cleared by a kernel, so nothing is lost when snapshots overtake each other.
void nm_status_snapshot(nm_ctx* ctx, hipStream_t s)
{
    if (!ctx->d_status) return;
    // the status words live in host memory: what is left to do behind a call is to mark its end.  inside a
    // hipGraph capture no event is recorded (querying a captured event is illegal): nm_check(ctx, 1) after the
    // caller has synchronised the replay reads the words as they stand
    if (nm_capturing(s)) return;
    if (hipEventRecord(ctx->status_event, s) == hipSuccess) ctx->status_pending = true;
}

int nm_status_poll(nm_ctx* ctx, bool wait)
{
    if (ctx->sticky) return ctx->sticky;
    if (ctx->status_pending) {
        if (wait) {
            if (hipEventSynchronize(ctx->status_event) != hipSuccess) return NM_OK;
        } else if (hipEventQuery(ctx->status_event) != hipSuccess) {
            return NM_OK;          // not there yet: look again next time
        }
        ctx->status_pending = false;
    } else if (!wait) {
        return NM_OK;              // nothing outstanding that this call could know about
    }
    // (wait without a pending event: the caller has synchronised a graph replay)
    const uint32_t* st = ctx->h_status;
    if (st[NM_ST_LATTICE]) {
        switch (st[NM_ST_LATTICE]) {
            case NM_LAT_TOO_SMALL_EDGE: ctx->error = "edge length is too small to address this space"; break;
            case NM_LAT_NO_EXTENT: ctx->error = "cloud has no extent beyond one voxel on some axis"; break;
            case NM_LAT_DEVICE_LIMIT:
                ctx->error = "an address width is outside the device path's range [1,30]";
                break;
            default: ctx->error = "cloud extrema are not finite"; break;
        }
        ctx->sticky = NM_ERR_LATTICE;
    } else if (st[NM_ST_INDEX_TIMEOUT]) {
        ctx->error = "occupancy index build timed out waiting for a leaf number: the features of an "
                     "earlier call are incomplete";
        ctx->sticky = NM_ERR_HIP;
    } else if (st[NM_ST_LEAF_OVERFLOW]) {
        ctx->error = "occupancy index ran out of leaves: the features of an earlier call are incomplete";
        ctx->sticky = NM_ERR_HIP;
    }
    return ctx->sticky;
}

extern "C" int nm_check(nm_ctx* ctx, int wait)
{
    if (!ctx) return NM_ERR_INVALID;
    nm_device_guard guard(ctx->device);
    return nm_status_poll(ctx, wait != 0);
}

extern "C" int nm_clear_error(nm_ctx* ctx)
{
    if (!ctx) return NM_ERR_INVALID;
    nm_device_guard guard(ctx->device);
    // outstanding work may still set a word: drain the device first, then start clean
    NM_HIP(ctx, hipDeviceSynchronize());
    for (int i = 0; i < NM_ST_WORDS; ++i) ctx->h_status[i] = 0u;
    ctx->status_pending = false;
    ctx->sticky = 0;
    ctx->error.clear();
    return NM_OK;
}

extern "C" void nm_destroy(nm_ctx* ctx)
{
    if (!ctx) return;
    nm_device_guard guard(ctx->device);
    if (ctx->status_event) (void)hipEventDestroy(ctx->status_event);
    if (ctx->h_status) (void)hipHostFree(ctx->h_status);      // (d_status is its device address)
    for (hipEvent_t e : ctx->events) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->sync_events) (void)hipEventDestroy(e);
    if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
    delete ctx;
}

extern "C" int nm_set_overlap(nm_ctx* ctx, int enabled)
{
    if (!ctx) return NM_ERR_INVALID;
    ctx->overlap = enabled != 0;
    return NM_OK;
}

extern "C" int nm_set_knn_fallback(nm_ctx* ctx, int k_min, double radius_factor)
{
    if (!ctx) return NM_ERR_INVALID;
    if (k_min < 0 || k_min > 16 || (k_min > 0 && !(radius_factor >= 1.0)))
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_set_knn_fallback: k_min in [0,16], radius_factor >= 1");
    ctx->knn_k = k_min;
    if (k_min > 0) ctx->knn_radius_factor = radius_factor;
    return NM_OK;
}

extern "C" int nm_set_covariance_output(nm_ctx* ctx, double* d_cov, int64_t cov_stride)
{
    if (!ctx) return NM_ERR_INVALID;
    if (d_cov && cov_stride < 6)
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_set_covariance_output: cov_stride must be at least 6");
    ctx->cov_out = d_cov;
    ctx->cov_stride = d_cov ? cov_stride : 0;
    return NM_OK;
}

extern "C" int nm_set_normal_output(nm_ctx* ctx, double* d_normal, int64_t normal_stride)
{
    if (!ctx) return NM_ERR_INVALID;
    if (d_normal && normal_stride < 3)
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_set_normal_output: normal_stride must be at least 3");
    ctx->normal_out = d_normal;
    ctx->normal_stride = d_normal ? normal_stride : 0;
    return NM_OK;
}

extern "C" int nm_set_fuse_scales(nm_ctx* ctx, int enabled)
{
    if (!ctx) return NM_ERR_INVALID;
    ctx->fuse_scales = enabled != 0;
    return NM_OK;
}

extern "C" int nm_set_forest_mode(nm_ctx* ctx, int in_search_kernel)
{
    if (!ctx) return NM_ERR_INVALID;
    ctx->forest_epilogue = in_search_kernel == 1;
    ctx->forest_mode = in_search_kernel;
    return NM_OK;
}

extern "C" int nm_set_forest_output(nm_ctx* ctx, const nm_forest* forest, double* d_proba,
                                    int64_t proba_stride, int32_t* d_label)
{
    if (!ctx) return NM_ERR_INVALID;
    if (!forest) {
        ctx->forest_on = false;
        return NM_OK;
    }
    if (!forest->d_packed8 || !forest->d_leaf_value || !forest->d_packed_roots)
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_set_forest_output: the forest needs the 8-byte node layout "
                                     "(d_packed8, d_leaf_value, d_packed_roots)");
    if (forest->n_features < 1 || forest->n_features > NM_FUSED_FOREST_FEATURES ||
        (forest->n_features & 3) || forest->n_classes < 1 ||
        forest->n_classes > NM_FUSED_FOREST_CLASSES || forest->n_trees < 1)
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_set_forest_output: behind the ladder a forest may have up to %d "
                                     "features (4 per scale) and %d classes; evaluate larger ones with "
                                     "nm_forest_eval", NM_FUSED_FOREST_FEATURES, NM_FUSED_FOREST_CLASSES);
    if ((!d_proba && !d_label) || (d_proba && proba_stride < forest->n_classes))
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_set_forest_output: bad output arguments");
    if (forest->leaf_stride && (forest->leaf_stride < forest->n_classes ||
                                (forest->leaf_stride == 8 && ((uintptr_t)forest->d_leaf_value & 63))))
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_set_forest_output: leaf_stride below n_classes, or rows of 8 doubles that "
                                     "are not 64-byte aligned");
    ForestDev F;
    F.nodes = (const uint2*)forest->d_packed8;
    F.leaf_value = forest->d_leaf_value;
    F.leaf_stride = forest->leaf_stride ? forest->leaf_stride : forest->n_classes;
    F.roots = forest->d_packed_roots;
    F.n_trees = forest->n_trees;
    F.n_classes = forest->n_classes;
    F.n_features = forest->n_features;
    F.n_nodes = forest->n_nodes;
    F.proba = d_proba;
    F.pstride = proba_stride;
    F.label = d_label;
    ctx->forest = F;
    ctx->forest_features = forest->n_features;
    ctx->forest_on = true;
    return NM_OK;
}

extern "C" int nm_profile_begin(nm_ctx* ctx)
{
    if (!ctx) return NM_ERR_INVALID;
    ctx->profiling = true;
    ctx->events_used = 0;
    ctx->profile_launches = 0;
    return NM_OK;
}

extern "C" int nm_profile_end(nm_ctx* ctx, double* ms, int64_t* launches)
{
    if (!ctx) return NM_ERR_INVALID;
    nm_device_guard guard(ctx->device);
    ctx->profiling = false;
    double acc[4] = {0, 0, 0, 0};
    int64_t calls = (int64_t)(ctx->events_used / 4);
    for (int64_t k = 0; k < calls; ++k) {
        NM_HIP(ctx, hipEventSynchronize(ctx->events[4 * k + 3]));
        for (int st = 0; st < 3; ++st) {
            float t = 0.f;
            NM_HIP(ctx, hipEventElapsedTime(&t, ctx->events[4 * k + st], ctx->events[4 * k + st + 1]));
            acc[st] += (double)t;
        }
    }
    if (ms) for (int i = 0; i < 4; ++i) ms[i] = acc[i];
    if (launches) *launches = ctx->profile_launches;
    ctx->events_used = 0;
    return NM_OK;
}

extern "C" const char* nm_last_error(const nm_ctx* ctx) { return ctx ? ctx->error.c_str() : "null context"; }

// one lane per point; all lanes of a wave walk the same tree at the same time, so node fetches of the
// upper levels are wave-uniform and cache resident.  proba is accumulated in tree order, like
// sklearn's accumulate-then-divide.
constexpr int NM_MAX_CLASSES = 16;

__global__ __launch_bounds__(256) void k_forest_eval(nm_forest F, const double* __restrict__ feat,
                                                     int64_t n, int64_t fstride,
                                                     double* __restrict__ proba,
                                                     int32_t* __restrict__ label)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* x = feat + i * fstride;
    double acc[NM_MAX_CLASSES];
#pragma unroll
    for (int c = 0; c < NM_MAX_CLASSES; ++c) acc[c] = 0.0;
    for (int t = 0; t < F.n_trees; ++t) {
        int32_t node = F.d_roots[t];
        for (;;) {
            const int32_t left = F.d_left[node];
            if (left < 0) break;
            // sklearn casts X to float32 and compares against the float64 threshold
            const double v = (double)(float)x[F.d_feature[node]];
            node = (v <= F.d_threshold[node]) ? left : F.d_right[node];
        }
        const double* val = F.d_value + (int64_t)node * F.n_classes;
#pragma unroll
        for (int c = 0; c < NM_MAX_CLASSES; ++c)
            if (c < F.n_classes) acc[c] += val[c];
    }
    int best = 0;
    double bestv = -1.0;
    const double inv = 1.0 / (double)F.n_trees;
#pragma unroll
    for (int c = 0; c < NM_MAX_CLASSES; ++c) {
        if (c < F.n_classes) {
            const double pr = acc[c] / (double)F.n_trees;
            (void)inv;
            if (proba) proba[i * F.n_classes + c] = pr;
            if (pr > bestv) {   // first maximum wins, like numpy.argmax
                bestv = pr;
                best = c;
            }
        }
    }
    if (label) label[i] = best;
}

// packed-node evaluator: the block's 256 feature rows are staged in LDS as fp32 (sklearn casts X to
// float32 anyway), transposed so that lane t reads bank t % 32 whatever feature it needs; a node visit
// is one 16-byte record load and one LDS read.
constexpr int NM_FOREST_MAX_FEATURES = 40;
constexpr int NM_FOREST_TREES = 8;     // trees descended side by side per thread

struct PackedNode {
    double threshold;
    int32_t left;
    int32_t feature;
};

__global__ __launch_bounds__(256) void k_forest_eval_packed(nm_forest F, const double* __restrict__ feat,
                                                            int64_t n, int64_t fstride,
                                                            double* __restrict__ proba,
                                                            int32_t* __restrict__ label)
{
    __shared__ float xs[NM_FOREST_MAX_FEATURES * 256];
    const int64_t row0 = (int64_t)blockIdx.x * 256;
    const int nf = F.n_features;
    const int rows_here = (int)((n - row0) < 256 ? (n - row0) : 256);
    // rows are fstride apart; consecutive threads read consecutive columns of one row
    for (int r = threadIdx.x / 32; r < rows_here; r += 8) {
        const double* src = feat + (row0 + r) * fstride;
        for (int f = threadIdx.x % 32; f < nf; f += 32) xs[f * 256 + r] = (float)src[f];
    }
    __syncthreads();
    const int64_t i = row0 + threadIdx.x;
    if (i >= n) return;
    const PackedNode* nodes = (const PackedNode*)F.d_packed;
    double acc[NM_MAX_CLASSES];
#pragma unroll
    for (int c = 0; c < NM_MAX_CLASSES; ++c) acc[c] = 0.0;
    // a descent is a chain of dependent record loads (L2 latency each): NM_FOREST_TREES trees go down
    // side by side so that their loads overlap (5.9 -> 3.6 ms for 10 M rows x 32 trees of depth 12;
    // keeping the top levels of the trees in LDS on top of that bought nothing).  the votes are still
    // added in tree order.
    for (int t0 = 0; t0 < F.n_trees; t0 += NM_FOREST_TREES) {
        PackedNode rec[NM_FOREST_TREES];
#pragma unroll
        for (int g = 0; g < NM_FOREST_TREES; ++g) {
            const int t = t0 + g < F.n_trees ? t0 + g : F.n_trees - 1;
            rec[g] = nodes[F.d_packed_roots[t]];
        }
        bool any = true;
        while (any) {
            any = false;
#pragma unroll
            for (int g = 0; g < NM_FOREST_TREES; ++g) {
                if (rec[g].left >= 0) {
                    const double v = (double)xs[rec[g].feature * 256 + threadIdx.x];
                    rec[g] = nodes[rec[g].left + (v <= rec[g].threshold ? 0 : 1)];
                    any = true;
                }
            }
        }
#pragma unroll
        for (int g = 0; g < NM_FOREST_TREES; ++g) {
            if (t0 + g >= F.n_trees) break;
            nm_forest_vote<NM_MAX_CLASSES>(F.d_leaf_value, F.leaf_stride ? F.leaf_stride : F.n_classes, F.n_classes,
                                           (uint32_t)rec[g].feature, acc);
        }
    }
    int best = 0;
    double bestv = -1.0;
#pragma unroll
    for (int c = 0; c < NM_MAX_CLASSES; ++c) {
        if (c < F.n_classes) {
            const double pr = acc[c] / (double)F.n_trees;
            if (proba) proba[i * F.n_classes + c] = pr;
            if (pr > bestv) {
                bestv = pr;
                best = c;
            }
        }
    }
    if (label) label[i] = best;
}

// the same evaluator on 8-byte nodes {fp32 threshold rounded down, packed} (nm_forest::d_packed8): half the
// bytes per node visit.  x_f32 <= threshold_f64  <=>  x_f32 <= largest fp32 not above the threshold.
constexpr int NM_FOREST8_MAX_FEATURES = 32;     // 5-bit feature field

__global__ __launch_bounds__(256) void k_forest_eval_packed8(ForestDev F, const double* __restrict__ feat,
                                                             int64_t n, int64_t fstride)
{
    __shared__ float xs[NM_FOREST8_MAX_FEATURES * 256];
    const int64_t row0 = (int64_t)blockIdx.x * 256;
    const int nf = F.n_features;
    const int rows_here = (int)((n - row0) < 256 ? (n - row0) : 256);
    for (int r = threadIdx.x / 32; r < rows_here; r += 8) {
        const double* src = feat + (row0 + r) * fstride;
        for (int f = threadIdx.x % 32; f < nf; f += 32) xs[f * 256 + r] = (float)src[f];
    }
    __syncthreads();
    const int64_t i = row0 + threadIdx.x;
    if (i >= n) return;
    double acc[NM_MAX_CLASSES];
#pragma unroll
    for (int c = 0; c < NM_MAX_CLASSES; ++c) acc[c] = 0.0;
    for (int t0 = 0; t0 < F.n_trees; t0 += NM_FOREST_TREES) {
        uint2 rec[NM_FOREST_TREES];
#pragma unroll
        for (int g = 0; g < NM_FOREST_TREES; ++g) {
            const int t = t0 + g < F.n_trees ? t0 + g : F.n_trees - 1;
            rec[g] = F.nodes[F.roots[t]];
        }
        bool any = true;
        while (any) {
            any = false;
#pragma unroll
            for (int g = 0; g < NM_FOREST_TREES; ++g) {
                if (!(rec[g].y >> 31)) {
                    const float v = xs[((rec[g].y >> 8) & 31u) * 256 + threadIdx.x];
                    rec[g] = F.nodes[(rec[g].y >> 13) + (v <= __uint_as_float(rec[g].x) ? 0u : 1u)];
                    any = true;
                }
            }
        }
#pragma unroll
        for (int g = 0; g < NM_FOREST_TREES; ++g) {
            if (t0 + g >= F.n_trees) break;
            nm_forest_vote<NM_MAX_CLASSES>(F.leaf_value, F.leaf_stride, F.n_classes, (rec[g].y >> 13) & 0x3FFFFu, acc);
        }
    }
    int best = 0;
    double bestv = -1.0;
#pragma unroll
    for (int c = 0; c < NM_MAX_CLASSES; ++c) {
        if (c < F.n_classes) {
            const double pr = acc[c] / (double)F.n_trees;
            if (F.proba) F.proba[i * F.pstride + c] = pr;
            if (pr > bestv) {
                bestv = pr;
                best = c;
            }
        }
    }
    if (F.label) F.label[i] = best;
}

// ---- the forest with its trees staged through LDS ------------------------------------------------------------
// the row-walking evaluators above fetch every node from memory: 64 rows that are neighbours in space part ways a
// few levels down, a wave-level fetch touches ~30 cache lines, and the walk is bound by the L1's lookup rate
// (DESIGN.md 3b).  here a 1024-thread workgroup keeps 1024 rows' features in LDS ([wave][feature][lane]: a lane
// reads its own column, conflict-free) and streams the trees through two LDS buffers, two at a time: a descent
// step is an LDS read of the node and an LDS read of the feature - one address space, ~100 cycles a step instead
// of ~700.  a tree's children are adjacent (breadth-first numbering, ForestModel.pack_nodes), so its nodes are
// one contiguous run of the node array; the staging rebases the child index on the tree's first node.  trees
// that do not fit a buffer (more than FT_TREE_CAP nodes) are walked from memory as before, pair by pair.
// votes are added in tree order, as everywhere.
constexpr int FT_THREADS = 1024;
constexpr int FT_TREES = 4;              // trees in LDS at a time, walked side by side by every thread
constexpr int FT_TREE_CAP = 2048;        // nodes per LDS tree buffer (16 KB)
constexpr int FT_STAGE = FT_TREES * FT_TREE_CAP / FT_THREADS;     // nodes a thread carries from memory to LDS
constexpr int FT_MAX_FEATURES = 21;      // 16 waves x 21 x 256 B = 84 KB of feature columns

__global__ __launch_bounds__(FT_THREADS) void k_forest_tiles(ForestDev F, const double* __restrict__ feat,
                                                             int64_t n, int64_t fstride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ft_lds[];
    uint2* tb = (uint2*)ft_lds;                                   // [FT_TREES][FT_TREE_CAP]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nf = F.n_features, nc = F.n_classes;
    char* xw = (char*)(ft_lds + FT_TREES * FT_TREE_CAP * sizeof(uint2)) + (size_t)w * nf * 256;   // [feature][lane]
    const uint32_t lane4 = (uint32_t)lane * 4u;
    const uint2* __restrict__ nodes = F.nodes;
    const int n_stages = (F.n_trees + FT_TREES - 1) / FT_TREES;

    // what a thread carries from memory into the buffers for stage `st`: node k of the flat range
    // [tree 0's nodes | tree 1's | ...] of that stage, FT_STAGE of them, rebased on their tree's first node
    auto fetch = [&](int st, uint2* regs, bool& fit) {
        fit = true;
#pragma unroll
        for (int g = 0; g < FT_TREES; ++g) {
            const int t = st * FT_TREES + g < F.n_trees ? st * FT_TREES + g : F.n_trees - 1;
            const int32_t r = F.roots[t];
            const int32_t sz = (t + 1 < F.n_trees ? F.roots[t + 1] : F.n_nodes) - r;
            fit = fit && sz <= FT_TREE_CAP && sz >= 3;
#pragma unroll
            for (int q = 0; q < FT_TREE_CAP / FT_THREADS; ++q) {
                const int k = tid + q * FT_THREADS;
                uint2 nd = make_uint2(0x7FC00000u, 1u);
                if (k < sz && sz <= FT_TREE_CAP && sz >= 3) {
                    nd = nodes[r + k];
                    // the LDS form of a node: y = byte offset of the left child inside the buffer << 16 |
                    // feature << 8; a leaf is a node that steps to itself whatever the row holds - a NaN
                    // threshold (no value is <= it: the step goes "right") with the leaf's row in its payload,
                    // and the node before itself as its "left" child - so the walk needs no test per step
                    if ((int32_t)nd.y >= 0) {
                        nd.y = ((((nd.y >> 13) - (uint32_t)r) * 8u) << 16) | (nd.y & 0x1F00u);
                    } else {
                        nd.x = 0x7FC00000u | ((nd.y >> 13) & 0x3FFFFu);
                        nd.y = (((uint32_t)k - 1u) * 8u) << 16 | 1u;
                    }
                }
                regs[g * (FT_TREE_CAP / FT_THREADS) + q] = nd;
            }
        }
    };

    for (int64_t chunk = blockIdx.x; chunk * FT_THREADS < n; chunk += gridDim.x) {
        const int64_t i = chunk * FT_THREADS + tid;
        const bool have = i < n;
        bool undefined = false;
        if (have) {
            const double* row = feat + i * fstride;
            for (int f = 0; f < nf; ++f) {
                const double v = row[f];
                undefined = undefined || v != v;
                *(float*)(xw + f * 256 + lane4) = v != v ? 0.0f : (float)v;    // (a NaN would step off a leaf)
            }
        } else {
            for (int f = 0; f < nf; ++f) *(float*)(xw + f * 256 + lane4) = 0.0f;
        }
        double acc[NM_FUSED_FOREST_CLASSES];
#pragma unroll
        for (int c = 0; c < NM_FUSED_FOREST_CLASSES; ++c) acc[c] = 0.0;
        uint2 carry[FT_STAGE];
        bool fit_next;
        fetch(0, carry, fit_next);
        double vote[FT_TREES][NM_FUSED_FOREST_CLASSES];
        int votes_pending = 0;               // trees of the previous stage whose leaf rows are on their way
        for (int st = 0; st < n_stages; ++st) {
            const bool fit = fit_next;
            __syncthreads();       // the previous stage's walks are over: the buffers are free
#pragma unroll
            for (int g = 0; g < FT_TREES; ++g)
#pragma unroll
                for (int q = 0; q < FT_TREE_CAP / FT_THREADS; ++q)
                    tb[g * FT_TREE_CAP + tid + q * FT_THREADS] = carry[g * (FT_TREE_CAP / FT_THREADS) + q];
            __syncthreads();
            // the previous stage's votes have had a staging's time to arrive: add them, in tree order
#pragma unroll
            for (int g = 0; g < FT_TREES; ++g)
                if (g < votes_pending)
#pragma unroll
                    for (int c = 0; c < NM_FUSED_FOREST_CLASSES; ++c)
                        if (c < nc) acc[c] += vote[g][c];
            // the next stage's nodes set out now and travel during this stage's walk
            if (st + 1 < n_stages) fetch(st + 1, carry, fit_next);
            uint2 cur[FT_TREES];
            if (fit) {
#pragma unroll
                for (int g = 0; g < FT_TREES; ++g) cur[g] = tb[g * FT_TREE_CAP];
                // every lane steps every tree, all the time (a leaf steps to itself): the four feature reads of
                // a level go out together, then the four node reads - two LDS round trips per level, not eight
                for (;;) {
                    uint32_t done = cur[0].y;
#pragma unroll
                    for (int g = 1; g < FT_TREES; ++g) done &= cur[g].y;
                    if (__all((done & 1u) != 0u)) break;
                    float v[FT_TREES];
#pragma unroll
                    for (int g = 0; g < FT_TREES; ++g)
                        v[g] = *(const float*)(xw + ((cur[g].y & 0x1F00u) | lane4));
#pragma unroll
                    for (int g = 0; g < FT_TREES; ++g) {
                        const uint32_t at = (cur[g].y >> 16) + (v[g] <= __uint_as_float(cur[g].x) ? 0u : 8u);
                        cur[g] = *(const uint2*)((const char*)(tb + g * FT_TREE_CAP) + at);
                    }
                }
                // back to the memory form of a leaf for the votes below
#pragma unroll
                for (int g = 0; g < FT_TREES; ++g) cur[g].y = 0x80000000u | ((cur[g].x & 0x3FFFFu) << 13);
            } else {
                // a tree of this stage is too large for its buffer: the four are walked from memory
#pragma unroll
                for (int g = 0; g < FT_TREES; ++g) {
                    const int t = st * FT_TREES + g < F.n_trees ? st * FT_TREES + g : F.n_trees - 1;
                    cur[g] = nodes[F.roots[t]];
                }
                for (;;) {
                    bool go = false;
#pragma unroll
                    for (int g = 0; g < FT_TREES; ++g) go = go || (int32_t)cur[g].y >= 0;
                    if (!__any(go)) break;
#pragma unroll
                    for (int g = 0; g < FT_TREES; ++g) {
                        if ((int32_t)cur[g].y >= 0) {
                            const float v = *(const float*)(xw + ((cur[g].y & 0x1F00u) | lane4));
                            cur[g] = nodes[(cur[g].y >> 13) + (v <= __uint_as_float(cur[g].x) ? 0u : 1u)];
                        }
                    }
                }
            }
            votes_pending = F.n_trees - st * FT_TREES < FT_TREES ? F.n_trees - st * FT_TREES : FT_TREES;
#pragma unroll
            for (int g = 0; g < FT_TREES; ++g) {
#pragma unroll
                for (int c = 0; c < NM_FUSED_FOREST_CLASSES; ++c) vote[g][c] = 0.0;
                nm_forest_vote<NM_FUSED_FOREST_CLASSES>(F.leaf_value, F.leaf_stride, nc, (cur[g].y >> 13) & 0x3FFFFu,
                                                        vote[g]);
            }
        }
#pragma unroll
        for (int g = 0; g < FT_TREES; ++g)
            if (g < votes_pending)
#pragma unroll
                for (int c = 0; c < NM_FUSED_FOREST_CLASSES; ++c)
                    if (c < nc) acc[c] += vote[g][c];
        if (have) {
            if (undefined) {     // a scale without an addressable lattice left NaN columns: no class
                if (F.proba)
                    for (int c = 0; c < nc; ++c) F.proba[i * F.pstride + c] = __builtin_nan("");
                if (F.label) F.label[i] = -1;
            } else {
                int best = 0;
                double bestv = -1.0;
#pragma unroll
                for (int c = 0; c < NM_FUSED_FOREST_CLASSES; ++c) {
                    if (c < nc) {
                        const double pr = acc[c] / (double)F.n_trees;
                        if (F.proba) F.proba[i * F.pstride + c] = pr;
                        if (pr > bestv) {   // first maximum wins, like numpy.argmax
                            bestv = pr;
                            best = c;
                        }
                    }
                }
                if (F.label) F.label[i] = best;
            }
        }
    }
}

int nm_forest_rows(nm_ctx* ctx, const ForestDev& F, const double* d_feat, int64_t feat_stride, int64_t n,
                   int32_t n_features, hipStream_t s)
{
    if (n <= 0) return NM_OK;
    ForestDev G = F;
    G.n_features = n_features;
    // rows in no particular order (the stand-alone evaluator, nm_set_forest_mode(2)): trees through LDS - 3.0 ms
    // for 10 M rows x 32 trees against 3.5 for the row walk from memory.  (behind a ladder the rows come in
    // spatial order and the walk from memory is the faster one - 2.5 ms, k_forest_ordered - and inside the search
    // kernel faster still; tools/forest_modes.py measures all of them)
    if (n_features <= FT_MAX_FEATURES && G.n_classes <= NM_FUSED_FOREST_CLASSES && G.n_nodes > 0 &&
        n >= 4 * FT_THREADS) {
        const size_t lds = (size_t)FT_TREES * FT_TREE_CAP * sizeof(uint2) + (size_t)(FT_THREADS / 64) * n_features * 256;
        if (!ctx->forest_tiles_attr) {
            NM_HIP(ctx, hipFuncSetAttribute((const void*)k_forest_tiles, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            160 * 1024));
            ctx->forest_tiles_attr = true;
        }
        int64_t blocks = (n + FT_THREADS - 1) / FT_THREADS;
        if (blocks > ctx->num_cus) blocks = ctx->num_cus;        // one workgroup to a CU: they share nothing
        k_forest_tiles<<<(int)blocks, FT_THREADS, lds, s>>>(G, d_feat, n, feat_stride);
        NM_HIP(ctx, hipGetLastError());
        return NM_OK;
    }
    k_forest_eval_packed8<<<(int)((n + 255) / 256), 256, 0, s>>>(G, d_feat, n, feat_stride);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}

extern "C" int nm_forest_eval(nm_ctx* ctx, const nm_forest* forest, const double* d_feat, int64_t n,
                              int64_t feat_stride, double* d_proba, int32_t* d_label, void* stream)
{
    NM_ENTER_STREAM(ctx, stream);
    if (!forest || n < 0 || (n > 0 && !d_feat) || (!d_proba && !d_label))
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_forest_eval: bad arguments");
    if (!forest->d_packed && (!forest->d_left || !forest->d_right || !forest->d_feature ||
                              !forest->d_threshold || !forest->d_value || !forest->d_roots))
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_forest_eval: node arrays missing");
    if (forest->n_classes < 1 || forest->n_classes > NM_MAX_CLASSES)
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_forest_eval: n_classes must be in [1,%d]", NM_MAX_CLASSES);
    if (forest->n_trees < 1 || forest->n_features < 1 || feat_stride < forest->n_features)
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_forest_eval: bad forest shape");
    if (forest->d_leaf_value && forest->leaf_stride &&
        (forest->leaf_stride < forest->n_classes ||
         (forest->leaf_stride == 8 && (forest->n_classes > 8 || ((uintptr_t)forest->d_leaf_value & 63)))))
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_forest_eval: leaf_stride below n_classes, or rows of 8 doubles that are not "
                                     "64-byte aligned");
    if (n == 0) return NM_OK;
    const bool packed = forest->d_packed && forest->d_leaf_value && forest->d_packed_roots &&
                        forest->n_features <= NM_FOREST_MAX_FEATURES;
    const bool packed8 = forest->d_packed8 && forest->d_leaf_value && forest->d_packed_roots &&
                         forest->n_features <= NM_FOREST8_MAX_FEATURES;
    if (packed8) {
        ForestDev F;
        F.nodes = (const uint2*)forest->d_packed8;
        F.leaf_value = forest->d_leaf_value;
        F.leaf_stride = forest->leaf_stride ? forest->leaf_stride : forest->n_classes;
        F.roots = forest->d_packed_roots;
        F.n_trees = forest->n_trees;
        F.n_classes = forest->n_classes;
        F.n_features = forest->n_features;
        F.n_nodes = forest->n_nodes;
        F.proba = d_proba;
        F.pstride = forest->n_classes;
        F.label = d_label;
        return nm_forest_rows(ctx, F, d_feat, feat_stride, n, forest->n_features, (hipStream_t)stream);
    }
    if (packed)
        k_forest_eval_packed<<<(int)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(
            *forest, d_feat, n, feat_stride, d_proba, d_label);
    else
        k_forest_eval<<<(int)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(
            *forest, d_feat, n, feat_stride, d_proba, d_label);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}


// ---- derived descriptors --------------------------------------------------------------------------------
// linearity (l1-l2)/l1, planarity (l2-l3)/l1, scatter l3/l1 from the normalised eigenvalues the feature
// matrix already holds (l1, l2 in columns 4s+2, 4s+3; l3 = 1 - l1 - l2).  rows whose eigen-features are
// undefined (zeros) give zeros.
__global__ __launch_bounds__(256) void k_descriptors(const double* __restrict__ feat, int64_t n,
                                                     int32_t n_scales, int64_t fstride,
                                                     double* __restrict__ out, int64_t ostride)
{
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= n * n_scales) return;
    const int64_t row = t / n_scales;
    const int32_t s = (int32_t)(t - row * n_scales);
    const double l1 = feat[row * fstride + 4 * s + 2], l2 = feat[row * fstride + 4 * s + 3];
    double lin = 0.0, pla = 0.0, sca = 0.0;
    if (l1 > 0.0) {
        double l3 = 1.0 - l1 - l2;
        l3 = l3 < 0.0 ? 0.0 : l3;
        const double inv = 1.0 / l1;
        lin = (l1 - l2) * inv;
        pla = (l2 - l3) * inv;
        sca = l3 * inv;
    }
    double* o = out + row * ostride + 3 * s;
    o[0] = lin;
    o[1] = pla;
    o[2] = sca;
}

extern "C" int nm_descriptors(nm_ctx* ctx, const double* d_feat, int64_t n, int32_t n_scales,
                              int64_t feat_stride, double* d_out, int64_t out_stride, void* stream)
{
    NM_ENTER_STREAM(ctx, stream);
    if (n < 0 || n_scales < 0 || feat_stride < 4 * (int64_t)n_scales ||
        out_stride < 3 * (int64_t)n_scales || (n > 0 && n_scales > 0 && (!d_feat || !d_out)))
        NM_FAIL(ctx, NM_ERR_INVALID, "nm_descriptors: bad arguments");
    const int64_t total = n * n_scales;
    if (total == 0) return NM_OK;
    k_descriptors<<<(int)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        d_feat, n, n_scales, feat_stride, d_out, out_stride);
    NM_HIP(ctx, hipGetLastError());
    return NM_OK;
}
