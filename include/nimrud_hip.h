/*
 * nimrud_hip.h - C ABI of libnimrud_hip.so: the MI355X (gfx950) implementation of the
 * grayhem/nimrud multiscale neighborhood-feature hot path (nimrud/minimal).
 *
 * The reference has no FFI layer: its boundary is the Python module surface of nimrud.minimal.  Each
 * entry point below names the reference code it replaces (paths relative to the reference checkout).
 * The Python package nimrud_amd binds these with ctypes (nimrud_amd/_ffi.py); INTEGRATION.md shows
 * the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only.  every pointer whose name starts with d_ is DEVICE memory (HBM) owned by the
 *     caller; the library never allocates or frees caller-visible memory.  scratch memory is passed in
 *     as (d_work, work_bytes); its required size comes from the matching *_workspace_bytes function.
 *   - point clouds are row-major fp64 arrays; `stride` is the row pitch in doubles (>= 3), so a
 *     (N, 3+F) cloud with feature columns is passed without a copy (minimal/README.md:38-40).
 *     the feature entry points take row pitches below 2^31 doubles (NM_ERR_INVALID otherwise).
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream)
 *     unless documented otherwise, and returns NM_OK or a negative nm_status.
 *     nm_last_error(ctx) returns a human-readable message for the last failure on that context.
 *   - one nm_ctx per (device, host thread).  no global state.
 */
#ifndef NIMRUD_HIP_H
#define NIMRUD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nm_ctx nm_ctx;

typedef enum nm_status {
    NM_OK = 0,
    NM_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, stride < 3 ...)            */
    NM_ERR_LATTICE = -2,     /* lattice cannot be addressed (geometry.py:59-60 ValueError), or is
                                outside the device path's limits (see nm_lattice)                     */
    NM_ERR_WORKSPACE = -3,   /* d_work too small                                                       */
    NM_ERR_HIP = -4,         /* a HIP runtime call failed                                              */
    NM_ERR_RADIUS = -5,      /* radius / edge ratio outside the supported range                        */
    NM_ERR_COMM = -6         /* an RCCL call failed                                                    */
} nm_status;

/*
 * The bounding lattice of one analysis scale: VoxelFilter.__init__ / _calculate_shift
 * (nimrud/utils/geometry.py:23-64).  Filled on the host (nimrud_amd/utils/geometry.py does the same
 * arithmetic as the reference) and passed by pointer.
 *   min_corner = cloud.min(0) - edge/2          geometry.py:37
 *   widths     = ceil(log2(span/edge)) per axis geometry.py:56   (sum <= 64, geometry.py:59)
 *   shifts     = cumsum(widths)[:-1]            geometry.py:62
 * Device-path limits: 3 spatial dimensions, every width in [1, 30].
 */
typedef struct nm_lattice {
    double  min_corner[3];
    double  edge;
    int32_t widths[3];
    int32_t shifts[2];
} nm_lattice;

/* ---- context ---------------------------------------------------------------------------------- */

int         nm_create(nm_ctx** out, int device);
void        nm_destroy(nm_ctx* ctx);
const char* nm_last_error(const nm_ctx* ctx);
/* library ABI version (bumped on any signature change) */
int         nm_abi_version(void);

/* ---- asynchronous failures -------------------------------------------------------------------------
 * every call is asynchronous, so a failure that only a kernel can detect (the occupancy-index builder
 * running out of its bounded wait or of leaves; a lattice built on the device that cannot be addressed,
 * geometry.py:59-60) cannot be the return value of the call that enqueued it.  such a failure is STICKY on
 * the context: the library leaves a snapshot of its device-side status words behind every feature call,
 * and the next nm_* call that launches work - or nm_check - returns NM_ERR_HIP / NM_ERR_LATTICE with a
 * message once that snapshot has arrived, and keeps doing so until nm_clear_error.  results of the
 * failed call must be discarded.  nm_check(ctx, 1) waits for the snapshot of the last call (call it after
 * synchronising the stream, before trusting results); nm_check(ctx, 0) only looks.  no reference
 * counterpart (the reference is synchronous and raises in place).                                     */
int nm_check(nm_ctx* ctx, int wait);
int nm_clear_error(nm_ctx* ctx);

/* ---- in-library stage timing ----------------------------------------------------------------------
 * between nm_profile_begin and nm_profile_end every nm_scale_features call brackets its stages with
 * HIP events on the caller's stream.  nm_profile_end synchronises on them and returns the summed
 * device time in milliseconds: ms[0] cell keys + radix sort, ms[1] occupancy index build,
 * ms[2] the fused search/feature kernel(s), ms[3] reserved; *launches = search-kernel launches timed.
 * used by bench.py for the roofline figure; no reference counterpart (the reference's verbose mode
 * prints wall time per scale, multiscale.py:47-65).                                                */
int nm_profile_begin(nm_ctx* ctx);
/* kept for ABI compatibility; no effect since ABI 5: one launch builds all indexes of a ladder before the
 * first search kernel, there is nothing left to overlap (round 1 measured 2 % for the overlapped form).  */
int nm_set_overlap(nm_ctx* ctx, int enabled);
int nm_profile_end(nm_ctx* ctx, double* ms, int64_t* launches);

/* ---- cloud bounds -------------------------------------------------------------------------------
 * per-axis min and max of a cloud: the `points.min(0)` / `points.max(0)` of geometry.py:37-38.
 * d_minmax receives 6 doubles {min x,y,z, max x,y,z}.  the host turns them into an nm_lattice.    */
int nm_bounds(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride,
              double* d_minmax, void* stream);

/* ---- spatial order ------------------------------------------------------------------------------
 * the order the ladder entry points work in internally (nm_order.hip), exposed for inspection and
 * tests; no reference counterpart (the reference never reorders a cloud).  rows sorted by the
 * compact Z-order key of their cell of `lat` (30 key bits, a wider key loses its low bits):
 * d_order[i] = row of the point in sorted slot i, d_sorted_xyz (n,3) = the coordinates in that order,
 * d_keys_sorted (nullable) = the key of every sorted slot.  radix sort of our own: three passes,
 * per-tile digit counts + one flat scan + a scatter kernel per pass, no look-back.                 */
size_t nm_spatial_order_workspace_bytes(int64_t n);
int nm_spatial_order(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride, const nm_lattice* lat,
                     uint32_t* d_order, double* d_sorted_xyz, uint32_t* d_keys_sorted, void* d_work,
                     size_t work_bytes, void* stream);

/* ---- voxelize -----------------------------------------------------------------------------------
 * VoxelFilter.coordinate_to_address + numpy.unique (geometry.py:103-116, 148-150): the sorted
 * distinct 64-bit voxel addresses x + (y << shifts[0]) + (z << shifts[1]) of the cells
 * floor((p - min_corner)/edge) occupied by the cloud.  position in this array is the reference's
 * search-voxel index.  d_addr_out needs room for n entries; *d_count (device int64) receives M.
 * points outside the lattice make the call fail the way _check_in_bounds does (geometry.py:95-97):
 * d_count[1] receives the number of out-of-bounds points (the host raises ValueError when != 0).  */
size_t nm_voxelize_workspace_bytes(int64_t n);
int nm_voxelize(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride, const nm_lattice* lat,
                int64_t* d_addr_out, int64_t* d_count, void* d_work, size_t work_bytes, void* stream);

/* VoxelFilter.coordinate_to_address (geometry.py:103-116) without the unique: one address per point,
 * in point order.  *d_oob (device int64, nullable) receives the number of points whose cell lies
 * outside [0, 2^width) on some axis (they are clamped); the bounds check proper
 * (_check_in_bounds, geometry.py:83-99) is done by the host with nm_bounds.                         */
int nm_coordinate_to_address(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride,
                             const nm_lattice* lat, int64_t* d_addr_out, int64_t* d_oob,
                             void* stream);

/* VoxelFilter.address_to_coordinate (geometry.py:120-138): centre = cell*e + min_corner + e*0.5,
 * evaluated left to right in fp64 without fused multiply-add.  d_xyz_out is (m,3) row-major.       */
int nm_address_to_coordinate(nm_ctx* ctx, const int64_t* d_addr, int64_t m, const nm_lattice* lat,
                             double* d_xyz_out, void* stream);

/* ---- one analysis scale, fused -------------------------------------------------------------------
 * one_scale_single_core (nimrud/minimal/multiscale.py:70-123) in one call:
 *   voxel-filter the search cloud (multiscale.py:76-77), find for every query point all voxel
 *   centres with Euclidean distance <= radius (cKDTree.query_ball_tree, multiscale.py:87-103;
 *   inclusive, ((dx*dx + dy*dy) + dz*dz) <= r*r in fp64), and write per query point
 *     [population, ||q - centroid||, l1/(l1+l2+l3), l2/(l1+l2+l3)]      features.py:21-57
 *   (eigenvalues of the ddof=1 covariance, descending) into d_feat[row*feat_stride + 0..3].
 * rows are in query order.  feat_stride is in doubles (4*S for the (Nq,4S) matrix of
 * process_single_core, with d_feat pre-offset by 4*s for scale s: multiscale.py:56).
 * neighborhoods with fewer than 2 voxels get zero eigen-features, empty ones zero centroid
 * (multiscale.py:4-5).  d_info (device int64[4], nullable) receives {M = number of occupied voxels,
 * number of neighborhoods with population < 2, passes taken by the search kernel, reserved}.
 * the query cloud may be the search cloud, or its leading n_query rows (same pointer and stride,
 * n_query <= n_search): the cloud is then sorted and indexed once.                                  */
size_t nm_scale_workspace_bytes(int64_t n_query, int64_t n_search, const nm_lattice* lat);
int nm_scale_features(nm_ctx* ctx,
                      const double* d_query, int64_t n_query, int64_t query_stride,
                      const double* d_search, int64_t n_search, int64_t search_stride,
                      const nm_lattice* lat, double radius,
                      double* d_feat, int64_t feat_stride, int64_t* d_info,
                      void* d_work, size_t work_bytes, void* stream);

/* ---- the whole scale ladder ---------------------------------------------------------------------------
 * process_single_core (nimrud/minimal/multiscale.py:27-67) in one call: scale i uses lats[i] and
 * radii[i] and writes columns 4i..4i+3 of the (n_query, feat_stride) matrix d_feat, scales in caller
 * order (multiscale.py:37,56).  numerically identical to calling nm_scale_features per scale, but the
 * cloud is sorted only once (by its cell keys at the finest lattice of the ladder; the coordinates
 * are kept in that order) and every scale's occupancy index is built from that spatially coherent
 * stream without a sort.  d_info (nullable) receives 4 int64 per scale as in nm_scale_features.      */
size_t nm_multiscale_workspace_bytes(int64_t n_query, int64_t n_search, const nm_lattice* lats,
                                     int32_t n_scales);
/* consecutive scales with the same radius/edge ratio run in ONE launch of the search kernel (a wave keeps its
 * 64 queries and walks the scales).  on by default; 0 = one launch per scale (same numbers).          */
int nm_set_fuse_scales(nm_ctx* ctx, int enabled);

/* ---- the whole scale ladder, lattices built on the device ----------------------------------------------
 * process_single_core (multiscale.py:27-67) without a single value visiting the host: the search cloud's
 * extrema (a bounds pass, or d_minmax: 6 doubles ON THE DEVICE {min xyz, max xyz}, e.g. the global extrema a
 * multi-GPU job has agreed on) are turned into every scale's lattice by a kernel - VoxelFilter.__init__ /
 * _calculate_shift, geometry.py:37-64, same arithmetic: min_corner = min - e/2, widths =
 * ceil(log2((max_corner - min_corner)/e)) - and all later kernels read the lattices from device memory.
 * nothing in the call waits for the device: it can be queued behind other work.  (it can also be captured in
 * a hipGraph - the library records no event and queries none while its stream is capturing - and a replay is
 * bit-identical to the eager call, on new data in the same buffers too: tools/graph_probe.py.  replay buys
 * nothing measurable - the step is bound by its kernels, not by their launches - and back-to-back replays of
 * the 10 M-point step stalled in that probe, so the product path does not use graphs.)
 * what a host-side VoxelFilter would have raised (geometry.py:59-60 "edge length is too small to address
 * this space"; no extent on an axis; a width outside the device path's [1,30]) is reported asynchronously
 * as NM_ERR_LATTICE through nm_check (see there).  the workspace depends on the point counts only.
 * bit-identical to nm_multiscale_features with the lattices the host would have built.                 */
size_t nm_ladder_workspace_bytes(int64_t n_query, int64_t n_search, int32_t n_scales);
/* the same when the edge lengths are known: scales of EQUAL edge length have one lattice and share one occupancy
 * index (the reference's ladders are one voxel edge with several radii, nimrud/utils/point_clouds.py:29-35), so
 * they need its memory once.  never more than nm_ladder_workspace_bytes.                                   */
size_t nm_ladder_workspace_bytes_for(int64_t n_query, int64_t n_search, const double* edges, int32_t n_scales);
int nm_ladder_features(nm_ctx* ctx,
                       const double* d_query, int64_t n_query, int64_t query_stride,
                       const double* d_search, int64_t n_search, int64_t search_stride,
                       const double* edges, const double* radii, int32_t n_scales,
                       const double* d_minmax,
                       double* d_feat, int64_t feat_stride, int64_t* d_info,
                       void* d_work, size_t work_bytes, void* stream);
int nm_multiscale_features(nm_ctx* ctx,
                           const double* d_query, int64_t n_query, int64_t query_stride,
                           const double* d_search, int64_t n_search, int64_t search_stride,
                           const nm_lattice* lats, const double* radii, int32_t n_scales,
                           double* d_feat, int64_t feat_stride, int64_t* d_info,
                           void* d_work, size_t work_bytes, void* stream);

/* ---- k-nearest-voxel fallback (BASELINE config 4; no reference counterpart) -----------------------------
 * build-defined extension: while k_min > 0, nm_scale_features / nm_multiscale_features re-evaluate every
 * query whose radius neighborhood holds fewer than k_min voxels on the k_min nearest voxel centres within
 * radius_factor * radius (fewer if there are not that many; ties broken by smaller voxel address).
 * column 0 keeps the radius population; centroid distance and eigen-features come from the k-NN set.
 * k_min <= 16.  the reference has nothing of the kind (its only kNN mention is a classifier with a
 * missing import, prototypes/apc.py:1479); parity is pinned by the build's own oracle
 * (oracle.one_scale_knn: cKDTree.query on the same voxel set).                                       */
int nm_set_knn_fallback(nm_ctx* ctx, int k_min, double radius_factor);

/* ---- covariance output (legacy C_MSO, prototypes/mso.py:1555) -----------------
 * while d_cov is not NULL, nm_scale_features / nm_multiscale_features also write, per query row and
 * scale s, the upper triangle [xx, xy, xz, yy, yz, zz] of the ddof=1 covariance of the neighborhood's
 * voxel centres - the matrix features.pca forms with numpy.cov (features.py:43) before it takes its
 * eigenvalues - to d_cov[row * cov_stride + 6*s + 0..5] (nm_scale_features: s = 0), in the cloud's
 * units squared.  neighborhoods with fewer than 2 voxels give zeros; rows re-evaluated by the kNN
 * fallback get the covariance of the kNN set.  pass NULL to switch it off again.  the reference returns
 * only the two normalised eigenvalues (features.py:57); this is an opt-in extra, checked against
 * numpy.cov on the oracle's neighborhoods.                                                          */
int nm_set_covariance_output(nm_ctx* ctx, double* d_cov, int64_t cov_stride);

/* ---- surface normal output (legacy OG_MSO, prototypes/mso.py:1315) -------------
 * while d_normal is not NULL, the same calls also write, per query row and scale s, the unit
 * eigenvector of the SMALLEST eigenvalue of that covariance - the normal of the best-fitting plane - to
 * d_normal[row * normal_stride + 3*s + 0..2].  the sign is fixed by making the last non-zero component
 * in the order x, y, z positive (normals point up).  zeros for neighborhoods with fewer than 3 voxels;
 * where the two smallest eigenvalues (nearly) coincide the direction is not defined by the data and
 * the vector is some unit vector of that eigenspace.  opt-in extra, checked against numpy.linalg.eigh
 * on the oracle's neighborhoods.  pass NULL to switch it off.                                        */
int nm_set_normal_output(nm_ctx* ctx, double* d_normal, int64_t normal_stride);

/* ---- neighbor lists (parity / inspection mode) -----------------------------------------------------
 * the neighbor_idx lists of multiscale.py:103 as CSR.  two calls: with d_nbr_index == NULL the
 * per-query counts are written to d_nbr_count (int32[n_query]); the caller turns them into offsets
 * (exclusive prefix sum, int64[n_query+1]) and calls again with d_nbr_offsets and d_nbr_index
 * (int64[total]).  indices are positions in the sorted unique address array d_addr (from
 * nm_voxelize), ascending within a query point.                                                     */
int nm_scale_neighbors(nm_ctx* ctx,
                       const double* d_query, int64_t n_query, int64_t query_stride,
                       const int64_t* d_addr, int64_t m, const nm_lattice* lat, double radius,
                       int32_t* d_nbr_count, const int64_t* d_nbr_offsets, int64_t* d_nbr_index,
                       void* stream);

/* ---- explicit neighborhoods -----------------------------------------------------------------------
 * features.population / centroid / pca (features.py:21-57) for a batch of explicit neighborhoods
 * given as CSR over a point array: neighborhood b is d_points[d_offsets[b] .. d_offsets[b+1]) (rows
 * of 3 doubles), d_query row b is its query point.  output as in nm_scale_features.                 */
int nm_neighborhood_features(nm_ctx* ctx, const double* d_points, const int64_t* d_offsets,
                             const double* d_query, int64_t n_neighborhoods,
                             double* d_feat, int64_t feat_stride, void* stream);

/* ---- multi-GPU tiling: halo selection -------------------------------------------------------------
 * no reference counterpart in nimrud/minimal (it is single-process); the legacy partitioners pair a
 * query tile with a search tile grown by the largest scale (prototypes/mso.py:892-927,
 * utils/geometry.py:203-253).  a rank owns one spatial tile; d_boxes holds one axis-aligned box per
 * rank (n_boxes x 6 doubles: lo xyz, hi xyz, already grown by the halo margin), `skip` is the
 * caller's own rank.  nm_halo_count writes, per destination box, how many of the caller's points lie
 * inside it (inclusive); nm_halo_pack writes those points as rows of 3 doubles into d_out, the rows
 * of destination b starting at row d_offsets[b] (order within a destination is unspecified);
 * d_cursor is int64[n_boxes] scratch.  the packed rows are what goes through RCCL's all-to-all.
 * nm_copy_xyz copies the geometry columns of a strided cloud into (n,3) contiguous rows.            */
int nm_halo_count(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride, const double* d_boxes,
                  int32_t n_boxes, int32_t skip, int64_t* d_counts, void* stream);
int nm_halo_pack(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride, const double* d_boxes,
                 int32_t n_boxes, int32_t skip, const int64_t* d_offsets, int64_t* d_cursor,
                 double* d_out, void* stream);
int nm_copy_xyz(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride, double* d_out,
                void* stream);

/* ---- multi-GPU tiling: cell-set halos -----------------------------------------------------------------
 * a Morton-contiguous tile (what BASELINE's north_star shards by) is L-shaped: its bounding box, and with
 * it a box halo, covers far more than the tile.  a tile's CELL SET is the set of cells of a coarse cubic
 * grid occupied by its points, dilated by the margin: NM_HALO_CELLSET_WORDS 32-bit words, one bit per
 * cell, cell index = (z * dim_y + y) * dim_x + x.  the grid is a pure function of the global extrema
 * (d_global_minmax: 6 doubles on the device, identical on every rank) and the margin - cell edge margin/4
 * unless that needs more than 2^21 cells - so every rank computes the same one without talking.
 * nm_halo_cellset writes the caller's own set; nm_halo_count_cells / nm_halo_pack_cells are
 * nm_halo_count / nm_halo_pack with "inside destination r's cell set" (d_cellsets: n_ranks sets back to
 * back, e.g. straight out of an all-gather) in place of the box test.  same idea as the reference's
 * query tile + grown search tile (prototypes/mso.py:892-927), for tiles of any shape.                */
#define NM_HALO_CELLSET_WORDS 65536
size_t nm_halo_cellset_workspace_bytes(void);
int nm_halo_cellset(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride,
                    const double* d_global_minmax, double margin, uint32_t* d_cellset,
                    void* d_work, size_t work_bytes, void* stream);
int nm_halo_count_cells(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride,
                        const double* d_global_minmax, double margin, const uint32_t* d_cellsets,
                        int32_t n_ranks, int32_t skip, int64_t* d_counts, void* stream);
int nm_halo_pack_cells(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride,
                       const double* d_global_minmax, double margin, const uint32_t* d_cellsets,
                       int32_t n_ranks, int32_t skip, const int64_t* d_offsets, int64_t* d_cursor,
                       double* d_out, void* stream);

/* ---- multi-GPU tiling: the halo exchange over RCCL ------------------------------------------------------
 * one process per GPU, one nm_ctx and one RCCL communicator per process.  the communicator is a plain
 * ncclComm_t (passed as void*): create it with nm_comm_create from a 128-byte unique id that rank 0 got
 * from nm_comm_unique_id and handed to the other ranks by any means (the Python host broadcasts it with
 * torch.distributed, which is all it uses torch.distributed for), or pass one the host already owns.
 *
 * nm_halo_exchange is collective over the communicator.  every rank passes its tile (d_xyz: n rows) and
 * the halo margin max_s(radius_s + sqrt(3)/2 edge_s); on return (the sends and receives are enqueued on
 * `stream`) d_recv holds, in rank order, the rows of all other tiles that lie within the margin of this
 * tile - by destination box (NM_HALO_BOXES) or by destination cell set (NM_HALO_CELLS, see above) -
 * *h_recv_rows / *h_sent_rows (HOST) the row counts, and d_global_minmax (device, 6 doubles) the extrema of
 * the WHOLE cloud, from which every rank builds the same lattices as a single-process run
 * (geometry.py:37).  traffic: all-gather of 6 doubles, (cell mode) all-gather of 256 KB, all-gather of
 * n_ranks + 3 int64, then grouped ncclSend / ncclRecv of 24-byte rows between the pairs that share a
 * boundary.  the call synchronises `stream` ONCE, to learn the sizes.  when some rank's buffers are too
 * small EVERY rank returns NM_ERR_WORKSPACE (with its own counts filled in) before anything is sent, so
 * the hosts can grow their buffers and call again.  a tile may be EMPTY (n = 0: it contributes an empty
 * box and receives nothing).  a rank that alone is unwell - a sticky failure of an earlier call, bad cloud
 * arguments - still goes through the collectives, with an empty contribution and its status in the
 * announced row: after the host synchronisation EVERY rank returns that status; nobody waits in an
 * all-gather for a rank that has left.  (nm_halo_plan_from_matrix is that decision as a host function
 * of the gathered matrix - n_ranks rows of n_ranks + 3 int64: rows sent to each rank, send capacity,
 * receive capacity, status - exported so that it can be tested without a communicator.)
 * NM_HALO_REUSE_PLAN (or-ed into mode, by EVERY rank or by none): the tiles have not changed since this
 * context's last full call with the same communicator size, rank, mode, tile size and workspace - boxes, cell
 * sets and the pair counts stand.  the call then packs and exchanges the rows again (the data moves every
 * step) but skips the three all-gathers and the host synchronisation: a step on a static cloud enqueues and
 * returns.  nm_halo_stats counts both kinds of call.  the send staging area is whatever d_work holds beyond
 * nm_halo_workspace_bytes(0, n_ranks).  NM_HALO_INCLUDE_SELF (or-ed into mode) makes a rank its own
 * neighbour as well - it then receives its own tile - which is how a one-rank communicator exercises the
 * whole path.  no reference counterpart: the reference is single-process.                              */
#define NM_COMM_ID_BYTES 128
enum { NM_HALO_BOXES = 0, NM_HALO_CELLS = 1, NM_HALO_INCLUDE_SELF = 4, NM_HALO_REUSE_PLAN = 8 };
int nm_comm_unique_id(void* id_out /* NM_COMM_ID_BYTES, host */);
int nm_comm_create(nm_ctx* ctx, int32_t n_ranks, int32_t rank, const void* id, void** comm_out);
int nm_comm_destroy(nm_ctx* ctx, void* comm);
size_t nm_halo_workspace_bytes(int64_t send_capacity_rows, int32_t n_ranks);
int nm_halo_stats(nm_ctx* ctx, int64_t* host_syncs, int64_t* exchanges);
int nm_halo_plan_from_matrix(const int64_t* matrix, int32_t n_ranks, int32_t rank, int64_t* send_off,
                             int64_t* recv_off, int64_t* sent_rows, int64_t* recv_rows, int32_t* culprit);
int nm_halo_exchange(nm_ctx* ctx, void* nccl_comm, int32_t n_ranks, int32_t rank,
                     const double* d_xyz, int64_t n, int64_t stride, double margin, int32_t mode,
                     double* d_recv, int64_t recv_capacity_rows,
                     int64_t* h_recv_rows, int64_t* h_sent_rows, double* d_global_minmax,
                     void* d_work, size_t work_bytes, void* stream);

/* ---- derived descriptors ------------------------------------------------
 * linearity (l1-l2)/l1, planarity (l2-l3)/l1 and scatter l3/l1 per scale, from the normalised
 * eigenvalues of a feature matrix produced by nm_(multi)scale_features (l3 = 1 - l1 - l2).  the
 * reference's minimal path stops at (l1, l2) (features.py:57); its legacy generation carries full
 * covariance / eigenvector variants (prototypes/mso.py:1315,1555).  d_out is (n, out_stride) with 3
 * columns per scale; undefined rows give zeros.                                                    */
int nm_descriptors(nm_ctx* ctx, const double* d_feat, int64_t n, int32_t n_scales, int64_t feat_stride,
                   double* d_out, int64_t out_stride, void* stream);

/* ---- vector-field operator (legacy V_MSO, prototypes/mso.py:12-173) ----------
 * neighborhood mean of arbitrary per-point attributes on the lattice of nimrud/minimal: the search cloud
 * is voxel-filtered as for the features (geometry.py:103-154); a voxel's attribute vector is the mean of
 * d_attr[row*attr_stride + 0..dims) over the search points in it; query row q receives, in
 * d_out[q*out_stride + 0..dims), the mean of the voxel attributes over the voxel centres within
 * `radius` (inclusive, the predicate of multiscale.py:87-103), zeros if there are none.  1 <= dims <= 16.
 * one scale per call.  the legacy operator (fp32, partition dependent) cannot be reproduced number for
 * number and the reference's current path has none: parity is pinned by the build's own oracle.      */
size_t nm_field_workspace_bytes(int64_t n_query, int64_t n_search, const nm_lattice* lat, int32_t dims);
int nm_field_mean(nm_ctx* ctx,
                  const double* d_query, int64_t n_query, int64_t query_stride,
                  const double* d_search, int64_t n_search, int64_t search_stride,
                  const double* d_attr, int64_t attr_stride, int32_t dims,
                  const nm_lattice* lat, double radius,
                  double* d_out, int64_t out_stride, void* d_work, size_t work_bytes, void* stream);

/* ---- classifier slot -------------------------------------------------------------------------------
 * nimrud/minimal/classification.py is a stub; the reference's classifier is sklearn's
 * RandomForestClassifier (prototypes/apc.py:1463) applied per point with predict / predict_proba
 * (apc.py:1022,1034).  the forest is passed flattened: node arrays of all trees concatenated,
 * children are absolute node indices (-1 at a leaf), `value` is the per-node class distribution
 * normalised to sum 1, `roots` the root node of each tree.  features are cast to fp32 and a sample
 * goes left when (double)(float)x[feature] <= threshold, as sklearn does.
 * d_proba (n, n_classes) row-major (nullable), d_label int32[n] = argmax class position (nullable). */
typedef struct nm_forest {
    const int32_t* d_left;
    const int32_t* d_right;
    const int32_t* d_feature;
    const double*  d_threshold;
    const double*  d_value;
    const int32_t* d_roots;
    int32_t n_nodes;
    int32_t n_trees;
    int32_t n_classes;
    int32_t n_features;
    /* optional fast layout (both or neither): nodes renumbered so that the right child of node i is
     * left(i) + 1, one 16-byte record per node {double threshold; int32 left; int32 feature}; a leaf has
     * left = -1 and feature = its row in d_leaf_value (n_leaves x n_classes, rows sum to 1).
     * d_packed_roots holds the root record of each tree.  with these set (and n_features <= 40) one
     * record load replaces four array loads per node visit.                                         */
    const void*    d_packed;
    const double*  d_leaf_value;
    const int32_t* d_packed_roots;
    int32_t n_leaves;
    /* doubles from one row of d_leaf_value to the next; 0 = n_classes (rows packed).  with rows of 8 doubles,
     * 64-byte aligned and zero-padded behind n_classes (what ForestModel builds for up to 8 classes), a lane's
     * leaf distribution is one cache line fetched with 16-byte loads: the gather of the distributions - 39 % of
     * the forest stage of config 5 with packed rows of 40 bytes - costs half                                    */
    int32_t leaf_stride;
    /* optional compact layout of the same renumbered nodes (needs d_leaf_value and d_packed_roots too, and
     * n_features <= 32): one 8-byte record per node {float threshold; uint32 packed}.  the threshold is the
     * largest fp32 value not above the fp64 threshold - sklearn compares the fp32-cast feature with the fp64
     * threshold, and x_f32 <= t_f64 holds exactly when x_f32 <= that value.  packed: bits 30..13 = the left
     * child's record (so at most 2^18 nodes), bits 12..8 the feature, bits 7..0 zero; a leaf has bit 31 set,
     * its row in d_leaf_value in bits 30..13 and the rest zero.  preferred by nm_forest_eval when present; required by
     * nm_set_forest_output.                                                                             */
    const void*    d_packed8;
} nm_forest;

int nm_forest_eval(nm_ctx* ctx, const nm_forest* forest, const double* d_feat, int64_t n,
                   int64_t feat_stride, double* d_proba, int32_t* d_label, void* stream);

/* the classifier behind the last scale (BASELINE config 5: "random-forest classifier evaluation fused after
 * feature assembly").  while a forest is set, nm_multiscale_features / nm_ladder_features evaluate it on
 * every query row right after the row's last scale, inside the search kernel: the wave that has just written
 * the row reads its 4*S features back while they are still in the XCD's L2, and its 64 lanes - neighbours in
 * space - walk nearly the same paths.  d_proba (n_query, proba_stride) and d_label int32[n_query], rows in
 * query order, either may be NULL; same numbers as nm_forest_eval on the finished matrix (which is what
 * runs instead when the last scale has an unusual radius/edge ratio or the kNN fallback is on).  limits:
 * n_features == 4 * n_scales <= 20, n_classes <= 8, the d_packed8 layout.  context state, like the
 * covariance output; pass forest = NULL to switch it off.  the struct is copied, the arrays must stay
 * alive.                                                                                               */
int nm_set_forest_output(nm_ctx* ctx, const nm_forest* forest, double* d_proba, int64_t proba_stride,
                         int32_t* d_label);
/* where the forest of nm_set_forest_output runs: 1 (default) in the epilogue of the last search kernel; 0 as a
 * launch of its own directly behind the last scale, over the rows in the ladder's spatial order; 2 as a launch of
 * its own over the finished matrix with the trees streamed through LDS (what nm_forest_eval runs).  same
 * numbers.  measured on MI355X, 10 M rows x 32 trees (tools/forest_modes.py): 2.1 ms in the epilogue - the tree
 * walk waits on memory, and there it waits while other waves of the same SIMD run their (vector-ALU bound)
 * search - against 2.5 ms (mode 0) and 3.0 ms (mode 2; 3.5 ms for a row walk from memory in row order). */
int nm_set_forest_mode(nm_ctx* ctx, int in_search_kernel);

#ifdef __cplusplus
}
#endif
#endif /* NIMRUD_HIP_H */
