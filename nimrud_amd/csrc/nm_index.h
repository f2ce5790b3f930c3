// nm_index.h - internal interface of the per-scale occupancy index (nm_index.hip)
#pragma once
#include "nm_common.h"

struct IndexLayout {
    uint32_t leaf_capacity;
    uint32_t hash_capacity;
    size_t hash_bytes, leaf_bytes, counter_bytes, total;
};

size_t nm_sort_pairs_temp_bytes(int64_t n);
void nm_index_layout(const LatticeDev& L, int64_t n_search, IndexLayout* out);

// cell keys of a cloud, sorted, with the permutation (val_sorted[i] = original row of sorted slot i)
int nm_sort_cells(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride, const LatticeDev& L,
                  uint64_t* key_tmp, uint32_t* val_tmp, uint64_t* key_sorted, uint32_t* val_sorted,
                  void* sort_temp, size_t sort_temp_bytes, hipStream_t s);

// hash table + leaves from the sorted keys of the search cloud
int nm_index_build(nm_ctx* ctx, const uint64_t* key_sorted, int64_t n, const IndexLayout& lay,
                   void* index_mem, IndexDev* out, hipStream_t s);

// whole-ladder path: one spatial order for all scales, every scale described by a ScaleDev in device memory.
// order[i] = original row of sorted slot i (sorted by the compact cell key of the lattice in *d_order_dev,
// nm_order.hip), sorted_xyz = the coordinates in that order, (n,3) contiguous - or null: the caller gathers them
// itself (nm_index_build_ladder_gather).  scratch: nm_order_scratch_bytes(n).  clear_ladder: the sort's three
// counting kernels also reset the indexes of that ladder, a third each (what nm_index_clear_all does as a launch
// of its own) - the caller then does not launch the clear
size_t nm_order_scratch_bytes(int64_t n);
int nm_order_build(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride, const OrderDev* d_order_dev,
                   void* scratch, size_t scratch_bytes, uint32_t* order, double* sorted_xyz, hipStream_t s,
                   const ScaleDev* clear_ladder = nullptr, int clear_scales = 0);

// the extrema of a cloud (nm_bounds) with NM_BOUNDS_SCRATCH_BYTES of caller scratch: no atomics
constexpr size_t NM_BOUNDS_SCRATCH_BYTES = 1024 * 6 * 8;
int nm_bounds_scratch(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride, double* d_minmax,
                      void* d_partial, hipStream_t s);

// the device-resident scale array: from host lattices, or from the cloud's extrema on the device
// leaf_alloc: leaves the workspace has room for per scale (> 0: coarse scales may be indexed densely);
// 0: keep every index exactly as the host sized it (the one-scale path)
int nm_ladder_put(nm_ctx* ctx, const LatticeDev* L, const IndexDev* I, const double* radii, int n_scales,
                  int finest, uint32_t leaf_alloc, ScaleDev* d_ladder, OrderDev* d_order, hipStream_t s);
// d_bounds_partial set: the extrema are the bounds pass's per-block pieces (nm_bounds_partial); the ladder kernel
// folds them and stores the six doubles in d_minmax_out.  else d_minmax holds them already
int nm_ladder_make(nm_ctx* ctx, const double* d_minmax, const void* d_bounds_partial, int bounds_blocks,
                   double* d_minmax_out, const double* edges, const double* radii,
                   int n_scales, int finest, void* const* hash, void* const* leaf, void* const* counters,
                   uint32_t hash_capacity, uint32_t leaf_capacity, ScaleDev* d_ladder, OrderDev* d_order,
                   hipStream_t s);
int nm_bounds_partial(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride, void* d_partial,
                      int* blocks_out, hipStream_t s);

// ---- the ladder's indexes: one per scale, cleared, built and counted together ---------------------------
IndexDev nm_index_at(nm_ctx* ctx, void* index_mem, const IndexLayout& lay);
int nm_index_clear_all(nm_ctx* ctx, const ScaleDev* d_ladder, int n, hipStream_t s);
int nm_index_count_all(nm_ctx* ctx, const ScaleDev* d_ladder, int n, hipStream_t s);
// indexes of scales [first, first + count) from a spatially coherent coordinate stream (no sort)
int nm_index_build_ladder(nm_ctx* ctx, const double* sorted_xyz, int64_t n, const ScaleDev* d_ladder,
                          int first, int count, hipStream_t s);
// the same from the cloud in the caller's order plus the permutation of the spatial sort: the builder gathers
// the coordinates into sorted_xyz itself (nm_order_build was then called with sorted_xyz = nullptr)
int nm_index_build_ladder_gather(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride,
                                 const uint32_t* order, double* sorted_xyz, const ScaleDev* d_ladder, int first,
                                 int count, hipStream_t s);
