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
// order[i] = original row of sorted slot i (sorted by the compact 32-bit cell key of the lattice in
// *d_order_dev), sorted_xyz = the coordinates in that order, (n,3) contiguous.
int nm_order_build(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride,
                   const OrderDev* d_order_dev, unsigned sort_bits, uint32_t* key_tmp,
                   uint32_t* val_tmp, uint32_t* key_sorted, uint32_t* order, void* sort_temp,
                   size_t sort_temp_bytes, double* sorted_xyz, hipStream_t s);

// bits of the spatial order's sort key (a wider key loses its low bits): three radix passes of 10 bits.
// (the benchmark scene's finest lattice has 31: the one bit makes no measurable difference to the index
// build or the search; round 1 measured that six do)
#ifndef NM_ORDER_KEY_BITS
#define NM_ORDER_KEY_BITS 30
#endif

// the extrema of a cloud (nm_bounds) with NM_BOUNDS_SCRATCH_BYTES of caller scratch: no atomics
constexpr size_t NM_BOUNDS_SCRATCH_BYTES = 1024 * 6 * 8;
int nm_bounds_scratch(nm_ctx* ctx, const double* d_xyz, int64_t n, int64_t stride, double* d_minmax,
                      void* d_partial, hipStream_t s);

// the device-resident scale array: from host lattices, or from the cloud's extrema on the device
// leaf_alloc: leaves the workspace has room for per scale (> 0: coarse scales may be indexed densely);
// 0: keep every index exactly as the host sized it (the one-scale path)
int nm_ladder_put(nm_ctx* ctx, const LatticeDev* L, const IndexDev* I, const double* radii, int n_scales,
                  int finest, uint32_t leaf_alloc, ScaleDev* d_ladder, OrderDev* d_order, hipStream_t s);
int nm_ladder_make(nm_ctx* ctx, const double* d_minmax, const double* edges, const double* radii,
                   int n_scales, int finest, void* const* hash, void* const* leaf, void* const* counters,
                   uint32_t hash_capacity, uint32_t leaf_capacity, ScaleDev* d_ladder, OrderDev* d_order,
                   hipStream_t s);

// ---- the ladder's indexes: one per scale, cleared, built and counted together ---------------------------
IndexDev nm_index_at(nm_ctx* ctx, void* index_mem, const IndexLayout& lay);
int nm_index_clear_all(nm_ctx* ctx, const ScaleDev* d_ladder, int n, hipStream_t s);
int nm_index_count_all(nm_ctx* ctx, const ScaleDev* d_ladder, int n, hipStream_t s);
// indexes of scales [first, first + count) from a spatially coherent coordinate stream (no sort)
int nm_index_build_ladder(nm_ctx* ctx, const double* sorted_xyz, int64_t n, const ScaleDev* d_ladder,
                          int first, int count, hipStream_t s);
