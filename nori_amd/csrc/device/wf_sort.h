/* wf_sort.h -- reordering the paths of a wavefront pass before wf_extend walks them (wf_sort.hip).
 *
 * The reference walks its samples in pixel order (src/main.cpp:38-55); a GPU is free to trace the rays of a pass in any order,
 * and after the first bounce the order the paths were generated in is not the order in which their rays are coherent.  A pass
 * is reordered by an INDEX: key per path slot -> radix sort of (key, slot) -> wf_extend takes slot perm[k] where it took slot
 * k.  Path records, hit records and wf_shade stay where they are.  Radiance per path is untouched (same rays, same hits), so
 * frames and ray counts do not change. */
#pragma once
#include <cstdint>
#include <string>

#include "rt_types.h"

namespace nrt {

struct WfSortBuffers {
    uint32_t *keys[2] = {nullptr, nullptr};
    uint32_t *vals[2] = {nullptr, nullptr};      /* vals[1]: the sorted slots = the permutation wf_extend reads */
    void *temp = nullptr;
    size_t temp_bytes = 0, capacity = 0;
    void release();
};

/* key kinds (NORI_HIP_WF_SORT): 1 = (Morton cell of the origin, `bits` per axis) << 3 | octant of the continuation direction;
   2 = the same with "has a shadow ray" on top; 3 = octant << 3 bits | cell.  Empty slots sort to the end. */
struct WfSortParams { int kind = 0, cell_bits = 5; };

/* sorts the n slots of a pass by key; afterwards b.vals[1][0 .. n) is the permutation.  o: origins (12 B), dA: (continuation
   direction, flags) as wf_records.h lays them out.  Returns "" or an error message. */
std::string wf_sort_pass(WfSortBuffers &b, const DevScene &sc, const void *o, const void *dA, uint32_t n, const WfSortParams &p, void *stream);

} // namespace nrt
