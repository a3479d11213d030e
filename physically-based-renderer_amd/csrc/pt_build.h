// pt_build.h — the BVH build ON THE DEVICE (pt_build.hip): SURVEY §8a P2 "host build first, GPU build later", north_star's "flattened LBVH".
//
// The definition of every value is the host's LBVH build (ptc_scene.cpp, build_or_refit with bvh_builder == PTC_BVH_LBVH), which the oracle pins
// (oracle/ptc_oracle.c: morton_order, morton_split, dp_compute, widen): 63-bit Morton codes of the triangle-box centres, order (code, primitive id),
// the radix tree of the keys (code, sorted position), bottom-up boxes and collapse-cost tables, the cost-optimal 8-wide collapse, octant slots, and the
// unit layout (breadth-first top, depth-first rest).  The device build writes THE SAME BYTES (tests/test_gpu_parity.py compares the unit arrays),
// from the world-space vertices a refit keeps in HBM (DevRefit::wverts / widx) — so a scene that has moved far from its commit is rebuilt where it
// lies (ptc_scene_rebuild), in milliseconds, instead of flattened, built and uploaded by the host again.
//
// Stages (one stream, no host arithmetic; the host reads back a node count per level of the 8-wide tree and the final sizes):
//   k_bld_prims    triangle boxes + the bounds of their centres                       k_bld_codes   Morton keys
//   k_sort_*       LSD radix sort of the 64-bit keys with their primitive ids, 8 bits per pass: per-wave tiles, digit ranking by ballot / mbcnt
//   k_bld_radix    every internal node of the radix tree from the sorted keys alone (Karras 2012)
//   k_bld_up       bottom-up from the leaves (second arrival proceeds): box and collapse-cost table of every internal node
//   k_bld_widen    level by level from the root: the <= 8 roots of the cheapest forest below a node, each in the slot of its octant
//   k_bld_sizes    bottom-up over the 8-wide tree: units of every children block and of every subtree
//   k_bld_top / k_bld_addr   addresses: the breadth-first prefix by one thread (<= toplet_budget nodes), the depth-first rest by prefix of subtree sizes
//   k_bld_emit     node headers (slot masks, children block) and triangle records (primitive id, class) — the planes, origins and triangle
//                  vertices are then written by the REFIT kernels (pt_refit.hip), exactly as a refit of that tree would
#pragma once
#include "pt_refit.h"

struct BuildScratch {       // device scratch of the build, grow-only, owned by the context
  void* p = nullptr; size_t bytes = 0;
};
struct BuildOut {
  // in: buffers the caller can spare (or null) and what they hold; out: the buffers the build wrote (the spare ones when they were large enough, else new
  // ones, 1/8 larger than needed, the spare ones freed).  The caller owns them either way.
  float4* recs = nullptr; size_t recs_cap = 0;            // the new unit array; capacity in 16-byte units
  uint32_t* level_nodes = nullptr; size_t level_cap = 0;  // unit addresses of the nodes by depth, deepest level first (the refit's plan); capacity in entries
  std::vector<uint32_t> level_first; // first index into level_nodes of each level (deepest first), plus the end
  uint32_t n_nodes = 0, n_units = 0, max_depth = 0, n_tri_records = 0;
};
// Builds the tree of the n_tris triangles (wverts, widx on the device; prim_cls = material class per primitive, on the device).  Returns an empty string or the error.
std::string pt_build_lbvh(hipStream_t st, const HostVertex* wverts, const uint32_t* widx, const uint32_t* prim_cls, uint32_t n_tris, uint32_t toplet_budget,
                          BuildScratch& scratch, BuildOut& out);
