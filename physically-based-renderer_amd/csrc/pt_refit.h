// pt_refit.h — refit of a committed scene ON THE DEVICE (pt_refit.hip), declarations shared with ptc_api.cpp / ptc_scene.cpp.
//
// ptc_refit_scene (ptc_scene.cpp) is the definition: same tree, same slots, same unit layout; everything that depends on vertex positions
// is recomputed.  The device path does the same arithmetic (same expressions, -ffp-contract=off, correctly rounded / and sqrt) element-parallel
// and IN PLACE on the scene's arrays in HBM, so a moved instance costs three kernel families and a 32-byte read-back instead of a host pass
// over every vertex plus an upload of the whole BVH.  What stays on the host: the instances' normal matrices (a handful of 3x3 inverses) and
// the emitter table (the emissive primitives only).  tests/test_gpu_parity.py holds the device result byte for byte against the host refit
// and the oracle's.
#pragma once
#include "ptc_internal.h"

// What the device needs of the committed scene beyond what already lies in HBM (built once per commit by ptc_refit_plan).  The tree's topology is
// NOT in it: a node record keeps its slot masks and its children-block address (words 2, 3) and a triangle record its primitive id and class
// (words 3, 7) through a refit, so the kernels read those from the records they rewrite.
struct RefitPlan {
  std::vector<HostVertex> mesh_verts;   // object-space vertices of every mesh, back to back in mesh order
  std::vector<uint32_t> vert_inst;      // per world vertex: its instance
  std::vector<uint32_t> inst_first;     // per instance: its first world vertex
  std::vector<uint32_t> inst_src;       // per instance: the first vertex of its mesh in mesh_verts
  std::vector<uint32_t> level_nodes;    // unit addresses of the 8-wide nodes ordered by depth, deepest level first
  std::vector<uint32_t> level_first;    // first index into level_nodes of each level, plus the end
  std::vector<int32_t> emit_prims;      // per primitive with an emissive material, in primitive order: prim, instance, its 3 vertex indices inside the instance's mesh
  uint32_t n_verts = 0, n_tris = 0;
};
void ptc_refit_plan(const std::vector<HostMaterial>&, const std::vector<HostMesh>&, const std::vector<HostInstance>&, const HostBuilt&, RefitPlan& out);
// per instance 21 floats: rows 0..2 of the model matrix column by column (m[0..2] m[4..6] m[8..10] m[12..14]) and the 9 of the normal matrix, both as
// the host flatten uses them; false if a matrix entry is not finite
bool ptc_refit_instance_transforms(const std::vector<HostInstance>&, std::vector<float>& out21);
// the emitter table of the moved scene (20 floats per emitter, cdf), from the emissive primitives alone; false when the SET of emitters is not the
// committed one (a triangle's area became or stopped being zero): prim_light changes then and the caller refits on the host
bool ptc_refit_emitters(const std::vector<HostMaterial>&, const std::vector<HostMesh>&, const std::vector<HostInstance>&, const RefitPlan&, const HostBuilt& committed,
                        std::vector<float>& lights, std::vector<float>& cdf);
// scene box → origin grid and ray offset, as ptc_refit_scene derives them
void ptc_refit_grid(const float lo[3], const float hi[3], float grid_lo[3], float grid_step[3], float* ray_eps);

struct DevRefit {     // device copies of the plan and the scratch arrays; owned by the context's scene allocations
  const HostVertex* mesh_verts; const uint32_t* vert_inst; const uint32_t* inst_first; const uint32_t* inst_src;
  float* inst_xf;               // 21 floats per instance, uploaded per refit
  const uint32_t* widx;         // 3 world-vertex indices per primitive
  const uint32_t* level_nodes;
  HostVertex* wverts;           // world vertices (R1 vertex record after the instance transform)
  float* wbt;                   // world bitangent, 3 floats per world vertex
  float* nbox;                  // 6 floats per 64-byte record of the unit array: box of the node at unit address 4*i (written deepest level first)
  uint32_t* bounds;             // [0..5] scene box lo.xyz hi.xyz as order-preserving integers (atomic min / max), [6] non-finite flag, [7] pad
  unsigned long long* cost;     // surface-area cost of the tree (ptc_stats.bvh_sa_cost) in units of 2^-20, summed by the node pass
  const uint32_t* prim_cls;     // material class per primitive (word 7 of its triangle record): what a rebuild (pt_build.hip) writes into the new records
  float4* recs; float4* shade;  // the scene's arrays, rewritten in place
  uint32_t n_verts, n_tris, shade_stride;
};
// a commit on the device: the material and emitter index of every primitive into its (zeroed) shading record, which k_refit_prims then completes
void pt_launch_refit_seed(hipStream_t, const DevRefit&, const int32_t* tri_mat, const int32_t* prim_light);
// flatten + per-primitive pass (shading records, scene box).  The caller reads `bounds` back before the node pass: the origin grid follows the box.
void pt_launch_refit_geometry(hipStream_t, const DevRefit&);
void pt_refit_decode_bounds(const uint32_t raw[8], float lo[3], float hi[3], bool* non_finite);
// nodes and triangle records, level by level from the leaves up
// scene_half_area: half area of the scene box (the denominator of the cost sum; <= 0: no sum)
void pt_launch_refit_nodes(hipStream_t, const DevRefit&, const std::vector<uint32_t>& level_first, const float grid_lo[3], const float grid_step[3], float scene_half_area);
// the cost term of one child slot in units of 2^-20 (shared by the host build, which sums the same terms)
#define PTC_SA_COST_ONE 1048576.0f
