// mcl_mesh.h -- triangle-mesh bathymetry: device acceleration structure + ray-cast kernel.
#pragma once
#include <string>

#include "../../include/mcl.h"
#include "mcl_mbes.h"

struct MeshDev {
  int dummy;
};

inline int mesh_build(const float*, int64_t, const uint32_t*, int64_t, MeshDev** out, std::string* err) {
  *out = nullptr;
  *err = "set_map_mesh: mesh ray-cast not built yet";
  return MCL_ERR_UNSUPPORTED;
}
inline void mesh_free(MeshDev* m) { delete m; }
inline int mesh_launch(MeshDev*, const MbesArgs&, hipStream_t) { return MCL_ERR_UNSUPPORTED; }
