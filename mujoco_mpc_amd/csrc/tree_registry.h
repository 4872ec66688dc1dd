// tree_registry.h -- registered configurations of the LDS-staged rollout kernel (lds_model.h, tree_kernel.h): the
// dimensions of a shipped model, fixed when the library is built. A run-time model with exactly these dimensions (and the
// features wave_tree.h covers) runs the specialised kernel; anything else runs the generic wavefront-per-candidate kernels.
// tests/test_capi_symbols.py::test_registered_tree_configs_match_the_shipped_models pins the numbers to the model files.
#pragma once

namespace mjpcx {

// Unitree A1 + QuadrupedFlat task (mujoco_mpc_amd/models/quadruped/task_flat.xml), BASELINE configs[2] / [4]
struct TreeCfgA1 {
  static constexpr int NQ = 19, NV = 18, NU = 12, NB = 16, NJ = 13, NS = 6, NG = 40, NKEY = 2, NMOCAP = 2;
  static constexpr int NSG = 4, NDG = 35, NRAY = 4;     // collidable static / moving geoms, geoms a ground ray can hit
  static constexpr int NR = 42, NTERM = 9, NTRACE = 1;  // residual entries, cost terms, traces
};

}  // namespace mjpcx
