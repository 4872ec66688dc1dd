// tree_registry.h -- registered configurations of the LDS-staged rollout kernel (lds_model.h, tree_kernel.h): the
// dimensions of a shipped model, fixed when the library is built. A run-time model with exactly these dimensions (and the
// features wave_tree.h covers) runs the specialised kernel; anything else runs the generic wavefront-per-candidate kernels.
// tests/test_capi_symbols.py::test_registered_tree_configs_match_the_shipped_models pins the numbers to the model files.
#pragma once

namespace mjpcx {

// Unitree A1 + QuadrupedFlat task (mujoco_mpc_amd/models/quadruped/task_flat.xml), BASELINE configs[2] / [4]
struct TreeCfgA1 {
  static constexpr int NQ = 19, NV = 18, NU = 12, NB = 16, NJ = 13, NS = 6, NG = 40, NKEY = 2, NMOCAP = 2;
  static constexpr int NBM = 16, NT = 0, NMAX = 18;    // bodies of the model (NB may leave inert trailing bodies out), limited fixed tendons, unroll width of the factorisations
  static constexpr int NSG = 4, NDG = 35, NRAY = 4;     // collidable static / moving geoms, geoms a ground ray can hit
  static constexpr int NR = 42, NTERM = 9, NTRACE = 1;  // residual entries, cost terms, traces
};

// dm_control humanoid + humanoid::Tracking (mujoco_mpc_amd/models/humanoid/tracking/task.xml), BASELINE configs[3].
// NB / NS count the live prefix (the 16 mocap marker bodies and their sites trail the model and have no dynamics: 37 / 38 in the
// model); NKEY = 0: the 1889 keyframes are NOT staged -- the tracking residual reads its marker table (key_mpos) from global memory
// and nothing reads key_qpos.
struct TreeCfgHumanoid {
  static constexpr int NQ = 28, NV = 27, NU = 21, NB = 21, NJ = 22, NS = 22, NG = 20, NKEY = 0, NMOCAP = 16;
  static constexpr int NBM = 37, NT = 2, NMAX = 28;
  static constexpr int NSG = 1, NDG = 19, NRAY = 4;
  static constexpr int NR = 141, NTERM = 21, NTRACE = 1;
};

}  // namespace mjpcx
