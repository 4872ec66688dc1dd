// model_io.h -- loads a compiled model blob (written by mujoco_mpc_amd/mjcf.py::save_blob) into the
// minimal mjModel. Stands in for mj_loadXML (mjpc/testspeed.cc:54-68) in this MuJoCo-less build.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <mujoco/mujoco.h>

namespace mjpc {

class ModelStorage {
 public:
  // throws std::runtime_error on a malformed blob
  static std::unique_ptr<ModelStorage> Load(const std::string& path);
  mjModel* model() { return &model_; }
  const mjModel* model() const { return &model_; }

 private:
  mjModel model_{};
  std::map<std::string, std::vector<int>> ints_;
  std::map<std::string, std::vector<double>> reals_;
  std::map<std::string, std::vector<unsigned char>> bytes_;
  int* I(const std::string& name, size_t n);
  double* R(const std::string& name, size_t n);
  unsigned char* B(const std::string& name, size_t n);
  void Bind();
};

// mj_name2id for the object kinds the hot path looks up; -1 if absent
int NameToId(const mjModel* m, int objtype, const std::string& name);

}  // namespace mjpc
