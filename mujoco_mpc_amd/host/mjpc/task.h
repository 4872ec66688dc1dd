// mjpc::Task / ResidualFn / BaseResidualFn with the reference's interface (mjpc/task.h). The one
// addition is Task::DeviceResidualId(): the id of the device function that implements the task's
// ResidualFn::Residual inside the GPU rollout kernels (include/mjpcx.h MJPCX_RESIDUAL_*).
#pragma once
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include <mujoco/mujoco.h>
#include "norm.h"

namespace mjpc {

inline constexpr double kRiskNeutralTolerance = 1.0e-6;
inline constexpr int kMaxCostTerms = 128;
inline constexpr int kMaxTraces = 99;

class Task;

class ResidualFn {
 public:
  virtual ~ResidualFn() = default;
  virtual void Residual(const mjModel* model, const mjData* data, double* residual) const = 0;
  virtual void CostTerms(double* terms, const double* residual, bool weighted) const = 0;
  virtual double CostValue(const double* residual) const = 0;
  virtual void Update() = 0;
};

class BaseResidualFn : public ResidualFn {
 public:
  explicit BaseResidualFn(const Task* task);
  void CostTerms(double* terms, const double* residual, bool weighted) const override;
  double CostValue(const double* residual) const override;
  void Update() override;

 protected:
  int num_residual_, num_term_, num_trace_;
  std::vector<int> dim_norm_residual_, num_norm_parameter_;
  std::vector<NormType> norm_;
  std::vector<double> weight_, norm_parameter_;
  double risk_;
  std::vector<double> parameters_;
  const Task* task_;
};

class Task {
 public:
  Task() = default;
  virtual ~Task() = default;

  std::unique_ptr<ResidualFn> Residual() const;  // frozen copy for the planner (agent.cc:319)
  void Residual(const mjModel* model, const mjData* data, double* residual) const;
  void UpdateResidual();
  void Transition(mjModel* model, mjData* data);
  void Reset(const mjModel* model);  // parses the cost specification; throws std::runtime_error where the reference aborts
  void CostTerms(double* terms, const double* residual) const;
  void UnweightedCostTerms(double* terms, const double* residual) const;
  double CostValue(const double* residual) const;

  virtual std::string Name() const = 0;
  virtual std::string XmlPath() const = 0;
  virtual int DeviceResidualId() const { return 0; }
  // the task-specific members of the frozen ResidualFn copy a device residual needs (mjpcx_task::residual_int/real);
  // layout per residual in csrc/wave_residual.h. Tasks whose residual only reads `parameters` leave them empty.
  virtual void ResidualState(std::vector<int32_t>* ints, std::vector<double>* reals) const { ints->clear(); reals->clear(); }

  int mode = 0;
  int reset = 0, visualize = 0;
  int num_residual = 0, num_term = 0, num_trace = 0;
  std::vector<int> dim_norm_residual, num_norm_parameter;
  std::vector<NormType> norm;
  std::vector<double> weight;
  std::vector<std::string> weight_names;
  std::vector<double> norm_parameter;
  double risk = 0;
  std::vector<double> parameters;
  // which of them are "residual_select_*" drop-downs (an integer's bits in the double: utilities.h ReinterpretAsInt), and the
  // decoded copy handed to the device, whose kernels read numbers
  std::vector<unsigned char> parameter_is_selection;
  std::vector<double> NumericParameters() const;
  std::vector<int> trace_site;  // site id behind sensor "trace%i" (resolved once instead of per GetTraces call)

 protected:
  virtual BaseResidualFn* InternalResidual() = 0;
  const BaseResidualFn* InternalResidual() const { return const_cast<Task*>(this)->InternalResidual(); }
  virtual std::unique_ptr<ResidualFn> ResidualLocked() const = 0;
  virtual void TransitionLocked(mjModel* model, mjData* data) {}
  virtual void ResetLocked(const mjModel* model) {}
  mutable std::mutex mutex_;

 private:
  void SetFeatureParameters(const mjModel* model);
};

}  // namespace mjpc
