// The tasks whose residuals have a device implementation: Cartpole (mjpc/tasks/cartpole) and the two
// particle test tasks of the reference's test-suite (mjpc/test/testdata/particle_residual.h,
// mjpc/test/agent/rollout_test.cc:28-58).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../task.h"

namespace mjpc {

#define MJPC_DECLARE_TASK(CLASS)                                                          \
  class CLASS : public Task {                                                             \
   public:                                                                                \
    CLASS() : residual_(this) {}                                                          \
    std::string Name() const override;                                                    \
    std::string XmlPath() const override;                                                 \
    int DeviceResidualId() const override;                                                \
    class ResidualFn : public BaseResidualFn {                                            \
     public:                                                                              \
      explicit ResidualFn(const CLASS* task) : BaseResidualFn(task) {}                    \
      void Residual(const mjModel* model, const mjData* data, double* residual) const override; \
    };                                                                                    \
                                                                                          \
   protected:                                                                             \
    std::unique_ptr<mjpc::ResidualFn> ResidualLocked() const override {                   \
      return std::make_unique<ResidualFn>(residual_);                                     \
    }                                                                                     \
    ResidualFn* InternalResidual() override { return &residual_; }                        \
                                                                                          \
   private:                                                                               \
    ResidualFn residual_;                                                                 \
  };

MJPC_DECLARE_TASK(Cartpole)
MJPC_DECLARE_TASK(ParticleTestTask)
MJPC_DECLARE_TASK(ParticleCopyTestTask)
#undef MJPC_DECLARE_TASK

std::vector<std::shared_ptr<Task>> GetTasks();

}  // namespace mjpc
