// iLQGSettings (mjpc/planners/ilqg/settings.h)
#pragma once
namespace mjpc {
struct iLQGSettings {
  double min_linesearch_step = 1.0e-3;  // minimum step size for the line search
  double fd_tolerance = 1.0e-6;         // finite-difference tolerance
  double fd_mode = 0;                   // 0: forward, 1: centred
  double min_regularization = 1.0e-6;
  double max_regularization = 1.0e6;
  int regularization_type = 0;          // 0: control, 1: state-control, 2: value
  int max_regularization_iterations = 5;
  int action_limits = 1;                // box-QP on ctrlrange in the backward pass
  int nominal_feedback_scaling = 1;     // line search over feedback scaling in the nominal rollouts
  int verbose = 0;
};
}  // namespace mjpc
