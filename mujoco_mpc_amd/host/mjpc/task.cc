#include "task.h"

#include <cmath>
#include <cstring>
#include <stdexcept>

#include "utilities.h"

namespace mjpc {
namespace {
bool StartsWith(const char* s, const char* prefix) { return std::strncmp(s, prefix, std::strlen(prefix)) == 0; }
[[noreturn]] void MissingParameter(const mjModel* m, int sensor) {
  throw std::runtime_error(std::string("Cost construction from XML: Missing parameter value. sensor ID = ") +
                           std::to_string(sensor) + " (" + (m->names + m->name_sensoradr[sensor]) + ")");
}
}  // namespace

BaseResidualFn::BaseResidualFn(const Task* task) : task_(task) { Update(); }

void BaseResidualFn::Update() {
  num_residual_ = task_->num_residual;
  num_term_ = task_->num_term;
  num_trace_ = task_->num_trace;
  dim_norm_residual_ = task_->dim_norm_residual;
  num_norm_parameter_ = task_->num_norm_parameter;
  norm_ = task_->norm;
  weight_ = task_->weight;
  norm_parameter_ = task_->norm_parameter;
  risk_ = task_->risk;
  parameters_ = task_->parameters;
}

void BaseResidualFn::CostTerms(double* terms, const double* residual, bool weighted) const {
  const double* r = residual;
  const double* p = norm_parameter_.data();
  for (int k = 0; k < num_term_; k++) {
    terms[k] = (weighted ? weight_[k] : 1.0) * Norm(nullptr, nullptr, r, p, dim_norm_residual_[k], norm_[k]);
    r += dim_norm_residual_[k];
    p += num_norm_parameter_[k];
  }
}

double BaseResidualFn::CostValue(const double* residual) const {
  double terms[kMaxCostTerms];
  CostTerms(terms, residual, /*weighted=*/true);
  double cost = 0.0;
  for (int k = 0; k < num_term_; k++) cost += terms[k];
  if (std::fabs(risk_) < kRiskNeutralTolerance) return cost;
  return (std::exp(risk_ * cost) - 1.0) / risk_;  // exponential risk transformation
}

std::unique_ptr<ResidualFn> Task::Residual() const {
  std::lock_guard<std::mutex> lock(mutex_);
  return ResidualLocked();
}
void Task::Residual(const mjModel* model, const mjData* data, double* residual) const {
  std::lock_guard<std::mutex> lock(mutex_);
  InternalResidual()->Residual(model, data, residual);
}
void Task::UpdateResidual() {
  std::lock_guard<std::mutex> lock(mutex_);
  InternalResidual()->Update();
}
void Task::Transition(mjModel* model, mjData* data) {
  std::lock_guard<std::mutex> lock(mutex_);
  TransitionLocked(model, data);
  InternalResidual()->Update();
}
void Task::CostTerms(double* terms, const double* residual) const {
  std::lock_guard<std::mutex> lock(mutex_);
  InternalResidual()->CostTerms(terms, residual, true);
}
void Task::UnweightedCostTerms(double* terms, const double* residual) const {
  std::lock_guard<std::mutex> lock(mutex_);
  InternalResidual()->CostTerms(terms, residual, false);
}
double Task::CostValue(const double* residual) const {
  std::lock_guard<std::mutex> lock(mutex_);
  return InternalResidual()->CostValue(residual);
}

void Task::SetFeatureParameters(const mjModel* model) {
  // task.cc:38-64: "residual_select_*" fields carry an integer's bits (DefaultResidualSelection), the others their value
  parameters.clear();
  parameter_is_selection.clear();
  for (int i = 0; i < model->nnumeric; i++) {
    const char* name = model->names + model->name_numericadr[i];
    if (StartsWith(name, "residual_select_")) {
      parameters.push_back(DefaultResidualSelection(model, i));
      parameter_is_selection.push_back(1);
    } else if (StartsWith(name, "residual_")) {
      parameters.push_back(model->numeric_data[model->numeric_adr[i]]);
      parameter_is_selection.push_back(0);
    }
  }
}

// what the device kernels read (include/mjpcx.h mjpcx_task.parameters): every field as a plain number
std::vector<double> Task::NumericParameters() const {
  std::vector<double> out(parameters);
  for (size_t i = 0; i < out.size() && i < parameter_is_selection.size(); i++)
    if (parameter_is_selection[i]) out[i] = (double)ReinterpretAsInt(parameters[i]);
  return out;
}

void Task::Reset(const mjModel* model) {
  std::lock_guard<std::mutex> lock(mutex_);
  mode = 0;
  risk = GetNumberOrDefault(0.0, model, "task_risk");
  // the leading run of user sensors defines the cost terms: [norm, weight, w_lo, w_hi, params...]
  if (model->nsensor == 0 || model->sensor_type[0] != mjSENS_USER)
    throw std::runtime_error("Cost construction from XML: User sensors specifying residuals must be specified first and sequentially");
  num_term = model->nsensor;
  for (int i = 1; i < model->nsensor; i++)
    if (model->sensor_type[i] != mjSENS_USER) { num_term = i; break; }
  if (num_term > kMaxCostTerms) throw std::runtime_error("Number of cost terms exceeds maximum.");
  num_trace = 0;
  for (int i = 0; i < model->nsensor; i++)
    if (StartsWith(model->names + model->name_sensoradr[i], "trace")) num_trace++;
  if (num_trace > kMaxTraces) throw std::runtime_error("Number of traces should be less than 100");
  trace_site.assign(num_trace, -1);
  for (int i = 0; i < model->nsensor; i++) {
    const char* name = model->names + model->name_sensoradr[i];
    if (!StartsWith(name, "trace")) continue;
    const int k = std::atoi(name + 5);
    if (k < 0 || k >= num_trace || model->sensor_type[i] != mjSENS_FRAMEPOS) continue;
    // site id, or -1 - body id for the frame of a body (mjpcx_task::trace_site)
    if (model->sensor_objtype[i] == mjOBJ_SITE) trace_site[k] = model->sensor_objid[i];
    else if (model->sensor_objtype[i] == mjOBJ_BODY || model->sensor_objtype[i] == mjOBJ_XBODY) trace_site[k] = -1 - model->sensor_objid[i];
  }
  num_residual = 0;
  dim_norm_residual.assign(num_term, 0);
  num_norm_parameter.assign(num_term, 0);
  norm.assign(num_term, kQuadratic);
  weight.assign(num_term, 0.0);
  weight_names.assign(num_term, "");
  norm_parameter.clear();
  for (int i = 0; i < num_term; i++) {
    const double* s = model->sensor_user + (size_t)i * model->nuser_sensor;
    const int npar = NormParameterDimension((int)s[0]);
    if (4 + npar > model->nuser_sensor) MissingParameter(model, i);
    for (int j = 0; j < npar; j++)
      if (s[4 + j] <= 0.0) MissingParameter(model, i);
    if ((int)s[0] == kNull && model->sensor_dim[i] != 1) MissingParameter(model, i);
    num_residual += model->sensor_dim[i];
    dim_norm_residual[i] = model->sensor_dim[i];
    norm[i] = (NormType)(int)s[0];
    weight[i] = s[1];
    weight_names[i] = model->names + model->name_sensoradr[i];
    num_norm_parameter[i] = npar;
    norm_parameter.insert(norm_parameter.end(), s + 4, s + 4 + npar);
  }
  SetFeatureParameters(model);
  ResetLocked(model);
  InternalResidual()->Update();
}

}  // namespace mjpc
