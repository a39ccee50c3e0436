// Host-side model of an XGBoost gbtree booster as the OH path uses it.
//
// The reference never touches tree data itself: it hands a file name to
// libxgboost 1.6.0 through XGBoosterLoadModel
// (/root/reference Shared/xgb_fortran_api.F90:19-23, called at
// OH_GridComp/OH_GridCompMod.F90:261).  What a booster must hold is therefore
// defined by xgboost 1.6.0's model schema; this header is the
// product's own in-memory form of it.
#pragma once
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace ohx {

struct OhxError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// One regression tree, node-indexed arrays in the file's node order.
struct Tree {
  std::vector<int32_t> left;        // -1 => leaf
  std::vector<int32_t> right;
  std::vector<int32_t> parent;      // file form: bit31 = "is left child", root = -1
  std::vector<uint32_t> feature;    // split feature index (0 for leaves)
  std::vector<uint8_t> default_left;
  std::vector<float> value;         // split condition, or leaf value at leaves
  std::vector<uint8_t> deleted;     // pruned slots kept by the file format
  // per-node training statistics carried through save/load untouched
  std::vector<float> loss_chg, sum_hess, base_weight;
  std::vector<int32_t> leaf_child_cnt;
  int32_t num_feature = 0;

  size_t size() const { return left.size(); }
  void resize(size_t n);
  bool is_leaf(size_t i) const { return left[i] == -1; }
};

struct Forest {
  float base_score = 0.5f;
  uint32_t num_feature = 0;
  int32_t num_class = 0;
  uint32_t num_target = 1;
  uint32_t major_version = 1, minor_version = 6;
  std::string objective = "reg:squarederror";
  std::string booster = "gbtree";
  std::vector<Tree> trees;
  std::vector<int32_t> tree_info;   // output group of each tree
  std::vector<std::pair<std::string, std::string>> attributes;
  std::vector<std::string> metrics;
  std::string poisson_max_delta_step;   // legacy binary, count:poisson only: one string between attributes and metrics
  bool legacy_binary = false;           // parsed from the legacy binary format (decides how base_score is read)
  // what the readers forgave: trailer sections prediction does not need, bookkeeping that disagrees
  std::vector<std::string> warnings;

  size_t total_nodes() const;
  int max_depth() const;
  // Checks every invariant the traversal kernels rely on; throws OhxError.
  void validate() const;
  // The value a prediction starts from.  xgboost 1.6.0 keeps the user's base_score in the file and starts
  // margins from obj->ProbToMargin(base_score) (learner.cc, LearnerConfiguration::ConfigureModelParam);
  // binary files written before 1.0 hold the transformed value already.  Identity for the OH model
  // (reg:squarederror / reg:linear).  Throws for an objective whose ProbToMargin is not known here.
  float margin_base() const;
};

// ---- file formats ----
// Dispatch as xgboost 1.6.0's XGBoosterLoadModel does: ".json" => JSON,
// ".ubj" => UBJSON, anything else => legacy binary, where a leading '{' still
// selects JSON (or UBJSON, told apart by the byte that follows).
Forest load_model_file(const std::string& path);
Forest load_model_buffer(const void* buf, size_t len);
Forest parse_legacy_binary(const uint8_t* p, size_t len);
Forest parse_json_model(const char* text, size_t len);
Forest parse_ubjson_model(const uint8_t* p, size_t len);

void save_model_file(const Forest& f, const std::string& path);
std::vector<uint8_t> write_legacy_binary(const Forest& f);
std::string write_json_model(const Forest& f);
std::vector<uint8_t> write_ubjson_model(const Forest& f);

// Objectives whose prediction transform is the identity (the OH model is
// reg:squarederror; files written by xgboost < 1.0 call it reg:linear).
bool objective_is_identity(const std::string& name);
// obj->ProbToMargin of xgboost 1.6.0 by objective name; false when the name is not known
bool prob_to_margin(const std::string& objective, float base_score, float* margin);

}  // namespace ohx
