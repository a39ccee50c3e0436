// Model file reader/writer: XGBoost legacy binary, JSON and UBJSON (schema as of 1.6.0).
//
// Replaces what libxgboost does behind XGBoosterLoadModel / XGBoosterSaveModel
// (/root/reference Shared/xgb_fortran_api.F90:19-31).  The production OH models
// are legacy-binary ".model"/".bin" files (OH_GridComp/OH_instance_OH.rc:17-20).
// Layout restated from the published xgboost 1.6.0 schema;
// there is no model file in the reference tree to check it against.
#include <algorithm>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

#include "forest.hpp"
#include "json_min.hpp"

namespace ohx {

// ---------------------------------------------------------------- Tree / Forest

void Tree::resize(size_t n) {
  left.assign(n, -1);
  right.assign(n, -1);
  parent.assign(n, -1);
  feature.assign(n, 0);
  default_left.assign(n, 0);
  value.assign(n, 0.0f);
  deleted.assign(n, 0);
  loss_chg.assign(n, 0.0f);
  sum_hess.assign(n, 0.0f);
  base_weight.assign(n, 0.0f);
  leaf_child_cnt.assign(n, 0);
}

size_t Forest::total_nodes() const {
  size_t n = 0;
  for (const auto& t : trees) n += t.size();
  return n;
}

static int tree_depth(const Tree& t) {
  // iterative: depth of the deepest leaf, root at depth 0
  int best = 0;
  std::vector<std::pair<int32_t, int>> stack;
  stack.emplace_back(0, 0);
  while (!stack.empty()) {
    auto [n, d] = stack.back();
    stack.pop_back();
    if (t.left[(size_t)n] == -1) {
      best = std::max(best, d);
    } else {
      stack.emplace_back(t.left[(size_t)n], d + 1);
      stack.emplace_back(t.right[(size_t)n], d + 1);
    }
  }
  return best;
}

int Forest::max_depth() const {
  int d = 0;
  for (const auto& t : trees) d = std::max(d, tree_depth(t));
  return d;
}

bool objective_is_identity(const std::string& name) {
  return name == "reg:squarederror" || name == "reg:linear" || name == "reg:squaredlogerror" ||
         name == "reg:pseudohubererror" || name == "reg:absoluteerror";
}

// src/objective/regression_obj.cu, regression_loss.h, aft_obj.cu, rank_obj.cu, hinge.cu of xgboost 1.6.0:
// logistic losses return -log(1/p - 1), the log-link objectives log(p), everything else the value itself.
bool prob_to_margin(const std::string& o, float base_score, float* margin) {
  if (objective_is_identity(o) || o == "binary:hinge" || o == "rank:pairwise" || o == "rank:ndcg" || o == "rank:map") {
    *margin = base_score;
    return true;
  }
  if (o == "reg:logistic" || o == "binary:logistic" || o == "binary:logitraw") {
    if (!(base_score > 0.0f && base_score < 1.0f))
      throw OhxError("base_score must be in (0,1) for logistic loss, got: " + std::to_string(base_score));
    *margin = -logf(1.0f / base_score - 1.0f);
    return true;
  }
  if (o == "count:poisson" || o == "reg:gamma" || o == "reg:tweedie" || o == "survival:cox" || o == "survival:aft") {
    *margin = logf(base_score);
    return true;
  }
  return false;
}

// JSON/UBJSON hold the user's base_score; a binary file from xgboost < 1.0 holds the margin.  The two are
// the same number for identity objectives only.
static void check_base_score_portable(const Forest& f) {
  if (f.legacy_binary && f.major_version < 1 && !objective_is_identity(f.objective))
    throw OhxError("a pre-1.0 binary model with objective '" + f.objective + "' cannot be re-saved as JSON/UBJSON");
}

float Forest::margin_base() const {
  if (legacy_binary && major_version < 1) return base_score;   // written by xgboost < 1.0: already a margin
  float m = base_score;
  if (!prob_to_margin(objective, base_score, &m))
    throw OhxError("objective '" + objective + "': the margin its base_score stands for is not known to this library");
  return m;
}

void Forest::validate() const {
  if (booster != "gbtree" && booster != "dart")
    throw OhxError("unsupported booster '" + booster + "' (the OH path uses gbtree)");
  if (booster == "dart") throw OhxError("dart boosters are not supported (tree weights)");
  if (tree_info.size() != trees.size()) throw OhxError("tree_info length does not match the number of trees");
  for (size_t ti = 0; ti < trees.size(); ++ti) {
    const Tree& t = trees[ti];
    const size_t n = t.size();
    if (n == 0) throw OhxError("tree " + std::to_string(ti) + " has no nodes");
    std::vector<uint8_t> seen(n, 0);
    std::vector<int32_t> stack{0};
    size_t reached = 0;
    while (!stack.empty()) {
      int32_t i = stack.back();
      stack.pop_back();
      if (i < 0 || (size_t)i >= n) throw OhxError("tree " + std::to_string(ti) + ": child index out of range");
      if (seen[(size_t)i]) throw OhxError("tree " + std::to_string(ti) + ": node reached twice (not a tree)");
      if (t.deleted[(size_t)i]) throw OhxError("tree " + std::to_string(ti) + ": a deleted node is reachable");
      seen[(size_t)i] = 1;
      ++reached;
      if (t.left[(size_t)i] == -1) continue;
      // xgboost 1.6.0's predictor steps to LeftChild() + !(fvalue < cond): it
      // takes right == left + 1 for granted; a file that breaks it is refused.
      if ((int64_t)t.right[(size_t)i] != (int64_t)t.left[(size_t)i] + 1)
        throw OhxError("tree " + std::to_string(ti) + ": right child is not left child + 1");
      if (num_feature != 0 && t.feature[(size_t)i] >= num_feature)
        throw OhxError("tree " + std::to_string(ti) + ": split feature " + std::to_string(t.feature[(size_t)i]) +
                       " >= num_feature " + std::to_string(num_feature));
      stack.push_back(t.left[(size_t)i]);
      stack.push_back(t.right[(size_t)i]);
    }
    (void)reached;
  }
  for (int32_t g : tree_info)
    if (g != 0) throw OhxError("multi-group (multi-class) boosters are not supported by the OH predictor");
}

// ---------------------------------------------------------------- legacy binary

namespace {

struct Reader {
  const uint8_t* p;
  size_t len, off = 0;
  void need(size_t n, const char* what) const {
    if (off + n > len) throw OhxError(std::string("legacy binary model truncated while reading ") + what);
  }
  template <class T>
  T get(const char* what) {
    need(sizeof(T), what);
    T v;
    memcpy(&v, p + off, sizeof(T));
    off += sizeof(T);
    return v;
  }
  void skip(size_t n, const char* what) {
    need(n, what);
    off += n;
  }
  std::string str(const char* what) {
    uint64_t n = get<uint64_t>(what);
    if (n > (1u << 20)) throw OhxError(std::string("legacy binary model: implausible string length for ") + what);
    need((size_t)n, what);
    std::string s((const char*)p + off, (size_t)n);
    off += (size_t)n;
    return s;
  }
};

struct Writer {
  std::vector<uint8_t> buf;
  template <class T>
  void put(T v) {
    size_t o = buf.size();
    buf.resize(o + sizeof(T));
    memcpy(buf.data() + o, &v, sizeof(T));
  }
  void zeros(size_t n) { buf.resize(buf.size() + n, 0); }
  void str(const std::string& s) {
    put<uint64_t>(s.size());
    buf.insert(buf.end(), s.begin(), s.end());
  }
};

constexpr uint32_t kDeletedMarker = 0xFFFFFFFFu;

}  // namespace

Forest parse_legacy_binary(const uint8_t* p, size_t len) {
  Reader r{p, len};
  Forest f;
  if (len >= 4 && memcmp(p, "bs64", 4) == 0) throw OhxError("base64 model files are not supported");
  if (len >= 4 && memcmp(p, "binf", 4) == 0) r.off = 4;
  // LearnerModelParamLegacy, 136 bytes
  f.base_score = r.get<float>("base_score");
  f.num_feature = r.get<uint32_t>("num_feature");
  f.num_class = r.get<int32_t>("num_class");
  int32_t contain_extra_attrs = r.get<int32_t>("contain_extra_attrs");
  int32_t contain_eval_metrics = r.get<int32_t>("contain_eval_metrics");
  f.major_version = r.get<uint32_t>("major_version");
  f.minor_version = r.get<uint32_t>("minor_version");
  f.num_target = r.get<uint32_t>("num_target");
  if (f.num_target == 0) f.num_target = 1;
  r.skip(26 * 4, "learner reserved");
  f.objective = r.str("objective name");
  f.booster = r.str("booster name");
  if (f.booster != "gbtree") throw OhxError("legacy binary model: booster '" + f.booster + "' is not gbtree");
  // GBTreeModelParam, 160 bytes
  int32_t num_trees = r.get<int32_t>("num_trees");
  r.skip(4 + 4 + 4 + 8 + 4, "gbtree deprecated fields");
  int32_t size_leaf_vector = r.get<int32_t>("size_leaf_vector");
  r.skip(32 * 4, "gbtree reserved");
  if (num_trees < 0 || num_trees > (1 << 24)) throw OhxError("legacy binary model: implausible num_trees");
  if (size_leaf_vector != 0) throw OhxError("legacy binary model: vector leaves are not supported");
  f.trees.resize((size_t)num_trees);
  for (int32_t ti = 0; ti < num_trees; ++ti) {
    Tree& t = f.trees[(size_t)ti];
    // TreeParam, 148 bytes
    r.skip(4, "tree num_roots");
    int32_t num_nodes = r.get<int32_t>("tree num_nodes");
    int32_t num_deleted = r.get<int32_t>("tree num_deleted");
    r.skip(4, "tree max_depth");
    t.num_feature = r.get<int32_t>("tree num_feature");
    int32_t slv = r.get<int32_t>("tree size_leaf_vector");
    r.skip(31 * 4, "tree reserved");
    if (num_nodes <= 0) throw OhxError("legacy binary model: tree with no nodes");
    if (slv != 0) throw OhxError("legacy binary model: vector leaves are not supported");
    r.need((size_t)num_nodes * 36, "tree nodes");
    t.resize((size_t)num_nodes);
    int32_t deleted = 0;
    for (int32_t i = 0; i < num_nodes; ++i) {
      // Node, 20 bytes: parent, cleft, cright, sindex, info
      t.parent[(size_t)i] = r.get<int32_t>("node");
      t.left[(size_t)i] = r.get<int32_t>("node");
      t.right[(size_t)i] = r.get<int32_t>("node");
      uint32_t sindex = r.get<uint32_t>("node");
      t.value[(size_t)i] = r.get<float>("node");
      if (sindex == kDeletedMarker) {
        t.deleted[(size_t)i] = 1;
        if (i > 0) ++deleted;
        t.left[(size_t)i] = -1;
        t.right[(size_t)i] = -1;
      } else {
        t.default_left[(size_t)i] = (uint8_t)(sindex >> 31);
        t.feature[(size_t)i] = sindex & 0x7FFFFFFFu;
      }
      if (t.left[(size_t)i] == -1) {
        t.feature[(size_t)i] = t.deleted[(size_t)i] ? 0u : t.feature[(size_t)i];
      }
    }
    for (int32_t i = 0; i < num_nodes; ++i) {
      // RTreeNodeStat, 16 bytes
      t.loss_chg[(size_t)i] = r.get<float>("node stat");
      t.sum_hess[(size_t)i] = r.get<float>("node stat");
      t.base_weight[(size_t)i] = r.get<float>("node stat");
      t.leaf_child_cnt[(size_t)i] = r.get<int32_t>("node stat");
    }
    // bookkeeping only: which slots are deleted is read from the node table itself
    if (deleted != num_deleted)
      f.warnings.push_back("tree " + std::to_string(ti) + ": num_deleted = " + std::to_string(num_deleted) + " but " +
                           std::to_string(deleted) + " slots of the node table are marked deleted");
  }
  f.tree_info.resize((size_t)num_trees);
  for (int32_t ti = 0; ti < num_trees; ++ti) f.tree_info[(size_t)ti] = r.get<int32_t>("tree_info");
  f.legacy_binary = true;
  // What follows - attributes, count:poisson's max_delta_step, metric names (learner.cc LearnerIO::Load) -
  // is not needed to predict: a trailer that does not parse is dropped with a warning, never an error.
  const size_t trailer = r.off;
  try {
    if (contain_extra_attrs != 0) {
      uint64_t n = r.get<uint64_t>("attribute count");
      if (n > (1u << 20)) throw OhxError("legacy binary model: implausible attribute count");
      for (uint64_t i = 0; i < n; ++i) {
        std::string k = r.str("attribute key");
        std::string v = r.str("attribute value");
        f.attributes.emplace_back(std::move(k), std::move(v));
      }
    }
    if (f.objective == "count:poisson") f.poisson_max_delta_step = r.str("max_delta_step");
    if (contain_eval_metrics != 0) {
      uint64_t n = r.get<uint64_t>("metric count");
      if (n > (1u << 20)) throw OhxError("legacy binary model: implausible metric count");
      for (uint64_t i = 0; i < n; ++i) f.metrics.push_back(r.str("metric name"));
    }
  } catch (const OhxError& e) {
    f.attributes.clear();
    f.metrics.clear();
    f.poisson_max_delta_step.clear();
    f.warnings.push_back(std::string("trailer after tree_info ignored (") + e.what() + "), " +
                         std::to_string(len - trailer) + " bytes");
  }
  return f;
}

std::vector<uint8_t> write_legacy_binary(const Forest& f) {
  Writer w;
  size_t nodes = f.total_nodes();
  w.buf.reserve(4 + 136 + 64 + 160 + f.trees.size() * 152 + nodes * 36 + 64);
  w.buf.insert(w.buf.end(), {'b', 'i', 'n', 'f'});
  w.put<float>(f.base_score);
  w.put<uint32_t>(f.num_feature);
  w.put<int32_t>(f.num_class);
  w.put<int32_t>(f.attributes.empty() ? 0 : 1);
  w.put<int32_t>(f.metrics.empty() ? 0 : 1);
  w.put<uint32_t>(f.major_version);
  w.put<uint32_t>(f.minor_version);
  w.put<uint32_t>(f.num_target);
  w.zeros(26 * 4);
  w.str(f.objective);
  w.str(f.booster);
  w.put<int32_t>((int32_t)f.trees.size());
  w.put<int32_t>(1);                      // num_roots (deprecated)
  w.put<int32_t>((int32_t)f.num_feature); // num_feature (deprecated)
  w.put<int32_t>(0);                      // pad
  w.put<int64_t>(0);                      // num_pbuffer (deprecated)
  w.put<int32_t>(1);                      // num_output_group (deprecated)
  w.put<int32_t>(0);                      // size_leaf_vector
  w.zeros(32 * 4);
  for (const Tree& t : f.trees) {
    int32_t n = (int32_t)t.size();
    int32_t num_deleted = 0;
    for (int32_t i = 1; i < n; ++i) num_deleted += t.deleted[(size_t)i] ? 1 : 0;
    w.put<int32_t>(1);
    w.put<int32_t>(n);
    w.put<int32_t>(num_deleted);
    w.put<int32_t>(0);
    w.put<int32_t>(t.num_feature ? t.num_feature : (int32_t)f.num_feature);
    w.put<int32_t>(0);
    w.zeros(31 * 4);
    for (int32_t i = 0; i < n; ++i) {
      w.put<int32_t>(t.parent[(size_t)i]);
      w.put<int32_t>(t.left[(size_t)i]);
      w.put<int32_t>(t.right[(size_t)i]);
      uint32_t sindex = t.deleted[(size_t)i] ? kDeletedMarker
                                             : (t.feature[(size_t)i] | ((uint32_t)(t.default_left[(size_t)i] ? 1u : 0u) << 31));
      w.put<uint32_t>(sindex);
      w.put<float>(t.value[(size_t)i]);
    }
    for (int32_t i = 0; i < n; ++i) {
      w.put<float>(t.loss_chg[(size_t)i]);
      w.put<float>(t.sum_hess[(size_t)i]);
      w.put<float>(t.base_weight[(size_t)i]);
      w.put<int32_t>(t.leaf_child_cnt[(size_t)i]);
    }
  }
  for (int32_t g : f.tree_info) w.put<int32_t>(g);
  if (!f.attributes.empty()) {
    w.put<uint64_t>(f.attributes.size());
    for (auto& kv : f.attributes) {
      w.str(kv.first);
      w.str(kv.second);
    }
  }
  if (f.objective == "count:poisson") w.str(f.poisson_max_delta_step.empty() ? "0.7" : f.poisson_max_delta_step);
  if (!f.metrics.empty()) {
    w.put<uint64_t>(f.metrics.size());
    for (auto& m : f.metrics) w.str(m);
  }
  return std::move(w.buf);
}

// ---------------------------------------------------------------- JSON

namespace {

double num_of(const json::Value& v, const char* what) {
  // xgboost writes scalar parameters as strings ("num_trees": "100")
  if (v.type == json::Value::String) {
    char* e = nullptr;
    double d = strtod(v.str.c_str(), &e);
    if (e == v.str.c_str()) throw OhxError(std::string("JSON model: '") + what + "' is not numeric");
    return d;
  }
  if (v.type == json::Value::Number) return v.num.d;
  if (v.type == json::Value::Bool) return v.b ? 1.0 : 0.0;
  throw OhxError(std::string("JSON model: '") + what + "' is not numeric");
}

float float_of(const json::Value& v, const char* what) {
  if (v.type == json::Value::String) {
    char* e = nullptr;
    float d = strtof(v.str.c_str(), &e);
    if (e == v.str.c_str()) throw OhxError(std::string("JSON model: '") + what + "' is not numeric");
    return d;
  }
  if (v.type == json::Value::Number) return v.num.f;
  throw OhxError(std::string("JSON model: '") + what + "' is not numeric");
}

const std::vector<json::Num>& nums_of(const json::Value& v, const char* what, size_t expect) {
  if (v.type != json::Value::NumArray)
    throw OhxError(std::string("JSON model: '") + what + "' is not a numeric array");
  if (v.nums.size() != expect)
    throw OhxError(std::string("JSON model: '") + what + "' has the wrong length");
  return v.nums;
}

void fmt_float(std::string& out, float v) {
  char b[40];
  if (v != v) { out += "NaN"; return; }
  if (v == INFINITY) { out += "Infinity"; return; }
  if (v == -INFINITY) { out += "-Infinity"; return; }
  snprintf(b, sizeof b, "%.9g", (double)v);
  out += b;
}

template <class T, class F>
void put_array(std::string& out, const char* key, const std::vector<T>& v, F fmt, bool comma = true) {
  out += '"';
  out += key;
  out += "\":[";
  for (size_t i = 0; i < v.size(); ++i) {
    if (i) out += ',';
    fmt(out, v[i]);
  }
  out += ']';
  if (comma) out += ',';
}

}  // namespace

static Forest forest_from_document(const json::Value& doc);

Forest parse_json_model(const char* text, size_t len) {
  json::Parser parser(text, len);
  return forest_from_document(parser.parse());
}

// ---------------------------------------------------------------- UBJSON
// Universal Binary JSON (draft 12) as xgboost >= 1.6 writes it for ".ubj": every scalar is
// big-endian, object keys are length-prefixed without the 'S' marker, and the per-node arrays are
// "optimized containers" ([$<type>#<count> followed by raw payloads).  The reader accepts the
// whole standard, the writer emits what xgboost emits; both build on the JSON document model.

namespace {

struct UbjReader {
  const uint8_t* p;
  size_t len, off = 0;
  [[noreturn]] void fail(const char* what) const { throw OhxError(std::string("UBJSON model: ") + what); }
  uint8_t byte() {
    if (off >= len) fail("truncated");
    return p[off++];
  }
  uint8_t peek() const {
    if (off >= len) throw OhxError("UBJSON model: truncated");
    return p[off];
  }
  template <class T>
  T be() {
    if (off + sizeof(T) > len) fail("truncated");
    uint8_t tmp[sizeof(T)];
    for (size_t i = 0; i < sizeof(T); ++i) tmp[i] = p[off + sizeof(T) - 1 - i];
    off += sizeof(T);
    T v;
    memcpy(&v, tmp, sizeof(T));
    return v;
  }
  int64_t integer(uint8_t t) {
    switch (t) {
      case 'i': return be<int8_t>();
      case 'U': return be<uint8_t>();
      case 'I': return be<int16_t>();
      case 'l': return be<int32_t>();
      case 'L': return be<int64_t>();
      default: fail("expected an integer type");
    }
  }
  std::string text(uint8_t len_type) {
    int64_t n = integer(len_type);
    if (n < 0 || (uint64_t)n > len - off) fail("bad string length");
    std::string s((const char*)p + off, (size_t)n);
    off += (size_t)n;
    return s;
  }
  json::Num number(uint8_t t) {
    if (t == 'd') { float f = be<float>(); return {(double)f, f}; }
    if (t == 'D') { double d = be<double>(); return {d, (float)d}; }
    int64_t v = integer(t);
    return {(double)v, (float)v};
  }
  static bool is_number(uint8_t t) { return strchr("iUIlLdD", (int)t) != nullptr && t != 0; }

  int depth = 0;
  // a count read from the file: there cannot be more elements than there are bytes left (payload-free
  // types - T, F, Z - included: no writer emits millions of those)
  int64_t container_count() {
    ++off;
    const int64_t count = integer(byte());
    if (count < 0 || (uint64_t)count > len - off + 64) fail("implausible element count");
    return count;
  }

  json::Value value(uint8_t t) {
    struct Depth {
      int& d;
      explicit Depth(int& dd) : d(dd) { ++d; }
      ~Depth() { --d; }
    } guard(depth);
    if (depth > 64) fail("nested too deeply");
    json::Value v;
    if (t == 'Z') return v;
    if (t == 'T' || t == 'F') { v.type = json::Value::Bool; v.b = (t == 'T'); return v; }
    if (is_number(t)) { v.type = json::Value::Number; v.num = number(t); return v; }
    if (t == 'C') { v.type = json::Value::String; v.str = std::string(1, (char)byte()); return v; }
    if (t == 'S') { v.type = json::Value::String; v.str = text(byte()); return v; }
    if (t == 'H') { v.type = json::Value::Number; std::string s = text(byte()); v.num = {strtod(s.c_str(), nullptr), strtof(s.c_str(), nullptr)}; return v; }
    if (t == '[') {
      uint8_t elem = 0;
      int64_t count = -1;
      if (peek() == '$') { ++off; elem = byte(); if (peek() != '#') fail("'$' without '#'"); }
      if (peek() == '#') count = container_count();
      if (elem != 0 && (is_number(elem) || elem == 'T' || elem == 'F')) {
        v.type = json::Value::NumArray;
        v.nums.reserve((size_t)count);
        for (int64_t i = 0; i < count; ++i) {
          if (elem == 'T') v.nums.push_back({1.0, 1.0f});
          else if (elem == 'F') v.nums.push_back({0.0, 0.0f});
          else v.nums.push_back(number(elem));
        }
        return v;
      }
      // general array; numeric-only ones are kept flat, like the JSON reader does
      std::vector<json::Value> items;
      bool numeric = true;
      for (int64_t i = 0; count < 0 || i < count; ++i) {
        uint8_t et = elem ? elem : byte();
        if (count < 0 && et == ']') break;
        if (et == 'N') { --i; continue; }
        items.push_back(value(et));
        numeric = numeric && (items.back().type == json::Value::Number || items.back().type == json::Value::Bool);
      }
      if (numeric) {
        v.type = json::Value::NumArray;
        for (auto& it : items) v.nums.push_back(it.type == json::Value::Bool ? json::Num{it.b ? 1.0 : 0.0, it.b ? 1.0f : 0.0f} : it.num);
      } else {
        v.type = json::Value::Array;
        v.arr = std::move(items);
      }
      return v;
    }
    if (t == '{') {
      v.type = json::Value::Object;
      uint8_t elem = 0;
      int64_t count = -1;
      if (peek() == '$') { ++off; elem = byte(); if (peek() != '#') fail("'$' without '#'"); }
      if (peek() == '#') count = container_count();
      for (int64_t i = 0; count < 0 || i < count; ++i) {
        uint8_t kt = byte();
        if (count < 0 && kt == '}') break;
        if (kt == 'N') { --i; continue; }
        std::string key = text(kt);
        uint8_t vt = elem ? elem : byte();
        v.obj.emplace(std::move(key), value(vt));
      }
      return v;
    }
    fail("unknown type marker");
  }
};

struct UbjWriter {
  std::vector<uint8_t> out;
  template <class T>
  void be(T v) {
    uint8_t tmp[sizeof(T)];
    memcpy(tmp, &v, sizeof(T));
    for (size_t i = 0; i < sizeof(T); ++i) out.push_back(tmp[sizeof(T) - 1 - i]);
  }
  void key(const std::string& k) { out.push_back('L'); be<int64_t>((int64_t)k.size()); out.insert(out.end(), k.begin(), k.end()); }
  void str(const std::string& s) { out.push_back('S'); key(s); }
  void kstr(const std::string& k, const std::string& v) { key(k); str(v); }
  void integer(int64_t v) { out.push_back('L'); be<int64_t>(v); }
  template <class T, class F>
  void typed(const std::string& k, char marker, const std::vector<T>& v, F conv) {
    key(k);
    out.push_back('[');
    out.push_back('$');
    out.push_back((uint8_t)marker);
    out.push_back('#');
    out.push_back('L');
    be<int64_t>((int64_t)v.size());
    for (const T& x : v) conv(x);
  }
};

}  // namespace

Forest parse_ubjson_model(const uint8_t* p, size_t len) {
  UbjReader r{p, len};
  if (len == 0 || p[0] != '{') throw OhxError("UBJSON model: the document is not an object");
  ++r.off;
  return forest_from_document(r.value('{'));
}

std::vector<uint8_t> write_ubjson_model(const Forest& f) {
  check_base_score_portable(f);
  UbjWriter w;
  auto f32 = [&](const std::string& k, const std::vector<float>& v) { w.typed(k, 'd', v, [&](float x) { w.be<float>(x); }); };
  auto i32 = [&](const std::string& k, const std::vector<int32_t>& v) { w.typed(k, 'l', v, [&](int32_t x) { w.be<int32_t>(x); }); };
  auto u8 = [&](const std::string& k, const std::vector<uint8_t>& v) { w.typed(k, 'U', v, [&](uint8_t x) { w.out.push_back(x); }); };
  auto empty = [&](const std::string& k) { w.key(k); w.out.push_back('['); w.out.push_back(']'); };
  auto fstr = [](float v) { char b[40]; snprintf(b, sizeof b, "%.9g", (double)v); return std::string(b); };
  w.out.push_back('{');
  w.key("learner"); w.out.push_back('{');
  w.key("attributes"); w.out.push_back('{');
  for (auto& kv : f.attributes) w.kstr(kv.first, kv.second);
  w.out.push_back('}');
  empty("feature_names");
  empty("feature_types");
  w.key("gradient_booster"); w.out.push_back('{');
  w.key("model"); w.out.push_back('{');
  w.key("gbtree_model_param"); w.out.push_back('{');
  w.kstr("num_parallel_tree", "1");
  w.kstr("num_trees", std::to_string(f.trees.size()));
  w.kstr("size_leaf_vector", "0");
  w.out.push_back('}');
  i32("tree_info", f.tree_info);
  w.key("trees"); w.out.push_back('[');
  for (size_t ti = 0; ti < f.trees.size(); ++ti) {
    const Tree& t = f.trees[ti];
    w.out.push_back('{');
    f32("base_weights", t.base_weight);
    empty("categories"); empty("categories_nodes"); empty("categories_segments"); empty("categories_sizes");
    u8("default_left", t.default_left);
    w.key("id"); w.integer((int64_t)ti);
    i32("left_children", t.left);
    f32("loss_changes", t.loss_chg);
    std::vector<int32_t> parents(t.size());
    for (size_t i = 0; i < t.size(); ++i)
      parents[i] = (i == 0 || t.parent[i] == -1) ? 2147483647 : (int32_t)((uint32_t)t.parent[i] & 0x7FFFFFFFu);
    i32("parents", parents);
    i32("right_children", t.right);
    f32("split_conditions", t.value);
    std::vector<int32_t> sidx(t.size());
    for (size_t i = 0; i < t.size(); ++i) sidx[i] = (int32_t)t.feature[i];
    i32("split_indices", sidx);
    std::vector<uint8_t> stype(t.size(), 0);
    u8("split_type", stype);
    f32("sum_hessian", t.sum_hess);
    w.key("tree_param"); w.out.push_back('{');
    int32_t num_deleted = 0;
    for (size_t i = 1; i < t.size(); ++i) num_deleted += t.deleted[i] ? 1 : 0;
    w.kstr("num_deleted", std::to_string(num_deleted));
    w.kstr("num_feature", std::to_string(t.num_feature ? t.num_feature : (int32_t)f.num_feature));
    w.kstr("num_nodes", std::to_string(t.size()));
    w.kstr("size_leaf_vector", "0");
    w.out.push_back('}');
    w.out.push_back('}');
  }
  w.out.push_back(']');
  w.out.push_back('}');           // model
  w.kstr("name", "gbtree");
  w.out.push_back('}');           // gradient_booster
  w.key("learner_model_param"); w.out.push_back('{');
  w.kstr("base_score", fstr(f.base_score));
  w.kstr("num_class", std::to_string(f.num_class));
  w.kstr("num_feature", std::to_string(f.num_feature));
  w.kstr("num_target", std::to_string(f.num_target));
  w.out.push_back('}');
  w.key("objective"); w.out.push_back('{');
  w.kstr("name", f.objective);
  w.key("reg_loss_param"); w.out.push_back('{'); w.kstr("scale_pos_weight", "1"); w.out.push_back('}');
  w.out.push_back('}');
  w.out.push_back('}');           // learner
  w.key("version"); w.out.push_back('[');
  w.integer(1); w.integer(6); w.integer(0);
  w.out.push_back(']');
  w.out.push_back('}');
  return std::move(w.out);
}

static Forest forest_from_document(const json::Value& doc) {
  Forest f;
  if (const json::Value* ver = doc.find("version")) {
    if (ver->type == json::Value::NumArray && ver->nums.size() >= 2) {
      f.major_version = (uint32_t)ver->nums[0].d;
      f.minor_version = (uint32_t)ver->nums[1].d;
    }
  }
  const json::Value& learner = doc.at("learner");
  const json::Value& lmp = learner.at("learner_model_param");
  f.base_score = float_of(lmp.at("base_score"), "base_score");
  f.num_feature = (uint32_t)num_of(lmp.at("num_feature"), "num_feature");
  f.num_class = (int32_t)num_of(lmp.at("num_class"), "num_class");
  if (const json::Value* nt = lmp.find("num_target")) f.num_target = (uint32_t)num_of(*nt, "num_target");
  f.objective = learner.at("objective").at("name").str;
  const json::Value& gb = learner.at("gradient_booster");
  f.booster = gb.at("name").str;
  if (f.booster != "gbtree") throw OhxError("JSON model: booster '" + f.booster + "' is not gbtree");
  const json::Value& model = gb.at("model");
  size_t num_trees = (size_t)num_of(model.at("gbtree_model_param").at("num_trees"), "num_trees");
  const json::Value& trees = model.at("trees");
  if (trees.array_size() != num_trees) throw OhxError("JSON model: num_trees does not match the trees array");
  f.trees.resize(num_trees);
  for (size_t ti = 0; ti < num_trees; ++ti) {
    const json::Value& jt = trees.arr[ti];
    Tree& t = f.trees[ti];
    const json::Value& tp = jt.at("tree_param");
    size_t n = (size_t)num_of(tp.at("num_nodes"), "num_nodes");
    if (n == 0) throw OhxError("JSON model: tree with no nodes");
    t.num_feature = (int32_t)num_of(tp.at("num_feature"), "tree num_feature");
    if (const json::Value* slv = tp.find("size_leaf_vector"))
      if (num_of(*slv, "size_leaf_vector") > 1) throw OhxError("JSON model: vector leaves are not supported");
    t.resize(n);
    auto& lc = nums_of(jt.at("left_children"), "left_children", n);
    auto& rc = nums_of(jt.at("right_children"), "right_children", n);
    auto& pa = nums_of(jt.at("parents"), "parents", n);
    auto& si = nums_of(jt.at("split_indices"), "split_indices", n);
    auto& sc = nums_of(jt.at("split_conditions"), "split_conditions", n);
    auto& dl = nums_of(jt.at("default_left"), "default_left", n);
    const std::vector<json::Num>* st = nullptr;
    if (const json::Value* v = jt.find("split_type")) st = &nums_of(*v, "split_type", n);
    if (const json::Value* cn = jt.find("categories_nodes"))
      if (cn->array_size() != 0) throw OhxError("JSON model: categorical splits are not supported");
    const std::vector<json::Num>* lch = nullptr;
    const std::vector<json::Num>* sh = nullptr;
    const std::vector<json::Num>* bw = nullptr;
    if (const json::Value* v = jt.find("loss_changes")) lch = &nums_of(*v, "loss_changes", n);
    if (const json::Value* v = jt.find("sum_hessian")) sh = &nums_of(*v, "sum_hessian", n);
    if (const json::Value* v = jt.find("base_weights")) bw = &nums_of(*v, "base_weights", n);
    for (size_t i = 0; i < n; ++i) {
      t.left[i] = (int32_t)lc[i].d;
      t.right[i] = (int32_t)rc[i].d;
      // JSON stores the plain parent id, 2147483647 for the root
      int64_t par = (int64_t)pa[i].d;
      t.parent[i] = (par >= 2147483647LL || par < 0) ? -1 : (int32_t)par;
      t.feature[i] = (uint32_t)si[i].d;
      t.value[i] = sc[i].f;
      t.default_left[i] = dl[i].d != 0.0 ? 1 : 0;
      if (st && (*st)[i].d != 0.0) throw OhxError("JSON model: categorical splits are not supported");
      if (lch) t.loss_chg[i] = (*lch)[i].f;
      if (sh) t.sum_hess[i] = (*sh)[i].f;
      if (bw) t.base_weight[i] = (*bw)[i].f;
    }
    // the binary form tags "is left child" in bit 31 of parent: rebuild it
    for (size_t i = 0; i < n; ++i) {
      if (t.left[i] != -1) {
        int32_t l = t.left[i], r = t.right[i];
        if (l >= 0 && (size_t)l < n) t.parent[(size_t)l] = (int32_t)((uint32_t)i | 0x80000000u);
        if (r >= 0 && (size_t)r < n) t.parent[(size_t)r] = (int32_t)i;
      }
    }
    // deleted slots in JSON: xgboost marks them with split index == max
    for (size_t i = 1; i < n; ++i)
      if (si[i].d >= 4294967295.0) { t.deleted[i] = 1; t.left[i] = t.right[i] = -1; t.feature[i] = 0; }
  }
  auto& info = nums_of(model.at("tree_info"), "tree_info", num_trees);
  f.tree_info.resize(num_trees);
  for (size_t i = 0; i < num_trees; ++i) f.tree_info[i] = (int32_t)info[i].d;
  if (const json::Value* attrs = learner.find("attributes")) {
    if (attrs->type == json::Value::Object)
      for (auto& kv : attrs->obj)
        if (kv.second.type == json::Value::String) f.attributes.emplace_back(kv.first, kv.second.str);
  }
  return f;
}

std::string write_json_model(const Forest& f) {
  check_base_score_portable(f);
  std::string o;
  o.reserve(256 + f.total_nodes() * 96);
  auto fi = [](std::string& s, int32_t v) { s += std::to_string(v); };
  auto fu = [](std::string& s, uint32_t v) { s += std::to_string(v); };
  auto fb = [](std::string& s, uint8_t v) { s += (v ? '1' : '0'); };
  auto ff = [](std::string& s, float v) { fmt_float(s, v); };
  o += "{\"learner\":{\"attributes\":{";
  for (size_t i = 0; i < f.attributes.size(); ++i) {
    if (i) o += ',';
    o += '"' + f.attributes[i].first + "\":\"" + f.attributes[i].second + '"';
  }
  o += "},\"feature_names\":[],\"feature_types\":[],\"gradient_booster\":{\"model\":{";
  o += "\"gbtree_model_param\":{\"num_parallel_tree\":\"1\",\"num_trees\":\"" + std::to_string(f.trees.size()) +
       "\",\"size_leaf_vector\":\"0\"},";
  put_array(o, "tree_info", f.tree_info, fi);
  o += "\"trees\":[";
  for (size_t ti = 0; ti < f.trees.size(); ++ti) {
    const Tree& t = f.trees[ti];
    if (ti) o += ',';
    o += '{';
    put_array(o, "base_weights", t.base_weight, ff);
    o += "\"categories\":[],\"categories_nodes\":[],\"categories_segments\":[],\"categories_sizes\":[],";
    put_array(o, "default_left", t.default_left, fb);
    o += "\"id\":" + std::to_string(ti) + ',';
    put_array(o, "left_children", t.left, fi);
    put_array(o, "loss_changes", t.loss_chg, ff);
    std::vector<int32_t> parents(t.size());
    for (size_t i = 0; i < t.size(); ++i)
      parents[i] = (i == 0 || t.parent[i] == -1) ? 2147483647 : (int32_t)((uint32_t)t.parent[i] & 0x7FFFFFFFu);
    put_array(o, "parents", parents, fi);
    put_array(o, "right_children", t.right, fi);
    put_array(o, "split_conditions", t.value, ff);
    std::vector<uint32_t> sidx(t.size());
    for (size_t i = 0; i < t.size(); ++i) sidx[i] = t.deleted[i] ? 0xFFFFFFFFu : t.feature[i];
    put_array(o, "split_indices", sidx, fu);
    std::vector<uint8_t> stype(t.size(), 0);
    put_array(o, "split_type", stype, fb);
    put_array(o, "sum_hessian", t.sum_hess, ff);
    int32_t num_deleted = 0;
    for (size_t i = 1; i < t.size(); ++i) num_deleted += t.deleted[i] ? 1 : 0;
    o += "\"tree_param\":{\"num_deleted\":\"" + std::to_string(num_deleted) + "\",\"num_feature\":\"" +
         std::to_string(t.num_feature ? t.num_feature : (int32_t)f.num_feature) + "\",\"num_nodes\":\"" +
         std::to_string(t.size()) + "\",\"size_leaf_vector\":\"0\"}}";
  }
  o += "]},\"name\":\"gbtree\"},\"learner_model_param\":{\"base_score\":\"";
  fmt_float(o, f.base_score);
  o += "\",\"num_class\":\"" + std::to_string(f.num_class) + "\",\"num_feature\":\"" + std::to_string(f.num_feature) +
       "\",\"num_target\":\"" + std::to_string(f.num_target) + "\"},\"objective\":{\"name\":\"" + f.objective +
       "\",\"reg_loss_param\":{\"scale_pos_weight\":\"1\"}}},\"version\":[1,6,0]}";
  return o;
}

// ---------------------------------------------------------------- dispatch

static std::string file_extension(const std::string& path) {
  size_t slash = path.find_last_of('/');
  size_t dot = path.find_last_of('.');
  if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return "";
  std::string e = path.substr(dot + 1);
  std::transform(e.begin(), e.end(), e.begin(), [](unsigned char c) { return (char)tolower(c); });
  return e;
}

Forest load_model_buffer(const void* buf, size_t len) {
  if (buf == nullptr || len == 0) throw OhxError("empty model buffer");
  const uint8_t* p = (const uint8_t*)buf;
  if (p[0] == '{') {
    // JSON text continues with white space, '"' or '}'; UBJSON with a length-type marker, '$' or '#'
    const uint8_t c = len > 1 ? p[1] : (uint8_t)'}';
    const bool ubj = c == 'L' || c == 'l' || c == 'I' || c == 'U' || c == 'i' || c == '$' || c == '#';
    return ubj ? parse_ubjson_model(p, len) : parse_json_model((const char*)p, len);
  }
  return parse_legacy_binary(p, len);
}

Forest load_model_file(const std::string& path) {
  std::ifstream in(path, std::ios::binary);
  if (!in) throw OhxError("cannot open model file '" + path + "'");
  std::vector<char> data((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  std::string ext = file_extension(path);
  if (ext == "ubj") return parse_ubjson_model((const uint8_t*)data.data(), data.size());
  if (ext == "json") {
    if (data.size() < 2 || data[0] != '{') throw OhxError("'" + path + "' does not hold a JSON document");
    return parse_json_model(data.data(), data.size());
  }
  return load_model_buffer(data.data(), data.size());
}

void save_model_file(const Forest& f, const std::string& path) {
  std::string ext = file_extension(path);
  std::ofstream out(path, std::ios::binary);
  if (!out) throw OhxError("cannot open '" + path + "' for writing");
  if (ext == "ubj") {
    std::vector<uint8_t> b = write_ubjson_model(f);
    out.write((const char*)b.data(), (std::streamsize)b.size());
  } else if (ext == "json") {
    std::string s = write_json_model(f);
    out.write(s.data(), (std::streamsize)s.size());
  } else {
    std::vector<uint8_t> b = write_legacy_binary(f);
    out.write((const char*)b.data(), (std::streamsize)b.size());
  }
  if (!out) throw OhxError("failed writing '" + path + "'");
}

}  // namespace ohx
