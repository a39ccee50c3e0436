// libohx_synth.so — host-only generator of the synthetic benchmark inputs of
// the 27-feature batches (same definition as the device
// generator, synth_common.h) and the synthetic OH booster (T trees of depth <= D
// grown by recursive partitioning of a feature sample, thresholds taken from the
// sample itself so that x == threshold ties occur).
//
// The reference ships no model file and no data (its production models live on
// NCCS paths, OH_GridComp/OH_instance_OH.rc:17-20), so every test and benchmark
// input comes from here.  Nothing in this file predicts anything.
#include <algorithm>
#include <cmath>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include "flatten.hpp"
#include "forest.hpp"
#include "synth_common.h"

using namespace ohx;

namespace {

thread_local std::string g_err;

struct Work {
  int32_t node;
  uint32_t lo, hi;
  int depth;
};

inline void cell_of(uint64_t m, int im, int jm, int* i, int* j, int* k) {
  const uint64_t plane = (uint64_t)im * (uint64_t)jm;
  *k = (int)(m / plane);
  const uint64_t c = m - (uint64_t)(*k) * plane;
  *j = (int)(c / (uint64_t)im);
  *i = (int)(c - (uint64_t)(*j) * (uint64_t)im);
}

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

const char* ohx_synth_last_error(void) { return g_err.c_str(); }

// torch.distributed.run exports OMP_NUM_THREADS=1 to every rank; the generators are told their
// share of the host cores explicitly instead.
void ohx_synth_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

// rows [nrows][27], PL in hPa (as xx_carr holds it, OH_GridCompMod.F90:314)
int ohx_synth_rows_cpu(uint32_t seed, int im, int jm, int km, uint64_t row_begin, uint64_t nrows, float* out) {
  if (im <= 0 || jm <= 0 || km <= 0 || row_begin + nrows > (uint64_t)im * jm * km) {
    g_err = "ohx_synth_rows_cpu: bad grid or row range";
    return -1;
  }
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < (int64_t)nrows; ++r) {
    int i, j, k;
    cell_of(row_begin + (uint64_t)r, im, jm, &i, &j, &k);
    for (int f = 0; f < OHX_NFEAT; ++f) out[(uint64_t)r * OHX_NFEAT + f] = ohx_synth_feature(seed, f, i, j, k, im, jm, km);
  }
  return 0;
}

// one MAPL field: feature 0..26 ((im,jm) for the 2-D ones, else (im,jm,km); PL in Pa), -1 = TROPP in Pa
int ohx_synth_field_cpu(uint32_t seed, int feature, int im, int jm, int km, float* out) {
  if (im <= 0 || jm <= 0 || km <= 0 || feature < -1 || feature >= OHX_NFEAT) {
    g_err = "ohx_synth_field_cpu: bad argument";
    return -1;
  }
  const bool two_d = feature < 0 || ohx_feature_is_2d(feature);
  const int64_t total = (int64_t)im * jm * (two_d ? 1 : km);
#pragma omp parallel for schedule(static)
  for (int64_t m = 0; m < total; ++m) {
    int i, j, k;
    cell_of((uint64_t)m, im, jm, &i, &j, &k);
    float v;
    if (feature < 0) v = ohx_synth_tropp_pa(seed, i, j);
    else if (feature == OHX_F_PL) v = ohx_synth_pl_pa(seed, i, j, k, im, jm, km);
    else v = ohx_synth_feature(seed, feature, i, j, k, im, jm, km);
    out[m] = v;
  }
  return 0;
}

// Grow the synthetic booster and return it as a model file image (malloc'd):
// format 0 = XGBoost legacy binary, 1 = JSON.  The feature sample is 2**sample_log2
// cells of the (im,jm,km) grid generated with feature_seed.
//   stats[0] = total nodes, stats[1] = total leaves, stats[2] = max depth,
//   stats[3] = sum over sample rows and trees of the walk length (internal nodes visited)
int ohx_synth_model(uint32_t model_seed, int num_trees, int max_depth, int sample_log2, int min_leaf,
                    uint32_t feature_seed, int im, int jm, int km, float base_score, float leaf_sigma, int format,
                    uint8_t** out_buf, uint64_t* out_len, uint64_t stats[4]) {
  try {
    if (num_trees < 0 || max_depth < 0 || max_depth > 40 || sample_log2 < 1 || sample_log2 > 26 || min_leaf < 1 ||
        im <= 0 || jm <= 0 || km <= 0 || out_buf == nullptr || out_len == nullptr)
      throw OhxError("ohx_synth_model: bad argument");
    const uint32_t S = 1u << sample_log2;
    const uint64_t ncell = (uint64_t)im * jm * km;
    // feature-major sample: X[f][s]
    std::vector<float> X((size_t)OHX_NFEAT * S);
#pragma omp parallel for schedule(static)
    for (int64_t s = 0; s < (int64_t)S; ++s) {
      const uint32_t h1 = ohx_hash4(model_seed, 0xA1u, (uint32_t)s, 0u, 0u);
      const uint32_t h2 = ohx_hash4(model_seed, 0xA2u, (uint32_t)s, 0u, 0u);
      const uint64_t m = (((uint64_t)h1 << 32) | h2) % ncell;
      int i, j, k;
      cell_of(m, im, jm, &i, &j, &k);
      for (int f = 0; f < OHX_NFEAT; ++f) X[(size_t)f * S + (size_t)s] = ohx_synth_feature(feature_seed, f, i, j, k, im, jm, km);
    }
    Forest forest;
    forest.base_score = base_score;
    forest.num_feature = OHX_NFEAT;
    forest.objective = "reg:squarederror";
    forest.trees.resize((size_t)num_trees);
    forest.tree_info.assign((size_t)num_trees, 0);
    std::vector<uint64_t> walk_sum((size_t)num_trees, 0), leaves((size_t)num_trees, 0);
    std::vector<int> depth_max((size_t)num_trees, 0);
#pragma omp parallel for schedule(dynamic, 1)
    for (int t = 0; t < num_trees; ++t) {
      std::vector<uint32_t> idx(S);
      for (uint32_t s = 0; s < S; ++s) idx[s] = s;
      std::vector<int32_t> left, right, parent;
      std::vector<uint32_t> feat;
      std::vector<uint8_t> dl;
      std::vector<float> value, hess;
      auto new_node = [&](int32_t par) {
        left.push_back(-1);
        right.push_back(-1);
        parent.push_back(par);
        feat.push_back(0);
        dl.push_back(0);
        value.push_back(0.0f);
        hess.push_back(0.0f);
        return (int32_t)left.size() - 1;
      };
      std::deque<Work> queue;
      queue.push_back({new_node(-1), 0u, S, 0});
      while (!queue.empty()) {
        const Work w = queue.front();
        queue.pop_front();
        const uint32_t cnt = w.hi - w.lo;
        hess[(size_t)w.node] = (float)cnt;
        bool split = false;
        if (w.depth < max_depth && cnt >= 2u * (uint32_t)min_leaf) {
          for (uint32_t attempt = 0; attempt < 8 && !split; ++attempt) {
            const uint32_t h = ohx_hash4(model_seed, 0xB0u + attempt, (uint32_t)t, (uint32_t)w.node, 0u);
            const uint32_t f = h % OHX_NFEAT;
            const uint32_t pick = w.lo + ohx_hash4(model_seed, 0xC0u + attempt, (uint32_t)t, (uint32_t)w.node, 1u) % cnt;
            const float* xf = &X[(size_t)f * S];
            const float thr = xf[idx[pick]];
            uint32_t* b = idx.data() + w.lo;
            uint32_t* e = idx.data() + w.hi;
            uint32_t* mid = std::partition(b, e, [&](uint32_t s) { return xf[s] < thr; });
            const uint32_t nl = (uint32_t)(mid - b);
            if (nl >= (uint32_t)min_leaf && cnt - nl >= (uint32_t)min_leaf) {
              const int32_t l = new_node((int32_t)((uint32_t)w.node | 0x80000000u));
              const int32_t r = new_node(w.node);
              left[(size_t)w.node] = l;
              right[(size_t)w.node] = r;
              feat[(size_t)w.node] = f;
              dl[(size_t)w.node] = (uint8_t)((h >> 16) & 1u);
              value[(size_t)w.node] = thr;
              queue.push_back({l, w.lo, w.lo + nl, w.depth + 1});
              queue.push_back({r, w.lo + nl, w.hi, w.depth + 1});
              split = true;
            }
          }
        }
        if (!split) {
          const uint32_t a = ohx_hash4(model_seed, 0xD0u, (uint32_t)t, (uint32_t)w.node, 0u);
          const uint32_t c = ohx_hash4(model_seed, 0xD1u, (uint32_t)t, (uint32_t)w.node, 0u);
          // sum of four uniforms: variance 1/3
          const float g = (ohx_u01(a) + ohx_u01(a * 0x9E3779B1u + 1u) + ohx_u01(c) + ohx_u01(c * 0x85EBCA77u + 1u)) - 2.0f;
          value[(size_t)w.node] = g * 1.7320508f * leaf_sigma;
          leaves[(size_t)t] += 1;
          walk_sum[(size_t)t] += (uint64_t)cnt * (uint64_t)w.depth;
          depth_max[(size_t)t] = std::max(depth_max[(size_t)t], w.depth);
        }
      }
      Tree& tr = forest.trees[(size_t)t];
      tr.resize(left.size());
      tr.left = left;
      tr.right = right;
      tr.parent = parent;
      tr.feature = feat;
      tr.default_left = dl;
      tr.value = value;
      tr.sum_hess = hess;
      tr.num_feature = OHX_NFEAT;
    }
    forest.validate();
    if (stats) {
      stats[0] = forest.total_nodes();
      stats[1] = 0;
      stats[2] = 0;
      stats[3] = 0;
      for (int t = 0; t < num_trees; ++t) {
        stats[1] += leaves[(size_t)t];
        stats[2] = std::max<uint64_t>(stats[2], (uint64_t)depth_max[(size_t)t]);
        stats[3] += walk_sum[(size_t)t];
      }
    }
    if (format == 1) {
      std::string s = write_json_model(forest);
      *out_buf = (uint8_t*)malloc(s.size());
      if (*out_buf == nullptr) throw OhxError("ohx_synth_model: out of memory");
      memcpy(*out_buf, s.data(), s.size());
      *out_len = s.size();
    } else {
      std::vector<uint8_t> b = write_legacy_binary(forest);
      *out_buf = (uint8_t*)malloc(b.size());
      if (*out_buf == nullptr) throw OhxError("ohx_synth_model: out of memory");
      memcpy(*out_buf, b.data(), b.size());
      *out_len = b.size();
    }
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

void ohx_synth_free(uint8_t* buf) { free(buf); }

// Convert a model file image between the formats (host logic of XGBoosterLoadModel/SaveModel
// without a booster): 0 = legacy binary, 1 = JSON, 2 = UBJSON.
int ohx_model_convert(const uint8_t* in_buf, uint64_t in_len, int format, uint8_t** out_buf, uint64_t* out_len) {
  try {
    Forest f = load_model_buffer(in_buf, (size_t)in_len);
    f.validate();
    if (format == 1) {
      std::string s = write_json_model(f);
      *out_buf = (uint8_t*)malloc(s.size());
      memcpy(*out_buf, s.data(), s.size());
      *out_len = s.size();
    } else {
      std::vector<uint8_t> b = format == 2 ? write_ubjson_model(f) : write_legacy_binary(f);
      *out_buf = (uint8_t*)malloc(b.size());
      memcpy(*out_buf, b.data(), b.size());
      *out_len = b.size();
    }
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// Host check of the super-node layout (flatten.hpp), for the CPU test-suite: walks emit_super's arrays
// the way the kernels do - the root from the head record (phase 1), a fixed `steps` iterations per tree,
// no finished state, the leaf OR-ed in whenever the child's code is 31, fillers after it - so that a
// layout bug shows without a GPU.  Not a prediction path: scalar, test support only.
// out[nrow] margins; info[0] = super-nodes, info[1] = trees that start below the root (phase 1),
// info[2] = total steps.  Returns 1 if the booster does not fit the format.
int ohx_super_walk_cpu(const uint8_t* model, uint64_t model_len, const float* rows, uint64_t nrow, uint32_t ncol,
                       float missing, float* out, uint64_t* info) {
  try {
    Forest f = load_model_buffer(model, (size_t)model_len);
    f.validate();
    SuperForest sf;
    if (!emit_super(f, &sf)) return 1;
    uint64_t phase1 = 0, steps = 0;
    for (const SuperTreeHead& h : sf.heads) {
      phase1 += (h.root_meta >> 8) & 1u;
      steps += h.steps;
    }
    if (info) {
      info[0] = sf.nodes.size();
      info[1] = phase1;
      info[2] = steps;
    }
    const bool missing_is_nan = missing != missing;
    std::vector<float> x(32, 0.0f);   // rows 27..31 "belong to nobody"
    for (uint64_t r = 0; r < nrow; ++r) {
      for (uint32_t c = 0; c < 32; ++c) {
        float v = c < ncol ? rows[r * ncol + c] : (c < f.num_feature ? NAN : 0.0f);
        if (c < ncol && !missing_is_nan && v == missing) v = NAN;
        x[c] = v;
      }
      float acc = f.base_score;
      for (const SuperTreeHead& h : sf.heads) {
        auto left = [](float xv, float thr, bool dl) { return xv != xv ? dl : xv < thr; };
        uint32_t rel = 4u;
        if (h.root_meta & 0x100u) rel += left(x[h.root_meta & 31u], h.root_thr, (h.root_meta & 32u) != 0) ? 0u : 1u;
        uint32_t leaf_bits = 0, taken = 0;
        // as walk_super does: the records of the first three steps among the tree's first kSuperTopSlots
        // (its one "top" load); two more steps than the tree has, as when it shares a group of chains with a
        // deeper tree (a walk past its leaf only meets fillers)
        const uint32_t nsteps = h.steps + 2u;
        for (uint32_t step = 0; step < nsteps; ++step) {
          if (step < 3 && rel >= kSuperTopSlots) throw OhxError("tree top outside the first records of its tree");
          if ((size_t)h.base + rel >= sf.nodes.size()) throw OhxError("walk left the super-node array");
          const SuperNode& s = sf.nodes[h.base + rel];
          const uint32_t w = s.meta;
          const bool l0 = left(x[(w >> 8) & 31u], s.thr0, ((w >> 5) & 1u) != 0);
          const float thr1 = l0 ? s.thrL : s.thrR;
          const uint32_t f1 = (w >> (l0 ? 0u : 13u)) & 31u;
          if (f1 == 31u) {
            memcpy(&leaf_bits, &thr1, 4);
            ++taken;
          }
          const bool l1 = left(x[f1], thr1, ((w >> (l0 ? 6u : 7u)) & 1u) != 0);
          rel = ((w >> 18) << 2) + (l0 ? 0u : 2u) + (l1 ? 0u : 1u);
        }
        if (taken != 1) throw OhxError("a walk must meet exactly one leaf code, met " + std::to_string(taken));
        float leaf;
        memcpy(&leaf, &leaf_bits, 4);
        acc += leaf;
      }
      out[r] = acc;
    }
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

}  // extern "C"
#pragma GCC visibility pop
