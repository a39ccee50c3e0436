#include "flatten.hpp"

#include <limits>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>

namespace ohx {

namespace {

struct Pair {
  int32_t l, r;
  int depth;
};

}  // namespace

Placement place_forest(const Forest& f, const LayoutParams& lp) {
  Placement p;
  p.layout = lp;
  const int line = lp.line_slots;
  if (line != 0 && (line < 2 || (line & 1))) throw OhxError("layout: line_slots must be even and >= 2");
  const int top = std::max(1, lp.top_levels);
  p.roots.resize(f.trees.size());
  p.slot_of.resize(f.trees.size());
  uint64_t next = 0;
  auto align_line = [&]() {
    if (line > 0) next = (next + (uint64_t)line - 1) / (uint64_t)line * (uint64_t)line;
  };
  for (size_t ti = 0; ti < f.trees.size(); ++ti) {
    const Tree& t = f.trees[ti];
    std::vector<uint32_t>& slot = p.slot_of[ti];
    slot.assign(t.size(), kNoSlot);
    align_line();
    // ---- breadth-first top ----
    p.roots[ti] = (uint32_t)next;
    slot[0] = (uint32_t)next++;
    p.real_nodes += 1;
    std::vector<int32_t> frontier{0}, nextf;
    int depth = 0;
    std::vector<Pair> pending;
    const bool bfs_all = (line == 0);
    while (!frontier.empty()) {
      nextf.clear();
      const bool last_top_level = (!bfs_all && depth == top - 1);
      for (int32_t n : frontier) {
        if (t.left[(size_t)n] == -1) continue;
        const int32_t l = t.left[(size_t)n], r = t.right[(size_t)n];
        if (last_top_level) {
          pending.push_back({l, r, depth + 1});
        } else {
          slot[(size_t)l] = (uint32_t)next++;
          slot[(size_t)r] = (uint32_t)next++;
          p.real_nodes += 2;
          nextf.push_back(l);
          nextf.push_back(r);
        }
      }
      if (!nextf.empty()) p.max_depth = std::max(p.max_depth, depth + 1);
      if (last_top_level) break;
      frontier.swap(nextf);
      ++depth;
    }
    if (pending.empty()) continue;
    // ---- line-packed deep part: depth-first over subtrees, breadth-first inside a line ----
    align_line();
    std::vector<Pair> stack(pending.rbegin(), pending.rend());
    std::deque<Pair> q;
    while (!stack.empty()) {
      Pair start = stack.back();
      stack.pop_back();
      const uint64_t free_in_line = (uint64_t)line - next % (uint64_t)line;
      if (free_in_line < (uint64_t)lp.min_chunk && free_in_line != (uint64_t)line) align_line();
      q.clear();
      q.push_back(start);
      bool placed_any = false;
      while (!q.empty()) {
        if (placed_any && next % (uint64_t)line == 0) break;  // the line is full
        Pair pr = q.front();
        q.pop_front();
        slot[(size_t)pr.l] = (uint32_t)next++;
        slot[(size_t)pr.r] = (uint32_t)next++;
        p.real_nodes += 2;
        placed_any = true;
        p.max_depth = std::max(p.max_depth, pr.depth);
        for (int32_t c : {pr.l, pr.r})
          if (t.left[(size_t)c] != -1) q.push_back({t.left[(size_t)c], t.right[(size_t)c], pr.depth + 1});
      }
      // whatever did not fit starts new chunks, nearest relatives first
      for (auto it = q.rbegin(); it != q.rend(); ++it) stack.push_back(*it);
    }
  }
  align_line();
  p.num_slots = next;
  if (p.num_slots >= 0xFFFFFFF0ull) throw OhxError("booster too large: more than 2**32 node slots");
  return p;
}

bool packed_format_fits(const Forest& f, const Placement& p) {
  // < 2**26 slots also keeps the array under 4 GiB, which the buffer descriptor needs
  return f.num_feature <= (1u << kPackedFeatureBits) && p.num_slots < kPackedMaxSlots;
}

std::vector<PackedNode> emit_packed(const Forest& f, const Placement& p, std::vector<int32_t>* orig_id) {
  if (!packed_format_fits(f, p)) throw OhxError("booster does not fit the packed node format");
  std::vector<PackedNode> out((size_t)p.num_slots, PackedNode{0u, 0u});
  if (orig_id) orig_id->assign((size_t)p.num_slots, -1);
  for (size_t ti = 0; ti < f.trees.size(); ++ti) {
    const Tree& t = f.trees[ti];
    const auto& slot = p.slot_of[ti];
    for (size_t i = 0; i < t.size(); ++i) {
      if (slot[i] == kNoSlot) continue;
      PackedNode nd;
      memcpy(&nd.value_bits, &t.value[i], 4);
      if (t.left[i] == -1) {
        nd.meta = 0u;
      } else {
        const uint32_t ls = slot[(size_t)t.left[i]], rs = slot[(size_t)t.right[i]];
        if (rs != ls + 1) throw OhxError("internal error: placement broke sibling adjacency");
        nd.meta = (rs << 6) | ((uint32_t)(t.default_left[i] ? 1u : 0u) << 5) | (t.feature[i] & 31u);
      }
      out[slot[i]] = nd;
      if (orig_id) (*orig_id)[slot[i]] = (int32_t)i;
    }
  }
  return out;
}

namespace {

// Which levels start a super-node.  A leaf that sits in a child slot of its parent's
// super-node costs nothing; a leaf that would START a super-node costs one more gather, and
// that gather is the most divergent one of the walk.  With phase 1 the root is hung under a
// virtual always-left node, so super-nodes start at odd levels and leaves at EVEN depth (the
// depth cap of an 18-level model) ride in child slots; the odd gather moves to the top of the
// tree, where all lanes read the same 16 bytes.  Choose per tree by training cover.
int choose_super_phase(const Tree& t) {
  double cost[2] = {0.0, 0.0};
  bool have_cover = false;
  for (float h : t.sum_hess) have_cover = have_cover || h > 0.0f;
  std::vector<std::pair<int32_t, int>> stack{{0, 0}};
  while (!stack.empty()) {
    auto [n, d] = stack.back();
    stack.pop_back();
    if (t.left[(size_t)n] == -1) {
      const double w = have_cover ? (double)t.sum_hess[(size_t)n] : 1.0;
      cost[d & 1] += w;  // phase p (super-nodes start at levels of parity p) pays for leaves of depth parity p
    } else {
      stack.emplace_back(t.left[(size_t)n], d + 1);
      stack.emplace_back(t.right[(size_t)n], d + 1);
    }
  }
  return cost[1] < cost[0] ? 1 : 0;   // fewer paid leaves wins; ties keep phase 0
}

}  // namespace

bool emit_super(const Forest& f, SuperForest* out) {
  out->nodes.clear();
  out->heads.clear();
  if (f.num_feature > kSuperLeaf) return false;
  // a filler looks like an internal node that sends everything - numbers below its +inf thresholds, NaN by its
  // default-left bits - to slot 0 of group 0, i.e. to a filler; none of its codes says "leaf" (flatten.hpp)
  const float inf = std::numeric_limits<float>::infinity();
  const SuperNode unused{inf, inf, inf, super_meta(0, 0, 0, 1, 1, 1, 0)};
  const uint32_t leaf_meta = super_meta(kSuperLeaf, kSuperLeaf, kSuperLeaf, 0, 0, 0, 0);
  struct Item {
    int32_t node;
    uint32_t slot;   // relative to the tree base
    uint32_t level;  // super-nodes above this one on its path
  };
  std::vector<Item> queue;
  for (const Tree& t : f.trees) {
    const uint32_t base = (uint32_t)out->nodes.size();
    SuperTreeHead head{base, 0u, 0.0f, 0u};
    std::vector<SuperNode> sn(8, unused);  // group 0: four fillers; group 1: where the walk starts
    uint32_t next_group = 2;
    queue.clear();
    const bool root_is_leaf = t.left[0] == -1;
    const int phase = root_is_leaf ? 0 : choose_super_phase(t);
    if (phase == 1) {
      // the root is evaluated from the head record; its children start the super-nodes
      head.root_meta = (t.feature[0] & 31u) | ((uint32_t)(t.default_left[0] ? 1u : 0u) << 5) | (1u << 8);
      head.root_thr = t.value[0];
      queue.push_back({t.left[0], 4u, 0u});
      queue.push_back({t.right[0], 5u, 0u});
    } else {
      queue.push_back({0, 4u, 0u});
    }
    for (size_t qi = 0; qi < queue.size(); ++qi) {
      const Item it = queue[qi];
      const size_t n = (size_t)it.node;
      SuperNode s = unused;
      if (t.left[n] == -1) {
        // a leaf on top: its value in all three slots, all three feature codes 31, so that whichever
        // child slot the walk looks at says "leaf" and holds the value; group 0: fillers from here on
        s.thr0 = s.thrL = s.thrR = t.value[n];
        s.meta = leaf_meta;
      } else {
        const size_t l = (size_t)t.left[n], r = (size_t)t.right[n];
        const bool l_int = t.left[l] != -1, r_int = t.left[r] != -1;
        s.thr0 = t.value[n];
        s.thrL = t.value[l];
        s.thrR = t.value[r];
        uint32_t meta = super_meta(t.feature[n], l_int ? t.feature[l] : kSuperLeaf, r_int ? t.feature[r] : kSuperLeaf,
                                   t.default_left[n] ? 1u : 0u, (l_int && t.default_left[l]) ? 1u : 0u,
                                   (r_int && t.default_left[r]) ? 1u : 0u, 0u);
        if (l_int || r_int) {
          const uint32_t grp = next_group++;
          if (grp >= kSuperMaxGroups) {
            out->nodes.clear();
            out->heads.clear();
            return false;
          }
          meta |= grp << 18;
          sn.resize((size_t)next_group * 4, unused);
          if (l_int) {
            queue.push_back({t.left[l], grp * 4 + 0, it.level + 1});
            queue.push_back({t.right[l], grp * 4 + 1, it.level + 1});
          }
          if (r_int) {
            queue.push_back({t.left[r], grp * 4 + 2, it.level + 1});
            queue.push_back({t.right[r], grp * 4 + 3, it.level + 1});
          }
        }
        s.meta = meta;
      }
      // breadth first: the super-nodes of a walk's first three steps (levels 0-2: at most 2 + 8 + 32) and the
      // fillers of group 0 lie in the tree's first kSuperTopSlots records - what walk_super's one coalesced
      // "tree top" load per tree relies on
      if (it.level <= 2 && it.slot >= kSuperTopSlots) throw OhxError("internal error: tree top outside its first records");
      if (it.level <= 3 && it.slot >= kSuperRingSlots) throw OhxError("internal error: a record of the first four steps outside the tree's first 176");
      // (below level 3 the order is free - a record names its child group by number - and was tried group by group depth
      // first, a subtree one stretch of the array, child group a median 368 B from its parent instead of 60 KB: the C360
      // step 24.39 against 24.16 ms, +1.0 %; profiles/r06_not_kept_dfs_order.patch, r06_sweeps.txt)
      sn[it.slot] = s;
      if (it.level + 1 > head.steps) head.steps = it.level + 1;
    }
    out->heads.push_back(head);
    out->nodes.insert(out->nodes.end(), sn.begin(), sn.end());
    if (out->nodes.size() >= 0xFFFFFFF0ull) throw OhxError("booster too large for the super-node format");
  }
  // the tree-top load of the last (small) tree reads kSuperTopSlots records from its base: keep that in bounds
  out->nodes.insert(out->nodes.end(), kSuperTopSlots, unused);
  return true;
}

std::vector<WideNode> emit_wide(const Forest& f, const Placement& p) {
  std::vector<WideNode> out((size_t)p.num_slots, WideNode{0.0f, 0u, 0u, -1});
  for (size_t ti = 0; ti < f.trees.size(); ++ti) {
    const Tree& t = f.trees[ti];
    const auto& slot = p.slot_of[ti];
    for (size_t i = 0; i < t.size(); ++i) {
      if (slot[i] == kNoSlot) continue;
      WideNode nd;
      nd.value = t.value[i];
      nd.orig_id = (int32_t)i;
      if (t.left[i] == -1) {
        nd.left = 0u;
        nd.feat_dl = 0u;
      } else {
        const uint32_t ls = slot[(size_t)t.left[i]], rs = slot[(size_t)t.right[i]];
        if (rs != ls + 1) throw OhxError("internal error: placement broke sibling adjacency");
        nd.left = ls;
        nd.feat_dl = t.feature[i] | ((uint32_t)(t.default_left[i] ? 1u : 0u) << 31);
      }
      out[slot[i]] = nd;
    }
  }
  return out;
}

}  // namespace ohx
