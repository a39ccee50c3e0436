// Flattening of a Forest into the node arrays the HIP traversal kernels read.
//
// Two device formats, both with absolute child slots into one array for the whole
// booster and "right child = left child + 1" by construction:
//
//  * packed (8 B/node; boosters with <= 32 features and < 2**26 slots):
//      .value_bits = float bits of the split condition (or the leaf value)
//      .meta       = (right_child_slot << 6) | (default_left << 5) | feature
//                    0 => leaf
//    Storing the RIGHT child lets the step be one subtract-with-borrow:
//      next = right - (x < cond).
//  * wide (16 B/node, any feature count): {value, left_child_slot (0 => leaf),
//    feature | default_left << 31, original node id}.
//
// Slot 0 is the root of tree 0 and therefore never a child, which is what makes
// "0 => leaf" unambiguous.
//
// Placement ("layout") is what makes the deep levels affordable on MI355X: the
// top `top_levels` levels of every tree are stored breadth-first (a few KB per
// tree that stay in L1/L2), and below that sibling pairs are packed greedily,
// breadth first, into 128-byte lines (16 packed slots), so that a walk touches
// about one new cache line per three levels instead of one per level.
#pragma once
#include <cstdint>
#include <vector>

#include "forest.hpp"

namespace ohx {

struct LayoutParams {
  int top_levels = 9;    // levels stored breadth-first per tree (root = level 0)
  int line_slots = 16;   // slots per packed line below the top; 0 => breadth-first all the way
  int min_chunk = 6;     // do not start a new subtree in a line with fewer free slots than this
};

struct PackedNode {  // 8 bytes
  uint32_t value_bits;
  uint32_t meta;
};

struct WideNode {  // 16 bytes
  float value;
  uint32_t left;     // absolute slot of the left child, 0 => leaf
  uint32_t feat_dl;  // feature | default_left << 31
  int32_t orig_id;   // node id inside its tree in the model file (for pred_leaf)
};

// Where every node of every tree lives in the flattened array.
struct Placement {
  LayoutParams layout;
  uint64_t num_slots = 0;
  uint64_t real_nodes = 0;                       // nodes reachable from the roots
  int max_depth = 0;
  std::vector<uint32_t> roots;                   // slot of each tree's root
  std::vector<std::vector<uint32_t>> slot_of;    // [tree][node] -> slot, 0xFFFFFFFF if unreachable
};

constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
constexpr int kPackedFeatureBits = 5;
constexpr uint64_t kPackedMaxSlots = 1ull << 26;

// "Super-node" format (16 B, boosters with <= 31 features): one node and BOTH its children,
// i.e. two tree levels per 16-byte gather.  On gfx950 a divergent wave64 gather costs the
// texture addresser the same ~39 cycles whether a lane reads 4, 8 or 16 bytes
// (tools/gather_microbench.hip), so fetching two levels at once halves the dominant cost.
//   thr0 / thrL / thrR  split condition of the node / its left / its right child,
//                       or the leaf value where that slot is a leaf; a super-node whose own
//                       node is a leaf holds the value in all three (and three codes 31)
//   meta  bits  0-4  feature of the left child  (31 = leaf)
//               5,6,7 default_left of node / left child / right child
//               8-12 feature of the node (31 = leaf); at bit 8 so that `meta & 0x1F00` IS the
//                    byte offset of that feature's row in the LDS tile (64 lanes x 4 B)
//              13-17 feature of the right child (31 = leaf)
//              18-31 group index, relative to the tree's base, of the four grandchild
//                    super-nodes [LL, LR, RL, RR] stored contiguously (64 B)
// next = tree_base + 4 * group + 2 * go_right(node) + go_right(child)
//
// Fillers.  Group 0 of every tree is four filler super-nodes and so is every slot of a group that lies
// below a leaf.  A filler looks like an internal node: feature 0 three times, thresholds +inf, default
// left, group 0 - whatever the row holds, the walk goes on to slot 0 of group 0, and no code of a filler
// is 31.  A lane that has taken its leaf therefore keeps stepping through fillers (shared by all
// finished lanes of the tree): the kernel needs no per-lane "finished" state, it keeps "the child's value
// if its code is 31" - and a walk meets code 31 exactly once, at its leaf (two ALU operations per
// step; with all-31 fillers of value +0.0 that were OR-ed in it was three).
// Group 1 is where a walk starts: the root's super-node at slot 4 (phase 0) or the super-nodes of
// the root's two children at slots 4 and 5 (phase 1).
struct SuperNode {
  float thr0, thrL, thrR;
  uint32_t meta;
};
constexpr uint32_t kSuperLeaf = 31u;
constexpr uint32_t super_meta(uint32_t f0, uint32_t fl, uint32_t fr, uint32_t dl0, uint32_t dll, uint32_t dlr,
                              uint32_t group) {
  return (fl & 31u) | (dl0 << 5) | (dll << 6) | (dlr << 7) | ((f0 & 31u) << 8) | ((fr & 31u) << 13) | (group << 18);
}
constexpr uint32_t kSuperMaxGroups = 1u << 14;
// Groups are numbered breadth first, so the records a walk can stand on during its first three steps are among
// the tree's first 48 (groups 0-11): the kernels fetch them with one coalesced load per tree and wave (one record
// per lane) and hand them from lane to lane.  emit_super pads the array by this much behind the last tree.
constexpr uint32_t kSuperTopSlots = 48;
// ... and the records of the first FOUR steps among its first 176 (4 fillers + 4 + 8 + 32 + 128): what the ring kernels
// keep in LDS per tree (kernels.hip)
constexpr uint32_t kSuperRingSlots = 176;

// Per tree: where the walk starts.  Phase-0 trees start at super-node base + 4.  Phase-1 trees
// (super-nodes start at odd levels) evaluate the root from this record - it is the same for
// every lane, so it comes through scalar loads - and start at base + 4 + go_right(root).
struct SuperTreeHead {
  uint32_t base;        // index of the tree's first super-node (multiple of 4)
  uint32_t root_meta;   // bit 8: phase; bits 0-4: root feature; bit 5: root default_left
  float root_thr;
  uint32_t steps;       // super-nodes on the longest path: the walk's trip count (wave-uniform)
};

struct SuperForest {
  std::vector<SuperNode> nodes;
  std::vector<SuperTreeHead> heads;
};

// Returns false (and leaves `out` empty) when the booster does not fit the format.
bool emit_super(const Forest& f, SuperForest* out);

Placement place_forest(const Forest& f, const LayoutParams& lp);
bool packed_format_fits(const Forest& f, const Placement& p);
std::vector<PackedNode> emit_packed(const Forest& f, const Placement& p, std::vector<int32_t>* orig_id);
std::vector<WideNode> emit_wide(const Forest& f, const Placement& p);

}  // namespace ohx
