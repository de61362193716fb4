// pcp_internal.h — shared between the C-ABI host code (pcp_api.hip) and the gfx950 kernels (pcp_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pcp_hip.h"

namespace pcp {

// One elementary filter as the kernels see it: 16 bytes, one dwordx4 load per lane.
//   xk : operand-x slot | kind << 28          (slot = index into the node's extended domain array)
//   y,z: operand slots (z unused for binary kinds)
//   d  : the single folded offset.  Binary kinds relate  X = dom[x]  and  Y = dom[y] + d.
//        Ternary kinds relate  x  and  y + z + d.   (EQ3: geq uses d-1, leq uses d+1.)
// Constants are interned as pseudo-variables in slots [n_vars, n_slots): their domain is the singleton
// {c}; narrowing it empties it, which is exactly Constant::update returning false (term/constant.rs:49-52).
struct __attribute__((aligned(16))) Rec {
  uint32_t xk;
  uint32_t y;
  uint32_t z;
  int32_t d;
};
static_assert(sizeof(Rec) == 16, "Rec must be 16 bytes");

constexpr uint32_t kSlotMask = 0x0FFFFFFFu;
constexpr uint32_t kMaxSlots = 1u << 26;  // (node, slot) pairs are packed into 32 bits in the kernels

// Compact 8-byte form of a binary record, used for the sweep's stream when every record of the model is binary and
// every slot index fits 15 bits (always true when the domains are LDS-resident with B > 1): the stream, which every
// workgroup reads in full, is half as long.   x = slot_x | slot_y << 15 | kind << 30,   y = d.
struct __attribute__((aligned(8))) Rec8 {
  uint32_t xyk;
  int32_t d;
};
static_assert(sizeof(Rec8) == 8, "Rec8 must be 8 bytes");
constexpr uint32_t kCompactSlots = 1u << 15;

// Word descriptor for the sweep's level -1 test (packed tiles): what the 64 records of one live-mask word have in
// common.  cls = 1 (all XNeqY) or 2 (all XLessY) when the operand slots of the word span at most kRangeMax consecutive
// slots each and every offset fits 16 bits, else 0 (the word always goes to the record-level tests).  A slot range
// [lo, hi] is queried in the tile's range-minimum tables as min(T[k][lo], T[k][second]) with 2^k <= hi-lo+1 < 2^(k+1).
struct __attribute__((aligned(16))) WordPart {
  uint32_t x;  // xlo | second_x << 16
  uint32_t y;  // ylo | second_y << 16
  uint32_t k;  // kx | ky << 4 | cls << 8 | (part b present) << 12
  uint32_t d;  // (dmin & 0xffff) | dmax << 16     (int16 each)
};
// A word that straddles two x-blocks of a table sorted by x (its y operands jump back) is described as two parts, the
// records before and after the first change of x; the word passes level -1 when both parts do.
struct __attribute__((aligned(16))) WordDesc {
  WordPart a, b;
};
static_assert(sizeof(WordDesc) == 32, "WordDesc must be 32 bytes");
// Group descriptor (implicit nodes): what the 64 words = 4096 records of one GROUP have in common, for a test one level above the
// word test: x slots within a short range (as in WordPart), y slots somewhere in [ylo, n_slots) — queried as a SUFFIX minimum —
// and offsets in [dmin, dmax].  cls as in WordPart (0 = no group test).
struct __attribute__((aligned(16))) GroupDesc {
  uint32_t x;    // xlo | second_x << 16
  uint32_t k;    // kx | cls << 8
  uint32_t ylo;  // smallest y slot of the group
  uint32_t d;    // (dmin & 0xffff) | dmax << 16
};
constexpr uint32_t kRangeMax = 64;    // longest slot range a descriptor may cover
constexpr uint32_t kRangeLevels = 7;  // table levels 2^0 .. 2^6

// term::Sum views (term/sum.rs:56-92): pseudo-slots [first, first + count) whose domain is the interval sum of their member
// variables, computed on demand; an update through a Sum of several variables never narrows, it only has to overlap.
struct SumTab {
  const uint32_t* off;  // [count + 1] into mem
  const uint32_t* mem;  // member variable indices
  uint32_t first;       // first sum slot (= n_vars)
  uint32_t count;       // 0 = the model has no Sum view
  const int32_t* mul_off;  // XEqYMulZ records: (dx, dy, dz) Addition offsets of the three operands, indexed by the record's `d`
};

struct ModelDev {
  const Rec* recs;          // [n_recs]
  const Rec8* recs8;        // [n_recs] or null when the model is not compactable
  const WordDesc* wdesc;    // [ceil(n_recs/64)] word descriptors, or null
  const GroupDesc* gdesc;   // [ceil(n_recs/4096)] group descriptors (with wdesc), or null
  const uint32_t* adj_off;  // [n_vars + 1]  CSR var -> incident record ids (constants have no adjacency)
  const uint32_t* adj;      // [adj_off[n_vars]]
  const uint2* adjp;        // [adj_off[n_vars]] payload of each adjacency entry of a binary-only model, or null:
                            //   .x = the record's OTHER operand slot | kind << 28 | (this variable is the record's y) << 31,  .y = d
                            // — the wake-up rounds rebuild the record from it instead of gathering 16 bytes per entry
  const int32_t* const_val; // [n_slots - n_vars]
  uint32_t n_recs;
  uint32_t n_vars;
  uint32_t n_slots;  // n_vars + number of interned constants
  uint32_t has_ternary;
  uint32_t uniform_kind;  // the kind shared by ALL records when that is NEQ or LT, else 0xFFFFFFFF (the sweep then classifies each chunk)
  uint32_t max_deg;       // longest adjacency list of a variable
  SumTab sums;            // Sum views; with count > 0 every record takes the generic path (no compact stream, no payloads)
};

// Both record tables are padded with copies of their last record up to a multiple of 256 records plus kStreamPadRecs, so
// that the sweep's unconditional prefetch (up to two rounds of 16 wavefronts x 4 words ahead) needs no index clamping.
constexpr uint32_t kStreamPadRecs = 2 * 16 * 4 * 64;

// Per-launch arguments of the fixpoint kernel.
struct LaunchArgs {
  ModelDev m;
  uint32_t n_nodes;
  uint32_t nodes_per_block;  // B: nodes whose domains one workgroup keeps in LDS (team == 1)
  uint32_t team;             // G: workgroups cooperating on ONE node (nodes_per_block == 1 when team > 1)
  uint32_t list_cap;         // capacity of the per-round changed-(node,var) list in LDS
  uint32_t global_dom;       // 1 = domains stay in lb_out/ub_out (HBM/L2), for variable stores larger than LDS
  uint32_t word_level;       // packed tiles only: 1 = sweep by word groups with the level -1 range test (needs m.wdesc),
                             //     2 = the same with maximum tables as well (the model has XLessY words)
  uint32_t packed;           // 1 = 16-bit packed LDS domains (every bound within +-kPackedMax); tiles that do not fit mark
                             //     their nodes kStatusRetry, raise *retry_flag to `epoch` and leave the outputs untouched
  uint32_t adj_cache;        // 1 = (n_vars + 1) words of LDS behind the carve hold a copy of m.adj_off
  uint32_t solo;             // 1 = a round with a single changed variable re-runs its records in place and jumps over forbidden values (rounds, c0)
  uint32_t* violation;       // sticky device word (pcp_stats_read reports and clears it): a node was refused with PCP_STATUS_HULL
  uint32_t dom10;            // global_dom launches only: 1 = the node's domains sit in LDS after all, as 10-bit (lb - lo, ub - lo) cells, three
                             //     per u64 (a declared hull of at most 1024 values: 50 000 variables = 130 KB), behind the carve
  int32_t dom10_lo;          // the hull's lower bound
  uint32_t only_marked;      // 1 = second launch of a packed call: run only tiles whose nodes carry kStatusRetry
  uint32_t epoch;            // launch stamp compared with *retry_flag
  uint32_t* retry_flag;      // device word of the context
  const uint32_t* sp_ptr;    // device-side DFS (pcp_dfs_device): the node to run is row *sp_ptr - 1 of the buffers below (nothing to do
                             //     when *sp_ptr == 0 or *stop_ptr != 0); null = rows [0, n_nodes)
  const uint32_t* stop_ptr;
  const int32_t* lb_in;
  const int32_t* ub_in;
  int32_t* lb_out;
  int32_t* ub_out;
  const uint64_t* live_in;  // [n_nodes][words] or null = all live
  uint64_t* live;           // [n_nodes][words] working + output live mask; null = IMPLICIT: no live rows at all, every record is
                            //     taken as live and the status comes from an entailment scan of the final domains
  uint8_t* status;
  pcp_stats* stats;         // device counters
  // team mode scratch (per node): arrival ticket, merged changed mask, remaining counter, fail flag
  uint32_t* team_ticket;    // [n_nodes]
  uint32_t* team_chg;       // [n_nodes][chg_words]
  uint32_t* team_remaining; // [n_nodes]
  uint32_t* team_fail;      // [n_nodes]
  uint64_t* team_counters;  // [n_nodes][kTeamCounters] steps, steps3, narrowings, evaluated, full_evals (merged by the tail block)
};

constexpr uint32_t kTeamCounters = 6;
constexpr int32_t kPackedMax = 16383;   // |bound| limit of the packed tiles: sums of two bounds fit int16
constexpr uint32_t kStatSlots = 64;  // the device counters are striped over this many pcp_stats structs (workgroup b adds to slot b % kStatSlots;
                                      // pcp_stats_read sums them): same-address device atomics serialise at ~12 ns each, chip-wide
constexpr int kBoundMax = (1 << 29) - 1;  // the engine's arithmetic (sums of two bounds and an offset) is exact for |bound| <= kBoundMax
constexpr uint8_t kStatusRetry = 0xFE;  // internal: never visible to the caller (the second launch overwrites it)

struct LaunchPlan {
  uint32_t grid;
  uint32_t block;
  size_t lds_bytes;
};

// Computes the dynamic-LDS footprint for (n_slots, B, list_cap); returns 0 if it cannot fit.
size_t lds_bytes_for(uint32_t n_slots, uint32_t nodes_per_block, uint32_t list_cap, uint32_t block, bool packed = false, uint32_t word_level = 0);
size_t lds_bytes_global(uint32_t n_vars, uint32_t n_slots, uint32_t list_cap);
inline size_t dom10_bytes(uint32_t n_vars) { return ((((size_t)n_vars + 2) / 3) * 8 + 15) & ~(size_t)15; }

hipError_t launch_fixpoint(const LaunchArgs& a, const LaunchPlan& p, hipStream_t stream);

// Grouped models (Conjunction / Distinct units): unit-level `active` rows <-> record-level live rows.
hipError_t launch_expand_units(const uint32_t* rec_unit, uint32_t n_recs, uint32_t unit_words, const uint64_t* active_in, uint64_t* live,
                               uint32_t n_nodes, hipStream_t stream);
hipError_t launch_contract_units(const uint32_t* unit_first, uint32_t n_units, uint32_t n_recs, const uint64_t* live, uint64_t* active_out,
                                 uint32_t n_nodes, hipStream_t stream);

// Implicit-active nodes: record-level live rows from the final domains (live bit = the record is not entailed).
hipError_t launch_derive_active(const ModelDev& m, const int32_t* lb, const int32_t* ub, uint64_t* live, uint32_t n_nodes, hipStream_t stream);

// Set mode (pcp_set.hip): IntervalSet<i32> domains as bitsets, one workgroup per node.  live == nullptr: implicit-active nodes;
// derive_into != nullptr: record-level live rows materialised from the final sets afterwards.
size_t lds_bytes_set(uint32_t n_vars, uint32_t n_slots, uint32_t set_words, uint32_t list_cap);
hipError_t launch_setfix(const ModelDev& m, uint32_t n_nodes, uint32_t set_words, int32_t base, uint32_t list_cap, const uint64_t* bits_in,
                         uint64_t* bits_out, int32_t* lb_out, int32_t* ub_out, const uint64_t* live_in, uint64_t* live, uint8_t* status,
                         pcp_stats* stats, uint64_t* derive_into, hipStream_t stream);

// Set mode, the search loop on the device: one tree per workgroup, the current node in LDS, an undo trail in HBM (pcp_set.hip).
struct SetDfsArgs {
  ModelDev m;
  uint32_t set_words, list_cap;
  int32_t base;
  uint32_t n_trees, level_cap, trail_cap, n_steps, stop_on_solution;
  unsigned long long node_limit;
  uint64_t* bits;                 // [n_trees][n_vars][set_words]: each tree's current node
  uint32_t* tree;                 // [n_trees][4]: levels, trail length, pending variable, finished
  uint4* levels;                  // [n_trees][level_cap]: (variable, value, trail mark, -)
  uint4* trail;                   // [n_trees][trail_cap]: (word index, variable, removed bits lo, hi)
  unsigned long long* counters;   // [n_trees][4]: nodes, solutions, failed, error
  unsigned long long* total_nodes;
  uint32_t* stop;
  int32_t* first_solution;
  uint32_t* solution_flag;
  pcp_stats* stats;
};
size_t lds_bytes_set_dfs(uint32_t n_vars, uint32_t n_slots, uint32_t set_words, uint32_t list_cap);
hipError_t launch_setdfs(const SetDfsArgs& a, hipStream_t stream);
hipError_t launch_setdfs_split(const SetDfsArgs& a, uint32_t n_pairs, const uint32_t* pairs, uint32_t* done, hipStream_t stream);

hipError_t launch_branch_scan(uint32_t n_nodes, const uint8_t* status, uint32_t* child_base, uint32_t* counts, hipStream_t stream);
// Set-mode branching: FirstSmallestVar by CARDINALITY, MiddleVal on the bounds, BinarySplit on the sets.
hipError_t launch_set_branch(uint32_t n_nodes, uint32_t n_vars, uint32_t set_words, int32_t base, uint32_t words, const uint64_t* bits, const int32_t* lb,
                             const int32_t* ub, const uint64_t* active, const uint8_t* status, uint64_t* child_bits, uint64_t* child_active,
                             uint32_t* child_base, uint32_t* counts, uint32_t reverse, hipStream_t stream);

// One step of the device-side DFS after the fixpoint of the top node: count it, branch it in place (right child over the parent's
// row, left child on top) or pop it, keep the first solution.
hipError_t launch_dfs_step(uint32_t n_vars, int32_t* lb, int32_t* ub, const uint8_t* status, uint32_t capacity, uint32_t* sp, uint32_t* stop,
                           unsigned long long* counters, int32_t* first_solution, uint32_t stop_on_solution, unsigned long long node_limit,
                           uint32_t* team_scratch, uint32_t team_words, hipStream_t stream);

// On-device branching (FirstSmallestVar / MiddleVal / BinarySplit): scan of the Unknown flags, then one block per node.
hipError_t launch_branch(uint32_t n_nodes, uint32_t n_vars, uint32_t words, const int32_t* lb, const int32_t* ub, const uint64_t* active,
                         const uint8_t* status, int32_t* child_lb, int32_t* child_ub, uint64_t* child_active, uint32_t* child_dirty, uint32_t* child_base,
                         uint32_t* counts, uint32_t reverse, hipStream_t stream);

}  // namespace pcp
