// pcp_neq.h — launch interfaces of the specialised fixpoint kernels (pcp_neq.hip: all-XNeqY models, assignment-driven;
// pcp_big.hip: binary models on 10-bit LDS cells), shared with the C-ABI host code.
#pragma once
#include "pcp_internal.h"

namespace pcp {

// Assignment-driven fixpoint of all-XNeqY models, implicit-active nodes (pcp_neq.hip): no sweep over the record table — round 0
// walks the adjacency lists of the variables that are singletons in the staged domains.
struct NeqArgs {
  ModelDev m;                   // needs adj_off, adjp, const_val, n_vars, n_slots (< 65536), n_recs, max_deg
  const uint32_t* adjp4;        // [adj_off[n_vars]] 4-byte payloads (other | is_y << 15 | t << 16) when slots < 32768 and |offsets| < 32768, else null
  const uint32_t* seed_always;  // [ceil(n_slots/32)] bit v = variable v has a Constant neighbour (walked in round 0 whatever its domain), or null
  uint32_t n_nodes;
  uint32_t nodes_per_block;     // B <= 16 nodes per workgroup, domains in LDS node-major
  uint32_t packed;              // 1 = 16-bit (-lb, ub) cells (declared hull within +-kPackedMax), 0 = int2 cells
  uint32_t lds_wgs;             // workgroups meant to share a CU's LDS (sizes the jump-window area; must match lds_bytes_neq's argument)
  uint32_t* tile_ctr;           // or null: nine words 128 bytes apart, zero between launches — [32 r] the next ticket of residue r = blockIdx.x & 7 (tile = tile_static * gridDim.x + 8 * ticket + r), [256] workgroups that finished (the last one zeroes all nine)
  uint32_t tile_static;         // tiles a workgroup takes by the fixed stride before it draws tickets (>= 1)
  uint32_t stagger;             // option "neq_stagger": shader cycles by which the workgroup in a CU's second pair of wave slots delays its start (0 = none)
  uint32_t debug;               // profiling only ("neq_debug"; results are WRONG when non-zero): 1 = no rounds, 2 = no status scan, 4 = round 0 only
  uint32_t* violation;          // sticky device word: a node was refused with PCP_STATUS_HULL
  unsigned long long* dbg;      // [PCP_DBG_COUNT] diagnostic counters of the context (pcp_debug_counters); slots 5..15: phase timers under debug & 32
  unsigned long long* trace;    // profiling (option "neq_trace_ptr"): [grid][16 wavefronts][16 events] s_memtime stamps, or null
  const uint32_t* sp_ptr;       // host-stepped device-side DFS: the node to run is row *sp_ptr - 1 (see LaunchArgs)
  const uint32_t* stop_ptr;
  struct {                      // n_steps > 0: the search loop itself runs in ONE workgroup (grid 1, nodes_per_block 1): pcp_dfs_device
    uint32_t* sp;               //   lb_in/ub_in = lb_out/ub_out = the stack's rows [capacity][n_vars]; status [capacity]
    uint32_t* stop;
    unsigned long long* counters;  // nodes, solutions, failed nodes, error, internal (pcp_hip.h)
    int32_t* first_solution;
    uint32_t* dirty;            //   [capacity] per stack row: the variable it was branched on (>= n_vars: none), or null
    uint32_t capacity;
    uint32_t n_steps;
    uint32_t stop_on_solution;
    unsigned long long node_limit;
  } dfs;
  uint32_t cell_rows;           // 1 = pcp_device_batch.cell_format PCP_CELLS_PACKED16: lb_in / lb_out are rows of 32-bit cells (-lb & 0xffff | ub << 16), ub_* unused
  const uint32_t* dirty;        // pcp_device_batch.dirty_var: [n_nodes] the one variable in which node i differs from a fixpoint (>= n_vars: none), or null
  const int32_t* lb_in;
  const int32_t* ub_in;
  int32_t* lb_out;
  int32_t* ub_out;
  uint8_t* status;
  pcp_stats* stats;
};
size_t lds_bytes_neq(uint32_t n_slots, uint32_t n_vars, uint32_t nodes_per_block, bool packed, uint32_t wgs = 2);
hipError_t launch_neqfix(const NeqArgs& a, const LaunchPlan& p, hipStream_t stream);
// int32 bounds rows <-> rows of packed cells (n = nodes x variables entries); a bound that does not fit sets *violation
hipError_t launch_pack_rows(const int32_t* lb, const int32_t* ub, uint32_t* cells, size_t n, uint32_t* violation, hipStream_t stream);
hipError_t launch_unpack_rows(const uint32_t* cells, int32_t* lb, int32_t* ub, size_t n, hipStream_t stream);
// the brancher over rows of cells; child_base from launch_branch_scan (pcp_kernels.hip), which also fills counts
hipError_t launch_branch_cells(uint32_t n_nodes, uint32_t n_vars, const uint32_t* cells, const uint32_t* child_base, uint32_t* child_cells, uint32_t* child_dirty,
                               const uint32_t* counts, uint32_t reverse, hipStream_t stream);

// Binary models whose store fits LDS only as 10-bit cells (declared hull of at most 1024 values), implicit-active nodes, one node
// per workgroup (pcp_big.hip).
// The records as that kernel reads them: 8 bytes with the CELL COORDINATES of both operands (slot s lives in word s / 3, field s % 3):
//   .x = word_x | field_x << 15 | (d & 0x1fff) << 17 | kind << 30        .y = word_y | field_y << 15
// and the adjacency payload of a variable's list entry (BigAdj), in the order of ModelDev::adj:
//   .x = word_other | field_other << 15 | (this variable is the record's y) << 17 | kind << 18        .y = d
struct BigArgs {
  ModelDev m;          // needs adj_off, const_val, n_recs, n_vars, n_slots (< 98304)
  const uint2* brec;   // [padded like ModelDev::recs] BigRec, sorted by kind (stable)
  const uint2* badj;   // [adj_off[n_vars]] BigAdj
  uint32_t dense_k;    // a wake-up round is dense iff dense_k * (list entries of the changed variables) >= records (option "big_dense_k")
  uint32_t has_unary;  // the model has records with a Constant operand (unary records, kind 3)
  uint32_t n_nodes;
  int32_t lo10;        // the hull's lower bound
  uint32_t round_mode; // option "big_round": 0 = every wake-up round takes the cheaper form, 1 = always dense, 2 = always sparse (tests)
  uint32_t* violation; // sticky device word: a node was refused with PCP_STATUS_HULL
  unsigned long long* dbg;  // [PCP_DBG_COUNT] diagnostic counters of the context (pcp_debug_counters)
  const uint32_t* sp_ptr;   // host-stepped device-side DFS: the node to run is row *sp_ptr - 1 (see LaunchArgs)
  const uint32_t* stop_ptr;
  const int32_t* lb_in;
  const int32_t* ub_in;
  int32_t* lb_out;
  int32_t* ub_out;
  uint8_t* status;
  pcp_stats* stats;
};
size_t lds_bytes_big(uint32_t n_vars, uint32_t n_slots);
hipError_t launch_bigfix(const BigArgs& a, const LaunchPlan& p, hipStream_t stream);

// Stores with formula propagators (the reified layer): every unit as a tree, one lane per unit (pcp_formula.hip).
using FNode = pcp_fnode;  // leaf: first = record index; inner node: first = index of its first child (children consecutive)
struct FormArgs {
  ModelDev m;                // needs recs, const_val, sums, n_vars, n_slots
  const FNode* nodes;        // the trees of all units
  const uint32_t* unit_root; // [n_units + 1] root node of each unit, then the number of nodes (a unit's nodes are unit_root[u] .. unit_root[u + 1])
  uint32_t n_units;
  uint32_t n_nodes;
  uint32_t* violation;
  const uint32_t* sp_ptr;    // host-stepped device-side DFS: the node to run is row *sp_ptr - 1 (see LaunchArgs)
  const uint32_t* stop_ptr;
  const int32_t* lb_in;
  const int32_t* ub_in;
  int32_t* lb_out;
  int32_t* ub_out;
  const uint64_t* active_in; // [n_nodes][ceil(n_units/64)] or null = every unit active
  uint64_t* active_out;      // or null
  uint8_t* status;
  pcp_stats* stats;
};
size_t lds_bytes_formula(uint32_t n_slots, uint32_t n_units, uint32_t waves);  // one slice per wavefront (= per node in flight)
hipError_t launch_formfix(const FormArgs& a, const LaunchPlan& p, hipStream_t stream);

// Small stores — a few dozen variables, up to a few thousand filters — over explicit `active` rows or implicit nodes: one wavefront per node
// (pcp_small.hip).
struct SmallArgs {
  ModelDev m;                 // needs recs, const_val, sums, n_recs, n_vars, n_slots
  const uint32_t* rec_unit;   // [n_recs] unit of each record, or null: every record is its own unit
  uint32_t n_units;
  uint32_t n_nodes;
  const uint32_t* ad_tab;     // all-different units, or null: [n, then (unit, count <= 64, first index into ad_vars) each]
  const uint32_t* ad_vars;    //   their variables
  const uint32_t* ad_mask;    //   bit u = unit u is one of them
  uint32_t* violation;
  unsigned long long* dbg;    // [kStatSlots][PCP_DBG_COUNT]
  const uint32_t* sp_ptr;     // host-stepped device-side DFS: the node to run is row *sp_ptr - 1 (see LaunchArgs)
  const uint32_t* stop_ptr;
  const uint32_t* nu_off;     // pcp_propagate_device_units: [n_nodes + 1] CSR offsets into nu, or null
  const pcp_prop* nu;         //   the nodes' own unary propagators (one variable against one Constant)
  const int32_t* lb_in;
  const int32_t* ub_in;
  int32_t* lb_out;
  int32_t* ub_out;
  const uint64_t* active_in;  // [n_nodes][ceil(n_units/64)] or null = every unit active
  uint64_t* active_out;       // or null
  uint8_t* status;
  pcp_stats* stats;
};
size_t lds_bytes_small(uint32_t n_slots, uint32_t n_units, uint32_t n_recs, uint32_t waves);
hipError_t launch_smallfix(const SmallArgs& a, const LaunchPlan& p, hipStream_t stream);

}  // namespace pcp
