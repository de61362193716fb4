// pcp_api.hip — the C ABI of libpcp_hip.so (include/pcp_hip.h): context, model lowering, launches.
// Host code only; the kernels are in pcp_kernels.hip.  No CPU fallback exists anywhere in this library.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "pcp_internal.h"
#include "pcp_neq.h"

using namespace pcp;

struct pcp_ctx {
  int device = 0;
  int num_cu = 256;
  size_t lds_max = 160 * 1024;
  std::string err;

  // host model
  uint32_t n_vars = 0;
  uint32_t set_words = 0;   // > 0: IntervalSet<i32> domains as bitsets (pcp_model_reset)
  std::vector<pcp_prop> props;          // as pushed
  std::vector<uint32_t> unit_of_prop;   // unit index of each prop
  std::vector<int32_t> formula_of_prop; // formula number of each prop (a leaf of that tree), or -1
  std::vector<std::vector<pcp_fnode>> formulas;  // pcp_model_push_formula: the trees (leaf.first = index among the formula's own leaves)
  bool has_formulas = false;            // a formula unit or a Boolean / BooleanNeg leaf: the store runs pcp_formula.hip
  pcp_fnode* d_fnodes = nullptr; size_t cap_fnodes = 0;
  uint32_t* d_unit_root = nullptr; size_t cap_unit_root = 0;
  uint32_t n_units = 0;
  std::vector<std::vector<uint32_t>> sums;  // term::Sum views: member variables of each term (pcp_model_push_sum)
  uint32_t* d_sum_off = nullptr; uint32_t* d_sum_mem = nullptr; size_t cap_sum_off = 0, cap_sum_mem = 0;
  int32_t* d_mul_off = nullptr; size_t cap_mul_off = 0;  // XEqYMulZ offsets (dx, dy, dz) per MUL3 record
  uint32_t n_sum_slots = 0;             // Sum terms with more than one member (those have a pseudo-slot)
  bool has_groups = false;
  bool dirty = true;

  // device model
  Rec* d_recs = nullptr;
  Rec8* d_recs8 = nullptr; size_t cap_recs8 = 0; bool compact = false;
  WordDesc* d_wdesc = nullptr; size_t cap_wdesc = 0;
  GroupDesc* d_gdesc = nullptr; size_t cap_gdesc = 0;
  uint32_t word_level = 0;           // 0 = no word descriptors worth using, 1 = XNeqY words only, 2 = XLessY words too
  uint32_t* d_adj_off = nullptr;
  uint32_t* d_adj = nullptr;
  uint2* d_adjp = nullptr; size_t cap_adjp = 0; bool have_adjp = false;
  int32_t* d_const = nullptr;
  // pcp_big.hip: the records sorted by kind and the adjacency payloads, both with the operands' cell coordinates (BigRec / BigAdj, pcp_neq.h)
  uint2* d_brec = nullptr; size_t cap_brec = 0; uint2* d_badj = nullptr; size_t cap_badj = 0;
  bool recs_by_kind_valid = false;  // (built on first use; `big_ok`: the model fits the format — offsets within +-4095, fewer than 98304 variables, no record over two constants)
  bool big_ok = false;
  uint32_t* d_adjp4 = nullptr; size_t cap_adjp4 = 0; bool have_adjp4 = false;  // 4-byte adjacency payloads (pcp_neq.hip)
  uint32_t* d_seed_always = nullptr; size_t cap_seed_always = 0; bool have_seed_always = false;  // variables with a Constant neighbour (pcp_neq.hip)
  bool neq_model = false;            // every record is an XNeqY with at least one variable operand, payload adjacency, slots < 65536
  // all-different units (pcp_small.hip): a Conjunction / Distinct unit whose members are x != y (no offsets, no constants) over EVERY pair of a
  // variable set of at most 64 variables — what Distinct::new builds (propagators/distinct.rs:63-83).  ad_tab = [n, then per unit: unit id,
  // count, first index into ad_vars]; ad_unit_mask bit u = unit u is one.
  uint32_t* d_ad_tab = nullptr; uint32_t* d_ad_vars = nullptr; uint32_t* d_ad_mask = nullptr; size_t cap_ad_tab = 0, cap_ad_vars = 0, cap_ad_mask = 0;
  uint32_t n_alldiff = 0;
  uint32_t* d_rec_unit = nullptr;    // grouped models only: unit of each record
  uint32_t* d_unit_first = nullptr;  // grouped models only: first record of each unit (+ sentinel)
  size_t cap_rec_unit = 0, cap_unit_first = 0;
  uint32_t n_slots = 0;
  bool has_ternary = false;
  uint32_t uniform_kind = 0xFFFFFFFFu;
  uint32_t max_deg = 0;
  bool consts_fit16 = true;      // every interned constant within +-kPackedMax (packed tiles)
  size_t cap_recs = 0, cap_adj = 0, cap_adj_off = 0, cap_const = 0;

  // scratch
  pcp_stats* d_stats = nullptr;
  unsigned long long* d_dbg = nullptr;  // [PCP_DBG_COUNT] diagnostic counters (pcp_debug_counters)
  const uint32_t* cur_nu_off = nullptr;  // pcp_propagate_device_units: the call's node units (device pointers), for the duration of the call
  const pcp_prop* cur_nu = nullptr;
  uint32_t* d_tile_ctr = nullptr; // pcp_neq.hip's tile tickets (NeqArgs::tile_ctr): zero between launches
  bool tickets_suspect = false;   // a HIP error was seen on this context since the tickets were last known to be zero: the next ticketed launch zeroes them first
  hipStream_t tickets_stream = nullptr;  // the stream of the last ticketed launch, and the event recorded behind it: a ticketed launch on ANOTHER stream waits
  hipEvent_t ev_tickets = nullptr;       // for that event, so that two launches of one context can never draw from the words at the same time
  bool tickets_used = false;
  uint32_t* d_retry = nullptr;   // packed launches: stamped with `epoch` by a tile that has to be re-run with 32-bit cells
  uint32_t epoch = 0;
  bool hull_set = false; int32_t hull_lo = 0, hull_hi = 0;  // pcp_model_set_hull
  uint32_t trusted_epoch = 0;   // epoch of the last packed launch without a retry launch (hull declared)
  uint64_t* d_live = nullptr; size_t cap_live = 0;       // working live mask when the caller passes none
  uint32_t* d_child_base = nullptr; size_t cap_child_base = 0;  // branching scratch
  uint32_t* d_team = nullptr; size_t cap_team = 0;       // team-mode scratch (u32 words)
  // host-buffer path staging
  void* d_stage = nullptr; size_t cap_stage = 0;

  pcp_plan last_plan{};   // geometry of the last launch (pcp_last_plan)
  const uint32_t* dfs_sp = nullptr;    // set by pcp_dfs_device around its launches: LaunchArgs::sp_ptr / stop_ptr
  size_t dfs_team_words = 0;           // words of d_team the step kernel zeroes after each step (0: the launch clears them itself)
  const uint32_t* dfs_stop = nullptr;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  bool ev_valid = false;
  int max_dyn_lds_set = 0;

  // options
  int64_t opt_block = 1024;
  int64_t opt_nodes_per_block = 0;  // 0 = auto
  int64_t opt_force_path = 0;       // 0 auto, 1 batch, 2 team
  int64_t opt_team = 0;             // 0 = auto
  int64_t opt_solo = 1;             // 1 = single-variable rounds re-run in place with the forbidden-value jump (pcp_kernels.hip, rounds c0)
  int64_t opt_list_cap = 2048;
  int64_t opt_global_dom = 0;       // 1 = force the HBM-resident-domain variant (tests)
  int64_t opt_branch_reverse = 0;   // 1 = pcp_branch_device writes the children in reverse order (row n_children-1-k)
  int64_t opt_packed = 1;           // 1 = auto (16-bit packed tiles when the batch is large enough), 0 = never
  int64_t opt_word_level = 1;       // 1 = auto (word-group sweep with the level -1 range test on packed tiles), 0 = never
  int64_t opt_dom10 = 1;            // 1 = variable stores larger than LDS use 10-bit LDS cells when the declared hull allows
  int64_t opt_group_level = 1;      // 1 = implicit nodes test whole groups of 64 words first (needs word descriptors)
  int64_t opt_implicit = 1;         // 1 = active_in == NULL runs without live rows (liveness derived), 0 = materialise all-ones rows
  int64_t opt_neq_path = 1;         // 1 = all-XNeqY models with implicit nodes run the assignment-driven kernel (pcp_neq.hip), 0 = the generic sweep kernels
  int64_t opt_neq_block = 0;        // threads per workgroup of that kernel (0 = auto)
  int64_t opt_neq_persist = 1;      // 1 = the tile kernel's workgroups are persistent (at most what the chip holds at once; each runs several tiles)
  int64_t opt_neq_debug = 0;        // profiling only: NeqArgs::debug
  int64_t opt_neq_trace = 0;        // profiling only: device pointer of NeqArgs::trace
  int64_t opt_neq_dynamic = 2;      // pcp_neq.hip: tiles a persistent workgroup takes by the fixed stride before it draws its tiles from a ticket (0 = never draws)
  int64_t opt_neq_wgs = 2;          // workgroups of that kernel meant to share a CU (sizes the jump-window area in LDS)
  int64_t opt_neq_dfs_block = 0;    // threads per tree of the in-kernel search loop: 256 or 512; 0 = 512 for one tree (pcp_dfs_device: latency per node),
                                    // 256 for a forest (four independent chains per CU instead of two: 20 % more nodes/s measured)
  int64_t opt_neq_dfs_wgs = 0;      // trees meant to share a CU's LDS in the in-kernel search loop (sizes the jump-window area): 0 = auto (4 in a forest, 2 for one tree)
  int64_t opt_neq_stagger = 0;      // shader cycles by which the second workgroup of a CU delays its start in large all-XNeqY batches (pcp_neq.hip)
  int64_t opt_neq_hint = 1;         // 0 = pcp_device_batch.dirty_var is ignored (every node is propagated from scratch): A/B and parity tests
  int64_t opt_neq_dfs = 1;          // 1 = pcp_dfs_device on an all-XNeqY model runs the whole search loop in one workgroup, 0 = one launch per step
  int64_t opt_small_alldiff = 1;    // 1 = pcp_small.hip filters an all-different unit through its value mask, 0 = pair by pair
  int64_t opt_time_kernels = 1;     // 1 = a pair of HIP events brackets every fixpoint launch (pcp_last_kernel_ms); 0 = nothing but the kernel is enqueued
  int64_t opt_small_path = 1;       // 1 = small stores (<= 128 slots, <= 2048 records) run one wavefront per node (pcp_small.hip)
  int64_t opt_big_dense_k = 2;      // pcp_big.hip: dense iff k * list entries >= records
  int64_t opt_big_bank = 1;         // pcp_big.hip: the kind-sorted table's records are ordered, within a kind, so that the 32 lanes of a half wavefront read 32 different LDS bank pairs (0 = the model's own order)
  int64_t opt_big_round = 0;        // tests: 1 = dense wake-up rounds only, 2 = sparse only (pcp_big.hip)
  int64_t opt_big_path = 1;         // 1 = binary models on 10-bit cells with implicit nodes run pcp_big.hip, 0 = the generic kernel's dom10 variant
};

namespace {

int32_t fail(pcp_ctx* c, int32_t code, const std::string& msg) {
  if (c) c->err = msg;
  // (a launch that died may have left pcp_neq.hip's tile tickets non-zero — only the last workgroup of a launch that ends cleanly zeroes them)
  if (c && code == PCP_ERR_HIP) c->tickets_suspect = true;
  return code;
}
int32_t hip_fail(pcp_ctx* c, hipError_t e, const char* what) {
  return fail(c, PCP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(c, call)                                     \
  do {                                                       \
    hipError_t e__ = (call);                                 \
    if (e__ != hipSuccess) return hip_fail((c), e__, #call); \
  } while (0)

template <class T>
int32_t ensure(pcp_ctx* c, T*& p, size_t& cap, size_t n) {
  if (n <= cap && p) return PCP_OK;
  if (p) { hipError_t e = hipFree(p); (void)e; p = nullptr; cap = 0; }
  size_t want = std::max<size_t>(n, 16);
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T));
  if (e != hipSuccess) return fail(c, PCP_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
  cap = want;
  return PCP_OK;
}

int arity(uint8_t kind) { return kind >= PCP_BOOL ? 1 : (kind <= PCP_LT ? 2 : 3); }
bool is_sum_operand(uint32_t var) { return var >= PCP_SUM && var < PCP_NOVAR; }

// Reference-panic checks on one prop (SURVEY.md §8b "Error conventions").
int32_t validate_prop(pcp_ctx* c, const pcp_prop& p) {
  if (p.kind > PCP_NBOOL) return fail(c, PCP_ERR_ARG, "unknown propagator kind");
  if (p.kind >= PCP_BOOL && c->set_words) return fail(c, PCP_ERR_UNSUPPORTED, "the reified layer (Boolean / formulas) is interval mode only");
  if (p.group_kind > 2 || p.reserved != 0) return fail(c, PCP_ERR_ARG, "bad group_kind/reserved");
  const int n = arity(p.kind);
  std::vector<uint32_t> seen;  // every variable the propagator subscribes to, Sum members included
  for (int i = 0; i < n; ++i) {
    if (p.var[i] == PCP_NOVAR) return fail(c, PCP_ERR_ARG, "missing operand");
    if (p.off[i] > PCP_BOUND_MAX || p.off[i] < -PCP_BOUND_MAX) return fail(c, PCP_ERR_CONTRACT, "offset outside +-PCP_BOUND_MAX");
    if (p.var[i] == PCP_CONST) continue;
    if (is_sum_operand(p.var[i])) {
      const uint32_t t = p.var[i] & ~PCP_SUM;
      if (t >= c->sums.size()) return fail(c, PCP_ERR_ARG, "unknown Sum term (pcp_model_push_sum)");
      if (c->set_words) return fail(c, PCP_ERR_UNSUPPORTED, "Sum views over IntervalSet domains are not supported (interval mode only)");
      if (p.kind == PCP_MUL3) return fail(c, PCP_ERR_UNSUPPORTED, "XEqYMulZ over a Sum view is not supported");
      for (uint32_t m : c->sums[t]) seen.push_back(m);
      continue;
    }
    if (p.var[i] >= c->n_vars) return fail(c, PCP_ERR_CONTRACT, "variable index out of range (variable/store.rs:176-179)");
    seen.push_back(p.var[i]);
  }
  std::sort(seen.begin(), seen.end());
  if (std::adjacent_find(seen.begin(), seen.end()) != seen.end())
    return fail(c, PCP_ERR_CONTRACT, "propagator already subscribed to this variable (reactors/indexed_deps.rs:69-77)");
  if (p.kind == PCP_MUL3 && c->set_words)
    return fail(c, PCP_ERR_UNSUPPORTED, "XEqYMulZ over IntervalSet domains is not supported (interval mode only)");
  return PCP_OK;
}

// Lower the host props to device records + CSR (done lazily, once per model change).
int32_t finalize_model(pcp_ctx* c) {
  if (!c->dirty) return PCP_OK;
  const size_t P = c->props.size();
  std::map<int32_t, uint32_t> const_slot;
  // slots: [0, n_vars) variables, [n_vars, n_vars + n_sum) Sum views of several members, then the interned constants.
  // `consts` covers every slot >= n_vars (the Sum slots hold 0: their domain is computed from the members on demand).
  std::vector<uint32_t> sum_slot(c->sums.size(), 0), sum_off(1, 0), sum_mem;
  uint32_t n_sum = 0;
  for (size_t t = 0; t < c->sums.size(); ++t) {
    if (c->sums[t].size() == 1) { sum_slot[t] = c->sums[t][0]; continue; }  // a Sum of one variable forwards to it (sum.rs:63-64)
    sum_slot[t] = c->n_vars + n_sum++;
    sum_mem.insert(sum_mem.end(), c->sums[t].begin(), c->sums[t].end());
    sum_off.push_back((uint32_t)sum_mem.size());
  }
  c->n_sum_slots = n_sum;
  std::vector<int32_t> consts(n_sum, 0);
  auto slot_of = [&](uint32_t var, int32_t value) -> uint32_t {
    if (is_sum_operand(var)) return sum_slot[var & ~PCP_SUM];
    if (var != PCP_CONST) return var;
    auto it = const_slot.find(value);
    if (it != const_slot.end()) return it->second;
    uint32_t s = c->n_vars + (uint32_t)consts.size();
    const_slot.emplace(value, s);
    consts.push_back(value);
    return s;
  };
  // every variable an operand makes the propagator depend on (ViewDependencies: identity.rs:66-69, sum.rs:85-91)
  auto for_each_dep = [&](uint32_t var, auto&& f) {
    if (var == PCP_CONST) return;
    if (is_sum_operand(var)) { for (uint32_t m : c->sums[var & ~PCP_SUM]) f(m); return; }
    f(var);
  };
  std::vector<Rec> recs(P);
  std::vector<int32_t> mul_off;
  std::vector<uint32_t> deg(c->n_vars + 1, 0);
  bool tern = false;
  for (size_t r = 0; r < P; ++r) {
    const pcp_prop& p = c->props[r];
    const int n = arity(p.kind);
    uint32_t s[3] = {0, 0, 0};
    int64_t off[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) {
      // a Constant operand carries its value in off[i]; as a pseudo-variable its offset is 0
      s[i] = slot_of(p.var[i], p.off[i]);
      off[i] = (p.var[i] == PCP_CONST) ? 0 : p.off[i];
    }
    int64_t d;
    if (n == 1) d = off[0];                       // Boolean / BooleanNeg over the view x + d
    else if (n == 2) d = off[1] - off[0];         // X = x, Y = y + d
    else if (p.kind == PCP_MUL3) {                // (x + dx) = (y + dy) * (z + dz): the offsets go to a side table, d = its index
      d = (int64_t)(mul_off.size() / 3);
      for (int i = 0; i < 3; ++i) mul_off.push_back((int32_t)off[i]);
    }
    else d = off[1] + off[2] - off[0];            // x  vs  y + z + d
    if (d > PCP_BOUND_MAX || d < -PCP_BOUND_MAX) return fail(c, PCP_ERR_CONTRACT, "folded offset outside +-PCP_BOUND_MAX");
    recs[r].xk = s[0] | ((uint32_t)p.kind << 28);
    recs[r].y = s[1];
    recs[r].z = (n == 3) ? s[2] : 0;
    recs[r].d = (int32_t)d;
    tern |= (n != 2);  // (Boolean leaves too: such stores take the formula kernel, never the binary fast paths)
    for (int i = 0; i < n; ++i) for_each_dep(p.var[i], [&](uint32_t v) { ++deg[v]; });
  }
  tern |= n_sum != 0;  // Sum views: generic path only (no compact stream, no adjacency payloads, no word descriptors)
  const uint32_t n_slots = c->n_vars + (uint32_t)consts.size();
  if (n_slots >= kMaxSlots) return fail(c, PCP_ERR_UNSUPPORTED, "too many variables");
  std::vector<uint32_t> adj_off(c->n_vars + 1, 0);
  for (uint32_t v = 0; v < c->n_vars; ++v) adj_off[v + 1] = adj_off[v] + deg[v];
  c->max_deg = 0;
  for (uint32_t v = 0; v < c->n_vars; ++v) c->max_deg = std::max(c->max_deg, deg[v]);
  std::vector<uint32_t> adj(adj_off[c->n_vars]);
  {
    std::vector<uint32_t> fill(adj_off.begin(), adj_off.end() - 1);
    for (size_t r = 0; r < P; ++r) {
      const pcp_prop& p = c->props[r];
      for (int i = 0; i < arity(p.kind); ++i) for_each_dep(p.var[i], [&](uint32_t v) { adj[fill[v]++] = (uint32_t)r; });
    }
  }
  HIP_TRY(c, hipSetDevice(c->device));
  int32_t rc;
  const size_t Ppad = P ? ((P + 255) / 256) * 256 + kStreamPadRecs : 0;  // see kStreamPadRecs
  if (P) { const Rec last = recs[P - 1]; recs.resize(Ppad, last); }
  if ((rc = ensure(c, c->d_recs, c->cap_recs, Ppad))) return rc;
  if ((rc = ensure(c, c->d_adj_off, c->cap_adj_off, adj_off.size()))) return rc;
  if ((rc = ensure(c, c->d_adj, c->cap_adj, adj.size()))) return rc;
  if ((rc = ensure(c, c->d_const, c->cap_const, consts.size()))) return rc;
  if (P) HIP_TRY(c, hipMemcpy(c->d_recs, recs.data(), Ppad * sizeof(Rec), hipMemcpyHostToDevice));
  c->uniform_kind = 0xFFFFFFFFu;
  if (P && !n_sum) {
    const uint32_t k0 = recs[0].xk >> 28;
    bool same = (k0 == PCP_NEQ || k0 == PCP_LT);
    for (size_t r = 1; r < P && same; ++r) same = (recs[r].xk >> 28) == k0;
    if (same) c->uniform_kind = k0;
  }
  HIP_TRY(c, hipMemcpy(c->d_adj_off, adj_off.data(), adj_off.size() * 4, hipMemcpyHostToDevice));
  if (!adj.empty()) HIP_TRY(c, hipMemcpy(c->d_adj, adj.data(), adj.size() * 4, hipMemcpyHostToDevice));
  c->have_adjp = false;
  if (!tern && !adj.empty()) {  // adjacency payloads (ModelDev::adjp)
    std::vector<uint2> adjp(adj.size());
    std::vector<uint32_t> fill(adj_off.begin(), adj_off.end() - 1);
    for (size_t r = 0; r < P; ++r) {
      const uint32_t x = recs[r].xk & kSlotMask, y = recs[r].y, kind = recs[r].xk >> 28;
      if (x < c->n_vars) adjp[fill[x]++] = make_uint2(y | (kind << 28), (uint32_t)recs[r].d);
      if (y < c->n_vars) adjp[fill[y]++] = make_uint2(x | (kind << 28) | (1u << 31), (uint32_t)recs[r].d);
    }
    if ((rc = ensure(c, c->d_adjp, c->cap_adjp, adjp.size()))) return rc;
    HIP_TRY(c, hipMemcpy(c->d_adjp, adjp.data(), adjp.size() * sizeof(uint2), hipMemcpyHostToDevice));
    c->have_adjp = true;
  }
  if (!consts.empty()) HIP_TRY(c, hipMemcpy(c->d_const, consts.data(), consts.size() * 4, hipMemcpyHostToDevice));
  // assignment-driven path (pcp_neq.hip): all-XNeqY models.  A record over two constants has no variable whose list would
  // run it: such (degenerate) models keep the generic kernels.  Variables with a Constant neighbour are walked in round 0
  // whatever their domain — the constant is a singleton without a list of its own.
  c->neq_model = c->uniform_kind == PCP_NEQ && c->have_adjp && n_slots < 65536u;
  c->have_seed_always = false;
  if (c->neq_model) {
    std::vector<uint32_t> seed((n_slots + 31) / 32, 0u);
    bool any = false;
    for (size_t r = 0; r < P && c->neq_model; ++r) {
      const uint32_t x = recs[r].xk & kSlotMask, y = recs[r].y;
      if (x >= c->n_vars && y >= c->n_vars) c->neq_model = false;
      else if (x >= c->n_vars) { seed[y >> 5] |= 1u << (y & 31); any = true; }
      else if (y >= c->n_vars) { seed[x >> 5] |= 1u << (x & 31); any = true; }
    }
    c->have_adjp4 = false;
    if (c->neq_model && n_slots <= 32768u) {
      bool fits = true;
      for (size_t r = 0; r < P && fits; ++r) fits = recs[r].d >= -32767 && recs[r].d <= 32767;
      if (fits) {
        std::vector<uint32_t> p4(adj.size());
        std::vector<uint32_t> fill(adj_off.begin(), adj_off.end() - 1);
        for (size_t r = 0; r < P; ++r) {
          const uint32_t x = recs[r].xk & kSlotMask, y = recs[r].y;
          const int32_t d = recs[r].d;
          if (x < c->n_vars) p4[fill[x]++] = y | ((uint32_t)(uint16_t)(int16_t)(-d) << 16);
          if (y < c->n_vars) p4[fill[y]++] = x | (1u << 15) | ((uint32_t)(uint16_t)(int16_t)d << 16);
        }
        if ((rc = ensure(c, c->d_adjp4, c->cap_adjp4, p4.size()))) return rc;
        HIP_TRY(c, hipMemcpy(c->d_adjp4, p4.data(), p4.size() * 4, hipMemcpyHostToDevice));
        c->have_adjp4 = true;
      }
    }
    if (c->neq_model && any) {
      if ((rc = ensure(c, c->d_seed_always, c->cap_seed_always, seed.size()))) return rc;
      HIP_TRY(c, hipMemcpy(c->d_seed_always, seed.data(), seed.size() * 4, hipMemcpyHostToDevice));
      c->have_seed_always = true;
    }
  }
  if (!mul_off.empty()) {
    if ((rc = ensure(c, c->d_mul_off, c->cap_mul_off, mul_off.size()))) return rc;
    HIP_TRY(c, hipMemcpy(c->d_mul_off, mul_off.data(), mul_off.size() * 4, hipMemcpyHostToDevice));
  }
  if (n_sum) {
    if ((rc = ensure(c, c->d_sum_off, c->cap_sum_off, sum_off.size()))) return rc;
    if ((rc = ensure(c, c->d_sum_mem, c->cap_sum_mem, sum_mem.size()))) return rc;
    HIP_TRY(c, hipMemcpy(c->d_sum_off, sum_off.data(), sum_off.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_sum_mem, sum_mem.data(), sum_mem.size() * 4, hipMemcpyHostToDevice));
  }
  c->compact = !tern && n_slots <= kCompactSlots && P > 0;
  c->consts_fit16 = true;
  for (int32_t v : consts) c->consts_fit16 &= (v >= -kPackedMax && v <= kPackedMax);
  if (c->compact) {
    std::vector<Rec8> r8(Ppad);
    for (size_t r = 0; r < Ppad; ++r) {
      r8[r].xyk = (recs[r].xk & kSlotMask) | (recs[r].y << 15) | ((recs[r].xk >> 28) << 30);
      r8[r].d = recs[r].d;
    }
    if ((rc = ensure(c, c->d_recs8, c->cap_recs8, Ppad))) return rc;
    HIP_TRY(c, hipMemcpy(c->d_recs8, r8.data(), Ppad * sizeof(Rec8), hipMemcpyHostToDevice));
  }
  // word descriptors for the level -1 test of packed tiles (WordDesc, pcp_internal.h)
  c->word_level = 0;
  if (c->compact) {
    const size_t W = (P + 63) / 64;
    std::vector<WordDesc> wd(W + kStreamPadRecs / 64);
    size_t good = 0;
    bool any_lt = false;
    auto lg = [](uint32_t len) { uint32_t k = 0; while ((2u << k) <= len) ++k; return k; };
    // one part: records [r0, r1) of one binary kind whose x and y slots each span < kRangeMax and whose offsets fit int16
    auto make_part = [&](size_t r0, size_t r1, WordPart& out) {
      const uint32_t kind = recs[r0].xk >> 28;
      if (kind != PCP_NEQ && kind != PCP_LT) return false;
      uint32_t xlo = ~0u, xhi = 0, ylo = ~0u, yhi = 0;
      int32_t dmin = INT32_MAX, dmax = INT32_MIN;
      for (size_t r = r0; r < r1; ++r) {
        if ((recs[r].xk >> 28) != kind) return false;
        const uint32_t x = recs[r].xk & kSlotMask, y = recs[r].y;
        xlo = std::min(xlo, x); xhi = std::max(xhi, x); ylo = std::min(ylo, y); yhi = std::max(yhi, y);
        dmin = std::min(dmin, recs[r].d); dmax = std::max(dmax, recs[r].d);
      }
      if (xhi - xlo >= kRangeMax || yhi - ylo >= kRangeMax || dmin < -30000 || dmax > 30000) return false;
      const uint32_t kx = lg(xhi - xlo + 1), ky = lg(yhi - ylo + 1);
      out.x = xlo | ((xhi - (1u << kx) + 1) << 16);
      out.y = ylo | ((yhi - (1u << ky) + 1) << 16);
      out.k = kx | (ky << 4) | ((kind == PCP_NEQ ? 1u : 2u) << 8);
      out.d = ((uint32_t)dmin & 0xffffu) | ((uint32_t)dmax << 16);
      any_lt |= kind == PCP_LT;
      return true;
    };
    for (size_t w = 0; w < W; ++w) {
      const size_t r0 = w * 64, r1 = std::min(P, r0 + 64);
      WordDesc q;
      memset(&q, 0, sizeof(q));
      if (make_part(r0, r1, q.a)) {
        ++good;
      } else {
        size_t rs = r0 + 1;  // first change of x
        while (rs < r1 && (recs[rs].xk & kSlotMask) == (recs[r0].xk & kSlotMask)) ++rs;
        WordPart pa, pb;
        if (rs < r1 && make_part(r0, rs, pa) && make_part(rs, r1, pb) && (pa.k >> 8) == (pb.k >> 8)) {
          q.a = pa; q.b = pb; q.a.k |= 1u << 12;
          ++good;
        } else {
          memset(&q, 0, sizeof(q));
        }
      }
      wd[w] = q;
    }
    if (W && good * 2 >= W && W <= 512u * 1024u) {  // worth a sweep organised by word groups (16-bit lane counters: <= 1023 groups per wavefront)
      if ((rc = ensure(c, c->d_wdesc, c->cap_wdesc, wd.size()))) return rc;
      HIP_TRY(c, hipMemcpy(c->d_wdesc, wd.data(), wd.size() * sizeof(WordDesc), hipMemcpyHostToDevice));
      c->word_level = any_lt ? 2 : 1;
      // group descriptors: the same idea one level up (64 words at a time; the y operands as a suffix [ylo, n_slots))
      const size_t G = (W + 63) / 64;
      std::vector<GroupDesc> gd(G);
      for (size_t g = 0; g < G; ++g) {
        GroupDesc q;
        memset(&q, 0, sizeof(q));
        const size_t r0 = g * 4096, r1 = std::min(P, r0 + 4096);
        const uint32_t kind = recs[r0].xk >> 28;
        bool ok = kind == PCP_NEQ || kind == PCP_LT;
        uint32_t xlo = ~0u, xhi = 0, ylo = ~0u;
        int32_t dmin = INT32_MAX, dmax = INT32_MIN;
        for (size_t r = r0; r < r1 && ok; ++r) {
          ok = (recs[r].xk >> 28) == kind;
          const uint32_t x = recs[r].xk & kSlotMask;
          xlo = std::min(xlo, x); xhi = std::max(xhi, x); ylo = std::min(ylo, recs[r].y);
          dmin = std::min(dmin, recs[r].d); dmax = std::max(dmax, recs[r].d);
        }
        if (ok && xhi - xlo < kRangeMax && dmin >= -30000 && dmax <= 30000) {
          const uint32_t kx = lg(xhi - xlo + 1);
          q.x = xlo | ((xhi - (1u << kx) + 1) << 16);
          q.k = kx | ((kind == PCP_NEQ ? 1u : 2u) << 8);
          q.ylo = ylo;
          q.d = ((uint32_t)dmin & 0xffffu) | ((uint32_t)dmax << 16);
        }
        gd[g] = q;
      }
      if ((rc = ensure(c, c->d_gdesc, c->cap_gdesc, gd.size()))) return rc;
      HIP_TRY(c, hipMemcpy(c->d_gdesc, gd.data(), gd.size() * sizeof(GroupDesc), hipMemcpyHostToDevice));
    }
  }
  if (c->has_groups) {
    std::vector<uint32_t> first(c->n_units + 1, (uint32_t)P);
    for (size_t r = P; r-- > 0;) first[c->unit_of_prop[r]] = (uint32_t)r;
    if ((rc = ensure(c, c->d_rec_unit, c->cap_rec_unit, P))) return rc;
    if ((rc = ensure(c, c->d_unit_first, c->cap_unit_first, first.size()))) return rc;
    HIP_TRY(c, hipMemcpy(c->d_rec_unit, c->unit_of_prop.data(), P * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_unit_first, first.data(), first.size() * 4, hipMemcpyHostToDevice));
  }
  c->n_alldiff = 0;
  if (c->has_groups && !c->has_formulas) {
    // all-different units: every pair of a set of m <= 64 variables exactly once, as x != y without offsets
    std::vector<uint32_t> tab{0u}, vars, mask((c->n_units + 31) / 32, 0u);
    size_t r = 0;
    while (r < P) {
      const uint32_t u = c->unit_of_prop[r];
      size_t e = r;
      while (e < P && c->unit_of_prop[e] == u) ++e;
      if (e - r >= 3 && c->props[r].group_kind != 0) {
        std::vector<uint32_t> vs;
        std::vector<std::pair<uint32_t, uint32_t>> pairs;
        bool ok = true;
        for (size_t k = r; k < e && ok; ++k) {
          const uint32_t x = recs[k].xk & kSlotMask, y = recs[k].y;
          ok = (recs[k].xk >> 28) == PCP_NEQ && recs[k].d == 0 && x < c->n_vars && y < c->n_vars && x != y;
          if (ok) { vs.push_back(x); vs.push_back(y); pairs.emplace_back(std::min(x, y), std::max(x, y)); }
        }
        if (ok) {
          std::sort(vs.begin(), vs.end()); vs.erase(std::unique(vs.begin(), vs.end()), vs.end());
          std::sort(pairs.begin(), pairs.end());
          const size_t m = vs.size();
          // (pcp_small.hip keeps the tables of up to 8 such units over up to 256 variables in LDS: kSmallAdUnits, kSmallAdVars)
          ok = m <= 64 && tab[0] < 8u && vars.size() + m <= 256 && pairs.size() == m * (m - 1) / 2 && std::adjacent_find(pairs.begin(), pairs.end()) == pairs.end();
          if (ok) {
            tab.push_back(u); tab.push_back((uint32_t)m); tab.push_back((uint32_t)vars.size());
            vars.insert(vars.end(), vs.begin(), vs.end());
            mask[u >> 5] |= 1u << (u & 31u);
            ++tab[0];
          }
        }
      }
      r = e;
    }
    if (tab[0]) {
      if ((rc = ensure(c, c->d_ad_tab, c->cap_ad_tab, tab.size()))) return rc;
      if ((rc = ensure(c, c->d_ad_vars, c->cap_ad_vars, vars.size()))) return rc;
      if ((rc = ensure(c, c->d_ad_mask, c->cap_ad_mask, mask.size()))) return rc;
      HIP_TRY(c, hipMemcpy(c->d_ad_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
      HIP_TRY(c, hipMemcpy(c->d_ad_vars, vars.data(), vars.size() * 4, hipMemcpyHostToDevice));
      HIP_TRY(c, hipMemcpy(c->d_ad_mask, mask.data(), mask.size() * 4, hipMemcpyHostToDevice));
      c->n_alldiff = tab[0];
    }
  }
  if (c->has_formulas) {
    // every unit as a tree for pcp_formula.hip: a standalone propagator = one leaf, a Conjunction / Distinct group = an AND over
    // its members, a formula = its own tree with the leaves renumbered to record indices
    std::vector<pcp_fnode> fn;
    std::vector<uint32_t> root(c->n_units + 1, 0);
    size_t r = 0;
    while (r < P) {
      const uint32_t u = c->unit_of_prop[r];
      size_t e = r;
      while (e < P && c->unit_of_prop[e] == u) ++e;
      root[u] = (uint32_t)fn.size();
      const int32_t f = c->formula_of_prop[r];
      if (f >= 0) {
        const auto& tree = c->formulas[(size_t)f];
        // (the kernel keeps a tree's node statuses in 64-bit masks; only a FLAT Conjunction of leaves may be wider — it is a loop)
        if (tree.size() > 64) {
          bool flat = tree[0].type == PCP_F_AND && (size_t)tree[0].n_children + 1 == tree.size();
          for (size_t i = 1; i < tree.size() && flat; ++i) flat = tree[i].type == PCP_F_LEAF;
          if (!flat) return fail(c, PCP_ERR_UNSUPPORTED, "a formula of more than 64 nodes (other than a flat Conjunction of propagators)");
        }
        const uint32_t base = (uint32_t)fn.size();
        for (const pcp_fnode& nd : tree) {
          pcp_fnode q = nd;
          q.first = nd.type == PCP_F_LEAF ? (uint32_t)r + nd.first : base + nd.first;
          fn.push_back(q);
        }
      } else if (e - r == 1) {
        fn.push_back(pcp_fnode{PCP_F_LEAF, 0, 0, (uint32_t)r});
      } else {
        if (e - r > 65535) return fail(c, PCP_ERR_UNSUPPORTED, "a Conjunction of more than 65535 members next to formula propagators");
        const uint32_t base = (uint32_t)fn.size();
        fn.push_back(pcp_fnode{PCP_F_AND, 0, (uint16_t)(e - r), base + 1});
        for (size_t k = r; k < e; ++k) fn.push_back(pcp_fnode{PCP_F_LEAF, 0, 0, (uint32_t)k});
      }
      r = e;
    }
    root[c->n_units] = (uint32_t)fn.size();  // (sentinel: a unit's nodes are root[u] .. root[u + 1])
    if ((rc = ensure(c, c->d_fnodes, c->cap_fnodes, fn.size()))) return rc;
    if ((rc = ensure(c, c->d_unit_root, c->cap_unit_root, root.size()))) return rc;
    if (!fn.empty()) HIP_TRY(c, hipMemcpy(c->d_fnodes, fn.data(), fn.size() * sizeof(pcp_fnode), hipMemcpyHostToDevice));
    if (!root.empty()) HIP_TRY(c, hipMemcpy(c->d_unit_root, root.data(), root.size() * 4, hipMemcpyHostToDevice));
  }
  c->n_slots = n_slots;
  c->has_ternary = tern;
  c->recs_by_kind_valid = false;
  c->dirty = false;
  return PCP_OK;
}

// Set mode (IntervalSet<i32> domains): one workgroup per node, the sets in LDS (pcp_set.hip).
int32_t propagate_set_device(pcp_ctx* c, uint32_t n_nodes, const pcp_device_batch* bt, hipStream_t stream) {
  if (!c->hull_set) return fail(c, PCP_ERR_CONTRACT, "set mode needs the hull of the initial domains (pcp_model_set_hull): value v is bit v - lo");
  if ((int64_t)c->hull_hi - c->hull_lo >= (int64_t)c->set_words * 64) return fail(c, PCP_ERR_CONTRACT, "the declared hull does not fit set_words * 64 values");
  if (c->n_vars && (!bt->bits_in || !bt->bits_out || !bt->lb_out || !bt->ub_out)) return fail(c, PCP_ERR_ARG, "set mode: bits_in, bits_out, lb_out and ub_out must not be null");
  const uint32_t P = (uint32_t)c->props.size(), words = (P + 63) / 64, S = c->n_slots;
  uint32_t cap = (uint32_t)std::min<int64_t>(c->opt_list_cap, 1024);
  while (cap > 64 && !lds_bytes_set(c->n_vars, S, c->set_words, cap)) cap /= 2;
  const size_t lds = lds_bytes_set(c->n_vars, S, c->set_words, cap);
  if (!lds || lds > c->lds_max) return fail(c, PCP_ERR_UNSUPPORTED, "set-mode variable store does not fit one CU's LDS (n_vars * (set_words + 1) * 8 bytes)");
  ModelDev m;
  memset(&m, 0, sizeof(m));
  m.recs = c->d_recs; m.adj_off = c->d_adj_off; m.adj = c->d_adj; m.adjp = c->have_adjp ? c->d_adjp : nullptr; m.const_val = c->d_const;
  m.n_recs = P; m.n_vars = c->n_vars; m.n_slots = S; m.has_ternary = c->has_ternary; m.uniform_kind = c->uniform_kind; m.max_deg = c->max_deg;
  const bool implicit = bt->active_in == nullptr && c->opt_implicit;
  const uint32_t unit_words = (c->n_units + 63) / 64;
  const uint64_t* live_in = nullptr;
  uint64_t* live = nullptr;
  uint64_t* derive = nullptr;
  int32_t rc;
  if (implicit) {
    if (bt->active_out && P) {
      if (c->has_groups) { if ((rc = ensure(c, c->d_live, c->cap_live, (size_t)n_nodes * std::max<uint32_t>(words, 1)))) return rc; derive = c->d_live; }
      else derive = bt->active_out;
    }
  } else if (c->has_groups) {
    if ((rc = ensure(c, c->d_live, c->cap_live, (size_t)n_nodes * std::max<uint32_t>(words, 1)))) return rc;
    HIP_TRY(c, launch_expand_units(c->d_rec_unit, P, unit_words, bt->active_in, c->d_live, n_nodes, stream));
    live_in = c->d_live; live = c->d_live;
  } else if (bt->active_out) {
    live_in = bt->active_in; live = bt->active_out;
  } else {
    if ((rc = ensure(c, c->d_live, c->cap_live, (size_t)n_nodes * std::max<uint32_t>(words, 1)))) return rc;
    live_in = bt->active_in; live = c->d_live;
  }
  c->last_plan = pcp_plan{1u, 1u, 0u, 0u, 0u, 0u, implicit ? 1u : 0u, 1u, n_nodes, 1024u, (uint32_t)lds, cap, 0u};
  if (c->opt_time_kernels) HIP_TRY(c, hipEventRecord(c->ev_start, stream));
  HIP_TRY(c, launch_setfix(m, n_nodes, c->set_words, c->hull_lo, cap, bt->bits_in, bt->bits_out, bt->lb_out, bt->ub_out, live_in, live, bt->status,
                           c->d_stats, derive, stream));
  if (c->opt_time_kernels) HIP_TRY(c, hipEventRecord(c->ev_stop, stream));
  if (c->has_groups && bt->active_out && P)
    HIP_TRY(c, launch_contract_units(c->d_unit_first, c->n_units, P, c->d_live, bt->active_out, n_nodes, stream));
  c->ev_valid = c->opt_time_kernels != 0;
  return PCP_OK;
}

// All-XNeqY models, implicit-active nodes: the assignment-driven kernel (pcp_neq.hip).  Returns 1 when the variable store does
// not fit LDS even with one node per workgroup (the caller falls through to the generic kernels).
int32_t propagate_neq_device(pcp_ctx* c, uint32_t n_nodes, const pcp_device_batch* bt, hipStream_t stream) {
  const uint32_t S = c->n_slots, V = c->n_vars, P = (uint32_t)c->props.size();
  const bool hull_fits16 = c->hull_set && c->hull_lo >= -kPackedMax && c->hull_hi <= kPackedMax;
  const bool packed = hull_fits16 && c->consts_fit16 && c->opt_packed;
  // nodes per workgroup: a tile per CU, as large as that allows — a round's list walk decodes an entry once for all the nodes of the
  // tile in which the variable changed, and deep tiles (many assigned variables, shared by neighbouring nodes) live on that
  // (measured, 4096 nodes 3000 nodes down a dive: 16-node tiles 0.64 ms, 8-node tiles 0.83 ms)
  uint32_t want = c->opt_nodes_per_block ? (uint32_t)c->opt_nodes_per_block : std::max<uint32_t>(1, n_nodes / (uint32_t)c->num_cu);
  want = std::min<uint32_t>(want, 16);
  uint32_t B = 0;
  for (uint32_t t : {16u, 8u, 4u, 2u, 1u}) {
    if (t > want) continue;
    const size_t need = lds_bytes_neq(S, V, t, packed, (uint32_t)c->opt_neq_wgs);
    if (need && need <= c->lds_max) { B = t; break; }
  }
  if (!B) return 1;
  if (bt->cell_format && !packed) return fail(c, PCP_ERR_UNSUPPORTED, "cell_format PCP_CELLS_PACKED16 needs a declared hull (and constants) within +-16383");
  LaunchPlan plan;
  plan.grid = (n_nodes + B - 1) / B;
  plan.lds_bytes = lds_bytes_neq(S, V, B, packed, (uint32_t)c->opt_neq_wgs);
  // few tiles: all the lanes a CU has on each; many tiles: 512 threads, so that two or three workgroups share a CU and one's
  // staging overlaps the other's list walk
  plan.block = c->opt_neq_block ? (uint32_t)c->opt_neq_block : (plan.grid <= (uint32_t)c->num_cu ? 1024u : 512u);
  const uint32_t lds_wgs = (uint32_t)c->opt_neq_wgs;
  if (c->opt_neq_persist) {
    // persistent tiles: no more workgroups than the chip holds at once (LDS and threads per CU); each runs the tiles g, g + grid, ...
    const uint32_t per_cu = std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)(c->lds_max / plan.lds_bytes), 2048u / plan.block));
    plan.grid = std::min<uint32_t>(plan.grid, per_cu * (uint32_t)c->num_cu);
  }
  NeqArgs a;
  memset(&a, 0, sizeof(a));
  a.m.recs = c->d_recs; a.m.adj_off = c->d_adj_off; a.m.adj = c->d_adj; a.m.adjp = c->d_adjp; a.m.const_val = c->d_const;
  a.m.n_recs = P; a.m.n_vars = V; a.m.n_slots = S; a.m.uniform_kind = c->uniform_kind; a.m.max_deg = c->max_deg;
  a.seed_always = c->have_seed_always ? c->d_seed_always : nullptr;
  a.adjp4 = c->have_adjp4 ? c->d_adjp4 : nullptr;
  a.n_nodes = n_nodes; a.nodes_per_block = B; a.packed = packed ? 1u : 0u;
  a.violation = c->d_retry + 1; a.dbg = c->d_dbg;
  a.debug = (uint32_t)c->opt_neq_debug; a.trace = reinterpret_cast<unsigned long long*>(c->opt_neq_trace);
  a.lds_wgs = lds_wgs;
  // persistent workgroups with more tiles than workgroups draw their second and later tiles from a ticket (the first is blockIdx.x): tiles of
  // unequal cost — deep nodes next to shallow ones — then spread over the workgroups as they come free instead of by a fixed stride
  a.tile_static = (uint32_t)std::max<int64_t>(1, c->opt_neq_dynamic);
  // (the tiles behind the static ones are dealt to EIGHT residues of blockIdx.x: a grid of fewer than eight workgroups — a device or an option
  // that leaves fewer than eight resident — would never draw some of them: fixed stride there)
  a.tile_ctr = (c->opt_neq_persist && c->opt_neq_dynamic && !c->dfs_sp && plan.grid >= 8u && (uint64_t)a.tile_static * plan.grid < (n_nodes + B - 1) / B) ? c->d_tile_ctr : nullptr;
  if (a.tile_ctr) {
    // The tickets belong to the context and must be zero when a launch starts.  A launch that ended cleanly left them zero; after a HIP error on
    // this context they are zeroed here; and a launch on a different stream than the last ticketed one first waits for that one to end.
    if (c->tickets_suspect) { HIP_TRY(c, hipMemsetAsync(c->d_tile_ctr, 0, 9 * 128, stream)); c->tickets_suspect = false; }
    if (c->tickets_used && c->tickets_stream != stream) HIP_TRY(c, hipStreamWaitEvent(stream, c->ev_tickets, 0));
  }
  a.stagger = (c->opt_neq_persist && plan.grid > (uint32_t)c->num_cu && !c->dfs_sp) ? (uint32_t)c->opt_neq_stagger : 0u;
  a.sp_ptr = c->dfs_sp; a.stop_ptr = c->dfs_stop;
  a.lb_in = bt->lb_in; a.ub_in = bt->ub_in; a.lb_out = bt->lb_out; a.ub_out = bt->ub_out;
  a.status = bt->status;
  a.stats = c->d_stats;
  a.cell_rows = bt->cell_format;
  if (a.cell_rows) { a.ub_in = nullptr; a.ub_out = nullptr; }  // (rows of cells: one row per node)
  a.dirty = (c->opt_neq_hint && !c->dfs_sp) ? bt->dirty_var : nullptr;  // (pcp_device_batch.dirty_var: round 0 = that variable's lists only)
  c->dfs_team_words = 0;
  c->last_plan = pcp_plan{B, 1u, packed ? 1u : 0u, 0u, 0u, 0u, 1u, 0u, plan.grid, plan.block, (uint32_t)plan.lds_bytes, S, 1u};
  if (!c->dfs_sp && c->opt_time_kernels) HIP_TRY(c, hipEventRecord(c->ev_start, stream));
  HIP_TRY(c, launch_neqfix(a, plan, stream));
  if (!c->dfs_sp && c->opt_time_kernels) HIP_TRY(c, hipEventRecord(c->ev_stop, stream));
  if (a.tile_ctr) { HIP_TRY(c, hipEventRecord(c->ev_tickets, stream)); c->tickets_stream = stream; c->tickets_used = true; }
  if (bt->active_out && P) {
    // the `active` rows on request: record r is live iff it is not entailed under the final domains
    ModelDev m = a.m;
    m.sums = SumTab{c->d_sum_off, c->d_sum_mem, c->n_vars, c->n_sum_slots, c->d_mul_off};
    int32_t rc;
    const uint32_t words = (P + 63) / 64;
    if (c->has_groups) {
      if ((rc = ensure(c, c->d_live, c->cap_live, (size_t)n_nodes * std::max<uint32_t>(words, 1)))) return rc;
      HIP_TRY(c, launch_derive_active(m, bt->lb_out, bt->ub_out, c->d_live, n_nodes, stream));
      HIP_TRY(c, launch_contract_units(c->d_unit_first, c->n_units, P, c->d_live, bt->active_out, n_nodes, stream));
    } else {
      HIP_TRY(c, launch_derive_active(m, bt->lb_out, bt->ub_out, bt->active_out, n_nodes, stream));
    }
  }
  c->ev_valid = !c->dfs_sp && c->opt_time_kernels != 0;
  return PCP_OK;
}

}  // namespace

extern "C" {

uint32_t pcp_abi_version(void) { return PCP_ABI_VERSION; }

const char* pcp_strerror(int32_t err) {
  switch (err) {
    case PCP_OK: return "ok";
    case PCP_ERR_ARG: return "invalid argument";
    case PCP_ERR_CONTRACT: return "contract violation (the reference would panic)";
    case PCP_ERR_HIP: return "HIP runtime error";
    case PCP_ERR_NOMEM: return "out of device memory";
    case PCP_ERR_UNSUPPORTED: return "unsupported";
    case PCP_ERR_NODEVICE: return "no HIP device";
    default: return "unknown error";
  }
}

const char* pcp_last_error(const pcp_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int32_t pcp_ctx_create(int32_t hip_device, pcp_ctx** out) {
  if (!out) return PCP_ERR_ARG;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return PCP_ERR_NODEVICE;  // fail loudly: there is no CPU path
  if (hip_device < 0 || hip_device >= n) return PCP_ERR_ARG;
  if (hipSetDevice(hip_device) != hipSuccess) return PCP_ERR_HIP;
  pcp_ctx* c = new pcp_ctx();
  c->device = hip_device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, hip_device) == hipSuccess) {
    c->num_cu = prop.multiProcessorCount;
    c->lds_max = (size_t)prop.maxSharedMemoryPerMultiProcessor >= 160 * 1024 ? 160 * 1024 : 64 * 1024;
  }
  if (hipMalloc(reinterpret_cast<void**>(&c->d_stats), kStatSlots * sizeof(pcp_stats)) != hipSuccess ||
      hipMemset(c->d_stats, 0, kStatSlots * sizeof(pcp_stats)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&c->d_retry), 8) != hipSuccess || hipMemset(c->d_retry, 0, 8) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&c->d_tile_ctr), 9 * 128) != hipSuccess || hipMemset(c->d_tile_ctr, 0, 9 * 128) != hipSuccess ||  // (pcp_neq.hip: eight tile tickets and the count of finished workgroups, a cache line each)
      hipMalloc(reinterpret_cast<void**>(&c->d_dbg), kStatSlots * PCP_DBG_COUNT * 8) != hipSuccess || hipMemset(c->d_dbg, 0, kStatSlots * PCP_DBG_COUNT * 8) != hipSuccess ||
      hipEventCreate(&c->ev_start) != hipSuccess || hipEventCreate(&c->ev_stop) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_tickets, hipEventDisableTiming) != hipSuccess) {
    delete c;
    return PCP_ERR_HIP;
  }
  *out = c;
  return PCP_OK;
}

void pcp_ctx_destroy(pcp_ctx* c) {
  if (!c) return;
  hipError_t e = hipSetDevice(c->device);
  (void)e;
  void* ptrs[] = {c->d_fnodes, c->d_unit_root, c->d_ad_tab, c->d_ad_vars, c->d_ad_mask, c->d_brec, c->d_badj, c->d_adjp4, c->d_seed_always, c->d_mul_off, c->d_gdesc, c->d_sum_off, c->d_sum_mem, c->d_recs, c->d_adj_off, c->d_adj, c->d_const, c->d_stats, c->d_live, c->d_team, c->d_stage, c->d_rec_unit, c->d_unit_first, c->d_recs8, c->d_child_base, c->d_retry, c->d_tile_ctr, c->d_dbg, c->d_wdesc, c->d_adjp};
  for (void* p : ptrs)
    if (p) { e = hipFree(p); (void)e; }
  if (c->ev_start) { e = hipEventDestroy(c->ev_start); (void)e; }
  if (c->ev_stop) { e = hipEventDestroy(c->ev_stop); (void)e; }
  if (c->ev_tickets) { e = hipEventDestroy(c->ev_tickets); (void)e; }
  delete c;
}

int32_t pcp_model_reset(pcp_ctx* c, uint32_t n_vars, uint32_t set_words) {
  if (!c) return PCP_ERR_ARG;
  if (set_words > 4096) return fail(c, PCP_ERR_ARG, "set_words too large");
  if (n_vars >= kMaxSlots) return fail(c, PCP_ERR_UNSUPPORTED, "too many variables");
  c->n_vars = n_vars;
  c->set_words = set_words;
  c->props.clear();
  c->unit_of_prop.clear();
  c->formula_of_prop.clear();
  c->formulas.clear();
  c->has_formulas = false;
  c->n_units = 0;
  c->sums.clear();
  c->has_groups = false;
  c->hull_set = false;
  c->dirty = true;
  return PCP_OK;
}

int32_t pcp_model_push_props(pcp_ctx* c, uint32_t n, const pcp_prop* props) {
  if (!c || (n && !props)) return PCP_ERR_ARG;
  for (uint32_t i = 0; i < n; ++i) {
    int32_t rc = validate_prop(c, props[i]);
    if (rc) return rc;
  }
  for (uint32_t i = 0; i < n; ++i) {
    const pcp_prop& p = props[i];
    // a unit is a run of members WITHIN ONE CALL: props pushed by an earlier call (or kept by pcp_model_truncate) never
    // join a later call's group, whatever their `group` value
    bool same_unit = false;
    if (p.group_kind != 0 && i > 0) {
      const pcp_prop& q = props[i - 1];
      same_unit = q.group_kind == p.group_kind && q.group == p.group;
    }
    if (!same_unit) ++c->n_units;
    if (p.group_kind != 0) c->has_groups = true;
    c->props.push_back(p);
    c->unit_of_prop.push_back(c->n_units - 1);
    c->formula_of_prop.push_back(-1);
    if (p.kind >= PCP_BOOL) c->has_formulas = true;
  }
  c->dirty = true;
  return PCP_OK;
}

int32_t pcp_model_push_formula(pcp_ctx* c, uint32_t n_nodes, const pcp_fnode* nodes, uint32_t n_leaves, const pcp_prop* leaves) {
  if (!c || !nodes || !leaves || n_nodes == 0 || n_leaves == 0) return PCP_ERR_ARG;
  if (c->set_words) return fail(c, PCP_ERR_UNSUPPORTED, "formula propagators are interval mode only");
  // the tree: children behind their parent and consecutive, every node reached exactly once, every leaf used exactly once, depth <= 8
  std::vector<uint32_t> depth(n_nodes, 0), uses(n_nodes, 0), leaf_uses(n_leaves, 0);
  depth[0] = 1; uses[0] = 1;
  for (uint32_t i = 0; i < n_nodes; ++i) {
    const pcp_fnode& nd = nodes[i];
    if (nd.reserved != 0 || nd.type > PCP_F_OR) return fail(c, PCP_ERR_ARG, "bad formula node");
    if (uses[i] != 1) return fail(c, PCP_ERR_ARG, "formula node not reached exactly once from the root");
    if (depth[i] > 8) return fail(c, PCP_ERR_UNSUPPORTED, "formula deeper than 8 levels");
    if (nd.type == PCP_F_LEAF) {
      if (nd.first >= n_leaves) return fail(c, PCP_ERR_ARG, "formula leaf out of range");
      if (++leaf_uses[nd.first] != 1) return fail(c, PCP_ERR_ARG, "formula leaf used twice");
      continue;
    }
    if (nd.n_children == 0) return fail(c, PCP_ERR_CONTRACT, "a Conjunction / Disjunction needs at least one child");
    if (nd.first <= i || (uint64_t)nd.first + nd.n_children > n_nodes) return fail(c, PCP_ERR_ARG, "formula children out of range");
    for (uint32_t k = 0; k < nd.n_children; ++k) { ++uses[nd.first + k]; depth[nd.first + k] = depth[i] + 1; }
  }
  for (uint32_t i = 0; i < n_leaves; ++i) {
    if (leaf_uses[i] != 1) return fail(c, PCP_ERR_ARG, "formula leaf not used");
    int32_t rc = validate_prop(c, leaves[i]);
    if (rc) return rc;
  }
  c->formulas.emplace_back(nodes, nodes + n_nodes);
  ++c->n_units;
  for (uint32_t i = 0; i < n_leaves; ++i) {
    pcp_prop p = leaves[i];
    p.group_kind = 0; p.group = 0;
    c->props.push_back(p);
    c->unit_of_prop.push_back(c->n_units - 1);
    c->formula_of_prop.push_back((int32_t)c->formulas.size() - 1);
  }
  c->has_formulas = true;
  c->dirty = true;
  return PCP_OK;
}

int32_t pcp_model_push_sum(pcp_ctx* c, uint32_t n_members, const uint32_t* vars, uint32_t* term) {
  if (!c || !vars || !term) return PCP_ERR_ARG;
  if (n_members == 0) return fail(c, PCP_ERR_CONTRACT, "At least one variable in sum. (term/sum.rs:78)");
  if (c->sums.size() >= 0x3FFFFFF0u) return fail(c, PCP_ERR_UNSUPPORTED, "too many Sum terms");
  for (uint32_t i = 0; i < n_members; ++i)
    if (vars[i] >= c->n_vars) return fail(c, PCP_ERR_CONTRACT, "variable index out of range (variable/store.rs:176-179)");
  c->sums.emplace_back(vars, vars + n_members);
  *term = (uint32_t)c->sums.size() - 1;
  c->dirty = true;
  return PCP_OK;
}

int32_t pcp_model_truncate(pcp_ctx* c, uint32_t n_units) {
  if (!c) return PCP_ERR_ARG;
  if (n_units > c->n_units) return fail(c, PCP_ERR_ARG, "truncate beyond the current size");
  size_t keep = 0;
  while (keep < c->props.size() && c->unit_of_prop[keep] < n_units) ++keep;
  c->props.resize(keep);
  c->unit_of_prop.resize(keep);
  c->formula_of_prop.resize(keep);
  c->n_units = n_units;
  c->has_groups = false;
  for (auto& p : c->props) c->has_groups |= p.group_kind != 0;
  int32_t last_formula = -1;
  c->has_formulas = false;
  for (size_t i = 0; i < keep; ++i) { last_formula = std::max(last_formula, c->formula_of_prop[i]); c->has_formulas |= c->props[i].kind >= PCP_BOOL; }
  c->formulas.resize((size_t)(last_formula + 1));
  c->has_formulas |= last_formula >= 0;
  c->dirty = true;
  return PCP_OK;
}

int32_t pcp_model_set_hull(pcp_ctx* c, int32_t lo, int32_t hi) {
  if (!c) return PCP_ERR_ARG;
  if (lo > hi) return fail(c, PCP_ERR_ARG, "empty hull");
  if (lo < -PCP_BOUND_MAX || hi > PCP_BOUND_MAX) return fail(c, PCP_ERR_CONTRACT, "bound outside +-PCP_BOUND_MAX");
  c->hull_set = true; c->hull_lo = lo; c->hull_hi = hi;
  return PCP_OK;
}

int32_t pcp_model_n_units(const pcp_ctx* c, uint32_t* n_units, uint32_t* n_props) {
  if (!c) return PCP_ERR_ARG;
  if (n_units) *n_units = c->n_units;
  if (n_props) *n_props = (uint32_t)c->props.size();
  return PCP_OK;
}

int32_t pcp_set_option(pcp_ctx* c, const char* key, int64_t value) {
  if (!c || !key) return PCP_ERR_ARG;
  std::string k(key);
  if (k == "block_threads") {
    if (value != 256 && value != 512 && value != 1024) return fail(c, PCP_ERR_ARG, "block_threads must be 256, 512 or 1024");
    c->opt_block = value;
  } else if (k == "nodes_per_block") {
    if (value < 0 || value > 32) return fail(c, PCP_ERR_ARG, "nodes_per_block must be in [0,32]");
    c->opt_nodes_per_block = value;
  } else if (k == "force_path") {
    if (value < 0 || value > 2) return fail(c, PCP_ERR_ARG, "force_path must be 0, 1 or 2");
    c->opt_force_path = value;
  } else if (k == "solo_cascade") {
    if (value < 0 || value > 1) return fail(c, PCP_ERR_ARG, "solo_cascade must be 0 or 1");
    c->opt_solo = value;
  } else if (k == "team") {
    if (value < 0 || value > 4096) return fail(c, PCP_ERR_ARG, "team must be in [0,4096]");
    c->opt_team = value;
  } else if (k == "global_dom") {
    if (value < 0 || value > 2) return fail(c, PCP_ERR_ARG, "global_dom must be 0, 1 (force the HBM-resident variant) or 2 (force it, 10-bit LDS cells allowed)");
    c->opt_global_dom = value;
  } else if (k == "branch_reverse") {
    if (value < 0 || value > 1) return fail(c, PCP_ERR_ARG, "branch_reverse must be 0 or 1");
    c->opt_branch_reverse = value;
  } else if (k == "word_level") {
    if (value < 0 || value > 1) return fail(c, PCP_ERR_ARG, "word_level must be 0 or 1");
    c->opt_word_level = value;
  } else if (k == "packed") {
    if (value < 0 || value > 1) return fail(c, PCP_ERR_ARG, "packed must be 0 or 1");
    c->opt_packed = value;
  } else if (k == "dom10") {
    if (value < 0 || value > 1) return fail(c, PCP_ERR_ARG, "dom10 must be 0 or 1");
    c->opt_dom10 = value;
  } else if (k == "group_level") {
    if (value < 0 || value > 1) return fail(c, PCP_ERR_ARG, "group_level must be 0 or 1");
    c->opt_group_level = value;
  } else if (k == "implicit_active") {
    if (value < 0 || value > 1) return fail(c, PCP_ERR_ARG, "implicit_active must be 0 or 1");
    c->opt_implicit = value;
  } else if (k == "neq_path") {
    if (value < 0 || value > 1) return fail(c, PCP_ERR_ARG, "neq_path must be 0 or 1");
    c->opt_neq_path = value;
  } else if (k == "big_path") {
    if (value < 0 || value > 1) return fail(c, PCP_ERR_ARG, "big_path must be 0 or 1");
    c->opt_big_path = value;
  } else if (k == "small_alldiff") {
    if (value < 0 || value > 1) return fail(c, PCP_ERR_ARG, "small_alldiff must be 0 or 1");
    c->opt_small_alldiff = value;
  } else if (k == "time_kernels") {
    if (value < 0 || value > 1) return fail(c, PCP_ERR_ARG, "time_kernels must be 0 or 1");
    c->opt_time_kernels = value;
  } else if (k == "small_path") {
    if (value < 0 || value > 1) return fail(c, PCP_ERR_ARG, "small_path must be 0 or 1");
    c->opt_small_path = value;
  } else if (k == "big_bank") {
    if (value != 0 && value != 1) return fail(c, PCP_ERR_ARG, "big_bank must be 0 or 1");
    if (c->opt_big_bank != value) c->recs_by_kind_valid = false;  // (the table is rebuilt on its next use)
    c->opt_big_bank = value;
  } else if (k == "big_dense_k") {
    if (value < 1 || value > 64) return fail(c, PCP_ERR_ARG, "big_dense_k must be in [1,64]");
    c->opt_big_dense_k = value;
  } else if (k == "big_round") {
    if (value < 0 || value > 2) return fail(c, PCP_ERR_ARG, "big_round must be 0 (auto), 1 (dense rounds only) or 2 (sparse rounds only)");
    c->opt_big_round = value;
  } else if (k == "neq_dynamic") {
    if (value < 0 || value > 1024) return fail(c, PCP_ERR_ARG, "neq_dynamic must be in [0,1024]");
    c->opt_neq_dynamic = value;
  } else if (k == "neq_wgs") {
    if (value < 1 || value > 8) return fail(c, PCP_ERR_ARG, "neq_wgs must be in [1,8]");
    c->opt_neq_wgs = value;
  } else if (k == "neq_dfs_wgs") {
    if (value < 0 || value > 8) return fail(c, PCP_ERR_ARG, "neq_dfs_wgs must be in [0,8]");
    c->opt_neq_dfs_wgs = value;
  } else if (k == "neq_stagger") {
    if (value < 0 || value > 1000000) return fail(c, PCP_ERR_ARG, "neq_stagger must be in [0, 1000000] cycles");
    c->opt_neq_stagger = value;
  } else if (k == "neq_hint") {
    if (value < 0 || value > 1) return fail(c, PCP_ERR_ARG, "neq_hint must be 0 or 1");
    c->opt_neq_hint = value;
  } else if (k == "neq_dfs_block") {
    if (value != 0 && value != 128 && value != 256 && value != 512) return fail(c, PCP_ERR_ARG, "neq_dfs_block must be 0, 128, 256 or 512");
    c->opt_neq_dfs_block = value;
  } else if (k == "neq_dfs") {
    if (value < 0 || value > 1) return fail(c, PCP_ERR_ARG, "neq_dfs must be 0 or 1");
    c->opt_neq_dfs = value;
  } else if (k == "neq_persist") {
    if (value < 0 || value > 1) return fail(c, PCP_ERR_ARG, "neq_persist must be 0 or 1");
    c->opt_neq_persist = value;
  } else if (k == "neq_debug") {
    c->opt_neq_debug = value;
  } else if (k == "neq_trace_ptr") {
    c->opt_neq_trace = value;
  } else if (k == "neq_block") {
    if (value < 0 || value > 1024 || (value & 63)) return fail(c, PCP_ERR_ARG, "neq_block must be 0 or a multiple of 64 up to 1024");
    c->opt_neq_block = value;
  } else if (k == "list_cap") {
    if (value < 64 || value > 16384) return fail(c, PCP_ERR_ARG, "list_cap must be in [64,16384]");
    c->opt_list_cap = value;
  } else {
    return fail(c, PCP_ERR_ARG, "unknown option");
  }
  return PCP_OK;
}

int32_t pcp_propagate_device(pcp_ctx* c, uint32_t n_nodes, const pcp_device_batch* bt, void* hip_stream) {
  if (!c || !bt) return PCP_ERR_ARG;
  hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
  HIP_TRY(c, hipSetDevice(c->device));
  int32_t rc = finalize_model(c);
  if (rc) return rc;
  c->ev_valid = false;
  if (n_nodes == 0) return PCP_OK;
  if (!bt->status) return fail(c, PCP_ERR_ARG, "status must not be null");
  if (bt->cell_format > PCP_CELLS_PACKED16 || bt->reserved) return fail(c, PCP_ERR_ARG, "unknown cell_format (or reserved != 0)");
  if (bt->cell_format == PCP_CELLS_PACKED16) {
    // rows of packed cells: the all-XNeqY kernel alone reads and writes them
    if (c->set_words || c->has_formulas || bt->active_in || bt->active_out || !c->neq_model || !c->opt_neq_path || !c->opt_implicit || c->opt_force_path == 2 ||
        c->opt_global_dom || c->dfs_sp)
      return fail(c, PCP_ERR_UNSUPPORTED, "cell_format PCP_CELLS_PACKED16: an all-XNeqY model over implicit nodes in interval mode only");
    if (!bt->lb_in || !bt->lb_out) return fail(c, PCP_ERR_ARG, "cell rows (lb_in / lb_out) must not be null");
    const int32_t rcn = propagate_neq_device(c, n_nodes, bt, stream);
    return rcn == 1 ? fail(c, PCP_ERR_UNSUPPORTED, "cell_format PCP_CELLS_PACKED16: the store does not fit LDS") : rcn;
  }
  if (c->n_vars && !c->set_words && (!bt->lb_in || !bt->ub_in || !bt->lb_out || !bt->ub_out)) return fail(c, PCP_ERR_ARG, "domain pointers must not be null");
  const bool has_nu = c->cur_nu_off != nullptr;
  if (has_nu && (c->set_words || c->has_formulas || c->dfs_sp))
    return fail(c, PCP_ERR_UNSUPPORTED, "node units: interval mode, stores without formula propagators (set mode needs none: exact set operations on `bits`)");

  if (c->set_words) return propagate_set_device(c, n_nodes, bt, stream);
  const uint32_t P = (uint32_t)c->props.size();
  const uint32_t words = (P + 63) / 64;
  const uint32_t S = c->n_slots, Wv = (S + 31) / 32;
  const uint32_t block = (uint32_t)c->opt_block;
  const uint32_t list_cap = (uint32_t)c->opt_list_cap;

  if (c->has_formulas) {
    // the reified layer: every unit is evaluated as a tree, one lane per unit (pcp_formula.hip); `active` rows are unit-level there
    // one wavefront per node, four per workgroup; a store too large for four slices runs with fewer
    uint32_t fwaves = 4;
    while (fwaves > 1 && (!lds_bytes_formula(S, c->n_units, fwaves) || lds_bytes_formula(S, c->n_units, fwaves) > c->lds_max / 2)) fwaves /= 2;
    const size_t lds = lds_bytes_formula(S, c->n_units, fwaves);
    if (!lds || lds > c->lds_max) return fail(c, PCP_ERR_UNSUPPORTED, "a store with formula propagators must fit one CU's LDS (8 bytes per variable)");
    FormArgs a;
    memset(&a, 0, sizeof(a));
    a.m.recs = c->d_recs; a.m.const_val = c->d_const; a.m.n_recs = P; a.m.n_vars = c->n_vars; a.m.n_slots = S;
    a.m.sums = SumTab{c->d_sum_off, c->d_sum_mem, c->n_vars, c->n_sum_slots, c->d_mul_off};
    a.nodes = c->d_fnodes; a.unit_root = c->d_unit_root; a.n_units = c->n_units; a.n_nodes = n_nodes; a.violation = c->d_retry + 1;
    a.sp_ptr = c->dfs_sp; a.stop_ptr = c->dfs_stop;
    a.lb_in = bt->lb_in; a.ub_in = bt->ub_in; a.lb_out = bt->lb_out; a.ub_out = bt->ub_out;
    a.active_in = bt->active_in; a.active_out = bt->active_out; a.status = bt->status; a.stats = c->d_stats;
    LaunchPlan plan;
    plan.block = 64 * fwaves; plan.lds_bytes = lds;
    {  // persistent workgroups: what the chip holds at once
      const uint32_t per_cu = std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)(c->lds_max / lds), 2048u / plan.block));
      plan.grid = std::min<uint32_t>((n_nodes + fwaves - 1) / fwaves, per_cu * (uint32_t)c->num_cu);
    }
    c->last_plan = pcp_plan{1u, 1u, 0u, 0u, 0u, 0u, bt->active_in ? 0u : 1u, 0u, plan.grid, plan.block, (uint32_t)plan.lds_bytes, 0u, 3u};
    if (c->opt_time_kernels) HIP_TRY(c, hipEventRecord(c->ev_start, stream));
    HIP_TRY(c, launch_formfix(a, plan, stream));
    if (c->opt_time_kernels) HIP_TRY(c, hipEventRecord(c->ev_stop, stream));
    c->ev_valid = c->opt_time_kernels != 0;
    return PCP_OK;
  }
  // a small store — at most 128 slots, 2048 records, no formulas: one wavefront per node (pcp_small.hip, plan.path 4).  Explicit rows and
  // implicit nodes alike.  Any option that asks for a particular geometry of the generic kernels keeps those kernels.
  // (a call with node units takes this kernel whatever the options say: it is the one that reads them)
  if (has_nu && !(S <= 128u && P <= 2048u && (bt->active_in != nullptr || c->opt_implicit)))
    return fail(c, PCP_ERR_UNSUPPORTED, "node units: stores of at most 128 variables (incl. interned constants) and 2048 elementary filters (the one-wavefront-per-node kernel)");
  if (has_nu || (c->opt_small_path && S <= 128u && P <= 2048u && !c->opt_force_path && !c->opt_nodes_per_block && !c->opt_global_dom && !c->opt_team &&
      (bt->active_in != nullptr || c->opt_implicit) && !(c->neq_model && c->opt_neq_path && bt->active_in == nullptr && n_nodes >= 64))) {
    const uint32_t waves = 4;
    const size_t lds = lds_bytes_small(S, c->n_units, P, waves);
    if (has_nu && !(lds && lds <= c->lds_max)) return fail(c, PCP_ERR_UNSUPPORTED, "node units: the store does not fit the one-wavefront-per-node kernel");
    if (lds && lds <= c->lds_max) {
      SmallArgs a;
      memset(&a, 0, sizeof(a));
      a.m.recs = c->d_recs; a.m.const_val = c->d_const; a.m.n_recs = P; a.m.n_vars = c->n_vars; a.m.n_slots = S; a.m.has_ternary = c->has_ternary;
      a.m.sums = SumTab{c->d_sum_off, c->d_sum_mem, c->n_vars, c->n_sum_slots, c->d_mul_off};
      a.rec_unit = c->has_groups ? c->d_rec_unit : nullptr; a.n_units = c->n_units; a.n_nodes = n_nodes;
      if (c->n_alldiff && c->opt_small_alldiff) { a.ad_tab = c->d_ad_tab; a.ad_vars = c->d_ad_vars; a.ad_mask = c->d_ad_mask; }
      a.violation = c->d_retry + 1; a.dbg = c->d_dbg; a.sp_ptr = c->dfs_sp; a.stop_ptr = c->dfs_stop;
      a.nu_off = c->cur_nu_off; a.nu = c->cur_nu;
      a.lb_in = bt->lb_in; a.ub_in = bt->ub_in; a.lb_out = bt->lb_out; a.ub_out = bt->ub_out;
      a.active_in = bt->active_in; a.active_out = bt->active_out; a.status = bt->status; a.stats = c->d_stats;
      LaunchPlan plan;
      plan.block = 64 * waves; plan.lds_bytes = lds;
      const uint32_t per_cu = std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)(c->lds_max / lds), 2048u / plan.block));
      plan.grid = std::min<uint32_t>((n_nodes + waves - 1) / waves, per_cu * (uint32_t)c->num_cu);
      c->last_plan = pcp_plan{1u, 1u, 0u, 0u, 0u, 0u, bt->active_in ? 0u : 1u, 0u, plan.grid, plan.block, (uint32_t)plan.lds_bytes, 0u, 4u};
      if (!c->dfs_sp && c->opt_time_kernels) HIP_TRY(c, hipEventRecord(c->ev_start, stream));
      HIP_TRY(c, launch_smallfix(a, plan, stream));
      if (!c->dfs_sp && c->opt_time_kernels) HIP_TRY(c, hipEventRecord(c->ev_stop, stream));
      c->ev_valid = !c->dfs_sp && c->opt_time_kernels != 0;
      return PCP_OK;
    }
  }
  const bool implicit = bt->active_in == nullptr && c->opt_implicit;  // see below (a.live == nullptr)
  if (implicit && c->neq_model && c->opt_neq_path && c->opt_force_path != 2 && !c->opt_global_dom) {
    int32_t rcn = propagate_neq_device(c, n_nodes, bt, stream);
    if (rcn != 1) return rcn;  // 1 = the store does not fit LDS: the generic kernels (HBM-resident domains) take it
  }
  // a store too large for (lb, ub) pairs in LDS, binary records only, a declared hull of at most 1024 values, implicit nodes, enough
  // nodes to give every CU one: the 10-bit-cell kernel (pcp_big.hip)
  bool big_fits = false;
  if (implicit && P && c->opt_big_path && c->opt_dom10 && c->opt_global_dom != 1 && c->opt_force_path != 2 && c->have_adjp && !c->n_sum_slots && c->hull_set &&
      (int64_t)c->hull_hi - c->hull_lo <= 1023 && (c->opt_global_dom == 2 || !lds_bytes_for(S, 1, 256, block)) && lds_bytes_big(c->n_vars, S) &&
      lds_bytes_big(c->n_vars, S) <= c->lds_max && (c->opt_global_dom == 2 || n_nodes * 2 > (uint32_t)c->num_cu || words < 64)) {
    if (!c->recs_by_kind_valid) {  // built on first use: only stores that take this path need it
      const size_t Ppad = (size_t)P ? ((size_t)P + 255) / 256 * 256 + kStreamPadRecs : 0;
      std::vector<Rec> all(Ppad);
      HIP_TRY(c, hipMemcpy(all.data(), c->d_recs, all.size() * sizeof(Rec), hipMemcpyDeviceToHost));
      std::vector<uint32_t> adj_off(c->n_vars + 1), adj;
      HIP_TRY(c, hipMemcpy(adj_off.data(), c->d_adj_off, adj_off.size() * 4, hipMemcpyDeviceToHost));
      adj.resize(adj_off[c->n_vars]);
      if (!adj.empty()) HIP_TRY(c, hipMemcpy(adj.data(), c->d_adj, adj.size() * 4, hipMemcpyDeviceToHost));
      std::vector<int32_t> consts(S - c->n_vars);
      if (!consts.empty()) HIP_TRY(c, hipMemcpy(consts.data(), c->d_const, consts.size() * 4, hipMemcpyDeviceToHost));
      c->big_ok = c->n_vars < 98304u;
      // a record with a Constant operand becomes a unary record  var (op) K:  x (kind) c + d  /  c (kind) y + d  <=>  y (>, =, !=) c - d
      struct BR { uint2 r; uint32_t key; };
      std::vector<BR> br(P);
      auto coord = [](uint32_t slot) { return (slot / 3u) | ((slot % 3u) << 15); };
      const uint32_t nv = c->n_vars;
      for (size_t r = 0; r < P && c->big_ok; ++r) {
        const uint32_t x = all[r].xk & kSlotMask, y = all[r].y, kind = all[r].xk >> 28;
        const int64_t d = all[r].d;
        if (kind > PCP_LT || (x >= nv && y >= nv)) { c->big_ok = false; break; }
        // (the folded constant K is computed in 64 bits and must stay far inside int32: the kernel forms K - lo, K - 1 and K + 1; a model
        // whose constant and offset add up to more than 2^30 in magnitude goes to the generic kernels instead of being wrapped)
        constexpr int64_t kFoldMax = (int64_t)1 << 30;
        if (y >= nv) {         // x (kind) K,  K = c + d
          const uint32_t op = kind == PCP_LT ? 0u : kind == PCP_EQ ? 2u : 3u;
          const int64_t K = (int64_t)consts[y - nv] + d;
          if (K < -kFoldMax || K > kFoldMax) { c->big_ok = false; break; }
          br[r] = BR{make_uint2(coord(x) | (op << 17) | (3u << 30), (uint32_t)(int32_t)K), 3u};
        } else if (x >= nv) {  // c (kind) y + d:  LT  y > c - d  |  EQ  y = c - d  |  NEQ  y != c - d
          const uint32_t op = kind == PCP_LT ? 1u : kind == PCP_EQ ? 2u : 3u;
          const int64_t K = (int64_t)consts[x - nv] - d;
          if (K < -kFoldMax || K > kFoldMax) { c->big_ok = false; break; }
          br[r] = BR{make_uint2(coord(y) | (op << 17) | (3u << 30), (uint32_t)(int32_t)K), 3u};
        } else {
          if (d < -4095 || d > 4095) { c->big_ok = false; break; }
          br[r] = BR{make_uint2(coord(x) | (((uint32_t)(int32_t)d & 0x1fffu) << 17) | (kind << 30), coord(y)), kind};
        }
      }
      if (c->big_ok) {
        // the adjacency payloads first (they follow ModelDev::adj, which names records of the UNSORTED table)
        std::vector<uint2> badj(adj.size());
        for (uint32_t v = 0; v < nv; ++v)
          for (uint32_t k = adj_off[v]; k < adj_off[v + 1]; ++k) {
            const Rec& rec = all[adj[k]];
            const BR& b = br[adj[k]];
            const uint32_t x = rec.xk & kSlotMask, y = rec.y, kind = rec.xk >> 28;
            if (b.key == 3u) { badj[k] = make_uint2(0x7fffu | (((b.r.x >> 17) & 3u) << 18) | (1u << 20), b.r.y); continue; }
            const bool is_y = x != v;
            badj[k] = make_uint2(coord(is_y ? x : y) | ((is_y ? 1u : 0u) << 17) | (kind << 18), (uint32_t)rec.d);
          }
        std::stable_sort(br.begin(), br.end(), [](const BR& p, const BR& q) { return p.key < q.key; });
        // Bank-aware order within a kind (round 6; the order of the table is free — every fair schedule reaches the same fixpoint, DESIGN.md §2 —
        // and the model is immutable).  A cell word is 8 bytes = one PAIR of LDS banks, and a wavefront's ds_read_b64 / 64-bit compare-and-swap is
        // served one 32-lane half at a time: it is conflict-free iff the half's 32 word indices differ mod 32.  With the model's own (random) order a
        // half hit ~12 distinct bank pairs out of 32 twice or more: SQ_LDS_BANK_CONFLICT was 0.44 of SQ_LDS_IDX_ACTIVE (profiles/r05_c3_*).  Greedy:
        // the records of a kind are dealt from 32 buckets (x word mod 32), one per lane of a half, preferring among a bucket's next few records one
        // whose y word falls on a bank pair the half has not used yet.  Neighbouring lanes then never compare-and-swap the same word either.
        if (c->opt_big_bank) {
          size_t s0 = 0;
          while (s0 < P) {
            size_t e0 = s0;
            while (e0 < P && br[e0].key == br[s0].key) ++e0;
            std::vector<uint32_t> bucket[32];
            for (size_t r = s0; r < e0; ++r) bucket[br[r].r.x & 31u].push_back((uint32_t)r);
            size_t head[32] = {0};
            std::vector<BR> out;
            out.reserve(e0 - s0);
            const bool unary = br[s0].key == 3u;
            size_t pos = s0;           // table position of the next record: halves are positions [32 h, 32 h + 32)
            uint32_t usedx = 0, usedy = 0;
            uint32_t rot = 0;
            while (out.size() < e0 - s0) {
              if ((pos & 31u) == 0) { usedx = 0; usedy = 0; }
              // the fullest bucket whose bank pair this half has not used (ties: rotate), else the fullest bucket at all
              int best = -1; size_t best_n = 0;
              for (uint32_t i = 0; i < 32; ++i) {
                const uint32_t b = (i + rot) & 31u;
                const size_t n = bucket[b].size() - head[b];
                if (n > best_n && !((usedx >> b) & 1u)) { best_n = n; best = (int)b; }
              }
              if (best < 0)
                for (uint32_t b = 0; b < 32; ++b) { const size_t n = bucket[b].size() - head[b]; if (n > best_n) { best_n = n; best = (int)b; } }
              std::vector<uint32_t>& bk = bucket[best];
              size_t pick = head[best];
              if (!unary)
                for (size_t k2 = head[best]; k2 < std::min(bk.size(), head[best] + 16); ++k2)
                  if (!((usedy >> (br[bk[k2]].r.y & 31u)) & 1u)) { pick = k2; break; }
              std::swap(bk[pick], bk[head[best]]);
              const BR& chosen = br[bk[head[best]++]];
              usedx |= 1u << (chosen.r.x & 31u);
              if (!unary) usedy |= 1u << (chosen.r.y & 31u);
              out.push_back(chosen);
              ++pos; ++rot;
            }
            std::copy(out.begin(), out.end(), br.begin() + s0);
            s0 = e0;
          }
        }
        std::vector<uint2> brec(Ppad);
        for (size_t r = 0; r < Ppad; ++r) brec[r] = br[std::min<size_t>(r, P - 1)].r;
        if ((rc = ensure(c, c->d_brec, c->cap_brec, brec.size()))) return rc;
        if ((rc = ensure(c, c->d_badj, c->cap_badj, std::max<size_t>(badj.size(), 1)))) return rc;
        HIP_TRY(c, hipMemcpy(c->d_brec, brec.data(), brec.size() * sizeof(uint2), hipMemcpyHostToDevice));
        if (!badj.empty()) HIP_TRY(c, hipMemcpy(c->d_badj, badj.data(), badj.size() * sizeof(uint2), hipMemcpyHostToDevice));
      }
      c->recs_by_kind_valid = true;
    }
    big_fits = c->big_ok;
  }
  if (big_fits) {
    BigArgs a;
    memset(&a, 0, sizeof(a));
    a.brec = c->d_brec; a.badj = c->d_badj; a.has_unary = S != c->n_vars ? 1u : 0u; a.dense_k = (uint32_t)c->opt_big_dense_k;
    a.m.recs = c->d_recs; a.m.adj_off = c->d_adj_off; a.m.adj = c->d_adj; a.m.adjp = c->d_adjp; a.m.const_val = c->d_const;
    a.m.n_recs = P; a.m.n_vars = c->n_vars; a.m.n_slots = S; a.m.uniform_kind = c->uniform_kind; a.m.max_deg = c->max_deg;
    a.n_nodes = n_nodes; a.lo10 = c->hull_lo; a.violation = c->d_retry + 1; a.dbg = c->d_dbg; a.round_mode = (uint32_t)c->opt_big_round;
    a.sp_ptr = c->dfs_sp; a.stop_ptr = c->dfs_stop;
    a.lb_in = bt->lb_in; a.ub_in = bt->ub_in; a.lb_out = bt->lb_out; a.ub_out = bt->ub_out; a.status = bt->status; a.stats = c->d_stats;
    LaunchPlan plan;
    plan.grid = n_nodes; plan.block = 1024; plan.lds_bytes = lds_bytes_big(c->n_vars, S);
    c->last_plan = pcp_plan{1u, 1u, 0u, 0u, 2u, 0u, 1u, 0u, plan.grid, plan.block, (uint32_t)plan.lds_bytes, 0u, 2u};
    if (c->opt_time_kernels) HIP_TRY(c, hipEventRecord(c->ev_start, stream));
    HIP_TRY(c, launch_bigfix(a, plan, stream));
    if (c->opt_time_kernels) HIP_TRY(c, hipEventRecord(c->ev_stop, stream));
    if (bt->active_out && P) {  // the `active` rows on request: record r is live iff it is not entailed under the final domains
      ModelDev m = a.m;
      m.sums = SumTab{c->d_sum_off, c->d_sum_mem, c->n_vars, c->n_sum_slots, c->d_mul_off};
      if (c->has_groups) {
        if ((rc = ensure(c, c->d_live, c->cap_live, (size_t)n_nodes * std::max<uint32_t>(words, 1)))) return rc;
        HIP_TRY(c, launch_derive_active(m, bt->lb_out, bt->ub_out, c->d_live, n_nodes, stream));
        HIP_TRY(c, launch_contract_units(c->d_unit_first, c->n_units, P, c->d_live, bt->active_out, n_nodes, stream));
      } else {
        HIP_TRY(c, launch_derive_active(m, bt->lb_out, bt->ub_out, bt->active_out, n_nodes, stream));
      }
    }
    c->ev_valid = c->opt_time_kernels != 0;
    return PCP_OK;
  }
  // ---- choose the path: B nodes per workgroup (batch) or a team of G workgroups per node ------------------
  // tile sizes the kernel is instantiated for (pcp_kernels.hip launch_fixpoint)
  static const uint32_t kTiles[] = {16, 12, 8, 4, 2, 1};
  // The changed-(node,var) list shares LDS with the domains: shrink it (down to 256 entries) before giving up
  // a tile size; a round with more changed pairs than the list holds falls back to a filtered sweep.
  uint32_t list_cap_used = list_cap;
  auto fits_cap = [&](uint32_t b, uint32_t cap) { size_t need = lds_bytes_for(S, b, cap, block); return need && need <= c->lds_max; };
  auto fits = [&](uint32_t b) {
    for (uint32_t cap = list_cap;; cap /= 2) {
      if (fits_cap(b, cap)) return cap;
      if (cap <= 256) return 0u;
    }
  };
  auto tile_le = [&](uint32_t want) {  // largest instantiated tile <= want that fits in LDS (0 if none)
    for (uint32_t t : kTiles)
      if (t <= want)
        if (uint32_t cap = fits(t)) { list_cap_used = cap; return t; }
    return 0u;
  };
  // Variable stores too large for LDS (8 bytes per variable and node) run with the domains left in HBM/L2.
  const bool global_dom = c->opt_global_dom || !fits(1);
  if (global_dom) {
    uint32_t cap = list_cap;
    while (cap > 256 && !lds_bytes_global(c->n_vars, S, cap)) cap /= 2;
    if (!lds_bytes_global(c->n_vars, S, cap)) return fail(c, PCP_ERR_UNSUPPORTED, "variable store too large even for the bitmasks in LDS");
    list_cap_used = cap;
  }
  const uint32_t slots = (uint32_t)c->num_cu;  // one resident workgroup per CU is what large tiles allow
  uint32_t B = 1, team = 1;
  bool use_team = false;
  if (c->opt_force_path == 2) use_team = true;
  else if (c->opt_force_path == 0) use_team = (n_nodes * 2 <= slots) && words >= 64;
  if (use_team) {
    // measured on N-queens-1000 (23 415 words): 128-256 blocks per node is the sweet spot; more blocks pay
    // their fixed cost (domain staging, merge, ticket) without shortening the slice stream further.
    uint32_t g = c->opt_team ? (uint32_t)c->opt_team : std::max<uint32_t>(1, slots / n_nodes);
    const uint32_t max_by_work = std::max<uint32_t>(1, words / (c->opt_team ? 1 : 128));
    team = std::max<uint32_t>(1, std::min(g, max_by_work));
  } else {
    const uint32_t want = c->opt_nodes_per_block ? (uint32_t)c->opt_nodes_per_block : std::max<uint32_t>(1, (n_nodes + slots - 1) / slots);
    B = global_dom ? 1u : std::max<uint32_t>(1, tile_le(want));
  }
  // Packed tiles (16-bit cells, pcp_kernels.hip LdsDom16): twice the nodes per workgroup in the same LDS and half the
  // LDS traffic per filter step.  They need every bound within +-kPackedMax; that is checked per tile on the device
  // while the domains are staged, and a tile that does not fit is handed back to a second launch with 32-bit cells
  // and half the tile size (same LDS footprint, so it fits whenever the packed tile did).
  uint32_t Bp = 0, cap_p = 0, cap_half = 0, wl_used = 0;
  const bool hull_fits16 = c->hull_set && c->hull_lo >= -kPackedMax && c->hull_hi <= kPackedMax;
  if (!use_team && !global_dom && c->opt_packed && c->compact && c->consts_fit16 && (!c->hull_set || hull_fits16)) {
    const uint32_t want = c->opt_nodes_per_block ? (uint32_t)c->opt_nodes_per_block : std::max<uint32_t>(1, (n_nodes + slots - 1) / slots);
    // The word-group sweep keeps B live registers per lane (B <= 16) and notes the words for the record level in a
    // bitmap that borrows the changed-pair list's LDS: one u64 per 64 words, so list_cap must not drop below that.
    const uint32_t groups = (words + 63) / 64;
    // (the sweep counts live records in 16-bit lane counters: at most 1023 groups of 64 words per wavefront)
    const uint32_t wl_model = (c->opt_word_level && groups <= 1000u * (block / 64u)) ? c->word_level : 0;
    for (uint32_t t : {32u, 16u, 8u}) {
      if (t > want) continue;
      // implicit nodes: the maximum tables as well (word_level 2) — the level -1 test then also clears words that are
      // entailed throughout the tile, which it otherwise has to hand to the record level at every node again
      const uint32_t wl_first = (implicit && wl_model) ? 2u : wl_model;
      for (uint32_t wl : {wl_first, wl_model, 0u}) {
        if (wl && t > 16) continue;
        auto need_for = [&](uint32_t cap) { return lds_bytes_for(S, t, cap, block, true, wl); };
        const uint32_t cap_min = std::max<uint32_t>(256, wl ? groups : 0);
        uint32_t cap = std::max(list_cap, cap_min);
        while (cap / 2 >= cap_min && !(need_for(cap) && need_for(cap) <= c->lds_max)) cap /= 2;
        const size_t need = need_for(cap);
        if (!need || need > c->lds_max) continue;
        const uint32_t ch = fits(t / 2);
        if (!ch) continue;
        Bp = t; cap_p = cap; cap_half = ch; wl_used = wl;
        break;
      }
      if (Bp) break;
      if (!wl_model) continue;
    }
    if (Bp * 2 <= B) Bp = 0;  // a 32-bit tile with at least twice the nodes wins
  }
  LaunchPlan plan;
  plan.block = block;
  if (use_team && !global_dom) list_cap_used = fits(1);
  if (Bp) { B = Bp; list_cap_used = cap_p; }
  plan.lds_bytes = global_dom ? lds_bytes_global(c->n_vars, S, list_cap_used) : lds_bytes_for(S, B, list_cap_used, block, Bp != 0, Bp ? wl_used : 0);
  plan.grid = team > 1 ? n_nodes * team : (n_nodes + B - 1) / B;
  const size_t adj_cache_bytes = (((size_t)c->n_vars + 1) * 4 + 15) & ~(size_t)15;
  const bool adj_cache = plan.lds_bytes + adj_cache_bytes <= c->lds_max;  // LDS copy of adj_off behind the carve
  if (adj_cache) plan.lds_bytes += adj_cache_bytes;

  LaunchArgs a;
  memset(&a, 0, sizeof(a));
  a.m.recs = c->d_recs; a.m.recs8 = c->compact ? c->d_recs8 : nullptr; a.m.adj_off = c->d_adj_off; a.m.adj = c->d_adj; a.m.adjp = c->have_adjp ? c->d_adjp : nullptr; a.m.const_val = c->d_const;
  a.m.n_recs = P; a.m.n_vars = c->n_vars; a.m.n_slots = S; a.m.has_ternary = c->has_ternary; a.m.uniform_kind = c->uniform_kind; a.m.max_deg = c->max_deg;
  a.m.sums = SumTab{c->d_sum_off, c->d_sum_mem, c->n_vars, c->n_sum_slots, c->d_mul_off};
  a.n_nodes = n_nodes; a.nodes_per_block = B; a.team = team; a.list_cap = list_cap_used; a.global_dom = global_dom ? 1u : 0u;
  a.adj_cache = adj_cache ? 1u : 0u;
  a.solo = (uint32_t)c->opt_solo;
  a.violation = c->d_retry + 1;
  a.packed = Bp ? 1u : 0u; a.word_level = Bp ? wl_used : 0u; a.m.wdesc = c->d_wdesc; a.m.gdesc = (c->word_level && c->opt_group_level) ? c->d_gdesc : nullptr; a.retry_flag = c->d_retry; a.epoch = Bp ? ++c->epoch : 0u;
  // with a declared hull there is no retry launch: a tile outside the hull raises the STICKY violation word (d_retry[1]),
  // which stays set until pcp_stats_read has reported it — whatever is launched in between
  if (Bp && c->hull_set && c->hull_lo >= -kPackedMax && c->hull_hi <= kPackedMax) a.retry_flag = c->d_retry + 1;
  // HBM-resident variable store with a declared hull of at most 1024 values: the domains fit LDS after all as 10-bit cells
  bool dom10 = global_dom && team == 1 && c->opt_global_dom != 1 && c->opt_dom10 && c->hull_set && (int64_t)c->hull_hi - c->hull_lo <= 1023;
  if (dom10) {
    // the cells share LDS with the changed-pair lists: shrink those (down to 256 entries) before giving up on the cells
    uint32_t cap = list_cap_used;
    auto total = [&](uint32_t cp) { return lds_bytes_global(c->n_vars, S, cp) + dom10_bytes(c->n_vars); };
    while (cap > 256 && (!lds_bytes_global(c->n_vars, S, cap) || total(cap) > c->lds_max)) cap /= 2;
    dom10 = lds_bytes_global(c->n_vars, S, cap) && total(cap) <= c->lds_max;
    if (dom10) {
      list_cap_used = cap; a.list_cap = cap; a.adj_cache = 0;
      plan.lds_bytes = total(cap);
      a.dom10 = 1u; a.dom10_lo = c->hull_lo; a.retry_flag = c->d_retry + 1;
    }
  }
  a.sp_ptr = c->dfs_sp; a.stop_ptr = c->dfs_stop;
  a.lb_in = bt->lb_in; a.ub_in = bt->ub_in; a.lb_out = bt->lb_out; a.ub_out = bt->ub_out;
  a.live_in = bt->active_in;
  a.status = bt->status;
  a.stats = c->d_stats;
  const uint32_t unit_words = (c->n_units + 63) / 64;
  // active_in == NULL: every unit active on entry.  Then liveness needs no rows at all: a unit that is entailed runs as a
  // no-op, so "active" can be DERIVED (inactive <=> entailed under the current domains) and the node is its
  // domains only — the implicit-active path (a.live == nullptr).  Results are identical to the explicit path by construction.
  if (implicit) {
    a.live_in = nullptr;
    a.live = nullptr;
  } else if (c->has_groups) {
    // record-level live rows in scratch, seeded from the unit-level `active` rows (in place for the kernel)
    if ((rc = ensure(c, c->d_live, c->cap_live, (size_t)n_nodes * std::max<uint32_t>(words, 1)))) return rc;
    HIP_TRY(c, launch_expand_units(c->d_rec_unit, P, unit_words, bt->active_in, c->d_live, n_nodes, stream));
    a.live_in = c->d_live;
    a.live = c->d_live;
  } else if (bt->active_out) {
    a.live = bt->active_out;
  } else {
    if ((rc = ensure(c, c->d_live, c->cap_live, (size_t)n_nodes * std::max<uint32_t>(words, 1)))) return rc;
    a.live = c->d_live;
  }
  if (team > 1) {
    // per node: ticket, remaining, fail (3 words) + Wv changed words + 4 u64 counters
    const size_t per_node_words = 4 + Wv + 2 * kTeamCounters;
    const size_t nwords = (size_t)n_nodes * per_node_words + 2;
    if ((rc = ensure(c, c->d_team, c->cap_team, nwords))) return rc;
    // (inside pcp_dfs_device the step kernel hands the scratch back zeroed: only the first step of a call clears it here)
    if (!(c->dfs_sp && c->dfs_team_words == nwords)) HIP_TRY(c, hipMemsetAsync(c->d_team, 0, nwords * 4, stream));
    c->dfs_team_words = c->dfs_sp ? nwords : 0;
    uint32_t* base = c->d_team;
    a.team_counters = reinterpret_cast<uint64_t*>(base);                 // [n_nodes][kTeamCounters] u64 (8-byte aligned at base)
    a.team_ticket = base + (size_t)n_nodes * 2 * kTeamCounters;
    a.team_remaining = a.team_ticket + n_nodes;
    a.team_fail = a.team_remaining + n_nodes;
    a.team_chg = a.team_fail + n_nodes + (n_nodes & 1);                  // keep alignment tidy
  }
  if ((team > 1 || global_dom) && !dom10 && bt->lb_out != bt->lb_in && c->n_vars) {
    // team: every slice narrows the node's rows in lb_out/ub_out with atomics; global_dom: they are the working set
    HIP_TRY(c, hipMemcpyAsync(bt->lb_out, bt->lb_in, (size_t)n_nodes * c->n_vars * 4, hipMemcpyDeviceToDevice, stream));
    HIP_TRY(c, hipMemcpyAsync(bt->ub_out, bt->ub_in, (size_t)n_nodes * c->n_vars * 4, hipMemcpyDeviceToDevice, stream));
  }
  c->last_plan = pcp_plan{B, team, Bp ? 1u : 0u, a.word_level, dom10 ? 2u : a.global_dom, a.m.recs8 ? 1u : 0u, implicit ? 1u : 0u, 0u, plan.grid, plan.block, (uint32_t)plan.lds_bytes, list_cap_used, 0u};
  if (!c->dfs_sp && c->opt_time_kernels) HIP_TRY(c, hipEventRecord(c->ev_start, stream));
  HIP_TRY(c, launch_fixpoint(a, plan, stream));
  if (Bp && hull_fits16) c->trusted_epoch = a.epoch;  // no retry launch: a tile outside the hull is the caller's contract violation (d_retry[1])
  if (Bp && !hull_fits16) {
    // the tiles the packed kernel handed back (normally none: every block of this launch returns at once)
    LaunchArgs a2 = a;
    a2.packed = 0; a2.word_level = 0; a2.only_marked = 1; a2.nodes_per_block = Bp / 2; a2.list_cap = cap_half;
    LaunchPlan plan2;
    plan2.block = block;
    plan2.lds_bytes = lds_bytes_for(S, Bp / 2, cap_half, block);
    a2.adj_cache = plan2.lds_bytes + adj_cache_bytes <= c->lds_max ? 1u : 0u;
    if (a2.adj_cache) plan2.lds_bytes += adj_cache_bytes;
    plan2.grid = (n_nodes + Bp / 2 - 1) / (Bp / 2);
    HIP_TRY(c, launch_fixpoint(a2, plan2, stream));
  }
  if (!c->dfs_sp && c->opt_time_kernels) HIP_TRY(c, hipEventRecord(c->ev_stop, stream));
  if (implicit && bt->active_out && P) {
    // the `active` rows on request: record r is live iff it is not entailed under the final domains
    if (c->has_groups) {
      if ((rc = ensure(c, c->d_live, c->cap_live, (size_t)n_nodes * std::max<uint32_t>(words, 1)))) return rc;
      HIP_TRY(c, launch_derive_active(a.m, bt->lb_out, bt->ub_out, c->d_live, n_nodes, stream));
      HIP_TRY(c, launch_contract_units(c->d_unit_first, c->n_units, P, c->d_live, bt->active_out, n_nodes, stream));
    } else {
      HIP_TRY(c, launch_derive_active(a.m, bt->lb_out, bt->ub_out, bt->active_out, n_nodes, stream));
    }
  } else if (c->has_groups && bt->active_out)
    HIP_TRY(c, launch_contract_units(c->d_unit_first, c->n_units, P, c->d_live, bt->active_out, n_nodes, stream));
  c->ev_valid = !c->dfs_sp && c->opt_time_kernels != 0;
  return PCP_OK;
}

static int32_t dfs_enqueue_steps(pcp_ctx* c, const pcp_dfs_state* st, uint32_t n, uint32_t stop_on_solution, uint64_t node_limit, hipStream_t stream) {
  pcp_device_batch bt;
  memset(&bt, 0, sizeof(bt));
  bt.lb_in = st->lb; bt.ub_in = st->ub; bt.lb_out = st->lb; bt.ub_out = st->ub; bt.status = st->status;  // in place, implicit-active
  int32_t rc = PCP_OK;
  for (uint32_t i = 0; i < n && rc == PCP_OK; ++i) {
    rc = pcp_propagate_device(c, 1, &bt, stream);
    if (rc == PCP_OK) {
      hipError_t e = launch_dfs_step(c->n_vars, st->lb, st->ub, st->status, st->capacity, st->sp, st->stop,
                                     reinterpret_cast<unsigned long long*>(st->counters), st->first_solution, stop_on_solution,
                                     (unsigned long long)node_limit, c->dfs_team_words ? c->d_team : nullptr, (uint32_t)c->dfs_team_words, stream);
      if (e != hipSuccess) rc = hip_fail(c, e, "launch_dfs_step");
    }
  }
  return rc;
}

int32_t pcp_dfs_forest_device_set(pcp_ctx* c, const pcp_forest_state* st, uint32_t n_steps, uint32_t stop_on_solution, uint64_t node_limit, void* hip_stream) {
  if (!c || !st) return PCP_ERR_ARG;
  if (!c->set_words) return fail(c, PCP_ERR_ARG, "pcp_dfs_forest_device_set needs a set-mode model (pcp_model_reset with set_words > 0)");
  if (!c->hull_set) return fail(c, PCP_ERR_CONTRACT, "set mode needs the hull of the initial domains (pcp_model_set_hull)");
  if ((int64_t)c->hull_hi - c->hull_lo >= (int64_t)c->set_words * 64) return fail(c, PCP_ERR_CONTRACT, "the declared hull does not fit set_words * 64 values");
  if (!st->n_trees || !st->bits || !st->tree || !st->levels || !st->trail || !st->counters || !st->total_nodes || !st->stop || !st->level_capacity || !st->trail_capacity)
    return fail(c, PCP_ERR_ARG, "null buffer / zero capacity");
  if ((st->first_solution == nullptr) != (st->solution_flag == nullptr)) return fail(c, PCP_ERR_ARG, "first_solution and solution_flag go together");
  hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
  HIP_TRY(c, hipSetDevice(c->device));
  { const int32_t rcf = finalize_model(c); if (rcf) return rcf; }
  const uint32_t P = (uint32_t)c->props.size(), S = c->n_slots;
  uint32_t cap = (uint32_t)std::min<int64_t>(c->opt_list_cap, 1024);
  while (cap > 64 && !lds_bytes_set_dfs(c->n_vars, S, c->set_words, cap)) cap /= 2;
  const size_t lds = lds_bytes_set_dfs(c->n_vars, S, c->set_words, cap);
  if (!lds || lds > c->lds_max) return fail(c, PCP_ERR_UNSUPPORTED, "set-mode variable store does not fit one CU's LDS (n_vars * (set_words + 1) * 8 bytes)");
  SetDfsArgs a;
  memset(&a, 0, sizeof(a));
  a.m.recs = c->d_recs; a.m.adj_off = c->d_adj_off; a.m.adj = c->d_adj; a.m.adjp = c->have_adjp ? c->d_adjp : nullptr; a.m.const_val = c->d_const;
  a.m.n_recs = P; a.m.n_vars = c->n_vars; a.m.n_slots = S; a.m.has_ternary = c->has_ternary; a.m.uniform_kind = c->uniform_kind; a.m.max_deg = c->max_deg;
  a.set_words = c->set_words; a.list_cap = cap; a.base = c->hull_lo;
  a.n_trees = st->n_trees; a.level_cap = st->level_capacity; a.trail_cap = st->trail_capacity; a.n_steps = n_steps; a.stop_on_solution = stop_on_solution;
  a.node_limit = node_limit;
  a.bits = st->bits; a.tree = st->tree; a.levels = reinterpret_cast<uint4*>(st->levels); a.trail = reinterpret_cast<uint4*>(st->trail);
  a.counters = reinterpret_cast<unsigned long long*>(st->counters); a.total_nodes = reinterpret_cast<unsigned long long*>(st->total_nodes);
  a.stop = st->stop; a.first_solution = st->first_solution; a.solution_flag = st->solution_flag; a.stats = c->d_stats;
  if (!n_steps) return PCP_OK;
  c->last_plan = pcp_plan{1u, 1u, 0u, 0u, 0u, 0u, 1u, 1u, st->n_trees, 1024u, (uint32_t)lds, cap, 0u};
  if (c->opt_time_kernels) HIP_TRY(c, hipEventRecord(c->ev_start, stream));
  HIP_TRY(c, launch_setdfs(a, stream));
  if (c->opt_time_kernels) HIP_TRY(c, hipEventRecord(c->ev_stop, stream));
  c->ev_valid = c->opt_time_kernels != 0;
  return PCP_OK;
}

int32_t pcp_dfs_forest_split_set(pcp_ctx* c, const pcp_forest_state* st, uint32_t n_pairs, const uint32_t* pairs, uint32_t* done, void* hip_stream) {
  if (!c || !st) return PCP_ERR_ARG;
  if (!c->set_words) return fail(c, PCP_ERR_ARG, "pcp_dfs_forest_split_set needs a set-mode model");
  if (!st->n_trees || !st->bits || !st->tree || !st->levels || !st->trail || (n_pairs && (!pairs || !done))) return fail(c, PCP_ERR_ARG, "null buffer");
  HIP_TRY(c, hipSetDevice(c->device));
  SetDfsArgs a;
  memset(&a, 0, sizeof(a));
  a.m.n_vars = c->n_vars; a.m.n_slots = c->n_slots;
  a.set_words = c->set_words; a.base = c->hull_lo;
  a.n_trees = st->n_trees; a.level_cap = st->level_capacity; a.trail_cap = st->trail_capacity;
  a.bits = st->bits; a.tree = st->tree; a.levels = reinterpret_cast<uint4*>(st->levels); a.trail = reinterpret_cast<uint4*>(st->trail);
  HIP_TRY(c, launch_setdfs_split(a, n_pairs, pairs, done, reinterpret_cast<hipStream_t>(hip_stream)));
  return PCP_OK;
}

// The search loop of an all-XNeqY model inside the kernel (neqfix_kernel<.., DFS>): workgroup t runs tree t, n_steps nodes per launch.
// Returns 1 when launched, 0 when this model / store cannot take the path, < 0 on error.
static int32_t launch_neq_dfs(pcp_ctx* c, const pcp_dfs_state* st, uint32_t n_trees, uint32_t n_steps, uint32_t stop_on_solution, uint64_t node_limit, hipStream_t stream) {
  if (!(c->neq_model && c->opt_neq_path && c->opt_neq_dfs)) return 0;
  const uint32_t S = c->n_slots, V = c->n_vars;
  // 32-bit cells: a one-node tile has LDS to spare, and the 16-bit-cell instantiation of the search loop needs 147 VGPRs — one
  // 512-thread workgroup per CU — where this one needs 115: two trees per CU in a forest, no scratch either way
  const bool packed = false;
  // A forest shares a CU among FOUR trees (the kernel's 127 VGPRs allow four wavefronts per SIMD = four 256-thread trees; the jump windows
  // get what is left of a quarter of the LDS), one tree alone keeps half a CU's LDS for its windows.  (Option "neq_dfs_wgs" overrides.)
  // Measured on N-queens-1000, first 2 M nodes (tools: PCP_SET_OPTIONS=... bench.py --mode search): 4 096 trees 2 x 256 threads per CU
  // (round 4) 4.3e7 nodes/s, 4 x 256 5.2e7, 8 x 128 5.6e7; 8 192 trees: 4 x 256 6.0e7, 8 x 128 7.2e7; 2 048 trees: 4 x 256 4.2e7, 8 x 128 3.5e7 —
  // a tree is a latency chain, so a CU wants as many of them as its registers allow, and narrower trees once there are enough of them.
  const bool many = n_trees >= 16u * (uint32_t)c->num_cu;
  const uint32_t wgs = c->opt_neq_dfs_wgs ? (uint32_t)c->opt_neq_dfs_wgs : (n_trees > 1 ? (many ? 8u : 4u) : 2u);
  const size_t lds = lds_bytes_neq(S, V, 1, packed, wgs);
  if (!lds || lds > c->lds_max || !n_steps) return 0;
  NeqArgs a;
  memset(&a, 0, sizeof(a));
  a.m.recs = c->d_recs; a.m.adj_off = c->d_adj_off; a.m.adj = c->d_adj; a.m.adjp = c->d_adjp; a.m.const_val = c->d_const;
  a.m.n_recs = (uint32_t)c->props.size(); a.m.n_vars = V; a.m.n_slots = S; a.m.uniform_kind = c->uniform_kind; a.m.max_deg = c->max_deg;
  a.seed_always = c->have_seed_always ? c->d_seed_always : nullptr;
  a.adjp4 = c->have_adjp4 ? c->d_adjp4 : nullptr;
  a.n_nodes = 1; a.nodes_per_block = 1; a.packed = packed ? 1u : 0u; a.violation = c->d_retry + 1; a.dbg = c->d_dbg; a.lds_wgs = wgs;
  a.lb_in = st->lb; a.ub_in = st->ub; a.lb_out = st->lb; a.ub_out = st->ub; a.status = st->status; a.stats = c->d_stats;
  a.dfs.sp = st->sp; a.dfs.stop = st->stop; a.dfs.counters = reinterpret_cast<unsigned long long*>(st->counters); a.dfs.first_solution = st->first_solution;
  a.dfs.dirty = c->opt_neq_hint ? st->dirty : nullptr;
  a.dfs.capacity = st->capacity; a.dfs.n_steps = n_steps; a.dfs.stop_on_solution = stop_on_solution; a.dfs.node_limit = node_limit;
  LaunchPlan plan;
  plan.grid = n_trees; plan.block = c->opt_neq_dfs_block ? (uint32_t)c->opt_neq_dfs_block : (n_trees > 1 ? (many ? 128u : 256u) : 512u); plan.lds_bytes = lds;
  c->last_plan = pcp_plan{1u, 1u, packed ? 1u : 0u, 0u, 0u, 0u, 1u, 0u, n_trees, plan.block, (uint32_t)plan.lds_bytes, S, 1u};
  if (c->opt_time_kernels) HIP_TRY(c, hipEventRecord(c->ev_start, stream));
  HIP_TRY(c, launch_neqfix(a, plan, stream));
  if (c->opt_time_kernels) HIP_TRY(c, hipEventRecord(c->ev_stop, stream));
  c->ev_valid = c->opt_time_kernels != 0;
  return 1;
}

int32_t pcp_dfs_forest_device(pcp_ctx* c, const pcp_dfs_state* st, uint32_t n_trees, uint32_t n_steps, uint32_t stop_on_solution, uint64_t node_limit, void* hip_stream) {
  if (!c || !st) return PCP_ERR_ARG;
  if (c->set_words) return fail(c, PCP_ERR_ARG, "pcp_dfs_forest_device runs interval-mode models (sets: pcp_dfs_forest_device_set)");
  if (!n_trees || !st->lb || !st->ub || !st->sp || !st->stop || !st->status || !st->counters || st->capacity < 2) return fail(c, PCP_ERR_ARG, "null buffer / no tree / capacity < 2");
  HIP_TRY(c, hipSetDevice(c->device));
  { const int32_t rcf = finalize_model(c); if (rcf) return rcf; }
  if (c->opt_force_path) return fail(c, PCP_ERR_UNSUPPORTED, "pcp_dfs_forest_device: force_path is set");
  const int32_t rc = launch_neq_dfs(c, st, n_trees, n_steps, stop_on_solution, node_limit, reinterpret_cast<hipStream_t>(hip_stream));
  if (rc < 0) return rc;
  if (rc == 0 && n_steps) return fail(c, PCP_ERR_UNSUPPORTED, "pcp_dfs_forest_device needs an all-XNeqY model whose variable store fits the in-kernel search (pcp_dfs_device runs any model, one tree)");
  return PCP_OK;
}

int32_t pcp_dfs_device(pcp_ctx* c, const pcp_dfs_state* st, uint32_t n_steps, uint32_t stop_on_solution, uint64_t node_limit, void* hip_stream) {
  if (!c || !st) return PCP_ERR_ARG;
  if (c->set_words) return fail(c, PCP_ERR_UNSUPPORTED, "pcp_dfs_device runs interval-mode models only");
  if (!st->lb || !st->ub || !st->sp || !st->stop || !st->status || !st->counters || st->capacity < 2) return fail(c, PCP_ERR_ARG, "null buffer / capacity < 2");
  // Plain launches on the caller's stream, three per step (team scratch memset, fixpoint, step).  Replaying the steps from captured
  // HIP graphs was built and measured: no faster (a step is bound by its kernels, ~55 us of fixpoint at this depth, not by the
  // host's enqueue rate) and not reliable across re-used buffers on this ROCm — dropped.
  const int64_t keep_path = c->opt_force_path;
  HIP_TRY(c, hipSetDevice(c->device));
  { const int32_t rcf = finalize_model(c); if (rcf) return rcf; }
  if (!keep_path) {
    // an all-XNeqY model: the whole loop — pop, propagate, count, branch — runs in ONE workgroup, n_steps nodes per launch
    // (neqfix_kernel<.., DFS>); a node is the lists of its assigned variables, there is no sweep to share out over the chip
    const int32_t rcn = launch_neq_dfs(c, st, 1u, n_steps, stop_on_solution, node_limit, reinterpret_cast<hipStream_t>(hip_stream));
    if (rcn < 0) return rcn;
    if (rcn == 1) return PCP_OK;
  }
  // one node per step: the team geometry unless the caller forced a path — or the model takes the assignment-driven kernel, which
  // has no sweep to share out (a node is one workgroup: its assigned variables' lists)
  if (!keep_path && !(c->neq_model && c->opt_neq_path)) c->opt_force_path = 2;
  c->dfs_sp = st->sp; c->dfs_stop = st->stop; c->dfs_team_words = 0;
  const int32_t rc = dfs_enqueue_steps(c, st, n_steps, stop_on_solution, node_limit, reinterpret_cast<hipStream_t>(hip_stream));
  c->dfs_sp = nullptr; c->dfs_stop = nullptr; c->dfs_team_words = 0;
  c->opt_force_path = keep_path;
  return rc;
}

int32_t pcp_pack_rows(pcp_ctx* c, uint32_t n_nodes, const int32_t* lb, const int32_t* ub, uint32_t* cells, void* hip_stream) {
  if (!c) return PCP_ERR_ARG;
  if (n_nodes && c->n_vars && (!lb || !ub || !cells)) return fail(c, PCP_ERR_ARG, "null pointer");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, launch_pack_rows(lb, ub, cells, (size_t)n_nodes * c->n_vars, c->d_retry + 1, reinterpret_cast<hipStream_t>(hip_stream)));
  return PCP_OK;
}

int32_t pcp_unpack_rows(pcp_ctx* c, uint32_t n_nodes, const uint32_t* cells, int32_t* lb, int32_t* ub, void* hip_stream) {
  if (!c) return PCP_ERR_ARG;
  if (n_nodes && c->n_vars && (!lb || !ub || !cells)) return fail(c, PCP_ERR_ARG, "null pointer");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, launch_unpack_rows(cells, lb, ub, (size_t)n_nodes * c->n_vars, reinterpret_cast<hipStream_t>(hip_stream)));
  return PCP_OK;
}

// ≡ Consistency::consistency for nodes whose cstores differ by a few propagators of their OWN (ABI v8): what Branch::distribute appends to one
// node (search/branching/branch.rs:36-55) and cannot be folded into its bounds — Enumerate's x != v (enumerate.rs:48-59).
int32_t pcp_propagate_device_units(pcp_ctx* c, uint32_t n_nodes, const pcp_device_batch* bt, const uint32_t* node_unit_off, const pcp_prop* node_units, void* hip_stream) {
  if (!c || !bt) return PCP_ERR_ARG;
  if (!node_unit_off) return pcp_propagate_device(c, n_nodes, bt, hip_stream);
  if (!node_units) return fail(c, PCP_ERR_ARG, "node_units must not be null when node_unit_off is given");
  if (bt->cell_format) return fail(c, PCP_ERR_UNSUPPORTED, "node units: int32 rows only");
  c->cur_nu_off = node_unit_off; c->cur_nu = node_units;
  const int32_t rc = pcp_propagate_device(c, n_nodes, bt, hip_stream);
  c->cur_nu_off = nullptr; c->cur_nu = nullptr;
  return rc;
}

int32_t pcp_branch_device(pcp_ctx* c, uint32_t n_nodes, const int32_t* lb, const int32_t* ub, const uint64_t* active,
                          const uint8_t* status, int32_t* child_lb, int32_t* child_ub, uint64_t* child_active,
                          uint32_t* counts, void* hip_stream) {
  return pcp_branch_device_hint(c, n_nodes, lb, ub, active, status, child_lb, child_ub, child_active, nullptr, counts, hip_stream);
}

int32_t pcp_branch_device_hint(pcp_ctx* c, uint32_t n_nodes, const int32_t* lb, const int32_t* ub, const uint64_t* active,
                               const uint8_t* status, int32_t* child_lb, int32_t* child_ub, uint64_t* child_active,
                               uint32_t* child_dirty, uint32_t* counts, void* hip_stream) {
  if (!c || !counts) return PCP_ERR_ARG;
  hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
  HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t words = (c->n_units + 63) / 64;
  // active == child_active == NULL: implicit-active nodes (domains only; liveness is derived by pcp_propagate_device)
  if (n_nodes && (!status || (c->n_vars && (!lb || !ub || !child_lb || !child_ub)) || (words && ((active == nullptr) != (child_active == nullptr)))))
    return fail(c, PCP_ERR_ARG, "null buffer");
  int32_t rc = ensure(c, c->d_child_base, c->cap_child_base, std::max<uint32_t>(n_nodes, 1));
  if (rc) return rc;
  if (n_nodes == 0) { HIP_TRY(c, hipMemsetAsync(counts, 0, 20, stream)); return PCP_OK; }
  HIP_TRY(c, launch_branch(n_nodes, c->n_vars, active ? words : 0u, lb, ub, active, status, child_lb, child_ub, child_active, child_dirty, c->d_child_base, counts, (uint32_t)c->opt_branch_reverse, stream));
  return PCP_OK;
}

int32_t pcp_branch_device_cells(pcp_ctx* c, uint32_t n_nodes, const uint32_t* cells, const uint8_t* status, uint32_t* child_cells, uint32_t* child_dirty,
                                uint32_t* counts, void* hip_stream) {
  if (!c || !counts) return PCP_ERR_ARG;
  if (c->set_words) return fail(c, PCP_ERR_UNSUPPORTED, "pcp_branch_device_cells: interval mode only");
  hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
  HIP_TRY(c, hipSetDevice(c->device));
  if (n_nodes && (!status || (c->n_vars && (!cells || !child_cells)))) return fail(c, PCP_ERR_ARG, "null buffer");
  int32_t rc = ensure(c, c->d_child_base, c->cap_child_base, std::max<uint32_t>(n_nodes, 1));
  if (rc) return rc;
  if (n_nodes == 0) { HIP_TRY(c, hipMemsetAsync(counts, 0, 20, stream)); return PCP_OK; }
  HIP_TRY(c, launch_branch_scan(n_nodes, status, c->d_child_base, counts, stream));
  HIP_TRY(c, launch_branch_cells(n_nodes, c->n_vars, cells, c->d_child_base, child_cells, child_dirty, counts, (uint32_t)c->opt_branch_reverse, stream));
  return PCP_OK;
}

int32_t pcp_branch_device_set(pcp_ctx* c, uint32_t n_nodes, const uint64_t* bits, const int32_t* lb, const int32_t* ub, const uint64_t* active,
                              const uint8_t* status, uint64_t* child_bits, uint64_t* child_active, uint32_t* counts, void* hip_stream) {
  if (!c || !counts) return PCP_ERR_ARG;
  if (!c->set_words) return fail(c, PCP_ERR_ARG, "pcp_branch_device_set needs a set-mode model (pcp_model_reset with set_words > 0)");
  if (!c->hull_set) return fail(c, PCP_ERR_CONTRACT, "set mode needs the hull of the initial domains (pcp_model_set_hull)");
  hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
  HIP_TRY(c, hipSetDevice(c->device));
  const uint32_t words = (c->n_units + 63) / 64;
  if (n_nodes && (!status || (c->n_vars && (!bits || !lb || !ub || !child_bits)) || (words && ((active == nullptr) != (child_active == nullptr)))))
    return fail(c, PCP_ERR_ARG, "null buffer");
  int32_t rc = ensure(c, c->d_child_base, c->cap_child_base, std::max<uint32_t>(n_nodes, 1));
  if (rc) return rc;
  if (n_nodes == 0) { HIP_TRY(c, hipMemsetAsync(counts, 0, 20, stream)); return PCP_OK; }
  HIP_TRY(c, launch_set_branch(n_nodes, c->n_vars, c->set_words, c->hull_lo, active ? words : 0u, bits, lb, ub, active, status, child_bits, child_active,
                               c->d_child_base, counts, (uint32_t)c->opt_branch_reverse, stream));
  return PCP_OK;
}

int32_t pcp_stats_reset(pcp_ctx* c, void* hip_stream) {
  if (!c) return PCP_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemsetAsync(c->d_stats, 0, kStatSlots * sizeof(pcp_stats), reinterpret_cast<hipStream_t>(hip_stream)));
  HIP_TRY(c, hipMemsetAsync(c->d_retry + 1, 0, 4, reinterpret_cast<hipStream_t>(hip_stream)));  // and the sticky hull-violation word
  HIP_TRY(c, hipMemsetAsync(c->d_dbg, 0, kStatSlots * PCP_DBG_COUNT * 8, reinterpret_cast<hipStream_t>(hip_stream)));
  return PCP_OK;
}

int32_t pcp_debug_counters(pcp_ctx* c, uint64_t* out, uint32_t n, void* hip_stream) {
  if (!c || !out || n > PCP_DBG_COUNT) return PCP_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
  // striped like the counters of pcp_stats (workgroup b adds to stripe b % kStatSlots); slots 5..7 are maxima (a launch's timeline)
  unsigned long long slots[kStatSlots][PCP_DBG_COUNT];
  HIP_TRY(c, hipMemcpyAsync(slots, c->d_dbg, sizeof(slots), hipMemcpyDeviceToHost, stream));
  HIP_TRY(c, hipStreamSynchronize(stream));
  for (uint32_t i = 0; i < n; ++i) {
    uint64_t v = 0;
    for (uint32_t s = 0; s < kStatSlots; ++s) v = (i >= 5 && i <= 7) ? std::max<uint64_t>(v, slots[s][i]) : v + slots[s][i];
    out[i] = v;
  }
  return PCP_OK;
}

// The counters, and whether the sticky hull-violation word is raised (left as it is: only pcp_stats_read consumes it).
static int32_t read_counters(pcp_ctx* c, pcp_stats* out, uint32_t* flag, hipStream_t stream) {
  pcp_stats slots[kStatSlots];
  HIP_TRY(c, hipMemcpyAsync(slots, c->d_stats, sizeof(slots), hipMemcpyDeviceToHost, stream));
  HIP_TRY(c, hipMemcpyAsync(flag, c->d_retry + 1, 4, hipMemcpyDeviceToHost, stream));
  HIP_TRY(c, hipStreamSynchronize(stream));
  static_assert(sizeof(pcp_stats) % sizeof(uint64_t) == 0, "pcp_stats is a struct of u64 counters");
  memset(out, 0, sizeof(*out));
  for (uint32_t s = 0; s < kStatSlots; ++s)
    for (size_t i = 0; i < sizeof(pcp_stats) / sizeof(uint64_t); ++i)
      reinterpret_cast<uint64_t*>(out)[i] += reinterpret_cast<const uint64_t*>(&slots[s])[i];
  return PCP_OK;
}

int32_t pcp_stats_read(pcp_ctx* c, pcp_stats* out, void* hip_stream) {
  if (!c || !out) return PCP_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  uint32_t flag = 0;
  const int32_t rcr = read_counters(c, out, &flag, reinterpret_cast<hipStream_t>(hip_stream));
  if (rcr) return rcr;
  if (flag) {  // some launch since the last read met a node outside the declared hull
    HIP_TRY(c, hipMemsetAsync(c->d_retry + 1, 0, 4, reinterpret_cast<hipStream_t>(hip_stream)));
    c->trusted_epoch = 0;
    return fail(c, PCP_ERR_CONTRACT, "a node's bounds lie outside the hull declared with pcp_model_set_hull, or beyond +-(2^29 - 1) (status PCP_STATUS_HULL)");
  }
  return PCP_OK;
}

int32_t pcp_last_plan(const pcp_ctx* c, pcp_plan* out) {
  if (!c || !out) return PCP_ERR_ARG;
  *out = c->last_plan;
  return PCP_OK;
}

int32_t pcp_last_kernel_ms(pcp_ctx* c, float* ms) {
  if (!c || !ms) return PCP_ERR_ARG;
  if (!c->ev_valid) return fail(c, PCP_ERR_ARG, "no timed launch (option time_kernels = 0, or nothing was launched yet)");
  HIP_TRY(c, hipEventSynchronize(c->ev_stop));
  HIP_TRY(c, hipEventElapsedTime(ms, c->ev_start, c->ev_stop));
  return PCP_OK;
}

int32_t pcp_propagate(pcp_ctx* c, uint32_t n_nodes, int32_t* lb, int32_t* ub, uint64_t* bits, uint64_t* active,
                      uint8_t* status, pcp_stats* stats) {
  if (!c) return PCP_ERR_ARG;
  if ((bits != nullptr) != (c->set_words != 0)) return fail(c, PCP_ERR_ARG, "`bits` must be given exactly when the model was reset with set_words > 0");
  if (n_nodes && (!status || (c->n_vars && (!lb || !ub)))) return fail(c, PCP_ERR_ARG, "null buffer");
  HIP_TRY(c, hipSetDevice(c->device));
  int32_t rc = finalize_model(c);
  if (rc) return rc;
  if (stats) memset(stats, 0, sizeof(*stats));
  if (n_nodes == 0) return PCP_OK;
  // The reference panics on an empty initial domain (variable/store.rs:136) and the build rejects bounds
  // outside +-PCP_BOUND_MAX.
  const size_t nv = (size_t)n_nodes * c->n_vars;
  const size_t sw = c->set_words;
  if (sw) {
    // set mode: the domains are the bitsets; an empty initial domain is the reference's alloc panic (variable/store.rs:136)
    for (size_t i = 0; i < nv; ++i) {
      bool any = false;
      for (size_t k = 0; k < sw && !any; ++k) any = bits[i * sw + k] != 0;
      if (!any) return fail(c, PCP_ERR_CONTRACT, "empty initial domain (variable/store.rs:136)");
    }
  }
  for (size_t i = 0; i < nv && !sw; ++i) {
    if (lb[i] > ub[i]) return fail(c, PCP_ERR_CONTRACT, "empty initial domain (variable/store.rs:136)");
    if (lb[i] < -PCP_BOUND_MAX || ub[i] > PCP_BOUND_MAX) return fail(c, PCP_ERR_CONTRACT, "bound outside +-PCP_BOUND_MAX");
    if (c->hull_set && (lb[i] < c->hull_lo || ub[i] > c->hull_hi)) return fail(c, PCP_ERR_CONTRACT, "bound outside the hull declared with pcp_model_set_hull");
  }
  const uint32_t words = ((uint32_t)c->n_units + 63) / 64;
  if (active && words && (c->n_units & 63)) {
    // bits at or above n_units name no unit: they would count as "still active" and keep a node from ever being True
    const uint64_t tail = ~0ull << (c->n_units & 63);
    for (uint32_t n = 0; n < n_nodes; ++n)
      if (active[(size_t)n * words + words - 1] & tail) return fail(c, PCP_ERR_ARG, "`active` row has bits set at or above n_units");
  }
  const size_t dom_bytes = nv * 4, act_bytes = (size_t)n_nodes * words * 8;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t bits_bytes = nv * sw * 8;
  const size_t o_lb = 0, o_ub = up(o_lb + dom_bytes), o_act = up(o_ub + dom_bytes), o_st = up(o_act + act_bytes), o_bits = up(o_st + n_nodes), total = up(o_bits + bits_bytes);
  {
    unsigned char* p = static_cast<unsigned char*>(c->d_stage);
    size_t cap = c->cap_stage;
    if ((rc = ensure(c, p, cap, total))) return rc;
    c->d_stage = p; c->cap_stage = cap;
  }
  unsigned char* base = static_cast<unsigned char*>(c->d_stage);
  hipStream_t stream = nullptr;
  if (dom_bytes && !sw) {
    HIP_TRY(c, hipMemcpyAsync(base + o_lb, lb, dom_bytes, hipMemcpyHostToDevice, stream));
    HIP_TRY(c, hipMemcpyAsync(base + o_ub, ub, dom_bytes, hipMemcpyHostToDevice, stream));
  }
  if (bits_bytes) HIP_TRY(c, hipMemcpyAsync(base + o_bits, bits, bits_bytes, hipMemcpyHostToDevice, stream));
  if (active && act_bytes) HIP_TRY(c, hipMemcpyAsync(base + o_act, active, act_bytes, hipMemcpyHostToDevice, stream));
  pcp_stats before;
  uint32_t flag_before = 0;  // (a violation raised by an EARLIER device launch stays pending for its owner's pcp_stats_read)
  if (stats) { rc = read_counters(c, &before, &flag_before, stream); if (rc) return rc; }
  pcp_device_batch bt;
  memset(&bt, 0, sizeof(bt));
  if (sw) { bt.bits_in = reinterpret_cast<uint64_t*>(base + o_bits); bt.bits_out = reinterpret_cast<uint64_t*>(base + o_bits); }
  bt.lb_in = reinterpret_cast<int32_t*>(base + o_lb); bt.ub_in = reinterpret_cast<int32_t*>(base + o_ub);
  bt.lb_out = reinterpret_cast<int32_t*>(base + o_lb); bt.ub_out = reinterpret_cast<int32_t*>(base + o_ub);
  bt.active_in = active ? reinterpret_cast<uint64_t*>(base + o_act) : nullptr;
  bt.active_out = active ? reinterpret_cast<uint64_t*>(base + o_act) : nullptr;
  bt.status = base + o_st;
  rc = pcp_propagate_device(c, n_nodes, &bt, stream);
  if (rc) return rc;
  if (dom_bytes) {
    HIP_TRY(c, hipMemcpyAsync(lb, base + o_lb, dom_bytes, hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipMemcpyAsync(ub, base + o_ub, dom_bytes, hipMemcpyDeviceToHost, stream));
  }
  if (active && act_bytes) HIP_TRY(c, hipMemcpyAsync(active, base + o_act, act_bytes, hipMemcpyDeviceToHost, stream));
  if (bits_bytes) HIP_TRY(c, hipMemcpyAsync(bits, base + o_bits, bits_bytes, hipMemcpyDeviceToHost, stream));
  HIP_TRY(c, hipMemcpyAsync(status, base + o_st, n_nodes, hipMemcpyDeviceToHost, stream));
  HIP_TRY(c, hipStreamSynchronize(stream));
  if (stats) {
    pcp_stats after;
    uint32_t flag_after = 0;
    rc = read_counters(c, &after, &flag_after, stream);
    if (rc) return rc;
    stats->steps = after.steps - before.steps; stats->steps3 = after.steps3 - before.steps3;
    stats->narrowings = after.narrowings - before.narrowings; stats->waves = after.waves - before.waves;
    stats->failed_nodes = after.failed_nodes - before.failed_nodes; stats->nodes = after.nodes - before.nodes;
    stats->evaluated = after.evaluated - before.evaluated; stats->full_evals = after.full_evals - before.full_evals;
  }
  return PCP_OK;
}

}  // extern "C"
