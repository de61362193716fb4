// pcp_host.hpp — C++17 host side above the C ABI (include/pcp_hip.h): a mirror of libpcp's model-building and
// search surface for the propagation path, so that code written against the reference reads the same here.
//
// The reference is Rust (compiled code); Rust is not available where this repository is built, so the host side is
// C++ (header-only, links against libpcp_hip.so).  It mirrors, with the reference's names and argument meaning
// (paths relative to /root/reference/src/libpcp):
//   Interval                      — crate intervallum, as used by variable/store.rs
//   VStore::alloc / operator[]    — variable/store.rs:129-141, 168-182            (VStoreFD, Interval<i32> domains)
//   Identity / Addition / Constant — term/identity.rs:47-70, term/addition.rs:80-110, term/constant.rs:43-68
//   XNeqY XEqY XLessY XLessYPlusZ XGreaterYPlusZ XEqYPlusZ XEqYMulZ, x_greater_y x_geq_y x_leq_y
//   x_geq_y_plus_z x_leq_y_plus_z — propagators/cmp/*.rs, propagators/cmp/mod.rs:34-86
//   Distinct / join_distinct      — propagators/distinct.rs:26-126
//   AllEqual                      — propagators/all_equal.rs:47-64
//   GpuCStore::alloc / consistency / label / restore — propagation/store.rs:223-230, 247-257, 306-324
//   Space::consistency            — search/space.rs:41-43
//   one_solution / all_solutions with FirstSmallestVar, MiddleVal, BinarySplit, StopNode
//                                 — search/mod.rs:45-52, search/branching/*.rs, search/engine/*.rs, search/stop_node.rs
// Contract violations the reference reports with panic!/assert! are thrown as pcp_host::Panic.
// There is no CPU propagation path in this header: consistency() always goes through pcp_propagate().
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pcp_hip.h"

namespace pcp_host {

struct Panic : std::runtime_error {
  using std::runtime_error::runtime_error;
};

enum class SKleene : uint8_t { False = 0, True = 1, Unknown = 2 };

struct Interval {
  int32_t lb, ub;
  Interval(int32_t l, int32_t u) : lb(l), ub(u) {}
  bool is_empty() const { return lb > ub; }
  bool is_singleton() const { return lb == ub; }
  int32_t lower() const { return lb; }
  int32_t upper() const { return ub; }
  uint32_t size() const { return is_empty() ? 0u : (uint32_t)((int64_t)ub - lb + 1); }
};

// ---- views -------------------------------------------------------------------------------------------------
struct Operand {  // a flattened view: Identity / Addition chain over a variable, or a Constant
  uint32_t var;   // variable index or PCP_CONST
  int32_t off;    // Addition offset, or the constant's value
};
struct View {
  virtual ~View() = default;
  virtual Operand flat() const = 0;
};
using Var = std::shared_ptr<const View>;
struct Identity final : View {
  size_t idx;
  explicit Identity(size_t i) : idx(i) {}
  size_t index() const { return idx; }
  Operand flat() const override { return {(uint32_t)idx, 0}; }
};
struct Addition final : View {  // term/addition.rs: read = x + v, update(d) = x.update(d - v)
  Var x;
  int32_t v;
  Addition(Var x_, int32_t v_) : x(std::move(x_)), v(v_) {}
  Operand flat() const override { Operand o = x->flat(); o.off += v; return o; }  // Addition(Constant c, v) == Constant(c+v)
};
struct Constant final : View {
  int32_t value;
  explicit Constant(int32_t c) : value(c) {}
  Operand flat() const override { return {PCP_CONST, value}; }
};
inline Var identity(size_t i) { return std::make_shared<Identity>(i); }
inline Var addition(Var x, int32_t v) { return std::make_shared<Addition>(std::move(x), v); }
inline Var constant(int32_t c) { return std::make_shared<Constant>(c); }

// ---- variable store ------------------------------------------------------------------------------------------
class VStore {
 public:
  Var alloc(Interval dom) {  // variable/store.rs:129-141
    if (dom.is_empty()) throw Panic("alloc: empty domain");
    if (dom.lb < -PCP_BOUND_MAX || dom.ub > PCP_BOUND_MAX) throw Panic("bound outside +-PCP_BOUND_MAX");
    lb_.push_back(dom.lb);
    ub_.push_back(dom.ub);
    return identity(lb_.size() - 1);
  }
  size_t size() const { return lb_.size(); }
  Interval operator[](size_t i) const {  // variable/store.rs:168-182
    if (i >= lb_.size()) throw Panic("Variable not registered in the store.");
    return Interval(lb_[i], ub_[i]);
  }
  bool update(size_t i, Interval d) {  // variable/store.rs:151-166 (monotone; empty => false, store untouched)
    Interval cur = (*this)[i];
    if (!(d.is_empty() || (d.lb >= cur.lb && d.ub <= cur.ub))) throw Panic("Domain update must be monotonic.");
    if (d.is_empty()) return false;
    lb_[i] = d.lb;
    ub_[i] = d.ub;
    return true;
  }
  std::vector<int32_t>& lbs() { return lb_; }
  std::vector<int32_t>& ubs() { return ub_; }
  const std::vector<int32_t>& lbs() const { return lb_; }
  const std::vector<int32_t>& ubs() const { return ub_; }

 private:
  std::vector<int32_t> lb_, ub_;
};

// ---- propagators ----------------------------------------------------------------------------------------------
struct Propagator {  // one unit of the constraint store: one elementary filter or a Conjunction/Distinct of them
  std::vector<pcp_prop> rows;
};
inline pcp_prop make_row(pcp_kind k, std::initializer_list<Var> ops) {
  pcp_prop p{};
  p.kind = (uint8_t)k;
  for (int i = 0; i < 3; ++i) { p.var[i] = PCP_NOVAR; p.off[i] = 0; }
  int i = 0;
  for (const Var& v : ops) { Operand o = v->flat(); p.var[i] = o.var; p.off[i] = o.off; ++i; }
  return p;
}
inline Propagator XNeqY(Var x, Var y) { return {{make_row(PCP_NEQ, {x, y})}}; }
inline Propagator XEqY(Var x, Var y) { return {{make_row(PCP_EQ, {x, y})}}; }
inline Propagator XLessY(Var x, Var y) { return {{make_row(PCP_LT, {x, y})}}; }
inline Propagator XLessYPlusZ(Var x, Var y, Var z) { return {{make_row(PCP_LT3, {x, y, z})}}; }
inline Propagator XGreaterYPlusZ(Var x, Var y, Var z) { return {{make_row(PCP_GT3, {x, y, z})}}; }
inline Propagator XEqYPlusZ(Var x, Var y, Var z) { return {{make_row(PCP_EQ3, {x, y, z})}}; }
inline Propagator XEqYMulZ(Var x, Var y, Var z) { return {{make_row(PCP_MUL3, {x, y, z})}}; }
// propagators/cmp/mod.rs:34-86
inline Propagator x_greater_y(Var x, Var y) { return XLessY(y, x); }
inline Propagator x_geq_y(Var x, Var y) { return x_greater_y(addition(x, 1), y); }
inline Propagator x_leq_y(Var x, Var y) { return XLessY(x, addition(y, 1)); }
inline Propagator x_geq_y_plus_z(Var x, Var y, Var z) { return XGreaterYPlusZ(addition(x, 1), y, z); }
inline Propagator x_leq_y_plus_z(Var x, Var y, Var z) { return XLessYPlusZ(addition(x, -1), y, z); }
inline Propagator AllEqual(const std::vector<Var>& vars) {  // propagators/all_equal.rs:47-64: XEqY(v[i], v[i+1]) as ONE unit
  if (vars.empty()) throw Panic("Variable array in `AllEqual` must be non-empty.");
  Propagator p;
  for (size_t i = 0; i + 1 < vars.size(); ++i) {
    pcp_prop r = make_row(PCP_EQ, {vars[i], vars[i + 1]});
    r.group_kind = 2;
    p.rows.push_back(r);
  }
  if (p.rows.empty()) {  // one variable: an empty conjunction, entailed at its first evaluation
    pcp_prop r = make_row(PCP_NEQ, {constant(0), constant(1)});
    r.group_kind = 2;
    p.rows.push_back(r);
  }
  return p;
}
inline Propagator Distinct(const std::vector<Var>& vars) {  // propagators/distinct.rs:63-83: ONE unit
  if (vars.empty()) throw Panic("Variable array in `Distinct` must be non-empty.");
  Propagator p;
  for (size_t i = 0; i + 1 < vars.size(); ++i)
    for (size_t j = i + 1; j < vars.size(); ++j) {
      pcp_prop r = make_row(PCP_NEQ, {vars[i], vars[j]});
      r.group_kind = 2;
      p.rows.push_back(r);
    }
  if (p.rows.empty()) {  // Distinct over one variable: an empty conjunction, entailed at its first evaluation
    pcp_prop r = make_row(PCP_NEQ, {constant(0), constant(1)});
    r.group_kind = 2;
    p.rows.push_back(r);
  }
  return p;
}

// ---- constraint store on the GPU --------------------------------------------------------------------------------
// IntCStore-shaped (concept.rs:120-138): alloc, consistency, label/restore.  Branch constraints of the search —
// a variable against a Constant through XLessY/XEqY — narrow their one variable on their first run and are then
// entailed and unlinked (x_less_y.rs:87-109, store.rs:171), so they are applied to the vstore at the next
// consistency() instead of growing the device model on every search node (SURVEY.md §8b "per-node propagators").
class GpuCStore {
 public:
  explicit GpuCStore(int hip_device = 0) {
    int32_t rc = pcp_ctx_create(hip_device, &ctx_);
    if (rc != PCP_OK) throw std::runtime_error(std::string("pcp_ctx_create: ") + pcp_strerror(rc) + " (there is no CPU path)");
  }
  ~GpuCStore() { pcp_ctx_destroy(ctx_); }
  GpuCStore(const GpuCStore&) = delete;
  GpuCStore& operator=(const GpuCStore&) = delete;

  size_t alloc(Propagator p) {  // Store::alloc, propagation/store.rs:223-230
    const size_t idx = units_.size();
    const uint32_t gid = (uint32_t)idx;
    for (auto& r : p.rows) r.group = gid;
    units_.push_back(std::move(p));
    set_bit(active_, idx, true);
    return idx;
  }
  size_t size() const { return units_.size(); }

  using Label = std::pair<size_t, std::vector<uint64_t>>;  // (propagators.len(), active.clone()), store.rs:315-317
  Label label() const { return {units_.size(), active_}; }
  void restore(const Label& l) {  // store.rs:319-323
    units_.resize(l.first);
    active_ = l.second;
    // device-side units that no longer exist on the host: a unit allocated later at the same index is a DIFFERENT
    // propagator, so the device model is truncated with the host's (sync_model compares unit indices only)
    size_t keep = 0;
    while (keep < dev_units_.size() && dev_units_[keep] < l.first) ++keep;
    if (keep < dev_units_.size()) {
      check(pcp_model_truncate(ctx_, (uint32_t)keep));
      dev_units_.resize(keep);
    }
  }

  SKleene consistency(VStore& vs) {  // Consistency::consistency, propagation/store.rs:247-257
    // 1. fold active var-vs-constant units into the vstore (they would be entailed right after their first run)
    for (size_t u = 0; u < units_.size(); ++u) {
      if (!get_bit(active_, u) || !is_unary(units_[u])) continue;
      if (!apply_unary(units_[u].rows[0], vs)) return SKleene::False;  // store.rs:155-158
      set_bit(active_, u, false);
    }
    // 2. device model = the non-unary units, in order; sync it when it changed (append-only or truncated)
    sync_model(vs.size());
    // 3. one node through the engine
    std::vector<uint64_t> act((dev_units_.size() + 63) / 64, 0);
    for (size_t k = 0; k < dev_units_.size(); ++k)
      if (get_bit(active_, dev_units_[k])) act[k >> 6] |= 1ull << (k & 63);
    uint8_t status = 0;
    int32_t rc = pcp_propagate(ctx_, 1, vs.lbs().data(), vs.ubs().data(), nullptr, act.empty() ? nullptr : act.data(), &status, &last_stats_);
    if (rc == PCP_ERR_CONTRACT) throw Panic(pcp_last_error(ctx_));
    if (rc != PCP_OK) throw std::runtime_error(std::string("pcp_propagate: ") + pcp_last_error(ctx_));
    if (status != PCP_FALSE)
      for (size_t k = 0; k < dev_units_.size(); ++k) set_bit(active_, dev_units_[k], (act[k >> 6] >> (k & 63)) & 1);
    return (SKleene)status;
  }
  const pcp_stats& last_stats() const { return last_stats_; }
  pcp_ctx* ctx() { return ctx_; }

 private:
  static bool get_bit(const std::vector<uint64_t>& b, size_t i) { return (i >> 6) < b.size() && ((b[i >> 6] >> (i & 63)) & 1); }
  static void set_bit(std::vector<uint64_t>& b, size_t i, bool v) {
    if ((i >> 6) >= b.size()) b.resize((i >> 6) + 1, 0);
    if (v) b[i >> 6] |= 1ull << (i & 63); else b[i >> 6] &= ~(1ull << (i & 63));
  }
  static bool is_unary(const Propagator& p) {
    if (p.rows.size() != 1) return false;
    const pcp_prop& r = p.rows[0];
    if (r.kind != PCP_LT && r.kind != PCP_EQ) return false;
    return (r.var[0] == PCP_CONST) != (r.var[1] == PCP_CONST);
  }
  // XLessY / XEqY between a variable view and a Constant: exactly the narrowing of one propagate()
  // (x_less_y.rs:104-109, x_eq_y.rs:102-107); false when the update would be empty.
  static bool apply_unary(const pcp_prop& r, VStore& vs) {
    const bool x_is_var = r.var[0] != PCP_CONST;
    const uint32_t v = x_is_var ? r.var[0] : r.var[1];
    const int64_t voff = x_is_var ? r.off[0] : r.off[1];
    const int64_t c = x_is_var ? r.off[1] : r.off[0];
    Interval d = vs[v];
    int64_t lb = d.lb, ub = d.ub;
    if (r.kind == PCP_EQ) { lb = std::max<int64_t>(lb, c - voff); ub = std::min<int64_t>(ub, c - voff); }
    else if (x_is_var) ub = std::min<int64_t>(ub, c - 1 - voff);  // x + voff < c
    else lb = std::max<int64_t>(lb, c + 1 - voff);                  // c < y + voff
    if (lb > ub) return false;
    return vs.update(v, Interval((int32_t)lb, (int32_t)ub));
  }
  void sync_model(size_t n_vars) {
    std::vector<size_t> want;
    for (size_t u = 0; u < units_.size(); ++u)
      if (!is_unary(units_[u])) want.push_back(u);
    size_t common = 0;
    while (common < want.size() && common < dev_units_.size() && want[common] == dev_units_[common]) ++common;
    if (n_vars != dev_vars_) {
      check(pcp_model_reset(ctx_, (uint32_t)n_vars, 0));
      dev_vars_ = n_vars;
      common = 0;
    } else if (common < dev_units_.size()) {
      check(pcp_model_truncate(ctx_, (uint32_t)common));
    }
    dev_units_.resize(common);
    std::vector<pcp_prop> rows;
    for (size_t k = common; k < want.size(); ++k) {
      for (const pcp_prop& r : units_[want[k]].rows) rows.push_back(r);
      dev_units_.push_back(want[k]);
    }
    if (!rows.empty()) check(pcp_model_push_props(ctx_, (uint32_t)rows.size(), rows.data()));
  }
  void check(int32_t rc) {
    if (rc == PCP_ERR_CONTRACT) throw Panic(pcp_last_error(ctx_));
    if (rc != PCP_OK) throw std::runtime_error(pcp_last_error(ctx_));
  }

  pcp_ctx* ctx_ = nullptr;
  std::vector<Propagator> units_;
  std::vector<uint64_t> active_;
  std::vector<size_t> dev_units_;  // unit index of each device-side unit
  size_t dev_vars_ = (size_t)-1;
  pcp_stats last_stats_{};
};

inline void join_distinct(VStore&, GpuCStore& cstore, const std::vector<Var>& vars) {  // propagators/distinct.rs:26-45
  if (vars.empty()) throw Panic("Variable array in `Distinct` must be non-empty.");
  for (size_t i = 0; i + 1 < vars.size(); ++i)
    for (size_t j = i + 1; j < vars.size(); ++j) cstore.alloc(XNeqY(vars[i], vars[j]));
}

// ---- space and search ----------------------------------------------------------------------------------------------
struct Space {
  VStore vstore;
  GpuCStore cstore;
  explicit Space(int hip_device = 0) : cstore(hip_device) {}
  SKleene consistency() { return cstore.consistency(vstore); }  // search/space.rs:41-43
};

enum class Status { Satisfiable, Unsatisfiable, EndOfSearch };
struct Statistics { uint64_t num_solution = 0, num_failed_node = 0, num_nodes = 0; };  // search/statistics.rs:19-24

inline size_t first_smallest_var(const VStore& vs) {  // first_smallest_var.rs:30-39
  size_t best = SIZE_MAX;
  uint32_t bs = 0;
  for (size_t i = 0; i < vs.size(); ++i) {
    uint32_t s = vs[i].size();
    if (s > 1 && (best == SIZE_MAX || s < bs)) { best = i; bs = s; }
  }
  if (best == SIZE_MAX) throw Panic("Cannot select a variable in a space where all variables are assigned.");
  return best;
}
inline int32_t middle_val(Interval d) { return (int32_t)(((int64_t)d.lower() + d.upper()) / 2); }  // middle_val.rs:25-27

// OneSolution<Propagation<Brancher<FirstSmallestVar, MiddleVal, BinarySplit>>, VectorStack> (search/mod.rs:45-52),
// optionally under AllSolution and StopNode(node_limit).  on_solution is called with the space at every solution.
inline Status search(Space& space, bool all_solutions, uint64_t node_limit, Statistics& st,
                     const std::function<void(const Space&)>& on_solution = nullptr) {
  struct Branch { std::vector<int32_t> lb, ub; GpuCStore::Label clabel; size_t var; int32_t val; bool left; };
  std::vector<Branch> stack;  // VectorStack: LIFO
  bool first = true, found = false;
  for (;;) {
    if (!first) {
      if (stack.empty()) break;
      Branch b = std::move(stack.back());
      stack.pop_back();
      space.vstore.lbs() = b.lb;  // Branch::commit (branch.rs:51-55): restore the labels, then add the branch propagator
      space.vstore.ubs() = b.ub;
      space.cstore.restore(b.clabel);
      if (b.left) space.cstore.alloc(x_leq_y(identity(b.var), constant(b.val)));       // binary_split.rs:46-51
      else space.cstore.alloc(x_greater_y(identity(b.var), constant(b.val)));          // binary_split.rs:52-57
    }
    first = false;
    SKleene k = space.consistency();  // Propagation::enter, search/propagation.rs:49
    ++st.num_nodes;
    if (node_limit && st.num_nodes >= node_limit) return Status::EndOfSearch;  // stop_node.rs:57-62
    if (k == SKleene::True) {
      ++st.num_solution;
      found = true;
      if (on_solution) on_solution(space);
      if (!all_solutions) return Status::Satisfiable;
    } else if (k == SKleene::False) {
      ++st.num_failed_node;
    } else {
      const size_t var = first_smallest_var(space.vstore);
      const int32_t val = middle_val(space.vstore[var]);
      GpuCStore::Label cl = space.cstore.label();
      stack.push_back({space.vstore.lbs(), space.vstore.ubs(), cl, var, val, false});  // reversed: left explored first
      stack.push_back({space.vstore.lbs(), space.vstore.ubs(), cl, var, val, true});
    }
  }
  if (all_solutions) return Status::EndOfSearch;
  return found ? Status::Satisfiable : Status::Unsatisfiable;
}

}  // namespace pcp_host
