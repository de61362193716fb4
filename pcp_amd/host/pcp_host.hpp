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
//   Sum                           — term/sum.rs:56-92
//   Boolean / BooleanNeg / Conjunction / Disjunction / not_ / implication / equivalence
//                                 — logic/boolean.rs:111-140, boolean_neg.rs:71-96, conjunction.rs:70-119, disjunction.rs:68-141,
//                                   logic/ops.rs:17-19, logic/mod.rs:30-45   (formula units: pcp_model_push_formula)
//   Cumulative::join              — propagators/cumulative.rs:59-114
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
#include <map>
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
struct Operand {  // a flattened view: Identity / Addition chain over a variable, a Constant, or a Sum
  uint32_t var;   // variable index, PCP_CONST, or PCP_SUM (the members below)
  int32_t off;    // Addition offset, or the constant's value
  std::vector<uint32_t> members;  // var == PCP_SUM: the variables of the Sum
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
  Operand flat() const override { return {(uint32_t)idx, 0, {}}; }
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
  Operand flat() const override { return {PCP_CONST, value, {}}; }
};
struct Sum final : View {  // term/sum.rs:56-92: read = the interval sum of the members; an update through it never narrows a member
  std::vector<Var> vars;
  explicit Sum(std::vector<Var> v) : vars(std::move(v)) {}
  Operand flat() const override {
    if (vars.empty()) throw Panic("At least one variable in sum.");
    Operand o{PCP_SUM, 0, {}};
    for (const Var& v : vars) {
      Operand m = v->flat();
      if (m.var == PCP_SUM) throw Panic("a Sum of Sums has no lowering");
      o.off += m.off;  // constant members and Addition offsets fold into one offset
      if (m.var != PCP_CONST) o.members.push_back(m.var);
    }
    if (o.members.empty()) o.var = PCP_CONST;
    return o;
  }
};
inline Var identity(size_t i) { return std::make_shared<Identity>(i); }
inline Var sum(std::vector<Var> vars) { return std::make_shared<Sum>(std::move(vars)); }
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
struct Propagator {  // one unit of the constraint store (one Box<dyn PropagatorConcept>, one bit of `active`)
  // type == PCP_F_LEAF: one elementary filter, or the members of a flat Conjunction / Distinct of them (group_kind 1 / 2)
  std::vector<pcp_prop> rows;
  std::vector<std::vector<uint32_t>> sums;  // the Sum views of this unit: a row names one as PCP_SUM | index in this list
  // type == PCP_F_AND / PCP_F_OR: logic::Conjunction / Disjunction over arbitrary formulas (a formula unit: pcp_model_push_formula)
  uint8_t type = PCP_F_LEAF;
  std::vector<Propagator> fs;
  bool is_tree() const { return type != PCP_F_LEAF; }
};
inline Propagator make_unit(pcp_kind k, std::initializer_list<Var> ops) {
  Propagator u;
  pcp_prop p{};
  p.kind = (uint8_t)k;
  for (int i = 0; i < 3; ++i) { p.var[i] = PCP_NOVAR; p.off[i] = 0; }
  int i = 0;
  for (const Var& v : ops) {
    Operand o = v->flat();
    if (o.var == PCP_SUM) { p.var[i] = PCP_SUM | (uint32_t)u.sums.size(); u.sums.push_back(std::move(o.members)); }
    else p.var[i] = o.var;
    p.off[i] = o.off;
    ++i;
  }
  u.rows.push_back(p);
  return u;
}
inline pcp_prop make_row(pcp_kind k, std::initializer_list<Var> ops) {  // (views without Sums)
  Propagator u = make_unit(k, ops);
  if (!u.sums.empty()) throw Panic("Sum views are not allowed here");
  return u.rows[0];
}
inline Propagator XNeqY(Var x, Var y) { return make_unit(PCP_NEQ, {x, y}); }
inline Propagator XEqY(Var x, Var y) { return make_unit(PCP_EQ, {x, y}); }
inline Propagator XLessY(Var x, Var y) { return make_unit(PCP_LT, {x, y}); }
inline Propagator XLessYPlusZ(Var x, Var y, Var z) { return make_unit(PCP_LT3, {x, y, z}); }
inline Propagator XGreaterYPlusZ(Var x, Var y, Var z) { return make_unit(PCP_GT3, {x, y, z}); }
inline Propagator XEqYPlusZ(Var x, Var y, Var z) { return make_unit(PCP_EQ3, {x, y, z}); }
inline Propagator XEqYMulZ(Var x, Var y, Var z) { return make_unit(PCP_MUL3, {x, y, z}); }
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

// ---- the reified layer (logic/) ------------------------------------------------------------------------------------
inline Propagator Boolean(Var b) { return make_unit(PCP_BOOL, {b}); }        // logic/boolean.rs:111-140: the formula "b = 1" over a 0/1 view
inline Propagator BooleanNeg(Var b) { return make_unit(PCP_NBOOL, {b}); }    // logic/boolean_neg.rs:71-96: "b = 0"
namespace detail {
// a child formula as tree nodes: a flat Conjunction of elementary members (rows) becomes an AND node over one leaf per member
inline Propagator as_tree_child(const Propagator& f) {
  if (f.is_tree() || f.rows.size() <= 1) return f;
  Propagator t;
  t.type = PCP_F_AND;
  for (const pcp_prop& r : f.rows) {
    Propagator leaf;
    pcp_prop q = r;
    q.group_kind = 0;
    for (int k = 0; k < 3; ++k)
      if (q.var[k] != PCP_CONST && q.var[k] != PCP_NOVAR && (q.var[k] & PCP_SUM) == PCP_SUM) {
        leaf.sums.push_back(f.sums[q.var[k] & ~PCP_SUM]);
        q.var[k] = PCP_SUM | (uint32_t)(leaf.sums.size() - 1);
      }
    leaf.rows.push_back(q);
    t.fs.push_back(std::move(leaf));
  }
  return t;
}
inline Propagator node(uint8_t type, std::vector<Propagator> fs) {
  if (fs.empty()) throw Panic("a Conjunction / Disjunction needs at least one child");
  Propagator t;
  t.type = type;
  for (Propagator& f : fs) t.fs.push_back(as_tree_child(f));
  return t;
}
inline bool is_sum(uint32_t var) { return var != PCP_CONST && var != PCP_NOVAR && (var & PCP_SUM) == PCP_SUM; }
}  // namespace detail
// logic::Conjunction (logic/conjunction.rs:77-119) and logic::Disjunction (logic/disjunction.rs:78-141) over arbitrary formulas: ONE unit
inline Propagator Conjunction(std::vector<Propagator> fs) { return detail::node(PCP_F_AND, std::move(fs)); }
inline Propagator Disjunction(std::vector<Propagator> fs) { return detail::node(PCP_F_OR, std::move(fs)); }
// NotFormula::not (logic/ops.rs:17-19), applied when the formula is built, as the reference does
inline Propagator not_(const Propagator& f) {
  if (f.is_tree()) {  // De Morgan: conjunction.rs:70-74, disjunction.rs:68-75
    std::vector<Propagator> nf;
    for (const Propagator& g : f.fs) nf.push_back(not_(g));
    return f.type == PCP_F_AND ? Disjunction(std::move(nf)) : Conjunction(std::move(nf));
  }
  if (f.rows.size() != 1) return not_(detail::as_tree_child(f));
  Propagator n = f;
  pcp_prop& r = n.rows[0];
  auto swap_ops = [&](int a, int b) { std::swap(r.var[a], r.var[b]); std::swap(r.off[a], r.off[b]); };
  switch (r.kind) {
    case PCP_NEQ: r.kind = PCP_EQ; break;                             // x_neq_y.rs:61-63
    case PCP_EQ: r.kind = PCP_NEQ; break;                             // x_eq_y.rs:62-64
    case PCP_LT: r.off[0] += 1; swap_ops(0, 1); break;                // x_less_y.rs:62-64: x >= y  =  y < x + 1
    case PCP_LT3: r.kind = PCP_GT3; r.off[0] += 1; break;             // x_less_y_plus_z.rs:66-72: x >= y + z  =  x + 1 > y + z
    case PCP_GT3: r.kind = PCP_LT3; r.off[0] -= 1; break;             // x_greater_y_plus_z.rs:66-72: x <= y + z  =  x - 1 < y + z
    case PCP_BOOL: r.kind = PCP_NBOOL; break;                         // boolean.rs:74-76
    case PCP_NBOOL: r.kind = PCP_BOOL; break;                         // boolean_neg.rs:66-68
    default: throw Panic("not implemented");                          // XEqYPlusZ / XEqYMulZ: unimplemented!() (x_eq_y_plus_z.rs:74-76)
  }
  return n;
}
inline Propagator implication(const Propagator& f, const Propagator& g) { return Disjunction({f, not_(g)}); }                    // logic/mod.rs:30-36
inline Propagator equivalence(const Propagator& f, const Propagator& g) { return Conjunction({implication(f, g), implication(g, f)}); }  // logic/mod.rs:38-45

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
    const uint8_t status = run_node(vs, act);
    if (status != PCP_FALSE)
      for (size_t k = 0; k < dev_units_.size(); ++k) set_bit(active_, dev_units_[k], (act[k >> 6] >> (k & 63)) & 1);
    return (SKleene)status;
  }
  const pcp_stats& last_stats() const { return last_stats_; }
  pcp_ctx* ctx() { return ctx_; }
  virtual ~GpuCStore() { pcp_ctx_destroy(ctx_); }

 protected:
  // One node through the engine: the vstore's bounds and the device units' `active` words in, the fixpoint out (both in place).
  // This form hands host buffers to pcp_propagate (two PCIe copies of the whole node per call); pcp_host_resident.hpp overrides it
  // with rows that stay in HBM.
  virtual uint8_t run_node(VStore& vs, std::vector<uint64_t>& act) {
    uint8_t status = 0;
    int32_t rc = pcp_propagate(ctx_, 1, vs.lbs().data(), vs.ubs().data(), nullptr, act.empty() ? nullptr : act.data(), &status, &last_stats_);
    if (rc == PCP_ERR_CONTRACT) throw Panic(pcp_last_error(ctx_));
    if (rc != PCP_OK) throw std::runtime_error(std::string("pcp_propagate: ") + pcp_last_error(ctx_));
    return status;
  }
  static bool get_bit(const std::vector<uint64_t>& b, size_t i) { return (i >> 6) < b.size() && ((b[i >> 6] >> (i & 63)) & 1); }
  static void set_bit(std::vector<uint64_t>& b, size_t i, bool v) {
    if ((i >> 6) >= b.size()) b.resize((i >> 6) + 1, 0);
    if (v) b[i >> 6] |= 1ull << (i & 63); else b[i >> 6] &= ~(1ull << (i & 63));
  }
  static bool is_unary(const Propagator& p) {
    if (p.is_tree() || p.rows.size() != 1 || !p.sums.empty()) return false;
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
    // runs of plain units go up in one pcp_model_push_props; a formula unit is one pcp_model_push_formula.  The Sum views of a unit are
    // registered first (pcp_model_push_sum) and its rows re-pointed from their unit-local numbers to the terms the engine returned.
    std::vector<pcp_prop> rows;
    auto flush = [&]() {
      if (!rows.empty()) check(pcp_model_push_props(ctx_, (uint32_t)rows.size(), rows.data()));
      rows.clear();
    };
    auto with_terms = [&](pcp_prop r, const std::vector<std::vector<uint32_t>>& sums) {
      for (int k = 0; k < 3; ++k)
        if (detail::is_sum(r.var[k])) {
          const std::vector<uint32_t>& m = sums[r.var[k] & ~PCP_SUM];
          uint32_t term = 0;
          check(pcp_model_push_sum(ctx_, (uint32_t)m.size(), m.data(), &term));
          r.var[k] = PCP_SUM | term;
        }
      return r;
    };
    for (size_t k = common; k < want.size(); ++k) {
      const Propagator& u = units_[want[k]];
      if (!u.is_tree()) {
        for (const pcp_prop& r : u.rows) rows.push_back(with_terms(r, u.sums));
      } else {
        flush();
        // breadth-first layout: nodes[0] is the root, the children of an inner node are consecutive
        std::vector<pcp_fnode> nodes(1);
        std::vector<pcp_prop> leaves;
        std::vector<std::pair<const Propagator*, size_t>> queue{{&u, 0}};
        for (size_t qi = 0; qi < queue.size(); ++qi) {
          const Propagator* g = queue[qi].first;
          const size_t at = queue[qi].second;
          if (g->is_tree()) {
            nodes[at] = pcp_fnode{g->type, 0, (uint16_t)g->fs.size(), (uint32_t)nodes.size()};
            for (const Propagator& c : g->fs) { nodes.push_back(pcp_fnode{}); queue.push_back({&c, nodes.size() - 1}); }
          } else {
            nodes[at] = pcp_fnode{PCP_F_LEAF, 0, 0, (uint32_t)leaves.size()};
            leaves.push_back(with_terms(g->rows[0], g->sums));
          }
        }
        check(pcp_model_push_formula(ctx_, (uint32_t)nodes.size(), nodes.data(), (uint32_t)leaves.size(), leaves.data()));
      }
      dev_units_.push_back(want[k]);
    }
    flush();
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

// propagators/cumulative.rs:59-114 — the decomposition of Schutt et al.: for each task j, the resources of the tasks that overlap j's start must
// fit the capacity.  join() allocates, per ordered pair (j, i != j): a Boolean b_i, the unit b_i <=> (s_i <= s_j /\ s_j < s_i + d_i), an
// intermediate r = b_i * r_i (XEqYMulZ), then c >= r_j + Sum(r).  Variable and propagator allocation order as in the reference.
class Cumulative {
 public:
  Cumulative(std::vector<Var> starts, std::vector<Var> durations, std::vector<Var> resources, Var capacity)
      : starts_(std::move(starts)), durations_(std::move(durations)), resources_(std::move(resources)), capacity_(std::move(capacity)) {
    if (starts_.size() != durations_.size() || starts_.size() != resources_.size()) throw Panic("Cumulative: starts, durations and resources differ in length");
  }
  void join(VStore& vstore, GpuCStore& cstore) {
    const size_t tasks = starts_.size();
    if (tasks == 1) {
      cstore.alloc(x_geq_y(capacity_, resources_[0]));  // c >= r[j]   (cumulative.rs:67-70)
      return;
    }
    for (size_t j = 0; j < tasks; ++j) {
      std::vector<Var> resource_vars;
      intermediate_.emplace_back();
      for (size_t i = 0; i < tasks; ++i) {
        if (i == j) continue;
        Propagator conj = Conjunction({x_leq_y(starts_[i], starts_[j]),                       // s[i] <= s[j]
                                       XLessYPlusZ(starts_[j], starts_[i], durations_[i])});  // s[j] < s[i] + d[i]
        Var bi = vstore.alloc(Interval(0, 1));  // Boolean::new (boolean.rs:38-43)
        cstore.alloc(equivalence(Boolean(bi), conj));
        const Operand ro = resources_[i]->flat();
        if (ro.var == PCP_SUM) throw Panic("Cumulative: a resource must be a variable view or a constant");
        const int32_t ri_ub = ro.var == PCP_CONST ? ro.off : vstore[ro.var].upper() + ro.off;
        Var r = vstore.alloc(Interval(0, ri_ub));
        intermediate_.back().push_back(r->flat().var);
        cstore.alloc(XEqYMulZ(r, bi, resources_[i]));  // r = bi * r[i]
        resource_vars.push_back(r);
      }
      cstore.alloc(x_geq_y_plus_z(capacity_, resources_[j], sum(resource_vars)));  // c >= r[j] + sum
    }
  }
  const std::vector<std::vector<uint32_t>>& intermediate() const { return intermediate_; }

 private:
  std::vector<Var> starts_, durations_, resources_;
  Var capacity_;
  std::vector<std::vector<uint32_t>> intermediate_;
};

// ---- space and search ----------------------------------------------------------------------------------------------
template <class CStoreT>
struct BasicSpace {  // Space<VStore, CStore, R> (search/space.rs:23-43) over any store with GpuCStore's interface
  VStore vstore;
  CStoreT cstore;
  explicit BasicSpace(int hip_device = 0) : cstore(hip_device) {}
  SKleene consistency() { return cstore.consistency(vstore); }  // search/space.rs:41-43
};
using Space = BasicSpace<GpuCStore>;

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
template <class SpaceT, class OnSolution>
inline Status search(SpaceT& space, bool all_solutions, uint64_t node_limit, Statistics& st, OnSolution&& on_solution) {
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
      on_solution(space);
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
template <class SpaceT>
inline Status search(SpaceT& space, bool all_solutions, uint64_t node_limit, Statistics& st) {
  return search(space, all_solutions, node_limit, st, [](const SpaceT&) {});
}

}  // namespace pcp_host
