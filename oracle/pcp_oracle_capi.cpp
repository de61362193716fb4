// pcp_oracle_capi.cpp — C entry points over the CPU oracle (pcp_oracle.hpp) for ctypes.
// TEST INFRASTRUCTURE ONLY: loaded by tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke().
// Models are described with the same pcp_prop records as the HIP ABI (include/pcp_hip.h) so that one
// lowered model feeds both sides of a parity test.
#include "pcp_oracle.hpp"

#include <cstring>

#include "../include/pcp_hip.h"

using namespace orc;

namespace {

thread_local std::string g_err;

// Everything below is written once over a `Types` (orc::fd::Types: Interval<i32> domains, VStoreFD; orc::fdset::Types:
// IntervalSet<i32> domains, VStoreSet — variable/mod.rs:35-38) and instantiated for both.
template <class T>
typename T::VarT make_view(uint32_t var, int32_t off, uint32_t n_vars, const std::vector<std::vector<uint32_t>>& sums) {
  if (var == PCP_CONST) return std::make_unique<typename T::ConstantT>(off);
  if (var >= PCP_SUM && var < PCP_NOVAR) {  // term::Sum over Identity members (+ a constant through Addition)
    const uint32_t t = var & ~PCP_SUM;
    if (t >= sums.size()) throw Panic("unknown Sum term");
    std::vector<typename T::VarT> mem;
    for (uint32_t m : sums[t]) {
      if (m >= n_vars) throw Panic("variable index out of range");
      mem.push_back(std::make_unique<typename T::IdentityT>(m));
    }
    typename T::VarT sv = std::make_unique<typename T::SumT>(std::move(mem));
    if (off != 0) return std::make_unique<typename T::AdditionT>(std::move(sv), off);
    return sv;
  }
  if (var >= n_vars) throw Panic("variable index out of range");
  typename T::VarT id = std::make_unique<typename T::IdentityT>(var);
  if (off != 0) return std::make_unique<typename T::AdditionT>(std::move(id), off);
  return id;
}

template <class T>
typename T::FormulaT make_elementary(const pcp_prop& p, uint32_t n_vars, const std::vector<std::vector<uint32_t>>& sums) {
  auto v = [&](int i) { return make_view<T>(p.var[i], p.off[i], n_vars, sums); };
  switch (p.kind) {
    case PCP_NEQ: return std::make_unique<typename T::XNeqYT>(v(0), v(1));
    case PCP_EQ: return std::make_unique<typename T::XEqYT>(v(0), v(1));
    case PCP_LT: return std::make_unique<typename T::XLessYT>(v(0), v(1));
    case PCP_LT3: return std::make_unique<typename T::XLessYPlusZT>(v(0), v(1), v(2));
    case PCP_GT3: return std::make_unique<typename T::XGreaterYPlusZT>(v(0), v(1), v(2));
    case PCP_EQ3: return std::make_unique<typename T::XEqYPlusZT>(v(0), v(1), v(2));
    case PCP_MUL3: return std::make_unique<typename T::XEqYMulZT>(v(0), v(1), v(2));
    case PCP_BOOL: return std::make_unique<typename T::BooleanT>(v(0));
    case PCP_NBOOL: return std::make_unique<typename T::BooleanNegT>(v(0));
    default: throw Panic("unknown propagator kind");
  }
}

// A Distinct-ordered conjunction: Conjunction semantics, Distinct's dependency order (distinct.rs:117-124).
template <class T>
struct DistinctGroup final : T::PropagatorT {
  using VStoreT = typename T::VStoreT;
  std::unique_ptr<typename T::ConjunctionT> conj;
  typename T::DepsT deps;  // Inner on each var, order of first appearance, no dedup beyond first appearance
  DistinctGroup(std::unique_ptr<typename T::ConjunctionT> c, typename T::DepsT d) : conj(std::move(c)), deps(std::move(d)) {}
  bool propagate(VStoreT& s) override { return conj->propagate(s); }
  SKleene is_subsumed(const VStoreT& s) const override { return conj->is_subsumed(s); }
  typename T::DepsT dependencies() const override { return deps; }
  typename T::FormulaT bclone() const override {
    auto cj = std::unique_ptr<typename T::ConjunctionT>(static_cast<typename T::ConjunctionT*>(conj->bclone().release()));
    return std::make_unique<DistinctGroup<T>>(std::move(cj), deps);
  }
  uint64_t num_elementary() const override { return conj->num_elementary(); }
};

// A formula unit (pcp_model_push_formula): nodes[0] is the root; an inner node's children are nodes[first .. first + n_children).
template <class T>
typename T::FormulaT build_formula(const std::vector<pcp_fnode>& nodes, uint32_t at, const pcp_prop* leaves, size_t n_leaves, uint32_t n_vars,
                                   const std::vector<std::vector<uint32_t>>& sums, int depth = 0) {
  if (at >= nodes.size() || depth > 16) throw Panic("malformed formula");
  const pcp_fnode& nd = nodes[at];
  if (nd.type == PCP_F_LEAF) {
    if (nd.first >= n_leaves) throw Panic("formula leaf out of range");
    return make_elementary<T>(leaves[nd.first], n_vars, sums);
  }
  if (nd.n_children == 0) throw Panic("a Conjunction / Disjunction needs at least one child");
  std::vector<typename T::FormulaT> fs;
  for (uint32_t c = 0; c < nd.n_children; ++c) fs.push_back(build_formula<T>(nodes, nd.first + c, leaves, n_leaves, n_vars, sums, depth + 1));
  if (nd.type == PCP_F_AND) return std::make_unique<typename T::ConjunctionT>(std::move(fs));
  if (nd.type == PCP_F_OR) return std::make_unique<typename T::DisjunctionT>(std::move(fs));
  throw Panic("unknown formula node type");
}

template <class T>
std::vector<typename T::FormulaT> build_units(const std::vector<pcp_prop>& props, uint32_t n_vars, const std::vector<std::vector<uint32_t>>& sums,
                                              const std::vector<std::vector<pcp_fnode>>& formulas) {
  std::vector<typename T::FormulaT> units;
  size_t i = 0;
  while (i < props.size()) {
    const pcp_prop& p = props[i];
    if (p.group_kind == 0) { units.push_back(make_elementary<T>(p, n_vars, sums)); ++i; continue; }
    if (p.group_kind == 3) {  // the leaves of formula number p.group, in a row
      size_t j = i;
      while (j < props.size() && props[j].group_kind == 3 && props[j].group == p.group) ++j;
      units.push_back(build_formula<T>(formulas.at(p.group), 0, &props[i], j - i, n_vars, sums));
      i = j;
      continue;
    }
    size_t j = i;
    std::vector<typename T::FormulaT> fs;
    typename T::DepsT ddeps;
    while (j < props.size() && props[j].group_kind == p.group_kind && props[j].group == p.group) {
      fs.push_back(make_elementary<T>(props[j], n_vars, sums));
      for (int k = 0; k < 3; ++k) {
        uint32_t v = props[j].var[k];
        if (v == PCP_CONST || v == PCP_NOVAR) continue;
        bool seen = false;
        for (auto& d : ddeps) if (d.first == v) { seen = true; break; }
        if (!seen) ddeps.emplace_back(v, Inner);
      }
      ++j;
    }
    auto conj = std::make_unique<typename T::ConjunctionT>(std::move(fs));
    if (p.group_kind == 2) units.push_back(std::make_unique<DistinctGroup<T>>(std::move(conj), std::move(ddeps)));
    else units.push_back(std::move(conj));
    i = j;
  }
  return units;
}

struct Model {
  uint32_t n_vars = 0;
  std::vector<pcp_prop> props;
  std::vector<std::vector<uint32_t>> sums;  // term::Sum views (pcp_model_push_sum)
  std::vector<std::vector<pcp_fnode>> formulas;  // formula units (pcp_model_push_formula): the trees; their leaves sit in `props`
  std::vector<fd::Formula> units;       // one per reference-level propagator, over Interval<i32>
  std::vector<fdset::Formula> units_s;  // the same model over IntervalSet<i32>
  bool units_s_valid = false;           // built on the first set-mode call
  void rebuild() {
    units = build_units<fd::Types>(props, n_vars, sums, formulas);
    units_s.clear();
    units_s_valid = false;
  }
  template <class T> const std::vector<typename T::FormulaT>& units_of();
};
template <> const std::vector<fd::Formula>& Model::units_of<fd::Types>() { return units; }
template <> const std::vector<fdset::Formula>& Model::units_of<fdset::Types>() {
  if (!units_s_valid) { units_s = build_units<fdset::Types>(props, n_vars, sums, formulas); units_s_valid = true; }
  return units_s;
}

template <class F>
int guard(F&& f) {
  try { f(); return 0; }
  catch (const Panic& e) { g_err = e.what(); return PCP_ERR_CONTRACT; }
  catch (const std::exception& e) { g_err = e.what(); return PCP_ERR_ARG; }
}

template <class T>
void fill_cstore(typename T::CStoreT& cs, Model& m) {
  cs.propagators.clear();
  for (auto& u : m.units_of<T>()) cs.propagators.push_back(u->bclone());
}

template <class CS>
void set_active(CS& cs, const uint64_t* row, size_t n_units) {
  cs.active = BitSet();
  for (size_t u = 0; u < n_units; ++u)
    if (!row || ((row[u >> 6] >> (u & 63)) & 1)) cs.active.insert(u);
}
template <class CS>
void get_active(const CS& cs, uint64_t* row, size_t n_units) {
  size_t words = (n_units + 63) / 64;
  for (size_t w = 0; w < words; ++w) row[w] = 0;
  for (size_t u = 0; u < n_units; ++u)
    if (cs.active.contains(u)) row[u >> 6] |= 1ull << (u & 63);
}

// IntervalSet <-> the bitset words of the C ABI: value v is bit (v - base) of the variable's set_words u64 words.
IntervalSet set_from_bits(const uint64_t* w, uint32_t set_words, int32_t base) {
  IntervalSet s;
  int64_t run_lo = 0; bool in_run = false;
  const int64_t nbits = (int64_t)set_words * 64;
  for (int64_t b = 0; b <= nbits; ++b) {
    const bool on = b < nbits && ((w[b >> 6] >> (b & 63)) & 1);
    if (on && !in_run) { in_run = true; run_lo = b; }
    if (!on && in_run) { in_run = false; s.iv.push_back(Interval::make(base + run_lo, base + b - 1)); }
  }
  return s;
}
void set_to_bits(const IntervalSet& s, uint64_t* w, uint32_t set_words, int32_t base) {
  for (uint32_t k = 0; k < set_words; ++k) w[k] = 0;
  for (auto& i : s.iv)
    for (int64_t v = i.lb; v <= i.ub; ++v) {
      const int64_t b = v - base;
      if (b < 0 || b >= (int64_t)set_words * 64) throw Panic("IntervalSet value outside the bitset universe");
      w[b >> 6] |= 1ull << (b & 63);
    }
}

}  // namespace

extern "C" {

struct orc_stats_c { uint64_t steps, pops, narrowings, nodes, failed_nodes, subscriptions; };
struct orc_search_stats_c { uint64_t num_solution, num_failed_node, num_prune, num_nodes; uint32_t end_of_search; };

const char* orc_last_error() { return g_err.c_str(); }

void* orc_model_new(uint32_t n_vars) { auto* m = new Model(); m->n_vars = n_vars; return m; }
void orc_model_free(void* h) { delete static_cast<Model*>(h); }
int orc_model_push_props(void* h, uint32_t n, const pcp_prop* props) {
  auto* m = static_cast<Model*>(h);
  return guard([&] {
    size_t old = m->props.size();
    m->props.insert(m->props.end(), props, props + n);
    try { m->rebuild(); } catch (...) { m->props.resize(old); m->rebuild(); throw; }
  });
}
int orc_model_push_sum(void* h, uint32_t n, const uint32_t* vars, uint32_t* term) {
  auto* m = static_cast<Model*>(h);
  return guard([&] {
    if (n == 0) throw Panic("At least one variable in sum.");
    m->sums.emplace_back(vars, vars + n);
    *term = (uint32_t)m->sums.size() - 1;
  });
}
// One formula unit: a tree of Conjunction / Disjunction nodes over elementary leaves (negations already applied by the caller,
// as the reference's NotFormula::not does at construction time).
int orc_model_push_formula(void* h, uint32_t n_nodes, const pcp_fnode* nodes, uint32_t n_leaves, const pcp_prop* leaves) {
  auto* m = static_cast<Model*>(h);
  return guard([&] {
    if (n_nodes == 0 || n_leaves == 0) throw Panic("empty formula");
    const size_t old = m->props.size();
    m->formulas.emplace_back(nodes, nodes + n_nodes);
    for (uint32_t i = 0; i < n_leaves; ++i) {
      pcp_prop p = leaves[i];
      p.group_kind = 3;
      p.group = (uint32_t)m->formulas.size() - 1;
      m->props.push_back(p);
    }
    try { m->rebuild(); } catch (...) { m->props.resize(old); m->formulas.pop_back(); m->rebuild(); throw; }
  });
}
uint32_t orc_model_n_units(void* h) { return (uint32_t)static_cast<Model*>(h)->units.size(); }
// Store::is_subsumed (propagation/store.rs:232-238): the Kleene conjunction over ALL propagators of the store.
int orc_is_subsumed(void* h, const int32_t* lb, const int32_t* ub, uint8_t* out) {
  auto* m = static_cast<Model*>(h);
  return guard([&] {
    fd::VStore vs;
    for (uint32_t v = 0; v < m->n_vars; ++v) vs.alloc(Interval::make(lb[v], ub[v]));
    SKleene k = SKleene::True;
    for (auto& u : m->units) k = kand(k, u->is_subsumed(vs));
    *out = (uint8_t)k;
  });
}

// ≡ Consistency::consistency on n_nodes independent spaces (same contract as pcp_propagate).
int orc_consistency(void* h, uint32_t n_nodes, int32_t* lb, int32_t* ub, uint64_t* active, uint8_t* status,
                    orc_stats_c* stats, int check_dup) {
  auto* m = static_cast<Model*>(h);
  return guard([&] {
    Stats st;
    CStore cs;
    cs.check_dup = check_dup != 0;
    cs.stats = &st;
    fill_cstore<fd::Types>(cs, *m);
    size_t nu = m->units.size(), words = (nu + 63) / 64;
    for (uint32_t n = 0; n < n_nodes; ++n) {
      VStore vs;
      for (uint32_t v = 0; v < m->n_vars; ++v) vs.alloc(Interval{lb[(size_t)n * m->n_vars + v], ub[(size_t)n * m->n_vars + v]});
      set_active(cs, active ? active + (size_t)n * words : nullptr, nu);
      SKleene k = cs.consistency(vs);
      status[n] = (uint8_t)k;
      for (uint32_t v = 0; v < m->n_vars; ++v) {
        lb[(size_t)n * m->n_vars + v] = vs.memory[v].lb;
        ub[(size_t)n * m->n_vars + v] = vs.memory[v].ub;
      }
      if (active) get_active(cs, active + (size_t)n * words, nu);
    }
    if (stats) *stats = orc_stats_c{st.steps, st.pops, st.narrowings, st.nodes, st.failed_nodes, st.subscriptions};
  });
}

// The reference's test fixture propagators/mod.rs:108-129 (test_propagation) on ONE unit:
// is_subsumed before, one propagate(), the drained delta (ascending var), is_subsumed after, final domains.
int orc_kat(uint32_t n_vars, int32_t* lb, int32_t* ub, uint32_t n_props, const pcp_prop* props, uint8_t* before,
            uint8_t* ok, uint8_t* after, uint32_t* delta_n, uint32_t* delta_var, uint8_t* delta_ev) {
  return guard([&] {
    Model m;
    m.n_vars = n_vars;
    m.props.assign(props, props + n_props);
    m.rebuild();
    if (m.units.size() != 1) throw Panic("orc_kat expects exactly one unit");
    VStore vs;
    for (uint32_t v = 0; v < n_vars; ++v) vs.alloc(Interval{lb[v], ub[v]});
    Propagator& p = *m.units[0];
    *before = (uint8_t)p.is_subsumed(vs);
    bool r = p.propagate(vs);
    *ok = r ? 1 : 0;
    *delta_n = 0;
    if (r) {
      for (auto& [v, ev] : vs.drain_delta()) { delta_var[*delta_n] = (uint32_t)v; delta_ev[*delta_n] = (uint8_t)ev; ++*delta_n; }
    }
    *after = (uint8_t)p.is_subsumed(vs);
    for (uint32_t v = 0; v < n_vars; ++v) { lb[v] = vs.memory[v].lb; ub[v] = vs.memory[v].ub; }
  });
}

// variable/store.rs test_op (:369-393): one update on a one-variable store; returns update() and the event (-1 none).
int orc_vstore_update(int32_t lb, int32_t ub, int32_t nlb, int32_t nub, uint8_t* ok, int32_t* event) {
  return guard([&] {
    VStore vs;
    vs.alloc(Interval{lb, ub});
    bool r = vs.update(0, Interval{nlb, nub});
    *ok = r;
    *event = -1;
    for (auto& [v, ev] : vs.drain_delta()) { (void)v; *event = ev; }
  });
}
// Interval ops used by the store tests (variable/store.rs:467-525): 0 shrink_left, 1 shrink_right, 2 intersection, 3 difference(value a)
int orc_interval_op(int op, int32_t lb, int32_t ub, int32_t a, int32_t b, int32_t* rlb, int32_t* rub) {
  return guard([&] {
    Interval x{lb, ub}, r{0, 0};
    switch (op) {
      case 0: r = x.shrink_left(a); break;
      case 1: r = x.shrink_right(a); break;
      case 2: r = x.intersection(Interval{a, b}); break;
      case 3: r = x.difference(a); break;
      case 4: r = x.strict_shrink_left(a); break;
      case 5: r = x.strict_shrink_right(a); break;
      default: throw Panic("bad op");
    }
    *rlb = r.lb; *rub = r.ub;
  });
}

// IndexedDeps / RelaxedFifo handles for the table tests (indexed_deps.rs:159-231, relaxed_fifo.rs:78-132).
void* orc_reactor_new(uint32_t num_vars) { return new IndexedDeps(num_vars, kNumEvents, true); }
void orc_reactor_free(void* r) { delete static_cast<IndexedDeps*>(r); }
int orc_reactor_subscribe(void* r, uint32_t var, uint32_t ev, uint32_t prop) { return guard([&] { static_cast<IndexedDeps*>(r)->subscribe(var, (FDEvent)ev, prop); }); }
int orc_reactor_unsubscribe(void* r, uint32_t var, uint32_t ev, uint32_t prop) {
  return guard([&] {
    auto* x = static_cast<IndexedDeps*>(r);
    if (x->num_subscriptions == 0) throw Panic("attempt to subtract with overflow");  // Rust debug: usize underflow panics first
    x->unsubscribe(var, (FDEvent)ev, prop);
  });
}
int orc_reactor_react(void* r, uint32_t var, uint32_t ev, uint32_t* out, uint32_t cap, uint32_t* n) {
  return guard([&] {
    auto v = static_cast<IndexedDeps*>(r)->react(var, (FDEvent)ev);
    *n = (uint32_t)v.size();
    for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = (uint32_t)v[i];
  });
}
int orc_reactor_is_empty(void* r) { return static_cast<IndexedDeps*>(r)->is_empty(); }

void* orc_fifo_new(uint32_t cap) { return new RelaxedFifo(cap); }
void orc_fifo_free(void* f) { delete static_cast<RelaxedFifo*>(f); }
int orc_fifo_schedule(void* f, uint32_t i) { return guard([&] { static_cast<RelaxedFifo*>(f)->schedule(i); }); }
int orc_fifo_unschedule(void* f, uint32_t i) { return guard([&] { static_cast<RelaxedFifo*>(f)->unschedule(i); }); }
int64_t orc_fifo_pop(void* f) { auto r = static_cast<RelaxedFifo*>(f)->pop(); return r ? (int64_t)*r : -1; }
int orc_fifo_is_empty(void* f) { return static_cast<RelaxedFifo*>(f)->is_empty(); }

int32_t orc_middle_val(int32_t lb, int32_t ub) { return middle_val(Interval{lb, ub}); }
int64_t orc_first_smallest_var(uint32_t n, const int32_t* lb, const int32_t* ub) {
  VStore vs;
  try {
    for (uint32_t i = 0; i < n; ++i) vs.alloc(Interval{lb[i], ub[i]});
    return (int64_t)first_smallest_var(vs);
  } catch (const Panic& e) { g_err = e.what(); return -1; }
}

// DFS with the default engine (search/mod.rs:45-52), optionally AllSolution and StopNode(node_limit).
// Records up to max_records explored nodes.  Recorded inputs are in the FOLDED form the HIP driver uses:
// the branch propagator `x <= v` / `x > v` (binary_split.rs:46-57) is applied to x's bounds in lb_in/ub_in, and the
// active rows cover the model's units only.  A node whose folded domain is empty is recorded with lb_in > ub_in.
int orc_search(void* h, const int32_t* lb0, const int32_t* ub0, int all_solutions, uint64_t node_limit, int check_dup,
               orc_search_stats_c* out, orc_stats_c* pstats, uint32_t max_records, int32_t* rec_lb_in, int32_t* rec_ub_in,
               int32_t* rec_lb_out, int32_t* rec_ub_out, uint64_t* rec_active_in, uint64_t* rec_active_out,
               uint8_t* rec_status, uint32_t* n_recorded, int32_t* first_solution) {
  auto* m = static_cast<Model*>(h);
  return guard([&] {
    Stats st;
    Space sp;
    sp.cstore.check_dup = check_dup != 0;
    sp.cstore.stats = &st;
    for (uint32_t v = 0; v < m->n_vars; ++v) sp.vstore.alloc(Interval{lb0[v], ub0[v]});
    for (auto& u : m->units) sp.cstore.alloc(u->bclone());
    size_t nu = m->units.size(), words = (nu + 63) / 64, V = m->n_vars;
    uint32_t nrec = 0;
    bool have_solution = false;
    SearchStats ss = fd::dfs(sp, all_solutions != 0, node_limit,
                         [&](const std::vector<Interval>& before, const BitSet& active_before, Space& s, SKleene k) {
      if (k == SKleene::True && !have_solution && first_solution) {
        have_solution = true;
        for (size_t v = 0; v < V; ++v) first_solution[v] = s.vstore.memory[v].lb;
      }
      if (nrec >= max_records) return;
      size_t r = nrec++;
      for (size_t v = 0; v < V; ++v) { rec_lb_in[r * V + v] = before[v].lb; rec_ub_in[r * V + v] = before[v].ub; }
      // fold the newest branch propagator (the last unit, if beyond the model) into the input domain
      if (s.cstore.propagators.size() > nu) {
        auto* br = dynamic_cast<XLessY*>(s.cstore.propagators.back().get());
        if (!br) throw Panic("branch propagator is not XLessY");
        // left: XLessY(Identity x, Addition(Constant v,1))  => x.ub = min(ub, v)
        // right: XLessY(Constant v, Identity x)             => x.lb = max(lb, v+1)
        if (auto* idx = dynamic_cast<Identity*>(br->x.get())) {
          Interval y = br->y->read(s.vstore);  // {v+1}
          size_t x = idx->idx;
          rec_ub_in[r * V + x] = std::min(rec_ub_in[r * V + x], y.ub - 1);
        } else {
          auto* idy = dynamic_cast<Identity*>(br->y.get());
          if (!idy) throw Panic("unexpected branch propagator shape");
          Interval c = br->x->read(s.vstore);  // {v}
          size_t x = idy->idx;
          rec_lb_in[r * V + x] = std::max(rec_lb_in[r * V + x], c.lb + 1);
        }
      }
      for (size_t w = 0; w < words; ++w) { rec_active_in[r * words + w] = 0; rec_active_out[r * words + w] = 0; }
      for (size_t u = 0; u < nu; ++u) {
        if (active_before.contains(u)) rec_active_in[r * words + (u >> 6)] |= 1ull << (u & 63);
        if (s.cstore.active.contains(u)) rec_active_out[r * words + (u >> 6)] |= 1ull << (u & 63);
      }
      for (size_t v = 0; v < V; ++v) { rec_lb_out[r * V + v] = s.vstore.memory[v].lb; rec_ub_out[r * V + v] = s.vstore.memory[v].ub; }
      rec_status[r] = (uint8_t)k;
    });
    if (n_recorded) *n_recorded = nrec;
    if (out) *out = orc_search_stats_c{ss.num_solution, ss.num_failed_node, ss.num_prune, ss.num_nodes, ss.end_of_search ? 1u : 0u};
    if (pstats) *pstats = orc_stats_c{st.steps, st.pops, st.narrowings, st.nodes, st.failed_nodes, st.subscriptions};
  });
}

// ---- set mode: the same two entry points over IntervalSet<i32> domains (VStoreSet, the FDSpace default) ----------------
// bits: [n_nodes][n_vars][set_words] u64 in/out, value v = bit (v - base); lb/ub: [n_nodes][n_vars] out (bounds of the sets).
int orc_consistency_set(void* h, uint32_t n_nodes, int32_t* lb, int32_t* ub, uint64_t* bits, uint32_t set_words, int32_t base,
                        uint64_t* active, uint8_t* status, orc_stats_c* stats, int check_dup) {
  auto* m = static_cast<Model*>(h);
  return guard([&] {
    using T = fdset::Types;
    Stats st;
    fdset::CStore cs;
    cs.check_dup = check_dup != 0;
    cs.stats = &st;
    fill_cstore<T>(cs, *m);
    const size_t nu = m->units.size(), words = (nu + 63) / 64, V = m->n_vars;
    for (uint32_t n = 0; n < n_nodes; ++n) {
      fdset::VStore vs;
      for (size_t v = 0; v < V; ++v) vs.alloc(set_from_bits(bits + ((size_t)n * V + v) * set_words, set_words, base));
      set_active(cs, active ? active + (size_t)n * words : nullptr, nu);
      SKleene k = cs.consistency(vs);
      status[n] = (uint8_t)k;
      for (size_t v = 0; v < V; ++v) {
        const IntervalSet& d = vs.memory[v];
        set_to_bits(d, bits + ((size_t)n * V + v) * set_words, set_words, base);
        lb[(size_t)n * V + v] = d.is_empty() ? 1 : d.lower();
        ub[(size_t)n * V + v] = d.is_empty() ? 0 : d.upper();
      }
      if (active) get_active(cs, active + (size_t)n * words, nu);
    }
    if (stats) *stats = orc_stats_c{st.steps, st.pops, st.narrowings, st.nodes, st.failed_nodes, st.subscriptions};
  });
}

// DFS over FDSpace (search/mod.rs:41-52): variables allocated as IntervalSet::new(lb0, ub0) (example/src/nqueens.rs:32-35).
// Records as orc_search, plus the sets: rec_bits_in is the FOLDED input (the branch propagator x <= v / x > v applied to the
// set), rec_bits_out the fixpoint.
// root_bits != NULL: the root's domains are these sets ([n_vars][set_words] words) instead of the intervals lb0..ub0 — a subtree of the
// FDSpace search (an open node of a breadth-first expansion has holes: no interval describes it).  Same loop otherwise.
int orc_search_set_root(void* h, const int32_t* lb0, const int32_t* ub0, const uint64_t* root_bits, uint32_t set_words, int32_t base, int all_solutions,
                        uint64_t node_limit, int check_dup, orc_search_stats_c* out, orc_stats_c* pstats, uint32_t max_records, uint64_t* rec_bits_in,
                        uint64_t* rec_bits_out, int32_t* rec_lb_out, int32_t* rec_ub_out, uint64_t* rec_active_in, uint64_t* rec_active_out,
                        uint8_t* rec_status, uint32_t* n_recorded, int32_t* first_solution) {
  auto* m = static_cast<Model*>(h);
  return guard([&] {
    using T = fdset::Types;
    Stats st;
    fdset::Space sp;
    sp.cstore.check_dup = check_dup != 0;
    sp.cstore.stats = &st;
    for (uint32_t v = 0; v < m->n_vars; ++v)
      sp.vstore.alloc(root_bits ? set_from_bits(root_bits + (size_t)v * set_words, set_words, base) : IntervalSet::from_interval(lb0[v], ub0[v]));
    for (auto& u : m->units_of<T>()) sp.cstore.alloc(u->bclone());
    const size_t nu = m->units.size(), words = (nu + 63) / 64, V = m->n_vars;
    uint32_t nrec = 0;
    bool have_solution = false;
    SearchStats ss = T::run_dfs(sp, all_solutions != 0, node_limit,
                                [&](const std::vector<IntervalSet>& before, const BitSet& active_before, fdset::Space& s, SKleene k) {
      if (k == SKleene::True && !have_solution && first_solution) {
        have_solution = true;
        for (size_t v = 0; v < V; ++v) first_solution[v] = s.vstore.memory[v].lower();
      }
      if (nrec >= max_records) return;
      const size_t r = nrec++;
      std::vector<IntervalSet> in = before;
      if (s.cstore.propagators.size() > nu) {  // fold the newest branch propagator into the input set
        auto* br = dynamic_cast<fdset::XLessY*>(s.cstore.propagators.back().get());
        if (!br) throw Panic("branch propagator is not XLessY");
        if (auto* idx = dynamic_cast<fdset::Identity*>(br->x.get())) {
          IntervalSet y = br->y->read(s.vstore);  // {v+1}
          in[idx->idx] = in[idx->idx].strict_shrink_right(y.upper());
        } else {
          auto* idy = dynamic_cast<fdset::Identity*>(br->y.get());
          if (!idy) throw Panic("unexpected branch propagator shape");
          IntervalSet c = br->x->read(s.vstore);  // {v}
          in[idy->idx] = in[idy->idx].strict_shrink_left(c.lower());
        }
      }
      for (size_t v = 0; v < V; ++v) {
        set_to_bits(in[v], rec_bits_in + (r * V + v) * set_words, set_words, base);
        const IntervalSet& d = s.vstore.memory[v];
        set_to_bits(d, rec_bits_out + (r * V + v) * set_words, set_words, base);
        rec_lb_out[r * V + v] = d.is_empty() ? 1 : d.lower();
        rec_ub_out[r * V + v] = d.is_empty() ? 0 : d.upper();
      }
      for (size_t w = 0; w < words; ++w) { rec_active_in[r * words + w] = 0; rec_active_out[r * words + w] = 0; }
      for (size_t u = 0; u < nu; ++u) {
        if (active_before.contains(u)) rec_active_in[r * words + (u >> 6)] |= 1ull << (u & 63);
        if (s.cstore.active.contains(u)) rec_active_out[r * words + (u >> 6)] |= 1ull << (u & 63);
      }
      rec_status[r] = (uint8_t)k;
    });
    if (n_recorded) *n_recorded = nrec;
    if (out) *out = orc_search_stats_c{ss.num_solution, ss.num_failed_node, ss.num_prune, ss.num_nodes, ss.end_of_search ? 1u : 0u};
    if (pstats) *pstats = orc_stats_c{st.steps, st.pops, st.narrowings, st.nodes, st.failed_nodes, st.subscriptions};
  });
}

int orc_search_set(void* h, const int32_t* lb0, const int32_t* ub0, uint32_t set_words, int32_t base, int all_solutions, uint64_t node_limit,
                   int check_dup, orc_search_stats_c* out, orc_stats_c* pstats, uint32_t max_records, uint64_t* rec_bits_in,
                   uint64_t* rec_bits_out, int32_t* rec_lb_out, int32_t* rec_ub_out, uint64_t* rec_active_in, uint64_t* rec_active_out,
                   uint8_t* rec_status, uint32_t* n_recorded, int32_t* first_solution) {
  return orc_search_set_root(h, lb0, ub0, nullptr, set_words, base, all_solutions, node_limit, check_dup, out, pstats, max_records, rec_bits_in, rec_bits_out,
                             rec_lb_out, rec_ub_out, rec_active_in, rec_active_out, rec_status, n_recorded, first_solution);
}

// IntervalSet algebra for the table tests: op 0 difference(a), 1 shrink_left(a), 2 shrink_right(a), 3 intersection with the second
// set, 4 shift by a, 5 is_disjoint (result in *flag), 6 is_subset.  Sets as bitset words over [base, base + 64 set_words).
int orc_set_op(int op, const uint64_t* x, const uint64_t* y, uint32_t set_words, int32_t base, int32_t a, uint64_t* out, int32_t* flag) {
  return guard([&] {
    IntervalSet s = set_from_bits(x, set_words, base), r;
    *flag = 0;
    switch (op) {
      case 0: r = s.difference(a); break;
      case 1: r = s.shrink_left(a); break;
      case 2: r = s.shrink_right(a); break;
      case 3: r = s.intersection(set_from_bits(y, set_words, base)); break;
      case 4: r = s.add(a); break;
      case 5: *flag = s.is_disjoint(set_from_bits(y, set_words, base)); r = s; break;
      case 6: *flag = s.is_subset(set_from_bits(y, set_words, base)); r = s; break;
      default: throw Panic("bad op");
    }
    set_to_bits(r, out, set_words, base);
  });
}

}  // extern "C"
