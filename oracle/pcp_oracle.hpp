// pcp_oracle.hpp — CPU restatement of libpcp 0.7.0's propagation fixpoint.  TEST INFRASTRUCTURE ONLY.
//
// This file is the *oracle* (checker) for the HIP engine in pcp_amd/csrc.  Only tests/, bench.py's
// cpu_baseline leg and __graft_entry__.smoke() may build, load or call it; the product path never does.
//
// The reference (/root/reference, pure Rust) cannot be compiled here (no cargo/rustc, crates not vendored),
// so this is a structure-faithful C++17 restatement: boxed virtual views and propagators, per-node
// prepare() rebuild of the reactor including its O(degree) duplicate-subscription check, VecDeque+bitset
// FIFO, reaction after every propagator.  Every function cites the reference file:line it follows
// (paths relative to /root/reference/src/libpcp unless stated).
//
// PARITY PIN: the reference cannot run here, so the oracle is pinned against the known-answer vectors
// held by the reference's own #[test]s, transcribed in tests/golden/*.json (each with file:line
// provenance) and replayed by tests/test_oracle_golden.py.  The domain algebra lives in the third-party
// crate `intervallum` ("^1.2.0", Cargo.toml:22; no lockfile, not vendored): its published Interval<i32>
// semantics are restated in struct Interval below and are consistent with all transcribed vectors.
// Unpinned corners (no reference test observes them): i32 overflow at extreme bounds, Interval×Mul with
// negative operands, IntervalSet interior removal (set mode is not restated at all).
#pragma once
#include <algorithm>
#include <cstdint>
#include <deque>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace orc {

// A reference `assert!`/panic (contract violation) is reported as this exception.
struct Panic : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// trilean::SKleene (crate trilean ^1.0.1): False/True/Unknown with Kleene and/not.
enum class SKleene : uint8_t { False = 0, True = 1, Unknown = 2 };
inline SKleene kand(SKleene a, SKleene b) {
  if (a == SKleene::False || b == SKleene::False) return SKleene::False;
  if (a == SKleene::True && b == SKleene::True) return SKleene::True;
  return SKleene::Unknown;
}
inline SKleene knot(SKleene a) {
  if (a == SKleene::False) return SKleene::True;
  if (a == SKleene::True) return SKleene::False;
  return SKleene::Unknown;
}

// ---------------------------------------------------------------------------------------------------
// Interval<i32> — crate intervallum (lib name `interval`), SURVEY.md Appendix A.1.  [lb,ub], empty <=> lb>ub.
// Arithmetic is done in i64 and range-checked so that an overflow is a Panic rather than silent UB
// (Rust: debug panics, release wraps; the build rejects such inputs at the boundary).
// ---------------------------------------------------------------------------------------------------
struct Interval {
  int32_t lb, ub;
  static Interval make(int64_t l, int64_t u) {
    if (l < INT32_MIN || l > INT32_MAX || u < INT32_MIN || u > INT32_MAX) throw Panic("i32 overflow in interval arithmetic");
    return Interval{(int32_t)l, (int32_t)u};
  }
  static Interval empty() { return Interval{1, 0}; }
  static Interval singleton(int32_t v) { return Interval{v, v}; }
  bool is_empty() const { return lb > ub; }
  bool is_singleton() const { return lb == ub; }
  uint32_t size() const { return is_empty() ? 0u : (uint32_t)((int64_t)ub - (int64_t)lb + 1); }
  int32_t lower() const { return lb; }
  int32_t upper() const { return ub; }
  bool contains(int32_t v) const { return lb <= v && v <= ub; }
  bool is_subset(const Interval& b) const { return is_empty() || (lb >= b.lb && ub <= b.ub); }
  bool is_disjoint(const Interval& b) const { return is_empty() || b.is_empty() || lb > b.ub || b.lb > ub; }
  bool overlap(const Interval& b) const { return !is_disjoint(b); }
  Interval intersection(const Interval& b) const { return Interval{std::max(lb, b.lb), std::min(ub, b.ub)}; }
  // Difference<Bound>: removes v only if it is a bound (interior values are NOT removed on Interval;
  // pinned by propagators/cmp/x_neq_y.rs:128 and term/constant.rs:165).
  Interval difference(int32_t v) const {
    if (is_empty()) return *this;
    if (v == lb) return make((int64_t)lb + 1, ub);
    if (v == ub) return make(lb, (int64_t)ub - 1);
    return *this;
  }
  Interval shrink_left(int64_t b) const { return b > lb ? make(b, ub) : *this; }
  Interval shrink_right(int64_t b) const { return b < ub ? make(lb, b) : *this; }
  Interval strict_shrink_left(int64_t b) const { return shrink_left(b + 1); }
  Interval strict_shrink_right(int64_t b) const { return shrink_right(b - 1); }
  Interval add(int32_t v) const { return is_empty() ? *this : make((int64_t)lb + v, (int64_t)ub + v); }
  Interval sub(int32_t v) const { return is_empty() ? *this : make((int64_t)lb - v, (int64_t)ub - v); }
  Interval add(const Interval& b) const {
    if (is_empty() || b.is_empty()) return empty();
    return make((int64_t)lb + b.lb, (int64_t)ub + b.ub);
  }
  Interval mul(const Interval& b) const {
    if (is_empty() || b.is_empty()) return empty();
    int64_t p[4] = {(int64_t)lb * b.lb, (int64_t)lb * b.ub, (int64_t)ub * b.lb, (int64_t)ub * b.ub};
    return make(*std::min_element(p, p + 4), *std::max_element(p, p + 4));
  }
  bool operator==(const Interval& o) const { return (is_empty() && o.is_empty()) || (lb == o.lb && ub == o.ub); }
};

// propagation/events/mod.rs:24-70 — FDEvent lattice, Merge = min, MonotonicEvent::new.
enum FDEvent : uint8_t { Assignment = 0, Bound = 1, Inner = 2 };
constexpr size_t kNumEvents = 3;  // EventIndex::size(), events/mod.rs:42-44
inline FDEvent merge(FDEvent e, FDEvent f) { return std::min(e, f); }  // events/mod.rs:31-35
inline std::optional<FDEvent> event_new(const Interval& little, const Interval& big) {  // events/mod.rs:51-69
  if (!little.is_subset(big)) throw Panic("Events are computed on the difference between `little` and `big`.");
  if (little.size() != big.size()) {
    if (little.is_singleton()) return Assignment;
    if (little.lower() != big.lower() || little.upper() != big.upper()) return Bound;
    return Inner;
  }
  return std::nullopt;
}

// ---------------------------------------------------------------------------------------------------
// variable::Store<Memory,Event> with copy memory — variable/store.rs:28-237.  `delta` is a VecMap (drain
// yields ascending keys); the trail (a22) is search-side and replaced by whole-store copies.
// ---------------------------------------------------------------------------------------------------
struct VStore {
  std::vector<Interval> memory;
  std::vector<int8_t> delta;  // -1 = absent, else FDEvent (VecMap<Event>)
  std::vector<uint32_t> delta_keys;  // keys present (unsorted; drained sorted)
  bool has_changed_ = false;

  size_t size() const { return memory.size(); }
  size_t alloc(Interval dom) {  // variable/store.rs:129-141
    if (dom.is_empty()) throw Panic("alloc: empty domain");
    memory.push_back(dom);
    delta.push_back(-1);
    return memory.size() - 1;
  }
  const Interval& at(size_t idx) const {  // variable/store.rs:168-182
    if (idx >= memory.size()) throw Panic("Variable not registered in the store.");
    return memory[idx];
  }
  void update_delta(size_t key, const Interval& old_dom) {  // variable/store.rs:94-106
    if (auto ev = event_new(memory[key], old_dom)) {
      has_changed_ = true;
      if (delta[key] >= 0) delta[key] = (int8_t)merge((FDEvent)delta[key], *ev);
      else { delta[key] = (int8_t)*ev; delta_keys.push_back((uint32_t)key); }
    }
  }
  bool update(size_t idx, Interval dom) {  // variable/store.rs:151-166
    if (!dom.is_subset(at(idx))) throw Panic("Domain update must be monotonic.");
    if (dom.is_empty()) return false;
    if (dom.size() < memory[idx].size()) {
      Interval old = memory[idx];
      memory[idx] = dom;  // memory.replace (trail elided)
      update_delta(idx, old);
    }
    return true;
  }
  std::vector<std::pair<size_t, FDEvent>> drain_delta() {  // variable/store.rs:225-229 (ascending var index)
    std::sort(delta_keys.begin(), delta_keys.end());
    std::vector<std::pair<size_t, FDEvent>> out;
    out.reserve(delta_keys.size());
    for (uint32_t k : delta_keys) { out.emplace_back(k, (FDEvent)delta[k]); delta[k] = -1; }
    delta_keys.clear();
    return out;
  }
  bool has_changed() const { return has_changed_; }   // variable/store.rs:230-232
  void reset_changed() { has_changed_ = false; }      // variable/store.rs:234-236
};

using Deps = std::vector<std::pair<size_t, FDEvent>>;

// ---------------------------------------------------------------------------------------------------
// Views — term/ops.rs:18-28 (StoreRead / StoreMonotonicUpdate / ViewDependencies), boxed like Var<VStore>.
// ---------------------------------------------------------------------------------------------------
struct View {
  virtual ~View() = default;
  virtual Interval read(const VStore&) const = 0;
  virtual bool update(VStore&, Interval) = 0;
  virtual Deps dependencies(FDEvent) const = 0;
  virtual std::unique_ptr<View> bclone() const = 0;
};
using Var = std::unique_ptr<View>;

struct Identity final : View {  // term/identity.rs:47-70
  size_t idx;
  explicit Identity(size_t i) : idx(i) {}
  Interval read(const VStore& s) const override { return s.at(idx); }
  bool update(VStore& s, Interval v) override { return s.update(idx, v); }
  Deps dependencies(FDEvent e) const override { return Deps{{idx, e}}; }
  Var bclone() const override { return std::make_unique<Identity>(idx); }
};
struct Addition final : View {  // term/addition.rs:80-110
  Var x; int32_t v;
  Addition(Var x_, int32_t v_) : x(std::move(x_)), v(v_) {}
  Interval read(const VStore& s) const override { return x->read(s).add(v); }
  bool update(VStore& s, Interval value) override { return x->update(s, value.sub(v)); }
  Deps dependencies(FDEvent e) const override { return x->dependencies(e); }
  Var bclone() const override { return std::make_unique<Addition>(x->bclone(), v); }
};
struct Constant final : View {  // term/constant.rs:43-68
  int32_t value;
  explicit Constant(int32_t v) : value(v) {}
  Interval read(const VStore&) const override { return Interval::singleton(value); }
  bool update(VStore&, Interval v) override { return !v.is_empty() && v.contains(value); }
  Deps dependencies(FDEvent) const override { return Deps{}; }
  Var bclone() const override { return std::make_unique<Constant>(value); }
};
struct Sum final : View {  // term/sum.rs:56-92
  std::vector<Var> vars;
  explicit Sum(std::vector<Var> v) : vars(std::move(v)) {}
  Interval read(const VStore& s) const override {
    if (vars.empty()) throw Panic("At least one variable in sum.");
    Interval a = vars[0]->read(s);
    for (size_t i = 1; i < vars.size(); ++i) a = a.add(vars[i]->read(s));
    return a;
  }
  bool update(VStore& s, Interval value) override {
    if (vars.size() == 1) return vars[0]->update(s, value);
    return read(s).overlap(value);
  }
  Deps dependencies(FDEvent e) const override {
    Deps d;
    for (auto& v : vars) { Deps x = v->dependencies(e); d.insert(d.end(), x.begin(), x.end()); }
    return d;
  }
  Var bclone() const override {
    std::vector<Var> c;
    for (auto& v : vars) c.push_back(v->bclone());
    return std::make_unique<Sum>(std::move(c));
  }
};

// ---------------------------------------------------------------------------------------------------
// Propagators — propagation/ops.rs:17-29 (Propagator / Subsumption / PropagatorDependencies), boxed like
// Box<dyn PropagatorConcept> (propagation/concept.rs:21-53).
// ---------------------------------------------------------------------------------------------------
struct Propagator {
  virtual ~Propagator() = default;
  virtual bool propagate(VStore&) = 0;
  virtual SKleene is_subsumed(const VStore&) const = 0;
  virtual Deps dependencies() const = 0;
  virtual std::unique_ptr<Propagator> bclone() const = 0;
  virtual uint64_t num_elementary() const { return 1; }  // children evaluated per pop (metric unit)
};
using Formula = std::unique_ptr<Propagator>;

struct XEqY final : Propagator {  // propagators/cmp/x_eq_y.rs:67-116
  Var x, y;
  XEqY(Var x_, Var y_) : x(std::move(x_)), y(std::move(y_)) {}
  SKleene is_subsumed(const VStore& s) const override {  // :73-94
    Interval a = x->read(s), b = y->read(s);
    if (a.lower() == b.upper() && a.upper() == b.lower()) return SKleene::True;
    if (a.is_disjoint(b)) return SKleene::False;
    return SKleene::Unknown;
  }
  bool propagate(VStore& s) override {  // :102-107
    Interval a = x->read(s), b = y->read(s);
    Interval n = a.intersection(b);
    return x->update(s, n) && y->update(s, n);
  }
  Deps dependencies() const override {  // :110-115
    Deps d = x->dependencies(Inner), e = y->dependencies(Inner);
    d.insert(d.end(), e.begin(), e.end());
    return d;
  }
  Formula bclone() const override { return std::make_unique<XEqY>(x->bclone(), y->bclone()); }
};

struct XNeqY final : Propagator {  // propagators/cmp/x_neq_y.rs:66-104
  Var x, y;
  XNeqY(Var x_, Var y_) : x(std::move(x_)), y(std::move(y_)) {}
  SKleene is_subsumed(const VStore& s) const override {  // :71-73 — builds a temporary XEqY from bclones
    return knot(XEqY(x->bclone(), y->bclone()).is_subsumed(s));
  }
  bool propagate(VStore& s) override {  // :82-93
    Interval a = x->read(s), b = y->read(s);
    if (a.is_singleton()) return y->update(s, b.difference(a.lower()));
    if (b.is_singleton()) return x->update(s, a.difference(b.lower()));
    return true;
  }
  Deps dependencies() const override {  // :101-103
    return XEqY(x->bclone(), y->bclone()).dependencies();
  }
  Formula bclone() const override { return std::make_unique<XNeqY>(x->bclone(), y->bclone()); }
};

struct XLessY final : Propagator {  // propagators/cmp/x_less_y.rs:67-117
  Var x, y;
  XLessY(Var x_, Var y_) : x(std::move(x_)), y(std::move(y_)) {}
  SKleene is_subsumed(const VStore& s) const override {  // :73-96
    Interval a = x->read(s), b = y->read(s);
    if (a.lower() >= b.upper()) return SKleene::False;
    if (a.upper() < b.lower()) return SKleene::True;
    return SKleene::Unknown;
  }
  bool propagate(VStore& s) override {  // :104-109 (both updates computed from the pre-read values)
    Interval a = x->read(s), b = y->read(s);
    return x->update(s, a.strict_shrink_right(b.upper())) && y->update(s, b.strict_shrink_left(a.lower()));
  }
  Deps dependencies() const override {  // :112-117
    Deps d = x->dependencies(Bound), e = y->dependencies(Bound);
    d.insert(d.end(), e.begin(), e.end());
    return d;
  }
  Formula bclone() const override { return std::make_unique<XLessY>(x->bclone(), y->bclone()); }
};

struct XLessYPlusZ final : Propagator {  // propagators/cmp/x_less_y_plus_z.rs:75-128
  Var x, y, z;
  XLessYPlusZ(Var x_, Var y_, Var z_) : x(std::move(x_)), y(std::move(y_)), z(std::move(z_)) {}
  SKleene is_subsumed(const VStore& s) const override {  // :81-97
    Interval a = x->read(s), b = y->read(s), c = z->read(s);
    if ((int64_t)a.lower() >= (int64_t)b.upper() + c.upper()) return SKleene::False;
    if ((int64_t)a.upper() < (int64_t)b.lower() + c.lower()) return SKleene::True;
    return SKleene::Unknown;
  }
  bool propagate(VStore& s) override {  // :105-119
    Interval a = x->read(s), b = y->read(s), c = z->read(s);
    return x->update(s, a.strict_shrink_right((int64_t)b.upper() + c.upper())) &&
           y->update(s, b.strict_shrink_left((int64_t)a.lower() - c.upper())) &&
           z->update(s, c.strict_shrink_left((int64_t)a.lower() - b.upper()));
  }
  Deps dependencies() const override {  // :122-128
    Deps d = x->dependencies(Bound), e = y->dependencies(Bound), f = z->dependencies(Bound);
    d.insert(d.end(), e.begin(), e.end());
    d.insert(d.end(), f.begin(), f.end());
    return d;
  }
  Formula bclone() const override { return std::make_unique<XLessYPlusZ>(x->bclone(), y->bclone(), z->bclone()); }
};

struct XGreaterYPlusZ final : Propagator {  // propagators/cmp/x_greater_y_plus_z.rs:75-128
  Var x, y, z;
  XGreaterYPlusZ(Var x_, Var y_, Var z_) : x(std::move(x_)), y(std::move(y_)), z(std::move(z_)) {}
  SKleene is_subsumed(const VStore& s) const override {  // :81-98
    Interval a = x->read(s), b = y->read(s), c = z->read(s);
    if ((int64_t)a.upper() <= (int64_t)b.lower() + c.lower()) return SKleene::False;
    if ((int64_t)a.lower() > (int64_t)b.upper() + c.upper()) return SKleene::True;
    return SKleene::Unknown;
  }
  bool propagate(VStore& s) override {  // :106-118
    Interval a = x->read(s), b = y->read(s), c = z->read(s);
    return x->update(s, a.strict_shrink_left((int64_t)b.lower() + c.lower())) &&
           y->update(s, b.strict_shrink_right((int64_t)a.upper() - c.lower())) &&
           z->update(s, c.strict_shrink_right((int64_t)a.upper() - b.lower()));
  }
  Deps dependencies() const override {  // :121-128
    Deps d = x->dependencies(Bound), e = y->dependencies(Bound), f = z->dependencies(Bound);
    d.insert(d.end(), e.begin(), e.end());
    d.insert(d.end(), f.begin(), f.end());
    return d;
  }
  Formula bclone() const override { return std::make_unique<XGreaterYPlusZ>(x->bclone(), y->bclone(), z->bclone()); }
};

// cmp/mod.rs:34-86 — constructor sugar.
inline std::unique_ptr<XLessY> x_greater_y(Var x, Var y) { return std::make_unique<XLessY>(std::move(y), std::move(x)); }
inline std::unique_ptr<XLessY> x_geq_y(Var x, Var y) { return x_greater_y(std::make_unique<Addition>(std::move(x), 1), std::move(y)); }
inline std::unique_ptr<XLessY> x_leq_y(Var x, Var y) { return std::make_unique<XLessY>(std::move(x), std::make_unique<Addition>(std::move(y), 1)); }
inline std::unique_ptr<XGreaterYPlusZ> x_geq_y_plus_z(Var x, Var y, Var z) {
  return std::make_unique<XGreaterYPlusZ>(std::make_unique<Addition>(std::move(x), 1), std::move(y), std::move(z));
}
inline std::unique_ptr<XLessYPlusZ> x_leq_y_plus_z(Var x, Var y, Var z) {
  return std::make_unique<XLessYPlusZ>(std::make_unique<Addition>(std::move(x), -1), std::move(y), std::move(z));
}

struct XEqYPlusZ final : Propagator {  // propagators/cmp/x_eq_y_plus_z.rs:26-105
  std::unique_ptr<XGreaterYPlusZ> geq;
  std::unique_ptr<XLessYPlusZ> leq;
  XEqYPlusZ(Var x, Var y, Var z) {  // :36-41
    geq = x_geq_y_plus_z(x->bclone(), y->bclone(), z->bclone());
    leq = x_leq_y_plus_z(std::move(x), std::move(y), std::move(z));
  }
  XEqYPlusZ(std::unique_ptr<XGreaterYPlusZ> g, std::unique_ptr<XLessYPlusZ> l) : geq(std::move(g)), leq(std::move(l)) {}
  SKleene is_subsumed(const VStore& s) const override { return kand(geq->is_subsumed(s), leq->is_subsumed(s)); }  // :65-67
  bool propagate(VStore& s) override { return geq->propagate(s) && leq->propagate(s); }                        // :85-87
  Deps dependencies() const override {  // :96-104
    Deps g = geq->dependencies(), l = leq->dependencies();
    if (g != l) throw Panic("This function assumed both dependencies of X >= Y + Z and X <= Y + Z are equals.");
    return g;
  }
  Formula bclone() const override {
    auto g = std::unique_ptr<XGreaterYPlusZ>(static_cast<XGreaterYPlusZ*>(geq->bclone().release()));
    auto l = std::unique_ptr<XLessYPlusZ>(static_cast<XLessYPlusZ*>(leq->bclone().release()));
    return std::make_unique<XEqYPlusZ>(std::move(g), std::move(l));
  }
};

struct XEqYMulZ final : Propagator {  // propagators/cmp/x_eq_y_mul_z.rs:68-115
  Var x, y, z;
  XEqYMulZ(Var x_, Var y_, Var z_) : x(std::move(x_)), y(std::move(y_)), z(std::move(z_)) {}
  SKleene is_subsumed(const VStore& s) const override {  // :73-91
    Interval a = x->read(s), yz = y->read(s).mul(z->read(s));
    if (yz.overlap(a)) return (yz.is_singleton() && a.is_singleton()) ? SKleene::True : SKleene::Unknown;
    return SKleene::False;
  }
  bool propagate(VStore& s) override {  // :99-105
    Interval a = x->read(s), yz = y->read(s).mul(z->read(s));
    return x->update(s, a.intersection(yz));
  }
  Deps dependencies() const override {  // :108-114
    Deps d = x->dependencies(Bound), e = y->dependencies(Bound), f = z->dependencies(Bound);
    d.insert(d.end(), e.begin(), e.end());
    d.insert(d.end(), f.begin(), f.end());
    return d;
  }
  Formula bclone() const override { return std::make_unique<XEqYMulZ>(x->bclone(), y->bclone(), z->bclone()); }
};

struct Conjunction final : Propagator {  // logic/conjunction.rs:77-119
  std::vector<Formula> fs;
  explicit Conjunction(std::vector<Formula> f) : fs(std::move(f)) {}
  mutable uint64_t last_children = 0;
  SKleene is_subsumed(const VStore& s) const override {  // :78-94
    bool all_entailed = true;
    for (auto& f : fs) {
      SKleene k = f->is_subsumed(s);
      if (k == SKleene::False) return SKleene::False;
      if (k == SKleene::Unknown) all_entailed = false;
    }
    return all_entailed ? SKleene::True : SKleene::Unknown;
  }
  bool propagate(VStore& s) override {  // :97-104
    last_children = 0;
    for (auto& f : fs) { ++last_children; if (!f->propagate(s)) return false; }
    return true;
  }
  Deps dependencies() const override {  // :107-118 (sorted + dedup union)
    Deps d;
    for (auto& f : fs) { Deps x = f->dependencies(); d.insert(d.end(), x.begin(), x.end()); }
    std::sort(d.begin(), d.end());
    d.erase(std::unique(d.begin(), d.end()), d.end());
    return d;
  }
  Formula bclone() const override {
    std::vector<Formula> c;
    for (auto& f : fs) c.push_back(f->bclone());
    return std::make_unique<Conjunction>(std::move(c));
  }
  uint64_t num_elementary() const override { return last_children; }
};

struct Distinct final : Propagator {  // propagators/distinct.rs:47-126
  std::unique_ptr<Conjunction> conj;
  std::vector<Var> vars;
  explicit Distinct(std::vector<Var> v) : vars(std::move(v)) {  // :63-83
    if (vars.empty()) throw Panic("Variable array in `Distinct` must be non-empty.");
    std::vector<Formula> props;
    for (size_t i = 0; i + 1 < vars.size(); ++i)
      for (size_t j = i + 1; j < vars.size(); ++j)
        props.push_back(std::make_unique<XNeqY>(vars[i]->bclone(), vars[j]->bclone()));
    conj = std::make_unique<Conjunction>(std::move(props));
  }
  Distinct(std::unique_ptr<Conjunction> c, std::vector<Var> v) : conj(std::move(c)), vars(std::move(v)) {}
  SKleene is_subsumed(const VStore& s) const override { return conj->is_subsumed(s); }  // :105-107
  bool propagate(VStore& s) override { return conj->propagate(s); }                      // :111-113
  Deps dependencies() const override {  // :117-124 (Inner on each var, NO dedup)
    Deps d;
    for (auto& v : vars) { Deps x = v->dependencies(Inner); d.insert(d.end(), x.begin(), x.end()); }
    return d;
  }
  Formula bclone() const override {
    std::vector<Var> c;
    for (auto& v : vars) c.push_back(v->bclone());
    auto cj = std::unique_ptr<Conjunction>(static_cast<Conjunction*>(conj->bclone().release()));
    return std::make_unique<Distinct>(std::move(cj), std::move(c));
  }
  uint64_t num_elementary() const override { return conj->num_elementary(); }
};

// ---------------------------------------------------------------------------------------------------
// Growable bitset — crate bit-set ^0.5.3 (iter yields ascending indices).
// ---------------------------------------------------------------------------------------------------
struct BitSet {
  std::vector<uint64_t> w;
  bool contains(size_t i) const { return (i >> 6) < w.size() && ((w[i >> 6] >> (i & 63)) & 1); }
  void insert(size_t i) { if ((i >> 6) >= w.size()) w.resize((i >> 6) + 1, 0); w[i >> 6] |= 1ull << (i & 63); }
  void remove(size_t i) { if ((i >> 6) < w.size()) w[i >> 6] &= ~(1ull << (i & 63)); }
  template <class F> void for_each(F&& f) const {
    for (size_t k = 0; k < w.size(); ++k) {
      uint64_t x = w[k];
      while (x) { unsigned b = __builtin_ctzll(x); f(k * 64 + b); x &= x - 1; }
    }
  }
};

// propagation/reactors/indexed_deps.rs:23-121.  `check_dup` = the release-mode assert! in subscribe (:69-77).
struct IndexedDeps {
  size_t num_events = 0, num_subscriptions = 0;
  std::vector<std::vector<size_t>> deps;
  bool check_dup = true;
  IndexedDeps() = default;
  IndexedDeps(size_t num_vars, size_t num_events_, bool check) : num_events(num_events_), deps(num_vars * num_events_), check_dup(check) {}  // :57-63
  size_t num_vars() const { return num_events ? deps.size() / num_events : 0; }
  void assert_var_idx(size_t var, const char* op) const {  // :49-53
    if (var >= num_vars()) throw Panic(std::string("Reactor IndexedDeps: bad variable index in ") + op);
  }
  void subscribe(size_t var, FDEvent ev, size_t prop) {  // :65-82
    if (check_dup) {
      // skip(var*num_events).take(num_events) tolerates var out of range (empty iteration), like the Rust iterator chain
      for (size_t e = 0; e < num_events && var * num_events + e < deps.size(); ++e)
        for (size_t x : deps[var * num_events + e])
          if (x == prop) throw Panic("propagator already subscribed to this variable");
    }
    assert_var_idx(var, "subscription");
    ++num_subscriptions;
    deps[num_events * var + ev].push_back(prop);
  }
  void unsubscribe(size_t var, FDEvent ev, size_t prop) {  // :84-97
    assert_var_idx(var, "unsubscription");
    --num_subscriptions;
    auto& props = deps[num_events * var + ev];
    auto it = std::find(props.begin(), props.end(), prop);
    if (it == props.end()) throw Panic("cannot unsubscribe propagator not registered.");
    *it = props.back();  // Vec::swap_remove
    props.pop_back();
  }
  std::vector<size_t> react(size_t var, FDEvent ev) const {  // :99-113 — a fresh Vec per call, lists ev..=Inner
    assert_var_idx(var, "react");
    std::vector<size_t> out;
    for (size_t e = ev; e < num_events; ++e) {
      const auto& l = deps[num_events * var + e];
      out.insert(out.end(), l.begin(), l.end());
    }
    return out;
  }
  size_t size() const { return num_subscriptions; }  // :116-121
  bool is_empty() const { return num_subscriptions == 0; }
};

// propagation/schedulers/relaxed_fifo.rs:27-71.
struct RelaxedFifo {
  BitSet inside_queue;
  std::deque<size_t> queue;
  size_t capacity = 0;
  RelaxedFifo() = default;
  explicit RelaxedFifo(size_t cap) : capacity(cap) {}
  void schedule(size_t idx) {  // :42-48
    if (idx >= capacity) throw Panic("RelaxedFifo::schedule out of bounds");
    if (!inside_queue.contains(idx)) { inside_queue.insert(idx); queue.push_back(idx); }
  }
  void unschedule(size_t idx) {  // :50-58 (VecDeque::swap_remove_front: swap with the front, pop front)
    if (idx >= capacity) throw Panic("RelaxedFifo::unschedule out of bounds");
    if (inside_queue.contains(idx)) {
      auto it = std::find(queue.begin(), queue.end(), idx);
      if (it == queue.end()) throw Panic("RelaxedFifo: inside_queue out of sync");
      std::swap(*it, queue.front());
      queue.pop_front();
      inside_queue.remove(idx);
    }
  }
  std::optional<size_t> pop() {  // :60-66
    if (queue.empty()) return std::nullopt;
    size_t r = queue.front();
    queue.pop_front();
    inside_queue.remove(r);
    return r;
  }
  bool is_empty() const { return queue.empty(); }  // :68-70
};

// Counters the reference does not keep (SURVEY §5 "metrics"): filter steps etc.
struct Stats {
  uint64_t steps = 0;         // elementary propagate()+is_subsumed() evaluations (Conjunction pop = children run)
  uint64_t pops = 0;          // scheduler pops
  uint64_t narrowings = 0;    // pops that changed at least one domain (has_changed)
  uint64_t nodes = 0, failed_nodes = 0;
  uint64_t subscriptions = 0; // Σ subscribe() calls in prepare()
};

// ---------------------------------------------------------------------------------------------------
// propagation::store::Store — propagation/store.rs:32-324.
// ---------------------------------------------------------------------------------------------------
struct CStore {
  std::vector<Formula> propagators;
  BitSet active;
  IndexedDeps reactor;
  RelaxedFifo scheduler;
  bool check_dup = true;  // false = the `restatement-noassert` baseline
  Stats* stats = nullptr;

  size_t alloc(Formula p) {  // :223-230
    size_t idx = propagators.size();
    propagators.push_back(std::move(p));
    active.insert(idx);
    return idx;
  }
  size_t size() const { return propagators.size(); }

  void init_reactor(const VStore& vs) {  // :130-142
    reactor = IndexedDeps(vs.size(), kNumEvents, check_dup);
    active.for_each([&](size_t p) {
      Deps d = propagators[p]->dependencies();
      for (auto& [v, ev] : d) {
        reactor.subscribe(v, ev, p);
        if (stats) ++stats->subscriptions;
      }
    });
  }
  void init_scheduler() {  // :144-149
    scheduler = RelaxedFifo(propagators.size());
    active.for_each([&](size_t p) { scheduler.schedule(p); });
  }
  void prepare(const VStore& vs) { init_reactor(vs); init_scheduler(); }  // :125-128

  SKleene propagator_consistency(size_t p, VStore& vs) {  // :177-183
    if (propagators[p]->propagate(vs)) return propagators[p]->is_subsumed(vs);
    return SKleene::False;
  }
  void unlink_prop(size_t p) {  // :200-207
    active.remove(p);
    scheduler.unschedule(p);
    Deps d = propagators[p]->dependencies();
    for (auto& [v, ev] : d) reactor.unsubscribe(v, ev, p);
  }
  void reschedule_prop(size_t p, VStore& vs) {  // :185-189
    if (vs.has_changed()) scheduler.schedule(p);
  }
  bool propagate_one(size_t p, VStore& vs) {  // :166-175
    vs.reset_changed();
    SKleene s = propagator_consistency(p, vs);
    if (stats) {
      ++stats->pops;
      stats->steps += propagators[p]->num_elementary();
      if (vs.has_changed()) ++stats->narrowings;
    }
    if (s == SKleene::False) return false;
    if (s == SKleene::True) unlink_prop(p); else reschedule_prop(p, vs);
    return true;
  }
  void react(VStore& vs) {  // :191-198
    for (auto& [v, ev] : vs.drain_delta()) {
      std::vector<size_t> reactions = reactor.react(v, ev);
      for (size_t p : reactions) scheduler.schedule(p);
    }
  }
  bool propagation_loop(VStore& vs) {  // :151-164
    bool consistent = true;
    while (!scheduler.is_empty() && consistent) {
      while (auto p = scheduler.pop()) {
        if (!propagate_one(*p, vs)) { consistent = false; break; }
        react(vs);
      }
    }
    return consistent;
  }
  SKleene consistency(VStore& vs) {  // :247-257
    prepare(vs);
    bool consistent = propagation_loop(vs);
    if (stats) { ++stats->nodes; if (!consistent) ++stats->failed_nodes; }
    if (!consistent) return SKleene::False;
    if (reactor.is_empty()) return SKleene::True;
    return SKleene::Unknown;
  }
  SKleene is_subsumed(const VStore& vs) const {  // :232-238
    SKleene x = SKleene::True;
    for (auto& p : propagators) x = kand(x, p->is_subsumed(vs));
    return x;
  }
  // Snapshot label/restore — :306-324
  using Label = std::pair<size_t, BitSet>;
  Label label() const { return {propagators.size(), active}; }
  void restore(const Label& l) { propagators.resize(l.first); active = l.second; }
};

// ---------------------------------------------------------------------------------------------------
// Search (caller side of the path): search/space.rs:21-44, search/propagation.rs:42-55,
// search/branching/{first_smallest_var.rs:30-39, middle_val.rs:25-27, binary_split.rs:33-60, brancher.rs:52-71,
// branch.rs:36-55}, search/engine/one_solution.rs:46-105, all_solution.rs:37-47, stop_node.rs:47-62,
// statistics via monitor.rs:19-68.  Restoration = whole-store copies (CopyMemory semantics,
// variable/memory/copy_memory.rs:125-151) — the trail is an optimisation with identical observable state.
// ---------------------------------------------------------------------------------------------------
struct Space {
  VStore vstore;
  CStore cstore;
  SKleene consistency() { return cstore.consistency(vstore); }  // search/space.rs:41-43
};

inline size_t first_smallest_var(const VStore& vs) {  // first_smallest_var.rs:30-39 (min_by_key keeps the FIRST minimum)
  size_t best = SIZE_MAX; uint32_t best_size = 0;
  for (size_t i = 0; i < vs.size(); ++i) {
    uint32_t sz = vs.memory[i].size();
    if (sz > 1 && (best == SIZE_MAX || sz < best_size)) { best = i; best_size = sz; }
  }
  if (best == SIZE_MAX) throw Panic("Cannot select a variable in a space where all variables are assigned.");
  return best;
}
inline int32_t middle_val(const Interval& d) {  // middle_val.rs:25-27 (Rust `/` truncates toward zero, as C++)
  return (int32_t)(((int64_t)d.lower() + d.upper()) / 2);
}

struct SearchStats { uint64_t num_solution = 0, num_failed_node = 0, num_prune = 0, num_nodes = 0; bool end_of_search = false; };

// One DFS over `root` with OneSolution<Propagation<Brancher<FirstSmallestVar,MiddleVal,BinarySplit>>, VectorStack>
// wrapped in AllSolution when `all_solutions` and StopNode(node_limit) when node_limit > 0 (search/mod.rs:45-52).
// `on_node(space_before_domains, space_after, status)` is called once per explored node.
struct Branch {
  std::vector<Interval> vlabel;  // vstore label (copy)
  CStore::Label clabel;          // cstore label (len, active)
  size_t var; int32_t val; bool left;
};

template <class OnNode>
inline SearchStats dfs(Space& root, bool all_solutions, uint64_t node_limit, OnNode&& on_node) {
  SearchStats st;
  std::vector<Branch> stack;  // VectorStack: LIFO
  bool first = true;
  // frozen "immutable_state" = current space contents; commit = restore label + run the alternative
  while (true) {
    if (!first) {
      if (stack.empty()) break;
      Branch b = std::move(stack.back());
      stack.pop_back();
      // Branch::commit (branch.rs:51-55): restore, then add the branch propagator.
      root.vstore.memory = b.vlabel;
      root.vstore.delta.assign(root.vstore.memory.size(), -1);  // Store::from_memory: fresh delta (variable/store.rs:68-76)
      root.vstore.delta_keys.clear();
      root.vstore.has_changed_ = false;
      root.cstore.restore(b.clabel);
      Var x = std::make_unique<Identity>(b.var);
      Var v = std::make_unique<Constant>(b.val);
      if (b.left) root.cstore.alloc(x_leq_y(std::move(x), std::move(v)));      // binary_split.rs:46-51
      else root.cstore.alloc(x_greater_y(std::move(x), std::move(v)));         // binary_split.rs:52-57
    }
    first = false;
    std::vector<Interval> before = root.vstore.memory;
    BitSet active_before = root.cstore.active;
    SKleene k = root.consistency();  // Propagation::enter, search/propagation.rs:49
    ++st.num_nodes;
    on_node(before, active_before, root, k);
    // StopNode replaces the status by EndOfSearch once the limit is reached (stop_node.rs:57-62), so the
    // monitor dispatches on_end_of_search for that node and neither a solution nor a failure is counted.
    if (node_limit && st.num_nodes >= node_limit) { st.end_of_search = true; return st; }
    if (k == SKleene::True) {
      ++st.num_solution;
      if (!all_solutions) return st;  // OneSolution::enter returns Satisfiable (one_solution.rs:98-103)
    } else if (k == SKleene::False) {
      ++st.num_failed_node;
    } else {
      // Brancher::enter (brancher.rs:52-71) + BinarySplit::distribute + Branch::distribute (label AFTER consistency).
      size_t var = first_smallest_var(root.vstore);
      Interval dom = root.vstore.memory[var];
      if (dom.is_singleton() || dom.is_empty()) throw Panic("Can not distribute over assigned or failed variables.");
      int32_t val = middle_val(dom);
      CStore::Label cl = root.cstore.label();
      // push_branches reversed so that the left branch is explored first (one_solution.rs:46-51)
      stack.push_back(Branch{root.vstore.memory, cl, var, val, false});
      stack.push_back(Branch{root.vstore.memory, cl, var, val, true});
    }
  }
  st.end_of_search = true;
  return st;
}

}  // namespace orc
