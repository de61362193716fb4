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
// negative operands.  SET MODE (IntervalSet<i32>, struct IntervalSet below; the engine is generic in the domain type:
// pcp_oracle_engine.inc is instantiated for Interval and for IntervalSet, like the reference's Store<Memory, Event>) is pinned
// ONLY at the search level — the reference's FDSpace tests (all_solution.rs:70, one_solution.rs:121-128, stop_node.rs:83-104) —
// because no reference test observes an IntervalSet after propagation: "parity unpinned" for set-mode DOMAINS, stated in DESIGN.md §7.
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

struct SearchStats { uint64_t num_solution = 0, num_failed_node = 0, num_prune = 0, num_nodes = 0; bool end_of_search = false; };

// ---------------------------------------------------------------------------------------------------
// IntervalSet<i32> — crate intervallum (interval_set.rs; not in the tree, "^1.2.0", no lockfile): a set of integers kept as
// sorted, pairwise disjoint, NON-ADJACENT intervals plus its cardinality.  Restated from the crate's published behaviour
// [3P-unverified]: every operation below is the SET-theoretic one — in particular `difference(&v)` removes an interior
// value (splitting an interval), which is what makes XNeqY raise `Inner` events on VStoreSet (the FDSpace default,
// variable/mod.rs:38, search/mod.rs:41-43; example/src/nqueens.rs:34 allocates IntervalSet::new(1, n)).
// PARITY: the REFERENCE pins this mode only at the search level (solution counts all_solution.rs:70, first-solution statuses
// one_solution.rs:121-128, StopNode stop_node.rs:83-104): no reference test observes an IntervalSet domain after propagation,
// and the crate is not in the tree.  Below that level the pin is an independent restatement: tests/test_intervalset_model.py
// re-derives the algebra and the propagation fixpoint over plain Python sets from the reference's propagator sources and agrees
// with this file on every random case (set algebra operation by operation, consistency over six propagator kinds).
// ---------------------------------------------------------------------------------------------------
struct IntervalSet {
  std::vector<Interval> iv;  // sorted by lb; iv[i].ub + 1 < iv[i+1].lb
  static IntervalSet empty() { return IntervalSet{}; }
  static IntervalSet singleton(int32_t v) { return from_interval(v, v); }
  static IntervalSet from_interval(int64_t l, int64_t u) {
    IntervalSet s;
    if (l <= u) s.iv.push_back(Interval::make(l, u));
    return s;
  }
  static IntervalSet make(int64_t l, int64_t u) { return from_interval(l, u); }
  bool is_empty() const { return iv.empty(); }
  uint32_t size() const { uint64_t n = 0; for (auto& i : iv) n += i.size(); return (uint32_t)n; }
  bool is_singleton() const { return iv.size() == 1 && iv[0].is_singleton(); }
  int32_t lower() const { if (iv.empty()) throw Panic("lower() of an empty IntervalSet"); return iv.front().lb; }
  int32_t upper() const { if (iv.empty()) throw Panic("upper() of an empty IntervalSet"); return iv.back().ub; }
  bool contains(int32_t v) const { for (auto& i : iv) if (i.contains(v)) return true; return false; }
  IntervalSet intersection(const IntervalSet& b) const {
    IntervalSet r;
    size_t i = 0, j = 0;
    while (i < iv.size() && j < b.iv.size()) {
      Interval x = iv[i].intersection(b.iv[j]);
      if (!x.is_empty()) r.iv.push_back(x);
      if (iv[i].ub < b.iv[j].ub) ++i; else ++j;
    }
    return r;
  }
  bool is_subset(const IntervalSet& b) const { return intersection(b).size() == size(); }
  bool is_disjoint(const IntervalSet& b) const { return intersection(b).is_empty(); }
  bool overlap(const IntervalSet& b) const { return !is_disjoint(b); }
  IntervalSet difference(int32_t v) const {  // Difference<Bound>: remove ONE value, wherever it sits
    IntervalSet r;
    for (auto& i : iv) {
      if (!i.contains(v)) { r.iv.push_back(i); continue; }
      if (i.lb < v) r.iv.push_back(Interval{i.lb, v - 1});
      if (v < i.ub) r.iv.push_back(Interval{v + 1, i.ub});
    }
    return r;
  }
  IntervalSet shrink_left(int64_t b) const {  // keep the values >= b
    IntervalSet r;
    for (auto& i : iv) { if (i.ub < b) continue; r.iv.push_back(i.lb >= b ? i : Interval::make(b, i.ub)); }
    return r;
  }
  IntervalSet shrink_right(int64_t b) const {  // keep the values <= b
    IntervalSet r;
    for (auto& i : iv) { if (i.lb > b) break; r.iv.push_back(i.ub <= b ? i : Interval::make(i.lb, b)); }
    return r;
  }
  IntervalSet strict_shrink_left(int64_t b) const { return shrink_left(b + 1); }
  IntervalSet strict_shrink_right(int64_t b) const { return shrink_right(b - 1); }
  IntervalSet add(int32_t v) const { IntervalSet r; for (auto& i : iv) r.iv.push_back(i.add(v)); return r; }
  IntervalSet sub(int32_t v) const { IntervalSet r; for (auto& i : iv) r.iv.push_back(i.sub(v)); return r; }
  IntervalSet add(const IntervalSet& b) const {  // { x + y }: union of the pairwise interval sums, normalised
    std::vector<Interval> all;
    for (auto& i : iv) for (auto& j : b.iv) all.push_back(i.add(j));
    std::sort(all.begin(), all.end(), [](const Interval& p, const Interval& q) { return p.lb < q.lb; });
    IntervalSet r;
    for (auto& x : all) {
      if (!r.iv.empty() && (int64_t)x.lb <= (int64_t)r.iv.back().ub + 1) r.iv.back().ub = std::max(r.iv.back().ub, x.ub);
      else r.iv.push_back(x);
    }
    return r;
  }
  IntervalSet mul(const IntervalSet&) const { throw Panic("IntervalSet * IntervalSet is not restated (XEqYMulZ is interval-mode only)"); }
  bool operator==(const IntervalSet& o) const {
    if (iv.size() != o.iv.size()) return false;
    for (size_t k = 0; k < iv.size(); ++k) if (!(iv[k].lb == o.iv[k].lb && iv[k].ub == o.iv[k].ub)) return false;
    return true;
  }
};

// The domain-generic engine (events, variable store, views, propagators, constraint store, search), once per domain type.
#define ORC_DOM Interval
namespace fd {
#include "pcp_oracle_engine.inc"
}  // namespace fd
#undef ORC_DOM
#define ORC_DOM IntervalSet
namespace fdset {
#include "pcp_oracle_engine.inc"
}  // namespace fdset
#undef ORC_DOM
using namespace fd;  // the unqualified names are the Interval<i32> engine (VStoreFD)

}  // namespace orc
