// nqueens.cpp — the model of the reference's example (example/src/nqueens.rs:28-74) written against pcp_host.hpp with the
// reference's constructor names (variables in [1,n]; for i<j  q_i != q_j + (j-i),  q_i != q_j - (j-i); join_distinct); the
// propagation fixpoint of every search node runs on the MI355X through libpcp_hip.so.  NOTE: this host mirror allocates
// Interval<i32> domains (VStoreFD); the reference's example runs FDSpace = IntervalSet domains (nqueens.rs:34), which the
// engine offers as set mode (pcp_model_reset(set_words > 0)) — same solutions, a different (smaller) search tree.
//
//   nqueens <n>                 first solution with the default engine (one_solution_engine, search/mod.rs:45-52)
//   nqueens <n> all [limit]     all solutions (AllSolution), optional StopNode(limit)
//   nqueens restore-test        label / alloc(x < y) / consistency / restore / alloc(x > y) / consistency on one GpuCStore
//   nqueens resident <n> [all [limit]]   the same searches over ResidentGpuCStore (pcp_host_resident.hpp: the node stays in HBM between
//                               consistency() calls); adds "pcie_in", "pcie_out", "pcie_whole_node" (bytes) to the JSON line
// Prints one JSON line: {"n":..,"status":..,"solutions":..,"nodes":..,"failed":..,"filter_steps":..,"first":[..]}
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../pcp_host_resident.hpp"

using namespace pcp_host;

// FrozenStore::restore followed by a different propagator at the same index (propagation/store.rs:306-324): the device model
// must follow the host's truncation.
static int restore_test() {
  Space space(0);
  Var x = space.vstore.alloc(Interval(0, 9)), y = space.vstore.alloc(Interval(0, 9));
  const std::vector<int32_t> lb0 = space.vstore.lbs(), ub0 = space.vstore.ubs();
  const GpuCStore::Label l = space.cstore.label();
  space.cstore.alloc(XLessY(x, y));
  space.consistency();
  printf("{\"first\": [[%d, %d], [%d, %d]], ", space.vstore[0].lower(), space.vstore[0].upper(), space.vstore[1].lower(), space.vstore[1].upper());
  space.vstore.lbs() = lb0; space.vstore.ubs() = ub0;
  space.cstore.restore(l);
  space.cstore.alloc(x_greater_y(x, y));
  space.consistency();
  printf("\"second\": [[%d, %d], [%d, %d]]}\n", space.vstore[0].lower(), space.vstore[0].upper(), space.vstore[1].lower(), space.vstore[1].upper());
  return 0;
}

static uint64_t pcie_in(const GpuCStore&) { return 0; }
static uint64_t pcie_in(const ResidentGpuCStore& c) { return c.bytes_in(); }
static uint64_t pcie_out(const GpuCStore&) { return 0; }
static uint64_t pcie_out(const ResidentGpuCStore& c) { return c.bytes_out(); }
static uint64_t pcie_whole(const GpuCStore&) { return 0; }
static uint64_t pcie_whole(const ResidentGpuCStore& c) { return c.bytes_whole_node(); }

template <class SpaceT>
static int run(int n, bool all, uint64_t limit) {
  SpaceT space(0);
  std::vector<Var> queens;
  for (int i = 0; i < n; ++i) queens.push_back(space.vstore.alloc(Interval(1, n)));
  for (int i = 0; i + 1 < n; ++i) {
    for (int j = i + 1; j < n; ++j) {
      const int q1 = i + 1, q2 = j + 1;  // the two diagonals (nqueens.rs:37-47)
      space.cstore.alloc(XNeqY(queens[i], addition(queens[j], q2 - q1)));
      space.cstore.alloc(XNeqY(queens[i], addition(queens[j], -q2 + q1)));
    }
  }
  if (n > 0) join_distinct(space.vstore, space.cstore, queens);  // the columns (nqueens.rs:50)

  Statistics st;
  std::vector<int32_t> first;
  Status status = search(space, all, limit, st, [&](const SpaceT& s) {
    if (first.empty())
      for (size_t i = 0; i < s.vstore.size(); ++i) first.push_back(s.vstore[i].lower());  // dom.lower() == dom.upper()
  });
  const uint64_t steps = space.cstore.last_stats().steps;
  const char* name = status == Status::Satisfiable ? "Satisfiable" : status == Status::Unsatisfiable ? "Unsatisfiable" : "EndOfSearch";
  printf("{\"n\": %d, \"status\": \"%s\", \"solutions\": %llu, \"nodes\": %llu, \"failed\": %llu, \"last_node_filter_steps\": %llu, "
         "\"pcie_in\": %llu, \"pcie_out\": %llu, \"pcie_whole_node\": %llu, \"first\": [",
         n, name, (unsigned long long)st.num_solution, (unsigned long long)st.num_nodes, (unsigned long long)st.num_failed_node,
         (unsigned long long)steps, (unsigned long long)pcie_in(space.cstore), (unsigned long long)pcie_out(space.cstore),
         (unsigned long long)pcie_whole(space.cstore));
  for (size_t i = 0; i < first.size(); ++i) printf("%s%d", i ? ", " : "", first[i]);
  printf("]}\n");
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "restore-test")) {
    try { return restore_test(); } catch (const std::exception& e) { fprintf(stderr, "error: %s\n", e.what()); return 2; }
  }
  const bool resident = argc > 1 && !strcmp(argv[1], "resident");
  if (resident) { --argc; ++argv; }
  const int n = argc > 1 ? atoi(argv[1]) : 8;
  const bool all = argc > 2 && !strcmp(argv[2], "all");
  const uint64_t limit = argc > 3 ? strtoull(argv[3], nullptr, 10) : 0;
  try {
    return resident ? run<ResidentSpace>(n, all, limit) : run<Space>(n, all, limit);
  } catch (const std::exception& e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 2;
  }
}
