// nqueens.cpp — the reference's example/src/nqueens.rs:28-74 written against pcp_host.hpp: the model code is the
// reference's, line for line (variables in [1,n]; for i<j  q_i != q_j + (j-i),  q_i != q_j - (j-i); join_distinct),
// the propagation fixpoint of every search node runs on the MI355X through libpcp_hip.so.
//
//   nqueens <n>                 first solution with the default engine (one_solution_engine, search/mod.rs:45-52)
//   nqueens <n> all [limit]     all solutions (AllSolution), optional StopNode(limit)
// Prints one JSON line: {"n":..,"status":..,"solutions":..,"nodes":..,"failed":..,"filter_steps":..,"first":[..]}
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../pcp_host.hpp"

using namespace pcp_host;

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8;
  const bool all = argc > 2 && !strcmp(argv[2], "all");
  const uint64_t limit = argc > 3 ? strtoull(argv[3], nullptr, 10) : 0;
  try {
    Space space(0);
    std::vector<Var> queens;
    // 2 queens can't share the same line.
    for (int i = 0; i < n; ++i) queens.push_back(space.vstore.alloc(Interval(1, n)));
    for (int i = 0; i + 1 < n; ++i) {
      for (int j = i + 1; j < n; ++j) {
        // 2 queens can't share the same diagonal.
        const int q1 = i + 1, q2 = j + 1;
        // Xi + i != Xj + j reformulated as: Xi != Xj + j - i
        space.cstore.alloc(XNeqY(queens[i], addition(queens[j], q2 - q1)));
        // Xi - i != Xj - j reformulated as: Xi != Xj - j + i
        space.cstore.alloc(XNeqY(queens[i], addition(queens[j], -q2 + q1)));
      }
    }
    // 2 queens can't share the same column.
    if (n > 0) join_distinct(space.vstore, space.cstore, queens);

    Statistics st;
    std::vector<int32_t> first;
    uint64_t steps = 0;
    Status status = search(space, all, limit, st, [&](const Space& s) {
      if (first.empty())
        for (size_t i = 0; i < s.vstore.size(); ++i) first.push_back(s.vstore[i].lower());  // dom.lower() == dom.upper()
    });
    steps = space.cstore.last_stats().steps;
    const char* name = status == Status::Satisfiable ? "Satisfiable" : status == Status::Unsatisfiable ? "Unsatisfiable" : "EndOfSearch";
    printf("{\"n\": %d, \"status\": \"%s\", \"solutions\": %llu, \"nodes\": %llu, \"failed\": %llu, \"last_node_filter_steps\": %llu, \"first\": [",
           n, name, (unsigned long long)st.num_solution, (unsigned long long)st.num_nodes, (unsigned long long)st.num_failed_node,
           (unsigned long long)steps);
    for (size_t i = 0; i < first.size(); ++i) printf("%s%d", i ? ", " : "", first[i]);
    printf("]}\n");
  } catch (const std::exception& e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 2;
  }
  return 0;
}
