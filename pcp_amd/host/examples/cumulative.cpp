// cumulative.cpp — CumulativeTest::instantiate + Cumulative::join + consistency() (propagators/cumulative.rs:201-233, 59-114, 236-252) written
// against pcp_host.hpp: the reified layer (Boolean, Conjunction, Disjunction, implication, equivalence), XEqYMulZ and Sum views of the C++ host
// twin, with the fixpoint on the MI355X through pcp_model_push_formula / pcp_propagate.
//
//   cumulative <constant 0|1> <tasks>  s_lb s_ub ... (tasks pairs)  d_lb d_ub ...  r_lb r_ub ...  c_lb c_ub
//   cumulative logic-test      implication / equivalence / not_ on two variables, statuses and bounds
// `constant` = singleton domains become Constant views instead of variables (cumulative.rs:214-222).
// Prints one JSON line: {"status": 0|1|2, "units": .., "vars": .., "lb": [..], "ub": [..]}
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../pcp_host.hpp"

using namespace pcp_host;

static void print_space(const char* head, SKleene k, const Space& space) {
  printf("%s{\"status\": %d, \"units\": %zu, \"vars\": %zu, \"lb\": [", head, (int)k, space.cstore.size(), space.vstore.size());
  for (size_t i = 0; i < space.vstore.size(); ++i) printf("%s%d", i ? ", " : "", space.vstore[i].lower());
  printf("], \"ub\": [");
  for (size_t i = 0; i < space.vstore.size(); ++i) printf("%s%d", i ? ", " : "", space.vstore[i].upper());
  printf("]}");
}

// x in [0,9], y in [0,9], b in [0,1]:  b <=> x < y, then b = 1 (the shape of a branch constraint) narrows x and y through the
// equivalence; then, on a fresh space, implication(x < y, x + 3 < y) = Disjunction[x < y, y < x + 4] — entailed by its second
// child on these domains — and not_(x >= y) = x < y on top of it.
static int logic_test() {
  printf("[");
  {
    Space space(0);
    Var x = space.vstore.alloc(Interval(0, 9)), y = space.vstore.alloc(Interval(0, 9)), b = space.vstore.alloc(Interval(0, 1));
    space.cstore.alloc(equivalence(Boolean(b), XLessY(x, y)));
    print_space("", space.consistency(), space);
    space.cstore.alloc(XEqY(b, constant(1)));
    print_space(", ", space.consistency(), space);
  }
  {
    Space space(0);
    Var x = space.vstore.alloc(Interval(5, 9)), y = space.vstore.alloc(Interval(0, 7));
    space.cstore.alloc(implication(XLessY(x, y), XLessY(addition(x, 3), y)));  // Disjunction[x < y, not(x + 3 < y)]
    print_space(", ", space.consistency(), space);
    space.cstore.alloc(not_(x_geq_y(x, y)));  // = x + 1 < y + 1
    print_space(", ", space.consistency(), space);
  }
  printf("]\n");
  return 0;
}

int main(int argc, char** argv) {
  try {
    if (argc > 1 && !strcmp(argv[1], "logic-test")) return logic_test();
    if (argc < 3) { fprintf(stderr, "usage: cumulative <constant> <tasks> bounds...\n"); return 2; }
    const bool constant_views = atoi(argv[1]) != 0;
    const int tasks = atoi(argv[2]);
    if (argc != 3 + 6 * tasks + 2) { fprintf(stderr, "expected %d bounds\n", 6 * tasks + 2); return 2; }
    int at = 3;
    Space space(0);
    auto mk = [&]() -> Var {
      const int lb = atoi(argv[at]), ub = atoi(argv[at + 1]);
      at += 2;
      if (constant_views && lb == ub) return constant(lb);
      return space.vstore.alloc(Interval(lb, ub));
    };
    std::vector<Var> starts, durations, resources;
    for (int i = 0; i < tasks; ++i) starts.push_back(mk());
    for (int i = 0; i < tasks; ++i) durations.push_back(mk());
    for (int i = 0; i < tasks; ++i) resources.push_back(mk());
    const int clb = atoi(argv[at]), cub = atoi(argv[at + 1]);
    Var capacity = space.vstore.alloc(Interval(clb, cub));
    Cumulative cumulative(starts, durations, resources, capacity);
    cumulative.join(space.vstore, space.cstore);
    print_space("", space.consistency(), space);
    printf("\n");
  } catch (const std::exception& e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 2;
  }
  return 0;
}
