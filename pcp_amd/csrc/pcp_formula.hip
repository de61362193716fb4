// pcp_formula.hip — the propagation fixpoint of stores that hold FORMULA propagators (the reified layer, logic/): units that are
// trees of Conjunction (logic/conjunction.rs:77-119) and Disjunction (logic/disjunction.rs:78-141) nodes over elementary leaves,
// Boolean / BooleanNeg (logic/boolean.rs:111-140, logic/boolean_neg.rs:71-96) among them — what implication / equivalence
// (logic/mod.rs:30-45) and propagators::Cumulative::join (propagators/cumulative.rs:59-114) allocate.  gfx950.
//
// Same contract as the other fixpoint kernels: Store::consistency (propagation/store.rs:125-164, 247-257) for a batch of nodes,
// explicit `active` rows (one bit per UNIT) or implicit-active nodes.  What is new is that a pop of a unit is not a filter but a
// small program: Disjunction::propagate first asks every child for is_subsumed() and propagates a child only when it is the
// single one left that is not disentailed (unit propagation, disjunction.rs:97-117); Conjunction::propagate runs its children in
// order.  Every unit of such a store — formula or not — is held as a tree here (a standalone propagator is a tree of one leaf,
// a Conjunction / Distinct group an AND over its members), and ONE LANE evaluates one unit.
//
// MI355X mapping: one workgroup of 256 threads per node; the node's domains in LDS as (-lb, ub) cells (LdsDom, narrowing =
// ds_min); a round = every live unit once, lane-strided; rounds until a round narrows nothing (Jacobi waves: the filters are
// monotone and contracting, is_subsumed() only moves from Unknown to True or False as domains shrink, so a decision taken on a
// stale domain is a conservative one that the next round repeats — same fixpoint as the reference's FIFO).  An
// entailed unit is unlinked (store.rs:200-207) — in implicit mode too: its propagate() is a no-op from then on.
// These stores are small and branchy (15 units for three tasks): the kernel is written for correctness and occupancy of the
// chip by nodes, not for bandwidth; no MFMA, no bulk tests.
#include <algorithm>

#include "pcp_device.hpp"
#include "pcp_neq.h"

namespace pcp {

namespace {

enum { F_FAIL = 0, F_OOB = 1, F_CHANGED = 2, F_STEPS2 = 4, F_STEPS3 = 6, F_NARROW = 8, F_WAVES = 9, F_WORDS = 10 };

struct FormCarve {
  size_t dom, chg, live, misc, total;
};
__host__ __device__ inline FormCarve form_carve(uint32_t S, uint32_t U) {
  auto up = [](size_t x) { return (x + 15) & ~(size_t)15; };
  FormCarve c;
  size_t o = 0;
  c.dom = o; o = up(o + (size_t)S * 8);
  c.chg = o; o = up(o + (size_t)((S + 31) / 32) * 4);
  c.live = o; o = up(o + (size_t)((U + 31) / 32) * 4);
  c.misc = o; o = up(o + 16 * 4);
  c.total = o;
  return c;
}

// is_subsumed() of one elementary propagator as SKleene (0 False, 1 True, 2 Unknown), on Interval<i32> domains:
//   XEqY x_eq_y.rs:73-94 | XNeqY x_neq_y.rs:71-73 (not XEqY) | XLessY x_less_y.rs:73-95 | XLessYPlusZ x_less_y_plus_z.rs:82-97 |
//   XGreaterYPlusZ x_greater_y_plus_z.rs:82-98 | XEqYPlusZ x_eq_y_plus_z.rs:60-67 (Kleene and of its halves, cmp/mod.rs:62-86) |
//   XEqYMulZ x_eq_y_mul_z.rs:73-91 | Boolean boolean.rs:111-127 | BooleanNeg boolean_neg.rs:71-79.
__device__ __forceinline__ uint32_t kleene_and(uint32_t a, uint32_t b) { return (a == 0u || b == 0u) ? 0u : ((a == 1u && b == 1u) ? 1u : 2u); }
__device__ __forceinline__ uint32_t kleene_not(uint32_t a) { return a == 2u ? 2u : 1u - a; }

__device__ uint32_t rec_subsumed(const Rec& rec, const LdsDom& dm) {
  const uint32_t kind = rec.xk >> 28, x = rec.xk & kSlotMask;
  const long long d = rec.d;
  const int2 X = dm.load(x);
  if (kind == PCP_BOOL || kind == PCP_NBOOL) {  // the view is x + d; Boolean: singleton ? (value == 1) : Unknown
    uint32_t b = 2u;
    if (X.x == X.y) b = ((long long)X.x + d == 1) ? 1u : 0u;
    return kind == PCP_BOOL ? b : kleene_not(b);
  }
  const int2 Y = dm.load(rec.y);
  if (kind <= PCP_LT) {
    const long long Yl = (long long)Y.x + d, Yu = (long long)Y.y + d;
    if (kind == PCP_LT) return X.x >= Yu ? 0u : (X.y < Yl ? 1u : 2u);
    const uint32_t eq = (X.x == Yu && X.y == Yl) ? 1u : ((X.x > Yu || Yl > X.y) ? 0u : 2u);
    return kind == PCP_EQ ? eq : kleene_not(eq);
  }
  const int2 Z = dm.load(rec.z);
  auto lt3 = [&](long long dd) -> uint32_t {  // x < y + z + dd
    return (long long)X.x >= (long long)Y.y + Z.y + dd ? 0u : ((long long)X.y < (long long)Y.x + Z.x + dd ? 1u : 2u);
  };
  auto gt3 = [&](long long dd) -> uint32_t {  // x > y + z + dd
    return (long long)X.y <= (long long)Y.x + Z.x + dd ? 0u : ((long long)X.x > (long long)Y.y + Z.y + dd ? 1u : 2u);
  };
  if (kind == PCP_LT3) return lt3(d);
  if (kind == PCP_GT3) return gt3(d);
  if (kind == PCP_EQ3) return kleene_and(gt3(d - 1), lt3(d + 1));
  // XEqYMulZ: (x + dx) = (y + dy) * (z + dz)
  const int32_t* mo = dm.mul_offsets() + 3 * (size_t)rec.d;
  const long long yl = Y.x + (long long)mo[1], yu = Y.y + (long long)mo[1], zl = Z.x + (long long)mo[2], zu = Z.y + (long long)mo[2];
  const long long p0 = yl * zl, p1 = yl * zu, p2 = yu * zl, p3 = yu * zu;
  const long long pl = min(min(p0, p1), min(p2, p3)), pu = max(max(p0, p1), max(p2, p3));
  const long long xl = X.x + (long long)mo[0], xu = X.y + (long long)mo[0];
  if (pl > xu || xl > pu) return 0u;
  return (pl == pu && xl == xu) ? 1u : 2u;
}

// propagate() of one elementary leaf.  A failure raises the node's fail flag (every caller ends the node on it).
__device__ void rec_propagate(const Rec& rec, const LdsDom& dm) {
  const uint32_t kind = rec.xk >> 28;
  if (kind == PCP_BOOL || kind == PCP_NBOOL) {
    // Boolean::propagate = update(var, {1}) (boolean.rs:134-137); BooleanNeg: {0} (boolean_neg.rs:86-89).  A domain without that
    // value is a non-monotonic update — the reference panics (variable/store.rs:153-156); here the node fails (pcp_hip.h).
    const uint32_t x = rec.xk & kSlotMask;
    const int want = (kind == PCP_BOOL ? 1 : 0) - rec.d;
    const int2 X = dm.load(x);
    if (want < X.x || want > X.y) { dm.set_fail(); return; }
    if (want > X.x) dm.raise_lb(x, want);
    if (want < X.y) dm.lower_ub(x, want);
    return;
  }
  (void)eval_record(rec, dm);
}

}  // namespace

// One WAVEFRONT per node (round 4; it was a 256-thread workgroup per node with one lane per unit and the tree walked by recursion unrolled
// eight levels deep: 178 VGPRs, 256 bytes of scratch per lane).  A unit's tree is laid out breadth-first — every child behind its parent,
// the children of a node consecutive (pcp_model_push_formula checks it) — so both walks are LOOPS over the unit's nodes:
//   bottom-up  (last node to first): is_subsumed() of every node from its children's, kept as two bit masks (true / false) in registers;
//   top-down   (first to last): propagate() — an `active` bit mask starts at the root; an active Conjunction activates all its children
//              (conjunction.rs:97-104), an active Disjunction none if a child is entailed, its single not-disentailed child if there is exactly
//              one, and fails the node if there is none (disjunction.rs:97-117); an active leaf runs its filter.
// The Disjunction decides on the statuses the bottom-up walk found a moment earlier, where the reference asks its children while it runs: a
// decision on an older domain is the conservative one (is_subsumed() only ever moves from Unknown to True or False as domains shrink), the
// round after repeats it on the newer one, and the last round — the one that narrows nothing — sees final domains throughout.
// A unit whose root is entailed is unlinked (store.rs:200-207) before it is propagated: its propagate() is a no-op (disjunction.rs:103).
// Units of more than 64 nodes are flat Conjunctions of leaves (Distinct next to formulas): a loop over the members.
__global__ void __launch_bounds__(256) formfix_kernel(const FormArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = blockDim.x >> 6;
  const uint32_t V = a.m.n_vars, S = a.m.n_slots, Wv = (S + 31) >> 5, U = a.n_units, Wu = (U + 31) >> 5;
  const FormCarve cv = form_carve(S, U);
  unsigned char* const mine = smem + (size_t)wv * cv.total;
  int2* const dom = reinterpret_cast<int2*>(mine + cv.dom);
  uint32_t* const chg = reinterpret_cast<uint32_t*>(mine + cv.chg);
  uint32_t* const live = reinterpret_cast<uint32_t*>(mine + cv.live);
  uint32_t* const misc = reinterpret_cast<uint32_t*>(mine + cv.misc);
  const uint32_t words64 = (U + 63) >> 6;
  auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); };
  pcp_stats* const stats = a.stats + (blockIdx.x & (kStatSlots - 1));
  unsigned long long acc_s2 = 0, acc_s3 = 0, acc_narrow = 0, acc_waves = 0, acc_nodes = 0, acc_failed = 0;

  uint32_t n_nodes = a.n_nodes, st_base = 0;
  size_t row_base = 0;
  if (a.sp_ptr) {  // host-stepped device-side DFS (pcp_dfs_device): ONE node, the one on top of the stack
    const uint32_t sp = *a.sp_ptr;
    if (sp == 0 || *a.stop_ptr) return;
    row_base = (size_t)(sp - 1) * V; st_base = sp - 1; n_nodes = 1;
  }
  for (uint32_t node = blockIdx.x * nwv + wv; node < n_nodes; node += gridDim.x * nwv) {
    const size_t row = row_base + (size_t)node * V;
    if (lane < (uint32_t)F_WORDS) misc[lane] = 0;
    for (uint32_t w = lane; w < Wv; w += 64) chg[w] = 0;
    // Store::active (one bit per unit): the caller's row, or every unit (implicit-active nodes)
    for (uint32_t w = lane; w < Wu; w += 64) {
      uint32_t bits = 0xFFFFFFFFu;
      if (a.active_in) { const uint64_t q = a.active_in[(size_t)node * words64 + (w >> 1)]; bits = (uint32_t)(q >> (32 * (w & 1))); }
      if (w == Wu - 1 && (U & 31u)) bits &= (1u << (U & 31u)) - 1u;
      live[w] = bits;
    }
    bool bad = false, wide = false;
    for (uint32_t v = lane; v < S; v += 64) {
      int l = 0, u = 0;
      if (v < V) {
        l = a.lb_in[row + v]; u = a.ub_in[row + v];
        bad |= l > u;
        wide |= (l < -kBoundMax) | (l > kBoundMax) | (u < -kBoundMax) | (u > kBoundMax);
      } else if (v - V >= a.m.sums.count) {
        l = u = a.m.const_val[v - V];  // (the Sum slots in between hold nothing: their domain is computed from the members)
      }
      dom[v] = make_int2(-l, u);
    }
    if (__ballot(wide)) {  // a bound beyond +-(2^29 - 1): refused, not wrapped (pcp_hip.h)
      if (lane == 0) { a.status[st_base + node] = kStatusRetry; atomicMax(a.violation, 1u); }
      continue;
    }
    if (__ballot(bad) && lane == 0) misc[F_FAIL] = 1u;
    wave_sync();

    Ctr ctr;
    const LdsDom dm{dom, 1u, chg, &misc[F_FAIL], 1u, &ctr, a.m.sums};
    uint32_t steps2 = 0, steps3 = 0, rounds = 0;
    auto leaf_propagate = [&](const Rec& rec) {
      const uint32_t kind = rec.xk >> 28;
      if (kind >= PCP_LT3 && kind <= PCP_MUL3) ++steps3; else ++steps2;
      rec_propagate(rec, dm);
    };
    bool failed = __builtin_amdgcn_readfirstlane(misc[F_FAIL]) != 0;
    while (!failed) {
      ++rounds;
      const uint32_t before = ctr.narrow;
      for (uint32_t u = lane; u < U; u += 64) {
        if (!((live[u >> 5] >> (u & 31u)) & 1u)) continue;
        const uint32_t root = a.unit_root[u], n = a.unit_root[u + 1] - root;
        const FNode rn = a.nodes[root];
        bool entailed;
        if (n > 64u || rn.type == PCP_F_LEAF) {
          // a single propagator, or a flat Conjunction of leaves too wide for the masks: the members in order (conjunction.rs:97-104)
          const uint32_t m0 = rn.type == PCP_F_LEAF ? root : rn.first, m1 = rn.type == PCP_F_LEAF ? root + 1 : rn.first + rn.n_children;
          entailed = true;
          for (uint32_t k = m0; k < m1; ++k) {
            const Rec rec = a.m.recs[a.nodes[k].first];
            leaf_propagate(rec);
            entailed = entailed && rec_subsumed(rec, dm) == 1u;
          }
        } else {
          // bottom-up: is_subsumed() of every node of the tree (bit i = node root + i)
          unsigned long long t_true = 0, t_false = 0;
          for (uint32_t i = n; i-- > 0;) {
            const FNode nd = a.nodes[root + i];
            uint32_t s_;
            if (nd.type == PCP_F_LEAF) {
              s_ = rec_subsumed(a.m.recs[nd.first], dm);
            } else {
              const unsigned long long cm = (nd.n_children >= 64 ? ~0ull : ((1ull << nd.n_children) - 1ull)) << (nd.first - root);
              if (nd.type == PCP_F_AND) s_ = (t_false & cm) ? 0u : ((t_true & cm) == cm ? 1u : 2u);   // conjunction.rs:78-94
              else s_ = (t_true & cm) ? 1u : ((t_false & cm) == cm ? 0u : 2u);                          // disjunction.rs:78-94
            }
            if (s_ == 1u) t_true |= 1ull << i; else if (s_ == 0u) t_false |= 1ull << i;
          }
          entailed = (t_true & 1ull) != 0;
          if (!entailed) {
            // top-down: propagate()
            unsigned long long active = 1ull;
            for (uint32_t i = 0; i < n; ++i) {
              if (!((active >> i) & 1ull)) continue;
              const FNode nd = a.nodes[root + i];
              if (nd.type == PCP_F_LEAF) { leaf_propagate(a.m.recs[nd.first]); continue; }
              const unsigned long long cm = (nd.n_children >= 64 ? ~0ull : ((1ull << nd.n_children) - 1ull)) << (nd.first - root);
              if (nd.type == PCP_F_AND) { active |= cm; continue; }
              if (t_true & cm) continue;                                 // an entailed child: the Disjunction holds (disjunction.rs:103)
              const unsigned long long open = cm & ~t_false;             // the children that are not disentailed
              if (open == 0ull) dm.set_fail();                           // all disentailed (disjunction.rs:112-114)
              else if ((open & (open - 1ull)) == 0ull) active |= open;   // exactly one left: unit propagation (disjunction.rs:108-111)
            }
          }
        }
        if (entailed) atomicAnd(&live[u >> 5], ~(1u << (u & 31u)));      // unlink_prop (store.rs:200-207)
      }
      wave_sync();
      failed = __builtin_amdgcn_readfirstlane(misc[F_FAIL]) != 0;
      if (!__ballot(ctr.narrow != before)) break;
    }

    // ---- write back -----------------------------------------------------------------------------------------------------------------
    bool emptied = false;
    for (uint32_t v = lane; v < V; v += 64) {
      const int2 d = dom[v];
      emptied |= -d.x > d.y;
      a.lb_out[row + v] = -d.x; a.ub_out[row + v] = d.y;
    }
    failed = failed || __ballot(emptied) != 0;
    bool any_live = false;
    for (uint32_t w = lane; w < Wu; w += 64) any_live |= live[w] != 0;
    if (a.active_out)
      for (uint32_t w = lane; w < words64; w += 64) {
        const uint64_t lo = live[2 * w], hi = (2 * w + 1 < Wu) ? live[2 * w + 1] : 0u;
        a.active_out[(size_t)node * words64 + w] = lo | (hi << 32);
      }
    const bool unknown = __ballot(any_live) != 0;
    // Consistency::consistency (store.rs:250-256): False if a propagate failed, True if no subscription remains, else Unknown
    if (lane == 0) a.status[st_base + node] = failed ? (uint8_t)PCP_FALSE : (unknown ? (uint8_t)PCP_UNKNOWN : (uint8_t)PCP_TRUE);
    for (int o = 32; o > 0; o >>= 1) { steps2 += __shfl_down(steps2, o); steps3 += __shfl_down(steps3, o); ctr.narrow += __shfl_down(ctr.narrow, o); }
    acc_s2 += steps2; acc_s3 += steps3; acc_narrow += ctr.narrow; acc_waves += rounds ? rounds : 1; acc_nodes += 1; acc_failed += failed ? 1 : 0;
    wave_sync();
  }
  if (lane == 0) {
    if (acc_s2) atomicAdd((unsigned long long*)&stats->steps, acc_s2);
    if (acc_s3) atomicAdd((unsigned long long*)&stats->steps3, acc_s3);
    if (acc_s2 + acc_s3) { atomicAdd((unsigned long long*)&stats->evaluated, acc_s2 + acc_s3); atomicAdd((unsigned long long*)&stats->full_evals, acc_s2 + acc_s3); }
    if (acc_narrow) atomicAdd((unsigned long long*)&stats->narrowings, acc_narrow);
    if (acc_waves) atomicAdd((unsigned long long*)&stats->waves, acc_waves);
    if (acc_nodes) atomicAdd((unsigned long long*)&stats->nodes, acc_nodes);
    if (acc_failed) atomicAdd((unsigned long long*)&stats->failed_nodes, acc_failed);
  }
}

size_t lds_bytes_formula(uint32_t n_slots, uint32_t n_units, uint32_t waves) {
  const FormCarve c = form_carve(n_slots, n_units);
  return c.total * waves <= 160 * 1024 ? c.total * waves : 0;
}

hipError_t launch_formfix(const FormArgs& a, const LaunchPlan& p, hipStream_t stream) {
  if (p.lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(formfix_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(formfix_kernel, dim3(p.grid), dim3(p.block), p.lds_bytes, stream, a);
  return hipGetLastError();
}

}  // namespace pcp
