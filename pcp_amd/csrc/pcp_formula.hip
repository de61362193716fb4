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
constexpr int kMaxDepth = 8;

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

struct FormCtx {
  const FNode* nodes;
  const Rec* recs;
  LdsDom dm;
  uint32_t steps2, steps3;
};

template <int DEPTH>
__device__ uint32_t f_subsumed(const FormCtx& c, uint32_t at) {
  const FNode nd = c.nodes[at];
  if (nd.type == PCP_F_LEAF) return rec_subsumed(c.recs[nd.first], c.dm);
  if constexpr (DEPTH > 1) {
    if (nd.type == PCP_F_AND) {  // conjunction.rs:78-94
      bool all_entailed = true;
      for (uint32_t k = 0; k < nd.n_children; ++k) {
        const uint32_t s = f_subsumed<DEPTH - 1>(c, nd.first + k);
        if (s == 0u) return 0u;
        if (s == 2u) all_entailed = false;
      }
      return all_entailed ? 1u : 2u;
    }
    bool all_disentailed = true;  // disjunction.rs:78-94
    for (uint32_t k = 0; k < nd.n_children; ++k) {
      const uint32_t s = f_subsumed<DEPTH - 1>(c, nd.first + k);
      if (s == 1u) return 1u;
      if (s == 2u) all_disentailed = false;
    }
    return all_disentailed ? 0u : 2u;
  }
  return 2u;  // (deeper than the host accepts: never reached)
}

template <int DEPTH>
__device__ void f_propagate(FormCtx& c, uint32_t at) {
  const FNode nd = c.nodes[at];
  if (nd.type == PCP_F_LEAF) {
    const Rec rec = c.recs[nd.first];
    const uint32_t kind = rec.xk >> 28;
    if (kind >= PCP_LT3 && kind <= PCP_MUL3) ++c.steps3; else ++c.steps2;
    rec_propagate(rec, c.dm);
    return;
  }
  if constexpr (DEPTH > 1) {
    if (nd.type == PCP_F_AND) {  // conjunction.rs:97-104: the children in order (a failure ends the node anyway)
      for (uint32_t k = 0; k < nd.n_children; ++k) f_propagate<DEPTH - 1>(c, nd.first + k);
      return;
    }
    // disjunction.rs:97-117
    uint32_t num_disentailed = 0, unknown_formula = 0;
    for (uint32_t k = 0; k < nd.n_children; ++k) {
      const uint32_t s = f_subsumed<DEPTH - 1>(c, nd.first + k);
      if (s == 1u) return;
      if (s == 0u) ++num_disentailed; else unknown_formula = k;
    }
    if (num_disentailed + 1 == nd.n_children) f_propagate<DEPTH - 1>(c, nd.first + unknown_formula);
    else if (num_disentailed == nd.n_children) c.dm.set_fail();
  }
}

}  // namespace

__global__ void __launch_bounds__(256) formfix_kernel(const FormArgs a_in) {
  FormArgs a = a_in;
  a.stats += blockIdx.x & (kStatSlots - 1);
  if (a.sp_ptr) {  // host-stepped device-side DFS (pcp_dfs_device): the node on top of the stack, as in fixpoint_kernel
    const uint32_t sp = *a.sp_ptr;
    if (sp == 0 || *a.stop_ptr) return;
    const size_t off = (size_t)(sp - 1) * a.m.n_vars;
    a.lb_in += off; a.ub_in += off; a.lb_out += off; a.ub_out += off; a.status += sp - 1;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t tid = threadIdx.x, nth = blockDim.x, lane = tid & 63;
  const uint32_t V = a.m.n_vars, S = a.m.n_slots, Wv = (S + 31) >> 5, U = a.n_units, Wu = (U + 31) >> 5;
  const FormCarve cv = form_carve(S, U);
  int2* const dom = reinterpret_cast<int2*>(smem + cv.dom);
  uint32_t* const chg = reinterpret_cast<uint32_t*>(smem + cv.chg);
  uint32_t* const live = reinterpret_cast<uint32_t*>(smem + cv.live);
  uint32_t* const misc = reinterpret_cast<uint32_t*>(smem + cv.misc);
  const uint32_t node = blockIdx.x;
  const size_t row = (size_t)node * V;
  const uint32_t words64 = (U + 63) >> 6;

  if (tid < (uint32_t)F_WORDS) misc[tid] = 0;
  for (uint32_t w = tid; w < Wv; w += nth) chg[w] = 0;
  // Store::active (one bit per unit): the caller's row, or every unit (implicit-active nodes)
  for (uint32_t w = tid; w < Wu; w += nth) {
    uint32_t bits = 0xFFFFFFFFu;
    if (a.active_in) { const uint64_t q = a.active_in[(size_t)node * words64 + (w >> 1)]; bits = (uint32_t)(q >> (32 * (w & 1))); }
    if (w == Wu - 1 && (U & 31u)) bits &= (1u << (U & 31u)) - 1u;
    live[w] = bits;
  }
  __syncthreads();
  {
    bool bad = false, wide = false;
    for (uint32_t v = tid; v < S; v += nth) {
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
    if (bad) atomicOr(&misc[F_FAIL], 1u);
    if (wide) atomicOr(&misc[F_OOB], 1u);
  }
  __syncthreads();
  if (misc[F_OOB]) {  // a bound beyond +-(2^29 - 1): refused, not wrapped (pcp_hip.h)
    if (tid == 0) { a.status[node] = kStatusRetry; atomicMax(a.violation, 1u); }
    return;
  }

  Ctr ctr;
  FormCtx fc{a.nodes, a.m.recs, LdsDom{dom, 1u, chg, &misc[F_FAIL], 1u, &ctr, a.m.sums}, 0u, 0u};
  uint32_t rounds = 0;
  bool failed_now = misc[F_FAIL] != 0;  // (nobody writes the flag between a round's first barrier and the next round)
  while (!failed_now) {
    ++rounds;
    for (uint32_t u = tid; u < U; u += nth) {
      if (!((live[u >> 5] >> (u & 31u)) & 1u)) continue;
      const uint32_t root = a.unit_root[u];
      f_propagate<kMaxDepth>(fc, root);                       // propagate_one (store.rs:166-175) ...
      if (f_subsumed<kMaxDepth>(fc, root) == 1u)              // ... is_subsumed() == True: unlink_prop (store.rs:200-207)
        atomicAnd(&live[u >> 5], ~(1u << (u & 31u)));
    }
    __syncthreads();
    bool any = false;
    for (uint32_t w = tid; w < Wv; w += nth) { any |= chg[w] != 0; }
    if (any) misc[F_CHANGED] = rounds;  // (benign race: every writer stores the same value)
    __syncthreads();
    const bool again = misc[F_CHANGED] == rounds;
    failed_now = misc[F_FAIL] != 0;
    for (uint32_t w = tid; w < Wv; w += nth) chg[w] = 0;
    __syncthreads();
    if (!again) break;
  }

  // ---- write back -----------------------------------------------------------------------------------------------------------------
  {
    bool bad = false;
    for (uint32_t v = tid; v < V; v += nth) {
      const int2 d = dom[v];
      bad |= -d.x > d.y;
      a.lb_out[row + v] = -d.x; a.ub_out[row + v] = d.y;
    }
    if (bad) atomicOr(&misc[F_FAIL], 1u);
  }
  for (int o = 32; o > 0; o >>= 1) { ctr.narrow += __shfl_down(ctr.narrow, o); fc.steps2 += __shfl_down(fc.steps2, o); fc.steps3 += __shfl_down(fc.steps3, o); }
  if (lane == 0) {
    if (ctr.narrow) atomicAdd(&misc[F_NARROW], ctr.narrow);
    if (fc.steps2) atomicAdd(reinterpret_cast<unsigned long long*>(&misc[F_STEPS2]), (unsigned long long)fc.steps2);
    if (fc.steps3) atomicAdd(reinterpret_cast<unsigned long long*>(&misc[F_STEPS3]), (unsigned long long)fc.steps3);
  }
  __syncthreads();
  if (a.active_out)
    for (uint32_t w = tid; w < words64; w += nth) {
      const uint64_t lo = live[2 * w], hi = (2 * w + 1 < Wu) ? live[2 * w + 1] : 0u;
      a.active_out[(size_t)node * words64 + w] = lo | (hi << 32);
    }
  if (tid == 0) {
    bool any_live = false;
    for (uint32_t w = 0; w < Wu; ++w) any_live |= live[w] != 0;
    const bool failed = misc[F_FAIL] != 0;
    // Consistency::consistency (store.rs:250-256): False if a propagate failed, True if no subscription remains, else Unknown
    a.status[node] = failed ? (uint8_t)PCP_FALSE : (any_live ? (uint8_t)PCP_UNKNOWN : (uint8_t)PCP_TRUE);
    const unsigned long long s2 = *reinterpret_cast<unsigned long long*>(&misc[F_STEPS2]), s3 = *reinterpret_cast<unsigned long long*>(&misc[F_STEPS3]);
    if (s2) atomicAdd((unsigned long long*)&a.stats->steps, s2);
    if (s3) atomicAdd((unsigned long long*)&a.stats->steps3, s3);
    if (s2 + s3) { atomicAdd((unsigned long long*)&a.stats->evaluated, s2 + s3); atomicAdd((unsigned long long*)&a.stats->full_evals, s2 + s3); }
    if (misc[F_NARROW]) atomicAdd((unsigned long long*)&a.stats->narrowings, (unsigned long long)misc[F_NARROW]);
    atomicAdd((unsigned long long*)&a.stats->waves, (unsigned long long)rounds);
    atomicAdd((unsigned long long*)&a.stats->nodes, 1ull);
    if (failed) atomicAdd((unsigned long long*)&a.stats->failed_nodes, 1ull);
  }
}

size_t lds_bytes_formula(uint32_t n_slots, uint32_t n_units) {
  const FormCarve c = form_carve(n_slots, n_units);
  return c.total <= 160 * 1024 ? c.total : 0;
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
