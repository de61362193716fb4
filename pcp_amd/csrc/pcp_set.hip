// pcp_set.hip — the propagation fixpoint over IntervalSet<i32> domains (set mode), gfx950.
//
// The reference's default space is FDSpace = Space<VStoreSet, ...> with VStoreSet = VStoreTrail<IntervalSet<i32>>
// (variable/mod.rs:38, search/mod.rs:41-43; example/src/nqueens.rs:34 allocates IntervalSet::new(1, n)): domains are SETS, so
// XNeqY removes interior values (x_neq_y.rs:82-93 with IntervalSet::difference), variable::Store::update raises `Inner` events
// (events/mod.rs:57-64) and XEqY intersects sets (x_eq_y.rs:102-107).  Same engine semantics as the interval kernels
// (Store::consistency, propagation/store.rs:125-258): every live propagator once, then wake-up rounds until nothing changes.
//
// MI355X design: ONE workgroup (1024 threads, one CU) owns a node.  Its domains live in LDS for the whole fixpoint:
//   bits[V][set_words] u64   value v of variable x <-> bit (v - base) of bits[x]        (N-queens-1000: 1000 x 16 words = 125 KB)
//   bnd[V] (lb, ub)          the sets' bounds, exact at the start of every round
// A filter narrows a set with ds_and_b64 on the words it touches and marks the variable changed; it never writes the bounds.
// The first step of the next round re-derives the exact bounds of every changed variable from its words (ffs / clz) — between
// two such steps a filter may read bounds that are wider than the set, i.e. a superset of the current domain, which is the same
// benign race as in the interval kernels (monotone, contracting filters: the greatest fixpoint is unique).
// Wake-up rule: any change of a variable wakes all its propagators.  The reference wakes Bound-subscribers (XLessY and the
// ternary kinds) only on Bound/Assignment events, not on Inner ones (indexed_deps.rs:99-113) — but those filters read and
// write bounds only, so running them after an interior removal is a no-op: same fixpoint, a few more steps.
// Entailment (`active`, status True): XNeqY is entailed iff the two SETS are disjoint (x_eq_y.rs:87-93 through
// IntervalSet::is_disjoint), decided on the words (shifted intersection).
#include <algorithm>

#include "pcp_internal.h"
#ifndef PCP_ABLATE
#define PCP_ABLATE 0  // profiling builds (tools/build_variant.py): bit 2048 = phase timers of setfix_kernel in the counters
#endif

namespace pcp {

namespace {

constexpr uint32_t kSetThreads = 1024;

struct SetCarve {
  size_t bits, bnd, chg_a, chg_b, list_id, list_off, list_deg, misc, total;
};
__host__ __device__ inline SetCarve set_carve(uint32_t V, uint32_t S, uint32_t sw, uint32_t cap) {
  auto up = [](size_t x) { return (x + 15) & ~(size_t)15; };
  const size_t Wv = (S + 31) / 32;
  SetCarve c;
  size_t o = 0;
  c.bits = o; o = up(o + (size_t)V * sw * 8);
  c.bnd = o; o = up(o + (size_t)V * 8);
  c.chg_a = o; o = up(o + Wv * 4);
  c.chg_b = o; o = up(o + Wv * 4);
  c.list_id = o; o = up(o + (size_t)cap * 4);
  c.list_off = o; o = up(o + (size_t)cap * 4);
  c.list_deg = o; o = up(o + (size_t)cap * 4);
  c.misc = o; o = up(o + 32 * 4);
  c.total = o;
  return c;
}

enum { S_FAIL = 0, S_TOTAL = 1, S_ITEMS = 2, S_TOTAL2 = 3, S_ITEMS2 = 4, S_WAVES = 5, S_OPEN = 6, S_NARROW = 7, S_STEPS2 = 8, S_STEPS3 = 10, S_EVAL = 12, S_LIVE = 14, S_NSING = 15,
       S_TRAILOVF = 16, S_TRAILLEN = 17, S_CTL = 18 /* .. 23: the DFS loop's broadcast words */ };

// One node's domains in LDS.  TRAIL = true (the device-side DFS, setdfs_kernel): every narrowing also appends (word, removed bits)
// to the tree's undo trail, which is what the reference's VStoreTrail keeps (variable/memory/trail_memory.rs:100-104): a
// backtrack ORs the removed bits back instead of reloading a node.
template <bool TRAIL>
struct SetDomT {
  unsigned long long* bits;  // [V][sw]
  int2* bnd;                 // [V] (lb, ub)
  const int32_t* cval;       // constants, slots >= V
  uint32_t V, sw;
  int32_t base;
  uint32_t* chg;             // changed mask to mark
  uint32_t* misc;
  uint32_t* narrow;          // per-thread counter
  uint4* trail = nullptr;    // TRAIL: the tree's trail in HBM, its length (an LDS word) and its capacity
  uint32_t* trail_len = nullptr;
  uint32_t trail_cap = 0;

  __device__ __forceinline__ void log(uint32_t var, uint32_t widx, unsigned long long removed) const {
    if constexpr (TRAIL) {
      if (!removed) return;
      const uint32_t pos = atomicAdd(trail_len, 1u);
      if (pos < trail_cap) trail[pos] = make_uint4(widx, var, (uint32_t)removed, (uint32_t)(removed >> 32));
      else atomicOr(&misc[S_TRAILOVF], 1u);
    }
  }

  __device__ __forceinline__ bool is_const(uint32_t s) const { return s >= V; }
  __device__ __forceinline__ int2 bounds(uint32_t s) const {
    if (s >= V) { const int c = cval[s - V]; return make_int2(c, c); }
    return bnd[s];
  }
  __device__ __forceinline__ void fail() const { atomicOr(&misc[S_FAIL], 1u); }
  __device__ __forceinline__ void mark(uint32_t s) const { atomicOr(&chg[s >> 5], 1u << (s & 31)); ++*narrow; }
  __device__ __forceinline__ bool test(uint32_t s, int v) const {  // v in the set of slot s?
    if (s >= V) return cval[s - V] == v;
    const long long b = (long long)v - base;
    if (b < 0 || b >= (long long)sw * 64) return false;
    return (bits[(size_t)s * sw + (b >> 6)] >> (b & 63)) & 1ull;
  }
  // IntervalSet::difference(&v): remove ONE value (x_neq_y.rs:86-89).  Removing the value of a Constant empties it:
  // Constant::update returns false (term/constant.rs:49-52).
  __device__ __forceinline__ void remove(uint32_t s, int v) const {
    if (s >= V) { if (cval[s - V] == v) fail(); return; }
    const long long b = (long long)v - base;
    if (b < 0 || b >= (long long)sw * 64) return;
    const unsigned long long m = 1ull << (b & 63);
    unsigned long long* w = &bits[(size_t)s * sw + (b >> 6)];
    if (!(*w & m)) return;
    if (atomicAnd(w, ~m) & m) { log(s, s * sw + (uint32_t)(b >> 6), m); mark(s); }
  }
  // keep only the values <= t  (shrink_right) / >= t (shrink_left), within the cached bounds [lo, hi] of slot s
  __device__ __forceinline__ void keep_le(uint32_t s, long long t, const int2 cur) const {
    if (t >= cur.y) return;
    if (s >= V) { fail(); return; }  // a constant above t
    clear_range(s, t + 1, cur.y);
  }
  __device__ __forceinline__ void keep_ge(uint32_t s, long long t, const int2 cur) const {
    if (t <= cur.x) return;
    if (s >= V) { fail(); return; }
    clear_range(s, cur.x, t - 1);
  }
  __device__ __forceinline__ void clear_range(uint32_t s, long long lo, long long hi) const {  // values lo..hi inclusive
    long long b0 = lo - base, b1 = hi - base;
    if (b0 < 0) b0 = 0;
    if (b1 >= (long long)sw * 64) b1 = (long long)sw * 64 - 1;
    if (b0 > b1) return;
    bool changed = false;
    for (long long k = b0 >> 6; k <= (b1 >> 6); ++k) {
      unsigned long long m = ~0ull;
      if (k == (b0 >> 6)) m &= ~0ull << (b0 & 63);
      if (k == (b1 >> 6)) m &= ~0ull >> (63 - (b1 & 63));
      unsigned long long* w = &bits[(size_t)s * sw + k];
      if (*w & m) {
        const unsigned long long gone = atomicAnd(w, ~m) & m;
        log(s, s * sw + (uint32_t)k, gone);
        changed |= gone != 0;
      }
    }
    if (changed) mark(s);
  }
  // 64 bits of slot s starting at bit position pos (positions outside the universe read as 0)
  __device__ __forceinline__ unsigned long long window(uint32_t s, long long pos) const {
    const long long nb = (long long)sw * 64;
    if (pos <= -64 || pos >= nb) return 0ull;
    const long long k = pos >> 6;  // floor
    const int sh = (int)(pos & 63);
    const unsigned long long lo = (k >= 0 && k < (long long)sw) ? bits[(size_t)s * sw + k] : 0ull;
    if (sh == 0) return lo;
    const unsigned long long hi = (k + 1 >= 0 && k + 1 < (long long)sw) ? bits[(size_t)s * sw + k + 1] : 0ull;
    return (lo >> sh) | (hi << (64 - sh));
  }
  // x := x ∩ (y + d) on the words of x
  __device__ __forceinline__ void intersect_shifted(uint32_t x, uint32_t y, long long d) const {
    if (x >= V) return;
    bool changed = false;
    for (uint32_t k = 0; k < sw; ++k) {
      unsigned long long* w = &bits[(size_t)x * sw + k];
      const unsigned long long cur = *w;
      if (!cur) continue;
      unsigned long long other;
      if (y >= V) {
        const long long b = (long long)cval[y - V] + d - base - (long long)k * 64;
        other = (b >= 0 && b < 64) ? (1ull << b) : 0ull;
      } else {
        other = window(y, (long long)k * 64 - d);  // value v of x  <->  value v - d of y
      }
      if (cur & ~other) {
        const unsigned long long gone = atomicAnd(w, other) & ~other;
        log(x, x * sw + k, gone);
        changed |= gone != 0;
      }
    }
    if (changed) mark(x);
  }
  // is x ∩ (y + d) empty?
  __device__ __forceinline__ bool disjoint_shifted(uint32_t x, uint32_t y, long long d) const {
    for (uint32_t k = 0; k < sw; ++k) {
      const unsigned long long cur = bits[(size_t)x * sw + k];
      if (cur && (cur & window(y, (long long)k * 64 - d))) return false;
    }
    return true;
  }
};
using SetDom = SetDomT<false>;

// One filter step on sets: propagate() + is_subsumed().  `want_entailed` = false skips the (possibly expensive) subsumption
// test — implicit-active nodes need it only in the final scan.  Returns whether the propagator is entailed.
template <class DM>
__device__ __forceinline__ bool eval_set(const Rec& rec, const DM& dm, const bool want_entailed) {
  const uint32_t kind = rec.xk >> 28, x = rec.xk & kSlotMask, y = rec.y;
  const long long d = rec.d;
  const int2 X = dm.bounds(x), Y = dm.bounds(y);
  if (X.x > X.y || Y.x > Y.y) return false;  // an emptied set: the node has failed (found by the next bounds step)
  if (kind == PCP_NEQ) {
    // XNeqY::propagate (x_neq_y.rs:82-93): a singleton side is removed from the other SET, wherever the value sits
    if (X.x == X.y) dm.remove(y, (int)(X.x - d));
    else if (Y.x == Y.y) dm.remove(x, (int)(Y.x + d));
    if (!want_entailed) return false;
    // !XEqY::is_subsumed (x_neq_y.rs:71-73, x_eq_y.rs:87-93): True iff the sets are disjoint
    if (X.x > Y.y + d || Y.x + d > X.y) return true;
    if (X.x == X.y) return !dm.test(y, (int)(X.x - d));
    if (Y.x == Y.y) return !dm.test(x, (int)(Y.x + d));
    if (dm.is_const(x) || dm.is_const(y)) return false;
    return dm.disjoint_shifted(x, y, d);
  }
  if (kind == PCP_EQ) {
    // XEqY::propagate (x_eq_y.rs:102-107): both become the intersection of the sets
    dm.intersect_shifted(x, y, d);
    dm.intersect_shifted(y, x, -d);
    if (dm.is_const(x) && dm.is_const(y) && X.x != Y.x + d) dm.fail();
    return want_entailed && X.x == X.y && Y.x == Y.y && X.x == Y.x + d;  // x_eq_y.rs:87-88
  }
  if (kind == PCP_LT) {
    // XLessY::propagate (x_less_y.rs:104-109): x.strict_shrink_right(y.upper()), y.strict_shrink_left(x.lower())
    dm.keep_le(x, (long long)Y.y + d - 1, X);
    dm.keep_ge(y, (long long)X.x - d + 1, Y);
    return (long long)X.y < (long long)Y.x + d;  // x_less_y.rs:90-91 (on the bounds read; re-evaluated while anything changes)
  }
  const uint32_t z = rec.z;
  const int2 Z = dm.bounds(z);
  if (Z.x > Z.y) return false;
  auto lt3 = [&](long long dd) {  // x < y + z + dd   (x_less_y_plus_z.rs:105-119)
    dm.keep_le(x, (long long)Y.y + Z.y + dd - 1, X);
    dm.keep_ge(y, (long long)X.x - Z.y - dd + 1, Y);
    dm.keep_ge(z, (long long)X.x - Y.y - dd + 1, Z);
  };
  auto gt3 = [&](long long dd) {  // x > y + z + dd   (x_greater_y_plus_z.rs:106-118)
    dm.keep_ge(x, (long long)Y.x + Z.x + dd + 1, X);
    dm.keep_le(y, (long long)X.y - Z.x - dd - 1, Y);
    dm.keep_le(z, (long long)X.y - Y.x - dd - 1, Z);
  };
  if (kind == PCP_LT3) { lt3(d); return (long long)X.y < (long long)Y.x + Z.x + d; }
  if (kind == PCP_GT3) { gt3(d); return (long long)X.x > (long long)Y.y + Z.y + d; }
  if (kind == PCP_EQ3) {  // geq && leq (x_eq_y_plus_z.rs:85-87; cmp/mod.rs:62-86)
    gt3(d - 1);
    lt3(d + 1);
    return (long long)X.x > (long long)Y.y + Z.y + d - 1 && (long long)X.y < (long long)Y.x + Z.x + d + 1;
  }
  dm.fail();  // XEqYMulZ on sets is rejected on the host (pcp_model_push_props)
  return false;
}

__device__ __forceinline__ int2 scan_bounds(const unsigned long long* w, uint32_t sw, int32_t base) {
  int lo = 1, hi = 0;
  for (uint32_t k = 0; k < sw; ++k)
    if (w[k]) { lo = base + (int)k * 64 + (int)__builtin_ctzll(w[k]); break; }
  for (uint32_t k = sw; k-- > 0;)
    if (w[k]) { hi = base + (int)k * 64 + 63 - (int)__builtin_clzll(w[k]); break; }
  return make_int2(lo, hi);
}

}  // namespace

size_t lds_bytes_set(uint32_t n_vars, uint32_t n_slots, uint32_t set_words, uint32_t list_cap) {
  const SetCarve c = set_carve(n_vars, n_slots, set_words, list_cap);
  return c.total <= 160 * 1024 ? c.total : 0;
}

struct SetArgs {
  ModelDev m;
  uint32_t n_nodes, set_words, list_cap;
  int32_t base;
  const uint64_t* bits_in;
  uint64_t* bits_out;
  int32_t* lb_out;
  int32_t* ub_out;
  const uint64_t* live_in;  // record-level rows or null
  uint64_t* live;           // null = implicit
  uint8_t* status;
  pcp_stats* stats;
};

template <bool IMPLICIT>
__global__ void __launch_bounds__(kSetThreads) setfix_kernel(const SetArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t tid = threadIdx.x, nth = blockDim.x, lane = tid & 63;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = nth >> 6;
  const uint32_t V = a.m.n_vars, S = a.m.n_slots, sw = a.set_words, C = a.list_cap, P = a.m.n_recs, words = (P + 63) >> 6;
  const uint32_t Wv = (S + 31) >> 5;
  const SetCarve cv = set_carve(V, S, sw, C);
  unsigned long long* bits = reinterpret_cast<unsigned long long*>(smem + cv.bits);
  int2* bnd = reinterpret_cast<int2*>(smem + cv.bnd);
  uint32_t* cur = reinterpret_cast<uint32_t*>(smem + cv.chg_a);
  uint32_t* nxt = reinterpret_cast<uint32_t*>(smem + cv.chg_b);
  uint32_t* list_id = reinterpret_cast<uint32_t*>(smem + cv.list_id);
  uint32_t* list_off = reinterpret_cast<uint32_t*>(smem + cv.list_off);
  uint32_t* list_deg = reinterpret_cast<uint32_t*>(smem + cv.list_deg);
  uint32_t* misc = reinterpret_cast<uint32_t*>(smem + cv.misc);
  const uint32_t node = blockIdx.x;
  pcp_stats* const st_slot = a.stats + (PCP_ABLATE ? 0u : (blockIdx.x & (kStatSlots - 1)));  // striped counters (pcp_internal.h)
  const uint64_t tail_mask = (P & 63) ? ((1ull << (P & 63)) - 1) : ~0ull;

  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};  // profiling build: phase boundaries (100 MHz ticks)
  if (PCP_ABLATE & 2048) tph[0] = wall_clock64();
  // ---- phase 0: stage the sets, derive their bounds ---------------------------------------------------------------------
  if (tid < 32) misc[tid] = 0;
  for (uint32_t i = tid; i < Wv; i += nth) { cur[i] = 0; nxt[i] = 0; }
  {
    const uint64_t* src = a.bits_in + (size_t)node * V * sw;
    const uint32_t nwords = V * sw;
    if (!(nwords & 1u) && !((size_t)src & 15)) {  // 16 bytes per lane: half the memory instructions of the node's 125 KB
      const uint4* s4 = reinterpret_cast<const uint4*>(src);
      uint4* d4 = reinterpret_cast<uint4*>(bits);
      for (uint32_t i = tid; i < nwords / 2; i += nth) d4[i] = s4[i];
    } else {
      for (uint32_t i = tid; i < nwords; i += nth) bits[i] = src[i];
    }
  }
  __syncthreads();
  for (uint32_t v = tid; v < V; v += nth) {
    const int2 b = scan_bounds(bits + (size_t)v * sw, sw, a.base);
    bnd[v] = b;
    if (b.x > b.y) atomicOr(&misc[S_FAIL], 1u);  // empty input domain
  }
  __syncthreads();

  uint32_t narrow = 0;
  uint64_t steps2 = 0, steps3 = 0;
  uint64_t* live_row = IMPLICIT ? nullptr : a.live + (size_t)node * words;
  if (PCP_ABLATE & 2048) tph[1] = wall_clock64();

  // ---- phase 1: every live propagator once (init_scheduler, store.rs:144-149) --------------------------------------------
  // Implicit nodes of an all-XNeqY model (N-queens): XNeqY::propagate is a no-op unless one operand is a singleton
  // (x_neq_y.rs:82-93), so of the whole sweep only the records of ASSIGNED variables can act: they are reached through the
  // variables' adjacency lists instead of streaming the table (N-queens-1000: a few thousand records instead of 1.5 million).
  // The other records count as reference-equivalent steps (the reference pops them) but are never looked at.
  uint64_t ev_only = 0;  // evaluated pairs that are not to be added to the credited steps again
  bool bulk_sweep = false;
  if constexpr (IMPLICIT) {
    bulk_sweep = a.m.uniform_kind == PCP_NEQ && S == V && !misc[S_FAIL];  // (a Constant operand has no adjacency: S == V excludes them)
    if (bulk_sweep) {
      for (uint32_t v = tid; v < V; v += nth) {
        const int2 b = bnd[v];
        if (b.x == b.y) {
          const uint32_t pos = atomicAdd(&misc[S_NSING], 1u);
          if (pos < C) list_id[pos] = v;
        }
      }
      __syncthreads();
      const uint32_t ns = misc[S_NSING];
      bulk_sweep = ns <= C;  // more assigned variables than the list holds: stream the table after all
      if (bulk_sweep) {
        const SetDom dm{bits, bnd, a.m.const_val, V, sw, a.base, cur, misc, &narrow};
        // four records per thread in flight; binary models rebuild the record from the adjacency payload (one coalesced
        // load instead of an index load and a gather that depends on it)
        constexpr int U = 4;
        for (uint32_t e = 0; e < ns; ++e) {
          const uint32_t v = list_id[e], o0 = a.m.adj_off[v], o1 = a.m.adj_off[v + 1];
          for (uint32_t i0 = o0 + tid; i0 < o1; i0 += nth * U) {
            Rec rc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const uint32_t i = i0 + u * nth;
              const uint32_t at = i < o1 ? i : o0;
              if (a.m.adjp) {
                const uint2 q = a.m.adjp[at];
                const uint32_t other = q.x & kSlotMask, kind = (q.x >> 28) & 7u;
                const bool is_y = (q.x >> 31) != 0;
                rc[u].xk = (is_y ? other : v) | (kind << 28);
                rc[u].y = is_y ? v : other;
                rc[u].z = 0;
                rc[u].d = (int32_t)q.y;
              } else {
                rc[u] = a.m.recs[a.m.adj[at]];
              }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
              if (i0 + u * nth < o1) { (void)eval_set(rc[u], dm, false); ++ev_only; }
          }
        }
        if (tid == 0) { *reinterpret_cast<unsigned long long*>(&misc[S_STEPS2]) += (unsigned long long)P; misc[S_LIVE] = 1u; }
      }
    }
  }
  if (bulk_sweep) {
  } else if (!misc[S_FAIL]) {
    const uint64_t* in_row = (IMPLICIT || !a.live_in) ? nullptr : a.live_in + (size_t)node * words;
    const SetDom dm{bits, bnd, a.m.const_val, V, sw, a.base, cur, misc, &narrow};
    for (uint32_t w = wv; w < words; w += nwv) {
      const uint64_t raw = in_row ? in_row[w] : ~0ull;
      uint64_t word = raw;
      if (w == words - 1) word &= tail_mask;  // bits at or above n_recs name no record: they are dropped from the caller's row
      const uint32_t r = (w << 6) + lane;
      bool e = false;
      if ((word >> lane) & 1ull) {
        const Rec rec = a.m.recs[r];
        e = eval_set(rec, dm, !IMPLICIT);
        if ((rec.xk >> 28) > PCP_LT) ++steps3; else ++steps2;
      }
      if constexpr (!IMPLICIT) {
        const uint64_t nw = word & ~__ballot(e);
        if (lane == 0 && (in_row != live_row || nw != raw)) live_row[w] = nw;
      }
    }
  } else if constexpr (!IMPLICIT) {
    // failed while staging: the row that is exported must still be defined (the caller's row, or all units active)
    if (a.live_in != a.live || !a.live_in)
      for (uint32_t w = tid; w < words; w += nth) {
        uint64_t word = a.live_in ? a.live_in[(size_t)node * words + w] : ~0ull;
        if (w == words - 1) word &= tail_mask;
        live_row[w] = word;
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  if (PCP_ABLATE & 2048) tph[2] = wall_clock64();
  // ---- phase 2: wake-up rounds (react + schedule as waves, store.rs:191-198) --------------------------------------------
  const bool neq_only = IMPLICIT && a.m.uniform_kind == PCP_NEQ && S == V;
  for (uint32_t round = 0;; ++round) {
    const uint32_t m_total = (round & 1u) ? S_TOTAL2 : S_TOTAL, m_items = (round & 1u) ? S_ITEMS2 : S_ITEMS;
    // the exact bounds of every changed variable, from its words (an emptied set: the node has failed) — one thread per
    // variable: left to the compaction pass below (one thread per 32 variables) this rescan was most of a node's time
    for (uint32_t v = tid; v < V; v += nth) {
      if (!((cur[v >> 5] >> (v & 31)) & 1u)) continue;
      const int2 b = scan_bounds(bits + (size_t)v * sw, sw, a.base);
      bnd[v] = b;
      if (b.x > b.y) atomicOr(&misc[S_FAIL], 1u);
    }
    __syncthreads();
    {
      uint32_t degsum = 0;
      for (uint32_t w = tid; w < Wv; w += nth) {
        if (round) nxt[w] = 0;
        uint32_t bitsw = cur[w];
        if (!bitsw) continue;
        uint32_t dropped = 0;  // changed variables that wake nobody: taken out of `cur`, which the FIFO dedup below reads
        while (bitsw) {
          const uint32_t v = (w << 5) + __builtin_ctz(bitsw);
          bitsw &= bitsw - 1;
          if (v < V) {
            const int2 b = bnd[v];
            // all-XNeqY model, implicit node: a propagator acts only through a SINGLETON operand, so a variable that lost
            // values but is not assigned wakes nobody (the reference wakes all of them on its Inner event: no-op steps)
            if (neq_only && b.x != b.y) { dropped |= 1u << (v & 31); continue; }
          }
          const uint32_t pos = atomicAdd(&misc[m_total], 1u);
          if (pos < C) {
            const uint32_t o0 = (v < V) ? a.m.adj_off[v] : 0u, o1 = (v < V) ? a.m.adj_off[v + 1] : 0u;
            list_id[pos] = v; list_off[pos] = o0; list_deg[pos] = o1 - o0;
            degsum += o1 - o0;
          }
        }
        if (dropped) cur[w] &= ~dropped;
      }
      if (degsum) atomicAdd(&misc[m_items], degsum);
    }
    __syncthreads();
    const uint32_t total = misc[m_total];
    if (total == 0 || misc[S_FAIL]) break;
    if (tid == 0) { misc[S_WAVES] += 1; misc[(round & 1u) ? S_TOTAL : S_TOTAL2] = 0; misc[(round & 1u) ? S_ITEMS : S_ITEMS2] = 0; }
    const SetDom dm{bits, bnd, a.m.const_val, V, sw, a.base, nxt, misc, &narrow};
    auto run = [&](uint32_t v, uint32_t r) {
      // a record woken from variable v runs unless a lower-numbered changed variable of the same record runs it (FIFO dedup)
      if constexpr (!IMPLICIT) {
        if (!((live_row[r >> 6] >> (r & 63)) & 1ull)) return;  // unlinked (store.rs:200-207)
      }
      const Rec rec = a.m.recs[r];
      const uint32_t x = rec.xk & kSlotMask;
      const bool tern = (rec.xk >> 28) > PCP_LT;
      if (x < v && ((cur[x >> 5] >> (x & 31)) & 1u)) return;
      if (rec.y < v && ((cur[rec.y >> 5] >> (rec.y & 31)) & 1u)) return;
      if (tern && rec.z < v && ((cur[rec.z >> 5] >> (rec.z & 31)) & 1u)) return;
      if (tern) ++steps3; else ++steps2;
      const bool e = eval_set(rec, dm, !IMPLICIT);
      if constexpr (!IMPLICIT) {
        if (e) atomicAnd(reinterpret_cast<unsigned long long*>(&live_row[r >> 6]), ~(1ull << (r & 63)));
      }
    };
    if (total <= C) {
      // every (changed variable, incident record) pair: flat over the block, one list entry at a time per wavefront group
      for (uint32_t e = 0; e < total; ++e) {
        const uint32_t v = list_id[e], deg = list_deg[e], off = list_off[e];
        for (uint32_t i = tid; i < deg; i += nth) run(v, a.m.adj[off + i]);
      }
    } else {
      // more changed variables than the list holds: every record that touches a changed variable
      for (uint32_t r = tid; r < P; r += nth) {
        const Rec rec = a.m.recs[r];
        const uint32_t x = rec.xk & kSlotMask;
        const bool tern = (rec.xk >> 28) > PCP_LT;
        uint32_t vmin = 0xFFFFFFFFu;
        if ((cur[x >> 5] >> (x & 31)) & 1u) vmin = x;
        if (((cur[rec.y >> 5] >> (rec.y & 31)) & 1u) && rec.y < vmin) vmin = rec.y;
        if (tern && ((cur[rec.z >> 5] >> (rec.z & 31)) & 1u) && rec.z < vmin) vmin = rec.z;
        if (vmin != 0xFFFFFFFFu) run(vmin, r);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t* t = cur; cur = nxt; nxt = t;
  }
  __syncthreads();

  if (PCP_ABLATE & 2048) tph[3] = wall_clock64();
  // ---- phase 3: status.  True iff no propagator is left that is not entailed (store.rs:250-256) -------------------------------
  const bool failed = misc[S_FAIL] != 0;
  if (!failed) {
    if constexpr (IMPLICIT) {
      const SetDom dm{bits, bnd, a.m.const_val, V, sw, a.base, nxt, misc, &narrow};
      for (uint32_t w = wv; w < words; w += nwv) {
        if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&misc[S_OPEN], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) break;
        const uint32_t r = (w << 6) + lane;
        bool open_rec = false;
        if (r < P) open_rec = !eval_set(a.m.recs[r], dm, true);  // at the fixpoint the filters are no-ops: only is_subsumed()
        if (__ballot(open_rec) != 0 && lane == 0) atomicOr(&misc[S_OPEN], 1u);
      }
    } else {
      uint32_t cnt = 0;
      for (uint32_t w = tid; w < words; w += nth) cnt += (uint32_t)__popcll(live_row[w]);
      if (cnt) atomicOr(&misc[S_OPEN], 1u);
    }
  }
  // counters
  for (int o = 32; o > 0; o >>= 1) { narrow += __shfl_down(narrow, o); steps2 += __shfl_down(steps2, o); steps3 += __shfl_down(steps3, o); ev_only += __shfl_down(ev_only, o); }
  if (lane == 0) {
    if (narrow) atomicAdd(&misc[S_NARROW], narrow);
    if (ev_only) atomicAdd(reinterpret_cast<unsigned long long*>(&misc[S_EVAL]), (unsigned long long)ev_only);
    if (steps2) atomicAdd(reinterpret_cast<unsigned long long*>(&misc[S_STEPS2]), (unsigned long long)steps2);
    if (steps3) atomicAdd(reinterpret_cast<unsigned long long*>(&misc[S_STEPS3]), (unsigned long long)steps3);
  }
  __syncthreads();

  if (PCP_ABLATE & 2048) tph[4] = wall_clock64();
  // ---- phase 4: write back ------------------------------------------------------------------------------------------------
  {
    uint64_t* dst = a.bits_out + (size_t)node * V * sw;
    const uint32_t nwords = V * sw;
    if (!(nwords & 1u) && !((size_t)dst & 15)) {
      const uint4* s4 = reinterpret_cast<const uint4*>(bits);
      uint4* d4 = reinterpret_cast<uint4*>(dst);
      for (uint32_t i = tid; i < nwords / 2; i += nth) d4[i] = s4[i];
    } else {
      for (uint32_t i = tid; i < nwords; i += nth) dst[i] = bits[i];
    }
    for (uint32_t v = tid; v < V; v += nth) {
      const int2 b = bnd[v];
      a.lb_out[(size_t)node * V + v] = b.x;
      a.ub_out[(size_t)node * V + v] = b.y;
    }
  }
  if (tid == 0) {
    a.status[node] = failed ? (uint8_t)PCP_FALSE : (misc[S_OPEN] ? (uint8_t)PCP_UNKNOWN : (uint8_t)PCP_TRUE);
    const unsigned long long s2 = *reinterpret_cast<unsigned long long*>(&misc[S_STEPS2]);
    const unsigned long long s3 = *reinterpret_cast<unsigned long long*>(&misc[S_STEPS3]);
    if (s2) atomicAdd((unsigned long long*)&st_slot->steps, s2);
    if (s3) atomicAdd((unsigned long long*)&st_slot->steps3, s3);
    // evaluated = the pairs that were looked at: everything counted as a step except the bulk sweep's credit, plus its own few
    const unsigned long long evo = *reinterpret_cast<unsigned long long*>(&misc[S_EVAL]);
    const unsigned long long looked = s2 + s3 - (misc[S_LIVE] ? (unsigned long long)P : 0ull) + evo;
    if (looked) { atomicAdd((unsigned long long*)&st_slot->evaluated, looked); atomicAdd((unsigned long long*)&st_slot->full_evals, looked); }
    if (misc[S_NARROW]) atomicAdd((unsigned long long*)&st_slot->narrowings, (unsigned long long)misc[S_NARROW]);
    atomicAdd((unsigned long long*)&st_slot->waves, (unsigned long long)(1 + misc[S_WAVES]));
    atomicAdd((unsigned long long*)&st_slot->nodes, 1ull);
    if (failed) atomicAdd((unsigned long long*)&st_slot->failed_nodes, 1ull);
  }
  if (PCP_ABLATE & 2048) {  // staging / sweep / rounds / status / write-back ticks, summed over the nodes
    __syncthreads();
    if (tid == 0) {
      tph[5] = wall_clock64();
      atomicAdd((unsigned long long*)&st_slot->steps3, tph[1] - tph[0]);
      atomicAdd((unsigned long long*)&st_slot->narrowings, tph[2] - tph[1]);
      atomicAdd((unsigned long long*)&st_slot->failed_nodes, tph[3] - tph[2]);
      atomicAdd((unsigned long long*)&st_slot->waves, tph[4] - tph[3]);
      atomicAdd((unsigned long long*)&st_slot->evaluated, tph[5] - tph[4]);
    }
  }
}

// `active` rows of implicit set-mode nodes on request: bit r = record r is not entailed under the final sets.
__global__ void __launch_bounds__(kSetThreads) set_derive_active_kernel(const SetArgs a, uint64_t* live) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t tid = threadIdx.x, nth = blockDim.x, lane = tid & 63;
  const uint32_t wv = tid >> 6, nwv = nth >> 6;
  const uint32_t V = a.m.n_vars, S = a.m.n_slots, sw = a.set_words, P = a.m.n_recs, words = (P + 63) >> 6;
  const SetCarve cv = set_carve(V, S, sw, a.list_cap);
  unsigned long long* bits = reinterpret_cast<unsigned long long*>(smem + cv.bits);
  int2* bnd = reinterpret_cast<int2*>(smem + cv.bnd);
  uint32_t* misc = reinterpret_cast<uint32_t*>(smem + cv.misc);
  uint32_t* scratch = reinterpret_cast<uint32_t*>(smem + cv.chg_a);
  const uint32_t node = blockIdx.x;
  if (tid < 32) misc[tid] = 0;
  const uint64_t* src = a.bits_out + (size_t)node * V * sw;
  for (size_t i = tid; i < (size_t)V * sw; i += nth) bits[i] = src[i];
  __syncthreads();
  for (uint32_t v = tid; v < V; v += nth) bnd[v] = scan_bounds(bits + (size_t)v * sw, sw, a.base);
  __syncthreads();
  uint32_t narrow = 0;
  const SetDom dm{bits, bnd, a.m.const_val, V, sw, a.base, scratch, misc, &narrow};
  for (uint32_t w = wv; w < words; w += nwv) {
    const uint32_t r = (w << 6) + lane;
    bool on = false;
    if (r < P) on = !eval_set(a.m.recs[r], dm, true);
    const uint64_t word = __ballot(on);
    if (lane == 0) live[(size_t)node * words + w] = word;
  }
}

// ------------------------------------------------------------------------------------------------
// The reference's search loop over FDSpace on the device, ONE TREE PER WORKGROUP (pcp_dfs_forest_device_set):
// OneSolution / AllSolution<Propagation<Brancher<FirstSmallestVar, MiddleVal, BinarySplit>>> (search/mod.rs:45-52) over
// VStoreSet = VStoreTrail<IntervalSet<i32>> (variable/mod.rs:38).  The reference restores a node by undoing a TRAIL
// (variable/memory/trail_memory.rs:100-104); so does this kernel: the current node never leaves LDS, every narrowing appends
// (variable, word, removed bits) to the tree's trail in HBM, a backtrack ORs the bits back down to the level's mark and applies the
// right branch.  Per node the HBM traffic is the trail (a few KB after an assignment, a few bytes otherwise) instead of the
// 125 KB-per-node rows of the batched search (pcp_propagate_device + pcp_branch_device_set).
// A child differs from its parent's fixpoint in the branched variable only, so only that variable is marked changed (at a fixpoint
// every other propagator is a no-op until one of its variables changes); a root (pending == kDfsFull) runs the sweep of setfix_kernel.
// Exactly the reference's left-first order within a tree; several trees = the subtrees of an open-node frontier, each on its CU.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kDfsFull = 0xFFFFFFFFu;
constexpr uint32_t kDfsChunk = 8;  // nodes a tree reserves from the forest's counter at a time

struct SetDfsCarve { SetCarve c; size_t touched, adjo, total; };
__host__ __device__ inline SetDfsCarve set_dfs_carve(uint32_t V, uint32_t S, uint32_t sw, uint32_t cap) {
  SetDfsCarve d;
  d.c = set_carve(V, S, sw, cap);
  d.touched = d.c.total;
  d.adjo = (d.touched + ((size_t)(S + 31) / 32) * 4 + 15) & ~(size_t)15;   // a copy of adj_off[V + 1]: no global load in a node's chains for it
  d.total = (d.adjo + ((size_t)V + 1) * 4 + 15) & ~(size_t)15;
  return d;
}

__global__ void __launch_bounds__(kSetThreads) setdfs_kernel(const SetDfsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t tid = threadIdx.x, nth = blockDim.x, lane = tid & 63;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = nth >> 6;
  const uint32_t V = a.m.n_vars, S = a.m.n_slots, sw = a.set_words, C = a.list_cap, P = a.m.n_recs, words = (P + 63) >> 6;
  const uint32_t Wv = (S + 31) >> 5;
  const SetDfsCarve dcv = set_dfs_carve(V, S, sw, C);
  const SetCarve& cv = dcv.c;
  unsigned long long* bits = reinterpret_cast<unsigned long long*>(smem + cv.bits);
  int2* bnd = reinterpret_cast<int2*>(smem + cv.bnd);
  uint32_t* cur = reinterpret_cast<uint32_t*>(smem + cv.chg_a);
  uint32_t* nxt = reinterpret_cast<uint32_t*>(smem + cv.chg_b);
  uint32_t* list_id = reinterpret_cast<uint32_t*>(smem + cv.list_id);
  uint32_t* list_off = reinterpret_cast<uint32_t*>(smem + cv.list_off);
  uint32_t* list_deg = reinterpret_cast<uint32_t*>(smem + cv.list_deg);
  uint32_t* misc = reinterpret_cast<uint32_t*>(smem + cv.misc);
  uint32_t* touched = reinterpret_cast<uint32_t*>(smem + dcv.touched);
  uint32_t* adjo = reinterpret_cast<uint32_t*>(smem + dcv.adjo);
  const uint32_t t = blockIdx.x;
  pcp_stats* const st_slot = a.stats + (blockIdx.x & (kStatSlots - 1));
  uint32_t* const tree = a.tree + (size_t)t * 4;
  uint32_t n_levels = tree[0], pending = tree[2];
  if (tree[3] & 1u) return;  // this tree is finished
  const uint32_t given0 = tree[3] >> 8;  // its oldest levels whose right branch went to another tree (setdfs_split_kernel)
  uint4* const trail = a.trail + (size_t)t * a.trail_cap;
  uint4* const levels = a.levels + (size_t)t * a.level_cap;
  unsigned long long* const gbits = reinterpret_cast<unsigned long long*>(a.bits) + (size_t)t * V * sw;

  // ---- the tree's current node into LDS, its bounds ------------------------------------------------------------------------
  if (tid < 32) misc[tid] = 0;
  for (uint32_t i = tid; i < Wv; i += nth) { cur[i] = 0; nxt[i] = 0; touched[i] = 0; }
  for (uint32_t v = tid; v <= V; v += nth) adjo[v] = a.m.adj_off[v];
  {
    const uint32_t nwords = V * sw;
    if (!(nwords & 1u) && !((size_t)gbits & 15)) {
      const uint4* s4 = reinterpret_cast<const uint4*>(gbits);
      uint4* d4 = reinterpret_cast<uint4*>(bits);
      for (uint32_t i = tid; i < nwords / 2; i += nth) d4[i] = s4[i];
    } else {
      for (uint32_t i = tid; i < nwords; i += nth) bits[i] = gbits[i];
    }
  }
  __syncthreads();
  if (tid == 0) misc[S_TRAILLEN] = tree[1];
  for (uint32_t v = tid; v < V; v += nth) {
    const int2 b = scan_bounds(bits + (size_t)v * sw, sw, a.base);
    bnd[v] = b;
    if (b.x > b.y) atomicOr(&misc[S_FAIL], 1u);
  }
  if (pending != kDfsFull && tid == 0) cur[pending >> 5] = 1u << (pending & 31u);  // (a node persisted by the launch before)
  __syncthreads();

  uint32_t narrow = 0;
  uint64_t ev = 0;
  unsigned long long c_nodes = 0, c_sols = 0, c_fail = 0;
  uint32_t c_err = 0;
  bool finished = false;
  uint32_t reserved = 0;       // nodes this tree may still run before it asks the forest's counter again
  bool res_has_last = false;   // ... the last of which is the node that reaches the limit
  const bool neq_only = a.m.uniform_kind == PCP_NEQ && S == V;
  using DM = SetDomT<true>;
  // a record as seen from variable v's list, entry idx: binary models rebuild it from the 8-byte adjacency payload (one coalesced load
  // instead of an index load and the gather that depends on it)
  const bool use_pay = a.m.adjp != nullptr && !a.m.has_ternary;
  auto rec_at = [&](uint32_t v, uint32_t idx) -> Rec {
    if (use_pay) {
      const uint2 q = a.m.adjp[idx];
      const uint32_t other = q.x & kSlotMask, kind = (q.x >> 28) & 7u;
      const bool is_y = (q.x >> 31) != 0;
      Rec rc;
      rc.xk = (is_y ? other : v) | (kind << 28); rc.y = is_y ? v : other; rc.z = 0; rc.d = (int32_t)q.y;
      return rc;
    }
    return a.m.recs[a.m.adj[idx]];
  };

  // a branch constraint folded into the variable's set: the values lo..hi leave it — one lane per word (one thread doing the up
  // to set_words atomics and trail entries in turn was a quarter of a cheap node)
  auto restrict_var = [&](uint32_t var, long long lo, long long hi) {
    if (wv != 0) return;
    long long b0 = lo - a.base, b1 = hi - a.base;
    if (b0 < 0) b0 = 0;
    if (b1 >= (long long)sw * 64) b1 = (long long)sw * 64 - 1;
    bool changed = false;
    for (long long k = (b0 >> 6) + lane; b0 <= b1 && k <= (b1 >> 6); k += 64) {
      unsigned long long m = ~0ull;
      if (k == (b0 >> 6)) m &= ~0ull << (b0 & 63);
      if (k == (b1 >> 6)) m &= ~0ull >> (63 - (b1 & 63));
      unsigned long long* w = &bits[(size_t)var * sw + k];
      const unsigned long long gone = atomicAnd(w, ~m) & m;
      if (gone) {
        changed = true;
        const uint32_t pos = atomicAdd(&misc[S_TRAILLEN], 1u);
        if (pos < a.trail_cap) trail[pos] = make_uint4(var * sw + (uint32_t)k, var, (uint32_t)gone, (uint32_t)(gone >> 32));
        else atomicOr(&misc[S_TRAILOVF], 1u);
      }
    }
    if (__ballot(changed) != 0 && lane == 0) { atomicOr(&cur[var >> 5], 1u << (var & 31u)); ++narrow; }
  };

  for (uint32_t step = 0; step < a.n_steps; ++step) {
    // ---- may this node run?  (stop flag of the forest, the node limit of all trees together: StopNode, stop_node.rs:57-62) ----
    // Nodes are reserved kDfsChunk at a time from the forest's counter (two device-scope atomics and a barrier per node were a fifth
    // of a cheap node); what a tree does not use goes back when the launch ends.  The node that reaches the limit is explored and
    // counted, but StopNode replaces its status by EndOfSearch (stop_node.rs:55-62): it is neither a solution nor a failure.
    if (reserved == 0) {
      if (tid == 0) {
        uint32_t got = 0, has_last = 0;
        if (!__hip_atomic_load(a.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          const unsigned long long want = min((unsigned long long)kDfsChunk, (unsigned long long)(a.n_steps - step));
          const unsigned long long old = atomicAdd(a.total_nodes, want);
          if (!a.node_limit) got = (uint32_t)want;
          else if (old >= a.node_limit) atomicAdd(a.total_nodes, 0ull - want);
          else {
            got = (uint32_t)min(want, a.node_limit - old);
            if (got < want) atomicAdd(a.total_nodes, 0ull - (want - got));
            has_last = old + got >= a.node_limit ? 1u : 0u;
          }
        }
        misc[S_CTL] = got; misc[S_CTL + 2] = has_last;
      }
      __syncthreads();
      reserved = misc[S_CTL];
      res_has_last = misc[S_CTL + 2] != 0;
      if (!reserved) break;
    }
    --reserved;
    const bool last = res_has_last && reserved == 0;

    // ---- propagate ---------------------------------------------------------------------------------------------------------
    if (pending == kDfsFull && !misc[S_FAIL]) {
      const DM dm{bits, bnd, a.m.const_val, V, sw, a.base, cur, misc, &narrow, trail, &misc[S_TRAILLEN], a.trail_cap};
      bool bulk = false;
      if (neq_only) {  // the lists of the assigned variables stand for the whole sweep (see setfix_kernel)
        if (tid == 0) misc[S_NSING] = 0;
        __syncthreads();
        for (uint32_t v = tid; v < V; v += nth) {
          const int2 b = bnd[v];
          if (b.x == b.y) { const uint32_t pos = atomicAdd(&misc[S_NSING], 1u); if (pos < C) list_id[pos] = v; }
        }
        __syncthreads();
        const uint32_t ns = misc[S_NSING];
        bulk = ns <= C;
        if (bulk)
          for (uint32_t e = 0; e < ns; ++e) {
            const uint32_t v = list_id[e], o0 = adjo[v], o1 = adjo[v + 1];
            for (uint32_t i = o0 + tid; i < o1; i += nth) { (void)eval_set(rec_at(v, i), dm, false); ++ev; }
          }
      }
      if (!bulk)
        for (uint32_t r = tid; r < P; r += nth) { (void)eval_set(a.m.recs[r], dm, false); ++ev; }
      __syncthreads();
    }
    bool runaway = false;
    for (uint32_t round = 0;; ++round) {
      if (round >= (1u << 22)) { runaway = true; break; }  // (a round that runs narrows something: the cap only makes a runaway impossible)
      const uint32_t m_total = (round & 1u) ? S_TOTAL2 : S_TOTAL;
      for (uint32_t v = tid; v < V; v += nth) {
        if (!((cur[v >> 5] >> (v & 31)) & 1u)) continue;
        const int2 b = scan_bounds(bits + (size_t)v * sw, sw, a.base);
        bnd[v] = b;
        if (b.x > b.y) atomicOr(&misc[S_FAIL], 1u);
      }
      __syncthreads();
      for (uint32_t w = tid; w < Wv; w += nth) {
        if (round) nxt[w] = 0;
        uint32_t bitsw = cur[w];
        if (!bitsw) continue;
        uint32_t dropped = 0;
        while (bitsw) {
          const uint32_t v = (w << 5) + __builtin_ctz(bitsw);
          bitsw &= bitsw - 1;
          if (v < V && neq_only) { const int2 b = bnd[v]; if (b.x != b.y) { dropped |= 1u << (v & 31); continue; } }
          const uint32_t pos = atomicAdd(&misc[m_total], 1u);
          if (pos < C) {
            const uint32_t o0 = (v < V) ? adjo[v] : 0u, o1 = (v < V) ? adjo[v + 1] : 0u;
            list_id[pos] = v; list_off[pos] = o0; list_deg[pos] = o1 - o0;
          }
        }
        if (dropped) cur[w] &= ~dropped;
      }
      __syncthreads();
      const uint32_t total = misc[m_total];
      if (total == 0 || misc[S_FAIL]) break;
      if (tid == 0) misc[(round & 1u) ? S_TOTAL : S_TOTAL2] = 0;
      const DM dm{bits, bnd, a.m.const_val, V, sw, a.base, nxt, misc, &narrow, trail, &misc[S_TRAILLEN], a.trail_cap};
      auto run = [&](uint32_t v, const Rec rec) {  // FIFO dedup as in setfix_kernel: the lowest changed variable of a record runs it
        const uint32_t x = rec.xk & kSlotMask;
        const bool tern = (rec.xk >> 28) > PCP_LT;
        if (x < v && ((cur[x >> 5] >> (x & 31)) & 1u)) return;
        if (rec.y < v && ((cur[rec.y >> 5] >> (rec.y & 31)) & 1u)) return;
        if (tern && rec.z < v && ((cur[rec.z >> 5] >> (rec.z & 31)) & 1u)) return;
        ++ev;
        (void)eval_set(rec, dm, false);
      };
      if (total <= C) {
        for (uint32_t e = 0; e < total; ++e) {
          const uint32_t v = list_id[e], deg = list_deg[e], off = list_off[e];
          for (uint32_t i = tid; i < deg; i += nth) run(v, rec_at(v, off + i));
        }
      } else {
        for (uint32_t r = tid; r < P; r += nth) {
          const Rec rec = a.m.recs[r];
          const uint32_t x = rec.xk & kSlotMask;
          const bool tern = (rec.xk >> 28) > PCP_LT;
          uint32_t vmin = 0xFFFFFFFFu;
          if ((cur[x >> 5] >> (x & 31)) & 1u) vmin = x;
          if (((cur[rec.y >> 5] >> (rec.y & 31)) & 1u) && rec.y < vmin) vmin = rec.y;
          if (tern && ((cur[rec.z >> 5] >> (rec.z & 31)) & 1u) && rec.z < vmin) vmin = rec.z;
          if (vmin != 0xFFFFFFFFu) run(vmin, rec);
        }
      }
      __syncthreads();
      uint32_t* sw_ = cur; cur = nxt; nxt = sw_;
    }
    __syncthreads();
    // ---- status (store.rs:250-256): failed / every propagator entailed / open -------------------------------------------------
    // The candidate for branching first — the variable of minimal CARDINALITY > 1, first index (first_smallest_var.rs:30-39 on
    // Domain::size()): an open node almost always has an open record in that variable's list, while the table's first records
    // belong to the variables a dive assigned first and are all entailed.  Only when the list holds no open record is the whole
    // table scanned (exact either way).
    const bool failed = misc[S_FAIL] != 0;
    unsigned long long key = ~0ull;
    if (!failed) {
      for (uint32_t v = tid; v < V; v += nth) {
        unsigned long long size = 0;
        for (uint32_t k = 0; k < sw; ++k) size += (unsigned long long)__popcll(bits[(size_t)v * sw + k]);
        if (size > 1) key = min(key, (size << 32) | v);
      }
      for (int o = 32; o > 0; o >>= 1) key = min(key, (unsigned long long)__shfl_down(key, o));
      unsigned long long* best = reinterpret_cast<unsigned long long*>(list_off);  // (idle between rounds; 16 x 8 bytes)
      if (lane == 0) best[wv] = key;
      __syncthreads();
      key = best[0];
      for (uint32_t w = 1; w < nwv; ++w) key = min(key, best[w]);
      const DM dm{bits, bnd, a.m.const_val, V, sw, a.base, nxt, misc, &narrow, trail, &misc[S_TRAILLEN], a.trail_cap};
      if (key != ~0ull) {
        const uint32_t u = (uint32_t)key, o0 = adjo[u], o1 = adjo[u + 1];
        for (uint32_t i0 = o0 + wv * 64; i0 < o1; i0 += nth) {
          if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&misc[S_OPEN], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) break;
          bool open_rec = false;
          if (i0 + lane < o1) open_rec = !eval_set(rec_at(u, i0 + lane), dm, true);
          if (__ballot(open_rec) != 0 && lane == 0) atomicOr(&misc[S_OPEN], 1u);
        }
      }
      __syncthreads();
      if (!misc[S_OPEN]) {
        for (uint32_t w = wv; w < words; w += nwv) {
          if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&misc[S_OPEN], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) break;
          const uint32_t r = (w << 6) + lane;
          bool open_rec = false;
          if (r < P) open_rec = !eval_set(a.m.recs[r], dm, true);  // at the fixpoint the filters are no-ops: only is_subsumed()
          if (__ballot(open_rec) != 0 && lane == 0) atomicOr(&misc[S_OPEN], 1u);
        }
      }
    }
    __syncthreads();
    if (failed)  // (only a failed node leaves marks behind; its backtrack has barriers before anything is marked again)
      for (uint32_t i = tid; i < Wv; i += nth) { cur[i] = 0; nxt[i] = 0; }
    const bool open = misc[S_OPEN] != 0;
    const uint32_t tlen = misc[S_TRAILLEN];
    if (misc[S_TRAILOVF]) { c_err = 4; break; }  // the trail is full: the tree cannot be restored any more (terminal)
    if (runaway) { c_err = 5; break; }
    ++c_nodes;
    bool descend = false;
    if (failed) {
      if (!last) ++c_fail;
    } else if (!open && last) {
    } else if (!open) {  // a solution (monitor.rs:19-68); the first one of the forest is kept
      ++c_sols;
      if (a.first_solution) {
        if (tid == 0) misc[S_CTL + 1] = atomicCAS(a.solution_flag, 0u, 1u) == 0u ? 1u : 0u;
        __syncthreads();
        if (misc[S_CTL + 1])
          for (uint32_t v = tid; v < V; v += nth) a.first_solution[v] = bnd[v].x;
      }
      if (a.stop_on_solution && tid == 0) atomicExch(a.stop, 1u);
    } else {
      // Brancher<FirstSmallestVar, MiddleVal, BinarySplit>::enter on sets: the candidate found above
      if (key == ~0ull) { c_err = 3; --c_nodes; if (tid == 0) atomicAdd(a.total_nodes, ~0ull); break; }  // Unknown, yet nothing to branch on: the reference panics
      if (n_levels >= a.level_cap) { c_err = 1; --c_nodes; pending = kDfsFull; if (tid == 0) atomicAdd(a.total_nodes, ~0ull); break; }
      const uint32_t var = (uint32_t)key;
      const int2 d = bnd[var];
      const int val = (int)(((long long)d.x + (long long)d.y) / 2);  // MiddleVal (middle_val.rs:25-27)
      if (tid == 0) levels[n_levels] = make_uint4(var, (uint32_t)val, tlen, 0u);
      restrict_var(var, (long long)val + 1, d.y);  // the left child x <= val (binary_split.rs:46-57)
      ++n_levels;
      pending = var;
      descend = true;
    }
    if (!descend) {
      // ---- backtrack: the deepest level whose right branch is still open (a level whose right branch was given to another tree
      // is undone like any other and skipped) ------------------------------------------------------------------------------
      uint4 lv = make_uint4(0u, 0u, 0u, 1u);
      uint32_t from = tlen;
      while (n_levels) {
        --n_levels;
        lv = levels[n_levels];
        for (uint32_t i = lv.z + tid; i < from; i += nth) {
          const uint4 e = trail[i];
          atomicOr(&bits[e.x], ((unsigned long long)e.w << 32) | e.z);
          atomicOr(&touched[e.y >> 5], 1u << (e.y & 31u));
        }
        from = lv.z;
        if (!lv.w) break;
      }
      if (lv.w) { finished = true; break; }  // nothing left that is this tree's
      __syncthreads();
      if (tid == 0) { misc[S_TRAILLEN] = from; misc[S_FAIL] = 0; }
      for (uint32_t v = tid; v < V; v += nth)
        if ((touched[v >> 5] >> (v & 31u)) & 1u) bnd[v] = scan_bounds(bits + (size_t)v * sw, sw, a.base);
      __syncthreads();
      for (uint32_t i = tid; i < Wv; i += nth) touched[i] = 0;
      restrict_var(lv.x, bnd[lv.x].x, (long long)(int)lv.y);  // the right child x > val
      pending = lv.x;
    }
    if (tid == 0) { misc[S_OPEN] = 0; misc[S_TOTAL] = 0; misc[S_TOTAL2] = 0; if (last) atomicExch(a.stop, 1u); }  // (a node's last round leaves its count behind)
    if (last || (a.stop_on_solution && c_sols)) { __syncthreads(); break; }
    __syncthreads();
  }
  if (reserved && tid == 0) atomicAdd(a.total_nodes, 0ull - (unsigned long long)reserved);  // what this tree reserved and did not run

  // ---- persist the tree: its current node, its stacks' lengths, its counters --------------------------------------------------
  __syncthreads();
  {
    const uint32_t nwords = V * sw;
    if (!(nwords & 1u) && !((size_t)gbits & 15)) {
      const uint4* s4 = reinterpret_cast<const uint4*>(bits);
      uint4* d4 = reinterpret_cast<uint4*>(gbits);
      for (uint32_t i = tid; i < nwords / 2; i += nth) d4[i] = s4[i];
    } else {
      for (uint32_t i = tid; i < nwords; i += nth) gbits[i] = bits[i];
    }
  }
  for (int o = 32; o > 0; o >>= 1) { narrow += __shfl_down(narrow, o); ev += __shfl_down(ev, o); }
  if (lane == 0) {
    if (narrow) atomicAdd((unsigned long long*)&st_slot->narrowings, (unsigned long long)narrow);
    if (ev) { atomicAdd((unsigned long long*)&st_slot->evaluated, (unsigned long long)ev); atomicAdd((unsigned long long*)&st_slot->full_evals, (unsigned long long)ev); }
  }
  if (tid == 0) {
    tree[0] = n_levels; tree[1] = misc[S_TRAILLEN]; tree[2] = pending; tree[3] = (finished ? 1u : 0u) | (min(given0, n_levels) << 8);
    unsigned long long* cn = a.counters + (size_t)t * 4;
    cn[0] += c_nodes; cn[1] += c_sols; cn[2] += c_fail;
    if (c_err) cn[3] = c_err;
    // reference-equivalent steps: every propagator of every node once (init_scheduler) — the wake-ups are in `evaluated`
    atomicAdd((unsigned long long*)&st_slot->steps, c_nodes * (unsigned long long)P);
    atomicAdd((unsigned long long*)&st_slot->nodes, c_nodes);
    if (c_fail) atomicAdd((unsigned long long*)&st_slot->failed_nodes, c_fail);
    if (c_err) atomicExch(a.stop, 1u);
  }
}

// A finished tree takes over the OLDEST open right branch of a tree that still has some (the subtree nearest the donor's root: the
// largest it can give).  That node = the donor's current node with its trail undone down to the level's mark (the parent's fixpoint)
// and the right branch applied; it is built straight into the receiver's row.  The donor's level is marked as given (levels[..].w):
// its search loop undoes it like any other level and does not take the right branch.  Runs BETWEEN launches of setdfs_kernel.
__global__ void __launch_bounds__(256) setdfs_split_kernel(const SetDfsArgs a, const uint32_t* __restrict__ pairs, uint32_t* __restrict__ done) {
  const uint32_t d = pairs[2 * blockIdx.x], r = pairs[2 * blockIdx.x + 1], tid = threadIdx.x, nth = blockDim.x;
  const uint32_t V = a.m.n_vars, sw = a.set_words;
  uint32_t* const td = a.tree + (size_t)d * 4;
  uint32_t* const tr = a.tree + (size_t)r * 4;
  const uint32_t n_levels = td[0], tlen = td[1], given = td[3] >> 8;
  const bool ok = !(td[3] & 1u) && (tr[3] & 1u) && given < n_levels && d != r;
  if (!ok) { if (tid == 0) done[blockIdx.x] = 0; return; }
  uint4* const lvp = a.levels + (size_t)d * a.level_cap + given;
  const uint4 lv = *lvp;
  const unsigned long long* src = reinterpret_cast<const unsigned long long*>(a.bits) + (size_t)d * V * sw;
  unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.bits) + (size_t)r * V * sw;
  for (uint32_t i = tid; i < V * sw; i += nth) dst[i] = src[i];
  __syncthreads();
  const uint4* trail = a.trail + (size_t)d * a.trail_cap;
  for (uint32_t i = lv.z + tid; i < tlen; i += nth) {
    const uint4 e = trail[i];
    atomicOr(&dst[e.x], ((unsigned long long)e.w << 32) | e.z);
  }
  __syncthreads();
  // the right child x > val: the values up to val leave the variable's set (binary_split.rs:52-57)
  for (uint32_t k = tid; k < sw; k += nth) {
    const long long t = (long long)(int)lv.y - ((long long)a.base + 64ll * (long long)k);
    const unsigned long long le = t < 0 ? 0ull : (t >= 63 ? ~0ull : ((2ull << t) - 1ull));
    dst[(size_t)lv.x * sw + k] &= ~le;
  }
  if (tid == 0) {
    lvp->w = 1u;
    td[3] = (td[3] & 0xFFu) | ((given + 1) << 8);
    tr[0] = 0; tr[1] = 0; tr[2] = lv.x; tr[3] = 0;
    done[blockIdx.x] = 1;
  }
}

size_t lds_bytes_set_dfs(uint32_t n_vars, uint32_t n_slots, uint32_t set_words, uint32_t list_cap) {
  const SetDfsCarve c = set_dfs_carve(n_vars, n_slots, set_words, list_cap);
  return c.total <= 160 * 1024 ? c.total : 0;
}

hipError_t launch_setdfs(const SetDfsArgs& a, hipStream_t stream) {
  const size_t lds = set_dfs_carve(a.m.n_vars, a.m.n_slots, a.set_words, a.list_cap).total;
  hipError_t e;
  if (lds > 64 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void*>(setdfs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
  hipLaunchKernelGGL(setdfs_kernel, dim3(a.n_trees), dim3(kSetThreads), lds, stream, a);
  return hipGetLastError();
}

hipError_t launch_setdfs_split(const SetDfsArgs& a, uint32_t n_pairs, const uint32_t* pairs, uint32_t* done, hipStream_t stream) {
  if (!n_pairs) return hipSuccess;
  hipLaunchKernelGGL(setdfs_split_kernel, dim3(n_pairs), dim3(256), 0, stream, a, pairs, done);
  return hipGetLastError();
}

hipError_t launch_setfix(const ModelDev& m, uint32_t n_nodes, uint32_t set_words, int32_t base, uint32_t list_cap, const uint64_t* bits_in,
                         uint64_t* bits_out, int32_t* lb_out, int32_t* ub_out, const uint64_t* live_in, uint64_t* live, uint8_t* status,
                         pcp_stats* stats, uint64_t* derive_into, hipStream_t stream) {
  SetArgs a{m, n_nodes, set_words, list_cap, base, bits_in, bits_out, lb_out, ub_out, live_in, live, status, stats};
  const size_t lds = set_carve(m.n_vars, m.n_slots, set_words, list_cap).total;
  hipError_t e;
  if (live) {
    if (lds > 64 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void*>(setfix_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
    hipLaunchKernelGGL(setfix_kernel<false>, dim3(n_nodes), dim3(kSetThreads), lds, stream, a);
  } else {
    if (lds > 64 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void*>(setfix_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
    hipLaunchKernelGGL(setfix_kernel<true>, dim3(n_nodes), dim3(kSetThreads), lds, stream, a);
  }
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if (derive_into) {
    if (lds > 64 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void*>(set_derive_active_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
    hipLaunchKernelGGL(set_derive_active_kernel, dim3(n_nodes), dim3(kSetThreads), lds, stream, a, derive_into);
    e = hipGetLastError();
  }
  return e;
}

// ------------------------------------------------------------------------------------------------
// Brancher<FirstSmallestVar, MiddleVal, BinarySplit>::enter over FDSpace (search/branching/brancher.rs:52-71): the variable of
// minimal CARDINALITY > 1, first index (first_smallest_var.rs:30-39 uses Domain::size()), the value (lower + upper) / 2
// (middle_val.rs:25-27), children `x <= v` / `x > v` (binary_split.rs:46-57) folded into the variable's set.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) set_branch_kernel(uint32_t V, uint32_t sw, int32_t base, uint32_t words, const uint64_t* __restrict__ bits,
                                                         const int32_t* __restrict__ lb, const int32_t* __restrict__ ub,
                                                         const uint64_t* __restrict__ active, const uint32_t* __restrict__ child_base,
                                                         uint64_t* __restrict__ child_bits, uint64_t* __restrict__ child_active,
                                                         const uint32_t* __restrict__ counts, uint32_t reverse) {
  const uint32_t node = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
  const uint32_t slot = child_base[node];
  if (slot == 0xFFFFFFFFu) return;
  const uint32_t rowL = reverse ? counts[0] - 1 - slot : slot;
  const uint32_t rowR = reverse ? rowL - 1 : slot + 1;
  __shared__ unsigned long long best[4];
  const uint64_t* pb = bits + (size_t)node * V * sw;
  unsigned long long key = ~0ull;
  for (uint32_t v = tid; v < V; v += nth) {
    unsigned long long size = 0;
    for (uint32_t k = 0; k < sw; ++k) size += (unsigned long long)__popcll(pb[(size_t)v * sw + k]);
    if (size > 1) key = min(key, (size << 32) | v);
  }
  for (int o = 32; o > 0; o >>= 1) key = min(key, (unsigned long long)__shfl_down(key, o));
  if ((tid & 63) == 0) best[tid >> 6] = key;
  __syncthreads();
  key = best[0];
  for (uint32_t w = 1; w < (nth >> 6); ++w) key = min(key, best[w]);
  const uint32_t var = key == ~0ull ? 0xFFFFFFFFu : (uint32_t)key;
  long long val = 0;
  if (var != 0xFFFFFFFFu) val = ((long long)lb[(size_t)node * V + var] + (long long)ub[(size_t)node * V + var]) / 2;
  uint64_t* b0 = child_bits + (size_t)rowL * V * sw;
  uint64_t* b1 = child_bits + (size_t)rowR * V * sw;
  for (size_t i = tid; i < (size_t)V * sw; i += nth) {
    const uint64_t w = pb[i];
    uint64_t le = ~0ull;  // the values <= val within this word
    if ((uint32_t)(i / sw) == var) {
      const long long t = val - ((long long)base + 64ll * (long long)(i % sw));
      le = t < 0 ? 0ull : (t >= 63 ? ~0ull : ((2ull << t) - 1ull));
      b0[i] = w & le;
      b1[i] = w & ~le;
    } else {
      b0[i] = w;
      b1[i] = w;
    }
  }
  if (active) {
    const uint64_t* pa = active + (size_t)node * words;
    uint64_t* a0 = child_active + (size_t)rowL * words;
    uint64_t* a1 = child_active + (size_t)rowR * words;
    for (uint32_t w = tid; w < words; w += nth) { const uint64_t x = pa[w]; a0[w] = x; a1[w] = x; }
  }
}

hipError_t launch_set_branch(uint32_t n_nodes, uint32_t n_vars, uint32_t set_words, int32_t base, uint32_t words, const uint64_t* bits, const int32_t* lb,
                             const int32_t* ub, const uint64_t* active, const uint8_t* status, uint64_t* child_bits, uint64_t* child_active,
                             uint32_t* child_base, uint32_t* counts, uint32_t reverse, hipStream_t stream) {
  hipError_t e = launch_branch_scan(n_nodes, status, child_base, counts, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(set_branch_kernel, dim3(n_nodes), dim3(256), 0, stream, n_vars, set_words, base, words, bits, lb, ub, active, child_base, child_bits,
                     child_active, counts, reverse);
  return hipGetLastError();
}

}  // namespace pcp
