// pcp_kernels.hip — gfx950 (MI355X, CDNA4) kernels of the propagation fixpoint.
//
// Replaces, for a batch of independent search nodes, the reference's
//   Store::consistency = prepare() + propagation_loop()      (propagation/store.rs:125-164, 247-257)
// i.e. the Reactor/Scheduler loop that pops a propagator, runs propagate()+is_subsumed()
// (store.rs:166-183), unlinks it when entailed (store.rs:200-207) and wakes the propagators of every
// changed variable (store.rs:191-198, reactors/indexed_deps.rs:99-113).
//
// MI355X design:
//  * one workgroup owns the domains of a tile of B nodes in LDS, node-minor, as (-lb, ub) cells — int2, or two int16
//    in one dword when every bound of the tile is within +-16383 (LdsDom16: twice the nodes per LDS byte, packed
//    v_pk_* arithmetic); narrowing is ds_min / a CAS loop; the model is shared by all tiles;
//  * the sweep (the reference's "schedule every active propagator", store.rs:144-149) is a hierarchy of tests that
//    prove (record, node) pairs no-ops in bulk: level -1 one lane per 64-record word x all nodes on range tables of
//    tile summaries (sweep_words), level 0 per record x all nodes on the summaries, level 1 per record x node
//    (fast_signs / fast_signs16), level 2 per node with liveness, then the full filter (eval_record) with LDS atomics.
//    The 64 lanes of a wavefront hold 64 consecutive propagators = one u64 word of a node's `active` BitSet, so
//    entailment is published with one __ballot and one 8-byte store;
//  * later waves visit only the propagators incident to changed variables (CSR var->records), found by a
//    ballot/prefix-sum compaction of the per-node changed-variable bitmask — the IndexedDeps::react step;
//  * when few nodes are in flight (the reference's one-node-per-call use) a node is split over a TEAM of
//    workgroups: each sweeps a slice with a private LDS copy, merges narrowings with device-scope
//    atomicMax/atomicMin, and the last arriver (ticket counter, agent-scope release/acquire) finishes
//    the fixpoint alone — no grid barrier, no co-residency requirement.
// Integer bound work only: no MFMA.  All filters are monotone and contracting, so the wave schedule reaches
// the same greatest fixpoint as the reference's FIFO (SURVEY.md §7 "chaotic-iteration equivalence").
#include <algorithm>
#include <type_traits>
#ifndef PCP_SOLO
#define PCP_SOLO 1
#endif

#include "pcp_internal.h"

// PCP_ABLATE (profiling builds only, tools/ablate.sh, tools/seg_*.sh; results are WRONG or counters overloaded when non-zero):
//   1 no LDS reads in level 1 | 2 no arithmetic in level 1 | 4 no record stream | 8 no live-word I/O | 16 skip the sweep's
//   cold part / phase B | 32 skip the wake-up rounds | 64 block phase timers (+256: staging / whole block) | 128 s_memtime
//   segment timers of the sweeps | 512 count noted words | 1024 / 2048 stop phase B after level 1 / 0 | 4096 / 8192 phase B
//   loads only / preamble only.
#ifndef PCP_ABLATE
#define PCP_ABLATE 0
#endif
// This file is compiled twice (__graft_entry__.build, in parallel): PCP_TU == 0 emits the kernels that work on explicit
// `active` rows plus every utility kernel and host helper; PCP_TU == 1 emits only the IMPLICIT-active instantiations of the
// fixpoint kernel (nodes are domains only, liveness is derived: a propagator is inactive iff it is entailed).
#ifndef PCP_TU
#define PCP_TU 0
#endif

#include "pcp_device.hpp"

namespace pcp {

// LDS carve (all offsets multiples of 16 bytes).  Domains are stored NODE-MINOR: dom[slot * BP + b] with
// BP = B + 2 (B > 1): the B nodes of one slot sit at compile-time immediate offsets of one address register,
// node pairs are 16-byte aligned (one ds_read_b128 reads two nodes), and consecutive slots are (B+2)*8 bytes
// = 12/20/28/36 banks apart for B = 4/8/12/16, so 16 consecutive slots start on 16 distinct 4-bank groups.
constexpr uint32_t kSummMinTile = 4;  // tiles of at least this many nodes keep per-slot summaries
// Packed summary / range tables are indexed through tsw(): one padding dword after every 64 slots.  The level -1 test reads
// T[k][ylo] with ylo = y0 + 64 * lane (consecutive words of an x-block cover consecutive 64-slot y ranges): without the
// padding all lanes of a wavefront hit ONE LDS bank (measured: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 64 %, the whole
// sweep of a near-root tile was that serialisation); with it the stride is 65 dwords and the lanes spread over all banks.
__host__ __device__ inline uint32_t tsw(uint32_t v) { return v + (v >> 6); }
__host__ __device__ inline uint32_t tsw_slots(uint32_t slots) { return slots + (slots >> 6) + 1u; }
__host__ __device__ inline uint32_t rmq_levels(uint32_t word_level, bool max_table) {
  return word_level == 0 ? 1u : ((max_table && word_level < 2) ? 1u : kRangeLevels);
}
struct Carve {
  size_t dom, summ, chg_a, chg_b, list_id, list_pre, list_off, tmp, remaining, misc, total;
};
__host__ __device__ inline uint32_t row_stride(uint32_t B) { return B == 1 ? 1u : B + 2u; }
// packed tiles: dwords per slot; B + 4 = 12/20/36 for B = 8/16/32 keeps rows 16-byte aligned (one ds_read_b128 = four
// nodes) and 16 consecutive slots on 16 distinct 4-bank groups, exactly as B/2 int2 nodes would.
__host__ __device__ inline uint32_t row_stride16(uint32_t B) { return B + 4u; }
// dom_slots: slots whose domains live in LDS (all of them, or only the constants in the global variant);
// mask_slots: slots covered by the changed-variable bitmasks (always all).
__host__ __device__ inline Carve carve(uint32_t dom_slots, uint32_t B, uint32_t list_cap, uint32_t mask_slots, bool packed = false, uint32_t word_level = 0) {
  auto up = [](size_t x) { return (x + 15) & ~(size_t)15; };
  const size_t Wv = (mask_slots + 31) / 32;
  Carve c;
  size_t o = 0;
  c.dom = o; o = up(o + (packed ? (size_t)row_stride16(B) * dom_slots * 4 : (size_t)row_stride(B) * dom_slots * 8));
  // tile summaries (level-0 test of the sweep): two cells per slot, (min -lb, min ub) and (max -lb, max ub) over the nodes
  // packed tiles keep them as two tables Tmin[levels][slots], Tmax[levels][slots] of dwords; with the word-group sweep
  // levels 1..5 hold the minima / maxima over 2^level consecutive slots (range queries of the level -1 test).
  c.summ = o; o = up(o + (B >= kSummMinTile ? (packed ? (size_t)tsw_slots(dom_slots) * 4 * (rmq_levels(word_level, false) + rmq_levels(word_level, true))
                                                     : (size_t)dom_slots * 2 * 8) : 0));
  c.chg_a = o; o = up(o + (size_t)B * Wv * 4);
  c.chg_b = o; o = up(o + (size_t)B * Wv * 4);
  c.list_id = o; o = up(o + (size_t)list_cap * 4);
  c.list_pre = o; o = up(o + ((size_t)list_cap + 1) * 4);
  c.list_off = o; o = up(o + (size_t)list_cap * 4);
  c.tmp = o; o = up(o + 40 * 4);
  c.remaining = o; o = up(o + (size_t)B * 4);
  c.misc = o; o = up(o + 32 * 4);
  c.total = o;
  return c;
}

#if PCP_TU == 0
size_t lds_bytes_for(uint32_t n_slots, uint32_t nodes_per_block, uint32_t list_cap, uint32_t block, bool packed, uint32_t word_level) {
  (void)block;
  Carve c = carve(n_slots, nodes_per_block, list_cap, n_slots, packed, packed ? word_level : 0);
  return c.total <= 160 * 1024 ? c.total : 0;
}
size_t lds_bytes_global(uint32_t n_vars, uint32_t n_slots, uint32_t list_cap) {
  Carve c = carve(n_slots - n_vars, 1, list_cap, n_slots);
  return c.total <= 160 * 1024 ? c.total : 0;
}
#endif  // PCP_TU == 0

// misc[] indices (u32 words; STEPS2/STEPS3 are 64-bit counters occupying two words each)
enum { M_FAIL = 0, M_TOTAL = 1, M_ITEMS = 2, M_ISLAST = 3, M_NARROW = 4, M_WAVES = 5, M_ROUNDMASK = 6, M_OOB = 7, M_STEPS2 = 8, M_STEPS3 = 10, M_HARD = 12, M_EVAL = 14, M_FULL = 16, M_UNK = 18, M_TOTAL2 = 19, M_ITEMS2 = 20, M_ROUNDMASK2 = 21, M_SCAN = 22, M_OPEN0 = 23, M_SOLO = 24, M_BASE_LO = 25, M_BASE_HI = 26, M_WORDS = 28 };

// Summaries of a packed tile: tmin[slot] = (min -lb, min ub), tmax[slot] = (max -lb, max ub) as 16-bit pairs; of an
// unpacked tile: summ[2*slot] = int2 minima, summ[2*slot+1] = int2 maxima.
struct SummPtr {
  const void* a;  // packed: tmin;  unpacked: summ
  const void* b;  // packed: tmax
};
struct BlockCtx {
  void* dom;    // LDS domains [slot][bp]: int2 (-lb,ub), or packed dwords; global variant: the constants' singletons [slot - n_vars]
  uint32_t bp;  // row stride of dom, in cells
  SummPtr summ; // LDS tile summaries (B >= kSummMinTile)
  uint32_t rmq_stride;  // packed: dwords between two levels of a table (= tsw_slots(slots))
  uint32_t S, Wv;
  uint32_t* misc;
  int32_t* glb;  // global variant: this block's node rows in lb_out / ub_out
  int32_t* gub;
  uint32_t V;
  SumTab sums;
  unsigned long long* c10;  // global variant with dom10: the LDS cells (else null)
  int lo10;
};

template <bool PACKED> struct CellOf { using type = int2; };
template <> struct CellOf<true> { using type = uint32_t; };

template <bool GLOBAL, bool PACKED> struct DomOf { using type = LdsDom; };
template <> struct DomOf<false, true> { using type = LdsDom16; };
template <bool PACKED> struct DomOf<true, PACKED> { using type = GlobalDom; };

template <bool GLOBAL, bool PACKED>
__device__ __forceinline__ typename DomOf<GLOBAL, PACKED>::type make_dom(const BlockCtx& k, uint32_t b, uint32_t* chg_next, Ctr* ctr) {
  if constexpr (GLOBAL) {
    return GlobalDom{k.glb, k.gub, static_cast<int2*>(k.dom), k.V, chg_next, &k.misc[M_FAIL], 1u, ctr, k.sums, k.c10, k.lo10};  // one node per block
  } else if constexpr (PACKED) {
    return LdsDom16{static_cast<uint32_t*>(k.dom) + b, k.bp, chg_next + (size_t)b * k.Wv, &k.misc[M_FAIL], 1u << b, ctr};
  } else {
    return LdsDom{static_cast<int2*>(k.dom) + b, k.bp, chg_next + (size_t)b * k.Wv, &k.misc[M_FAIL], 1u << b, ctr, k.sums};
  }
}

// Fast predicate of one binary kind on the domains read for one node: a wave mask straight out of v_cmp.
// A clear bit proves that running the filter on that lane would change NOTHING: no domain narrows, the
// propagator is not entailed, nothing fails — so its live bit and the domains stay as they are.
//  NEQ (x_neq_y.rs:82-93, x_eq_y.rs:87-93): nothing happens iff X.x < Yu && Yl < X.y.  With neither side a
//      singleton this is "overlapping in more than a point" (no singleton => no narrowing; overlap => not
//      entailed); with X = {v} it reads Yl < v < Yu, i.e. v strictly inside Y (Interval::difference leaves Y
//      alone and the two are not disjoint); symmetrically for Y = {u}; two singletons never satisfy it.
//  LT  (x_less_y.rs:87-109): x.ub drops iff X.y >= Yu, y.lb rises iff Yl <= X.x, entailed iff X.y < Yl.
//  EQ  (x_eq_y.rs:87-107): x ∩ y differs from x or y iff a bound differs; entailed iff both are one singleton.
//
// IMPLICIT (no `active` rows: nothing is ever unlinked, so entailment is of no interest during the fixpoint): a clear bit
// only has to prove that no domain narrows and nothing fails.
//  NEQ narrows only with a singleton on one side that equals a bound of the other side: X = {v}, v == Yl or v == Yu, or
//      Y = {u}, u == X.lb or u == X.ub; all four cases (and the failure X = Y = {v}) have X.lb == Yu or X.ub == Yl.
//  LT  narrows iff X.ub >= Yu or Yl <= X.lb (a failure X.lb >= Yu implies the first);  EQ iff a bound differs.
template <int KIND, bool IMPLICIT = false>
__device__ __forceinline__ uint64_t fast_flag(const int2 X, const int Yl, const int Yu) {
  if constexpr (IMPLICIT) {
    if (KIND == PCP_NEQ) return __ballot(X.x == Yu) | __ballot(Yl == X.y);
    if (KIND == PCP_LT) return __ballot(X.y >= Yu) | __ballot(Yl <= X.x);
    return __ballot(X.x != Yl) | __ballot(X.y != Yu);
  }
  if (KIND == PCP_NEQ) return __ballot(X.x >= Yu) | __ballot(Yl >= X.y);
  if (KIND == PCP_LT) return __ballot(X.y >= Yu) | __ballot(Yl <= X.x) | __ballot(X.y < Yl);
  return __ballot(X.x != Yl) | __ballot(X.y != Yu) | __ballot(X.x == X.y);
}

// The lanes on which the propagator is entailed WITHOUT touching a domain (its propagate() is a no-op and is_subsumed()
// is True): such lanes only lose their live bit, which the caller does in bulk.
//  NEQ: the intervals are disjoint (x_neq_y.rs:71-73 via x_eq_y.rs:87-93; Interval::difference of a value outside is a no-op).
//  LT : X.ub < Y.lb  (x_less_y.rs:90-91; then min(X.ub, Yu-1) = X.ub and max(Yl, X.lb+1) = Yl).
//  EQ : both are the same singleton (x_eq_y.rs:87-88).
template <int KIND>
__device__ __forceinline__ uint64_t pure_entailed(const int2 X, const int Yl, const int Yu) {
  if (KIND == PCP_NEQ) return __ballot(X.x > Yu) | __ballot(Yl > X.y);
  if (KIND == PCP_LT) return __ballot(X.y < Yl);
  return __ballot(X.x == X.y && Yl == Yu && X.x == Yl);
}

// The unrolled per-node loop of the fast path.  Nodes are taken four at a time: their x- and y-domains are two
// ds_read_b128 each (nodes b, b+1 of one slot are 16 adjacent, 16-byte-aligned bytes) at immediate offsets of
// one address register, all issued before the first compare so that the LDS latency is paid once per group;
// then two or three compares per node and scalar mask logic.  (word,node) pairs with a flagged live lane are
// returned in the bitmask for the full filter.
template <int KIND, int B, bool PACKED, int NL, bool IMPLICIT>
__device__ __forceinline__ uint32_t fast_nodes(const typename CellOf<PACKED>::type* px, const typename CellOf<PACKED>::type* py, const int d,
                                               const uint64_t (&live)[NL], const uint32_t j) {
  // live[h]: lane (b & 15) * kChunk + j holds word j of node b = 16 h + (b & 15)   (see sweep_fast)
  uint32_t todo = 0;
  if constexpr (B == 1) {
    const uint64_t word = readlane64(live[0], j);
    if (word) {
      const int2 X = px[0], Y = py[0];  // stored as (-lb, ub)
      if (fast_flag<KIND, IMPLICIT>(make_int2(-X.x, X.y), d - Y.x, Y.y + d) & word) todo = 1;
    }
    return todo;
  } else {
    constexpr int G = (B % 4 == 0) ? 4 : 2;
#pragma unroll
    for (int g = 0; g < B; g += G) {
      uint64_t wd[G];
      uint64_t any = 0;
#pragma unroll
      for (int jj = 0; jj < G; ++jj) { wd[jj] = readlane64(live[(g + jj) >> 4], (uint32_t)((g + jj) & 15) * 4u + j); any |= wd[jj]; }
      if (any == 0) continue;
      if constexpr (PACKED) {
        static_assert(!PACKED || G == 4, "packed tiles hold a multiple of four nodes");
        const uint4 Xq = *static_cast<const uint4*>(__builtin_assume_aligned(px + g, 16));
        const uint4 Yq = *static_cast<const uint4*>(__builtin_assume_aligned(py + g, 16));
        const uint32_t xs[4] = {Xq.x, Xq.y, Xq.z, Xq.w}, ys[4] = {Yq.x, Yq.y, Yq.z, Yq.w};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int2 X = unpack16(xs[jj]), Y = unpack16(ys[jj]);
          const uint64_t f = fast_flag<KIND, IMPLICIT>(X, Y.x + d, Y.y + d) & wd[jj];
          todo |= f ? (1u << (g + jj)) : 0u;
        }
      } else {
        int4 Xp[G / 2], Yp[G / 2];
#pragma unroll
        for (int jj = 0; jj < G / 2; ++jj) {
          Xp[jj] = *static_cast<const int4*>(__builtin_assume_aligned(px + g + 2 * jj, 16));
          Yp[jj] = *static_cast<const int4*>(__builtin_assume_aligned(py + g + 2 * jj, 16));
        }
#pragma unroll
        for (int jj = 0; jj < G / 2; ++jj) {
          // LDS holds (-lb, ub)
          const uint64_t f0 = fast_flag<KIND, IMPLICIT>(make_int2(-Xp[jj].x, Xp[jj].y), d - Yp[jj].x, Yp[jj].y + d) & wd[2 * jj];
          const uint64_t f1 = fast_flag<KIND, IMPLICIT>(make_int2(-Xp[jj].z, Xp[jj].w), d - Yp[jj].z, Yp[jj].w + d) & wd[2 * jj + 1];
          todo |= f0 ? (1u << (g + 2 * jj)) : 0u;
          todo |= f1 ? (1u << (g + 2 * jj + 1)) : 0u;
        }
      }
    }
    return __builtin_amdgcn_readfirstlane(todo);
  }
}

// slot * BP (BP = B + 2) with the full-rate 24-bit multiply (slots < 2^26 / B fit): a 32-bit v_mul_lo_u32 is a
// quarter-rate VALU op on gfx950, and hipcc folds shift-add sequences back into it.
template <int B, bool PACKED = false>
__device__ __forceinline__ uint32_t slot_row(uint32_t v) {
  return B == 1 ? v : __umul24(v, (uint32_t)(PACKED ? B + 4 : B + 2));
}

// First-level test of the fast path: ONE sign word per lane for ALL the nodes of the tile, liveness ignored.
// For each node the "nothing happens" condition of fast_flag is rewritten as a conjunction of non-negative
// differences (every bound and offset is below 2^29 in magnitude, so three-term sums cannot wrap).  LDS holds
// (-lb, ub) per (slot,node) — write Xn = -X.lb, Xu = X.ub, Yn = -Y.lb, Yu = Y.ub:
//   NEQ:  X.lb < Y.ub+d && Y.lb+d < X.ub   <=>  Yu + (d-1) + Xn >= 0   &&  Xu + (-d-1) + Yn >= 0      (two v_add3_u32)
//   LT :  X.ub < Y.ub+d && Y.lb+d > X.lb && X.ub >= Y.lb+d
//                                           <=>  (Yu - Xu) + (d-1) >= 0  &&  (Xn - Yn) + (d-1) >= 0  &&  Xu + (-d) + Yn >= 0
// and the terms of all nodes are OR-ed together (v_or3_b32): the sign bit of the result is clear iff nothing happens
// on this lane's record in ANY node.  No scalar work, no cross-lane work: 3 VALU (NEQ) and half a ds_read_b128 per
// (record,node).
// IMPLICIT: XNeqY can only narrow where t1 = Yu + d + Xn or t2 = Xu + Yn - d is exactly zero (fast_flag): the running
// UNSIGNED minimum of the terms over the nodes is zero iff some node has a zero.  XLessY: the entailment term is dropped.
template <int KIND, int B, bool IMPLICIT = false>
__device__ __forceinline__ int fast_signs(const int2* px, const int2* py, const int d) {
  const int c1 = d - 1, c2 = -d - 1, c3 = -d;
  int o = 0;
  if constexpr (IMPLICIT && KIND == PCP_NEQ) {
    uint32_t z1 = 0xffffffffu, z2 = 0xffffffffu;
    if (B == 1) {
      const int2 X = px[0], Y = py[0];
      z1 = (uint32_t)(Y.y + d + X.x); z2 = (uint32_t)(X.y - d + Y.x);
    } else {
#pragma unroll
      for (int g = 0; g < B; g += 2) {
        const int4 Xp = *static_cast<const int4*>(__builtin_assume_aligned(px + g, 16));
        const int4 Yp = *static_cast<const int4*>(__builtin_assume_aligned(py + g, 16));
        z1 = min(min(z1, (uint32_t)(Yp.y + d + Xp.x)), (uint32_t)(Yp.w + d + Xp.z));
        z2 = min(min(z2, (uint32_t)(Xp.y - d + Yp.x)), (uint32_t)(Xp.w - d + Yp.z));
      }
    }
    return (z1 == 0u || z2 == 0u) ? -1 : 0;
  }
  if (PCP_ABLATE & 1) {
    const int f = (int)(size_t)px ^ (int)(size_t)py;
#pragma unroll
    for (int g = 0; g < B; ++g) {
      if (PCP_ABLATE & 2) o |= f + g;
      else o |= ((f + g) + c1 + (f - g)) | ((f ^ g) + c2 + (f + 2 * g));
    }
    return o & 0x7fffffff;
  }
  // The constants are the same for every node of the tile, so the per-node work is one add per term and a running
  // minimum (v_min3_i32 takes two nodes at a time); the constant is added once at the end: some node has
  // t + c < 0  <=>  min(t) + c < 0.  All sums stay below 2^31 (bounds < 2^29, folded offsets < 2^30).
  int m1 = 0x7fffffff, m2 = 0x7fffffff;
  auto two = [&](int xn0, int xu0, int yn0, int yu0, int xn1, int xu1, int yn1, int yu1) {
    if (PCP_ABLATE & 2) { o |= (xn0 ^ yu0) & (xu0 ^ yn0) & (xn1 ^ yu1) & (xu1 ^ yn1) & 0x7fffffff; return; }
    if (KIND == PCP_NEQ) {
      m1 = min(min(m1, yu0 + xn0), yu1 + xn1);
      m2 = min(min(m2, xu0 + yn0), xu1 + yn1);
    } else {
      m1 = min(min(m1, yu0 - xu0), xn0 - yn0);
      m1 = min(min(m1, yu1 - xu1), xn1 - yn1);
      m2 = min(min(m2, xu0 + yn0), xu1 + yn1);
    }
  };
  if (B == 1) {
    const int2 X = px[0], Y = py[0];
    two(X.x, X.y, Y.x, Y.y, X.x, X.y, Y.x, Y.y);
  } else {
#pragma unroll
    for (int g = 0; g < B; g += 2) {
      // node pairs are 16-byte aligned by construction (row stride (B+2)*8 bytes, even g): keep it one ds_read_b128
      const int4 Xp = *static_cast<const int4*>(__builtin_assume_aligned(px + g, 16));
      const int4 Yp = *static_cast<const int4*>(__builtin_assume_aligned(py + g, 16));
      two(Xp.x, Xp.y, Yp.x, Yp.y, Xp.z, Xp.w, Yp.z, Yp.w);
    }
  }
  if (PCP_ABLATE & 2) return o;
  if (KIND == PCP_NEQ) return (m1 + c1) | (m2 + c2);
  if (IMPLICIT) return m1 + c1;  // XLessY narrows iff X.ub >= Yu or Yl <= X.lb; whether it is entailed does not matter
  return (m1 + c1) | (m2 + c3);
}

// The same test on packed tiles (LdsDom16: low half -lb, high half ub).  One v_pk_add_u16 forms both NEQ terms of a
// node — X + swap(Y) = (Xn + Yu, Xu + Yn), the half swap is an op_sel modifier — and one v_pk_min_i16 keeps the two
// running minima: 2 VALU and a quarter of a ds_read_b128 per (record,node).  |bound| <= kPackedMax keeps every sum
// inside int16; the folded offset is added in 32 bits at the end.
// The packed ops are written as inline asm, four nodes (one ds_read_b128 per operand) per block: from the generic vector
// form hipcc scalarised the running minimum into v_add_u16_sdwa / v_lshrrev / v_min3_i16 chains (about twice the
// instructions), and between single-instruction asm statements it pads every dependence with an s_nop.
// op_sel:[0,1] op_sel_hi:[1,0] = (x.lo + y.hi, x.hi + y.lo).
__device__ __forceinline__ void neq4_16(uint32_t& mn, const uint4 X, const uint4 Y) {
  uint32_t t0, t1, t2, t3;
  asm("v_pk_add_u16 %1, %5, %9 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      "v_pk_add_u16 %2, %6, %10 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      "v_pk_add_u16 %3, %7, %11 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      "v_pk_add_u16 %4, %8, %12 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      "v_pk_min_i16 %1, %1, %2\n\t"
      "v_pk_min_i16 %3, %3, %4\n\t"
      "v_pk_min_i16 %0, %0, %1\n\t"
      "v_pk_min_i16 %0, %0, %3"
      : "+v"(mn), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(X.x), "v"(X.y), "v"(X.z), "v"(X.w), "v"(Y.x), "v"(Y.y), "v"(Y.z), "v"(Y.w));
}
__device__ __forceinline__ void lt4_16(uint32_t& mn, uint32_t& mx, uint32_t& mn3, const uint4 X, const uint4 Y) {
  uint32_t t0, t1, t2, t3;
  asm("v_pk_sub_i16 %3, %7, %11\n\t"
      "v_pk_sub_i16 %4, %8, %12\n\t"
      "v_pk_sub_i16 %5, %9, %13\n\t"
      "v_pk_sub_i16 %6, %10, %14\n\t"
      "v_pk_min_i16 %0, %0, %3\n\t"
      "v_pk_max_i16 %1, %1, %3\n\t"
      "v_pk_min_i16 %0, %0, %4\n\t"
      "v_pk_max_i16 %1, %1, %4\n\t"
      "v_pk_min_i16 %0, %0, %5\n\t"
      "v_pk_max_i16 %1, %1, %5\n\t"
      "v_pk_min_i16 %0, %0, %6\n\t"
      "v_pk_max_i16 %1, %1, %6\n\t"
      "v_pk_add_u16 %3, %7, %11 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      "v_pk_add_u16 %4, %8, %12 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      "v_pk_add_u16 %5, %9, %13 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      "v_pk_add_u16 %6, %10, %14 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      "v_pk_min_i16 %3, %3, %4\n\t"
      "v_pk_min_i16 %5, %5, %6\n\t"
      "v_pk_min_i16 %2, %2, %3\n\t"
      "v_pk_min_i16 %2, %2, %5"
      : "+v"(mn), "+v"(mx), "+v"(mn3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(X.x), "v"(X.y), "v"(X.z), "v"(X.w), "v"(Y.x), "v"(Y.y), "v"(Y.z), "v"(Y.w));
}
// IMPLICIT XNeqY: U = X + swap(Y) + (d, -d) = (t1, t2) per node; running unsigned minimum (zero iff some node has a zero)
__device__ __forceinline__ void neq4z_16(uint32_t& mn, const uint4 X, const uint4 Y, const uint32_t c0) {
  uint32_t t0, t1, t2, t3;
  asm("v_pk_add_u16 %1, %5, %9 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      "v_pk_add_u16 %2, %6, %10 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      "v_pk_add_u16 %3, %7, %11 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      "v_pk_add_u16 %4, %8, %12 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      "v_pk_add_u16 %1, %1, %13\n\t"
      "v_pk_add_u16 %2, %2, %13\n\t"
      "v_pk_add_u16 %3, %3, %13\n\t"
      "v_pk_add_u16 %4, %4, %13\n\t"
      "v_pk_min_u16 %1, %1, %2\n\t"
      "v_pk_min_u16 %3, %3, %4\n\t"
      "v_pk_min_u16 %0, %0, %1\n\t"
      "v_pk_min_u16 %0, %0, %3"
      : "+v"(mn), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(X.x), "v"(X.y), "v"(X.z), "v"(X.w), "v"(Y.x), "v"(Y.y), "v"(Y.z), "v"(Y.w), "v"(c0));
}
// IMPLICIT XLessY: only the narrowing terms (running min of Xn - Yn and running max of Xu - Yu)
__device__ __forceinline__ void lt4i_16(uint32_t& mn, uint32_t& mx, const uint4 X, const uint4 Y) {
  uint32_t t0, t1, t2, t3;
  asm("v_pk_sub_i16 %2, %6, %10\n\t"
      "v_pk_sub_i16 %3, %7, %11\n\t"
      "v_pk_sub_i16 %4, %8, %12\n\t"
      "v_pk_sub_i16 %5, %9, %13\n\t"
      "v_pk_min_i16 %0, %0, %2\n\t"
      "v_pk_max_i16 %1, %1, %2\n\t"
      "v_pk_min_i16 %0, %0, %3\n\t"
      "v_pk_max_i16 %1, %1, %3\n\t"
      "v_pk_min_i16 %0, %0, %4\n\t"
      "v_pk_max_i16 %1, %1, %4\n\t"
      "v_pk_min_i16 %0, %0, %5\n\t"
      "v_pk_max_i16 %1, %1, %5"
      : "+v"(mn), "+v"(mx), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(X.x), "v"(X.y), "v"(X.z), "v"(X.w), "v"(Y.x), "v"(Y.y), "v"(Y.z), "v"(Y.w));
}
// (d, -d) as two int16 halves; |d| beyond the packed range can never meet a sum of two packed bounds: any non-zero filler
__device__ __forceinline__ uint32_t pack_c0(int d) {
  const int dc = max(-32767, min(32767, d));
  return ((uint32_t)dc & 0xffffu) | ((uint32_t)(-dc) << 16);
}
__device__ __forceinline__ uint32_t pk_min(uint32_t x, uint32_t y) {
  uint32_t r;
  asm("v_pk_min_i16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ uint32_t pk_max(uint32_t x, uint32_t y) {
  uint32_t r;
  asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ int lo16(uint32_t v) { return (int)(short)(v & 0xffffu); }
__device__ __forceinline__ int hi16(uint32_t v) { return (int)v >> 16; }

template <int KIND, int B, bool IMPLICIT = false>
__device__ __forceinline__ int fast_signs16(const uint32_t* px, const uint32_t* py, const int d) {
  static_assert(B % 4 == 0, "packed tiles hold a multiple of four nodes");
  const int c1 = d - 1, c2 = -d - 1, c3 = -d;
  uint32_t mn = (IMPLICIT && KIND == PCP_NEQ) ? 0xffffffffu : 0x7fff7fffu, mn3 = 0x7fff7fffu, mx = 0x80008000u;
  const uint32_t c0 = pack_c0(d);
#pragma unroll
  for (int g = 0; g < B; g += 4) {
    uint32_t xs[4], ys[4];
    if (PCP_ABLATE & 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { xs[i] = (uint32_t)(size_t)px + g + i; ys[i] = (uint32_t)(size_t)py ^ (g + i); }
    } else {
      const uint4 Xq = *static_cast<const uint4*>(__builtin_assume_aligned(px + g, 16));
      const uint4 Yq = *static_cast<const uint4*>(__builtin_assume_aligned(py + g, 16));
      xs[0] = Xq.x; xs[1] = Xq.y; xs[2] = Xq.z; xs[3] = Xq.w;
      ys[0] = Yq.x; ys[1] = Yq.y; ys[2] = Yq.z; ys[3] = Yq.w;
    }
    if (PCP_ABLATE & 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) mn &= xs[i] ^ ys[i];
      continue;
    }
    const uint4 X4 = make_uint4(xs[0], xs[1], xs[2], xs[3]), Y4 = make_uint4(ys[0], ys[1], ys[2], ys[3]);
    if constexpr (IMPLICIT) {
      if (KIND == PCP_NEQ) neq4z_16(mn, X4, Y4, c0);  // running unsigned min of (t1, t2)
      else lt4i_16(mn, mx, X4, Y4);
    } else {
      if (KIND == PCP_NEQ) neq4_16(mn, X4, Y4);   // running min of (Xn + Yu, Xu + Yn)
      else lt4_16(mn, mx, mn3, X4, Y4);           // mn.lo: min (Xn - Yn); mx.hi: max (Xu - Yu); mn3.hi: min (Xu + Yn)
    }
  }
  if constexpr (IMPLICIT) {
    if (KIND == PCP_NEQ) return ((mn & 0xffffu) == 0u || (mn >> 16) == 0u) ? -1 : 0;
    return (lo16(mn) + c1) | (c1 - hi16(mx));
  }
  if (KIND == PCP_NEQ) return (lo16(mn) + c1) | (hi16(mn) + c2);
  return (lo16(mn) + c1) | (c1 - hi16(mx)) | (hi16(mn3) + c3);
}

// ---- the same test in pipelined form (whole-chunk fast block of sweep_fast) -----------------------------------------
// A unit = the rows of one record for 8 packed / 4 unpacked nodes = 2 + 2 ds_read_b128 = 16 VGPRs.  The fast block walks
// the units of a chunk with two row buffers: the reads of unit u+1 are issued before the arithmetic of unit u.
constexpr int kUnitRows = 2;  // ds_read_b128 per operand and unit
struct Rows { uint4 x[kUnitRows], y[kUnitRows]; };
struct SignAcc { uint32_t a, b, c; };  // packed: mn, mx, mn3;  unpacked: m1, m2 (as int bits)
template <bool PACKED> struct UnitNodes { static constexpr int value = (PACKED ? 4 : 2) * kUnitRows; };

template <bool PACKED>
__device__ __forceinline__ void load_unit(Rows& r, const typename CellOf<PACKED>::type* px, const typename CellOf<PACKED>::type* py, int g0) {
  // g0: first node of the unit; 16 bytes = 4 packed cells or 2 int2 cells
  constexpr int kPer = PACKED ? 4 : 2;
#pragma unroll
  for (int i = 0; i < kUnitRows; ++i) {
    r.x[i] = *static_cast<const uint4*>(__builtin_assume_aligned(px + g0 + kPer * i, 16));
    r.y[i] = *static_cast<const uint4*>(__builtin_assume_aligned(py + g0 + kPer * i, 16));
  }
}
template <int KIND, bool PACKED, bool IMPLICIT>
__device__ __forceinline__ SignAcc acc_init() {
  if (IMPLICIT && KIND == PCP_NEQ) return SignAcc{0xffffffffu, 0xffffffffu, 0u};  // running unsigned minima
  if (PACKED) return SignAcc{0x7fff7fffu, 0x80008000u, 0x7fff7fffu};
  return SignAcc{0x7fffffffu, 0x7fffffffu, 0u};
}
template <int KIND, bool PACKED, bool IMPLICIT>
__device__ __forceinline__ void accumulate(SignAcc& s, const Rows& r, const int d) {
#pragma unroll
  for (int i = 0; i < kUnitRows; ++i) {
    if constexpr (PACKED) {
      if constexpr (IMPLICIT) {
        if (KIND == PCP_NEQ) neq4z_16(s.a, r.x[i], r.y[i], pack_c0(d));
        else lt4i_16(s.a, s.b, r.x[i], r.y[i]);
      } else {
        if (KIND == PCP_NEQ) neq4_16(s.a, r.x[i], r.y[i]);
        else lt4_16(s.a, s.b, s.c, r.x[i], r.y[i]);
      }
    } else if constexpr (IMPLICIT && KIND == PCP_NEQ) {
      const int xn0 = (int)r.x[i].x, xu0 = (int)r.x[i].y, xn1 = (int)r.x[i].z, xu1 = (int)r.x[i].w;
      const int yn0 = (int)r.y[i].x, yu0 = (int)r.y[i].y, yn1 = (int)r.y[i].z, yu1 = (int)r.y[i].w;
      s.a = min(min(s.a, (uint32_t)(yu0 + d + xn0)), (uint32_t)(yu1 + d + xn1));
      s.b = min(min(s.b, (uint32_t)(xu0 - d + yn0)), (uint32_t)(xu1 - d + yn1));
    } else {
      // (-lb, ub) pairs of two nodes per 16 bytes
      const int xn0 = (int)r.x[i].x, xu0 = (int)r.x[i].y, xn1 = (int)r.x[i].z, xu1 = (int)r.x[i].w;
      const int yn0 = (int)r.y[i].x, yu0 = (int)r.y[i].y, yn1 = (int)r.y[i].z, yu1 = (int)r.y[i].w;
      int m1 = (int)s.a, m2 = (int)s.b;
      if (KIND == PCP_NEQ) {
        m1 = min(min(m1, yu0 + xn0), yu1 + xn1);
        m2 = min(min(m2, xu0 + yn0), xu1 + yn1);
      } else {
        m1 = min(min(m1, yu0 - xu0), xn0 - yn0);
        m1 = min(min(m1, yu1 - xu1), xn1 - yn1);
        m2 = min(min(m2, xu0 + yn0), xu1 + yn1);
      }
      s.a = (uint32_t)m1; s.b = (uint32_t)m2;
    }
  }
}
template <int KIND, bool PACKED, bool IMPLICIT>
__device__ __forceinline__ int acc_finish(const SignAcc& s, const int d) {
  const int c1 = d - 1, c2 = -d - 1, c3 = -d;
  if constexpr (IMPLICIT) {
    if constexpr (KIND == PCP_NEQ) {
      if (PACKED) return ((s.a & 0xffffu) == 0u || (s.a >> 16) == 0u) ? -1 : 0;
      return (s.a == 0u || s.b == 0u) ? -1 : 0;
    } else {
      if (PACKED) return (lo16(s.a) + c1) | (c1 - hi16(s.b));
      return (int)s.a + c1;
    }
  }
  if constexpr (PACKED) {
    if (KIND == PCP_NEQ) return (lo16(s.a) + c1) | (hi16(s.a) + c2);
    return (lo16(s.a) + c1) | (c1 - hi16(s.b)) | (hi16(s.c) + c3);
  } else {
    if (KIND == PCP_NEQ) return ((int)s.a + c1) | ((int)s.b + c2);
    return ((int)s.a + c1) | ((int)s.b + c3);
  }
}
// o[j] for the four words of a chunk, units double-buffered.
template <int KIND, int B, bool PACKED, bool IMPLICIT>
__device__ __forceinline__ void chunk_signs(int (&o)[4], const typename CellOf<PACKED>::type* const (&px)[4],
                                            const typename CellOf<PACKED>::type* const (&py)[4], const int (&d)[4]) {
  constexpr int UN = UnitNodes<PACKED>::value, UPW = B / UN, U = 4 * UPW;
  static_assert(B % UN == 0 && UPW >= 1, "tile must be a whole number of units");
  Rows r[2];
  SignAcc acc = acc_init<KIND, PACKED, IMPLICIT>();
  load_unit<PACKED>(r[0], px[0], py[0], 0);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (u + 1 < U) load_unit<PACKED>(r[(u + 1) & 1], px[(u + 1) / UPW], py[(u + 1) / UPW], ((u + 1) % UPW) * UN);
    accumulate<KIND, PACKED, IMPLICIT>(acc, r[u & 1], d[u / UPW]);
    if (u % UPW == UPW - 1) { o[u / UPW] = acc_finish<KIND, PACKED, IMPLICIT>(acc, d[u / UPW]); acc = acc_init<KIND, PACKED, IMPLICIT>(); }
    __builtin_amdgcn_sched_barrier(0);  // keep the source order: hipcc otherwise clusters the reads of a whole word up front
  }
}

// OR of a 32-bit value over the 16 lanes of each DPP row (quad swaps, then half-mirror, then mirror).
__device__ __forceinline__ uint32_t row_or16(uint32_t v) {
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);  // row_mirror
  return v;
}

// ------------------------------------------------------------------------------------------------
// Wave 0: one pass over 64-record words [w0, w1) for the nb nodes of this block — the reference's
// init_scheduler (store.rs:144-149): every live propagator runs once.  The 64 lanes of a wavefront hold 64
// consecutive records == one u64 word of each node's live mask; lane b also carries node b's word.
//
// Fast path (all 64 records of one binary kind, no failed node yet): per-lane predicates become wave masks
// and only (word,node) pairs in which some live lane would narrow are re-run with the full filter and its
// LDS atomics.  The fast predicates restate exactly the no-op conditions of XNeqY/XLessY/XEqY::propagate and
// the True case of their is_subsumed (files cited in eval_record).
// ------------------------------------------------------------------------------------------------
// Each wavefront owns CHUNK consecutive 64-record words at a time.  For the live-mask I/O the lanes are laid out as
// lane = node*CHUNK + j: one 8-byte load (and one store) per lane moves the CHUNK words of every node of the tile —
// 32 contiguous bytes per node instead of a lone 8-byte access per (word,node), which the memory side turns into a
// 32-byte transaction each (measured: WRITE_SIZE 3.4x the bytes stored).  Word j of node b is then lane b*CHUNK+j.
constexpr int kChunk = 4;

// OR of a 32-bit value over all lanes congruent to this lane modulo 4 (the kChunk lanes-per-node layout): two DPP row
// rotations combine the four quads of each 16-lane row, v_permlane16_swap / v_permlane32_swap (gfx950) combine the rows.
__device__ __forceinline__ uint32_t or_mod4(uint32_t v) {
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, true);  // row_ror:4
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true);  // row_ror:8
  const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = r[0] | r[1];
  const auto q = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return q[0] | q[1];
}
__device__ __forceinline__ uint64_t or_mod4_64(uint64_t v) { return ((uint64_t)or_mod4((uint32_t)(v >> 32)) << 32) | or_mod4((uint32_t)v); }

__device__ __forceinline__ Rec expand(const Rec8 q) {
  Rec r;
  r.xk = (q.xyk & 0x7fffu) | ((q.xyk >> 30) << 28);
  r.y = (q.xyk >> 15) & 0x7fffu;
  r.z = 0;
  r.d = q.d;
  return r;
}
__device__ __forceinline__ Rec expand(const Rec r) { return r; }

// Level-0 test of the sweep: the level-1 conditions evaluated ONCE for the whole tile on per-slot summaries
// S0 = (min over nodes of -lb, min over nodes of ub), S1 = (max ..., max ...), taken when the domains were staged.  The minimum
// of a sum is at least the sum of the minima (and min (a - b) >= min a - max b), so a non-negative result proves that the
// record is a no-op in EVERY node of the tile: 2 LDS reads and ~5 VALU per record instead of 2 VALU per (record,node).
// The summaries are not maintained while the domains narrow: a record that touches a variable narrowed during the
// sweep is re-run by the wake-up rounds anyway, so a stale summary is the same race as reading the domain a moment
// before the narrowing.
// IMPLICIT: XNeqY narrows only where t1 = Xn + Yu + d or t2 = Xu + Yn - d is exactly zero: a term whose lower bound over the
// tile (sum of the minima) is positive or whose upper bound (sum of the maxima) is negative has no zero in any node — this also
// clears records that are ENTAILED in every node of the tile, which the explicit form (it must unlink them) cannot.
__device__ __forceinline__ int neq_no_zero(const uint32_t tlo, const uint32_t thi, const int dmin, const int dmax) {
  // tlo <= (Xn + Yu, Xu + Yn) <= thi over the nodes; offsets in [dmin, dmax].  < 0 iff a zero cannot be excluded
  const bool z1 = (lo16(tlo) + dmin > 0) || (lo16(thi) + dmax < 0);
  const bool z2 = (hi16(tlo) - dmax > 0) || (hi16(thi) - dmin < 0);
  return (z1 && z2) ? 0 : -1;
}
template <int KIND, bool PACKED, bool IMPLICIT>
__device__ __forceinline__ void level0_chunk(int (&o)[4], const SummPtr sp, const uint32_t (&x)[4], const uint32_t (&y)[4], const int (&d)[4]) {
  if constexpr (PACKED) {
    const uint32_t* tmin = static_cast<const uint32_t*>(sp.a);
    const uint32_t* tmax = static_cast<const uint32_t*>(sp.b);
    if constexpr (IMPLICIT && KIND == PCP_NEQ) {
      uint32_t xn[4], xx[4], yn[4], yx[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { xn[j] = tmin[tsw(x[j])]; xx[j] = tmax[tsw(x[j])]; yn[j] = tmin[tsw(y[j])]; yx[j] = tmax[tsw(y[j])]; }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t tlo, thi;
        asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(tlo) : "v"(xn[j]), "v"(yn[j]));
        asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(thi) : "v"(xx[j]), "v"(yx[j]));
        o[j] = neq_no_zero(tlo, thi, d[j], d[j]);
      }
    } else if (KIND == PCP_NEQ) {  // only the minima: (Xn + Yu, Xu + Yn) in one packed add
      uint32_t xs[4], ys[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { xs[j] = tmin[tsw(x[j])]; ys[j] = tmin[tsw(y[j])]; }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t t;
        asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(xs[j]), "v"(ys[j]));
        o[j] = (lo16(t) + (d[j] - 1)) | (hi16(t) + (-d[j] - 1));
      }
    } else {
      uint32_t xn[4], xx[4], yn[4], yx[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { xn[j] = tmin[tsw(x[j])]; xx[j] = tmax[tsw(x[j])]; yn[j] = tmin[tsw(y[j])]; yx[j] = tmax[tsw(y[j])]; }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c1 = d[j] - 1, c3 = -d[j];
        o[j] = (hi16(yn[j]) - hi16(xx[j]) + c1) | (lo16(xn[j]) - lo16(yx[j]) + c1) | (IMPLICIT ? 0 : (hi16(xn[j]) + lo16(yn[j]) + c3));
      }
    }
  } else {
    const int4* summ = static_cast<const int4*>(sp.a);  // (min n, min u, max n, max u) per slot
    int4 X[4], Y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { X[j] = summ[x[j]]; Y[j] = summ[y[j]]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c1 = d[j] - 1, c2 = -d[j] - 1, c3 = -d[j];
      if constexpr (IMPLICIT) {
        if (KIND == PCP_NEQ) {
          const bool z1 = (X[j].x + Y[j].y + d[j] > 0) || (X[j].z + Y[j].w + d[j] < 0);
          const bool z2 = (X[j].y + Y[j].x - d[j] > 0) || (X[j].w + Y[j].z - d[j] < 0);
          o[j] = (z1 && z2) ? 0 : -1;
        } else {
          o[j] = (Y[j].y - X[j].w + c1) | (X[j].x - Y[j].z + c1);
        }
      } else if (KIND == PCP_NEQ) o[j] = (X[j].x + Y[j].y + c1) | (X[j].y + Y[j].x + c2);
      else o[j] = (Y[j].y - X[j].w + c1) | (X[j].x - Y[j].z + c1) | (X[j].y + Y[j].x + c3);
    }
  }
}

template <int B, bool GLOBAL, bool COMPACT, bool PACKED, bool IMPLICIT>
__device__ __forceinline__ void sweep_fast(const LaunchArgs& a, const BlockCtx& k, uint32_t w0, uint32_t w1, uint32_t node0, uint32_t nb,
                                           uint32_t* chg_next, uint32_t* remaining, uint64_t& steps2, uint64_t& steps3, Ctr& ctr,
                                           const uint64_t* hard64 = nullptr) {
  // hard64 != nullptr: second pass behind phase A of sweep_words (which has counted the live records, copied the live
  // rows to a.live and noted in hard64 the words its range test could not clear): only chunks with a noted word are
  // processed, in place on a.live, and `remaining` is corrected by the records that get unlinked.
  const bool post_a = hard64 != nullptr;
  static_assert(B <= 32 && (B <= 16 || PACKED), "node*CHUNK+j must fit in the 64 lanes of one or two live registers");
  static_assert(kChunk == 4, "the hot loop is written for four record buffers");
  constexpr int NL = B > 16 ? 2 : 1;  // live registers per lane: register h carries nodes 16h .. 16h+15
  using Cell = typename CellOf<PACKED>::type;
  const Cell* const kdom = static_cast<const Cell*>(k.dom);
  const uint32_t lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps the loop control scalar
  const uint32_t P = a.m.n_recs, words = (P + 63) >> 6;
  const uint64_t tail_mask = (P & 63) ? ((1ull << (P & 63)) - 1) : ~0ull;
  // IMPLICIT: no live rows anywhere — every record of every node is taken as live (an entailed record's filter is a no-op)
  const uint64_t* live_src = IMPLICIT ? nullptr : (post_a ? a.live : a.live_in);
  const uint32_t bq = lane / kChunk, jq = lane % kChunk;  // this lane's (node mod 16, word-in-chunk) for the live-mask I/O
  bool io[NL];
  const uint64_t* my_in[NL];
  uint64_t* my_out[NL];
#pragma unroll
  for (int h = 0; h < NL; ++h) {
    io[h] = bq + 16u * h < nb;
    const uint32_t my_node = node0 + (io[h] ? bq + 16u * h : 0u);
    my_in[h] = live_src ? live_src + (size_t)my_node * words : nullptr;
    my_out[h] = IMPLICIT ? nullptr : a.live + (size_t)my_node * words;
  }
  const uint32_t c0 = w0 / kChunk, c1 = (w1 + kChunk - 1) / kChunk;  // w0 is a multiple of kChunk
  // The stream is latency-bound, not bandwidth-bound: with one 1-KiB record load in flight per wavefront a CU moves
  // 16 KiB per memory round trip (measured: the loop skeleton alone took 73 % of the kernel).  So every wavefront
  // keeps FOUR record loads (one whole chunk ahead) plus the next chunk's live words in flight.  All stream loads
  // are UNCONDITIONAL (indices clamped into the buffers, validity applied afterwards) and sit in straight-line code:
  // hipcc can then count them and wait with vmcnt(N>0) for the oldest load while the younger ones stay in flight; a
  // load under a divergent guard makes it fall back to vmcnt(0).
  const uint32_t last_word = words - 1;
  auto fetch_live = [&](uint32_t c, int h) -> uint64_t {
    const uint32_t w = c * kChunk + jq;
    const uint64_t v = (my_in[h] && !(PCP_ABLATE & 8)) ? my_in[h][min(w, last_word)] : ~0ull;
    const bool ok = c < c1 && io[h] && w < w1;
    return ok ? (w == last_word ? v & tail_mask : v) : 0ull;
  };
  using RecT = typename std::conditional<COMPACT, Rec8, Rec>::type;
  const RecT* rec_stream;
  if constexpr (COMPACT) rec_stream = a.m.recs8; else rec_stream = a.m.recs;
  auto fetch_rec = [&](uint32_t w) -> RecT {  // w is wave-uniform; the tables are padded (kStreamPadRecs): no clamping
    if (PCP_ABLATE & 4) {  // profiling: no record stream (a synthetic NEQ record)
      const uint32_t r = (w << 6) + lane;
      RecT q;
      if constexpr (COMPACT) { q.xyk = (r & 511u) | (((r >> 3) & 511u) << 15); q.d = (int)(r & 7u); }
      else { q.xk = r & 511u; q.y = (r >> 3) & 511u; q.z = 0; q.d = (int)(r & 7u); }
      return q;
    }
    return (rec_stream + (size_t)w * 64)[lane];
  };
  // word j of node b sits in lane (b & 15) * kChunk + j of live register b >> 4
  auto word_of = [&](const uint64_t (&reg)[NL], uint32_t b, uint32_t j) -> uint64_t {
    const uint32_t l = (b & 15u) * kChunk + j;
    if (NL == 1) return readlane64(reg[0], l);
    const uint64_t lo = readlane64(reg[0], l), hi = readlane64(reg[NL - 1], l);
    return b < 16u ? lo : hi;
  };
  uint32_t steps_lane = 0, rem_acc[NL];
#pragma unroll
  for (int h = 0; h < NL; ++h) rem_acc[h] = 0;
  // Two stages in ping-pong: while one chunk is processed the loads of this wavefront's next chunk are in flight, and
  // each stage is reloaded only after its last use, straight into the registers it is read from.  (A single rotating
  // buffer looked equivalent in the source, but its reload overlapped the last use of the old value, so hipcc loaded
  // into fresh registers and ended every iteration with s_waitcnt vmcnt(0) + copies: the prefetch was waited for in
  // the iteration that issued it.)
  struct Stage { RecT buf[kChunk]; uint64_t live[NL]; };
  auto issue = [&](Stage& st, uint32_t c) {
#pragma unroll
    for (int h = 0; h < NL; ++h) st.live[h] = fetch_live(c, h);
#pragma unroll
    for (int j = 0; j < kChunk; ++j) st.buf[j] = fetch_rec(c * kChunk + j);
  };
  uint64_t seg[4] = {0, 0, 0, 0}, segw = 0;  // PCP_ABLATE & 128: s_memtime ticks per segment of process()
  auto process = [&](const uint32_t c, const Stage& st) {
    if (post_a && ((hard64[c >> 4] >> ((c & 15u) * 4u)) & 0xFull) == 0) return;  // nothing noted in this chunk
    const uint64_t tm0 = (PCP_ABLATE & 128) ? __builtin_amdgcn_s_memtime() : 0;
    uint64_t loaded[NL], my_new[NL];
    uint64_t any_live = 0;
#pragma unroll
    for (int h = 0; h < NL; ++h) {
      loaded[h] = st.live[h];
      my_new[h] = loaded[h];
      any_live |= loaded[h];
    }
    const uint32_t failm = __hip_atomic_load(&k.misc[M_FAIL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    // alive4: lane l holds the OR over the tile's nodes of word (l & 3) — the records of that word that are live in
    // at least one node (a record that is dead in every node of the tile must not drag its word onto the cold path).
    // Computed on demand: the common chunk is cleared by level 0 without looking at the live words at all.
    uint64_t alive4 = 0;
    bool have_alive = false;
    auto get_alive = [&]() { if (!have_alive) { alive4 = or_mod4_64(any_live); have_alive = true; } };
    uint64_t tm1 = 0;
    if (PCP_ABLATE & 128) { tm1 = __builtin_amdgcn_s_memtime() + (__builtin_amdgcn_readfirstlane((uint32_t)any_live) & 0u); seg[0] += tm1 - tm0; }
    // ---- hot part: level-1 test of the four words; anything else is only noted in `slow` -------------------------
    uint32_t slow = 0;
    // Whole-chunk fast block: the 4 x 64 records are of ONE binary kind (NEQ or LT) and no node of the tile has failed.
    // Level 0 (tile summaries) clears most words outright; the per-node level 1 runs only for what is left — as one
    // straight-line pipelined block when all four words need it, else word by word.
    bool chunk_fast = false;
    uint32_t ckind = 0;
    if constexpr (!GLOBAL) {
      const bool shape_ok = failm == 0 && c * kChunk + (kChunk - 1) < w1;
      if (a.m.uniform_kind <= PCP_LT) {  // the host has checked the whole table
        ckind = a.m.uniform_kind;
        chunk_fast = shape_ok;
      } else if (shape_ok && a.m.sums.count == 0) {
        uint32_t k_or = 0, k_and = ~0u;
#pragma unroll
        for (int j = 0; j < kChunk; ++j) {
          uint32_t kj;
          if constexpr (COMPACT) kj = st.buf[j].xyk >> 30; else kj = st.buf[j].xk >> 28;
          k_or |= kj; k_and &= kj;
        }
        ckind = __builtin_amdgcn_readfirstlane(k_or);
        chunk_fast = __all(k_or == k_and && k_or == ckind) && (ckind == PCP_NEQ || ckind == PCP_LT);
      }
    }
    if (chunk_fast) {
#pragma unroll
      for (int h = 0; h < NL; ++h) steps_lane += post_a ? 0u : (uint32_t)__popcll(loaded[h]);  // every live record of every node runs once
      uint64_t tw = 0;
      if (PCP_ABLATE & 128) tw = __builtin_amdgcn_s_memtime() + (__builtin_amdgcn_readfirstlane(st.buf[0].d) & 0);
      // level 0: the whole tile at once, on the per-slot summaries
      uint32_t need = 0;  // words with a live record that level 0 could not clear
      if constexpr (B >= (int)kSummMinTile) {
        int o0[kChunk];
        uint32_t sx[kChunk], sy[kChunk];
        int d0[kChunk];
#pragma unroll
        for (int j = 0; j < kChunk; ++j) {
          const Rec rec = expand(st.buf[j]);
          sx[j] = rec.xk & kSlotMask;
          sy[j] = rec.y;
          d0[j] = rec.d;
        }
        // one kind branch per chunk, and the eight summary reads of a chunk go out together
        if (ckind == PCP_NEQ) level0_chunk<PCP_NEQ, PACKED, IMPLICIT>(o0, k.summ, sx, sy, d0);
        else level0_chunk<PCP_LT, PACKED, IMPLICIT>(o0, k.summ, sx, sy, d0);
        uint64_t bal[kChunk];
        bool any0 = false;
#pragma unroll
        for (int j = 0; j < kChunk; ++j) { bal[j] = __ballot(o0[j] < 0); any0 |= bal[j] != 0; }
        if (any0) {
          get_alive();
#pragma unroll
          for (int j = 0; j < kChunk; ++j)
            if (bal[j] & readlane64(alive4, j)) need |= 1u << j;
        }
      } else {
        get_alive();
#pragma unroll
        for (int j = 0; j < kChunk; ++j)
          if (readlane64(alive4, j)) need |= 1u << j;
      }
      // level 1: every node of the tile, for the words level 0 left over
#pragma unroll
      for (int h = 0; h < NL; ++h) ctr.ev += ((need >> jq) & 1u) ? (uint32_t)__popcll(loaded[h]) : 0u;
      bool done1 = false;
      if constexpr (B % UnitNodes<PACKED>::value == 0) {
        if (need == 0xFu) {
          int o[kChunk];
          const Cell* px[kChunk];
          const Cell* py[kChunk];
          int dd[kChunk];
#pragma unroll
          for (int j = 0; j < kChunk; ++j) {
            const Rec rec = expand(st.buf[j]);
            px[j] = kdom + slot_row<B, PACKED>(rec.xk & kSlotMask);
            py[j] = kdom + slot_row<B, PACKED>(rec.y);
            dd[j] = rec.d;
          }
          if (ckind == PCP_NEQ) chunk_signs<PCP_NEQ, B, PACKED, IMPLICIT>(o, px, py, dd);
          else chunk_signs<PCP_LT, B, PACKED, IMPLICIT>(o, px, py, dd);
#pragma unroll
          for (int j = 0; j < kChunk; ++j)
            if (__ballot(o[j] < 0) & readlane64(alive4, j)) slow |= 1u << j;
          done1 = true;
        }
      }
      if (!done1) {
        while (need) {  // rolled: one copy of the per-node test
          const uint32_t j = __builtin_ctz(need);
          need &= need - 1;
          RecT rsel = st.buf[0];  // the stage register of word j, selected without a dynamic register index
#pragma unroll
          for (int u = 1; u < kChunk; ++u)
            if (j == (uint32_t)u) rsel = st.buf[u];
          const Rec rec = expand(rsel);
          const Cell* px = kdom + slot_row<B, PACKED>(rec.xk & kSlotMask);
          const Cell* py = kdom + slot_row<B, PACKED>(rec.y);
          int o;
          if constexpr (PACKED) o = (ckind == PCP_NEQ) ? fast_signs16<PCP_NEQ, B, IMPLICIT>(px, py, rec.d) : fast_signs16<PCP_LT, B, IMPLICIT>(px, py, rec.d);
          else o = (ckind == PCP_NEQ) ? fast_signs<PCP_NEQ, B, IMPLICIT>(px, py, rec.d) : fast_signs<PCP_LT, B, IMPLICIT>(px, py, rec.d);
          if (__ballot(o < 0) & readlane64(alive4, j)) slow |= 1u << j;
        }
      }
      if (PCP_ABLATE & 128) segw += __builtin_amdgcn_s_memtime() + (slow & 0u) - tw;
    } else {
      get_alive();
  #pragma unroll
      for (int j = 0; j < kChunk; ++j) {
        const Rec rec = expand(st.buf[j]);
        const uint32_t w = c * kChunk + j;
        if (w < w1) {
          const uint32_t kind = rec.xk >> 28;
          const uint32_t kind0 = __builtin_amdgcn_readfirstlane(kind);
          if (!GLOBAL && __all(kind == kind0) && kind0 <= PCP_LT && failm == 0 && a.m.sums.count == 0) {
            const uint64_t alive = readlane64(alive4, j);
            if (alive) {  // some record of this word is live in some node
              if (kind0 == PCP_EQ) {
                slow |= 1u << j;
              } else {
                const Cell* px = kdom + slot_row<B, PACKED>(rec.xk & kSlotMask);
                const Cell* py = kdom + slot_row<B, PACKED>(rec.y);
                int o;
                uint64_t tw = 0;
                if (PCP_ABLATE & 128) tw = __builtin_amdgcn_s_memtime() + (__builtin_amdgcn_readfirstlane(rec.d) & 0);
                if constexpr (PACKED) o = (kind0 == PCP_NEQ) ? fast_signs16<PCP_NEQ, B, IMPLICIT>(px, py, rec.d) : fast_signs16<PCP_LT, B, IMPLICIT>(px, py, rec.d);
                else o = (kind0 == PCP_NEQ) ? fast_signs<PCP_NEQ, B, IMPLICIT>(px, py, rec.d) : fast_signs<PCP_LT, B, IMPLICIT>(px, py, rec.d);
                if (PCP_ABLATE & 128) segw += __builtin_amdgcn_s_memtime() + (__builtin_amdgcn_readfirstlane(o) & 0) - tw;
                if (__ballot(o < 0) & alive) slow |= 1u << j;
              }
              if (jq == (uint32_t)j) {  // every live pair of this word is tested on its node's domains (level 1, or level 2 for XEqY)
#pragma unroll
                for (int h = 0; h < NL; ++h) ctr.ev += (uint32_t)__popcll(loaded[h]);
              }
            }
            if (jq == (uint32_t)j && !post_a) {  // every live record of every node runs once
  #pragma unroll
              for (int h = 0; h < NL; ++h) steps_lane += __popcll(loaded[h]);
            }
          } else {
            slow |= 16u << j;  // mixed kinds / ternary / a failed node in the tile / HBM-resident domains
          }
        }
      }
    }
    if (PCP_ABLATE & 16) slow = 0;
    uint64_t tm2 = 0;
    if (PCP_ABLATE & 128) { tm2 = __builtin_amdgcn_s_memtime() + (slow & 0u); seg[1] += tm2 - tm1; }
    // ---- cold part, one rolled copy: flagged words (level 2 + full filter) and words outside the fast path ---------
    while (slow) {
      const uint32_t jb = __builtin_ctz(slow);
      slow &= slow - 1;
      const uint32_t j = jb & 3u;
      const bool generic = jb >= 4;
      const uint32_t w = c * kChunk + j;
      const Rec rec = expand(fetch_rec(w));  // re-read (L1/L2 hit) instead of keeping four more records alive
      const uint32_t kind = rec.xk >> 28;
      uint32_t todo = 0;  // nodes to run with the full filter
      if (!generic) {
        const uint32_t kind0 = __builtin_amdgcn_readfirstlane(kind);
        const Cell* px = kdom + slot_row<B, PACKED>(rec.xk & kSlotMask);
        const Cell* py = kdom + slot_row<B, PACKED>(rec.y);
        // (for NEQ / LT the hot part has already established that a record live somewhere in the tile is flagged)
        if (kind0 == PCP_EQ) todo = fast_nodes<PCP_EQ, B, PACKED, NL, IMPLICIT>(px, py, rec.d, loaded, j);
        else if (kind0 == PCP_NEQ) todo = fast_nodes<PCP_NEQ, B, PACKED, NL, IMPLICIT>(px, py, rec.d, loaded, j);
        else todo = fast_nodes<PCP_LT, B, PACKED, NL, IMPLICIT>(px, py, rec.d, loaded, j);
      } else {
        const bool tern = kind > PCP_LT;
        for (uint32_t b = 0; b < nb; ++b) {
          const uint64_t word = word_of(loaded, b, j);
          if (word == 0 || ((failm >> b) & 1u)) continue;
          todo |= 1u << b;
          const uint64_t t3 = __ballot(((word >> lane) & 1ull) && tern);
          if (!post_a) { steps3 += __popcll(t3); steps2 += __popcll(word) - __popcll(t3); }
        }
      }
      while (todo) {
        const uint32_t b = __builtin_ctz(todo);
        todo &= todo - 1;
        const uint64_t word = word_of(loaded, b, j);
        bool e = false;
        if (generic) ctr.add_ev_uniform((uint32_t)__popcll(word));  // not counted by a level-1 / level-2 test above
        bool run = (word >> lane) & 1ull;
        if constexpr (IMPLICIT) {
          // only the lanes on which a domain can narrow (fast_flag); an entailed or untouched record is left alone
          if (!generic && run) {
            const auto dmr = make_dom<GLOBAL, PACKED>(k, b, chg_next, &ctr);
            const int2 X = dmr.load(rec.xk & kSlotMask), Y = dmr.load(rec.y);
            const int Yl = Y.x + rec.d, Yu = Y.y + rec.d;
            if (kind == PCP_NEQ) run = (X.x == Yu) | (Yl == X.y);
            else if (kind == PCP_LT) run = (X.y >= Yu) | (Yl <= X.x);
            else run = (X.x != Yl) | (X.y != Yu);
          }
        }
        ctr.add_full_uniform((uint32_t)__popcll(__ballot(run)));
        if (run) {
          const auto dm = make_dom<GLOBAL, PACKED>(k, b, chg_next, &ctr);
          e = eval_record(rec, dm);
        }
        const uint64_t nw_word = word & ~__ballot(e);
        const uint32_t l = (b & 15u) * kChunk + j;
        if (NL == 1 || b < 16u) my_new[0] = writelane64(my_new[0], nw_word, l);
        else my_new[NL - 1] = writelane64(my_new[NL - 1], nw_word, l);
      }
    }
    uint64_t tm3 = 0;
    if (PCP_ABLATE & 128) { tm3 = __builtin_amdgcn_s_memtime(); seg[2] += tm3 - tm2; }
    const uint32_t wl = c * kChunk + jq;
#pragma unroll
    for (int h = 0; h < NL; ++h) {
      if (!IMPLICIT && io[h] && wl < w1) {
        rem_acc[h] += post_a ? (uint32_t)(__popcll(loaded[h]) - __popcll(my_new[h])) : (uint32_t)__popcll(my_new[h]);
        if (!(PCP_ABLATE & 8) && (live_src != a.live || my_new[h] != loaded[h])) my_out[h][wl] = my_new[h];
      }
    }
    if (PCP_ABLATE & 128) seg[3] += __builtin_amdgcn_s_memtime() - tm3;
  };
  Stage sa, sb;
  issue(sa, c0 + wave);
  for (uint32_t c = c0 + wave; c < c1; c += 2 * nw) {
    issue(sb, c + nw);
    process(c, sa);
    issue(sa, c + 2 * nw);
    if (c + nw < c1) process(c + nw, sb);
  }
#pragma unroll
  for (int h = 0; h < NL; ++h)
    if (!IMPLICIT && io[h] && rem_acc[h]) { if (post_a) atomicSub(&remaining[bq + 16u * h], rem_acc[h]); else atomicAdd(&remaining[bq + 16u * h], rem_acc[h]); }
  for (int o = 32; o > 0; o >>= 1) steps_lane += __shfl_down(steps_lane, o);
  steps2 += __builtin_amdgcn_readfirstlane(steps_lane);
  if ((PCP_ABLATE & 128) && lane == 0) {  // profiling build: per-segment ticks summed over all wavefronts
    atomicAdd((unsigned long long*)&a.stats->nodes, (unsigned long long)segw);
    atomicAdd((unsigned long long*)&a.stats->steps3, (unsigned long long)seg[0]);
    atomicAdd((unsigned long long*)&a.stats->narrowings, (unsigned long long)seg[1]);
    atomicAdd((unsigned long long*)&a.stats->failed_nodes, (unsigned long long)seg[2]);
    atomicAdd((unsigned long long*)&a.stats->waves, (unsigned long long)seg[3]);
  }
}

// ------------------------------------------------------------------------------------------------
// Word-group sweep (packed tiles, models whose words have descriptors — WordDesc, pcp_internal.h).
// Phase A streams the live masks, ONE WORD PER LANE, 64 consecutive words per wavefront step:
//   * the B nodes' live words are B fully coalesced 512-byte row loads (lane l = word l of that node);
//   * level -1: the lane tests its word as a whole — 64 records x B nodes — on the range-minimum tables of the tile
//     summaries: the slots of the word's x (y) operands span a short range [lo,hi], whose (min -lb, min ub) is two
//     table reads, and the offsets span [dmin,dmax]; the level-0 inequalities with those bounds prove that no record
//     of the word does anything in any node.  Then nothing of the word is touched: not its records, not the domains;
//   * words that fail are only noted in a bitmap (LDS: the changed-pair list's space, idle until the rounds).
// Phase B hands the noted words (a few percent: the variables branched on near this tile, words that straddle two
// x-blocks) to the record-level tests, lane = record: level 0, level 1, level 2, full filter.  Word w goes to wavefront
// w mod 16; the records and the node column of four words are fetched together.
// ------------------------------------------------------------------------------------------------
// Returns false when phase A noted more than an eighth of the words (a tile deep in the search tree: many assigned
// variables, whose words no range test clears): the caller then runs the chunked record-level sweep over the noted
// words instead of phase B, which is organised for a few words.
template <int B, bool COMPACT, bool IMPLICIT>
__device__ __forceinline__ bool sweep_words(const LaunchArgs& a, const BlockCtx& k, uint32_t node0, uint32_t nb, uint32_t* chg_next,
                                            uint32_t* remaining, uint32_t* hardmap, uint64_t& steps2, uint64_t& steps3, Ctr& ctr,
                                            uint32_t* gscratch, uint32_t gscratch_words) {
  static_assert(B >= 4 && B <= 16 && B % 4 == 0, "one live register per node and lane; a node column fits one DPP row");
  constexpr bool PACKED = true, GLOBAL = false;
  using Cell = uint32_t;
  const Cell* const kdom = static_cast<const Cell*>(k.dom);
  const uint32_t* const tmin = static_cast<const uint32_t*>(k.summ.a);
  const uint32_t* const tmax = static_cast<const uint32_t*>(k.summ.b);
  const uint32_t lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t P = a.m.n_recs, words = (P + 63) >> 6, groups = (words + 63) >> 6;
  const uint64_t tail_mask = (P & 63) ? ((1ull << (P & 63)) - 1) : ~0ull;
  const uint64_t* live_src = a.live_in;
  const uint32_t S = k.rmq_stride;
  using RecT = typename std::conditional<COMPACT, Rec8, Rec>::type;
  const RecT* rec_stream;
  if constexpr (COMPACT) rec_stream = a.m.recs8; else rec_stream = a.m.recs;
  uint64_t* const hard64 = reinterpret_cast<uint64_t*>(hardmap);  // [groups]
  uint32_t steps_lane = 0, n_hard = 0;
  uint32_t racc[B / 2];  // remaining live records per node, two 16-bit lane counters per register
#pragma unroll
  for (int i = 0; i < B / 2; ++i) racc[i] = 0;
  const uint32_t failm = __hip_atomic_load(&k.misc[M_FAIL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // empty input domains
  uint64_t tseg[2] = {0, 0};  // PCP_ABLATE & 128: s_memtime ticks of phase A / phase B
  uint32_t n_l0 = 0, n_l1 = 0, n_l2 = 0, n_bulk = 0;  // words that reached level 0 / 1 / 2; (word,node) pairs unlinked in bulk
  (void)n_bulk;
  const uint64_t tA = (PCP_ABLATE & 128) ? __builtin_amdgcn_s_memtime() : 0;
  // ================= group level (implicit nodes) =================
  // One lane per GROUP of 64 words: the x operands through the range tables, the y operands as a SUFFIX [ylo, n_slots) —
  // the minimum (maximum) of the rest of ylo's 64-slot chunk from the tables and of all later chunks from a small
  // suffix array — with the group's offset range.  A group that passes needs no word descriptors and no word tests at all
  // (near the root: every x-block whose operands were not branched on).  Same soundness argument as level -1.
  uint32_t* gpass = gscratch;  // [ceil(groups/32)] pass bits, then the chunk suffix arrays
  bool have_g = false;
  if constexpr (IMPLICIT) {
    const uint32_t slots = k.S, chunks = (slots + 63) >> 6, gw = (groups + 31) >> 5;
    have_g = a.m.gdesc != nullptr && a.word_level >= 2 && gw + 2 * (chunks + 1) <= gscratch_words && failm == 0;
    if (have_g) {
      uint32_t* sufmin = gscratch + gw;             // [chunks + 1]: min over the slots of chunks >= c
      uint32_t* sufmax = sufmin + chunks + 1;
      const uint32_t tid = threadIdx.x;
      if (tid <= chunks) {
        uint32_t mn = 0x7fff7fffu, mx = 0x80008000u;
        for (uint32_t c = tid; c < chunks; ++c) {  // level 6 = the chunk starting at slot 64 c
          mn = pk_min(mn, tmin[6 * S + tsw(c << 6)]);
          mx = pk_max(mx, tmax[6 * S + tsw(c << 6)]);
        }
        sufmin[tid] = mn; sufmax[tid] = mx;
      }
      __syncthreads();
      for (uint32_t g0 = 0; g0 < groups; g0 += blockDim.x) {
        const uint32_t gi = g0 + tid;
        bool pass = false, open0 = false;
        if (gi < groups) {
          const GroupDesc q = a.m.gdesc[gi];
          const uint32_t cls = (q.k >> 8) & 15u, kx = q.k & 15u;
          if (cls) {
            const uint32_t xa = __umul24(kx, S) + tsw(q.x & 0xffffu), xb = __umul24(kx, S) + tsw(q.x >> 16);
            const uint32_t yend = min(q.ylo | 63u, slots - 1), len = yend - q.ylo + 1, ky = 31u - (uint32_t)__builtin_clz(len);
            const uint32_t ya = __umul24(ky, S) + tsw(q.ylo), yb = __umul24(ky, S) + tsw(yend + 1 - (1u << ky));
            const uint32_t cn = (q.ylo >> 6) + 1;
            const uint32_t Xn = pk_min(tmin[xa], tmin[xb]), Xx = pk_max(tmax[xa], tmax[xb]);
            const uint32_t Yn = pk_min(pk_min(tmin[ya], tmin[yb]), sufmin[cn]), Yx = pk_max(pk_max(tmax[ya], tmax[yb]), sufmax[cn]);
            const int dmin = lo16(q.d), dmax = hi16(q.d);
            if (cls == 1) {
              uint32_t tlo, thi;
              asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(tlo) : "v"(Xn), "v"(Yn));
              asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(thi) : "v"(Xx), "v"(Yx));
              pass = neq_no_zero(tlo, thi, dmin, dmax) == 0;
              open0 = (lo16(tlo) + dmin > 0) && (hi16(tlo) - dmax > 0);
            } else {
              pass = ((hi16(Yn) - hi16(Xx) + dmin - 1) | (lo16(Xn) - lo16(Yx) + dmin - 1)) >= 0;
            }
          }
        }
        const uint64_t bal = __ballot(pass);
        const uint32_t wi = (g0 >> 5) + 2 * wave;
        if (lane == 0) { if (wi < gw) gpass[wi] = (uint32_t)bal; if (wi + 1 < gw) gpass[wi + 1] = (uint32_t)(bal >> 32); }
        // a group of XNeqY words whose intervals overlap in more than a point in every node: none of its records is entailed
        // at staging time — if the tile then narrows nothing, every node is known to be Unknown without the final scan
        if (__ballot(open0) != 0 && lane == 0) atomicOr(&k.misc[M_OPEN0], 1u);
      }
      __syncthreads();
    }
  }
  // ================= phase A =================
  // The word's descriptor is fetched one group ahead, so that the level -1 arithmetic runs while the group's row loads
  // are in flight instead of behind them.
  WordPart qa_next = a.m.wdesc[min(wave * 64 + lane, words - 1)].a;
  for (uint32_t g = wave; g < groups; g += nw) {
    const uint32_t w = g * 64 + lane;
    const bool wv = w < words;
    const uint32_t wc = min(w, words - 1);
    if constexpr (IMPLICIT) {
      if (have_g && ((gpass[g >> 5] >> (g & 31u)) & 1u)) {  // the whole group was cleared one level up
        const uint64_t v = wv ? (wc == words - 1 ? tail_mask : ~0ull) : 0ull;
        steps_lane += nb * (uint32_t)__popcll(v);
        if (lane == 0) hard64[g] = 0;
        const uint32_t gn = g + nw;
        if (gn < groups && !((gpass[gn >> 5] >> (gn & 31u)) & 1u)) qa_next = a.m.wdesc[min(gn * 64 + lane, words - 1)].a;
        continue;
      }
    }
    const WordPart qa = qa_next;
    uint64_t lv[IMPLICIT ? 1 : B];
    if constexpr (IMPLICIT) {
      lv[0] = wv ? (wc == words - 1 ? tail_mask : ~0ull) : 0ull;  // every node: all records live
    } else {
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const uint32_t node = node0 + ((uint32_t)b < nb ? (uint32_t)b : 0u);
        uint64_t v = live_src ? live_src[(size_t)node * words + wc] : ~0ull;
        if (wc == words - 1) v &= tail_mask;
        lv[b] = (wv && (uint32_t)b < nb) ? v : 0ull;
      }
    }
    qa_next = a.m.wdesc[min((g + nw) * 64 + lane, words - 1)].a;
    // level -1: the whole word against the range tables
    auto part_fails = [&](const WordPart q) -> bool {
      const uint32_t cls = (q.k >> 8) & 15u;
      const uint32_t kx = q.k & 15u, ky = (q.k >> 4) & 15u;
      const uint32_t xa = __umul24(kx, S) + tsw(q.x & 0xffffu), xb = __umul24(kx, S) + tsw(q.x >> 16);
      const uint32_t ya = __umul24(ky, S) + tsw(q.y & 0xffffu), yb = __umul24(ky, S) + tsw(q.y >> 16);
      const uint32_t Xn = pk_min(tmin[xa], tmin[xb]), Yn = pk_min(tmin[ya], tmin[yb]);  // (min -lb, min ub) over the range
      const int dmin = lo16(q.d), dmax = hi16(q.d);
      if constexpr (IMPLICIT) {
        // no record of the word can narrow in any node (neq_no_zero / the XLessY narrowing terms); with the maximum tables
        // (word_level 2) this also clears words that are entailed throughout the tile
        if (cls == 1 && a.word_level >= 2) {
          const uint32_t Xx = pk_max(tmax[xa], tmax[xb]), Yx = pk_max(tmax[ya], tmax[yb]);
          uint32_t tlo, thi;
          asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(tlo) : "v"(Xn), "v"(Yn));
          asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(thi) : "v"(Xx), "v"(Yx));
          return neq_no_zero(tlo, thi, dmin, dmax) < 0;
        }
        if (cls == 2 && a.word_level >= 2) {
          const uint32_t Xx = pk_max(tmax[xa], tmax[xb]), Yx = pk_max(tmax[ya], tmax[yb]);
          return ((hi16(Yn) - hi16(Xx) + dmin - 1) | (lo16(Xn) - lo16(Yx) + dmin - 1)) < 0;
        }
      }
      if (cls == 1) {
        uint32_t t;
        asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(Xn), "v"(Yn));  // (Xn + Yu, Xu + Yn)
        return ((lo16(t) + dmin - 1) | (hi16(t) - dmax - 1)) < 0;
      }
      if (cls == 2 && a.word_level >= 2) {
        const uint32_t Xx = pk_max(tmax[xa], tmax[xb]), Yx = pk_max(tmax[ya], tmax[yb]);
        return ((hi16(Yn) - hi16(Xx) + dmin - 1) | (lo16(Xn) - lo16(Yx) + dmin - 1) | (hi16(Xn) + lo16(Yn) - dmax)) < 0;
      }
      return true;  // no descriptor: always to the record level
    };
    bool fail = part_fails(qa);
    if (!fail && ((qa.k >> 12) & 1u)) fail = part_fails(a.m.wdesc[wc].b);  // a word that straddles two x-blocks
    uint64_t alive = 0;
    if constexpr (IMPLICIT) {
      alive = lv[0];
      steps_lane += nb * (uint32_t)__popcll(lv[0]);  // every record of every node runs once
    } else {
#pragma unroll
      for (int b = 0; b < B; ++b) {
        alive |= lv[b];
        const uint32_t pc = (uint32_t)__popcll(lv[b]);
        steps_lane += pc;                    // every live record of every node runs once
        racc[b / 2] += pc << (16 * (b & 1));
        if (wv && (uint32_t)b < nb && live_src != a.live) a.live[(size_t)(node0 + b) * words + w] = lv[b];
      }
    }
    uint64_t hard = __ballot(alive != 0 && (fail || failm != 0));
    if (PCP_ABLATE & 16) hard = 0;
    if (PCP_ABLATE & 128) n_l0 += __popcll(hard);
    if (PCP_ABLATE & 512) steps3 += __popcll(hard);  // profiling: words noted for the record level (tools/hard_words.sh)
    n_hard += __popcll(hard);
    if (lane == 0) hard64[g] = hard;
  }
  if (lane == 0 && n_hard) atomicAdd(&k.misc[M_HARD], n_hard);
  if constexpr (!IMPLICIT) {
#pragma unroll
    for (int b = 0; b < B; ++b) {
      uint32_t r = (racc[b / 2] >> (16 * (b & 1))) & 0xffffu;
      for (int o = 32; o > 0; o >>= 1) r += __shfl_down(r, o);
      if (lane == 0 && (uint32_t)b < nb && r) atomicAdd(&remaining[b], r);
    }
  }
  for (int o = 32; o > 0; o >>= 1) steps_lane += __shfl_down(steps_lane, o);
  steps2 += __builtin_amdgcn_readfirstlane(steps_lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's live stores are out before anyone re-reads the rows
  __syncthreads();
  if (k.misc[M_HARD] * 8u > words) return false;
  const uint64_t tB = (PCP_ABLATE & 128) ? __builtin_amdgcn_s_memtime() : 0;
  // ================= phase B =================
  uint64_t pattern = 0;  // the words of a group this wavefront takes: bit positions congruent to `wave` modulo the wavefront count
  for (uint32_t i = wave; i < 64; i += nw) pattern |= 1ull << i;
  constexpr int kBatch = 4;
  uint32_t pendv = 0;  // the pending words, word t in lane t (a dynamically indexed array would live in scratch memory)
  uint32_t npend = 0;
  uint64_t* const my_live = IMPLICIT ? nullptr : a.live + (size_t)(node0 + (lane < nb ? lane : 0u)) * words;  // lane b = node b: the word's column
  uint64_t tfl[2] = {0, 0};  // PCP_ABLATE & 128: ticks waiting for a batch's loads / processing it
  auto flush = [&]() {
    const uint64_t tf0 = (PCP_ABLATE & 128) ? __builtin_amdgcn_s_memtime() : 0;
    RecT rb[kBatch];
    uint64_t cb[kBatch];
#pragma unroll
    for (int t = 0; t < kBatch; ++t) {
      const uint32_t ww = (uint32_t)__builtin_amdgcn_readlane((int)pendv, (uint32_t)t < npend ? t : 0);
      rb[t] = (rec_stream + (size_t)ww * 64)[lane];
      uint64_t v = ~0ull;
      if constexpr (!IMPLICIT) v = __hip_atomic_load(my_live + ww, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // past the L1: phase A stored it
      cb[t] = lane < nb ? (ww == words - 1 ? v & tail_mask : v) : 0ull;
    }
    uint64_t tf1 = 0;
    if (PCP_ABLATE & 128) {
      uint32_t dep = 0;
#pragma unroll
      for (int t = 0; t < kBatch; ++t) dep |= (uint32_t)cb[t] | (uint32_t)rb[t].d;
      tf1 = __builtin_amdgcn_s_memtime() + (__builtin_amdgcn_readfirstlane(dep) & 0u);
      tfl[0] += tf1 - tf0;
    }
    for (uint32_t t = 0; t < npend; ++t) {
      const uint32_t ww = (uint32_t)__builtin_amdgcn_readlane((int)pendv, (int)t);
      RecT rsel = rb[0];
      uint64_t col = cb[0];
#pragma unroll
      for (int u = 1; u < kBatch; ++u)
        if (t == (uint32_t)u) { rsel = rb[u]; col = cb[u]; }
      if (PCP_ABLATE & 4096) continue;  // profiling: nothing per word
      // alive_w: OR of the column over the nodes (lanes 0..B-1 sit in one DPP row)
      const uint64_t alive_w = IMPLICIT ? ((ww == words - 1) ? tail_mask : ~0ull)
                                        : (((uint64_t)__builtin_amdgcn_readfirstlane(row_or16((uint32_t)(col >> 32))) << 32) |
                                           __builtin_amdgcn_readfirstlane(row_or16((uint32_t)col)));
      const Rec rec = expand(rsel);
      const uint32_t kind = rec.xk >> 28;
      const uint32_t kind0 = __builtin_amdgcn_readfirstlane(kind);
      const Cell* px = kdom + slot_row<B, PACKED>(rec.xk & kSlotMask);
      const Cell* py = kdom + slot_row<B, PACKED>(rec.y);
      const uint32_t failnow = __hip_atomic_load(&k.misc[M_FAIL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if ((PCP_ABLATE & 8192) && (alive_w | failnow | kind0 | (uint64_t)(size_t)px | (uint64_t)(size_t)py) != 0x123456789ull) continue;  // profiling: preamble only
      uint32_t todo = 0;
      uint64_t ncol = col;
      const bool tested = __all(kind == kind0) && kind0 <= PCP_LT && failnow == 0;
      if (tested) {
        bool run2 = true;
        if (kind0 != PCP_EQ) {
          // level 0 on the per-slot summaries, then level 1 on every node
          int o0;
          if constexpr (IMPLICIT) {
            const uint32_t xs = tmin[tsw(rec.xk & kSlotMask)], ys = tmin[tsw(rec.y)], xx = tmax[tsw(rec.xk & kSlotMask)], yx = tmax[tsw(rec.y)];
            if (kind0 == PCP_NEQ) {
              uint32_t tlo, thi;
              asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(tlo) : "v"(xs), "v"(ys));
              asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(thi) : "v"(xx), "v"(yx));
              o0 = neq_no_zero(tlo, thi, rec.d, rec.d);
            } else {
              o0 = (hi16(ys) - hi16(xx) + rec.d - 1) | (lo16(xs) - lo16(yx) + rec.d - 1);
            }
          } else {
            const uint32_t xs = tmin[tsw(rec.xk & kSlotMask)], ys = tmin[tsw(rec.y)];
            if (kind0 == PCP_NEQ) {
              uint32_t t2;
              asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(t2) : "v"(xs), "v"(ys));
              o0 = (lo16(t2) + (rec.d - 1)) | (hi16(t2) + (-rec.d - 1));
            } else {
              const uint32_t xx = tmax[tsw(rec.xk & kSlotMask)], yx = tmax[tsw(rec.y)];
              o0 = (hi16(ys) - hi16(xx) + rec.d - 1) | (lo16(xs) - lo16(yx) + rec.d - 1) | (hi16(xs) + lo16(ys) - rec.d);
            }
          }
          run2 = (__ballot(o0 < 0) & alive_w) != 0;
          if (PCP_ABLATE & 2048) run2 = false;  // profiling: stop after level 0
          // Explicit rows: no level 1 here — of the words that level -1 AND level 0 leave over, 97 % go on to level 2 anyway
          // (their entailed lanes have to be unlinked), which decides node by node at about the same cost.
          // Implicit nodes: level 1 over all nodes of the tile first.  Only a lane that can narrow matters, and the words
          // that reach this point are mostly those of assigned variables whose records are entailed or untouched: the
          // zero-detection test clears them for the whole tile in one pass.
          if constexpr (IMPLICIT) {
            if (run2) {
              const int o1 = (kind0 == PCP_NEQ) ? fast_signs16<PCP_NEQ, B, true>(px, py, rec.d) : fast_signs16<PCP_LT, B, true>(px, py, rec.d);
              ctr.ev += (uint32_t)__popcll(col);  // lane b < nb: the pairs of node b (the column register)
              run2 = (__ballot(o1 < 0) & alive_w) != 0;
            }
          }
          if ((PCP_ABLATE & 128) && run2) { ++n_l1; ++n_l2; }
        }
        if (run2) {
          // level 2, node by node: lanes that are merely entailed lose their live bit here, in bulk; only nodes with a
          // lane whose domains would change go on to the full filter
          if (kind0 == PCP_NEQ) {
            // packed form of fast_flag / pure_entailed for XNeqY: T = (Xn + Yu, Xu + Yn); something happens iff
            // T.lo + d - 1 < 0 or T.hi - d - 1 < 0, and it is a pure entailment iff T.lo + d < 0 or T.hi - d < 0
            // (saturating adds: |T| <= 32766, |d| may be larger)
            const int dc = max(-32767, min(32767, rec.d));
            const uint32_t c1 = ((uint32_t)(dc - 1) & 0xffffu) | ((uint32_t)(-dc - 1) << 16);
            const uint32_t c0 = ((uint32_t)dc & 0xffffu) | ((uint32_t)(-dc) << 16);
            const uint32_t cz = c0;  // IMPLICIT: (d, -d)
            (void)c1; (void)cz;
#pragma unroll
            for (int g4 = 0; g4 < B; g4 += 4) {
              const uint4 Xq = *static_cast<const uint4*>(__builtin_assume_aligned(px + g4, 16));
              const uint4 Yq = *static_cast<const uint4*>(__builtin_assume_aligned(py + g4, 16));
              const uint32_t xs[4] = {Xq.x, Xq.y, Xq.z, Xq.w}, ys[4] = {Yq.x, Yq.y, Yq.z, Yq.w};
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) {
                const uint64_t wd = readlane64(col, (uint32_t)(g4 + jj));
                if (wd == 0) continue;
                if (!IMPLICIT) ctr.add_ev_uniform((uint32_t)__popcll(wd));
                if constexpr (IMPLICIT) {
                  // (t1, t2) = X + swap(Y) + (d, -d): a lane can narrow only where one of them is zero
                  uint32_t Uz;
                  asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                      "v_pk_add_u16 %0, %0, %3"
                      : "=&v"(Uz) : "v"(xs[jj]), "v"(ys[jj]), "v"(cz));
                  const uint64_t f = __ballot((Uz & 0xffffu) == 0u || (Uz >> 16) == 0u) & wd;
                  todo |= f ? (1u << (g4 + jj)) : 0u;
                  continue;
                }
                uint32_t F, E;
                asm("v_pk_add_u16 %0, %2, %3 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                    "v_pk_add_i16 %1, %0, %5 clamp\n\t"
                    "v_pk_add_i16 %0, %0, %4 clamp"
                    : "=&v"(F), "=&v"(E) : "v"(xs[jj]), "v"(ys[jj]), "v"(c1), "v"(c0));
                const uint64_t f = __ballot((F & 0x80008000u) != 0) & wd;
                const uint64_t en = __ballot((E & 0x80008000u) != 0) & wd;
                if (en) { ncol = writelane64(ncol, wd & ~en, (uint32_t)(g4 + jj)); if (PCP_ABLATE & 128) ++n_bulk; }
                todo |= (f & ~en) ? (1u << (g4 + jj)) : 0u;
              }
            }
          } else {
#pragma unroll
            for (int g4 = 0; g4 < B; g4 += 4) {
              const uint4 Xq = *static_cast<const uint4*>(__builtin_assume_aligned(px + g4, 16));
              const uint4 Yq = *static_cast<const uint4*>(__builtin_assume_aligned(py + g4, 16));
              const uint32_t xs[4] = {Xq.x, Xq.y, Xq.z, Xq.w}, ys[4] = {Yq.x, Yq.y, Yq.z, Yq.w};
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) {
                const uint64_t wd = readlane64(col, (uint32_t)(g4 + jj));
                if (wd == 0) continue;
                if (!IMPLICIT || kind0 == PCP_EQ) ctr.add_ev_uniform((uint32_t)__popcll(wd));
                const int2 X = unpack16(xs[jj]), Y = unpack16(ys[jj]);
                uint64_t f, en = 0;
                if (kind0 == PCP_LT) { f = fast_flag<PCP_LT, IMPLICIT>(X, Y.x + rec.d, Y.y + rec.d); if (!IMPLICIT) en = pure_entailed<PCP_LT>(X, Y.x + rec.d, Y.y + rec.d); }
                else { f = fast_flag<PCP_EQ, IMPLICIT>(X, Y.x + rec.d, Y.y + rec.d); if (!IMPLICIT) en = pure_entailed<PCP_EQ>(X, Y.x + rec.d, Y.y + rec.d); }
                en &= wd;
                if (en) { ncol = writelane64(ncol, wd & ~en, (uint32_t)(g4 + jj)); if (PCP_ABLATE & 128) ++n_bulk; }
                todo |= (f & wd & ~en) ? (1u << (g4 + jj)) : 0u;
              }
            }
          }
          todo = __builtin_amdgcn_readfirstlane(todo);
        }
      } else {
        // mixed kinds or a failed node in the tile: every node with a live record, straight to the full filter
        for (uint32_t b = 0; b < nb; ++b)
          if (readlane64(col, b) != 0 && !((failnow >> b) & 1u)) { todo |= 1u << b; ctr.add_ev_uniform((uint32_t)__popcll(readlane64(col, b))); }
      }
      while (todo) {
        const uint32_t b = __builtin_ctz(todo);
        todo &= todo - 1;
        const uint64_t word = readlane64(ncol, b);  // without the lanes already unlinked in bulk
        bool e = false;
        bool run = (word >> lane) & 1ull;
        if constexpr (IMPLICIT) {
          if (tested && run) {  // only the lanes on which a domain can narrow (fast_flag)
            const auto dmr = make_dom<GLOBAL, PACKED>(k, b, chg_next, &ctr);
            const int2 X = dmr.load(rec.xk & kSlotMask), Y = dmr.load(rec.y);
            const int Yl = Y.x + rec.d, Yu = Y.y + rec.d;
            if (kind0 == PCP_NEQ) run = (X.x == Yu) | (Yl == X.y);
            else if (kind0 == PCP_LT) run = (X.y >= Yu) | (Yl <= X.x);
            else run = (X.x != Yl) | (X.y != Yu);
          }
        }
        ctr.add_full_uniform((uint32_t)__popcll(__ballot(run)));
        if (run) {
          const auto dm = make_dom<GLOBAL, PACKED>(k, b, chg_next, &ctr);
          e = eval_record(rec, dm);
        }
        ncol = writelane64(ncol, word & ~__ballot(e), b);
      }
      if (!IMPLICIT && lane < nb && ncol != col) {  // entailed records: unlink them (store.rs:200-207)
        my_live[ww] = ncol;
        atomicSub(&remaining[lane], (uint32_t)(__popcll(col) - __popcll(ncol)));
      }
    }
    npend = 0;
    if (PCP_ABLATE & 128) tfl[1] += __builtin_amdgcn_s_memtime() - tf1;
  };
  for (uint32_t base = 0; base < groups; base += 64) {  // 64 groups per step: lane l looks at group base + l
    const uint64_t mine = (base + lane < groups) ? (hard64[base + lane] & pattern) : 0ull;
    uint64_t have = __ballot(mine != 0);
    while (have) {
      const uint32_t l = __builtin_ctzll(have);
      have &= have - 1;
      uint64_t bits = readlane64(mine, l);
      const uint32_t g = base + l;
      while (bits) {
        const uint32_t j = __builtin_ctzll(bits);
        bits &= bits - 1;
        pendv = (lane == npend) ? g * 64 + j : pendv;  // (a v_cndmask on the lane id)
        if (++npend == kBatch) flush();
      }
    }
  }
  if (npend) flush();
  (void)steps3;
  if ((PCP_ABLATE & 128) && lane == 0) {  // profiling build (tools/seg_words.py)
    const uint64_t tE = __builtin_amdgcn_s_memtime();
    tseg[0] = tB - tA; tseg[1] = tE - tB;
    atomicAdd((unsigned long long*)&a.stats->steps3, (unsigned long long)tseg[0]);
    atomicAdd((unsigned long long*)&a.stats->narrowings, (unsigned long long)tseg[1]);
    atomicAdd((unsigned long long*)&a.stats->failed_nodes, (unsigned long long)n_l0 | ((unsigned long long)n_l1 << 24) | ((unsigned long long)n_l2 << 44));
    atomicAdd((unsigned long long*)&a.stats->waves, (unsigned long long)tfl[0]);
    atomicAdd((unsigned long long*)&a.stats->nodes, (unsigned long long)tfl[1]);
  }
  return true;
}

// Dense wake-up round: more changed variables than the LDS list holds, so stream the whole table again and run
// the live records that touch a variable in `cur`.  Rare path, generic code.
template <bool GLOBAL, bool PACKED, bool IMPLICIT>
__device__ __forceinline__ void sweep_filtered(const LaunchArgs& a, const BlockCtx& k, uint32_t node0, uint32_t nb, const uint32_t* cur,
                                               uint32_t* chg_next, uint32_t& rem_sub, uint64_t& steps2, uint64_t& steps3, Ctr& ctr) {
  const uint32_t lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps the loop control scalar
  const uint32_t P = a.m.n_recs, words = (P + 63) >> 6;
  for (uint32_t w = wave; w < words; w += nw) {
    const uint32_t r = (w << 6) + lane;
    Rec rec;
    if (r < P) rec = a.m.recs[r];
    else { rec.xk = 0; rec.y = 0; rec.z = 0; rec.d = 0; }
    uint64_t my_word = 0;
    if (lane < nb) {
      if constexpr (IMPLICIT) my_word = (w == words - 1 && (P & 63)) ? ((1ull << (P & 63)) - 1) : ~0ull;
      else my_word = a.live[(size_t)(node0 + lane) * words + w];
    }
    uint64_t my_new = my_word;
    const uint32_t failm = __hip_atomic_load(&k.misc[M_FAIL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t x = rec.xk & kSlotMask;
    const bool tern = (rec.xk >> 28) > PCP_LT;
    for (uint32_t b = 0; b < nb; ++b) {
      const uint64_t word = readlane64(my_word, b);
      if (word == 0 || ((failm >> b) & 1u)) continue;
      const uint32_t* cb = cur + (size_t)b * k.Wv;
      bool touched = ((cb[x >> 5] >> (x & 31)) & 1u) | ((cb[rec.y >> 5] >> (rec.y & 31)) & 1u);
      if (tern) touched |= (cb[rec.z >> 5] >> (rec.z & 31)) & 1u;
      if (a.m.sums.count)  // a Sum operand changes with any of its members: always re-run (conservative)
        touched |= (x - a.m.sums.first < a.m.sums.count) | (rec.y - a.m.sums.first < a.m.sums.count) | (tern && rec.z - a.m.sums.first < a.m.sums.count);
      const bool mine = ((word >> lane) & 1ull) && touched;
      bool e = false;
      if (mine) {
        const auto dm = make_dom<GLOBAL, PACKED>(k, b, chg_next, &ctr);
        e = eval_record(rec, dm);
      }
      const uint64_t run = __ballot(mine), t3 = __ballot(mine && tern);
      steps3 += __popcll(t3);
      steps2 += __popcll(run) - __popcll(t3);
      ctr.add_ev_uniform((uint32_t)__popcll(run));
      ctr.add_full_uniform((uint32_t)__popcll(run));
      my_new = writelane64(my_new, word & ~__ballot(e), b);
    }
    if (!IMPLICIT && lane < nb && my_new != my_word) {
      rem_sub += __popcll(my_word) - __popcll(my_new);  // newly entailed
      a.live[(size_t)(node0 + lane) * words + w] = my_new;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The fixpoint kernel.  grid = ceil(n_nodes / B) (team == 1)  or  n_nodes * team (B == 1).
// ------------------------------------------------------------------------------------------------
template <int B, bool GLOBAL, bool COMPACT, bool PACKED, bool IMPLICIT>
__global__ void __launch_bounds__(1024) fixpoint_kernel(const LaunchArgs a_in) {
  LaunchArgs a = a_in;
  if (!PCP_ABLATE) a.stats += blockIdx.x & (kStatSlots - 1);  // striped counters (pcp_internal.h); profiling builds keep maxima in slot 0
  if (a.sp_ptr) {
    // device-side DFS: this launch runs the node on top of the stack (one logical node: n_nodes == 1, team geometry)
    const uint32_t sp = *a.sp_ptr;
    if (sp == 0 || *a.stop_ptr) return;
    const size_t off = (size_t)(sp - 1) * a.m.n_vars;
    a.lb_in += off; a.ub_in += off; a.lb_out += off; a.ub_out += off; a.status += sp - 1;
  }
  static_assert(!GLOBAL || B == 1, "the global-domain variant runs one node per block");
  static_assert(!PACKED || (!GLOBAL && B >= 8 && B % 4 == 0), "packed tiles: LDS-resident, a multiple of four nodes");
  static_assert(B <= 32, "fail / todo masks are 32 bits wide");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned long long t_entry = (PCP_ABLATE & 64) ? wall_clock64() : 0ull;
  const uint32_t tid = threadIdx.x, nth = blockDim.x, lane = tid & 63;
  const uint32_t V = a.m.n_vars, S = a.m.n_slots, Wv = (S + 31) >> 5, P = a.m.n_recs, words = (P + 63) >> 6;
  const uint32_t team = a.team, C = a.list_cap;
  constexpr uint32_t BP = PACKED ? (uint32_t)B + 4u : ((B == 1) ? 1u : (uint32_t)B + 2u);
  using Cell = typename CellOf<PACKED>::type;
  // global variant: only the constants' singleton domains are kept in LDS (slots n_vars..S-1)
  const Carve cv = carve(GLOBAL ? S - V : S, B, C, S, PACKED, PACKED ? a.word_level : 0);
  Cell* dom = reinterpret_cast<Cell*>(smem + cv.dom);
  uint32_t* cur = reinterpret_cast<uint32_t*>(smem + cv.chg_a);
  uint32_t* nxt = reinterpret_cast<uint32_t*>(smem + cv.chg_b);
  uint32_t* list_id = reinterpret_cast<uint32_t*>(smem + cv.list_id);
  uint32_t* list_pre = reinterpret_cast<uint32_t*>(smem + cv.list_pre);
  uint32_t* list_off = reinterpret_cast<uint32_t*>(smem + cv.list_off);
  uint32_t* tmp = reinterpret_cast<uint32_t*>(smem + cv.tmp);
  uint32_t* remaining = reinterpret_cast<uint32_t*>(smem + cv.remaining);
  uint32_t* misc = reinterpret_cast<uint32_t*>(smem + cv.misc);

  uint32_t node0, nb, g;
  if (team > 1) { node0 = blockIdx.x / team; g = blockIdx.x % team; nb = 1; }
  else { node0 = blockIdx.x * B; g = 0; nb = min((uint32_t)B, a.n_nodes - node0); }
  if (a.only_marked) {
    // second launch of a packed call (pcp_api.hip): only the tiles the packed kernel handed back.  A packed tile is
    // two of these tiles, so all nodes of this tile carry the same mark.
    if (__hip_atomic_load(a.retry_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) return;
    if (a.status[node0] != kStatusRetry) return;
  }
  const BlockCtx k{dom, BP, SummPtr{smem + cv.summ, smem + cv.summ + (size_t)4 * tsw_slots(S) * rmq_levels(PACKED ? a.word_level : 0, false)}, tsw_slots(S), S, Wv, misc, a.lb_out + (size_t)node0 * V, a.ub_out + (size_t)node0 * V, V, a.m.sums,
                   (GLOBAL && a.dom10) ? reinterpret_cast<unsigned long long*>(smem + cv.total + (a.adj_cache ? ((((size_t)V + 1) * 4 + 15) & ~(size_t)15) : 0)) : nullptr,
                   a.dom10_lo};

  // ---- phase 0: stage the nodes' domains in LDS (coalesced SoA reads), zero the masks ------------------
  // adjacency offsets of the variables: an LDS copy behind the carve when the launch has room for it (a round's
  // compaction then has no global load in its dependence chain)
  const uint32_t* adjo = a.m.adj_off;
  if (a.adj_cache) {
    uint32_t* adj_lds = reinterpret_cast<uint32_t*>(smem + cv.total);
    for (uint32_t v = tid; v <= V; v += nth) adj_lds[v] = a.m.adj_off[v];
    adjo = adj_lds;
  }
  if (tid < (uint32_t)M_WORDS) misc[tid] = 0;
  if (tid < (uint32_t)B) remaining[tid] = 0;
  for (uint32_t i = tid; i < (uint32_t)B * Wv; i += nth) { cur[i] = 0; nxt[i] = 0; }
  __syncthreads();
  if constexpr (GLOBAL) {
    if (a.dom10) {
      // 10-bit LDS cells: three variables per u64, one thread per word (no atomics while staging)
      bool bad = false, oob = false;
      const uint32_t nw3 = (V + 2) / 3;
      for (uint32_t w = tid; w < nw3; w += nth) {
        unsigned long long word = 0;
        for (uint32_t j = 0; j < 3; ++j) {
          const uint32_t v = 3 * w + j;
          int l = 0, u = 0;
          if (v < V) {
            const int lbv = a.lb_in[(size_t)node0 * V + v], ubv = a.ub_in[(size_t)node0 * V + v];
            bad |= lbv > ubv;
            oob |= (lbv < a.dom10_lo) | (ubv > a.dom10_lo + 1023) | (lbv > a.dom10_lo + 1023) | (ubv < a.dom10_lo);
            l = min(max(lbv - a.dom10_lo, 0), 1023); u = min(max(ubv - a.dom10_lo, 0), 1023);
          }
          word |= (unsigned long long)((uint32_t)l | ((uint32_t)u << 10)) << (20 * j);
        }
        k.c10[w] = word;
      }
      for (uint32_t v = V + tid; v < S; v += nth) { int2 d; d.x = d.y = a.m.const_val[v - V]; dom[v - V] = d; }
      if (bad) atomicOr(&misc[M_FAIL], 1u);
      if (oob) atomicOr(&misc[M_OOB], 1u);
      __syncthreads();
      if (misc[M_OOB]) {  // a bound outside the declared hull: the caller's contract violation (as for packed tiles)
        if (tid == 0) { a.status[node0] = kStatusRetry; atomicMax(a.retry_flag, 1u); }
        return;
      }
    } else {
    // the node's rows in lb_out/ub_out ARE the working domains (the host copied the inputs there); only check them
    bool bad = false, wide = false;
    if (g == 0)
      for (uint32_t v = tid; v < V; v += nth) {
        const int l = a.lb_in[(size_t)node0 * V + v], u = a.ub_in[(size_t)node0 * V + v];
        bad |= l > u;
        wide |= (l < -kBoundMax) | (l > kBoundMax) | (u < -kBoundMax) | (u > kBoundMax);
      }
    for (uint32_t v = V + tid; v < S; v += nth) { int2 d; d.x = d.y = a.m.const_val[v - V]; dom[v - V] = d; }
    if (bad | wide) atomicOr(&misc[M_FAIL], 1u);
    if (wide) atomicOr(&misc[M_OOB], 1u);
    }
  } else {
    // slot-major: the 2*B bound loads of a slot (one per node row, lanes = consecutive slots: coalesced) are issued
    // together — one memory round trip per pass instead of one per node
    uint32_t badm = 0, widem = 0;
    bool oob = false;
    for (uint32_t v = tid; v < S; v += nth) {
      int lbv[B], ubv[B];
      if (v < V) {
#pragma unroll
        for (int b = 0; b < B; ++b) {
          const size_t row = (size_t)(node0 + ((uint32_t)b < nb ? (uint32_t)b : 0u)) * V;  // missing nodes of a tail tile mirror node 0
          lbv[b] = a.lb_in[row + v];
          ubv[b] = a.ub_in[row + v];
        }
      } else {
        const int cv0 = a.m.const_val[v - V];
#pragma unroll
        for (int b = 0; b < B; ++b) lbv[b] = ubv[b] = cv0;
      }
#pragma unroll
      for (int b = 0; b < B; ++b) {
        if (lbv[b] > ubv[b] && (uint32_t)b < nb) badm |= 1u << b;  // empty input domain: the node is failed
        if constexpr (PACKED) {
          oob |= (lbv[b] < -kPackedMax) | (lbv[b] > kPackedMax) | (ubv[b] < -kPackedMax) | (ubv[b] > kPackedMax);
          dom[(size_t)v * BP + b] = pack16(lbv[b], ubv[b]);
        } else {
          // the engine's arithmetic is exact for |bound| < 2^29 (pcp_hip.h): a node beyond that is refused, not wrapped
          if (((lbv[b] < -kBoundMax) | (lbv[b] > kBoundMax) | (ubv[b] < -kBoundMax) | (ubv[b] > kBoundMax)) && (uint32_t)b < nb) widem |= 1u << b;
          dom[(size_t)v * BP + b] = make_int2(-lbv[b], ubv[b]);  // LDS holds (-lb, ub)
        }
      }
    }
    if (badm | widem) atomicOr(&misc[M_FAIL], badm | widem);  // (a refused node is inert like a failed one)
    if (PACKED && oob) atomicOr(&misc[M_OOB], 1u);
    if (!PACKED && widem) atomicOr(&misc[M_OOB], widem);
  }
  __syncthreads();
  if (!PACKED && misc[M_OOB] && tid == 0) atomicMax(a.violation, 1u);  // sticky: reported by pcp_stats_read
  if (PACKED && misc[M_OOB]) {
    // some bound of this tile does not fit the packed cells: hand the tile back untouched (pcp_api.hip launches the
    // 32-bit kernel right behind this one; it runs exactly the tiles marked here)
    if (tid < nb) a.status[node0 + tid] = kStatusRetry;
    if (tid == 0) atomicMax(a.retry_flag, a.epoch);
    return;
  }
  if constexpr (!GLOBAL && B >= (int)kSummMinTile) {
    // tile summaries for the sweep's level-0 test: per slot (min -lb, min ub) and (max -lb, max ub) over the tile's nodes
    if constexpr (PACKED) {
      uint32_t* tmin = reinterpret_cast<uint32_t*>(smem + cv.summ);
      const uint32_t Sp = tsw_slots(S);
      uint32_t* tmax = tmin + (size_t)Sp * rmq_levels(a.word_level, false);
      for (uint32_t v = tid; v < S; v += nth) {
        uint32_t mn = 0x7fff7fffu, mx = 0x80008000u;
#pragma unroll
        for (int b = 0; b < B; b += 4) {
          const uint4 q = *reinterpret_cast<const uint4*>(dom + (size_t)v * BP + b);
          mn = pk_min(pk_min(mn, q.x), pk_min(q.y, pk_min(q.z, q.w)));
          mx = pk_max(pk_max(mx, q.x), pk_max(q.y, pk_max(q.z, q.w)));
        }
        tmin[tsw(v)] = mn; tmax[tsw(v)] = mx;
      }
      // range tables of the level -1 test: level l holds the minimum (maximum) over slots [v, v + 2^l)
      if (a.word_level) {
        for (uint32_t l = 1; l < kRangeLevels; ++l) {
          __syncthreads();
          for (uint32_t v = tid; v < S; v += nth) {
            const uint32_t v2 = min(v + (1u << (l - 1)), S - 1);
            tmin[l * Sp + tsw(v)] = pk_min(tmin[(l - 1) * Sp + tsw(v)], tmin[(l - 1) * Sp + tsw(v2)]);
            if (a.word_level >= 2) tmax[l * Sp + tsw(v)] = pk_max(tmax[(l - 1) * Sp + tsw(v)], tmax[(l - 1) * Sp + tsw(v2)]);
          }
        }
      }
    } else {
      Cell* summ = reinterpret_cast<Cell*>(smem + cv.summ);
      for (uint32_t v = tid; v < S; v += nth) {
        int2 mn = make_int2(0x7fffffff, 0x7fffffff), mx = make_int2(-0x7fffffff - 1, -0x7fffffff - 1);
#pragma unroll
        for (int b = 0; b < B; ++b) {
          const int2 q = dom[(size_t)v * BP + b];
          mn.x = min(mn.x, q.x); mn.y = min(mn.y, q.y); mx.x = max(mx.x, q.x); mx.y = max(mx.y, q.y);
        }
        summ[2 * v] = mn; summ[2 * v + 1] = mx;
      }
    }
    __syncthreads();
  }

  Ctr ctr;
  uint64_t steps2 = 0, steps3 = 0;
  const unsigned long long t_begin = (PCP_ABLATE & 64) ? wall_clock64() : 0ull;  // profiling build: phase timers (100 MHz ticks)

  // ---- phase 1: wave 0 = every live propagator once (a slice of the table when team > 1) -----------------
  {
    uint32_t w0 = 0, w1 = words;
    if (team > 1) {  // slices are whole chunks
      const uint32_t ws = (((words + team - 1) / team) + kChunk - 1) / kChunk * kChunk;
      w0 = min(words, g * ws); w1 = min(words, w0 + ws);
    }
    // narrowings of wave 0 are recorded in `cur`, which the first wake-up round reads as its current set
    bool swept = false;
    if constexpr (PACKED && B <= 16) {
      if (a.word_level) {  // team == 1 here
        swept = true;
        if (!sweep_words<B, COMPACT, IMPLICIT>(a, k, node0, nb, cur, remaining, list_id, steps2, steps3, ctr, list_off, C))
          sweep_fast<B, GLOBAL, COMPACT, PACKED, IMPLICIT>(a, k, w0, w1, node0, nb, cur, remaining, steps2, steps3, ctr, reinterpret_cast<const uint64_t*>(list_id));
      }
    }
    if (!swept && w0 < w1) sweep_fast<B, GLOBAL, COMPACT, PACKED, IMPLICIT>(a, k, w0, w1, node0, nb, cur, remaining, steps2, steps3, ctr);  // w0 is chunk-aligned unless the slice is empty
  }
  // Every wave drains its own global stores (live words) before the barrier: a later atomicAnd on the same
  // word, or the team's release fence, must not be overtaken by them (cdna_hip_programming.md G16, R1).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // (the barrier also answers "did the sweep narrow anything in this tile?": if not, no variable is marked and the rounds —
  // their compaction pass, barrier and the registers spilled around them — are skipped altogether)
  const bool sweep_narrowed = __syncthreads_or(ctr.narrow != 0) != 0;

  // ---- phase 2 (team mode): merge this slice into the node's global arrays; the last arriver continues ---
  if constexpr (!PACKED) if (team > 1) {  // (packed tiles never run as a team: B >= 8)
    int32_t* glb = a.lb_out + (size_t)node0 * V;
    int32_t* gub = a.ub_out + (size_t)node0 * V;
    uint32_t* gchg = a.team_chg + (size_t)node0 * Wv;
    for (uint32_t w = tid; w < Wv; w += nth) {
      uint32_t bits = cur[w];
      if (bits) atomicOr(&gchg[w], bits);
      while (!GLOBAL && bits) {
        const uint32_t v = (w << 5) + __builtin_ctz(bits);
        bits &= bits - 1;
        if (v < V) { atomicMax(&glb[v], -dom[(size_t)v * BP].x); atomicMin(&gub[v], dom[(size_t)v * BP].y); }
      }
    }
    // counters: wave -> LDS -> ONE global atomic per block and counter (same-address device atomics serialise
    // at ~12 ns each: per-wave adds from a 512-block team would cost more than the sweep itself)
    for (int o = 32; o > 0; o >>= 1) { ctr.narrow += __shfl_down(ctr.narrow, o); ctr.ev += __shfl_down(ctr.ev, o); ctr.full += __shfl_down(ctr.full, o); }
    if (lane == 0) {
      if (steps2) atomicAdd(reinterpret_cast<unsigned long long*>(&misc[M_STEPS2]), (unsigned long long)steps2);
      if (steps3) atomicAdd(reinterpret_cast<unsigned long long*>(&misc[M_STEPS3]), (unsigned long long)steps3);
      if (ctr.narrow) atomicAdd(&misc[M_NARROW], ctr.narrow);
      if (ctr.ev) atomicAdd(reinterpret_cast<unsigned long long*>(&misc[M_EVAL]), (unsigned long long)ctr.ev);
      if (ctr.full) atomicAdd(reinterpret_cast<unsigned long long*>(&misc[M_FULL]), (unsigned long long)ctr.full);
    }
    ctr = Ctr(); steps2 = 0; steps3 = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its merge atomics have been performed
    __syncthreads();
    if (tid == 0) {
      const unsigned long long s2 = *reinterpret_cast<unsigned long long*>(&misc[M_STEPS2]);
      const unsigned long long s3 = *reinterpret_cast<unsigned long long*>(&misc[M_STEPS3]);
      if (s2) atomicAdd((unsigned long long*)&a.team_counters[(size_t)node0 * kTeamCounters + 0], s2);
      if (s3) atomicAdd((unsigned long long*)&a.team_counters[(size_t)node0 * kTeamCounters + 1], s3);
      if (misc[M_NARROW]) atomicAdd((unsigned long long*)&a.team_counters[(size_t)node0 * kTeamCounters + 2], (unsigned long long)misc[M_NARROW]);
      const unsigned long long sev = *reinterpret_cast<unsigned long long*>(&misc[M_EVAL]);
      const unsigned long long sfu = *reinterpret_cast<unsigned long long*>(&misc[M_FULL]);
      if (sev) atomicAdd((unsigned long long*)&a.team_counters[(size_t)node0 * kTeamCounters + 3], sev);
      if (sfu) atomicAdd((unsigned long long*)&a.team_counters[(size_t)node0 * kTeamCounters + 4], sfu);
      *reinterpret_cast<unsigned long long*>(&misc[M_STEPS2]) = 0;
      *reinterpret_cast<unsigned long long*>(&misc[M_STEPS3]) = 0;
      *reinterpret_cast<unsigned long long*>(&misc[M_EVAL]) = 0;
      *reinterpret_cast<unsigned long long*>(&misc[M_FULL]) = 0;
      misc[M_NARROW] = 0;
      if (remaining[0]) atomicAdd(&a.team_remaining[node0], remaining[0]);
      if (misc[M_FAIL]) atomicOr(&a.team_fail[node0], 1u);
      // publish: agent-scope release, drain, then take a ticket (MI355X_MICROARCH "valid forms")
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const uint32_t ticket = __hip_atomic_fetch_add(&a.team_ticket[node0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t last = (ticket == team - 1) ? 1u : 0u;
      if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      misc[M_ISLAST] = last;
    }
    __syncthreads();
    if (!misc[M_ISLAST]) return;
    // tail block: reload the merged state
    for (uint32_t v = tid; !GLOBAL && v < V; v += nth) {
      int2 d;
      d.x = -__hip_atomic_load(&glb[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      d.y = __hip_atomic_load(&gub[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      dom[(size_t)v * BP] = d;
    }
    for (uint32_t w = tid; w < Wv; w += nth) {
      cur[w] = __hip_atomic_load(&gchg[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      nxt[w] = 0;
    }
    if (tid == 0) {
      remaining[0] = __hip_atomic_load(&a.team_remaining[node0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      misc[M_FAIL] = __hip_atomic_load(&a.team_fail[node0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
    }
    __syncthreads();
  }

  // ---- phase 3: wake-up rounds until no variable changes (IndexedDeps::react + RelaxedFifo, as waves) ------
  const unsigned long long t_sweep = (PCP_ABLATE & 64) ? wall_clock64() : 0ull;
  const uint32_t TW = nb * Wv;
  // A round costs latency, not throughput (the tail of a cascade is one changed variable per round), so it is built from as
  // few block-wide steps as possible: ONE pass compacts the changed (node, variable) pairs with LDS atomics (their order is
  // irrelevant), a barrier, the incident records run, a barrier.  The counters alternate between two slots and the old
  // `cur` mask is cleared inside the next round's pass, so nothing else needs a barrier of its own.
  bool tile_clean = false;  // no variable of any node of the tile changed: the domains are the inputs
  uint32_t dbg_r[6] = {0, 0, 0, 0, 0, 0};  // profiling build (PCP_ABLATE & 1024): rounds by number of changed pairs, solo passes
  unsigned long long dbg_t[5] = {0, 0, 0, 0, 0}, dbg_t0 = 0;  // and their time (100 MHz ticks)
  uint32_t dbg_class = 0;
  unsigned long long dbg_slow = 0;  // slowest round: ticks << 40 | pairs << 28 | items
  if (team == 1 && !sweep_narrowed && misc[M_FAIL] == 0) tile_clean = true;
  else
  for (uint32_t round = 0;; ++round) {
    const uint32_t m_total = (round & 1u) ? M_TOTAL2 : M_TOTAL, m_items = (round & 1u) ? M_ITEMS2 : M_ITEMS, m_mask = (round & 1u) ? M_ROUNDMASK2 : M_ROUNDMASK;
    // (a) compact the changed pairs of the live nodes: (node, var), the variable's adjacency offset and its degree
    const uint32_t failm = misc[M_FAIL];
    {
      uint32_t degsum = 0;
      for (uint32_t w = tid; w < TW; w += nth) {
        if (round) nxt[w] = 0;  // the mask two rounds back (`cur` of the previous round): free since the last barrier
        const uint32_t b = w / Wv;
        if ((failm >> b) & 1u) continue;
        uint32_t bits = cur[w];
        if (!bits) continue;
        atomicOr(&misc[m_mask], 1u << b);
        uint32_t pos = atomicAdd(&misc[m_total], (uint32_t)__popc(bits));
        const uint32_t vbase = (w - b * Wv) << 5;
        while (bits) {
          const uint32_t v = vbase + __builtin_ctz(bits);
          bits &= bits - 1;
          if (pos < C) {
            const uint32_t o0 = (v < V) ? adjo[v] : 0u, o1 = (v < V) ? adjo[v + 1] : 0u;
            list_id[pos] = (b << 26) | v;
            list_off[pos] = o0;
            list_pre[pos] = o1 - o0;
            degsum += o1 - o0;
          }
          ++pos;
        }
      }
      if (degsum) atomicAdd(&misc[m_items], degsum);
    }
    __syncthreads();
    const uint32_t total = misc[m_total];
    if (total == 0 || (PCP_ABLATE & 32)) { tile_clean = round == 0 && total == 0 && failm == 0; break; }
    if (PCP_ABLATE & 1024) { dbg_class = total == 1 ? 0 : total <= 4 ? 1 : total <= 16 ? 2 : total <= C ? 3 : 4; ++dbg_r[dbg_class]; dbg_t0 = wall_clock64(); }
    if (tid == 0) {  // the other slot: last read before this round's barrier
      misc[M_WAVES] += __popc(misc[m_mask]);
      misc[(round & 1u) ? M_TOTAL : M_TOTAL2] = 0; misc[(round & 1u) ? M_ITEMS : M_ITEMS2] = 0; misc[(round & 1u) ? M_ROUNDMASK : M_ROUNDMASK2] = 0;
    }
    if (total <= C) {
      const bool high_degree = misc[m_items] >= 32u * total;
      if (!high_degree) {
        // low-degree path: turn the degrees into prefix sums (the flat item space below is balanced by binary search)
        const uint32_t per = (total + nth - 1) / nth;
        const uint32_t s = min(total, tid * per), e = min(total, s + per);
        uint32_t sum = 0;
        for (uint32_t i = s; i < e; ++i) sum += list_pre[i];
        uint32_t base = block_exclusive_scan(sum, tmp, &misc[M_SCAN]);
        for (uint32_t i = s; i < e; ++i) { const uint32_t dg = list_pre[i]; list_pre[i] = base; base += dg; }
        if (tid == 0) list_pre[total] = misc[M_SCAN];
        __syncthreads();
      }
      const uint32_t T = misc[m_items];
      uint32_t my2 = 0, my3 = 0, myf = 0;
      // one item = one (changed var, incident record).  Runs a live record woken from variable v of node b unless a
      // lower-numbered changed variable of the same record will run it (RelaxedFifo dedup, relaxed_fifo.rs:42-48).
      auto run_item = [&](uint32_t b, uint32_t v, uint32_t r, uint32_t lbits, const Rec rec) -> bool {  // true = ran and is entailed
        const uint32_t bit = 1u << (r & 31);
        if (!IMPLICIT && !(lbits & bit)) return false;  // unlinked (store.rs:200-207)
        const uint32_t* cb = cur + (size_t)b * Wv;
        const uint32_t x = rec.xk & kSlotMask;
        const bool tern = (rec.xk >> 28) > PCP_LT;
        if (x < v && ((cb[x >> 5] >> (x & 31)) & 1u)) return false;
        if (rec.y < v && ((cb[rec.y >> 5] >> (rec.y & 31)) & 1u)) return false;
        if (tern && rec.z < v && ((cb[rec.z >> 5] >> (rec.z & 31)) & 1u)) return false;
        const auto dm = make_dom<GLOBAL, PACKED>(k, b, nxt, &ctr);
        if (tern) ++my3; else ++my2;
        ++myf;
        const bool entailed = eval_record(rec, dm);
        if constexpr (!IMPLICIT) {
          if (entailed) {
            uint32_t* lw = reinterpret_cast<uint32_t*>(a.live + (size_t)(node0 + b) * words) + (r >> 5);
            const uint32_t old = atomicAnd(lw, ~bit);
            if (old & bit) atomicSub(&remaining[b], 1u);
          }
        }
        return entailed;
      };
      if (high_degree) {
        // (c1) high-degree variables (N-queens: 2997 records each): the adjacency lists are walked in pieces of 4 x 64 entries,
        // one item per lane and u — the index / payload loads are coalesced and all four are in flight together.
        const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = nth >> 6;
        constexpr int U = 4;
        bool solo = false;
        // (compiled for packed tiles and one-node blocks only: with it the 32-bit tile kernels spill ~70 VGPRs for a case
        // — a lone straggler among the B nodes of such a tile — that their workloads hardly have)
        constexpr bool kSolo = !GLOBAL && PCP_SOLO && (PACKED || B == 1);
        if constexpr (kSolo) solo = a.solo && total == 1 && list_pre[0] <= nwv * 64 * U && C >= 192;
        {
          // every piece is two stages: the coalesced streams (record ids, payloads) and what depends on
          // them (live words, records without payloads).  The first stage of the NEXT piece is issued before the current one
          // is evaluated: at 16 wavefronts per CU nothing else hides the ~2 us those loads take, and a round of a few dozen
          // changed variables is a chain of ~40 pieces per wavefront.
          struct Piece { uint32_t b, v, deg, aoff, k0; };
          // Every wavefront visits every pair and takes the 256-entry pieces p with (p + e) % nwv == wv of pair e's list: a
          // round with a single changed variable (the tail of a long cascade) is spread over the whole workgroup, and a round
          // of a few dozen pairs is balanced to within a piece whatever the number of pairs.
          const uint32_t k_step = nwv * 64 * U;
          // (two copies of the loop: binary-only models carry the record in the adjacency payload and need no ids unless
          // `active` rows are explicit; keeping them apart keeps each within the register budget)
          auto walk = [&](auto with_payload) {
            constexpr bool PAY = decltype(with_payload)::value;
            constexpr bool IDS = !(IMPLICIT && PAY);
            auto stage1 = [&](const Piece& pc, uint32_t (&r)[U], uint2 (&q)[U]) {
#pragma unroll
              for (int u = 0; u < U; ++u) {
                const uint32_t idx = pc.k0 + u * 64 + lane;
                const uint32_t at = pc.aoff + (idx < pc.deg ? idx : 0u);
                r[u] = IDS ? a.m.adj[at] : 0u;
                q[u] = PAY ? a.m.adjp[at] : make_uint2(0u, 0u);
              }
            };
            auto process = [&](auto solo_c, const Piece& pc, const uint32_t (&r)[U], const uint2 (&q)[U]) {
              constexpr bool SOLO = decltype(solo_c)::value;
              const uint32_t* cb = cur + (size_t)pc.b * Wv;
              const auto dm = make_dom<GLOBAL, PACKED>(k, pc.b, nxt, &ctr);
              if constexpr (IMPLICIT && PAY && !GLOBAL) if (PACKED || !dm.any_sums()) {
                // The common case written out: binary records from payloads over LDS cells, no `active` rows, no Sum views.  A wave is
                // VALU-bound here (4 items per lane, ~20 of them a round per changed variable), so an item that cannot act
                // costs the payload decode, one cell, one dedup word and the test.  With v on either side of x ◇ y + d and
                // t = d if v is y, -d if v is x:  x != y + d can act iff lb(v) + t == ub(other) or ub(v) + t == lb(other)
                // (fast_flag's condition), and a singleton other side forbids the value other - t for v.
                using Cell = typename CellOf<PACKED>::type;
                const Cell* dcol = static_cast<const Cell*>(k.dom) + pc.b;
                auto bounds = [](const Cell c) -> int2 {  // (lb, ub)
                  if constexpr (PACKED) return unpack16(c);
                  else return make_int2(-c.x, c.y);
                };
                const int2 Vd = bounds(dcol[(size_t)pc.v * k.bp]);
                Cell oc[U];
                uint32_t cw[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                  const uint32_t other = q[u].x & kSlotMask;
                  oc[u] = dcol[(size_t)other * k.bp];
                  cw[u] = cb[other >> 5];
                }
                // no branch until an item can act: the predicates are folded into one mask per piece
                const bool all_neq = a.m.uniform_kind == PCP_NEQ;
                uint32_t actm = 0;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                  const uint32_t other = q[u].x & kSlotMask;
                  const bool is_y = (q[u].x >> 31) != 0;
                  const bool live = (pc.k0 + u * 64 + lane < pc.deg) && !(other < pc.v && ((cw[u] >> (other & 31)) & 1u));  // RelaxedFifo dedup (relaxed_fifo.rs:42-48)
                  my2 += live ? 1u : 0u;
                  const int2 O = bounds(oc[u]);
                  const int d = (int32_t)q[u].y;
                  const int t = is_y ? d : -d;
                  bool act = (Vd.x + t == O.y) || (Vd.y + t == O.x);
                  if (!all_neq) {
                    const uint32_t kind = (q[u].x >> 28) & 7u;
                    const int2 X = is_y ? O : Vd, Y = is_y ? Vd : O;
                    const int Yl = Y.x + d, Yu = Y.y + d;
                    if (kind != PCP_NEQ) act = kind == PCP_LT ? (X.y >= Yu || Yl <= X.x) : (X.x != Yl || X.y != Yu);
                  }
                  actm |= (live && act) ? 1u << u : 0u;
                  if constexpr (SOLO) {
                    if (live && O.x == O.y && other != pc.v && (all_neq || ((q[u].x >> 28) & 7u) == PCP_NEQ)) {
                      const int f = O.x - t;
                      const uint32_t dl = (uint32_t)(f - (int)misc[M_BASE_LO]), dh = (uint32_t)((int)misc[M_BASE_HI] - f);
                      if (dl < 4096u) atomicOr(&list_off[64 + (dl >> 5)], 1u << (dl & 31));
                      if (dh < 4096u) atomicOr(&list_pre[64 + (dh >> 5)], 1u << (dh & 31));
                    }
                  }
                }
                if (actm) {
#pragma unroll
                  for (int u = 0; u < U; ++u) {
                    if (!((actm >> u) & 1u)) continue;
                    const uint32_t other = q[u].x & kSlotMask, kind = (q[u].x >> 28) & 7u;
                    const bool is_y = (q[u].x >> 31) != 0;
                    ++myf;
                    Rec rec;
                    rec.xk = (is_y ? other : pc.v) | (kind << 28);
                    rec.y = is_y ? pc.v : other;
                    rec.z = 0;
                    rec.d = (int32_t)q[u].y;
                    eval_record(rec, dm);
                  }
                }
                return;
              }
              uint32_t lw[U];
              Rec rc[U];
              if constexpr (!IMPLICIT) {
                const uint32_t* lrow = reinterpret_cast<const uint32_t*>(a.live + (size_t)(node0 + pc.b) * words);
#pragma unroll
                for (int u = 0; u < U; ++u) lw[u] = __hip_atomic_load(lrow + (r[u] >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
#pragma unroll
              for (int u = 0; u < U; ++u) {
                if constexpr (PAY) {
                  const uint32_t other = q[u].x & kSlotMask, kind = (q[u].x >> 28) & 7u;
                  const bool is_y = (q[u].x >> 31) != 0;
                  rc[u].xk = (is_y ? other : pc.v) | (kind << 28);
                  rc[u].y = is_y ? pc.v : other;
                  rc[u].z = 0;
                  rc[u].d = (int32_t)q[u].y;
                } else {
                  rc[u] = a.m.recs[r[u]];
                }
              }
              // the four items together: dedup words and domains are read for all of them before the first test, and the
              // filter proper only runs where the domains say it can act (fast_flag's conditions, per lane)
              bool run[U];
              int2 X[U], Y[U];
#pragma unroll
              for (int u = 0; u < U; ++u) {
                const uint32_t x = rc[u].xk & kSlotMask, y = rc[u].y;
                const uint32_t wx = cb[x >> 5], wy = cb[y >> 5];
                bool go = pc.k0 + u * 64 + lane < pc.deg;
                if constexpr (!IMPLICIT) go = go && ((lw[u] >> (r[u] & 31)) & 1u);  // unlinked (store.rs:200-207)
                go = go && !(x < pc.v && ((wx >> (x & 31)) & 1u)) && !(y < pc.v && ((wy >> (y & 31)) & 1u));  // RelaxedFifo dedup (relaxed_fifo.rs:42-48)
                run[u] = go;
                X[u] = dm.load(x);
                Y[u] = dm.load(y);
              }
#pragma unroll
              for (int u = 0; u < U; ++u) {
                if (!run[u]) continue;
                const uint32_t kind = rc[u].xk >> 28;
                if constexpr (SOLO) {
                  // x != y + d with a singleton on the other side of v: its value is forbidden for v (window marks, see c0)
                  const uint32_t x = rc[u].xk & kSlotMask;
                  const int2 O = (x == pc.v) ? Y[u] : X[u];
                  if (kind == PCP_NEQ && x != rc[u].y && O.x == O.y) {
                    const int f = (x == pc.v) ? O.x + rc[u].d : O.x - rc[u].d;
                    const uint32_t dl = (uint32_t)(f - (int)misc[M_BASE_LO]), dh = (uint32_t)((int)misc[M_BASE_HI] - f);
                    if (dl < 4096u) atomicOr(&list_off[64 + (dl >> 5)], 1u << (dl & 31));
                    if (dh < 4096u) atomicOr(&list_pre[64 + (dh >> 5)], 1u << (dh & 31));
                  }
                }
                if (!PAY && kind > PCP_LT) {
                  const uint32_t z = rc[u].z;
                  if (z < pc.v && ((cb[z >> 5] >> (z & 31)) & 1u)) continue;
                  ++my3;
                } else {
                  ++my2;
                  if (!dm.any_sums()) {
                    const int Yl = Y[u].x + rc[u].d, Yu = Y[u].y + rc[u].d;
                    bool act;
                    if constexpr (IMPLICIT) act = kind == PCP_NEQ ? (X[u].x == Yu || Yl == X[u].y) : kind == PCP_LT ? (X[u].y >= Yu || Yl <= X[u].x) : (X[u].x != Yl || X[u].y != Yu);
                    else act = kind == PCP_NEQ ? (X[u].x >= Yu || Yl >= X[u].y) : kind == PCP_LT ? (X[u].y >= Yu || Yl <= X[u].x || X[u].y < Yl) : (X[u].x != Yl || X[u].y != Yu || X[u].x == X[u].y);
                    if (!act) continue;
                  }
                }
                ++myf;
                const bool entailed = eval_record(rc[u], dm);
                if constexpr (!IMPLICIT) {
                  if (entailed) {
                    const uint32_t bit = 1u << (r[u] & 31);
                    uint32_t* lwp = reinterpret_cast<uint32_t*>(a.live + (size_t)(node0 + pc.b) * words) + (r[u] >> 5);
                    const uint32_t old = atomicAnd(lwp, ~bit);
                    if (old & bit) atomicSub(&remaining[pc.b], 1u);
                  }
                }
              }
            };
            // (lists of at most one piece: pair e simply belongs to wavefront e % nwv)
            const bool one_piece = a.m.max_deg <= 64u * U;
            const uint32_t e_step = one_piece ? nwv : 1u;
            auto k_first = [&](uint32_t e_) { return one_piece ? 0u : ((wv + nwv - (e_ % nwv)) % nwv) * 64 * U; };
            uint32_t e = one_piece ? wv : 0u, k0 = k_first(e);
            auto settle = [&]() { while (e < total && k0 >= list_pre[e]) { e += e_step; k0 = k_first(e); } };
            auto piece_at = [&](uint32_t e_, uint32_t k_) { const uint32_t id = list_id[e_]; return Piece{id >> 26, id & ((1u << 26) - 1), list_pre[e_], list_off[e_], k_}; };
            settle();
            bool have = e < total;
            Piece pa{0, 0, 0, 0, 0};
            uint32_t rA[U];
            uint2 qA[U];
            if (have) { pa = piece_at(e, k0); stage1(pa, rA, qA); }
            if constexpr (kSolo) if (solo) {
              // (c0) the tail of a cascade: ONE changed variable v of ONE node.  Its whole adjacency list is held in the
              // workgroup's registers (one piece per wavefront), so the records are fetched once and re-run while v itself
              // keeps changing — a pass costs two barriers instead of a round (compaction, fetch, three barriers).  And the
              // commonest chain, a bound of v walking through values that assigned neighbours forbid (x != y + c removes a
              // value only at a bound, x_neq_y.rs:82-93, so the reference takes one wake-up per value), is taken in one
              // jump: every live x != y + c whose other side is a singleton marks its forbidden value in a 4096-bit window
              // above lb(v) / below ub(v) (entries >= 64 of the pair list are free in this round), and wave 0 moves the
              // bound to the first unmarked value.  Each skipped value is one the NEQ filter would remove at the bound, so
              // the fixpoint is the same (it is unique); v leaves the wake list because all its records have run after its
              // last change.
              const uint32_t id = list_id[0], b = id >> 26, v = id & ((1u << 26) - 1);
              uint32_t* flo = list_off + 64;
              uint32_t* fhi = list_pre + 64;
              uint32_t* vw = nxt + (size_t)b * Wv + (v >> 5);
              const uint32_t vbit = 1u << (v & 31);
              const auto dmv = make_dom<GLOBAL, PACKED>(k, b, nxt, &ctr);
              if (tid < 128) { flo[tid] = 0; fhi[tid] = 0; }
              if (tid == 0) { const int2 d = dmv.load(v); misc[M_BASE_LO] = (uint32_t)d.x; misc[M_BASE_HI] = (uint32_t)d.y; }
              __syncthreads();
              for (;;) {
                if (have) process(std::true_type{}, pa, rA, qA);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (wv == 0) {
                  const int blo = (int)misc[M_BASE_LO], bhi = (int)misc[M_BASE_HI];
                  const int2 d = dmv.load(v);
                  const bool dead = ((misc[M_FAIL] >> b) & 1u) != 0 || d.x > d.y;
                  // first unmarked position >= off of a 4096-bit window, 64 bits per lane (4096 if there is none)
                  auto first_free = [&](const uint32_t* win, uint32_t off) -> uint32_t {
                    unsigned long long w = ~(((unsigned long long)win[2 * lane + 1] << 32) | win[2 * lane]);
                    const uint32_t base = lane * 64;
                    if (base + 63 < off) w = 0;
                    else if (base < off) w &= ~0ull << (off - base);
                    const unsigned long long has = __ballot(w != 0);
                    if (!has) return 4096u;
                    const int src = __builtin_ctzll(has);
                    const uint32_t mine = base + (w ? (uint32_t)__builtin_ctzll(w) : 0u);
                    return (uint32_t)__shfl((int)mine, src);
                  };
                  uint32_t go = 0;
                  if (!dead) {
                    const uint32_t pl = first_free(flo, (uint32_t)(d.x - blo));
                    const uint32_t ph = first_free(fhi, (uint32_t)(bhi - d.y));
                    if (lane == 0) {
                      const int nl = min(blo + (int)pl, d.y + 1);
                      if (nl > d.x) dmv.raise_lb(v, nl);
                      if (nl <= d.y) {
                        const int nu = bhi - (int)ph;
                        if (nu < d.y) dmv.lower_ub(v, nu);
                      }
                      const bool failed = ((__hip_atomic_load(&misc[M_FAIL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> b) & 1u) != 0;
                      const bool changed = (__hip_atomic_load(vw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & vbit) != 0;
                      if (changed && !failed) {
                        go = 1;
                        atomicAnd(vw, ~vbit);
                        const int2 d2 = dmv.load(v);
                        misc[M_BASE_LO] = (uint32_t)d2.x; misc[M_BASE_HI] = (uint32_t)d2.y;
                        misc[M_WAVES] += 1;
                      }
                    }
                  }
                  if (lane == 0) misc[M_SOLO] = go;
                  flo[lane] = 0; flo[lane + 64] = 0; fhi[lane] = 0; fhi[lane + 64] = 0;
                }
                __syncthreads();
                if (!misc[M_SOLO]) break;
                if (PCP_ABLATE & 1024) ++dbg_r[5];
              }
              have = false;
            }
            while (have) {
              k0 += k_step;
              if (k0 >= pa.deg) { e += e_step; k0 = k_first(e); }
              settle();
              const bool have_n = e < total;
              Piece pb = pa;
              uint32_t rB[U];
              uint2 qB[U];
              if (have_n) { pb = piece_at(e, k0); stage1(pb, rB, qB); }
              process(std::false_type{}, pa, rA, qA);
              if (have_n) {
                pa = pb;
#pragma unroll
                for (int u = 0; u < U; ++u) { rA[u] = rB[u]; qA[u] = qB[u]; }
              }
              have = have_n;
            }
          };
          if (a.m.adjp) walk(std::true_type{}); else walk(std::false_type{});
        }
      } else {
        // (c2) low-degree variables: flat item space, load-balanced over the whole block by binary search in the
        // degree prefix sums
        for (uint32_t i = tid; i < T; i += nth) {
          uint32_t lo = 0, hi = total;
          while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (list_pre[mid] <= i) lo = mid; else hi = mid; }
          const uint32_t id = list_id[lo], b = id >> 26, v = id & ((1u << 26) - 1);
          const uint32_t ai = list_off[lo] + (i - list_pre[lo]);
          const uint32_t r = a.m.adj[ai];
          uint32_t lbits = ~0u;
          if constexpr (!IMPLICIT) {
            const uint32_t* lrow = reinterpret_cast<const uint32_t*>(a.live + (size_t)(node0 + b) * words);
            lbits = __hip_atomic_load(lrow + (r >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          Rec rec;
          if (a.m.adjp) {
            const uint2 q = a.m.adjp[ai];
            const uint32_t other = q.x & kSlotMask, kind = (q.x >> 28) & 7u;
            const bool is_y = (q.x >> 31) != 0;
            rec.xk = (is_y ? other : v) | (kind << 28); rec.y = is_y ? v : other; rec.z = 0; rec.d = (int32_t)q.y;
          } else {
            rec = a.m.recs[r];
          }
          run_item(b, v, r, lbits, rec);
        }
      }
      for (int o = 32; o > 0; o >>= 1) { my2 += __shfl_down(my2, o); my3 += __shfl_down(my3, o); myf += __shfl_down(myf, o); }
      my2 = __builtin_amdgcn_readfirstlane(my2); my3 = __builtin_amdgcn_readfirstlane(my3); myf = __builtin_amdgcn_readfirstlane(myf);
      steps2 += my2; steps3 += my3;
      ctr.add_ev_uniform(my2 + my3);
      ctr.add_full_uniform(myf);
    } else {
      __syncthreads();
      uint32_t rem_sub = 0;
      sweep_filtered<GLOBAL, PACKED, IMPLICIT>(a, k, node0, nb, cur, nxt, rem_sub, steps2, steps3, ctr);
      if (lane < nb && rem_sub) atomicSub(&remaining[lane], rem_sub);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (PCP_ABLATE & 1024) {
      const unsigned long long dt = wall_clock64() - dbg_t0;
      dbg_t[dbg_class] += dt;
      if (dt > (dbg_slow >> 40)) dbg_slow = dt << 40 | (unsigned long long)(total & 0xfffu) << 28 | (misc[m_items] & 0xfffffffu);
    }
    uint32_t* t = cur; cur = nxt; nxt = t;  // the old `cur` is cleared by the next round's pass
  }

  // ---- phase 3b (implicit `active`): is anything NOT entailed under the final domains? -------------------------
  // Consistency::consistency returns True iff no subscription remains (store.rs:250-256), i.e. iff every propagator is
  // entailed under the final domains (A.4).  At the fixpoint every filter is a no-op, so eval_record only reports
  // is_subsumed(); the scan stops as soon as every node of the tile has shown one propagator that is not entailed —
  // for an Unknown node that is its first or second word.
  if constexpr (IMPLICIT) {
    __syncthreads();
    const uint32_t wv3 = __builtin_amdgcn_readfirstlane(tid >> 6), nwv3 = nth >> 6;
    const uint32_t want = (nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1u)) & ~misc[M_FAIL];
    if (tile_clean && misc[M_OPEN0]) {  // shown at staging time and nothing changed since (sweep_words, group level)
      if (tid == 0) misc[M_UNK] = want;
    } else
    for (uint32_t w = wv3; w < words; w += nwv3) {
      uint32_t have = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&misc[M_UNK], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      if ((have & want) == want) break;
      const uint32_t r = (w << 6) + lane;
      const bool valid = r < P;
      const Rec rec = a.m.recs[valid ? r : P - 1];
      uint32_t todo = want & ~have;
      while (todo) {
        const uint32_t b = __builtin_ctz(todo);
        todo &= todo - 1;
        bool open_rec = false;
        if (valid) {
          const auto dm = make_dom<GLOBAL, PACKED>(k, b, nxt, &ctr);
          open_rec = !eval_record(rec, dm);
        }
        if (__ballot(open_rec) != 0 && lane == 0) atomicOr(&misc[M_UNK], 1u << b);
      }
    }
  }

  // ---- phase 4: write back domains, status, counters ---------------------------------------------------
  __syncthreads();
  if ((PCP_ABLATE & 64) && tid == 0) {  // profiling build: steps3 := sweep ticks, narrowings := rounds ticks (summed over blocks)
    const unsigned long long t_end = wall_clock64();
    atomicAdd((unsigned long long*)&a.stats->steps3, t_sweep - t_begin);
    atomicAdd((unsigned long long*)&a.stats->narrowings, t_end - t_sweep);
    if (PCP_ABLATE & 256) {
      atomicAdd((unsigned long long*)&a.stats->failed_nodes, t_begin - t_entry);  // staging (phase 0)
    } else {
      atomicMax((unsigned long long*)&a.stats->failed_nodes, t_sweep - t_begin);   // slowest block's sweep
      atomicMax((unsigned long long*)&a.stats->waves, t_end - t_sweep);            // slowest block's rounds
    }
  }
  if constexpr (GLOBAL) {
    bool bad = false;  // the domains are already in place; a missed failure shows as an empty domain here
    if (a.dom10) {
      const GlobalDom dmw = make_dom<GLOBAL, PACKED>(k, 0, nxt, &ctr);
      for (uint32_t v = tid; v < V; v += nth) {
        const int2 d = dmw.load10(v);
        bad |= d.x > d.y;
        k.glb[v] = d.x; k.gub[v] = d.y;
      }
    } else
    for (uint32_t v = tid; v < V; v += nth)
      bad |= __hip_atomic_load(&k.glb[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > __hip_atomic_load(&k.gub[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (bad) atomicOr(&misc[M_FAIL], 1u);
  } else if (tile_clean && a.lb_in == a.lb_out && a.ub_in == a.ub_out) {
    // in place and nothing narrowed in this tile: the rows in HBM already hold the result
  } else {
    for (uint32_t b = 0; b < nb; ++b) {
      if (!PACKED && ((misc[M_OOB] >> b) & 1u)) continue;  // a refused node's outputs are left alone
      int32_t* lbp = a.lb_out + (size_t)(node0 + b) * V;
      int32_t* ubp = a.ub_out + (size_t)(node0 + b) * V;
      bool bad = false;
      for (uint32_t v = tid; v < S; v += nth) {
        int2 d;  // (lb, ub)
        if constexpr (PACKED) d = unpack16(dom[(size_t)v * BP + b]);
        else { const int2 t = dom[(size_t)v * BP + b]; d = make_int2(-t.x, t.y); }
        bad |= d.x > d.y;
        if (v < V) { lbp[v] = d.x; ubp[v] = d.y; }
      }
      if (bad) atomicOr(&misc[M_FAIL], 1u << b);
    }
  }
  for (int o = 32; o > 0; o >>= 1) { ctr.narrow += __shfl_down(ctr.narrow, o); ctr.ev += __shfl_down(ctr.ev, o); ctr.full += __shfl_down(ctr.full, o); }
  if (lane == 0) {
    if (ctr.narrow) atomicAdd(&misc[M_NARROW], ctr.narrow);
    if (ctr.ev) atomicAdd(reinterpret_cast<unsigned long long*>(&misc[M_EVAL]), (unsigned long long)ctr.ev);
    if (ctr.full) atomicAdd(reinterpret_cast<unsigned long long*>(&misc[M_FULL]), (unsigned long long)ctr.full);
    if (steps2) atomicAdd(reinterpret_cast<unsigned long long*>(&misc[M_STEPS2]), (unsigned long long)steps2);
    if (steps3) atomicAdd(reinterpret_cast<unsigned long long*>(&misc[M_STEPS3]), (unsigned long long)steps3);
  }
  __syncthreads();
  if (tid < nb) {
    const bool failed = (misc[M_FAIL] >> tid) & 1u;
    // Consistency::consistency (store.rs:250-256): False if a propagate failed, True if no subscription
    // remains (every live propagator got entailed), else Unknown.
    const bool none_open = IMPLICIT ? !((misc[M_UNK] >> tid) & 1u) : remaining[tid] == 0;
    const bool refused = !PACKED && ((misc[M_OOB] >> tid) & 1u);  // a bound beyond +-kBoundMax
    a.status[node0 + tid] = refused ? kStatusRetry : failed ? (uint8_t)PCP_FALSE : (none_open ? (uint8_t)PCP_TRUE : (uint8_t)PCP_UNKNOWN);
  }
  if (tid == 0) {
    unsigned long long s2 = *reinterpret_cast<unsigned long long*>(&misc[M_STEPS2]);
    unsigned long long s3 = *reinterpret_cast<unsigned long long*>(&misc[M_STEPS3]);
    unsigned long long nr = misc[M_NARROW];
    unsigned long long sev = *reinterpret_cast<unsigned long long*>(&misc[M_EVAL]);
    unsigned long long sfu = *reinterpret_cast<unsigned long long*>(&misc[M_FULL]);
    if (team > 1) {
      sev += __hip_atomic_load(&a.team_counters[(size_t)node0 * kTeamCounters + 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sfu += __hip_atomic_load(&a.team_counters[(size_t)node0 * kTeamCounters + 4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s2 += __hip_atomic_load(&a.team_counters[(size_t)node0 * kTeamCounters + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s3 += __hip_atomic_load(&a.team_counters[(size_t)node0 * kTeamCounters + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      nr += __hip_atomic_load(&a.team_counters[(size_t)node0 * kTeamCounters + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (s2) atomicAdd((unsigned long long*)&a.stats->steps, s2);
    if (!(PCP_ABLATE & 1024)) {
    if (s3) atomicAdd((unsigned long long*)&a.stats->steps3, s3);
    if (nr) atomicAdd((unsigned long long*)&a.stats->narrowings, nr);
    if (sev) atomicAdd((unsigned long long*)&a.stats->evaluated, sev);
    if (sfu) atomicAdd((unsigned long long*)&a.stats->full_evals, sfu);
    atomicAdd((unsigned long long*)&a.stats->waves, (unsigned long long)(nb + misc[M_WAVES]));
    }
    if (PCP_ABLATE & 1024) atomicMax((unsigned long long*)&a.stats->nodes, dbg_slow);
    else atomicAdd((unsigned long long*)&a.stats->nodes, (unsigned long long)nb);
    const uint32_t nf = __popc(misc[M_FAIL] & (nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1)));
    if (nf && !(PCP_ABLATE & 1024)) atomicAdd((unsigned long long*)&a.stats->failed_nodes, (unsigned long long)nf);
    if ((PCP_ABLATE & 320) == 320) atomicAdd((unsigned long long*)&a.stats->nodes, wall_clock64() - t_entry);  // whole block
    if (PCP_ABLATE & 1024) {  // rounds histogram: per class count << 40 | ticks, summed over tiles; slowest tile's rounds time in waves
      atomicAdd((unsigned long long*)&a.stats->steps3, (unsigned long long)dbg_r[0] << 40 | dbg_t[0]);
      atomicAdd((unsigned long long*)&a.stats->narrowings, (unsigned long long)dbg_r[1] << 40 | dbg_t[1]);
      atomicAdd((unsigned long long*)&a.stats->failed_nodes, (unsigned long long)dbg_r[2] << 40 | dbg_t[2]);
      atomicAdd((unsigned long long*)&a.stats->evaluated, (unsigned long long)dbg_r[3] << 40 | dbg_t[3]);
      atomicAdd((unsigned long long*)&a.stats->full_evals, (unsigned long long)dbg_r[4] << 40 | dbg_t[4]);
      const unsigned long long tt = dbg_t[0] + dbg_t[1] + dbg_t[2] + dbg_t[3] + dbg_t[4];
      // slowest tile: total ticks << 40 | ticks in (>C) << 20 | ticks in (17..C)
      atomicMax((unsigned long long*)&a.stats->waves, tt << 40 | (dbg_t[4] & 0xfffff) << 20 | (dbg_t[3] & 0xfffff));
    }
  }
}

#if PCP_TU == 0
// ------------------------------------------------------------------------------------------------
// Conjunction / Distinct units (logic/conjunction.rs:77-119, propagators/distinct.rs:63-126).  The reference keeps
// ONE active bit for the whole conjunction: it is entailed iff every member is (conjunction.rs:78-94), and a pop
// runs every member (conjunction.rs:97-104).  The fixpoint kernel works on elementary records, so a grouped model
// carries a record-level live mask: expand_units seeds it (member live <=> its unit active) and contract_units
// folds it back (unit active <=> some member still live, i.e. not all members entailed under the final domains).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) expand_units_kernel(const uint32_t* __restrict__ rec_unit, uint32_t n_recs, uint32_t unit_words,
                                                           const uint64_t* __restrict__ active_in, uint64_t* __restrict__ live, uint32_t n_nodes) {
  const uint32_t rec_words = (n_recs + 63) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint64_t i = wave; i < (uint64_t)n_nodes * rec_words; i += nwaves) {
    const uint32_t node = (uint32_t)(i / rec_words), w = (uint32_t)(i % rec_words);
    const uint32_t r = (w << 6) + lane;
    bool on = false;
    if (r < n_recs) {
      const uint32_t u = rec_unit[r];
      on = active_in ? ((active_in[(size_t)node * unit_words + (u >> 6)] >> (u & 63)) & 1ull) : true;
    }
    const uint64_t word = __ballot(on);
    if (lane == 0) live[(size_t)node * rec_words + w] = word;
  }
}

__global__ void __launch_bounds__(256) contract_units_kernel(const uint32_t* __restrict__ unit_first, uint32_t n_units, uint32_t n_recs,
                                                             const uint64_t* __restrict__ live, uint64_t* __restrict__ active_out, uint32_t n_nodes) {
  const uint32_t rec_words = (n_recs + 63) >> 6, unit_words = (n_units + 63) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint64_t i = wave; i < (uint64_t)n_nodes * unit_words; i += nwaves) {
    const uint32_t node = (uint32_t)(i / unit_words), uw = (uint32_t)(i % unit_words);
    const uint32_t u = (uw << 6) + lane;
    const uint64_t* row = live + (size_t)node * rec_words;
    bool on = false;
    if (u < n_units) {
      const uint32_t r0 = unit_first[u], r1 = unit_first[u + 1];
      if (r1 - r0 == 1) {
        on = (row[r0 >> 6] >> (r0 & 63)) & 1ull;
      } else {
        for (uint32_t w = r0 >> 6; w <= ((r1 - 1) >> 6) && !on; ++w) {
          uint64_t m = row[w];
          if (w == (r0 >> 6)) m &= ~0ull << (r0 & 63);
          if (w == ((r1 - 1) >> 6) && (r1 & 63)) m &= (1ull << (r1 & 63)) - 1;
          on = m != 0;
        }
      }
    }
    const uint64_t word = __ballot(on);
    if (lane == 0) active_out[(size_t)node * unit_words + uw] = word;
  }
}

// ------------------------------------------------------------------------------------------------
// `active` rows on request for implicit-active nodes: bit r = record r is NOT entailed under the node's final domains
// (a propagator is inactive iff it is entailed under the final domains).  At a fixpoint every filter is a
// no-op, so eval_record over a read-only view of the final bounds only reports is_subsumed().
// ------------------------------------------------------------------------------------------------
struct ConstDom {
  const int32_t* lb;
  const int32_t* ub;
  const int32_t* cval;
  uint32_t n_vars;
  SumTab sums;
  __device__ __forceinline__ int2 load_var(uint32_t v) const { return make_int2(lb[v], ub[v]); }
  __device__ __forceinline__ bool any_sums() const { return sums.count != 0; }
  __device__ __forceinline__ const int32_t* mul_offsets() const { return sums.mul_off; }
  __device__ __forceinline__ int2 load(uint32_t v) const {
    if (v - sums.first < sums.count) return sum_read(*this, sums, v);
    if (v >= n_vars) { const int c = cval[v - n_vars]; return make_int2(c, c); }
    return make_int2(lb[v], ub[v]);
  }
  __device__ __forceinline__ void raise_lb(uint32_t, int) const {}
  __device__ __forceinline__ void lower_ub(uint32_t, int) const {}
  __device__ __forceinline__ void set_fail() const {}
};
__global__ void __launch_bounds__(256) derive_active_kernel(const ModelDev m, const int32_t* __restrict__ lb, const int32_t* __restrict__ ub,
                                                            uint64_t* __restrict__ live, uint32_t n_nodes) {
  const uint32_t words = (m.n_recs + 63) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint64_t i = wave; i < (uint64_t)n_nodes * words; i += nwaves) {
    const uint32_t node = (uint32_t)(i / words), w = (uint32_t)(i % words);
    const uint32_t r = (w << 6) + lane;
    bool on = false;
    if (r < m.n_recs) {
      const ConstDom dm{lb + (size_t)node * m.n_vars, ub + (size_t)node * m.n_vars, m.const_val, m.n_vars, m.sums};
      on = !eval_record(m.recs[r], dm);
    }
    const uint64_t word = __ballot(on);
    if (lane == 0) live[(size_t)node * words + w] = word;
  }
}
hipError_t launch_derive_active(const ModelDev& m, const int32_t* lb, const int32_t* ub, uint64_t* live, uint32_t n_nodes, hipStream_t stream) {
  const uint64_t items = (uint64_t)n_nodes * ((m.n_recs + 63) >> 6);
  if (items == 0) return hipSuccess;
  const uint32_t grid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, (items + 3) / 4), 16384);
  hipLaunchKernelGGL(derive_active_kernel, dim3(grid), dim3(256), 0, stream, m, lb, ub, live, n_nodes);
  return hipGetLastError();
}

hipError_t launch_expand_units(const uint32_t* rec_unit, uint32_t n_recs, uint32_t unit_words, const uint64_t* active_in, uint64_t* live,
                               uint32_t n_nodes, hipStream_t stream) {
  const uint64_t items = (uint64_t)n_nodes * ((n_recs + 63) >> 6);
  const uint32_t grid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, (items + 3) / 4), 8192);
  hipLaunchKernelGGL(expand_units_kernel, dim3(grid), dim3(256), 0, stream, rec_unit, n_recs, unit_words, active_in, live, n_nodes);
  return hipGetLastError();
}
hipError_t launch_contract_units(const uint32_t* unit_first, uint32_t n_units, uint32_t n_recs, const uint64_t* live, uint64_t* active_out,
                                 uint32_t n_nodes, hipStream_t stream) {
  const uint64_t items = (uint64_t)n_nodes * ((n_units + 63) >> 6);
  const uint32_t grid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, (items + 3) / 4), 8192);
  hipLaunchKernelGGL(contract_units_kernel, dim3(grid), dim3(256), 0, stream, unit_first, n_units, n_recs, live, active_out, n_nodes);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// On-device branching — Brancher<FirstSmallestVar, MiddleVal, BinarySplit>::enter (search/branching/brancher.rs:52-71)
// for every Unknown node of a propagated batch.  branch_scan_kernel (one block) turns the statuses into child slots
// in tree order; branch_kernel (one block per node) selects the variable with a block-wide (size, index) minimum,
// splits it, and streams the parent's bounds and `active` row into its two children.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) branch_scan_kernel(const uint8_t* __restrict__ status, uint32_t n_nodes, uint32_t* __restrict__ child_base,
                                                           uint32_t* __restrict__ counts) {
  __shared__ uint32_t tmp[40];
  __shared__ uint32_t total;
  uint32_t run = 0, n_true = 0, n_false = 0, n_other = 0;
  for (uint32_t base = 0; base < n_nodes; base += blockDim.x) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t st = i < n_nodes ? status[i] : 255u;
    const uint32_t unk = st == PCP_UNKNOWN ? 1u : 0u;
    n_true += st == PCP_TRUE;
    n_false += st == PCP_FALSE;
    n_other += (i < n_nodes) && st > PCP_UNKNOWN;  // e.g. PCP_STATUS_HULL: a node the engine refused, neither counted nor branched
    const uint32_t ex = block_exclusive_scan(unk, tmp, &total);
    if (i < n_nodes) child_base[i] = unk ? 2u * (run + ex) : 0xFFFFFFFFu;
    run += total;
    __syncthreads();
  }
  for (int o = 32; o > 0; o >>= 1) { n_true += __shfl_down(n_true, o); n_false += __shfl_down(n_false, o); n_other += __shfl_down(n_other, o); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&counts[1], n_true); atomicAdd(&counts[2], n_false); if (n_other) atomicAdd(&counts[4], n_other); }
  if (threadIdx.x == 0) { counts[0] = 2u * run; counts[3] = run; }
}

__global__ void __launch_bounds__(256) branch_kernel(uint32_t n_vars, uint32_t words, const int32_t* __restrict__ lb, const int32_t* __restrict__ ub,
                                                     const uint64_t* __restrict__ active, const uint32_t* __restrict__ child_base,
                                                     int32_t* __restrict__ child_lb, int32_t* __restrict__ child_ub, uint64_t* __restrict__ child_active,
                                                     uint32_t* __restrict__ child_dirty, const uint32_t* __restrict__ counts, uint32_t reverse) {
  const uint32_t node = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
  const uint32_t slot = child_base[node];
  if (slot == 0xFFFFFFFFu) return;  // not Unknown: nothing to branch on
  // reverse: child k of the batch goes to row n_children-1-k, so that a caller that appends the rows to a LIFO stack
  // pops the first node's left child first (left-first DFS) without reordering anything
  const uint32_t rowL = reverse ? counts[0] - 1 - slot : slot;
  const uint32_t rowR = reverse ? rowL - 1 : slot + 1;
  __shared__ unsigned long long best[4];
  const int32_t* plb = lb + (size_t)node * n_vars;
  const int32_t* pub = ub + (size_t)node * n_vars;
  // FirstSmallestVar: minimum of (size << 32 | index) over the variables of size > 1 (first index wins ties)
  unsigned long long key = ~0ull;
  for (uint32_t v = tid; v < n_vars; v += nth) {
    const unsigned long long size = (unsigned long long)((long long)pub[v] - (long long)plb[v] + 1);
    if (size > 1) key = min(key, (size << 32) | v);
  }
  for (int o = 32; o > 0; o >>= 1) key = min(key, (unsigned long long)__shfl_down(key, o));
  if ((tid & 63) == 0) best[tid >> 6] = key;
  __syncthreads();
  key = best[0];
  for (uint32_t w = 1; w < (nth >> 6); ++w) key = min(key, best[w]);
  // An Unknown node always has an unassigned variable; if it has none (the reference panics,
  // first_smallest_var.rs:36) the children are plain copies.
  const uint32_t var = key == ~0ull ? 0xFFFFFFFFu : (uint32_t)key;
  int32_t val = 0;
  if (var != 0xFFFFFFFFu) {
    const long long s = (long long)plb[var] + (long long)pub[var];
    val = (int32_t)(s / 2);  // MiddleVal: C++ `/` truncates toward zero like Rust's
  }
  // the children differ from the parent's row — a fixpoint, if the caller propagated it — in this one variable (pcp_device_batch.dirty_var)
  if (child_dirty && tid == 0) { child_dirty[rowL] = var; child_dirty[rowR] = var; }
  int32_t* l0 = child_lb + (size_t)rowL * n_vars;
  int32_t* u0 = child_ub + (size_t)rowL * n_vars;
  int32_t* l1 = child_lb + (size_t)rowR * n_vars;
  int32_t* u1 = child_ub + (size_t)rowR * n_vars;
  for (uint32_t v = tid; v < n_vars; v += nth) {
    const int32_t a = plb[v], b = pub[v];
    l0[v] = a;                                   // left:  x <= val
    u0[v] = (v == var) ? min(b, val) : b;
    l1[v] = (v == var) ? max(a, val + 1) : a;    // right: x > val
    u1[v] = b;
  }
  const uint64_t* pa = active + (size_t)node * words;
  uint64_t* a0 = child_active + (size_t)rowL * words;
  uint64_t* a1 = child_active + (size_t)rowR * words;
  for (uint32_t w = tid; w < words; w += nth) { const uint64_t x = pa[w]; a0[w] = x; a1[w] = x; }
}

// ------------------------------------------------------------------------------------------------
// Device-side DFS step (pcp_dfs_device): what OneSolution / AllSolution do with the node Propagation::enter has just propagated
// (search/engine/one_solution.rs:92-105, search/monitor.rs:19-68, search/stop_node.rs:47-62): count it; True -> a solution
// (the first one is kept), False -> a failure, both pop the node; Unknown -> Brancher<FirstSmallestVar, MiddleVal, BinarySplit>
// (brancher.rs:52-71): the right child `x > v` replaces the parent's row, the left child `x <= v` goes on top, so the next
// step takes the left child first like the reference's reversed push (one_solution.rs:46-51).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) dfs_step_kernel(uint32_t V, int32_t* __restrict__ lb, int32_t* __restrict__ ub, const uint8_t* __restrict__ status,
                                                       uint32_t capacity, uint32_t* __restrict__ sp_ptr, uint32_t* __restrict__ stop,
                                                       unsigned long long* __restrict__ counters, int32_t* __restrict__ first_solution,
                                                       uint32_t stop_on_solution, unsigned long long node_limit,
                                                       uint32_t* __restrict__ team_scratch, uint32_t team_words) {
  const uint32_t tid = threadIdx.x, nth = blockDim.x;
  // the team kernel's per-node scratch (tickets, merged masks, counters) is handed back zeroed for the next step's launch
  for (uint32_t i = tid; i < team_words; i += nth) team_scratch[i] = 0;
  const uint32_t sp = *sp_ptr;
  if (sp == 0 || *stop) return;
  const uint32_t node = sp - 1;
  const uint32_t st = status[node];
  int32_t* plb = lb + (size_t)node * V;
  int32_t* pub = ub + (size_t)node * V;
  __shared__ unsigned long long best[4];
  uint32_t new_sp = sp - 1;
  if (st == PCP_UNKNOWN) {
    unsigned long long key = ~0ull;
    for (uint32_t v = tid; v < V; v += nth) {
      const unsigned long long size = (unsigned long long)((long long)pub[v] - (long long)plb[v] + 1);
      if (size > 1) key = min(key, (size << 32) | v);
    }
    for (int o = 32; o > 0; o >>= 1) key = min(key, (unsigned long long)__shfl_down(key, o));
    if ((tid & 63) == 0) best[tid >> 6] = key;
    __syncthreads();
    key = best[0];
    for (uint32_t w = 1; w < (nth >> 6); ++w) key = min(key, best[w]);
    if (key != ~0ull && sp < capacity) {
      const uint32_t var = (uint32_t)key;
      const long long s2 = (long long)plb[var] + (long long)pub[var];
      const int32_t val = (int32_t)(s2 / 2);  // MiddleVal (middle_val.rs:25-27)
      int32_t* l0 = lb + (size_t)sp * V;      // the left child, on top
      int32_t* u0 = ub + (size_t)sp * V;
      __syncthreads();
      for (uint32_t v = tid; v < V; v += nth) {
        const int32_t a0 = plb[v], b0 = pub[v];
        l0[v] = a0;
        u0[v] = (v == var) ? min(b0, val) : b0;
        if (v == var) plb[v] = max(a0, val + 1);  // the parent's row becomes the right child
      }
      new_sp = sp + 1;
    } else if (key != ~0ull) {
      // stack overflow: terminal for this call.  The node stays on the stack, uncounted: a caller that resumes with a larger
      // stack propagates it again (idempotent) and counts it then.
      if (tid == 0) { counters[3] = 1; *stop = 1u; }
      return;
    } else {
      // Unknown, yet no variable with more than one value: the reference panics here (first_smallest_var.rs:36); error 3
      if (tid == 0) { counters[3] = 3; *stop = 1u; }
      return;
    }
  }
  if (tid == 0) {
    const unsigned long long n = ++counters[0];
    // StopNode hands EndOfSearch to the monitor for the node that reaches the limit (stop_node.rs:57-62, nesting of stop_node.rs:90-97):
    // it is a node, but neither a solution nor a failure
    const bool last = node_limit && n >= node_limit;
    if (st == PCP_TRUE && !last) {
      if (counters[1]++ == 0 && first_solution) counters[4] = 1;  // flag for the copy below
      if (stop_on_solution) *stop = 1u;
    } else if (st == PCP_FALSE && !last) {
      ++counters[2];
    } else if (st > PCP_UNKNOWN) {
      counters[3] = 2; *stop = 1u;  // a node the engine refused (PCP_STATUS_HULL)
    }
    if (node_limit && n >= node_limit) *stop = 1u;  // StopNode (stop_node.rs:57-62)
    *sp_ptr = new_sp;
  }
  if (st == PCP_TRUE && first_solution) {
    __syncthreads();
    if (counters[4] == 1) {
      for (uint32_t v = tid; v < V; v += nth) first_solution[v] = plb[v];
      __syncthreads();
      if (tid == 0) counters[4] = 2;
    }
  }
}
hipError_t launch_dfs_step(uint32_t n_vars, int32_t* lb, int32_t* ub, const uint8_t* status, uint32_t capacity, uint32_t* sp, uint32_t* stop,
                           unsigned long long* counters, int32_t* first_solution, uint32_t stop_on_solution, unsigned long long node_limit,
                           uint32_t* team_scratch, uint32_t team_words, hipStream_t stream) {
  hipLaunchKernelGGL(dfs_step_kernel, dim3(1), dim3(256), 0, stream, n_vars, lb, ub, status, capacity, sp, stop, counters, first_solution,
                     stop_on_solution, node_limit, team_scratch, team_words);
  return hipGetLastError();
}

// the scan alone (set mode branches with its own kernel, pcp_set.hip)
hipError_t launch_branch_scan(uint32_t n_nodes, const uint8_t* status, uint32_t* child_base, uint32_t* counts, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(counts, 0, 5 * sizeof(uint32_t), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(branch_scan_kernel, dim3(1), dim3(1024), 0, stream, status, n_nodes, child_base, counts);
  return hipGetLastError();
}

hipError_t launch_branch(uint32_t n_nodes, uint32_t n_vars, uint32_t words, const int32_t* lb, const int32_t* ub, const uint64_t* active,
                         const uint8_t* status, int32_t* child_lb, int32_t* child_ub, uint64_t* child_active, uint32_t* child_dirty, uint32_t* child_base,
                         uint32_t* counts, uint32_t reverse, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(counts, 0, 5 * sizeof(uint32_t), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(branch_scan_kernel, dim3(1), dim3(1024), 0, stream, status, n_nodes, child_base, counts);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  hipLaunchKernelGGL(branch_kernel, dim3(n_nodes), dim3(256), 0, stream, n_vars, words, lb, ub, active, child_base, child_lb, child_ub, child_active, child_dirty, counts, reverse);
  return hipGetLastError();
}

#endif  // PCP_TU == 0

template <int B, bool GLOBAL, bool COMPACT, bool PACKED = false>
static hipError_t launch_k(const LaunchArgs& a, const LaunchPlan& p, hipStream_t stream) {
  constexpr bool IMPLICIT = PCP_TU == 1;
  if (p.lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fixpoint_kernel<B, GLOBAL, COMPACT, PACKED, IMPLICIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((fixpoint_kernel<B, GLOBAL, COMPACT, PACKED, IMPLICIT>), dim3(p.grid), dim3(p.block), p.lds_bytes, stream, a);
  return hipGetLastError();
}
template <int B>
static hipError_t launch_b(const LaunchArgs& a, const LaunchPlan& p, hipStream_t stream) {
  return a.m.recs8 ? launch_k<B, false, true>(a, p, stream) : launch_k<B, false, false>(a, p, stream);
}

// nodes_per_block must be one of the instantiated tile sizes; global_dom selects the HBM-resident-domain variant,
// packed the 16-bit tiles (compact record stream only).  This translation unit serves a.live != nullptr (PCP_TU == 0:
// explicit `active` rows) or a.live == nullptr (PCP_TU == 1: implicit).
#if PCP_TU == 0
hipError_t launch_fixpoint_implicit(const LaunchArgs& a, const LaunchPlan& p, hipStream_t stream);
hipError_t launch_fixpoint(const LaunchArgs& a, const LaunchPlan& p, hipStream_t stream) {
  if (a.live == nullptr) return launch_fixpoint_implicit(a, p, stream);
#else
hipError_t launch_fixpoint_implicit(const LaunchArgs& a, const LaunchPlan& p, hipStream_t stream) {
  if (a.live != nullptr) return hipErrorInvalidValue;
#endif
  if (a.global_dom) return a.nodes_per_block == 1 ? launch_k<1, true, false>(a, p, stream) : hipErrorInvalidValue;
  if (a.packed) {
    if (!a.m.recs8) return hipErrorInvalidValue;
    switch (a.nodes_per_block) {
      case 8: return launch_k<8, false, true, true>(a, p, stream);
      case 16: return launch_k<16, false, true, true>(a, p, stream);
      case 32: return launch_k<32, false, true, true>(a, p, stream);
      default: return hipErrorInvalidValue;
    }
  }
  switch (a.nodes_per_block) {
    case 1: return launch_b<1>(a, p, stream);
    case 2: return launch_b<2>(a, p, stream);
    case 4: return launch_b<4>(a, p, stream);
    case 8: return launch_b<8>(a, p, stream);
    case 12: return launch_b<12>(a, p, stream);
    case 16: return launch_b<16>(a, p, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace pcp
