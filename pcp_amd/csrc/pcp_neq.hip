// pcp_neq.hip — the propagation fixpoint of all-XNeqY models over Interval<i32> domains, ASSIGNMENT-DRIVEN (gfx950).
//
// Same contract as fixpoint_kernel (pcp_kernels.hip): Store::consistency = prepare() + propagation_loop()
// (propagation/store.rs:125-164, 247-257) for a batch of implicit-active nodes (a node is its domains; liveness is derived,
// SURVEY.md A.4).  What differs is how the reference's "schedule every active propagator once" (init_scheduler,
// store.rs:144-149) is honoured.  XNeqY::propagate (propagators/cmp/x_neq_y.rs:82-93) does something only when one side is a
// singleton — and then only when that value is a BOUND of the other side (Interval::difference removes a value only at a
// bound, pinned by x_neq_y.rs:128 and term/constant.rs:165).  So of the P records of a node only those incident to an
// ASSIGNED variable can act in the initial sweep, and they are exactly the adjacency lists (IndexedDeps, reactors/
// indexed_deps.rs:56-121) of the assigned variables.  The kernel therefore has no sweep over the record table at all:
//   round 0  = the adjacency lists of every variable that is a singleton in the staged domains (plus the variables with a
//              Constant neighbour: a constant is a singleton that has no list of its own);
//   round r  = the adjacency lists of the variables changed in round r-1 (Store::react, store.rs:191-198).
// Every other (record, node) pair of the sweep is a proven no-op: both sides non-singletons => propagate() returns at once,
// and entailment is of no interest while the fixpoint runs (implicit-active nodes: nothing is unlinked).
// N-queens-1000 near the root: 1-2 assigned queens per node => ~3-6 thousand item tests instead of 1.5 million records.
//
// MI355X mapping
//  * one workgroup = a tile of B nodes; their domains sit in LDS NODE-MINOR (the B cells of a slot are adjacent: one
//    ds_read_b128 fetches a slot's cells of FOUR nodes), as 16-bit packed (-lb, ub) cells under a declared hull within +-16383,
//    else as int2 (-lb, ub); 16 bytes of padding after every 256 bytes of rows put the 16 lanes of a ds_read_b128 group, which
//    read 16 consecutive slots, on 16 distinct 4-bank groups; B * S * 4 bytes * 1.06: two tiles of 16 nodes of N-queens-1000 per CU;
//  * staging is the only HBM traffic: 16-byte row loads, four in flight per lane; a tile that narrows nothing writes
//    nothing back when the call is in place (Store::consistency(&mut vstore) works in place);
//  * a round is VARIABLE-major: the changed / assigned variables of all B nodes are compacted into one list of
//    (variable, node mask) entries; a list is walked in pieces of 4 x 64 entries, payload loads (8 B per entry: other slot,
//    offset) coalesced and issued one piece ahead; each piece decodes its entries ONCE and tests them against every node
//    of the mask: per (entry, node) a quarter of a ds_read_b128, two v_pk_add_u16 and a v_pk_min_u16 — the filter can act iff
//    lb(v) + t == ub(o) or ub(v) + t == lb(o) (fast_flag's condition, pcp_kernels.hip), i.e. iff a 16-bit half of
//    cell(v) + swap(cell(o)) + (-t, t) is zero; the running unsigned minimum over the nodes is tested once per entry and
//    only flagged entries run the full filter (eval_record: propagate() + is_subsumed() literally, LDS atomics);
//  * chains: x != y + c removes a value only at a bound, so a bound that runs into values forbidden by assigned neighbours moves one
//    value per round in the reference (one wake-up each).  A list walked for one or two nodes (the tail of a cascade) also marks,
//    for every ASSIGNED neighbour, the value it forbids in a 64-bit window above lb(v) / below ub(v) (LDS atomics); after the
//    round's barrier the bound jumps to the first unmarked value.  Every skipped value is one the filter would remove at the
//    bound, so the (unique) fixpoint is unchanged — the argument of the generic kernel's solo cascade;
//  * status (store.rs:250-256: True iff no subscription remains): at the fixpoint every record of an assigned variable has
//    been run after that variable's last change, so only records of UNASSIGNED variables can be open: one wavefront per
//    node walks them with early exit (an Unknown node shows an open record in its first 64 entries).
// Integer bound work: no MFMA.  Bound by HBM (staging) for shallow nodes, by VALU issue for deep ones.
#include <algorithm>
#include <type_traits>

#include "pcp_device.hpp"
#include "pcp_neq.h"

// Profiling builds only (tools/build_variant.py 0 <out.so> -DPCP_NEQ_PROFILE=1): the s_memtime phase timers ("neq_debug" bits 8 and 32)
// and the per-wavefront event trace ("neq_trace_ptr").  They hold a dozen registers across the whole kernel, which the product build
// needs for its row loads.
#ifndef PCP_NEQ_PROFILE
#define PCP_NEQ_PROFILE 0
#endif
#ifndef PCP_PUT_FAST
#define PCP_PUT_FAST 1
#endif
// cache policy of a full tile's row loads (raw_buffer_load aux: 1 = sc0, 2 = nt, 16 = sc1): the rows are read once — streaming them past the
// L1 leaves it to the list payloads every tile reads again
#ifndef PCP_STAGE_AUX
#define PCP_STAGE_AUX 0
#endif
// A/B switches of two round-5 experiments (tools/build_neq_variant.py): the constant compared by exclusive-or instead of a second packed
// add, and pieces shorter than 256 entries when a round walks few lists.
#ifndef PCP_NEQ_XOR
#define PCP_NEQ_XOR 1
#endif
#ifndef PCP_NEQ_PLEN
#define PCP_NEQ_PLEN 0
#endif

// profiling builds: one s_memtime stamp per wavefront and event (needs `tr_on`, `trbuf`, `lane`, `wv` in scope)
#define PCP_TR(k) do { if (tr_on && lane == 0) trbuf[wv * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)

namespace pcp {

namespace {

enum { N_FAIL = 0, N_OOB = 1, N_COUNT0 = 2, N_COUNT1 = 3, N_RMASK0 = 4, N_RMASK1 = 5, N_DIRTY = 6, N_WAVES = 7, N_UNK = 8, N_NARROW = 9,
       N_EV = 10, N_FULL = 12, N_STEPS = 14, N_WIN0 = 16, N_WIN1 = 17, N_MORE = 18, N_WORDS = 19,
       N_NID = 19 /* .. 34: the global node index of the tile's node b */,
       N_R0OVF = 38 /* round 0's direct list overflowed: the marks are scanned instead */,
       N_HINT = 35 /* bit b: node b came with a dirty-variable hint (pcp_device_batch.dirty_var): round 0 walks that variable's lists only */,
       N_VOTE = 39 /* .. 41: the votes of the lean round 0's passes, word (pass mod 3): 1 = a lane narrowed something, 2 = it assigned or emptied a variable,
                      4 = it left a flagged entry untested */ };
constexpr uint32_t kMaxRounds = 1u << 22;  // a round that runs narrows something, so a fixpoint has far fewer; the cap only makes a runaway impossible
constexpr uint32_t kCascadeThreads = 8;  // threads that narrowed something in a round before the next round's cover is priced at all
constexpr uint32_t kResweepMin = 32;  // marked variables of a node before the assigned-lists alternative is priced

constexpr uint32_t kR0Cap = 32;      // assigned variables of a tile that staging lists by itself (round 0's list built in passing); more: the marks are scanned
constexpr uint32_t kNextTileWord = 2 * 48 + 14;  // word of the status area (behind its two copies and the seven u64 sums) that holds a persistent workgroup's next tile
constexpr uint32_t kListCap = 256;   // entries of a round's list; more changed variables than that wait for the next round
constexpr uint32_t kWinCap = 1024;   // jump windows per round (fewer when LDS is short: NeqCarve::wcap)
constexpr uint32_t kNoWin = 0xFFFFu;
constexpr uint32_t kFastLists = 4;   // assigned variables of a full tile up to which round 0 takes the lean form (neq_fast_walk)
constexpr uint32_t kFastPer = 6;     // list entries per lane of that form (512 threads: lists of up to 3072 entries)

// A jump window: the values of variable v (node b) that assigned neighbours forbid, as bits above lb0 / below ub0.
struct __attribute__((aligned(16))) Win {
  uint32_t lo[2], hi[2];
  int lb0, ub0;
  uint32_t vb;  // v | b << 16
  uint32_t pad;
};
static_assert(sizeof(Win) == 32, "Win must be 32 bytes");

struct NeqCarve {
  size_t dom, chg, list, adj, win, misc, total;
  uint32_t sh;    // log2 of the rows between two paddings
  uint32_t wcap;  // jump windows that fit
  size_t vmk;     // 16-node tiles: per slot, the 16-bit mask of the tile's nodes in which staging found it assigned (round 0's list, built in passing); = win
};
// cell index of (slot, node 0): rows of B cells, four cells of padding after every 2^sh rows (2^sh rows of packed cells = 256 bytes)
__host__ __device__ inline uint32_t neq_row(uint32_t slot, uint32_t B, uint32_t sh) { return slot * B + ((slot >> sh) << 2); }
__host__ __device__ inline NeqCarve neq_carve(uint32_t S, uint32_t V, uint32_t B, bool packed, uint32_t wgs = 2) {
  auto up = [](size_t x) { return (x + 15) & ~(size_t)15; };
  NeqCarve c;
  c.sh = B >= 16 ? 2u : B >= 8 ? 3u : B >= 4 ? 4u : B >= 2 ? 5u : 6u;
  const size_t Wv = (S + 31) / 32;
  size_t o = 0;
  c.dom = o; o = up(o + ((size_t)neq_row(S, B, c.sh) + 4) * (packed ? 4 : 8));
  c.chg = o; o = up(o + (size_t)B * Wv * 4);
  c.list = o; o = up(o + (size_t)kListCap * 16);
  c.adj = o; o = up(o + ((size_t)V + 1) * 4);
  c.misc = o; o = up(o + (2 * 48 + 20) * 4);  // two copies: a persistent workgroup's next tile clears ITS copy while stragglers still read the last tile's; then seven u64 sums over the workgroup's tiles
  // the windows take what is left of the CU's LDS divided by the workgroups that are to share it (two by default; one when the
  // tile needs more than its share)
  // (256 bytes short of an even share: __syncthreads_or and friends take a few bytes of static LDS on top of the dynamic carve)
  const size_t share = (((size_t)160 * 1024 / (wgs ? wgs : 2)) & ~(size_t)15) - 256;
  const size_t budget = o <= share ? share : 160 * 1024;
  c.wcap = (uint32_t)std::min<size_t>(kWinCap, o < budget ? (budget - o) / sizeof(Win) : 0);
  c.win = o; o = up(o + (size_t)c.wcap * sizeof(Win));
  c.vmk = c.win;  // (the masks live in the window area: staging and round 0 use the masks, only later rounds use windows)
  c.total = o;
  return c;
}

// Sum over the 64 lanes of a wavefront, in every lane's... lane 63, handed out wave-uniform: six DPP adds (row_shr 1, 2, 4, 8 inside the rows
// of 16 lanes, row_bcast 15 and 31 across them) and one readlane — VALU only.  __shfl_down is a ds_bpermute, i.e. an LDS-pipeline
// round trip per step, and this kernel's LDS pipeline is where a frontier tile's serial steps queue up.
// A kernel argument read from the kernel-argument segment at the point of use (the kernel's one parameter, NeqArgs, starts the segment).
template <class T>
__device__ __forceinline__ T neq_karg(size_t off) {
  typedef __attribute__((address_space(4))) const char* KP;
  typedef __attribute__((address_space(4))) const volatile T* TP;
  KP kp = (KP)__builtin_amdgcn_kernarg_segment_ptr();
  return *(TP)(kp + off);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8  -> lane 15 of a row holds the row's sum
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the sum
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
// the same for counters that may be large: the halves are summed separately, so nothing can wrap
__device__ __forceinline__ unsigned long long wave_sum64(uint32_t x) {
  return (unsigned long long)wave_sum(x & 0xffffu) + ((unsigned long long)wave_sum(x >> 16) << 16);
}

__device__ __forceinline__ bool zero_half(uint32_t u) { return (u & 0xffffu) == 0u || (u >> 16) == 0u; }

// both 16-bit halves added separately (no carry from the low half into the high one)
__device__ __forceinline__ uint32_t pk_add16(uint32_t x, uint32_t y) {
  uint32_t r;
  asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}

// (cell(v) + swap(cell(o))) ^ (t, -t): a half is zero iff lb(v) + t == ub(o) (low) or ub(v) + t == lb(o) (high).  The constant is
// compared by an exclusive-or, not added: v_pk_add_u16 issues at half the rate of a 32-bit VALU operation on gfx950 (4.6 against 2.5
// cycles per wave instruction with four wavefronts per SIMD, tools/micro/box_probe.hip), v_xor_b32 at full rate.
__device__ __forceinline__ uint32_t neq_terms16(uint32_t cv, uint32_t co, uint32_t k) {
  uint32_t u;
#if PCP_NEQ_XOR
  asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(u) : "v"(cv), "v"(co));
  return u ^ k;
#else
  asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      "v_pk_add_u16 %0, %0, %3"
      : "=&v"(u) : "v"(cv), "v"(co), "v"(k));
  return u;
#endif
}
// The cell an entry's other side must MATCH in one half, when the walked variable's cell cv = (-lb(v), ub(v)) is the same in every node
// tested: T = (-(ub(v) + t), lb(v) + t) = -((t, -t) + swap(cv)) per half: cell(o).lo == T.lo <=> lb(o) == ub(v) + t, cell(o).hi == T.hi <=>
// ub(o) == lb(v) + t.  Computed once per ENTRY; the (entry, node) test is then one exclusive-or and one packed minimum.
__device__ __forceinline__ uint32_t neq_target16(uint32_t cv, uint32_t k) {
  uint32_t u;
  asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      "v_pk_sub_u16 %0, 0, %0"
      : "=&v"(u) : "v"(k), "v"(cv));
  return u;
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t x, uint32_t y) {
  uint32_t r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
// (t, -t) as two int16 halves — what the halves of cell(v) + swap(cell(o)) are compared with; |t| beyond the packed range can never meet a
// sum of two packed bounds: clamp (pack_c0's argument)
__device__ __forceinline__ uint32_t pack_mt(int t) {
  const int tc = max(-32767, min(32767, t));
#if PCP_NEQ_XOR
  return ((uint32_t)tc & 0xffffu) | ((uint32_t)(-tc) << 16);
#else
  return ((uint32_t)(-tc) & 0xffffu) | ((uint32_t)tc << 16);  // (added, not compared: (-t, t))
#endif
}

template <bool PACKED> struct NeqCell { using type = int2; };
template <> struct NeqCell<true> { using type = uint32_t; };

// Adjacency payloads: 8 bytes per entry (ModelDev::adjp: other | kind << 28 | is_y << 31, d), or — when every slot fits 15 bits and
// every offset 16 — 4 bytes: other | is_y << 15 | t << 16 with t = d for the record's y side, -d for its x side (half the stream).
__device__ __forceinline__ uint32_t pay_other(const uint2 q) { return q.x & kSlotMask; }
__device__ __forceinline__ bool pay_is_y(const uint2 q) { return (q.x >> 31) != 0; }
__device__ __forceinline__ int pay_t(const uint2 q) { const int d = (int32_t)q.y; return (q.x >> 31) ? d : -d; }
__device__ __forceinline__ uint32_t pay_other(const uint32_t q) { return q & 0x7fffu; }
__device__ __forceinline__ bool pay_is_y(const uint32_t q) { return ((q >> 15) & 1u) != 0; }
__device__ __forceinline__ int pay_t(const uint32_t q) { return (int32_t)q >> 16; }

template <bool PACKED>
__device__ __forceinline__ int2 cell_bounds(const typename NeqCell<PACKED>::type c) {  // (lb, ub)
  if constexpr (PACKED) return unpack16(c);
  else return make_int2(-c.x, c.y);
}

// variable::Store::update (variable/store.rs:151-166) on the tile's cells: LdsDom16 / LdsDom (pcp_device.hpp) with the padded
// node-minor index.  `dom` points at node b's cell of slot 0.  A narrowing marks the variable changed and the node dirty.
struct TileDom16 {
  uint32_t* dom;
  uint32_t B, sh;
  uint32_t* chg;
  uint32_t* misc;
  uint32_t fbit;
  Ctr* c;
  __device__ __forceinline__ bool any_sums() const { return false; }
  __device__ __forceinline__ const int32_t* mul_offsets() const { return nullptr; }
  __device__ __forceinline__ uint32_t* cell(uint32_t v) const { return dom + neq_row(v, B, sh); }
  __device__ __forceinline__ int2 load(uint32_t v) const { return unpack16(*cell(v)); }
  __device__ __forceinline__ void mark(uint32_t v) const { atomicOr(&chg[v >> 5], 1u << (v & 31)); atomicOr(&misc[N_DIRTY], fbit); }
  __device__ __forceinline__ void set_fail() const { atomicOr(&misc[N_FAIL], fbit); }
  // A narrowing that need not wake the variable's propagators (a window jump that ended inside its window: see neq_apply_jumps): the node is
  // dirty, the variable is marked only when it became assigned or empty.
  __device__ __forceinline__ void touch(uint32_t v, bool wake) const { if (wake) mark(v); else atomicOr(&misc[N_DIRTY], fbit); }
  __device__ __forceinline__ void raise_lb(uint32_t v, int nlb, bool quiet = false) const {
    uint32_t* p = cell(v);
    uint32_t old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (;;) {
      const int2 d = unpack16(old);
      if (nlb <= d.x) return;  // somebody else got there first
      const uint32_t prev = atomicCAS(p, old, pack16(min(nlb, d.y + 1), d.y));
      if (prev == old) { ++c->narrow; touch(v, !quiet || nlb >= d.y); if (nlb > d.y) set_fail(); return; }
      old = prev;
    }
  }
  __device__ __forceinline__ void lower_ub(uint32_t v, int nub, bool quiet = false) const {
    uint32_t* p = cell(v);
    uint32_t old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (;;) {
      const int2 d = unpack16(old);
      if (nub >= d.y) return;
      const uint32_t prev = atomicCAS(p, old, pack16(d.x, max(nub, d.x - 1)));
      if (prev == old) { ++c->narrow; touch(v, !quiet || nub <= d.x); if (nub < d.x) set_fail(); return; }
      old = prev;
    }
  }
};
struct TileDom32 {
  int2* dom;
  uint32_t B, sh;
  uint32_t* chg;
  uint32_t* misc;
  uint32_t fbit;
  Ctr* c;
  __device__ __forceinline__ bool any_sums() const { return false; }
  __device__ __forceinline__ const int32_t* mul_offsets() const { return nullptr; }
  __device__ __forceinline__ int2* cell(uint32_t v) const { return dom + neq_row(v, B, sh); }
  __device__ __forceinline__ int2 load(uint32_t v) const { const int2 d = *cell(v); return make_int2(-d.x, d.y); }
  __device__ __forceinline__ void mark(uint32_t v) const { atomicOr(&chg[v >> 5], 1u << (v & 31)); atomicOr(&misc[N_DIRTY], fbit); }
  __device__ __forceinline__ void set_fail() const { atomicOr(&misc[N_FAIL], fbit); }
  __device__ __forceinline__ void touch(uint32_t v, bool wake) const { if (wake) mark(v); else atomicOr(&misc[N_DIRTY], fbit); }
  __device__ __forceinline__ void raise_lb(uint32_t v, int nlb, bool quiet = false) const {
    int2* p = cell(v);
    const int old = atomicMin(&p->x, -nlb);
    if (old > -nlb) {
      const int ubv = __hip_atomic_load(&p->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      ++c->narrow; touch(v, !quiet || nlb >= ubv); if (nlb > ubv) set_fail();
    }
  }
  __device__ __forceinline__ void lower_ub(uint32_t v, int nub, bool quiet = false) const {
    int2* p = cell(v);
    const int old = atomicMin(&p->y, nub);
    if (old > nub) {
      const int lbv = -__hip_atomic_load(&p->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      ++c->narrow; touch(v, !quiet || nub <= lbv); if (lbv > nub) set_fail();
    }
  }
};
// TileDom16 for the lean round 0 (neq_fast_walk): a narrowing wakes its variable — marks it, raises *woke — only when it assigned or emptied it.
// (Why the others need no wake-up: see neq_fast_walk.)
struct TileDom16Q : TileDom16 {
  bool* woke;
  __device__ __forceinline__ void raise_lb(uint32_t v, int nlb) const {
    uint32_t* p = cell(v);
    uint32_t old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (;;) {
      const int2 d = unpack16(old);
      if (nlb <= d.x) return;
      const uint32_t prev = atomicCAS(p, old, pack16(min(nlb, d.y + 1), d.y));
      if (prev == old) { ++c->narrow; const bool wake = nlb >= d.y; touch(v, wake); *woke |= wake; if (nlb > d.y) set_fail(); return; }
      old = prev;
    }
  }
  __device__ __forceinline__ void lower_ub(uint32_t v, int nub) const {
    uint32_t* p = cell(v);
    uint32_t old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (;;) {
      const int2 d = unpack16(old);
      if (nub >= d.y) return;
      const uint32_t prev = atomicCAS(p, old, pack16(d.x, max(nub, d.x - 1)));
      if (prev == old) { ++c->narrow; const bool wake = nub <= d.x; touch(v, wake); *woke |= wake; if (nub < d.x) set_fail(); return; }
      old = prev;
    }
  }
};
template <bool PACKED> struct TileDomOf { using type = TileDom32; };
template <> struct TileDomOf<true> { using type = TileDom16; };

}  // namespace

// The cheaper of two covers for one node's next round (see (a0) in the kernel): ONE wavefront, node b.  Kept out of line: inlined,
// its loops cost the kernel 14 VGPRs and 9 % of the headline launch although frontier tiles hardly ever reach it.
template <bool PACKED>
__device__ __noinline__ void resweep_marks(const typename NeqCell<PACKED>::type* dom, uint32_t* row, const uint32_t* adjo,
                                           const uint32_t* seed_always, uint32_t V, uint32_t Wv, uint32_t B, uint32_t sh, uint32_t b,
                                           uint32_t lane) {
  uint32_t c = 0;
  for (uint32_t w = lane; w < Wv; w += 64) c += __popc(row[w]);
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if (c < kResweepMin) return;
  uint32_t cdeg = 0, adeg = 0;
  for (uint32_t v = lane; v < V; v += 64) {
    const uint32_t dg = adjo[v + 1] - adjo[v];
    const int2 d = cell_bounds<PACKED>(dom[neq_row(v, B, sh) + b]);
    if ((row[v >> 5] >> (v & 31u)) & 1u) cdeg += dg;
    if (d.x == d.y) adeg += dg;
  }
  for (int o = 32; o > 0; o >>= 1) { cdeg += __shfl_xor(cdeg, o); adeg += __shfl_xor(adeg, o); }
  if (adeg >= cdeg) return;
  for (uint32_t base = 0; base < V; base += 64) {
    const uint32_t v = base + lane;
    bool single = false;
    if (v < V) { const int2 d = cell_bounds<PACKED>(dom[neq_row(v, B, sh) + b]); single = d.x == d.y; }
    const uint64_t bal = __ballot(single);
    const uint32_t w = (base >> 5) + lane;
    if (lane < 2 && w < Wv) {
      const uint32_t keep = seed_always ? (row[w] & seed_always[w]) : 0u;
      row[w] = (uint32_t)(bal >> (32u * lane)) | keep;
    }
  }
}

// What the phases of a tile share: its LDS arrays and its geometry.  (`tid`, `lane` are refreshed per tile by the kernel: see its loop.)
template <bool PACKED>
struct NeqTile {
  using TDom = typename TileDomOf<PACKED>::type;
  typename NeqCell<PACKED>::type* dom;
  uint32_t* chg;
  uint32_t* misc;
  const uint32_t* adjo;
  uint4* list;
  Win* win;
  uint32_t V, Wv, B, sh, nb;
  uint32_t tid, lane, wv, nwv, nth;
  __device__ __forceinline__ uint32_t rowof(uint32_t slot) const { return neq_row(slot, B, sh); }  // index of node 0's cell of a slot
  __device__ __forceinline__ TDom dom_of(uint32_t b, Ctr* c) const { return TDom{dom + b, B, sh, chg + (size_t)b * Wv, misc, 1u << b, c}; }  // node b's store
};

// ---- status: is any record NOT entailed under the final domains? (store.rs:250-256) ------------------------------
// Records of two assigned variables are entailed at a fixpoint that did not fail (two different values: disjoint), so only the
// lists of unassigned variables can hold an open record; x != y + d is entailed iff the intervals are disjoint
// (x_neq_y.rs:71-73 via x_eq_y.rs:87-93).
// Two nodes per wavefront at a time, one per 32-lane half: a node's scan is a chain of dependent LDS and memory reads (cells -> the
// list's offsets -> its payload -> the other sides' cells), and a tile of sixteen nodes on eight wavefronts used to run two such
// chains one after the other in every wavefront.  Sets bit b of misc[N_UNK] for a node with an open record.
template <bool PACKED, bool DFS, class Tile, class Pay>
__device__ __forceinline__ void neq_status_scan(const Tile& tl, const Pay* pay, bool skip, const uint32_t only = 0xFFFFFFFFu) {
const uint32_t inert = tl.misc[N_FAIL] | tl.misc[N_OOB] | ~only;  // (`only`: the nodes to scan — the others keep the bit they have)
  for (uint32_t b0 = tl.wv; b0 < tl.nb; b0 += 2 * tl.nwv) {
    // (a wavefront with one node left gives it all 64 lanes: the search loop's single node, the odd node of a ragged tile)
    const bool pair = !DFS && b0 + tl.nwv < tl.nb;                            // wave-uniform (the search loop has one node: folded away)
    const uint32_t hw = pair ? 32u : 64u, hl = tl.lane & (hw - 1u), hb = pair ? tl.lane >> 5 : 0u;
    const uint32_t b = b0 + hb * tl.nwv;                                   // this half's node
    bool done = ((inert >> b) & 1u) || skip;                  // (uniform within a half)
    bool open = false;
    auto cellb = [&](uint32_t slot) { return cell_bounds<PACKED>(tl.dom[tl.rowof(slot) + b]); };
    auto mine = [&](unsigned long long bal) { return pair ? (unsigned long long)(uint32_t)(bal >> (32u * hb)) : bal; };
    for (uint32_t base = 0; base < tl.V; base += hw) {
      if (!__ballot(!done)) break;
      const uint32_t vv = base + hl;
      bool wide = false;
      if (!done && vv < tl.V) { const int2 d = cellb(vv); wide = d.x < d.y; }
      unsigned long long cand = mine(__ballot(wide));                   // this half's unassigned variables among these
      for (;;) {
        const bool has = !done && cand != 0ull;
        if (!__ballot(has)) break;
        const uint32_t u = has ? base + (uint32_t)__builtin_ctzll(cand) : 0u;
        cand &= cand - 1ull;
        int2 Ud = make_int2(0, 0);
        uint32_t o0 = 0, deg = 0;
        if (has) { Ud = cellb(u); o0 = tl.adjo[u]; deg = tl.adjo[u + 1] - o0; }
        for (uint32_t k = 0;; k += hw) {
          const bool go = has && !open && k < deg;
          if (!__ballot(go)) break;
          bool op = false;
          if (go && k + hl < deg) {
            const Pay q = pay[o0 + k + hl];
            const int t = pay_t(q);
            const int2 O = cellb(pay_other(q));
            op = !((Ud.x + t > O.y) || (Ud.y + t < O.x));  // not disjoint
          }
          if (mine(__ballot(op))) open = true;
        }
        if (open) done = true;
      }
    }
    if (open && hl == 0) atomicOr(&tl.misc[N_UNK], 1u << b);
  }
}

// The usual outcome of the status scan, split in two so that its one memory round trip is in flight while the tile's list walk runs (the lean
// round 0, neq_fast_walk).  Full tiles on eight wavefronts only: node wv in a wavefront's lanes 0-31, node wv + 8 in its lanes 32-63.
// issue: the node's first unassigned variable u among its first 32, u's cell and the first 32 entries of u's list (one load per lane);
// finish: is one of those records open — not entailed: the intervals meet (x_neq_y.rs:71-73 via x_eq_y.rs:87-93)?  Then the node is Unknown
// (store.rs:250-256) and its bit of misc[N_UNK] is set.  Anything else — no unassigned variable among the first 32, no open record among the
// first 32 — is left to the full scan (neq_status_scan), which finish() asks for by returning true (wave-uniform).
struct StatusPre { uint32_t q; int lbu, ubu; uint32_t flags; };  // flags: 1 = q holds an entry, 4 = the node is inert (failed or refused); bits 8..: the variable u
template <class Tile>
__device__ __forceinline__ StatusPre neq_status_issue(const Tile& tl, const uint32_t* __restrict__ pay) {
  const uint32_t hb = tl.lane >> 5, hl = tl.lane & 31u, b = tl.wv + hb * tl.nwv;
  const bool live = !(((tl.misc[N_FAIL] | tl.misc[N_OOB]) >> b) & 1u);
  StatusPre r{0u, 0, 0, live ? 0u : 4u};
  bool wide = false;
  if (live && hl < tl.V) { const int2 d = unpack16(tl.dom[tl.rowof(hl) + b]); wide = d.x < d.y; }
  const uint32_t cand = (uint32_t)(__ballot(wide) >> (32u * hb));
  if (cand) {
    const uint32_t u = (uint32_t)__builtin_ctz(cand);
    const int2 Ud = unpack16(tl.dom[tl.rowof(u) + b]);
    const uint32_t o0 = tl.adjo[u], deg = tl.adjo[u + 1] - o0;
    r.lbu = Ud.x; r.ubu = Ud.y;
    r.flags |= u << 8;
    if (hl < deg) { r.q = pay[o0 + hl]; r.flags |= 1u; }
  }
  return r;
}
template <class Tile>
// `refresh`: u's cell is read again (a later pass of the lean round 0: the node narrowed since issue(); u is still its first unassigned variable —
// those passes assign nothing — and the entries are u's, whatever its bounds).
__device__ __forceinline__ bool neq_status_finish(const Tile& tl, const StatusPre& r, const uint32_t only = 0xFFFFFFFFu, const bool refresh = false) {
  const uint32_t hb = tl.lane >> 5, b = tl.wv + hb * tl.nwv;
  if (!__ballot((only >> b) & 1u)) return false;  // (none of this wavefront's two nodes is asked for)
  bool op = false;
  if (r.flags & 1u) {
    int lbu = r.lbu, ubu = r.ubu;
    if (refresh) { const int2 Ud = unpack16(tl.dom[tl.rowof(r.flags >> 8) + b]); lbu = Ud.x; ubu = Ud.y; }
    const int t = pay_t(r.q);
    const int2 O = unpack16(tl.dom[tl.rowof(pay_other(r.q)) + b]);
    op = !((lbu + t > O.y) || (ubu + t < O.x));  // not disjoint
  }
  const bool open = (uint32_t)(__ballot(op) >> (32u * hb)) != 0u, mine = ((only >> b) & 1u) != 0u;
  if (open && mine && (tl.lane & 31u) == 0u) atomicOr(&tl.misc[N_UNK], 1u << b);
  return __ballot(mine && !(r.flags & 4u) && !open) != 0ull;
}

// ---- write back: the rows of the nodes that changed (every node when the call is not in place).  A refused node's outputs are left
// alone.  An empty cell found on the way out fails its node (misc[N_FAIL]).  Returns the mask of the nodes written (workgroup-uniform).
template <bool PACKED, class Tile>
__device__ __forceinline__ uint32_t neq_write_back(const Tile& tl, const int32_t* lb_in, const int32_t* ub_in, int32_t* lb_out, int32_t* ub_out, const bool cells,
                                                   const bool force_all = false) {
  const bool in_place = !force_all && lb_in == lb_out && (cells || ub_in == ub_out);
  const uint32_t all_nodes = tl.nb >= 32 ? 0xFFFFFFFFu : ((1u << tl.nb) - 1u);
  const uint32_t dirty = __builtin_amdgcn_readfirstlane(tl.misc[N_DIRTY]), refused = __builtin_amdgcn_readfirstlane(tl.misc[N_OOB]);
  uint32_t badm = 0;
  const bool vec_out = (tl.V & 3u) == 0 && (((size_t)lb_out | (size_t)ub_out) & 15u) == 0;
  // a refused node's outputs are left alone; in place, the rows of an unchanged node already hold the result in HBM: a frontier
  // tile writes nothing and does not even look at its sixteen nodes one by one
  const uint32_t wb_need = (in_place ? dirty : all_nodes) & ~refused & all_nodes;  // (workgroup-uniform)
  for (uint32_t need = wb_need; need; need &= need - 1u) {
    const uint32_t b = (uint32_t)__builtin_ctz(need);
    auto cellb = [&](uint32_t slot) { return cell_bounds<PACKED>(tl.dom[tl.rowof(slot) + b]); };
    int32_t* lbp = lb_out + (size_t)tl.misc[N_NID + b] * tl.V;
    int32_t* ubp = ub_out + (size_t)tl.misc[N_NID + b] * tl.V;
    bool bad = false;
    if constexpr (PACKED) if (cells) {  // rows of cells (PCP_CELLS_PACKED16): the LDS column of the node, as it is
      uint32_t* const cp = reinterpret_cast<uint32_t*>(lb_out) + (size_t)tl.misc[N_NID + b] * tl.V;
      auto emp = [](uint32_t c) { return ((c + (c >> 16)) & 0x8000u) != 0u; };
      if ((tl.V & 3u) == 0 && ((size_t)lb_out & 15u) == 0) {
        for (uint32_t q = tl.tid; q < (tl.V >> 2); q += tl.nth) {
          const uint32_t r0 = tl.rowof(4 * q) + b;  // (four slots of one quad: no padding between their rows)
          const uint32_t c0 = tl.dom[r0], c1 = tl.dom[r0 + tl.B], c2 = tl.dom[r0 + 2 * tl.B], c3 = tl.dom[r0 + 3 * tl.B];
          bad |= emp(c0) | emp(c1) | emp(c2) | emp(c3);
          reinterpret_cast<uint4*>(cp)[q] = make_uint4(c0, c1, c2, c3);
        }
      } else {
        for (uint32_t v = tl.tid; v < tl.V; v += tl.nth) { const uint32_t c = tl.dom[tl.rowof(v) + b]; bad |= emp(c); cp[v] = c; }
      }
      if (bad) badm |= 1u << b;
      continue;
    }
    if (vec_out) {
      for (uint32_t q = tl.tid; q < (tl.V >> 2); q += tl.nth) {
        int l[4], u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int2 d = cellb(4 * q + i); l[i] = d.x; u[i] = d.y; bad |= d.x > d.y; }
        reinterpret_cast<int4*>(lbp)[q] = make_int4(l[0], l[1], l[2], l[3]);
        reinterpret_cast<int4*>(ubp)[q] = make_int4(u[0], u[1], u[2], u[3]);
      }
    } else {
      for (uint32_t v = tl.tid; v < tl.V; v += tl.nth) { const int2 d = cellb(v); bad |= d.x > d.y; lbp[v] = d.x; ubp[v] = d.y; }
    }
    if (bad) badm |= 1u << b;
  }
  if (badm) atomicOr(&tl.misc[N_FAIL], badm);
  return wb_need;
}

// ---- the jumps of a round: each window's bound moves to the first value no assigned neighbour forbids (the windows were filled by the
// list walk of one- and two-node masks: see the kernel's header).  Every skipped value is one the filter would remove at the bound.
// QUIET jumps (round 5).  The walk that filled a window saw EVERY entry of the variable's list, so a bound that jumped to a value inside the
// window is known not to be forbidden by any neighbour that was assigned when its cell was read.  Who could still act on the new bound?  Only
// a neighbour assigned since — and that neighbour is a changed variable whose OWN list is walked next round and tests this bound from its
// side — or, if the jump assigned the variable (or emptied it), the variable itself.  So a jump that ends inside its window and leaves the
// variable with more than one value does not mark it changed: the reference would pop its 3V propagators once more and find every one a no-op
// (x_neq_y.rs:82-93: no side is a singleton whose value is a bound of the other).  Option bit neq_debug 65536 switches the rule off (A/B, tests).
template <bool PACKED, class Tile>
__device__ __forceinline__ void neq_apply_jumps(const Tile& tl, uint32_t nwin, Ctr& ctr, const bool quiet) {
  for (uint32_t wi = tl.tid; wi < nwin; wi += tl.nth) {
    const Win w = tl.win[wi];
    const uint32_t v = w.vb & 0xffffu, b = w.vb >> 16;
    if ((tl.misc[N_FAIL] >> b) & 1u) continue;
    const auto dm = tl.dom_of(b, &ctr);
    const int2 d = dm.load(v);
    if (d.x > d.y) continue;
    const unsigned long long Lm = ((unsigned long long)w.lo[1] << 32) | w.lo[0], Hm = ((unsigned long long)w.hi[1] << 32) | w.hi[0];
    {
      const uint32_t off = (uint32_t)(d.x - w.lb0);  // the bound may have moved during the walk
      if (off < 64u) {
        const unsigned long long m = Lm | ((1ull << off) - 1ull);
        const int nl = w.lb0 + (m == ~0ull ? 64 : (int)__builtin_ctzll(~m));
        if (nl > d.x) dm.raise_lb(v, nl, quiet && m != ~0ull);
      }
    }
    {
      const uint32_t off = (uint32_t)(w.ub0 - d.y);
      if (off < 64u) {
        const unsigned long long m = Hm | ((1ull << off) - 1ull);
        const int nu = w.ub0 - (m == ~0ull ? 64 : (int)__builtin_ctzll(~m));
        if (nu < d.y) dm.lower_ub(v, nu, quiet && m != ~0ull);
      }
    }
  }
}

// The in-kernel search loop's registers (replicated in every thread of the tree's workgroup).
struct NeqDfsRegs {
  uint32_t sp, stop, resume_var, hint, err;
  unsigned long long nodes, sols, fail;
  uint32_t stale;  // the node on top of the stack is in LDS only (a left child whose row was not pushed): its write-back writes the whole row
};

// ---- the search step on the node the tile has just propagated: OneSolution / AllSolution over Propagation<Brancher<FirstSmallestVar,
// MiddleVal, BinarySplit>> (one_solution.rs:92-105, brancher.rs:52-71) under StopNode (stop_node.rs:47-62).  `a` points at the node's
// row (lb_out / ub_out: row sp - 1 of the tree's stack).  Count it; failed / solution: pop; Unknown: the right child x > v over the
// parent's row, the left child x <= v on top and — its domains being in LDS already — marked as the node to continue with.
template <bool PACKED, class Tile>
__device__ __forceinline__ void neq_dfs_step(const Tile& tl, const NeqArgs& a, NeqDfsRegs& r, const bool last_step) {
  const bool failed = (tl.misc[N_FAIL] & 1u) != 0, refused = (tl.misc[N_OOB] & 1u) != 0, open = (tl.misc[N_UNK] & 1u) != 0;
  r.resume_var = 0xFFFFFFFFu;
  uint32_t new_sp = r.sp - 1;
  // StopNode (stop_node.rs:57-62) replaces the status of the node that reaches the limit by EndOfSearch BEFORE the monitor sees it
  // (Monitor<Statistics, StopNode<..>>, stop_node.rs:90-97): that node is counted as a node, never as a solution or a failure.
  const bool last = a.dfs.node_limit && r.nodes + 1 >= a.dfs.node_limit;
  if (refused) {
    r.err = 2; r.stop = 1;  // a node the engine refused (PCP_STATUS_HULL)
    ++r.nodes;
  } else if (failed) {
    ++r.nodes;
    if (!last) ++r.fail;
  } else if (!open) {  // True: a solution (monitor.rs:19-68); the first one is kept
    ++r.nodes;
    if (!last) {
      if (r.sols == 0 && a.dfs.first_solution)
        for (uint32_t v = tl.tid; v < tl.V; v += tl.nth) a.dfs.first_solution[v] = cell_bounds<PACKED>(tl.dom[tl.rowof(v)]).x;
      ++r.sols;
      if (a.dfs.stop_on_solution) r.stop = 1;
    }
  } else {
    // Brancher<FirstSmallestVar, MiddleVal, BinarySplit>::enter (brancher.rs:52-71): the first variable of minimal size > 1
    unsigned long long key = ~0ull;
    for (uint32_t v = tl.tid; v < tl.V; v += tl.nth) {
      const int2 d = cell_bounds<PACKED>(tl.dom[tl.rowof(v)]);
      const unsigned long long size = (unsigned long long)((long long)d.y - (long long)d.x + 1);
      if (size > 1) key = min(key, (size << 32) | v);
    }
    for (int o = 32; o > 0; o >>= 1) key = min(key, (unsigned long long)__shfl_down(key, o));
    unsigned long long* best = reinterpret_cast<unsigned long long*>(tl.list);  // (the round list is idle here)
    if (tl.lane == 0) best[tl.wv] = key;
    __syncthreads();
    key = best[0];
    for (uint32_t w = 1; w < tl.nwv; ++w) key = min(key, best[w]);
    __syncthreads();
    if (key == ~0ull) {
      r.err = 3; r.stop = 1;  // Unknown, yet nothing to branch on: the reference panics (first_smallest_var.rs:36)
      new_sp = r.sp;
    } else if (r.sp >= a.dfs.capacity) {
      r.err = 1; r.stop = 1;  // stack overflow: the node stays on the stack, uncounted
      new_sp = r.sp;
    } else {
      ++r.nodes;
      const uint32_t var = (uint32_t)key;
      const int2 d = cell_bounds<PACKED>(tl.dom[tl.rowof(var)]);
      const int val = (int)(((long long)d.x + (long long)d.y) / 2);  // MiddleVal (middle_val.rs:25-27: `/` truncates toward zero)
      // the right child x > val takes the parent's row (which holds the fixpoint: written back above if it changed)
      if (tl.tid == 0) a.lb_out[var] = max(d.x, val + 1);
      if (tl.tid == 0 && a.dfs.dirty) { a.dfs.dirty[r.sp - 1] = var; a.dfs.dirty[r.sp] = var; }  // both children differ from this fixpoint in `var`
      // the left child x <= val: one bound of one LDS cell, and its row on top of the stack
      if (tl.tid == 0) {
        if constexpr (PACKED) tl.dom[tl.rowof(var)] = pack16(d.x, min(d.y, val)); else tl.dom[tl.rowof(var)] = make_int2(-d.x, min(d.y, val));
      }
      __syncthreads();
      // The left child is the next node of this very loop and is propagated from LDS; the write-back of ITS fixpoint puts its row on the
      // stack (in full: r.stale).  Its unpropagated row is written here only when the loop ends with it on top — the launch's last step,
      // or the node limit reached —, so that the stack in HBM is complete whenever anyone else can look at it (between launches: the
      // steal, the refill across ranks, the host).  Saves one 8 n_vars-byte row of scalar stores per node.
      if (last_step || (a.dfs.node_limit && r.nodes >= a.dfs.node_limit)) {
        int32_t* l0 = a.lb_out + tl.V;
        int32_t* u0 = a.ub_out + tl.V;
        for (uint32_t v = tl.tid; v < tl.V; v += tl.nth) { const int2 c = cell_bounds<PACKED>(tl.dom[tl.rowof(v)]); l0[v] = c.x; u0[v] = c.y; }
      } else {
        r.stale = 1u;
      }
      new_sp = r.sp + 1;
      r.resume_var = var;
    }
  }
  if (a.dfs.node_limit && r.nodes >= a.dfs.node_limit) r.stop = 1;  // StopNode (stop_node.rs:57-62)
  r.sp = new_sp;
}

// ---- (b) of a round: walk the lists.  Piece p (4 x 64 entries) of list e goes to wavefront (p + e) mod nwv: one long list is spread over
// the workgroup, many lists are balanced to within a piece.  The payload loads of the next TWO pieces are in flight while a
// piece is tested: three register stages in rotation, the loop unrolled three times so that no stage is ever copied (a copy
// would wait for the load it copies).  A piece decodes its entries ONCE and tests them against every node of the entry's mask (quads of
// nodes per ds_read_b128; masks of one or two nodes also fill the jump windows); only flagged entries run the full filter (eval_record).
// Returns this thread's (entry, node) tests.
template <bool PACKED, bool PAY4, class Tile, class Pay>
__device__ __forceinline__ uint32_t neq_walk_lists(const Tile& tl, const NeqArgs& a, const Pay* pay, const uint32_t total, const bool one_piece, const uint32_t round, Ctr& ctr,
                                                    const bool tr_on, unsigned long long* const trbuf) {
  const uint32_t U4 = 4;
  const uint32_t wv = tl.wv, nwv = tl.nwv, lane = tl.lane, B = tl.B;
  auto* const dom = tl.dom;
  uint4* const list = tl.list;
  Win* const win = tl.win;
  auto rowof = [&](uint32_t slot) { return tl.rowof(slot); };
  auto dom_of = [&](uint32_t b, Ctr* c) { return tl.dom_of(b, c); };
  struct Piece { uint32_t v, M, aoff, deg, k0, wsel; };
  // The length of a piece: 4 x 64 entries, or — when the round walks so few lists that whole pieces per wavefront do not come out even —
  // fewer: one list of 2997 entries (a frontier tile: the one queen its nodes have in common) is 12 pieces of 256 for 8 wavefronts,
  // i.e. two rounds of pieces with half of the wavefronts idle in the second, but 16 pieces of 192: two even rounds, a quarter less time.
  uint32_t plen = 64u * U4;
  if (PCP_NEQ_PLEN && !one_piece && total <= 4u) {
    uint32_t work = 0;
    for (uint32_t e_ = 0; e_ < total; ++e_) work += list[e_].z;
    work = (uint32_t)__builtin_amdgcn_readfirstlane(work);
    const uint32_t per = nwv * 64u * U4, r = (work + per - 1u) / per;  // rounds of pieces at full length
    if (r) plen = min(64u * U4, 64u * ((work + 64u * nwv * r - 1u) / (64u * nwv * r)));
  }
  const uint32_t nu = plen >> 6;  // payload loads / entries per lane of a piece (wave-uniform)
  const uint32_t k_step = nwv * plen;
  const uint32_t e_step = one_piece ? nwv : 1u;
  auto k_first = [&](uint32_t e_) { return one_piece ? 0u : ((wv + nwv - (e_ % nwv)) % nwv) * plen; };
  uint32_t e = one_piece ? wv : 0u, k0 = k_first(e);
  // the next piece of this wavefront (deg == 0: none left; its loads then read entry 0 of list 0 and are ignored)
  auto next_piece = [&]() -> Piece {
    while (e < total) {
      const uint4 ent = list[e];
      const uint32_t dg = __builtin_amdgcn_readfirstlane(ent.z);  // wave-uniform: keeps the loop control scalar
      if (k0 < dg) {
        const uint32_t vm = __builtin_amdgcn_readfirstlane(ent.x);
        const Piece pc{vm & 0xffffu, vm >> 16, (uint32_t)__builtin_amdgcn_readfirstlane(ent.y), dg, k0, (uint32_t)__builtin_amdgcn_readfirstlane(ent.w)};
        k0 += k_step;
        return pc;
      }
      e += e_step; k0 = k_first(e);
    }
    return Piece{0u, 0u, 0u, 0u, 0u, 0u};
  };
  auto load = [&](const Piece& pc, Pay (&q)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if ((uint32_t)u >= nu) { q[u] = q[0]; continue; }  // (a short piece: entry 0's payload stands in, masked off below)
      const uint32_t idx = pc.k0 + u * 64 + lane;
      q[u] = pay[pc.aoff + (idx < pc.deg ? idx : 0u)];
    }
  };
  uint32_t my_ev = 0;
  const bool timing = PCP_NEQ_PROFILE && (a.debug & 8u) != 0;  // profiling: s_memtime ticks of the walk / of the node loops, pieces (counters overloaded)
  uint64_t t_walk0 = 0, t_inner = 0, n_pieces = 0;
  if (timing) t_walk0 = __builtin_amdgcn_s_memtime();
  auto process = [&](const Piece& pc, const Pay (&q)[4]) {
    uint32_t other[4];
    int t[4];
    bool valid[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      valid[u] = (uint32_t)u < nu && pc.k0 + u * 64 + lane < pc.deg;
      other[u] = pay_other(q[u]);
      t[u] = pay_t(q[u]);  // v is the record's y: x != v + d  <=>  o != v + d (t = d);  v is x: o != v - d (t = -d)
    }
    bool hit[4] = {false, false, false, false};
    uint64_t ti0 = 0;
    if (timing) { ti0 = __builtin_amdgcn_s_memtime() + (__builtin_amdgcn_readfirstlane(other[0] ^ other[3] ^ (uint32_t)t[1]) & 0u); ++n_pieces; }
    const uint32_t rv = rowof(pc.v);
    uint32_t ro[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ro[u] = rowof(other[u]);
    if constexpr (PACKED) {
      uint32_t K[4], acc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { K[u] = pack_mt(t[u]); acc[u] = 0xffffffffu; }
      if (B >= 4 && __popc(pc.M) > 2) {
        // a quad of nodes per ds_read_b128, two quads per step (ten reads in flight); nodes of a quad outside the mask are
        // tested along: they can only raise a flag that the full-filter pass below, which walks the mask, ignores
        uint32_t qm = 0;
        for (uint32_t g = 0; g < (B >> 2); ++g) qm |= ((pc.M >> (4 * g)) & 0xFu) ? 1u << g : 0u;
#if PCP_NEQ_XOR
        // The walked variable has the SAME cell in every node of these quads — the rule, not the exception: the nodes of a tile are
        // neighbours in the search tree and hold the queens of their common ancestors at the same values.  Then what an entry's other
        // side must match is a property of the entry alone (neq_target16), and an (entry, node) test is one exclusive-or and one
        // packed minimum instead of a packed add on top (v_pk_* issue at half rate: tools/micro/box_probe.hip).
        const uint32_t cvu = dom[rv + 4u * (uint32_t)__builtin_ctz(qm)];
        uint32_t differ = 0;
        for (uint32_t qq = qm; qq; qq &= qq - 1u) {
          const uint4 c = *reinterpret_cast<const uint4*>(dom + rv + 4u * (uint32_t)__builtin_ctz(qq));
          differ |= (c.x ^ cvu) | (c.y ^ cvu) | (c.z ^ cvu) | (c.w ^ cvu);
        }
        if (__builtin_amdgcn_readfirstlane(differ) == 0u) {  // (every lane read the same cells: wave-uniform)
          uint32_t T[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) T[u] = neq_target16(cvu, K[u]);
          if (__popc(qm) & 1) {
            const uint32_t g0 = (uint32_t)__builtin_ctz(qm);
            qm &= qm - 1;
            uint4 o0[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) o0[u] = *reinterpret_cast<const uint4*>(dom + ro[u] + 4 * g0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
              acc[u] = pk_min_u16(acc[u], pk_min_u16(pk_min_u16(o0[u].x ^ T[u], o0[u].y ^ T[u]), pk_min_u16(o0[u].z ^ T[u], o0[u].w ^ T[u])));
          }
          while (qm) {
            const uint32_t g0 = (uint32_t)__builtin_ctz(qm);
            qm &= qm - 1;
            const uint32_t g1 = (uint32_t)__builtin_ctz(qm);
            qm &= qm - 1;
            uint4 o0[4], o1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { o0[u] = *reinterpret_cast<const uint4*>(dom + ro[u] + 4 * g0); o1[u] = *reinterpret_cast<const uint4*>(dom + ro[u] + 4 * g1); }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const uint32_t m0 = pk_min_u16(pk_min_u16(o0[u].x ^ T[u], o0[u].y ^ T[u]), pk_min_u16(o0[u].z ^ T[u], o0[u].w ^ T[u]));
              const uint32_t m1 = pk_min_u16(pk_min_u16(o1[u].x ^ T[u], o1[u].y ^ T[u]), pk_min_u16(o1[u].z ^ T[u], o1[u].w ^ T[u]));
              acc[u] = pk_min_u16(acc[u], pk_min_u16(m0, m1));
            }
          }
        }
#endif
        if (__popc(qm) & 1) {  // an odd quad out, by itself
          const uint32_t g0 = (uint32_t)__builtin_ctz(qm);
          qm &= qm - 1;
          const uint4 c0 = *reinterpret_cast<const uint4*>(dom + rv + 4 * g0);
          uint4 o0[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) if ((uint32_t)u < nu) o0[u] = *reinterpret_cast<const uint4*>(dom + ro[u] + 4 * g0);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if ((uint32_t)u >= nu) continue;  // (a short piece)
            uint32_t m0 = pk_min_u16(neq_terms16(c0.x, o0[u].x, K[u]), neq_terms16(c0.y, o0[u].y, K[u]));
            uint32_t m1 = pk_min_u16(neq_terms16(c0.z, o0[u].z, K[u]), neq_terms16(c0.w, o0[u].w, K[u]));
            acc[u] = pk_min_u16(acc[u], pk_min_u16(m0, m1));
          }
        }
        while (qm) {
          const uint32_t g0 = (uint32_t)__builtin_ctz(qm);
          qm &= qm - 1;
          const uint32_t g1 = (uint32_t)__builtin_ctz(qm);
          qm &= qm - 1;
          const uint4 c0 = *reinterpret_cast<const uint4*>(dom + rv + 4 * g0), c1 = *reinterpret_cast<const uint4*>(dom + rv + 4 * g1);
          uint4 o0[4], o1[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) if ((uint32_t)u < nu) { o0[u] = *reinterpret_cast<const uint4*>(dom + ro[u] + 4 * g0); o1[u] = *reinterpret_cast<const uint4*>(dom + ro[u] + 4 * g1); }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if ((uint32_t)u >= nu) continue;
            uint32_t m0 = pk_min_u16(neq_terms16(c0.x, o0[u].x, K[u]), neq_terms16(c0.y, o0[u].y, K[u]));
            uint32_t m1 = pk_min_u16(neq_terms16(c0.z, o0[u].z, K[u]), neq_terms16(c0.w, o0[u].w, K[u]));
            uint32_t m2 = pk_min_u16(neq_terms16(c1.x, o1[u].x, K[u]), neq_terms16(c1.y, o1[u].y, K[u]));
            uint32_t m3 = pk_min_u16(neq_terms16(c1.z, o1[u].z, K[u]), neq_terms16(c1.w, o1[u].w, K[u]));
            acc[u] = pk_min_u16(acc[u], pk_min_u16(pk_min_u16(m0, m1), pk_min_u16(m2, m3)));
          }
        }
      } else {
        uint32_t k = 0;
        for (uint32_t m = pc.M; m; m &= m - 1, ++k) {
          const uint32_t b = (uint32_t)__builtin_ctz(m);
          const uint32_t c0 = dom[rv + b];
          uint32_t oc[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { oc[u] = dom[ro[u] + b]; acc[u] = pk_min_u16(acc[u], neq_terms16(c0, oc[u], K[u])); }
          const uint32_t wi = k == 0 ? (pc.wsel & 0xffffu) : k == 1 ? (pc.wsel >> 16) : kNoWin;
          if (wi != kNoWin) {
            // the values assigned neighbours forbid for v, near its bounds (the bounds the window was opened with)
            Win* wp = win + wi;
            const int lb0 = wp->lb0, ub0 = wp->ub0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int2 O = unpack16(oc[u]);
              if (valid[u] && O.x == O.y) {
                const int f = O.x - t[u];  // lb(v) + t == O  <=>  lb(v) == f
                const uint32_t dl = (uint32_t)(f - lb0), dh = (uint32_t)(ub0 - f);
                if (dl < 64u) atomicOr(&wp->lo[dl >> 5], 1u << (dl & 31u));
                if (dh < 64u) atomicOr(&wp->hi[dh >> 5], 1u << (dh & 31u));
              }
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) hit[u] = valid[u] && zero_half(acc[u]);
    } else {
      uint32_t k = 0;
      for (uint32_t m = pc.M; m; m &= m - 1, ++k) {
        const uint32_t b = (uint32_t)__builtin_ctz(m);
        const int2 c0 = dom[rv + b];
        const uint32_t wi = k == 0 ? (pc.wsel & 0xffffu) : k == 1 ? (pc.wsel >> 16) : kNoWin;
        Win* wp = win + (wi != kNoWin ? wi : 0u);
        int lb0 = 0, ub0 = 0;
        if (wi != kNoWin) { lb0 = wp->lb0; ub0 = wp->ub0; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // lb(v) + t == ub(o)  |  ub(v) + t == lb(o)
          const int2 o = dom[ro[u] + b];
          hit[u] |= (c0.x + o.y == t[u]) | (c0.y + o.x == -t[u]);
          if (wi != kNoWin && valid[u] && -o.x == o.y) {
            const int f = o.y - t[u];
            const uint32_t dl = (uint32_t)(f - lb0), dh = (uint32_t)(ub0 - f);
            if (dl < 64u) atomicOr(&wp->lo[dl >> 5], 1u << (dl & 31u));
            if (dh < 64u) atomicOr(&wp->hi[dh >> 5], 1u << (dh & 31u));
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) hit[u] = hit[u] && valid[u];
    }
    if (timing) t_inner += __builtin_amdgcn_s_memtime() + (__builtin_amdgcn_readfirstlane((uint32_t)hit[0] | (uint32_t)hit[3]) & 0u) - ti0;
    const uint32_t nm = (uint32_t)__popc(pc.M);
#pragma unroll
    for (int u = 0; u < 4; ++u) my_ev += valid[u] ? nm : 0u;
    if (hit[0] | hit[1] | hit[2] | hit[3]) {
      // flagged entries: the full filter, in the nodes whose domains meet the condition
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!hit[u]) continue;
        const bool is_y = pay_is_y(q[u]);
        Rec rec;
        rec.xk = (is_y ? other[u] : pc.v) | ((uint32_t)PCP_NEQ << 28);
        rec.y = is_y ? pc.v : other[u];
        rec.z = 0;
        rec.d = is_y ? t[u] : -t[u];
        uint32_t k = 0;
        for (uint32_t m = pc.M; m; m &= m - 1, ++k) {
          const uint32_t b = (uint32_t)__builtin_ctz(m);
          const int2 Vd = cell_bounds<PACKED>(dom[rv + b]), O = cell_bounds<PACKED>(dom[ro[u] + b]);
          if (Vd.x + t[u] != O.y && Vd.y + t[u] != O.x) continue;
          // a node with a jump window: an ASSIGNED neighbour at a bound of the unassigned variable is a bit of the window — the jump behind
          // the barrier removes it together with the values behind it, once, instead of one value here and the rest there
          const uint32_t wi = k == 0 ? (pc.wsel & 0xffffu) : k == 1 ? (pc.wsel >> 16) : kNoWin;
          if (wi != kNoWin && !(a.debug & 65536u) && O.x == O.y && Vd.x < Vd.y) {
            const Win* wp = win + wi;
            if ((uint32_t)(Vd.x - wp->lb0) < 64u && (uint32_t)(wp->ub0 - Vd.y) < 64u) continue;
          }
          ++ctr.full;
          eval_record(rec, dom_of(b, &ctr));
        }
      }
    }
  };
  Piece pa = next_piece(), pb, pc3;
  Pay qA[4], qB[4], qC[4];
  load(pa, qA);
  if (round == 0) PCP_TR(6);
  pb = next_piece(); load(pb, qB);
  while (pa.deg) {
    pc3 = next_piece(); load(pc3, qC);
    process(pa, qA);
    if (!pb.deg) break;
    pa = next_piece(); load(pa, qA);
    process(pb, qB);
    if (!pc3.deg) break;
    pb = next_piece(); load(pb, qB);
    process(pc3, qC);
  }
  if (timing && lane == 0 && round == 0) {
    atomicAdd((unsigned long long*)&a.stats->steps3, (unsigned long long)(__builtin_amdgcn_s_memtime() - t_walk0));
    atomicAdd((unsigned long long*)&a.stats->failed_nodes, (unsigned long long)t_inner);
    atomicAdd((unsigned long long*)&a.stats->waves, (unsigned long long)n_pieces);
  }
  return my_ev;
}

// ---- round 0 of a FRONTIER tile, lean (round 6).  A full tile of sixteen nodes in 16-bit cells whose nodes have at most kFastLists assigned
// variables between them — the tiles of a breadth-first frontier, the headline batch — used to go through the general round: complete the list's
// entries, barrier, pieces of 4 x 64 entries dealt to the wavefronts (twelve pieces on eight wavefronts: two rounds of pieces, the second half
// empty), barrier with a vote, status scan, barrier: 19 000 of a tile's 31 000 ticks (tools/neq_trace.py), most of them waiting.  Here every
// wavefront reads the (few) list heads by itself — no barrier —, a list's entries are dealt EVENLY, entry j * threads + tid to thread tid (2997
// entries on 512 threads: six loads per lane, all in flight at once), every entry is tested against all sixteen nodes (nodes outside the variable's
// mask ride along: they can only raise a flag that the full filter, which walks the mask, ignores), and the status scan runs BEFORE the round's one
// barrier, so that its memory round trip overlaps the other wavefronts' walks.
// A flagged entry runs the full filter (eval_record: XNeqY::propagate, x_neq_y.rs:82-93) on a QUIET store (TileDom16Q): the narrowed variable is not
// woken unless it was assigned or emptied.  Instead, when a pass narrowed something and woke nothing, the SAME lists are walked again, until a pass
// narrows nothing.  Same fixpoint: a propagator x != y + c acts only when one side is assigned, so every propagator that can act on a narrowed,
// still unassigned variable sits in the list of an assigned variable — and the lists walked here are those of ALL assigned variables of every node
// of the tile (staging listed them: vmk), none of which changed.  (The general rounds make the same choice when it is cheaper: resweep_marks.)
// A pass that assigned or emptied a variable hands the tile to the general rounds (the new variable's own list must run).
// The payload of the tile's FIRST list is requested once per tile and kept in registers (neq_fast_load): a later pass — and there is one only in
// the few tiles that narrow anything — tests the same entries again without another memory round trip (in the launch's first generation of tiles,
// when every CU of the chip is streaming rows, a round trip was 3-5 thousand ticks, and a tile that narrowed paid four of them in a row).  Further
// lists (a frontier tile has one, seldom two) are loaded where they are tested.
template <class Tile>
__device__ __forceinline__ void neq_fast_load(const Tile& tl, const uint32_t* __restrict__ pay, const uint32_t e, uint32_t (&q)[kFastPer]) {
  const uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane(tl.list[e].x) & 0xffffu;
  const uint32_t o0 = (uint32_t)__builtin_amdgcn_readfirstlane(tl.adjo[v]), dg = (uint32_t)__builtin_amdgcn_readfirstlane(tl.adjo[v + 1]) - o0;
  // (all kFastPer loads, unconditionally — a shorter list reads its last entry again, masked off by the test —: a conditional load into a register
  // array makes the compiler thread the whole array through memory)
  const uint32_t last = o0 + (dg ? dg - 1u : 0u);
#pragma unroll
  for (uint32_t j = 0; j < kFastPer; ++j) q[j] = pay[min(o0 + j * tl.nth + tl.tid, last)];
}
// One pass over the tile's lists for the nodes `only`; q0 = list 0's payload (neq_fast_load).  Returns: this lane narrowed something; `more`: it left
// a flagged entry untested.
template <class Tile>
__device__ __forceinline__ bool neq_fast_test(const Tile& tl, const uint32_t* __restrict__ pay, const uint32_t* vmk, const uint32_t cnt, const uint32_t only,
                                               const uint32_t (&q0)[kFastPer], Ctr& ctr, uint32_t& my_ev, bool& woke, bool& more) {
  const uint32_t narrow0 = ctr.narrow;
  auto* const dom = tl.dom;
  auto one_list = [&](const uint32_t v, const uint32_t M, const uint32_t dg, const uint32_t (&q)[kFastPer]) {
    const uint32_t qm = ((M & 0xFu) ? 1u : 0u) | ((M & 0xF0u) ? 2u : 0u) | ((M & 0xF00u) ? 4u : 0u) | ((M & 0xF000u) ? 8u : 0u);  // quads of nodes with a node of the mask
    const uint32_t per = (dg + tl.nth - 1u) / tl.nth;
    const uint32_t rv = tl.rowof(v);
    uint4 c[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) c[g] = *reinterpret_cast<const uint4*>(dom + rv + 4 * g);  // the walked variable's cells in the sixteen nodes
    const bool all4 = qm == 15u;  // (uniform) the first pass of a frontier tile: every quad; a later pass: the quads of the nodes that narrowed
    uint32_t dif = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) dif |= (c[g].x ^ c[0].x) | (c[g].y ^ c[0].x) | (c[g].z ^ c[0].x) | (c[g].w ^ c[0].x);
    const bool same = PCP_NEQ_XOR && __builtin_amdgcn_readfirstlane(dif) == 0u;  // (every lane read the same cells: wave-uniform)
    const uint32_t nm = (uint32_t)__popc(M);
    // a lane's flagged entry (one is kept; a second one in the same lane and list — entries 512 apart both at a bound — sets `more`: the caller
    // runs another pass over every node, which finds it)
    bool have = false;
    uint32_t hq = 0;
#pragma unroll
    for (uint32_t j = 0; j < kFastPer; ++j) {
      if (j >= per) break;  // (uniform)
      const bool valid = j * tl.nth + tl.tid < dg;
      const uint32_t ro = tl.rowof(pay_other(q[j])), K = pack_mt(pay_t(q[j]));
      uint32_t acc = 0xffffffffu;
      if (all4 && same) {
        // the walked variable has ONE cell in all sixteen nodes (the nodes of a frontier tile are siblings and cousins: they hold their common
        // ancestors' queens at the same values): what the other side must match is a property of the entry (neq_target16), and an (entry, node)
        // test is one exclusive-or and one packed minimum
        const uint32_t T = neq_target16(c[0].x, K);
        uint4 o[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) o[g] = *reinterpret_cast<const uint4*>(dom + ro + 4 * g);
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc = pk_min_u16(acc, pk_min_u16(pk_min_u16(o[g].x ^ T, o[g].y ^ T), pk_min_u16(o[g].z ^ T, o[g].w ^ T)));
      } else if (all4) {
        uint4 o[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) o[g] = *reinterpret_cast<const uint4*>(dom + ro + 4 * g);
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc = pk_min_u16(acc, pk_min_u16(pk_min_u16(neq_terms16(c[g].x, o[g].x, K), neq_terms16(c[g].y, o[g].y, K)),
                                           pk_min_u16(neq_terms16(c[g].z, o[g].z, K), neq_terms16(c[g].w, o[g].w, K))));
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (!((qm >> g) & 1u)) continue;  // (uniform)
          const uint4 og = *reinterpret_cast<const uint4*>(dom + ro + 4 * g);
          acc = pk_min_u16(acc, pk_min_u16(pk_min_u16(neq_terms16(c[g].x, og.x, K), neq_terms16(c[g].y, og.y, K)),
                                           pk_min_u16(neq_terms16(c[g].z, og.z, K), neq_terms16(c[g].w, og.w, K))));
        }
      }
      const bool hit = valid && zero_half(acc);
      more |= hit && have;
      hq = (hit && !have) ? q[j] : hq;
      have |= hit;
      my_ev += valid ? nm : 0u;
    }
    if (have) {  // the flagged entry: the full filter, in the nodes of the mask whose domains meet the condition
      const uint32_t other = pay_other(hq);
      const int t = pay_t(hq);
      const bool is_y = pay_is_y(hq);
      Rec rec;
      rec.xk = (is_y ? other : v) | ((uint32_t)PCP_NEQ << 28);
      rec.y = is_y ? v : other;
      rec.z = 0;
      rec.d = is_y ? t : -t;
      const uint32_t ro = tl.rowof(other);
      for (uint32_t m = M; m; m &= m - 1u) {
        const uint32_t b = (uint32_t)__builtin_ctz(m);
        const int2 Vd = unpack16(dom[rv + b]), O = unpack16(dom[ro + b]);
        if (Vd.x + t != O.y && Vd.y + t != O.x) continue;
        ++ctr.full;
        const TileDom16 base = tl.dom_of(b, &ctr);
        TileDom16Q dq;
        static_cast<TileDom16&>(dq) = base;
        dq.woke = &woke;
        eval_record(rec, dq);
      }
    }
  };
  auto head = [&](const uint32_t e, uint32_t& v, uint32_t& M, uint32_t& dg) {
    v = (uint32_t)__builtin_amdgcn_readfirstlane(tl.list[e].x) & 0xffffu;
    M = ((uint32_t)__builtin_amdgcn_readfirstlane(vmk[v >> 1]) >> (16u * (v & 1u))) & 0xffffu & only;  // (`only`: the nodes this pass is about)
    dg = (uint32_t)__builtin_amdgcn_readfirstlane(tl.adjo[v + 1]) - (uint32_t)__builtin_amdgcn_readfirstlane(tl.adjo[v]);
  };
  uint32_t q[kFastPer];
#pragma unroll
  for (uint32_t j = 0; j < kFastPer; ++j) q[j] = q0[j];
  for (uint32_t e = 0; e < cnt; ++e) {
    uint32_t v, M, dg;
    head(e, v, M, dg);
    if (!M || !dg) continue;
    if (e) neq_fast_load(tl, pay, e, q);
    one_list(v, M, dg, q);
  }
  return ctr.narrow != narrow0;
}

// ---- (a) of a round: one list for the tile, (variable, mask of the nodes in which it changed).  The marks of the listed variables are
// consumed here (the narrowings of this round set them again behind the barrier); variables beyond the list's capacity keep their
// marks and are listed by the next round.
// One WAVEFRONT per mask word, lane b = node b.  Four words per step with their LDS reads in flight together; the variables of
// a word come out of ballots and readlanes alone: the lanes that still hold an unlisted bit are balloted, the first of them names
// a bit, a second ballot over that bit is the variable's node mask.  (One ballot per bit position of every non-empty word, each
// behind a dependent LDS read, made this pass 10 000 cycles of a frontier tile's 45 000 for ONE listed variable.)
// Round 0 of a frontier tile finds its list built by the staging loop (`vmk`: the node masks) and only completes the entries.
template <bool PACKED, class Tile>
__device__ __forceinline__ void neq_build_list(const Tile& tl, uint32_t* const vmk, const uint32_t round, const bool r0_direct, const uint32_t inert, const uint32_t wcap) {
  const uint32_t tid = tl.tid, nth = tl.nth, lane = tl.lane, wv = tl.wv, nwv = tl.nwv, nb = tl.nb, B = tl.B, V = tl.V, Wv = tl.Wv;
  auto* const dom = tl.dom;
  uint32_t* const chg = tl.chg;
  uint32_t* const misc = tl.misc;
  const uint32_t* const adjo = tl.adjo;
  uint4* const list = tl.list;
  Win* const win = tl.win;
  auto rowof = [&](uint32_t slot) { return tl.rowof(slot); };
  const uint32_t m_count = (round & 1u) ? N_COUNT1 : N_COUNT0, m_rmask = (round & 1u) ? N_RMASK1 : N_RMASK0, m_win = (round & 1u) ? N_WIN1 : N_WIN0;
  const bool r0_listed = round == 0 && r0_direct && misc[N_R0OVF] == 0u;  // (workgroup-uniform: written before the staging barrier)
  if (r0_listed) {
    // round 0's list is there already (staging): complete its entries — mask without the failed and refused nodes, list offset, degree —
    // and drop the marks staging set for the same variables (kept until here for the overflow case below)
    for (uint32_t i = tid; i < B * Wv; i += nth) chg[i] = 0;
    if (tid < misc[N_COUNT0]) {
      const uint32_t v = list[tid].x;
      const uint32_t M = (vmk[v >> 1] >> (16u * (v & 1u))) & 0xffffu & ~inert;
      const uint32_t o0 = adjo[v], dg = adjo[v + 1] - o0;
      list[tid] = make_uint4(v | (M << 16), o0, dg, kNoWin | (kNoWin << 16));
      if (M) atomicOr(&misc[m_rmask], M);
    }
  } else {
    if (round == 0 && r0_direct) {  // more assigned variables than the list holds: the marks are scanned as in every other round
      __syncthreads();
      if (tid == 0) misc[N_COUNT0] = 0;
      __syncthreads();
    }
    uint32_t rm = 0;
    bool list_full = false;
    for (uint32_t w0 = wv; w0 < Wv && !list_full; w0 += 4 * nwv) {
      uint32_t xs[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const uint32_t w = w0 + j * nwv; xs[j] = (lane < nb && w < Wv) ? chg[lane * Wv + w] : 0u; }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t w = w0 + j * nwv;
        uint32_t x = xs[j];
        if ((inert >> lane) & 1u) { if (x) chg[lane * Wv + w] = 0; x = 0; }  // a failed or refused node is inert  (lanes >= nb hold 0)
        uint32_t taken = 0;  // wave-uniform: the bits of this word listed so far
        while (!list_full) {
          const unsigned long long holders = __ballot((x & ~taken) != 0u);
          if (!holders) break;
          const uint32_t xf = (uint32_t)__builtin_amdgcn_readlane((int)(x & ~taken), (int)__builtin_ctzll(holders));
          const uint32_t i = (uint32_t)__builtin_ctz(xf);
          const uint32_t M = (uint32_t)__ballot((x >> i) & 1u);
          // lane 0 emits the entry (wave-uniform values: every lane computes them, one writes)
          uint32_t pos = 0;
          if (lane == 0) pos = atomicAdd(&misc[m_count], 1u);
          pos = __builtin_amdgcn_readfirstlane(pos);
          if (pos >= kListCap) {  // full: this variable and the rest wait for the next round
            if (lane == 0) { atomicSub(&misc[m_count], 1u); misc[N_MORE] = round + 1; }
            list_full = true;
            break;
          }
          taken |= 1u << i;
          const uint32_t v = (w << 5) + i;
          const uint32_t o0 = v < V ? adjo[v] : 0u, dg = v < V ? adjo[v + 1] - o0 : 0u;
          // jump windows: a list walked for one or two nodes only, not in the sweep round, the variable not assigned
          uint32_t wsel = kNoWin | (kNoWin << 16);
          if (round && wcap && __popc(M) <= 2) {
            uint32_t k = 0;
            for (uint32_t m = M; m; m &= m - 1, ++k) {
              const uint32_t b = (uint32_t)__builtin_ctz(m);
              const int2 d = cell_bounds<PACKED>(dom[rowof(v) + b]);
              if (d.x >= d.y) continue;
              uint32_t wi = 0;
              if (lane == 0) wi = atomicAdd(&misc[m_win], 1u);
              wi = __builtin_amdgcn_readfirstlane(wi);
              if (wi >= wcap) continue;
              if (lane == 0) {
                Win nw;
                nw.lo[0] = nw.lo[1] = nw.hi[0] = nw.hi[1] = 0u; nw.lb0 = d.x; nw.ub0 = d.y; nw.vb = v | (b << 16); nw.pad = 0u;
                win[wi] = nw;
              }
              wsel = k == 0 ? ((wsel & 0xffff0000u) | wi) : ((wsel & 0xffffu) | (wi << 16));
            }
          }
          if (lane == 0) list[pos] = make_uint4(v | (M << 16), o0, dg, wsel);
          rm |= M;
        }
        if (lane < nb && (x & taken)) chg[lane * Wv + w] = x & ~taken;
      }
    }
    if (rm && lane == 0) atomicOr(&misc[m_rmask], rm);
  }
}

// ------------------------------------------------------------------------------------------------
// Staging of a FULL tile (16 nodes, 16-bit cells, whole 16-byte quads, aligned rows) — the frontier launch's dominant phase by
// instructions: 59 % of its VALU and 67 % of its SALU wave-instructions were staging (profiles/r05_phases_*: 100 VALU + 73 SALU per
// wave-task of 64 row quads, most of them predicates, address arithmetic and scalars spilled into VGPR lanes by the pressure of the
// kernel around the loop).  This function is the same loop with nothing around it: out of line, so that it has its own register
// allocation; no lane predicate at all — a full tile has no ragged node, and the quads beyond the row's end in the last chunk of 16 are
// CLAMPED to the row's last quad (those lanes load and write the last quad's cells a second time: same values, same addresses);
// a wave-task's row offset is a scalar (buffer_load soffset), its LDS offset a scalar added to a per-lane constant.
// Semantics of the kernel's `put` (fast path and its rare branch) exactly; returns bad | oob << 16 (bit b = node b).
// A wave-task = FOUR nodes x SIXTEEN consecutive quads (see the kernel: the cells a wavefront writes per store fall on all LDS banks).
// CELLS (pcp_device_batch.cell_format PCP_CELLS_PACKED16): the rows ARE cells — one 16-byte load per quad instead of two, no packing; a
// singleton shows as a zero 16-bit sum of the two halves, an empty domain as a negative one, a field outside +-16384 refuses the node.
struct StageTile16Args {
  const int32_t* lb;            // the tile's rows: [16][V], 16-byte aligned (CELLS: the rows of cells)
  const int32_t* ub;            // (CELLS: unused)
  const uint32_t* seed_always;  // or null
  uint32_t V, Wv;
  uint32_t dom_off, chg_off, vmk_off, list_off, misc_off;  // byte offsets into the workgroup's dynamic LDS
  uint32_t hintm;               // nodes that came with a dirty-variable hint: their assigned variables are not marked
  uint32_t r0_direct;           // round 0's list is built here (vmk masks, up to kR0Cap variables)
  uint32_t wv, nwv;
};
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#ifndef PCP_STAGE_UF
#define PCP_STAGE_UF 4
#endif
// wave-tasks (one 16-byte load of lb and one of ub per lane each) in flight per wavefront in a full tile's staging loop.  FOUR, i.e. two rounds of
// loads per tile at 1000 variables on 512 threads: with all eight at once (the whole tile requested in one go: rounds 4-5) every CU of the chip puts
// 256 KB into the memory system's queues at the same moment and who is served last waits for everybody — 8 / 4 / 3 / 2 in flight, one box, two runs each:
// 41.0-41.5 / 40.0-40.8 / 40.4-41.2 / 41.0-42.3 us for the headline launch, 28.0 / 27.3 / 27.0 / 27.1 for one generation of tiles (tools/ab_variants.sh).
constexpr int kStage16UF = PCP_STAGE_UF;
template <bool CELLS>
__device__ __forceinline__ uint32_t stage_tile16(const StageTile16Args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_[];
  constexpr int UF = kStage16UF;
  constexpr int lim = kPackedMax;
  const uint32_t lane = threadIdx.x & 63u, lb4 = lane >> 4, lq = lane & 15u;
  const uint32_t V = g.V, SQ = V >> 2, QC = (SQ + 15u) >> 4, WT = 4u * QC;
  const uint32_t lq_last = min(lq, SQ - 1u - 16u * (QC - 1u));  // the last chunk of a row may hold fewer than 16 quads
  const uint32_t vo_full = lb4 * V * 4u + 16u * lq, vo_last = lb4 * V * 4u + 16u * lq_last;      // row bytes: node lb4 of the group, quad lq of the chunk
  const uint32_t do_full = g.dom_off + lq * 272u + lb4 * 4u, do_last = g.dom_off + lq_last * 272u + lb4 * 4u;  // cell bytes: row(4 q) = 68 q words
  uint32_t* const chg = reinterpret_cast<uint32_t*>(smem_ + g.chg_off);
  uint32_t* const vmk = reinterpret_cast<uint32_t*>(smem_ + g.vmk_off);
  uint4* const list = reinterpret_cast<uint4*>(smem_ + g.list_off);
  uint32_t* const misc = reinterpret_cast<uint32_t*>(smem_ + g.misc_off);
  const __amdgpu_buffer_rsrc_t rs_lb = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(g.lb), 0, (int)(16u * V * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_ub = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(CELLS ? g.lb : g.ub), 0, (int)(16u * V * 4u), 0x00020000);
  const uint32_t dqc = g.nwv % QC, dng = g.nwv / QC;
  uint32_t ng = g.wv / QC, qc = g.wv - ng * QC;  // (wave-uniform: scalar registers)
  uint32_t badm = 0, oobm = 0;
  for (uint32_t w0 = g.wv; w0 < WT; w0 += UF * g.nwv) {
    u32x4 L[UF], U[UF];
    uint32_t ngj[UF], qcj[UF];
#pragma unroll
    for (int j = 0; j < UF; ++j) {
      ngj[j] = ng; qcj[j] = qc;
      if (w0 + j * g.nwv < WT) {  // (uniform)
        const uint32_t so = ng * (16u * V) + qc * 256u, vo = qc == QC - 1u ? vo_last : vo_full;
        L[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_lb, (int)vo, (int)so, PCP_STAGE_AUX);
        if constexpr (!CELLS) U[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_ub, (int)vo, (int)so, PCP_STAGE_AUX);
      }
      qc += dqc; ng += dng;
      if (qc >= QC) { qc -= QC; ++ng; }
    }
#pragma unroll
    for (int j = 0; j < UF; ++j) {
      if (w0 + j * g.nwv >= WT) break;  // (uniform)
      const bool lastc = qcj[j] == QC - 1u;
      if constexpr (CELLS) {
        const uint32_t c0 = L[j].x, c1 = L[j].y, c2 = L[j].z, c3 = L[j].w;
        uint32_t* const p0 = reinterpret_cast<uint32_t*>(smem_ + ((lastc ? do_last : do_full) + qcj[j] * (16u * 272u) + ngj[j] * 16u));
        p0[0] = c0; p0[16] = c1; p0[32] = c2; p0[48] = c3;
        const uint32_t s0 = (c0 + (c0 >> 16)) & 0xffffu, s1 = (c1 + (c1 >> 16)) & 0xffffu, s2 = (c2 + (c2 >> 16)) & 0xffffu, s3 = (c3 + (c3 >> 16)) & 0xffffu;  // ub - lb, 16 bits
        const uint32_t rng = (pk_add16(c0, 0x40004000u) | pk_add16(c1, 0x40004000u) | pk_add16(c2, 0x40004000u) | pk_add16(c3, 0x40004000u)) & 0x80008000u;
        if ((min(min(s0, s1), min(s2, s3)) == 0u) | (((s0 | s1 | s2 | s3) & 0x8000u) != 0u) | (rng != 0u) || g.seed_always) {
          const uint32_t b = 4u * ngj[j] + lb4, v0 = 4u * (16u * qcj[j] + (lastc ? lq_last : lq));
          uint32_t nib = (s0 == 0u ? 1u : 0u) | (s1 == 0u ? 2u : 0u) | (s2 == 0u ? 4u : 0u) | (s3 == 0u ? 8u : 0u);
          if (g.seed_always) nib |= (g.seed_always[v0 >> 5] >> (v0 & 31u)) & 15u;
          if ((g.hintm >> b) & 1u) nib = 0;
          if (nib) atomicOr(&chg[b * g.Wv + (v0 >> 5)], nib << (v0 & 31u));
          if (g.r0_direct && misc[N_R0OVF] == 0u) {
            for (uint32_t m = nib; m; m &= m - 1u) {
              const uint32_t v = v0 + (uint32_t)__builtin_ctz(m), hs = 16u * (v & 1u);
              const uint32_t old = atomicOr(&vmk[v >> 1], (1u << b) << hs);
              if (((old >> hs) & 0xffffu) == 0u) {
                const uint32_t pos = atomicAdd(&misc[N_COUNT0], 1u);
                if (pos < kR0Cap) list[pos].x = v; else misc[N_R0OVF] = 1u;
              }
            }
          }
          if (rng) oobm |= 1u << b;
          else if ((s0 | s1 | s2 | s3) & 0x8000u) badm |= 1u << b;
        }
        continue;
      }
      const int l0 = (int)L[j].x, l1 = (int)L[j].y, l2 = (int)L[j].z, l3 = (int)L[j].w, u0 = (int)U[j].x, u1 = (int)U[j].y, u2 = (int)U[j].z, u3 = (int)U[j].w;
      const int mn = min(min(min(l0, l1), min(l2, l3)), min(min(u0, u1), min(u2, u3)));
      const int mx = max(max(max(l0, l1), max(l2, l3)), max(max(u0, u1), max(u2, u3)));
      const int dmin = min(min(u0 - l0, u1 - l1), min(u2 - l2, u3 - l3));
      uint32_t* const p0 = reinterpret_cast<uint32_t*>(smem_ + ((lastc ? do_last : do_full) + qcj[j] * (16u * 272u) + ngj[j] * 16u));
      p0[0] = __builtin_amdgcn_perm((uint32_t)u0, (uint32_t)(-l0), 0x05040100u);  // ub << 16 | (-lb & 0xffff)
      p0[16] = __builtin_amdgcn_perm((uint32_t)u1, (uint32_t)(-l1), 0x05040100u);
      p0[32] = __builtin_amdgcn_perm((uint32_t)u2, (uint32_t)(-l2), 0x05040100u);
      p0[48] = __builtin_amdgcn_perm((uint32_t)u3, (uint32_t)(-l3), 0x05040100u);
      // everything else is ONE rarely taken branch: a bound out of range (the node is refused), an empty domain (failed), a singleton (an
      // assigned variable: marked for the sweep round) or a model with Constant neighbours
      if (((mn < -lim) | (mx > lim) | (dmin <= 0)) || g.seed_always) {
        const uint32_t b = 4u * ngj[j] + lb4, v0 = 4u * (16u * qcj[j] + (lastc ? lq_last : lq));
        uint32_t nib = (l0 == u0 ? 1u : 0u) | (l1 == u1 ? 2u : 0u) | (l2 == u2 ? 4u : 0u) | (l3 == u3 ? 8u : 0u);
        if (g.seed_always) nib |= (g.seed_always[v0 >> 5] >> (v0 & 31u)) & 15u;
        if ((g.hintm >> b) & 1u) nib = 0;
        if (nib) atomicOr(&chg[b * g.Wv + (v0 >> 5)], nib << (v0 & 31u));
        if (g.r0_direct && misc[N_R0OVF] == 0u) {
          for (uint32_t m = nib; m; m &= m - 1u) {
            const uint32_t v = v0 + (uint32_t)__builtin_ctz(m), hs = 16u * (v & 1u);
            const uint32_t old = atomicOr(&vmk[v >> 1], (1u << b) << hs);
            if (((old >> hs) & 0xffffu) == 0u) {  // the first node of the tile with this variable: it goes on the list
              const uint32_t pos = atomicAdd(&misc[N_COUNT0], 1u);
              if (pos < kR0Cap) list[pos].x = v; else misc[N_R0OVF] = 1u;
            }
          }
        }
        if ((mn < -lim) | (mx > lim)) oobm |= 1u << b;
        if (dmin < 0) badm |= 1u << b;
      }
    }
  }
  return badm | (oobm << 16);
}

// ---- phase 0 of a tile: stage the nodes' rows as cells (16-byte row loads where the rows allow it), find the assigned variables — they
// are marked for round 0, or listed directly (`r0_direct`: vmk masks + the list's first kR0Cap entries) —, fail nodes with an empty
// domain, refuse nodes with a bound outside the cells' range, and mark the ONE changed variable of hinted nodes instead of their assigned
// ones.  The first tile of a workgroup also stores the list offsets it requested before (adj_pre) behind its row loads.
template <bool PACKED, bool DFS, int BT, bool CELLS, class Tile>
__device__ __forceinline__ void neq_stage_tile(const Tile& tl, const NeqArgs& a, uint32_t* const vmk, const NeqCarve& cv, unsigned char* const smem, const uint32_t S,
                                                const uint32_t node0, const bool r0_direct, const bool vec, const uint32_t dfs_hint, uint32_t (&adj_pre)[4], bool& adj_stored) {
  using Cell = typename NeqCell<PACKED>::type;
  const uint32_t tid = tl.tid, nth = tl.nth, nb = tl.nb, B = tl.B, V = tl.V, Wv = tl.Wv;
  Cell* const dom = tl.dom;
  uint32_t* const chg = tl.chg;
  uint32_t* const misc = tl.misc;
  uint32_t* const adjo = const_cast<uint32_t*>(tl.adjo);
  uint4* const list = tl.list;
  auto rowof = [&](uint32_t slot) { return tl.rowof(slot); };
  const int lim = PACKED ? kPackedMax : kBoundMax;
  uint32_t badm = 0, oobm = 0;
  const uint32_t hintm = DFS ? (dfs_hint < V ? 1u : 0u) : a.dirty ? (uint32_t)__builtin_amdgcn_readfirstlane(misc[N_HINT]) : 0u;  // (written before the barrier above)
  // returns bit 0 = an empty domain among the four, bit 1 = a bound out of range (the callers collect them per node)
  auto put = [&](uint32_t b, uint32_t v0, const int (&l)[4], const int (&u)[4], uint32_t cnt) -> uint32_t {
    if (PCP_PUT_FAST && cnt == 4) {
      // Four whole slots — every put of a store whose size is a multiple of four.  The kernel is bound by instruction issue (four
      // wavefronts share a SIMD), so this is counted in instructions: the range check is a minimum and a maximum over the eight
      // bounds (v_min3 / v_max3) instead of sixteen compares; an empty or a singleton domain shows as min(ub - lb) <= 0, and only
      // then are the four looked at one by one; a cell is packed by one subtraction and one byte permute.
      const int mn = min(min(min(l[0], l[1]), min(l[2], l[3])), min(min(u[0], u[1]), min(u[2], u[3])));
      const int mx = max(max(max(l[0], l[1]), max(l[2], l[3])), max(max(u[0], u[1]), max(u[2], u[3])));
      const int dmin = min(min(u[0] - l[0], u[1] - l[1]), min(u[2] - l[2], u[3] - l[3]));
      // (v0 is a multiple of four: the four rows are B cells apart, no padding between them.  With 16-node tiles row(4q) = 68 q: a 24-bit
      // multiply, full rate — the 32-bit v_mul_lo_u32 the compiler picks for "q * 272 bytes" is quarter rate, sixteen cycles per wavefront)
      Cell* const p0 = dom + (BT >= 16 ? __umul24(v0 >> 2, 4u * B + 4u) : rowof(v0)) + b;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (PACKED) p0[i * B] = __builtin_amdgcn_perm((uint32_t)u[i], (uint32_t)(-l[i]), 0x05040100u);  // ub << 16 | (-lb & 0xffff)
        else p0[i * B] = make_int2(-l[i], u[i]);
      }
      // everything else is ONE rarely taken branch: a bound out of range (the node is refused, not wrapped: pcp_hip.h), an empty domain
      // (the node is failed), a singleton (an assigned variable: marked for the sweep round) or a model with Constant neighbours
      if (((mn < -lim) | (mx > lim) | (dmin <= 0)) || a.seed_always) {
        uint32_t nib = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) nib |= (l[i] == u[i]) ? 1u << i : 0u;
        if (a.seed_always) nib |= (a.seed_always[v0 >> 5] >> (v0 & 31u)) & 15u;
        if ((hintm >> b) & 1u) nib = 0;  // a hinted node: its assigned variables' records ran at the parent's fixpoint
        if (nib) atomicOr(&chg[b * Wv + (v0 >> 5)], nib << (v0 & 31u));
        // (only while the tile has few assigned variables — a frontier: deep tiles, where most quads come through here, give up after
        // kR0Cap variables and pay one LDS read per quad from then on; they are listed by the scan, which is a small part of THEIR time)
        if (r0_direct && misc[N_R0OVF] == 0u) {
          for (uint32_t m = nib; m; m &= m - 1u) {
            const uint32_t v = v0 + (uint32_t)__builtin_ctz(m), hs = 16u * (v & 1u);
            const uint32_t old = atomicOr(&vmk[v >> 1], (1u << b) << hs);
            if (((old >> hs) & 0xffffu) == 0u) {  // the first node of the tile with this variable: it goes on the list
              const uint32_t pos = atomicAdd(&misc[N_COUNT0], 1u);
              if (pos < kR0Cap) list[pos].x = v; else misc[N_R0OVF] = 1u;
            }
          }
        }
        return (((mn < -lim) | (mx > lim)) ? 2u : 0u) | (dmin < 0 ? 1u : 0u);
      }
      return 0u;
    }
    uint32_t nib = 0;
    bool bad = false, oob = false;
    Cell cl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool on = (uint32_t)i < cnt;
      bad |= on && l[i] > u[i];                                                    // empty input domain: the node is failed
      oob |= on && ((l[i] < -lim) | (l[i] > lim) | (u[i] < -lim) | (u[i] > lim));  // refused, not wrapped (pcp_hip.h)
      nib |= (on && l[i] == u[i]) ? 1u << i : 0u;
      if constexpr (PACKED) cl[i] = pack16(l[i], u[i]); else cl[i] = make_int2(-l[i], u[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if ((uint32_t)i < cnt) dom[rowof(v0 + i) + b] = cl[i];
    if (a.seed_always) nib |= (a.seed_always[v0 >> 5] >> (v0 & 31u)) & ((1u << cnt) - 1u);
    if ((hintm >> b) & 1u) nib = 0;
    if (nib) atomicOr(&chg[b * Wv + (v0 >> 5)], nib << (v0 & 31u));
    return (bad ? 1u : 0u) | (oob ? 2u : 0u);
  };
  auto note = [&](uint32_t f, uint32_t b) { if (f) { badm |= (f & 1u) << b; oobm |= (f >> 1) << b; } };
  const uint32_t SQ = (V + 3) >> 2, tasks = nb * SQ;
  if (CELLS) {
    // the rows are cells already (PCP_CELLS_PACKED16): a copy into the node-minor layout plus the checks
    if constexpr (CELLS) {
      const uint32_t* const rows = reinterpret_cast<const uint32_t*>(a.lb_in);
      if (!adj_stored) {
        adj_stored = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint32_t v = tid + j * nth; if (v <= V) adjo[v] = adj_pre[j]; }
      }
      if (BT == 16 && nb == 16u && (V & 3u) == 0 && ((size_t)rows & 15u) == 0) {
        const uint32_t r = stage_tile16<true>(StageTile16Args{a.lb_in + (size_t)node0 * V, nullptr, a.seed_always, V, Wv, (uint32_t)cv.dom, (uint32_t)cv.chg,
                                                              (uint32_t)cv.vmk, (uint32_t)cv.list, (uint32_t)(reinterpret_cast<unsigned char*>(misc) - smem), hintm,
                                                              r0_direct ? 1u : 0u, (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6), nth >> 6});
        badm |= r & 0xffffu; oobm |= r >> 16;
      } else {
        if (r0_direct && tid == 0) misc[N_R0OVF] = 1u;  // (a ragged tile: round 0's list comes from the scan of the marks)
        for (uint32_t t = tid; t < tasks; t += nth) {
          const uint32_t b = t / SQ, q = t - b * SQ, v0 = 4 * q, cnt = min(4u, V - v0);
          const uint32_t* const rp = rows + (size_t)misc[N_NID + b] * V + v0;
          uint32_t nib = 0;
          bool bad = false, oob = false;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if ((uint32_t)i < cnt) {
              const uint32_t c = rp[i], s = (c + (c >> 16)) & 0xffffu;
              dom[rowof(v0 + i) + b] = c;
              nib |= s == 0u ? 1u << i : 0u;
              bad |= (s & 0x8000u) != 0u;
              oob |= (pk_add16(c, 0x40004000u) & 0x80008000u) != 0u;
            }
          if (a.seed_always) nib |= (a.seed_always[v0 >> 5] >> (v0 & 31u)) & ((1u << cnt) - 1u);
          if ((hintm >> b) & 1u) nib = 0;
          if (nib) atomicOr(&chg[b * Wv + (v0 >> 5)], nib << (v0 & 31u));
          if (oob) oobm |= 1u << b; else if (bad) badm |= 1u << b;
        }
      }
    }
  } else if (vec) {
    // ALL of a tile's row loads in flight at once where the registers allow (16 nodes of 1000 variables on 512 threads: eight
    // 16-byte pairs per lane = 64 VGPRs): one memory round trip per tile instead of two in a row
    constexpr int UF = PACKED ? 8 : 6;  // (the int2-cell instantiations have fewer registers to spare)
    // task t = (node t / SQ, quad t % SQ); a lane's tasks are nth apart: one division per lane, then (node, quad) move by a fixed step
    const uint32_t dq = nth % SQ, db = nth / SQ;
    // (opaque per tile: everything below depends on the thread index alone, and hoisted out of the tile loop it would sit in two dozen
    // registers — spilled — for the whole kernel)
    const uint32_t tid_o = tid;
    uint32_t bs = tid_o / SQ, qs = tid_o - bs * SQ;
    auto step = [&](uint32_t& bq, uint32_t& qq) { qq += dq; bq += db; if (qq >= SQ) { qq -= SQ; ++bq; } };
    // (a tile's rows are contiguous: buffer loads — a descriptor of the tile's rows in SGPRs and ONE 32-bit byte offset per
    // pair of loads, which lb and ub share; sixteen 64-bit addresses would take 32 of the VGPRs the loaded rows need)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs_lb = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.lb_in + (size_t)node0 * V), 0, (int)(nb * V * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ub = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(a.ub_in + (size_t)node0 * V), 0, (int)(nb * V * 4u), 0x00020000);
    if (BT >= 16) {
      // 16-node tiles.  A wavefront's task = FOUR nodes x SIXTEEN consecutive quads (lanes 0-15 node 4g, 16-31 node 4g+1, ...): the loads are
      // still 256 contiguous bytes per row, and the 64 cells a wavefront writes per store — word 68 q + b, q = 16 consecutive, b = 4 consecutive —
      // fall on all 32 LDS banks, two lanes each.  (64 consecutive quads of ONE node, the obvious mapping, put word 68 q + b on 8 banks: every
      // ds_write of the staging loop ran 8-way conflicted — most of the launch's SQ_LDS_BANK_CONFLICT cycles.)
      const uint32_t QC = (SQ + 15u) >> 4;
      const uint32_t lane_o = tid_o & 63u, lb4 = lane_o >> 4, lq = lane_o & 15u;
      const uint32_t wv_s = __builtin_amdgcn_readfirstlane(tid_o >> 6), nwv_s = nth >> 6;
      const uint32_t dqc = nwv_s % QC, dng = nwv_s / QC;
      const uint32_t ng_first = wv_s / QC, qc_first = wv_s - ng_first * QC;  // (wave-uniform: scalar registers)
      const uint32_t ro_lane = lb4 * V * 4u + 16u * lq;
      // UF wave-tasks from (w0; ng, qc) of a tile of tnb nodes behind the descriptors (r_lb, r_ub): the loads / the cells
      auto loadw = [&](const __amdgpu_buffer_rsrc_t r_lb, const __amdgpu_buffer_rsrc_t r_ub, uint32_t tnb, uint32_t w0, uint32_t ng, uint32_t qc, int4 (&L)[UF], int4 (&U)[UF]) {
        const uint32_t wt = ((tnb + 3u) >> 2) * QC;
#pragma unroll
        for (int j = 0; j < UF; ++j) {
          const uint32_t b = 4u * ng + lb4, q = 16u * qc + lq;
          const bool on = w0 + j * nwv_s < wt && q < SQ && b < tnb;
          const uint32_t off = on ? ro_lane + ng * (16u * V) + qc * 256u : 0u;
          const u32x4 lv = __builtin_amdgcn_raw_buffer_load_b128(r_lb, (int)off, 0, 0), uv = __builtin_amdgcn_raw_buffer_load_b128(r_ub, (int)off, 0, 0);
          L[j] = make_int4((int)lv.x, (int)lv.y, (int)lv.z, (int)lv.w);
          U[j] = make_int4((int)uv.x, (int)uv.y, (int)uv.z, (int)uv.w);
          qc += dqc; ng += dng;
          if (qc >= QC) { qc -= QC; ++ng; }
        }
      };
      const uint32_t wtasks = ((nb + 3u) >> 2) * QC;
      auto putw = [&](const int4 (&L)[UF], const int4 (&U)[UF], uint32_t w0, uint32_t& ng, uint32_t& qc) {
#pragma unroll
        for (int j = 0; j < UF; ++j) {
          if (w0 + j * nwv_s >= wtasks) break;  // (uniform)
          const uint32_t b = 4u * ng + lb4, q = 16u * qc + lq;
          if (q < SQ && b < nb) {
            const int l[4] = {L[j].x, L[j].y, L[j].z, L[j].w}, u[4] = {U[j].x, U[j].y, U[j].z, U[j].w};
            note(put(b, 4 * q, l, u, 4), b);
          }
          qc += dqc; ng += dng;
          if (qc >= QC) { qc -= QC; ++ng; }
        }
      };
      auto store_adj = [&]() {
        if (!adj_stored) {
          adj_stored = true;
#pragma unroll
          for (int j = 0; j < 4; ++j) { const uint32_t v = tid + j * nth; if (v <= V) adjo[v] = adj_pre[j]; }
        }
      };
      uint32_t ngs = ng_first, qcs = qc_first;
      if (PACKED && BT == 16 && nb == 16u && !(a.debug & 32768u)) {
        // a full tile: the lean loop (stage_tile16), out of line
        store_adj();
        const uint32_t r = stage_tile16<false>(StageTile16Args{a.lb_in + (size_t)node0 * V, a.ub_in + (size_t)node0 * V, a.seed_always, V, Wv, (uint32_t)cv.dom, (uint32_t)cv.chg,
                                                        (uint32_t)cv.vmk, (uint32_t)cv.list, (uint32_t)(reinterpret_cast<unsigned char*>(misc) - smem), hintm,
                                                        r0_direct ? 1u : 0u, wv_s, nwv_s});
        badm |= r & 0xffffu; oobm |= r >> 16;
      } else
      for (uint32_t w0 = wv_s; w0 < wtasks; w0 += UF * nwv_s) {
        int4 L[UF], U[UF];
        loadw(rs_lb, rs_ub, nb, w0, ngs, qcs, L, U);
        store_adj();
        putw(L, U, w0, ngs, qcs);
      }
    } else
    for (uint32_t t0 = tid_o; t0 < tasks; t0 += UF * nth) {
      int4 L[UF], U[UF];
      uint32_t bq = bs, qq = qs;
      uint32_t ro = bs * V * 4u;
#pragma unroll
      for (int j = 0; j < UF; ++j) {
        const uint32_t off = t0 + j * nth < tasks ? ro + 16u * qq : 0u;  // < 16 rows * 4 bytes * n_vars: 32 bits are plenty
        const u32x4 lv = __builtin_amdgcn_raw_buffer_load_b128(rs_lb, (int)off, 0, 0), uv = __builtin_amdgcn_raw_buffer_load_b128(rs_ub, (int)off, 0, 0);
        L[j] = make_int4((int)lv.x, (int)lv.y, (int)lv.z, (int)lv.w);
        U[j] = make_int4((int)uv.x, (int)uv.y, (int)uv.z, (int)uv.w);
        // the row's byte offset moves with the node by additions (a 32-bit multiply per load pair is a quarter-rate instruction)
        qq += dq; bq += db; ro += db * V * 4u;
        if (qq >= SQ) { qq -= SQ; ++bq; ro += V * 4u; }
      }
      if (!adj_stored) {
        adj_stored = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint32_t v = tid + j * nth; if (v <= V) adjo[v] = adj_pre[j]; }
      }
      bq = bs; qq = qs;
#pragma unroll
      for (int j = 0; j < UF; ++j) {
        if (t0 + j * nth >= tasks) break;
        const int l[4] = {L[j].x, L[j].y, L[j].z, L[j].w}, u[4] = {U[j].x, U[j].y, U[j].z, U[j].w};
        note(put(bq, 4 * qq, l, u, 4), bq);
        step(bq, qq);
      }
      bs = bq; qs = qq;
    }
  } else {
    for (uint32_t t = tid; t < tasks; t += nth) {
      const uint32_t b = t / SQ, q = t - b * SQ, v0 = 4 * q, cnt = min(4u, V - v0);
      const size_t row = (size_t)misc[N_NID + b] * V;
      int l[4] = {0, 0, 0, 0}, u[4] = {0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if ((uint32_t)i < cnt) {
          if constexpr (DFS) {
            l[i] = __hip_atomic_load(a.lb_in + row + v0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            u[i] = __hip_atomic_load(a.ub_in + row + v0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            l[i] = a.lb_in[row + v0 + i]; u[i] = a.ub_in[row + v0 + i];
          }
        }
      note(put(b, v0, l, u, cnt), b);
    }
  }
  // interned constants: singleton pseudo-variables behind the variables (term/constant.rs:43-68)
  for (uint32_t t = tid; t < nb * (S - V); t += nth) {
    const uint32_t b = t / (S - V), s = V + (t - b * (S - V));
    const int c = a.m.const_val[s - V];
    if constexpr (PACKED) dom[rowof(s) + b] = pack16(c, c); else dom[rowof(s) + b] = make_int2(-c, c);
  }
  if (badm) atomicOr(&misc[N_FAIL], badm);
  if (oobm) atomicOr(&misc[N_OOB], oobm);
  if (hintm && tid < nb && ((hintm >> tid) & 1u)) {  // the hinted nodes' one changed variable
    const uint32_t dv = DFS ? dfs_hint : a.dirty[node0 + tid];
    atomicOr(&chg[tid * Wv + (dv >> 5)], 1u << (dv & 31u));
  }
}

// DFS = true: ONE workgroup runs the reference's search loop itself (pcp_dfs_device) — OneSolution / AllSolution over
// Propagation<Brancher<FirstSmallestVar, MiddleVal, BinarySplit>> on a VectorStack (search/mod.rs:45-52, one_solution.rs:92-105),
// under StopNode (stop_node.rs:47-62): up to a.dfs.n_steps nodes per launch, each popped from the device stack, propagated, counted
// and branched in place (the right child x > v over the parent's row, the left child x <= v on top: left first, one_solution.rs:46-51).
// The left child is the next node and its domains are already in LDS — the parent's fixpoint with one bound moved — so it is neither
// staged again nor swept again: the only variable whose lists must run is the one branched on (Store::react's argument: at the
// parent's fixpoint every propagator is a no-op until one of its variables changes).  A node popped from the stack is staged from
// global memory and swept in full, like any node handed in by a caller.
// BT = the tile size as a compile-time constant — 16 (the batch default), 1 (the search loop) — or 0: taken from the launch.  With it the
// cell index of (slot, node) is shifts and immediates; a run-time tile size costs a multiplication per access and a handful of SGPRs the
// kernel does not have (it spills scalars into VGPR lanes as it is).
// CELLS: the bounds rows in HBM are rows of packed cells (pcp_device_batch.cell_format PCP_CELLS_PACKED16; PACKED batch launches only).
// TICKETS: the launch has more tiles than `tile_static` per workgroup and deals the rest by tickets (NeqArgs::tile_ctr != null).  A template flag,
// not a run-time test: the ticket's registers cost the 16 384-node headline launch — which never draws — 16 B of scratch per lane in round 5.
template <bool PACKED, bool PAY4, bool DFS, int BT, bool CELLS = false, bool TICKETS = false>
__global__ void __launch_bounds__(DFS ? 512 : 1024) __attribute__((amdgpu_waves_per_eu(4))) neqfix_kernel(const NeqArgs a_in) {  // (four wavefronts per SIMD: the forest runs 16 per CU)
  NeqArgs a = a_in;
  a.stats += blockIdx.x & (kStatSlots - 1);
  if (a.dbg) a.dbg += (size_t)(blockIdx.x & (kStatSlots - 1)) * PCP_DBG_COUNT;
  if (!DFS && a.sp_ptr) {  // host-stepped device-side DFS: the node on top of the stack
    const uint32_t sp = *a.sp_ptr;
    if (sp == 0 || *a.stop_ptr) return;
    const size_t off = (size_t)(sp - 1) * a.m.n_vars;
    a.lb_in += off; a.ub_in += off; a.lb_out += off; a.ub_out += off; a.status += sp - 1;
  }
  if constexpr (!DFS) {
    // Two workgroups share a CU and start together: they stage together (every CU of the chip at once: HBM saturated), then compute
    // together (HBM idle).  The one in the CU's second pair of wave slots may start late, so that one streams while the other computes.
    if (a.stagger) {
      uint32_t hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      if ((hw & 15u) >= 2u) {
        const uint64_t t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (uint64_t)a.stagger) __builtin_amdgcn_s_sleep(16);
      }
    }
  }
  using Cell = typename NeqCell<PACKED>::type;
  using TDom = typename TileDomOf<PACKED>::type;
  using Pay = typename std::conditional<PAY4, uint32_t, uint2>::type;
  const Pay* pay;
  if constexpr (PAY4) pay = a.adjp4; else pay = a.m.adjp;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t tid = threadIdx.x, lane = tid & 63;  // (refreshed, opaquely, at the top of every tile: see the loop)
  const uint32_t nth = blockDim.x;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = nth >> 6;
  const uint32_t V = a.m.n_vars, S = a.m.n_slots, Wv = (S + 31) >> 5, B = BT ? (uint32_t)BT : a.nodes_per_block;
  auto bar = [&]() { __syncthreads(); };
  const NeqCarve cv = neq_carve(S, V, B, PACKED, a.lds_wgs);
  // profiling: per-wavefront event stamps behind the windows (16 stamps of 8 bytes per wavefront = four windows' worth each: with 512 threads the
  // masks of round 0's direct list — they borrow the window area — still fit in front of them, so the traced launch runs what the product runs)
  const bool tr_on = PCP_NEQ_PROFILE && !DFS && a.trace != nullptr && cv.wcap >= 4u * nwv;
  const uint32_t sh = BT >= 16 ? 2u : BT == 1 ? 6u : cv.sh, wcap = tr_on ? cv.wcap - 4u * nwv : cv.wcap;
  unsigned long long* const trbuf = reinterpret_cast<unsigned long long*>(smem + cv.win + (size_t)wcap * sizeof(Win));
  PCP_TR(0);
  auto rowof = [&](uint32_t slot) { return neq_row(slot, B, sh); };  // index of node 0's cell of a slot
  Cell* const dom = reinterpret_cast<Cell*>(smem + cv.dom);
  uint32_t* const chg = reinterpret_cast<uint32_t*>(smem + cv.chg);
  uint4* const list = reinterpret_cast<uint4*>(smem + cv.list);  // (v | M << 16, list offset, degree, windows w0 | w1 << 16)
  uint32_t* const vmk = reinterpret_cast<uint32_t*>(smem + cv.vmk);
  Win* const win = reinterpret_cast<Win*>(smem + cv.win);
  uint32_t* misc = reinterpret_cast<uint32_t*>(smem + cv.misc);  // (the copy of the current tile: see the loop's end)
  const uint32_t n_eff = a.n_nodes;
  // PERSISTENT tiles: workgroup g runs the tiles g, g + gridDim.x, ... (the host launches at most as many workgroups as fit the chip at
  // once).  The kernel's entry (arguments, the lists' offsets) is paid once per workgroup, not per tile, and nothing drains between
  // a tile's last barrier and the next tile's first loads.
  const uint32_t n_tiles = DFS ? 1u : (n_eff + B - 1) / B;
  uint32_t tile = blockIdx.x;
  if (!DFS && tile >= n_tiles) return;
  uint32_t node0 = DFS ? 0u : tile * B;
  uint32_t nb = DFS ? 1u : min(B, n_eff - node0);
  auto dom_of = [&](uint32_t b, Ctr* c) { return TDom{dom + b, B, sh, chg + (size_t)b * Wv, misc, 1u << b, c}; };

  // ---- phase 0: stage the domains (16-byte row loads), find the assigned variables ------------------------------------------
  // the lists' offsets: an LDS copy (the build pass of a round then has no global load in its chain)
  uint32_t* const adjo = reinterpret_cast<uint32_t*>(smem + cv.adj);
  // (the first four offsets per thread are only LOADED here and stored behind the staging loads below: one memory round trip
  // for both instead of two in a row)
  const bool ptime = PCP_NEQ_PROFILE && (a.debug & 32u) != 0;  // profiling: s_memtime ticks per phase, summed over workgroups into the counters
  uint64_t pt0 = ptime ? __builtin_amdgcn_s_memtime() : 0;
  uint64_t rt0 = ptime ? __builtin_amdgcn_s_memrealtime() : 0;  // (100 MHz: the launch's timeline across workgroups)
  uint64_t pt1 = 0, pt2 = 0, pt3 = 0, pta = 0, ptb = 0, ptc = 0;
  // The copy goes straight into LDS (global_load_lds: LDS-DMA, no registers, nobody waits for it until the first tile's staging barrier, whose
  // __syncthreads drains it): requested here, at the kernel's entry, its round trip — every workgroup of the launch asks for the same 4 KB at the
  // same moment — runs under the cold start and the first tile's row loads.  (Through registers, stored behind the first zeroing barrier, the
  // slowest workgroups waited 6 000 ticks for it there; held across the staging instead they cost the kernel a spilled register.)
  for (uint32_t base = 0; base <= V; base += nth) {
    const uint32_t v = base + tid;
    auto* const dst = (__attribute__((address_space(3))) void*)(adjo + base + 64u * wv);  // (wave-uniform: lane i's dword lands at dst + 4 i)
    if (v <= V) __builtin_amdgcn_global_load_lds(a.m.adj_off + v, dst, 4, 0, 0);
  }
  uint32_t adj_pre[4] = {0u, 0u, 0u, 0u};
  bool adj_stored = true;  // (the register path of earlier rounds: its stores are dead code now)
  // DFS: the stack pointer and the stop flag live in registers for the launch (every thread keeps the same copy)
  NeqDfsRegs dfs{0u, 0u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0ull, 0ull, 0ull, 0u};
  // DFS: workgroup t searches tree t — its own stack rows, stack pointer, stop word, counters and first solution (pcp_dfs_device is
  // the forest of one tree; pcp_dfs_forest_device launches many, each an independent instance of the same loop)
  const size_t tree_row0 = DFS ? (size_t)blockIdx.x * a_in.dfs.capacity : 0;
  if constexpr (DFS) {
    a.dfs.sp += blockIdx.x; a.dfs.stop += blockIdx.x; a.dfs.counters += (size_t)blockIdx.x * 5;
    if (a.dfs.first_solution) a.dfs.first_solution += (size_t)blockIdx.x * V;
    if (a.dfs.dirty) a.dfs.dirty += tree_row0;
  }
  unsigned long long acc_steps = 0, acc_narrow = 0, acc_ev = 0, acc_full = 0, acc_waves = 0, acc_nodes = 0, acc_failed = 0;  // DFS: pcp_stats, per launch
  if constexpr (DFS) { dfs.sp = *a.dfs.sp; dfs.stop = *a.dfs.stop; dfs.nodes = a.dfs.counters[0]; dfs.sols = a.dfs.counters[1]; dfs.fail = a.dfs.counters[2]; }
  const int lim = PACKED ? kPackedMax : kBoundMax;
  uint32_t* const misc_base = misc;
  // a persistent workgroup's counters over its tiles: seven u64 in LDS behind the two copies of the status words, touched by thread 0 only
  // (in registers they cost the tile loop fourteen VGPRs it does not have)
  unsigned long long* const accl = reinterpret_cast<unsigned long long*>(misc_base + 2 * 48);
  // ... and, behind the next-tile word, three diagnostic counts of the lean round 0 (pcp_debug_counters): tiles that took it, passes beyond the
  // first, tiles it handed to the general rounds — thread 0 only
  uint32_t* const lean_ctr = misc_base + 2 * 48 + 16;
  if (!DFS && tid == 0) { for (int i = 0; i < 7; ++i) accl[i] = 0ull; lean_ctr[0] = lean_ctr[1] = lean_ctr[2] = 0u; }
  // A persistent workgroup's per-thread counters (narrowings, pairs tested, full filter runs, wake-ups) are carried over its tiles in
  // registers and added up ONCE, behind the last tile: the wave sums, the LDS atomics and the barrier they needed were 1 500 cycles of every
  // tile.  (Not in profiling builds, whose timers read the per-tile words.)
  constexpr bool DEFER = !DFS && !PCP_NEQ_PROFILE && PAY4;  // (the 8-byte-payload instantiations have no registers to spare: they spilled)
  uint32_t tot_narrow = 0, tot_ev = 0, tot_full = 0, tot_later = 0;
  // (a counter nobody touched — narrowings and full filter runs of a frontier tile — costs one ballot; the high halves likewise)
  auto total = [&](uint32_t x) -> unsigned long long {
    if (!__ballot(x != 0u)) return 0ull;
    const unsigned long long lo = wave_sum(x & 0xffffu);
    return __ballot((x >> 16) != 0u) ? lo + ((unsigned long long)wave_sum(x >> 16) << 16) : lo;
  };
  uint32_t dfs_it = 0;
  for (;; ++dfs_it) {  // DFS: the search loop's nodes; otherwise this workgroup's tiles
  bool resume = false;
  if constexpr (!DFS) {
    misc = misc_base + (dfs_it & 1u) * 48u;
    // the thread index, made opaque per tile: what depends on it alone (lane masks, task coordinates, cell addresses) is then computed
    // where it is used instead of being hoisted out of the tile loop into registers that stay occupied — and spill — for the whole kernel
    asm volatile("" : "+v"(tid));
    lane = tid & 63u;
  }
  if constexpr (DFS) {
    if (dfs.sp == 0 || dfs.stop || dfs_it >= a.dfs.n_steps) break;
    const size_t off = (tree_row0 + (dfs.sp - 1)) * V;
    a.lb_in = a_in.lb_in + off; a.ub_in = a_in.ub_in + off; a.lb_out = a_in.lb_out + off; a.ub_out = a_in.ub_out + off; a.status = a_in.status + tree_row0 + (dfs.sp - 1);
    resume = dfs.resume_var != 0xFFFFFFFFu;
    // a popped row: the variable it was branched on, if the stack keeps them (pcp_dfs_state.dirty) — it is a propagated parent with that
    // one variable moved, so its first round is that variable's lists (the rows of this very launch are read past the L1, like the bounds)
    dfs.hint = (!resume && a.dfs.dirty) ? __hip_atomic_load(a.dfs.dirty + (dfs.sp - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
  }
  if (tid < (uint32_t)N_WORDS) misc[tid] = 0;
  if (tid == (uint32_t)N_R0OVF || (tid >= (uint32_t)N_VOTE && tid < (uint32_t)N_VOTE + 3u)) misc[tid] = 0;
  // Round 0's list — the assigned variables of the tile's nodes, each with the mask of the nodes it is assigned in — is built BY the staging
  // loop where it finds a singleton (rare branch of put): the mask in vmk, the first node to see a variable appends it.  The ballot scan over
  // the marks that used to build it was 3 000 of a frontier tile's 32 000 cycles, for one listed variable.
  const bool r0_direct = !DFS && BT >= 16 && (V & 3u) == 0 && (((size_t)a.lb_in | (size_t)a.ub_in) & 15u) == 0 && !(a.debug & 16384u) &&
                         (size_t)wcap * sizeof(Win) >= (((size_t)S + 1) / 2) * 4 &&  // (the masks borrow the window area)
                         a.dirty == nullptr;                                          // (hinted batches are deep nodes: their round 0 comes from the hints)
  if (r0_direct) for (uint32_t i = tid; i < (S + 1u) / 2u; i += nth) vmk[i] = 0;
  if (tid < nb) misc[N_NID + tid] = node0 + tid;
  if constexpr (!DFS) {
    // Hinted nodes (pcp_device_batch.dirty_var): the row is a fixpoint of this model but for ONE variable — a child of a propagated node.
    // At a fixpoint every propagator is a no-op until one of its variables changes (Store::react, store.rs:191-198), so the node's first
    // round is that variable's lists instead of the lists of all its assigned variables: what the in-kernel search loop (DFS) does for a
    // left child.  One wavefront, one lane per node of the tile; the variable is fetched again behind the barrier by the same lanes.
    if (a.dirty && tid < 64u) {
      const uint32_t dv = tid < nb ? a.dirty[node0 + tid] : 0xFFFFFFFFu;
      const unsigned long long hm = __ballot(dv < V);
      if (tid == 0) misc[N_HINT] = (uint32_t)hm;
    }
  }
  for (uint32_t i = tid; i < B * Wv; i += nth) chg[i] = 0;
  // (only LDS words were written: a batch tile's barrier here waits for LDS alone, so that the offsets' DMA stays in flight; the search loop's
  // rows may have been written by this very workgroup a moment ago: its barrier also waits for those stores)
  if constexpr (!DFS) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else bar();
  PCP_TR(1);
  if (!adj_stored) {
    adj_stored = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const uint32_t v = tid + j * nth; if (v <= V) adjo[v] = adj_pre[j]; }
    for (uint32_t v = tid + 4 * nth; v <= V; v += nth) adjo[v] = a.m.adj_off[v];
  }
  // DFS rows may have been written by this very workgroup a moment ago: they are read past the L1 (relaxed agent-scope loads)
  const bool vec = !DFS && (V & 3u) == 0 && (((size_t)a.lb_in | (size_t)a.ub_in) & 15u) == 0;
  if (resume) {
    if (tid == 0) chg[dfs.resume_var >> 5] = 1u << (dfs.resume_var & 31u);  // the left child: only the variable branched on has changed
  } else {
    neq_stage_tile<PACKED, DFS, BT, CELLS>(NeqTile<PACKED>{dom, chg, misc, adjo, list, win, V, Wv, B, sh, nb, tid, lane, wv, nwv, nth}, a, vmk, cv, smem, S, node0, r0_direct, vec, dfs.hint,
                                    adj_pre, adj_stored);
  }
  if (!adj_stored) {
    adj_stored = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const uint32_t v = tid + j * nth; if (v <= V) adjo[v] = adj_pre[j]; }
  }
  PCP_TR(2);
  bar();
  PCP_TR(3);
  if (ptime) pt1 = __builtin_amdgcn_s_memtime();
  if (misc[N_OOB] && tid == 0) atomicMax(a.violation, 1u);  // sticky: reported by pcp_stats_read

  uint32_t ticket = 0;
  // Tiles of unequal cost on persistent workgroups: a workgroup's first `tile_static` tiles are blockIdx.x + k gridDim.x, the later ones come
  // from a ticket — one of eight (blockIdx.x & 7: the tiles behind the static ones are dealt to the eight residues, so that 512 workgroups
  // do not queue on one address: same-address device atomics serialise, and when every workgroup drew from one word in the launch's
  // lockstep first generation the last of them waited ~8 us).  The ticket of the tile after next is drawn behind this tile's staging by the last
  // wavefront's first lane (of a frontier tile's twelve pieces that wavefront walks one, not two) and published behind the status scan; the
  // barrier there makes it visible, and the word's last reader (the end of the previous tile) is three barriers back.  (Drawn in front of the row
  // loads and kept across the staging it cost the headline instantiation 16 B of scratch.)
  // (the two arguments are read from the kernel-argument segment where they are used — volatile scalar loads, three per tile — instead of being
  // held for the whole kernel in scalar registers it does not have: held, they cost the headline instantiation 16 B of scratch per lane)
  auto draws = [&]() { if constexpr (DFS || !TICKETS) return false; else return dfs_it + 1u >= neq_karg<uint32_t>(offsetof(NeqArgs, tile_static)); };
  if constexpr (!DFS && TICKETS) { if (tid == nth - 64u && draws()) ticket = atomicAdd(neq_karg<uint32_t*>(offsetof(NeqArgs, tile_ctr)) + 32u * (blockIdx.x & 7u), 1u); }
  // ---- rounds: round 0 = the lists of the assigned variables (the sweep), round r = the lists of the changed variables ------
  Ctr ctr;
  uint32_t ev0 = 0;  // item tests of round 0 (they stand for the sweep: counted as evaluated, not as extra steps)
  const uint32_t U4 = 4;
  const bool one_piece = a.m.max_deg <= 64u * U4;
  bool cascade = false;  // the round before narrowed in many threads at once (workgroup-uniform)
  // ---- round 0 of a frontier tile, lean (neq_fast_walk): a full tile whose nodes have few assigned variables between them -------------------
  uint32_t fstate = 0;  // (workgroup-uniform) 0 / 1: the general rounds from round 0 / 1;  2: the rounds are done, the statuses are not;  3: both are
  if constexpr (!DFS && BT == 16 && PACKED && PAY4) {
    if (r0_direct && nb == 16u && a.seed_always == nullptr && !(a.debug & (131072u | 1u | 4u)) && a.m.max_deg <= kFastPer * nth) {
      const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane(misc[N_COUNT0]);
      if (cnt <= kFastLists && (uint32_t)__builtin_amdgcn_readfirstlane(misc[N_R0OVF]) == 0u) {
        const NeqTile<PACKED> tf{dom, chg, misc, adjo, list, win, V, Wv, B, sh, nb, tid, lane, wv, nwv, nth};
        PCP_TR(4); PCP_TR(5); PCP_TR(6);
        bool woke = false;
        uint32_t my_ev = 0;
        // the lists' payload first (one memory round trip, kept in registers for every pass); behind it the status scan's own round trip — the
        // statuses are taken BEFORE the round's one barrier: valid iff the tile narrowed nothing at all (else the nodes that narrowed are
        // scanned again below, on their final domains)
        const bool no_status = (a.debug & 2u) != 0;
        uint32_t q0[kFastPer] = {0u, 0u, 0u, 0u, 0u, 0u};
        static_assert(kFastPer == 6, "q0's initialiser");
        if (cnt) neq_fast_load(tf, pay, 0u, q0);
        StatusPre sp{0u, 0, 0, 4u};
        if (!no_status) sp = neq_status_issue(tf, pay);
        // pass 0: every node; a later pass — there is one only when something narrowed, in a handful of tiles per frontier launch — tests the same
        // lists again for the nodes that narrowed, until a pass narrows nothing; a pass that assigned or emptied a variable hands the tile to the
        // general rounds (the variable's own list must run)
        uint32_t only = 0xffffu & ~(misc[N_FAIL] | misc[N_OOB]);
        for (uint32_t pass = 0;; ++pass) {
          woke = false; my_ev = 0;
          bool more = false;
          const bool nar = neq_fast_test(tf, pay, vmk, cnt, only, q0, ctr, my_ev, woke, more);
          ctr.ev += my_ev;
          if (pass == 0) {
            ev0 += my_ev;
            PCP_TR(7);
            if (!no_status && neq_status_finish(tf, sp)) neq_status_scan<PACKED, DFS>(tf, pay, false);
            if constexpr (TICKETS) { if (tid == nth - 64u && draws()) misc_base[kNextTileWord] = neq_karg<uint32_t>(offsetof(NeqArgs, tile_static)) * gridDim.x + 8u * ticket + (blockIdx.x & 7u); }
            PCP_TR(10);
          }
          // the pass's vote: one word, one barrier that waits for LDS only (__syncthreads_count is a reduction through LDS and two barriers)
          const uint32_t vw = (uint32_t)N_VOTE + pass % 3u;
          const uint32_t bits = (nar ? 1u : 0u) | (woke ? 2u : 0u) | (more ? 4u : 0u);
          if (bits) atomicOr(&misc[vw], bits);
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          const uint32_t vote = (uint32_t)__builtin_amdgcn_readfirstlane(misc[vw]);
          if (pass == 0) { PCP_TR(8); PCP_TR(9); PCP_TR(11); } else { PCP_TR(4); }  // (profiling: a later pass re-uses stamp 4 — "pass voted")
          if (tid == 0) { if (pass == 0) ++lean_ctr[0]; else ++lean_ctr[1]; }
          if (vote == 0u) {
            if (pass == 0) { fstate = 3; break; }
            // quiet again: the nodes that moved get their statuses from their final domains — from the entries requested for them before
            // pass 0 where those still decide, else from the full scan
            const uint32_t moved = misc[N_DIRTY];
            if (tid == 0) atomicAnd(&misc[N_UNK], ~moved);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (!no_status && neq_status_finish(tf, sp, moved, true)) neq_status_scan<PACKED, DFS>(tf, pay, false, moved);
            PCP_TR(5);  // (profiling, tiles that narrowed: statuses done)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            PCP_TR(6);
            fstate = 3;
            break;
          }
          if (tid == 0) misc[(uint32_t)N_VOTE + (pass + 2u) % 3u] = 0;  // (the word of the pass after next: last read before the barrier above)
          if ((vote & 2u) || pass >= 64u) {
            // the general rounds, from round 1: the marks of the narrowings are set; staging's marks of the assigned variables are still set
            // too (their lists run once more: correct, and this is a tile in a thousand)
            fstate = 1;
            if (tid == 0) { misc[N_UNK] = 0; ++lean_ctr[2]; }
            bar();
            break;
          }
          // (only the nodes that narrowed can have anything left to do — unless a lane left a flagged entry untested: then every node again)
          only = ((vote & 4u) ? 0xffffu : misc[N_DIRTY]) & ~(misc[N_FAIL] | misc[N_OOB]);
        }
      }
    }
  }
  if (fstate < 2u)
  for (uint32_t round = fstate;; ++round) {
    if (round >= kMaxRounds) { if (tid == 0) { atomicOr(&misc[N_OOB], nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1u)); atomicMax(a.violation, 1u); } bar(); break; }  // (refused, not hung)
    const uint32_t m_count = (round & 1u) ? N_COUNT1 : N_COUNT0, m_rmask = (round & 1u) ? N_RMASK1 : N_RMASK0, m_win = (round & 1u) ? N_WIN1 : N_WIN0;
    const uint32_t inert = misc[N_FAIL] | misc[N_OOB];
    const uint32_t narrow_before = ctr.narrow;
    // (a0) the cheaper of two covers.  A changed variable that is NOT assigned wakes only propagators that can act if their OTHER
    // side is assigned (x_neq_y.rs:82-93), and those sit in the assigned variables' lists too.  After an assignment near the root
    // ~V variables of a node lose a bound and each would re-walk its whole list to find nothing; the lists of the node's few
    // assigned variables cover the same propagators.  Per node, whichever set of lists is shorter is walked: the node's marks
    // become {assigned variables} + {changed variables with a Constant neighbour: that record is in their own list only}.  Every
    // propagator incident to a changed variable that can act is still evaluated, so the fixpoint is the same; the choice is made
    // again every round, so the tail of a cascade (few changed variables) goes back to the changed lists and their jump windows.
    // One wavefront per node; nothing leaves the wavefront until the barrier.  (-DPCP_NEQ_NO_RESWEEP: A/B builds without it.)
#ifndef PCP_NEQ_NO_RESWEEP
#ifdef PCP_NEQ_RESWEEP_DEAD  // A/B: the code is there, never run
    if (cascade && (a.debug & 64u)) {
#else
    if (cascade) {
#endif
      for (uint32_t b = wv; b < nb; b += nwv)
        if (!((inert >> b) & 1u)) resweep_marks<PACKED>(dom, chg + (size_t)b * Wv, adjo, a.seed_always, V, Wv, B, sh, b, lane);
      bar();
    }
#endif
    // (a) one list for the tile: (variable, mask of the nodes in which it changed) — neq_build_list
    neq_build_list<PACKED>(NeqTile<PACKED>{dom, chg, misc, adjo, list, win, V, Wv, B, sh, nb, tid, lane, wv, nwv, nth}, vmk, round, r0_direct, inert, wcap);
    if (round == 0) PCP_TR(4);
    bar();
    if (round == 0) PCP_TR(5);
    if (ptime && round == 0) pta = __builtin_amdgcn_s_memtime();
    const uint32_t total = min(misc[m_count], kListCap);
    const uint32_t nwin = min(misc[m_win], wcap);
    if (total == 0 || (a.debug & 1u) || ((a.debug & 4u) && round == 1)) break;
    if (tid == 0) {  // the other slots: last read before this round's barrier
      if (round) misc[N_WAVES] += __popc(misc[m_rmask]);
      misc[(round & 1u) ? N_COUNT0 : N_COUNT1] = 0; misc[(round & 1u) ? N_RMASK0 : N_RMASK1] = 0; misc[(round & 1u) ? N_WIN0 : N_WIN1] = 0;
    }
    // (b) walk the lists (neq_walk_lists): every entry of every listed variable against the nodes of its mask
    {
      const uint32_t my_ev = neq_walk_lists<PACKED, PAY4>(NeqTile<PACKED>{dom, chg, misc, adjo, list, win, V, Wv, B, sh, nb, tid, lane, wv, nwv, nth}, a, pay, total, one_piece, round, ctr,
                                                           tr_on, trbuf);
      if (round == 0) ev0 += my_ev;
      ctr.ev += my_ev;
    }
    // (the barrier also answers "did this round narrow anything?": if not — the usual case of a shallow tile's sweep round — no
    // variable is marked and the next round's list pass and barrier are skipped)
    // (the count of narrowing threads also says whether this round was a cascade: only then is the next round's cover priced)
    if (ptime && round == 0) ptb = __builtin_amdgcn_s_memtime();
    if (round == 0) PCP_TR(7);
    const uint32_t n_narrowing = (uint32_t)__syncthreads_count(ctr.narrow != narrow_before);
    if (round == 0) PCP_TR(8);
    if (ptime && round == 0) ptc = __builtin_amdgcn_s_memtime();
    const bool narrowed = n_narrowing != 0;
    cascade = n_narrowing >= kCascadeThreads;
    if (!narrowed && !nwin && misc[N_MORE] != round + 1) break;
    // (c) the jumps: each window's bound moves to the first value no assigned neighbour forbids
    if (nwin) {
      neq_apply_jumps<PACKED>(NeqTile<PACKED>{dom, chg, misc, adjo, list, win, V, Wv, B, sh, nb, tid, lane, wv, nwv, nth}, nwin, ctr, !(a.debug & 65536u));
      bar();
    }
  }

  if (ptime) pt2 = __builtin_amdgcn_s_memtime();
  PCP_TR(9);
  const NeqTile<PACKED> tl{dom, chg, misc, adjo, list, win, V, Wv, B, sh, nb, tid, lane, wv, nwv, nth};
  // ---- status: is any record NOT entailed under the final domains? (store.rs:250-256) ------------------------
  // Records of two assigned variables are entailed at a fixpoint that did not fail (two different values: disjoint), so only the
  // lists of unassigned variables can hold an open record; x != y + d is entailed iff the intervals are disjoint
  // (x_neq_y.rs:71-73 via x_eq_y.rs:87-93).
  // Two nodes per wavefront at a time, one per 32-lane half: a node's scan is a chain of dependent LDS and memory reads (cells -> the
  // list's offsets -> its payload -> the other sides' cells), and a tile of sixteen nodes on eight wavefronts used to run two such
  // chains one after the other in every wavefront.
  if (fstate != 3u) {
    neq_status_scan<PACKED, DFS>(tl, pay, (a.debug & 2u) != 0);
    if constexpr (!DFS && TICKETS) { if (tid == nth - 64u && draws()) misc_base[kNextTileWord] = neq_karg<uint32_t>(offsetof(NeqArgs, tile_static)) * gridDim.x + 8u * ticket + (blockIdx.x & 7u); }
  }

  // ---- write back: the rows of the nodes that changed (every node when the call is not in place) ----------------------------
  uint32_t wb_need = 0;
  if (fstate != 3u) {
    PCP_TR(10);
    bar();
    PCP_TR(11);
  }
  if (ptime) pt3 = __builtin_amdgcn_s_memtime();
  wb_need = neq_write_back<PACKED>(tl, a.lb_in, a.ub_in, a.lb_out, a.ub_out, CELLS, DFS && dfs.stale != 0u);
  if constexpr (DFS) dfs.stale = 0u;
  // the counters: wave sums by DPP (VALU only), then one lane adds them to the tile's LDS words.  (The wave reductions that used to
  // stand here were 24 dependent ds_bpermute round trips, 4 000 cycles of a frontier tile's 45 000; 64-lane LDS atomics on one
  // address were tried instead and cost 8 800.)
  if constexpr (DEFER) {
    tot_narrow += ctr.narrow; tot_ev += ctr.ev; tot_full += ctr.full; tot_later += ctr.ev - ev0;
  } else {
    const unsigned long long s_narrow = total(ctr.narrow), s_ev = total(ctr.ev), s_full = total(ctr.full), s_later = total(ctr.ev - ev0);
    if (lane == 0) {
      if (s_narrow) atomicAdd(&misc[N_NARROW], (uint32_t)s_narrow);
      if (s_ev) atomicAdd(reinterpret_cast<unsigned long long*>(&misc[N_EV]), s_ev);
      if (s_full) atomicAdd(reinterpret_cast<unsigned long long*>(&misc[N_FULL]), s_full);
      if (s_later) atomicAdd(reinterpret_cast<unsigned long long*>(&misc[N_STEPS]), s_later);
    }
  }
  PCP_TR(12);
  // (the write-back may fail a node — an empty cell found on the way out —: the statuses wait for it; a tile that wrote nothing back
  // has nothing to wait for, its words are final since the barrier behind the status scan)
  // (a barrier that waits for LDS only: the rows just written need not have landed — __syncthreads would wait for the stores)
  if (!DEFER || wb_need) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  PCP_TR(13);
  if (tid < nb) {
    const bool failed = (misc[N_FAIL] >> tid) & 1u, refused = (misc[N_OOB] >> tid) & 1u;
    const bool none_open = !((misc[N_UNK] >> tid) & 1u);
    a.status[misc[N_NID + tid]] = refused ? kStatusRetry : failed ? (uint8_t)PCP_FALSE : (none_open ? (uint8_t)PCP_TRUE : (uint8_t)PCP_UNKNOWN);
  }
  if (tid == 0) {
    const uint32_t active_nodes = (uint32_t)__popc(((nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1u)) & ~misc[N_OOB]));
    // reference-equivalent steps: every propagator of every node once (init_scheduler) + every wake-up of the later rounds
    const unsigned long long s2 = (unsigned long long)active_nodes * a.m.n_recs + *reinterpret_cast<unsigned long long*>(&misc[N_STEPS]);
    const unsigned long long sev = *reinterpret_cast<unsigned long long*>(&misc[N_EV]), sfu = *reinterpret_cast<unsigned long long*>(&misc[N_FULL]);
    const uint32_t nf = __popc(misc[N_FAIL] & (nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1u)));
    if constexpr (DFS) {
      // the counters are added up in registers and handed over once per launch: six same-address atomics per node are nothing for one
      // tree and serialise a forest of hundreds (pcp_dfs_forest_device)
      acc_steps += s2; acc_narrow += misc[N_NARROW]; acc_ev += sev; acc_full += sfu; acc_waves += nb + misc[N_WAVES]; acc_nodes += nb; acc_failed += nf;
    } else if (DEFER) {
      // (atomics: the wavefronts that are through with their last tile add their carried counters to the same words)
      atomicAdd(&accl[0], (unsigned long long)active_nodes * a.m.n_recs); atomicAdd(&accl[4], (unsigned long long)(nb + misc[N_WAVES]));
      atomicAdd(&accl[5], (unsigned long long)nb); if (nf) atomicAdd(&accl[6], (unsigned long long)nf);
    } else if (!PCP_NEQ_PROFILE || !(a.debug & (8u | 32u))) {
      accl[0] += s2; accl[1] += misc[N_NARROW]; accl[2] += sev; accl[3] += sfu; accl[4] += nb + misc[N_WAVES]; accl[5] += nb; accl[6] += nf;
    } else {  // (profiling builds of a launch: the counters carry timers, per tile)
      atomicAdd((unsigned long long*)&a.stats->steps, s2);
      if (misc[N_NARROW]) atomicAdd((unsigned long long*)&a.stats->narrowings, (unsigned long long)misc[N_NARROW]);
      if (sev) atomicAdd((unsigned long long*)&a.stats->evaluated, sev);
      if (sfu) atomicAdd((unsigned long long*)&a.stats->full_evals, sfu);
      if (ptime) {
        atomicAdd((unsigned long long*)&a.stats->steps3, (unsigned long long)(pt1 - pt0));        // staging
        atomicAdd((unsigned long long*)&a.stats->failed_nodes, (unsigned long long)(pt2 - pt1));  // rounds
        atomicAdd((unsigned long long*)&a.stats->waves, (unsigned long long)(pt3 - pt2));         // status scan
        const uint64_t pt4 = __builtin_amdgcn_s_memtime();
        atomicAdd((unsigned long long*)&a.stats->full_evals, (unsigned long long)(pt4 - pt3));  // write-back, counters
        if (a.dbg && !((a.debug & 2048u) && blockIdx.x >= gridDim.x / 2) && !((a.debug & 4096u) && blockIdx.x < gridDim.x / 2)) {  // (2048 / 4096: first / second half of the grid only) the same, finer, and the launch's timeline (100 MHz ticks): pcp_debug_counters slots 5..15
          if (pta == 0) pta = ptb = ptc = pt2;  // (the tile never reached round 0's walk)
          if (ptb == 0) ptb = ptc = pt2;
          atomicAdd(&a.dbg[8], pt1 - pt0); atomicAdd(&a.dbg[9], pta - pt1); atomicAdd(&a.dbg[10], ptb - pta); atomicAdd(&a.dbg[11], ptc - ptb);
          atomicAdd(&a.dbg[12], pt2 - ptc); atomicAdd(&a.dbg[13], pt3 - pt2); atomicAdd(&a.dbg[14], pt4 - pt3); atomicAdd(&a.dbg[15], 1ull);
          atomicMax(&a.dbg[5], ~rt0); atomicMax(&a.dbg[6], (unsigned long long)__builtin_amdgcn_s_memrealtime()); atomicMax(&a.dbg[7], rt0);
        }
      }
      if (!(a.debug & (8u | 32u))) atomicAdd((unsigned long long*)&a.stats->waves, (unsigned long long)(nb + misc[N_WAVES]));
      atomicAdd((unsigned long long*)&a.stats->nodes, (unsigned long long)nb);
      if (nf && !(a.debug & (8u | 32u))) atomicAdd((unsigned long long*)&a.stats->failed_nodes, (unsigned long long)nf);
    }
  }
  if constexpr (!DFS) {
    PCP_TR(14);
    if (tr_on && lane < 16) a.trace[((size_t)tile * 16 + wv) * 16 + lane] = lane == 15 ? (unsigned long long)__builtin_amdgcn_s_memrealtime() : trbuf[wv * 16 + lane];
    // the next tile of this workgroup.  No barrier here: the next tile clears the OTHER copy of the status words, its cells are not
    // written before its own first barrier, and every wavefront left the last tile's cells behind the barrier above.
    tile = draws() ? (uint32_t)__builtin_amdgcn_readfirstlane(misc_base[kNextTileWord]) : tile + gridDim.x;
    if (tile >= n_tiles) break;
    node0 = tile * B; nb = min(B, n_eff - node0);
    if (ptime) { pt0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); pta = ptb = ptc = 0; }
    PCP_TR(0);
  } else {
    // ---- the search step on the node just propagated (what dfs_step_kernel does for the generic kernels) ---------------------
    bar();
    neq_dfs_step<PACKED>(NeqTile<PACKED>{dom, chg, misc, adjo, list, win, V, Wv, B, sh, nb, tid, lane, wv, nwv, nth}, a, dfs, dfs_it + 1u >= a.dfs.n_steps);
    bar();
  }
  }  // the DFS loop / the tile loop
  if constexpr (DEFER) {
    const unsigned long long s_narrow = total(tot_narrow), s_ev = total(tot_ev), s_full = total(tot_full), s_later = total(tot_later);
    if (lane == 0) {
      if (s_later) atomicAdd(&accl[0], s_later);
      if (s_narrow) atomicAdd(&accl[1], s_narrow);
      if (s_ev) atomicAdd(&accl[2], s_ev);
      if (s_full) atomicAdd(&accl[3], s_full);
    }
    bar();
  }
  if (tid == 0) {
    if constexpr (!DFS) { acc_steps = accl[0]; acc_narrow = accl[1]; acc_ev = accl[2]; acc_full = accl[3]; acc_waves = accl[4]; acc_nodes = accl[5]; acc_failed = accl[6]; }
    if (acc_steps) atomicAdd((unsigned long long*)&a.stats->steps, acc_steps);
    if (acc_narrow) atomicAdd((unsigned long long*)&a.stats->narrowings, acc_narrow);
    if (acc_ev) atomicAdd((unsigned long long*)&a.stats->evaluated, acc_ev);
    if (acc_full) atomicAdd((unsigned long long*)&a.stats->full_evals, acc_full);
    if (acc_waves) atomicAdd((unsigned long long*)&a.stats->waves, acc_waves);
    if (acc_nodes) atomicAdd((unsigned long long*)&a.stats->nodes, acc_nodes);
    if (acc_failed) atomicAdd((unsigned long long*)&a.stats->failed_nodes, acc_failed);
    if constexpr (DFS) {
      *a.dfs.sp = dfs.sp; *a.dfs.stop = dfs.stop;
      a.dfs.counters[0] = dfs.nodes; a.dfs.counters[1] = dfs.sols; a.dfs.counters[2] = dfs.fail;
      if (dfs.err) a.dfs.counters[3] = dfs.err;
    }
    if constexpr (!DFS && TICKETS) {
      uint32_t* const tile_ctr = neq_karg<uint32_t*>(offsetof(NeqArgs, tile_ctr));
      // every workgroup drew the ticket that ended it before it comes here: the last one to arrive leaves the words zero for the next launch
      __threadfence();
      if (atomicAdd(tile_ctr + 32u * 8u, 1u) == gridDim.x - 1u) { for (uint32_t i = 0; i <= 8u; ++i) atomicExch(tile_ctr + 32u * i, 0u); }
    }
    if (!DFS && a.dbg && !PCP_NEQ_PROFILE) {  // (profiling builds keep phase timers in these slots)
      if (lean_ctr[0]) atomicAdd(&a.dbg[PCP_DBG_NEQ_LEAN], (unsigned long long)lean_ctr[0]);
      if (lean_ctr[1]) atomicAdd(&a.dbg[PCP_DBG_NEQ_LEAN_PASSES], (unsigned long long)lean_ctr[1]);
      if (lean_ctr[2]) atomicAdd(&a.dbg[PCP_DBG_NEQ_LEAN_HANDOVER], (unsigned long long)lean_ctr[2]);
    }
    if (!DFS && a.dbg) atomicAdd(&a.dbg[PCP_DBG_NEQ_TILES], (unsigned long long)(DFS ? 0u : dfs_it + 1u));  // the tiles this workgroup ran (by stride or by ticket)
  }
}

size_t lds_bytes_neq(uint32_t n_slots, uint32_t n_vars, uint32_t nodes_per_block, bool packed, uint32_t wgs) {
  const NeqCarve c = neq_carve(n_slots, n_vars, nodes_per_block, packed, wgs);
  return c.total <= 160 * 1024 ? c.total : 0;
}

template <bool PACKED, bool PAY4, bool DFS, int BT, bool CELLS, bool TICKETS>
static hipError_t launch_neq_t(const NeqArgs& a, const LaunchPlan& p, hipStream_t stream) {
  if (p.lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(neqfix_kernel<PACKED, PAY4, DFS, BT, CELLS, TICKETS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((neqfix_kernel<PACKED, PAY4, DFS, BT, CELLS, TICKETS>), dim3(p.grid), dim3(p.block), p.lds_bytes, stream, a);
  return hipGetLastError();
}
template <bool PACKED, bool PAY4, bool DFS, int BT, bool CELLS = false>
static hipError_t launch_neq_k(const NeqArgs& a, const LaunchPlan& p, hipStream_t stream) {
  if constexpr (!DFS)
    if (a.tile_ctr) return launch_neq_t<PACKED, PAY4, DFS, BT, CELLS, true>(a, p, stream);
  return launch_neq_t<PACKED, PAY4, DFS, BT, CELLS, false>(a, p, stream);
}
template <bool DFS, int BT>
static hipError_t launch_neq_d(const NeqArgs& a, const LaunchPlan& p, hipStream_t stream) {
  if constexpr (!DFS)
    if (a.cell_rows) {
      if (!a.packed) return hipErrorInvalidValue;
      return a.adjp4 ? launch_neq_k<true, true, false, BT, true>(a, p, stream) : launch_neq_k<true, false, false, BT, true>(a, p, stream);
    }
  if (a.adjp4) return a.packed ? launch_neq_k<true, true, DFS, BT>(a, p, stream) : launch_neq_k<false, true, DFS, BT>(a, p, stream);
  return a.packed ? launch_neq_k<true, false, DFS, BT>(a, p, stream) : launch_neq_k<false, false, DFS, BT>(a, p, stream);
}

// ---- int32 bounds rows <-> rows of packed cells (pcp_pack_rows / pcp_unpack_rows): element-wise, four per thread where the rows allow
__global__ void __launch_bounds__(256) pack_rows_kernel(const int32_t* __restrict__ lb, const int32_t* __restrict__ ub, uint32_t* __restrict__ cells, size_t n, uint32_t* violation) {
  bool oob = false;
  auto one = [&](int l, int u) {
    oob |= (l < -kPackedMax) | (l > kPackedMax) | (u < -kPackedMax) | (u > kPackedMax);
    return pack16(max(-kPackedMax, min(kPackedMax, l)), max(-kPackedMax, min(kPackedMax, u)));
  };
  const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if ((((size_t)lb | (size_t)ub | (size_t)cells) & 15u) == 0) {
    for (size_t q = i0; q < (n >> 2); q += stride) {
      const int4 l = reinterpret_cast<const int4*>(lb)[q], u = reinterpret_cast<const int4*>(ub)[q];
      reinterpret_cast<uint4*>(cells)[q] = make_uint4(one(l.x, u.x), one(l.y, u.y), one(l.z, u.z), one(l.w, u.w));
    }
    for (size_t i = (n & ~(size_t)3) + i0; i < n; i += stride) cells[i] = one(lb[i], ub[i]);
  } else {
    for (size_t i = i0; i < n; i += stride) cells[i] = one(lb[i], ub[i]);
  }
  if (oob) atomicMax(violation, 1u);
}
__global__ void __launch_bounds__(256) unpack_rows_kernel(const uint32_t* __restrict__ cells, int32_t* __restrict__ lb, int32_t* __restrict__ ub, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if ((((size_t)lb | (size_t)ub | (size_t)cells) & 15u) == 0) {
    for (size_t q = i0; q < (n >> 2); q += stride) {
      const uint4 c = reinterpret_cast<const uint4*>(cells)[q];
      const int2 a = unpack16(c.x), b = unpack16(c.y), d = unpack16(c.z), e = unpack16(c.w);
      reinterpret_cast<int4*>(lb)[q] = make_int4(a.x, b.x, d.x, e.x);
      reinterpret_cast<int4*>(ub)[q] = make_int4(a.y, b.y, d.y, e.y);
    }
    for (size_t i = (n & ~(size_t)3) + i0; i < n; i += stride) { const int2 d = unpack16(cells[i]); lb[i] = d.x; ub[i] = d.y; }
  } else {
    for (size_t i = i0; i < n; i += stride) { const int2 d = unpack16(cells[i]); lb[i] = d.x; ub[i] = d.y; }
  }
}

// ---- Brancher<FirstSmallestVar, MiddleVal, BinarySplit>::enter (search/branching/brancher.rs:52-71) on a row of cells: branch_kernel
// (pcp_kernels.hip) with the bounds read from and written as cells.  One block per node; child_base[node] = the node's first child slot
// (branch_scan_kernel), 0xFFFFFFFF = not Unknown.
__global__ void __launch_bounds__(256) branch_cells_kernel(uint32_t n_vars, const uint32_t* __restrict__ cells, const uint32_t* __restrict__ child_base,
                                                           uint32_t* __restrict__ child_cells, uint32_t* __restrict__ child_dirty, const uint32_t* __restrict__ counts,
                                                           uint32_t reverse) {
  const uint32_t node = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
  const uint32_t slot = child_base[node];
  if (slot == 0xFFFFFFFFu) return;
  const uint32_t rowL = reverse ? counts[0] - 1 - slot : slot;
  const uint32_t rowR = reverse ? rowL - 1 : slot + 1;
  __shared__ unsigned long long best[4];
  const uint32_t* const p = cells + (size_t)node * n_vars;
  unsigned long long key = ~0ull;  // FirstSmallestVar: minimum of (size << 32 | index) over the variables of size > 1
  for (uint32_t v = tid; v < n_vars; v += nth) {
    const int2 d = unpack16(p[v]);
    const unsigned long long size = (unsigned long long)((long long)d.y - (long long)d.x + 1);
    if (d.y > d.x) key = min(key, (size << 32) | v);
  }
  for (int o = 32; o > 0; o >>= 1) key = min(key, (unsigned long long)__shfl_down(key, o));
  if ((tid & 63) == 0) best[tid >> 6] = key;
  __syncthreads();
  key = best[0];
  for (uint32_t w = 1; w < (nth >> 6); ++w) key = min(key, best[w]);
  const uint32_t var = key == ~0ull ? 0xFFFFFFFFu : (uint32_t)key;
  int32_t val = 0;
  if (var != 0xFFFFFFFFu) { const int2 d = unpack16(p[var]); val = (d.x + d.y) / 2; }  // MiddleVal: `/` truncates toward zero like Rust's
  if (child_dirty && tid == 0) { child_dirty[rowL] = var; child_dirty[rowR] = var; }
  uint32_t* const c0 = child_cells + (size_t)rowL * n_vars;
  uint32_t* const c1 = child_cells + (size_t)rowR * n_vars;
  for (uint32_t v = tid; v < n_vars; v += nth) {
    const uint32_t c = p[v];
    uint32_t l = c, r = c;
    if (v == var) { const int2 d = unpack16(c); l = pack16(d.x, min(d.y, val)); r = pack16(max(d.x, val + 1), d.y); }  // x <= val | x > val
    c0[v] = l; c1[v] = r;
  }
}

hipError_t launch_branch_cells(uint32_t n_nodes, uint32_t n_vars, const uint32_t* cells, const uint32_t* child_base, uint32_t* child_cells, uint32_t* child_dirty,
                               const uint32_t* counts, uint32_t reverse, hipStream_t stream) {
  if (!n_nodes) return hipSuccess;
  hipLaunchKernelGGL(branch_cells_kernel, dim3(n_nodes), dim3(256), 0, stream, n_vars, cells, child_base, child_cells, child_dirty, counts, reverse);
  return hipGetLastError();
}

hipError_t launch_neqfix(const NeqArgs& a, const LaunchPlan& p, hipStream_t stream) {
  if (a.nodes_per_block == 0 || a.nodes_per_block > 16 || a.m.n_slots >= 65536u || !a.m.adjp) return hipErrorInvalidValue;
  if (a.dfs.n_steps) {
    if (a.nodes_per_block != 1 || p.grid < 1 || !a.dfs.sp || !a.dfs.stop || !a.dfs.counters) return hipErrorInvalidValue;  // (grid = trees)
    return launch_neq_d<true, 1>(a, p, stream);
  }
  return a.nodes_per_block == 16 ? launch_neq_d<false, 16>(a, p, stream) : launch_neq_d<false, 0>(a, p, stream);
}

hipError_t launch_pack_rows(const int32_t* lb, const int32_t* ub, uint32_t* cells, size_t n, uint32_t* violation, hipStream_t stream) {
  if (!n) return hipSuccess;
  const uint32_t grid = (uint32_t)std::min<size_t>((n / 4 + 255) / 256 + 1, 4096);
  hipLaunchKernelGGL(pack_rows_kernel, dim3(grid), dim3(256), 0, stream, lb, ub, cells, n, violation);
  return hipGetLastError();
}
hipError_t launch_unpack_rows(const uint32_t* cells, int32_t* lb, int32_t* ub, size_t n, hipStream_t stream) {
  if (!n) return hipSuccess;
  const uint32_t grid = (uint32_t)std::min<size_t>((n / 4 + 255) / 256 + 1, 4096);
  hipLaunchKernelGGL(unpack_rows_kernel, dim3(grid), dim3(256), 0, stream, cells, lb, ub, n);
  return hipGetLastError();
}

}  // namespace pcp
