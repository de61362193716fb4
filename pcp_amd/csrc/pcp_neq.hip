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
//  * one workgroup = a tile of B nodes; their domains sit in LDS NODE-MAJOR (dom[b][slot]: the items of a list touch
//    consecutive slots of ONE node: consecutive banks), as 16-bit packed (-lb, ub) cells under a declared hull within
//    +-16383 (LdsDom16), else as int2 (-lb, ub) (LdsDom); B * S * 4 bytes: two tiles of 16 nodes of N-queens-1000 per CU;
//  * staging is the only HBM traffic: 16-byte row loads, four in flight per lane; a tile that narrows nothing writes
//    nothing back when the call is in place (Store::consistency(&mut vstore) works in place);
//  * a round is VARIABLE-major: the changed / assigned variables of all B nodes are compacted into one list of
//    (variable, node mask) entries; a list is walked in pieces of 4 x 64 entries, payload loads (8 B per entry: other slot,
//    offset) coalesced and issued one piece ahead; each piece decodes its entries ONCE and tests them against every node
//    of the mask: per (entry, node) one ds_read_b32, two v_pk_add_u16 and a v_pk_min_u16 — the filter can act iff
//    lb(v) + t == ub(o) or ub(v) + t == lb(o) (fast_flag's condition, pcp_kernels.hip), i.e. iff a 16-bit half of
//    cell(v) + swap(cell(o)) + (-t, t) is zero; the running unsigned minimum over the nodes is tested once per entry and
//    only flagged entries run the full filter (eval_record: propagate() + is_subsumed() literally, LDS atomics);
//  * status (store.rs:250-256: True iff no subscription remains): at the fixpoint every record of an assigned variable has
//    been run after that variable's last change, so only records of UNASSIGNED variables can be open: one wavefront per
//    node walks them with early exit (an Unknown node shows an open record in its first 64 entries).
// Integer bound work: no MFMA.  Bound by HBM (staging) for shallow nodes, by VALU issue for deep ones.
#include <algorithm>
#include <type_traits>

#include "pcp_device.hpp"
#include "pcp_neq.h"

namespace pcp {

namespace {

enum { N_FAIL = 0, N_OOB = 1, N_COUNT0 = 2, N_COUNT1 = 3, N_RMASK0 = 4, N_RMASK1 = 5, N_DIRTY = 6, N_WAVES = 7, N_UNK = 8, N_NARROW = 9,
       N_EV = 10, N_FULL = 12, N_STEPS = 14, N_WORDS = 16 };

struct NeqCarve {
  size_t dom, chg, list, adj, misc, total;
  uint32_t SP;
};
__host__ __device__ inline NeqCarve neq_carve(uint32_t S, uint32_t V, uint32_t B, bool packed, bool adj_cache) {
  auto up = [](size_t x) { return (x + 15) & ~(size_t)15; };
  NeqCarve c;
  c.SP = (S + 3u) & ~3u;  // rows of whole 16-byte groups
  const size_t Wv = (S + 31) / 32;
  size_t o = 0;
  c.dom = o; o = up(o + (size_t)B * c.SP * (packed ? 4 : 8));
  c.chg = o; o = up(o + (size_t)B * Wv * 4);
  c.list = o; o = up(o + (size_t)S * 4);
  c.adj = o; o = up(o + (adj_cache ? ((size_t)V + 1) * 4 : 0));
  c.misc = o; o = up(o + 32 * 4);
  c.total = o;
  return c;
}

__device__ __forceinline__ bool zero_half(uint32_t u) { return (u & 0xffffu) == 0u || (u >> 16) == 0u; }

// cell(v) + swap(cell(o)) + (-t, t): a half is zero iff lb(v) + t == ub(o) (low) or ub(v) + t == lb(o) (high)
__device__ __forceinline__ uint32_t neq_terms16(uint32_t cv, uint32_t co, uint32_t k) {
  uint32_t u;
  asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
      "v_pk_add_u16 %0, %0, %3"
      : "=&v"(u) : "v"(cv), "v"(co), "v"(k));
  return u;
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t x, uint32_t y) {
  uint32_t r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
// (-t, t) as two int16 halves; |t| beyond the packed range can never meet a sum of two packed bounds: clamp (pack_c0's argument)
__device__ __forceinline__ uint32_t pack_mt(int t) {
  const int tc = max(-32767, min(32767, t));
  return ((uint32_t)(-tc) & 0xffffu) | ((uint32_t)tc << 16);
}

template <bool PACKED> struct NeqCell { using type = int2; };
template <> struct NeqCell<true> { using type = uint32_t; };

template <bool PACKED>
__device__ __forceinline__ int2 cell_bounds(const typename NeqCell<PACKED>::type c) {  // (lb, ub)
  if constexpr (PACKED) return unpack16(c);
  else return make_int2(-c.x, c.y);
}

}  // namespace

template <bool PACKED>
__global__ void __launch_bounds__(1024) neqfix_kernel(const NeqArgs a_in) {
  NeqArgs a = a_in;
  a.stats += blockIdx.x & (kStatSlots - 1);
  if (a.sp_ptr) {  // device-side DFS (pcp_dfs_device): the node on top of the stack
    const uint32_t sp = *a.sp_ptr;
    if (sp == 0 || *a.stop_ptr) return;
    const size_t off = (size_t)(sp - 1) * a.m.n_vars;
    a.lb_in += off; a.ub_in += off; a.lb_out += off; a.ub_out += off; a.status += sp - 1;
  }
  using Cell = typename NeqCell<PACKED>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t tid = threadIdx.x, nth = blockDim.x, lane = tid & 63;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = nth >> 6;
  const uint32_t V = a.m.n_vars, S = a.m.n_slots, Wv = (S + 31) >> 5, B = a.nodes_per_block;
  const NeqCarve cv = neq_carve(S, V, B, PACKED, a.adj_cache != 0);
  const uint32_t SP = cv.SP;
  Cell* const dom = reinterpret_cast<Cell*>(smem + cv.dom);
  uint32_t* const chg = reinterpret_cast<uint32_t*>(smem + cv.chg);
  uint32_t* const list = reinterpret_cast<uint32_t*>(smem + cv.list);
  uint32_t* const misc = reinterpret_cast<uint32_t*>(smem + cv.misc);
  const uint32_t node0 = blockIdx.x * B, nb = min(B, a.n_nodes - node0);

  // ---- phase 0: stage the domains (16-byte row loads), find the assigned variables ------------------------------------------
  const uint32_t* adjo = a.m.adj_off;
  if (a.adj_cache) {
    uint32_t* adj_lds = reinterpret_cast<uint32_t*>(smem + cv.adj);
    for (uint32_t v = tid; v <= V; v += nth) adj_lds[v] = a.m.adj_off[v];
    adjo = adj_lds;
  }
  if (tid < (uint32_t)N_WORDS) misc[tid] = 0;
  for (uint32_t i = tid; i < B * Wv; i += nth) chg[i] = 0;
  __syncthreads();
  const int lim = PACKED ? kPackedMax : kBoundMax;
  const bool vec = (V & 3u) == 0 && (((size_t)a.lb_in | (size_t)a.ub_in) & 15u) == 0;
  {
    uint32_t badm = 0, oobm = 0;
    auto put = [&](uint32_t b, uint32_t v0, const int (&l)[4], const int (&u)[4], uint32_t cnt) {
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
      Cell* row = dom + (size_t)b * SP + v0;
      if (cnt == 4) {
        if constexpr (PACKED) *reinterpret_cast<uint4*>(row) = make_uint4(cl[0], cl[1], cl[2], cl[3]);
        else { reinterpret_cast<int4*>(row)[0] = make_int4(cl[0].x, cl[0].y, cl[1].x, cl[1].y); reinterpret_cast<int4*>(row)[1] = make_int4(cl[2].x, cl[2].y, cl[3].x, cl[3].y); }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if ((uint32_t)i < cnt) row[i] = cl[i];
      }
      if (a.seed_always) nib |= (a.seed_always[v0 >> 5] >> (v0 & 31u)) & ((1u << cnt) - 1u);
      if (nib) atomicOr(&chg[b * Wv + (v0 >> 5)], nib << (v0 & 31u));
      if (bad) badm |= 1u << b;
      if (oob) oobm |= 1u << b;
    };
    const uint32_t SQ = (V + 3) >> 2, tasks = nb * SQ;
    if (vec) {
      constexpr int UF = 4;  // row loads in flight per lane: 2 * UF * 16 bytes
      for (uint32_t t0 = tid; t0 < tasks; t0 += UF * nth) {
        int4 L[UF], U[UF];
        uint32_t bq[UF], qq[UF];
#pragma unroll
        for (int j = 0; j < UF; ++j) {
          const uint32_t t = min(t0 + j * nth, tasks - 1);
          bq[j] = t / SQ; qq[j] = t - bq[j] * SQ;
          const size_t row = (size_t)(node0 + bq[j]) * V;
          L[j] = reinterpret_cast<const int4*>(a.lb_in + row)[qq[j]];
          U[j] = reinterpret_cast<const int4*>(a.ub_in + row)[qq[j]];
        }
#pragma unroll
        for (int j = 0; j < UF; ++j) {
          if (t0 + j * nth >= tasks) break;
          const int l[4] = {L[j].x, L[j].y, L[j].z, L[j].w}, u[4] = {U[j].x, U[j].y, U[j].z, U[j].w};
          put(bq[j], 4 * qq[j], l, u, 4);
        }
      }
    } else {
      for (uint32_t t = tid; t < tasks; t += nth) {
        const uint32_t b = t / SQ, q = t - b * SQ, v0 = 4 * q, cnt = min(4u, V - v0);
        const size_t row = (size_t)(node0 + b) * V;
        int l[4] = {0, 0, 0, 0}, u[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if ((uint32_t)i < cnt) { l[i] = a.lb_in[row + v0 + i]; u[i] = a.ub_in[row + v0 + i]; }
        put(b, v0, l, u, cnt);
      }
    }
    // interned constants: singleton pseudo-variables behind the variables (term/constant.rs:43-68)
    for (uint32_t t = tid; t < nb * (S - V); t += nth) {
      const uint32_t b = t / (S - V), s = V + (t - b * (S - V));
      const int c = a.m.const_val[s - V];
      if constexpr (PACKED) dom[(size_t)b * SP + s] = pack16(c, c); else dom[(size_t)b * SP + s] = make_int2(-c, c);
    }
    if (badm) atomicOr(&misc[N_FAIL], badm);
    if (oobm) atomicOr(&misc[N_OOB], oobm);
  }
  __syncthreads();
  if (misc[N_OOB] && tid == 0) atomicMax(a.violation, 1u);  // sticky: reported by pcp_stats_read

  // ---- rounds: round 0 = the lists of the assigned variables (the sweep), round r = the lists of the changed variables ------
  Ctr ctr;
  uint32_t ev0 = 0;  // item tests of round 0 (they stand for the sweep: counted as evaluated, not as extra steps)
  const uint32_t U4 = 4;
  const bool one_piece = a.m.max_deg <= 64u * U4;
  for (uint32_t round = 0;; ++round) {
    const uint32_t m_count = (round & 1u) ? N_COUNT1 : N_COUNT0, m_rmask = (round & 1u) ? N_RMASK1 : N_RMASK0;
    const uint32_t inert = misc[N_FAIL] | misc[N_OOB];
    // (a) one list for the tile: (variable, mask of the nodes in which it changed); the masks are consumed (zeroed) here,
    // the narrowings of this round mark the same words again behind the barrier
    {
      uint32_t rm = 0;
      for (uint32_t w = tid; w < Wv; w += nth) {
        uint32_t uni = 0;
        for (uint32_t b = 0; b < nb; ++b) uni |= ((inert >> b) & 1u) ? 0u : chg[b * Wv + w];
        if (uni) {
          uint32_t pos = atomicAdd(&misc[m_count], (uint32_t)__popc(uni));
          uint32_t bits = uni;
          while (bits) {
            const uint32_t i = __builtin_ctz(bits);
            bits &= bits - 1;
            uint32_t M = 0;
            for (uint32_t b = 0; b < nb; ++b) M |= (((chg[b * Wv + w] >> i) & 1u) & ~(inert >> b)) << b;
            list[pos++] = ((w << 5) + i) | (M << 16);
            rm |= M;
          }
        }
        for (uint32_t b = 0; b < nb; ++b) chg[b * Wv + w] = 0;
      }
      if (rm) atomicOr(&misc[m_rmask], rm);
    }
    __syncthreads();
    const uint32_t total = misc[m_count];
    if (total == 0) break;
    if (tid == 0) {  // the other slots: last read before this round's barrier
      if (round) { misc[N_WAVES] += __popc(misc[m_rmask]); misc[N_DIRTY] |= misc[m_rmask]; }
      misc[(round & 1u) ? N_COUNT0 : N_COUNT1] = 0; misc[(round & 1u) ? N_RMASK0 : N_RMASK1] = 0;
    }
    // (b) walk the lists.  Piece p (4 x 64 entries) of list e goes to wavefront (p + e) mod nwv: one long list is spread over
    // the workgroup, many lists are balanced to within a piece.
    {
      struct Piece { uint32_t v, M, aoff, deg, k0; };
      auto piece_at = [&](uint32_t e_, uint32_t k_) {
        const uint32_t ent = __builtin_amdgcn_readfirstlane(list[e_]), v = ent & 0xffffu;  // wave-uniform: keeps the loop control scalar
        const uint32_t o0 = __builtin_amdgcn_readfirstlane(adjo[v]), o1 = __builtin_amdgcn_readfirstlane(adjo[v + 1]);
        return Piece{v, ent >> 16, o0, o1 - o0, k_};
      };
      auto load = [&](const Piece& pc, uint2 (&q)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t idx = pc.k0 + u * 64 + lane;
          q[u] = a.m.adjp[pc.aoff + (idx < pc.deg ? idx : 0u)];
        }
      };
      uint32_t my_ev = 0;
      auto process = [&](const Piece& pc, const uint2 (&q)[4]) {
        uint32_t other[4];
        int t[4];
        bool valid[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          valid[u] = pc.k0 + u * 64 + lane < pc.deg;
          other[u] = q[u].x & kSlotMask;
          const int d = (int32_t)q[u].y;
          t[u] = (q[u].x >> 31) ? d : -d;  // v is the record's y: x != v + d  <=>  o != v + d;  v is x: o != v - d
        }
        bool hit[4] = {false, false, false, false};
        if constexpr (PACKED) {
          uint32_t K[4], acc[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { K[u] = pack_mt(t[u]); acc[u] = 0xffffffffu; }
          for (uint32_t m = pc.M; m; m &= m - 1) {
            const Cell* row = dom + (size_t)__builtin_ctz(m) * SP;
            const uint32_t cvv = row[pc.v];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = pk_min_u16(acc[u], neq_terms16(cvv, row[other[u]], K[u]));
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) hit[u] = valid[u] && zero_half(acc[u]);
        } else {
          for (uint32_t m = pc.M; m; m &= m - 1) {
            const Cell* row = dom + (size_t)__builtin_ctz(m) * SP;
            const int2 cvv = row[pc.v];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int2 co = row[other[u]];
              hit[u] |= (cvv.x + co.y == t[u]) | (cvv.y + co.x == -t[u]);  // lb(v) + t == ub(o)  |  ub(v) + t == lb(o)
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) hit[u] = hit[u] && valid[u];
        }
        const uint32_t nm = (uint32_t)__popc(pc.M);
#pragma unroll
        for (int u = 0; u < 4; ++u) my_ev += valid[u] ? nm : 0u;
        if (hit[0] | hit[1] | hit[2] | hit[3]) {
          // flagged entries: the full filter, in the nodes whose domains meet the condition
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (!hit[u]) continue;
            const bool is_y = (q[u].x >> 31) != 0;
            Rec rec;
            rec.xk = (is_y ? other[u] : pc.v) | ((uint32_t)PCP_NEQ << 28);
            rec.y = is_y ? pc.v : other[u];
            rec.z = 0;
            rec.d = (int32_t)q[u].y;
            for (uint32_t m = pc.M; m; m &= m - 1) {
              const uint32_t b = (uint32_t)__builtin_ctz(m);
              const Cell* row = dom + (size_t)b * SP;
              const int2 Vd = cell_bounds<PACKED>(row[pc.v]), O = cell_bounds<PACKED>(row[other[u]]);
              if (Vd.x + t[u] != O.y && Vd.y + t[u] != O.x) continue;
              ++ctr.full;
              if constexpr (PACKED) eval_record(rec, LdsDom16{dom + (size_t)b * SP, 1u, chg + (size_t)b * Wv, &misc[N_FAIL], 1u << b, &ctr});
              else eval_record(rec, LdsDom{dom + (size_t)b * SP, 1u, chg + (size_t)b * Wv, &misc[N_FAIL], 1u << b, &ctr, SumTab{nullptr, nullptr, 0u, 0u, nullptr}});
            }
          }
        }
      };
      const uint32_t k_step = nwv * 64 * U4;
      const uint32_t e_step = one_piece ? nwv : 1u;
      auto k_first = [&](uint32_t e_) { return one_piece ? 0u : ((wv + nwv - (e_ % nwv)) % nwv) * 64 * U4; };
      auto deg_of = [&](uint32_t e_) { const uint32_t v = __builtin_amdgcn_readfirstlane(list[e_]) & 0xffffu; return __builtin_amdgcn_readfirstlane(adjo[v + 1] - adjo[v]); };
      uint32_t e = one_piece ? wv : 0u, k0 = k_first(e);
      auto settle = [&]() { while (e < total && k0 >= deg_of(e)) { e += e_step; k0 = k_first(e); } };
      settle();
      bool have = e < total;
      Piece pa{0, 0, 0, 0, 0};
      uint2 qA[4];
      if (have) { pa = piece_at(e, k0); load(pa, qA); }
      while (have) {
        k0 += k_step;
        if (k0 >= pa.deg) { e += e_step; k0 = k_first(e); }
        settle();
        const bool have_n = e < total;
        Piece pb = pa;
        uint2 qB[4];
        if (have_n) { pb = piece_at(e, k0); load(pb, qB); }  // the next piece's stream is in flight across this piece's tests
        process(pa, qA);
        if (have_n) {
          pa = pb;
#pragma unroll
          for (int u = 0; u < 4; ++u) qA[u] = qB[u];
        }
        have = have_n;
      }
      if (round == 0) ev0 += my_ev;
      ctr.ev += my_ev;
    }
    __syncthreads();
  }

  // ---- status: is any record NOT entailed under the final domains? (store.rs:250-256) ------------------------
  // Records of two assigned variables are entailed at a fixpoint that did not fail (two different values: disjoint), so only the
  // lists of unassigned variables can hold an open record; x != y + d is entailed iff the intervals are disjoint
  // (x_neq_y.rs:71-73 via x_eq_y.rs:87-93).
  {
    const uint32_t inert = misc[N_FAIL] | misc[N_OOB];
    for (uint32_t b = wv; b < nb; b += nwv) {
      if ((inert >> b) & 1u) continue;
      const Cell* row = dom + (size_t)b * SP;
      bool open = false;
      for (uint32_t base = 0; base < V && !open; base += 64) {
        const uint32_t vv = base + lane;
        bool wide = false;
        if (vv < V) { const int2 d = cell_bounds<PACKED>(row[vv]); wide = d.x < d.y; }
        uint64_t bal = __ballot(wide);
        while (bal && !open) {
          const uint32_t u = base + (uint32_t)__builtin_ctzll(bal);
          bal &= bal - 1;
          const int2 Ud = cell_bounds<PACKED>(row[u]);
          const uint32_t o0 = adjo[u], deg = adjo[u + 1] - o0;
          for (uint32_t k = 0; k < deg && !open; k += 64) {
            bool op = false;
            if (k + lane < deg) {
              const uint2 q = a.m.adjp[o0 + k + lane];
              const int d = (int32_t)q.y, t = (q.x >> 31) ? d : -d;
              const int2 O = cell_bounds<PACKED>(row[q.x & kSlotMask]);
              op = !((Ud.x + t > O.y) || (Ud.y + t < O.x));  // not disjoint
            }
            open = __ballot(op) != 0;
          }
        }
      }
      if (open && lane == 0) atomicOr(&misc[N_UNK], 1u << b);
    }
  }

  // ---- write back: the rows of the nodes that changed (every node when the call is not in place) ----------------------------
  __syncthreads();
  {
    const bool in_place = a.lb_in == a.lb_out && a.ub_in == a.ub_out;
    const uint32_t dirty = misc[N_DIRTY], refused = misc[N_OOB];
    uint32_t badm = 0;
    const bool vec_out = (V & 3u) == 0 && (((size_t)a.lb_out | (size_t)a.ub_out) & 15u) == 0;
    for (uint32_t b = 0; b < nb; ++b) {
      if ((refused >> b) & 1u) continue;                   // a refused node's outputs are left alone
      if (in_place && !((dirty >> b) & 1u)) continue;      // the rows in HBM already hold the result
      const Cell* row = dom + (size_t)b * SP;
      int32_t* lbp = a.lb_out + (size_t)(node0 + b) * V;
      int32_t* ubp = a.ub_out + (size_t)(node0 + b) * V;
      bool bad = false;
      if (vec_out) {
        for (uint32_t q = tid; q < (V >> 2); q += nth) {
          int l[4], u[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) { const int2 d = cell_bounds<PACKED>(row[4 * q + i]); l[i] = d.x; u[i] = d.y; bad |= d.x > d.y; }
          reinterpret_cast<int4*>(lbp)[q] = make_int4(l[0], l[1], l[2], l[3]);
          reinterpret_cast<int4*>(ubp)[q] = make_int4(u[0], u[1], u[2], u[3]);
        }
      } else {
        for (uint32_t v = tid; v < V; v += nth) { const int2 d = cell_bounds<PACKED>(row[v]); bad |= d.x > d.y; lbp[v] = d.x; ubp[v] = d.y; }
      }
      if (bad) badm |= 1u << b;
    }
    if (badm) atomicOr(&misc[N_FAIL], badm);
  }
  for (int o = 32; o > 0; o >>= 1) { ctr.narrow += __shfl_down(ctr.narrow, o); ctr.ev += __shfl_down(ctr.ev, o); ctr.full += __shfl_down(ctr.full, o); ev0 += __shfl_down(ev0, o); }
  if (lane == 0) {
    if (ctr.narrow) atomicAdd(&misc[N_NARROW], ctr.narrow);
    if (ctr.ev) atomicAdd(reinterpret_cast<unsigned long long*>(&misc[N_EV]), (unsigned long long)ctr.ev);
    if (ctr.full) atomicAdd(reinterpret_cast<unsigned long long*>(&misc[N_FULL]), (unsigned long long)ctr.full);
    if (ctr.ev - ev0) atomicAdd(reinterpret_cast<unsigned long long*>(&misc[N_STEPS]), (unsigned long long)(ctr.ev - ev0));
  }
  __syncthreads();
  if (tid < nb) {
    const bool failed = (misc[N_FAIL] >> tid) & 1u, refused = (misc[N_OOB] >> tid) & 1u;
    const bool none_open = !((misc[N_UNK] >> tid) & 1u);
    a.status[node0 + tid] = refused ? kStatusRetry : failed ? (uint8_t)PCP_FALSE : (none_open ? (uint8_t)PCP_TRUE : (uint8_t)PCP_UNKNOWN);
  }
  if (tid == 0) {
    const uint32_t active_nodes = (uint32_t)__popc(((nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1u)) & ~misc[N_OOB]));
    // reference-equivalent steps: every propagator of every node once (init_scheduler) + every wake-up of the later rounds
    const unsigned long long s2 = (unsigned long long)active_nodes * a.m.n_recs + *reinterpret_cast<unsigned long long*>(&misc[N_STEPS]);
    atomicAdd((unsigned long long*)&a.stats->steps, s2);
    if (misc[N_NARROW]) atomicAdd((unsigned long long*)&a.stats->narrowings, (unsigned long long)misc[N_NARROW]);
    const unsigned long long sev = *reinterpret_cast<unsigned long long*>(&misc[N_EV]), sfu = *reinterpret_cast<unsigned long long*>(&misc[N_FULL]);
    if (sev) atomicAdd((unsigned long long*)&a.stats->evaluated, sev);
    if (sfu) atomicAdd((unsigned long long*)&a.stats->full_evals, sfu);
    atomicAdd((unsigned long long*)&a.stats->waves, (unsigned long long)(nb + misc[N_WAVES]));
    atomicAdd((unsigned long long*)&a.stats->nodes, (unsigned long long)nb);
    const uint32_t nf = __popc(misc[N_FAIL] & (nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1u)));
    if (nf) atomicAdd((unsigned long long*)&a.stats->failed_nodes, (unsigned long long)nf);
  }
}

size_t lds_bytes_neq(uint32_t n_slots, uint32_t n_vars, uint32_t nodes_per_block, bool packed, bool adj_cache) {
  const NeqCarve c = neq_carve(n_slots, n_vars, nodes_per_block, packed, adj_cache);
  return c.total <= 160 * 1024 ? c.total : 0;
}

hipError_t launch_neqfix(const NeqArgs& a, const LaunchPlan& p, hipStream_t stream) {
  if (a.nodes_per_block == 0 || a.nodes_per_block > 16 || a.m.n_slots >= 65536u || !a.m.adjp) return hipErrorInvalidValue;
  if (a.packed) {
    if (p.lds_bytes > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(neqfix_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(neqfix_kernel<true>, dim3(p.grid), dim3(p.block), p.lds_bytes, stream, a);
  } else {
    if (p.lds_bytes > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(neqfix_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(neqfix_kernel<false>, dim3(p.grid), dim3(p.block), p.lds_bytes, stream, a);
  }
  return hipGetLastError();
}

}  // namespace pcp
