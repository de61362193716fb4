// pcp_big.hip — the propagation fixpoint of binary models whose variable store does not fit LDS as (lb, ub) pairs but does as
// 10-bit cells (gfx950): BASELINE config 3, 50 000 Interval<i32> variables over a declared hull of at most 1024 values, 500 000
// `x <> y + c` propagators.  Implicit-active nodes (a node is its domains; liveness is derived), one node per
// workgroup = per CU.
//
// Same contract as fixpoint_kernel (pcp_kernels.hip): Store::consistency = prepare() + propagation_loop()
// (propagation/store.rs:125-164, 247-257): every propagator once (init_scheduler, store.rs:144-149), then the propagators of
// the changed variables (Store::react, store.rs:191-198) until nothing changes.
//
// MI355X mapping
//  * the node's store sits in LDS as (lb - lo, ub - lo) cells of 20 bits, three per u64 (50 000 variables = 130 KB); a narrowing is
//    ONE compare-and-swap on the cell's word whatever bounds it moves; the two changed-variable bitmasks (6 KB each) are the only
//    other state.  A record with a Constant operand (term/constant.rs:43-68) is lowered by the host to a UNARY record "var < K", "var > K",
//    "var = K" or "var != K" with K exact in 32 bits (the Addition offset folded in): constants need no cells and may lie anywhere;
//  * the kernel is bound by INSTRUCTION ISSUE (16 wavefronts share four SIMDs; rocprof r03: ~110 VALU + ~85 SALU instructions per 64
//    filter steps), so round 4 rebuilt the data path around the instruction count:
//      - the record a lane reads is 8 bytes and carries the CELL COORDINATES of its operands — word index and field (0..2) of x and
//        of y, the offset, the kind (BigRec) — computed once on the host: no division by three, no 64-bit multiply per cell access
//        (they were a third of the VALU instructions), and half the bytes per record;
//      - the table is sorted by kind and a wavefront's 64 records are ONE kind except at the two seams: the kind is read with
//        readfirstlane and the filter runs as straight-line code for that kind, with selects instead of lane-masked branches;
//      - every kind ends in "X := X ∩ [xl, xu], Y := Y ∩ [yl, yu]", so the narrowing code exists once per operand;
//  * the sweep streams the records, 64 per wavefront load, four loads in flight per wavefront; a wake-up round is DENSE (the table is
//    streamed again; a record runs iff one of its operands is marked: two LDS bit tests) or SPARSE (lane = item of a flat item space
//    built per 64 variables from the degrees of the changed ones — wave prefix sums, binary search by ds_bpermute — over adjacency
//    payloads that carry the other operand's cell coordinates), whichever touches fewer records: dense iff the degrees of the changed
//    variables add up to more than half the table (round 3 compared the NUMBER of changed variables with V / 8);
//  * status (store.rs:250-256): a scan for one record that is not entailed, with early exit.
// Integer bound work: no MFMA; the node's 400 KB of bounds cross HBM once in, once out.
#include <algorithm>

#include "pcp_device.hpp"
#include "pcp_neq.h"

namespace pcp {

namespace {

enum { G_FAIL = 0, G_OOB = 1, G_TOTAL0 = 2, G_TOTAL1 = 3, G_UNK = 4, G_WAVES = 5, G_NARROW = 6, G_EV = 8, G_FULL = 10, G_DENSE = 12, G_SPARSE = 13, G_DEG0 = 14, G_DEG1 = 15, G_WORDS = 16 };

struct BigCarve {
  size_t cells, cdom, chg_a, chg_b, misc, total;
};
__host__ __device__ inline BigCarve big_carve(uint32_t V, uint32_t S) {
  auto up = [](size_t x) { return (x + 15) & ~(size_t)15; };
  const size_t Wv = (S + 31) / 32;
  BigCarve c;
  size_t o = 0;
  c.cells = o; o = up(o + (((size_t)V + 2) / 3) * 8);
  c.cdom = o;
  c.chg_a = o; o = up(o + Wv * 4);
  c.chg_b = o; o = up(o + Wv * 4);
  c.misc = o; o = up(o + 16 * 4);
  c.total = o;
  return c;
}

// BigRec / BigAdj field access (layout: pcp_neq.h)
__device__ __forceinline__ uint32_t br_wx(uint2 r) { return r.x & 0x7fffu; }
__device__ __forceinline__ uint32_t br_fx(uint2 r) { return (r.x >> 15) & 3u; }
__device__ __forceinline__ int br_d(uint2 r) { return ((int)(r.x << 2)) >> 19; }  // bits 17..29, sign-extended
__device__ __forceinline__ uint32_t br_kind(uint2 r) { return r.x >> 30; }
__device__ __forceinline__ uint32_t br_wy(uint2 r) { return r.y & 0x7fffu; }
__device__ __forceinline__ uint32_t br_fy(uint2 r) { return (r.y >> 15) & 3u; }

// The node's store: variable::Store::update (variable/store.rs:151-166) on 20-bit cells addressed by (word, field).
struct Cells {
  unsigned long long* c10;
  uint32_t* chg;   // marks of the variables narrowed (bit = variable index = 3 * word + field)
  uint32_t* misc;
  Ctr* c;
  int lo;          // the hull's lower bound (the cells are relative to it; a unary record's K is absolute)
  // (lb - lo, ub - lo) of the cell
  __device__ __forceinline__ int2 load(uint32_t w, uint32_t f) const {
    const uint32_t cell = (uint32_t)(c10[w] >> (f * 20u)) & 0xFFFFFu;
    return make_int2((int)(cell & 1023u), (int)(cell >> 10));
  }
  __device__ __forceinline__ void set_fail() const { atomicOr(&misc[G_FAIL], 1u); }
  // the cell := the cell ∩ [rl, ru] (hull-relative; either may lie outside 0..1023): BOTH bounds in ONE compare-and-swap.  An empty
  // result fails the node (the cell then holds a clamped, unspecified value: SURVEY.md A.4).
  __device__ __forceinline__ void update(uint32_t w, uint32_t f, int rl, int ru) const {
    const uint32_t sh = f * 20u;
    unsigned long long* p = &c10[w];
    unsigned long long old = *p;
    for (;;) {
      const uint32_t cell = (uint32_t)(old >> sh) & 0xFFFFFu;
      const int l = (int)(cell & 1023u), u = (int)(cell >> 10);
      const int ml = max(l, rl), mu = min(u, ru);
      if (ml == l && mu == u) return;  // (somebody else got there first)
      const uint32_t neu_cell = (uint32_t)min(ml, 1023) | ((uint32_t)max(mu, 0) << 10);
      const unsigned long long neu = (old & ~(0xFFFFFull << sh)) | ((unsigned long long)neu_cell << sh);
      const unsigned long long prev = atomicCAS(p, old, neu);
      if (prev == old) {
        c->narrow += (ml != l) + (mu != u);
        const uint32_t v = 3u * w + f;
        atomicOr(&chg[v >> 5], 1u << (v & 31u));
        if (ml > mu) set_fail();
        return;
      }
      old = prev;
    }
  }
};

// propagate() of one binary propagator on hull-relative cells — without is_subsumed(): implicit-active nodes unlink nothing, so whether
// the propagator is entailed is of no interest while the fixpoint runs.  From the pre-read values, as the reference:
//   XNeqY x_neq_y.rs:82-93 (a value is removed only at a bound) | XEqY x_eq_y.rs:102-107 | XLessY x_less_y.rs:104-109.
// KIND is a compile-time constant on the fast path (a wavefront's 64 records are one kind) and 3 = "look at r" at the seams.
template <int KIND>
__device__ __forceinline__ void propagate_big(const uint2 r, const Cells& cs) {
  const uint32_t wx = br_wx(r), fx = br_fx(r), wy = br_wy(r), fy = br_fy(r);
  const int d = br_d(r);
  const uint32_t kind = KIND == 3 ? br_kind(r) : (uint32_t)KIND;
  const int2 X = cs.load(wx, fx), Y = cs.load(wy, fy);
  const int Yl = Y.x + d, Yu = Y.y + d;  // Y as seen through Addition(y, d)  (term/addition.rs:98)
  int xl = X.x, xu = X.y, yl = Yl, yu = Yu;  // the new bounds of X and of Y + d
  if (kind == PCP_LT) {
    xu = min(X.y, Yu - 1); yl = max(Yl, X.x + 1);
  } else if (kind == PCP_NEQ) {
    const bool xs = X.x == X.y, ys = Yl == Yu;
    yl = (xs && X.x == Yl) ? Yl + 1 : Yl;
    yu = (xs && X.x != Yl && X.x == Yu) ? Yu - 1 : Yu;
    xl = (!xs && ys && Yl == X.x) ? X.x + 1 : X.x;
    xu = (!xs && ys && Yl != X.x && Yl == X.y) ? X.y - 1 : X.y;
  } else {
    xl = yl = max(X.x, Yl); xu = yu = min(X.y, Yu);
  }
  if ((xl != X.x) | (xu != X.y)) cs.update(wx, fx, xl, xu);
  if ((yl != Yl) | (yu != Yu)) cs.update(wy, fy, yl - d, yu - d);
}

// A record with a Constant operand, lowered to  var (op) K  with op = 0 "<", 1 ">", 2 "=", 3 "!=" in the low bits of the offset field and K
// hull-relative in .y (any 32-bit value).  Same filters, with the constant side a singleton that an update can only empty
// (Constant::update, term/constant.rs:49-52): emptying it and emptying the variable coincide, see the case analysis in DESIGN.md 4.2.
__device__ __forceinline__ uint32_t br_uop(uint2 r) { return (r.x >> 17) & 3u; }
__device__ __forceinline__ void propagate_unary(const uint2 r, const Cells& cs) {
  const uint32_t w = br_wx(r), f = br_fx(r), op = br_uop(r);
  const int K = (int)r.y - cs.lo;
  const int2 X = cs.load(w, f);
  int xl = X.x, xu = X.y;
  if (op == 0u) xu = min(X.y, K - 1);
  else if (op == 1u) xl = max(X.x, K + 1);
  else if (op == 2u) { xl = max(X.x, K); xu = min(X.y, K); }
  else {
    const bool xs = X.x == X.y;
    if (xs && X.x == K) { cs.set_fail(); return; }  // {c} - {x} is empty: Constant::update returns false
    xl = (!xs && K == X.x) ? X.x + 1 : X.x;
    xu = (!xs && K != X.x && K == X.y) ? X.y - 1 : X.y;
  }
  if ((xl != X.x) | (xu != X.y)) cs.update(w, f, xl, xu);
}
__device__ __forceinline__ bool open_unary(const uint2 r, const Cells& cs) {
  const int2 X = cs.load(br_wx(r), br_fx(r));
  const int K = (int)r.y - cs.lo;
  const uint32_t op = br_uop(r);
  if (op == 0u) return !(X.y < K);
  if (op == 1u) return !(X.x > K);
  if (op == 2u) return !(X.x == K && X.y == K);
  return !(K < X.x || K > X.y);
}

// is the propagator NOT entailed under these domains?  (is_subsumed() != True: x_less_y.rs:73-95, x_eq_y.rs:73-94, x_neq_y.rs:71-73)
__device__ __forceinline__ bool open_big(const uint2 r, const Cells& cs) {
  const int2 X = cs.load(br_wx(r), br_fx(r)), Y = cs.load(br_wy(r), br_fy(r));
  const int d = br_d(r), Yl = Y.x + d, Yu = Y.y + d;
  const uint32_t kind = br_kind(r);
  if (kind == PCP_LT) return !(X.y < Yl);
  if (kind == PCP_EQ) return !(X.x == Yu && X.y == Yl);
  return !(X.x > Yu || Yl > X.y);  // XNeqY is entailed iff the two are disjoint
}

}  // namespace

// UNARY = the model has records with a Constant operand (kind 3 in the stream); compiled out otherwise: their tests in the stream's inner
// loop cost config 3, which has none, 10 %.
template <bool UNARY>
__global__ void __launch_bounds__(1024) bigfix_kernel(const BigArgs a_in) {
  BigArgs a = a_in;
  a.stats += blockIdx.x & (kStatSlots - 1);
  if (a.dbg) a.dbg += (size_t)(blockIdx.x & (kStatSlots - 1)) * PCP_DBG_COUNT;
  if (a.sp_ptr) {  // host-stepped device-side DFS (pcp_dfs_device): the node on top of the stack, as in fixpoint_kernel
    const uint32_t sp = *a.sp_ptr;
    if (sp == 0 || *a.stop_ptr) return;
    const size_t off = (size_t)(sp - 1) * a.m.n_vars;
    a.lb_in += off; a.ub_in += off; a.lb_out += off; a.ub_out += off; a.status += sp - 1;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t tid = threadIdx.x, nth = blockDim.x, lane = tid & 63;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = nth >> 6;
  const uint32_t V = a.m.n_vars, S = a.m.n_slots, Wv = (V + 31) >> 5, P = a.m.n_recs, words = (P + 63) >> 6;
  const BigCarve cv = big_carve(V, S);
  unsigned long long* const cells = reinterpret_cast<unsigned long long*>(smem + cv.cells);
  uint32_t* cur = reinterpret_cast<uint32_t*>(smem + cv.chg_a);
  uint32_t* nxt = reinterpret_cast<uint32_t*>(smem + cv.chg_b);
  uint32_t* const misc = reinterpret_cast<uint32_t*>(smem + cv.misc);
  const uint32_t node = blockIdx.x;
  const size_t row = (size_t)node * V;
  const int lo = a.lo10;

  // ---- phase 0: stage the node's bounds as 10-bit cells, three slots per u64, one thread per word; constants behind the variables ---
  if (tid < (uint32_t)G_WORDS) misc[tid] = 0;
  for (uint32_t w = tid; w < Wv; w += nth) { cur[w] = 0; nxt[w] = 0; }
  __syncthreads();
  {
    bool bad = false, oob = false;
    const uint32_t nw3 = (V + 2) / 3;
    for (uint32_t w = tid; w < nw3; w += nth) {
      unsigned long long word = 0;
#pragma unroll
      for (uint32_t j = 0; j < 3; ++j) {
        const uint32_t v = 3 * w + j;
        int l = 0, u = 0;
        if (v < V) {
          const int lbv = a.lb_in[row + v], ubv = a.ub_in[row + v];
          bad |= lbv > ubv;
          oob |= (lbv < lo) | (ubv > lo + 1023) | (lbv > lo + 1023) | (ubv < lo);
          l = min(max(lbv - lo, 0), 1023); u = min(max(ubv - lo, 0), 1023);
        }
        word |= (unsigned long long)((uint32_t)l | ((uint32_t)u << 10)) << (20 * j);
      }
      cells[w] = word;
    }
    if (bad) atomicOr(&misc[G_FAIL], 1u);
    if (oob) atomicOr(&misc[G_OOB], 1u);
  }
  __syncthreads();
  if (misc[G_OOB]) {  // a bound outside the declared hull: the caller's contract violation (pcp_hip.h)
    if (tid == 0) { a.status[node] = kStatusRetry; atomicMax(a.violation, 1u); }
    return;
  }

  Ctr ctr;
  uint32_t items = 0;  // filter steps of this thread (sweep + rounds)
  // ---- the record stream: every propagator once (FILTER = false) or every propagator with a marked operand (a dense round) -------------
  auto stream = [&](auto touched_only, const uint32_t* mask, uint32_t* mark_into) {
    constexpr bool FILTER = decltype(touched_only)::value;
    const Cells cs{cells, mark_into, misc, &ctr, lo};
    constexpr int D = 4;
    for (uint32_t w0 = wv; w0 < words; w0 += D * nwv) {
      uint2 rc[D];
#pragma unroll
      for (int j = 0; j < D; ++j) rc[j] = a.brec[(size_t)min(w0 + j * nwv, words - 1) * 64 + lane];  // (the table is padded to whole words)
      if (misc[G_FAIL]) break;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const uint32_t w = w0 + j * nwv;
        bool run = w < words && (w << 6) + lane < P;
        if constexpr (FILTER) {
          const uint32_t x = 3u * br_wx(rc[j]) + br_fx(rc[j]);
          const uint32_t y = (UNARY && br_kind(rc[j]) == 3u) ? x : 3u * br_wy(rc[j]) + br_fy(rc[j]);  // (a unary record has one variable)
          run = run && ((((mask[x >> 5] >> (x & 31u)) | (mask[y >> 5] >> (y & 31u))) & 1u) != 0);
        }
        const unsigned long long on = __ballot(run);
        if (!on) continue;
        items += run ? 1u : 0u;
        // one kind per wavefront step except at the table's two seams: the kind becomes a scalar and the filter straight-line code
        const uint32_t k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)br_kind(rc[j]));
        const bool uniform = __ballot(run && br_kind(rc[j]) != k0) == 0ull;
        if (run) {
          if (!uniform) { if (UNARY && br_kind(rc[j]) == 3u) propagate_unary(rc[j], cs); else propagate_big<3>(rc[j], cs); }
          else if (k0 == PCP_LT) propagate_big<PCP_LT>(rc[j], cs);
          else if (k0 == PCP_NEQ) propagate_big<PCP_NEQ>(rc[j], cs);
          else if (!UNARY || k0 == PCP_EQ) propagate_big<PCP_EQ>(rc[j], cs);
          else propagate_unary(rc[j], cs);
        }
      }
    }
  };
  stream(std::false_type{}, nullptr, cur);
  __syncthreads();

  // ---- rounds -----------------------------------------------------------------------------------------------------------------
  for (uint32_t round = 0;; ++round) {
    const uint32_t m_total = (round & 1u) ? G_TOTAL1 : G_TOTAL0, m_deg = (round & 1u) ? G_DEG1 : G_DEG0;
    {
      // how many variables changed, and how many records hang on them (the cost of a sparse round)
      uint32_t n = 0, dg = 0;
      for (uint32_t w = tid; w < Wv; w += nth) {
        uint32_t m = cur[w];
        n += (uint32_t)__popc(m);
        for (; m; m &= m - 1u) { const uint32_t v = (w << 5) + (uint32_t)__builtin_ctz(m); if (v < V) dg += a.m.adj_off[v + 1] - a.m.adj_off[v]; }
      }
      const unsigned long long any = __ballot(n != 0u);
      if (any) {
        for (int o = 32; o > 0; o >>= 1) { n += __shfl_down(n, o); dg += __shfl_down(dg, o); }
        if (lane == 0) { atomicAdd(&misc[m_total], n); atomicAdd(&misc[m_deg], dg); }
      }
    }
    __syncthreads();
    const uint32_t total = misc[m_total], sum_deg = misc[m_deg];
    if (total == 0 || misc[G_FAIL]) break;
    // dense: the table is streamed (P records looked at, two bit tests each); sparse: the changed variables' lists (sum_deg payloads, each
    // a binary search and a gathered load): the list entries cost about twice a streamed record
    const bool dense = a.round_mode ? a.round_mode == 1u : a.dense_k * sum_deg >= P;
    if (tid == 0) {
      misc[G_WAVES] += 1; misc[(round & 1u) ? G_TOTAL0 : G_TOTAL1] = 0; misc[(round & 1u) ? G_DEG0 : G_DEG1] = 0;
      misc[dense ? G_DENSE : G_SPARSE] += 1;
    }
    if (dense) {
      stream(std::true_type{}, cur, nxt);
    } else {
      const Cells cs{cells, nxt, misc, &ctr, lo};
      for (uint32_t w = 2 * wv; w < Wv; w += 2 * nwv) {  // 64 variables per wavefront step
        const uint32_t m0 = cur[w], m1 = (w + 1 < Wv) ? cur[w + 1] : 0u;
        if ((m0 | m1) == 0) continue;
        const uint32_t vb = w << 5;
        const bool on = ((((lane < 32u) ? m0 : m1) >> (lane & 31u)) & 1u) && vb + lane < V;
        uint32_t aoff = 0, deg = 0;
        if (on) { aoff = a.m.adj_off[vb + lane]; deg = a.m.adj_off[vb + lane + 1] - aoff; }
        uint32_t inc = deg;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
          const uint32_t t = __shfl_up(inc, o);
          if (lane >= (uint32_t)o) inc += t;
        }
        const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63), pre = inc - deg;
        // this lane's variable as cell coordinates (one division per changed variable, not per record)
        const uint32_t my_w = (vb + lane) / 3u, my_c = my_w | ((vb + lane - 3u * my_w) << 15);
        constexpr int U = 4;
        for (uint32_t i0 = 0; i0 < T; i0 += 64 * U) {
          uint32_t vc[U], vv[U];
          uint2 q[U];
          bool ok[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const uint32_t i = i0 + u * 64 + lane;
            ok[u] = i < T;
            uint32_t lo_ = 0, hi_ = 64;  // the last lane whose prefix is <= i (a lane of degree 0 is never the last one: its successor shares its prefix)
#pragma unroll
            for (int st = 0; st < 6; ++st) {
              const uint32_t mid = (lo_ + hi_) >> 1;
              const uint32_t pm = (uint32_t)__shfl((int)pre, (int)mid);
              if (pm <= i) lo_ = mid; else hi_ = mid;
            }
            vv[u] = vb + lo_;
            // (the permutes outside any lane predicate: a ds_bpermute reads nothing from a lane that is switched off)
            const uint32_t a_l = (uint32_t)__shfl((int)aoff, (int)lo_), p_l = (uint32_t)__shfl((int)pre, (int)lo_);
            vc[u] = (uint32_t)__shfl((int)my_c, (int)lo_);
            q[u] = a.badj[ok[u] ? a_l + (i - p_l) : 0u];
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            const uint32_t mw = vc[u] & 0x7fffu, mf = vc[u] >> 15;
            if (UNARY && ((q[u].x >> 20) & 1u)) {  // a record with a Constant operand: var (op) K, op in the kind field, K in .y
              ++items;
              propagate_unary(make_uint2(mw | (mf << 15) | (((q[u].x >> 18) & 3u) << 17) | (3u << 30), q[u].y), cs);
              continue;
            }
            // BigAdj: .x = the other operand's word | field << 15 | (this variable is the record's y) << 17 | kind << 18, .y = d
            const uint32_t ow = q[u].x & 0x7fffu, of = (q[u].x >> 15) & 3u, kind = (q[u].x >> 18) & 3u;
            const bool is_y = ((q[u].x >> 17) & 1u) != 0;
            const uint32_t other = 3u * ow + of;
            // RelaxedFifo dedup (relaxed_fifo.rs:42-48): a lower-numbered changed variable of the same record runs it
            if (other < vv[u] && ((cur[other >> 5] >> (other & 31u)) & 1u)) continue;
            ++items;
            // rebuild the record: x first
            uint2 r;
            r.x = (is_y ? ow | (of << 15) : mw | (mf << 15)) | (((uint32_t)(int)q[u].y & 0x1fffu) << 17) | (kind << 30);
            r.y = is_y ? mw | (mf << 15) : ow | (of << 15);
            propagate_big<3>(r, cs);
          }
        }
      }
    }
    __syncthreads();
    for (uint32_t w = tid; w < Wv; w += nth) cur[w] = 0;
    uint32_t* t = cur; cur = nxt; nxt = t;
  }

  // ---- status: is any propagator NOT entailed under the final domains? (store.rs:250-256, A.4) ------------------------------
  __syncthreads();
  if (!misc[G_FAIL]) {
    const Cells cs{cells, nxt, misc, &ctr, lo};
    for (uint32_t w = wv; w < words; w += nwv) {
      if (__hip_atomic_load(&misc[G_UNK], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
      const uint32_t r = (w << 6) + lane;
      const uint2 rec = a.brec[r];  // (padded)
      const bool open_rec = r < P && ((UNARY && br_kind(rec) == 3u) ? open_unary(rec, cs) : open_big(rec, cs));
      if (__ballot(open_rec) != 0 && lane == 0) atomicOr(&misc[G_UNK], 1u);
    }
  }
  __syncthreads();
  // ---- write back: one thread per word, three variables -------------------------------------------------------------------------------
  {
    bool bad = false;
    const uint32_t nw3 = (V + 2) / 3;
    for (uint32_t w = tid; w < nw3; w += nth) {
      const unsigned long long word = cells[w];
#pragma unroll
      for (uint32_t j = 0; j < 3; ++j) {
        const uint32_t v = 3 * w + j;
        if (v >= V) break;
        const uint32_t cell = (uint32_t)(word >> (20 * j)) & 0xFFFFFu;
        const int l = lo + (int)(cell & 1023u), u = lo + (int)(cell >> 10);
        bad |= l > u;
        a.lb_out[row + v] = l; a.ub_out[row + v] = u;
      }
    }
    if (bad) atomicOr(&misc[G_FAIL], 1u);
  }
  for (int o = 32; o > 0; o >>= 1) { ctr.narrow += __shfl_down(ctr.narrow, o); items += __shfl_down(items, o); }
  if (lane == 0) {
    if (ctr.narrow) atomicAdd(&misc[G_NARROW], ctr.narrow);
    if (items) atomicAdd(reinterpret_cast<unsigned long long*>(&misc[G_EV]), (unsigned long long)items);
  }
  __syncthreads();
  if (tid == 0) {
    const bool failed = misc[G_FAIL] != 0;
    a.status[node] = failed ? (uint8_t)PCP_FALSE : (misc[G_UNK] ? (uint8_t)PCP_UNKNOWN : (uint8_t)PCP_TRUE);
    const unsigned long long ev = *reinterpret_cast<unsigned long long*>(&misc[G_EV]);
    atomicAdd((unsigned long long*)&a.stats->steps, ev);
    atomicAdd((unsigned long long*)&a.stats->evaluated, ev);
    atomicAdd((unsigned long long*)&a.stats->full_evals, ev);
    if (misc[G_NARROW]) atomicAdd((unsigned long long*)&a.stats->narrowings, (unsigned long long)misc[G_NARROW]);
    atomicAdd((unsigned long long*)&a.stats->waves, (unsigned long long)(1 + misc[G_WAVES]));
    atomicAdd((unsigned long long*)&a.stats->nodes, 1ull);
    if (failed) atomicAdd((unsigned long long*)&a.stats->failed_nodes, 1ull);
    if (a.dbg) {
      if (misc[G_DENSE]) atomicAdd(&a.dbg[PCP_DBG_BIG_DENSE], (unsigned long long)misc[G_DENSE]);
      if (misc[G_SPARSE]) atomicAdd(&a.dbg[PCP_DBG_BIG_SPARSE], (unsigned long long)misc[G_SPARSE]);
    }
  }
}

size_t lds_bytes_big(uint32_t n_vars, uint32_t n_slots) {
  const BigCarve c = big_carve(n_vars, n_slots);
  return c.total <= 160 * 1024 ? c.total : 0;
}

hipError_t launch_bigfix(const BigArgs& a, const LaunchPlan& p, hipStream_t stream) {
  if (!a.brec || !a.badj || !a.m.adj_off) return hipErrorInvalidValue;
  const void* fn = a.has_unary ? reinterpret_cast<const void*>(bigfix_kernel<true>) : reinterpret_cast<const void*>(bigfix_kernel<false>);
  if (p.lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
    if (e != hipSuccess) return e;
  }
  if (a.has_unary) hipLaunchKernelGGL(bigfix_kernel<true>, dim3(p.grid), dim3(p.block), p.lds_bytes, stream, a);
  else hipLaunchKernelGGL(bigfix_kernel<false>, dim3(p.grid), dim3(p.block), p.lds_bytes, stream, a);
  return hipGetLastError();
}

}  // namespace pcp
