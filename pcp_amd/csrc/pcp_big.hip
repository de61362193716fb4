// pcp_big.hip — the propagation fixpoint of binary models whose variable store does not fit LDS as (lb, ub) pairs but does as
// 10-bit cells (gfx950): BASELINE config 3, 50 000 Interval<i32> variables over a declared hull of at most 1024 values, 500 000
// `x <> y + c` propagators.  Implicit-active nodes (a node is its domains; liveness is derived), one node per
// workgroup = per CU.
//
// Same contract as fixpoint_kernel (pcp_kernels.hip): Store::consistency = prepare() + propagation_loop()
// (propagation/store.rs:125-164, 247-257): every propagator once (init_scheduler, store.rs:144-149), then the propagators of
// the changed variables (Store::react, store.rs:191-198) until nothing changes; one filter step = eval_record (pcp_device.hpp).
//
// MI355X mapping
//  * the node's store sits in LDS as (lb - lo, ub - lo) cells of 20 bits, three per u64 (50 000 variables = 130 KB); a narrowing
//    is a compare-and-swap on the cell's word; the two changed-variable bitmasks (6 KB each) are the only other state;
//  * the sweep streams the 16-byte records, 64 per wavefront load, four loads in flight per wavefront (a CU has one workgroup of
//    16 wavefronts here: the stream is latency-bound, not bandwidth-bound) — from a copy of the table SORTED BY KIND (the order
//    in which the propagators of an implicit node run is free): the 64 lanes of a wavefront then run the same filter, where the
//    model's own order made them run all three in turn;
//  * a wake-up round takes one of two forms, chosen by how many variables changed:
//      dense  (an eighth of the variables or more): the record table is streamed again and a record runs iff one of its
//             operands is marked — two LDS bit tests per record, coalesced 16-byte loads, no indirection.  With 20 records per
//             variable a round in which a tenth of the variables changed touches most records anyway;
//      sparse: lane = item of a flat item space built per 64 variables from the degrees of the changed ones (wave prefix sums,
//             binary search by ds_bpermute); the items' payload loads (8 bytes: other slot, kind, offset) are independent, four
//             rounds of 64 in flight per wavefront.  A round costs the records incident to the changed variables;
//    the generic kernel's list of changed (node, variable) pairs does not exist here: it had 256 entries next to the cells and
//    every round of config 3 overflowed it into a sweep with one load in flight per wavefront (161 ms per 4096 nodes);
//  * status (store.rs:250-256): a scan for one record that is not entailed, with early exit.
// Integer bound work: no MFMA.  2.8 million filter steps per node on LDS cells: bound by VALU issue and LDS atomics, not by HBM
// (the node's 400 KB of bounds cross HBM once in, once out).
#include <algorithm>

#include "pcp_device.hpp"
#include "pcp_neq.h"

namespace pcp {

namespace {

enum { G_FAIL = 0, G_OOB = 1, G_TOTAL0 = 2, G_TOTAL1 = 3, G_UNK = 4, G_WAVES = 5, G_NARROW = 6, G_EV = 8, G_FULL = 10, G_DENSE = 12, G_SPARSE = 13, G_WORDS = 14 };

struct BigCarve {
  size_t cells, cdom, chg_a, chg_b, misc, total;
};
__host__ __device__ inline BigCarve big_carve(uint32_t V, uint32_t S) {
  auto up = [](size_t x) { return (x + 15) & ~(size_t)15; };
  const size_t Wv = (S + 31) / 32;
  BigCarve c;
  size_t o = 0;
  c.cells = o; o = up(o + (((size_t)V + 2) / 3) * 8);
  c.cdom = o; o = up(o + (size_t)(S - V) * 8);
  c.chg_a = o; o = up(o + Wv * 4);
  c.chg_b = o; o = up(o + Wv * 4);
  c.misc = o; o = up(o + 16 * 4);
  c.total = o;
  return c;
}

// variable::Store::update (variable/store.rs:151-166) on 10-bit cells; constants (slots >= n_vars) are singletons that an update
// can only empty (Constant::update, term/constant.rs:49-52).
struct Dom10 {
  unsigned long long* c10;
  const int2* cdom;
  uint32_t n_vars;
  int lo10;
  uint32_t* chg;
  uint32_t* misc;
  Ctr* c;
  __device__ __forceinline__ bool any_sums() const { return false; }
  __device__ __forceinline__ const int32_t* mul_offsets() const { return nullptr; }
  __device__ __forceinline__ static uint32_t word3(uint32_t v) { return __umulhi(v, 0xAAAAAAABu) >> 1; }  // v / 3
  __device__ __forceinline__ int2 load(uint32_t v) const {
    if (v >= n_vars) return cdom[v - n_vars];
    const uint32_t w = word3(v), sh = (v - 3u * w) * 20u;
    const uint32_t cell = (uint32_t)(c10[w] >> sh) & 0xFFFFFu;
    return make_int2(lo10 + (int)(cell & 1023u), lo10 + (int)(cell >> 10));
  }
  __device__ __forceinline__ void mark(uint32_t v) const { atomicOr(&chg[v >> 5], 1u << (v & 31)); }
  __device__ __forceinline__ void set_fail() const { atomicOr(&misc[G_FAIL], 1u); }
  // which = 0 raises lb to nv, which = 1 lowers ub to nv (values relative to lo10)
  __device__ __forceinline__ void narrow(uint32_t v, int nv, int which) const {
    const uint32_t w = word3(v), sh = (v - 3u * w) * 20u;
    unsigned long long* p = &c10[w];
    unsigned long long old = *p;
    for (;;) {
      const uint32_t cell = (uint32_t)(old >> sh) & 0xFFFFFu;
      const int l = (int)(cell & 1023u), u = (int)(cell >> 10);
      int nl = l, nu = u;
      if (which == 0) { if (nv <= l) return; nl = min(nv, 1023); } else { if (nv >= u) return; nu = max(nv, 0); }
      const unsigned long long neu = (old & ~(0xFFFFFull << sh)) | ((unsigned long long)((uint32_t)nl | ((uint32_t)nu << 10)) << sh);
      const unsigned long long prev = atomicCAS(p, old, neu);
      if (prev == old) {
        ++c->narrow;
        mark(v);
        if ((which == 0 ? nv : l) > (which == 0 ? u : nv)) set_fail();
        return;
      }
      old = prev;
    }
  }
  __device__ __forceinline__ void raise_lb(uint32_t v, int nlb) const {
    if (v >= n_vars) { if (nlb > cdom[v - n_vars].y) set_fail(); return; }
    narrow(v, nlb - lo10, 0);
  }
  __device__ __forceinline__ void lower_ub(uint32_t v, int nub) const {
    if (v >= n_vars) { if (nub < cdom[v - n_vars].x) set_fail(); return; }
    narrow(v, nub - lo10, 1);
  }
};

// propagate() of one binary propagator — eval_record without is_subsumed(): implicit-active nodes unlink nothing, so whether the
// propagator is entailed is of no interest while the fixpoint runs.  Same updates, from the pre-read values:
//   XNeqY x_neq_y.rs:82-93 (a value is removed only at a bound) | XEqY x_eq_y.rs:102-107 | XLessY x_less_y.rs:104-109.
__device__ __forceinline__ void propagate_binary(const uint32_t kind, const uint32_t x, const uint32_t y, const int d, const Dom10& dm) {
  const int2 X = dm.load(x), Y = dm.load(y);
  const int Yl = Y.x + d, Yu = Y.y + d;  // Y as seen through Addition(y, d)  (term/addition.rs:98)
  if (kind == PCP_LT) {
    const int nxu = min(X.y, Yu - 1), nYl = max(Yl, X.x + 1);
    if (nxu < X.y) dm.lower_ub(x, nxu);
    if (nYl > Yl) dm.raise_lb(y, nYl - d);
    if (X.x > nxu || nYl > Yu) dm.set_fail();
  } else if (kind == PCP_NEQ) {
    if (X.x == X.y) {
      if (X.x == Yl) { dm.raise_lb(y, Yl + 1 - d); if (Yl + 1 > Yu) dm.set_fail(); }
      else if (X.x == Yu) { dm.lower_ub(y, Yu - 1 - d); if (Yl > Yu - 1) dm.set_fail(); }
    } else if (Yl == Yu) {
      if (Yl == X.x) dm.raise_lb(x, X.x + 1);       // (X is not a singleton: it cannot become empty)
      else if (Yl == X.y) dm.lower_ub(x, X.y - 1);
    }
  } else {
    const int nl = max(X.x, Yl), nu = min(X.y, Yu);
    if (nl > X.x) dm.raise_lb(x, nl);
    if (nu < X.y) dm.lower_ub(x, nu);
    if (nl > Yl) dm.raise_lb(y, nl - d);
    if (nu < Yu) dm.lower_ub(y, nu - d);
    if (nl > nu) dm.set_fail();
  }
}

}  // namespace

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
  const uint32_t V = a.m.n_vars, S = a.m.n_slots, Wv = (S + 31) >> 5, P = a.m.n_recs, words = (P + 63) >> 6;
  const BigCarve cv = big_carve(V, S);
  unsigned long long* const cells = reinterpret_cast<unsigned long long*>(smem + cv.cells);
  int2* const cdom = reinterpret_cast<int2*>(smem + cv.cdom);
  uint32_t* cur = reinterpret_cast<uint32_t*>(smem + cv.chg_a);
  uint32_t* nxt = reinterpret_cast<uint32_t*>(smem + cv.chg_b);
  uint32_t* const misc = reinterpret_cast<uint32_t*>(smem + cv.misc);
  const uint32_t node = blockIdx.x;
  const size_t row = (size_t)node * V;

  // ---- phase 0: stage the node's bounds as 10-bit cells, three variables per u64, one thread per word ----------------------
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
          oob |= (lbv < a.lo10) | (ubv > a.lo10 + 1023) | (lbv > a.lo10 + 1023) | (ubv < a.lo10);
          l = min(max(lbv - a.lo10, 0), 1023); u = min(max(ubv - a.lo10, 0), 1023);
        }
        word |= (unsigned long long)((uint32_t)l | ((uint32_t)u << 10)) << (20 * j);
      }
      cells[w] = word;
    }
    for (uint32_t v = V + tid; v < S; v += nth) { int2 d; d.x = d.y = a.m.const_val[v - V]; cdom[v - V] = d; }
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
  // ---- phase 1: every propagator once.  Four record loads per wavefront in flight. --------------------------------------------
  auto stream = [&](auto touched_only, const uint32_t* mask, uint32_t* mark_into) {
    constexpr bool FILTER = decltype(touched_only)::value;
    const Dom10 dm{cells, cdom, V, a.lo10, mark_into, misc, &ctr};
    constexpr int D = 4;
    for (uint32_t w0 = wv; w0 < words; w0 += D * nwv) {
      Rec rc[D];
#pragma unroll
      for (int j = 0; j < D; ++j) rc[j] = a.recs_by_kind[(size_t)min(w0 + j * nwv, words - 1) * 64 + lane];  // (the table is padded to whole words)
      if (misc[G_FAIL]) break;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const uint32_t w = w0 + j * nwv;
        bool run = w < words && (w << 6) + lane < P;
        if constexpr (FILTER) {
          const uint32_t x = rc[j].xk & kSlotMask, y = rc[j].y;
          run = run && ((((mask[x >> 5] >> (x & 31u)) | (mask[y >> 5] >> (y & 31u))) & 1u) != 0);
        }
        if (run) { ++items; propagate_binary(rc[j].xk >> 28, rc[j].xk & kSlotMask, rc[j].y, rc[j].d, dm); }
      }
    }
  };
  stream(std::false_type{}, nullptr, cur);
  __syncthreads();

  // ---- rounds -----------------------------------------------------------------------------------------------------------------
  for (uint32_t round = 0;; ++round) {
    const uint32_t m_total = (round & 1u) ? G_TOTAL1 : G_TOTAL0;
    {
      uint32_t n = 0;
      for (uint32_t w = tid; w < Wv; w += nth) n += (uint32_t)__popc(cur[w]);
      for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
      if (lane == 0 && n) atomicAdd(&misc[m_total], n);
    }
    __syncthreads();
    const uint32_t total = misc[m_total];
    if (total == 0 || misc[G_FAIL]) break;
    const bool dense = a.round_mode ? a.round_mode == 1u : total * 8u >= V;
    if (tid == 0) { misc[G_WAVES] += 1; misc[(round & 1u) ? G_TOTAL0 : G_TOTAL1] = 0; misc[dense ? G_DENSE : G_SPARSE] += 1; }
    if (dense) {
      stream(std::true_type{}, cur, nxt);
    } else {
      const Dom10 dm{cells, cdom, V, a.lo10, nxt, misc, &ctr};
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
        constexpr int U = 4;
        for (uint32_t i0 = 0; i0 < T; i0 += 64 * U) {
          uint32_t vv[U];
          uint2 q[U];
          bool ok[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const uint32_t i = i0 + u * 64 + lane;
            ok[u] = i < T;
            uint32_t lo = 0, hi = 64;  // the last lane whose prefix is <= i (a lane of degree 0 is never the last one: its successor shares its prefix)
#pragma unroll
            for (int st = 0; st < 6; ++st) {
              const uint32_t mid = (lo + hi) >> 1;
              const uint32_t pm = (uint32_t)__shfl((int)pre, (int)mid);
              if (pm <= i) lo = mid; else hi = mid;
            }
            vv[u] = vb + lo;
            // (both permutes outside any lane predicate: a ds_bpermute reads nothing from a lane that is switched off)
            const uint32_t a_l = (uint32_t)__shfl((int)aoff, (int)lo), p_l = (uint32_t)__shfl((int)pre, (int)lo);
            q[u] = a.m.adjp[ok[u] ? a_l + (i - p_l) : 0u];
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            const uint32_t other = q[u].x & kSlotMask, kind = (q[u].x >> 28) & 7u;
            const bool is_y = (q[u].x >> 31) != 0;
            // RelaxedFifo dedup (relaxed_fifo.rs:42-48): a lower-numbered changed variable of the same record runs it
            if (other < vv[u] && ((cur[other >> 5] >> (other & 31u)) & 1u)) continue;
            ++items;
            propagate_binary(kind, is_y ? other : vv[u], is_y ? vv[u] : other, (int32_t)q[u].y, dm);
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
    const Dom10 dm{cells, cdom, V, a.lo10, nxt, misc, &ctr};
    for (uint32_t w = wv; w < words; w += nwv) {
      if (__hip_atomic_load(&misc[G_UNK], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
      const uint32_t r = (w << 6) + lane;
      const Rec rec = a.m.recs[r];  // (padded)
      const bool open_rec = r < P && !eval_record(rec, dm);  // at the fixpoint every filter is a no-op: this only reports is_subsumed()
      if (__ballot(open_rec) != 0 && lane == 0) atomicOr(&misc[G_UNK], 1u);
    }
  }
  __syncthreads();
  // ---- write back ---------------------------------------------------------------------------------------------------------------
  {
    const Dom10 dm{cells, cdom, V, a.lo10, nxt, misc, &ctr};
    bool bad = false;
    for (uint32_t v = tid; v < V; v += nth) {
      const int2 d = dm.load(v);
      bad |= d.x > d.y;
      a.lb_out[row + v] = d.x; a.ub_out[row + v] = d.y;
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
  if (!a.m.adjp || !a.m.recs) return hipErrorInvalidValue;
  if (p.lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bigfix_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(bigfix_kernel, dim3(p.grid), dim3(p.block), p.lds_bytes, stream, a);
  return hipGetLastError();
}

}  // namespace pcp
