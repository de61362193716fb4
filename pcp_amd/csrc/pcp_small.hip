// pcp_small.hip — the propagation fixpoint of SMALL stores (gfx950): a few dozen variables, up to a few thousand elementary filters — BASELINE
// config 4 (the Golomb-ruler network: 55 variables, one Distinct of 990 XNeqY, 45 XEqYPlusZ, 10 XLessY / XEqY), the reference's own test models.
//
// Same contract as fixpoint_kernel (pcp_kernels.hip): Store::consistency = prepare() + propagation_loop() (propagation/store.rs:125-164,
// 247-257) for a batch of nodes, explicit unit-level `active` rows or implicit-active nodes: every active propagator once
// (init_scheduler, store.rs:144-149), then again while anything changes (Store::react wakes the propagators of a changed variable,
// store.rs:191-198; on a store this small "all of them" is cheaper than finding out which); an entailed unit is unlinked (store.rs:200-207).
//
// MI355X mapping: ONE WAVEFRONT PER NODE.  The generic kernel gives a tile of 16 such nodes a 1024-thread workgroup and runs the machinery
// it needs for million-record models (packed tiles, bulk range tests, changed-pair lists, workgroup barriers between rounds): 0.70 ms for 4096
// nodes of 440 bytes, none of it spent on filters.  Here a node's domains are an LDS slice of its wavefront ((-lb, ub) cells: both narrowings
// are ds_min), the records come from an LDS copy the workgroup shares, a round is every live record once, lane-strided, and rounds are
// separated by a wavefront barrier only; sixteen wavefronts per CU run sixteen nodes at sixteen different points of their fixpoints.  A round's
// last pass — the one that narrows nothing — has evaluated every live record on the final domains, so its is_subsumed() results ARE the units'
// entailment: a unit all of whose members are entailed is unlinked.
// ALL-DIFFERENT units (Distinct::new, propagators/distinct.rs:63-83: x != y over every pair of a variable set; the host recognises them)
// are filtered as a group, lane = variable: the assigned variables' values go into a 128-bit mask above the unit's smallest lower bound
// (an LDS atomicOr: a bit that was already set is two variables with one value — the node fails, as the pair's filter would make it);
// every other variable moves its bounds past the values in the mask.  Pair by pair the reference removes a value only at a bound and only
// against an assigned partner (x_neq_y.rs:82-93), one value per wake-up: the mask does the same removals, all at once — same fixpoint
// ("jump windows").  990 pair filters of config 4's Distinct become one step of 45 lanes.  A unit whose values span more than
// 128 falls back to its pairs for that round.  Integer bound work: no MFMA.
#include <algorithm>

#include "pcp_device.hpp"
#include "pcp_neq.h"

namespace pcp {

namespace {

enum { S_FAIL = 0, S_OOB = 1, S_TAKEN = 4 /* .. 7 */, S_WORDS = 8 };
constexpr uint32_t kSmallAdUnits = 8;    // all-different units filtered as groups (more of them: the later ones run pair by pair)
constexpr uint32_t kSmallAdVars = 256;   // their variables, all together
constexpr uint32_t kSmallAdWords = 1 + 3 * kSmallAdUnits + kSmallAdVars;

// the workgroup's shared part of LDS: the records, their unit ids (u16), the all-different tables
__host__ __device__ inline size_t small_shared_off(uint32_t P, int part) {
  auto up = [](size_t x) { return (x + 15) & ~(size_t)15; };
  const size_t o_unit = up((size_t)P * sizeof(Rec)), o_ad = o_unit + up((size_t)P * 2), o_end = o_ad + up((size_t)kSmallAdWords * 4);
  return part == 0 ? o_unit : part == 1 ? o_ad : o_end;
}
__host__ __device__ inline size_t small_shared_bytes(uint32_t P) { return small_shared_off(P, 2); }

__host__ __device__ inline size_t small_wave_bytes(uint32_t S, uint32_t U) {
  auto up = [](size_t x) { return (x + 15) & ~(size_t)15; };
  return up((size_t)S * 8) + up((size_t)((S + 31) / 32) * 4) + 2 * up((size_t)((U + 31) / 32) * 4) + up(S_WORDS * 4);
}

}  // namespace

__global__ void __launch_bounds__(256) smallfix_kernel(const SmallArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = blockDim.x >> 6;
  const uint32_t V = a.m.n_vars, S = a.m.n_slots, P = a.m.n_recs, U = a.n_units, Wu = (U + 31) >> 5, Wv = (S + 31) >> 5, words64 = (U + 63) >> 6;
  auto up = [](size_t x) { return (x + 15) & ~(size_t)15; };
  // the workgroup's copy of the records, then one slice per wavefront: cells, a (dummy) changed mask, the live units, the entailed units, flags
  // (everything a round reads comes from LDS: with the unit ids and the all-different tables in global memory a lane's round was a chain
  // of dependent L2 round trips, one per record — 0.26 ms for config 4 where the filters themselves take a fifth of that)
  Rec* const recs = reinterpret_cast<Rec*>(smem);
  uint16_t* const runit = reinterpret_cast<uint16_t*>(smem + up((size_t)P * sizeof(Rec)));
  uint32_t* const adt = reinterpret_cast<uint32_t*>(smem + small_shared_off(P, 1));
  unsigned char* const mine = smem + small_shared_bytes(P) + (size_t)wv * small_wave_bytes(S, U);
  int2* const dom = reinterpret_cast<int2*>(mine);
  uint32_t* const chg = reinterpret_cast<uint32_t*>(mine + up((size_t)S * 8));
  uint32_t* const live = chg + (up((size_t)Wv * 4) >> 2);
  uint32_t* const ent = live + (up((size_t)Wu * 4) >> 2);
  uint32_t* const misc = ent + (up((size_t)Wu * 4) >> 2);
  for (uint32_t r = tid; r < P; r += blockDim.x) { recs[r] = a.m.recs[r]; runit[r] = (uint16_t)(a.rec_unit ? a.rec_unit[r] : r); }
  // the all-different tables: [n, (unit, count, first) * n] then the variables, kSmallAdWords words in all (the host checks that they fit)
  const uint32_t n_ad = a.ad_tab ? min(a.ad_tab[0], kSmallAdUnits) : 0u;
  if (n_ad) {
    const uint32_t n_vars_ad = a.ad_tab[3 * n_ad] + a.ad_tab[3 * n_ad - 1];  // first + count of the last unit
    for (uint32_t i = tid; i < 1 + 3 * n_ad; i += blockDim.x) adt[i] = a.ad_tab[i];
    for (uint32_t i = tid; i < n_vars_ad; i += blockDim.x) adt[1 + 3 * kSmallAdUnits + i] = a.ad_vars[i];
  }
  const uint32_t* const adv = adt + 1 + 3 * kSmallAdUnits;
  __syncthreads();
  auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); };
  pcp_stats* const stats = a.stats + (blockIdx.x & (kStatSlots - 1));
  unsigned long long acc_s2 = 0, acc_s3 = 0, acc_narrow = 0, acc_waves = 0, acc_nodes = 0, acc_failed = 0;

  uint32_t first = blockIdx.x * nwv + wv, stride = gridDim.x * nwv, n_nodes = a.n_nodes;
  size_t row_base = 0;
  uint32_t st_base = 0;
  if (a.sp_ptr) {  // host-stepped device-side DFS (pcp_dfs_device): ONE node, the one on top of the stack
    const uint32_t sp = *a.sp_ptr;
    if (sp == 0 || *a.stop_ptr) return;
    row_base = (size_t)(sp - 1) * V; st_base = sp - 1; n_nodes = 1;
  }
  for (uint32_t node = first; node < n_nodes; node += stride) {
    const size_t row = row_base + (size_t)node * V;
    // ---- stage: the node's domains, its live units -----------------------------------------------------------------------------------------
    if (lane < (uint32_t)S_WORDS) misc[lane] = 0;
    for (uint32_t w = lane; w < Wv; w += 64) chg[w] = 0;
    for (uint32_t w = lane; w < Wu; w += 64) {  // Store::active (one bit per unit): the caller's row, or every unit (implicit-active nodes)
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
    if (__ballot(bad) && lane == 0) misc[S_FAIL] = 1u;
    wave_sync();

    // ---- rounds: every record of a live unit once per round, until a round narrows nothing ------------------------------------------------
    Ctr ctr;
    const LdsDom dm{dom, 1u, chg, &misc[S_FAIL], 1u, &ctr, a.m.sums};
    uint32_t s2 = 0, s3 = 0, rounds = 0;
    bool failed = __builtin_amdgcn_readfirstlane(misc[S_FAIL]) != 0;
    // the node's OWN propagators (pcp_propagate_device_units): what a branch appends to ONE node's cstore and what cannot be folded into its
    // bounds — Enumerate's  x != v  with v inside the domain (search/branching/enumerate.rs:48-59; it stays active until v reaches a bound).
    // One variable against one Constant, kinds XNeqY / XEqY / XLessY; lane-strided like the records; `nu_open`: one of them is not entailed.
    const uint32_t nu0 = a.nu_off ? a.nu_off[node] : 0u, nu1 = a.nu_off ? a.nu_off[node + 1] : 0u;
    bool nu_open = false;
    while (!failed) {
      ++rounds;
      for (uint32_t w = lane; w < Wu; w += 64) ent[w] = live[w];  // a live unit counts as entailed until one of its members says otherwise
      wave_sync();
      const uint32_t before = ctr.narrow;
      nu_open = false;
      for (uint32_t i = nu0 + lane; i < nu1; i += 64) {
        const pcp_prop p = a.nu[i];
        // x + ox (kind) c  <=>  x (kind) K = c - ox;   c (kind) y + oy  <=>  y != K | y = K | y > K  with K = c - oy   (term/addition.rs:98, cmp/mod.rs:40-60)
        const bool cy = p.var[1] == PCP_CONST, cx = p.var[0] == PCP_CONST;
        const uint32_t v = cy ? p.var[0] : p.var[1];
        if (cx == cy || v >= V || p.kind > PCP_LT) { dm.set_fail(); misc[S_OOB] = 1u; continue; }  // malformed: the node is refused
        const long long K64 = cy ? (long long)p.off[1] - p.off[0] : (long long)p.off[0] - p.off[1];
        const int K = (int)max(-(long long)kBoundMax - 2, min((long long)kBoundMax + 2, K64));  // (beyond every bound either way)
        const uint32_t op = p.kind == PCP_NEQ ? 3u : p.kind == PCP_EQ ? 2u : (cy ? 0u : 1u);  // 0 "< K", 1 "> K", 2 "= K", 3 "!= K"
        const int2 c_ = dom[v];
        const int xl0 = -c_.x, xu0 = c_.y;
        int xl = xl0, xu = xu0;
        ++s2;
        if (op == 0u) xu = min(xu0, K - 1);           // XLessY::propagate (x_less_y.rs:104-109) against Constant::update (term/constant.rs:49-52)
        else if (op == 1u) xl = max(xl0, K + 1);
        else if (op == 2u) { xl = max(xl0, K); xu = min(xu0, K); }  // XEqY (x_eq_y.rs:102-107)
        else {                                        // XNeqY (x_neq_y.rs:82-93): a value is removed only at a bound
          if (xl0 == xu0 && xl0 == K) { dm.set_fail(); continue; }
          if (xl0 != xu0) { if (K == xl0) xl = xl0 + 1; else if (K == xu0) xu = xu0 - 1; }
        }
        if (xl > xu) { dm.set_fail(); continue; }
        if (xl > xl0) dm.raise_lb(v, xl);
        if (xu < xu0) dm.lower_ub(v, xu);
        // is_subsumed() on what propagate() left (store.rs:166-175)
        const bool entailed = op == 0u ? xu < K : op == 1u ? xl > K : op == 2u ? (xl == K && xu == K) : (K < xl || K > xu);
        nu_open |= !entailed;
      }
      unsigned long long grouped = 0;  // (wave-uniform) the all-different units filtered as a group this round: bit k = entry k of ad_tab
      for (uint32_t k = 0; k < n_ad; ++k) {
        const uint32_t au = adt[1 + 3 * k], cnt = adt[2 + 3 * k], v0 = adt[3 + 3 * k];
        if (!((live[au >> 5] >> (au & 31u)) & 1u)) continue;
        const bool on = lane < cnt;
        const uint32_t v = on ? adv[v0 + lane] : 0u;
        int2 d = make_int2(0, 0);
        if (on) { const int2 c_ = dom[v]; d = make_int2(-c_.x, c_.y); }
        int lo = on ? d.x : 0x7fffffff, hi = on ? d.y : (int)0x80000000;
        for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
        if ((long long)hi - (long long)lo >= 128) continue;  // (wave-uniform) too wide for the mask: this unit's pairs run below
        grouped |= 1ull << k;
        if (lane < 4) misc[S_TAKEN + lane] = 0;
        wave_sync();
        if (lane == 0) s2 += cnt * (cnt - 1u) / 2u;  // the unit's cnt (cnt - 1) / 2 pair filters, what the reference (and small_alldiff = 0) runs and counts for it in one round
        const bool single = on && d.x == d.y;
        if (single) {
          const uint32_t b = (uint32_t)(d.x - lo);
          if (atomicOr(&misc[S_TAKEN + (b >> 5)], 1u << (b & 31u)) & (1u << (b & 31u))) dm.set_fail();  // two variables, one value
        }
        wave_sync();
        if (on && !single) {
          // the 128 value bits as two 64-bit registers.  (Made opaque: left to itself the compiler turns "pick one of four loaded words"
          // into "pick one of four ADDRESSES, then load", through an array of pointers in scratch memory — 72 bytes per lane.)
          unsigned long long tlo = (unsigned long long)misc[S_TAKEN] | ((unsigned long long)misc[S_TAKEN + 1] << 32);
          unsigned long long thi = (unsigned long long)misc[S_TAKEN + 2] | ((unsigned long long)misc[S_TAKEN + 3] << 32);
          asm volatile("" : "+v"(tlo), "+v"(thi));
          auto taken = [&](int val) { const uint32_t b = (uint32_t)(val - lo); return (((b < 64 ? tlo : thi) >> (b & 63u)) & 1ull) != 0; };
          int nl = d.x, nu = d.y;
          while (nl <= nu && taken(nl)) ++nl;
          while (nu >= nl && taken(nu)) --nu;
          if (nl > d.x) dm.raise_lb(v, nl);
          if (nu < d.y && nu >= nl) dm.lower_ub(v, nu);
          if (nl > nu) dm.set_fail();
        }
        wave_sync();
        // the unit's is_subsumed(): True iff every pair is disjoint (conjunction.rs:78-94 over x_neq_y.rs:71-73).  More values claimed than
        // the hull has room for is an overlap somewhere (the usual case, one wave sum); otherwise every lane looks at every other interval.
        {
          int l2 = 0, u2 = -1;
          if (on) { const int2 c2 = dom[v]; l2 = -c2.x; u2 = c2.y; }
          uint32_t claimed = on && u2 >= l2 ? (uint32_t)(u2 - l2 + 1) : 0u;
          for (int o = 32; o > 0; o >>= 1) claimed += __shfl_xor(claimed, o);
          bool overlap = claimed > (uint32_t)(hi - lo + 1);
          if (!overlap) {
            bool ov = false;
            for (uint32_t j = 0; j < cnt; ++j) {
              const int lj = __builtin_amdgcn_readlane(l2, (int)j), uj = __builtin_amdgcn_readlane(u2, (int)j);
              ov |= on && j != lane && !(u2 < lj || uj < l2);
            }
            overlap = __ballot(ov) != 0;
          }
          if (overlap && lane == 0) atomicAnd(&ent[au >> 5], ~(1u << (au & 31u)));
        }
      }
      for (uint32_t r = lane; r < P; r += 64) {
        const uint32_t u = runit[r];
        if (!((live[u >> 5] >> (u & 31u)) & 1u)) continue;
        if (grouped) {  // a member of an all-different unit that was filtered as a group this round?
          bool skip = false;
          for (uint32_t k = 0; k < n_ad; ++k) skip |= ((grouped >> k) & 1ull) && adt[1 + 3 * k] == u;
          if (skip) continue;
        }
        const Rec rec = recs[r];
        if ((rec.xk >> 28) >= PCP_LT3) ++s3; else ++s2;
        if (!eval_record(rec, dm)) atomicAnd(&ent[u >> 5], ~(1u << (u & 31u)));   // propagate_one + is_subsumed (store.rs:166-175)
      }
      wave_sync();
      failed = __builtin_amdgcn_readfirstlane(misc[S_FAIL]) != 0;
      if (!__ballot(ctr.narrow != before)) break;
    }

    if (__builtin_amdgcn_readfirstlane(misc[S_OOB]) != 0) {  // a malformed node unit: the node is refused, its outputs left alone
      if (lane == 0) { a.status[st_base + node] = kStatusRetry; atomicMax(a.violation, 1u); }
      wave_sync();
      continue;
    }
    // ---- write back, unlink the entailed units (store.rs:200-207), status (store.rs:250-256) --------------------------------------------------
    bool emptied = false;
    for (uint32_t v = lane; v < V; v += 64) {
      const int2 d = dom[v];
      emptied |= -d.x > d.y;
      a.lb_out[row + v] = -d.x; a.ub_out[row + v] = d.y;
    }
    failed = failed || __ballot(emptied) != 0;
    bool any_live = false;
    for (uint32_t w = lane; w < Wu; w += 64) {
      const uint32_t left = failed ? live[w] : live[w] & ~ent[w];
      live[w] = left;
      any_live |= left != 0;
    }
    wave_sync();
    if (a.active_out)
      for (uint32_t w = lane; w < words64; w += 64) {
        const uint64_t lo = live[2 * w], hi = (2 * w + 1 < Wu) ? live[2 * w + 1] : 0u;
        a.active_out[(size_t)node * words64 + w] = lo | (hi << 32);
      }
    const bool unknown = __ballot(any_live || nu_open) != 0;
    if (lane == 0) a.status[st_base + node] = failed ? (uint8_t)PCP_FALSE : (unknown ? (uint8_t)PCP_UNKNOWN : (uint8_t)PCP_TRUE);
    for (int o = 32; o > 0; o >>= 1) { s2 += __shfl_down(s2, o); s3 += __shfl_down(s3, o); ctr.narrow += __shfl_down(ctr.narrow, o); }
    acc_s2 += s2; acc_s3 += s3; acc_narrow += ctr.narrow; acc_waves += rounds ? rounds : 1; acc_nodes += 1; acc_failed += failed ? 1 : 0;
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
    if (a.dbg && acc_nodes) atomicAdd(&a.dbg[(size_t)(blockIdx.x & (kStatSlots - 1)) * PCP_DBG_COUNT + PCP_DBG_SMALL_NODES], acc_nodes);
  }
}

size_t lds_bytes_small(uint32_t n_slots, uint32_t n_units, uint32_t n_recs, uint32_t waves) {
  const size_t b = small_shared_bytes(n_recs) + (size_t)waves * small_wave_bytes(n_slots, n_units);
  return b <= 160 * 1024 ? b : 0;
}

hipError_t launch_smallfix(const SmallArgs& a, const LaunchPlan& p, hipStream_t stream) {
  if (!a.m.recs || !a.status) return hipErrorInvalidValue;
  if (p.lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(smallfix_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(smallfix_kernel, dim3(p.grid), dim3(p.block), p.lds_bytes, stream, a);
  return hipGetLastError();
}

}  // namespace pcp
