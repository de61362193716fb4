// pcp_device.hpp — device-side building blocks shared by the gfx950 kernels (pcp_kernels.hip, pcp_neq.hip):
// the domain access policies (variable::Store::update as LDS / global atomics, variable/store.rs:151-166) and
// eval_record = one filter step, propagate() + is_subsumed() of one elementary propagator (propagation/store.rs:166-183,
// propagators/cmp/*.rs).  Everything lives in an unnamed namespace: each translation unit gets its own copy.
#pragma once
#include "pcp_internal.h"

namespace pcp {
namespace {

constexpr int kWave = 64;

struct Ctr {  // per-thread counters, reduced once per block
  uint32_t narrow = 0;
  uint32_t ev = 0;    // pcp_stats.evaluated: (record, node) pairs tested on the node's own domains
  uint32_t full = 0;  // pcp_stats.full_evals: pairs that ran eval_record
  // wave-uniform amounts are added on lane 0 only (the block reduction sums all lanes)
  __device__ __forceinline__ void add_ev_uniform(uint32_t n) { ev += (threadIdx.x & 63u) == 0u ? n : 0u; }
  __device__ __forceinline__ void add_full_uniform(uint32_t n) { full += (threadIdx.x & 63u) == 0u ? n : 0u; }
};

// ------------------------------------------------------------------------------------------------
// Domain access policy: node-local (lb,ub) pairs in LDS.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int clamp_sum(long long v) {
  const long long lo = -2147483647LL, hi = 2147483647LL;
  return (int)(v < lo ? lo : (v > hi ? hi : v));
}
// Sum::read (term/sum.rs:76-81): the interval sum of the members' domains, through the policy's own variable load.
template <class D>
__device__ __forceinline__ int2 sum_read(const D& dm, const SumTab& t, uint32_t slot) {
  long long lo = 0, hi = 0;
  const uint32_t i0 = t.off[slot - t.first], i1 = t.off[slot - t.first + 1];
  for (uint32_t i = i0; i < i1; ++i) { const int2 d = dm.load_var(t.mem[i]); lo += d.x; hi += d.y; }
  return make_int2(clamp_sum(lo), clamp_sum(hi));
}

struct LdsDom {
  int2* dom;        // &dom[0*BP + b]: this node's column of the [slot][BP] array; element = (-lb, ub)
  uint32_t bp;      // row stride in int2 (nodes per block + padding)
  uint32_t* chg;    // next-wave changed bitmask of this node [ceil(n_slots/32)]
  uint32_t* fail;   // block fail mask
  uint32_t fbit;    // this node's bit in *fail
  Ctr* c;
  SumTab sums;

  // LDS holds (-lb, ub): both narrowings are ds_min, and the sweep's no-op test becomes one v_add3_u32 per bound
  // (see fast_signs).  Bounds are below 2^29 in magnitude, so the negation cannot overflow.
  __device__ __forceinline__ int2 load_var(uint32_t v) const { const int2 d = dom[(size_t)v * bp]; return make_int2(-d.x, d.y); }
  __device__ __forceinline__ bool is_sum(uint32_t v) const { return v - sums.first < sums.count; }
  __device__ __forceinline__ bool any_sums() const { return sums.count != 0; }
  __device__ __forceinline__ const int32_t* mul_offsets() const { return sums.mul_off; }
  __device__ __forceinline__ int2 load(uint32_t v) const { return is_sum(v) ? sum_read(*this, sums, v) : load_var(v); }
  __device__ __forceinline__ void mark(uint32_t v) const { atomicOr(&chg[v >> 5], 1u << (v & 31)); }
  __device__ __forceinline__ void set_fail() const { atomicOr(fail, fbit); }
  // lb := max(lb, nlb).  Called only when nlb exceeds the lb this thread read.
  __device__ __forceinline__ void raise_lb(uint32_t v, int nlb) const {
    // Sum::update with several members (term/sum.rs:66-69): no pruning, the new domain only has to overlap the sum
    if (is_sum(v)) { if (nlb > sum_read(*this, sums, v).y) set_fail(); return; }
    int2* p = dom + (size_t)v * bp;
    const int old = atomicMin(&p->x, -nlb);
    if (old > -nlb) {
      ++c->narrow;
      mark(v);
      const int ub = __hip_atomic_load(&p->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (nlb > ub) set_fail();
    }
  }
  __device__ __forceinline__ void lower_ub(uint32_t v, int nub) const {
    if (is_sum(v)) { if (nub < sum_read(*this, sums, v).x) set_fail(); return; }
    int2* p = dom + (size_t)v * bp;
    const int old = atomicMin(&p->y, nub);
    if (old > nub) {
      ++c->narrow;
      mark(v);
      const int nlb = __hip_atomic_load(&p->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (-nlb > nub) set_fail();
    }
  }
};

// ------------------------------------------------------------------------------------------------
// Packed variant of LdsDom for tiles whose every bound lies within +-kPackedMax: one dword per (slot,node),
// low half = -lb, high half = ub (both int16).  Half the LDS bytes and half the VALU work in the sweep's level-1 test
// (v_pk_add_u16 / v_pk_min_i16, see fast_signs).  Narrowing one half is a compare-and-swap loop (LDS has no 16-bit
// atomics); narrowings are rare next to tests.  A bound that crosses the other one is clamped to lb = ub + 1 resp.
// ub = lb - 1 so that it stays representable: the node is failed, and a failed node's domains are unspecified.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pack16(int lb, int ub) { return ((uint32_t)ub << 16) | ((uint32_t)(-lb) & 0xffffu); }
__device__ __forceinline__ int2 unpack16(uint32_t c) { return make_int2(-(int)(short)(c & 0xffffu), (int)c >> 16); }  // (lb, ub)

struct LdsDom16 {
  uint32_t* dom;    // &dom[0*BP + b]
  uint32_t bp;      // row stride in dwords
  uint32_t* chg;
  uint32_t* fail;
  uint32_t fbit;
  Ctr* c;

  // (packed tiles are binary-only models without Sum views: the compact record stream excludes them)
  __device__ __forceinline__ bool any_sums() const { return false; }
  __device__ __forceinline__ const int32_t* mul_offsets() const { return nullptr; }  // (packed tiles: binary models only)
  __device__ __forceinline__ int2 load(uint32_t v) const { return unpack16(dom[(size_t)v * bp]); }
  __device__ __forceinline__ void mark(uint32_t v) const { atomicOr(&chg[v >> 5], 1u << (v & 31)); }
  __device__ __forceinline__ void set_fail() const { atomicOr(fail, fbit); }
  __device__ __forceinline__ void raise_lb(uint32_t v, int nlb) const {
    uint32_t* p = dom + (size_t)v * bp;
    uint32_t old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (;;) {
      const int2 d = unpack16(old);
      if (nlb <= d.x) return;  // somebody else got there first
      const uint32_t neu = pack16(min(nlb, d.y + 1), d.y);
      const uint32_t prev = atomicCAS(p, old, neu);
      if (prev == old) {
        ++c->narrow;
        mark(v);
        if (nlb > d.y) set_fail();
        return;
      }
      old = prev;
    }
  }
  __device__ __forceinline__ void lower_ub(uint32_t v, int nub) const {
    uint32_t* p = dom + (size_t)v * bp;
    uint32_t old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (;;) {
      const int2 d = unpack16(old);
      if (nub >= d.y) return;
      const uint32_t neu = pack16(d.x, max(nub, d.x - 1));
      const uint32_t prev = atomicCAS(p, old, neu);
      if (prev == old) {
        ++c->narrow;
        mark(v);
        if (nub < d.x) set_fail();
        return;
      }
      old = prev;
    }
  }
};

// ------------------------------------------------------------------------------------------------
// Domain access policy for variable stores that do not fit in LDS (BASELINE config 3: 50 000 variables =
// 400 KB per node): the node's (lb,ub) rows stay in HBM/L2 — the lb_out/ub_out arrays themselves — and are
// narrowed with device-scope atomicMax/atomicMin.  Reads are relaxed agent-scope atomic loads (global_load … sc1):
// they bypass the per-CU L1, which is never refreshed by atomics performed at L2, so a workgroup always sees
// its own and its team-mates' narrowings.  Constants (pseudo-variable slots >= n_vars) sit in a small LDS array.
// A failure can be missed at the moment it happens (two narrowings of one variable racing on different CUs);
// the final scan of the node's domains (phase 4) catches it.
// ------------------------------------------------------------------------------------------------
struct GlobalDom {
  int32_t* lb;      // [n_vars] this node's rows
  int32_t* ub;
  int2* cdom;       // LDS: constants, indexed by slot - n_vars
  uint32_t n_vars;
  uint32_t* chg;
  uint32_t* fail;
  uint32_t fbit;
  Ctr* c;
  SumTab sums;
  // dom10: the variable store is in LDS after all — 10-bit cells (lb - lo10) | (ub - lo10) << 10, three per u64 — because the
  // declared hull has at most 1024 values (config 3: 50 000 variables over [0, 999] = 130 KB).  Same policy interface: the sweep
  // and the rounds do not know the difference; a narrowing is a CAS on the cell's word.
  unsigned long long* c10;
  int lo10;

  __device__ __forceinline__ static uint32_t word3(uint32_t v) { return __umulhi(v, 0xAAAAAAABu) >> 1; }  // v / 3
  __device__ __forceinline__ int2 load10(uint32_t v) const {
    const uint32_t w = word3(v), sh = (v - 3u * w) * 20u;
    const uint32_t cell = (uint32_t)(c10[w] >> sh) & 0xFFFFFu;
    return make_int2(lo10 + (int)(cell & 1023u), lo10 + (int)(cell >> 10));
  }
  // narrow one bound of the cell of v: which = 0 raises lb to nv, which = 1 lowers ub to nv (values relative to lo10)
  __device__ __forceinline__ void narrow10(uint32_t v, int nv, int which) const {
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

  __device__ __forceinline__ bool is_sum(uint32_t v) const { return v - sums.first < sums.count; }
  __device__ __forceinline__ bool any_sums() const { return sums.count != 0; }
  __device__ __forceinline__ const int32_t* mul_offsets() const { return sums.mul_off; }
  __device__ __forceinline__ int2 load_var(uint32_t v) const {
    if (c10) return load10(v);
    int2 d;
    d.x = __hip_atomic_load(&lb[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    d.y = __hip_atomic_load(&ub[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return d;
  }
  __device__ __forceinline__ int2 load(uint32_t v) const {
    if (is_sum(v)) return sum_read(*this, sums, v);
    if (v >= n_vars) return cdom[v - n_vars];
    if (c10) return load10(v);
    int2 d;
    d.x = __hip_atomic_load(&lb[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    d.y = __hip_atomic_load(&ub[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return d;
  }
  __device__ __forceinline__ void mark(uint32_t v) const { atomicOr(&chg[v >> 5], 1u << (v & 31)); }
  __device__ __forceinline__ void set_fail() const { atomicOr(fail, fbit); }
  __device__ __forceinline__ void raise_lb(uint32_t v, int nlb) const {
    if (is_sum(v)) { if (nlb > sum_read(*this, sums, v).y) set_fail(); return; }
    if (v >= n_vars) { if (nlb > cdom[v - n_vars].y) set_fail(); return; }  // Constant::update (term/constant.rs:49-52)
    if (c10) { narrow10(v, nlb - lo10, 0); return; }
    const int old = atomicMax(&lb[v], nlb);
    if (old < nlb) {
      ++c->narrow;
      mark(v);
      if (nlb > __hip_atomic_load(&ub[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) set_fail();
    }
  }
  __device__ __forceinline__ void lower_ub(uint32_t v, int nub) const {
    if (is_sum(v)) { if (nub < sum_read(*this, sums, v).x) set_fail(); return; }
    if (v >= n_vars) { if (nub < cdom[v - n_vars].x) set_fail(); return; }
    if (c10) { narrow10(v, nub - lo10, 1); return; }
    const int old = atomicMin(&ub[v], nub);
    if (old > nub) {
      ++c->narrow;
      mark(v);
      if (__hip_atomic_load(&lb[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > nub) set_fail();
    }
  }
};

__device__ __forceinline__ int clamp_i32(long long v) {
  const long long lo = -2147483647LL, hi = 2147483647LL;
  return (int)(v < lo ? lo : (v > hi ? hi : v));
}

// x < y + z + d      (XLessYPlusZ::propagate, x_less_y_plus_z.rs:105-119; all three updates from pre-read values)
template <class D>
__device__ __forceinline__ void filter_lt3(int2& X, int2& Y, int2& Z, uint32_t x, uint32_t y, uint32_t z, long long d, const D& dm) {
  int nxu = min(X.y, clamp_i32((long long)Y.y + Z.y + d - 1));
  int nyl = max(Y.x, clamp_i32((long long)X.x - Z.y - d + 1));
  int nzl = max(Z.x, clamp_i32((long long)X.x - Y.y - d + 1));
  if (nxu < X.y) { X.y = nxu; dm.lower_ub(x, nxu); }
  if (nyl > Y.x) { Y.x = nyl; dm.raise_lb(y, nyl); }
  if (nzl > Z.x) { Z.x = nzl; dm.raise_lb(z, nzl); }
  if (X.x > X.y || Y.x > Y.y || Z.x > Z.y) dm.set_fail();
}
// x > y + z + d      (XGreaterYPlusZ::propagate, x_greater_y_plus_z.rs:106-118)
template <class D>
__device__ __forceinline__ void filter_gt3(int2& X, int2& Y, int2& Z, uint32_t x, uint32_t y, uint32_t z, long long d, const D& dm) {
  int nxl = max(X.x, clamp_i32((long long)Y.x + Z.x + d + 1));
  int nyu = min(Y.y, clamp_i32((long long)X.y - Z.x - d - 1));
  int nzu = min(Z.y, clamp_i32((long long)X.y - Y.x - d - 1));
  if (nxl > X.x) { X.x = nxl; dm.raise_lb(x, nxl); }
  if (nyu < Y.y) { Y.y = nyu; dm.lower_ub(y, nyu); }
  if (nzu < Z.y) { Z.y = nzu; dm.lower_ub(z, nzu); }
  if (X.x > X.y || Y.x > Y.y || Z.x > Z.y) dm.set_fail();
}

// One filter step = propagate() + is_subsumed() of one elementary propagator (store.rs:166-183).
// Returns true when the propagator is entailed (SKleene::True) under the domains it leaves behind.
template <class D>
__device__ __forceinline__ bool eval_record(const Rec& rec, const D& dm) {
  const uint32_t kind = rec.xk >> 28;
  const uint32_t x = rec.xk & kSlotMask, y = rec.y;
  const int d = rec.d;
  if (kind <= PCP_LT) {
    int2 X = dm.load(x), Y = dm.load(y);
    int Yl = Y.x + d, Yu = Y.y + d;  // Y as seen through Addition(y, d)  (term/addition.rs:98)
    if (kind == PCP_NEQ) {
      // XNeqY::propagate (x_neq_y.rs:82-93) with Interval::difference removing a value only at a bound.
      if (X.x == X.y) {
        const int v = X.x;
        if (v == Yl) { ++Yl; dm.raise_lb(y, Yl - d); }
        else if (v == Yu) { --Yu; dm.lower_ub(y, Yu - d); }
      } else if (Yl == Yu) {
        const int v = Yl;
        if (v == X.x) { ++X.x; dm.raise_lb(x, X.x); }
        else if (v == X.y) { --X.y; dm.lower_ub(x, X.y); }
      }
      if (X.x > X.y || Yl > Yu) dm.set_fail();
      // An update through a Sum of several variables narrows nothing (term/sum.rs:66-69), so with Sum views in the model
      // is_subsumed() is evaluated on the domains as they ARE after propagate(), not on the values computed above.
      if (dm.any_sums()) { X = dm.load(x); Y = dm.load(y); Yl = Y.x + d; Yu = Y.y + d; }
      // !XEqY::is_subsumed (x_neq_y.rs:71-73, x_eq_y.rs:87-93): True iff disjoint.
      return (X.x > Yu) || (Yl > X.y);
    } else if (kind == PCP_EQ) {
      // XEqY::propagate (x_eq_y.rs:102-107): both become x ∩ y.
      const int nl = max(X.x, Yl), nu = min(X.y, Yu);
      if (nl > X.x) dm.raise_lb(x, nl);
      if (nu < X.y) dm.lower_ub(x, nu);
      if (nl > Yl) dm.raise_lb(y, nl - d);
      if (nu < Yu) dm.lower_ub(y, nu - d);
      if (nl > nu) dm.set_fail();
      if (dm.any_sums()) { X = dm.load(x); Y = dm.load(y); return X.x == X.y && Y.x == Y.y && X.x == Y.x + d; }
      return nl == nu;  // x_eq_y.rs:87-88: both the same singleton
    } else {
      // XLessY::propagate (x_less_y.rs:104-109), both updates from the pre-read values.
      const int nxu = min(X.y, Yu - 1);
      const int nYl = max(Yl, X.x + 1);
      if (nxu < X.y) dm.lower_ub(x, nxu);
      if (nYl > Yl) dm.raise_lb(y, nYl - d);
      if (X.x > nxu || nYl > Yu) dm.set_fail();
      if (dm.any_sums()) { X = dm.load(x); Y = dm.load(y); return X.y < Y.x + d; }
      return nxu < nYl;  // x_less_y.rs:90-91: x.upper() < y.lower()
    }
  }
  const uint32_t z = rec.z;
  int2 X = dm.load(x), Y = dm.load(y), Z = dm.load(z);
  if (kind == PCP_LT3) {
    filter_lt3(X, Y, Z, x, y, z, (long long)d, dm);
    if (dm.any_sums()) { X = dm.load(x); Y = dm.load(y); Z = dm.load(z); }
    return (long long)X.y < (long long)Y.x + Z.x + d;  // x_less_y_plus_z.rs:90-91
  } else if (kind == PCP_GT3) {
    filter_gt3(X, Y, Z, x, y, z, (long long)d, dm);
    if (dm.any_sums()) { X = dm.load(x); Y = dm.load(y); Z = dm.load(z); }
    return (long long)X.x > (long long)Y.y + Z.y + d;  // x_greater_y_plus_z.rs:91-92
  } else if (kind == PCP_EQ3) {
    // XEqYPlusZ = geq.propagate() && leq.propagate() (x_eq_y_plus_z.rs:85-87): leq reads what geq left.
    // geq: (x+1) > y+z  <=>  x > y+z+(d-1);   leq: (x-1) < y+z  <=>  x < y+z+(d+1)   (cmp/mod.rs:62-86)
    filter_gt3(X, Y, Z, x, y, z, (long long)d - 1, dm);
    if (dm.any_sums()) { X = dm.load(x); Y = dm.load(y); Z = dm.load(z); }  // a Sum operand was not narrowed
    filter_lt3(X, Y, Z, x, y, z, (long long)d + 1, dm);
    if (dm.any_sums()) { X = dm.load(x); Y = dm.load(y); Z = dm.load(z); }
    const bool geq_true = (long long)X.x > (long long)Y.y + Z.y + d - 1;
    const bool leq_true = (long long)X.y < (long long)Y.x + Z.x + d + 1;
    return geq_true && leq_true;  // Kleene and (x_eq_y_plus_z.rs:65-67)
  } else {
    // XEqYMulZ (x_eq_y_mul_z.rs:99-105): x := x ∩ (y·z) through the operands' Addition views: (x + dx) = (y + dy)·(z + dz);
    // the three offsets sit in a side table indexed by the record's `d` field.
    const int32_t* mo = dm.mul_offsets() + 3 * (size_t)d;
    const long long dx = mo[0], dy = mo[1], dz = mo[2];
    const long long yl = Y.x + dy, yu = Y.y + dy, zl = Z.x + dz, zu = Z.y + dz;
    const long long p0 = yl * zl, p1 = yl * zu, p2 = yu * zl, p3 = yu * zu;
    const long long pl = min(min(p0, p1), min(p2, p3)), pu = max(max(p0, p1), max(p2, p3));
    const int nl = max(X.x, clamp_i32(pl - dx)), nu = min(X.y, clamp_i32(pu - dx));
    if (nl > X.x) dm.raise_lb(x, nl);
    if (nu < X.y) dm.lower_ub(x, nu);
    if (nl > nu) dm.set_fail();
    return pl == pu && nl == nu;  // x_eq_y_mul_z.rs:81-86
  }
}

// ------------------------------------------------------------------------------------------------
// Block-wide exclusive scan of one value per thread (wave shuffle scan + LDS for wave totals).
// `tmp` has >= 33 words.  Returns the exclusive prefix; *total_out (LDS) holds the grand total after return.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* tmp, uint32_t* total_out) {
  const uint32_t lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps the loop control scalar
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    uint32_t t = __shfl_up(inc, o);
    if (lane >= (uint32_t)o) inc += t;
  }
  if (lane == 63) tmp[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    uint32_t w = lane < nw ? tmp[lane] : 0u;
    uint32_t winc = w;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      uint32_t t = __shfl_up(winc, o);
      if (lane >= (uint32_t)o) winc += t;
    }
    if (lane < nw) tmp[lane] = winc - w;  // exclusive wave offsets
    if (lane == nw - 1) *total_out = winc;
  }
  __syncthreads();
  return tmp[wave] + inc - v;
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t l) {
  uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
  uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l);
  return ((uint64_t)hi << 32) | lo;
}

// lane l of the result takes the wave-uniform value v, the other lanes keep `old`
__device__ __forceinline__ uint64_t writelane64(uint64_t old, uint64_t v, uint32_t l) {
  return ((threadIdx.x & 63u) == l) ? v : old;
}

}  // namespace
}  // namespace pcp
