// stage_probe.hip — how fast can the rows of an N-queens-1000 batch (lb[n][1000], ub[n][1000] i32) be brought into LDS as 16-bit packed
// node-minor cells, and what does overlapping that with a tile's compute phase buy?  Stand-alone (hipcc, gfx950); prints one line per
// variant.  Not part of the library: a measuring stick for pcp_neq.hip's staging.
//   v0  read ceiling: every 16-byte quad of both arrays read once, 8 loads in flight per lane, nothing stored
//   v1  the tile kernel's staging as it is: grid = tiles, 512 threads, 81 KB LDS (two per CU), 8 row loads in flight, pack, ds_write
//   v2  persistent, one 1024-thread workgroup per CU: wave 15 is a LOADER issuing global_load_lds_dwordx4 (LDS-DMA, inline asm, not
//       counted by hipcc) into a ring of 8 raw node slots; the other waves pack ring -> cells; flags in LDS; tile k+1's first half
//       lands while tile k "computes"
//   D   synthetic compute per tile in shader cycles (all waves spin), to model the rounds + status phases
// usage: stage_probe [nodes=16384] [D=0] [reps=5]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr uint32_t V = 1000, B = 16, SH = 2, SQ = V / 4;
__host__ __device__ inline uint32_t rowof(uint32_t slot) { return slot * B + ((slot >> SH) << 2); }
constexpr uint32_t kCellBytes = ((1000 * 16 + (1000 >> 2) * 4 + 4) * 4 + 15) & ~15u;  // 68 016 -> 68 016
__device__ __forceinline__ uint32_t pack16(int l, int u) { return ((uint32_t)(-l) & 0xffffu) | ((uint32_t)u << 16); }

__device__ __forceinline__ void spin(uint64_t cycles) {
  if (!cycles) return;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(2);
}

// one plain store per workgroup (8192 same-address device atomics alone cost ~100 us: they were this probe's first result)
__device__ __forceinline__ void block_sum(unsigned long long acc, unsigned long long* out) {
  __shared__ unsigned long long tot;
  if (threadIdx.x == 0) tot = 0;
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(&tot, acc);
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------------------------------------------- v0
__global__ void __launch_bounds__(256) read_ceiling(const int4* __restrict__ a, const int4* __restrict__ b, size_t nq, unsigned long long* out) {
  uint32_t acc = 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += 4 * stride) {
    int4 x[4], y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const size_t k = i + j * stride < nq ? i + j * stride : nq - 1; x[j] = a[k]; y[j] = b[k]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc += (uint32_t)(x[j].x ^ x[j].w ^ y[j].y ^ y[j].z);
  }
  if (acc == 0x12345678u) out[blockIdx.x] = 1ull;
}

// ---------------------------------------------------------------------------------------------------------------- v1
__global__ void __launch_bounds__(512) stage_tiles(const int32_t* __restrict__ lb, const int32_t* __restrict__ ub, uint32_t n_nodes, uint64_t D,
                                                   unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* dom = reinterpret_cast<uint32_t*>(smem);
  const uint32_t tid = threadIdx.x, nth = blockDim.x;
  const uint32_t node0 = blockIdx.x * B, nb = min(B, n_nodes - node0), tasks = nb * SQ;
  uint32_t sing = 0;
  for (uint32_t t0 = tid; t0 < tasks; t0 += 4 * nth) {
    int4 L[4], U[4];
    uint32_t bq[4], qq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t t = min(t0 + j * nth, tasks - 1);
      bq[j] = t / SQ; qq[j] = t - bq[j] * SQ;
      const size_t row = (size_t)(node0 + bq[j]) * V;
      L[j] = reinterpret_cast<const int4*>(lb + row)[qq[j]];
      U[j] = reinterpret_cast<const int4*>(ub + row)[qq[j]];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (t0 + j * nth >= tasks) break;
      const int l[4] = {L[j].x, L[j].y, L[j].z, L[j].w}, u[4] = {U[j].x, U[j].y, U[j].z, U[j].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) { dom[rowof(4 * qq[j] + i) + bq[j]] = pack16(l[i], u[i]); sing += l[i] == u[i]; }
    }
  }
  __syncthreads();
  spin(D);
  __syncthreads();
  unsigned long long acc = (unsigned long long)sing << 20;
  for (uint32_t i = tid; i < V * B; i += nth) { const uint32_t v = i / B, b = i % B; if (b < nb) acc += dom[rowof(v) + b]; }
  block_sum(acc, out);
}

// ---------------------------------------------------------------------------------------------------------------- v2
constexpr uint32_t kSlots = 8, kSlotBytes = 8192, kRowStride = 4096;
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)p; }

template <int DEPTH>
__global__ void __launch_bounds__(1024) stage_ring(const int32_t* __restrict__ lb, const int32_t* __restrict__ ub, uint32_t n_nodes, uint64_t D,
                                                   unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* const dom = reinterpret_cast<uint32_t*>(smem);
  unsigned char* const ring = smem + kCellBytes;
  uint32_t* const ready = reinterpret_cast<uint32_t*>(ring + kSlots * kSlotBytes);  // [8] fills landed in slot s
  uint32_t* const done = ready + kSlots;                                             // [8] consumer waves finished with slot s (cumulative)
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = blockDim.x >> 6;
  const bool loader = wv == nwv - 1;
  const uint32_t n_tiles = (n_nodes + B - 1) / B;
  if (tid < 2 * kSlots) ready[tid] = 0;
  __syncthreads();
  // consumer groups: waves {0-3}, {4-7}, {8-11}, {12-14}: group g packs the nodes i = g (mod 4) of a tile
  const uint32_t grp = min(wv >> 2, 3u), gw0 = grp * 4, gsz = grp == 3 ? nwv - 1 - 12 : 4, wig = wv - gw0;
  const uint32_t ring_base = lds_addr(ring);
  uint32_t seq = 0;   // nodes handed to the ring by this workgroup so far (the loader's count; consumers keep their own)
  uint32_t cseq = 0;  // consumer: nodes of earlier tiles (multiples of 16)
  unsigned long long acc = 0;
  // issue the DMA of node (tile t, node b) into slot s
  auto issue = [&](uint32_t t, uint32_t b, uint32_t s) {
    const uint32_t node = min(t * B + b, n_nodes - 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t q = min((uint32_t)(64 * j) + lane, SQ - 1);
      glds16(reinterpret_cast<const int4*>(lb + (size_t)node * V) + q, __builtin_amdgcn_readfirstlane(ring_base + s * kSlotBytes + j * 1024));
      glds16(reinterpret_cast<const int4*>(ub + (size_t)node * V) + q, __builtin_amdgcn_readfirstlane(ring_base + s * kSlotBytes + kRowStride + j * 1024));
    }
  };
  const uint32_t t_first = blockIdx.x;
  if (loader && t_first < n_tiles) {  // prologue: the first half of the first tile
    for (uint32_t b = 0; b < 8; ++b) issue(t_first, b, b);
  }
  for (uint32_t t = t_first; t < n_tiles; t += gridDim.x) {
    // ---- pack region: ring -> cells, all 16 nodes of tile t; the loader refills slots as they are released ------------------
    if (loader) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // first half landed (issued during the previous tile's compute)
      if (lane < 8) ready[lane] = seq / 8 + 1;            // fill number of every slot
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      for (uint32_t b = 8; b < 16; ++b) {
        const uint32_t s = b - 8, g = s & 3u, need = (seq / 8 + 1) * (g == 3 ? nwv - 1 - 12 : 4);
        while (__hip_atomic_load(&done[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
        issue(t, b, s);
        if (b >= 8 + DEPTH) {  // node b - DEPTH has landed (in-order returns): publish it
          asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 * DEPTH) : "memory");
          if (lane == 0) ready[b - DEPTH - 8] = seq / 8 + 2;
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane < 8) ready[lane] = seq / 8 + 2;
      seq += 16;
    } else {
      for (uint32_t b = grp; b < 16; b += 4) {
        const uint32_t s = b & 7u, want = cseq / 8 + 1 + (b >> 3);
        while (__hip_atomic_load(&ready[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const unsigned char* sl = ring + s * kSlotBytes;
        for (uint32_t q = wig * 64 + lane; q < SQ; q += gsz * 64) {
          const int4 L = *reinterpret_cast<const int4*>(sl + q * 16), U = *reinterpret_cast<const int4*>(sl + kRowStride + q * 16);
          const int l[4] = {L.x, L.y, L.z, L.w}, u[4] = {U.x, U.y, U.z, U.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) { dom[rowof(4 * q + i) + b] = pack16(l[i], u[i]); acc += (unsigned long long)(l[i] == u[i]) << 20; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) atomicAdd(&done[s], 1u);
      }
      cseq += 16;
    }
    __syncthreads();
    // ---- compute region: the next tile's first half is requested now and lands while this tile computes ----------------------
    const uint32_t tn = t + gridDim.x;
    if (loader && tn < n_tiles) {
      for (uint32_t b = 0; b < 8; ++b) issue(tn, b, b);  // (every slot was released in the pack region: the barrier above says so)
    }
    spin(D);
    const uint32_t nb = min(B, n_nodes - t * B);
    for (uint32_t i = tid; i < V * B; i += blockDim.x) { const uint32_t v = i / B, b = i % B; if (b < nb) acc += dom[rowof(v) + b]; }
    __syncthreads();
  }
  block_sum(acc, out);
}

int main(int argc, char** argv) {
  const uint32_t N = argc > 1 ? (uint32_t)atoi(argv[1]) : 16384;
  const uint64_t D = argc > 2 ? (uint64_t)atoll(argv[2]) : 0;
  const int reps = argc > 3 ? atoi(argv[3]) : 5;
  const size_t n = (size_t)N * V;
  std::vector<int32_t> hl(n), hu(n);
  uint64_t x = 88172645463325252ull;
  unsigned long long want = 0;
  for (size_t i = 0; i < n; ++i) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    const int l = 1 + (int)(x % 7), u = (x >> 20) % 97 == 0 ? l : 1000 - (int)((x >> 8) % 5);
    hl[i] = l; hu[i] = u;
  }
  // expected checksum per tile: sum of packed cells + singletons << 20, added up as u32 per thread then u64 — equal to a u32-wrapped sum only
  // per thread, so compare variants with each other through a wrap-free quantity instead: the count of singletons and the sum of (ub - lb)
  (void)want;
  int32_t *dl, *du;
  unsigned long long* dout;
  CK(hipMalloc(&dl, n * 4)); CK(hipMalloc(&du, n * 4)); CK(hipMalloc(&dout, 8 * 4096));
  CK(hipMemcpy(dl, hl.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(du, hu.data(), n * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint32_t tiles = (N + B - 1) / B;
  const double bytes = (double)n * 8;
  auto run = [&](const char* name, auto launch) {
    float best = 1e9f, sum = 0;
    unsigned long long h = 0;
    for (int r = 0; r < reps + 1; ++r) {
      CK(hipMemset(dout, 0, 8 * 4096));
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r) { best = ms < best ? ms : best; sum += ms; }
      std::vector<unsigned long long> hv(4096);
      CK(hipMemcpy(hv.data(), dout, 8 * 4096, hipMemcpyDeviceToHost));
      h = 0; for (auto v : hv) h += v;
    }
    printf("%-34s best %8.1f us  mean %8.1f us  %6.2f TB/s (best)  checksum %llu\n", name, best * 1e3, sum / reps * 1e3, bytes / (best * 1e-3) / 1e12, h);
  };
  run("v0 read ceiling", [&] { hipLaunchKernelGGL(read_ceiling, dim3(2048), dim3(256), 0, 0, (const int4*)dl, (const int4*)du, n / 4, dout); });
  {
    const size_t lds = 81664;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stage_tiles), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    char nm[64]; snprintf(nm, sizeof nm, "v1 tiles x512 (2/CU) D=%llu", (unsigned long long)D);
    run(nm, [&] { hipLaunchKernelGGL(stage_tiles, dim3(tiles), dim3(512), lds, 0, dl, du, N, D, dout); });
  }
  {
    const size_t lds = kCellBytes + kSlots * kSlotBytes + 64;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stage_ring<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stage_ring<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stage_ring<7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    char nm[64];
    snprintf(nm, sizeof nm, "v2 ring, loader wave, depth 2 D=%llu", (unsigned long long)D);
    run(nm, [&] { hipLaunchKernelGGL(stage_ring<2>, dim3(256), dim3(1024), lds, 0, dl, du, N, D, dout); });
    snprintf(nm, sizeof nm, "v2 ring, loader wave, depth 4 D=%llu", (unsigned long long)D);
    run(nm, [&] { hipLaunchKernelGGL(stage_ring<4>, dim3(256), dim3(1024), lds, 0, dl, du, N, D, dout); });
    snprintf(nm, sizeof nm, "v2 ring, loader wave, depth 7 D=%llu", (unsigned long long)D);
    run(nm, [&] { hipLaunchKernelGGL(stage_ring<7>, dim3(256), dim3(1024), lds, 0, dl, du, N, D, dout); });
  }
  return 0;
}
