// stream_probe.hip — the launch structure of pcp_neq.hip's streaming form, on its own, with a synthetic compute phase: does a CU that keeps
// LOADER wavefronts streaming rows into one LDS tile while COMPUTE wavefronts work on the other hide the staging behind the compute?
// Stand-alone (hipcc, gfx950); prints one line per variant.  Same data, cells and checksum as stage_probe.hip.
//   one 1024-thread workgroup per CU, persistent over its tiles (tile t = 16 nodes of N-queens-1000: 128 KB of i32 rows -> 68 KB of
//   16-bit packed node-minor cells);
//   wavefronts 0..5 = compute group 0, 6..11 = compute group 1, 12..15 = loaders;
//   TWO tile buffers in LDS: the workgroup's j-th tile goes to buffer / group j & 1.  Loaders: row loads (raw_buffer-free plain 16-byte
//   loads here), UFL pairs in flight per lane in two register stages, pack, ds_write; hand-over by two monotonic LDS counters per group
//   (`ready`: loader wavefronts that finished the tile; `freed`: compute wavefronts that are through with it) — no s_barrier after the
//   prologue, so neither side ever waits at a barrier for the other;
//   compute: wait for `ready`, spin D shader cycles (the rounds + status scan of a frontier tile), read every cell once (checksum).
// usage: stream_probe [nodes=16384] [D=20000] [reps=5]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr uint32_t V = 1000, B = 16, SH = 2, SQ = V / 4;
__host__ __device__ inline uint32_t rowof(uint32_t slot) { return slot * B + ((slot >> SH) << 2); }
constexpr uint32_t kCellBytes = ((1000 * 16 + (1000 >> 2) * 4 + 4) * 4 + 15) & ~15u;  // 68 016
constexpr uint32_t kSpinCap = 1u << 24;

__device__ __forceinline__ void spin(uint64_t cycles) {
  if (!cycles) return;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(2);
}
__device__ __forceinline__ bool wait_ge(uint32_t* p, uint32_t want) {
  for (uint32_t i = 0; i < kSpinCap; ++i) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;  // (never hang the box: a protocol bug shows as a wrong checksum)
}

template <int NLOAD, int UFL>
__global__ void __launch_bounds__(1024) stream_tiles(const int32_t* __restrict__ lb, const int32_t* __restrict__ ub, uint32_t n_nodes, uint64_t D,
                                                     unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* const ctl = reinterpret_cast<uint32_t*>(smem + 2 * kCellBytes);  // ready[2], freed[2], sum[2 x u64]
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = blockDim.x >> 6;
  const uint32_t ncomp = nwv - NLOAD, gsz = ncomp / 2;
  const uint32_t n_tiles = (n_nodes + B - 1) / B;
  if (tid < 8) ctl[tid] = 0;
  __syncthreads();
  const uint32_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - 1 - blockIdx.x) / gridDim.x + 1 : 0;
  if (wv >= ncomp) {
    // ---------------------------------------------------------------- loader wavefronts
    const uint32_t lw = wv - ncomp;
    const uint32_t lb4 = lane >> 4, lq = lane & 15u;
    constexpr uint32_t QC = (SQ + 15) / 16;        // 16 chunks of 16 quads
    constexpr uint32_t WT = 4 * QC;                // wave-tasks per tile: (node group of 4) x (chunk)
    // ONE sequence of wave-tasks over all of this workgroup's tiles: task i = (tile i / TPW, wave-task lw + NLOAD (i % TPW)); two register
    // stages of UFL pairs each, so that the loads of stage k + 1 are in flight while stage k is packed and written — also across the
    // seam between two tiles.  A tile's cells are written only once its buffer is free; its hand-over follows its last write.
    constexpr uint32_t TPW = WT / NLOAD;  // wave-tasks per tile and loader wavefront
    const uint32_t total = my_tiles * TPW;
    auto coords = [&](uint32_t i, uint32_t& j, uint32_t& b, uint32_t& q, uint32_t& nb, bool& on) {
      j = i / TPW;
      const uint32_t w = lw + NLOAD * (i - j * TPW), ng = w / QC, qc = w - ng * QC;
      const uint32_t t = blockIdx.x + j * gridDim.x;
      nb = min(B, n_nodes - min(t * B, n_nodes));
      b = 4 * ng + lb4; q = 16 * qc + lq;
      on = i < total && q < SQ && b < nb;
    };
    auto load = [&](uint32_t i0, int4 (&L)[UFL], int4 (&U)[UFL]) {
#pragma unroll
      for (int k = 0; k < UFL; ++k) {
        uint32_t j, b, q, nb; bool on;
        coords(i0 + k, j, b, q, nb, on);
        const uint32_t t = blockIdx.x + j * gridDim.x;
        const size_t off = on ? ((size_t)t * B + b) * V + 4 * q : 0;
        L[k] = *reinterpret_cast<const int4*>(lb + off);
        U[k] = *reinterpret_cast<const int4*>(ub + off);
      }
    };
    uint32_t freed_for = 2;  // tiles 0 and 1 find their buffers free
    auto put = [&](uint32_t i0, const int4 (&L)[UFL], const int4 (&U)[UFL]) {
#pragma unroll
      for (int k = 0; k < UFL; ++k) {
        uint32_t j, b, q, nb; bool on;
        coords(i0 + k, j, b, q, nb, on);
        if (i0 + k >= total) break;
        if (j >= freed_for) { wait_ge(&ctl[2 + (j & 1u)], gsz * (j / 2)); freed_for = j + 1; }
        if (on) {
          uint32_t* p0 = reinterpret_cast<uint32_t*>(smem + (j & 1u) * kCellBytes) + 68u * q + b;
          const int l[4] = {L[k].x, L[k].y, L[k].z, L[k].w}, u[4] = {U[k].x, U[k].y, U[k].z, U[k].w};
#pragma unroll
          for (int i = 0; i < 4; ++i) p0[i * B] = __builtin_amdgcn_perm((uint32_t)u[i], (uint32_t)(-l[i]), 0x05040100u);
        }
        if ((i0 + k) % TPW == TPW - 1) {  // this wavefront's last task of tile j
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) atomicAdd(&ctl[j & 1u], 1u);
        }
      }
    };
    int4 LA[UFL], UA[UFL], LB[UFL], UB[UFL];
    load(0, LA, UA);
    for (uint32_t i0 = 0; i0 < total; i0 += 2 * UFL) {
      load(i0 + UFL, LB, UB);
      put(i0, LA, UA);
      load(i0 + 2 * UFL, LA, UA);
      put(i0 + UFL, LB, UB);
    }
    return;
  }
  // ------------------------------------------------------------------ compute groups
  const uint32_t g = wv / gsz, gw = wv - g * gsz, gt = gw * 64 + lane, gn = gsz * 64;
  if (g >= 2) return;
  const uint32_t* const dom = reinterpret_cast<const uint32_t*>(smem + g * kCellBytes);
  unsigned long long acc = 0;
  uint32_t k = 0;
  for (uint32_t j = g; j < my_tiles; j += 2, ++k) {
    const uint32_t t = blockIdx.x + j * gridDim.x, nb = min(B, n_nodes - t * B);
    wait_ge(&ctl[g], NLOAD * (k + 1));
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    spin(D);
    for (uint32_t i = gt; i < V * B; i += gn) { const uint32_t v = i / B, b = i % B; if (b < nb) { const uint32_t c = dom[rowof(v) + b]; acc += c; acc += (unsigned long long)(((c + (c >> 16)) & 0xffffu) == 0u) << 20; } }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) atomicAdd(&ctl[2 + g], 1u);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if (lane == 0) atomicAdd(&out[blockIdx.x], acc);
}

int main(int argc, char** argv) {
  const uint32_t N = argc > 1 ? (uint32_t)atoi(argv[1]) : 16384;
  const uint64_t D = argc > 2 ? (uint64_t)atoll(argv[2]) : 20000;
  const int reps = argc > 3 ? atoi(argv[3]) : 5;
  const size_t n = (size_t)N * V;
  std::vector<int32_t> hl(n), hu(n);
  uint64_t x = 88172645463325252ull;
  for (size_t i = 0; i < n; ++i) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    const int l = 1 + (int)(x % 7), u = (x >> 20) % 97 == 0 ? l : 1000 - (int)((x >> 8) % 5);
    hl[i] = l; hu[i] = u;
  }
  int32_t *dl, *du;
  unsigned long long* dout;
  CK(hipMalloc(&dl, n * 4)); CK(hipMalloc(&du, n * 4)); CK(hipMalloc(&dout, 8 * 4096));
  CK(hipMemcpy(dl, hl.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(du, hu.data(), n * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
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
    printf("%-44s best %8.1f us  mean %8.1f us  %6.2f TB/s (best)  checksum %llu\n", name, best * 1e3, sum / reps * 1e3, bytes / (best * 1e-3) / 1e12, h);
  };
  const size_t lds = 2 * kCellBytes + 64;
#define VARIANT(NL, UF) { \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_tiles<NL, UF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    char nm[96]; snprintf(nm, sizeof nm, "v3 stream %d loaders x %d pairs, D=%llu", NL, UF, (unsigned long long)D); \
    run(nm, [&] { hipLaunchKernelGGL((stream_tiles<NL, UF>), dim3(256), dim3(1024), lds, 0, dl, du, N, D, dout); }); }
  VARIANT(4, 4) VARIANT(4, 2) VARIANT(2, 4) VARIANT(4, 1)
  return 0;
}
