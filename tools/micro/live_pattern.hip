// Microbenchmark: the sweep's live-mask traffic pattern in isolation (copy in -> out of [nodes][words] u64 rows).
//   mode 0: lane = node*4 + j, one 8-byte load/store per lane per 4-word chunk   (32 contiguous bytes per node)
//   mode 1: lane = node*4 + q, 32 bytes per lane per 16-word chunk               (128 contiguous bytes per node)
// 16 nodes per workgroup, 16 wavefronts per workgroup, chunks dealt round-robin to the wavefronts, DEPTH loads in flight.
// build: hipcc --offload-arch=gfx950 -O3 -o live_pattern live_pattern.hip ; run: ./live_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>

template <int DEPTH>
__global__ void __launch_bounds__(1024) k_mode0(const uint64_t* in, uint64_t* out, uint32_t words) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const uint32_t b = lane >> 2, j = lane & 3;
  const uint64_t* src = in + (size_t)(blockIdx.x * 16 + b) * words;
  uint64_t* dst = out + (size_t)(blockIdx.x * 16 + b) * words;
  const uint32_t chunks = words / 4;
  uint64_t buf[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) { uint32_t c = wave + d * nw; buf[d] = src[min(c, chunks - 1) * 4 + j]; }
  for (uint32_t c = wave; c < chunks; c += nw * DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const uint32_t cc = c + d * nw;
      const uint64_t v = buf[d];
      const uint32_t cn = cc + nw * DEPTH;
      buf[d] = src[min(cn, chunks - 1) * 4 + j];
      if (cc < chunks) dst[cc * 4 + j] = v ^ (v >> 63);
    }
  }
}
template <int DEPTH>
__global__ void __launch_bounds__(1024) k_mode1(const uint64_t* in, uint64_t* out, uint32_t words) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const uint32_t b = lane >> 2, q = lane & 3;
  const ulonglong4* src = reinterpret_cast<const ulonglong4*>(in + (size_t)(blockIdx.x * 16 + b) * words);
  ulonglong4* dst = reinterpret_cast<ulonglong4*>(out + (size_t)(blockIdx.x * 16 + b) * words);
  const uint32_t chunks = words / 16;
  ulonglong4 buf[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) { uint32_t c = wave + d * nw; buf[d] = src[min(c, chunks - 1) * 4 + q]; }
  for (uint32_t c = wave; c < chunks; c += nw * DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const uint32_t cc = c + d * nw;
      ulonglong4 v = buf[d];
      const uint32_t cn = cc + nw * DEPTH;
      buf[d] = src[min(cn, chunks - 1) * 4 + q];
      v.x ^= v.x >> 63;
      if (cc < chunks) dst[cc * 4 + q] = v;
    }
  }
}

// mode 2: the word-group sweep's pattern, read only: a wavefront step = 64 consecutive words of each of 16 node rows
// (16 coalesced 512-byte loads in flight), popcounts summed.
template <int ROWS>
__global__ void __launch_bounds__(1024) k_mode2(const uint64_t* in, uint32_t* out, uint32_t words) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const uint32_t groups = (words + 63) / 64;
  uint32_t acc = 0;
  for (uint32_t g = wave; g < groups; g += nw) {
    const uint32_t w = min(g * 64 + lane, words - 1);
    uint64_t v[ROWS];
#pragma unroll
    for (int b = 0; b < ROWS; ++b) v[b] = in[(size_t)(blockIdx.x * ROWS + b) * words + w];
#pragma unroll
    for (int b = 0; b < ROWS; ++b) acc += __popcll(v[b]);
  }
  if (acc == 0xFFFFFFFFu) out[0] = acc;
}

int main(int argc, char** argv) {
  const uint32_t nodes = 4096; const uint32_t words = argc > 1 ? atoi(argv[1]) : 23424;  // 23424 = 16 * 1464: rows 128-byte aligned
  const size_t n = (size_t)nodes * words;
  uint64_t *in, *out;
  hipMalloc(&in, n * 8); hipMalloc(&out, n * 8);
  hipMemset(in, 0x5a, n * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-28s %.3f ms  %.2f TB/s (read+write)\n", name, ms, 2.0 * n * 8 / ms / 1e9);
  };
  run("mode0 32B/node depth1", [&] { hipLaunchKernelGGL(k_mode0<1>, dim3(nodes / 16), dim3(1024), 0, 0, in, out, words); });
  run("mode0 32B/node depth2", [&] { hipLaunchKernelGGL(k_mode0<2>, dim3(nodes / 16), dim3(1024), 0, 0, in, out, words); });
  run("mode0 32B/node depth4", [&] { hipLaunchKernelGGL(k_mode0<4>, dim3(nodes / 16), dim3(1024), 0, 0, in, out, words); });
  run("mode1 128B/node depth1", [&] { hipLaunchKernelGGL(k_mode1<1>, dim3(nodes / 16), dim3(1024), 0, 0, in, out, words); });
  run("mode1 128B/node depth2", [&] { hipLaunchKernelGGL(k_mode1<2>, dim3(nodes / 16), dim3(1024), 0, 0, in, out, words); });
  run("mode1 128B/node depth4", [&] { hipLaunchKernelGGL(k_mode1<4>, dim3(nodes / 16), dim3(1024), 0, 0, in, out, words); });
  auto run_ro = [&](const char* name, auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-28s %.3f ms  %.2f TB/s (read only)\n", name, ms, 1.0 * n * 8 / ms / 1e9);
  };
  run_ro("mode2 16 rows x 512B", [&] { hipLaunchKernelGGL(k_mode2<16>, dim3(nodes / 16), dim3(1024), 0, 0, in, (uint32_t*)out, words); });
  run_ro("mode2 8 rows x 512B", [&] { hipLaunchKernelGGL(k_mode2<8>, dim3(nodes / 8), dim3(1024), 0, 0, in, (uint32_t*)out, words); });
  run_ro("mode2 8 rows, 512 thr", [&] { hipLaunchKernelGGL(k_mode2<8>, dim3(nodes / 8), dim3(512), 0, 0, in, (uint32_t*)out, words); });
  hipMemcpyAsync(out, in, n * 8, hipMemcpyDeviceToDevice, 0); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipMemcpyAsync(out, in, n * 8, hipMemcpyDeviceToDevice, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  printf("%-28s %.3f ms  %.2f TB/s (read+write)\n", "hipMemcpy D2D", ms, 2.0 * n * 8 / ms / 1e9);
  return 0;
}
