// box_probe.hip — two ceilings of the box a bench runs on, measured next to the bench (bench.py loads libbox_probe.so through ctypes;
// __graft_entry__.build() compiles it).  NOT part of libpcp_hip.so: measuring sticks, no product path calls them.
//   box_stream_read_ms   a read-only streaming kernel, 16 bytes per lane and load, eight loads in flight per lane, persistent grid — the
//                        ceiling of "bring these bytes from HBM into registers once" ("the measured ceiling of a plain
//                        copy/read kernel on the box"), over the SAME two buffers a fixpoint launch reads;
//   box_valu_issue       integer VALU issue rate: independent v_add_u32 / v_pk_add_u16 chains on every SIMD at full occupancy — wave64
//                        instructions per second per CU, i.e. how many cycles a wave64 integer op costs a SIMD (2 or 4: the guide gives
//                        both readings).
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC box_probe.hip -o libbox_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

template <int NT>
__global__ void __launch_bounds__(256) stream_read_kernel(const v4i* __restrict__ a, const v4i* __restrict__ b, size_t nq, unsigned int* sink) {
  // each lane: quads i, i + stride, ...; four of a and four of b in flight
  unsigned int acc = 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < nq; i += 4 * stride) {
    v4i x[4], y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (NT) { x[j] = __builtin_nontemporal_load(a + i + j * stride); y[j] = __builtin_nontemporal_load(b + i + j * stride); }
      else { x[j] = a[i + j * stride]; y[j] = b[i + j * stride]; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc += (unsigned int)(x[j].x ^ x[j].y ^ x[j].z ^ x[j].w ^ y[j].x ^ y[j].y ^ y[j].z ^ y[j].w);
  }
  for (; i < nq; i += stride) { const v4i x = a[i], y = b[i]; acc += (unsigned int)(x.x ^ x.w ^ y.y ^ y.z); }
  if (acc == 0x9e3779b9u) sink[0] = acc;  // (never true for the bench's data; keeps the loads alive)
}

template <int PK>
__global__ void __launch_bounds__(1024) valu_issue_kernel(unsigned int iters, unsigned int* sink) {
  unsigned int r[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) r[k] = threadIdx.x * 8u + (unsigned int)k;
  const unsigned int c = blockIdx.x | 1u;
  for (unsigned int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (PK) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r[k]) : "v"(c));
        else asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[k]) : "v"(c));
      }
    }
  }
  unsigned int s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s ^= r[k];
  if (s == 0x12345u) sink[0] = s;
}

}  // namespace

extern "C" {

// Average milliseconds of one pass over a[0..bytes_each) and b[0..bytes_each) (both 16-byte aligned device pointers), `reps` passes timed by
// one pair of HIP events on `stream`, one untimed pass first.  grid_per_cu workgroups of 256 threads per CU (0 = 8).  < 0 on error.
float box_stream_read_ms(const void* a, const void* b, size_t bytes_each, int reps, int grid_per_cu, int nontemporal, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1.f;
  unsigned int* sink = nullptr;
  if (hipMalloc(&sink, 4) != hipSuccess) return -1.f;
  const size_t nq = bytes_each / 16;
  const unsigned int grid = (unsigned int)cus * (unsigned int)(grid_per_cu > 0 ? grid_per_cu : 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&]() {
    if (nontemporal) hipLaunchKernelGGL(stream_read_kernel<1>, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const v4i*>(a), reinterpret_cast<const v4i*>(b), nq, sink);
    else hipLaunchKernelGGL(stream_read_kernel<0>, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const v4i*>(a), reinterpret_cast<const v4i*>(b), nq, sink);
  };
  launch();
  hipEventRecord(e0, stream);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1, stream);
  float ms = -1.f;
  if (hipEventSynchronize(e1) == hipSuccess) { hipEventElapsedTime(&ms, e0, e1); ms /= (float)(reps > 0 ? reps : 1); }
  hipEventDestroy(e0); hipEventDestroy(e1);
  hipFree(sink);
  return hipGetLastError() == hipSuccess ? ms : -1.f;
}

// Wave64 integer VALU instructions per second and CU (packed = 0: v_add_u32, 1: v_pk_add_u16), with `waves_per_simd` resident wavefronts per
// SIMD (1..4 with this kernel's 1024-thread workgroups: one workgroup per CU = 4 per SIMD; smaller values launch fewer threads per workgroup).
// Also returns, through *cycles_per_inst, SIMD cycles per wave instruction at the clock *mhz the runtime reports.
double box_valu_issue(int packed, int waves_per_simd, double* cycles_per_inst, double* mhz, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  int dev = 0, cus = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1.0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev);
  unsigned int* sink = nullptr;
  if (hipMalloc(&sink, 4) != hipSuccess) return -1.0;
  const int wps = waves_per_simd < 1 ? 1 : waves_per_simd > 4 ? 4 : waves_per_simd;
  const unsigned int block = 256u * (unsigned int)wps, iters = 20000u;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&](unsigned int n) {
    if (packed) hipLaunchKernelGGL(valu_issue_kernel<1>, dim3(cus), dim3(block), 0, stream, n, sink);
    else hipLaunchKernelGGL(valu_issue_kernel<0>, dim3(cus), dim3(block), 0, stream, n, sink);
  };
  launch(100u);
  hipEventRecord(e0, stream);
  launch(iters);
  hipEventRecord(e1, stream);
  float ms = -1.f;
  double rate = -1.0;
  if (hipEventSynchronize(e1) == hipSuccess) {
    hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_cu = (double)iters * 64.0 * (double)(block / 64u);  // 64 instructions per iteration and wavefront
    rate = inst_per_cu / (ms * 1e-3);
    if (mhz) *mhz = khz / 1000.0;
    if (cycles_per_inst) *cycles_per_inst = 4.0 * (khz * 1e3) / rate;  // four SIMDs per CU
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  hipFree(sink);
  return rate;
}

}  // extern "C"
