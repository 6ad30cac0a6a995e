// l2_gather_probe.hip — ceiling for 512-byte row gathers that HIT in the cache levels above the fabric
// (CU vector L1 16-32 KiB, XCD L2 4 MiB).  Companion of gather_probe.hip (which covers the Infinity Cache /
// HBM side).  Answers: once sgf_reorder has made the SpMM's gathers L2 hits (hit rate 70-80 %,
// profiles/r02_spmm_pmc.md), how far is the kernel from what the L2 -> L1 path can deliver?
//
//   hipcc --offload-arch=gfx950 -O3 scripts/l2_gather_probe.hip -o build/l2_gather_probe && build/l2_gather_probe
//
// Footprint W, M random 512-byte rows, fetched as 64 lanes x 8 B (one row per instruction) or 32 lanes x 16 B
// (two rows per instruction, the k_spmm_seg_bf16x2 access), U independent gathers in flight per lane.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename V, int U>
__global__ __launch_bounds__(256) void k_gather(const V* __restrict__ buf, const uint32_t* __restrict__ idx,
                                                 int64_t m, V* __restrict__ sink) {
  constexpr int LPP = 512 / sizeof(V);      // lanes per row
  constexpr int PPW = 64 / LPP;             // rows per wave instruction
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPP, grp = lane / LPP;
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  uint32_t acc = 0;
  for (int64_t p0 = wave * PPW * U; p0 + PPW * U <= m; p0 += nwaves * PPW * U) {
    uint32_t c[U];
    V v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) c[u] = idx[p0 + u * PPW + grp];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = buf[static_cast<int64_t>(c[u]) * LPP + sub];
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u].x + v[u].y;
  }
  if (acc == 0x12345678u) sink[0].x = acc;   // never true: keeps the loads alive
}

template <typename V, int U>
float run(const void* buf, const uint32_t* idx, int64_t m, void* sink, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int grid = 256 * 8;
  auto go = [&] {
    hipLaunchKernelGGL((k_gather<V, U>), dim3(grid), dim3(256), 0, 0, static_cast<const V*>(buf), idx, m,
                       static_cast<V*>(sink));
  };
  go();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) go();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  const size_t max_w = size_t(64) << 20;
  void* buf; uint32_t* idx; void* sink;
  CK(hipMalloc(&buf, max_w)); CK(hipMemset(buf, 1, max_w));
  const int64_t m = int64_t(1) << 24;       // 16 M rows = 8 GiB gathered per launch
  CK(hipMalloc(&idx, m * 4)); CK(hipMalloc(&sink, 64));
  std::vector<uint32_t> h(m);
  const size_t foot_kb[] = {8, 256, 1024, 2048, 4096, 8192, 16384, 65536};
  printf("footprint_KiB lane_B inflight ms GBps\n");
  for (size_t fk : foot_kb) {
    const uint64_t nrows = (fk << 10) / 512;
    uint64_t s = 0x9E3779B97F4A7C15ull ^ (fk * 1315423911u);
    for (int64_t i = 0; i < m; ++i) {        // xorshift64*
      s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
      h[i] = static_cast<uint32_t>(((s * 0x2545F4914F6CDD1Dull) >> 32) % nrows);
    }
    CK(hipMemcpy(idx, h.data(), m * 4, hipMemcpyHostToDevice));
    struct { const char* name; float ms; } r[] = {
        {"8 4", run<uint2, 4>(buf, idx, m, sink, 5)},   {"8 8", run<uint2, 8>(buf, idx, m, sink, 5)},
        {"16 4", run<uint4, 4>(buf, idx, m, sink, 5)},  {"16 8", run<uint4, 8>(buf, idx, m, sink, 5)},
    };
    for (auto& e : r) {
      printf("%zu %s %.3f %.1f\n", fk, e.name, e.ms, double(m) * 512 / (e.ms * 1e-3) / 1e9);
    }
    fflush(stdout);
  }
  return 0;
}
