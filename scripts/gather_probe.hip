// gather_probe.hip — how fast can MI355X serve RANDOM row gathers out of each level of the memory
// hierarchy?  Decides whether a cache-blocked SpMM can pay on graphs without locality: the SpMM of a
// uniform random graph is bound by nnz * (row bytes) of gather traffic out of HBM;
// if pieces that fit the 256 MiB Infinity Cache are served much faster than HBM, blocking the
// operand (by feature chunk x source range) is worth its extra passes; if not, it is not.
//
//   hipcc --offload-arch=gfx950 -O3 scripts/gather_probe.hip -o build/gather_probe && build/gather_probe
//
// For footprint W (bytes) and piece size G (bytes per gathered row slice): M random piece indices,
// each fetched once by G/8 lanes (8 B per lane, the bf16x4 access of k_spmm_wave) with 8 independent
// gathers in flight per lane, 32 waves per CU.  Prints GB/s of gathered bytes.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int LPP>   // lanes per piece (8 B per lane)
__global__ __launch_bounds__(256) void k_gather(const uint2* __restrict__ buf, const uint32_t* __restrict__ idx,
                                                 int64_t m, uint2* __restrict__ sink) {
  constexpr int PPW = 64 / LPP;             // pieces per wave instruction
  constexpr int U = 8;
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPP, grp = lane / LPP;
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  uint2 acc = make_uint2(0, 0);
  for (int64_t p0 = wave * PPW * U; p0 + PPW * U <= m; p0 += nwaves * PPW * U) {
    uint32_t c[U];
    uint2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) c[u] = idx[p0 + u * PPW + grp];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = buf[static_cast<int64_t>(c[u]) * LPP + sub];
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y += v[u].y; }
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;   // never true: keeps the loads alive
}

template <int LPP>
float run(const uint2* buf, const uint32_t* idx, int64_t m, uint2* sink, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int grid = 256 * 8;
  hipLaunchKernelGGL((k_gather<LPP>), dim3(grid), dim3(256), 0, 0, buf, idx, m, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_gather<LPP>), dim3(grid), dim3(256), 0, 0, buf, idx, m, sink);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  const size_t max_w = size_t(1280) << 20;
  uint2* buf; uint32_t* idx; uint2* sink;
  CK(hipMalloc(&buf, max_w)); CK(hipMemset(buf, 1, max_w));
  const int64_t m_max = int64_t(1) << 27;   // 128 M pieces max
  CK(hipMalloc(&idx, m_max * 4)); CK(hipMalloc(&sink, 64));
  std::vector<uint32_t> h(m_max);
  const size_t foot_mb[] = {16, 64, 128, 160, 192, 224, 256, 320, 512, 1280};
  const int piece[] = {64, 128, 256, 512};
  printf("footprint_MiB piece_B pieces ms GBps\n");
  for (size_t fm : foot_mb) {
    for (int g : piece) {
      const size_t w = fm << 20;
      const uint64_t npieces = w / g;
      int64_t m = (int64_t(8) << 30) / g;   // gather 8 GiB per launch
      if (m > m_max) m = m_max;
      uint64_t s = 0x9E3779B97F4A7C15ull ^ (fm * 1315423911u + g);
      for (int64_t i = 0; i < m; ++i) {      // xorshift64*
        s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
        h[i] = static_cast<uint32_t>(((s * 0x2545F4914F6CDD1Dull) >> 32) % npieces);
      }
      CK(hipMemcpy(idx, h.data(), m * 4, hipMemcpyHostToDevice));
      float ms = 0;
      if (g == 64) ms = run<8>(buf, idx, m, sink, 3);
      else if (g == 128) ms = run<16>(buf, idx, m, sink, 3);
      else if (g == 256) ms = run<32>(buf, idx, m, sink, 3);
      else ms = run<64>(buf, idx, m, sink, 3);
      printf("%zu %d %lld %.3f %.1f\n", fm, g, (long long)m, ms, double(m) * g / (ms * 1e-3) / 1e9);
      fflush(stdout);
    }
  }
  return 0;
}
