// stream_pattern_probe.hip — does the ACCESS PATTERN of the streaming matrix-core kernels (csrc/rowgemm.hip) cost HBM rate?
// Those kernels load a 32-row tile of a [N, 256] bf16 tensor straight into MFMA A fragments: per load instruction lane
// (i31, hi) takes 16 bytes of row i31 — 32 rows x 32 bytes per instruction, 16 instructions per tile — and store results as
// whole rows.  This probe copies [N, 256] bf16 (1.25 GB in, 1.25 GB out) with that pattern and with fully coalesced 1 KiB
// instructions, same launch shape (512 threads, 2 blocks per CU), with and without a one-tile prefetch.
//
//   hipcc --offload-arch=gfx950 -O3 scripts/stream_pattern_probe.hip -o build/stream_pattern_probe && build/stream_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// PATTERN 0: A-fragment loads (row per lane), 1: coalesced.  STORE 0: like the load pattern, 1: coalesced rows.
template <int PATTERN, int STORE, bool PREFETCH>
__global__ __launch_bounds__(512, 2) void k_copy(const uint4* __restrict__ x, uint4* __restrict__ y, int64_t ntiles) {
  const int lane = threadIdx.x & 63;
  const int i31 = lane & 31, hi = lane >> 5;
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  auto off = [&](int pat, int64_t t, int s) -> int64_t {       // in 16-byte units; a tile = 32 rows x 32 units
    return pat == 0 ? t * 1024 + i31 * 32 + 2 * s + hi : t * 1024 + s * 64 + lane;
  };
  uint4 cur[16], nxt[16];
  int64_t t = wave;
  if (PREFETCH && t < ntiles) {
#pragma unroll
    for (int s = 0; s < 16; ++s) cur[s] = x[off(PATTERN, t, s)];
  }
  for (; t < ntiles; t += nwaves) {
    if (PREFETCH) {
      if (t + nwaves < ntiles) {
#pragma unroll
        for (int s = 0; s < 16; ++s) nxt[s] = x[off(PATTERN, t + nwaves, s)];
      }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) cur[s] = x[off(PATTERN, t, s)];
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      uint4 v = cur[s];
      v.x ^= 0x00010001u;                                   // (something to do)
      y[off(STORE, t, s)] = v;
    }
    if (PREFETCH) {
#pragma unroll
      for (int s = 0; s < 16; ++s) cur[s] = nxt[s];
    }
  }
}

static int g_blocks = 512;
template <int P, int S, bool PF>
void run(const char* what, const uint4* x, uint4* y, int64_t ntiles) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f;
  for (int rep = 0; rep < 40; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_copy<P, S, PF>), dim3(g_blocks), dim3(512), 0, 0, x, y, ntiles);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep >= 20 && ms < best) best = ms;
  }
  printf("%-64s %.3f ms  %.2f TB/s\n", what, best, 2.0 * ntiles * 16384 / (best * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  if (argc > 1) g_blocks = atoi(argv[1]);
  const int64_t n = 2449029, ntiles = n / 32;
  uint4 *x, *y;
  CK(hipMalloc(&x, ntiles * 16384)); CK(hipMalloc(&y, ntiles * 16384));
  CK(hipMemset(x, 1, ntiles * 16384)); CK(hipMemset(y, 0, ntiles * 16384));
  for (int round = 0; round < 2; ++round) {
  run<0, 0, true>("A-fragment loads, A-fragment stores, prefetch", x, y, ntiles);
  run<0, 1, true>("A-fragment loads, whole-row stores, prefetch  (rowgemm.hip)", x, y, ntiles);
  run<1, 1, true>("coalesced loads, whole-row stores, prefetch", x, y, ntiles);
  run<0, 1, false>("A-fragment loads, whole-row stores, no prefetch", x, y, ntiles);
  run<1, 1, false>("coalesced loads, whole-row stores, no prefetch", x, y, ntiles);
  }
  CK(hipMemcpyAsync(y, x, ntiles * 16384, hipMemcpyDeviceToDevice, 0));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a)); CK(hipMemcpyAsync(y, x, ntiles * 16384, hipMemcpyDeviceToDevice, 0)); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("%-64s %.3f ms  %.2f TB/s\n", "hipMemcpyAsync device to device", ms, 2.0 * ntiles * 16384 / (ms * 1e-3) / 1e12);
  return 0;
}
