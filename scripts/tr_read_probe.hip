// tr_read_probe.hip — what ds_read_b64_tr_b16 returns for ARBITRARY per-lane addresses, and whether a
// global_load_lds_dwordx4 image lands lane-linear.  csrc/spmm_tile.hip relies on:
//   result[lane g*16 + i][j] = the 16-bit element (i & 3) of the 8-byte chunk addressed by lane g*16 + 4*j + (i >> 2)
// (cdna_hip_programming.md gives the formula for contiguous lane addresses only).
//   hipcc --offload-arch=gfx950 -O3 scripts/tr_read_probe.hip -o build/tr_read_probe && build/tr_read_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define LDS3(T, p) ((__attribute__((address_space(3))) T*)(p))

__global__ void k(const uint16_t* src, const int* addr, uint16_t* out) {
  __shared__ __attribute__((aligned(1024))) uint16_t lds[8192];
  const int lane = threadIdx.x;
  for (int i = lane; i < 8192; i += 64) lds[i] = src[i];
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS3(s16x4, lds + addr[lane]));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}

int main() {
  std::vector<uint16_t> h(8192);
  for (int i = 0; i < 8192; ++i) h[i] = static_cast<uint16_t>(i);
  std::vector<int> a(64);
  srand(7);
  for (int l = 0; l < 64; ++l) a[l] = (rand() % 2048) * 4;          // 8-byte aligned element offsets
  uint16_t *ds, *dout; int* da;
  CK(hipMalloc(&ds, 8192 * 2)); CK(hipMalloc(&dout, 256 * 2)); CK(hipMalloc(&da, 64 * 4));
  CK(hipMemcpy(ds, h.data(), 8192 * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(da, a.data(), 64 * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, ds, da, dout);
  std::vector<uint16_t> o(256);
  CK(hipMemcpy(o.data(), dout, 256 * 2, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const int g = l >> 4, i = l & 15;
      const int expect = a[g * 16 + 4 * j + (i >> 2)] + (i & 3);
      if (o[l * 4 + j] != expect) {
        if (bad < 16) printf("lane %d elem %d: got %d expected %d\n", l, j, o[l * 4 + j], expect);
        ++bad;
      }
    }
  printf("tr_read model %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
  return bad ? 1 : 0;
}
