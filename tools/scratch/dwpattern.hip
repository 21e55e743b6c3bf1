// Microbenchmark: what do the MEMORY ACCESS PATTERNS of the streaming depthwise backward cost, with the math removed?
// Tensors [B,H,W,C] bf16 (B=256,H=W=112,C=32): read dz, y, x, write dx.
//  A: the production pattern -- lane = (column pair, 2 channels): four 4-byte column loads per tensor per row, PF-row register ring
//  B: fully coalesced 16 B per lane, each element loaded once (what an LDS-staged kernel would issue), U loads in flight
//  C: as A but 8 B per lane (4 channels), two columns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32;
typedef __attribute__((ext_vector_type(4))) u32 u32x4;
typedef __attribute__((ext_vector_type(2))) u32 u32x2;
constexpr int B = 256, H = 112, W = 112, C = 32;

template <int PF, typename V>      // V = u32 (2 ch) or u32x2 (4 ch)
__global__ __launch_bounds__(256) void patA(const V* __restrict__ z, const V* __restrict__ y, const V* __restrict__ x, V* __restrict__ o,
                                            int nitems) {
  constexpr int CH = sizeof(V) / 2, CG = C / CH, Wp = W / 2;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= Wp * CG) return;
  const int cg = j % CG, xp = j / CG, x0 = 2 * xp;
  int col[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) col[c] = (min(max(x0 - 1 + c, 0), W - 1) * C + cg * CH) / CH;
  for (int q = blockIdx.y; q < nitems; q += gridDim.y) {          // item = (image, half of the rows)
    const int b = q / 2, r0 = (q % 2) * (H / 2), r1 = r0 + H / 2;
    const size_t img = (size_t)b * H * W * C / CH;
    V rz[PF][4], ry[PF][4], rx[PF][4];
    auto fetch = [&](int r, int s) {
      const size_t ro = img + (size_t)min(max(r, 0), H - 1) * W * C / CH;
#pragma unroll
      for (int c = 0; c < 4; ++c) { rz[s][c] = z[ro + col[c]]; ry[s][c] = y[ro + col[c]]; rx[s][c] = x[ro + col[c]]; }
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) fetch(r0 - 1 + u, u);
    for (int base = r0 - 1; base <= r1; base += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int r = base + u;
        if (r <= r1) {
          V a = rz[u][0] ^ ry[u][1] ^ rx[u][2] ^ rz[u][3], bb = ry[u][0] ^ rx[u][1] ^ rz[u][2] ^ ry[u][3] ^ rx[u][0] ^ rz[u][1] ^ ry[u][2] ^ rx[u][3];
          fetch(r + PF, u);
          if (r - 1 >= r0 && r - 1 < r1) {
            const size_t off = img + ((size_t)(r - 1) * W + x0) * C / CH + cg;
            o[off] = a;
            o[off + C / CH] = bb;
          }
        }
      }
    }
  }
}

template <int U>
__global__ __launch_bounds__(256) void patB(const u32x4* __restrict__ z, const u32x4* __restrict__ y, const u32x4* __restrict__ x,
                                            u32x4* __restrict__ o, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += stride * U) {
    u32x4 a[U], b[U], c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t k = min(i + u * stride, n16 - 1);
      a[u] = z[k]; b[u] = y[k]; c[u] = x[k];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * stride < n16) o[i + u * stride] = a[u] ^ b[u] ^ c[u];
  }
}

int main() {
  const size_t n = (size_t)B * H * W * C, bytes = n * 2;
  void *z, *y, *x, *o;
  hipMalloc(&z, bytes); hipMalloc(&y, bytes); hipMalloc(&x, bytes); hipMalloc(&o, bytes);
  hipMemset(z, 1, bytes); hipMemset(y, 2, bytes); hipMemset(x, 3, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, auto&& launch) {
    launch();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-44s %7.1f us  %5.2f TB/s (4 tensors x %zu MB)\n", name, ms * 1e3, 4.0 * bytes / (ms * 1e-3) / 1e12, bytes >> 20);
  };
  for (int gy : {64, 128, 256}) {
    char nm[96];
    snprintf(nm, 96, "A: 4 B/lane, 4 cols, PF=3, grid 4x%d", gy);
    time(nm, [&] { hipLaunchKernelGGL((patA<3, u32>), dim3(4, gy), dim3(256), 0, 0, (const u32*)z, (const u32*)y, (const u32*)x, (u32*)o, B * 2); });
    snprintf(nm, 96, "A: 4 B/lane, 4 cols, PF=6, grid 4x%d", gy);
    time(nm, [&] { hipLaunchKernelGGL((patA<6, u32>), dim3(4, gy), dim3(256), 0, 0, (const u32*)z, (const u32*)y, (const u32*)x, (u32*)o, B * 2); });
    snprintf(nm, 96, "C: 8 B/lane, 4 cols, PF=3, grid 2x%d", gy * 2);
    time(nm, [&] { hipLaunchKernelGGL((patA<3, u32x2>), dim3(2, gy * 2), dim3(256), 0, 0, (const u32x2*)z, (const u32x2*)y, (const u32x2*)x, (u32x2*)o, B * 2); });
  }
  for (int g : {512, 1024, 2048}) {
    char nm[96];
    snprintf(nm, 96, "B: 16 B/lane coalesced, U=2, grid %d", g);
    time(nm, [&] { hipLaunchKernelGGL(patB<2>, dim3(g), dim3(256), 0, 0, (const u32x4*)z, (const u32x4*)y, (const u32x4*)x, (u32x4*)o, bytes / 16); });
    snprintf(nm, 96, "B: 16 B/lane coalesced, U=4, grid %d", g);
    time(nm, [&] { hipLaunchKernelGGL(patB<4>, dim3(g), dim3(256), 0, 0, (const u32x4*)z, (const u32x4*)y, (const u32x4*)x, (u32x4*)o, bytes / 16); });
  }
  return 0;
}
