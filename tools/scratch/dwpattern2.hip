// Microbenchmark 2 (slab geometry, C = 128: a wave = one column pair x 64 two-channel groups = 256 B per pixel):
//  A: production pattern: per tensor and row four 4-byte loads per lane (columns x0-1 .. x0+2)
//  D: ONE 16-byte load per lane per tensor and row (lane -> column l>>4, 16-B chunk l&15 of the 256-B slab), transposed through
//     a per-wave LDS buffer (ds_write_b128, four ds_read_b32) -- same bytes through L1, a quarter of the vector-memory instructions
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32;
typedef __attribute__((ext_vector_type(4))) u32 u32x4;
constexpr int B = 256, H = 56, W = 56, C = 128, Wp = W / 2;

template <int PF, bool LDSX>
__global__ __launch_bounds__(256) void pat(const u32* __restrict__ z, const u32* __restrict__ y, const u32* __restrict__ x, u32* __restrict__ o, int nitems) {
  __shared__ u32 lbuf[4][2][3][256];                 // [wave][buffer][tensor][4 columns x 64 dwords]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int q = blockIdx.x * 4 + wave; q < nitems; q += gridDim.x * 4) {
    const int xp = q % Wp, rest = q / Wp, b = rest / 2, r0 = (rest % 2) * (H / 2), r1 = r0 + H / 2, x0 = 2 * xp;
    const size_t img = (size_t)b * H * W * (C / 2);
    int col[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) col[c] = min(max(x0 - 1 + c, 0), W - 1) * (C / 2);
    if constexpr (!LDSX) {
      u32 rz[PF][4], ry[PF][4], rx[PF][4];
      auto fetch = [&](int r, int s) {
        const size_t ro = img + (size_t)min(max(r, 0), H - 1) * W * (C / 2) + lane;
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
            u32 a = rz[u][0] ^ ry[u][1] ^ rx[u][2] ^ rz[u][3], bb = ry[u][0] ^ rx[u][1] ^ rz[u][2] ^ ry[u][3] ^ rx[u][0] ^ rz[u][1] ^ ry[u][2] ^ rx[u][3];
            fetch(r + PF, u);
            if (r - 1 >= r0 && r - 1 < r1) {
              const size_t off = img + ((size_t)(r - 1) * W + x0) * (C / 2) + lane;
              o[off] = a; o[off + C / 2] = bb;
            }
          }
        }
      }
    } else {
      u32x4 sz[PF], sy[PF], sx[PF];
      const int c = lane >> 4, ch = lane & 15;
      auto fetch = [&](int r, int s) {
        const size_t ro = (img + (size_t)min(max(r, 0), H - 1) * W * (C / 2) + col[c]) / 4 + ch;
        sz[s] = reinterpret_cast<const u32x4*>(z)[ro]; sy[s] = reinterpret_cast<const u32x4*>(y)[ro]; sx[s] = reinterpret_cast<const u32x4*>(x)[ro];
      };
#pragma unroll
      for (int u = 0; u < PF; ++u) fetch(r0 - 1 + u, u);
      int buf = 0;
      for (int base = r0 - 1; base <= r1; base += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const int r = base + u;
          if (r <= r1) {
            u32 (*lb)[256] = lbuf[wave][buf];
            reinterpret_cast<u32x4*>(lb[0])[lane] = sz[u];
            reinterpret_cast<u32x4*>(lb[1])[lane] = sy[u];
            reinterpret_cast<u32x4*>(lb[2])[lane] = sx[u];
            fetch(r + PF, u);
            u32 rz[4], ry[4], rx[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { rz[k] = lb[0][k * 64 + lane]; ry[k] = lb[1][k * 64 + lane]; rx[k] = lb[2][k * 64 + lane]; }
            u32 a = rz[0] ^ ry[1] ^ rx[2] ^ rz[3], bb = ry[0] ^ rx[1] ^ rz[2] ^ ry[3] ^ rx[0] ^ rz[1] ^ ry[2] ^ rx[3];
            if (r - 1 >= r0 && r - 1 < r1) {
              const size_t off = img + ((size_t)(r - 1) * W + x0) * (C / 2) + lane;
              o[off] = a; o[off + C / 2] = bb;
            }
            buf ^= 1;
          }
        }
      }
    }
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
    printf("%-52s %7.1f us  %5.2f TB/s (4 tensors x %zu MB)\n", name, ms * 1e3, 4.0 * bytes / (ms * 1e-3) / 1e12, bytes >> 20);
  };
  const int nitems = Wp * B * 2;
  for (int g : {512, 1024, 2048}) {
    char nm[96];
    snprintf(nm, 96, "A: 4 x 4-B loads per tensor-row, PF=3, grid %d", g);
    time(nm, [&] { hipLaunchKernelGGL((pat<3, false>), dim3(g), dim3(256), 0, 0, (const u32*)z, (const u32*)y, (const u32*)x, (u32*)o, nitems); });
    snprintf(nm, 96, "D: 1 x 16-B load + LDS transpose, PF=3, grid %d", g);
    time(nm, [&] { hipLaunchKernelGGL((pat<3, true>), dim3(g), dim3(256), 0, 0, (const u32*)z, (const u32*)y, (const u32*)x, (u32*)o, nitems); });
    snprintf(nm, 96, "D: 1 x 16-B load + LDS transpose, PF=6, grid %d", g);
    time(nm, [&] { hipLaunchKernelGGL((pat<6, true>), dim3(g), dim3(256), 0, 0, (const u32*)z, (const u32*)y, (const u32*)x, (u32*)o, nitems); });
  }
  return 0;
}
