// Pointwise (1x1) convolution forward and data-gradient as one MFMA GEMM family (gfx950).
//
//   out[m][n] = epilogue( sum_k  pro(A)[m][k] * W[n][k] ),   m = pixel (B*H*W), NHWC rows
//
// The contraction runs on the matrix cores (bf16: v_mfma_f32_16x16x32_bf16, fp32 parity
// mode: v_mfma_f32_16x16x4_f32 = exact fp32 fma chain), but the kernel is HBM-bound
// (<= 39 FLOP/B), so the design goal is ONE pass over the activations:
//   * the A operand is staged global -> registers -> LDS in full 128-B row segments; the
//     producer's BatchNorm affine + activation (+SE), or -- for the data gradient -- the
//     BatchNorm-backward affine  dy = alpha*dz + beta*y + gamma, is applied in that
//     register stage, so neither ever costs a pass over HBM;
//   * the product is computed transposed (D = W * A^T) with the weight rows permuted so
//     that every lane ends up with 4*NT CONSECUTIVE output channels of one pixel:
//     16-B vector stores, no LDS transpose of the accumulators;
//   * forward epilogue: raw output + per-channel sum / sum-of-squares for the following
//     BatchNorm; dgrad epilogue: multiply by act'(BN(y_in)), emit sum(dz), sum(dz*y_in)
//     for the BatchNorm backward of the producer (or per-sample sums when an SE gate
//     sits in between).  Sums are reduced over the 16 pixels of a DPP row, parked in
//     LDS across the block's persistent tile loop and leave as one fp64 atomic per
//     channel per block.
#include <cstdlib>
#include <type_traits>
#include "pwconv_common.h"

using namespace t3d_pw;

namespace {

template <typename T> struct MM;
template <> struct MM<bf16_t> { static constexpr int BK = 64, EPV = 8, LDK = 72; };
template <> struct MM<float> { static constexpr int BK = 32, EPV = 4, LDK = 36; };

constexpr int BM = 128;  // pixels per tile (4 waves x 2 x 16)

template <typename T, int NT>
__global__ __launch_bounds__(256, 2) void pw_gemm_kernel(const GemmArgs a) {
  constexpr int BK = MM<T>::BK, EPV = MM<T>::EPV, LDK = MM<T>::LDK;
  constexpr int BN = NT * 16;
  constexpr int WV = NT / 2;  // 16-B weight vectors per thread per chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* As = reinterpret_cast<T*>(smem);                         // [BM][LDK]
  T* Ws = As + BM * LDK;                                      // [BN][LDK]
  double* lstat = reinterpret_cast<double*>(Ws + BN * LDK);   // [BN][2] fp64: adds of the tiles' fp32 partial sums in any order
  float* coef = reinterpret_cast<float*>(lstat + BN * 2);     // [3][kpad]
  const int kpad = (a.Kin + BK - 1) / BK * BK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lg = lane >> 4, lc = lane & 15;
  const int n0 = blockIdx.y * BN;
  const T* __restrict__ A0 = reinterpret_cast<const T*>(a.a0);
  const T* __restrict__ A1 = reinterpret_cast<const T*>(a.a1);
  const T* __restrict__ Wg = reinterpret_cast<const T*>(a.w);
  T* __restrict__ out = reinterpret_cast<T*>(a.out);

  for (int i = tid; i < BN * 2; i += 256) lstat[i] = 0.0;
  for (int i = tid; i < kpad; i += 256) {
    const bool v = i < a.Kin;
    if (!a.dgrad) {
      coef[i] = (v && a.p0) ? a.p0[i] : 1.f;
      coef[kpad + i] = (v && a.p0) ? a.p1[i] : 0.f;
    } else {
      coef[i] = (v && !a.per_sample) ? a.p0[i] : 0.f;
      coef[kpad + i] = v ? a.p1[i] : 0.f;
      coef[2 * kpad + i] = (v && !a.per_sample) ? a.p2[i] : 0.f;
    }
  }
  const bool plainA = (!a.dgrad && !a.p0 && !a.p2 && a.act == T3D_ACT_NONE);
  const int arow = tid >> 3, aseg = tid & 7;  // A staging: rows arow + 32*i, 16-B segment aseg
  const int nk = kpad / BK;

  for (int mt = blockIdx.x; mt < a.mtiles; mt += gridDim.x) {
    const int m0 = mt * BM;
    f32x4 acc[2][NT];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 ra[4], rb[4], rw[WV];
    auto gload = [&](int kc) {
      const int k = kc * BK + aseg * EPV;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + arow + 32 * i;
        const bool ok = (m < a.M) && (k < a.Kin);
        ra[i] = ok ? *reinterpret_cast<const uint4*>(A0 + (size_t)m * a.Kin + k) : uint4{0, 0, 0, 0};
        if (a.dgrad) rb[i] = ok ? *reinterpret_cast<const uint4*>(A1 + (size_t)m * a.Kin + k) : uint4{0, 0, 0, 0};
      }
#pragma unroll
      for (int i = 0; i < WV; ++i) {
        const int idx = tid + 256 * i;
        const int n = n0 + (idx >> 3), kk = kc * BK + (idx & 7) * EPV;
        rw[i] = (n < a.Nout && kk < a.Kin) ? *reinterpret_cast<const uint4*>(Wg + (size_t)n * a.Kin + kk)
                                           : uint4{0, 0, 0, 0};
      }
    };
    auto lstore = [&](int kc) {
      const int kl = aseg * EPV, k = kc * BK + kl;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = arow + 32 * i;
        T* dst = As + row * LDK + kl;
        if (plainA) {
          *reinterpret_cast<uint4*>(dst) = ra[i];
          continue;
        }
        const int m = m0 + row;
        float v[EPV], y[EPV];
        ldvec<T>(reinterpret_cast<const T*>(&ra[i]), v);
        const bool ok = (m < a.M) && (k < a.Kin);
        if (!a.dgrad) {
          const float* se = (a.p2 && ok) ? a.p2 + (size_t)(m / a.HW) * a.Kin + k : nullptr;
#pragma unroll
          for (int j = 0; j < EPV; ++j) {
            float u = v[j] * coef[k + j] + coef[kpad + k + j];
            const float sv = se ? se[j] : 1.f;
            if (!a.se_after) u *= sv;
            u = act_apply(u, a.act);
            if (a.se_after) u *= sv;
            v[j] = ok ? u : 0.f;
          }
        } else {
          ldvec<T>(reinterpret_cast<const T*>(&rb[i]), y);
          const size_t pb = (a.per_sample && ok) ? (size_t)(m / a.HW) * a.Kin + k : 0;
#pragma unroll
          for (int j = 0; j < EPV; ++j) {
            const float al = a.per_sample ? (ok ? a.p0[pb + j] : 0.f) : coef[k + j];
            const float ga = a.per_sample ? (ok ? a.p2[pb + j] : 0.f) : coef[2 * kpad + k + j];
            v[j] = ok ? (al * v[j] + coef[kpad + k + j] * y[j] + ga) : 0.f;
          }
        }
        stvec<T>(dst, v);
      }
#pragma unroll
      for (int i = 0; i < WV; ++i) {
        const int idx = tid + 256 * i;
        *reinterpret_cast<uint4*>(Ws + (idx >> 3) * LDK + (idx & 7) * EPV) = rw[i];
      }
    };

    // split contraction (a.kz > 1: few-pixel, deep layers -- the classifier's data gradient is 256 x 1280 -> 960 on 16
    // workgroups, 40 serial chunk rounds): blockIdx.z owns chunks [kc0, kc1) and writes its partial product to the
    // workspace (atomics into the output were no faster than the unsplit launch: device-scope float atomics from 8 XCDs)
    const int kc0 = a.kz > 1 ? (int)((long long)blockIdx.z * nk / a.kz) : 0;
    const int kc1 = a.kz > 1 ? (int)((long long)(blockIdx.z + 1) * nk / a.kz) : nk;
    gload(kc0);
    for (int kc = kc0; kc < kc1; ++kc) {
      __syncthreads();  // previous chunk consumed (and coef/lstat init visible)
      lstore(kc);
      __syncthreads();
      if (kc + 1 < kc1) gload(kc + 1);
      const T* ap = As + (wave * 32 + lc) * LDK;
      // weight row owned by MFMA row lc of tile t:  n_local = (lc>>2)*4*NT + 4*t + (lc&3)
      const T* wp = Ws + ((lc >> 2) * 4 * NT + (lc & 3)) * LDK;
      if constexpr (std::is_same<T, bf16_t>::value) {
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
          const int ko = ks * 32 + lg * 8;
          const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(ap + ko);
          const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(ap + 16 * LDK + ko);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wp + 4 * t * LDK + ko);
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, b0, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, b1, acc[1][t], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int kg = 0; kg < BK / 16; ++kg) {
          const int ko = kg * 16 + lg * 4;
          const float4 b0 = *reinterpret_cast<const float4*>(ap + ko);
          const float4 b1 = *reinterpret_cast<const float4*>(ap + 16 * LDK + ko);
          const float b0v[4] = {b0.x, b0.y, b0.z, b0.w}, b1v[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const float4 wq = *reinterpret_cast<const float4*>(wp + 4 * t * LDK + ko);
            const float wv[4] = {wq.x, wq.y, wq.z, wq.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], b0v[j], acc[0][t], 0, 0, 0);
              acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], b1v[j], acc[1][t], 0, 0, 0);
            }
          }
        }
      }
    }

    // ---------------- epilogue: lane holds channels nb .. nb+4*NT-1 of pixel m ----------
    const int nb = n0 + lg * 4 * NT;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int mrow0 = m0 + wave * 32 + r * 16;
      const int m = mrow0 + lc;
      const bool mok = m < a.M;
      // all 16 pixels of the DPP row in one sample and in range -> reduce in-row, one lane adds
      const int mlast = mrow0 + 15;
      const bool uni = (mlast < a.M) && (mrow0 / a.HW == mlast / a.HW);
      const int bidx = mok ? m / a.HW : 0;
#pragma unroll
      for (int q = 0; q < NT / 2; ++q) {
        const int n = nb + 8 * q;
        if (n >= a.Nout) continue;  // whole 8-channel groups are in or out (Nout % 8 == 0)
        float v[8], yv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = acc[r][2 * q + (j >> 2)][j & 3];
        if constexpr (std::is_same<T, float>::value) {
          if (a.kz > 1) {      // plain product only (launcher): this split's partial tile; splitk_reduce_kernel finishes
            if (mok) Vec8<float>::store(a.part + ((size_t)blockIdx.z * a.M + m) * a.Nout + n, v);
            continue;
          }
        }
        if (a.bias) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += a.bias[n + j];
        }
        if (a.e_y) {
          if (mok) Vec8<T>::load(reinterpret_cast<const T*>(a.e_y) + (size_t)m * a.Nout + n, yv);
          else {
#pragma unroll
            for (int j = 0; j < 8; ++j) yv[j] = 0.f;
          }
          const float* se = (a.e_se && mok) ? a.e_se + (size_t)bidx * a.Nout + n : nullptr;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float u = yv[j] * (a.e_scale ? a.e_scale[n + j] : 1.f) + (a.e_scale ? a.e_shift[n + j] : 0.f);
            const float sv = se ? se[j] : 1.f;
            if (!a.e_se_after) {
              v[j] *= act_grad(u * sv, a.e_act);          // d/d(se*u) of act(se*u)
            } else {
              v[j] *= sv * act_grad(u, a.e_act);          // d/du of se*act(u)  (gate treated as constant here)
            }
          }
        }
        if (a.e_res && mok) {
          float rr[8];
          Vec8<T>::load(reinterpret_cast<const T*>(a.e_res) + (size_t)m * a.Nout + n, rr);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += rr[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = mok ? Vec8<T>::round(v[j]) : 0.f;
        if (mok) Vec8<T>::store(out + (size_t)m * a.Nout + n, v);
        if (a.stats || a.ps_stats) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float s1 = v[j];
            float s2 = a.e_y ? v[j] * yv[j] : v[j] * v[j];
            if (a.ps_stats) {
              if (uni) {
                s1 = row16_sum(s1);
                s2 = row16_sum(s2);
                if (lc == 0) {
                  unsafeAtomicAdd(a.ps_stats + ((size_t)bidx * a.Nout + n + j) * 2, s1);
                  unsafeAtomicAdd(a.ps_stats + ((size_t)bidx * a.Nout + n + j) * 2 + 1, s2);
                }
              } else if (mok) {
                unsafeAtomicAdd(a.ps_stats + ((size_t)bidx * a.Nout + n + j) * 2, s1);
                unsafeAtomicAdd(a.ps_stats + ((size_t)bidx * a.Nout + n + j) * 2 + 1, s2);
              }
            } else {
              s1 = row16_sum(s1);
              s2 = row16_sum(s2);
              if (lc == 0) {
                atomicAdd(lstat + (n - n0 + j) * 2, (double)s1);
                atomicAdd(lstat + (n - n0 + j) * 2 + 1, (double)s2);
              }
            }
          }
        }
      }
    }
  }

  if (a.stats) {
    __syncthreads();
    for (int i = tid; i < BN * 2; i += 256) {
      const int n = n0 + (i >> 1);
      if (n < a.Nout) atomicAdd(a.stats + (size_t)(i & 1) * a.Nout + n, lstat[i]);
    }
  }
}

// out[m][n] = sum_z part[z][m][n] (+ bias[n]);  stats [2][N] fp64 += column sums of out, out^2.  Block = 64 columns x 64 rows.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                            float* __restrict__ out, double* __restrict__ stats, int kz, int M,
                                                            int N) {
  __shared__ float red[2][4][64];
  const int n = blockIdx.x * 64 + (threadIdx.x & 63), rq = threadIdx.x >> 6;
  float s1 = 0.f, s2 = 0.f;
  if (n < N) {
    const float b = bias ? bias[n] : 0.f;
    const int mend = min(M, (int)blockIdx.y * 64 + 64);
    // four rows x up to eight partials in flight per thread (the plain nest was 16 x kz dependent-latency loads: 28 us for 10 MB);
    // every element still adds its partials in split order and the rows in row order: the same numbers
    for (int m0 = blockIdx.y * 64 + rq; m0 < mend; m0 += 16) {
      float p[4][8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = min(m0 + 4 * i, mend - 1);
#pragma unroll
        for (int z = 0; z < 8; ++z) p[i][z] = z < kz ? part[((size_t)z * M + m) * N + n] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + 4 * i;
        if (m < mend) {
          float v = 0.f;
#pragma unroll
          for (int z = 0; z < 8; ++z)
            if (z < kz) v += p[i][z];
          for (int z = 8; z < kz; ++z) v += part[((size_t)z * M + m) * N + n];
          v += b;
          out[(size_t)m * N + n] = v;
          s1 += v;
          s2 += v * v;
        }
      }
    }
  }
  if (!stats) return;
  red[0][rq][threadIdx.x & 63] = s1;
  red[1][rq][threadIdx.x & 63] = s2;
  __syncthreads();
  if (rq == 0 && n < N) {
    const int l = threadIdx.x;
    atomicAdd(stats + n, (double)(red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l]));
    atomicAdd(stats + N + n, (double)(red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l]));
  }
}

template <typename T, int NT>
int launch_nt(GemmArgs& a, hipStream_t st) {
  constexpr int BN = NT * 16;
  const int kpad = (a.Kin + MM<T>::BK - 1) / MM<T>::BK * MM<T>::BK;
  const size_t lds = (size_t)(BM + BN) * MM<T>::LDK * sizeof(T) + BN * 2 * 8 + (size_t)3 * kpad * 4;
  a.mtiles = cdiv(a.M, BM);
  const int ny = cdiv(a.Nout, BN);
  int gx = a.mtiles < 2048 / ny ? a.mtiles : 2048 / ny;
  if (gx < 1) gx = 1;
  if (lds > 64 * 1024)
    (void)t3d_max_lds((const void*)pw_gemm_kernel<T, NT>, (int)lds);
  a.kz = 1;
  const int nk = kpad / MM<T>::BK;
  if (std::is_same<T, float>::value && gx * ny <= 64 && nk >= 16 && !a.e_y && !a.e_res && !a.ps_stats) {
    int kz = nk / 4 < 8 ? nk / 4 : 8;
    const size_t need = (size_t)kz * a.M * a.Nout * sizeof(float);
    if (g_t3d_ws_main.ptr && (size_t)g_t3d_ws_main.bytes >= need) {
      a.kz = kz;
      a.part = reinterpret_cast<float*>(g_t3d_ws_main.ptr);
    }
  }
  T3D_LAUNCH_TIMED((pw_gemm_kernel<T, NT>), dim3(gx, ny, a.kz), dim3(256), lds, st, a);
  if (a.kz > 1)
    T3D_LAUNCH(splitk_reduce_kernel, dim3(cdiv(a.Nout, 64), cdiv(a.M, 64)), dim3(256), 0, st, a.part, a.bias,
                       reinterpret_cast<float*>(a.out), a.stats, a.kz, a.M, a.Nout);
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

template <typename T>
int launch(GemmArgs& a, hipStream_t st) {
  if (a.Nout <= 32) return launch_nt<T, 2>(a, st);
  if (a.Nout <= 64) return launch_nt<T, 4>(a, st);
  return launch_nt<T, 8>(a, st);  // wider outputs tile over grid.y (A re-read comes from L2)
}

int dispatch(int dtype, GemmArgs& a, void* stream) {
  if (a.M <= 0 || a.Kin <= 0 || a.Nout <= 0 || (a.Kin % 8) || (a.Nout % 8) || a.HW <= 0) return T3D_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // `w` is the fragment-order copy (include/t3d.h: T3D_W_FRAG): the deep-contraction kernel where it takes the shape, else the
  // streaming kernel; the LDS-tiled fallback below cannot read it
  a.wfrag = (dtype & T3D_W_FRAG) ? 1 : 0;
  dtype &= ~T3D_W_FRAG;
  if (a.wfrag && dtype != T3D_BF16 && dtype != T3D_F16) return T3D_ERR_ARG;
  if (dtype == T3D_BF16) {
    int rc = (a.wfrag && deep_shape(a.Kin, a.Nout)) ? deep_launch(a, st) : T3D_ERR_UNSUPPORTED;
    if (rc == T3D_ERR_UNSUPPORTED) rc = stream_launch(a, st);
    if (rc != T3D_ERR_UNSUPPORTED || a.wfrag) return rc;
  }
  if (dtype == T3D_F16) {                    // inference forward in fp16 storage: the streaming kernel or nothing
    if (const int rc = t3d_fold_fallback(a.p0, st)) return rc;
    return stream_launch_f16(a, st);
  }
  // the LDS-tiled kernel reads finished coefficients: a pending derive request for them becomes a launch of its own
  if (const int rc = t3d_fold_fallback(a.p0, st)) return rc;
  if (dtype == T3D_F32) {
    // inference forwards of many-pixel layers: the register-operand fp32 kernel (pwconv_f32_reg.hip; T3D_F32_TILED=1: round 1's)
    const bool tiled = getenv("T3D_F32_TILED") != nullptr;
    const int rc = tiled ? T3D_ERR_UNSUPPORTED : f32_reg_launch(a, st);
    return rc != T3D_ERR_UNSUPPORTED ? rc : launch<float>(a, st);
  }
  if (dtype == T3D_BF16) return launch<bf16_t>(a, st);
  return T3D_ERR_ARG;
}

}  // namespace

extern "C" int t3d_pwconv_fwd(int dtype, const void* x, const t3d_prologue* pro, const void* w, const float* bias,
                              void* y, double* stats, int M, int HW, int K, int N, void* stream) {
  if (!x || !w || (!y && !stats)) return T3D_ERR_ARG;
  GemmArgs a{};
  a.a0 = x;
  if (pro) { a.p0 = pro->scale; a.p1 = pro->shift; a.p2 = pro->se; a.act = pro->act; a.se_after = pro->se_after_act; }
  a.w = w; a.bias = bias; a.out = y; a.stats = stats;
  a.M = M; a.HW = HW; a.Kin = K; a.Nout = N;
  if (!y) {
    // statistics-only pass (the BatchNorm sums of a conv whose output is never stored: t3d_expdw_fwd recomputes it in
    // LDS): bf16 streaming kernel only
    a.wfrag = (dtype & T3D_W_FRAG) ? 1 : 0;
    if ((dtype & ~T3D_W_FRAG) != T3D_BF16 || (K % 8) || (N % 8) || M <= 0 || HW <= 0) return T3D_ERR_UNSUPPORTED;
    return stream_launch(a, reinterpret_cast<hipStream_t>(stream));
  }
  return dispatch(dtype, a, stream);
}

// include/t3d.h
extern "C" int t3d_bn_apply(int dtype, const void* y, const t3d_prologue* pro, const void* residual, void* z, int M, int C,
                            void* stream);
extern "C" int t3d_pwconv_fwd_mat(int dtype, const void* y_in, const t3d_prologue* pro_in, const void* residual, void* z_out,
                                  const void* w, void* y, double* stats, int M, int HW, int K, int N, void* stream) {
  if (!y_in || !pro_in || !z_out || !w || !y) return T3D_ERR_ARG;
  if (pro_in->se || M <= 0 || K <= 0 || N <= 0 || (K % 8) || (N % 8) || HW <= 0) return T3D_ERR_ARG;
  const bool wfrag = (dtype & T3D_W_FRAG) != 0;
  dtype &= ~T3D_W_FRAG;
  if (dtype == T3D_BF16) {
    GemmArgs a{};
    a.wfrag = wfrag ? 1 : 0;
    a.a0 = y_in;
    a.p0 = pro_in->scale; a.p1 = pro_in->shift; a.act = pro_in->act;
    a.z_res = residual; a.z_out = z_out;
    a.w = w; a.out = y; a.stats = stats;
    a.M = M; a.HW = HW; a.Kin = K; a.Nout = N;
    int rc = wfrag ? wide_launch(a, reinterpret_cast<hipStream_t>(stream)) : T3D_ERR_UNSUPPORTED;
    if (rc == T3D_ERR_UNSUPPORTED) rc = stream_launch(a, reinterpret_cast<hipStream_t>(stream));
    if (rc != T3D_ERR_UNSUPPORTED || wfrag) return rc;
  }
  if (wfrag) return T3D_ERR_ARG;
  if (dtype == T3D_F32 && !getenv("T3D_F32_TILED")) {
    // fp32 storage: the register-operand kernel materialises the block output itself (pwconv_f32_reg.hip)
    if (const int rc = t3d_fold_fallback(pro_in->scale, reinterpret_cast<hipStream_t>(stream))) return rc;
    GemmArgs a{};
    a.a0 = y_in;
    a.p0 = pro_in->scale; a.p1 = pro_in->shift; a.act = pro_in->act;
    a.z_res = residual; a.z_out = z_out;
    a.w = w; a.out = y; a.stats = stats;
    a.M = M; a.HW = HW; a.Kin = K; a.Nout = N;
    const int rc = f32_reg_launch(a, reinterpret_cast<hipStream_t>(stream));
    if (rc != T3D_ERR_UNSUPPORTED) return rc;
  }
  // no materialising kernel for this case (fp32 training forward, shapes outside the streaming kernel): the two launches it fuses
  if (const int rc = t3d_bn_apply(dtype, y_in, pro_in, residual, z_out, M, K, stream)) return rc;
  return t3d_pwconv_fwd(dtype, z_out, nullptr, w, nullptr, y, stats, M, HW, K, N, stream);
}

extern "C" int t3d_pwconv_dgrad(int dtype, const void* dz, const void* y, const t3d_bnbwd* bb, const void* wt,
                                const void* x_raw, const t3d_prologue* pro_in, const void* residual, void* dx,
                                double* stats, float* ps_stats, int M, int HW, int K, int N, void* stream) {
  if (!dz || !y || !bb || !wt || !dx || !bb->beta) return T3D_ERR_ARG;
  GemmArgs a{};
  a.dgrad = 1;
  a.a0 = dz; a.a1 = y;
  a.p0 = bb->alpha; a.p1 = bb->beta; a.p2 = bb->gamma; a.per_sample = bb->per_sample;
  a.w = wt;
  if (x_raw) {
    a.e_y = x_raw;
    if (pro_in) {
      a.e_scale = pro_in->scale; a.e_shift = pro_in->shift; a.e_se = pro_in->se;
      a.e_act = pro_in->act; a.e_se_after = pro_in->se_after_act;
    }
  }
  a.e_res = residual; a.out = dx; a.stats = stats; a.ps_stats = ps_stats;
  a.M = M; a.HW = HW; a.Kin = N; a.Nout = K;  // contraction runs over the forward OUTPUT channels
  return dispatch(dtype, a, stream);
}

