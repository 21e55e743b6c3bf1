// Shared device helpers for the gfx950 kernels (wave64, NHWC activations).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/t3d.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef _Float16 f16_t;       // fp16 activation storage: inference forward only (3 more mantissa bits than bf16 at every MFMA operand)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
// packed fp32 FMA: one v_pk_fma_f32 does two lanes' worth of fused multiply-adds per issue slot (the fp32 VALU peak
// on gfx950 is only reachable in this form); same rounding as two fmaf() calls.
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

#define T3D_CHECK_LAUNCH()                                  \
  do {                                                      \
    hipError_t e_ = hipGetLastError();                      \
    if (e_ != hipSuccess) return T3D_ERR_LAUNCH;            \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// hipFuncAttributeMaxDynamicSharedMemorySize, raised at most once per (kernel, size): the launch paths used to make this
// runtime call in front of every launch with more than 64 KB of LDS (~70 per training step; misc.hip)
hipError_t t3d_max_lds(const void* fn, int bytes);

// Reduction replicas (t3d_set_reduction_replicas, misc.hip): contended atomics into one small array run an order of
// magnitude below the chip's atomic rate, so block b adds its BatchNorm sums / depthwise weight gradient into
// replica b % nrep; the finalize kernels (and the caller, for dw) sum the replicas.
struct T3dReduceCfg {
  int nrep; long long stats_stride;
  int dw_slots; int* dw_used;      // t3d_set_dw_slots: one depthwise weight-gradient slot per workgroup (no atomics), see t3d_dw_flush
  int pool_exact;                  // t3d_set_exact_pool: squeeze-excite pooled sums as int64 fixed point (t3d_pool_add / t3d_pool_get)
};
extern T3dReduceCfg g_t3d_reduce;

// Environment switches on launch paths are read ONCE per process (getenv walks the whole environment block): a
// lambda per use site holds its own static.  The two switches the test suite flips inside one process (T3D_F32_TILED,
// T3D_DW_TILED: A/B against the round-1 kernels) and the fp32 sweep knob T3D_F32_SHAPE stay plain getenv calls.
#define T3D_ENV_SET(name) ([]() -> bool { static const bool v = getenv(name) != nullptr; return v; }())

// Kernel-exact timing of the NEXT depthwise launch (t3d_set_launch_events, misc.hip): the two events are attached to the
// kernel's own dispatch (hipExtLaunchKernelGGL), so their difference is the kernel's begin-to-end time -- what rocprofv3
// reports -- and not that plus the ~5-9 us of event-record packets and dispatch latency an event pair AROUND the launch
// call measures (bench.py's roofline block; the depthwise launches are 50-200 us long).
struct T3dLaunchEvents { hipEvent_t start, stop; };
extern T3dLaunchEvents g_t3d_time;

// Device-side hand-off to another stream (t3d_plan_run, plan.hip): while `ev` is set, every kernel launched on `stream`
// carries `ev` as the STOP event of its own dispatch packet (hipExtLaunchKernelGGL), so that a hipStreamWaitEvent(other, ev)
// issued afterwards waits for the completion signal of the last such kernel -- the producing queue gets NO event-record
// packet (an event recorded behind a kernel is a barrier packet of its own, and the kernel behind it started 6-30 us late:
// ~35 of them per training step were most of the main queue's 0.3 ms of gaps).  Also counts launches, so that a plan being
// recorded knows which calls launched a kernel and on which stream.
// RULE (tests/test_abi.py::test_every_kernel_launch_goes_through_the_launch_macro holds the sources to it): the
// hand-off waits for "the last kernel of the last call that launched on that stream", so (a) every kernel of the library is
// launched through T3D_LAUNCH / T3D_LAUNCH_TIMED -- never a bare hipLaunchKernelGGL or <<< >>> -- and (b) an entry point
// that enqueues non-kernel stream work (hipMemsetAsync) does so BEFORE its last kernel launch: the stream is in order, so
// that kernel's completion covers it.  A trailing memset / copy would not be waited for by a replayed fork.
struct T3dSignal { hipEvent_t ev; hipStream_t stream; };
extern T3dSignal g_t3d_signal;
extern unsigned long long g_t3d_launches;
extern hipStream_t g_t3d_last_stream;
#define T3D_LAUNCH(kernel, grid, block, lds, st, ...)                                                                    \
  do {                                                                                                                   \
    ++g_t3d_launches;                                                                                                    \
    g_t3d_last_stream = (st);                                                                                            \
    if (g_t3d_signal.ev && g_t3d_signal.stream == (st)) {                                                                \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, st, nullptr, g_t3d_signal.ev, 0, __VA_ARGS__);                     \
    } else {                                                                                                             \
      hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                                     \
    }                                                                                                                    \
  } while (0)
#define T3D_LAUNCH_TIMED(kernel, grid, block, lds, st, ...)                                                              \
  do {                                                                                                                   \
    if (g_t3d_time.start) {                                                                                              \
      ++g_t3d_launches;                                                                                                  \
      g_t3d_last_stream = (st);                                                                                          \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, st, g_t3d_time.start, g_t3d_time.stop, 0, __VA_ARGS__);            \
      g_t3d_time.start = g_t3d_time.stop = nullptr;                                                                      \
      /* (a timed launch that also has to signal: a plain event record behind it) */                                     \
      if (g_t3d_signal.ev && g_t3d_signal.stream == (st)) (void)hipEventRecord(g_t3d_signal.ev, st);                     \
    } else {                                                                                                             \
      T3D_LAUNCH(kernel, grid, block, lds, st, __VA_ARGS__);                                                             \
    }                                                                                                                    \
  } while (0)

// Scratch workspace in device memory (t3d_set_workspace, misc.hip): partial results of split reductions (the pointwise
// weight gradient) are written there with plain stores and summed by a second small kernel instead of leaving as atomics.
struct T3dWorkspace { void* ptr; long long bytes; };
extern T3dWorkspace g_t3d_ws;
extern T3dWorkspace g_t3d_ws_main;   // scratch of the launches on the caller's MAIN stream (t3d_set_main_workspace)

// BatchNorm finalize DERIVED BY THE CONSUMER (t3d_fold_request, misc.hip; descriptor: t3d_bn_fold in include/t3d.h, in
// DEVICE memory -- the kernels take one pointer, not the ~130 bytes: the streaming kernels sit at the SGPR limit and a
// by-value descriptor spilled scalars into their row loops, -40 %).  t3d_bn_finalize / t3d_bn_bwd_finalize are ~5-us
// launches between every convolution and its consumer, ~100 per training step on the critical stream: with the launch
// gaps 0.64 ms of an 8.4-ms step (skip-the-launch ablation, tools/ablate.sh).  Folding them into the PRODUCER's last
// workgroup needs a device-wide "all sums have arrived" and was slower than the launches (round 2: the release fence
// writes back the XCD's dirty L2 lines; without it every wave waits for its output stores before it may exit).  The
// CONSUMER needs no such thing -- the kernel boundary already orders the sums before it -- so the first kernel that
// reads a BatchNorm's coefficients derives them itself: every workgroup computes the channels it is going to use from
// the replica sums (same arithmetic as bn.hip, all loads L2 hits), and one designated workgroup per channel also
// PUBLISHES what the standalone finalize would have written (scale / shift / mean / invstd + running statistics, or
// alpha / beta / gamma + dgamma / dbeta) for every later reader (the backward, the weight-gradient stream).
typedef t3d_bn_fold T3dFold;
struct T3dFoldReq { const T3dFold* desc; const void* key; };
extern T3dFoldReq g_t3d_fold;
// host side: the pending request if it names the coefficient array this launch reads (consumes it), else null
inline const T3dFold* t3d_take_fold(const void* key) {
  if (!key || !g_t3d_fold.desc || g_t3d_fold.key != key) return nullptr;
  const T3dFold* d = g_t3d_fold.desc;
  g_t3d_fold.desc = nullptr;
  return d;
}

// host side, for the launch paths WITHOUT a derive prologue (fp32 parity kernels, tiled fallbacks): if a request for `key`
// is pending, run the finalize as a launch of its own on `st` (bn.hip) and consume the request
int t3d_fold_fallback(const void* key, hipStream_t st);

// Whole workgroup: channels [cbase, cbase + Cb) -> dst[0 .. Cb) = o0, dst[ld .. ld + Cb) = o1, dst[2 ld ..) = o2 (kind 2);
// ends with a barrier.  One thread per channel (consecutive threads = consecutive channels: coalesced 8-B loads), U
// channels per thread at once when the replica count leaves registers for it; every replica load is issued before the
// first add (the sums were written by device-scope atomics and live at the memory side, ~2 us away: a loop with a
// data-dependent trip count would make one round trip per iteration).  `between` runs once per thread between the issue
// of the first batch of loads and their use -- work that does not need the coefficients (weight staging) hides the round
// trip.  `publish`: this workgroup is the one that writes the channels' finalize outputs.
struct T3dNoop { __device__ __forceinline__ void operator()() const {} };
template <class Between = T3dNoop>
__device__ __forceinline__ void t3d_fold_block(const T3dFold* __restrict__ fp, int cbase, int Cb, float* dst, int ld,
                                               bool publish, Between between = Between()) {
  const T3dFold f = *fp;
  const int nthr = blockDim.x;
  auto finish = [&](int i, double s1, double s2) {
    const int c = cbase + i;
    if (f.kind == 1) {
      const double mean = s1 / f.count;
      double var = s2 / f.count - mean * mean;  // biased
      if (var < 0.0) var = 0.0;
      const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
      const float g = f.gamma ? f.gamma[c] : 1.f, b = f.beta ? f.beta[c] : 0.f;
      const float sc = g * invstd, sh = b - (float)mean * sc;
      dst[i] = sc;
      dst[ld + i] = sh;
      if (publish) {
        f.o0[c] = sc;
        f.o1[c] = sh;
        if (f.o2) f.o2[c] = (float)mean;
        if (f.o3) f.o3[c] = invstd;
        if (f.rm) f.rm[c] = (1.f - f.momentum) * f.rm[c] + f.momentum * (float)mean;
        if (f.rv) {
          const double unbiased = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
          f.rv[c] = (1.f - f.momentum) * f.rv[c] + f.momentum * (float)unbiased;
        }
        if (c == 0 && f.nbt) *f.nbt += 1;
      }
    } else {
      const double mu = (double)f.mean[c], is = (double)f.invstd[c];
      const double dg = is * (s2 - mu * s1);
      const double al = (double)(f.gamma ? f.gamma[c] : 1.f) * is;
      const double be = -al * is * dg / f.count;
      const float ga = (float)(-al * s1 / f.count - be * mu);
      dst[i] = (float)al;
      dst[ld + i] = (float)be;
      dst[2 * ld + i] = ga;
      if (publish) {
        f.o0[c] = (float)al;
        f.o1[c] = (float)be;
        f.o2[c] = ga;
        if (f.o3) f.o3[c] = (float)dg;
        if (f.o4) f.o4[c] = (float)s1;
      }
    }
  };
  auto pass = [&](auto nr_tag, auto u_tag) {
    constexpr int NR = decltype(nr_tag)::value, U = decltype(u_tag)::value;
    bool first = true;
    for (int i0 = threadIdx.x; i0 < Cb || first; i0 += U * nthr) {      // (every thread runs the first round: `between`)
      double v1[U][NR], v2[U][NR];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = min(i0 + u * nthr, Cb - 1);          // clamped: the surplus lanes' results are never used
        const double* p = f.stats + cbase + i;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const size_t o = (size_t)min(r, f.nrep - 1) * f.rstride;
          v1[u][r] = p[o];
          v2[u][r] = p[o + f.C];
        }
      }
      if (first) {
        between();
        first = false;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          s1 += r < f.nrep ? v1[u][r] : 0.0;
          s2 += r < f.nrep ? v2[u][r] : 0.0;
        }
        if (i0 + u * nthr < Cb) finish(i0 + u * nthr, s1, s2);
      }
    }
  };
  if (f.nrep <= 2) pass(std::integral_constant<int, 2>{}, std::integral_constant<int, 8>{});
  else if (f.nrep <= 4) pass(std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
  else if (f.nrep <= 8) pass(std::integral_constant<int, 8>{}, std::integral_constant<int, 2>{});
  else pass(std::integral_constant<int, 16>{}, std::integral_constant<int, 1>{});
  __syncthreads();
}

// ---- order-independent BatchNorm sums (forward) ------------------------------------------------------------------
// The batch sums reach memory as a few hundred atomic adds per channel, in whatever order the workgroups finish; fp64
// addition is not associative, the network at initialisation amplifies a last-bit difference of a sum into a flipped
// bf16 rounding a few layers later, and two runs of the same step end up 2.5e-3 apart in the loss.  Every partial sum
// is therefore snapped onto a fixed grid before it is added: with |partial| <= count * 2^10 (raw conv outputs beyond 1024
// would just lose the exactness, nothing else) a quantum of P * 2^-40, P = the power of two >= count, keeps every partial
// and the total below 2^52 quanta -- all adds are then EXACT in fp64, hence independent of their order.  Sums of squares:
// quantum * 2^10.  The grid costs ~1e-9 relative on the sums, ~1e-6 on the sums of squares.
struct T3dQuant { double q, iq; };        // q == 0: off (backward sums: their scale goes with the loss, not with the count)
static inline T3dQuant t3d_quant_for(long long count) {
  int e = 0;
  while ((1LL << e) < count) ++e;
  const double q = ldexp(1.0, e - 40);
  return T3dQuant{q, 1.0 / q};
}
__device__ __forceinline__ double t3d_snap(float v, const T3dQuant& t, bool squares) {
  if (t.q == 0.0) return (double)v;
  const double s = squares ? 1024.0 : 1.0;
  return rint((double)v * (t.iq * (1.0 / s))) * (t.q * s);
}

// ---- 8-channel vector load/store, storage type T, math in fp32 ------------
template <typename T> struct Vec8;
template <> struct Vec8<float> {
  static __device__ __forceinline__ void load(const float* p, float v[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ __forceinline__ void store(float* p, const float v[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
  static __device__ __forceinline__ float round(float x) { return x; }
};
template <> struct Vec8<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float v[8]) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float v[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = a;
  }
  static __device__ __forceinline__ float round(float x) { return (float)(bf16_t)x; }
};

template <> struct Vec8<f16_t> {
  static __device__ __forceinline__ void load(const f16_t* p, float v[8]) {
    const f16x8 a = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
  }
  static __device__ __forceinline__ void store(f16_t* p, const float v[8]) {
    f16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (f16_t)v[i];
    *reinterpret_cast<f16x8*>(p) = a;
  }
  static __device__ __forceinline__ float round(float x) { return (float)(f16_t)x; }
};

// ---- activations (reference: torchdet3d/models/mobilenetv3.py:74-89) ------
// h_swish(x) = x * relu6(x + 3) / 6.  The 1/6 is applied as a multiplication (an IEEE division costs ~10 VALU
// instructions per element; the results agree to 1 ulp, far inside the 1e-4 parity bound).
#define T3D_SIXTH 0.16666667163372040f
__device__ __forceinline__ float act_apply(float x, int act) {
  switch (act) {
    case T3D_ACT_RELU: return fmaxf(x, 0.f);
    case T3D_ACT_RELU6: return __builtin_amdgcn_fmed3f(x, 0.f, 6.f);           // one v_med3_f32
    case T3D_ACT_HSWISH: return x * (__builtin_amdgcn_fmed3f(x + 3.f, 0.f, 6.f) * T3D_SIXTH);
    default: return x;
  }
}
// derivative w.r.t. the pre-activation value x (PyTorch conventions at the kinks:
// relu'(0)=0, hardtanh/relu6 passes gradient only strictly inside (0,6))
__device__ __forceinline__ float act_grad(float x, int act) {
  switch (act) {
    case T3D_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case T3D_ACT_RELU6: return (x > 0.f && x < 6.f) ? 1.f : 0.f;
    case T3D_ACT_HSWISH: {
      // d/dx [x * relu6(x+3)/6] = relu6(x+3)/6 + x * [0<x+3<6]/6
      const float h = __builtin_amdgcn_fmed3f(x + 3.f, 0.f, 6.f) * T3D_SIXTH;
      return h + ((x > -3.f && x < 3.f) ? x * T3D_SIXTH : 0.f);
    }
    default: return 1.f;
  }
}
// Vector forms: the (wave-uniform) switch is taken ONCE per vector, not once per element -- inside an unrolled
// element loop hipcc keeps the per-element scalar compare/branch chain, which dominated the VALU-light kernels.
//   v[i] = act(v[i]*sc[i] + sh[i])
template <int NV>
__device__ __forceinline__ void act_affine_vec(float* v, const float* sc, const float* sh, int act) {
  switch (act) {
    case T3D_ACT_RELU:
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = fmaxf(fmaf(v[i], sc[i], sh[i]), 0.f);
      break;
    case T3D_ACT_RELU6:
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = __builtin_amdgcn_fmed3f(fmaf(v[i], sc[i], sh[i]), 0.f, 6.f);
      break;
    case T3D_ACT_HSWISH:
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float u = fmaf(v[i], sc[i], sh[i]);
        v[i] = u * (__builtin_amdgcn_fmed3f(u + 3.f, 0.f, 6.f) * T3D_SIXTH);
      }
      break;
    default:
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = fmaf(v[i], sc[i], sh[i]);
  }
}
//   g[i] *= act'(y[i]*sc[i] + sh[i])
template <int NV>
__device__ __forceinline__ void act_grad_affine_vec(float* g, const float* y, const float* sc, const float* sh, int act) {
  switch (act) {
    case T3D_ACT_RELU:
#pragma unroll
      for (int i = 0; i < NV; ++i) g[i] = (fmaf(y[i], sc[i], sh[i]) > 0.f) ? g[i] : 0.f;
      break;
    case T3D_ACT_RELU6:
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float u = fmaf(y[i], sc[i], sh[i]);
        g[i] = (u > 0.f && u < 6.f) ? g[i] : 0.f;
      }
      break;
    case T3D_ACT_HSWISH:
#pragma unroll
      for (int i = 0; i < NV; ++i) g[i] *= act_grad(fmaf(y[i], sc[i], sh[i]), T3D_ACT_HSWISH);
      break;
    default:
      break;
  }
}
__device__ __forceinline__ float hsigmoid(float x) { return __builtin_amdgcn_fmed3f(x + 3.f, 0.f, 6.f) * T3D_SIXTH; }

// ---- wave64 helpers --------------------------------------------------------
// sum over the 16 lanes of a DPP row (lanes sharing lane>>4); every lane gets the total
__device__ __forceinline__ float row16_sum(float v) {
#define T3D_ROR(n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (n), 0xf, 0xf, false))
  v += T3D_ROR(8);
  v += T3D_ROR(4);
  v += T3D_ROR(2);
  v += T3D_ROR(1);
#undef T3D_ROR
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- squeeze-excite pooled sums (per sample and channel, accumulated by many work items) ---------------------------
// exact == 0: fp32 atomics into float [B][C] (the plain semantics; arrival order moves the last bits).  exact != 0
// (t3d_set_exact_pool): the SAME pointer addresses int64 [B][C] in units of 2^-24 -- integer adds are associative, so the
// sums, the gates computed from them and everything downstream are bit-reproducible; one partial is rounded to 6e-8
// absolute, far below an fp32 accumulation's own error.
#define T3D_POOL_LSB 16777216.0
__device__ __forceinline__ void t3d_pool_add(float* gap, size_t idx, float v, int exact) {
  if (exact)
    atomicAdd(reinterpret_cast<unsigned long long*>(gap) + idx, (unsigned long long)__double2ll_rn((double)v * T3D_POOL_LSB));
  else
    unsafeAtomicAdd(gap + idx, v);
}
__device__ __forceinline__ float t3d_pool_get(const float* gap, size_t idx, int exact) {
  return exact ? (float)((double)reinterpret_cast<const long long*>(gap)[idx] * (1.0 / T3D_POOL_LSB)) : gap[idx];
}

// ---- end of a depthwise-backward workgroup -------------------------------------------------------------------------
// lacc [K2 + 2][Cb] fp64 in LDS: the workgroup's weight-gradient taps (rows 0 .. K2-1) and sum(dx), sum(dx*x) of its
// channel range [cbase, cbase + Cb), accumulated with fp64 LDS atomics of the lanes' fp32 partials (exact adds in any
// order for all but absurd dynamic ranges -- the fp32 LDS atomics of rounds 1-3a made two backward passes on one saved
// forward differ by 1e-2 in the gradients, through one BatchNorm coefficient ulp and bf16 rounding downstream).
//   weight gradient, slots > 0: STORED into the workgroup's own slot of dw [slots][C][K2] -- no atomics; the caller adds the
//     slots in index order (t3d_sum_slots_batched): bit-reproducible, and the flush is plain coalesced stores;
//   slots == 0: fp32 atomics into replica (workgroup % nrep), lanes along consecutive addresses (tap fastest: a wave's 64
//     adds fall into 2-3 cache lines; channel-fastest they were 64 lines and 12 % of the s=1 backward);
//   sums: fp64 atomics into the BatchNorm's replicas.
template <int K2, int NTH>
__device__ __forceinline__ void t3d_dw_flush(const double* lacc, int Cb, int cbase, int C, float* dw, double* stats, int nrep,
                                             long long rstride, int slots, int slot, int* used) {
  const int rep = (blockIdx.x + blockIdx.y) % nrep;
  if (dw) {
    float* dst = dw + (size_t)(slots > 0 ? slot : rep) * C * K2 + (size_t)cbase * K2;
    for (int i = threadIdx.x; i < K2 * Cb; i += NTH) {
      const float v = (float)lacc[(i % K2) * Cb + i / K2];
      if (slots > 0) dst[i] = v;
      else if (v != 0.f) unsafeAtomicAdd(dst + i, v);
    }
    if (used && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *used = slots > 0 ? slots : nrep;
  }
  if (stats) {
    for (int i = threadIdx.x; i < 2 * Cb; i += NTH) {
      const double v = lacc[K2 * Cb + i];
      if (v != 0.0) atomicAdd(stats + (size_t)rep * rstride + (size_t)(i / Cb) * C + cbase + i % Cb, v);
    }
  }
}
