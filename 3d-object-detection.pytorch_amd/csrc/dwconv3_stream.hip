// Depthwise 3x3 convolution forward (stride 1 / 2), barrier-free streaming kernel for gfx950 (NHWC).
//
// HBM-bound (3.4 FLOP/B): the whole job is to touch every input and output byte once with wide
// coalesced accesses and enough loads in flight.  No LDS, no barriers:
//   * a thread owns 8 channels (one 16-B bf16 vector) of ONE output column and walks down a chunk of
//     rows; lanes are laid out (channel-group fastest, then column), so a wave's load of one input row
//     is a contiguous 1 KiB segment, and the three column taps (x-1, x, x+1) are three such loads
//     shifted by one pixel (the overlap is served by L1 / L2, HBM sees each byte once);
//   * stride 1: every input row is loaded ONCE and scattered into three rotating accumulators (the
//     output rows it contributes to); stride 2: two new rows per output row, the shared odd row is
//     kept in registers;
//   * the producer's BatchNorm affine + activation is applied on load (zero padding AFTER the
//     activation, as the reference pads the activated tensor);
//   * per-channel sum / sum-of-squares for the following BatchNorm stay in the thread's registers
//     for its whole chunk, meet in LDS once per block and leave as one fp64 atomic per channel.
// The 9 x 8 weights of the thread's channels live in registers for the whole walk.
#include <cstdlib>
#include "common.h"

namespace {

struct Dw3Args {
  const void* x;
  void* y;
  const float* w;  // [C][9]
  const float *scale, *shift;
  int act;
  double* stats;
  int B, H, W, C, Ho, Wo;
  int rows_per_chunk, nchunks;
  int slab;        // 0: flattened (column, channel-group) mapping; 1: 64-group channel slabs (wide layers)
  int nitems;      // work items a thread walks: flattened: B*nchunks; slab: Wo*B*nchunks
  T3dQuant quant;  // BatchNorm sums snapped onto a fixed grid: order-independent (common.h)
  int nrep;        // reduction replicas (common.h)
  long long rstride;
  const T3dFold* fold;    // requested BatchNorm finalize of the INPUT's coefficients, derived in the prologue (common.h)
};

template <typename T, int CH> using rawvec = T __attribute__((ext_vector_type(CH)));

// raw buffer loads / stores: scalar descriptor + 32-bit scalar row offset + 32-bit lane offset -- no 64-bit vector address
// arithmetic in the row loop (20 v_lshl_add_u64 + 8 v_mul_lo_u32 per three rows of the s=1 forward before); an offset beyond
// num_records reads zeros / drops the store, which also replaces the store predicates of edge lanes
template <typename RV>
__device__ __forceinline__ RV bufload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  if constexpr (sizeof(RV) == 4) return __builtin_bit_cast(RV, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  else if constexpr (sizeof(RV) == 8) return __builtin_bit_cast(RV, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
  else return __builtin_bit_cast(RV, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
template <typename RV>
__device__ __forceinline__ void bufstore(const RV& v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  if constexpr (sizeof(RV) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
  else if constexpr (sizeof(RV) == 8) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, voff, soff, 0);
  else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}

// raw vector -> activated fp32.  Zero padding applies to the ACTIVATED tensor: taps outside the image are removed by
// zeroing the WEIGHTS of an out-of-image column (per item) and by skipping out-of-image rows (wave-uniform).
template <typename T, int CH>
__device__ __forceinline__ void activate(const rawvec<T, CH>& r, const float* sc, const float* sh, int act, bool affine,
                                         float* v) {
#pragma unroll
  for (int i = 0; i < CH; ++i) v[i] = (float)r[i];
  if (affine) act_affine_vec<CH>(v, sc, sh, act);
}

// ReLU6 of a BatchNorm affine in ONE packed instruction (bf16 storage): relu6(s x + t) = 6 clamp01((s/6) x + t/6), and
// v_pk_fma_f32 takes the clamp modifier (result clamped to [0, 1] per half; probed on the box, tools/scratch/pkclamp.hip).
// The caller scales (s, t) by 1/6 once per thread and folds the 6 into whatever multiplies the activated value next (the
// stencil weights): two v_med3_f32 per channel pair leave the row loop.  fp32 storage keeps the exact form.
__device__ __forceinline__ f32x2 pk_fma_clamp01(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// rounded store of one output vector + its BatchNorm sums, packed (two channels per v_pk_add / v_pk_fma)
template <typename T, int CH>
__device__ __forceinline__ rawvec<T, CH> round_sums2(const f32x2* acc, f32x2* psum, f32x2* psq, float m) {
  rawvec<T, CH> o;
#pragma unroll
  for (int h = 0; h < CH / 2; ++h) {
    o[2 * h] = (T)acc[h][0];
    o[2 * h + 1] = (T)acc[h][1];
    const f32x2 r = f32x2{(float)o[2 * h], (float)o[2 * h + 1]} * f32x2{m, m};      // (m = 0: a lane without this column)
    psum[h] = psum[h] + r;
    psq[h] = pk_fma(r, r, psq[h]);
  }
  return o;
}

template <typename T, int CH>
__device__ __forceinline__ void store_round(T* p, const float* acc, float* psum, float* psq) {
  rawvec<T, CH> o;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    o[i] = (T)acc[i];
    const float r = (float)o[i];
    psum[i] += r;
    psq[i] = fmaf(r, r, psq[i]);
  }
  *reinterpret_cast<rawvec<T, CH>*>(p) = o;
}

// CH channels per thread (8 -> 16-B bf16 vectors, 4 -> 8-B: more resident waves), PF input rows in flight
// ACT: compile-time activation (round 4); stride 2 in bf16 storage with ReLU6: the clamp form (pk_fma_clamp01) and packed
// stencil math -- a.act is only read by the launcher
template <typename T, int S, int CH, int PF, int ACT>
__global__ __launch_bounds__(256) void dw3_fwd_kernel(const Dw3Args a) {
  constexpr bool C6 = !std::is_same<T, float>::value && ACT == T3D_ACT_RELU6 && S == 2;
  constexpr int H2 = CH / 2;
  extern __shared__ __attribute__((aligned(16))) float lstat[];  // [2][C] doubles at the end of the kernel (sums); floats for a derived finalize
  using RV = rawvec<T, CH>;
  const int CG = a.C / CH;
  // A thread keeps ONE channel group for its whole life (weights + BatchNorm partial sums stay in registers)
  // and walks a list of (column, row-chunk, sample) items.
  //   flattened (CG < 64): thread = (column, group) of the row strip, items = (sample, chunk)
  //   slab      (CG >= 64): lane = group inside a 64-group slab (blockIdx.y), items = (column, sample, chunk);
  //                         a block then touches <= 64*CH channels when it flushes its sums
  int cg, ox_fixed = 0, q0, qstride;
  bool on;
  if (!a.slab) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    on = j < a.Wo * CG;
    cg = on ? j % CG : 0;
    ox_fixed = on ? j / CG : 0;
    q0 = blockIdx.y;
    qstride = gridDim.y;
  } else {
    cg = blockIdx.y * 64 + (threadIdx.x & 63);
    on = cg < CG;
    if (!on) cg = 0;
    q0 = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: scalar item decode
    qstride = gridDim.x * 4;
  }
  const int c0 = cg * CH;
  const bool affine = a.scale != nullptr || ACT != T3D_ACT_NONE;
  // the producer's BatchNorm finalize, derived here when requested (common.h): the block's own channel range through
  // the statistics scratch (free until the end of the kernel); one block per channel range publishes
  const int fbase = a.slab ? blockIdx.y * 64 * CH : 0, fCb = a.slab ? min(64 * CH, a.C - fbase) : a.C;
  if (a.fold) t3d_fold_block(a.fold, fbase, fCb, lstat, fCb, a.slab ? blockIdx.x == 0 : (blockIdx.x == 0 && blockIdx.y == 0));

  float wt[9][CH], sc[CH], sh[CH], psum[CH], psq[CH];
  {
    // the CH*9 weights of this thread's channels are contiguous: 16-B loads, transposed in registers
    float wb[CH * 9];
    const float4* wp = reinterpret_cast<const float4*>(a.w + (size_t)c0 * 9);   // c0*9*4 B is 16-B aligned (CH % 4 == 0)
#pragma unroll
    for (int i = 0; i < CH * 9 / 4; ++i) {
      const float4 q = wp[i];
      wb[4 * i] = q.x; wb[4 * i + 1] = q.y; wb[4 * i + 2] = q.z; wb[4 * i + 3] = q.w;
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      sc[i] = a.fold ? lstat[c0 - fbase + i] : (a.scale ? a.scale[c0 + i] : 1.f);
      sh[i] = a.fold ? lstat[fCb + c0 - fbase + i] : (a.scale ? a.shift[c0 + i] : 0.f);
      psum[i] = psq[i] = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) wt[t][i] = wb[i * 9 + t] * (C6 ? 6.f : 1.f);
      if (C6) { sc[i] *= T3D_SIXTH; sh[i] *= T3D_SIXTH; }
    }
  }
  f32x2 sc2[H2], sh2[H2];
#pragma unroll
  for (int h = 0; h < H2; ++h) { sc2[h] = f32x2{sc[2 * h], sc[2 * h + 1]}; sh2[h] = f32x2{sh[2 * h], sh[2 * h + 1]}; }
  if (a.fold) __syncthreads();      // the scratch is zeroed again at the end of the kernel
  // raw buffer addressing (scalar row offsets + 32-bit lane offsets) when the tensors allow it (< 2 GB); else 64-bit pointers
  const size_t xbytes = (size_t)a.B * a.H * a.W * a.C * sizeof(T), ybytes = (size_t)a.B * a.Ho * a.Wo * a.C * sizeof(T);
  constexpr bool CANBUF = sizeof(RV) <= 16;      // (fp32 storage with 8 channels per thread: 32-byte vectors keep the pointer path)
  const bool buf = CANBUF && xbytes < (1ull << 31);
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)ybytes, 0x00020000);
  for (int q = q0; q < a.nitems && on; q += qstride) {
  int ox, rest;
  if (!a.slab) { ox = ox_fixed; rest = q; } else { ox = q % a.Wo; rest = q / a.Wo; }
  const int chunk = rest % a.nchunks, b = rest / a.nchunks;
  const T* __restrict__ x = reinterpret_cast<const T*>(a.x) + (size_t)b * a.H * a.W * a.C + c0;
  T* __restrict__ y = reinterpret_cast<T*>(a.y) + (size_t)b * a.Ho * a.Wo * a.C + c0;
  const size_t imgrow = (size_t)b * a.H, outrow = (size_t)b * a.Ho;
  const unsigned vst = (unsigned)(ox * a.C + c0) * (unsigned)sizeof(T);
  const int oy0 = chunk * a.rows_per_chunk;
  const int oy1 = min(a.Ho, oy0 + a.rows_per_chunk);
  const int ix0 = ox * S - 1;
  const bool cok[3] = {ix0 >= 0, true, ix0 + 2 < a.W};  // ix0+1 = ox*S < W always
  float wk[9][CH];   // this item's weights: columns outside the image contribute nothing
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < CH; ++i) wk[t][i] = cok[t % 3] ? wt[t][i] : 0.f;
  // input rows this thread walks: iy_first .. iy_last (inclusive)
  const int iy_first = oy0 * S - 1, iy_last = (oy1 - 1) * S + 1;

  // loads are unconditional from clamped (always valid) addresses: out-of-image columns are cancelled by the
  // zeroed weights above, out-of-image rows are skipped below -- no divergent branches around the loads
  int coff[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) coff[c] = min(max(ix0 + c, 0), a.W - 1) * a.C;
  RV ring[PF][3];
  unsigned boff[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) boff[c] = (unsigned)(coff[c] + c0) * (unsigned)sizeof(T);
  auto fetch = [&](int iy, RV* dst) {
    if constexpr (CANBUF) {
      if (buf) {
        const unsigned so = (unsigned)((imgrow + min(max(iy, 0), a.H - 1)) * a.W * a.C * sizeof(T));     // scalar
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[c] = bufload<RV>(rsx, boff[c], so);
        return;
      }
    }
    {
      const T* rp = x + (size_t)min(max(iy, 0), a.H - 1) * a.W * a.C;
#pragma unroll
      for (int c = 0; c < 3; ++c) dst[c] = *reinterpret_cast<const RV*>(rp + coff[c]);
    }
  };
  auto put = [&](int oy, const float* acc) {       // rounded store + BatchNorm sums of one output vector
    if constexpr (CANBUF) {
      if (buf) {
        RV o;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          o[i] = (T)acc[i];
          const float r = (float)o[i];
          psum[i] += r;
          psq[i] = fmaf(r, r, psq[i]);
        }
        bufstore<RV>(o, rsy, vst, (unsigned)((outrow + oy) * a.Wo * a.C * sizeof(T)));
        return;
      }
    }
    store_round<T, CH>(y + ((size_t)oy * a.Wo + ox) * a.C, acc, psum, psq);
  };

  {
#pragma unroll
    for (int u = 0; u < PF; ++u) fetch(iy_first + u, ring[u]);
    static_assert(PF == 3, "the three rotating accumulators take their roles from the unroll index");
    float acc[3][CH];   // S=1: roles (row iy-1, iy, iy+1) = acc[u%3], acc[(u+1)%3], acc[(u+2)%3].  S=2: acc[0] only
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int i = 0; i < CH; ++i) acc[r][i] = 0.f;
    for (int base = iy_first; base <= iy_last; base += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int iy = base + u;
        if (iy <= iy_last) {
          const bool rok = iy >= 0 && iy < a.H;   // wave-uniform in the slab mapping, near-uniform otherwise
          float v[3][CH];
          if (C6) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
              for (int h = 0; h < H2; ++h) {
                const f32x2 t = pk_fma_clamp01(f32x2{(float)ring[u][c][2 * h], (float)ring[u][c][2 * h + 1]}, sc2[h], sh2[h]);
                v[c][2 * h] = t[0];
                v[c][2 * h + 1] = t[1];
              }
          } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) activate<T, CH>(ring[u][c], sc, sh, ACT, affine, v[c]);
          }
          fetch(iy + PF, ring[u]);   // refill this slot: PF rows ahead
          if constexpr (S == 1) {
            float* accA = acc[u % 3];
            float* accB = acc[(u + 1) % 3];
            float* accC = acc[(u + 2) % 3];
            // row iy feeds output rows iy+1 (ky=0), iy (ky=1), iy-1 (ky=2)
            if (rok) {
#pragma unroll
              for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < CH; i += 2) {
                  const f32x2 vv = {v[c][i], v[c][i + 1]};
                  f32x2 ra = pk_fma(vv, f32x2{wk[6 + c][i], wk[6 + c][i + 1]}, f32x2{accA[i], accA[i + 1]});
                  f32x2 rb = pk_fma(vv, f32x2{wk[3 + c][i], wk[3 + c][i + 1]}, f32x2{accB[i], accB[i + 1]});
                  f32x2 rc = pk_fma(vv, f32x2{wk[c][i], wk[c][i + 1]}, f32x2{accC[i], accC[i + 1]});
                  accA[i] = ra[0]; accA[i + 1] = ra[1];
                  accB[i] = rb[0]; accB[i + 1] = rb[1];
                  accC[i] = rc[0]; accC[i + 1] = rc[1];
                }
            }
            const int oy = iy - 1;
            if (oy >= oy0 && oy < oy1) put(oy, accA);
#pragma unroll
            for (int i = 0; i < CH; ++i) accA[i] = 0.f;   // becomes the "row iy+2" accumulator of the next row
          } else {
            float* accA = acc[0];
            // rows 2oy-1 (ky=0), 2oy (ky=1), 2oy+1 (ky=2, and ky=0 of the next output row)
            const int rel = iy - iy_first;      // 0: ky=0 of oy0; odd: ky=1; even>0: ky=2 of oy and ky=0 of oy+1
            if (!rok) {
#pragma unroll
              for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < CH; ++i) v[c][i] = 0.f;
            }
            if (rel & 1) {
#pragma unroll
              for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < CH; ++i) accA[i] = fmaf(v[c][i], wk[3 + c][i], accA[i]);
            } else {
              if (rel > 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                  for (int i = 0; i < CH; ++i) accA[i] = fmaf(v[c][i], wk[6 + c][i], accA[i]);
                const int oy = oy0 + (rel >> 1) - 1;
                put(oy, accA);
              }
#pragma unroll
              for (int i = 0; i < CH; ++i) accA[i] = v[0][i] * wk[0][i];
#pragma unroll
              for (int c = 1; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < CH; ++i) accA[i] = fmaf(v[c][i], wk[c][i], accA[i]);
            }
          }
        }
      }
    }
  }
  }  // item loop

  if (a.stats) {
    // only the block's own channel range (whole tensor, or one 64-group slab) goes through LDS and out
    const int cbase = a.slab ? blockIdx.y * 64 * CH : 0;
    const int Cb = a.slab ? min(64 * CH, a.C - cbase) : a.C;
    double* dstat = reinterpret_cast<double*>(lstat);     // [2][Cb], exact adds of snapped partial sums (common.h)
    for (int i = threadIdx.x; i < 2 * Cb; i += 256) dstat[i] = 0.0;
    __syncthreads();
    if (on) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        atomicAdd(dstat + c0 - cbase + i, t3d_snap(psum[i], a.quant, false));
        atomicAdd(dstat + Cb + c0 - cbase + i, t3d_snap(psq[i], a.quant, true));
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * Cb; i += 256)
      if (dstat[i] != 0.0)
        atomicAdd(a.stats + (size_t)((blockIdx.x + blockIdx.y) % a.nrep) * a.rstride + (size_t)(i / Cb) * a.C + cbase + i % Cb,
                  dstat[i]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Stride-1 variant with TWO output columns per thread.  The kernel above is VALU-bound (PMC: SIMD VALU ~60 % busy at
// 2.6 TB/s): every input element is converted + activated by the three threads whose 3-tap windows contain it.
// Here a thread owns columns (2p, 2p+1): four column loads feed two outputs (each element is activated twice instead
// of three times, address / loop overhead is shared), and all multiply-adds are packed (v_pk_fma_f32: two channels per
// issue slot).  Zero padding: a 0/1 mask per out-of-image column (per item), skipped rows.
// ACT: the input's activation as a compile-time constant (round 4; the runtime switch kept every variant's code in the row
// loop behind scalar branches); C6: ReLU6 through the clamp modifier (pk_fma_clamp01 above)
template <typename T, int PF, int ACT>
__global__ __launch_bounds__(256) void dw3_fwd2_kernel(const Dw3Args a) {
  constexpr int CH = 4, H2 = CH / 2;
  constexpr bool C6 = !std::is_same<T, float>::value && ACT == T3D_ACT_RELU6;
  extern __shared__ __attribute__((aligned(16))) float lstat[];  // [2][C] doubles at the end of the kernel (sums); floats for a derived finalize
  using RV = rawvec<T, CH>;
  const int CG = a.C / CH, Wp = (a.Wo + 1) / 2;
  int cg, xp_fixed = 0, q0, qstride;
  bool on;
  if (!a.slab) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    on = j < Wp * CG;
    cg = on ? j % CG : 0;
    xp_fixed = on ? j / CG : 0;
    q0 = blockIdx.y;
    qstride = gridDim.y;
  } else {
    cg = blockIdx.y * 64 + (threadIdx.x & 63);
    on = cg < CG;
    if (!on) cg = 0;
    q0 = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: scalar item decode
    qstride = gridDim.x * 4;
  }
  const int c0 = cg * CH;
  const bool affine = a.scale != nullptr || ACT != T3D_ACT_NONE;
  // requested BatchNorm finalize of the producer: derived here (see dw3_fwd_kernel)
  const int fbase = a.slab ? blockIdx.y * 64 * CH : 0, fCb = a.slab ? min(64 * CH, a.C - fbase) : a.C;
  f32x2 w2[9][H2], sc2[H2], sh2[H2];
  f32x2 psum[H2], psq[H2];
  {
    float wb[CH * 9];
    auto load_w = [&]() {
      const float4* wp = reinterpret_cast<const float4*>(a.w + (size_t)c0 * 9);
#pragma unroll
      for (int i = 0; i < CH * 9 / 4; ++i) {
        const float4 q = wp[i];
        wb[4 * i] = q.x; wb[4 * i + 1] = q.y; wb[4 * i + 2] = q.z; wb[4 * i + 3] = q.w;
      }
    };
    // (the stencil weights are fetched while the sums of a derived finalize are in flight)
    if (a.fold) t3d_fold_block(a.fold, fbase, fCb, lstat, fCb, a.slab ? blockIdx.x == 0 : (blockIdx.x == 0 && blockIdx.y == 0), load_w);
    else load_w();
#pragma unroll
    for (int h = 0; h < H2; ++h) {
      if (a.fold) {
        sc2[h] = f32x2{lstat[c0 - fbase + 2 * h], lstat[c0 - fbase + 2 * h + 1]};
        sh2[h] = f32x2{lstat[fCb + c0 - fbase + 2 * h], lstat[fCb + c0 - fbase + 2 * h + 1]};
      } else {
        sc2[h] = f32x2{a.scale ? a.scale[c0 + 2 * h] : 1.f, a.scale ? a.scale[c0 + 2 * h + 1] : 1.f};
        sh2[h] = f32x2{a.scale ? a.shift[c0 + 2 * h] : 0.f, a.scale ? a.shift[c0 + 2 * h + 1] : 0.f};
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) w2[t][h] = f32x2{wb[(2 * h) * 9 + t], wb[(2 * h + 1) * 9 + t]};
      if (C6) {                        // relu6(s x + t) = 6 clamp01((s/6) x + t/6): the 6 rides on the stencil weights
        sc2[h] = sc2[h] * f32x2{T3D_SIXTH, T3D_SIXTH};
        sh2[h] = sh2[h] * f32x2{T3D_SIXTH, T3D_SIXTH};
#pragma unroll
        for (int t = 0; t < 9; ++t) w2[t][h] = w2[t][h] * f32x2{6.f, 6.f};
      }
      psum[h] = psq[h] = f32x2{0.f, 0.f};
    }
  }
  if (a.fold) __syncthreads();      // the scratch is zeroed again at the end of the kernel

  const size_t xbytes = (size_t)a.B * a.H * a.W * a.C * sizeof(T), ybytes = (size_t)a.B * a.Ho * a.Wo * a.C * sizeof(T);
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)ybytes, 0x00020000);
  const unsigned OOB = 0x80000000u;         // beyond num_records of every tensor here (the launcher refuses >= 2 GB)
  for (int q = q0; q < a.nitems; q += qstride) {
    int xp, rest;
    if (!a.slab) { xp = xp_fixed; rest = q; } else { xp = q % Wp; rest = q / Wp; }
    const int chunk = rest % a.nchunks, b = rest / a.nchunks;          // wave-uniform
    const int oy0 = chunk * a.rows_per_chunk, oy1 = min(a.Ho, oy0 + a.rows_per_chunk);
    const int x0 = 2 * xp;                  // output columns x0 (always valid) and x0+1
    const bool validB = x0 + 1 < a.W;
    // input columns x0-1 .. x0+2; 0/1 masks for the ones outside the image (column x0 is always inside)
    const float m0 = x0 - 1 >= 0 ? 1.f : 0.f, m2 = validB ? 1.f : 0.f, m3 = x0 + 2 < a.W ? 1.f : 0.f;
    // (interior waves skip the mask multiplies altogether: wave-uniform)
    const bool edge = __any(x0 - 1 < 0 || x0 + 2 >= a.W);
    unsigned coff[4];                       // lane part of the load addresses (bytes)
#pragma unroll
    for (int c = 0; c < 4; ++c) coff[c] = (unsigned)(min(max(x0 - 1 + c, 0), a.W - 1) * a.C + c0) * (unsigned)sizeof(T);
    const unsigned stA = on ? (unsigned)(x0 * a.C + c0) * (unsigned)sizeof(T) : OOB;
    const unsigned stB = (on && validB) ? stA + (unsigned)(a.C * sizeof(T)) : OOB;
    const float mB = (on && validB) ? 1.f : 0.f;
    const int iy_first = oy0 - 1, iy_last = oy1;
    const size_t imgrow = (size_t)b * a.H, outrow = (size_t)b * a.Ho;

    RV ring[PF][4];
    auto fetch = [&](int iy, RV* dst) {
      const unsigned so = (unsigned)((imgrow + min(max(iy, 0), a.H - 1)) * a.W * a.C * sizeof(T));     // scalar
#pragma unroll
      for (int c = 0; c < 4; ++c) dst[c] = bufload<RV>(rsx, coff[c], so);
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) {      // ring fill in slot order, pinned (see dw3_bwd2_kernel's row loop)
      __builtin_amdgcn_sched_barrier(0);
      fetch(iy_first + u, ring[u]);
    }
    __builtin_amdgcn_sched_barrier(0);

    static_assert(PF == 3, "accumulator roles come from the unroll index");
    f32x2 accA[3][H2], accB[3][H2];   // roles (output row iy-1, iy, iy+1) = [u%3], [(u+1)%3], [(u+2)%3]
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int h = 0; h < H2; ++h) accA[r][h] = accB[r][h] = f32x2{0.f, 0.f};

    // padded walk without a guard, reads pinned above the refill: keeps the ring PF rows deep in the generated code
    // (dw3_bwd2_kernel in dwconv3_bwd_stream.hip has the full note); rows past iy_last contribute nothing
    const int iy_end = iy_first + (iy_last - iy_first + PF) / PF * PF;
    for (int base = iy_first; base < iy_end; base += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int iy = base + u;
        {
          const bool rok = iy >= 0 && iy < a.H && iy <= iy_last;
          f32x2 v[4][H2];
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int h = 0; h < H2; ++h) {
              v[c][h] = f32x2{(float)ring[u][c][2 * h], (float)ring[u][c][2 * h + 1]};
              asm volatile("" : "+v"(v[c][h]));
            }
          __builtin_amdgcn_sched_barrier(0);
          fetch(iy + PF, ring[u]);
          __builtin_amdgcn_sched_barrier(0);
          if (C6) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
              for (int h = 0; h < H2; ++h) v[c][h] = pk_fma_clamp01(v[c][h], sc2[h], sh2[h]);
          } else if (affine) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
              for (int h = 0; h < H2; ++h) v[c][h] = pk_fma(v[c][h], sc2[h], sh2[h]);
            switch (ACT) {
              case T3D_ACT_RELU:
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                  for (int h = 0; h < H2; ++h) v[c][h] = f32x2{fmaxf(v[c][h][0], 0.f), fmaxf(v[c][h][1], 0.f)};
                break;
              case T3D_ACT_RELU6:
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                  for (int h = 0; h < H2; ++h)
                    v[c][h] = f32x2{__builtin_amdgcn_fmed3f(v[c][h][0], 0.f, 6.f), __builtin_amdgcn_fmed3f(v[c][h][1], 0.f, 6.f)};
                break;
              case T3D_ACT_HSWISH:
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                  for (int h = 0; h < H2; ++h) {
                    const f32x2 t = v[c][h];
                    v[c][h] = f32x2{t[0] * (__builtin_amdgcn_fmed3f(t[0] + 3.f, 0.f, 6.f) * T3D_SIXTH),
                                    t[1] * (__builtin_amdgcn_fmed3f(t[1] + 3.f, 0.f, 6.f) * T3D_SIXTH)};
                  }
                break;
              default: break;
            }
          }
          if (edge) {
#pragma unroll
            for (int h = 0; h < H2; ++h) {   // zero padding of the activated tensor
              v[0][h] = v[0][h] * f32x2{m0, m0};
              v[2][h] = v[2][h] * f32x2{m2, m2};
              v[3][h] = v[3][h] * f32x2{m3, m3};
            }
          }
          f32x2* aA = accA[u % 3];
          f32x2* aB = accB[u % 3];
          if (rok) {
            f32x2* bA = accA[(u + 1) % 3];
            f32x2* cA = accA[(u + 2) % 3];
            f32x2* bB = accB[(u + 1) % 3];
            f32x2* cB = accB[(u + 2) % 3];
            // input row iy feeds output rows iy-1 (ky=2), iy (ky=1), iy+1 (ky=0); column A uses taps c=0..2, B c=1..3
#pragma unroll
            for (int h = 0; h < H2; ++h) {
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                aA[h] = pk_fma(v[c][h], w2[6 + c][h], aA[h]);
                bA[h] = pk_fma(v[c][h], w2[3 + c][h], bA[h]);
                cA[h] = pk_fma(v[c][h], w2[c][h], cA[h]);
                aB[h] = pk_fma(v[c + 1][h], w2[6 + c][h], aB[h]);
                bB[h] = pk_fma(v[c + 1][h], w2[3 + c][h], bB[h]);
                cB[h] = pk_fma(v[c + 1][h], w2[c][h], cB[h]);
              }
            }
          }
          const int oy = iy - 1;
          if (oy >= oy0 && oy < oy1) {      // wave-uniform
            const unsigned so = (unsigned)((outrow + oy) * a.Wo * a.C * sizeof(T));
            bufstore<RV>(round_sums2<T, CH>(aA, psum, psq, 1.f), rsy, stA, so);      // (lanes that are off never flush their sums)
            bufstore<RV>(round_sums2<T, CH>(aB, psum, psq, mB), rsy, stB, so);
          }
#pragma unroll
          for (int h = 0; h < H2; ++h) aA[h] = aB[h] = f32x2{0.f, 0.f};
        }
      }
    }
  }  // item loop

  if (a.stats) {
    // only the block's own channel range (whole tensor, or one 64-group slab) goes through LDS and out
    const int cbase = a.slab ? blockIdx.y * 64 * CH : 0;
    const int Cb = a.slab ? min(64 * CH, a.C - cbase) : a.C;
    double* dstat = reinterpret_cast<double*>(lstat);     // [2][Cb], exact adds of snapped partial sums (common.h)
    for (int i = threadIdx.x; i < 2 * Cb; i += 256) dstat[i] = 0.0;
    __syncthreads();
    if (on) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        atomicAdd(dstat + c0 - cbase + i, t3d_snap(psum[i / 2][i % 2], a.quant, false));
        atomicAdd(dstat + Cb + c0 - cbase + i, t3d_snap(psq[i / 2][i % 2], a.quant, true));
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * Cb; i += 256)
      if (dstat[i] != 0.0)
        atomicAdd(a.stats + (size_t)((blockIdx.x + blockIdx.y) % a.nrep) * a.rstride + (size_t)(i / Cb) * a.C + cbase + i % Cb,
                  dstat[i]);
  }
}

template <typename T, int CH>
int launch_ch(Dw3Args& a, int s, hipStream_t st) {
  constexpr int PF = 3;
  const int CG = a.C / CH;
  // row chunks: enough work items to fill the chip (several resident waves per SIMD), but long enough that
  // the 1-2 halo rows re-read per chunk stay a small fraction
  const long long per_row_chunk = (long long)a.B * a.Wo * CG;
  int nchunks = (int)((256LL * 64 * 40 + per_row_chunk - 1) / per_row_chunk);
  int max_chunks = a.Ho / 8;
  if (max_chunks < 1) max_chunks = 1;
  if (nchunks > max_chunks) nchunks = max_chunks;
  if (nchunks < 1) nchunks = 1;
  a.rows_per_chunk = cdiv(a.Ho, nchunks);
  a.nchunks = cdiv(a.Ho, a.rows_per_chunk);
  const bool two_col = true;
  // (the two-column kernel addresses through 32-bit buffer offsets: tensors below 2 GB)
  const bool use2 = (s == 1 && CH == 4 && two_col && (size_t)a.B * a.H * a.W * a.C * sizeof(T) < (1ull << 31));
  const int Wcols = use2 ? (a.Wo + 1) / 2 : a.Wo;     // work items per row: column pairs or columns
  dim3 grid;
  // persistent blocks: enough to fill the chip, few enough that the per-block flush stays cheap (tools/sweep_dwf.sh)
  static const int tb_env = getenv("T3D_DW_TB") ? atoi(getenv("T3D_DW_TB")) : 0;      // (sweep knob, tools/scratch/sweep_tb.sh)
  // (round 4, tools/scratch/sweep_tb.sh: the small-spatial s=1 layers gain 4-5 % at 768: 28x28x192 38.0 -> 35.7 us, 14x14x384 24.8 -> 23.7)
  const int target_blocks = tb_env ? tb_env : ((s == 1 && a.H > 28) ? 512 : 768);
  a.nrep = g_t3d_reduce.nrep;
  a.rstride = g_t3d_reduce.stats_stride;
  if (CG < 64) {
    a.slab = 0;
    a.nitems = a.B * a.nchunks;
    const int jb = cdiv(Wcols * CG, 256);
    int gy = target_blocks / jb;
    if (gy > a.nitems) gy = a.nitems;
    if (gy < 1) gy = 1;
    grid = dim3(jb, gy);
  } else {
    a.slab = 1;
    a.nitems = Wcols * a.B * a.nchunks;
    const int ns = cdiv(CG, 64);
    int gx = target_blocks / ns;
    if (gx > cdiv(a.nitems, 4)) gx = cdiv(a.nitems, 4);
    if (gx < 1) gx = 1;
    grid = dim3(gx, ns);
  }
  const size_t lds = (size_t)2 * a.C * sizeof(double);
  a.fold = t3d_take_fold(a.scale);
  // (throughput mode only: in fp32 storage the sums' order noise stays at 1e-6 of the outputs and the grid would be one more
  // perturbation between the parity mode and the reference)
  a.quant = (a.stats && std::is_same<T, bf16_t>::value && !T3D_ENV_SET("T3D_NO_SNAP")) ? t3d_quant_for((long long)a.B * a.Ho * a.Wo)
                                                                                  : T3dQuant{0.0, 0.0};
  if (use2) {
    switch (a.act) {
      case T3D_ACT_RELU: T3D_LAUNCH_TIMED((dw3_fwd2_kernel<T, PF, T3D_ACT_RELU>), grid, dim3(256), lds, st, a); break;
      case T3D_ACT_RELU6: T3D_LAUNCH_TIMED((dw3_fwd2_kernel<T, PF, T3D_ACT_RELU6>), grid, dim3(256), lds, st, a); break;
      case T3D_ACT_HSWISH: T3D_LAUNCH_TIMED((dw3_fwd2_kernel<T, PF, T3D_ACT_HSWISH>), grid, dim3(256), lds, st, a); break;
      default: T3D_LAUNCH_TIMED((dw3_fwd2_kernel<T, PF, T3D_ACT_NONE>), grid, dim3(256), lds, st, a); break;
    }
  }
  else {
#define T3D_DWF(SV, ACTV) T3D_LAUNCH_TIMED((dw3_fwd_kernel<T, SV, CH, PF, ACTV>), grid, dim3(256), lds, st, a)
#define T3D_DWF_S(SV)                                          \
  switch (a.act) {                                             \
    case T3D_ACT_RELU: T3D_DWF(SV, T3D_ACT_RELU); break;       \
    case T3D_ACT_RELU6: T3D_DWF(SV, T3D_ACT_RELU6); break;     \
    case T3D_ACT_HSWISH: T3D_DWF(SV, T3D_ACT_HSWISH); break;   \
    default: T3D_DWF(SV, T3D_ACT_NONE); break;                 \
  }
    if (s == 1) { T3D_DWF_S(1) } else { T3D_DWF_S(2) }
#undef T3D_DWF_S
#undef T3D_DWF
  }
  T3D_CHECK_LAUNCH();
  return T3D_OK;
}

template <typename T>
int launch(Dw3Args& a, int s, hipStream_t st) {
  const int ch = 4;
  return ch == 8 ? launch_ch<T, 8>(a, s, st) : launch_ch<T, 4>(a, s, st);
}

}  // namespace

// Called by t3d_dwconv_fwd for k == 3 without a squeeze-excite pooled output.
int t3d_dw3_fwd_stream(int dtype, const void* x, const t3d_prologue* pro, const float* w, void* y, double* stats, int B,
                       int H, int W, int C, int stride, hipStream_t st) {
  Dw3Args a{};
  a.x = x; a.y = y; a.w = w; a.stats = stats;
  if (pro) { a.scale = pro->scale; a.shift = pro->shift; a.act = pro->act; }
  a.B = B; a.H = H; a.W = W; a.C = C;
  a.Ho = (H + 2 - 3) / stride + 1;
  a.Wo = (W + 2 - 3) / stride + 1;
  if (dtype == T3D_F32) return launch<float>(a, stride, st);
  if (dtype == T3D_BF16) return launch<bf16_t>(a, stride, st);
  if (dtype == T3D_F16) return launch<f16_t>(a, stride, st);       // inference forward
  return T3D_ERR_ARG;
}
