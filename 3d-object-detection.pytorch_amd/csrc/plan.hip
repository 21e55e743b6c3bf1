// Step plans (include/t3d.h: t3d_plan_*): a recorded list of entry-point calls, stream forks and read-backs that
// t3d_plan_run replays from ONE host call.
//
// Why: the train iteration (torchdet3d/trainer/train.py:44-55 of the reference: forward, losses, backward, optimizer step)
// is ~230 enqueue-only calls through this C ABI.  Issued one by one from the Python host they cost ~3 ms of host time per
// step (ctypes marshalling, the engine's bookkeeping between two launches) against ~6.9 ms of device time: any step much
// below that would be host-bound, and eight ranks multiply the jitter.  The entry points take only plain scalars, device
// pointers that do not move between steps (the engine's buffers are allocated once) and three small structs, so one
// recorded step IS the next step -- except for a handful of values (the batch's input pointers, the dropout counter, the
// optimizer's step count and learning rate, the read-back slot), which are SLOTS filled in by the caller at every run.
//
// The replay goes through the same exported entry points as the recording (a typed thunk per entry point, generated
// from the declarations of include/t3d.h -- no ABI tricks), so host-side state such as the pending BatchNorm fold
// request, the reduction replicas or the workspace pointers is driven exactly as in the eager step and the launches are
// bit-identical (tests/test_gpu_step_plan.py).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "common.h"

namespace {

// ---- typed thunks: u64 argument words -> the entry point's own C signature -------------------------------------------
template <class T>
inline T word_to(uint64_t v) {
  if constexpr (std::is_pointer<T>::value) {
    return reinterpret_cast<T>(static_cast<uintptr_t>(v));
  } else if constexpr (std::is_same<T, float>::value) {
    const uint32_t b = static_cast<uint32_t>(v);
    float f;
    std::memcpy(&f, &b, 4);
    return f;
  } else if constexpr (std::is_same<T, double>::value) {
    double d;
    std::memcpy(&d, &v, 8);
    return d;
  } else {
    return static_cast<T>(static_cast<int64_t>(v));       // int, long long, unsigned long long, int64_t
  }
}

template <class... A, size_t... I>
inline int call_words(int (*fn)(A...), const uint64_t* w, std::index_sequence<I...>) {
  return fn(word_to<A>(w[I])...);
}

template <class... A>
constexpr int arity(int (*)(A...)) { return (int)sizeof...(A); }

template <auto Fn>
struct Thunk;
template <class... A, int (*Fn)(A...)>
struct Thunk<Fn> {
  static int run(const uint64_t* w) { return call_words(Fn, w, std::index_sequence_for<A...>{}); }
};

struct Entry { const char* name; int (*run)(const uint64_t*); int nargs; };
#define T3D_E(fn) {#fn, &Thunk<&fn>::run, arity(&fn)}
// every entry point a step can contain (queries that return sizes / flags are not enqueue calls and are not listed)
const Entry kEntries[] = {
    T3D_E(t3d_dwconv_fwd), T3D_E(t3d_bn_finalize), T3D_E(t3d_bn_eval_affine), T3D_E(t3d_bn_eval_affine_batched),
    T3D_E(t3d_pwconv_fwd), T3D_E(t3d_pwconv_dgrad), T3D_E(t3d_pack_weight), T3D_E(t3d_sum_replicas_batched),
    T3D_E(t3d_set_dw_slots), T3D_E(t3d_set_exact_pool), T3D_E(t3d_sum_slots_batched), T3D_E(t3d_dwconv_bwd),
    T3D_E(t3d_pwconv_wgrad), T3D_E(t3d_pwconv_yfree_prep), T3D_E(t3d_pwconv_dgrad_yfree), T3D_E(t3d_pwconv_wgrad_yfree),
    T3D_E(t3d_pwconv_yfree_prep2), T3D_E(t3d_pwconv_bwd_yfree), T3D_E(t3d_pwconv_bwd_yfree_w), T3D_E(t3d_pwconv_wgrad_yfree_finish),
    T3D_E(t3d_bn_bwd_finalize), T3D_E(t3d_stem_im2col), T3D_E(t3d_stem_im2col_u8), T3D_E(t3d_crop_resize_u8),
    T3D_E(t3d_pwconv_fwd_mat), T3D_E(t3d_im2col), T3D_E(t3d_im2col_nchw), T3D_E(t3d_col2im_bwd), T3D_E(t3d_pack_conv_weight),
    T3D_E(t3d_unpack_conv_grad), T3D_E(t3d_maxpool_fwd), T3D_E(t3d_maxpool_bwd), T3D_E(t3d_res_relu_fwd),
    T3D_E(t3d_res_relu_bwd), T3D_E(t3d_subsample), T3D_E(t3d_ir_block_eval), T3D_E(t3d_bn_apply), T3D_E(t3d_bn_act_bwd),
    T3D_E(t3d_bn_apply_gram), T3D_E(t3d_gram_bn_finalize), T3D_E(t3d_gap_fwd), T3D_E(t3d_gap_bwd), T3D_E(t3d_pool_fwd), T3D_E(t3d_pool_bwd), T3D_E(t3d_head_fwd),
    T3D_E(t3d_linear_fwd), T3D_E(t3d_head_fwd_all), T3D_E(t3d_head_bwd), T3D_E(t3d_head_bwd_weights), T3D_E(t3d_se_fwd),
    T3D_E(t3d_se_bwd), T3D_E(t3d_se_fwd_fused), T3D_E(t3d_se_bwd_data), T3D_E(t3d_se_bwd_weights), T3D_E(t3d_se_after_sums),
    T3D_E(t3d_se_after_apply), T3D_E(t3d_set_reduction_replicas), T3D_E(t3d_set_workspace), T3D_E(t3d_set_main_workspace),
    T3D_E(t3d_fold_request), T3D_E(t3d_pack_weights_batched), T3D_E(t3d_pwconv_pack_frag), T3D_E(t3d_adamw_step), T3D_E(t3d_set_grad_watch),
    T3D_E(t3d_zero_batched), T3D_E(t3d_copy_cols), T3D_E(t3d_bn_bias_grad), T3D_E(t3d_se_bwd_affine), T3D_E(t3d_dropout_mask),
    T3D_E(t3d_loss_fwd_bwd), T3D_E(t3d_metrics_per_sample), T3D_E(t3d_iou3d), T3D_E(t3d_box_iou3d), T3D_E(t3d_ssd_decode_nms),
    T3D_E(t3d_expdw_fwd), T3D_E(t3d_conv3x3_fwd), T3D_E(t3d_conv3x3_dgrad), T3D_E(t3d_conv3x3_wgrad), T3D_E(t3d_pack_conv3x3_dgrad_weight),
};
#undef T3D_E
constexpr int kNumEntries = (int)(sizeof(kEntries) / sizeof(kEntries[0]));
constexpr int kMaxArgs = 24;

enum OpKind { OP_CALL = 0, OP_FORK = 1, OP_COPY_D2H = 2, OP_EVENT_RECORD = 3 };
enum ArgKind { ARG_WORD = 0, ARG_STRUCT = 1, ARG_SLOT = 2 };

struct Op {
  int kind;
  int entry;                    // OP_CALL: index into kEntries
  int nargs;
  uint64_t w[kMaxArgs];         // literal words (ARG_STRUCT: offset into the plan's struct arena, fixed up to a pointer at run time)
  uint8_t ak[kMaxArgs];         // ArgKind per argument
  // OP_FORK: w[0] = stream recorded on, w[1] = stream that waits, w[2] = 1: wait only (`ev` is the stop event of an earlier
  // call's last kernel, owned by that op);  OP_COPY_D2H: dst, src, bytes, stream;  OP_EVENT_RECORD: event, stream
  // OP_CALL: ev != null: the kernels this call launches on stream `sig_stream` carry `ev` as their stop event (common.h:
  // T3dSignal) -- the hand-off of t3d_plan_add_fork_after
  hipEvent_t ev;
  hipStream_t sig_stream;
};

}  // namespace

struct t3d_plan {
  // T3D_PLAN_PROFILE=1 (debugging aid): host nanoseconds spent inside each entry point / fork, printed by t3d_plan_destroy
  std::vector<unsigned long long> host_ns, host_calls;
  bool profile = false;
  std::vector<Op> ops;
  std::vector<int> seg_end;     // op index one past each closed segment
  std::vector<uint64_t> arena;  // struct copies (8-byte aligned)
  std::vector<uint8_t> timed;   // per entry: attach caller events to this entry point's launches (t3d_plan_time_entry)
  int failed_op = -1, failed_rc = 0;
  t3d_plan() : timed(kNumEntries, 0) {
    profile = getenv("T3D_PLAN_PROFILE") != nullptr;
    if (profile) { host_ns.assign(kNumEntries + 4, 0); host_calls.assign(kNumEntries + 4, 0); }
  }
};

static int find_entry(const char* name) {
  for (int i = 0; i < kNumEntries; ++i)
    if (!std::strcmp(kEntries[i].name, name)) return i;
  return -1;
}

extern "C" int t3d_plan_create(t3d_plan** out) {
  if (!out) return T3D_ERR_ARG;
  *out = new t3d_plan();
  return T3D_OK;
}

extern "C" int t3d_plan_destroy(t3d_plan* p) {
  if (!p) return T3D_OK;
  if (p->profile) {
    unsigned long long tot = 0;
    for (size_t i = 0; i < p->host_ns.size(); ++i) tot += p->host_ns[i];
    fprintf(stderr, "t3d_plan host profile: %.1f us per op list run in total\n", tot / 1e3);
    for (size_t i = 0; i < p->host_ns.size(); ++i)
      if (p->host_calls[i])
        fprintf(stderr, "  %-34s %7llu calls %9.1f us  %6.2f us/call\n", i < (size_t)kNumEntries ? kEntries[i].name : (i == (size_t)kNumEntries + 1 ? "fork" : "copy/event"),
                p->host_calls[i], p->host_ns[i] / 1e3, p->host_ns[i] / 1e3 / p->host_calls[i]);
  }
  for (Op& o : p->ops)
    if (o.ev && !(o.kind == OP_FORK && o.w[2])) (void)hipEventDestroy(o.ev);
  delete p;
  return T3D_OK;
}

extern "C" int t3d_plan_add_call(t3d_plan* p, const char* entry, int nargs, const int* kinds, const unsigned long long* words,
                                 const int* struct_bytes) {
  if (!p || !entry || nargs < 0 || nargs > kMaxArgs || (nargs && (!kinds || !words))) return T3D_ERR_ARG;
  const int e = find_entry(entry);
  if (e < 0) return T3D_ERR_UNSUPPORTED;
  if (kEntries[e].nargs != nargs) return T3D_ERR_ARG;
  Op o{};
  o.kind = OP_CALL; o.entry = e; o.nargs = nargs;
  for (int i = 0; i < nargs; ++i) {
    o.ak[i] = (uint8_t)kinds[i];
    if (kinds[i] == ARG_STRUCT) {
      const int nb = struct_bytes ? struct_bytes[i] : 0;
      if (nb <= 0 || !words[i]) return T3D_ERR_ARG;
      const size_t off = p->arena.size();
      p->arena.resize(off + (nb + 7) / 8);
      std::memcpy(&p->arena[off], reinterpret_cast<const void*>(static_cast<uintptr_t>(words[i])), nb);
      o.w[i] = off;
    } else if (kinds[i] == ARG_WORD || kinds[i] == ARG_SLOT) {
      o.w[i] = words[i];
    } else {
      return T3D_ERR_ARG;
    }
  }
  p->ops.push_back(o);
  return T3D_OK;
}

extern "C" int t3d_plan_add_fork(t3d_plan* p, void* from_stream, void* to_stream) {
  if (!p) return T3D_ERR_ARG;
  Op o{};
  o.kind = OP_FORK;
  o.w[0] = reinterpret_cast<uintptr_t>(from_stream);
  o.w[1] = reinterpret_cast<uintptr_t>(to_stream);
  if (hipEventCreateWithFlags(&o.ev, hipEventDisableTiming) != hipSuccess) return T3D_ERR_LAUNCH;
  p->ops.push_back(o);
  return T3D_OK;
}

// The stream `to_stream` waits for the LAST kernel that call op `producer_op` launches on `from_stream` -- through the stop
// event of that kernel's own dispatch packet: nothing is enqueued on `from_stream` (include/t3d.h).
extern "C" int t3d_plan_add_fork_after(t3d_plan* p, int producer_op, void* from_stream, void* to_stream) {
  if (!p || producer_op < 0 || producer_op >= (int)p->ops.size() || p->ops[producer_op].kind != OP_CALL) return T3D_ERR_ARG;
  Op& prod = p->ops[producer_op];
  hipStream_t from = reinterpret_cast<hipStream_t>(from_stream);
  if (prod.ev && prod.sig_stream != from) return T3D_ERR_ARG;
  if (!prod.ev) {
    if (hipEventCreateWithFlags(&prod.ev, hipEventDisableTiming) != hipSuccess) return T3D_ERR_LAUNCH;
    prod.sig_stream = from;
  }
  Op o{};
  o.kind = OP_FORK;
  o.w[0] = reinterpret_cast<uintptr_t>(from_stream);
  o.w[1] = reinterpret_cast<uintptr_t>(to_stream);
  o.w[2] = 1;
  o.ev = prod.ev;
  p->ops.push_back(o);
  return T3D_OK;
}

extern "C" int t3d_plan_add_copy_d2h(t3d_plan* p, int dst_slot, const void* src, long long bytes, void* stream) {
  if (!p || dst_slot < 0 || !src || bytes <= 0) return T3D_ERR_ARG;
  Op o{};
  o.kind = OP_COPY_D2H;
  o.w[0] = (uint64_t)dst_slot;
  o.w[1] = reinterpret_cast<uintptr_t>(src);
  o.w[2] = (uint64_t)bytes;
  o.w[3] = reinterpret_cast<uintptr_t>(stream);
  p->ops.push_back(o);
  return T3D_OK;
}

extern "C" int t3d_plan_add_event_record(t3d_plan* p, int event_slot, void* stream) {
  if (!p || event_slot < 0) return T3D_ERR_ARG;
  Op o{};
  o.kind = OP_EVENT_RECORD;
  o.w[0] = (uint64_t)event_slot;
  o.w[1] = reinterpret_cast<uintptr_t>(stream);
  p->ops.push_back(o);
  return T3D_OK;
}

extern "C" int t3d_plan_end_segment(t3d_plan* p) {
  if (!p) return T3D_ERR_ARG;
  p->seg_end.push_back((int)p->ops.size());
  return (int)p->seg_end.size() - 1;        // index of the segment just closed
}

extern "C" int t3d_plan_num_ops(const t3d_plan* p, int kind) {
  if (!p) return T3D_ERR_ARG;
  if (kind < 0) return (int)p->ops.size();
  int n = 0;
  for (const Op& o : p->ops) n += o.kind == kind;
  return n;
}

extern "C" int t3d_plan_time_entry(t3d_plan* p, const char* entry, int on) {
  if (!p || !entry) return T3D_ERR_ARG;
  const int e = find_entry(entry);
  if (e < 0) return T3D_ERR_UNSUPPORTED;
  p->timed[e] = on ? 1 : 0;
  return T3D_OK;
}

extern "C" int t3d_plan_failed_op(const t3d_plan* p, int* rc_out) {
  if (!p) return T3D_ERR_ARG;
  if (rc_out) *rc_out = p->failed_rc;
  return p->failed_op;
}

// Runs segment `segment` (-1: every op).  `events` (optional): hipEvent_t pairs attached, in op order, to the launches of the
// entry points switched on with t3d_plan_time_entry (through t3d_set_launch_events, i.e. kernel-exact for the depthwise
// entry points).  Returns the number of events consumed (>= 0) or a negative T3D_ERR_*.
extern "C" int t3d_plan_run(t3d_plan* p, int segment, const unsigned long long* slots, int nslots, void** events, int nevents) {
  if (!p) return T3D_ERR_ARG;
  int lo = 0, hi = (int)p->ops.size();
  if (segment >= 0) {
    if (segment >= (int)p->seg_end.size()) return T3D_ERR_ARG;
    lo = segment ? p->seg_end[segment - 1] : 0;
    hi = p->seg_end[segment];
  }
  int used = 0;
  uint64_t w[kMaxArgs];
  for (int i = lo; i < hi; ++i) {
    const Op& o = p->ops[i];
    int rc = T3D_OK;
    std::chrono::steady_clock::time_point t_begin;
    if (p->profile) t_begin = std::chrono::steady_clock::now();
    switch (o.kind) {
      case OP_CALL: {
        for (int a = 0; a < o.nargs; ++a) {
          if (o.ak[a] == ARG_WORD) w[a] = o.w[a];
          else if (o.ak[a] == ARG_STRUCT) w[a] = reinterpret_cast<uintptr_t>(&p->arena[o.w[a]]);
          else {
            if ((int)o.w[a] >= nslots || !slots) { rc = T3D_ERR_ARG; break; }
            w[a] = slots[o.w[a]];
          }
        }
        if (rc) break;
        const bool timed = events && p->timed[o.entry] && used + 2 <= nevents;
        if (timed) (void)t3d_set_launch_events(events[used], events[used + 1]);
        if (o.ev) g_t3d_signal = {o.ev, o.sig_stream};
        rc = kEntries[o.entry].run(w);
        if (o.ev) g_t3d_signal = {nullptr, nullptr};
        if (timed) {
          // (an entry point whose path had no kernel-exact launch site: the pair is recorded back to back, reads ~0)
          if (g_t3d_time.start && g_t3d_last_stream) {
            (void)hipEventRecord(g_t3d_time.start, g_t3d_last_stream);
            (void)hipEventRecord(g_t3d_time.stop, g_t3d_last_stream);
          }
          (void)t3d_set_launch_events(nullptr, nullptr);
          used += 2;
        }
        break;
      }
      case OP_FORK: {
        hipStream_t from = reinterpret_cast<hipStream_t>(static_cast<uintptr_t>(o.w[0]));
        hipStream_t to = reinterpret_cast<hipStream_t>(static_cast<uintptr_t>(o.w[1]));
        if (!o.w[2] && hipEventRecord(o.ev, from) != hipSuccess) rc = T3D_ERR_LAUNCH;
        if (rc == T3D_OK && hipStreamWaitEvent(to, o.ev, 0) != hipSuccess) rc = T3D_ERR_LAUNCH;
        break;
      }
      case OP_COPY_D2H: {
        if ((int)o.w[0] >= nslots || !slots) { rc = T3D_ERR_ARG; break; }
        void* dst = reinterpret_cast<void*>(static_cast<uintptr_t>(slots[o.w[0]]));
        if (hipMemcpyAsync(dst, reinterpret_cast<const void*>(static_cast<uintptr_t>(o.w[1])), (size_t)o.w[2], hipMemcpyDeviceToHost,
                           reinterpret_cast<hipStream_t>(static_cast<uintptr_t>(o.w[3]))) != hipSuccess)
          rc = T3D_ERR_LAUNCH;
        break;
      }
      case OP_EVENT_RECORD: {
        if ((int)o.w[0] >= nslots || !slots) { rc = T3D_ERR_ARG; break; }
        hipEvent_t ev = reinterpret_cast<hipEvent_t>(static_cast<uintptr_t>(slots[o.w[0]]));
        if (hipEventRecord(ev, reinterpret_cast<hipStream_t>(static_cast<uintptr_t>(o.w[1]))) != hipSuccess) rc = T3D_ERR_LAUNCH;
        break;
      }
      default: rc = T3D_ERR_ARG;
    }
    if (p->profile) {
      const int slot = o.kind == OP_CALL ? o.entry : kNumEntries + o.kind;
      p->host_ns[slot] += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_begin).count();
      p->host_calls[slot] += 1;
    }
    if (rc != T3D_OK) {
      p->failed_op = i;
      p->failed_rc = rc;
      return rc < 0 ? rc : T3D_ERR_LAUNCH;
    }
  }
  return used;
}
