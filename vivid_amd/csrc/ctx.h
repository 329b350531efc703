// Shared host-side plumbing for libvivid_hip.so: context, record/replay plan, error reporting.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "../../include/vivid_hip.h"

struct vh_op {
    std::function<int(hipStream_t)> launch;
    int tag;          // VH_TAG_*
    double flops;     // algorithmic FLOPs of this launch
    double bytes;     // algorithmic HBM bytes (operands read once + result written once)
};

struct vh_plan {
    std::vector<vh_op> ops;
    hipGraph_t graph = nullptr;          // the same launches captured into a hipGraph (vh_plan_capture_graph)
    hipGraphExec_t exec = nullptr;
};

struct vh_prof_rec {
    hipEvent_t e0, e1;
    int tag;
    double flops, bytes;
};

struct vh_ctx {
    hipStream_t stream = nullptr;
    bool recording = false;
    vh_plan* cur = nullptr;
    bool profiling = false;
    std::vector<vh_prof_rec> prof;          // one per launch while profiling
    std::vector<hipEvent_t> event_pool;
};

int vh_run_op(vh_ctx* ctx, const vh_op& op);

// process-wide scheduling knobs (vh_set_knob)
enum { VH_KNOB_ATTN_XCD = 0, VH_KNOB_DBG_LO = 1, VH_KNOB_DBG_HI = 2, VH_KNOB_ATTN_M16 = 3,
       VH_KNOB_CONV_KORDER = 4,      // -1: by input size / vh_conv_args.korder (default); 0 tap-major, 1 chunk-major (A/B runs)
       VH_KNOB_CONV_STAGGER = 5,     // -1: vh_conv_args.stagger (default); 0 never, 1 always
       VH_KNOB_ATTN_PIPE = 6,        // 0: plain instead of software-pipelined attention kernels
       VH_KNOB_ATTN_NOMAX = 7,       // 0: keep the running maximum although logit_bound allows dropping it
       VH_KNOB_CONV_SLIM2 = 8,       // -1 (default): Cout <= 64 layers take the 256x64 two-per-CU tile; 0: the 512x64 one (A/B runs)
       VH_KNOB_CONV_KORDER_MB = 9,   // input size (MB) above which 3x3 convolutions take the chunk-major K order (default 60)
       VH_KNOB_CONV_KSPLIT = 10,     // > 0: force this many K slices in the glds convolutions (A/B runs of the split-K rule)
       VH_KNOB_CONV_PATCH = 11,      // -1 (default): 3x3 Cout == 64 layers at large M take the patch-resident kernel (conv_patch.hip); 0 never; 1 whenever eligible
       VH_KNOB_CONV_PATCH_DELAY = 12, // conv_x3_patch: start delay (units of 2048 shader cycles) of the second workgroup per CU in a launch's first round
       VH_KNOB_FUSE_CONCAT = 13,     // vh_net walk: decoder concat inputs written by their producers (vh_s8_sink): 0 never, 1 the x half only, 2 both halves
       VH_KNOB_CONV_PATCH96 = 14,    // 1 (default): Cout = 192 layers take the patch kernel as two 96-channel blocks; 0: the 256x192 tile of conv_x3_glds (A/B)
       VH_KNOB_CONV_PATCH_TAIL = 15, // conv_x3_patch with a tail segment: 2 (default) the wave-private register-staged tail, 1 the LDS-DMA staged one (64-channel blocks; A/B)
       VH_KNOB_CONV_TAIL_F32 = 16,   // vh_conv_args.tail_f32 launches: 1 (default) taken where the patch kernel takes tails, 0 never (vh_conv_takes_patch answers 0: callers
                                     // fall back to the raw S8 form of vh_split), 2 also at Cout = 256 / 512 (ties on the S8 tail; A/B)
       VH_KNOB_CONV_SRC_F32 = 17,    // vh_conv_args.src_f32 launches: 2 (default) every block width under the patch kernel's size rule, 1 the 64- / 96-channel blocks only (the look-ahead form),
                                     // 0 never (vh_conv_takes_patch answers 0: the walks fall back to a vh_split pass and the S8 input)
       VH_NUM_KNOBS = 18 };
int vh_knob(int id);
// device buffer for the stamps of diagnostic builds (-DVH_CLOCK), set through the knobs "dbg_lo"/"dbg_hi"; null otherwise
inline unsigned long long* vh_debug_ptr() {
    return reinterpret_cast<unsigned long long*>(((unsigned long long)(unsigned)vh_knob(VH_KNOB_DBG_HI) << 32) | (unsigned)vh_knob(VH_KNOB_DBG_LO));
}

std::string& vh_err();

inline int vh_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    vh_err() = buf;
    return code;
}

#define VH_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) return vh_fail(VH_EINVAL, __VA_ARGS__);    \
    } while (0)

inline int vh_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vh_fail(VH_EHIP, "%s: %s", what, hipGetErrorString(e));
    return VH_OK;
}

// Launch now, or append to the plan being recorded.
template <class F>
inline int vh_dispatch(vh_ctx* ctx, int tag, double flops, double bytes, F&& launch) {
    if (!ctx) return vh_fail(VH_EINVAL, "null context");
    vh_op op{std::forward<F>(launch), tag, flops, bytes};
    if (ctx->recording) {
        ctx->cur->ops.emplace_back(std::move(op));
        return VH_OK;
    }
    return vh_run_op(ctx, op);
}

inline bool vh_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Diagnostic switches (-DVH_CLOCK stamps, -DVH_CONV_ABLATE / -DVH_ATTN_ABLATE timing ablations that compute WRONG results) compile
// only together with -DVH_DIAG, which `make` gives to `make variant` builds alone; a translation unit built with it reports so
// through vh_diag_flags(), the Python binding refuses such a library as the product, and __graft_entry__.build() asserts 0.
#if (defined(VH_CLOCK) || defined(VH_CONV_ABLATE) || defined(VH_ATTN_ABLATE)) && !defined(VH_DIAG)
#error "VH_CLOCK / VH_CONV_ABLATE / VH_ATTN_ABLATE are diagnostic switches: build them with `make variant ... DEFS='-DVH_DIAG ...'`, never into libvivid_hip.so"
#endif
#ifdef VH_DIAG
#define VH_DIAG_FLAG 1
#else
#define VH_DIAG_FLAG 0
#endif
