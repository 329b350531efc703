// Context, record/replay plan and error reporting of libvivid_hip.so (see include/vivid_hip.h).
#include "ctx.h"

std::string& vh_err() {
    static thread_local std::string e;
    return e;
}

static int g_knobs[VH_NUM_KNOBS] = {1, 0, 0, 1, -1, -1, 1, 1, -1, 60, 0, -1, 0, 0, 1, 2, 1, 2};
static const char* const g_knob_names[VH_NUM_KNOBS] = {"attn_xcd", "dbg_lo", "dbg_hi", "attn_m16", "conv_korder", "conv_stagger", "attn_pipe", "attn_nomax", "conv_slim2", "conv_korder_mb", "conv_ksplit", "conv_patch", "conv_patch_delay", "fuse_concat", "conv_patch96", "conv_patch_tail", "conv_tail_f32", "conv_src_f32"};

int vh_knob(int id) { return (id >= 0 && id < VH_NUM_KNOBS) ? g_knobs[id] : 0; }

extern "C" int vh_set_knob(const char* name, int value) {
    if (!name) return vh_fail(VH_EINVAL, "vh_set_knob: null name");
    for (int i = 0; i < VH_NUM_KNOBS; ++i)
        if (std::string(name) == g_knob_names[i]) { g_knobs[i] = value; return VH_OK; }
    return vh_fail(VH_EINVAL, "vh_set_knob: unknown knob '%s'", name);
}

extern "C" int vh_abi_version(void) { return VH_ABI_VERSION; }

// bit i set = translation unit i was compiled with -DVH_DIAG (stamps / timing ablations): not a product library
int vh_diag_conv3();
int vh_diag_conv1();
int vh_diag_attn();
int vh_diag_conv_patch();
extern "C" int vh_diag_flags(void) { return VH_DIAG_FLAG | (vh_diag_conv3() << 1) | (vh_diag_conv1() << 2) | (vh_diag_attn() << 3) | (vh_diag_conv_patch() << 4); }

extern "C" const char* vh_last_error(void) { return vh_err().c_str(); }

extern "C" int vh_ctx_create(void* stream, vh_ctx** out) {
    if (!out) return vh_fail(VH_EINVAL, "vh_ctx_create: null out");
    vh_ctx* c = new vh_ctx();
    c->stream = static_cast<hipStream_t>(stream);
    *out = c;
    return VH_OK;
}

static hipEvent_t vh_get_event(vh_ctx* ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

// Launch one op; when profiling, bracket it with events on the launch stream.
int vh_run_op(vh_ctx* ctx, const vh_op& op) {
    if (!ctx->profiling) return op.launch(ctx->stream);
    vh_prof_rec r{vh_get_event(ctx), vh_get_event(ctx), op.tag, op.flops, op.bytes};
    if (!r.e0 || !r.e1) return vh_fail(VH_EHIP, "profiling: hipEventCreate failed");
    (void)hipEventRecord(r.e0, ctx->stream);
    const int rc = op.launch(ctx->stream);
    (void)hipEventRecord(r.e1, ctx->stream);
    ctx->prof.push_back(r);
    return rc;
}

extern "C" int vh_ctx_destroy(vh_ctx* ctx) {
    if (!ctx) return VH_OK;
    for (auto& r : ctx->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    delete ctx->cur;
    delete ctx;
    return VH_OK;
}

extern "C" int vh_profile_enable(vh_ctx* ctx, int on) {
    if (!ctx) return vh_fail(VH_EINVAL, "vh_profile_enable: null context");
    if (ctx->recording) return vh_fail(VH_ESTATE, "vh_profile_enable: context is recording");
    ctx->profiling = on != 0;
    return VH_OK;
}

extern "C" int vh_profile_read(vh_ctx* ctx, int ntags, double* ms, double* flops, double* bytes, long long* launches) {
    if (!ctx || !ms || !flops || !bytes || !launches) return vh_fail(VH_EINVAL, "vh_profile_read: null argument");
    for (int i = 0; i < ntags; ++i) { ms[i] = 0; flops[i] = 0; bytes[i] = 0; launches[i] = 0; }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return vh_fail(VH_EHIP, "vh_profile_read: %s", hipGetErrorString(e));
    for (auto& r : ctx->prof) {
        float t = 0.f;
        e = hipEventElapsedTime(&t, r.e0, r.e1);
        if (e != hipSuccess) return vh_fail(VH_EHIP, "vh_profile_read: %s", hipGetErrorString(e));
        if (r.tag >= 0 && r.tag < ntags) {
            ms[r.tag] += t; flops[r.tag] += r.flops; bytes[r.tag] += r.bytes; launches[r.tag] += 1;
        }
        ctx->event_pool.push_back(r.e0);
        ctx->event_pool.push_back(r.e1);
    }
    ctx->prof.clear();
    return VH_OK;
}

extern "C" int vh_profile_read_list(vh_ctx* ctx, int max_n, int* tags, double* ms, double* flops, double* bytes, int* n_out) {
    if (!ctx || !tags || !ms || !flops || !bytes || !n_out) return vh_fail(VH_EINVAL, "vh_profile_read_list: null argument");
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return vh_fail(VH_EHIP, "vh_profile_read_list: %s", hipGetErrorString(e));
    int n = 0;
    for (auto& r : ctx->prof) {
        if (n < max_n) {
            float t = 0.f;
            (void)hipEventElapsedTime(&t, r.e0, r.e1);
            tags[n] = r.tag; ms[n] = t; flops[n] = r.flops; bytes[n] = r.bytes;
            ++n;
        }
        ctx->event_pool.push_back(r.e0);
        ctx->event_pool.push_back(r.e1);
    }
    ctx->prof.clear();
    *n_out = n;
    return VH_OK;
}

extern "C" int vh_ctx_set_stream(vh_ctx* ctx, void* stream) {
    if (!ctx) return vh_fail(VH_EINVAL, "vh_ctx_set_stream: null context");
    ctx->stream = static_cast<hipStream_t>(stream);
    return VH_OK;
}

extern "C" int vh_plan_begin(vh_ctx* ctx) {
    if (!ctx) return vh_fail(VH_EINVAL, "vh_plan_begin: null context");
    if (ctx->recording) return vh_fail(VH_ESTATE, "vh_plan_begin: already recording");
    ctx->cur = new vh_plan();
    ctx->recording = true;
    return VH_OK;
}

extern "C" int vh_plan_end(vh_ctx* ctx, vh_plan** out) {
    if (!ctx || !out) return vh_fail(VH_EINVAL, "vh_plan_end: null argument");
    if (!ctx->recording) return vh_fail(VH_ESTATE, "vh_plan_end: not recording");
    ctx->recording = false;
    *out = ctx->cur;
    ctx->cur = nullptr;
    return VH_OK;
}

extern "C" int vh_plan_abort(vh_ctx* ctx) {
    if (!ctx) return vh_fail(VH_EINVAL, "vh_plan_abort: null context");
    delete ctx->cur;
    ctx->cur = nullptr;
    ctx->recording = false;
    return VH_OK;
}

// Capture the plan's launches into a hipGraph (on a private capture stream: the legacy default stream cannot be
// captured) and instantiate it; vh_plan_run then replays the whole evaluation with one hipGraphLaunch.
extern "C" int vh_plan_capture_graph(vh_ctx* ctx, vh_plan* plan) {
    if (!ctx || !plan) return vh_fail(VH_EINVAL, "vh_plan_capture_graph: null argument");
    if (ctx->recording) return vh_fail(VH_ESTATE, "vh_plan_capture_graph: context is recording");
    if (plan->exec) return VH_OK;
    hipStream_t cs = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    if (e != hipSuccess) return vh_fail(VH_EHIP, "vh_plan_capture_graph: %s", hipGetErrorString(e));
    e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { (void)hipStreamDestroy(cs); return vh_fail(VH_EHIP, "vh_plan_capture_graph: begin: %s", hipGetErrorString(e)); }
    int rc = VH_OK;
    for (const auto& op : plan->ops) {
        rc = op.launch(cs);
        if (rc != VH_OK) break;
    }
    hipGraph_t g = nullptr;
    e = hipStreamEndCapture(cs, &g);
    (void)hipStreamDestroy(cs);
    if (rc != VH_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess || !g) return vh_fail(VH_EHIP, "vh_plan_capture_graph: end: %s", hipGetErrorString(e));
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(g); return vh_fail(VH_EHIP, "vh_plan_capture_graph: instantiate: %s", hipGetErrorString(e)); }
    plan->graph = g;
    plan->exec = ex;
    return VH_OK;
}

extern "C" int vh_plan_run(vh_ctx* ctx, const vh_plan* plan) {
    if (!ctx || !plan) return vh_fail(VH_EINVAL, "vh_plan_run: null argument");
    if (ctx->recording) return vh_fail(VH_ESTATE, "vh_plan_run: context is recording");
    if (plan->exec && !ctx->profiling) {
        const hipError_t e = hipGraphLaunch(plan->exec, ctx->stream);
        if (e != hipSuccess) return vh_fail(VH_EHIP, "vh_plan_run: hipGraphLaunch: %s", hipGetErrorString(e));
        return VH_OK;
    }
    for (const auto& op : plan->ops) {
        const int rc = vh_run_op(ctx, op);
        if (rc != VH_OK) return rc;
    }
    return VH_OK;
}

extern "C" int vh_plan_num_ops(const vh_plan* plan) { return plan ? (int)plan->ops.size() : 0; }

extern "C" int vh_plan_destroy(vh_plan* plan) {
    if (plan) {
        if (plan->exec) (void)hipGraphExecDestroy(plan->exec);
        if (plan->graph) (void)hipGraphDestroy(plan->graph);
    }
    delete plan;
    return VH_OK;
}
