// Context, record/replay plan and error reporting of libvivid_hip.so (see include/vivid_hip.h).
#include "ctx.h"

std::string& vh_err() {
    static thread_local std::string e;
    return e;
}

extern "C" int vh_abi_version(void) { return 1; }

extern "C" const char* vh_last_error(void) { return vh_err().c_str(); }

extern "C" int vh_ctx_create(void* stream, vh_ctx** out) {
    if (!out) return vh_fail(VH_EINVAL, "vh_ctx_create: null out");
    vh_ctx* c = new vh_ctx();
    c->stream = static_cast<hipStream_t>(stream);
    *out = c;
    return VH_OK;
}

extern "C" int vh_ctx_destroy(vh_ctx* ctx) {
    if (!ctx) return VH_OK;
    delete ctx->cur;
    delete ctx;
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

extern "C" int vh_plan_run(vh_ctx* ctx, const vh_plan* plan) {
    if (!ctx || !plan) return vh_fail(VH_EINVAL, "vh_plan_run: null argument");
    if (ctx->recording) return vh_fail(VH_ESTATE, "vh_plan_run: context is recording");
    for (const auto& op : plan->ops) {
        const int rc = op(ctx->stream);
        if (rc != VH_OK) return rc;
    }
    return VH_OK;
}

extern "C" int vh_plan_num_ops(const vh_plan* plan) { return plan ? (int)plan->ops.size() : 0; }

extern "C" int vh_plan_destroy(vh_plan* plan) {
    delete plan;
    return VH_OK;
}
