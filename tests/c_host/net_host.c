/* A C host of libvivid_hip.so: no Python, no C++ - what INTEGRATION.md section 3 describes, as a program.
 *
 *   net_host <blob> <out.bin>
 *
 * <blob> (written by tests/test_hip_c_host.py) holds a vh_net_config, the network's parameters under the reference's state_dict
 * keys (training/models.py NVPrecond), one set of inputs and a mode word; the program creates the network, binds every parameter the
 * library asks for BY NAME, prepares, records and runs one evaluation (or the split evaluation vh_net_encode + vh_net_run_bound), and
 * writes D_x.  The test compares it with the reference's golden D_x and with vivid_amd.NVPrecond.
 *
 * Build: cc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include net_host.c -L vivid_amd -lvivid_hip -L /opt/rocm/lib -lamdhip64
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "vivid_hip.h"

#define CHECK_VH(call) do { int rc_ = (call); if (rc_ != VH_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, vh_last_error()); return 2; } } while (0)
#define CHECK_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 3; } } while (0)

typedef struct { char name[160]; long long numel; float* host; } tensor;

static int read_exact(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }

static int read_tensor(FILE* f, tensor* t) {
    int len;
    if (read_exact(f, &len, 4) || len < 0 || len >= (int)sizeof t->name) return -1;
    if (read_exact(f, t->name, (size_t)len)) return -1;
    t->name[len] = 0;
    if (read_exact(f, &t->numel, 8) || t->numel < 0) return -1;
    t->host = NULL;
    if (t->numel) {
        t->host = (float*)malloc((size_t)t->numel * 4);
        if (!t->host || read_exact(f, t->host, (size_t)t->numel * 4)) return -1;
    }
    return 0;
}

static float* to_device(const tensor* t) {
    float* d = NULL;
    if (!t->numel) return NULL;
    if (hipMalloc((void**)&d, (size_t)t->numel * 4) != hipSuccess) return NULL;
    if (hipMemcpy(d, t->host, (size_t)t->numel * 4, hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return d;
}

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s blob out.bin\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int magic, cfg_bytes, batch, mode, nparams;
    vh_net_config cfg;
    if (read_exact(f, &magic, 4) || magic != 0x56484E54 || read_exact(f, &cfg_bytes, 4)) { fprintf(stderr, "bad blob\n"); return 1; }
    if (cfg_bytes != (int)sizeof cfg) { fprintf(stderr, "vh_net_config is %d bytes in the blob, %d in vivid_hip.h\n", cfg_bytes, (int)sizeof cfg); return 1; }
    if (read_exact(f, &cfg, sizeof cfg) || read_exact(f, &batch, 4) || read_exact(f, &mode, 4) || read_exact(f, &nparams, 4)) { fprintf(stderr, "bad blob\n"); return 1; }
    tensor* params = (tensor*)calloc((size_t)nparams, sizeof(tensor));
    for (int i = 0; i < nparams; ++i) if (read_tensor(f, &params[i])) { fprintf(stderr, "bad parameter %d\n", i); return 1; }
    tensor in[5];      /* src, x, sigma, geometry, cond (numel 0 = absent) */
    for (int i = 0; i < 5; ++i) if (read_tensor(f, &in[i])) { fprintf(stderr, "bad input %d\n", i); return 1; }
    fclose(f);

    if (vh_abi_version() != VH_ABI_VERSION) { fprintf(stderr, "library ABI %d, header %d\n", vh_abi_version(), VH_ABI_VERSION); return 1; }
    vh_ctx* ctx; vh_net* net;
    CHECK_VH(vh_ctx_create(NULL, &ctx));
    CHECK_VH(vh_net_create(ctx, &cfg, &net));
    const int want = vh_net_num_params(net);
    for (int i = 0; i < want; ++i) {
        const char* name; int ndim, shape[4];
        CHECK_VH(vh_net_param_info(net, i, &name, &ndim, shape));
        long long numel = 1;
        for (int k = 0; k < ndim; ++k) numel *= shape[k];
        int found = -1;
        for (int j = 0; j < nparams && found < 0; ++j) if (!strcmp(params[j].name, name)) found = j;
        if (found < 0 || params[found].numel != numel) { fprintf(stderr, "parameter %s: %s\n", name, found < 0 ? "not in the blob" : "wrong size"); return 1; }
        float* d = to_device(&params[found]);
        if (!d) { fprintf(stderr, "device copy of %s failed\n", name); return 3; }
        CHECK_VH(vh_net_bind_param(net, name, d));
    }
    void* prepared; size_t pb = vh_net_prepared_bytes(net);
    CHECK_HIP(hipMalloc(&prepared, pb));
    CHECK_VH(vh_net_prepare(net, prepared, pb));
    float* dev[5];
    for (int i = 0; i < 5; ++i) { dev[i] = to_device(&in[i]); if (in[i].numel && !dev[i]) return 3; }
    const int R = cfg.img_resolution;
    const size_t out_n = (size_t)batch * 3 * R * R;
    float* out; CHECK_HIP(hipMalloc((void**)&out, out_n * 4));
    if (mode == 0) {                       /* one whole evaluation */
        size_t wb = vh_net_workspace_bytes(net, batch);
        if (!wb) { fprintf(stderr, "vh_net_workspace_bytes: %s\n", vh_last_error()); return 2; }
        void* ws; CHECK_HIP(hipMalloc(&ws, wb));
        CHECK_VH(vh_net_record(net, batch, ws, wb));
        CHECK_VH(vh_net_run(net, batch, dev[0], dev[1], dev[2], dev[3], dev[4], /*cond_noise=*/NULL, out));
    } else {                               /* the sampler's split evaluation: encoder into feature slot 1, UNet on that slot in place */
        size_t fb = vh_net_workspace_bytes_mode(net, VH_NET_FEATURES, batch), bb = vh_net_workspace_bytes_mode(net, VH_NET_BOUND, batch);
        if (!fb || !bb) { fprintf(stderr, "vh_net_workspace_bytes_mode: %s\n", vh_last_error()); return 2; }
        void *wf, *wbnd; CHECK_HIP(hipMalloc(&wf, fb)); CHECK_HIP(hipMalloc(&wbnd, bb));
        CHECK_VH(vh_net_record_mode(net, VH_NET_FEATURES, 1, batch, wf, fb));
        CHECK_VH(vh_net_record_mode(net, VH_NET_BOUND, 1, batch, wbnd, bb));
        CHECK_VH(vh_net_encode(net, 1, batch, dev[0], dev[2], dev[3]));
        CHECK_VH(vh_net_run_bound(net, 1, batch, dev[0], dev[1], dev[2], dev[3], dev[4], /*cond_noise=*/NULL, out));
    }
    CHECK_HIP(hipDeviceSynchronize());
    float* host_out = (float*)malloc(out_n * 4);
    CHECK_HIP(hipMemcpy(host_out, out, out_n * 4, hipMemcpyDeviceToHost));
    FILE* g = fopen(argv[2], "wb");
    if (!g || fwrite(host_out, 4, out_n, g) != out_n) { perror(argv[2]); return 1; }
    fclose(g);
    CHECK_VH(vh_net_destroy(net));
    CHECK_VH(vh_ctx_destroy(ctx));
    printf("net_host: %d parameters bound, batch %d, mode %d, D_x %zu floats\n", want, batch, mode, out_n);
    return 0;
}
