// vh_net: one whole NVPrecond evaluation behind the C ABI (include/vivid_hip.h, "whole-network evaluation").
//
// reference: NVPrecond._forward_dualsource training/models.py:628-689 (and the single-source forward :691-749) ->
// UNetEncoder.forward :536-570 -> XAttnUNet.forward :483-518 -> Block.forward :165-206 / XAttnBlock.forward :251-315, on the
// architecture tables UNet.__init__ builds (:322-384, :413-480, :524-534, :576-582).
//
// This is the production walk of vivid_amd/engine.py restated in C++ for hosts without Python: bf16x3 arithmetic on the
// direct-to-LDS convolution kernels with the fused q/k/v epilogue, any even resample filter ([1,1] fused), modes "full" (encoder + UNet)
// and "uncond" (UNet with the zero features in closed form).  It emits the SAME op sequence with the same arguments as the Python
// engine (tests/test_hip_net_c.py compares the two outputs bit for bit), records it once per batch size into a vh_plan over a
// caller-supplied workspace, and replays it; nothing is allocated inside a call.
#include "ctx.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <tuple>

namespace {

constexpr double LOG2E = 1.4426950408889634;
inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ---------------------------------------------------------------- architecture table (training/models.py:322-384, 413-480, 524-534)
struct Block {
    std::string name;
    bool conv = false;            // bare first MPConv 3x3
    int cin = 0, cout = 0, res = 0;
    bool dec = false;
    int resample = 0;             // 0 keep, 1 up, 2 down
    int heads = 0;
    bool xattn = false, takes_skip = false, live = true;
};
struct Spec {
    int in_channels = 0, label_dim = 0, cnoise = 0, cemb = 0, out_channels = 0, last_ch = 0;
    std::vector<Block> enc, dec;
};

Spec make_spec(const vh_net_config& c, bool encoder) {
    Spec s;
    const int R = c.img_resolution;
    const int warp = c.logvar_channels * (c.warp_depth_coor ? 1 : 0);
    int img_ch, cph;
    if (encoder) { img_ch = c.img_channels + (c.depth_input ? 1 : 0) + warp; s.label_dim = c.source_label_dim; cph = 64; s.in_channels = img_ch + 1; }
    else {
        img_ch = c.img_channels + warp; s.label_dim = c.target_label_dim; cph = c.super_res ? 32 : 64; s.in_channels = img_ch + 1;
        if (c.super_res) s.in_channels = 2 * (s.in_channels - 1) + 1;
    }
    std::vector<int> cblock;
    for (int i = 0; i < c.num_levels; ++i) cblock.push_back(c.model_channels * c.channel_mult[i]);
    s.cnoise = c.channel_mult_noise > 0 ? c.model_channels * c.channel_mult_noise : cblock[0];
    s.cemb = c.channel_mult_emb > 0 ? c.model_channels * c.channel_mult_emb : *std::max_element(cblock.begin(), cblock.end());
    const bool xattn = !encoder;
    auto is_attn_res = [&](int res) { for (int i = 0; i < c.num_attn_resolutions; ++i) if (c.attn_resolutions[i] == res) return true; return false; };
    auto rname = [](int res, const char* what, int idx = -1) {
        std::string n = std::to_string(res) + "x" + std::to_string(res) + "_" + what;
        if (idx >= 0) n += std::to_string(idx);
        return n;
    };
    int cout = s.in_channels;
    const int L = (int)cblock.size();
    for (int level = 0; level < L; ++level) {
        const int ch = cblock[level], res = R >> level;
        if (level == 0) { Block b; b.name = rname(res, "conv"); b.conv = true; b.cin = cout; b.cout = ch; b.res = res; s.enc.push_back(b); cout = ch; }
        else { Block b; b.name = rname(res, "down"); b.cin = b.cout = cout; b.res = res; b.resample = 2; s.enc.push_back(b); }
        for (int idx = 0; idx < c.num_blocks; ++idx) {
            Block b; b.name = rname(res, "block", idx); b.cin = cout; b.cout = cout = ch; b.res = res;
            const bool attn = is_attn_res(res) || (c.extra_attn >= 0 && c.extra_attn == idx && level != 0);
            b.heads = attn ? cout / cph : 0; b.xattn = xattn && attn;
            s.enc.push_back(b);
        }
    }
    std::vector<int> skips;
    for (auto& b : s.enc) skips.push_back(b.cout);
    for (int level = L - 1; level >= 0; --level) {
        const int ch = cblock[level], res = R >> level;
        if (level == L - 1) {
            Block b0; b0.name = rname(res, "in", 0); b0.cin = b0.cout = cout; b0.res = res; b0.dec = true; b0.heads = cout / cph; b0.xattn = xattn; s.dec.push_back(b0);
            Block b1; b1.name = rname(res, "in", 1); b1.cin = b1.cout = cout; b1.res = res; b1.dec = true; s.dec.push_back(b1);
        } else { Block b; b.name = rname(res, "up"); b.cin = b.cout = cout; b.res = res; b.dec = true; b.resample = 1; s.dec.push_back(b); }
        for (int idx = 0; idx <= c.num_blocks; ++idx) {
            const int sk = skips.back(); skips.pop_back();
            Block b; b.name = rname(res, "block", idx); b.cin = cout + sk; b.cout = cout = ch; b.res = res; b.dec = true; b.takes_skip = true;
            const bool attn = is_attn_res(res) || (c.extra_attn >= 0 && c.extra_attn == c.num_blocks - idx && level != 0);
            b.heads = attn ? cout / cph : 0; b.xattn = xattn && attn;
            s.dec.push_back(b);
        }
    }
    s.last_ch = cout;
    if (!encoder) s.out_channels = 3;
    else for (int i = (int)s.dec.size() - 1; i >= 0 && s.dec[i].heads == 0; --i) s.dec[i].live = false;      // :530-534
    return s;
}

// ---------------------------------------------------------------- parameters (the reference's state_dict keys)
struct Param { std::string name; int ndim; int shape[4]; const float* ptr = nullptr; };

void spec_params(const Spec& sp, const std::string& prefix, bool has_out_gain, std::vector<Param>& out) {
    auto add = [&](const std::string& n, std::initializer_list<int> shp) {
        Param p; p.name = n; p.ndim = (int)shp.size(); int i = 0; for (int v : shp) p.shape[i++] = v; for (; i < 4; ++i) p.shape[i] = 1; out.push_back(p);
    };
    if (has_out_gain) add(prefix + "out_gain", {});
    add(prefix + "emb_fourier.freqs", {sp.cnoise});
    add(prefix + "emb_fourier.phases", {sp.cnoise});
    add(prefix + "emb_noise.weight", {sp.cemb, sp.cnoise});
    if (sp.label_dim) add(prefix + "emb_label.weight", {sp.cemb, sp.label_dim});
    for (int g = 0; g < 2; ++g)
        for (const Block& b : (g ? sp.dec : sp.enc)) {
            if (!b.live) continue;
            const std::string p = prefix + (g ? "dec." : "enc.") + b.name + ".";
            if (b.conv) { add(p + "weight", {b.cout, b.cin, 3, 3}); continue; }
            add(p + "emb_gain", {});
            add(p + "conv_res0.weight", {b.cout, b.dec ? b.cin : b.cout, 3, 3});
            add(p + "emb_linear.weight", {b.cout, sp.cemb});
            add(p + "conv_res1.weight", {b.cout, b.cout, 3, 3});
            if (b.cin != b.cout) add(p + "conv_skip.weight", {b.cout, b.cin, 1, 1});
            if (b.heads) {
                add(p + "attn_qkv.weight", {3 * b.cout, b.cout, 1, 1});
                if (b.xattn) add(p + "x_attn_kv.weight", {2 * b.cout, b.cout, 1, 1});
                add(p + "attn_proj.weight", {b.cout, b.cout, 1, 1});
            }
        }
    if (sp.out_channels) add(prefix + "out_conv.weight", {sp.out_channels, sp.last_ch, 3, 3});
}

// ---------------------------------------------------------------- workspace arena (first fit, 64-float granules)
struct Arena {
    std::vector<std::pair<long long, long long>> free_{{0, 1LL << 62}};
    long long peak = 0;
    long long alloc(long long n) {
        n = (std::max<long long>(n, 1) + 63) / 64 * 64;
        for (size_t i = 0; i < free_.size(); ++i)
            if (free_[i].second >= n) {
                const long long off = free_[i].first;
                if (free_[i].second == n) free_.erase(free_.begin() + i); else free_[i] = {off + n, free_[i].second - n};
                peak = std::max(peak, off + n);
                return off;
            }
        return -1;
    }
    void release(long long off, long long n) {
        n = (std::max<long long>(n, 1) + 63) / 64 * 64;
        free_.push_back({off, n});
        std::sort(free_.begin(), free_.end());
        std::vector<std::pair<long long, long long>> m{free_[0]};
        for (size_t i = 1; i < free_.size(); ++i)
            if (m.back().first + m.back().second == free_[i].first) m.back().second += free_[i].second; else m.push_back(free_[i]);
        free_ = m;
    }
};
struct Buf { long long off = -1; long long n = 0; int c = 0; float* abs = nullptr; bool ok() const { return off >= 0; } };   // n floats; c = channels (last dim); abs: lives in ANOTHER program's workspace

struct Weight { float* wt = nullptr; size_t off = 0; int cin_pad = 0, k_pad = 0, cout = 0, taps = 0, nj = 0, D = 0, fused_c1 = 0; const float* gain = nullptr; bool has_gain = false; };

struct FeatBuf { Buf f32, s8, k, v; };      // k, v: this block's cross-attention keys / values, when the VH_NET_FEATURES program computed them with the features
// one decoder block's concat input mp_silu(mp_cat(x, skip)) (training/models.py:78-84, :174) whose halves are written by their producers (vh_s8_sink)
struct CatState { int rows = 0, R = 0, Na = 0, Nb = 0; float sc0 = 1.f, sc1 = 1.f; bool raw = false, ok = false, x_done = false, skip_done = false; Buf cs, craw; };
struct Program {
    int B = 0, mode = 0;
    vh_plan* plan = nullptr;
    float* base = nullptr;
    float* feat_base = nullptr;       // VH_NET_BOUND: workspace base of the VH_NET_FEATURES program whose buffers this plan reads in place
    Buf sigma, geometry, src, x, cond, D;
    std::vector<FeatBuf> feats;       // VH_NET_FEATURES: the encoder's feature list, (fp32, S8) pairs inside this program's workspace
    std::vector<Buf> feats_in;        // VH_NET_INJECT: NHWC input buffers of the caller's feature list
    long long peak_floats = 0;
};

}  // namespace

struct vh_net {
    vh_ctx* ctx = nullptr;
    vh_net_config cfg{};
    bool has_enc = false;
    Spec enc, unet;
    std::vector<Param> params;
    std::map<std::string, int> pindex;
    // prepared weights (inside the caller's buffer)
    std::map<std::string, Weight> W;
    struct EmbW { float* wt = nullptr; size_t off = 0; std::map<std::string, int> cols; int total = 0; } embE, embU;
    std::vector<std::string> prep_order;          // keys of W in preparation order
    size_t prepared_floats = 0;
    float* zeros = nullptr; float* scratch_enc = nullptr; float* scratch_unet = nullptr; float* scratch = nullptr;
    bool prepared = false;
    bool fp32 = false;            // vh_net_config.fp32: exact-fp32 arithmetic (engine.Engine(precision="fp32")) instead of bf16x3
    bool std_filter = true; int ntaps = 2; float taps[8] = {0.5f, 0.5f};          // Block.resample_filter (:139), normalised (f / sum f, :53)
    std::map<std::tuple<int, int, int>, std::unique_ptr<Program>> programs;      // (mode, slot, batch)
    // walk state
    Arena* A = nullptr; float* base = nullptr; bool emit = false; int rc = VH_OK;
    std::map<int, CatState> cat;          // by decoder block index, for the network being walked
    // vh_edm_sampler: streams / events of its scheduling (created on first use, destroyed with the net)
    hipStream_t side_stream = nullptr, gside_stream = nullptr;
    hipEvent_t ev_main = nullptr, ev_enc[2] = {nullptr, nullptr}, ev_g = nullptr;
};

namespace {

constexpr size_t ZEROS_FLOATS = 16384, SCRATCH_FLOATS = size_t(16) << 20;

const float* P(vh_net* n, const std::string& key) {
    auto it = n->pindex.find(key);
    return it == n->pindex.end() ? nullptr : n->params[it->second].ptr;
}

// --------------------------------------------------------------- emission helpers (mirror engine.Engine._alloc/_free/_conv/_split)
Buf alloc(vh_net* n, long long rows, long long h, long long w, long long c) {
    Buf b; b.n = rows * h * w * c; b.c = (int)c; b.off = n->A->alloc(b.n);
    if (b.off < 0 && n->rc == VH_OK) n->rc = vh_fail(VH_EINVAL, "vh_net: arena exhausted");
    return b;
}
void release(vh_net* n, Buf& b) { if (b.ok()) { n->A->release(b.off, b.n); b.off = -1; } }
float* ptr(vh_net* n, const Buf& b) { return !b.ok() ? nullptr : b.abs ? b.abs : n->base + b.off; }
template <class F, class A> void call(vh_net* n, F fn, const A& a) { if (n->emit && n->rc == VH_OK) n->rc = fn(n->ctx, &a); }

struct ConvOpt {
    int up = 0, epi = VH_EPI_STORE; const float* cvec = nullptr; int cvec_ld = 0; const Buf* res = nullptr; int res_up = 0;
    float ta = 0.f, tb = 0.f, clip = 0.f; Buf* out = nullptr; bool s8_only = false, also_s8 = false; const vh_qkv_epilogue* qkv = nullptr;
    const Buf* src1 = nullptr;        // S8 second source: the 1-tap tail segment of a fused conv_res1 + conv_skip
    bool src_f32 = false; const Buf* f1 = nullptr; float fsc0 = 1.f, fsc1 = 1.f; int pro = VH_PRO_NONE;   // `src` (and f1) are the fp32 tensors of the input concat (vh_conv_args.src_f32)
    const Buf* tail0 = nullptr; const Buf* tail1 = nullptr; float tsc0 = 1.f, tsc1 = 0.f;     // ... or the fp32 tensors x, skip themselves with their mp_cat weights (vh_conv_args.tail_f32)
    const Buf* res_scale = nullptr;   // per-pixel factor of the residual
    int sink_j = -1, sink_half = 0;   // this result is half 0 (x) / 1 (skip) of decoder block sink_j's concat input: written as S8 by this launch if it takes the patch kernel
    bool fp32_optional = false;       // ... and nothing else reads its fp32 form
};
Buf ghost(int c) { Buf b; b.c = c; return b; }       // a tensor that was never materialised in fp32 (off < 0): only its channel count is known
// bf16x3 glds convolution of an S8 source; returns (fp32 out, S8 out) - either may be empty
std::pair<Buf, Buf> conv(vh_net* n, const Buf& src, const Weight& W, int rows, int h, int w, ConvOpt o) {
    Buf out, out8;
    if (o.qkv) o.epi = VH_EPI_QKV;
    vh_s8_sink sinks[2] = {};
    bool sunk = false;
    if (o.sink_j >= 0 && !o.qkv) {
        vh_conv_args q{};
        q.src0 = reinterpret_cast<const float*>(16); q.src1 = (o.src1 || o.tail0) ? reinterpret_cast<const float*>(16) : nullptr; q.c0 = src.c;
        q.c1 = o.tail0 ? o.tail0->c : o.src1 ? o.src1->c : 0; q.tail_f32 = o.tail0 ? 1 : 0; q.src2 = o.tail1 ? reinterpret_cast<const float*>(16) : nullptr; q.c2 = o.tail1 ? o.tail1->c : 0;
        q.rows = rows; q.h = h; q.w = w; q.up = o.up; q.taps = W.taps; q.cout = W.cout; q.prec = VH_PREC_BF16X3; q.kernel = VH_CONV_GLDS256; q.epi = o.epi; q.res_up = o.res_up;
        if (vh_conv_takes_patch(&q) == 1) {
            CatState& st = n->cat.at(o.sink_j);
            const int Ct = st.Na + st.Nb;
            if (!st.cs.ok()) { st.cs = alloc(n, st.rows, st.R, st.R, Ct); if (st.raw) st.craw = alloc(n, st.rows, st.R, st.R, Ct); }
            const int off = o.sink_half == 0 ? 0 : st.Na;
            const float scale = o.sink_half == 0 ? st.sc0 : st.sc1;
            (o.sink_half == 0 ? st.x_done : st.skip_done) = true;
            sinks[0] = vh_s8_sink{ptr(n, st.cs), Ct, off, scale, 1};
            if (st.raw) sinks[1] = vh_s8_sink{ptr(n, st.craw), Ct, off, scale, 0};
            sunk = true;
        }
    }
    const bool skip_fp32 = sunk && o.fp32_optional && !o.also_s8 && !o.s8_only;
    if ((o.s8_only || o.also_s8) && !o.qkv) out8 = alloc(n, rows, h, w, W.cout);
    if (!o.s8_only && !o.qkv && !skip_fp32) { if (o.out) out = *o.out; else out = alloc(n, rows, h, w, W.cout); }
    vh_conv_args a{};
    a.src0 = ptr(n, src); a.src1 = o.src1 ? ptr(n, *o.src1) : nullptr; a.c0 = src.c; a.c1 = o.src1 ? o.src1->c : 0; a.scale0 = 1.f; a.scale1 = 1.f;
    if (o.src_f32) {
        a.src_f32 = 1; a.scale0 = o.fsc0;
        if (o.f1) { a.src1 = ptr(n, *o.f1); a.c1 = o.f1->c; a.scale1 = o.fsc1; }
    }
    if (o.tail0) {
        a.tail_f32 = 1; a.src1 = ptr(n, *o.tail0); a.c1 = o.tail0->c; a.scale1 = o.tsc0;
        if (o.tail1) { a.src2 = ptr(n, *o.tail1); a.c2 = o.tail1->c; a.scale2 = o.tsc1; }
    }
    a.rows = rows; a.h = h; a.w = w; a.up = o.up; a.taps = W.taps; a.pro = o.src_f32 ? o.pro : VH_PRO_NONE;
    a.wt = W.wt; a.cin_pad = W.cin_pad; a.k_pad = W.k_pad; a.zeros = n->zeros; a.zeros_bytes = ZEROS_FLOATS * 4; a.cout = W.cout;
    a.scratch = n->scratch; a.scratch_floats = SCRATCH_FLOATS;
    a.out = ptr(n, out); a.out_s8 = ptr(n, out8); a.out_s8_c = out8.ok() ? W.cout : 0;
    a.prec = VH_PREC_BF16X3; a.kernel = VH_CONV_GLDS256; a.epi = o.epi; a.cvec = o.cvec; a.cvec_ld = o.cvec_ld;
    a.res = o.res ? ptr(n, *o.res) : nullptr; a.res_up = o.res_up; a.res_scale = o.res_scale ? ptr(n, *o.res_scale) : nullptr; a.ta = o.ta; a.tb = o.tb; a.clip = o.clip; a.qkv = o.qkv;
    a.stagger = 0; a.korder = VH_KORDER_AUTO; a.tile = VH_TILE_AUTO;
    a.sink[0] = sinks[0]; a.sink[1] = sinks[1];
    call(n, vh_conv, a);
    if (skip_fp32) out = ghost(W.cout);
    return {out, out8};
}
// fp32 NHWC (1-2 sources, mp_cat weights) -> S8; raw_too: second S8 output without the prologue
std::pair<Buf, Buf> split(vh_net* n, const Buf& s0, float sc0, const Buf* s1, float sc1, long long npix, int rows, int h, int w, int pro, bool raw_too) {
    const int ctot = s0.c + (s1 ? s1->c : 0), cpad = round_up(ctot, 32);
    Buf out = alloc(n, rows, h, w, cpad), raw;
    if (raw_too) raw = alloc(n, rows, h, w, cpad);
    vh_split_args a{};
    a.src0 = ptr(n, s0); a.src1 = s1 ? ptr(n, *s1) : nullptr; a.c0 = s0.c; a.c1 = s1 ? s1->c : 0; a.scale0 = sc0; a.scale1 = sc1;
    a.pro = pro; a.npix = npix; a.c_pad = cpad; a.out = ptr(n, out); a.out_raw = ptr(n, raw);
    call(n, vh_split, a);
    return {out, raw};
}
// vh_split of ONE half of a concat input into its channel range of the S8 concat tensors (the other half came from a sink)
void split_half(vh_net* n, const Buf& src, float scale, const CatState& st, int off, long long npix) {
    vh_split_args a{};
    a.src0 = ptr(n, src); a.src1 = nullptr; a.c0 = src.c; a.c1 = 0; a.scale0 = scale; a.scale1 = 1.f; a.pro = VH_PRO_SILU; a.npix = npix; a.c_pad = src.c;
    a.out = ptr(n, st.cs); a.out_raw = ptr(n, st.craw); a.out_c_total = st.Na + st.Nb; a.out_c_off = off;
    call(n, vh_split, a);
}
void mp_sum_coeffs(double t, float& a, float& b) { const double nn = std::sqrt((1.0 - t) * (1.0 - t) + t * t); a = (float)((1.0 - t) / nn); b = (float)(t / nn); }

using Feat = FeatBuf;

// resample() with a non-default filter (training/models.py:48-61) as its own launch (engine.Engine._resample); [1,1] is fused into
// vh_pixnorm (down) and vh_conv (up) instead
Buf resample(vh_net* n, const Buf& x, int rows, int h, int w, bool up) {
    Buf out = up ? alloc(n, rows, h * 2, w * 2, x.c) : alloc(n, rows, h / 2, w / 2, x.c);
    vh_resample_args a{}; a.in = ptr(n, x); a.out = ptr(n, out); a.rows = rows; a.h = h; a.w = w; a.c = x.c; a.up = up ? 1 : 0; a.ntaps = n->ntaps;
    for (int i = 0; i < n->ntaps; ++i) a.taps[i] = n->taps[i];
    call(n, vh_resample, a);
    return out;
}

// Block.forward :165-206 / XAttnBlock.forward :251-315 (bf16x3 path of engine.Engine._block)
std::pair<Buf, Buf> block(vh_net* n, const std::string& prefix, const Block& b, int rows, const Buf& x, const Buf* skip, const Buf& cvec_all,
                          const vh_net::EmbW& emb, const Feat* feat, int nsrc, float n_zero, bool want_s8, int cat_j = -1, int out_sink_j = -1, int out_sink_half = 0,
                          bool fp32_optional = false, bool s8_final = false) {
    const vh_net_config& cfg = n->cfg;
    const std::string p = prefix + (b.dec ? "dec." : "enc.") + b.name + ".";
    const int R = b.res, C = b.cout, D = b.heads ? C / b.heads : 0;
    const float* cv = ptr(n, cvec_all) ? ptr(n, cvec_all) + emb.cols.at(p) : (const float*)nullptr;
    if (!n->emit) cv = reinterpret_cast<const float*>(16);          // (dry walk: never dereferenced)
    float ta, tb; mp_sum_coeffs(cfg.res_balance, ta, tb);
    const float clip = cfg.clip_act > 0.0 ? (float)cfg.clip_act : 0.f, clip_res = b.heads ? 0.f : clip;
    const bool has_skip_conv = b.cin != b.cout;
    const bool res1_s8 = b.heads > 0, fin_s8 = want_s8 && !b.heads;
    // the UNet's last block: out_conv is its only reader and reads S8 - conv_res1 writes that form alone (engine.Engine._block: fin_only)
    const bool fin_only = s8_final && !b.heads && b.dec;
    if (b.heads) { out_sink_j = -1; fp32_optional = false; }      // (the block's last op is attn_proj, a 1x1 convolution: no sinks there)
    const long long npix = (long long)rows * R * R;
    Buf out, r_s8, out_s8;
    if (!b.dec) {
        Buf xs = alloc(n, rows, R, R, C), xn, res_scale;
        vh_pixnorm_args pa{}; pa.rows = rows; pa.h = R; pa.w = R; pa.c = C; pa.norm = 1; pa.out_s8 = nullptr;
        if (b.resample == 2 && n->std_filter) {
            xn = alloc(n, rows, R, R, C);
            pa.in = ptr(n, x); pa.out = ptr(n, xn); pa.pool = 1; pa.out_s8 = ptr(n, xs);
            call(n, vh_pixnorm, pa);
        } else if (b.resample == 2) {                                  // general FIR filter (:48-59), then the plain pixel norm
            xn = resample(n, x, rows, R * 2, R * 2, false);
            pa.in = ptr(n, xn); pa.out = ptr(n, xn); pa.pool = 0; pa.out_s8 = ptr(n, xs);
            call(n, vh_pixnorm, pa);
        } else if (has_skip_conv) {
            auto xr = split(n, x, 1.f, nullptr, 1.f, npix, rows, R, R, VH_PRO_NONE, false);
            xn = conv(n, xr.first, n->W.at(p + "conv_skip.weight"), rows, R, R, ConvOpt{}).first;
            release(n, xr.first);
            pa.in = ptr(n, xn); pa.out = ptr(n, xn); pa.pool = 0; pa.out_s8 = ptr(n, xs);
            call(n, vh_pixnorm, pa);
        } else {
            // plain block: the normalised tensor is never materialised - conv_res1's residual is x * scale[pixel]
            res_scale = alloc(n, rows, R, R, 1);
            pa.in = ptr(n, x); pa.out = nullptr; pa.pool = 0; pa.out_s8 = ptr(n, xs); pa.scale_out = ptr(n, res_scale);
            call(n, vh_pixnorm, pa);
        }
        ConvOpt o0; o0.epi = VH_EPI_SCALE_SILU; o0.cvec = cv; o0.cvec_ld = emb.total; o0.s8_only = true;
        Buf y = conv(n, xs, n->W.at(p + "conv_res0.weight"), rows, R, R, o0).second;
        release(n, xs);
        ConvOpt o1; o1.epi = VH_EPI_MPSUM; o1.res = xn.ok() ? &xn : &x; o1.res_scale = res_scale.ok() ? &res_scale : nullptr;
        o1.ta = ta; o1.tb = tb; o1.clip = clip_res; o1.also_s8 = res1_s8 || fin_s8; o1.sink_j = out_sink_j; o1.sink_half = out_sink_half;
        auto r = conv(n, y, n->W.at(p + "conv_res1.weight"), rows, R, R, o1);
        release(n, y); release(n, xn); release(n, res_scale);
        out = r.first; r_s8 = r.second;
    } else {
        int up = b.resample == 1 ? 1 : 0;
        Buf xup; const Buf* xp = &x;
        if (up && !n->std_filter) {                                // general FIR filter (:60-61): materialise the upsampled input
            xup = resample(n, x, rows, R / 2, R / 2, true);
            xp = &xup; up = 0;
        }
        const Buf& x = *xp;                                       // (shadows the parameter from here on, like engine._block's `x = xup`)
        float sc0 = 1.f, sc1 = 1.f;
        if (skip) {                                               // mp_cat :78-84
            const double t = cfg.concat_balance, Na = x.c, Nb = skip->c;
            const double Cc = std::sqrt((Na + Nb) / ((1 - t) * (1 - t) + t * t));
            sc0 = (float)(Cc / std::sqrt(Na) * (1 - t)); sc1 = (float)(Cc / std::sqrt(Nb) * t);
        }
        const long long npix_in = up ? npix / 4 : npix;
        const int Rin = up ? R / 2 : R;
        std::pair<Buf, Buf> cs;
        bool tail32 = false, src32 = false;
        CatState* st = (cat_j >= 0 && skip) ? &n->cat.at(cat_j) : nullptr;
        if (st && (st->x_done || st->skip_done)) {
            // at least one half of mp_silu(mp_cat(x, skip)) was written by its producer (vh_s8_sink); vh_split fills in the other, if any
            if (!st->x_done) split_half(n, x, st->sc0, *st, 0, npix_in);
            if (!st->skip_done) split_half(n, *skip, st->sc1, *st, st->Na, npix_in);
            cs = {st->cs, st->craw};
        } else {
            // the fused conv_res1 + conv_skip launch reads x and skip as fp32 where the library's size rule gives it the patch-resident kernel
            // (vh_conv_args.tail_f32): vh_split then writes the mp_silu form only (engine.Engine._tail_f32)
            if (has_skip_conv && !up && x.c % 32 == 0 && (!skip || skip->c % 32 == 0)) {
                vh_conv_args q{};
                q.src0 = q.src1 = reinterpret_cast<const float*>(16); q.out = reinterpret_cast<float*>(16); q.c0 = C; q.c1 = x.c; q.tail_f32 = 1;
                q.src2 = skip ? reinterpret_cast<const float*>(16) : nullptr; q.c2 = skip ? skip->c : 0;
                q.rows = rows; q.h = R; q.w = R; q.up = 0; q.taps = 9; q.cout = C; q.prec = VH_PREC_BF16X3; q.kernel = VH_CONV_GLDS256; q.epi = VH_EPI_STORE;
                tail32 = vh_conv_takes_patch(&q) == 1;
            }
            // ... and conv_res0 stages its patches from the fp32 tensors itself (vh_conv_args.src_f32; engine.Engine._src_f32): then no vh_split pass at all
            if (x.c % 32 == 0 && (!skip || skip->c % 32 == 0)) {
                vh_conv_args q{};
                q.src0 = reinterpret_cast<const float*>(16); q.src1 = skip ? reinterpret_cast<const float*>(16) : nullptr; q.out_s8 = reinterpret_cast<void*>(16);
                q.c0 = x.c; q.c1 = skip ? skip->c : 0; q.src_f32 = 1; q.pro = VH_PRO_SILU;
                q.rows = rows; q.h = R; q.w = R; q.up = up; q.taps = 9; q.cout = C; q.prec = VH_PREC_BF16X3; q.kernel = VH_CONV_GLDS256; q.epi = VH_EPI_SCALE_SILU;
                src32 = vh_conv_takes_patch(&q) == 1;
            }
            if (has_skip_conv && !tail32) {
                if (src32) cs.second = split(n, x, sc0, skip, sc1, npix_in, rows, Rin, Rin, VH_PRO_NONE, false).first;
                else cs = split(n, x, sc0, skip, sc1, npix_in, rows, Rin, Rin, VH_PRO_SILU, true);
            } else if (!src32) {
                cs = split(n, x, sc0, skip, sc1, npix_in, rows, Rin, Rin, VH_PRO_SILU, false);
            }
        }
        ConvOpt o0; o0.up = up; o0.epi = VH_EPI_SCALE_SILU; o0.cvec = cv; o0.cvec_ld = emb.total; o0.s8_only = true;
        Buf y;
        if (src32) {
            o0.src_f32 = true; o0.f1 = skip; o0.fsc0 = sc0; o0.fsc1 = sc1; o0.pro = VH_PRO_SILU;
            y = conv(n, x, n->W.at(p + "conv_res0.weight"), rows, R, R, o0).second;
        } else {
            y = conv(n, cs.first, n->W.at(p + "conv_res0.weight"), rows, R, R, o0).second;
            release(n, cs.first);
        }
        std::pair<Buf, Buf> r;
        if (has_skip_conv) {
            // conv_res1 + conv_skip as one GEMM: the raw concat is the 1-tap tail segment, ta / tb are folded into the weights
            ConvOpt o1; o1.epi = VH_EPI_STORE; o1.clip = clip_res; o1.also_s8 = (res1_s8 || fin_s8) && !fin_only; o1.s8_only = fin_only;
            if (tail32) { o1.tail0 = &x; o1.tail1 = skip; o1.tsc0 = sc0; o1.tsc1 = skip ? sc1 : 0.f; } else o1.src1 = &cs.second;
            o1.sink_j = out_sink_j; o1.sink_half = out_sink_half; o1.fp32_optional = fp32_optional;
            r = conv(n, y, n->W.at(p + "conv_res1+skip"), rows, R, R, o1);
            release(n, cs.second);
        } else {
            ConvOpt o1; o1.epi = VH_EPI_MPSUM; o1.res = &x; o1.res_up = up; o1.ta = ta; o1.tb = tb; o1.clip = clip_res; o1.also_s8 = (res1_s8 || fin_s8) && !fin_only;
            o1.s8_only = fin_only; o1.sink_j = out_sink_j; o1.sink_half = out_sink_half; o1.fp32_optional = fp32_optional;
            r = conv(n, y, n->W.at(p + "conv_res1.weight"), rows, R, R, o1);
        }
        release(n, y); release(n, xup);
        out = r.first; r_s8 = r.second;
        if (fin_only) out = ghost(C);
    }
    if (fin_s8 || fin_only) out_s8 = r_s8;
    if (b.heads) {
        const int S = R * R;
        const bool use_feat = b.xattn && feat != nullptr;
        const int kl = use_feat ? S * (1 + nsrc) : S;
        const float nz = (b.xattn && !use_feat) ? n_zero * S : 0.f;
        const int klp = round_up(kl, 64);
        const bool fused = S % 32 == 0;
        // the cross keys / values of this block were written when the features were (VH_NET_FEATURES: they depend on the features only,
        // training/models.py:279-297); attn_qkv adds the self keys at offset 0 of the same tensors
        const bool kv_pre = use_feat && feat->k.ok() && fused;
        Buf q = alloc(n, rows, b.heads, S, D), k, v;
        if (kv_pre) { k = feat->k; v = feat->v; } else { k = alloc(n, rows, b.heads, klp, D); v = alloc(n, rows, b.heads, klp, D); }
        const float qscale = (float)(LOG2E / std::sqrt((double)D));
        if (fused) {
            vh_qkv_epilogue e{}; e.q = ptr(n, q); e.k = ptr(n, k); e.v = ptr(n, v); e.heads = b.heads; e.nj = 3; e.rows_per_b = 1; e.koff = 0; e.kl = kl; e.qscale = qscale;
            ConvOpt oq; oq.qkv = &e;
            conv(n, r_s8, n->W.at(p + "attn_qkv.weight"), rows, R, R, oq);
        } else {
            Buf qkv = conv(n, r_s8, n->W.at(p + "attn_qkv.weight"), rows, R, R, ConvOpt{}).first;
            vh_qkv_split_args sa{}; sa.in = ptr(n, qkv); sa.rows = rows; sa.s = S; sa.heads = b.heads; sa.d = D; sa.nj = 3; sa.rows_per_b = 1; sa.koff = 0; sa.kl = kl;
            sa.qscale = qscale; sa.q = ptr(n, q); sa.k = ptr(n, k); sa.v = ptr(n, v);
            call(n, vh_qkv_split_x3, sa);
            release(n, qkv);
        }
        release(n, r_s8);
        Buf fs_own;
        Feat fsplit;
        if (use_feat && !kv_pre && !feat->s8.ok()) {                         // an injected list comes as fp32 only: its S8 form is made here (engine._block: own = True)
            fs_own = split(n, feat->f32, 1.f, nullptr, 1.f, (long long)rows * nsrc * R * R, rows * nsrc, R, R, VH_PRO_NONE, false).first;
            fsplit.f32 = feat->f32; fsplit.s8 = fs_own;
            feat = &fsplit;
        }
        if (use_feat && !kv_pre) {
            if (fused) {
                vh_qkv_epilogue e2{}; e2.q = nullptr; e2.k = ptr(n, k); e2.v = ptr(n, v); e2.heads = b.heads; e2.nj = 2; e2.rows_per_b = nsrc; e2.koff = S; e2.kl = kl; e2.qscale = 1.f;
                ConvOpt ok; ok.qkv = &e2;
                conv(n, feat->s8, n->W.at(p + "x_attn_kv.weight"), rows * nsrc, R, R, ok);
            } else {
                Buf kv = conv(n, feat->s8, n->W.at(p + "x_attn_kv.weight"), rows * nsrc, R, R, ConvOpt{}).first;
                vh_qkv_split_args sa{}; sa.in = ptr(n, kv); sa.rows = rows * nsrc; sa.s = S; sa.heads = b.heads; sa.d = D; sa.nj = 2; sa.rows_per_b = nsrc; sa.koff = S; sa.kl = kl;
                sa.qscale = 1.f; sa.q = nullptr; sa.k = ptr(n, k); sa.v = ptr(n, v);
                call(n, vh_qkv_split_x3, sa);
                release(n, kv);
            }
            release(n, fs_own);
        }
        Buf att = alloc(n, rows, R, R, C);
        vh_attention_args aa{}; aa.q = ptr(n, q); aa.k = ptr(n, k); aa.v = ptr(n, v); aa.b = rows; aa.heads = b.heads; aa.s = S; aa.kl = kl; aa.d = D;
        aa.n_zero_keys = nz; aa.out = ptr(n, att); aa.out_s8 = 1; aa.logit_bound = (float)(LOG2E * std::sqrt((double)D) * 1.001);
        call(n, vh_attention_x3, aa);
        release(n, q);
        if (!kv_pre) { release(n, k); release(n, v); }
        float ta2, tb2; mp_sum_coeffs(cfg.attn_balance, ta2, tb2);
        ConvOpt op; op.epi = VH_EPI_MPSUM; op.res = &out; op.ta = ta2; op.tb = tb2; op.clip = clip; op.out = &out; op.also_s8 = want_s8;
        auto r2 = conv(n, att, n->W.at(p + "attn_proj.weight"), rows, R, R, op);
        if (want_s8) out_s8 = r2.second;
        release(n, att);
    }
    return {out, out_s8};
}

// ---- the exact-fp32 walk (vh_net_config.fp32; the fp32 branches of engine.Engine._conv / _block): fp32 NHWC tensors only, mp_silu / mp_cat in
// the convolution's loader, VH_PREC_F32 on the register-staged 128x128 tile, vh_qkv_split / vh_attention - no S8 forms, no fusions
struct ConvF {
    const Buf* s1 = nullptr; float sc0 = 1.f, sc1 = 1.f;      // second source of a channel concat and the mp_cat weights
    int up = 0, pro = VH_PRO_NONE, epi = VH_EPI_STORE; const float* cvec = nullptr; int cvec_ld = 0;
    const Buf* res = nullptr; int res_up = 0; float ta = 0.f, tb = 0.f, clip = 0.f; Buf* out = nullptr;
};
Buf conv_f32(vh_net* n, const Buf& s0, const Weight& W, int rows, int h, int w, const ConvF& o) {
    Buf out = o.out ? *o.out : alloc(n, rows, h, w, W.cout);
    vh_conv_args a{};
    a.src0 = ptr(n, s0); a.src1 = o.s1 ? ptr(n, *o.s1) : nullptr; a.c0 = s0.c; a.c1 = o.s1 ? o.s1->c : 0; a.scale0 = o.sc0; a.scale1 = o.sc1;
    a.rows = rows; a.h = h; a.w = w; a.up = o.up; a.taps = W.taps; a.pro = o.pro;
    a.wt = W.wt; a.cin_pad = W.cin_pad; a.k_pad = W.k_pad; a.zeros = n->zeros; a.zeros_bytes = ZEROS_FLOATS * 4; a.cout = W.cout;
    a.scratch = n->scratch; a.scratch_floats = SCRATCH_FLOATS;
    a.out = ptr(n, out); a.out_s8 = nullptr; a.out_s8_c = 0;
    a.prec = VH_PREC_F32; a.kernel = VH_CONV_TILE128; a.epi = o.epi; a.cvec = o.cvec; a.cvec_ld = o.cvec_ld;
    a.res = o.res ? ptr(n, *o.res) : nullptr; a.res_up = o.res_up; a.res_scale = nullptr; a.ta = o.ta; a.tb = o.tb; a.clip = o.clip; a.qkv = nullptr;
    a.stagger = 0; a.korder = VH_KORDER_AUTO; a.tile = VH_TILE_AUTO;
    call(n, vh_conv, a);
    return out;
}

// Block.forward :165-206 / XAttnBlock.forward :251-315, fp32
Buf block_f32(vh_net* n, const std::string& prefix, const Block& b, int rows, const Buf& x_in, const Buf* skip, const Buf& cvec_all,
              const vh_net::EmbW& emb, const Feat* feat, int nsrc, float n_zero) {
    const vh_net_config& cfg = n->cfg;
    const std::string p = prefix + (b.dec ? "dec." : "enc.") + b.name + ".";
    const int R = b.res, C = b.cout, D = b.heads ? C / b.heads : 0;
    const float* cv = ptr(n, cvec_all) ? ptr(n, cvec_all) + emb.cols.at(p) : (const float*)nullptr;
    if (!n->emit) cv = reinterpret_cast<const float*>(16);          // (dry walk: never dereferenced)
    float ta, tb; mp_sum_coeffs(cfg.res_balance, ta, tb);
    const float clip = cfg.clip_act > 0.0 ? (float)cfg.clip_act : 0.f, clip_res = b.heads ? 0.f : clip;
    const bool has_skip_conv = b.cin != b.cout;
    Buf out;
    if (!b.dec) {
        Buf xn;
        vh_pixnorm_args pa{}; pa.rows = rows; pa.h = R; pa.w = R; pa.c = C; pa.norm = 1; pa.out_s8 = nullptr;
        if (b.resample == 2 && n->std_filter) {
            xn = alloc(n, rows, R, R, C);
            pa.in = ptr(n, x_in); pa.out = ptr(n, xn); pa.pool = 1;
        } else if (b.resample == 2) {
            xn = resample(n, x_in, rows, R * 2, R * 2, false);
            pa.in = ptr(n, xn); pa.out = ptr(n, xn); pa.pool = 0;
        } else if (has_skip_conv) {
            xn = conv_f32(n, x_in, n->W.at(p + "conv_skip.weight"), rows, R, R, ConvF{});
            pa.in = ptr(n, xn); pa.out = ptr(n, xn); pa.pool = 0;
        } else {
            xn = alloc(n, rows, R, R, C);
            pa.in = ptr(n, x_in); pa.out = ptr(n, xn); pa.pool = 0;
        }
        call(n, vh_pixnorm, pa);
        ConvF o0; o0.pro = VH_PRO_SILU; o0.epi = VH_EPI_SCALE_SILU; o0.cvec = cv; o0.cvec_ld = emb.total;
        Buf y = conv_f32(n, xn, n->W.at(p + "conv_res0.weight"), rows, R, R, o0);
        ConvF o1; o1.epi = VH_EPI_MPSUM; o1.res = &xn; o1.ta = ta; o1.tb = tb; o1.clip = clip_res;
        out = conv_f32(n, y, n->W.at(p + "conv_res1.weight"), rows, R, R, o1);
        release(n, y); release(n, xn);
    } else {
        int up = b.resample == 1 ? 1 : 0;
        Buf xup; const Buf* xp = &x_in;
        if (up && !n->std_filter) { xup = resample(n, x_in, rows, R / 2, R / 2, true); xp = &xup; up = 0; }
        const Buf& x = *xp;
        ConvF src;                                                // the (1-2) sources every convolution of this block that reads the input shares
        if (skip) {                                               // mp_cat :78-84
            const double t = cfg.concat_balance, Na = x.c, Nb = skip->c;
            const double Cc = std::sqrt((Na + Nb) / ((1 - t) * (1 - t) + t * t));
            src.sc0 = (float)(Cc / std::sqrt(Na) * (1 - t)); src.sc1 = (float)(Cc / std::sqrt(Nb) * t); src.s1 = skip;
        }
        ConvF o0 = src; o0.up = up; o0.pro = VH_PRO_SILU; o0.epi = VH_EPI_SCALE_SILU; o0.cvec = cv; o0.cvec_ld = emb.total;
        Buf y = conv_f32(n, x, n->W.at(p + "conv_res0.weight"), rows, R, R, o0);
        Buf xsk;
        ConvF o1; o1.epi = VH_EPI_MPSUM; o1.ta = ta; o1.tb = tb; o1.clip = clip_res;
        if (has_skip_conv) {
            ConvF os = src; os.up = up;
            xsk = conv_f32(n, x, n->W.at(p + "conv_skip.weight"), rows, R, R, os);
            o1.res = &xsk; o1.res_up = 0;
        } else { o1.res = &x; o1.res_up = up; }
        out = conv_f32(n, y, n->W.at(p + "conv_res1.weight"), rows, R, R, o1);
        release(n, y); release(n, xsk); release(n, xup);
    }
    if (b.heads) {
        const int S = R * R;
        const bool use_feat = b.xattn && feat != nullptr;
        const int kl = use_feat ? S * (1 + nsrc) : S;
        const float nz = (b.xattn && !use_feat) ? n_zero * S : 0.f;
        Buf q = alloc(n, rows, b.heads, S, D), k = alloc(n, rows, b.heads, kl, D), v = alloc(n, rows, b.heads, kl, D);
        const float qscale = (float)(LOG2E / std::sqrt((double)D));
        Buf qkv = conv_f32(n, out, n->W.at(p + "attn_qkv.weight"), rows, R, R, ConvF{});
        vh_qkv_split_args sa{}; sa.in = ptr(n, qkv); sa.rows = rows; sa.s = S; sa.heads = b.heads; sa.d = D; sa.nj = 3; sa.rows_per_b = 1; sa.koff = 0; sa.kl = kl;
        sa.qscale = qscale; sa.q = ptr(n, q); sa.k = ptr(n, k); sa.v = ptr(n, v);
        call(n, vh_qkv_split, sa);
        release(n, qkv);
        if (use_feat) {
            Buf kv = conv_f32(n, feat->f32, n->W.at(p + "x_attn_kv.weight"), rows * nsrc, R, R, ConvF{});
            vh_qkv_split_args sb{}; sb.in = ptr(n, kv); sb.rows = rows * nsrc; sb.s = S; sb.heads = b.heads; sb.d = D; sb.nj = 2; sb.rows_per_b = nsrc; sb.koff = S; sb.kl = kl;
            sb.qscale = 1.f; sb.q = nullptr; sb.k = ptr(n, k); sb.v = ptr(n, v);
            call(n, vh_qkv_split, sb);
            release(n, kv);
        }
        Buf att = alloc(n, rows, R, R, C);
        vh_attention_args aa{}; aa.q = ptr(n, q); aa.k = ptr(n, k); aa.v = ptr(n, v); aa.b = rows; aa.heads = b.heads; aa.s = S; aa.kl = kl; aa.d = D;
        aa.n_zero_keys = nz; aa.out = ptr(n, att); aa.out_s8 = 0; aa.logit_bound = (float)(LOG2E * std::sqrt((double)D) * 1.001);
        call(n, vh_attention, aa);
        release(n, q); release(n, k); release(n, v);
        float ta2, tb2; mp_sum_coeffs(cfg.attn_balance, ta2, tb2);
        ConvF op; op.epi = VH_EPI_MPSUM; op.res = &out; op.ta = ta2; op.tb = tb2; op.clip = clip; op.out = &out;
        conv_f32(n, att, n->W.at(p + "attn_proj.weight"), rows, R, R, op);
        release(n, att);
    }
    return out;
}

// emb = mp_silu(mp_sum(emb_noise(fourier(c_noise)), emb_label(geometry))) :388-391 and every block's emb_linear in one launch
Buf embedding(vh_net* n, const std::string& prefix, const Spec& sp, const vh_net::EmbW& emb, int rows, const Buf& sigma, int sigma_stride, float time_scale,
              const Buf& geometry, int label_dim) {
    const vh_net_config& cfg = n->cfg;
    Buf e = alloc(n, rows, 1, 1, sp.cemb);
    const Weight& wn = n->W.at(prefix + "emb_noise.weight");
    const bool has_label = sp.label_dim > 0;
    vh_embed_args a{};
    a.sigma = ptr(n, sigma); a.sigma_stride = sigma_stride; a.time_scale = time_scale;
    a.geometry = has_label ? ptr(n, geometry) : nullptr; a.label_dim = has_label ? label_dim : 0; a.geometry_scale = cfg.uncond ? 0.f : 1.f;
    a.freqs = P(n, prefix + "emb_fourier.freqs"); a.phases = P(n, prefix + "emb_fourier.phases"); a.cnoise = sp.cnoise;
    a.w_noise = wn.wt; a.w_noise_kpad = wn.k_pad;
    if (has_label) { const Weight& wl = n->W.at(prefix + "emb_label.weight"); a.w_label = wl.wt; a.w_label_kpad = wl.k_pad; }
    a.label_balance = (float)cfg.label_balance; a.rows = rows; a.cemb = sp.cemb; a.raw = 0; a.emb = ptr(n, e);
    call(n, vh_embed, a);
    Buf cvec = alloc(n, rows, 1, 1, emb.total);
    vh_linear_args l{}; l.emb = ptr(n, e); l.rows = rows; l.cemb = sp.cemb; l.wt = emb.wt; l.k_pad = round_up(sp.cemb, 32); l.cols = emb.total; l.bias = 1.f; l.out = ptr(n, cvec);
    call(n, vh_linear, l);
    release(n, e);
    return cvec;
}

// UNetEncoder.forward :536-570 (collect) / XAttnUNet.forward :483-518
Buf run_unet(vh_net* n, const std::string& prefix, const Spec& sp, const vh_net::EmbW& emb, int rows, Buf x_in, const Buf& cvec,
             const std::vector<Feat>* feats, bool collect, float n_zero, int nsrc, std::vector<Feat>* out_feats, Buf* last_s8 = nullptr) {
    std::vector<Buf> skips;
    size_t fi = 0;
    Buf x = x_in;
    auto kept = [&](const Buf& b) { if (out_feats) for (auto& f : *out_feats) if (f.f32.off == b.off) return true; return false; };
    auto in_skips = [&](const Buf& b) { for (auto& s : skips) if (s.off == b.off) return true; return false; };
    auto next_feat = [&](const Block& b) -> const Feat* {
        const Feat* f = nullptr;
        if (b.xattn) { if (feats) f = &(*feats)[fi]; ++fi; }
        return f;
    };
    // decoder concat inputs written by their producers: which encoder entry each skip-taking decoder block pops (UNet.forward's skip stack,
    // training/models.py:507-510), its two channel counts and mp_cat weights (:78-84), whether conv_skip needs the raw form too
    n->cat.clear();
    std::map<int, int> consumer;          // encoder entry index -> decoder block index
    {
        int k = (int)sp.enc.size() - 1, cprev = sp.enc.back().cout;
        const double t = n->cfg.concat_balance;
        for (int j = 0; j < (int)sp.dec.size(); ++j) {
            const Block& b = sp.dec[j];
            if (!b.live) break;
            if (b.takes_skip) {
                CatState st; st.rows = rows; st.R = b.res; st.Nb = sp.enc[k].cout; st.Na = b.cin - st.Nb;
                const double Cc = std::sqrt((double)(st.Na + st.Nb) / ((1 - t) * (1 - t) + t * t));
                st.sc0 = (float)(Cc / std::sqrt((double)st.Na) * (1 - t)); st.sc1 = (float)(Cc / std::sqrt((double)st.Nb) * t);
                st.raw = b.cin != b.cout; st.ok = st.Na % 32 == 0 && st.Nb % 32 == 0 && b.resample != 1 && st.Na == cprev;
                n->cat[j] = st;
                consumer[k] = j;
                --k;
            }
            cprev = b.cout;
        }
    }
    const int fuse = n->fp32 ? 0 : vh_knob(VH_KNOB_FUSE_CONCAT);
    if (fuse <= 0) { n->cat.clear(); consumer.clear(); }
    auto skip_sink = [&](int ei) { auto it = consumer.find(ei); return (fuse >= 2 && it != consumer.end() && n->cat.at(it->second).ok) ? it->second : -1; };
    for (int ei = 0; ei < (int)sp.enc.size(); ++ei) {
        const Block& b = sp.enc[ei];
        Buf nx;
        if (b.conv && n->fp32) {
            nx = conv_f32(n, x, n->W.at(prefix + "enc." + b.name + ".weight"), rows, b.res, b.res, ConvF{});
            release(n, x);
        } else if (n->fp32) {
            const Feat* f = next_feat(b);
            nx = block_f32(n, prefix, b, rows, x, nullptr, cvec, emb, f, nsrc, n_zero);
            if (collect && b.heads > 0) out_feats->push_back(Feat{nx, Buf{}});
        } else if (b.conv) {
            auto xs8 = split(n, x, 1.f, nullptr, 1.f, (long long)rows * b.res * b.res, rows, b.res, b.res, VH_PRO_NONE, false);
            ConvOpt oc; oc.sink_j = skip_sink(ei); oc.sink_half = 1;
            nx = conv(n, xs8.first, n->W.at(prefix + "enc." + b.name + ".weight"), rows, b.res, b.res, oc).first;
            release(n, xs8.first);
            release(n, x);
        } else {
            const Feat* f = next_feat(b);
            auto r = block(n, prefix, b, rows, x, nullptr, cvec, emb, f, nsrc, n_zero, collect && b.heads > 0, -1, skip_sink(ei), 1, false);
            nx = r.first;
            if (collect && b.heads > 0) out_feats->push_back(Feat{r.first, r.second});
            // x was the output of entry ei-1; if its skip half already sits in its consumer's concat tensors, this block was its last fp32 reader
            auto pj = consumer.find(ei - 1);
            if (pj != consumer.end() && n->cat.at(pj->second).skip_done && x.ok() && !kept(x) && !skips.empty() && skips.back().off == x.off) {
                release(n, x);
                skips.back() = ghost(x.c);
            }
        }
        skips.push_back(nx);
        x = nx;
    }
    for (int j = 0; j < (int)sp.dec.size(); ++j) {
        const Block& b = sp.dec[j];
        if (!b.live) break;
        Buf skip; const Buf* sk = nullptr;
        if (b.takes_skip) { skip = skips.back(); skips.pop_back(); sk = &skip; }
        const Feat* f = next_feat(b);
        // this block's result is the x half of the NEXT block's concat input, and nothing else reads it
        const Block* nb = (j + 1 < (int)sp.dec.size() && sp.dec[j + 1].live) ? &sp.dec[j + 1] : nullptr;
        const bool xs_ok = nb && nb->takes_skip && n->cat.count(j + 1) && n->cat.at(j + 1).ok && !b.heads;
        const bool s8_final = !collect && !nb && last_s8 && sp.out_channels > 0;
        auto r = n->fp32 ? std::pair<Buf, Buf>{block_f32(n, prefix, b, rows, x, sk, cvec, emb, f, nsrc, n_zero), Buf{}}
                         : block(n, prefix, b, rows, x, sk, cvec, emb, f, nsrc, n_zero, collect && b.heads > 0, n->cat.count(j) ? j : -1, xs_ok ? j + 1 : -1, 0, xs_ok, s8_final);
        if (s8_final && !r.first.ok() && r.second.ok()) *last_s8 = r.second;
        if (x.ok() && !kept(x) && !in_skips(x)) release(n, x);
        if (sk && skip.ok() && !kept(skip) && !in_skips(skip) && skip.off != x.off) release(n, skip);
        if (collect && b.heads > 0) out_feats->push_back(Feat{r.first, r.second});
        x = r.first;
    }
    for (auto& s : skips) if (s.ok() && !kept(s) && s.off != x.off) release(n, s);
    return x;
}

Buf assemble(vh_net* n, const vh_segment* segs, int nseg, int rows, int R, int c_pad, const Buf& sigma) {
    Buf out = alloc(n, rows, R, R, c_pad);
    vh_assemble_args a{};
    a.nseg = nseg; for (int i = 0; i < nseg; ++i) a.seg[i] = segs[i];
    a.sigma = ptr(n, sigma); a.sigma_data = (float)n->cfg.sigma_data; a.rows = rows; a.h = R; a.w = R; a.c_pad = c_pad; a.out = ptr(n, out);
    call(n, vh_assemble, a);
    return out;
}

// NVPrecond._forward_dualsource :628-689 / forward :691-749 ("full" when the net has an encoder, "uncond" otherwise)
// mode VH_NET_FULL: encoder (when the net has one) + UNet; VH_NET_FEATURES: encoder only, the features stay in this program's workspace;
// VH_NET_BOUND: UNet only, reading `ext` = the feature buffers of a VH_NET_FEATURES program in place (the sampler's split evaluation)
void walk(vh_net* n, int B, Program& pr, int mode, const std::vector<Feat>* ext) {
    const vh_net_config& cfg = n->cfg;
    const int R = cfg.img_resolution, nsrc = cfg.dual_source ? 2 : 1, rm = nsrc, rows_all = B * rm;
    const int src_c = 3 + ((cfg.depth_input || cfg.warp_depth_coor) ? 1 : 0);
    const bool need_enc = n->has_enc && mode != VH_NET_BOUND && mode != VH_NET_INJECT, need_unet = mode != VH_NET_FEATURES;
    pr.mode = mode;
    pr.sigma = alloc(n, rows_all, 1, 1, 1);
    pr.geometry = alloc(n, rows_all, 1, 1, cfg.source_label_dim);
    if (need_enc || cfg.warp_depth_coor) pr.src = alloc(n, rows_all, src_c, R, R);
    if (need_unet) {
        pr.x = alloc(n, rows_all, cfg.img_channels, R, R);
        pr.D = alloc(n, B, cfg.img_channels, R, R);
        if (cfg.super_res) pr.cond = alloc(n, B, cfg.img_channels, R, R);
    }
    std::vector<Feat> inj;
    if (mode == VH_NET_INJECT) {                                  // inject_features :664-665: the caller's list, copied into these NHWC buffers per call
        for (int g = 0; g < 2; ++g)
            for (const Block& b : (g ? n->unet.dec : n->unet.enc))
                if (!b.conv && b.xattn) { Buf f = alloc(n, rows_all, b.res, b.res, b.cout); pr.feats_in.push_back(f); Feat ft; ft.f32 = f; inj.push_back(ft); }
    }
    Buf sgrid, dgrid;
    if (cfg.warp_depth_coor) {                                    // depth-warp Fourier features :643-652
        sgrid = alloc(n, rows_all, R, R, 128); dgrid = alloc(n, rows_all, R, R, 128);
        Buf flag = alloc(n, 1, 1, 1, 1);
        vh_nonzero_args nz{}; nz.in = ptr(n, pr.src); nz.rows = rows_all; nz.c_used = 3; nz.c_total = src_c; nz.hw = R * R; nz.flag = ptr(n, flag);
        call(n, vh_nonzero_flag, nz);
        vh_warp_args wa{}; wa.depth = ptr(n, pr.src); wa.src_c = src_c; wa.depth_ch = 3; wa.geometry = ptr(n, pr.geometry);
        std::memcpy(wa.mean, cfg.geom_mean, sizeof wa.mean); std::memcpy(wa.std, cfg.geom_std, sizeof wa.std);
        wa.freqs = P(n, "logvar_fourier.freqs"); wa.phases = P(n, "logvar_fourier.phases"); wa.rows = rows_all; wa.s = R;
        wa.grid_feat = ptr(n, sgrid); wa.warp_feat = ptr(n, dgrid); wa.nonzero_flag = ptr(n, flag);
        call(n, vh_warp_features, wa);
        release(n, flag);
    }
    std::vector<Feat> feats;
    if (need_enc) {
        n->scratch = n->scratch_enc;
        vh_segment segs[2]; int ns = 0;
        segs[ns++] = vh_segment{ptr(n, pr.src), 0, cfg.warp_depth_coor ? 3 : src_c, src_c, 1, 0};
        if (cfg.warp_depth_coor) segs[ns++] = vh_segment{ptr(n, sgrid), 1, 128, 128, 1, 0};
        Buf xin = assemble(n, segs, ns, rows_all, R, round_up(n->enc.in_channels, n->fp32 ? 4 : 8), pr.sigma);
        release(n, sgrid);
        Buf cvec = embedding(n, "encoder.", n->enc, n->embE, rows_all, pr.sigma, 1, cfg.no_time_enc ? 0.f : 1.f, pr.geometry, cfg.source_label_dim);
        Buf last = run_unet(n, "encoder.", n->enc, n->embE, rows_all, xin, cvec, nullptr, true, 0.f, nsrc, &feats);
        bool last_kept = false; for (auto& f : feats) if (f.f32.off == last.off) last_kept = true;
        if (!last_kept) release(n, last);
        release(n, cvec);
    }
    release(n, sgrid);
    if (mode == VH_NET_FEATURES) {
        // cross-attention K / V of every XAttnBlock of the UNet, computed with the features (engine.Engine._cross_kv)
        size_t fi = 0;
        for (int g = 0; g < 2; ++g)
            for (const Block& b : (g ? n->unet.dec : n->unet.enc)) {
                if (b.conv || !b.xattn) continue;
                Feat& f = feats.at(fi++);
                const int S = b.res * b.res, D = b.cout / b.heads;
                if (S % 32 != 0 || !f.s8.ok()) continue;
                const int kl = S * (1 + nsrc), klp = round_up(kl, 64);
                f.k = alloc(n, B, b.heads, klp, D); f.v = alloc(n, B, b.heads, klp, D);
                vh_qkv_epilogue e2{}; e2.q = nullptr; e2.k = ptr(n, f.k); e2.v = ptr(n, f.v); e2.heads = b.heads; e2.nj = 2; e2.rows_per_b = nsrc; e2.koff = S; e2.kl = kl; e2.qscale = 1.f;
                ConvOpt ok; ok.qkv = &e2;
                conv(n, f.s8, n->W.at(std::string("unet.") + (g ? "dec." : "enc.") + b.name + ".x_attn_kv.weight"), B * nsrc, b.res, b.res, ok);
            }
        pr.feats = feats;
    }
    if (need_unet) {
        const bool have_feats = need_enc || ext != nullptr || mode == VH_NET_INJECT;
        const std::vector<Feat>* use = mode == VH_NET_INJECT ? &inj : ext ? ext : &feats;
        n->scratch = n->scratch_unet;
        vh_segment segs[3]; int ns = 0;
        segs[ns++] = vh_segment{ptr(n, pr.x), 0, cfg.img_channels, cfg.img_channels, rm, 1};
        if (cfg.warp_depth_coor) segs[ns++] = vh_segment{ptr(n, dgrid), 1, 128, 128, rm, 0};
        if (cfg.super_res) segs[ns++] = vh_segment{ptr(n, pr.cond), 0, cfg.img_channels, cfg.img_channels, 1, 0};
        Buf xin = assemble(n, segs, ns, B, R, round_up(n->unet.in_channels, n->fp32 ? 4 : 8), pr.sigma);
        release(n, dgrid);
        Buf cvec = embedding(n, "unet.", n->unet, n->embU, B, pr.sigma, rm, 1.f, pr.geometry, cfg.target_label_dim);
        const float n_zero = have_feats ? 0.f : (float)nsrc;
        Buf last8;
        Buf last = run_unet(n, "unet.", n->unet, n->embU, B, xin, cvec, have_feats ? use : nullptr, false, n_zero, nsrc, nullptr, &last8);
        Buf F;
        if (n->fp32) F = conv_f32(n, last, n->W.at("unet.out_conv.weight"), B, R, R, ConvF{});
        else {
            Buf ls8 = last8.ok() ? last8 : split(n, last, 1.f, nullptr, 1.f, (long long)B * R * R, B, R, R, VH_PRO_NONE, false).first;
            F = conv(n, ls8, n->W.at("unet.out_conv.weight"), B, R, R, ConvOpt{}).first;
            release(n, ls8);
        }
        release(n, last); release(n, cvec);
        vh_precond_out_args po{}; po.x = ptr(n, pr.x); po.row_mul = rm; po.f = ptr(n, F); po.fc = F.c; po.sigma = ptr(n, pr.sigma); po.sigma_data = (float)cfg.sigma_data;
        po.rows = B; po.c = cfg.img_channels; po.h = R; po.w = R; po.out = ptr(n, pr.D);
        call(n, vh_precond_out, po);
        release(n, F);
    }
    release(n, dgrid);
}

int check_config(const vh_net_config& c) {
    VH_REQUIRE(c.img_resolution > 0 && c.img_channels == 3, "vh_net: img_channels must be 3 (UNet out_conv is hard-wired to 3, training/models.py:480)");
    VH_REQUIRE(c.num_levels >= 1 && c.num_levels <= 8 && c.num_blocks >= 1 && c.model_channels > 0, "vh_net: bad architecture");
    VH_REQUIRE(c.num_attn_resolutions >= 0 && c.num_attn_resolutions <= 8, "vh_net: bad attn_resolutions");
    VH_REQUIRE((c.img_resolution >> (c.num_levels - 1)) >= 1 && c.img_resolution % (1 << (c.num_levels - 1)) == 0, "vh_net: resolution not divisible by the level count");
    VH_REQUIRE(c.source_label_dim > 0 && c.target_label_dim >= 0 && c.logvar_channels > 0, "vh_net: bad label dims");
    for (int i = 0; i < c.num_levels && !c.fp32; ++i)
        VH_REQUIRE((c.model_channels * c.channel_mult[i]) % 32 == 0, "vh_net: the bf16x3 path needs channel counts that are multiples of 32 (level %d has %d)", i, c.model_channels * c.channel_mult[i]);
    VH_REQUIRE(c.resample_ntaps == 0 || (c.resample_ntaps >= 2 && c.resample_ntaps <= 8 && c.resample_ntaps % 2 == 0),
               "vh_net: resample_filter must have 2, 4, 6 or 8 taps (the reference asserts an even length, training/models.py:52); got %d", c.resample_ntaps);
    double tot = 0.0;
    for (int i = 0; i < c.resample_ntaps; ++i) tot += (double)c.resample_filter[i];
    VH_REQUIRE(c.resample_ntaps == 0 || tot != 0.0, "vh_net: resample_filter sums to zero");
    return VH_OK;
}

// Layout of the prepared-weight buffer (offsets in floats), fixed at creation: [zero page][split-K scratch x2] then, per network, the
// embedding matrices and every convolution weight in walk order; a permuted copy of one q/k/v weight fits behind the last one.
void layout(vh_net* n) {
    size_t cur = ZEROS_FLOATS + 2 * SCRATCH_FLOATS, biggest = 0;
    auto add = [&](const std::string& key, int taps, int nj, int D, bool has_gain) {
        const Param& p = n->params[n->pindex.at(key)];
        const bool conv4 = p.ndim == 4;
        Weight w; w.cout = p.shape[0]; w.taps = taps; w.cin_pad = round_up(p.shape[1], conv4 ? 32 : 4); w.k_pad = round_up(taps * w.cin_pad, 32);
        w.off = cur; w.nj = nj; w.D = D; w.has_gain = has_gain;
        cur += (size_t)w.k_pad / 4 * w.cout * 4;
        if (nj) biggest = std::max(biggest, (size_t)p.shape[0] * p.shape[1] * taps);
        n->W[key] = w;
        n->prep_order.push_back(key);
    };
    for (int which = 0; which < 2; ++which) {
        if (which == 0 && !n->has_enc) continue;
        const Spec& sp = which == 0 ? n->enc : n->unet;
        const std::string prefix = which == 0 ? "encoder." : "unet.";
        vh_net::EmbW& emb = which == 0 ? n->embE : n->embU;
        add(prefix + "emb_noise.weight", 1, 0, 0, false);
        if (sp.label_dim) add(prefix + "emb_label.weight", 1, 0, 0, false);
        emb.total = 0;
        for (int g = 0; g < 2; ++g) for (const Block& b : (g ? sp.dec : sp.enc)) if (b.live && !b.conv) { emb.cols[prefix + (g ? "dec." : "enc.") + b.name + "."] = emb.total; emb.total += b.cout; }
        emb.off = cur; cur += (size_t)round_up(sp.cemb, 32) / 4 * emb.total * 4;
        for (int g = 0; g < 2; ++g)
            for (const Block& b : (g ? sp.dec : sp.enc)) {
                if (!b.live) continue;
                const std::string p = prefix + (g ? "dec." : "enc.") + b.name + ".";
                if (b.conv) { add(p + "weight", 9, 0, 0, false); continue; }
                add(p + "conv_res0.weight", 9, 0, 0, false);
                if (b.dec && b.cin != b.cout && !n->fp32) {
                    // conv_res1 + conv_skip of a decoder block as one GEMM (vh_conv_args.src1): rows of 9*Cout + Cin_pad K elements
                    Weight w; w.cout = b.cout; w.taps = 9; w.cin_pad = b.cout; w.k_pad = 9 * b.cout + round_up(b.cin, 32); w.off = cur; w.fused_c1 = round_up(b.cin, 32);
                    cur += (size_t)w.k_pad / 4 * w.cout * 4;
                    n->W[p + "conv_res1+skip"] = w;
                    n->prep_order.push_back(p + "conv_res1+skip");
                } else {
                    add(p + "conv_res1.weight", 9, 0, 0, false);
                    if (b.cin != b.cout) add(p + "conv_skip.weight", 1, 0, 0, false);
                }
                if (b.heads) {
                    const int D = b.cout / b.heads;
                    const bool fused = !n->fp32 && (b.res * b.res) % 32 == 0;      // VH_EPI_QKV needs 32 | pixels per image; else vh_qkv_split_x3 on an fp32 tensor
                    add(p + "attn_qkv.weight", 1, fused ? 3 : 0, D, false);
                    add(p + "attn_proj.weight", 1, 0, 0, false);
                    if (b.xattn) add(p + "x_attn_kv.weight", 1, fused ? 2 : 0, D, false);
                }
            }
        if (sp.out_channels) add(prefix + "out_conv.weight", 9, 0, 0, true);
    }
    add("logvar_linear.weight", 1, 0, 0, false);                  // the logvar head :685-688 (vh_net_logvar)
    n->prepared_floats = cur + biggest + 64;
}

int prep_one(vh_net* n, const std::string& key, float* base) {
    Weight& w = n->W.at(key);
    if (w.fused_c1) {
        w.wt = base + w.off;
        const std::string p = key.substr(0, key.size() - std::string("conv_res1+skip").size());
        const Param& p1 = n->params[n->pindex.at(p + "conv_res1.weight")];
        const Param& ps = n->params[n->pindex.at(p + "conv_skip.weight")];
        float ta, tb; mp_sum_coeffs(n->cfg.res_balance, ta, tb);
        vh_prep_weight_args a{};
        a.w = p1.ptr; a.cout = w.cout; a.cin = w.cout; a.taps = 9; a.cin_pad = w.cout; a.k_pad = 9 * w.cout; a.gain_ptr = nullptr; a.gain_value = tb;
        a.wt = w.wt; a.dst_col0 = 0; a.dst_cols = w.cout; a.split = 2; a.k_off = 0; a.k_stride = w.k_pad;
        int rc = vh_prep_weight(n->ctx, &a);
        if (rc != VH_OK) return rc;
        a.w = ps.ptr; a.cin = ps.shape[1]; a.taps = 1; a.cin_pad = w.fused_c1; a.k_pad = w.fused_c1; a.gain_value = ta; a.k_off = 9 * w.cout;
        return vh_prep_weight(n->ctx, &a);
    }
    const Param& p = n->params[n->pindex.at(key)];
    w.wt = base + w.off;
    const int cout = p.shape[0], cin = p.shape[1];
    const float* src = p.ptr;
    if (w.nj) {
        // output channel (head*D + d)*nj + j  ->  (head*nj + j)*D + d (one (head, j) per D-column accumulator slab, VH_EPI_QKV): the rows are
        // gathered by one strided device-to-device copy per (head, j) into the space behind this weight's output, then normalised from there
        const int heads = cout / (w.D * w.nj);
        const size_t row = (size_t)cin * w.taps;
        float* tmp = w.wt + (size_t)w.k_pad / 4 * cout * 4;
        for (int hh = 0; hh < heads; ++hh)
            for (int j = 0; j < w.nj; ++j) {
                const hipError_t e = hipMemcpy2DAsync(tmp + ((size_t)(hh * w.nj + j) * w.D) * row, row * 4, src + ((size_t)hh * w.D * w.nj + j) * row, row * 4 * w.nj,
                                                      row * 4, w.D, hipMemcpyDeviceToDevice, n->ctx->stream);
                if (e != hipSuccess) return vh_fail(VH_EHIP, "vh_net_prepare: %s", hipGetErrorString(e));
            }
        src = tmp;
    }
    vh_prep_weight_args a{};
    a.w = src; a.cout = cout; a.cin = cin; a.taps = w.taps; a.cin_pad = w.cin_pad; a.k_pad = w.k_pad;
    a.gain_ptr = w.has_gain ? P(n, key.substr(0, key.size() - std::string("out_conv.weight").size()) + "out_gain") : nullptr; a.gain_value = 1.f;
    a.wt = w.wt; a.dst_col0 = 0; a.dst_cols = cout; a.split = (p.ndim == 4 && !n->fp32) ? 2 : 0;
    return vh_prep_weight(n->ctx, &a);
}

}  // namespace

extern "C" int vh_net_create(vh_ctx* ctx, const vh_net_config* cfg, vh_net** out) {
    if (!ctx || !cfg || !out) return vh_fail(VH_EINVAL, "vh_net_create: null argument");
    const int rc = check_config(*cfg);
    if (rc != VH_OK) return rc;
    auto n = std::make_unique<vh_net>();
    n->ctx = ctx; n->cfg = *cfg; n->has_enc = !cfg->uncond; n->fp32 = cfg->fp32 != 0;
    if (cfg->resample_ntaps > 0) {                      // f / sum(f) (:53), the division in double like engine.Engine._resample
        double tot = 0.0;
        for (int i = 0; i < cfg->resample_ntaps; ++i) tot += (double)cfg->resample_filter[i];
        n->ntaps = cfg->resample_ntaps;
        for (int i = 0; i < n->ntaps; ++i) n->taps[i] = (float)((double)cfg->resample_filter[i] / tot);
        n->std_filter = n->ntaps == 2 && cfg->resample_filter[0] == 1.f && cfg->resample_filter[1] == 1.f;
    }
    if (n->has_enc) { n->enc = make_spec(*cfg, true); spec_params(n->enc, "encoder.", false, n->params); }
    n->unet = make_spec(*cfg, false); spec_params(n->unet, "unet.", true, n->params);
    auto add = [&](const char* name, std::initializer_list<int> shp) {
        Param p; p.name = name; p.ndim = (int)shp.size(); int i = 0; for (int v : shp) p.shape[i++] = v; for (; i < 4; ++i) p.shape[i] = 1; n->params.push_back(p);
    };
    add("logvar_fourier.freqs", {cfg->logvar_channels}); add("logvar_fourier.phases", {cfg->logvar_channels}); add("logvar_linear.weight", {1, cfg->logvar_channels});
    for (size_t i = 0; i < n->params.size(); ++i) n->pindex[n->params[i].name] = (int)i;
    for (const Spec* sp : {n->has_enc ? (const Spec*)&n->enc : (const Spec*)nullptr, (const Spec*)&n->unet}) {
        if (!sp) continue;
        for (int g = 0; g < 2; ++g)
            for (const Block& b : (g ? sp->dec : sp->enc))
                if (b.live && b.heads) {
                    const int D = b.cout / b.heads;
                    if (D != 32 && D != 64) return vh_fail(VH_EINVAL, "vh_net: attention block %s needs 32- or 64-channel heads", b.name.c_str());
                }
    }
    layout(n.get());
    *out = n.release();
    return VH_OK;
}

extern "C" int vh_net_num_params(const vh_net* n) { return n ? (int)n->params.size() : 0; }

extern "C" int vh_net_param_info(const vh_net* n, int i, const char** name, int* ndim, int* shape) {
    if (!n || i < 0 || i >= (int)n->params.size() || !name || !ndim || !shape) return vh_fail(VH_EINVAL, "vh_net_param_info: bad argument");
    *name = n->params[i].name.c_str(); *ndim = n->params[i].ndim;
    for (int k = 0; k < 4; ++k) shape[k] = n->params[i].shape[k];
    return VH_OK;
}

extern "C" int vh_net_bind_param(vh_net* n, const char* name, const float* device_ptr) {
    if (!n || !name || !device_ptr) return vh_fail(VH_EINVAL, "vh_net_bind_param: null argument");
    auto it = n->pindex.find(name);
    if (it == n->pindex.end()) return vh_fail(VH_EINVAL, "vh_net_bind_param: no parameter named %s", name);
    if (!vh_aligned16(device_ptr) && n->params[it->second].ndim >= 2) return vh_fail(VH_EINVAL, "vh_net_bind_param: %s must be 16-byte aligned", name);
    n->params[it->second].ptr = device_ptr;
    n->prepared = false;
    return VH_OK;
}

extern "C" size_t vh_net_prepared_bytes(const vh_net* n) { return n ? n->prepared_floats * 4 : 0; }

// K1 once per weight version (the reference re-normalises every weight on every forward, training/models.py:115-120)
extern "C" int vh_net_prepare(vh_net* n, void* buffer, size_t bytes) {
    if (!n || !buffer) return vh_fail(VH_EINVAL, "vh_net_prepare: null argument");
    VH_REQUIRE(bytes >= vh_net_prepared_bytes(n) && vh_aligned16(buffer), "vh_net_prepare: buffer too small (%zu < %zu bytes) or unaligned", bytes, vh_net_prepared_bytes(n));
    VH_REQUIRE(!n->ctx->recording, "vh_net_prepare: context is recording");
    for (const Param& p : n->params) VH_REQUIRE(p.ptr, "vh_net_prepare: parameter %s is not bound", p.name.c_str());
    for (auto& kv : n->programs) if (kv.second && kv.second->plan) (void)vh_plan_destroy(kv.second->plan);
    n->programs.clear();
    float* base = static_cast<float*>(buffer);
    hipError_t e = hipMemsetAsync(base, 0, ZEROS_FLOATS * 4, n->ctx->stream);
    if (e != hipSuccess) return vh_fail(VH_EHIP, "vh_net_prepare: %s", hipGetErrorString(e));
    n->zeros = base; n->scratch_enc = base + ZEROS_FLOATS; n->scratch_unet = n->scratch_enc + SCRATCH_FLOATS;
    for (const std::string& key : n->prep_order) {
        const int rc = prep_one(n, key, base);
        if (rc != VH_OK) return rc;
    }
    for (int which = 0; which < 2; ++which) {
        if (which == 0 && !n->has_enc) continue;
        const Spec& sp = which == 0 ? n->enc : n->unet;
        vh_net::EmbW& emb = which == 0 ? n->embE : n->embU;
        const int kpad = round_up(sp.cemb, 32);
        emb.wt = base + emb.off;
        e = hipMemsetAsync(emb.wt, 0, (size_t)kpad / 4 * emb.total * 16, n->ctx->stream);
        if (e != hipSuccess) return vh_fail(VH_EHIP, "vh_net_prepare: %s", hipGetErrorString(e));
        for (auto& kv : emb.cols) {
            vh_prep_weight_args a{};
            a.w = P(n, kv.first + "emb_linear.weight"); a.cout = n->params[n->pindex.at(kv.first + "emb_linear.weight")].shape[0]; a.cin = sp.cemb; a.taps = 1;
            a.cin_pad = round_up(sp.cemb, 4); a.k_pad = kpad; a.gain_ptr = P(n, kv.first + "emb_gain"); a.gain_value = 1.f;
            a.wt = emb.wt; a.dst_col0 = kv.second; a.dst_cols = emb.total; a.split = 0;
            const int rc = vh_prep_weight(n->ctx, &a);
            if (rc != VH_OK) return rc;
        }
    }
    n->prepared = true;
    return VH_OK;
}

static int net_build(vh_net* n, int mode, int slot, int B, float* workspace, size_t bytes, Program** out) {
    VH_REQUIRE(B > 0, "vh_net: batch must be positive");
    VH_REQUIRE(mode >= VH_NET_FULL && mode <= VH_NET_INJECT && (slot == 0 || slot == 1), "vh_net: bad mode / slot");
    VH_REQUIRE(mode == VH_NET_FULL || mode == VH_NET_INJECT || n->has_enc, "vh_net: an uncond net has no encoder: only VH_NET_FULL / VH_NET_INJECT");
    VH_REQUIRE(n->prepared || !workspace, "vh_net_record: call vh_net_prepare after binding the parameters");
    auto pr = std::make_unique<Program>();
    pr->B = B;
    // the feature list a VH_NET_BOUND walk reads: the recorded VH_NET_FEATURES program of the same (slot, batch), or - for sizing - a dry one
    std::vector<Feat> ext;
    if (mode == VH_NET_BOUND) {
        auto it = n->programs.find(std::make_tuple((int)VH_NET_FEATURES, slot, B));
        if (workspace) {
            VH_REQUIRE(it != n->programs.end() && it->second->plan, "vh_net_record: record the VH_NET_FEATURES program of slot %d, batch %d first", slot, B);
            for (const Feat& f : it->second->feats) {
                Feat g = f;
                g.f32.abs = it->second->base + f.f32.off;
                if (g.s8.ok()) g.s8.abs = it->second->base + f.s8.off;
                if (g.k.ok()) { g.k.abs = it->second->base + f.k.off; g.v.abs = it->second->base + f.v.off; }
                ext.push_back(g);
            }
        } else {
            Arena tmp; Program scratch_prog;
            n->A = &tmp; n->base = nullptr; n->emit = false; n->rc = VH_OK;
            walk(n, B, scratch_prog, VH_NET_FEATURES, nullptr);
            n->A = nullptr;
            if (n->rc != VH_OK) return n->rc;
            ext = scratch_prog.feats;
        }
    }
    Arena dry;
    n->A = &dry; n->base = nullptr; n->emit = false; n->rc = VH_OK;
    walk(n, B, *pr, mode, mode == VH_NET_BOUND ? &ext : nullptr);
    n->A = nullptr;
    if (n->rc != VH_OK) return n->rc;
    pr->peak_floats = dry.peak;
    if (!workspace) { *out = pr.release(); return VH_OK; }
    VH_REQUIRE(bytes >= (size_t)dry.peak * 4 && vh_aligned16(workspace), "vh_net_record: workspace too small (%zu < %lld bytes) or unaligned", bytes, dry.peak * 4LL);
    Arena real;
    Program rec; rec.B = B;
    int rc = vh_plan_begin(n->ctx);
    if (rc != VH_OK) return rc;
    n->A = &real; n->base = workspace; n->emit = true; n->rc = VH_OK;
    walk(n, B, rec, mode, mode == VH_NET_BOUND ? &ext : nullptr);
    n->A = nullptr; n->emit = false;
    if (n->rc != VH_OK) { (void)vh_plan_abort(n->ctx); return n->rc; }
    rc = vh_plan_end(n->ctx, &rec.plan);
    if (rc != VH_OK) return rc;
    rec.base = workspace; rec.peak_floats = real.peak;
    if (mode == VH_NET_BOUND) rec.feat_base = n->programs.at(std::make_tuple((int)VH_NET_FEATURES, slot, B))->base;
    *pr = rec;
    *out = pr.release();
    return VH_OK;
}

extern "C" size_t vh_net_workspace_bytes_mode(vh_net* n, int mode, int batch) {
    if (!n) return 0;
    Program* p = nullptr;
    if (net_build(n, mode, 0, batch, nullptr, 0, &p) != VH_OK) return 0;
    const size_t b = (size_t)p->peak_floats * 4;
    delete p;
    return b;
}
extern "C" size_t vh_net_workspace_bytes(vh_net* n, int batch) { return vh_net_workspace_bytes_mode(n, VH_NET_FULL, batch); }

extern "C" int vh_net_record_mode(vh_net* n, int mode, int slot, int batch, void* workspace, size_t bytes) {
    if (!n || !workspace) return vh_fail(VH_EINVAL, "vh_net_record: null argument");
    Program* p = nullptr;
    const int rc = net_build(n, mode, slot, batch, static_cast<float*>(workspace), bytes, &p);
    if (rc != VH_OK) return rc;
    const auto key = std::make_tuple(mode, slot, batch);
    auto it = n->programs.find(key);
    if (it != n->programs.end() && it->second->plan) (void)vh_plan_destroy(it->second->plan);
    n->programs[key].reset(p);
    if (mode == VH_NET_FEATURES) {
        // a VH_NET_BOUND program holds ABSOLUTE pointers into the workspace of the VH_NET_FEATURES program it was recorded against: a re-recorded
        // (possibly moved) FEATURES program leaves it reading the old addresses - drop it, so that vh_net_run_bound reports "no program
        // recorded" until the host records the BOUND program again
        auto bt = n->programs.find(std::make_tuple((int)VH_NET_BOUND, slot, batch));
        if (bt != n->programs.end()) { if (bt->second && bt->second->plan) (void)vh_plan_destroy(bt->second->plan); n->programs.erase(bt); }
    }
    return VH_OK;
}
extern "C" int vh_net_record(vh_net* n, int batch, void* workspace, size_t bytes) { return vh_net_record_mode(n, VH_NET_FULL, 0, batch, workspace, bytes); }

// One evaluation: D = net(src, x, sigma, geometry, cond).  All pointers are device fp32, contiguous, in the reference's layouts:
// src [rows][3 or 4][R][R], x [rows][3][R][R], sigma [rows], geometry [rows][source_label_dim] (NULL for an uncond net: zeros),
// cond [B][3][R][R] (super_res only), out [B][3][R][R]; rows = B * (dual_source ? 2 : 1).
static int net_run(vh_net* n, int mode, int slot, int batch, const float* src, const float* x, const float* sigma, const float* geometry, const float* cond,
                   const float* cond_noise, const float* const* feats_in, float* out) {
    auto it = n->programs.find(std::make_tuple(mode, slot, batch));
    VH_REQUIRE(it != n->programs.end() && it->second->plan, "vh_net: no program recorded for mode %d, slot %d, batch %d (vh_net_record / vh_net_record_mode)", mode, slot, batch);
    Program& p = *it->second;
    if (!n->prepared) return vh_fail(VH_ESTATE, "vh_net: a parameter was re-bound after vh_net_prepare: call vh_net_prepare (and record) again");
    if (mode == VH_NET_BOUND) {
        // the features this program reads in place must still be where they were when it was recorded
        auto ft = n->programs.find(std::make_tuple((int)VH_NET_FEATURES, slot, batch));
        if (ft == n->programs.end() || ft->second->base != p.feat_base)
            return vh_fail(VH_ESTATE, "vh_net_run_bound: the VH_NET_FEATURES program of slot %d, batch %d was re-recorded: record the VH_NET_BOUND program again", slot, batch);
    }
    hipStream_t s = n->ctx->stream;
    auto put = [&](const Buf& b, const float* from) -> int {
        if (!b.ok()) return VH_OK;
        hipError_t e = from ? hipMemcpyAsync(p.base + b.off, from, (size_t)b.n * 4, hipMemcpyDeviceToDevice, s) : hipMemsetAsync(p.base + b.off, 0, (size_t)b.n * 4, s);
        return e == hipSuccess ? VH_OK : vh_fail(VH_EHIP, "vh_net_run: %s", hipGetErrorString(e));
    };
    VH_REQUIRE(!p.src.ok() || src, "vh_net_run: this program reads src");
    VH_REQUIRE(!p.cond.ok() || cond, "vh_net_run: a super_res net needs the conditioning image (training/models.py:656)");
    const bool noisy = p.cond.ok() && n->cfg.noisy_sr != 0.0;
    VH_REQUIRE(!noisy || cond_noise, "vh_net_run: this super_res net has noisy_sr = %g: pass cond_noise, [batch][3][R][R] standard-normal draws (training/models.py:658)", n->cfg.noisy_sr);
    VH_REQUIRE(geometry || n->cfg.uncond, "vh_net_run: geometry is required for a conditional net (training/models.py:631)");
    VH_REQUIRE(!p.x.ok() || (x && out), "vh_net_run: x / out missing");
    VH_REQUIRE(p.feats_in.empty() || feats_in, "vh_net_run_inject: feature list missing");
    int rc;
    if ((rc = put(p.sigma, sigma)) != VH_OK || (rc = put(p.geometry, geometry)) != VH_OK || (rc = put(p.src, src)) != VH_OK ||
        (rc = put(p.x, x)) != VH_OK) return rc;
    if (noisy) {                                                   // cond + noisy_sr * eps, rounded like torch's two kernels (:658)
        vh_axpy_args ax{cond, cond_noise, (float)n->cfg.noisy_sr, p.base + p.cond.off, (size_t)p.cond.n};
        if ((rc = vh_axpy(n->ctx, &ax)) != VH_OK) return rc;
    } else if ((rc = put(p.cond, cond)) != VH_OK) return rc;
    for (size_t i = 0; i < p.feats_in.size(); ++i) {               // NCHW list -> the program's NHWC buffers (copied per call, like the reference's deepcopy :665)
        const Buf& b = p.feats_in[i];
        VH_REQUIRE(feats_in[i], "vh_net_run_inject: feature %zu is NULL", i);
        const int rows = batch * (n->cfg.dual_source ? 2 : 1);
        vh_layout_args la{feats_in[i], p.base + b.off, rows, b.c, (int)(b.n / ((long long)rows * b.c)), 0};
        if ((rc = vh_layout(n->ctx, &la)) != VH_OK) return rc;
    }
    if ((rc = vh_plan_run(n->ctx, p.plan)) != VH_OK) return rc;
    if (!p.D.ok()) return VH_OK;
    const hipError_t e = hipMemcpyAsync(out, p.base + p.D.off, (size_t)p.D.n * 4, hipMemcpyDeviceToDevice, s);
    return e == hipSuccess ? VH_OK : vh_fail(VH_EHIP, "vh_net_run: %s", hipGetErrorString(e));
}

// One evaluation: D = net(src, x, sigma, geometry, cond).  All pointers are device fp32, contiguous, in the reference's layouts:
// src [rows][3 or 4][R][R], x [rows][3][R][R], sigma [rows], geometry [rows][source_label_dim] (NULL for an uncond net: zeros),
// cond [B][3][R][R] (super_res only), out [B][3][R][R]; rows = B * (dual_source ? 2 : 1).
extern "C" int vh_net_run(vh_net* n, int batch, const float* src, const float* x, const float* sigma, const float* geometry, const float* cond,
                          const float* cond_noise, float* out) {
    if (!n || !x || !sigma || !out) return vh_fail(VH_EINVAL, "vh_net_run: null argument");
    return net_run(n, VH_NET_FULL, 0, batch, src, x, sigma, geometry, cond, cond_noise, nullptr, out);
}
// The two halves of an evaluation (training/models.py:664-667 / :676-683): the encoder into feature slot `slot`, and the UNet on that slot's
// features in place.  The encoder sees (src, sigma, geometry) only, so a sampler can run it for its next noise level on another stream
// (vh_ctx_set_stream between the calls) and once per level instead of once per call (vh_edm_sampler does exactly this).
extern "C" int vh_net_encode(vh_net* n, int slot, int batch, const float* src, const float* sigma, const float* geometry) {
    if (!n || !src || !sigma) return vh_fail(VH_EINVAL, "vh_net_encode: null argument");
    return net_run(n, VH_NET_FEATURES, slot, batch, src, nullptr, sigma, geometry, nullptr, nullptr, nullptr, nullptr);
}
extern "C" int vh_net_run_bound(vh_net* n, int slot, int batch, const float* src, const float* x, const float* sigma, const float* geometry, const float* cond,
                                const float* cond_noise, float* out) {
    if (!n || !x || !sigma || !out) return vh_fail(VH_EINVAL, "vh_net_run_bound: null argument");
    return net_run(n, VH_NET_BOUND, slot, batch, src, x, sigma, geometry, cond, cond_noise, nullptr, out);
}

// ---- the rest of NVPrecond.forward's protocol: feature lists in and out, the logvar head (training/models.py:664-670, 685-688)
extern "C" int vh_net_num_features(const vh_net* n) {
    if (!n) return 0;
    int c = 0;
    for (int g = 0; g < 2; ++g) for (const Block& b : (g ? n->unet.dec : n->unet.enc)) if (!b.conv && b.xattn) ++c;
    return c;
}
extern "C" int vh_net_feature_shape(const vh_net* n, int i, int* channels, int* res) {
    if (!n || !channels || !res) return vh_fail(VH_EINVAL, "vh_net_feature_shape: null argument");
    int c = 0;
    for (int g = 0; g < 2; ++g)
        for (const Block& b : (g ? n->unet.dec : n->unet.enc))
            if (!b.conv && b.xattn) { if (c == i) { *channels = b.cout; *res = b.res; return VH_OK; } ++c; }
    return vh_fail(VH_EINVAL, "vh_net_feature_shape: index %d out of range (%d features)", i, c);
}
extern "C" int vh_net_features(vh_net* n, int batch, const float* src, const float* sigma, const float* geometry, float* const* features_out) {
    if (!n || !src || !sigma || !features_out) return vh_fail(VH_EINVAL, "vh_net_features: null argument");
    VH_REQUIRE(n->has_enc, "vh_net_features: an uncond net has no encoder (its feature list is all zeros, training/models.py:727-736)");
    int rc = net_run(n, VH_NET_FEATURES, 0, batch, src, nullptr, sigma, geometry, nullptr, nullptr, nullptr, nullptr);
    if (rc != VH_OK) return rc;
    const Program& p = *n->programs.at(std::make_tuple((int)VH_NET_FEATURES, 0, batch));
    const int rows = batch * (n->cfg.dual_source ? 2 : 1);
    for (size_t i = 0; i < p.feats.size(); ++i) {
        VH_REQUIRE(features_out[i], "vh_net_features: output %zu is NULL", i);
        const Buf& b = p.feats[i].f32;
        vh_layout_args la{p.base + b.off, features_out[i], rows, b.c, (int)(b.n / ((long long)rows * b.c)), 1};
        if ((rc = vh_layout(n->ctx, &la)) != VH_OK) return rc;
    }
    return VH_OK;
}
extern "C" int vh_net_run_inject(vh_net* n, int batch, const float* src, const float* x, const float* sigma, const float* geometry, const float* cond,
                                 const float* cond_noise, const float* const* features_in, float* out) {
    if (!n || !x || !sigma || !out || !features_in) return vh_fail(VH_EINVAL, "vh_net_run_inject: null argument");
    return net_run(n, VH_NET_INJECT, 0, batch, src, x, sigma, geometry, cond, cond_noise, features_in, out);
}
extern "C" int vh_net_logvar(vh_net* n, int batch, const float* sigma, float* logvar) {
    if (!n || !sigma || !logvar || batch <= 0) return vh_fail(VH_EINVAL, "vh_net_logvar: bad argument");
    VH_REQUIRE(n->prepared, "vh_net_logvar: call vh_net_prepare first");
    const Weight& wl = n->W.at("logvar_linear.weight");
    vh_embed_args a{};
    a.sigma = sigma; a.sigma_stride = n->cfg.dual_source ? 2 : 1; a.time_scale = 1.f; a.geometry = nullptr; a.label_dim = 0; a.geometry_scale = 0.f;
    a.freqs = P(n, "logvar_fourier.freqs"); a.phases = P(n, "logvar_fourier.phases"); a.cnoise = n->cfg.logvar_channels;
    a.w_noise = wl.wt; a.w_noise_kpad = wl.k_pad; a.w_label = nullptr; a.w_label_kpad = 0; a.label_balance = 0.f; a.rows = batch; a.cemb = 1; a.raw = 1; a.emb = logvar;
    return vh_embed(n->ctx, &a);
}

// ================================================================== edm_sampler (generate_images.py:43-118) over vh_net evaluations
// vivid_amd/sampler.py restated: same schedule arithmetic (fp32, rounded where torch rounds), same call order, same kernels
// (vh_sampler_step, vh_axpy), same scheduling (encoder once per noise level, one level ahead on a second stream; the guidance net on a
// third for small evaluations) - so that its samples equal vivid_amd.edm_sampler's bit for bit (tests/test_hip_net_c.py).
namespace {
#pragma clang fp contract(off)       // the schedule below is fp32 arithmetic with torch's roundings: no fused multiply-adds

size_t align_up(size_t v) { return (v + 63) / 64 * 64; }       // in floats
struct SamplerWs { size_t xbuf, dbuf, tt, total; };               // element counts
SamplerWs sampler_ws(const vh_net* n, int B) {
    const int R = n->cfg.img_resolution, rm = n->cfg.dual_source ? 2 : 1;
    SamplerWs w;
    w.xbuf = align_up((size_t)B * rm * 3 * R * R); w.dbuf = align_up((size_t)B * 3 * R * R); w.tt = align_up((size_t)B * rm);
    w.total = 4 * w.xbuf + 6 * w.dbuf + 3 * w.tt;      // x_hat, x_next, x_corr, churn noise | D, ref, Dp, refp, d_cur, cond noise | tt, tt_enc[2]
    return w;
}
bool has_prog(const vh_net* n, int mode, int slot, int B) {
    auto it = n->programs.find(std::make_tuple(mode, slot, B));
    return it != n->programs.end() && it->second->plan;
}
int hip_rc(hipError_t e, const char* what) { return e == hipSuccess ? VH_OK : vh_fail(VH_EHIP, "vh_edm_sampler: %s: %s", what, hipGetErrorString(e)); }
}  // namespace

extern "C" size_t vh_edm_sampler_workspace_bytes(const vh_net* n, int batch) { return (n && batch > 0) ? sampler_ws(n, batch).total * 4 : 0; }

extern "C" int vh_edm_sampler(vh_net* net, vh_net* gnet, const vh_sampler_config* cp, int B, const float* src, const float* noise, const float* labels,
                              const float* cond, void* workspace, size_t workspace_bytes, float* out) {
    if (!net || !cp || !src || !noise || !workspace || !out || B <= 0) return vh_fail(VH_EINVAL, "vh_edm_sampler: null argument");
    const vh_sampler_config c = *cp;
    const vh_net_config& nc = net->cfg;
    VH_REQUIRE(c.num_steps >= 2, "vh_edm_sampler: num_steps must be >= 2 (the schedule divides by num_steps - 1, generate_images.py:69)");
    const bool guided = c.guidance != 1.0 && gnet != nullptr;
    VH_REQUIRE(c.guidance == 1.0 || gnet, "vh_edm_sampler: guidance != 1 needs a guidance net");
    const SamplerWs W = sampler_ws(net, B);
    VH_REQUIRE(workspace_bytes >= W.total * 4 && vh_aligned16(workspace), "vh_edm_sampler: workspace too small (%zu < %zu bytes) or unaligned", workspace_bytes, W.total * 4);
    const int R = nc.img_resolution, rm = nc.dual_source ? 2 : 1, rows = B * rm, N = c.num_steps;
    const size_t xn = (size_t)rows * 3 * R * R, dn = (size_t)B * 3 * R * R;
    const bool churn_any = c.S_churn > 0.0;
    const bool sr_noise = nc.super_res && nc.noisy_sr != 0.0;
    VH_REQUIRE(!(churn_any || sr_noise) || c.randn, "vh_edm_sampler: S_churn > 0 / a super_res net with noisy_sr != 0 draw noise: give vh_sampler_config.randn");
    VH_REQUIRE(!nc.super_res || cond, "vh_edm_sampler: a super_res net needs the conditioning image");
    float* base = static_cast<float*>(workspace);
    float* X[3] = {base, base + W.xbuf, base + 2 * W.xbuf};
    float* eps = base + 3 * W.xbuf;
    float* Dd = eps + W.xbuf; float* Rf = Dd + W.dbuf; float* Dp = Rf + W.dbuf; float* Rp = Dp + W.dbuf; float* dcur = Rp + W.dbuf; float* cn = dcur + W.dbuf;
    float* tt = cn + W.dbuf; float* tte[2] = {tt + W.tt, tt + 2 * W.tt};
    vh_ctx* ctx = net->ctx;
    const hipStream_t main_s = ctx->stream;
    int rc;

    // ---- noise levels (:68-70), in fp32 like the reference (and vivid_amd/sampler.py:165-167)
    std::vector<float> t(N + 1);
    if (c.t_steps) { for (int i = 0; i <= N; ++i) t[i] = c.t_steps[i]; }
    else {
        const float a = (float)std::pow(c.sigma_max, 1.0 / c.rho), d = (float)(std::pow(c.sigma_min, 1.0 / c.rho) - std::pow(c.sigma_max, 1.0 / c.rho));
        for (int i = 0; i < N; ++i) {
            const float frac = (float)i / (float)(N - 1);
            const float prod = frac * d;
            const float v = a + prod;
            t[i] = powf(v, (float)c.rho);
        }
        t[N] = 0.f;
    }
    // per step: churned level t_hat (:78-84) and the noise level of every denoiser call in call order (Euler call at t_hat, Heun probe at t_next)
    const float gamma = (float)std::min(c.S_churn / N, std::sqrt(2.0) - 1.0);
    std::vector<float> that(N), levels;
    std::vector<char> churned(N);
    for (int i = 0; i < N; ++i) {
        churned[i] = churn_any && c.S_min <= (double)t[i] && (double)t[i] <= c.S_max;
        if (churned[i]) { const float gp = gamma * t[i]; that[i] = t[i] + gp; } else that[i] = t[i];
        levels.push_back(that[i]);
        if (i < N - 1) levels.push_back(t[i + 1]);
    }

    // ---- which programs carry the net's calls
    const bool split_ok = !nc.uncond && has_prog(net, VH_NET_FEATURES, 0, B) && has_prog(net, VH_NET_BOUND, 0, B);
    const bool nte = nc.no_time_enc != 0 && split_ok;                                                   // :52-53: features once, re-used by every call
    const bool pipe = !nc.no_time_enc && split_ok && has_prog(net, VH_NET_FEATURES, 1, B) && has_prog(net, VH_NET_BOUND, 1, B);
    VH_REQUIRE(nte || pipe || has_prog(net, VH_NET_FULL, 0, B), "vh_edm_sampler: record VH_NET_FULL (or VH_NET_FEATURES + VH_NET_BOUND of both slots) for batch %d first", B);
    VH_REQUIRE(!guided || has_prog(gnet, VH_NET_FULL, 0, B), "vh_edm_sampler: record the guidance net's VH_NET_FULL program for batch %d first", B);
    bool overlap = guided && gnet != net && (c.guidance_overlap > 0 || (c.guidance_overlap < 0 && (long long)rows * R * R <= 32LL * 2 * 64 * 64));
    if ((pipe || overlap) && !net->ev_main) {
        if ((rc = hip_rc(hipStreamCreateWithFlags(&net->side_stream, hipStreamNonBlocking), "stream")) != VH_OK) return rc;
        if ((rc = hip_rc(hipStreamCreateWithFlags(&net->gside_stream, hipStreamNonBlocking), "stream")) != VH_OK) return rc;
        for (hipEvent_t* e : {&net->ev_main, &net->ev_enc[0], &net->ev_enc[1], &net->ev_g})
            if ((rc = hip_rc(hipEventCreateWithFlags(e, hipEventDisableTiming), "event")) != VH_OK) return rc;
    }
    auto fill = [&](float* dst, float v, size_t n_) { vh_axpy_args a{nullptr, nullptr, v, dst, n_}; return vh_axpy(ctx, &a); };

    // ---- encoder features for the sequence of levels (vivid_amd.sampler._FeaturePipeline)
    int cur_slot = -1, ahead_slot = -1, next_slot = 0; float cur_level = 0.f, ahead_level = 0.f; bool have_cur = false, have_ahead = false;
    auto launch_enc = [&](float level, int* slot_out) -> int {
        int r;
        if ((r = hip_rc(hipEventRecord(net->ev_main, main_s), "record")) != VH_OK) return r;           // everything enqueued so far, incl. the UNet call still reading this slot
        if ((r = hip_rc(hipStreamWaitEvent(net->side_stream, net->ev_main, 0), "wait")) != VH_OK) return r;
        const int slot = next_slot; next_slot ^= 1;
        ctx->stream = net->side_stream;
        r = fill(tte[slot], level, (size_t)rows);
        if (r == VH_OK) r = net_run(net, VH_NET_FEATURES, slot, B, src, nullptr, tte[slot], labels, nullptr, nullptr, nullptr, nullptr);
        ctx->stream = main_s;
        if (r != VH_OK) return r;
        if ((r = hip_rc(hipEventRecord(net->ev_enc[slot], net->side_stream), "record")) != VH_OK) return r;
        *slot_out = slot;
        return VH_OK;
    };
    auto features_for = [&](size_t k, int* slot_out) -> int {
        const float level = levels[k];
        int r;
        if (!have_cur || cur_level != level) {
            int slot;
            if (have_ahead && ahead_level == level) slot = ahead_slot;
            else if ((r = launch_enc(level, &slot)) != VH_OK) return r;
            if ((r = hip_rc(hipStreamWaitEvent(main_s, net->ev_enc[slot], 0), "wait")) != VH_OK) return r;
            cur_slot = slot; cur_level = level; have_cur = true; have_ahead = false;
        }
        if (!have_ahead)
            for (size_t j = k + 1; j < levels.size(); ++j)
                if (levels[j] != level) { if ((r = launch_enc(levels[j], &ahead_slot)) != VH_OK) return r; ahead_level = levels[j]; have_ahead = true; break; }
        *slot_out = cur_slot;
        return VH_OK;
    };
    if (nte) {      // features = net(src, 0, ones, labels, cond, return_features=True) once (:52-53); the encoder ignores sigma (time_scale 0)
        if ((rc = fill(tte[0], 1.f, (size_t)rows)) != VH_OK) return rc;
        if ((rc = net_run(net, VH_NET_FEATURES, 0, B, src, nullptr, tte[0], labels, nullptr, nullptr, nullptr, nullptr)) != VH_OK) return rc;
    }

    // ---- the `denoise` closure (:55-62): D = net(src, x, t, labels, cond, features), ref = gnet(src, x, t)
    size_t call_k = 0;
    auto denoise = [&](const float* x, float level, float* D_out, float* ref_out) -> int {
        int r;
        if ((r = fill(tt, level, (size_t)rows)) != VH_OK) return r;
        int slot = 0;
        if (pipe && (r = features_for(call_k, &slot)) != VH_OK) return r;
        ++call_k;
        if (guided && overlap) {
            if ((r = hip_rc(hipEventRecord(net->ev_main, main_s), "record")) != VH_OK) return r;       // x and tt were produced on the main stream
            if ((r = hip_rc(hipStreamWaitEvent(net->gside_stream, net->ev_main, 0), "wait")) != VH_OK) return r;
            vh_ctx* gc = gnet->ctx; const hipStream_t gs = gc->stream;
            gc->stream = net->gside_stream;
            r = net_run(gnet, VH_NET_FULL, 0, B, src, x, tt, nullptr, nullptr, nullptr, nullptr, ref_out);
            gc->stream = gs;
            if (r != VH_OK) return r;
            if ((r = hip_rc(hipEventRecord(net->ev_g, net->gside_stream), "record")) != VH_OK) return r;
        }
        const float* cnp = nullptr;
        if (sr_noise) { c.randn(c.randn_user, cn, dn, (void*)main_s); cnp = cn; }
        if (pipe || nte) r = net_run(net, VH_NET_BOUND, slot, B, src, x, tt, labels, cond, cnp, nullptr, D_out);
        else r = net_run(net, VH_NET_FULL, 0, B, src, x, tt, labels, cond, cnp, nullptr, D_out);
        if (r != VH_OK) return r;
        if (guided && overlap) return hip_rc(hipStreamWaitEvent(main_s, net->ev_g, 0), "wait");
        if (guided) {
            vh_ctx* gc = gnet->ctx; const hipStream_t gs = gc->stream;
            gc->stream = main_s;
            r = net_run(gnet, VH_NET_FULL, 0, B, src, x, tt, nullptr, nullptr, nullptr, nullptr, ref_out);
            gc->stream = gs;
        }
        return r;
    };
    auto step = [&](const float* x_hat, const float* x_probe, const float* D, const float* ref, float th, float tn, float* x_next) -> int {
        vh_sampler_step_args a{};
        a.x_hat = x_hat; a.x_probe = x_probe; a.d_cond = D; a.d_ref = guided ? ref : nullptr; a.guidance = (float)c.guidance; a.d_cur = dcur;
        a.t_hat = th; a.t_next = tn; a.rows = B; a.row_mul = rm; a.row_elems = (size_t)3 * R * R; a.x_next = x_next;
        return vh_sampler_step(ctx, &a);
    };

    // ---- main loop (:72-114)
    int ix = 0;                                                    // X[ix] holds x_cur
    { vh_axpy_args a{nullptr, noise, t[0], X[ix], xn}; if ((rc = vh_axpy(ctx, &a)) != VH_OK) return rc; }       // x0 = noise * t0
    for (int i = 0; i < N; ++i) {
        const float t_cur = t[i], t_next = t[i + 1], t_hat = that[i];
        float* x_hat = X[ix];
        if (churned[i]) {                                          // x_hat = x_cur + sqrt(t_hat^2 - t_cur^2) * S_noise * eps, in place (:81)
            const float th2 = t_hat * t_hat, tc2 = t_cur * t_cur;
            const float df = th2 - tc2;
            const float sq = sqrtf(df);
            c.randn(c.randn_user, eps, xn, (void*)main_s);
            vh_axpy_args a{x_hat, eps, (float)((double)sq * c.S_noise), x_hat, xn};
            if ((rc = vh_axpy(ctx, &a)) != VH_OK) return rc;
        }
        float* x_nx = X[(ix + 1) % 3];
        if ((rc = denoise(x_hat, t_hat, Dd, Rf)) != VH_OK) return rc;
        if ((rc = step(x_hat, nullptr, Dd, Rf, t_hat, t_next, x_nx)) != VH_OK) return rc;                      // Euler (:93-98)
        if (i < N - 1) {                                                                                       // Heun correction (:104-111)
            float* x_co = X[(ix + 2) % 3];
            if ((rc = denoise(x_nx, t_next, Dp, Rp)) != VH_OK) return rc;
            if ((rc = step(x_hat, x_nx, Dp, Rp, t_hat, t_next, x_co)) != VH_OK) return rc;
            ix = (ix + 2) % 3;
        } else ix = (ix + 1) % 3;
    }
    // the even rows of the final state (:116-118)
    return hip_rc(hipMemcpy2DAsync(out, (size_t)3 * R * R * 4, X[ix], (size_t)rm * 3 * R * R * 4, (size_t)3 * R * R * 4, (size_t)B, hipMemcpyDeviceToDevice, main_s), "copy out");
}

extern "C" int vh_net_destroy(vh_net* n) {
    if (!n) return VH_OK;
    if (n->side_stream) (void)hipStreamDestroy(n->side_stream);
    if (n->gside_stream) (void)hipStreamDestroy(n->gside_stream);
    for (hipEvent_t e : {n->ev_main, n->ev_enc[0], n->ev_enc[1], n->ev_g}) if (e) (void)hipEventDestroy(e);
    for (auto& kv : n->programs) if (kv.second && kv.second->plan) (void)vh_plan_destroy(kv.second->plan);
    delete n;
    return VH_OK;
}
