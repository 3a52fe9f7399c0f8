// Plan builder + weight packer for the STTN hot path (see sttn_plan.h).
#include "sttn_plan.h"
#include "gather_gemm.h"
#include <assert.h>
#include <math.h>
#include <stdexcept>
#include <stdlib.h>

namespace vsr {

static int envInt(const char* name, int dflt)
{
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

const Tuning& Tuning::get(int precision)
{
    // defaults = best measured on the 1080p bench (interleaved A/B, profiles/): 128x64 tiles (3 workgroups
    // per CU) win in both modes; 128x128 is 8 % slower with the split-half kernels and 4 % with fp32
    static const Tuning t[2] = {
        [] {
            Tuning x;
            x.convTile = envInt("VSR_CONV_TILE", VSR_TILE_128x64);
            if (x.convTile == VSR_TILE_256x128) x.convTile = VSR_TILE_128x64;      // the 8-wave tile exists in the fp16-operand kernel only
            x.qkTile = envInt("VSR_QK_TILE", VSR_TILE_128x64);
            x.qkvTile = envInt("VSR_QKV_TILE", x.convTile);
            x.pvTile = envInt("VSR_PV_TILE", VSR_TILE_128x64);
            x.pvSplitChunks = envInt("VSR_PV_SPLIT_CHUNKS", 50);
            x.convChannelMajor = envInt("VSR_CONV_KORDER", 1);
            // the fused path needs the kernels it was written into: v3 for the scores, v1 (+ A_EXP) for P.V -- the defaults
            x.fuseSoftmax = envInt("VSR_FUSE_SOFTMAX", 1) && envInt("VSR_GG_VARIANT", 3) == 3 && envInt("VSR_PV_VARIANT", 1) == 1;
            x.outConvBlocked = envInt("VSR_OUT_CONV_BLOCKED", 1);
            x.trimLastBlock = envInt("VSR_TRIM_LAST_BLOCK", 1);
            x.shareQkv0 = envInt("VSR_QKV0_SHARED", 0);
            return x;
        }(),
        [] {
            Tuning x;
            x.convTile = envInt("VSR_CONV_TILE", VSR_TILE_128x64);
            x.qkTile = envInt("VSR_QK_TILE", VSR_TILE_128x64);
            x.qkvTile = envInt("VSR_QKV_TILE", x.convTile);
            x.pvTile = envInt("VSR_PV_TILE", VSR_TILE_128x64);
            x.pvSplitChunks = envInt("VSR_PV_SPLIT_CHUNKS", 50);
            x.convChannelMajor = envInt("VSR_CONV_KORDER", 1);
            x.fuseSoftmax = 0;      // split-format tensors: the probabilities are a GEMM operand in that format, k_softmax_rows writes it
            x.outConvBlocked = envInt("VSR_OUT_CONV_BLOCKED", 1);
            x.trimLastBlock = envInt("VSR_TRIM_LAST_BLOCK", 1);
            x.shareQkv0 = envInt("VSR_QKV0_SHARED", 0);
            return x;
        }()};
    return t[precision ? 1 : 0];
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t rup(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

Geometry Geometry::make(int variant)
{
    Geometry g{};
    g.variant = variant;
    g.channels = 256;
    g.blocks = 8;
    g.nscales = 4;
    g.neighborStride = 5;
    g.refLength = 10;
    if (variant == 0) { // auto_sttn.py:69 patchsize, sttn_auto_inpaint.py:39 model input
        g.modelW = 640; g.modelH = 120;
        const int pw[4] = {80, 32, 10, 5}, ph[4] = {15, 6, 5, 3};
        for (int i = 0; i < 4; ++i) { g.patchW[i] = pw[i]; g.patchH[i] = ph[i]; }
    } else {            // network_sttn.py:69 patchsize, sttn_det_inpaint.py model input 432x240
        g.modelW = 432; g.modelH = 240;
        const int pw[4] = {108, 36, 18, 9}, ph[4] = {60, 20, 10, 5};
        for (int i = 0; i < 4; ++i) { g.patchW[i] = pw[i]; g.patchH[i] = ph[i]; }
    }
    g.featW = g.modelW / 4;
    g.featH = g.modelH / 4;
    return g;
}

// ------------------------------------------------------------------------------------
// Model
// ------------------------------------------------------------------------------------
Model::Model(int variant) : g(Geometry::make(variant)) { blk.resize(g.blocks); }

std::vector<std::string> Model::expected_keys(int variant)
{
    (void)variant; // both generators share the parameter structure
    std::vector<std::string> k;
    for (int i = 0; i < 8; ++i) {
        const std::string p = "transformer." + std::to_string(i) + ".";
        for (const char* e : {"attention.query_embedding", "attention.value_embedding", "attention.key_embedding",
                              "attention.output_linear.0", "feed_forward.conv.0", "feed_forward.conv.2"}) {
            k.push_back(p + e + ".weight");
            k.push_back(p + e + ".bias");
        }
    }
    for (const char* e : {"encoder.0", "encoder.2", "encoder.4", "encoder.6", "decoder.0.conv", "decoder.2",
                          "decoder.4.conv", "decoder.6"}) {
        k.push_back(std::string(e) + ".weight");
        k.push_back(std::string(e) + ".bias");
    }
    return k;
}

bool Model::set_param(const std::string& name, const float* data, const int64_t* shape, int ndim, std::string& err)
{
    bool known = false;
    for (const auto& k : expected_keys(g.variant))
        if (k == name) { known = true; break; }
    if (!known) { err = "unexpected key in state_dict: " + name; return false; }
    Raw r;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) { r.shape.push_back(shape[i]); n *= shape[i]; }
    r.v.assign(data, data + n);
    raw_[name] = std::move(r);
    ready_ = false;
    return true;
}

// [cout][cin][kh][kw] -> [cout][K], k = (ky*kw + kx)*cin + ci ; K padded up to a multiple of 32
bool Model::pack_conv(const std::string& key, ConvW& cw, int /*cinPad*/, std::string& err)
{
    auto wi = raw_.find(key + ".weight"), bi = raw_.find(key + ".bias");
    if (wi == raw_.end() || bi == raw_.end()) { err = "missing key in state_dict: " + key; return false; }
    const Raw& w = wi->second;
    if (w.shape.size() != 4) { err = "bad weight rank: " + key; return false; }
    const int cout = (int)w.shape[0], cin = (int)w.shape[1], kh = (int)w.shape[2], kw = (int)w.shape[3];
    if ((int64_t)bi->second.v.size() != cout) { err = "bad bias shape: " + key; return false; }
    const int Kreal = kh * kw * cin;
    const int K = (int)rup(Kreal, VSR_GG_KC);
    cw.cout = cout;
    cw.K = K;
    cw.w = (int64_t)packed.size();
    packed.resize(packed.size() + (size_t)rup((int64_t)cout * K, 32), 0.f);
    float* dst = packed.data() + cw.w;
    // K order: (tap, ci) for the 3-channel first layer (it is consumed through an explicit im2col);
    // otherwise (channel chunk of 32, tap, ci%32) when convChannelMajor: the 9 taps of one channel
    // chunk are contracted back to back, so the gathered 128-byte lines are re-used from L1/L2
    // while they are hot instead of once per tap pass.  Must mirror Plan::tColsConv.
    const bool chanMajor = Tuning::get().convChannelMajor && (cin % VSR_GG_KC == 0);
    const int taps = kh * kw;
    for (int n = 0; n < cout; ++n)
        for (int ci = 0; ci < cin; ++ci)
            for (int ky = 0; ky < kh; ++ky)
                for (int kx = 0; kx < kw; ++kx) {
                    const int tap = ky * kw + kx;
                    const int k = chanMajor ? ((ci / VSR_GG_KC) * taps + tap) * VSR_GG_KC + (ci % VSR_GG_KC) : tap * cin + ci;
                    dst[(int64_t)n * K + k] = w.v[(((int64_t)n * cin + ci) * kh + ky) * kw + kx];
                }
    cw.b = (int64_t)packed.size();
    packed.resize(packed.size() + (size_t)rup(cout, 32), 0.f);
    for (int n = 0; n < cout; ++n) packed[cw.b + n] = bi->second.v[n];
    return true;
}

// A 3x3 conv with very few output channels as a GEMM over bh x bw output blocks: one GEMM row per block, N = bh*bw*cout
// columns (dy, dx, c), K = the (bh+2) x (bw+2) window the block's pixels share x cin; the taps outside a pixel's own 3x3 are zero
// rows of the packed matrix.  For the 64 -> 3 output conv of the decoder (auto_sttn.py:93) with 2x4 blocks: 24 of 32 MFMA
// columns carry outputs instead of 3, for 2.7x the multiply-adds per pixel -- a third of the matrix-core time (the LaMa plan
// does the same with its 7x7 output conv, lama_plan.cpp).  K order mirrors Plan::buildWindow's window table.
bool Model::pack_conv_blocked(const std::string& key, ConvW& cw, int bh, int bw, std::string& err)
{
    auto wi = raw_.find(key + ".weight"), bi = raw_.find(key + ".bias");
    if (wi == raw_.end() || bi == raw_.end()) { err = "missing key in state_dict: " + key; return false; }
    const Raw& w = wi->second;
    const int cout = (int)w.shape[0], cin = (int)w.shape[1], kh = (int)w.shape[2], kw = (int)w.shape[3];
    if (kh != 3 || kw != 3 || cin % VSR_GG_KC) { err = "blocked conv: 3x3 over whole channel chunks only: " + key; return false; }
    const int wh = bh + 2, ww = bw + 2, taps = wh * ww;
    const int K = taps * cin, N = bh * bw * cout;
    cw.cout = N;
    cw.K = K;
    cw.w = (int64_t)packed.size();
    packed.resize(packed.size() + (size_t)rup((int64_t)N * K, 32), 0.f);
    float* dst = packed.data() + cw.w;
    const bool chanMajor = Tuning::get().convChannelMajor != 0;
    for (int dy = 0; dy < bh; ++dy)
        for (int dx = 0; dx < bw; ++dx)
            for (int c = 0; c < cout; ++c) {
                const int n = (dy * bw + dx) * cout + c;
                for (int ci = 0; ci < cin; ++ci)
                    for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx) {
                            const int tap = (dy + ky) * ww + (dx + kx);     // window position of this pixel's tap
                            const int k = chanMajor ? ((ci / VSR_GG_KC) * taps + tap) * VSR_GG_KC + (ci % VSR_GG_KC) : tap * cin + ci;
                            dst[(int64_t)n * K + k] = w.v[(((int64_t)c * cin + ci) * 3 + ky) * 3 + kx];
                        }
            }
    cw.b = (int64_t)packed.size();
    packed.resize(packed.size() + (size_t)rup(N, 32), 0.f);
    for (int n = 0; n < N; ++n) packed[cw.b + n] = bi->second.v[n % cout];
    return true;
}

bool Model::pack(std::string& err)
{
    packed.clear();
    ready_ = false;
    for (const auto& k : expected_keys(g.variant))
        if (!raw_.count(k)) { err = "missing key in state_dict: " + k; return false; }
    static const char* encKeys[4] = {"encoder.0", "encoder.2", "encoder.4", "encoder.6"};
    static const char* decKeys[4] = {"decoder.0.conv", "decoder.2", "decoder.4.conv", "decoder.6"};
    static const int encShape[4][2] = {{64, 3}, {64, 64}, {128, 64}, {256, 128}};
    static const int decShape[4][2] = {{128, 256}, {64, 128}, {64, 64}, {3, 64}};
    auto check = [&](const std::string& key, int cout, int cin, int ksz) {
        const Raw& w = raw_[key + ".weight"];
        if (w.shape.size() != 4 || w.shape[0] != cout || w.shape[1] != cin || w.shape[2] != ksz || w.shape[3] != ksz) {
            err = "shape mismatch for " + key + ".weight";
            return false;
        }
        return true;
    };
    for (int i = 0; i < 4; ++i) {
        if (!check(encKeys[i], encShape[i][0], encShape[i][1], 3)) return false;
        if (!pack_conv(encKeys[i], enc[i], 0, err)) return false;
    }
    for (int i = 0; i < 4; ++i) {
        if (!check(decKeys[i], decShape[i][0], decShape[i][1], 3)) return false;
        if (!pack_conv(decKeys[i], dec[i], 0, err)) return false;
    }
    if (!pack_conv_blocked(decKeys[3], dec4blk, kOutBlkH, kOutBlkW, err)) return false;
    const int C = g.channels;
    for (int b = 0; b < g.blocks; ++b) {
        const std::string p = "transformer." + std::to_string(b) + ".";
        // fused QKV 1x1: rows [0,C) query, [C,2C) key, [2C,3C) value (auto_sttn.py:172-174)
        ConvW q, k, v;
        for (const char* e : {"attention.query_embedding", "attention.key_embedding", "attention.value_embedding"})
            if (!check(p + e, C, C, 1)) return false;
        if (!pack_conv(p + "attention.query_embedding", q, 0, err)) return false;
        if (!pack_conv(p + "attention.key_embedding", k, 0, err)) return false;
        if (!pack_conv(p + "attention.value_embedding", v, 0, err)) return false;
        ConvW& f = blk[b].qkv;
        f.cout = 3 * C;
        f.K = C;
        f.w = (int64_t)packed.size();
        packed.resize(packed.size() + (size_t)3 * C * C, 0.f);
        f.b = (int64_t)packed.size();
        packed.resize(packed.size() + (size_t)rup(3 * C, 32), 0.f);
        const ConvW* src[3] = {&q, &k, &v};
        for (int j = 0; j < 3; ++j) {
            for (int64_t i = 0; i < (int64_t)C * C; ++i) packed[f.w + (int64_t)j * C * C + i] = packed[src[j]->w + i];
            for (int i = 0; i < C; ++i) packed[f.b + j * C + i] = packed[src[j]->b + i];
        }
        if (!check(p + "attention.output_linear.0", C, C, 3)) return false;
        if (!pack_conv(p + "attention.output_linear.0", blk[b].out, 0, err)) return false;
        if (!check(p + "feed_forward.conv.0", C, C, 3)) return false;
        if (!pack_conv(p + "feed_forward.conv.0", blk[b].ffn1, 0, err)) return false;
        if (!check(p + "feed_forward.conv.2", C, C, 3)) return false;
        if (!pack_conv(p + "feed_forward.conv.2", blk[b].ffn2, 0, err)) return false;
    }
    ready_ = true;
    return true;
}

// ------------------------------------------------------------------------------------
// Plan
// ------------------------------------------------------------------------------------
static void tileDims(int cfg, int& BM, int& BN)
{
    if (cfg == VSR_TILE_128x128) { BM = 128; BN = 128; }
    else if (cfg == VSR_TILE_128x64) { BM = 128; BN = 64; }
    else if (cfg == VSR_TILE_256x64) { BM = 256; BN = 64; }
    else if (cfg == VSR_TILE_256x128) { BM = 256; BN = 128; }
    else { BM = 256; BN = 32; }
}
int n64Tile()
{
    static const int t = envInt("VSR_N64_TILE", VSR_TILE_128x64);
    return t;
}

// N = 64 convs (decoder, encoder): 128x64 since round 3 (decoder 99 -> 104 TF; VSR_N64_TILE=2 restores 256x64)
int Plan::pickTile(int N) const { return N <= 32 ? VSR_TILE_256x32 : (N <= 64 ? n64Tile() : tu_.convTile); }

static void checkFits(int64_t v)
{
    if (v > 2147483647LL || v < -2147483648LL) throw std::runtime_error("offset table entry exceeds int32");
}

int PlanBuilder::table(const std::string& key, std::vector<int32_t>&& v)
{
    auto it = tableKey_.find(key);
    if (it != tableKey_.end()) return it->second;
    tables.push_back(std::move(v));
    const int id = (int)tables.size() - 1;
    tableKey_[key] = id;
    return id;
}

void PlanBuilder::need(int buf, int64_t elems)
{
    if ((int)bufElems.size() <= buf) bufElems.resize(buf + 1, 0);
    if (bufElems[buf] < elems) bufElems[buf] = elems;
}

static std::string idsKey(const std::vector<int>& ids)
{
    std::string s;
    for (int i : ids) { s += std::to_string(i); s += ','; }
    return s;
}

int PlanBuilder::tRowsAct(const Act& a, const std::vector<int>& ids, int oh, int ow, int stride, int padTo, int64_t add)
{
    const std::string key = "RA:" + std::to_string(a.buf) + ":" + std::to_string(a.halo) + ":" + std::to_string(a.H) +
                            ":" + std::to_string(a.W) + ":" + std::to_string(a.C) + ":" + std::to_string(oh) + ":" +
                            std::to_string(ow) + ":" + std::to_string(stride) + ":" + std::to_string(add) + ":" +
                            std::to_string(padTo) + ":" + idsKey(ids);
    auto it = tableKey_.find(key);
    if (it != tableKey_.end()) return it->second;
    std::vector<int32_t> v;
    const int64_t M = (int64_t)ids.size() * oh * ow;
    v.reserve((size_t)rup(M, padTo));
    for (size_t ti = 0; ti < ids.size(); ++ti)
        for (int y = 0; y < oh; ++y)
            for (int x = 0; x < ow; ++x) {
                const int64_t o = a.pix(ids[ti], y * stride, x * stride) + add;
                checkFits(o);
                v.push_back((int32_t)o);
            }
    const int32_t first = v.empty() ? 0 : v[0];
    while ((int64_t)v.size() % padTo) v.push_back(first);
    return table(key, std::move(v));
}

int PlanBuilder::tRowsActRect(const Act& a, const std::vector<int>& ids, int ylo, int yhi, int xlo, int xhi, int padTo)
{
    const std::string key = "RR:" + std::to_string(a.buf) + ":" + std::to_string(a.halo) + ":" + std::to_string(a.H) + ":" +
                            std::to_string(a.W) + ":" + std::to_string(a.C) + ":" + std::to_string(ylo) + "-" + std::to_string(yhi) + ":" +
                            std::to_string(xlo) + "-" + std::to_string(xhi) + ":" + std::to_string(padTo) + ":" + idsKey(ids);
    auto it = tableKey_.find(key);
    if (it != tableKey_.end()) return it->second;
    std::vector<int32_t> v;
    v.reserve((size_t)rup((int64_t)ids.size() * (yhi - ylo) * (xhi - xlo), padTo));
    for (size_t ti = 0; ti < ids.size(); ++ti)
        for (int y = ylo; y < yhi; ++y)
            for (int x = xlo; x < xhi; ++x) {
                const int64_t o = a.pix(ids[ti], y, x);
                checkFits(o);
                v.push_back((int32_t)o);
            }
    const int32_t first = v.empty() ? 0 : v[0];
    while ((int64_t)v.size() % padTo) v.push_back(first);
    return table(key, std::move(v));
}

int PlanBuilder::tColsConvHW(const Act& a, int kh, int kw, int dil, int c0, int cin)
{
    if (cin < 0) cin = a.C - c0;
    const std::string key = "CC:" + std::to_string(a.halo) + ":" + std::to_string(a.W) + ":" + std::to_string(a.C) +
                            ":" + std::to_string(kh) + "x" + std::to_string(kw) + ":" + std::to_string(dil) + ":" +
                            std::to_string(c0) + ":" + std::to_string(cin);
    auto it = tableKey_.find(key);
    if (it != tableKey_.end()) return it->second;
    if (a.halo < dil * (kh / 2) || a.halo < dil * (kw / 2)) throw std::runtime_error("activation halo too small for conv");
    if (cin % VSR_GG_KC || c0 % VSR_GG_KC) throw std::runtime_error("conv input channels must be a multiple of 32");
    std::vector<int32_t> v;
    auto off = [&](int ky, int kx, int c) {
        return (int32_t)(((int64_t)(ky - kh / 2) * dil * a.Wp() + (kx - kw / 2) * dil) * a.C + c0 + c);
    };
    if (Tuning::get().convChannelMajor) { // mirrors Model::pack_conv
        for (int c = 0; c < cin; c += VSR_GG_KC)
            for (int ky = 0; ky < kh; ++ky)
                for (int kx = 0; kx < kw; ++kx) v.push_back(off(ky, kx, c));
    } else {
        for (int ky = 0; ky < kh; ++ky)
            for (int kx = 0; kx < kw; ++kx)
                for (int c = 0; c < cin; c += VSR_GG_KC) v.push_back(off(ky, kx, c));
    }
    return table(key, std::move(v));
}

int PlanBuilder::tRowsLinear(int count, int ld, int padTo)
{
    const std::string key = "RL:" + std::to_string(count) + ":" + std::to_string(ld) + ":" + std::to_string(padTo);
    auto it = tableKey_.find(key);
    if (it != tableKey_.end()) return it->second;
    std::vector<int32_t> v;
    for (int i = 0; i < count; ++i) { checkFits((int64_t)i * ld); v.push_back((int32_t)((int64_t)i * ld)); }
    while ((int)v.size() % padTo) v.push_back(0);
    return table(key, std::move(v));
}

int PlanBuilder::tColsLinear(int nchunks, int padTo)
{
    const std::string key = "CL:" + std::to_string(nchunks) + ":" + std::to_string(padTo);
    auto it = tableKey_.find(key);
    if (it != tableKey_.end()) return it->second;
    std::vector<int32_t> v;
    for (int i = 0; i < nchunks; ++i) v.push_back(i * VSR_GG_KC);
    while ((int)v.size() < padTo) v.push_back(0);
    return table(key, std::move(v));
}

// token (t, oy, ox) of scale s inside the plain QKV buffer [T*fh*fw][3C] (auto_sttn.py:182-190:
// view(b,t,d_k,out_h,height,out_w,width).permute(0,1,3,5,2,4,6) => tokens ordered t, out_h, out_w)
int Plan::tRowsTokens(int T, int s, int choff, int count, int padTo, int oy0, int oy1, int ox0, int ox1, const std::vector<int>* fids)
{
    const int pw = g.patchW[s], ph = g.patchH[s], ow = g.featW / pw, oh = g.featH / ph, C3 = 3 * g.channels;
    if (oy1 < 0) oy1 = oh;
    if (ox1 < 0) ox1 = ow;
    std::string key = "RT:" + std::to_string(T) + ":" + std::to_string(s) + ":" + std::to_string(choff) + ":" +
                      std::to_string(count) + ":" + std::to_string(padTo) + ":" + std::to_string(oy0) + "-" + std::to_string(oy1) +
                      ":" + std::to_string(ox0) + "-" + std::to_string(ox1);
    if (fids) key += ":f" + idsKey(std::vector<int>(fids->begin(), fids->begin() + T));
    auto it = tableKey_.find(key);
    if (it != tableKey_.end()) return it->second;
    std::vector<int32_t> v;
    for (int t = 0; t < T; ++t)
        for (int oy = oy0; oy < oy1; ++oy)
            for (int ox = ox0; ox < ox1; ++ox) {
                const int64_t o = (((int64_t)(fids ? (*fids)[t] : t) * g.featH + oy * ph) * g.featW + ox * pw) * C3 + choff;
                checkFits(o);
                if (o * 4 > 4294967295LL) throw std::runtime_error("q/k/v row offset exceeds the kernels' 32-bit byte offsets");
                v.push_back((int32_t)o);
            }
    assert((int)v.size() == count);
    (void)count;
    const int32_t first = v[0];
    while ((int)v.size() % padTo) v.push_back(first);
    return table(key, std::move(v));
}

// the D = d_k*ph*pw elements of a token as 32-float chunks (y, x, half); the order inside a
// token is free for QK^T (a sum) and is mirrored by tColsPatchAct for the PV scatter.
int Plan::tColsPatch(int s, int padTo)
{
    const std::string key = "CP:" + std::to_string(s) + ":" + std::to_string(padTo);
    auto it = tableKey_.find(key);
    if (it != tableKey_.end()) return it->second;
    const int pw = g.patchW[s], ph = g.patchH[s], C3 = 3 * g.channels, dk = g.channels / g.nscales;
    std::vector<int32_t> v;
    for (int y = 0; y < ph; ++y)
        for (int x = 0; x < pw; ++x)
            for (int c0 = 0; c0 < dk; c0 += VSR_GG_KC) v.push_back((int32_t)(((int64_t)y * g.featW + x) * C3 + c0));
    while ((int)v.size() < padTo) v.push_back(0);
    return table(key, std::move(v));
}

int Plan::tRowsTokensAct(const Act& a, int T, int s, int padTo, int oy0, int oy1, int ox0, int ox1)
{
    const int pw = g.patchW[s], ph = g.patchH[s], ow = g.featW / pw, oh = g.featH / ph, dk = g.channels / g.nscales;
    if (oy1 < 0) oy1 = oh;
    if (ox1 < 0) ox1 = ow;
    const std::string key = "RTA:" + std::to_string(a.buf) + ":" + std::to_string(a.halo) + ":" + std::to_string(T) +
                            ":" + std::to_string(s) + ":" + std::to_string(padTo) + ":" + std::to_string(oy0) + "-" + std::to_string(oy1) +
                            ":" + std::to_string(ox0) + "-" + std::to_string(ox1);
    auto it = tableKey_.find(key);
    if (it != tableKey_.end()) return it->second;
    std::vector<int32_t> v;
    for (int t = 0; t < T; ++t)
        for (int oy = oy0; oy < oy1; ++oy)
            for (int ox = ox0; ox < ox1; ++ox) {
                const int64_t o = a.pix(t, oy * ph, ox * pw) + (int64_t)dk * s;
                checkFits(o);
                v.push_back((int32_t)o);
            }
    const int32_t first = v[0];
    while ((int)v.size() % padTo) v.push_back(first);
    return table(key, std::move(v));
}

int Plan::tColsPatchAct(const Act& a, int s, int padTo)
{
    const std::string key = "CPA:" + std::to_string(a.halo) + ":" + std::to_string(a.W) + ":" + std::to_string(a.C) +
                            ":" + std::to_string(s) + ":" + std::to_string(padTo);
    auto it = tableKey_.find(key);
    if (it != tableKey_.end()) return it->second;
    const int pw = g.patchW[s], ph = g.patchH[s], dk = g.channels / g.nscales;
    std::vector<int32_t> v;
    for (int y = 0; y < ph; ++y)
        for (int x = 0; x < pw; ++x)
            for (int c0 = 0; c0 < dk; c0 += VSR_GG_KC) v.push_back((int32_t)(((int64_t)y * a.Wp() + x) * a.C + c0));
    while ((int)v.size() < padTo) v.push_back(0);
    return table(key, std::move(v));
}

static std::vector<int> iota(int n)
{
    std::vector<int> v(n);
    for (int i = 0; i < n; ++i) v[i] = i;
    return v;
}

// conv (ksz 1 or 3, stride, dilation) + bias + optional LeakyReLU(0.2) + optional residual,
// as one gather-GEMM: M = nOut*out.H*out.W pixels, N = cout, K = ksz*ksz*cin.
void Plan::addConv(const char* tag, const Act& in, const std::vector<int>& inIds, const Act& out, int nOut, int ksz,
                   int stride, int dil, const ConvW& w, int act, const Act* res, const std::vector<int>* resIds, int ylo, int yhi,
                   int xlo, int xhi)
{
    if (yhi < 0) yhi = out.H;
    if (xhi < 0) xhi = out.W;
    if (ylo < 0 || yhi > out.H || ylo >= yhi || (stride != 1 && (ylo != 0 || yhi != out.H))) throw std::runtime_error(std::string("conv row range: ") + tag);
    if (xlo < 0 || xhi > out.W || xlo >= xhi || (stride != 1 && (xlo != 0 || xhi != out.W))) throw std::runtime_error(std::string("conv column range: ") + tag);
    const bool rect = xlo != 0 || xhi != out.W;   // a column range needs its own table; a row range alone is a row offset of the plain one
    const int oh = yhi - ylo, ow = xhi - xlo;     // pix(f, y + ylo, x) = pix(f, y, x) + ylo * Wp * C
    if (w.K != ksz * ksz * in.C) throw std::runtime_error(std::string("conv K mismatch: ") + tag);
    if ((int)inIds.size() != nOut) throw std::runtime_error("conv frame list mismatch");
    Op op;
    op.kind = OP_GEMM;
    op.tag = tag;
    op.bmode = VSR_BMODE_NK;
    op.tileCfg = pickTile(w.cout);
    int BM, BN;
    tileDims(op.tileCfg, BM, BN);
    GemmItem it{};
    it.M = nOut * oh * ow;
    it.N = w.cout;
    it.K = w.K;
    it.tilesM = cdiv(it.M, BM);
    it.tilesN = cdiv(it.N, BN);
    it.splitK = 1;
    it.chunksPerSplit = it.K / VSR_GG_KC;
    it.splitStride = 0;
    it.alpha = 1.f;
    it.act = act;
    it.bufA = in.buf; it.offA = 0;
    it.tRowA = rect ? tRowsActRect(in, inIds, ylo, yhi, xlo, xhi, BM) : tRowsAct(in, inIds, oh, out.W, stride, BM, (int64_t)ylo * in.Wp() * in.C);
    it.tColA = tColsConv(in, ksz, dil);
    it.bufB = BUF_WEIGHTS; it.offB = w.w;
    it.tRowB = tRowsLinear(it.N, it.K, BN);
    it.tColB = tColsLinear(it.K / VSR_GG_KC, it.K / VSR_GG_KC);
    it.bufC = out.buf; it.offC = 0;
    it.tRowC = rect ? tRowsActRect(out, iota(nOut), ylo, yhi, xlo, xhi, BM) : tRowsAct(out, iota(nOut), oh, out.W, 1, BM, (int64_t)ylo * out.Wp() * out.C);
    it.tColC = tColsLinear(cdiv(it.N, VSR_GG_KC), it.tilesN * BN / VSR_GG_KC);
    it.offBias = w.b;
    if (res) {
        it.bufR = res->buf; it.offR = 0;
        it.tRowR = rect ? tRowsActRect(*res, *resIds, ylo, yhi, xlo, xhi, BM) : tRowsAct(*res, *resIds, oh, out.W, 1, BM, (int64_t)ylo * res->Wp() * res->C);
    } else {
        it.bufR = -1; it.offR = 0; it.tRowR = -1;
    }
    op.flops = 2.0 * it.M * it.N * (double)(ksz * ksz * in.C);
    op.gemm.push_back(it);
    need(out.buf, (int64_t)nOut * out.frameElems());
    flops += op.flops;
    ops.push_back(std::move(op));
}

// multi-scale patch attention of one block (auto_sttn.py:167-206, Attention :140-145):
// grouped QK^T (split-K on the coarse scales) -> grouped row softmax -> grouped PV that
// scatters straight back into the NHWC attention buffer.  Problems are listed largest first
// so that the long workgroups are dispatched first and the short ones fill the tail; a PV
// whose token contraction is long is split into slices (fewer, longer workgroups than CUs
// would otherwise leave half the chip idle) and combined by a reduce-scatter pass.
// Tq <= T: the QUERY frames are the first Tq of the T frames (the keys and values are all T): the last block of a window only needs
// the rows of its neighbour frames (Plan::buildWindow)
// [attLo, attHi): the feature rows of the attention output that anything reads (the last block of a window that feeds a ranged
// decoder): the query tokens are the patches that touch those rows
void Plan::addAttention(int Tq, int T, const BlockW&, int attLo, int attHi, int attXLo, int attXHi, int qkvBuf, const std::vector<int>* fids)
{
    if (qkvBuf < 0) qkvBuf = lb(BUF_QKV);
    if (attHi < 0) attHi = g.featH;
    if (attXHi < 0) attXHi = g.featW;
    const Tuning& tu = tu_;
    const int C = g.channels, dk = C / g.nscales;
    const Act att{lb(BUF_ATT), T, g.featH, g.featW, C, 1};
    Op qk, sm, pv;
    qk.kind = OP_GEMM; qk.tag = "attn.qk"; qk.tileCfg = tu.qkTile; qk.bmode = VSR_BMODE_NK;
    sm.kind = OP_SOFTMAX; sm.tag = "attn.softmax";
    pv.kind = OP_GEMM; pv.tag = "attn.pv"; pv.tileCfg = tu.pvTile; pv.bmode = VSR_BMODE_KN;
    int qBM, qBN, pBM, pBN;
    tileDims(qk.tileCfg, qBM, qBN);
    tileDims(pv.tileCfg, pBM, pBN);
    std::vector<Op> reduces;
    bool pvFused = false;
    int64_t sOff = 0, pOff = 0, partOff = 0, lOff = 0;
    for (int s = g.nscales - 1; s >= 0; --s) { // finest scale (most tokens, most work) first
        const int pw = g.patchW[s], ph = g.patchH[s];
        const int Pn = (g.featW / pw) * (g.featH / ph);
        const int Ntok = T * Pn;        // key / value tokens
        const int oy0 = attLo / ph, oy1 = cdiv(attHi, ph);       // patch rows / columns that touch what is read
        const int ox0 = attXLo / pw, ox1 = cdiv(attXHi, pw);
        const int Mtok = Tq * (oy1 - oy0) * (ox1 - ox0);         // query tokens: the first Tq frames (frame-major), those patches
        const int D = dk * pw * ph;
        const int ldS = (int)rup(Ntok, VSR_GG_KC);
        const int nchunks = D / VSR_GG_KC;
        int cps = nchunks < 32 ? nchunks : 32;
        int splitK = cdiv(nchunks, cps);
        cps = cdiv(nchunks, splitK);
        splitK = cdiv(nchunks, cps);
        const int64_t plane = (int64_t)Mtok * ldS;

        GemmItem a{};
        a.M = Mtok; a.N = Ntok; a.K = D;
        a.tilesM = cdiv(Mtok, qBM); a.tilesN = cdiv(Ntok, qBN);
        a.splitK = splitK; a.chunksPerSplit = cps; a.splitStride = plane;
        a.alpha = 1.f; a.act = VSR_ACT_NONE;
        a.bufA = qkvBuf; a.offA = 0;
        a.tRowA = tRowsTokens(Tq, s, dk * s, Mtok, qBM, oy0, oy1, ox0, ox1, fids);
        a.tColA = tColsPatch(s, nchunks);
        a.bufB = qkvBuf; a.offB = 0;
        a.tRowB = tRowsTokens(T, s, C + dk * s, Ntok, qBN, 0, -1, 0, -1, fids);
        a.tColB = a.tColA;
        a.bufC = lb(BUF_S); a.offC = sOff;
        a.tRowC = tRowsLinear(Mtok, ldS, qBM);
        a.tColC = tColsLinear(a.tilesN * qBN / VSR_GG_KC, a.tilesN * qBN / VSR_GG_KC);
        a.bufR = -1; a.tRowR = -1; a.offBias = -1;
        // Fused softmax (exact-fp32 mode): a scale whose scores are not split along K and whose token count is whole chunks keeps
        // no probability matrix -- the score GEMM scales by log2(e)/sqrt(D) itself and leaves the row maxima (VSR_ACT_ROW_MAX), the
        // P.V GEMM reads the scores, takes 2^(s - max) while staging and normalises by the row sums (VSR_ACT_A_EXP).  At the
        // 4800-token scale that is 92 MB written and read once instead of twice, and no k_softmax_rows pass over them.
        const bool fused = tu.fuseSoftmax && precision == 0 && splitK == 1 && Ntok % VSR_GG_KC == 0;
        const float scale = (float)(1.0 / sqrt((double)D)); // scores / math.sqrt(query.size(-1))
        int64_t rmaxOff = -1;
        if (fused) {
            rmaxOff = rowmaxElems_;
            rowmaxElems_ += rup((int64_t)a.tilesM * qBM > (int64_t)cdiv(Mtok, pBM) * pBM ? (int64_t)a.tilesM * qBM : (int64_t)cdiv(Mtok, pBM) * pBM, 32);
            a.alpha = (float)((double)scale * 1.4426950408889634);   // log2(e): the P.V kernel exponentiates with v_exp_f32 (2^x)
            a.act |= VSR_ACT_ROW_MAX;
            a.bufR = BUF_ROWMAX; a.offR = rmaxOff;
        }
        qk.gemm.push_back(a);
        qk.flops += 2.0 * Mtok * (double)Ntok * D;

        if (!fused) {
            SoftmaxItem m{};
            m.bufS = lb(BUF_S); m.offS = sOff; m.splitStride = plane; m.nsplit = splitK;
            m.bufP = lb(BUF_P); m.offP = pOff;
            m.M = Mtok; m.N = Ntok; m.ldS = ldS; m.ldP = ldS;
            m.scale = scale;
            sm.softmax.push_back(m);
        }

        const int kchunks = ldS / VSR_GG_KC;
        int pvSplit = 1, pvCps = kchunks;
        if (tu.pvSplitChunks > 0 && kchunks >= 2 * tu.pvSplitChunks) {
            pvSplit = cdiv(kchunks, tu.pvSplitChunks);
            pvCps = cdiv(kchunks, pvSplit);
            pvSplit = cdiv(kchunks, pvCps);
        }
        GemmItem b{};
        b.M = Mtok; b.N = D; b.K = ldS;
        b.tilesM = cdiv(Mtok, pBM); b.tilesN = cdiv(D, pBN);
        b.splitK = pvSplit; b.chunksPerSplit = pvCps;
        b.alpha = 1.f; b.act = VSR_ACT_NONE;
        b.bufA = fused ? lb(BUF_S) : lb(BUF_P); b.offA = fused ? sOff : pOff;
        b.tRowA = tRowsLinear(Mtok, ldS, pBM);
        b.tColA = tColsLinear(kchunks, kchunks);
        b.bufB = qkvBuf; b.offB = 0;
        b.tRowB = tRowsTokens(T, s, 2 * C + dk * s, Ntok, ldS, 0, -1, 0, -1, fids); // K rows, padded with token 0 (P pad cols are 0)
        b.tColB = tColsPatch(s, b.tilesN * pBN / VSR_GG_KC);
        const int tRowAtt = tRowsTokensAct(att, Tq, s, pBM, oy0, oy1, ox0, ox1);
        const int tColAtt = tColsPatchAct(att, s, b.tilesN * pBN / VSR_GG_KC);
        if (pvSplit == 1) {
            b.bufC = lb(BUF_ATT); b.offC = 0; b.splitStride = 0;
            b.tRowC = tRowAtt;
            b.tColC = tColAtt;
        } else {        // partial planes [pvSplit][Ntok][D], combined + scattered by the reduce op
            b.bufC = lb(BUF_PVPART); b.offC = partOff; b.splitStride = (int64_t)Mtok * D;
            b.tRowC = tRowsLinear(Mtok, D, pBM);
            b.tColC = tColsLinear(D / VSR_GG_KC, b.tilesN * pBN / VSR_GG_KC);
            Op r;
            r.kind = OP_REDUCE_SCATTER; r.tag = "attn.pv.reduce";
            r.bufSrc = lb(BUF_PVPART); r.offSrc = partOff; r.splitStride = b.splitStride; r.nsplit = pvSplit;
            r.bufDst = lb(BUF_ATT); r.offDst = 0; r.M = Mtok; r.N = D; r.tRowC = tRowAtt; r.tColC = tColAtt;
            if (fused) { r.ibuf[0] = lb(BUF_LSUM); r.ioff[0] = lOff; r.ipar[0] = b.tilesM * pBM; }
            reduces.push_back(std::move(r));
            partOff += rup(b.splitStride * pvSplit, 32);
        }
        b.bufR = -1; b.tRowR = -1; b.offBias = -1;
        if (fused) {
            b.act |= VSR_ACT_A_EXP;
            b.bufBias = BUF_ROWMAX; b.offBias = rmaxOff;
            if (pvSplit > 1) { b.bufR = lb(BUF_LSUM); b.offR = lOff; lOff += rup((int64_t)pvSplit * b.tilesM * pBM, 32); }
            pvFused = true;
        }
        pv.gemm.push_back(b);
        pv.flops += 2.0 * Mtok * (double)Ntok * D;

        sOff += rup(plane * splitK, 32);
        if (!fused) pOff += rup(plane, 32);
    }
    if (pvFused) pv.ipar[0] = 1;       // the launch may carry VSR_ACT_A_EXP problems: kernel variant 1 | VSR_VARIANT_A_EXP
    need(lb(BUF_S), sOff);
    need(lb(BUF_P), pOff);
    need(BUF_ROWMAX, rowmaxElems_);
    need(lb(BUF_LSUM), lOff);
    need(lb(BUF_PVPART), partOff);
    need(lb(BUF_ATT), att.elems());
    flops += qk.flops + pv.flops;
    ops.push_back(std::move(qk));
    if (!sm.softmax.empty()) ops.push_back(std::move(sm));
    ops.push_back(std::move(pv));
    for (Op& r : reduces) ops.push_back(std::move(r));
}

// one sliding window (sttn_auto_inpaint.py:142-162): infer over neighbours+refs, decode the
// neighbours, tanh -> u8, pairwise overlap average into comp.
void Plan::buildWindow(const std::vector<int>& neighbors, const std::vector<int>& refs, std::vector<int32_t>& visits)
{
    const int C = g.channels, fh = g.featH, fw = g.featW, mh = g.modelH, mw = g.modelW;
    std::vector<int> ids = neighbors;
    ids.insert(ids.end(), refs.begin(), refs.end());
    const int T = (int)ids.size(), nn = (int)neighbors.size();
    const Act feats{BUF_FEATS, L, fh, fw, C, 2};
    const Act x0{lb(BUF_X0), T, fh, fw, C, 2}, x1{lb(BUF_X1), T, fh, fw, C, 2};
    const Act att{lb(BUF_ATT), T, fh, fw, C, 1}, f1{lb(BUF_F1), T, fh, fw, C, 1};
    const std::vector<int> idT = iota(T);

    // Rows of every decoder stage that the output rows [decLo, decHi) depend on (the whole image when no range was given): a 3x3
    // conv widens by one row, the align_corners x2 upsampling of H source rows reads rows floor(y (H - 1) / (2 H - 1)) and the
    // next one for output row y -- the kernel's float arithmetic is not repeated here, one row of slack on each side covers its
    // rounding.  Rows outside a stage's range keep whatever an earlier launch left there; nothing inside the ranges reads them.
    struct Rng { int lo, hi; };
    auto widen = [](Rng r, int by, int H) { return Rng{r.lo - by > 0 ? r.lo - by : 0, r.hi + by < H ? r.hi + by : H}; };
    auto below = [](Rng r, int Hsrc) {          // source rows of the x2 upsampling that produces rows r of 2 * Hsrc
        const int OH = 2 * Hsrc;
        int lo = (int)((int64_t)r.lo * (Hsrc - 1) / (OH - 1)) - 1, hi = (int)((int64_t)(r.hi - 1) * (Hsrc - 1) / (OH - 1)) + 3;
        return Rng{lo > 0 ? lo : 0, hi < Hsrc ? hi : Hsrc};
    };
    const bool ranged = decHi > decLo && (decLo > 0 || decHi < mh);
    const Rng rOut = ranged ? Rng{decLo, decHi} : Rng{0, mh};
    const Rng rD3 = widen(rOut, 1, mh), rUp2 = widen(rD3, 1, mh);
    const Rng rD2 = below(rUp2, 2 * fh), rD1 = widen(rD2, 1, 2 * fh), rUp1 = widen(rD1, 1, 2 * fh);
    const Rng rX1 = below(rUp1, fh);            // rows of the last block's output the decoder reads
    const int lastLo = rX1.lo, lastHi = rX1.hi;
    // the same chain along x for the GEMMs (the decoder's two elementwise kernels keep whole rows)
    const bool rangedX = decXHi > decXLo && (decXLo > 0 || decXHi < mw);
    const Rng cOut = rangedX ? Rng{decXLo, decXHi} : Rng{0, mw};
    const Rng cD3 = widen(cOut, 1, mw), cUp2 = widen(cD3, 1, mw);
    const Rng cD2 = below(cUp2, 2 * fw), cD1 = widen(cD2, 1, 2 * fw), cUp1 = widen(cD1, 1, 2 * fw);
    const Rng cX1 = below(cUp1, fw);
    double fullBlockFlops = 0;

    Act cur = feats;
    std::vector<int> curIds = ids;
    for (int b = 0; b < g.blocks; ++b) {
        const BlockW& bw = m_.blk[b];
        const bool shared0 = b == 0 && qkv0_;     // this block's q/k/v of every frame of the chunk are in BUF_QKV0 (Plan::Plan)
        if (shared0) trimmedFlops_ += 2.0 * T * fh * fw * (3.0 * C) * C;
        if (!shared0) {   // fused Q/K/V 1x1 (auto_sttn.py:172-174) -> plain [T*fh*fw][3C]
            Op op;
            op.kind = OP_GEMM; op.tag = "attn.qkv"; op.tileCfg = tu_.qkvTile; op.bmode = VSR_BMODE_NK;
            int BM, BN;
            tileDims(op.tileCfg, BM, BN);
            GemmItem it{};
            it.M = T * fh * fw; it.N = 3 * C; it.K = C;
            it.tilesM = cdiv(it.M, BM); it.tilesN = cdiv(it.N, BN);
            it.splitK = 1; it.chunksPerSplit = it.K / VSR_GG_KC; it.alpha = 1.f; it.act = VSR_ACT_NONE;
            it.bufA = cur.buf; it.offA = 0;
            it.tRowA = tRowsAct(cur, curIds, fh, fw, 1, BM, 0);
            it.tColA = tColsConv(cur, 1, 1);
            it.bufB = BUF_WEIGHTS; it.offB = bw.qkv.w;
            it.tRowB = tRowsLinear(it.N, it.K, BN);
            it.tColB = tColsLinear(it.K / VSR_GG_KC, it.K / VSR_GG_KC);
            it.bufC = lb(BUF_QKV); it.offC = 0;
            it.tRowC = tRowsLinear(it.M, 3 * C, BM);
            it.tColC = tColsLinear(it.N / VSR_GG_KC, it.tilesN * BN / VSR_GG_KC);
            it.offBias = bw.qkv.b;
            it.bufR = -1; it.tRowR = -1;
            op.flops = 2.0 * it.M * (double)it.N * it.K;
            flops += op.flops;
            op.gemm.push_back(it);
            need(lb(BUF_QKV), (int64_t)it.M * 3 * C);
            ops.push_back(std::move(op));
        }
        // The LAST block of a window is read by the decoder alone, and the decoder takes the neighbour frames only
        // (sttn_auto_inpaint.py:150: pred_feat[:len(neighbor_ids)]).  Everything behind the K / V projection is per query row and
        // per frame -- a score row, its softmax and its P.V row belong to one query token, the 3x3 convs stay inside a frame -- so
        // the rows of the reference frames (the last T - nn of the window: `ids` lists the neighbours first) are computed by the
        // reference and never read.  They are not computed here (Tuning::trimLastBlock): the neighbour rows come out bit for bit as
        // before (same tiles' worth of products in the same order), 3 % of the chunk's FLOPs are not spent.  `refFlops` keeps the
        // reference's count.
        // With a decoder row range (Plan::decLo) the same holds for ROWS: the decoder's first upsampling reads feature rows
        // [lastLo, lastHi) of the last block's output (the chain is walked backwards at the top of this function), its second 3x3
        // conv needs one more row of the first conv's output on each side, the first (dilation 2) two more of the out-conv's, the
        // out-conv one more of the attention output, and the attention output rows belong to the patches that touch them.
        const bool last = tu_.trimLastBlock && b == g.blocks - 1;
        const int Tq = last ? nn : T;
        const int r2lo = last ? lastLo : 0, r2hi = last ? lastHi : fh;                       // ffn.2 output = the block's output
        auto wide = [&](int lo, int hi, int by, int& olo, int& ohi) { olo = lo - by > 0 ? lo - by : 0; ohi = hi + by < fh ? hi + by : fh; };
        int r1lo, r1hi, r0lo, r0hi, ralo, rahi;
        wide(r2lo, r2hi, 1, r1lo, r1hi);                                                       // ffn.1 output
        wide(r2lo, r2hi, 3, r0lo, r0hi);                                                       // out-conv output (and the residual of ffn.2)
        wide(r2lo, r2hi, 4, ralo, rahi);                                                       // attention output
        const Rng c2 = last ? cX1 : Rng{0, fw}, c1 = widen(c2, 1, fw), c0 = widen(c2, 3, fw), ca = widen(c2, 4, fw);   // the same along x
        const std::vector<int> idQ = iota(Tq);
        const std::vector<int> curIdsQ(curIds.begin(), curIds.begin() + Tq);
        const double before = flops;
        if (shared0) addAttention(Tq, T, bw, ralo, rahi, ca.lo, ca.hi, BUF_QKV0, &ids);
        else addAttention(Tq, T, bw, ralo, rahi, ca.lo, ca.hi);
        // x = x + LeakyReLU(conv3x3(att))            (auto_sttn.py:162-164,237)
        addConv("attn.out", att, idQ, x0, Tq, 3, 1, 1, bw.out, VSR_ACT_LRELU02, &cur, &curIdsQ, r0lo, r0hi, c0.lo, c0.hi);
        // x = x + LeakyReLU(conv3x3(LeakyReLU(conv3x3 dil2(x))))   (auto_sttn.py:214-218,238)
        addConv("ffn.1", x0, idQ, f1, Tq, 3, 1, 2, bw.ffn1, VSR_ACT_LRELU02, nullptr, nullptr, r1lo, r1hi, c1.lo, c1.hi);
        addConv("ffn.2", f1, idQ, x1, Tq, 3, 1, 1, bw.ffn2, VSR_ACT_LRELU02, &x0, &idQ, r2lo, r2hi, c2.lo, c2.hi);
        if (last && b > 0) trimmedFlops_ += fullBlockFlops - (flops - before);               // (the blocks of a window are alike in full form)
        else fullBlockFlops = flops - before;
        cur = x1;
        curIds = idT;
    }

    // decoder on the neighbour frames only (sttn_auto_inpaint.py:150; auto_sttn.py:87-95,118-127)
    const Act up1{lb(BUF_UP1), nn, 2 * fh, 2 * fw, C, 1}, d1{lb(BUF_D1), nn, 2 * fh, 2 * fw, 128, 1};
    const Act d2{lb(BUF_D2), nn, 2 * fh, 2 * fw, 64, 0}, up2{lb(BUF_UP2), nn, mh, mw, 64, 1}, d3{lb(BUF_D3), nn, mh, mw, 64, 1};
    const std::vector<int> idN = iota(nn);
    {
        Op op;
        op.kind = OP_UPSAMPLE2X; op.tag = "dec.up1";
        op.bufSrc = x1.buf; op.H = fh; op.W = fw; op.C = C; op.haloS = x1.halo; op.bufDst = up1.buf; op.haloD = up1.halo;
        op.n = nn;
        op.ipar[1] = rUp1.lo; op.ipar[2] = rUp1.hi;       // output rows
        need(up1.buf, up1.elems());
        ops.push_back(std::move(op));
    }
    // (refFlops keeps the reference's count: what a ranged conv leaves out is its flops x (H / rows - 1))
    auto skipped = [&](Rng r, int H, Rng c, int W) {
        const double part = (double)(r.hi - r.lo) * (c.hi - c.lo);
        trimmedFlops_ += ops.back().flops * ((double)H * W - part) / part;
    };
    addConv("dec.1", up1, idN, d1, nn, 3, 1, 1, m_.dec[0], VSR_ACT_LRELU02, nullptr, nullptr, rD1.lo, rD1.hi, cD1.lo, cD1.hi);
    skipped(rD1, 2 * fh, cD1, 2 * fw);
    addConv("dec.2", d1, idN, d2, nn, 3, 1, 1, m_.dec[1], VSR_ACT_LRELU02, nullptr, nullptr, rD2.lo, rD2.hi, cD2.lo, cD2.hi);
    skipped(rD2, 2 * fh, cD2, 2 * fw);
    {
        Op op;
        op.kind = OP_UPSAMPLE2X; op.tag = "dec.up2";
        op.bufSrc = d2.buf; op.H = 2 * fh; op.W = 2 * fw; op.C = 64; op.haloS = d2.halo; op.bufDst = up2.buf;
        op.haloD = up2.halo; op.n = nn;
        op.ipar[1] = rUp2.lo; op.ipar[2] = rUp2.hi;
        need(up2.buf, up2.elems());
        ops.push_back(std::move(op));
    }
    addConv("dec.3", up2, idN, d3, nn, 3, 1, 1, m_.dec[2], VSR_ACT_LRELU02, nullptr, nullptr, rD3.lo, rD3.hi, cD3.lo, cD3.hi);
    skipped(rD3, mh, cD3, mw);
    if (tu_.outConvBlocked && mh % Model::kOutBlkH == 0 && mw % Model::kOutBlkW == 0) {
        // 64 -> 3 conv over 2x4 output blocks (Model::pack_conv_blocked): row = block, columns (dy, dx, c) in a [blocks][32] buffer
        const ConvW& w = m_.dec4blk;
        const int bh = Model::kOutBlkH, bw = Model::kOutBlkW, wh = bh + 2, ww = bw + 2;
        Op op;
        op.kind = OP_GEMM; op.tag = "dec.4"; op.tileCfg = VSR_TILE_256x32; op.bmode = VSR_BMODE_NK;
        const int BM = 256, BN = 32;
        GemmItem it{};
        const int by0 = rOut.lo / bh, by1 = rOut.hi / bh;      // (decLo / decHi, decXLo / decXHi are whole blocks: Plan::Plan)
        const int bx0 = cOut.lo / bw, bx1 = cOut.hi / bw;
        it.M = nn * (by1 - by0) * (bx1 - bx0); it.N = w.cout; it.K = w.K;
        it.tilesM = cdiv(it.M, BM); it.tilesN = 1;
        it.splitK = 1; it.chunksPerSplit = it.K / VSR_GG_KC; it.alpha = 1.f; it.act = VSR_ACT_NONE;
        it.bufA = d3.buf; it.offA = 0;
        std::vector<int32_t> crows;          // where a block's 32 columns go: the [frame][block row][block col] slot of the FULL image
        {   // rows: the block's top-left pixel; columns: the (bh+2) x (bw+2) window around the block, K order as packed
            std::vector<int32_t> rows;
            for (int f = 0; f < nn; ++f)
                for (int by = by0; by < by1; ++by)
                    for (int bx = bx0; bx < bx1; ++bx) {
                        rows.push_back((int32_t)d3.pix(f, by * bh, bx * bw));
                        crows.push_back((int32_t)((((int64_t)f * (mh / bh) + by) * (mw / bw) + bx) * 32));
                    }
            while ((int)rows.size() % BM) { rows.push_back(rows[0]); crows.push_back(crows[0]); }
            it.tRowA = table("RBLK:" + std::to_string(nn) + ":" + std::to_string(mh) + "x" + std::to_string(mw) + ":" + std::to_string(by0) + "-" + std::to_string(by1) + ":" + std::to_string(bx0) + "-" + std::to_string(bx1), std::move(rows));
            std::vector<int32_t> cols;
            auto off = [&](int wy, int wx, int c) { return (int32_t)(((int64_t)(wy - 1) * d3.Wp() + (wx - 1)) * d3.C + c); };
            if (tu_.convChannelMajor) {
                for (int c = 0; c < d3.C; c += VSR_GG_KC)
                    for (int wy = 0; wy < wh; ++wy)
                        for (int wx = 0; wx < ww; ++wx) cols.push_back(off(wy, wx, c));
            } else {
                for (int wy = 0; wy < wh; ++wy)
                    for (int wx = 0; wx < ww; ++wx)
                        for (int c = 0; c < d3.C; c += VSR_GG_KC) cols.push_back(off(wy, wx, c));
            }
            it.tColA = table("CBLK:" + std::to_string(d3.Wp()) + ":" + std::to_string(d3.C) + ":" + std::to_string(tu_.convChannelMajor), std::move(cols));
        }
        it.bufB = BUF_WEIGHTS; it.offB = w.w;
        it.tRowB = tRowsLinear(it.N, it.K, BN);
        it.tColB = tColsLinear(it.K / VSR_GG_KC, it.K / VSR_GG_KC);
        it.bufC = lb(BUF_D4); it.offC = 0;
        it.tRowC = table("CBLKROW:" + std::to_string(nn) + ":" + std::to_string(mh) + "x" + std::to_string(mw) + ":" + std::to_string(by0) + "-" + std::to_string(by1) + ":" + std::to_string(bx0) + "-" + std::to_string(bx1), std::move(crows));
        it.tColC = tColsLinear(1, 1);
        it.offBias = w.b;
        it.bufR = -1; it.tRowR = -1;
        op.flops = 2.0 * (double)nn * (rOut.hi - rOut.lo) * (cOut.hi - cOut.lo) * 3.0 * (9.0 * d3.C);     // the algorithmic work of the 3x3 conv, not the padded window's
        flops += op.flops;
        op.gemm.push_back(it);
        need(lb(BUF_D4), (int64_t)cdiv(nn * (mh / bh) * (mw / bw), BM) * BM * 32);
        ops.push_back(std::move(op));
        skipped(rOut, mh, cOut, mw);
    } else
    {   // 64 -> 3 conv into a plain [M][32] buffer (columns 0..2 valid)
        const ConvW& w = m_.dec[3];
        Op op;
        op.kind = OP_GEMM; op.tag = "dec.4"; op.tileCfg = VSR_TILE_256x32; op.bmode = VSR_BMODE_NK;
        const int BM = 256, BN = 32;
        GemmItem it{};
        if (ranged) throw std::runtime_error("a decoder row range needs the blocked output conv (VSR_OUT_CONV_BLOCKED=1)");
        it.M = nn * mh * mw; it.N = w.cout; it.K = w.K;
        it.tilesM = cdiv(it.M, BM); it.tilesN = 1;
        it.splitK = 1; it.chunksPerSplit = it.K / VSR_GG_KC; it.alpha = 1.f; it.act = VSR_ACT_NONE;
        it.bufA = d3.buf; it.offA = 0;
        it.tRowA = tRowsAct(d3, idN, mh, mw, 1, BM, 0);
        it.tColA = tColsConv(d3, 3, 1);
        it.bufB = BUF_WEIGHTS; it.offB = w.w;
        it.tRowB = tRowsLinear(it.N, it.K, BN);
        it.tColB = tColsLinear(it.K / VSR_GG_KC, it.K / VSR_GG_KC);
        it.bufC = lb(BUF_D4); it.offC = 0;
        it.tRowC = tRowsLinear(it.M, 32, BM);
        it.tColC = tColsLinear(1, 1);
        it.offBias = w.b;
        it.bufR = -1; it.tRowR = -1;
        op.flops = 2.0 * it.M * (double)it.N * it.K;
        flops += op.flops;
        op.gemm.push_back(it);
        need(lb(BUF_D4), (int64_t)it.M * 32);
        ops.push_back(std::move(op));
    }
    {
        Op op;
        op.kind = OP_DECODE_OUT; op.tag = "dec.out";
        op.bufSrc = lb(BUF_D4); op.bufDst = BUF_COMP; op.ldy = 32; op.pix = mh * mw; op.n = nn;
        op.ipar[1] = rOut.lo; op.ipar[2] = rOut.hi;       // rows of the mw-wide image that are decoded and averaged
        if (g.variant == 1) { op.ipar[1] = 0; op.ipar[2] = mh; }   // sttn-det: the rows without mask take the input frame here (below), every row is written
        if (tu_.outConvBlocked && mh % Model::kOutBlkH == 0 && mw % Model::kOutBlkW == 0) op.W = mw;   // src rows are 2x4 blocks of a mw-wide image
        op.bufMask = g.variant == 1 ? BUF_MASK_U8 : -1;   // sttn-det: model-resolution blend with the input frames
        std::vector<int32_t> fi, fs;
        for (int i = 0; i < nn; ++i) {
            fi.push_back(neighbors[i]);
            fs.push_back(visits[neighbors[i]] == 0 ? 1 : 0);
            visits[neighbors[i]]++;
        }
        std::string k1 = "FI:", k2 = "FS:";
        for (int v : fi) k1 += std::to_string(v) + ",";
        for (int v : fs) k2 += std::to_string(v) + ",";
        op.tFrameIdx = table(k1, std::move(fi));
        op.tFirst = table(k2, std::move(fs));
        ops.push_back(std::move(op));
    }
    need(lb(BUF_X0), x0.elems());
    need(lb(BUF_X1), x1.elems());
    need(lb(BUF_F1), f1.elems());
    ++nwindows;
}

void Plan::decoder_bounds(const Geometry& g, int precision, int decLo_, int decHi_, int decXLo_, int decXHi_, int* lo, int* hi, int* xlo, int* xhi)
{
    int decLo = 0, decHi = 0, decXLo = 0, decXHi = 0;
    if (decHi_ > decLo_ && Tuning::get(precision).outConvBlocked && g.modelH % Model::kOutBlkH == 0 && g.modelW % Model::kOutBlkW == 0) {
        // whole 2-row blocks of the output conv, inside the image (the per-pixel form of that conv keeps the whole image)
        decLo = (decLo_ < 0 ? 0 : decLo_) / Model::kOutBlkH * Model::kOutBlkH;
        decHi = (decHi_ > g.modelH ? g.modelH : decHi_);
        decHi = (decHi + Model::kOutBlkH - 1) / Model::kOutBlkH * Model::kOutBlkH;
        if (decHi > g.modelH) decHi = g.modelH;
        if (decXHi_ > decXLo_) {         // columns only together with rows; whole 4-column blocks
            decXLo = (decXLo_ < 0 ? 0 : decXLo_) / Model::kOutBlkW * Model::kOutBlkW;
            decXHi = (decXHi_ > g.modelW ? g.modelW : decXHi_);
            decXHi = (decXHi + Model::kOutBlkW - 1) / Model::kOutBlkW * Model::kOutBlkW;
            if (decXHi > g.modelW) decXHi = g.modelW;
        }
    }
    *lo = decLo; *hi = decHi; *xlo = decXLo; *xhi = decXHi;
}

Plan::Plan(const Model& model, int L_, int precision_, int lanes_, int decLo_, int decHi_, int decXLo_, int decXHi_)
    : L(L_), precision(precision_), lanes(lanes_ < 1 ? 1 : (lanes_ > kMaxLanes ? kMaxLanes : lanes_)), g(model.g), m_(model), tu_(Tuning::get(precision_))
{
    decoder_bounds(g, precision_, decLo_, decHi_, decXLo_, decXHi_, &decLo, &decHi, &decXLo, &decXHi);
    if (!model.packed_ready()) throw std::runtime_error("model weights are not packed");
    if (L <= 0) throw std::runtime_error("empty frame list");
    bufElems.assign(BUF_COUNT, 0);
    bufElems[BUF_WEIGHTS] = (int64_t)model.packed.size();
    const int mh = g.modelH, mw = g.modelW, fh = g.featH, fw = g.featW, C = g.channels;
    need(BUF_IN_U8, (int64_t)L * mh * mw * 3);
    need(BUF_COMP, (int64_t)L * mh * mw * 3);
    const Act e1{BUF_E1, L, mh / 2, mw / 2, 64, 1}, e2{BUF_E2, L, mh / 2, mw / 2, 64, 1};
    const Act e3{BUF_E3, L, fh, fw, 128, 1}, feats{BUF_FEATS, L, fh, fw, C, 2};
    const std::vector<int> idL = iota(L);
    {   // Stack/ToTorchFormatTensor/*2-1 fused with the im2col of encoder conv1
        Op op;
        op.kind = OP_NORM_IM2COL; op.tag = "enc.im2col";
        op.bufSrc = BUF_IN_U8; op.bufDst = BUF_IM2COL; op.H = mh; op.W = mw; op.n = L;
        // sttn-det feeds feats*(1-mask) to the encoder (sttn_det_inpaint.py:143); sttn-auto never sees the mask
        op.premask = g.variant == 1 ? 1 : 0;
        op.bufMask = g.variant == 1 ? BUF_MASK_U8 : -1;
        if (g.variant == 1) need(BUF_MASK_U8, (int64_t)L * mh * mw);
        need(BUF_IM2COL, (int64_t)L * (mh / 2) * (mw / 2) * 32);
        ops.push_back(std::move(op));
    }
    {   // encoder conv1 3->64 s2 as a plain GEMM over the im2col rows (K = 27 padded to 32)
        const ConvW& w = model.enc[0];
        Op op;
        op.kind = OP_GEMM; op.tag = "enc.1"; op.tileCfg = VSR_TILE_256x64; op.bmode = VSR_BMODE_NK;
        const int BM = 256, BN = 64;
        GemmItem it{};
        it.M = L * e1.H * e1.W; it.N = w.cout; it.K = w.K;
        it.tilesM = cdiv(it.M, BM); it.tilesN = cdiv(it.N, BN);
        it.splitK = 1; it.chunksPerSplit = it.K / VSR_GG_KC; it.alpha = 1.f; it.act = VSR_ACT_LRELU02;
        it.bufA = BUF_IM2COL; it.offA = 0;
        it.tRowA = tRowsLinear(it.M, 32, BM);
        it.tColA = tColsLinear(1, 1);
        it.bufB = BUF_WEIGHTS; it.offB = w.w;
        it.tRowB = tRowsLinear(it.N, it.K, BN);
        it.tColB = tColsLinear(1, 1);
        it.bufC = e1.buf; it.offC = 0;
        it.tRowC = tRowsAct(e1, idL, e1.H, e1.W, 1, BM, 0);
        it.tColC = tColsLinear(it.N / VSR_GG_KC, it.tilesN * BN / VSR_GG_KC);
        it.offBias = w.b;
        it.bufR = -1; it.tRowR = -1;
        op.flops = 2.0 * it.M * (double)it.N * 27;
        flops += op.flops;
        op.gemm.push_back(it);
        need(e1.buf, e1.elems());
        ops.push_back(std::move(op));
    }
    addConv("enc.2", e1, idL, e2, L, 3, 1, 1, model.enc[1], VSR_ACT_LRELU02, nullptr, nullptr);
    addConv("enc.3", e2, idL, e3, L, 3, 2, 1, model.enc[2], VSR_ACT_LRELU02, nullptr, nullptr);
    addConv("enc.4", e3, idL, feats, L, 3, 1, 1, model.enc[3], VSR_ACT_LRELU02, nullptr, nullptr);

    // The first transformer block of EVERY window reads the encoder features, and its q/k/v projection is a 1x1 conv: the same rows of
    // the same GEMM for a frame whichever window it is in.  With Tuning::shareQkv0 it runs here, once per frame of the chunk, and the
    // first block's attention of a window addresses its frames' rows in BUF_QKV0 (row tables by chunk frame id; read-only, so the
    // lanes share it).  Same products in the same order per output element: the same bits.  (Exact fp32 only, and only while the
    // buffer stays inside the kernels' 32-bit byte offsets.)
    qkv0_ = tu_.shareQkv0 && precision == 0 && g.blocks > 1 && (int64_t)L * g.featH * g.featW * 3 * g.channels * 4 < 4294967296LL;
    if (qkv0_) {
        const int fh = g.featH, fw = g.featW, C = g.channels;
        const BlockW& bw = model.blk[0];
        Op op;
        op.kind = OP_GEMM; op.tag = "attn.qkv"; op.tileCfg = tu_.qkvTile; op.bmode = VSR_BMODE_NK;
        int BM, BN;
        tileDims(op.tileCfg, BM, BN);
        GemmItem it{};
        it.M = L * fh * fw; it.N = 3 * C; it.K = C;
        it.tilesM = cdiv(it.M, BM); it.tilesN = cdiv(it.N, BN);
        it.splitK = 1; it.chunksPerSplit = it.K / VSR_GG_KC; it.alpha = 1.f; it.act = VSR_ACT_NONE;
        it.bufA = feats.buf; it.offA = 0;
        it.tRowA = tRowsAct(feats, idL, fh, fw, 1, BM, 0);
        it.tColA = tColsConv(feats, 1, 1);
        it.bufB = BUF_WEIGHTS; it.offB = bw.qkv.w;
        it.tRowB = tRowsLinear(it.N, it.K, BN);
        it.tColB = tColsLinear(it.K / VSR_GG_KC, it.K / VSR_GG_KC);
        it.bufC = BUF_QKV0; it.offC = 0;
        it.tRowC = tRowsLinear(it.M, 3 * C, BM);
        it.tColC = tColsLinear(it.N / VSR_GG_KC, it.tilesN * BN / VSR_GG_KC);
        it.offBias = bw.qkv.b;
        it.bufR = -1; it.tRowR = -1;
        op.flops = 2.0 * it.M * (double)it.N * it.K;
        flops += op.flops;
        trimmedFlops_ -= op.flops;              // (buildWindow adds what every window would have spent)
        op.gemm.push_back(it);
        need(BUF_QKV0, (int64_t)it.M * 3 * C);
        ops.push_back(std::move(op));
    }

    std::vector<int32_t> visits(L, 0);
    const int ns = g.neighborStride;
    for (int f = 0; f < L; f += ns) {
        std::vector<int> neighbors, refs;
        for (int i = (f - ns > 0 ? f - ns : 0); i < (f + ns + 1 < L ? f + ns + 1 : L); ++i) neighbors.push_back(i);
        for (int i = 0; i < L; i += g.refLength) { // get_ref_index (sttn_auto_inpaint.py:107-120)
            bool inN = false;
            for (int n : neighbors) inN |= (n == i);
            if (!inN) refs.push_back(i);
        }
        // windows are independent until OP_DECODE_OUT averages their frames into BUF_COMP (in window order): window w works in
        // lane w % lanes' buffers and is issued on that lane's stream
        lane_ = nwindows % lanes;
        const size_t first = ops.size();
        if (firstWindowOp < 0) firstWindowOp = (int)first;
        buildWindow(neighbors, refs, visits);
        for (size_t i = first; i < ops.size(); ++i) ops[i].lane = lane_;
    }
    lane_ = 0;
    compCount = visits;
    refFlops = flops + trimmedFlops_;
}

// ------------------------------------------------------------------------------------
void cv2_linear_tables(int ssize, int dsize, bool clampX, std::vector<int32_t>& ofs, std::vector<int16_t>& icoef,
                       std::vector<float>& fcoef)
{
    const double inv_scale = (double)dsize / (double)ssize;
    const double scale = 1.0 / inv_scale;
    ofs.resize(dsize);
    icoef.resize(2 * (size_t)dsize);
    fcoef.resize(2 * (size_t)dsize);
    for (int d = 0; d < dsize; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (clampX) {
            if (s < 0) { f = 0.f; s = 0; }
            if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
        }
        ofs[d] = s;
        const float c0 = 1.f - f, c1 = f;
        fcoef[2 * d] = c0;
        fcoef[2 * d + 1] = c1;
        auto sat = [](float v) {
            long r = lrintf(v * 2048.f); // saturate_cast<short>(float) = cvRound (nearest-even) + clamp
            if (r > 32767) r = 32767;
            if (r < -32768) r = -32768;
            return (int16_t)r;
        };
        icoef[2 * d] = sat(c0);
        icoef[2 * d + 1] = sat(c1);
    }
}

} // namespace vsr
