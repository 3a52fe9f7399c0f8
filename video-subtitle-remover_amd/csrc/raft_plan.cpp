// Weight packer + plan builder for RAFT (see raft_plan.h).
#include "raft_plan.h"
#include "gather_gemm.h"
#include <math.h>
#include <stdexcept>

namespace vsr {

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t rup(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
static std::vector<int> iota(int n)
{
    std::vector<int> v(n);
    for (int i = 0; i < n; ++i) v[i] = i;
    return v;
}
static void tileDims(int cfg, int& BM, int& BN)
{
    if (cfg == VSR_TILE_128x128) { BM = 128; BN = 128; }
    else if (cfg == VSR_TILE_128x64) { BM = 128; BN = 64; }
    else if (cfg == VSR_TILE_256x64) { BM = 256; BN = 64; }
    else { BM = 256; BN = 32; }
}

// ------------------------------------------------------------------------------------
// RaftModel: state_dict of raft-things.pth minus DataParallel's "module." prefix
// ------------------------------------------------------------------------------------
static const int kDims[3] = {64, 96, 128};

static void encoderKeys(const std::string& p, bool bn, std::vector<std::string>& k)
{
    auto conv = [&](const std::string& n) { k.push_back(p + n + ".weight"); k.push_back(p + n + ".bias"); };
    auto norm = [&](const std::string& n) {
        if (!bn) return;
        for (const char* leaf : {"weight", "bias", "running_mean", "running_var", "num_batches_tracked"}) k.push_back(p + n + "." + leaf);
    };
    norm("norm1");
    conv("conv1");
    for (int li = 0; li < 3; ++li)
        for (int bi = 0; bi < 2; ++bi) {
            const std::string b = "layer" + std::to_string(li + 1) + "." + std::to_string(bi) + ".";
            conv(b + "conv1");
            conv(b + "conv2");
            norm(b + "norm1");
            norm(b + "norm2");
            if (li > 0 && bi == 0) {
                norm(b + "norm3");
                conv(b + "downsample.0");
                norm(b + "downsample.1");
            }
        }
    conv("conv2");
}

std::vector<std::string> RaftModel::expected_keys()
{
    std::vector<std::string> k;
    encoderKeys("fnet.", false, k);
    encoderKeys("cnet.", true, k);
    for (const char* e : {"encoder.convc1", "encoder.convc2", "encoder.convf1", "encoder.convf2", "encoder.conv", "gru.convz1",
                          "gru.convr1", "gru.convq1", "gru.convz2", "gru.convr2", "gru.convq2", "flow_head.conv1",
                          "flow_head.conv2", "mask.0", "mask.2"}) {
        k.push_back(std::string("update_block.") + e + ".weight");
        k.push_back(std::string("update_block.") + e + ".bias");
    }
    return k;
}

RaftModel::RaftModel() {}

bool RaftModel::set_param(const std::string& name, const float* data, const int64_t* shape, int ndim, std::string& err)
{
    static const std::vector<std::string> keys = expected_keys();
    bool known = false;
    for (const auto& k : keys)
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

bool RaftModel::pack_conv(const std::string& key, ConvW& cw, int cout, int cin, int kh, int kw, const std::string& bn, float scale,
                          std::string& err)
{
    auto wi = raw_.find(key + ".weight"), bi = raw_.find(key + ".bias");
    if (wi == raw_.end() || bi == raw_.end()) { err = "missing key in state_dict: " + key; return false; }
    const Raw& w = wi->second;
    if (w.shape.size() != 4 || w.shape[0] != cout || w.shape[1] != cin || w.shape[2] != kh || w.shape[3] != kw ||
        (int64_t)bi->second.v.size() != cout) {
        err = "shape mismatch for " + key;
        return false;
    }
    // eval-mode BatchNorm2d folded into the conv: y = (conv(x) - mean) * gamma / sqrt(var + eps) + beta  (extractor.py:24-28)
    std::vector<float> s(cout, scale), sh(cout, 0.f);
    if (!bn.empty()) {
        const char* leaves[4] = {"weight", "bias", "running_mean", "running_var"};
        const Raw* r[4];
        for (int i = 0; i < 4; ++i) {
            auto it = raw_.find(bn + "." + leaves[i]);
            if (it == raw_.end() || (int64_t)it->second.v.size() != cout) { err = "missing / bad BatchNorm entry: " + bn + "." + leaves[i]; return false; }
            r[i] = &it->second;
        }
        for (int n = 0; n < cout; ++n) {
            const float g = r[0]->v[n] / sqrtf(r[3]->v[n] + 1e-5f);
            s[n] = g * scale;
            sh[n] = (r[1]->v[n] - r[2]->v[n] * g) * scale;
        }
    }
    const int K = (int)rup((int64_t)kh * kw * cin, VSR_GG_KC);
    cw.cout = cout;
    cw.K = K;
    cw.w = (int64_t)packed.size();
    packed.resize(packed.size() + (size_t)rup((int64_t)cout * K, 32), 0.f);
    float* dst = packed.data() + cw.w;
    const bool chanMajor = Tuning::get().convChannelMajor && (cin % VSR_GG_KC == 0);   // mirrors PlanBuilder::tColsConvHW
    const int taps = kh * kw;
    for (int n = 0; n < cout; ++n)
        for (int ci = 0; ci < cin; ++ci)
            for (int tap = 0; tap < taps; ++tap) {
                const int k = chanMajor ? ((ci / VSR_GG_KC) * taps + tap) * VSR_GG_KC + (ci % VSR_GG_KC) : tap * cin + ci;
                dst[(int64_t)n * K + k] = w.v[((int64_t)n * cin + ci) * taps + tap] * s[n];
            }
    cw.b = (int64_t)packed.size();
    packed.resize(packed.size() + (size_t)rup(cout, 32), 0.f);
    for (int n = 0; n < cout; ++n) packed[cw.b + n] = bi->second.v[n] * s[n] + sh[n];
    return true;
}

bool RaftModel::pack_encoder(const std::string& p, RaftEncW& e, bool bn, std::string& err)
{
    auto nm = [&](const std::string& n) { return bn ? p + n : std::string(); };
    if (!pack_conv(p + "conv1", e.conv1, 64, 3, 7, 7, nm("norm1"), 1.f, err)) return false;
    int cin = 64;
    for (int li = 0; li < 3; ++li) {
        const int dim = kDims[li];
        for (int bi = 0; bi < 2; ++bi) {
            const std::string b = "layer" + std::to_string(li + 1) + "." + std::to_string(bi) + ".";
            if (!pack_conv(p + b + "conv1", e.b1[li][bi], dim, cin, 3, 3, nm(b + "norm1"), 1.f, err)) return false;
            if (!pack_conv(p + b + "conv2", e.b2[li][bi], dim, dim, 3, 3, nm(b + "norm2"), 1.f, err)) return false;
            if (li > 0 && bi == 0)
                if (!pack_conv(p + b + "downsample.0", e.ds[li], dim, cin, 1, 1, nm(b + "downsample.1"), 1.f, err)) return false;
            cin = dim;
        }
    }
    return pack_conv(p + "conv2", e.conv2, 256, 128, 1, 1, "", 1.f, err);
}

bool RaftModel::fuse_rows(const ConvW& a, const ConvW& b, ConvW& out)
{
    if (a.K != b.K) return false;
    out.cout = a.cout + b.cout;
    out.K = a.K;
    out.w = (int64_t)packed.size();
    packed.resize(packed.size() + (size_t)rup((int64_t)out.cout * out.K, 32), 0.f);
    out.b = (int64_t)packed.size();
    packed.resize(packed.size() + (size_t)rup(out.cout, 32), 0.f);
    for (int64_t i = 0; i < (int64_t)a.cout * a.K; ++i) packed[out.w + i] = packed[a.w + i];
    for (int64_t i = 0; i < (int64_t)b.cout * b.K; ++i) packed[out.w + (int64_t)a.cout * a.K + i] = packed[b.w + i];
    for (int i = 0; i < a.cout; ++i) packed[out.b + i] = packed[a.b + i];
    for (int i = 0; i < b.cout; ++i) packed[out.b + a.cout + i] = packed[b.b + i];
    return true;
}

bool RaftModel::slice_cin(const ConvW& src, int cin, int taps, int c0, int n, int c1, int n1, bool keepBias, ConvW& out)
{
    // channel-major K order (pack_conv): k = ((ci / 32) * taps + tap) * 32 + ci % 32 -- a 32-channel block is `taps` whole chunks
    if (!Tuning::get().convChannelMajor || cin % VSR_GG_KC || c0 % VSR_GG_KC || n % VSR_GG_KC || c1 % VSR_GG_KC || n1 % VSR_GG_KC) return false;
    if (src.K != taps * cin) return false;
    const int K = taps * (n + n1);
    out.cout = src.cout;
    out.K = K;
    out.w = (int64_t)packed.size();
    packed.resize(packed.size() + (size_t)rup((int64_t)out.cout * K, 32), 0.f);
    for (int row = 0; row < src.cout; ++row) {
        int64_t d = out.w + (int64_t)row * K;
        for (int part = 0; part < 2; ++part) {
            const int cb = part ? c1 : c0, cn = part ? n1 : n;
            const int64_t from = src.w + (int64_t)row * src.K + (int64_t)(cb / VSR_GG_KC) * taps * VSR_GG_KC;
            for (int64_t i = 0; i < (int64_t)(cn / VSR_GG_KC) * taps * VSR_GG_KC; ++i) packed[d++] = packed[from + i];
        }
    }
    out.b = -1;
    if (keepBias) {
        out.b = (int64_t)packed.size();
        packed.resize(packed.size() + (size_t)rup(out.cout, 32), 0.f);
        for (int i = 0; i < out.cout; ++i) packed[out.b + i] = packed[src.b + i];
    }
    return true;
}

bool RaftModel::pack(std::string& err)
{
    packed.clear();
    ready_ = false;
    for (const auto& k : expected_keys())
        if (!raw_.count(k)) { err = "missing key in state_dict: " + k; return false; }
    if (!pack_encoder("fnet.", fnet, false, err) || !pack_encoder("cnet.", cnet, true, err)) return false;
    const std::string u = "update_block.";
    if (!pack_conv(u + "encoder.convc1", convc1, 256, 324, 1, 1, "", 1.f, err)) return false;
    if (!pack_conv(u + "encoder.convc2", convc2, 192, 256, 3, 3, "", 1.f, err)) return false;
    if (!pack_conv(u + "encoder.convf1", convf1, 128, 2, 7, 7, "", 1.f, err)) return false;
    if (!pack_conv(u + "encoder.convf2", convf2, 64, 128, 3, 3, "", 1.f, err)) return false;
    if (!pack_conv(u + "encoder.conv", conv, 126, 256, 3, 3, "", 1.f, err)) return false;
    for (int pass = 0; pass < 2; ++pass) {          // SepConvGRU: (1,5) then (5,1)  (update.py:33-60); z and r share their input
        const std::string s = std::to_string(pass + 1);
        const int kh = pass == 0 ? 1 : 5, kw = pass == 0 ? 5 : 1;
        ConvW z, r;
        if (!pack_conv(u + "gru.convz" + s, z, 128, 384, kh, kw, "", 1.f, err)) return false;
        if (!pack_conv(u + "gru.convr" + s, r, 128, 384, kh, kw, "", 1.f, err)) return false;
        if (!fuse_rows(z, r, zr[pass])) { err = "gru z/r fuse"; return false; }
        if (!pack_conv(u + "gru.convq" + s, q[pass], 128, 384, kh, kw, "", 1.f, err)) return false;
        // input channels: 0..127 h (or r*h), 128..255 inp (the context), 256..383 motion features
        if (Tuning::get().convChannelMajor &&
            (!slice_cin(zr[pass], 384, 5, 128, 128, 0, 0, true, zrCtx[pass]) || !slice_cin(zr[pass], 384, 5, 0, 128, 256, 128, false, zrVar[pass]) ||
             !slice_cin(q[pass], 384, 5, 128, 128, 0, 0, true, qCtx[pass]) || !slice_cin(q[pass], 384, 5, 0, 128, 256, 128, false, qVar[pass]))) {
            err = "gru context slices";
            return false;
        }
    }
    if (!pack_conv(u + "flow_head.conv1", fh1, 256, 128, 3, 3, "", 1.f, err)) return false;
    if (!pack_conv(u + "flow_head.conv2", fh2, 2, 256, 3, 3, "", 1.f, err)) return false;
    if (!pack_conv(u + "mask.0", mask1, 256, 128, 3, 3, "", 1.f, err)) return false;
    if (!pack_conv(u + "mask.2", mask2, 576, 256, 1, 1, "", 0.25f, err)) return false;   // mask = .25 * self.mask(net)  (update.py:136)
    ready_ = true;
    return true;
}

// ------------------------------------------------------------------------------------
// RaftPlan
// ------------------------------------------------------------------------------------
int RaftPlan::pickTile(int N) const { return N <= 32 ? VSR_TILE_256x32 : (N <= 64 ? n64Tile() : VSR_TILE_128x64); }

Op& RaftPlan::ew(int kind, const char* tag)
{
    Op op;
    op.kind = OP_EW;
    op.ew = kind;
    op.tag = tag;
    ops.push_back(std::move(op));
    return ops.back();
}

void RaftPlan::conv(const char* tag, const Act& in, const std::vector<int>& inIds, int c0in, int cin, const Act& out, int c0out,
                    int nOut, int kh, int kw, int stride, const ConvW& w, int act, const Act* res, const std::vector<int>* chunks)
{
    const int K = kh * kw * cin;
    if (w.K != K) throw std::runtime_error(std::string("raft conv K mismatch: ") + tag);
    if ((int)inIds.size() != nOut) throw std::runtime_error("raft conv frame list mismatch");
    Op op;
    op.kind = OP_GEMM;
    op.tag = tag;
    op.bmode = VSR_BMODE_NK;
    op.tileCfg = pickTile(w.cout);
    int BM, BN;
    tileDims(op.tileCfg, BM, BN);
    GemmItem it{};
    it.M = nOut * out.H * out.W;
    it.N = w.cout;
    it.K = K;
    it.tilesM = cdiv(it.M, BM);
    it.tilesN = cdiv(it.N, BN);
    it.splitK = 1;
    it.chunksPerSplit = K / VSR_GG_KC;
    it.alpha = 1.f;
    it.act = act;
    it.bufA = in.buf; it.offA = 0;
    it.tRowA = tRowsAct(in, inIds, out.H, out.W, stride, BM, 0);
    if (chunks) {       // logical 32-channel chunk c of the conv input lives at physical channel (*chunks)[c]
        std::string key = "CCL:" + std::to_string(in.halo) + ":" + std::to_string(in.W) + ":" + std::to_string(in.C) + ":" +
                          std::to_string(kh) + "x" + std::to_string(kw) + ":";
        for (int c : *chunks) key += std::to_string(c) + ",";
        if (in.halo < kh / 2 || in.halo < kw / 2) throw std::runtime_error("activation halo too small for conv");
        std::vector<int32_t> v;
        for (int c : *chunks)       // channel-major K order (cin % 32 == 0), mirrors RaftModel::pack_conv
            for (int ky = 0; ky < kh; ++ky)
                for (int kx = 0; kx < kw; ++kx)
                    v.push_back((int32_t)(((int64_t)(ky - kh / 2) * in.Wp() + (kx - kw / 2)) * in.C + c));
        if (!Tuning::get().convChannelMajor) throw std::runtime_error("chunk-list convs need the channel-major K order");
        it.tColA = table(key, std::move(v));
    } else {
        it.tColA = tColsConvHW(in, kh, kw, 1, c0in, cin);
    }
    it.bufB = RB_WEIGHTS; it.offB = w.w;
    it.tRowB = tRowsLinear(it.N, it.K, BN);
    it.tColB = tColsLinear(it.K / VSR_GG_KC, it.K / VSR_GG_KC);
    it.bufC = out.buf; it.offC = 0;
    it.tRowC = tRowsAct(out, iota(nOut), out.H, out.W, 1, BM, c0out);
    it.tColC = tColsLinear(cdiv(it.N, VSR_GG_KC), it.tilesN * BN / VSR_GG_KC);
    it.offBias = w.b;
    if (res) {
        it.bufR = res->buf; it.offR = 0;
        it.tRowR = tRowsAct(*res, iota(nOut), out.H, out.W, 1, BM, 0);
    } else {
        it.bufR = -1; it.offR = 0; it.tRowR = -1;
    }
    op.flops = 2.0 * it.M * it.N * (double)K;
    op.gemm.push_back(it);
    need(out.buf, (int64_t)nOut * out.frameElems());
    flops += op.flops;
    ops.push_back(std::move(op));
}

void RaftPlan::inorm(const Act& x, bool relu, const Act* res)
{
    Op& s = ew(EW_INORM_STATS, "inorm.stats");
    s.ibuf[0] = x.buf; s.ibuf[1] = RB_STATS;
    s.ipar[0] = x.n; s.ipar[1] = x.H; s.ipar[2] = x.W; s.ipar[3] = x.C; s.ipar[4] = x.halo;
    Op& a = ew(EW_INORM_APPLY, "inorm.apply");
    a.ibuf[0] = x.buf; a.ibuf[1] = RB_STATS; a.ibuf[2] = res ? res->buf : -1;
    a.ipar[0] = x.n; a.ipar[1] = x.H; a.ipar[2] = x.W; a.ipar[3] = x.C; a.ipar[4] = x.halo;
    a.ipar[5] = relu ? 1 : 0; a.ipar[6] = res ? res->halo : 0;
    need(RB_STATS, (int64_t)x.n * x.C * 2);
}

// BasicEncoder.forward (extractor.py:166-192).  instanceNorm (fnet): conv -> statistics -> normalise(+ReLU)(+residual+ReLU)
// passes; otherwise (cnet) BatchNorm is folded into the weights and ReLU / residual / ReLU ride in the GEMM epilogue.
void RaftPlan::encoder(const RaftEncW& e, bool inst, int outBuf)
{
    const std::vector<int> idsT = iota(t);
    const int stageBuf[3][3] = {{RB_S1A, RB_S1B, RB_S1C}, {RB_S2A, RB_S2B, RB_S2C}, {RB_S3A, RB_S3B, RB_S3C}};
    auto stage = [&](int li, int j) {
        const int div = 2 << li;
        return Act{stageBuf[li][j], t, H / div, W / div, kDims[li], 1};
    };
    const Act cols{RB_IM2COL, t, H / 2, W / 2, 160, 0};
    Act x = stage(0, 0);
    conv(inst ? "fnet.stem" : "cnet.stem", cols, idsT, 0, 160, x, 0, t, 1, 1, 1, e.conv1, inst ? VSR_ACT_NONE : VSR_ACT_RELU, nullptr);
    if (inst) inorm(x, true, nullptr);
    const char* tg = inst ? "fnet.res" : "cnet.res";
    for (int li = 0; li < 3; ++li) {
        const bool down = li > 0;
        // block 0: stride 2 (with a 1x1 stride-2 shortcut) on layers 2, 3
        Act y = stage(li, down ? 0 : 1), z = stage(li, 2), res = x;
        conv(tg, x, idsT, 0, x.C, y, 0, t, 3, 3, down ? 2 : 1, e.b1[li][0], inst ? VSR_ACT_NONE : VSR_ACT_RELU, nullptr);
        if (inst) inorm(y, true, nullptr);
        if (down) {
            res = stage(li, 1);
            conv(tg, x, idsT, 0, x.C, res, 0, t, 1, 1, 2, e.ds[li], VSR_ACT_NONE, nullptr);
            if (inst) inorm(res, false, nullptr);
        }
        conv(tg, y, idsT, 0, y.C, z, 0, t, 3, 3, 1, e.b2[li][0], inst ? VSR_ACT_NONE : (VSR_ACT_RELU | VSR_ACT_POST_RELU), inst ? nullptr : &res);
        if (inst) inorm(z, true, &res);
        // block 1
        Act y1 = stage(li, 0), z1 = stage(li, 1);
        conv(tg, z, idsT, 0, z.C, y1, 0, t, 3, 3, 1, e.b1[li][1], inst ? VSR_ACT_NONE : VSR_ACT_RELU, nullptr);
        if (inst) inorm(y1, true, nullptr);
        conv(tg, y1, idsT, 0, y1.C, z1, 0, t, 3, 3, 1, e.b2[li][1], inst ? VSR_ACT_NONE : (VSR_ACT_RELU | VSR_ACT_POST_RELU), inst ? nullptr : &z);
        if (inst) inorm(z1, true, &z);
        x = z1;
    }
    const Act out{outBuf, t, h8, w8, 256, 0};
    conv(inst ? "fnet.out" : "cnet.out", x, idsT, 0, 128, out, 0, t, 1, 1, 1, e.conv2, VSR_ACT_NONE, nullptr);
}

RaftPlan::RaftPlan(const RaftModel& model, int t_, int H_, int W_, int iters_)
    : t(t_), H(H_), W(W_), iters(iters_), pairs(2 * (t_ - 1)), h8(H_ / 8), w8(W_ / 8), m_(model)
{
    if (!model.packed_ready()) throw std::runtime_error("RAFT model is not packed");
    if (t < 2 || iters < 1) throw std::runtime_error("RAFT needs at least two frames and one iteration");
    if (H % 8 || W % 8 || h8 < 16 || w8 < 16) throw std::runtime_error("RAFT frame size must be a multiple of 8 and at least 128x128");
    const int hw = h8 * w8;
    if ((int64_t)hw * hw > 2147483647LL) throw std::runtime_error("correlation volume row offsets exceed int32");
    bufElems.assign(RB_COUNT, 0);
    bufElems[RB_WEIGHTS] = (int64_t)model.packed.size();
    const std::vector<int> idsP = iota(pairs);
    const int M = pairs * hw;

    {   // stem im2col, shared by both encoders
        Op& op = ew(EW_IM2COL7_U8, "stem.im2col");
        op.ibuf[0] = RB_IN_U8; op.ibuf[1] = RB_IM2COL;
        op.ipar[0] = t; op.ipar[1] = H; op.ipar[2] = W;
        need(RB_IN_U8, (int64_t)t * H * W * 3);
        need(RB_IM2COL, (int64_t)t * (H / 2) * (W / 2) * 160);
    }
    encoder(model.fnet, true, RB_FMAP);
    encoder(model.cnet, false, RB_CMAP);

    // frame pair of every pair-direction (flow_comp_raft.py:45-49): forward p -> p+1, backward p+1 -> p
    std::vector<int32_t> fa(pairs), fb(pairs);
    for (int p = 0; p < pairs; ++p) {
        const int q = p < t - 1 ? p : p - (t - 1);
        fa[p] = p < t - 1 ? q : q + 1;
        fb[p] = p < t - 1 ? q + 1 : q;
    }
    const int tFrameA = table("PAIRA:" + std::to_string(t), std::vector<int32_t>(fa));

    {   // all-pairs correlation (corr.py:53-60): one problem per pair-direction, level 0 of the pyramid
        Op op;
        op.kind = OP_GEMM; op.tag = "corr.volume"; op.bmode = VSR_BMODE_NK; op.tileCfg = VSR_TILE_128x64;
        int BM, BN;
        tileDims(op.tileCfg, BM, BN);
        // The backward pair-direction of frames (q, q+1) correlates f(q+1) with f(q): its volume is the TRANSPOSE of the forward one's,
        // C_b[i][j] = f(q+1)_i . f(q)_j = C_f[j][i] -- the same products added in the same k order, so the same bits (a*b = b*a).  Half of the
        // correlation GEMMs (4 of RAFT's 151 TFLOP per 68-frame batch) become a tiled transpose pass.  VSR_RAFT_CORR_TRANSPOSE=0: 2 (t-1) GEMMs.
        static const bool transposeEnv = [] { const char* e = getenv("VSR_RAFT_CORR_TRANSPOSE"); return !(e && atoi(e) == 0); }();
        const int nGemm = transposeEnv ? t - 1 : pairs;
        for (int p = 0; p < nGemm; ++p) {
            GemmItem it{};
            it.M = hw; it.N = hw; it.K = 256;
            it.tilesM = cdiv(hw, BM); it.tilesN = cdiv(hw, BN);
            it.splitK = 1; it.chunksPerSplit = 8; it.alpha = 1.f / 16.f; it.act = VSR_ACT_NONE;   // / sqrt(dim = 256)
            it.bufA = RB_FMAP; it.offA = (int64_t)fa[p] * hw * 256;
            it.tRowA = tRowsLinear(hw, 256, BM);
            it.tColA = tColsLinear(8, 8);
            it.bufB = RB_FMAP; it.offB = (int64_t)fb[p] * hw * 256;
            it.tRowB = tRowsLinear(hw, 256, BN);
            it.tColB = it.tColA;
            it.bufC = RB_PYR; it.offC = (int64_t)p * hw * hw;
            it.tRowC = tRowsLinear(hw, hw, BM);
            it.tColC = tColsLinear(cdiv(hw, VSR_GG_KC), it.tilesN * BN / VSR_GG_KC);
            it.bufR = -1; it.tRowR = -1; it.offBias = -1;
            op.gemm.push_back(it);
            op.flops += 2.0 * hw * (double)hw * 256;
        }
        flops += op.flops;
        ops.push_back(std::move(op));
        if (nGemm < pairs) {
            Op& tr = ew(EW_CORR_TRANSPOSE, "corr.transpose");
            tr.ibuf[0] = RB_PYR; tr.ioff[0] = 0; tr.ioff[1] = (int64_t)(t - 1) * hw * hw;
            tr.ipar[0] = t - 1; tr.ipar[1] = hw;
        }
    }
    lvlH[0] = h8; lvlW[0] = w8; lvlOff[0] = 0;
    for (int l = 1; l < 4; ++l) {
        lvlH[l] = lvlH[l - 1] / 2; lvlW[l] = lvlW[l - 1] / 2;
        lvlOff[l] = lvlOff[l - 1] + rup((int64_t)M * lvlH[l - 1] * lvlW[l - 1], 32);
        Op& op = ew(EW_AVGPOOL2, "corr.pool");
        op.ibuf[0] = RB_PYR; op.ioff[0] = lvlOff[l - 1]; op.ioff[1] = lvlOff[l];
        op.ipar[0] = M; op.ipar[1] = lvlH[l - 1]; op.ipar[2] = lvlW[l - 1];
    }
    need(RB_PYR, lvlOff[3] + (int64_t)M * lvlH[3] * lvlW[3]);

    // ---- recurrent state: one NHWC buffer [h | inp | motion(126) flow(2) | r*h] so that the GRU convs gather their
    // 384-channel inputs (cat[h, x] and cat[r*h, x], update.py:47-59) without a concatenation copy
    const Act hxr{RB_HXR, pairs, h8, w8, 512, 2};
    const Act corrf{RB_CORRF, pairs, h8, w8, 352, 0}, c1{RB_C1, pairs, h8, w8, 256, 1}, corflo{RB_CORFLO, pairs, h8, w8, 256, 1};
    const Act flowcol{RB_FLOWCOL, pairs, h8, w8, 128, 0}, f1{RB_F1, pairs, h8, w8, 128, 1};
    const Act zr{RB_ZR, pairs, h8, w8, 256, 0}, qb{RB_Q, pairs, h8, w8, 128, 0};
    const Act fh{RB_FH1, pairs, h8, w8, 256, 1}, delta{RB_DELTA, pairs, h8, w8, 32, 0};
    const Act maskh{RB_MASKH, pairs, h8, w8, 256, 0}, maskb{RB_MASK, pairs, h8, w8, 576, 0};
    need(RB_HXR, hxr.elems());
    need(RB_COORDS, (int64_t)M * 2);
    need(RB_FLOW, (int64_t)M * 2);
    need(RB_DELTA, delta.elems());
    {
        Op& op = ew(EW_CTX_SPLIT, "ctx.split");      // net = tanh(cnet[:128]), inp = relu(cnet[128:])  (raft.py:114-116)
        op.ibuf[0] = RB_CMAP; op.ibuf[1] = RB_HXR;
        op.ipar[0] = pairs; op.ipar[1] = h8; op.ipar[2] = w8; op.ipar[3] = hxr.halo; op.ipar[4] = hxr.C; op.ipar[5] = tFrameA;
    }
    auto flowUpdate = [&](int init) {
        Op& op = ew(EW_FLOW_UPDATE, init ? "flow.init" : "flow.update");
        op.ibuf[0] = RB_DELTA; op.ibuf[1] = RB_COORDS; op.ibuf[2] = RB_FLOW; op.ibuf[3] = RB_HXR;
        op.ipar[0] = pairs; op.ipar[1] = h8; op.ipar[2] = w8; op.ipar[3] = init; op.ipar[4] = hxr.halo; op.ipar[5] = hxr.C;
        op.ipar[6] = 382; op.ipar[7] = delta.C;
    };
    flowUpdate(1);
    std::vector<int> chunksQ;                       // cat[r*h, x]: r*h at channels 384.., x at 128..383
    for (int c = 0; c < 4; ++c) chunksQ.push_back(384 + 32 * c);
    for (int c = 4; c < 12; ++c) chunksQ.push_back(32 * c);
    // The context's share of the GRU convs, once per call.  x = cat[inp, motion] (update.py:128) and inp = relu(cnet[128:]) never changes
    // over the iterations (raft.py:114-116,127-133), and a conv is linear in its input channels: conv(cat[h, inp, motion]) =
    // conv_inp(inp) + bias + conv_rest(cat[h, motion]).  The four context products (z|r and q of both passes, K = 5 x 128) are computed
    // here and ride into the per-iteration convs as their residual operand; those contract K = 5 x 256 instead of 5 x 384: a third of
    // the GRU's FLOPs (33 of RAFT's 178 TFLOP per 68-frame batch) is not recomputed twenty times.  Same sums, another association:
    // fp32 results move by rounding (the parity bars are unchanged, tests/test_gpu_raft.py).  VSR_RAFT_CTX_HOIST=0: the plain convs.
    static const bool hoistEnv = [] { const char* e = getenv("VSR_RAFT_CTX_HOIST"); return !(e && atoi(e) == 0); }();
    const bool ctxHoist = hoistEnv && Tuning::get().convChannelMajor;
    const Act zrCtx[2] = {{RB_ZRCTX0, pairs, h8, w8, 256, 0}, {RB_ZRCTX1, pairs, h8, w8, 256, 0}};
    const Act qCtx[2] = {{RB_QCTX0, pairs, h8, w8, 128, 0}, {RB_QCTX1, pairs, h8, w8, 128, 0}};
    std::vector<int> chunksZRv, chunksQv;           // the per-iteration inputs without the context: [h | motion], [r*h | motion]
    for (int c = 0; c < 4; ++c) { chunksZRv.push_back(32 * c); chunksQv.push_back(384 + 32 * c); }
    for (int c = 8; c < 12; ++c) { chunksZRv.push_back(32 * c); chunksQv.push_back(32 * c); }
    if (ctxHoist)
        for (int pass = 0; pass < 2; ++pass) {
            const int kh = pass == 0 ? 1 : 5, kw = pass == 0 ? 5 : 1;
            conv("gru.zr.ctx", hxr, idsP, 128, 128, zrCtx[pass], 0, pairs, kh, kw, 1, m_.zrCtx[pass], VSR_ACT_NONE, nullptr);
            conv("gru.q.ctx", hxr, idsP, 128, 128, qCtx[pass], 0, pairs, kh, kw, 1, m_.qCtx[pass], VSR_ACT_NONE, nullptr);
        }
    for (int it = 0; it < iters; ++it) {
        {
            Op& op = ew(EW_CORR_LOOKUP, "corr.lookup");
            op.ibuf[0] = RB_PYR; op.ibuf[1] = RB_COORDS; op.ibuf[2] = RB_CORRF;
            for (int l = 0; l < 4; ++l) { op.ioff[l] = lvlOff[l]; op.ipar[1 + l] = lvlH[l]; op.ipar[5 + l] = lvlW[l]; }
            op.ipar[0] = M; op.ipar[9] = corrf.C;
            need(RB_CORRF, corrf.elems());
        }
        // BasicMotionEncoder (update.py:79-98)
        conv("upd.convc1", corrf, idsP, 0, 352, c1, 0, pairs, 1, 1, 1, m_.convc1, VSR_ACT_RELU, nullptr);
        conv("upd.convc2", c1, idsP, 0, 256, corflo, 0, pairs, 3, 3, 1, m_.convc2, VSR_ACT_RELU, nullptr);
        {
            Op& op = ew(EW_IM2COL7_FLOW, "flow.im2col");
            op.ibuf[0] = RB_FLOW; op.ibuf[1] = RB_FLOWCOL;
            op.ipar[0] = pairs; op.ipar[1] = h8; op.ipar[2] = w8;
            need(RB_FLOWCOL, flowcol.elems());
        }
        conv("upd.convf1", flowcol, idsP, 0, 128, f1, 0, pairs, 1, 1, 1, m_.convf1, VSR_ACT_RELU, nullptr);
        conv("upd.convf2", f1, idsP, 0, 128, corflo, 192, pairs, 3, 3, 1, m_.convf2, VSR_ACT_RELU, nullptr);
        conv("upd.conv", corflo, idsP, 0, 256, hxr, 256, pairs, 3, 3, 1, m_.conv, VSR_ACT_RELU, nullptr);
        // SepConvGRU (update.py:33-60)
        for (int pass = 0; pass < 2; ++pass) {
            const int kh = pass == 0 ? 1 : 5, kw = pass == 0 ? 5 : 1;
            if (ctxHoist) conv("gru.zr", hxr, idsP, 0, 256, zr, 0, pairs, kh, kw, 1, m_.zrVar[pass], VSR_ACT_NONE, &zrCtx[pass], &chunksZRv);
            else
            conv("gru.zr", hxr, idsP, 0, 384, zr, 0, pairs, kh, kw, 1, m_.zr[pass], VSR_ACT_NONE, nullptr);
            {
                Op& op = ew(EW_GRU_RH, "gru.rh");
                op.ibuf[0] = RB_ZR; op.ibuf[1] = RB_HXR;
                op.ipar[0] = pairs; op.ipar[1] = h8; op.ipar[2] = w8; op.ipar[3] = hxr.halo; op.ipar[4] = hxr.C; op.ipar[5] = 0; op.ipar[6] = 384;
            }
            if (ctxHoist) conv("gru.q", hxr, idsP, 0, 256, qb, 0, pairs, kh, kw, 1, m_.qVar[pass], VSR_ACT_NONE, &qCtx[pass], &chunksQv);
            else
            conv("gru.q", hxr, idsP, 0, 384, qb, 0, pairs, kh, kw, 1, m_.q[pass], VSR_ACT_NONE, nullptr, &chunksQ);
            {
                Op& op = ew(EW_GRU_UPDATE, "gru.update");
                op.ibuf[0] = RB_ZR; op.ibuf[1] = RB_Q; op.ibuf[2] = RB_HXR;
                op.ipar[0] = pairs; op.ipar[1] = h8; op.ipar[2] = w8; op.ipar[3] = hxr.halo; op.ipar[4] = hxr.C; op.ipar[5] = 0;
            }
        }
        // FlowHead (update.py:6-14)
        conv("upd.fh1", hxr, idsP, 0, 128, fh, 0, pairs, 3, 3, 1, m_.fh1, VSR_ACT_RELU, nullptr);
        conv("upd.fh2", fh, idsP, 0, 256, delta, 0, pairs, 3, 3, 1, m_.fh2, VSR_ACT_NONE, nullptr);
        flowUpdate(0);
    }
    // upsampling mask of the last iteration only (raft.py:139-144 keeps just the final flow_up in test mode)
    conv("upd.mask1", hxr, idsP, 0, 128, maskh, 0, pairs, 3, 3, 1, m_.mask1, VSR_ACT_RELU, nullptr);
    conv("upd.mask2", maskh, idsP, 0, 256, maskb, 0, pairs, 1, 1, 1, m_.mask2, VSR_ACT_NONE, nullptr);
    {
        Op& op = ew(EW_CONVEX_UP, "flow.upsample");
        op.ibuf[0] = RB_FLOW; op.ibuf[1] = RB_MASK; op.ibuf[2] = RB_OUT;
        op.ipar[0] = pairs; op.ipar[1] = h8; op.ipar[2] = w8;
        need(RB_OUT, (int64_t)pairs * 2 * H * W);
    }
}

} // namespace vsr
