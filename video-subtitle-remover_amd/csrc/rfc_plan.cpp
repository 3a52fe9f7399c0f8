// Weight packer + plan builder for the recurrent flow completion network (see rfc_plan.h).
#include "rfc_plan.h"
#include <algorithm>
#include <stdlib.h>
#include "gather_gemm.h"
#include <stdexcept>

namespace vsr {

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t rup(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
static void tileDims(int cfg, int& BM, int& BN)
{
    if (cfg == VSR_TILE_128x128) { BM = 128; BN = 128; }
    else if (cfg == VSR_TILE_128x64) { BM = 128; BN = 64; }
    else if (cfg == VSR_TILE_256x64) { BM = 256; BN = 64; }
    else { BM = 256; BN = 32; }
}

// ------------------------------------------------------------------------------------
// RfcModel
// ------------------------------------------------------------------------------------
static const char* kMods[2] = {"backward_", "forward_"};

std::vector<std::string> RfcModel::expected_keys()
{
    std::vector<std::string> k;
    auto add = [&](const std::string& n) { k.push_back(n + ".weight"); k.push_back(n + ".bias"); };
    add("downsample.0");
    for (const char* e : {"encoder1.0", "encoder1.2", "encoder2.0", "encoder2.2"}) {
        add(std::string(e) + ".conv1.0");
        add(std::string(e) + ".conv2.0");
    }
    for (const char* e : {"mid_dilation.0", "mid_dilation.2", "mid_dilation.4"}) add(e);
    for (int m = 0; m < 2; ++m) {
        const std::string p = std::string("feat_prop_module.deform_align.") + kMods[m];
        add(p);
        for (const char* e : {".conv_offset.0", ".conv_offset.2", ".conv_offset.4", ".conv_offset.6"}) add(p + e);
    }
    for (int m = 0; m < 2; ++m) {
        add(std::string("feat_prop_module.backbone.") + kMods[m] + ".0");
        add(std::string("feat_prop_module.backbone.") + kMods[m] + ".2");
    }
    add("feat_prop_module.fusion");
    for (const char* e : {"decoder2.0", "decoder2.2.conv", "decoder1.0", "decoder1.2.conv", "upsample.0", "upsample.2.conv",
                          "edgeDetector.projection.0", "edgeDetector.mid_layer_1.0", "edgeDetector.mid_layer_2.0", "edgeDetector.out_layer"})
        add(e);
    return k;
}

bool RfcModel::set_param(const std::string& name, const float* data, const int64_t* shape, int ndim, std::string& err)
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

bool RfcModel::pack_conv(const std::string& key, ConvW& cw, int cout, int cin, int taps, std::string& err)
{
    auto wi = raw_.find(key + ".weight"), bi = raw_.find(key + ".bias");
    if (wi == raw_.end() || bi == raw_.end()) { err = "missing key in state_dict: " + key; return false; }
    const Raw& w = wi->second;
    int64_t kprod = 1;
    for (size_t i = 2; i < w.shape.size(); ++i) kprod *= w.shape[i];
    if (w.shape.size() < 4 || w.shape[0] != cout || w.shape[1] != cin || kprod != taps || (int64_t)bi->second.v.size() != cout) {
        err = "shape mismatch for " + key;
        return false;
    }
    const int K = (int)rup((int64_t)taps * cin, VSR_GG_KC);
    cw.cout = cout;
    cw.K = K;
    cw.w = (int64_t)packed.size();
    packed.resize(packed.size() + (size_t)rup((int64_t)cout * K, 32), 0.f);
    float* dst = packed.data() + cw.w;
    const bool chanMajor = Tuning::get().convChannelMajor && (cin % VSR_GG_KC == 0);   // mirrors the col tables of RfcPlan
    for (int n = 0; n < cout; ++n)
        for (int ci = 0; ci < cin; ++ci)
            for (int tap = 0; tap < taps; ++tap) {
                const int k = chanMajor ? ((ci / VSR_GG_KC) * taps + tap) * VSR_GG_KC + (ci % VSR_GG_KC) : tap * cin + ci;
                dst[(int64_t)n * K + k] = w.v[((int64_t)n * cin + ci) * taps + tap];
            }
    cw.b = (int64_t)packed.size();
    packed.resize(packed.size() + (size_t)rup(cout, 32), 0.f);
    for (int n = 0; n < cout; ++n) packed[cw.b + n] = bi->second.v[n];
    return true;
}

bool RfcModel::pack(std::string& err)
{
    packed.clear();
    ready_ = false;
    for (const auto& k : expected_keys())
        if (!raw_.count(k)) { err = "missing key in state_dict: " + k; return false; }
    if (!pack_conv("downsample.0", down, 32, 3, 25, err)) return false;
    static const char* enc[4] = {"encoder1.0", "encoder1.2", "encoder2.0", "encoder2.2"};
    static const int ch[4][2] = {{32, 32}, {32, 64}, {64, 64}, {64, 128}};
    for (int i = 0; i < 4; ++i) {
        if (!pack_conv(std::string(enc[i]) + ".conv1.0", p1[i], ch[i][1], ch[i][0], 9, err)) return false;
        if (!pack_conv(std::string(enc[i]) + ".conv2.0", p2[i], ch[i][1], ch[i][1], 3, err)) return false;
    }
    for (int i = 0; i < 3; ++i)
        if (!pack_conv("mid_dilation." + std::to_string(2 * i), mid[i], 128, 128, 9, err)) return false;
    for (int m = 0; m < 2; ++m) {
        const std::string p = std::string("feat_prop_module.deform_align.") + kMods[m];
        if (!pack_conv(p, deform[m], 128, 256, 9, err)) return false;
        if (!pack_conv(p + ".conv_offset.0", off[m][0], 128, 384, 9, err)) return false;
        if (!pack_conv(p + ".conv_offset.2", off[m][1], 128, 128, 9, err)) return false;
        if (!pack_conv(p + ".conv_offset.4", off[m][2], 128, 128, 9, err)) return false;
        if (!pack_conv(p + ".conv_offset.6", off[m][3], 432, 128, 9, err)) return false;
        const std::string b = std::string("feat_prop_module.backbone.") + kMods[m];
        if (!pack_conv(b + ".0", bb1[m], 128, m == 0 ? 256 : 384, 9, err)) return false;
        if (!pack_conv(b + ".2", bb2[m], 128, 128, 9, err)) return false;
    }
    if (!pack_conv("feat_prop_module.fusion", fusion, 128, 256, 1, err)) return false;
    if (!pack_conv("decoder2.0", dec2a, 128, 128, 9, err)) return false;
    if (!pack_conv("decoder2.2.conv", dec2b, 64, 128, 9, err)) return false;
    if (!pack_conv("decoder1.0", dec1a, 64, 64, 9, err)) return false;
    if (!pack_conv("decoder1.2.conv", dec1b, 32, 64, 9, err)) return false;
    if (!pack_conv("upsample.0", up0, 32, 32, 9, err)) return false;
    if (!pack_conv("upsample.2.conv", up1, 2, 32, 9, err)) return false;
    ready_ = true;
    return true;
}

// ------------------------------------------------------------------------------------
// RfcPlan
// ------------------------------------------------------------------------------------
int RfcPlan::pickTile(int N) const { return N <= 32 ? VSR_TILE_256x32 : (N <= 64 ? n64Tile() : VSR_TILE_128x64); }

// Frames are stored step-major, sequence-minor (frame = i*S + s); buffers read by a temporal conv carry two zero
// steps before and after the sequence, so a +-2 step shift is a constant offset and never crosses into the other sequence.
std::vector<int> RfcPlan::seqIds(bool temporalHalo) const
{
    std::vector<int> v;
    for (int j = 0; j < S * T; ++j) v.push_back((temporalHalo ? 2 * S : 0) + j);
    return v;
}

void RfcPlan::gemm(const char* tag, int bufA, int64_t offA, int tRowA, int tColA, int K, int M, int bufC, int64_t offC, int tRowC,
                   const ConvW& w, int act, int bufR, int64_t offR, int tRowR, int tile, bool append)
{
    if (w.K != K) throw std::runtime_error(std::string("rfc gemm K mismatch: ") + tag);
    if (append) {                                  // one more problem of the previous op's launch (conv() cutting a large batch into frame groups)
        if (ops.empty() || ops.back().kind != OP_GEMM || ops.back().tileCfg != tile) throw std::runtime_error("rfc gemm: nothing to append to");
    } else {
        ops.emplace_back();
        ops.back().kind = OP_GEMM;
        ops.back().tag = tag;
        ops.back().bmode = VSR_BMODE_NK;
        ops.back().tileCfg = tile;
    }
    Op& op = ops.back();
    int BM, BN;
    tileDims(tile, BM, BN);
    GemmItem it{};
    it.M = M; it.N = w.cout; it.K = K;
    it.tilesM = cdiv(M, BM); it.tilesN = cdiv(it.N, BN);
    it.splitK = 1; it.chunksPerSplit = K / VSR_GG_KC; it.alpha = 1.f; it.act = act;
    it.bufA = bufA; it.offA = offA; it.tRowA = tRowA; it.tColA = tColA;
    it.bufB = FB_WEIGHTS; it.offB = w.w;
    it.tRowB = tRowsLinear(it.N, K, BN);
    it.tColB = tColsLinear(K / VSR_GG_KC, K / VSR_GG_KC);
    it.bufC = bufC; it.offC = offC; it.tRowC = tRowC;
    it.tColC = tColsLinear(cdiv(it.N, VSR_GG_KC), it.tilesN * BN / VSR_GG_KC);
    it.offBias = w.b;
    it.bufR = bufR; it.offR = offR; it.tRowR = tRowR;
    const double fl = 2.0 * M * (double)it.N * K;
    op.flops += fl;
    flops += fl;
    op.gemm.push_back(it);
}

void RfcPlan::conv(const char* tag, const Act& in, const std::vector<int>& inIds, const Act& out, const std::vector<int>& outIds, int kh,
                   int kw, int stride, int dil, const ConvW& w, int act, const Act* res, const std::vector<int>* resIds)
{
    const int tile = pickTile(w.cout);
    int BM, BN;
    tileDims(tile, BM, BN);
    const int nf = (int)outIds.size();
    need(out.buf, out.elems());
    // Offset tables hold 32-bit element offsets from the problem's base pointer.  A 70-frame batch of 1080p strips (what
    // batch_generator(1200, 70) hands the plugin: 138 flow fields of 1920 x 360) has operands beyond 2^31 elements -- the stem's im2col
    // (2.3e9) and the full-resolution 32-channel map in front of the last conv (3.1e9).  Such a conv becomes several problems of ONE
    // launch: groups of consecutive frames, each with its own 64-bit base (GemmItem::offA / offC / offR) and tables relative to the
    // group's first frame (equal groups share their tables).  Same rows, same K order, same arithmetic per output element.
    auto span = [](const Act& a, const std::vector<int>& v) { return ((int64_t)*std::max_element(v.begin(), v.end()) + 1) * a.frameElems(); };
    // (VSR_RFC_SPAN_LIMIT: test hook -- a small limit makes small plans take the grouped form, tests/test_rfc_replay.py)
    static const int64_t kMaxSpan = [] { const char* e = getenv("VSR_RFC_SPAN_LIMIT"); const long long x = e ? atoll(e) : 0;
                                         return x > 0 ? (int64_t)x : (int64_t)(2147483647LL - 8 * 1024 * 1024); }();   // room for the column offsets added to a row offset
    const bool fits = span(in, inIds) < kMaxSpan && span(out, outIds) < kMaxSpan && (!res || span(*res, *resIds) < kMaxSpan);
    int per = nf;
    if (!fits) {
        const int64_t fe = std::max(std::max(in.frameElems(), out.frameElems()), res ? res->frameElems() : (int64_t)0);
        per = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)1 << 30, kMaxSpan / 2) / fe);
        for (size_t j = 1; j < inIds.size(); ++j)
            if (inIds[j] < inIds[j - 1] || outIds[j] < outIds[j - 1] || (res && (*resIds)[j] < (*resIds)[j - 1]))
                throw std::runtime_error(std::string("rfc conv: frame ids of an operand beyond 2^31 elements must ascend: ") + tag);
    }
    for (int j0 = 0; j0 < nf; j0 += per) {
        const int j1 = std::min(nf, j0 + per);
        auto rel = [&](const std::vector<int>& v) { std::vector<int> r(v.begin() + j0, v.begin() + j1); for (int& x : r) x -= v[j0]; return r; };
        const std::vector<int> ri = rel(inIds), ro = rel(outIds);
        gemm(tag, in.buf, (int64_t)inIds[j0] * in.frameElems(), tRowsAct(in, ri, out.H, out.W, stride, BM, 0), tColsConvHW(in, kh, kw, dil),
             kh * kw * in.C, (j1 - j0) * out.H * out.W, out.buf, (int64_t)outIds[j0] * out.frameElems(), tRowsAct(out, ro, out.H, out.W, 1, BM, 0), w, act,
             res ? res->buf : -1, res ? (int64_t)(*resIds)[j0] * res->frameElems() : 0,
             res ? tRowsAct(*res, rel(*resIds), out.H, out.W, 1, BM, 0) : -1, tile, j0 > 0);
    }
}

// Conv3d kernel (3,1,1), dilation (2,1,1), padding (2,0,0) (P3DBlock.conv2, :160-163): K = 3*C gathered from steps i-2, i, i+2
void RfcPlan::tconv(const char* tag, const Act& in, const std::vector<int>& ids, const Act& out, const ConvW& w, int act)
{
    const int tile = pickTile(w.cout);
    int BM, BN;
    tileDims(tile, BM, BN);
    const std::string key = "TC:" + std::to_string(in.buf) + ":" + std::to_string(in.C) + ":" + std::to_string(in.H) + ":" + std::to_string(in.W);
    auto it = tableKey_.find(key);
    int tCol;
    if (it != tableKey_.end()) {
        tCol = it->second;
    } else {
        if (!Tuning::get().convChannelMajor || in.C % VSR_GG_KC) throw std::runtime_error("temporal conv needs the channel-major K order");
        std::vector<int32_t> v;
        for (int c = 0; c < in.C; c += VSR_GG_KC)
            for (int tap = 0; tap < 3; ++tap) v.push_back((int32_t)((int64_t)(tap - 1) * 2 * S * in.frameElems() + c));
        tCol = table(key, std::move(v));
    }
    const int M = S * T * out.H * out.W;
    need(out.buf, out.elems());
    gemm(tag, in.buf, 0, tRowsAct(in, ids, out.H, out.W, 1, BM, 0), tCol, 3 * in.C, M, out.buf, 0,
         tRowsAct(out, seqIds(false), out.H, out.W, 1, BM, 0), w, act, -1, 0, -1, tile);
}

void RfcPlan::upsample(const Act& in, const Act& out)
{
    Op op;
    op.kind = OP_UPSAMPLE2X; op.tag = "dec.up";
    op.bufSrc = in.buf; op.H = in.H; op.W = in.W; op.C = in.C; op.haloS = in.halo; op.bufDst = out.buf; op.haloD = out.halo; op.n = in.n;
    need(out.buf, out.elems());
    ops.push_back(std::move(op));
}

RfcPlan::RfcPlan(const RfcModel& model, int t_, int H_, int W_) : t(t_), T(t_ - 1), H(H_), W(W_), m_(model)
{
    if (!model.packed_ready()) throw std::runtime_error("flow-completion model is not packed");
    if (t < 2) throw std::runtime_error("flow completion needs at least two frames");
    if (H % 8 || W % 8 || H < 16 || W < 16) throw std::runtime_error("flow size must be a multiple of 8");
    bufElems.assign(FB_COUNT, 0);
    bufElems[FB_WEIGHTS] = (int64_t)model.packed.size();
    const int n = S * T, nh = S * (T + 4);
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, h8 = H / 8, w8 = W / 8;
    const std::vector<int> ids = seqIds(false), idsH = seqIds(true);
    need(FB_IN_FLOW_F, (int64_t)T * 2 * H * W);
    need(FB_IN_FLOW_B, (int64_t)T * 2 * H * W);
    need(FB_IN_MASK, (int64_t)t * H * W);                 // bytes
    {
        Op op;
        op.kind = OP_EW; op.ew = EW_RFC_IM2COL5; op.tag = "stem.im2col";
        op.ibuf[0] = FB_IN_FLOW_F; op.ibuf[1] = FB_IN_FLOW_B; op.ibuf[2] = FB_IN_MASK; op.ibuf[3] = FB_IM2COL;
        op.ipar[0] = t; op.ipar[1] = H; op.ipar[2] = W;
        need(FB_IM2COL, (int64_t)n * H2 * W2 * 96);
        ops.push_back(std::move(op));
    }
    // encoder (:208-227): stem, 4 P3D blocks; buffers that feed a temporal conv have the temporal halo (nh frames)
    const Act cols{FB_IM2COL, n, H2, W2, 96, 0}, x0{FB_X0, n, H2, W2, 32, 1};
    const Act a{FB_A, nh, H2, W2, 32, 0}, b{FB_B, n, H2, W2, 32, 1}, c{FB_C, nh, H4, W4, 64, 0}, e1{FB_E1, n, H4, W4, 64, 1};
    const Act d{FB_D, nh, H4, W4, 64, 0}, e{FB_E, n, H4, W4, 64, 1}, f{FB_F, nh, h8, w8, 128, 0}, e2{FB_E2, n, h8, w8, 128, 3};
    const Act m1{FB_M1, n, h8, w8, 128, 2}, m2{FB_M2, n, h8, w8, 128, 1};
    conv("enc.stem", cols, ids, x0, ids, 1, 1, 1, 1, m_.down, VSR_ACT_LRELU02, nullptr, nullptr);
    conv("enc.p3d.s", x0, ids, a, idsH, 3, 3, 1, 1, m_.p1[0], VSR_ACT_LRELU02, nullptr, nullptr);
    tconv("enc.p3d.t", a, idsH, b, m_.p2[0], VSR_ACT_LRELU02);
    conv("enc.p3d.s", b, ids, c, idsH, 3, 3, 2, 1, m_.p1[1], VSR_ACT_LRELU02, nullptr, nullptr);
    tconv("enc.p3d.t", c, idsH, e1, m_.p2[1], VSR_ACT_LRELU02);
    conv("enc.p3d.s", e1, ids, d, idsH, 3, 3, 1, 1, m_.p1[2], VSR_ACT_LRELU02, nullptr, nullptr);
    tconv("enc.p3d.t", d, idsH, e, m_.p2[2], VSR_ACT_LRELU02);
    conv("enc.p3d.s", e, ids, f, idsH, 3, 3, 2, 1, m_.p1[3], VSR_ACT_LRELU02, nullptr, nullptr);
    tconv("enc.p3d.t", f, idsH, e2, m_.p2[3], VSR_ACT_LRELU02);
    // mid_dilation (:229-236), the last conv writes the "spatial" slots of the propagation buffer
    // propagation buffer: slots of S frames each -- spatial[T] | backward feats[T] | forward feats[T] | aligned | zero
    const int slotBK = T, slotFW = 2 * T, slotALN = 3 * T, slotZERO = 3 * T + 1;
    const Act prop{FB_PROP, S * (3 * T + 2), h8, w8, 128, 1};
    need(FB_PROP, prop.elems());
    auto slotIds = [&](int slot) { return std::vector<int>{slot * S, slot * S + 1}; };
    const int64_t slotElems = (int64_t)S * prop.frameElems();
    conv("mid.d3", e2, ids, m1, ids, 3, 3, 1, 3, m_.mid[0], VSR_ACT_LRELU02, nullptr, nullptr);
    conv("mid.d2", m1, ids, m2, ids, 3, 3, 1, 2, m_.mid[1], VSR_ACT_LRELU02, nullptr, nullptr);
    conv("mid.d1", m2, ids, prop, ids, 3, 3, 1, 1, m_.mid[2], VSR_ACT_LRELU02, nullptr, nullptr);   // spatial slot i = frames i*S + s = ids

    // BidirectionalPropagation.forward (:69-126).  A conv over cat[...] of 128-channel slots: rows address slot 0, the
    // column table adds the absolute slot offsets -- the concatenation is never materialised.
    const Act t1{FB_T1, S, h8, w8, 128, 1}, t2{FB_T2, S, h8, w8, 128, 1}, t3{FB_T3, S, h8, w8, 128, 1};
    const Act offb{FB_OFF, S, h8, w8, 448, 0}, colsb{FB_COLS, S, h8, w8, 2304, 0}, bb{FB_BB, S, h8, w8, 128, 1};
    const std::vector<int> idS{0, 1};
    int BM, BN;
    tileDims(VSR_TILE_128x64, BM, BN);
    const int Mstep = S * h8 * w8;
    auto catCols = [&](const std::vector<int>& slots) {       // 3x3 window over cat[slots], channel-major K order
        std::string key = "CAT:";
        for (int s : slots) key += std::to_string(s) + ",";
        auto itk = tableKey_.find(key);
        if (itk != tableKey_.end()) return itk->second;
        std::vector<int32_t> v;
        for (int s : slots)
            for (int c0 = 0; c0 < 128; c0 += VSR_GG_KC)
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) {
                        const int64_t o = (int64_t)s * slotElems + ((int64_t)(ky - 1) * prop.Wp() + (kx - 1)) * prop.C + c0;
                        if (o > 2147483647LL) throw std::runtime_error("propagation buffer offsets exceed int32");
                        v.push_back((int32_t)o);
                    }
        return table(key, std::move(v));
    };
    if (!Tuning::get().convChannelMajor) throw std::runtime_error("the propagation convs need the channel-major K order");
    const int tRowSlot0 = tRowsAct(prop, idS, h8, w8, 1, BM, 0);
    for (int mod = 0; mod < 2; ++mod) {
        const int slotMod = mod == 0 ? slotBK : slotFW;
        for (int i = 0; i < T; ++i) {
            const int idx = mod == 0 ? T - 1 - i : i;                        // frame_idx reversed for "backward_"
            const int prev = mod == 0 ? idx + 1 : idx - 1, prev2 = mod == 0 ? idx + 2 : idx - 2;
            int slotProp = slotZERO;
            if (i > 0) {
                const int sP = slotMod + prev, sN2 = i > 1 ? slotMod + prev2 : slotZERO;
                // conv_offset over cat[cond_n1 = feat_prop, feat_current, cond_n2] (:97-103; :17-25)
                gemm("prop.off0", FB_PROP, 0, tRowSlot0, catCols({sP, idx, sN2}), 9 * 384, Mstep, FB_T1, 0, tRowsAct(t1, idS, h8, w8, 1, BM, 0),
                     m_.off[mod][0], VSR_ACT_LRELU01, -1, 0, -1, VSR_TILE_128x64);
                need(FB_T1, t1.elems());
                conv("prop.off1", t1, idS, t2, idS, 3, 3, 1, 1, m_.off[mod][1], VSR_ACT_LRELU01, nullptr, nullptr);
                conv("prop.off2", t2, idS, t3, idS, 3, 3, 1, 1, m_.off[mod][2], VSR_ACT_LRELU01, nullptr, nullptr);
                conv("prop.off3", t3, idS, offb, idS, 3, 3, 1, 1, m_.off[mod][3], VSR_ACT_NONE, nullptr, nullptr);
                {   // offsets 5*tanh, mask sigmoid, modulated bilinear columns of cat[feat_prop, feat_n2] (:31-46,104)
                    Op op;
                    op.kind = OP_EW; op.ew = EW_DEFORM_COLS; op.tag = "prop.deform.cols";
                    op.ibuf[0] = FB_PROP; op.ibuf[1] = FB_OFF; op.ibuf[2] = FB_COLS;
                    op.ioff[0] = (int64_t)sP * slotElems; op.ioff[1] = (int64_t)sN2 * slotElems;
                    op.ipar[0] = S; op.ipar[1] = h8; op.ipar[2] = w8; op.ipar[3] = prop.halo; op.ipar[4] = prop.C; op.ipar[5] = offb.C;
                    op.fpar[0] = 5.f;
                    need(FB_COLS, colsb.elems());
                    ops.push_back(std::move(op));
                }
                conv("prop.deform", colsb, idS, prop, slotIds(slotALN), 1, 1, 1, 1, m_.deform[mod], VSR_ACT_NONE, nullptr, nullptr);
                slotProp = slotALN;
            }
            // backbone over cat[feat_current, (backward feats of this frame), feat_prop], residual feat_prop (:107-115)
            std::vector<int> parts{idx};
            if (mod == 1) parts.push_back(slotBK + idx);
            parts.push_back(slotProp);
            gemm("prop.bb0", FB_PROP, 0, tRowSlot0, catCols(parts), 9 * 128 * (int)parts.size(), Mstep, FB_BB, 0,
                 tRowsAct(bb, idS, h8, w8, 1, BM, 0), m_.bb1[mod], VSR_ACT_LRELU01, -1, 0, -1, VSR_TILE_128x64);
            need(FB_BB, bb.elems());
            const std::vector<int> outIds = slotIds(slotMod + idx), resIds = slotIds(slotProp);
            conv("prop.bb1", bb, idS, prop, outIds, 3, 3, 1, 1, m_.bb2[mod], VSR_ACT_NONE, &prop, &resIds);
        }
    }
    // fusion 1x1 over cat[backward, forward] + x (:119-126), all frames at once
    const Act fused{FB_FUSED, n, h8, w8, 128, 1};
    {
        std::vector<int> bkIds, spIds = ids;
        for (int j = 0; j < n; ++j) bkIds.push_back(slotBK * S + j);
        std::vector<int32_t> v;
        for (int src = 0; src < 2; ++src)
            for (int c0 = 0; c0 < 128; c0 += VSR_GG_KC) v.push_back((int32_t)((int64_t)src * T * slotElems + c0));
        const int tCol = table("FUSE:" + std::to_string(T), std::move(v));
        need(FB_FUSED, fused.elems());
        gemm("prop.fusion", FB_PROP, 0, tRowsAct(prop, bkIds, h8, w8, 1, BM, 0), tCol, 256, n * h8 * w8, FB_FUSED, 0,
             tRowsAct(fused, ids, h8, w8, 1, BM, 0), m_.fusion, VSR_ACT_NONE, FB_PROP, 0, tRowsAct(prop, spIds, h8, w8, 1, BM, 0), VSR_TILE_128x64);
    }
    // decoders (:241-259,293-304)
    const Act d2a{FB_D2A, n, h8, w8, 128, 0}, up2{FB_UP2, n, H4, W4, 128, 1}, d2{FB_D2, n, H4, W4, 64, 1};
    const Act d1a{FB_D1A, n, H4, W4, 64, 0}, up1{FB_UP1, n, H2, W2, 64, 1}, d1{FB_D1, n, H2, W2, 32, 1};
    const Act u0{FB_U0, n, H2, W2, 32, 0}, up0{FB_UP0, n, H, W, 32, 1}, pred{FB_PRED, n, H, W, 32, 0};
    conv("dec.2a", fused, ids, d2a, ids, 3, 3, 1, 1, m_.dec2a, VSR_ACT_LRELU02, nullptr, nullptr);
    upsample(d2a, up2);
    conv("dec.2b", up2, ids, d2, ids, 3, 3, 1, 1, m_.dec2b, VSR_ACT_LRELU02, &e1, &ids);     // + feat_e1 after the LeakyReLU
    conv("dec.1a", d2, ids, d1a, ids, 3, 3, 1, 1, m_.dec1a, VSR_ACT_LRELU02, nullptr, nullptr);
    upsample(d1a, up1);
    conv("dec.1b", up1, ids, d1, ids, 3, 3, 1, 1, m_.dec1b, VSR_ACT_LRELU02, nullptr, nullptr);
    conv("dec.u0", d1, ids, u0, ids, 3, 3, 1, 1, m_.up0, VSR_ACT_LRELU02, nullptr, nullptr);
    upsample(u0, up0);
    conv("dec.u1", up0, ids, pred, ids, 3, 3, 1, 1, m_.up1, VSR_ACT_NONE, nullptr, nullptr);
    {
        Op op;
        op.kind = OP_EW; op.ew = EW_RFC_COMBINE; op.tag = "combine";
        op.ibuf[0] = FB_PRED; op.ibuf[1] = FB_OUT_F; op.ibuf[2] = FB_OUT_B;
        op.ipar[0] = t; op.ipar[1] = H; op.ipar[2] = W; op.ipar[3] = pred.C;
        need(FB_OUT_F, (int64_t)T * 2 * H * W);
        need(FB_OUT_B, (int64_t)T * 2 * H * W);
        ops.push_back(std::move(op));
    }
}

} // namespace vsr
