// Weight packer + plan builders for the ProPainter generator (see pp_plan.h).
#include "pp_plan.h"
#include "gather_gemm.h"
#include <math.h>
#include <stdexcept>

namespace vsr {

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t rup(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
static std::vector<int> iota(int n, int start = 0)
{
    std::vector<int> v(n);
    for (int i = 0; i < n; ++i) v[i] = start + i;
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
// img_propagation: BidirectionalPropagation(3, learnable=False) (propainter.py:104-193 with :157-165)
// ------------------------------------------------------------------------------------
PpImgPropPlan::PpImgPropPlan(int t_, int H_, int W_) : t(t_), H(H_), W(W_)
{
    if (t < 1 || H < 2 || W < 2) throw std::runtime_error("image propagation needs at least one frame");
    bufElems.assign(PB_COUNT, 0);
    const int64_t hw = (int64_t)H * W;
    need(PB_IN_FRAMES, t * 3 * hw);
    need(PB_IN_MASK_U8, t * hw);                          // bytes
    need(PB_IN_FLOW_F, (int64_t)(t > 1 ? t - 1 : 1) * 2 * hw);
    need(PB_IN_FLOW_B, (int64_t)(t > 1 ? t - 1 : 1) * 2 * hw);
    need(PB_MASK_F, t * hw);
    need(PB_BK, t * 3 * hw); need(PB_BKM, t * hw);
    need(PB_FW, t * 3 * hw); need(PB_FWM, t * hw);
    need(PB_OUT_MASK_U8, t * hw);
    {
        Op op;
        op.kind = OP_EW; op.ew = EW_PP_MASK_F32; op.tag = "imgprop.mask";
        op.ibuf[0] = PB_IN_MASK_U8; op.ibuf[1] = PB_MASK_F;
        op.ipar[0] = (int)(t * hw);
        ops.push_back(std::move(op));
    }
    // backward_1 over the input frames (reversed order, flows_forward propagate / flows_backward check), then forward_1
    // over backward_1's results (flow index i-1, roles swapped)  (:123-141)
    for (int mod = 0; mod < 2; ++mod) {
        for (int i = 0; i < t; ++i) {
            const int idx = mod == 0 ? t - 1 - i : i;
            const int prev = mod == 0 ? idx + 1 : idx - 1;
            const int flow = mod == 0 ? idx : idx - 1;
            Op op;
            op.kind = OP_EW; op.ew = EW_PP_IMGPROP; op.tag = mod == 0 ? "imgprop.backward" : "imgprop.forward";
            // ibuf: [0] current frames, [1] current masks, [2] propagated frames (out, and previous step), [3] propagated masks
            op.ibuf[0] = mod == 0 ? PB_IN_FRAMES : PB_BK; op.ibuf[1] = mod == 0 ? PB_MASK_F : PB_BKM;
            op.ibuf[2] = mod == 0 ? PB_BK : PB_FW; op.ibuf[3] = mod == 0 ? PB_BKM : PB_FWM;
            op.ipar[0] = 3; op.ipar[1] = H; op.ipar[2] = W; op.ipar[3] = i == 0 ? 1 : 0;
            op.ipar[4] = idx; op.ipar[5] = i == 0 ? idx : prev; op.ipar[6] = i == 0 ? 0 : flow;
            op.ipar[7] = mod;                              // 0: propagate with flows_f, check with flows_b ; 1: swapped
            ops.push_back(std::move(op));
        }
    }
}

// ------------------------------------------------------------------------------------
// PpModel: state_dict of ProPainter.pth (propainter.py:250-314)
// ------------------------------------------------------------------------------------
static const char* kMods[2] = {"backward_1", "forward_1"};

std::vector<std::string> PpModel::expected_keys()
{
    std::vector<std::string> k;
    auto add = [&](const std::string& n) { k.push_back(n + ".weight"); k.push_back(n + ".bias"); };
    for (int i = 0; i <= 16; i += 2) add("encoder.layers." + std::to_string(i));
    for (const char* e : {"decoder.0.conv", "decoder.2", "decoder.4.conv", "decoder.6", "ss.embedding", "sc.embedding", "sc.bias_conv"}) add(e);
    for (int m = 0; m < 2; ++m) {
        const std::string p = std::string("feat_prop_module.deform_align.") + kMods[m];
        add(p);
        for (const char* e : {".conv_offset.0", ".conv_offset.2", ".conv_offset.4", ".conv_offset.6"}) add(p + e);
    }
    for (int m = 0; m < 2; ++m) {
        add(std::string("feat_prop_module.backbone.") + kMods[m] + ".0");
        add(std::string("feat_prop_module.backbone.") + kMods[m] + ".2");
    }
    add("feat_prop_module.fuse.0");
    add("feat_prop_module.fuse.2");
    for (int i = 0; i < 8; ++i) {
        const std::string p = "transformers.transformer." + std::to_string(i) + ".";
        k.push_back(p + "attention.valid_ind_rolled");
        for (const char* e : {"attention.key", "attention.query", "attention.value", "attention.proj", "attention.pool_layer", "norm1", "norm2",
                              "mlp.fc1.0", "mlp.fc2.1"})
            add(p + e);
    }
    return k;
}

bool PpModel::set_param(const std::string& name, const float* data, const int64_t* shape, int ndim, std::string& err)
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

bool PpModel::pack_conv(const std::string& key, ConvW& cw, int cout, int cin, int taps, int r0, int nrows, const std::vector<int>& ciPos,
                        int cinPad, const std::vector<int>& outPos, int nPad, std::string& err)
{
    auto wi = raw_.find(key + ".weight"), bi = raw_.find(key + ".bias");
    if (wi == raw_.end() || bi == raw_.end()) { err = "missing key in state_dict: " + key; return false; }
    const Raw& w = wi->second;
    // conv weights [cout][cin][k...] and Linear weights [cout][cin*taps] (soft split: unfold's (c, ky, kx) order) alike
    if (w.shape.size() < 2 || w.shape[0] != cout || (int64_t)w.v.size() != (int64_t)cout * cin * taps || (int64_t)bi->second.v.size() != cout) {
        err = "shape mismatch for " + key;
        return false;
    }
    if (cinPad % VSR_GG_KC || !Tuning::get().convChannelMajor) { err = "padded-channel packing needs the channel-major K order"; return false; }
    const int K = taps * cinPad;
    cw.cout = nPad;
    cw.K = K;
    cw.w = (int64_t)packed.size();
    packed.resize(packed.size() + (size_t)rup((int64_t)nPad * K, 32), 0.f);
    float* dst = packed.data() + cw.w;
    for (int n = 0; n < nrows; ++n)
        for (int ci = 0; ci < cin; ++ci)
            for (int tap = 0; tap < taps; ++tap) {
                const int pos = ciPos[ci];
                const int k = ((pos / VSR_GG_KC) * taps + tap) * VSR_GG_KC + (pos % VSR_GG_KC);     // mirrors PpGenPlan::tColsChunks
                dst[(int64_t)outPos[n] * K + k] = w.v[((int64_t)(r0 + n) * cin + ci) * taps + tap];
            }
    cw.b = (int64_t)packed.size();
    packed.resize(packed.size() + (size_t)rup(nPad, 32), 0.f);
    for (int n = 0; n < nrows; ++n) packed[cw.b + outPos[n]] = bi->second.v[r0 + n];
    return true;
}

// all rows, channels in place; cin % 32 != 0 -> tap-major K (k = tap*cin + ci) padded up to a multiple of 32
// F.fold / F.unfold order the 7x7 patch vectors channel-major (element c*49 + tap).  The fold / unfold kernels want the channel
// fastest: lanes that are neighbours in the NHWC map then read / write neighbouring floats of a token's row (a 49-float stride between
// lanes made k_pp_fold run at 0.4 TB/s, profiles/r05_propainter_f32_kernel_stats.csv).  The token rows are GEMM outputs / inputs, so
// their column order is free: it is the row order of the producing weight (fc1, soft composition embedding) and the K order of the
// consuming one (fc2).  patch_perm(C)[c*49 + tap] = tap*C + c.
static std::vector<int> patch_perm(int C)
{
    std::vector<int> v((size_t)C * 49);
    for (int c = 0; c < C; ++c)
        for (int tap = 0; tap < 49; ++tap) v[(size_t)c * 49 + tap] = tap * C + c;
    return v;
}

bool PpModel::pack_plain(const std::string& key, ConvW& cw, int cout, int cin, int taps, std::string& err, const std::vector<int>* outPos,
                         const std::vector<int>* ciPos)
{
    if (cin % VSR_GG_KC == 0 && !ciPos) return pack_conv(key, cw, cout, cin, taps, 0, cout, iota(cin), cin, outPos ? *outPos : iota(cout), cout, err);
    if (outPos || (ciPos && taps != 1)) { err = "pack_plain: unsupported permutation for " + key; return false; }
    auto wi = raw_.find(key + ".weight"), bi = raw_.find(key + ".bias");
    if (wi == raw_.end() || bi == raw_.end()) { err = "missing key in state_dict: " + key; return false; }
    const Raw& w = wi->second;
    if (w.shape.size() < 2 || w.shape[0] != cout || (int64_t)w.v.size() != (int64_t)cout * cin * taps || (int64_t)bi->second.v.size() != cout) {
        err = "shape mismatch for " + key;
        return false;
    }
    const int K = (int)rup((int64_t)taps * cin, VSR_GG_KC);
    cw.cout = cout;
    cw.K = K;
    cw.w = (int64_t)packed.size();
    packed.resize(packed.size() + (size_t)rup((int64_t)cout * K, 32), 0.f);
    for (int n = 0; n < cout; ++n)
        for (int ci = 0; ci < cin; ++ci)
            for (int tap = 0; tap < taps; ++tap)
                packed[cw.w + (int64_t)n * K + tap * cin + (ciPos ? (*ciPos)[ci] : ci)] = w.v[((int64_t)n * cin + ci) * taps + tap];
    cw.b = (int64_t)packed.size();
    packed.resize(packed.size() + (size_t)rup(cout, 32), 0.f);
    for (int n = 0; n < cout; ++n) packed[cw.b + n] = bi->second.v[n];
    return true;
}

int64_t PpModel::push_vec(const std::string& key, int n, std::string& err)
{
    auto it = raw_.find(key);
    if (it == raw_.end() || (int64_t)it->second.v.size() != n) { err = "missing / bad entry: " + key; return -1; }
    const int64_t off = (int64_t)packed.size();
    packed.resize(packed.size() + (size_t)rup(n, 32), 0.f);
    for (int i = 0; i < n; ++i) packed[off + i] = it->second.v[i];
    return off;
}

bool PpModel::pack(std::string& err)
{
    packed.clear();
    ready_ = false;
    for (const auto& k : expected_keys())
        if (!raw_.count(k)) { err = "missing key in state_dict: " + k; return false; }
    const std::string E = "encoder.layers.";
    if (!pack_plain(E + "0", enc0, 64, 5, 9, err) || !pack_plain(E + "2", enc2, 64, 64, 9, err) || !pack_plain(E + "4", enc4, 128, 64, 9, err) ||
        !pack_plain(E + "6", enc6, 256, 128, 9, err) || !pack_plain(E + "8", enc8, 384, 256, 9, err))
        return false;
    // grouped convs (:206-211,217-223): one GEMM per group.  Layer 12 writes each group's 96 outputs as two blocks of
    // 48 padded to 64 (zero rows), so that layer 14's eight 48-channel input groups start on 32-channel chunks.
    for (int j = 0; j < 2; ++j)
        if (!pack_conv(E + "10", enc10[j], 512, 320, 9, 256 * j, 256, iota(320), 320, iota(256), 256, err)) return false;
    {
        std::vector<int> outPos(96);
        for (int n = 0; n < 96; ++n) outPos[n] = n < 48 ? n : n + 16;
        for (int j = 0; j < 4; ++j)
            if (!pack_conv(E + "12", enc12[j], 384, 192, 9, 96 * j, 96, iota(192), 192, outPos, 128, err)) return false;
    }
    {
        std::vector<int> ciPos(80);           // 32 channels of x0, then 48 of layer 12 (padded to 64)
        for (int c = 0; c < 80; ++c) ciPos[c] = c;
        for (int j = 0; j < 8; ++j)
            if (!pack_conv(E + "14", enc14[j], 256, 80, 9, 32 * j, 32, ciPos, 96, iota(32), 32, err)) return false;
    }
    if (!pack_plain(E + "16", enc16, 128, 512, 9, err)) return false;
    for (int m = 0; m < 2; ++m) {
        const std::string p = std::string("feat_prop_module.deform_align.") + kMods[m];
        if (!pack_plain(p, deform[m], 128, 128, 9, err)) return false;
        if (!pack_conv(p + ".conv_offset.0", off[m][0], 128, 261, 9, 0, 128, iota(261), 288, iota(128), 128, err)) return false;
        if (!pack_plain(p + ".conv_offset.2", off[m][1], 128, 128, 9, err) || !pack_plain(p + ".conv_offset.4", off[m][2], 128, 128, 9, err) ||
            !pack_plain(p + ".conv_offset.6", off[m][3], 432, 128, 9, err))
            return false;
        const std::string b = std::string("feat_prop_module.backbone.") + kMods[m];
        if (!pack_conv(b + ".0", bb1[m], 128, 258, 9, 0, 128, iota(258), 288, iota(128), 128, err)) return false;
        if (!pack_plain(b + ".2", bb2[m], 128, 128, 9, err)) return false;
    }
    if (!pack_conv("feat_prop_module.fuse.0", fuse1, 128, 258, 9, 0, 128, iota(258), 288, iota(128), 128, err)) return false;
    if (!pack_plain("feat_prop_module.fuse.2", fuse2, 128, 128, 9, err)) return false;
    if (!pack_plain("ss.embedding", ss, 512, 128, 49, err)) return false;   // Linear(128*49, 512) on unfold's (c, ky, kx) order = conv [512][128][7][7]
    const std::vector<int> perm128 = patch_perm(128), perm40 = patch_perm(40);
    if (!pack_plain("sc.embedding", sc, 6272, 512, 1, err, &perm128) || !pack_plain("sc.bias_conv", scConv, 128, 128, 9, err)) return false;
    if (!pack_plain("decoder.0.conv", dec0, 128, 128, 9, err) || !pack_plain("decoder.2", dec2, 64, 128, 9, err) ||
        !pack_plain("decoder.4.conv", dec4, 64, 64, 9, err) || !pack_plain("decoder.6", dec6, 3, 64, 9, err))
        return false;
    for (int i = 0; i < 8; ++i) {
        const std::string p = "transformers.transformer." + std::to_string(i) + ".";
        ConvW q, k, v;
        if (!pack_plain(p + "attention.query", q, 512, 512, 1, err) || !pack_plain(p + "attention.key", k, 512, 512, 1, err) ||
            !pack_plain(p + "attention.value", v, 512, 512, 1, err))
            return false;
        ConvW& f = blk[i].qkv;                                      // rows [0,512) query, [512,1024) key, [1024,1536) value
        f.cout = 1536; f.K = 512;
        f.w = (int64_t)packed.size();
        packed.resize(packed.size() + (size_t)1536 * 512, 0.f);
        f.b = (int64_t)packed.size();
        packed.resize(packed.size() + 1536, 0.f);
        const ConvW* src[3] = {&q, &k, &v};
        for (int j = 0; j < 3; ++j) {
            for (int64_t e = 0; e < 512 * 512; ++e) packed[f.w + (int64_t)j * 512 * 512 + e] = packed[src[j]->w + e];
            for (int e = 0; e < 512; ++e) packed[f.b + j * 512 + e] = packed[src[j]->b + e];
        }
        if (!pack_plain(p + "attention.proj", blk[i].proj, 512, 512, 1, err) || !pack_plain(p + "mlp.fc1.0", blk[i].fc1, 1960, 512, 1, err, &perm40) ||
            !pack_plain(p + "mlp.fc2.1", blk[i].fc2, 512, 1960, 1, err, nullptr, &perm40))
            return false;
        blk[i].ln1g = push_vec(p + "norm1.weight", 512, err); blk[i].ln1b = push_vec(p + "norm1.bias", 512, err);
        blk[i].ln2g = push_vec(p + "norm2.weight", 512, err); blk[i].ln2b = push_vec(p + "norm2.bias", 512, err);
        blk[i].poolW = push_vec(p + "attention.pool_layer.weight", 512 * 16, err);
        blk[i].poolB = push_vec(p + "attention.pool_layer.bias", 512, err);
        if (blk[i].ln1g < 0 || blk[i].ln1b < 0 || blk[i].ln2g < 0 || blk[i].ln2b < 0 || blk[i].poolW < 0 || blk[i].poolB < 0) return false;
    }
    ready_ = true;
    return true;
}

// ------------------------------------------------------------------------------------
// PpGenPlan
// ------------------------------------------------------------------------------------
void PpGenPlan::token_grid(int H, int W, int& fh, int& fw, int& gh, int& gw)
{
    const int h = H / 4, w = W / 4;
    fh = (h + 6 - 7) / 3 + 1; fw = (w + 6 - 7) / 3 + 1;          // SoftSplit: kernel 7, stride 3, padding 3 (:280-287)
    gh = cdiv(fh, 5) * 5; gw = cdiv(fw, 9) * 9;                   // window (5, 9)
}

// Problems of N >= wideN output columns run on 128 x 128 tiles (BM is 128 either way: same row tables, same K order, same bits).
// wideN is the model's hint (PpModel::wideN, set with the engine's arithmetic: vsr_pp_set_precision): 512 for exact fp32 -- the token GEMMs,
// soft split / composition; the two shapes are within 1 % there -- and 128 for the fp16-operand arithmetic, where the operands are rounded in
// the staging registers (gather_gemm_f32_v4<HI_ONLY>) and a square tile halves the B fragments staged per MFMA: tr.fc2 66.5 -> 43.2 ms per
// 68-frame batch, the whole generator 0.83 -> 0.75 s, while exact fp32 LOSES 48 ms with N >= 128 (profiles/r06_pp_square_tiles_ab.log).
// VSR_PP_WIDE_N overrides the hint, VSR_PP_TR_TILE=128x64 switches the square tiles off (A/B runs).
int PpGenPlan::wideTile(int N) const
{
    static const bool narrowEnv = [] { const char* e = getenv("VSR_PP_TR_TILE"); return e && std::string(e) == "128x64"; }();
    static const int wideEnv = [] { const char* e = getenv("VSR_PP_WIDE_N"); const int x = e ? atoi(e) : 0; return x >= 128 ? x : 0; }();
    const int wideN = wideEnv ? wideEnv : m_.wideN;
    return (N >= wideN && !narrowEnv) ? VSR_TILE_128x128 : VSR_TILE_128x64;
}
int PpGenPlan::pickTile(int N) const { return N <= 32 ? VSR_TILE_256x32 : (N <= 64 ? n64Tile() : wideTile(N)); }

Op& PpGenPlan::ew(int kind, const char* tag)
{
    Op op;
    op.kind = OP_EW; op.ew = kind; op.tag = tag;
    ops.push_back(std::move(op));
    return ops.back();
}

// 32-channel chunk offsets of a kh x kw window over the channel chunks that start at chunkCh[] (channel-major K order)
int PpGenPlan::tColsChunks(const Act& a, int kh, int kw, int dil, const std::vector<int>& chunkCh)
{
    std::string key = "CH:" + std::to_string(a.halo) + ":" + std::to_string(a.W) + ":" + std::to_string(a.C) + ":" + std::to_string(kh) + "x" +
                      std::to_string(kw) + ":" + std::to_string(dil) + ":";
    for (int c : chunkCh) key += std::to_string(c) + ",";
    auto it = tableKey_.find(key);
    if (it != tableKey_.end()) return it->second;
    if (a.halo < dil * (kh / 2) || a.halo < dil * (kw / 2)) throw std::runtime_error("activation halo too small for conv");
    std::vector<int32_t> v;
    for (int c : chunkCh)
        for (int ky = 0; ky < kh; ++ky)
            for (int kx = 0; kx < kw; ++kx) v.push_back((int32_t)(((int64_t)(ky - kh / 2) * dil * a.Wp() + (kx - kw / 2) * dil) * a.C + c));
    return table(key, std::move(v));
}

void PpGenPlan::gemm(const char* tag, int bufA, int64_t offA, int tRowA, int tColA, int K, int M, int bufC, int64_t offC, int tRowC, int tColC,
                     const ConvW& w, int act, int bufR, int64_t offR, int tRowR, int tile, Op* appendTo)
{
    if (w.K != K) throw std::runtime_error(std::string("propainter gemm K mismatch: ") + tag);
    // (wide problems on square tiles: wideTile())
    if (!appendTo && tile == VSR_TILE_128x64) tile = wideTile(w.cout);
    int BM, BN;
    tileDims(tile, BM, BN);
    GemmItem it{};
    it.M = M; it.N = w.cout; it.K = K;
    it.tilesM = cdiv(M, BM); it.tilesN = cdiv(it.N, BN);
    it.splitK = 1; it.chunksPerSplit = K / VSR_GG_KC; it.alpha = 1.f; it.act = act;
    it.bufA = bufA; it.offA = offA; it.tRowA = tRowA; it.tColA = tColA;
    it.bufB = PB_WEIGHTS; it.offB = w.w;
    it.tRowB = tRowsLinear(it.N, K, BN);
    it.tColB = tColsLinear(K / VSR_GG_KC, K / VSR_GG_KC);
    it.bufC = bufC; it.offC = offC; it.tRowC = tRowC;
    it.tColC = tColC >= 0 ? tColC : tColsLinear(cdiv(it.N, VSR_GG_KC), it.tilesN * BN / VSR_GG_KC);
    it.offBias = w.b;
    it.bufR = bufR; it.offR = offR; it.tRowR = tRowR;
    const double fl = 2.0 * M * (double)it.N * K;
    flops += fl;
    if (appendTo) {
        if (appendTo->tileCfg != tile) throw std::runtime_error("grouped problems must share the tile shape");
        appendTo->gemm.push_back(it);
        appendTo->flops += fl;
        return;
    }
    Op op;
    op.kind = OP_GEMM; op.tag = tag; op.bmode = VSR_BMODE_NK; op.tileCfg = tile; op.flops = fl;
    op.gemm.push_back(it);
    ops.push_back(std::move(op));
}

void PpGenPlan::conv(const char* tag, const Act& in, const std::vector<int>& inIds, const std::vector<int>& chunkCh, int kh, int kw, int stride,
                     int dil, const Act& out, const std::vector<int>& outIds, int c0out, const ConvW& w, int act, const Act* res,
                     const std::vector<int>* resIds, Op* appendTo)
{
    const int tile = appendTo ? appendTo->tileCfg : pickTile(w.cout);
    int BM, BN;
    tileDims(tile, BM, BN);
    const int M = (int)outIds.size() * out.H * out.W;
    need(out.buf, out.elems());
    gemm(tag, in.buf, 0, tRowsAct(in, inIds, out.H, out.W, stride, BM, 0), tColsChunks(in, kh, kw, dil, chunkCh), kh * kw * 32 * (int)chunkCh.size(),
         M, out.buf, 0, tRowsAct(out, outIds, out.H, out.W, 1, BM, c0out), -1, w, act, res ? res->buf : -1, 0,
         res ? tRowsAct(*res, *resIds, out.H, out.W, 1, BM, 0) : -1, tile, appendTo);
}

void PpGenPlan::convRect(const char* tag, const Act& in, const std::vector<int>& ids, const std::vector<int>& chunkCh, const Act& out,
                         const ConvW& w, int act, const Act* res, int ylo, int yhi, int xlo, int xhi)
{
    if (in.H != out.H || in.W != out.W || ylo < 0 || yhi > out.H || xlo < 0 || xhi > out.W || ylo >= yhi || xlo >= xhi)
        throw std::runtime_error(std::string("propainter conv rectangle: ") + tag);
    const int tile = pickTile(w.cout);
    int BM, BN;
    tileDims(tile, BM, BN);
    const int M = (int)ids.size() * (yhi - ylo) * (xhi - xlo);
    trimmedFlops_ += 2.0 * ((double)ids.size() * out.H * out.W - M) * w.cout * (9.0 * 32 * chunkCh.size());
    need(out.buf, out.elems());
    gemm(tag, in.buf, 0, tRowsActRect(in, ids, ylo, yhi, xlo, xhi, BM), tColsChunks(in, 3, 3, 1, chunkCh), 9 * 32 * (int)chunkCh.size(), M, out.buf,
         0, tRowsActRect(out, ids, ylo, yhi, xlo, xhi, BM), -1, w, act, res ? res->buf : -1, 0,
         res ? tRowsActRect(*res, ids, ylo, yhi, xlo, xhi, BM) : -1, tile, nullptr);
}

void PpGenPlan::upsample(const Act& in, const Act& out)
{
    Op op;
    op.kind = OP_UPSAMPLE2X; op.tag = "dec.up";
    op.bufSrc = in.buf; op.H = in.H; op.W = in.W; op.C = in.C; op.haloS = in.halo; op.bufDst = out.buf; op.haloD = out.halo; op.n = in.n;
    need(out.buf, out.elems());
    ops.push_back(std::move(op));
}

static std::vector<int> chunks(int c0, int n)            // n consecutive 32-channel chunks starting at channel c0
{
    std::vector<int> v;
    for (int i = 0; i < n; ++i) v.push_back(c0 + 32 * i);
    return v;
}
static std::vector<int> cat(std::vector<int> a, const std::vector<int>& b)
{
    a.insert(a.end(), b.begin(), b.end());
    return a;
}

// SparseWindowAttention.forward (sparse_transformer.py:163-270) of transformer block `blk` as three grouped ops: QK^T,
// row softmax, P.V -- one problem per (window, head).  Keys of a masked window: its own tokens, the rolled-window tokens
// outside it (valid_ind_rolled) and all pooled tokens, on the frames of this block's temporal stride; an unmasked window
// attends to its own tokens on all frames.  Key order is irrelevant to softmax(QK^T)V, only the set matters.
void PpGenPlan::attention(int blk, const std::vector<uint8_t>& windowMasked, int tq, int ty0, int ty1, int tx0, int tx1)
{
    const bool allQ = tq == t && ty0 <= 0 && ty1 >= fh && tx0 <= 0 && tx1 >= fw;
    const std::string qSfx = allQ ? std::string() : ":q" + std::to_string(tq);      // (a window is taken whole or not at all)
    double fullFlops = 0;
    const int nwh = gh / 5, nww = gw / 9, C = 512, CH = 128, Q3 = 3 * C;
    const int parity = blk % 2;                                   // T_ind = arange(parity, t, 2)  (:330-334, t_dilation = 2)
    Op qk, sm, pv;
    qk.kind = OP_GEMM; qk.tag = "attn.qk"; qk.tileCfg = VSR_TILE_128x64; qk.bmode = VSR_BMODE_NK;
    sm.kind = OP_SOFTMAX; sm.tag = "attn.softmax";
    pv.kind = OP_GEMM; pv.tag = "attn.pv"; pv.tileCfg = VSR_TILE_128x64; pv.bmode = VSR_BMODE_KN;
    const int BM = 128, BN = 64;
    const int eh = 3, ew_ = 5;                                    // expand_size = ((5+1)/2, (9+1)/2)
    const int64_t pooledBase = (int64_t)t * gh * gw;
    int64_t sOff = 0;
    auto tokRow = [&](int f, int y, int x) { return ((int64_t)f * gh + y) * gw + x; };
    // one problem: queries qTok (token rows) x keys kTok, for every head
    auto addProblem = [&](const std::string& qkey, const std::vector<int64_t>& qTok, const std::string& kkey, const std::vector<int64_t>& kTok) {
        const int M = (int)qTok.size(), nk = (int)kTok.size();
        const int ldS = (int)rup(nk, VSR_GG_KC);
        std::vector<int32_t> qrow, arow, krowN, krowK;
        for (int64_t r : qTok) { qrow.push_back((int32_t)(r * Q3)); arow.push_back((int32_t)(r * C)); }
        while (qrow.size() % BM) { qrow.push_back(qrow[0]); arow.push_back(arow[0]); }
        for (int64_t r : kTok) krowN.push_back((int32_t)(r * Q3));
        krowK = krowN;
        while (krowN.size() % BN) krowN.push_back(krowN[0]);
        while ((int)krowK.size() < ldS) krowK.push_back(krowK[0]);          // P's padded columns are zero
        const int tQ = table("QROW:" + qkey, std::move(qrow)), tA = table("AROW:" + qkey, std::move(arow));
        const int tKn = table("KROWN:" + kkey, std::move(krowN)), tKk = table("KROWK:" + kkey, std::move(krowK));
        for (int head = 0; head < 4; ++head) {
            GemmItem a{};
            a.M = M; a.N = nk; a.K = CH;
            a.tilesM = cdiv(M, BM); a.tilesN = cdiv(nk, BN);
            a.splitK = 1; a.chunksPerSplit = CH / VSR_GG_KC; a.alpha = 1.f; a.act = VSR_ACT_NONE;
            a.bufA = PG_QKV; a.offA = head * CH; a.tRowA = tQ; a.tColA = tColsLinear(4, 4);
            a.bufB = PG_QKV; a.offB = C + head * CH; a.tRowB = tKn; a.tColB = a.tColA;
            a.bufC = PG_S; a.offC = sOff; a.tRowC = tRowsLinear(M, ldS, BM);
            a.tColC = tColsLinear(a.tilesN * BN / VSR_GG_KC, a.tilesN * BN / VSR_GG_KC);
            a.bufR = -1; a.tRowR = -1; a.offBias = -1;
            qk.gemm.push_back(a);
            SoftmaxItem sI{};
            sI.bufS = PG_S; sI.offS = sOff; sI.splitStride = 0; sI.nsplit = 1;
            sI.bufP = PG_P; sI.offP = sOff;
            sI.M = M; sI.N = nk; sI.ldS = ldS; sI.ldP = ldS;
            sI.scale = (float)(1.0 / sqrt((double)CH));
            sm.softmax.push_back(sI);
            GemmItem b{};
            b.M = M; b.N = CH; b.K = ldS;
            b.tilesM = cdiv(M, BM); b.tilesN = cdiv(CH, BN);
            b.splitK = 1; b.chunksPerSplit = ldS / VSR_GG_KC; b.alpha = 1.f; b.act = VSR_ACT_NONE;
            b.bufA = PG_P; b.offA = sOff; b.tRowA = tRowsLinear(M, ldS, BM); b.tColA = tColsLinear(ldS / VSR_GG_KC, ldS / VSR_GG_KC);
            b.bufB = PG_QKV; b.offB = 2 * C + head * CH; b.tRowB = tKk; b.tColB = tColsLinear(4, b.tilesN * BN / VSR_GG_KC);
            b.bufC = PG_ATT; b.offC = head * CH; b.tRowC = tA; b.tColC = tColsLinear(4, b.tilesN * BN / VSR_GG_KC);
            b.bufR = -1; b.tRowR = -1; b.offBias = -1;
            pv.gemm.push_back(b);
            const double fl = 2.0 * M * (double)nk * CH;
            qk.flops += fl; pv.flops += fl;
            fullFlops -= 2 * fl;
            sOff += rup((int64_t)M * ldS, 32);
        }
    };
    for (int win = 0; win < nwh * nww; ++win) {
        const int wy = win / nww, wx = win % nww;
        const std::string wkey = std::to_string(win);
        const bool unread = !allQ && (wy * 5 >= ty1 || wy * 5 + 5 <= ty0 || wx * 9 >= tx1 || wx * 9 + 9 <= tx0);      // none of its tokens is read
        if (!windowMasked[win]) {
            fullFlops += 4 * 2 * 2.0 * (45.0 * t) * 45 * CH;           // what the reference spends on it: 4 heads, QK^T and P.V
            if (unread) continue;
            // unmasked window (:253-262): every frame's 45 queries attend to the same frame's 45 window tokens (t is a batch dimension)
            for (int f = 0; f < tq; ++f) {
                std::vector<int64_t> tok;
                for (int i = 0; i < 5; ++i)
                    for (int j = 0; j < 9; ++j) tok.push_back(tokRow(f, wy * 5 + i, wx * 9 + j));
                const std::string key = wkey + ":f" + std::to_string(f);
                addProblem(key, tok, key, tok);
            }
            continue;
        }
        // masked window (:238-251): all frames' queries; keys on the frames of this block's temporal stride: the window tokens,
        // the rolled-window tokens outside the window, all pooled tokens
        std::vector<int64_t> qTok, kTok;
        for (int f = 0; f < tq; ++f)
            for (int i = 0; i < 5; ++i)
                for (int j = 0; j < 9; ++j) qTok.push_back(tokRow(f, wy * 5 + i, wx * 9 + j));
        for (int f = parity; f < t; f += 2) {
            for (int i = 0; i < 5; ++i)
                for (int j = 0; j < 9; ++j) kTok.push_back(tokRow(f, wy * 5 + i, wx * 9 + j));
            // rolled windows (:189-209): torch.roll(k, shifts=(sh, sw)) puts k[(y - sh) mod gh][(x - sw) mod gw] at (y, x);
            // of each rolled window only the tokens outside the current one are kept (mask_tl/tr/bl/br, :143-154)
            const int sh[4] = {-eh, -eh, eh, eh}, sw[4] = {-ew_, ew_, -ew_, ew_};
            for (int s = 0; s < 4; ++s)
                for (int i = 0; i < 5; ++i)
                    for (int j = 0; j < 9; ++j) {
                        const bool rowKeep = (s < 2) ? (i >= 5 - eh) : (i < eh);
                        const bool colKeep = (s % 2 == 0) ? (j >= 9 - ew_) : (j < ew_);
                        if (!(rowKeep || colKeep)) continue;
                        const int y = ((wy * 5 + i - sh[s]) % gh + gh) % gh, x = ((wx * 9 + j - sw[s]) % gw + gw) % gw;
                        kTok.push_back(tokRow(f, y, x));
                    }
            for (int p = 0; p < ph * pw; ++p) kTok.push_back(pooledBase + (int64_t)f * ph * pw + p);
        }
        fullFlops += 4 * 2 * 2.0 * (45.0 * t) * (double)kTok.size() * CH;
        if (unread) continue;
        addProblem(wkey + qSfx, qTok, wkey + ":m" + std::to_string(parity), kTok);
    }
    need(PG_S, sOff);
    need(PG_P, sOff);
    flops += qk.flops + pv.flops;
    if (!allQ && fullFlops > 0) trimmedFlops_ += fullFlops;
    ops.push_back(std::move(qk));
    ops.push_back(std::move(sm));
    ops.push_back(std::move(pv));
}

PpGenPlan::PpGenPlan(const PpModel& model, int t_, int lt_, int H_, int W_, const std::vector<uint8_t>& windowMasked, int decLo_, int decHi_,
                     int decXLo_, int decXHi_, int mode_)
    : t(t_), lt(lt_), H(H_), W(W_), mode(mode_), h(H_ / 4), w(W_ / 4), m_(model)
{
    if (mode < PP_PLAN_FULL || mode > PP_PLAN_CACHED) throw std::runtime_error("bad plan mode");
    const bool encodeOnly = mode == PP_PLAN_ENCODE, cached = mode == PP_PLAN_CACHED;
    const int encTok = encodeOnly ? lt_ : 0;       // PP_PLAN_ENCODE: soft-split tokens for the first lt_ frames only (the ones that can be reference frames)
    if (encodeOnly) {
        if (encTok < 0 || encTok > t) throw std::runtime_error("bad number of token frames");
        lt = t;
    }
    decLo = 0; decHi = H; decXLo = 0; decXHi = W;
    if (decHi_ > decLo_) {
        if (decLo_ < 0 || decHi_ > H) throw std::runtime_error("bad output row range");
        decLo = decLo_; decHi = decHi_;
    }
    if (decXHi_ > decXLo_) {
        if (decXLo_ < 0 || decXHi_ > W) throw std::runtime_error("bad output column range");
        decXLo = decXLo_; decXHi = decXHi_;
    }
    if (!model.packed_ready()) throw std::runtime_error("ProPainter model is not packed");
    if (lt < 1 || lt > t) throw std::runtime_error("bad number of local frames");
    if (H % 4 || W % 4 || h < 7 || w < 7) throw std::runtime_error("frame size must be a multiple of 4 and at least 28");
    if (!Tuning::get().convChannelMajor) throw std::runtime_error("the ProPainter plan needs the channel-major K order");
    token_grid(H, W, fh, fw, gh, gw);
    ph = (gh - 4) / 4 + 1; pw = (gw - 4) / 4 + 1;
    if (!encodeOnly && (int)windowMasked.size() != (gh / 5) * (gw / 9)) throw std::runtime_error("window mask flags do not match the token grid");
    bufElems.assign(PB_COUNT, 0);
    bufElems[PB_WEIGHTS] = (int64_t)model.packed.size();
    const int H2 = H / 2, W2 = W / 2;
    const int64_t HW = (int64_t)H * W;
    const std::vector<int> idsT = iota(t), idsL = iota(lt), idsR = iota(t - lt, lt);
    if (!cached) need(PB_IN_FRAMES, t * 3 * HW);
    need(PB_IN_MASK_U8, t * HW);
    need(PB_IN_MASK_UPD_U8, t * HW);
    if (!encodeOnly) {
        need(PB_IN_FLOW_F, (int64_t)(lt > 1 ? lt - 1 : 1) * 2 * HW);
        need(PB_IN_FLOW_B, (int64_t)(lt > 1 ? lt - 1 : 1) * 2 * HW);
    }
    const int L = VSR_ACT_LRELU02;
    // propagation buffer: slots of one frame [h+2][w+2][128]: input[lt] | backward[lt] | forward[lt] | masks[lt] | warped | misc | aligned
    const int sIN = 0, sBK = lt, sFW = 2 * lt, sMK = 3 * lt, sWARP = 4 * lt, sMISC = 4 * lt + 1, sALN = 4 * lt + 2;
    const Act prop{PG_PROP, 4 * lt + 3, h, w, 128, 1};
    const Act feat{PG_FEAT, t, h, w, 128, 3};
    propHalo = prop.halo; featHalo = feat.halo;
    const int ntok = t * fh * fw;
    const Act xTok{PG_X, t, fh, fw, 512, 0};

    // ---- encoder (:196-224)
    if (!cached) {
    {
        Op& op = ew(EW_PP_IM2COL3, "enc.im2col");
        op.ibuf[0] = PB_IN_FRAMES; op.ibuf[1] = PB_IN_MASK_U8; op.ibuf[2] = PB_IN_MASK_UPD_U8; op.ibuf[3] = PG_IM2COL;
        op.ipar[0] = t; op.ipar[1] = H; op.ipar[2] = W;
        need(PG_IM2COL, (int64_t)t * H2 * W2 * 64);
    }
    const Act cols{PG_IM2COL, t, H2, W2, 64, 0}, e0{PG_E0, t, H2, W2, 64, 1}, e1{PG_E1, t, H2, W2, 64, 1}, e2{PG_E2, t, h, w, 128, 1};
    // one NHWC buffer [x0 | layer 8 | layer 10 | layer 12 (padded) | layer 14] so that the grouped layers gather their
    // re-injected x0 slices and the previous layer's slices (view(bt, g, -1, h, w) + cat, :218-222) through chunk lists
    const int cX0 = 0, cL8 = 256, cL10 = 640, cL12 = 1152, cL14 = 1664;
    const Act enc{PG_ENC, t, h, w, 1920, 1};
    conv("enc.0", cols, idsT, chunks(0, 2), 1, 1, 1, 1, e0, idsT, 0, m_.enc0, L, nullptr, nullptr);
    conv("enc.2", e0, idsT, chunks(0, 2), 3, 3, 1, 1, e1, idsT, 0, m_.enc2, L, nullptr, nullptr);
    conv("enc.4", e1, idsT, chunks(0, 2), 3, 3, 2, 1, e2, idsT, 0, m_.enc4, L, nullptr, nullptr);
    conv("enc.6", e2, idsT, chunks(0, 4), 3, 3, 1, 1, enc, idsT, cX0, m_.enc6, L, nullptr, nullptr);
    conv("enc.8", enc, idsT, chunks(cX0, 8), 3, 3, 1, 1, enc, idsT, cL8, m_.enc8, L, nullptr, nullptr);
    {
        Op g;
        g.kind = OP_GEMM; g.tag = "enc.10"; g.bmode = VSR_BMODE_NK; g.tileCfg = VSR_TILE_128x64;
        for (int j = 0; j < 2; ++j)
            conv("enc.10", enc, idsT, cat(chunks(cX0 + 128 * j, 4), chunks(cL8 + 192 * j, 6)), 3, 3, 1, 1, enc, idsT, cL10 + 256 * j, m_.enc10[j], L,
                 nullptr, nullptr, &g);
        ops.push_back(std::move(g));
    }
    {
        Op g;
        g.kind = OP_GEMM; g.tag = "enc.12"; g.bmode = VSR_BMODE_NK; g.tileCfg = VSR_TILE_128x64;
        for (int j = 0; j < 4; ++j)
            conv("enc.12", enc, idsT, cat(chunks(cX0 + 64 * j, 2), chunks(cL10 + 128 * j, 4)), 3, 3, 1, 1, enc, idsT, cL12 + 128 * j, m_.enc12[j], L,
                 nullptr, nullptr, &g);
        ops.push_back(std::move(g));
    }
    {
        Op g;
        g.kind = OP_GEMM; g.tag = "enc.14"; g.bmode = VSR_BMODE_NK; g.tileCfg = VSR_TILE_256x32;
        for (int j = 0; j < 8; ++j)
            conv("enc.14", enc, idsT, cat(chunks(cX0 + 32 * j, 1), chunks(cL12 + 64 * j, 2)), 3, 3, 1, 1, enc, idsT, cL14 + 32 * j, m_.enc14[j], L,
                 nullptr, nullptr, &g);
        ops.push_back(std::move(g));
    }
    if (encodeOnly) {
        // every frame's features as a reference frame gets them (PG_FEAT, halo 3: the engine copies the interiors out for the windows
        // in which the frame is a local one), then its soft-split tokens.  Same GEMM rows and K order as in the full plan: same bits.
        need(PG_FEAT, feat.elems());
        Op g;
        g.kind = OP_GEMM; g.tag = "enc.16"; g.bmode = VSR_BMODE_NK; g.tileCfg = VSR_TILE_128x64;
        const std::vector<int> ch16 = cat(chunks(cX0, 8), chunks(cL14, 8));
        conv("enc.16", enc, idsT, ch16, 3, 3, 1, 1, feat, idsT, 0, m_.enc16, L, nullptr, nullptr, &g);
        ops.push_back(std::move(g));
        const Act tokOut{PG_TOKOUT, encTok > 0 ? encTok : 1, fh, fw, 512, 0};
        need(PG_TOKOUT, tokOut.elems());
        if (encTok > 0) conv("ss", feat, iota(encTok), chunks(0, 4), 7, 7, 3, 1, tokOut, iota(encTok), 0, m_.ss, VSR_ACT_NONE, nullptr, nullptr);
        refFlops = flops;
        return;
    }
    need(PG_PROP, prop.elems());
    need(PG_FEAT, feat.elems());
    {   // layer 16 over cat[x0, layer 14]: local frames go to the propagation input slots, reference frames straight to enc_feat
        Op g;
        g.kind = OP_GEMM; g.tag = "enc.16"; g.bmode = VSR_BMODE_NK; g.tileCfg = VSR_TILE_128x64;
        const std::vector<int> ch16 = cat(chunks(cX0, 8), chunks(cL14, 8));
        conv("enc.16", enc, idsL, ch16, 3, 3, 1, 1, prop, iota(lt, sIN), 0, m_.enc16, L, nullptr, nullptr, &g);
        if (t > lt) conv("enc.16", enc, idsR, ch16, 3, 3, 1, 1, feat, idsR, 0, m_.enc16, L, nullptr, nullptr, &g);
        ops.push_back(std::move(g));
    }
    } else {        // PP_PLAN_CACHED: the engine has put the local frames' features into slots sIN .. and the reference frames' tokens into PG_X
        need(PG_PROP, prop.elems());
        need(PG_FEAT, feat.elems());
    }

    // ---- flows and masks at 1/4 resolution (:341-350)
    const int64_t hw = (int64_t)h * w, slotElems = prop.frameElems();
    need(PG_DSF_F, (int64_t)(lt > 1 ? lt - 1 : 1) * 2 * hw);
    need(PG_DSF_B, (int64_t)(lt > 1 ? lt - 1 : 1) * 2 * hw);
    if (lt > 1) {
        for (int d = 0; d < 2; ++d) {
            Op& op = ew(EW_PP_DS_FLOW, "prop.dsflow");
            op.ibuf[0] = d == 0 ? PB_IN_FLOW_F : PB_IN_FLOW_B; op.ibuf[1] = d == 0 ? PG_DSF_F : PG_DSF_B;
            op.ipar[0] = (lt - 1) * 2; op.ipar[1] = H; op.ipar[2] = W;
        }
    }
    {
        Op& op = ew(EW_PP_DS_MASK, "prop.dsmask");
        op.ibuf[0] = PB_IN_MASK_U8; op.ibuf[1] = PB_IN_MASK_UPD_U8; op.ibuf[2] = PG_PROP; op.ioff[2] = (int64_t)sMK * slotElems;
        op.ipar[0] = lt; op.ipar[1] = H; op.ipar[2] = W; op.ipar[3] = prop.halo; op.ipar[4] = prop.C;
    }

    // ---- feature propagation: BidirectionalPropagation(128, learnable=True).forward (:104-193)
    const Act t1{PG_T1, 1, h, w, 128, 1}, t2{PG_T2, 1, h, w, 128, 1}, t3{PG_T3, 1, h, w, 128, 1};
    const Act offb{PG_OFF, 1, h, w, 448, 0}, colsb{PG_COLS, 1, h, w, 1152, 0}, bb{PG_BB, 1, h, w, 128, 1}, fu{PG_FU, lt, h, w, 128, 1};
    const std::vector<int> id0{0};
    int BM, BN;
    tileDims(VSR_TILE_128x64, BM, BN);
    const int tRowSlot0 = tRowsAct(prop, id0, h, w, 1, BM, 0);
    auto slotCols = [&](const std::vector<std::pair<int, int>>& parts) {     // (slot, chunks) list -> 3x3 column table with absolute slot offsets
        std::string key = "SLOT:";
        for (auto& p : parts) key += std::to_string(p.first) + "/" + std::to_string(p.second) + ",";
        auto itk = tableKey_.find(key);
        if (itk != tableKey_.end()) return itk->second;
        std::vector<int32_t> v;
        for (auto& p : parts)
            for (int c0 = 0; c0 < 32 * p.second; c0 += 32)
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) {
                        const int64_t o = (int64_t)p.first * slotElems + ((int64_t)(ky - 1) * prop.Wp() + (kx - 1)) * prop.C + c0;
                        if (o > 2147483647LL) throw std::runtime_error("propagation buffer offsets exceed int32");
                        v.push_back((int32_t)o);
                    }
        return table(key, std::move(v));
    };
    for (int mod = 0; mod < 2; ++mod) {
        const int sSrc = mod == 0 ? sIN : sBK, sDst = mod == 0 ? sBK : sFW;     // feats[cache_list[p_i]] (:121,140)
        for (int i = 0; i < lt; ++i) {
            const int idx = mod == 0 ? lt - 1 - i : i;
            const int prev = mod == 0 ? idx + 1 : idx - 1, flow = mod == 0 ? idx : idx - 1;
            int slotProp = sSrc + idx;                                           // i == 0: feat_prop = feat_current
            if (i > 0) {
                const int64_t foff = (int64_t)flow * 2 * hw;
                {
                    Op& op = ew(EW_PP_FEATPROP_PREP, "prop.prep");
                    op.ibuf[0] = PG_PROP; op.ibuf[1] = mod == 0 ? PG_DSF_F : PG_DSF_B; op.ibuf[2] = mod == 0 ? PG_DSF_B : PG_DSF_F;
                    op.ioff[0] = (int64_t)(sDst + prev) * slotElems;             // feat_prop = previous output of this direction
                    op.ioff[1] = foff; op.ioff[2] = foff; op.ioff[3] = (int64_t)(sMK + idx) * slotElems;
                    op.ipar[0] = h; op.ipar[1] = w; op.ipar[2] = prop.halo; op.ipar[3] = prop.C; op.ipar[4] = sWARP; op.ipar[5] = sMISC;
                }
                gemm("prop.off0", PG_PROP, 0, tRowSlot0, slotCols({{sSrc + idx, 4}, {sWARP, 4}, {sMISC, 1}}), 9 * 288, (int)hw, PG_T1, 0,
                     tRowsAct(t1, id0, h, w, 1, BM, 0), -1, m_.off[mod][0], VSR_ACT_LRELU01, -1, 0, -1, VSR_TILE_128x64);
                need(PG_T1, t1.elems());
                conv("prop.off1", t1, id0, chunks(0, 4), 3, 3, 1, 1, t2, id0, 0, m_.off[mod][1], VSR_ACT_LRELU01, nullptr, nullptr);
                conv("prop.off2", t2, id0, chunks(0, 4), 3, 3, 1, 1, t3, id0, 0, m_.off[mod][2], VSR_ACT_LRELU01, nullptr, nullptr);
                conv("prop.off3", t3, id0, chunks(0, 4), 3, 3, 1, 1, offb, id0, 0, m_.off[mod][3], VSR_ACT_NONE, nullptr, nullptr);
                {
                    Op& op = ew(EW_PP_DEFORM_COLS, "prop.deform.cols");
                    op.ibuf[0] = PG_PROP; op.ibuf[1] = PG_OFF; op.ibuf[2] = mod == 0 ? PG_DSF_F : PG_DSF_B; op.ibuf[3] = PG_COLS;
                    op.ioff[0] = (int64_t)(sDst + prev) * slotElems; op.ioff[2] = foff;
                    op.ipar[0] = h; op.ipar[1] = w; op.ipar[2] = prop.halo; op.ipar[3] = prop.C; op.ipar[4] = offb.C;
                    op.fpar[0] = 3.f;
                    need(PG_COLS, colsb.elems());
                }
                conv("prop.deform", colsb, id0, chunks(0, 36), 1, 1, 1, 1, prop, std::vector<int>{sALN}, 0, m_.deform[mod], VSR_ACT_NONE, nullptr, nullptr);
                slotProp = sALN;
            }
            gemm("prop.bb0", PG_PROP, 0, tRowSlot0, slotCols({{sSrc + idx, 4}, {slotProp, 4}, {sMK + idx, 1}}), 9 * 288, (int)hw, PG_BB, 0,
                 tRowsAct(bb, id0, h, w, 1, BM, 0), -1, m_.bb1[mod], VSR_ACT_LRELU02, -1, 0, -1, VSR_TILE_128x64);
            need(PG_BB, bb.elems());
            const std::vector<int> outIds{sDst + idx}, resIds{slotProp};
            conv("prop.bb1", bb, id0, chunks(0, 4), 3, 3, 1, 1, prop, outIds, 0, m_.bb2[mod], VSR_ACT_NONE, &prop, &resIds);
        }
    }
    {   // fuse over cat[backward, forward, mask] + x (:172-176) for the local frames -> enc_feat[:lt]
        std::vector<int32_t> v;
        const int rel[3] = {0, lt, 2 * lt};
        const int nch[3] = {4, 4, 1};
        for (int s = 0; s < 3; ++s)
            for (int c0 = 0; c0 < 32 * nch[s]; c0 += 32)
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) v.push_back((int32_t)((int64_t)rel[s] * slotElems + ((int64_t)(ky - 1) * prop.Wp() + (kx - 1)) * prop.C + c0));
        const int tCol = table("FUSE:" + std::to_string(lt), std::move(v));
        need(PG_FU, fu.elems());
        gemm("prop.fuse0", PG_PROP, 0, tRowsAct(prop, iota(lt, sBK), h, w, 1, BM, 0), tCol, 9 * 288, lt * (int)hw, PG_FU, 0,
             tRowsAct(fu, idsL, h, w, 1, BM, 0), -1, m_.fuse1, VSR_ACT_LRELU02, -1, 0, -1, VSR_TILE_128x64);
        const std::vector<int> inSlots = iota(lt, sIN);
        conv("prop.fuse1", fu, idsL, chunks(0, 4), 3, 3, 1, 1, feat, idsL, 0, m_.fuse2, VSR_ACT_NONE, &prop, &inSlots);
    }

    // ---- soft split (sparse_transformer.py:7-31): unfold 7x7 / stride 3 / pad 3 + Linear = a strided 7x7 conv
    if (cached) {
        need(PG_X, xTok.elems());
        conv("ss", feat, idsL, chunks(0, 4), 7, 7, 3, 1, xTok, idsL, 0, m_.ss, VSR_ACT_NONE, nullptr, nullptr);
    } else
    conv("ss", feat, idsT, chunks(0, 4), 7, 7, 3, 1, xTok, idsT, 0, m_.ss, VSR_ACT_NONE, nullptr, nullptr);

    // Only the local frames' output is produced, and PropainterInpaint reads it under the dilated mask only.  With a promise about
    // those rows / columns (decLo ...) the chain is walked backwards: a 3x3 conv widens a range by one, the align_corners x2
    // upsampling of n source rows reads rows floor(y (n - 1) / (2 n - 1)) and the next one for output row y (one more of slack on
    // each side for the kernel's float arithmetic), a token of the soft composition covers feature rows 3 ty - 3 .. 3 ty + 3.
    // The elementwise ops between the GEMMs (fold, the two upsamplings, tanh) keep whole images: what they compute outside the
    // ranges comes from rows no GEMM wrote in this pass, and nothing inside the ranges reads it.
    struct Rng { int lo, hi; };
    auto widen = [](Rng r, int by, int n) { return Rng{r.lo - by > 0 ? r.lo - by : 0, r.hi + by < n ? r.hi + by : n}; };
    auto below = [](Rng r, int nsrc) {
        const int on = 2 * nsrc;
        int lo = (int)((int64_t)r.lo * (nsrc - 1) / (on - 1)) - 1, hi = (int)((int64_t)(r.hi - 1) * (nsrc - 1) / (on - 1)) + 3;
        return Rng{lo > 0 ? lo : 0, hi < nsrc ? hi : nsrc};
    };
    auto tokens = [](Rng r, int ntok) {           // tokens whose 7x7 / stride 3 / padding 3 patch touches feature rows r
        int lo = r.lo - 3 <= 0 ? 0 : (r.lo - 3 + 2) / 3, hi = (r.hi + 2) / 3 + 1;
        return Rng{lo, hi < ntok ? hi : ntok};
    };
    const bool ranged = decLo > 0 || decHi < H || decXLo > 0 || decXHi < W;
    const Rng rD3{decLo, decHi}, rD2 = widen(rD3, 1, H), rUp1 = widen(rD2, 1, H);
    const Rng rD1 = below(rUp1, 2 * h), rD0 = widen(rD1, 1, 2 * h), rUp0 = widen(rD0, 1, 2 * h);
    const Rng rDin = below(rUp0, h), rScf = widen(rDin, 1, h), rTok = tokens(rScf, fh);
    const Rng cD3{decXLo, decXHi}, cD2 = widen(cD3, 1, W), cUp1 = widen(cD2, 1, W);
    const Rng cD1 = below(cUp1, 2 * w), cD0 = widen(cD1, 1, 2 * w), cUp0 = widen(cD0, 1, 2 * w);
    const Rng cDin = below(cUp0, w), cScf = widen(cDin, 1, w), cTok = tokens(cScf, fw);
    // The LAST transformer block is read by the soft composition alone.  Under a box promise its output is needed on tokens rTok x cTok
    // of the local frames; fc2 is per token, the fold / unfold pair in front of it reads the fc1 output of every token whose patch
    // touches the feature rows those tokens cover (rF1 x cF1), and so do norm2, the projection and the attention queries.  The keys
    // and values (all frames, all tokens) are never restricted; the elementwise ops keep whole token grids.
    auto patchRows = [](Rng r, int n) { return Rng{3 * r.lo - 3 > 0 ? 3 * r.lo - 3 : 0, 3 * r.hi + 1 < n ? 3 * r.hi + 1 : n}; };
    const Rng rF1 = tokens(patchRows(rTok, h), fh), cF1 = tokens(patchRows(cTok, w), fw);
    auto tokTable = [&](const char* name, Rng ry, Rng rx, int64_t ld, bool padded) {
        std::vector<int32_t> v;
        for (int f = 0; f < lt; ++f)
            for (int ty = ry.lo; ty < ry.hi; ++ty)
                for (int tx = rx.lo; tx < rx.hi; ++tx) {
                    const int64_t o = (padded ? ((int64_t)f * gh + ty) * gw + tx : ((int64_t)f * fh + ty) * fw + tx) * ld;
                    if (o > 2147483647LL) throw std::runtime_error("offset table entry exceeds int32");
                    v.push_back((int32_t)o);
                }
        while (v.size() % BM) v.push_back(v[0]);
        return table(std::string(name) + ":" + std::to_string(lt) + ":" + std::to_string(ry.lo) + "-" + std::to_string(ry.hi) + ":" +
                         std::to_string(rx.lo) + "-" + std::to_string(rx.hi) + ":" + std::to_string(ld) + (padded ? "p" : ""),
                     std::move(v));
    };
    // ---- 8 TemporalSparseTransformer blocks (:273-344)
    const int64_t gridRows = (int64_t)t * gh * gw, poolRows = (int64_t)t * ph * pw, qRows = gridRows + poolRows;
    need(PG_YQ, qRows * 512);
    need(PG_QKV, qRows * 1536);
    need(PG_ATT, gridRows * 512);
    need(PG_Y2, (int64_t)ntok * 512);
    need(PG_F1, (int64_t)ntok * 1984);
    need(PG_FMAP, (int64_t)t * hw * 40);
    need(PG_F2, (int64_t)ntok * 1984);
    std::vector<int32_t> validTok;                                       // un-padded tokens inside the padded grid, token order
    for (int f = 0; f < t; ++f)
        for (int y = 0; y < fh; ++y)
            for (int x = 0; x < fw; ++x) validTok.push_back((int32_t)((((int64_t)f * gh + y) * gw + x) * 512));
    while (validTok.size() % BM) validTok.push_back(validTok[0]);
    const int tValid = table("VALIDTOK", std::move(validTok));
    const int tRowX = tRowsLinear(ntok, 512, BM);
    for (int i = 0; i < 8; ++i) {
        const PpBlockW& bw = m_.blk[i];
        {
            Op& op = ew(EW_PP_LAYERNORM, "tr.norm1");
            op.ibuf[0] = PG_X; op.ibuf[1] = PG_YQ; op.ioff[0] = bw.ln1g; op.ioff[1] = bw.ln1b;
            op.ipar[0] = t; op.ipar[1] = fh; op.ipar[2] = fw; op.ipar[3] = 512; op.ipar[4] = gh; op.ipar[5] = gw;
        }
        {
            Op& op = ew(EW_PP_POOL, "tr.pool");
            op.ibuf[0] = PG_YQ; op.ioff[1] = gridRows * 512; op.ioff[2] = bw.poolW; op.ioff[3] = bw.poolB;
            op.ipar[0] = t; op.ipar[1] = gh; op.ipar[2] = gw; op.ipar[3] = 512; op.ipar[4] = ph; op.ipar[5] = pw;
        }
        gemm("tr.qkv", PG_YQ, 0, tRowsLinear((int)qRows, 512, BM), tColsLinear(16, 16), 512, (int)qRows, PG_QKV, 0, tRowsLinear((int)qRows, 1536, BM),
             -1, bw.qkv, VSR_ACT_NONE, -1, 0, -1, VSR_TILE_128x64);
        const bool boxed = ranged && i == 7;
        const int mF1 = lt * (rF1.hi - rF1.lo) * (cF1.hi - cF1.lo), mTok = lt * (rTok.hi - rTok.lo) * (cTok.hi - cTok.lo);
        if (boxed) attention(i, windowMasked, lt, rF1.lo, rF1.hi, cF1.lo, cF1.hi);
        else attention(i, windowMasked, t, 0, fh, 0, fw);
        // x = shortcut + proj(att) on the un-padded tokens (:263-269,288-291)
        if (boxed) {
            const int tx512 = tokTable("TOKBOX", rF1, cF1, 512, false);
            gemm("tr.proj", PG_ATT, 0, tokTable("TOKBOX", rF1, cF1, 512, true), tColsLinear(16, 16), 512, mF1, PG_X, 0, tx512, -1, bw.proj, VSR_ACT_NONE,
                 PG_X, 0, tx512, VSR_TILE_128x64);
            trimmedFlops_ += 2.0 * (ntok - mF1) * 512.0 * 512;
        } else
        gemm("tr.proj", PG_ATT, 0, tValid, tColsLinear(16, 16), 512, ntok, PG_X, 0, tRowX, -1, bw.proj, VSR_ACT_NONE, PG_X, 0, tRowX, VSR_TILE_128x64);
        {
            Op& op = ew(EW_PP_LAYERNORM, "tr.norm2");
            op.ibuf[0] = PG_X; op.ibuf[1] = PG_Y2; op.ioff[0] = bw.ln2g; op.ioff[1] = bw.ln2b;
            op.ipar[0] = t; op.ipar[1] = fh; op.ipar[2] = fw; op.ipar[3] = 512; op.ipar[4] = fh; op.ipar[5] = fw;
        }
        // FusionFeedForward (:67-104): fc1, fold / normalise / unfold, GELU, fc2
        if (boxed) {
            gemm("tr.fc1", PG_Y2, 0, tokTable("TOKBOX", rF1, cF1, 512, false), tColsLinear(16, 16), 512, mF1, PG_F1, 0,
                 tokTable("TOKBOX", rF1, cF1, 1984, false), -1, bw.fc1, VSR_ACT_NONE, -1, 0, -1, VSR_TILE_128x64);
            trimmedFlops_ += 2.0 * (ntok - mF1) * (double)bw.fc1.cout * 512;
        } else
        gemm("tr.fc1", PG_Y2, 0, tRowX, tColsLinear(16, 16), 512, ntok, PG_F1, 0, tRowsLinear(ntok, 1984, BM), -1, bw.fc1, VSR_ACT_NONE, -1, 0, -1,
             VSR_TILE_128x64);
        {
            Op& op = ew(EW_PP_FOLD, "tr.fold");
            op.ibuf[0] = PG_F1; op.ibuf[1] = PG_FMAP;
            op.ipar[0] = 1984; op.ipar[1] = t; op.ipar[2] = fh; op.ipar[3] = fw; op.ipar[4] = h; op.ipar[5] = w; op.ipar[6] = 40; op.ipar[7] = 0; op.ipar[8] = 1;
        }
        {
            Op& op = ew(EW_PP_UNFOLD_GELU, "tr.unfold");
            op.ibuf[0] = PG_FMAP; op.ibuf[1] = PG_F2;
            op.ipar[0] = t; op.ipar[1] = fh; op.ipar[2] = fw; op.ipar[3] = h; op.ipar[4] = w; op.ipar[5] = 40; op.ipar[6] = 1984;
        }
        if (boxed) {
            const int tx512 = tokTable("TOKBOX", rTok, cTok, 512, false);
            gemm("tr.fc2", PG_F2, 0, tokTable("TOKBOX", rTok, cTok, 1984, false), tColsLinear(62, 62), 1984, mTok, PG_X, 0, tx512, -1, bw.fc2,
                 VSR_ACT_NONE, PG_X, 0, tx512, VSR_TILE_128x64);
            trimmedFlops_ += 2.0 * (ntok - mTok) * 512.0 * 1984;
        } else
        gemm("tr.fc2", PG_F2, 0, tRowsLinear(ntok, 1984, BM), tColsLinear(62, 62), 1984, ntok, PG_X, 0, tRowX, -1, bw.fc2, VSR_ACT_NONE, PG_X, 0, tRowX,
             VSR_TILE_128x64);
    }

    // ---- soft composition (:34-64) of the local frames + enc_feat, decoder (:270-277,371-376)
    const int ntokL = lt * fh * fw;
    need(PG_SC, (int64_t)ntokL * 6272);
    if (!ranged) {
        gemm("sc.embed", PG_X, 0, tRowsLinear(ntokL, 512, BM), tColsLinear(16, 16), 512, ntokL, PG_SC, 0, tRowsLinear(ntokL, 6272, BM), -1, m_.sc,
             VSR_ACT_NONE, -1, 0, -1, VSR_TILE_128x64);
    } else {
        std::vector<int32_t> ra, rc;
        for (int f = 0; f < lt; ++f)
            for (int ty = rTok.lo; ty < rTok.hi; ++ty)
                for (int tx = cTok.lo; tx < cTok.hi; ++tx) {
                    const int64_t tok = ((int64_t)f * fh + ty) * fw + tx;
                    if (tok * 6272 > 2147483647LL) throw std::runtime_error("offset table entry exceeds int32");
                    ra.push_back((int32_t)(tok * 512));
                    rc.push_back((int32_t)(tok * 6272));
                }
        const int M = (int)ra.size();
        trimmedFlops_ += 2.0 * (ntokL - M) * 6272.0 * 512;
        while (ra.size() % BM) { ra.push_back(ra[0]); rc.push_back(rc[0]); }
        const std::string key = std::to_string(lt) + ":" + std::to_string(rTok.lo) + "-" + std::to_string(rTok.hi) + ":" + std::to_string(cTok.lo) +
                                "-" + std::to_string(cTok.hi);
        gemm("sc.embed", PG_X, 0, table("SCTOKA:" + key, std::move(ra)), tColsLinear(16, 16), 512, M, PG_SC, 0, table("SCTOKC:" + key, std::move(rc)),
             -1, m_.sc, VSR_ACT_NONE, -1, 0, -1, VSR_TILE_128x64);
    }
    const Act scf{PG_SCF, lt, h, w, 128, 1}, din{PG_DIN, lt, h, w, 128, 0};
    {
        Op& op = ew(EW_PP_FOLD, "sc.fold");
        op.ibuf[0] = PG_SC; op.ibuf[1] = PG_SCF;
        op.ipar[0] = 6272; op.ipar[1] = lt; op.ipar[2] = fh; op.ipar[3] = fw; op.ipar[4] = h; op.ipar[5] = w; op.ipar[6] = 128; op.ipar[7] = 1; op.ipar[8] = 0;
        need(PG_SCF, scf.elems());
    }
    if (ranged) convRect("sc.conv", scf, idsL, chunks(0, 4), din, m_.scConv, VSR_ACT_NONE, &feat, rDin.lo, rDin.hi, cDin.lo, cDin.hi);
    else conv("sc.conv", scf, idsL, chunks(0, 4), 3, 3, 1, 1, din, idsL, 0, m_.scConv, VSR_ACT_NONE, &feat, &idsL);
    const Act up0{PG_UP0, lt, 2 * h, 2 * w, 128, 1}, d0{PG_D0, lt, 2 * h, 2 * w, 128, 1}, d1{PG_D1, lt, 2 * h, 2 * w, 64, 0};
    const Act up1{PG_UP1, lt, H, W, 64, 1}, d2{PG_D2, lt, H, W, 64, 1}, d3{PG_D3, lt, H, W, 32, 0};
    upsample(din, up0);
    if (ranged) {
        convRect("dec.0", up0, idsL, chunks(0, 4), d0, m_.dec0, L, nullptr, rD0.lo, rD0.hi, cD0.lo, cD0.hi);
        convRect("dec.2", d0, idsL, chunks(0, 4), d1, m_.dec2, L, nullptr, rD1.lo, rD1.hi, cD1.lo, cD1.hi);
    } else {
        conv("dec.0", up0, idsL, chunks(0, 4), 3, 3, 1, 1, d0, idsL, 0, m_.dec0, L, nullptr, nullptr);
        conv("dec.2", d0, idsL, chunks(0, 4), 3, 3, 1, 1, d1, idsL, 0, m_.dec2, L, nullptr, nullptr);
    }
    upsample(d1, up1);
    if (ranged) {
        convRect("dec.4", up1, idsL, chunks(0, 2), d2, m_.dec4, L, nullptr, rD2.lo, rD2.hi, cD2.lo, cD2.hi);
        convRect("dec.6", d2, idsL, chunks(0, 2), d3, m_.dec6, VSR_ACT_NONE, nullptr, rD3.lo, rD3.hi, cD3.lo, cD3.hi);
    } else {
        conv("dec.4", up1, idsL, chunks(0, 2), 3, 3, 1, 1, d2, idsL, 0, m_.dec4, L, nullptr, nullptr);
        conv("dec.6", d2, idsL, chunks(0, 2), 3, 3, 1, 1, d3, idsL, 0, m_.dec6, VSR_ACT_NONE, nullptr, nullptr);
    }
    {
        Op& op = ew(EW_PP_TANH_OUT, "dec.tanh");
        op.ibuf[0] = PG_D3; op.ibuf[1] = PG_OUT;
        op.ipar[0] = 32; op.ipar[1] = lt; op.ipar[2] = H; op.ipar[3] = W;
        need(PG_OUT, (int64_t)lt * 3 * HW);
    }
    refFlops = flops + trimmedFlops_;
}

} // namespace vsr
