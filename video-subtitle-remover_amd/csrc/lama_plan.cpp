// Weight packer + plan builder for the big-LaMa generator (see lama_plan.h).
#include "lama_plan.h"
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
static int32_t fits(int64_t v)
{
    if (v > 2147483647LL || v < -2147483648LL) throw std::runtime_error("offset table entry does not fit 32 bits");
    return (int32_t)v;
}

// ------------------------------------------------------------------------------------
// LamaModel
// ------------------------------------------------------------------------------------
bool LamaModel::set_param(const std::string& name_, const float* data, const int64_t* shape, int ndim, std::string& err)
{
    std::string name = name_;
    if (name.compare(0, 10, "generator.") == 0) name = name.substr(10);          // keys of the whole exported module
    if (name.compare(0, 6, "model.") != 0) { err = "unexpected key in state_dict: " + name_; return false; }
    const size_t dot = name.rfind('.');
    if (dot != std::string::npos && name.substr(dot + 1) == "num_batches_tracked") return true;
    Raw r;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) { r.shape.push_back(shape[i]); n *= shape[i]; }
    r.v.assign(data, data + n);
    raw_[name] = std::move(r);
    ready_ = false;
    return true;
}

const LamaModel::Raw* LamaModel::get(const std::string& key, std::string& err) const
{
    auto it = raw_.find(key);
    if (it == raw_.end()) { err = "missing key in state_dict: " + key; return nullptr; }
    return &it->second;
}

// eval-mode BatchNorm2d as y = x * scale + shift
bool LamaModel::bn_affine(const std::string& bn, int c, std::vector<float>& scale, std::vector<float>& shift, std::string& err) const
{
    const char* leaves[4] = {"weight", "bias", "running_mean", "running_var"};
    const Raw* r[4];
    for (int i = 0; i < 4; ++i) {
        r[i] = get(bn + "." + leaves[i], err);
        if (!r[i]) return false;
        if ((int64_t)r[i]->v.size() != c) { err = "shape mismatch for " + bn + "." + leaves[i]; return false; }
    }
    scale.resize(c);
    shift.resize(c);
    for (int n = 0; n < c; ++n) {
        const float g = r[0]->v[n] / sqrtf(r[3]->v[n] + 1e-5f);
        scale[n] = g;
        shift[n] = r[1]->v[n] - r[2]->v[n] * g;
    }
    return true;
}

namespace {
struct Part { const float* w; int cout, cin, ci0, co0; };      // source [cout][cin][k][k] placed at rows co0.., channels ci0..
}

// [coutTotal][K] with K = rup(k*k*cinTotal, 32); K order mirrors PlanBuilder::tColsConvHW (channel-chunk major when possible)
static void pack_parts(std::vector<float>& packed, ConvW& cw, const std::vector<Part>& parts, int coutTotal, int cinTotal, int k,
                       const float* scale, const float* bias)
{
    const int taps = k * k;
    const int K = (int)rup((int64_t)taps * cinTotal, VSR_GG_KC);
    cw.cout = coutTotal;
    cw.K = K;
    cw.w = (int64_t)packed.size();
    packed.resize(packed.size() + (size_t)rup((int64_t)coutTotal * K, 32), 0.f);
    float* dst = packed.data() + cw.w;
    const bool chanMajor = Tuning::get().convChannelMajor && (cinTotal % VSR_GG_KC == 0);
    for (const Part& p : parts)
        for (int n = 0; n < p.cout; ++n)
            for (int c = 0; c < p.cin; ++c)
                for (int tap = 0; tap < taps; ++tap) {
                    const int ci = p.ci0 + c, row = p.co0 + n;
                    const int kk = chanMajor ? ((ci / VSR_GG_KC) * taps + tap) * VSR_GG_KC + (ci % VSR_GG_KC) : tap * cinTotal + ci;
                    dst[(int64_t)row * K + kk] = p.w[((int64_t)n * p.cin + c) * taps + tap] * (scale ? scale[row] : 1.f);
                }
    cw.b = (int64_t)packed.size();
    packed.resize(packed.size() + (size_t)rup(coutTotal, 32), 0.f);
    for (int n = 0; n < coutTotal; ++n) packed[cw.b + n] = bias ? bias[n] : 0.f;
}

static bool shape_is(const std::vector<int64_t>& s, int a, int b, int c, int d) { return s.size() == 4 && s[0] == a && s[1] == b && s[2] == c && s[3] == d; }

bool LamaModel::pack_ffc(const std::string& p, LamaFfcW& f, std::string& err)
{
    const Raw *l2l = get(p + ".ffc.convl2l.weight", err), *l2g = get(p + ".ffc.convl2g.weight", err), *g2l = get(p + ".ffc.convg2l.weight", err);
    const Raw *c1 = get(p + ".ffc.convg2g.conv1.0.weight", err), *fu = get(p + ".ffc.convg2g.fu.conv_layer.weight", err);
    const Raw* c2 = get(p + ".ffc.convg2g.conv2.weight", err);
    if (!l2l || !l2g || !g2l || !c1 || !fu || !c2) return false;
    if (!shape_is(l2l->shape, LAMA_CL, LAMA_CL, 3, 3) || !shape_is(l2g->shape, LAMA_CG, LAMA_CL, 3, 3) || !shape_is(g2l->shape, LAMA_CL, LAMA_CG, 3, 3) ||
        !shape_is(c1->shape, LAMA_CS, LAMA_CG, 1, 1) || !shape_is(fu->shape, LAMA_CG, LAMA_CG, 1, 1) || !shape_is(c2->shape, LAMA_CG, LAMA_CS, 1, 1)) {
        err = "shape mismatch in " + p;
        return false;
    }
    std::vector<float> sl, tl, sg, tg, s1, t1, sf, tf;
    if (!bn_affine(p + ".bn_l", LAMA_CL, sl, tl, err) || !bn_affine(p + ".bn_g", LAMA_CG, sg, tg, err) ||
        !bn_affine(p + ".ffc.convg2g.conv1.1", LAMA_CS, s1, t1, err) || !bn_affine(p + ".ffc.convg2g.fu.bn", LAMA_CG, sf, tf, err))
        return false;
    // out_xl = bn_l(convl2l(x_l) + convg2l(x_g)): one conv over the 512-channel tensor [x_l | x_g]
    pack_parts(packed, f.outL, {{l2l->v.data(), LAMA_CL, LAMA_CL, 0, 0}, {g2l->v.data(), LAMA_CL, LAMA_CG, LAMA_CL, 0}}, LAMA_CL, LAMA_C, 3, sl.data(), tl.data());
    // out_xg = bn_g(convl2g(x_l) + convg2g(x_g)): the affine is split -- scale on both summands, shift with the first
    pack_parts(packed, f.l2g, {{l2g->v.data(), LAMA_CG, LAMA_CL, 0, 0}}, LAMA_CG, LAMA_CL, 3, sg.data(), tg.data());
    pack_parts(packed, f.st1, {{c1->v.data(), LAMA_CS, LAMA_CG, 0, 0}}, LAMA_CS, LAMA_CG, 1, s1.data(), t1.data());
    // FourierUnit's conv sees channels (c, part) interleaved as 2c + part (ffc.py: stack(re, im).permute(0,1,4,2,3).view(b, 2c, ..));
    // here a frequency pixel holds [part][c], so both channel axes are permuted
    std::vector<float> wp((size_t)LAMA_CG * LAMA_CG), sp(LAMA_CG), tp(LAMA_CG);
    auto src = [](int packedIdx) { return 2 * (packedIdx % LAMA_CS) + packedIdx / LAMA_CS; };
    for (int n = 0; n < LAMA_CG; ++n) {
        sp[n] = sf[src(n)];
        tp[n] = tf[src(n)];
        for (int kx = 0; kx < LAMA_CG; ++kx) wp[(size_t)n * LAMA_CG + kx] = fu->v[(size_t)src(n) * LAMA_CG + src(kx)];
    }
    pack_parts(packed, f.fu, {{wp.data(), LAMA_CG, LAMA_CG, 0, 0}}, LAMA_CG, LAMA_CG, 1, sp.data(), tp.data());
    pack_parts(packed, f.st2, {{c2->v.data(), LAMA_CG, LAMA_CS, 0, 0}}, LAMA_CG, LAMA_CS, 1, sg.data(), nullptr);
    return true;
}

// ConvTranspose2d(cin, cout, 3, stride 2, padding 1, output_padding 1) + BatchNorm: out[2i+a][2j+b] only sees the taps with
// ky = 1 (a = 0) or ky in {0, 2} (a = 1), same along x; tap (ky, kx) reads input pixel (i + (ky == 0), j + (kx == 0))
bool LamaModel::pack_up(const std::string& key, const std::string& bn, int cin, int cout, ConvW out[4], std::string& err)
{
    const Raw *w = get(key + ".weight", err), *b = get(key + ".bias", err);
    if (!w || !b) return false;
    if (!shape_is(w->shape, cin, cout, 3, 3) || (int)b->v.size() != cout) { err = "shape mismatch for " + key; return false; }
    std::vector<float> s, t;
    if (!bn_affine(bn, cout, s, t, err)) return false;
    for (int a = 0; a < 2; ++a)
        for (int bb = 0; bb < 2; ++bb) {
            std::vector<int> kys = a ? std::vector<int>{0, 2} : std::vector<int>{1}, kxs = bb ? std::vector<int>{0, 2} : std::vector<int>{1};
            const int ntaps = (int)(kys.size() * kxs.size());
            ConvW& cw = out[a * 2 + bb];
            cw.cout = cout;
            cw.K = ntaps * cin;
            cw.w = (int64_t)packed.size();
            packed.resize(packed.size() + (size_t)rup((int64_t)cout * cw.K, 32), 0.f);
            float* dst = packed.data() + cw.w;
            for (int n = 0; n < cout; ++n)
                for (int ci = 0; ci < cin; ++ci) {
                    int tix = 0;
                    for (int ky : kys)
                        for (int kx : kxs) {
                            const int kk = ((ci / VSR_GG_KC) * ntaps + tix) * VSR_GG_KC + (ci % VSR_GG_KC);
                            dst[(int64_t)n * cw.K + kk] = w->v[(((int64_t)ci * cout + n) * 3 + ky) * 3 + kx] * s[n];
                            ++tix;
                        }
                }
            cw.b = (int64_t)packed.size();
            packed.resize(packed.size() + (size_t)rup(cout, 32), 0.f);
            for (int n = 0; n < cout; ++n) packed[cw.b + n] = b->v[n] * s[n] + t[n];
        }
    return true;
}

bool LamaModel::pack(std::string& err)
{
    packed.clear();
    ffc.clear();
    nBlocks = 0;
    while (raw_.count("model." + std::to_string(5 + nBlocks) + ".conv1.ffc.convl2l.weight")) ++nBlocks;
    if (nBlocks < 1) { err = "no FFC residual block (model.5.conv1...) in the state_dict"; return false; }
    auto plain = [&](const std::string& key, const std::string& bn, ConvW& cw, int cout, int cin, int k) -> bool {
        const Raw* w = get(key + ".weight", err);
        if (!w) return false;
        if (!shape_is(w->shape, cout, cin, k, k)) { err = "shape mismatch for " + key; return false; }
        std::vector<float> s, t;
        if (!bn_affine(bn, cout, s, t, err)) return false;
        pack_parts(packed, cw, {{w->v.data(), cout, cin, 0, 0}}, cout, cin, k, s.data(), t.data());
        return true;
    };
    if (!plain("model.1.ffc.convl2l", "model.1.bn_l", stem, 64, 4, 7)) return false;
    if (!plain("model.2.ffc.convl2l", "model.2.bn_l", down[0], 128, 64, 3)) return false;
    if (!plain("model.3.ffc.convl2l", "model.3.bn_l", down[1], 256, 128, 3)) return false;
    {
        const Raw *a = get("model.4.ffc.convl2l.weight", err), *b = get("model.4.ffc.convl2g.weight", err);
        if (!a || !b) return false;
        if (!shape_is(a->shape, LAMA_CL, 256, 3, 3) || !shape_is(b->shape, LAMA_CG, 256, 3, 3)) { err = "shape mismatch for model.4"; return false; }
        std::vector<float> sl, tl, sg, tg;
        if (!bn_affine("model.4.bn_l", LAMA_CL, sl, tl, err) || !bn_affine("model.4.bn_g", LAMA_CG, sg, tg, err)) return false;
        sl.insert(sl.end(), sg.begin(), sg.end());
        tl.insert(tl.end(), tg.begin(), tg.end());
        pack_parts(packed, down3, {{a->v.data(), LAMA_CL, 256, 0, 0}, {b->v.data(), LAMA_CG, 256, 0, LAMA_CL}}, LAMA_C, 256, 3, sl.data(), tl.data());
    }
    for (int i = 0; i < nBlocks; ++i)
        for (const char* cv : {".conv1", ".conv2"}) {
            LamaFfcW f;
            if (!pack_ffc("model." + std::to_string(5 + i) + cv, f, err)) return false;
            ffc.push_back(f);
        }
    const int base = 5 + nBlocks + 1;                       // ConcatTupleLayer sits at model.(5 + nBlocks)
    const int ch[4] = {512, 256, 128, 64};
    for (int j = 0; j < 3; ++j)
        if (!pack_up("model." + std::to_string(base + 3 * j), "model." + std::to_string(base + 3 * j + 1), ch[j], ch[j + 1], up[j], err)) return false;
    {
        const std::string key = "model." + std::to_string(base + 10);
        const Raw *w = get(key + ".weight", err), *b = get(key + ".bias", err);
        if (!w || !b) return false;
        if (!shape_is(w->shape, 3, 64, 7, 7) || b->v.size() != 3) { err = "shape mismatch for " + key; return false; }
        // The 64 -> 3 conv as a GEMM with N = 3 wastes a 32-wide tile and re-reads 49 x 64 inputs per output pixel.  It is packed
        // for 4 x 4 output blocks instead: row = block, N = (dy, dx, c) = 48 outputs, K = the block's 10 x 10 input window x 64
        // channels (weights of taps outside a pixel's own 7 x 7 window are zero): a quarter of the padded FLOPs and an eighth of the
        // operand traffic.
        const int BLK = LAMA_OUT_BLOCK, WIN = BLK + 6, taps = WIN * WIN, cin = 64, nOut = BLK * BLK * 3;
        last.cout = nOut;
        last.K = taps * cin;
        last.w = (int64_t)packed.size();
        packed.resize(packed.size() + (size_t)rup((int64_t)nOut * last.K, 32), 0.f);
        float* dst = packed.data() + last.w;
        for (int dy = 0; dy < BLK; ++dy)
            for (int dx = 0; dx < BLK; ++dx)
                for (int c = 0; c < 3; ++c) {
                    const int n = (dy * BLK + dx) * 3 + c;
                    for (int ky = 0; ky < 7; ++ky)
                        for (int kx = 0; kx < 7; ++kx)
                            for (int ci = 0; ci < cin; ++ci) {
                                const int tap = (dy + ky) * WIN + (dx + kx);
                                const int kk = ((ci / VSR_GG_KC) * taps + tap) * VSR_GG_KC + (ci % VSR_GG_KC);
                                dst[(int64_t)n * last.K + kk] = w->v[(((int64_t)c * cin + ci) * 7 + ky) * 7 + kx];
                            }
                }
        last.b = (int64_t)packed.size();
        packed.resize(packed.size() + (size_t)rup(nOut, 32), 0.f);
        for (int n = 0; n < nOut; ++n) packed[last.b + n] = b->v[n % 3];
    }
    ready_ = true;
    return true;
}

// ------------------------------------------------------------------------------------
// LamaPlan
// ------------------------------------------------------------------------------------
int LamaPlan::pickTile(int N) const { return N <= 32 ? VSR_TILE_256x32 : (N <= 64 ? n64Tile() : VSR_TILE_128x64); }

Op& LamaPlan::ew(int kind, const char* tag)
{
    Op op;
    op.kind = OP_EW;
    op.ew = kind;
    op.tag = tag;
    ops.push_back(std::move(op));
    return ops.back();
}

void LamaPlan::conv(const char* tag, const Act& in, int c0in, int cin, const Act& out, int c0out, int k, int stride, const ConvW& w, int act,
                    const Act* res, int c0res)
{
    if (w.K != (int)rup((int64_t)k * k * cin, VSR_GG_KC)) throw std::runtime_error(std::string("lama conv K mismatch: ") + tag);
    Op op;
    op.kind = OP_GEMM;
    op.tag = tag;
    op.bmode = VSR_BMODE_NK;
    op.tileCfg = pickTile(w.cout);
    int BM, BN;
    tileDims(op.tileCfg, BM, BN);
    const std::vector<int> ids = iota(B);
    GemmItem it{};
    it.M = B * out.H * out.W;
    it.N = w.cout;
    it.K = w.K;
    it.tilesM = cdiv(it.M, BM);
    it.tilesN = cdiv(it.N, BN);
    it.splitK = 1;
    it.chunksPerSplit = it.K / VSR_GG_KC;
    it.alpha = 1.f;
    it.act = act;
    it.bufA = in.buf; it.offA = 0;
    it.tRowA = tRowsAct(in, ids, out.H, out.W, stride, BM, 0);
    it.tColA = tColsConvHW(in, k, k, 1, c0in, cin);
    it.bufB = LB_WEIGHTS; it.offB = w.w;
    it.tRowB = tRowsLinear(it.N, it.K, BN);
    it.tColB = tColsLinear(it.K / VSR_GG_KC, it.K / VSR_GG_KC);
    it.bufC = out.buf; it.offC = 0;
    it.tRowC = tRowsAct(out, ids, out.H, out.W, 1, BM, c0out);
    it.tColC = tColsLinear(cdiv(it.N, VSR_GG_KC), it.tilesN * BN / VSR_GG_KC);
    it.offBias = w.b;
    if (res) {
        it.bufR = res->buf; it.offR = 0;
        it.tRowR = tRowsAct(*res, ids, out.H, out.W, 1, BM, c0res);
    } else {
        it.bufR = -1; it.offR = 0; it.tRowR = -1;
    }
    op.flops = 2.0 * it.M * it.N * (double)(k * k * cin);
    op.gemm.push_back(it);
    need(out.buf, out.elems());
    flops += op.flops;
    ops.push_back(std::move(op));
}

void LamaPlan::halo(const Act& a)
{
    Op& o = ew(EW_LAMA_HALO, "reflect.halo");
    o.ibuf[0] = a.buf;
    o.ipar[0] = a.n; o.ipar[1] = a.H; o.ipar[2] = a.W; o.ipar[3] = a.C; o.ipar[4] = a.halo;
}

void LamaPlan::addHalo(const Act& a, const Act& b, const Act& dst, bool reflect)
{
    if (a.halo != dst.halo || b.halo != dst.halo || a.C != dst.C || b.C != dst.C) throw std::runtime_error("residual add over unequal layouts");
    Op& o = ew(EW_LAMA_ADD_HALO, "block.residual");
    o.ibuf[0] = a.buf; o.ibuf[1] = b.buf; o.ibuf[2] = dst.buf;
    o.ipar[0] = dst.n; o.ipar[1] = dst.H; o.ipar[2] = dst.W; o.ipar[3] = dst.C; o.ipar[4] = dst.halo; o.ipar[5] = reflect ? 1 : 0;
    need(dst.buf, dst.elems());
}

// DFT matrices as plan constants, rows x rup(cols, 32) fp32, generated in double.  n = transform length.
//   kind 0: real -> half spectrum along W      rows m = 2 kx + part, cols x:      part 0: cos(2 pi kx x / n), part 1: -sin
//   kind 1: complex forward along H            rows 2 ky + po, cols 2 y + pi:     [[cos, sin], [-sin, cos]]
//   kind 2: complex inverse along H            rows 2 y + po, cols 2 ky + pi:     [[cos, -sin], [sin, cos]]
//   kind 3: half spectrum -> real along W      rows x, cols 2 kx + part:          a_kx cos, -a_kx sin (a = 1 for DC / Nyquist, else 2)
// every kind carries 1 / sqrt(n) (norm='ortho')
int64_t LamaPlan::dft(const std::string& key, int rows, int cols, int kind, int n)
{
    auto it = dftOff_.find(key);
    if (it != dftOff_.end()) return it->second;
    const int ld = (int)rup(cols, VSR_GG_KC);
    const int64_t off = (int64_t)consts.size();
    consts.resize(consts.size() + (size_t)rows * ld, 0.f);
    float* d = consts.data() + off;
    const double sc = 1.0 / sqrt((double)n), tau = 2.0 * M_PI / n;
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            double v = 0;
            if (kind == 0) {
                const int kx = r / 2, part = r % 2;
                const double th = tau * (double)(((int64_t)kx * c) % n);
                v = part == 0 ? cos(th) : -sin(th);
            } else if (kind == 1 || kind == 2) {
                const int o = r / 2, po = r % 2, i = c / 2, pi = c % 2;
                const double th = tau * (double)(((int64_t)o * i) % n);
                const double s = kind == 1 ? sin(th) : -sin(th);            // forward: e^{-i th}; inverse: e^{+i th}
                v = po == pi ? cos(th) : (po == 0 ? s : -s);               // re = a cos + b s ; im = -a s + b cos  (forward)
            } else {
                const int kx = c / 2, part = c % 2;
                const double a = (kx == 0 || (n % 2 == 0 && kx == n / 2)) ? 1.0 : 2.0;
                const double th = tau * (double)(((int64_t)kx * r) % n);
                v = part == 0 ? a * cos(th) : -a * sin(th);
            }
            d[(int64_t)r * ld + c] = (float)(v * sc);
        }
    dftOff_[key] = off;
    return off;
}

// FourierUnit.forward on S1 [B][h][w][192]: S2 = S1 + irfftn(relu(bn(conv(rfftn(S1)))))   (SpectralTransform adds x + fu(x)).
// Each DFT stage is ONE problem per image: the transform axis is the contraction (KN mode: B rows = positions along the axis),
// every other (position, channel) pair is a column -- 32-channel chunk j of the N axis is (other position j / 6, channels
// 32 (j % 6) ..), which the column tables express -- so a stage is [2 wf | 2 h | w] x [thousands of columns] in 128 x 128 tiles
// instead of hundreds of 3-tile problems.
void LamaPlan::fourier(const LamaFfcW& f)
{
    const int CS = LAMA_CS, CG = LAMA_CG, CH = CS / VSR_GG_KC;      // 6 chunks of 32 channels
    const int Kw = (int)rup(w, VSR_GG_KC), Kh = (int)rup(2 * h, VSR_GG_KC), Kc = (int)rup(2 * wf, VSR_GG_KC);
    const int cfg = VSR_TILE_128x128;
    int BM, BN;
    tileDims(cfg, BM, BN);
    auto makeTable = [&](const std::string& key, int count, int padTo, auto fn) {
        std::vector<int32_t> v;
        for (int i = 0; i < count; ++i) v.push_back(fits(fn(i)));
        const int32_t first = v[0];
        while ((int)v.size() % padTo) v.push_back(first);
        return table(key, std::move(v));
    };
    const std::string g = std::to_string(h) + "x" + std::to_string(w);
    const int padN = BN / VSR_GG_KC;
    // rows along W: pixel x of a spatial row / spectrum slot (kx, part) of a spectrum row
    const int tXk = makeTable("LF:xk:" + g, w, Kw, [&](int x) { return (int64_t)x * CS; });
    const int tXm = makeTable("LF:xm:" + g, w, BM, [&](int x) { return (int64_t)x * CS; });
    const int tFWm = makeTable("LF:fwm:" + g, 2 * wf, BM, [&](int m) { return (int64_t)(m / 2) * CG + (m % 2) * CS; });
    const int tFWk = makeTable("LF:fwk:" + g, 2 * wf, Kc, [&](int m) { return (int64_t)(m / 2) * CG + (m % 2) * CS; });
    // rows along H: (y, part) of a spectrum column
    const int tFHm = makeTable("LF:fhm:" + g, 2 * h, BM, [&](int m) { return (int64_t)(m / 2) * wf * CG + (m % 2) * CS; });
    const int tFHk = makeTable("LF:fhk:" + g, 2 * h, Kh, [&](int m) { return (int64_t)(m / 2) * wf * CG + (m % 2) * CS; });
    // columns: (image row y, channel chunk) of the spatial tensor / of the spectrum, (kx, channel chunk) of one spectrum part
    const int tColSpat = makeTable("LF:cs:" + g, h * CH, padN, [&](int j) { return (int64_t)(j / CH) * w * CS + (j % CH) * VSR_GG_KC; });
    const int tColSpecY = makeTable("LF:cy:" + g, h * CH, padN, [&](int j) { return (int64_t)(j / CH) * wf * CG + (j % CH) * VSR_GG_KC; });
    const int tColSpecX = makeTable("LF:cx:" + g, wf * CH, padN, [&](int j) { return (int64_t)(j / CH) * CG + (j % CH) * VSR_GG_KC; });
    const int64_t fw = dft("fw:" + g, 2 * wf, w, 0, w), fh = dft("fh:" + g, 2 * h, 2 * h, 1, h), fhi = dft("fhi:" + g, 2 * h, 2 * h, 2, h),
                  fwi = dft("fwi:" + g, w, 2 * wf, 3, w);
    const int64_t imgSpat = (int64_t)h * w * CS, imgSpec = (int64_t)h * wf * CG;
    auto stage = [&](const char* tag, int M, int N, int K, int Kpad, int64_t aOff, int tRowB, int tColB, int tRowC, int tColC, int bufB,
                     int64_t strideB, int bufC, int64_t strideC, int bufR) {
        Op op;
        op.kind = OP_GEMM;
        op.tag = tag;
        op.bmode = VSR_BMODE_KN;
        op.tileCfg = cfg;
        for (int b = 0; b < B; ++b) {
            GemmItem it{};
            it.M = M; it.N = N; it.K = Kpad;
            it.tilesM = cdiv(M, BM); it.tilesN = cdiv(N, BN);
            it.splitK = 1; it.chunksPerSplit = Kpad / VSR_GG_KC;
            it.alpha = 1.f; it.act = VSR_ACT_NONE;
            it.bufA = BUF_PLAN_CONST; it.offA = aOff;
            it.tRowA = tRowsLinear(M, Kpad, BM);
            it.tColA = tColsLinear(Kpad / VSR_GG_KC, Kpad / VSR_GG_KC);
            it.bufB = bufB; it.offB = b * strideB;
            it.tRowB = tRowB; it.tColB = tColB;
            it.bufC = bufC; it.offC = b * strideC;
            it.tRowC = tRowC; it.tColC = tColC;
            it.offBias = -1;
            it.bufR = bufR; it.offR = bufR >= 0 ? b * strideC : 0; it.tRowR = bufR >= 0 ? tRowC : -1;
            op.gemm.push_back(it);
            op.flops += 2.0 * M * N * (double)K;
        }
        flops += op.flops;
        ops.push_back(std::move(op));
    };
    const Act FA{LB_FA, B, h, wf, CG, 0}, FB{LB_FB, B, h, wf, CG, 0};
    need(LB_FA, FA.elems());
    need(LB_FB, FB.elems());
    need(LB_S2, (int64_t)B * h * w * CS);
    stage("fu.dft_w", 2 * wf, h * CS, w, Kw, fw, tXk, tColSpat, tFWm, tColSpecY, LB_S1, imgSpat, LB_FA, imgSpec, -1);
    stage("fu.dft_h", 2 * h, wf * CS, 2 * h, Kh, fh, tFHk, tColSpecX, tFHm, tColSpecX, LB_FA, imgSpec, LB_FB, imgSpec, -1);
    conv("fu.conv", FB, 0, CG, FA, 0, 1, 1, f.fu, VSR_ACT_RELU, nullptr, 0);
    stage("fu.idft_h", 2 * h, wf * CS, 2 * h, Kh, fhi, tFHk, tColSpecX, tFHm, tColSpecX, LB_FA, imgSpec, LB_FB, imgSpec, -1);
    stage("fu.idft_w", w, h * CS, 2 * wf, Kc, fwi, tFWk, tColSpecY, tXm, tColSpat, LB_FB, imgSpec, LB_S2, imgSpat, LB_S1);
}

// FFC_BN_ACT (ratio 0.75 in and out, 3x3, reflect padding): x, y are 512-channel tensors [local 128 | global 384]
void LamaPlan::ffc(const LamaFfcW& f, const Act& x, const Act& y)
{
    const Act S1{LB_S1, B, h, w, LAMA_CS, 0}, S2{LB_S2, B, h, w, LAMA_CS, 0};
    conv("ffc.local", x, 0, LAMA_C, y, 0, 3, 1, f.outL, VSR_ACT_RELU, nullptr, 0);
    conv("ffc.l2g", x, 0, LAMA_CL, y, LAMA_CL, 3, 1, f.l2g, VSR_ACT_NONE, nullptr, 0);
    conv("ffc.st1", x, LAMA_CL, LAMA_CG, S1, 0, 1, 1, f.st1, VSR_ACT_RELU, nullptr, 0);
    fourier(f);
    conv("ffc.st2", S2, 0, LAMA_CS, y, LAMA_CL, 1, 1, f.st2, VSR_ACT_NONE | VSR_ACT_POST_RELU, &y, LAMA_CL);
}

void LamaPlan::upconv(const char* tag, const Act& in, const Act& out, const ConvW wts[4])
{
    if (in.halo < 1 || in.C % VSR_GG_KC) throw std::runtime_error("transposed conv input needs a zero halo and 32-channel chunks");
    Op op;
    op.kind = OP_GEMM;
    op.tag = tag;
    op.bmode = VSR_BMODE_NK;
    op.tileCfg = pickTile(wts[0].cout);
    int BM, BN;
    tileDims(op.tileCfg, BM, BN);
    const std::vector<int> ids = iota(B);
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            const ConvW& w4 = wts[a * 2 + b];
            std::vector<int> dys = a ? std::vector<int>{1, 0} : std::vector<int>{0}, dxs = b ? std::vector<int>{1, 0} : std::vector<int>{0};
            const int ntaps = (int)(dys.size() * dxs.size());
            if (w4.K != ntaps * in.C) throw std::runtime_error("transposed conv K mismatch");
            std::vector<int32_t> cols;
            for (int c = 0; c < in.C; c += VSR_GG_KC)
                for (int dy : dys)
                    for (int dx : dxs) cols.push_back(fits(((int64_t)dy * in.Wp() + dx) * in.C + c));
            std::vector<int32_t> rows;
            for (int f = 0; f < B; ++f)
                for (int i = 0; i < in.H; ++i)
                    for (int j = 0; j < in.W; ++j) rows.push_back(fits(out.pix(f, 2 * i + a, 2 * j + b)));
            const int32_t first = rows[0];
            while ((int)rows.size() % BM) rows.push_back(first);
            const std::string key = std::to_string(in.buf) + ":" + std::to_string(in.H) + "x" + std::to_string(in.W) + ":" + std::to_string(in.C) + ":" +
                                    std::to_string(out.buf) + ":" + std::to_string(out.halo) + ":" + std::to_string(a) + std::to_string(b) + ":" +
                                    std::to_string(B) + ":" + std::to_string(BM);
            GemmItem it{};
            it.M = B * in.H * in.W; it.N = w4.cout; it.K = w4.K;
            it.tilesM = cdiv(it.M, BM); it.tilesN = cdiv(it.N, BN);
            it.splitK = 1; it.chunksPerSplit = it.K / VSR_GG_KC;
            it.alpha = 1.f; it.act = VSR_ACT_RELU;
            it.bufA = in.buf; it.offA = 0;
            it.tRowA = tRowsAct(in, ids, in.H, in.W, 1, BM, 0);
            it.tColA = table("UPC:" + key, std::move(cols));
            it.bufB = LB_WEIGHTS; it.offB = w4.w;
            it.tRowB = tRowsLinear(it.N, it.K, BN);
            it.tColB = tColsLinear(it.K / VSR_GG_KC, it.K / VSR_GG_KC);
            it.bufC = out.buf; it.offC = 0;
            it.tRowC = table("UPR:" + key, std::move(rows));
            it.tColC = tColsLinear(cdiv(it.N, VSR_GG_KC), it.tilesN * BN / VSR_GG_KC);
            it.offBias = w4.b;
            it.bufR = -1; it.tRowR = -1;
            op.flops += 2.0 * it.M * it.N * (double)it.K;
            op.gemm.push_back(it);
        }
    need(out.buf, out.elems());
    flops += op.flops;
    ops.push_back(std::move(op));
}

LamaPlan::LamaPlan(const LamaModel& model, int B_, int H_, int W_)
    : B(B_), H(H_), W(W_), Hp((int)rup(H_, 8)), Wp((int)rup(W_, 8)), h(Hp / 8), w(Wp / 8), wf(w / 2 + 1), m_(model)
{
    if (!model.packed_ready()) throw std::runtime_error("LaMa model is not packed");
    if (B < 1 || H < 16 || W < 16) throw std::runtime_error("LaMa needs at least one image of 16 x 16 pixels");
    if (Hp - H > H || Wp - W > W) throw std::runtime_error("symmetric padding larger than the image");
    bufElems.assign(LB_COUNT, 0);
    need(LB_WEIGHTS, (int64_t)model.packed.size());
    need(LB_IN_U8, (int64_t)B * H * W * 3);
    need(LB_MASK_U8, (int64_t)B * H * W);
    need(LB_OUT_U8, (int64_t)B * H * W * 3);
    const Act cols{LB_COLS, B, Hp, Wp, 224, 0};
    const Act d0{LB_D0, B, Hp, Wp, 64, 1}, d1{LB_D1, B, Hp / 2, Wp / 2, 128, 1}, d2{LB_D2, B, Hp / 4, Wp / 4, 256, 1};
    const Act xa{LB_XA, B, h, w, LAMA_C, 1}, xb{LB_XB, B, h, w, LAMA_C, 1}, y1{LB_Y1, B, h, w, LAMA_C, 1}, y2{LB_Y2, B, h, w, LAMA_C, 1},
        xt{LB_XT, B, h, w, LAMA_C, 1};
    const Act u1{LB_U1, B, Hp / 4, Wp / 4, 256, 1}, u2{LB_U2, B, Hp / 2, Wp / 2, 128, 1}, u3{LB_U3, B, Hp, Wp, 64, 3};
    {   // pad to x8, normalise, mask, cat, ReflectionPad2d(3), 7x7 window -> rows of the stem GEMM
        Op& o = ew(EW_LAMA_IM2COL7, "stem.im2col");
        o.ibuf[0] = LB_IN_U8; o.ibuf[1] = LB_MASK_U8; o.ibuf[2] = LB_COLS;
        o.ipar[0] = B; o.ipar[1] = H; o.ipar[2] = W; o.ipar[3] = Hp; o.ipar[4] = Wp;
        need(LB_COLS, cols.elems());
    }
    conv("stem", cols, 0, 224, d0, 0, 1, 1, m_.stem, VSR_ACT_RELU, nullptr, 0);
    halo(d0);
    conv("down1", d0, 0, 64, d1, 0, 3, 2, m_.down[0], VSR_ACT_RELU, nullptr, 0);
    halo(d1);
    conv("down2", d1, 0, 128, d2, 0, 3, 2, m_.down[1], VSR_ACT_RELU, nullptr, 0);
    halo(d2);
    conv("down3", d2, 0, 256, xa, 0, 3, 2, m_.down3, VSR_ACT_RELU, nullptr, 0);
    halo(xa);
    Act x = xa, other = xb;
    for (int i = 0; i < m_.nBlocks; ++i) {
        ffc(m_.ffc[2 * i], x, y1);
        halo(y1);
        ffc(m_.ffc[2 * i + 1], y1, y2);
        const bool lastBlock = i == m_.nBlocks - 1;
        const Act& dst = lastBlock ? xt : other;              // the transposed convs read a ZERO halo
        addHalo(x, y2, dst, !lastBlock);
        if (!lastBlock) std::swap(x, other);
    }
    upconv("up1", xt, u1, m_.up[0]);
    upconv("up2", u1, u2, m_.up[1]);
    upconv("up3", u2, u3, m_.up[2]);
    halo(u3);
    {   // 7x7 64 -> 3 over 4 x 4 output blocks (see LamaModel::pack): rows = blocks, K = 10 x 10 window, N = 48 -> pred [blocks][64]
        const int BLK = LAMA_OUT_BLOCK, WIN = BLK + 6, bh = Hp / BLK, bw = Wp / BLK;
        Op op;
        op.kind = OP_GEMM;
        op.tag = "last";
        op.bmode = VSR_BMODE_NK;
        op.tileCfg = VSR_TILE_128x64;
        int BM, BN;
        tileDims(op.tileCfg, BM, BN);
        GemmItem it{};
        it.M = B * bh * bw; it.N = m_.last.cout; it.K = m_.last.K;
        it.tilesM = cdiv(it.M, BM); it.tilesN = cdiv(it.N, BN);
        it.splitK = 1; it.chunksPerSplit = it.K / VSR_GG_KC; it.alpha = 1.f; it.act = VSR_ACT_NONE;
        std::vector<int32_t> rows, cols;
        for (int f = 0; f < B; ++f)
            for (int by = 0; by < bh; ++by)
                for (int bx = 0; bx < bw; ++bx) rows.push_back(fits(u3.pix(f, by * BLK - 3, bx * BLK - 3)));
        const int32_t first = rows[0];
        while ((int)rows.size() % BM) rows.push_back(first);
        for (int c = 0; c < u3.C; c += VSR_GG_KC)
            for (int ty = 0; ty < WIN; ++ty)
                for (int tx = 0; tx < WIN; ++tx) cols.push_back(fits(((int64_t)ty * u3.Wp() + tx) * u3.C + c));
        const std::string key = std::to_string(B) + ":" + std::to_string(Hp) + "x" + std::to_string(Wp);
        it.bufA = u3.buf; it.offA = 0;
        it.tRowA = table("LASTR:" + key, std::move(rows));
        it.tColA = table("LASTC:" + key, std::move(cols));
        it.bufB = LB_WEIGHTS; it.offB = m_.last.w;
        it.tRowB = tRowsLinear(it.N, it.K, BN);
        it.tColB = tColsLinear(it.K / VSR_GG_KC, it.K / VSR_GG_KC);
        it.bufC = LB_PRED; it.offC = 0;
        it.tRowC = tRowsLinear(it.M, LAMA_PRED_LD, BM);
        it.tColC = tColsLinear(LAMA_PRED_LD / VSR_GG_KC, it.tilesN * BN / VSR_GG_KC);
        it.offBias = m_.last.b;
        it.bufR = -1; it.tRowR = -1;
        op.flops = 2.0 * (double)B * Hp * Wp * 3 * (49 * 64);       // algorithmic: the zero taps of the blocked form are not counted
        op.gemm.push_back(it);
        need(LB_PRED, (int64_t)it.M * LAMA_PRED_LD);
        flops += op.flops;
        ops.push_back(std::move(op));
    }
    {
        Op& o = ew(EW_LAMA_OUT, "out.blend");
        o.ibuf[0] = LB_PRED; o.ibuf[1] = LB_IN_U8; o.ibuf[2] = LB_MASK_U8; o.ibuf[3] = LB_OUT_U8;
        o.ipar[0] = B; o.ipar[1] = H; o.ipar[2] = W; o.ipar[3] = Hp; o.ipar[4] = Wp;
    }
}

} // namespace vsr
