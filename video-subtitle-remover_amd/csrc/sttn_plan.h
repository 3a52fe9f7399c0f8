// Host-side description of one STTNInpaint.inpaint() call (reference
// backend/inpaint/sttn_auto_inpaint.py:122-164) as a flat list of device ops over
// symbolic buffers and offset tables.  Pure C++ (no HIP): the engine (sttn_engine.hip)
// materialises it on the GPU; tests read it back through the vsr_plan_* C entry points and
// replay it on the CPU against the oracle, which checks every table / descriptor / schedule
// decision without a GPU.
#pragma once
#include <stdint.h>
#include <map>
#include <string>
#include <vector>

namespace vsr {

// ---- network geometry (auto: auto_sttn.py:64-95 ; det: network_sttn.py:64-95) ----
struct Geometry {
    int variant;            // 0 = sttn-auto, 1 = sttn-det
    int modelW, modelH;     // 640x120 (sttn_auto_inpaint.py:39) / 432x240 (sttn_det_inpaint.py)
    int featW, featH;       // model / 4
    int channels;           // 256
    int blocks;             // 8
    int nscales;            // 4
    int patchW[4], patchH[4];
    int neighborStride;     // config.sttnNeighborStride (5)
    int refLength;          // config.sttnReferenceLength (10)
    static Geometry make(int variant);
};

// ---- tuning knobs (defaults = the measured best; env overrides are for A/B runs) ----
struct Tuning {
    int convTile;        // VSR_CONV_TILE: tile config of the N >= 128 convs / QKV GEMM (VSR_TILE_*)
    int qkTile;          // VSR_QK_TILE
    int qkvTile;         // VSR_QKV_TILE: the fused Q/K/V 1x1 GEMM (K = 256: eight chunks per tile)
    int pvTile;          // VSR_PV_TILE
    int pvSplitChunks;   // VSR_PV_SPLIT_CHUNKS: split the P.V contraction into slices of ~this many 32-token
                         //   chunks when it has at least twice as many (0 = never split)
    int convChannelMajor; // VSR_CONV_KORDER: 1 = K ordered (channel-chunk, tap), 0 = (tap, channel-chunk)
    int outConvBlocked;  // VSR_OUT_CONV_BLOCKED: 1 = the 64 -> 3 output conv runs over 2x4 output blocks (Model::pack_conv_blocked)
    int shareQkv0;       // VSR_QKV0_SHARED: 1 = the first block's q/k/v once per frame of the chunk (BUF_QKV0) instead of once per window;
                         // built and replayed on the CPU in round 4, bit-equal frames on the GPU in round 5: the Python side exports 1 (switches.py)
    int trimLastBlock;   // VSR_TRIM_LAST_BLOCK: 1 = the last transformer block of a window computes its attention output, out-conv and
                         //   FFN for the NEIGHBOUR frames only -- the decoder reads nothing else (Plan::buildWindow); same bits
    int fuseSoftmax;     // VSR_FUSE_SOFTMAX: 1 = exact-fp32 mode keeps no probability matrix for the scales whose scores are not
                         //   split along K: row max in the QK^T epilogue, exp + row sum while P.V stages its A tiles (0 = k_softmax_rows)
    // precision 0: exact fp32 MFMA kernels; 1: split-half f16 MFMA kernels (larger tiles pay there)
    static const Tuning& get(int precision = 0);
};

// ---- packed weights ----
struct ConvW {
    int64_t w = -1, b = -1; // element offsets into the packed weight buffer
    int cout = 0, K = 0;    // packed as [cout][K], K = taps*cin (k = tap*cin + ci)
};
struct BlockW { ConvW qkv, out, ffn1, ffn2; };

class Model {
public:
    explicit Model(int variant);
    // name = reference state_dict key, data = fp32 contiguous, shape as in the checkpoint
    bool set_param(const std::string& name, const float* data, const int64_t* shape, int ndim, std::string& err);
    bool pack(std::string& err);             // all 112 tensors present -> packed buffer
    bool packed_ready() const { return ready_; }
    Geometry g;
    ConvW enc[4], dec[4];
    ConvW dec4blk;          // the 64 -> 3 conv as a GEMM over 2x4 output blocks (pack_conv_blocked): cout = 24 = (dy, dx, c), K = 4x6x64
    static constexpr int kOutBlkH = 2, kOutBlkW = 4;
    std::vector<BlockW> blk;
    std::vector<float> packed;
    static std::vector<std::string> expected_keys(int variant);
private:
    struct Raw { std::vector<float> v; std::vector<int64_t> shape; };
    std::map<std::string, Raw> raw_;
    bool ready_ = false;
    bool pack_conv(const std::string& key, ConvW& cw, int cinPad, std::string& err);
    bool pack_conv_blocked(const std::string& key, ConvW& cw, int bh, int bw, std::string& err);
};

// ---- plan IR ----
enum BufId {
    BUF_WEIGHTS = 0, BUF_IN_U8, BUF_IM2COL, BUF_E1, BUF_E2, BUF_E3, BUF_FEATS, BUF_X0, BUF_X1, BUF_QKV,
    BUF_S, BUF_P, BUF_ATT, BUF_F1, BUF_UP1, BUF_D1, BUF_D2, BUF_UP2, BUF_D3, BUF_D4, BUF_COMP, BUF_PVPART, BUF_MASK_U8,
    BUF_ROWMAX,   // fused attention: row maxima of the scores (uint32 images of floats, one array per attention instance of the plan; zeroed per run)
    BUF_LSUM,     // fused attention: partial row sums of the exponentials [split][rows] of a split P.V
    // further instances of every buffer a sliding window works in: the windows of a chunk are independent until their decoded
    // frames are averaged into BUF_COMP, so window w runs on stream ("lane") w % lanes in lane-owned instances (Plan::lanes, laneBuf())
    BUF_LANE_FIRST,
    BUF_LANE_END = BUF_LANE_FIRST + 15 * 3,  // kLaneBufs * (kMaxLanes - 1)
    // Tuning::shareQkv0: the FIRST transformer block's q/k/v of every frame of the chunk [L * featH * featW][3C].  That block reads the
    // encoder features, which the windows share, and its q/k/v projection is a 1x1 conv -- a per-frame function -- so the reference
    // computes it once per window a frame appears in (about three times per frame); here once, read by every window and lane
    BUF_QKV0 = BUF_LANE_END,
    BUF_COUNT
};
constexpr int kLaneBufs = 15, kMaxLanes = 4;
// the window-scoped buffers, in the order of their lane instances
constexpr int kLaneBufList[kLaneBufs] = {BUF_X0, BUF_X1, BUF_QKV, BUF_S, BUF_P, BUF_ATT, BUF_F1, BUF_UP1, BUF_D1, BUF_D2, BUF_UP2,
                                         BUF_D3, BUF_D4, BUF_PVPART, BUF_LSUM};
// lane instance of a window-scoped buffer (every other buffer is shared: weights, encoder stages, features, comp, masks, row maxima)
inline int laneBuf(int buf, int lane)
{
    if (lane == 0) return buf;
    for (int i = 0; i < kLaneBufs; ++i)
        if (kLaneBufList[i] == buf) return BUF_LANE_FIRST + (lane - 1) * kLaneBufs + i;
    return buf;
}
inline int baseBuf(int buf) { return buf >= BUF_LANE_FIRST && buf < BUF_LANE_END ? kLaneBufList[(buf - BUF_LANE_FIRST) % kLaneBufs] : buf; }   // the lane-0 buffer an instance mirrors

enum OpKind { OP_NORM_IM2COL = 0, OP_GEMM = 1, OP_SOFTMAX = 2, OP_UPSAMPLE2X = 3, OP_DECODE_OUT = 4, OP_REDUCE_SCATTER = 5,
              OP_EW = 6 /* RAFT's elementwise / gather kernels, sub-kind in Op::ew (raft_plan.h) */ };

struct GemmItem {
    int bufA, bufB, bufC, bufR;          // bufR = -1: no residual
    int64_t offA, offB, offC, offR, offBias; // offBias (into BUF_WEIGHTS) = -1: none
    int tRowA, tColA, tRowB, tColB, tRowC, tColC, tRowR; // table ids (tRowR = -1: none)
    int M, N, K;
    int tilesM, tilesN, splitK, chunksPerSplit;
    int64_t splitStride;
    float alpha;
    int act;
    int bufBias;                         // buffer `offBias` indexes: 0 = BUF_WEIGHTS (the zero-initialised default); VSR_ACT_A_EXP: the row maxima
};
struct SoftmaxItem {
    int bufS, bufP;
    int64_t offS, offP, splitStride;
    int M, N, ldS, ldP, nsplit;
    float scale;
};
struct Op {
    int kind = 0;
    int tileCfg = 0, bmode = 0;          // GEMM
    std::vector<GemmItem> gemm;          // GEMM group
    std::vector<SoftmaxItem> softmax;    // SOFTMAX group
    // UPSAMPLE2X: src/dst buffers, H, W, C, halos, n ; DECODE_OUT: src, ldy, pix, n, tables
    int bufSrc = -1, bufDst = -1, H = 0, W = 0, C = 0, haloS = 0, haloD = 0, n = 0;
    int ldy = 0, pix = 0, tFrameIdx = -1, tFirst = -1;
    int premask = 0;
    int bufMask = -1;                    // sttn-det: model-res resized mask [L][mh][mw] u8 (BUF_MASK_U8)
    // REDUCE_SCATTER: out[bufDst+offDst][rowC[m]+colC[n/32]+n%32] = sum_s part[bufSrc+offSrc][s*splitStride + m*N + n]
    //   (ibuf[0] >= 0: divided by sum_s lsum[ibuf[0] + ioff[0]][s*ipar[0] + m], the row sums a VSR_ACT_A_EXP product left)
    int M = 0, N = 0, nsplit = 0, tRowC = -1, tColC = -1;
    int64_t offSrc = 0, offDst = 0, splitStride = 0;
    double flops = 0;                    // algorithmic flops of this op (2*M*N*K, unpadded)
    int lane = 0;                        // STTN: the stream this op is issued on (0 = the caller's; ops in list order are a valid serial schedule)
    std::string tag;
    // OP_EW: sub-kind + generic operands (buffers, element offsets, integer / float parameters; meaning per sub-kind)
    int ew = 0;
    int ibuf[4] = {-1, -1, -1, -1};
    int64_t ioff[4] = {0, 0, 0, 0};
    int ipar[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float fpar[4] = {0.f, 0.f, 0.f, 0.f};
};

struct Act {                              // NHWC activation with a physical zero halo
    int buf, n, H, W, C, halo;
    int Hp() const { return H + 2 * halo; }
    int Wp() const { return W + 2 * halo; }
    int64_t frameElems() const { return (int64_t)Hp() * Wp() * C; }
    int64_t pix(int f, int y, int x) const { return (((int64_t)f * Hp() + y + halo) * Wp() + x + halo) * C; }
    int64_t elems() const { return (int64_t)n * frameElems(); }
};

// GemmItem::bufA / bufB may name this pseudo buffer: PlanIR::consts, fp32 constants that belong to the plan (not to the
// checkpoint), uploaded with the offset tables -- the DFT matrices of the LaMa plan
constexpr int BUF_PLAN_CONST = -2;

// What the engines materialise and what tests replay: symbolic buffers, offset tables, op list.
struct PlanIR {
    std::vector<int64_t> bufElems;        // one entry per buffer id (u8 buffers in bytes, others floats)
    std::vector<float> consts;            // BUF_PLAN_CONST
    std::vector<std::vector<int32_t>> tables;
    std::vector<Op> ops;
    std::vector<int32_t> compCount;       // STTN: decodes per frame (1 => comp stays u8)
    double flops = 0;                     // algorithmic model flops of the whole call (what the plan's GEMMs contract)
    double refFlops = 0;                  // STTN: flops of the call as the reference computes it -- `flops` + the rows of the last block
                                          // that nothing reads (Plan::buildWindow); equal to `flops` for every other plan
    virtual ~PlanIR() {}
};

// Table / conv primitives shared by the STTN and RAFT plan builders.
class PlanBuilder : public PlanIR {
protected:
    std::map<std::string, int> tableKey_;
    int table(const std::string& key, std::vector<int32_t>&& v);
    // element offset of pixel (ids[i], y*stride, x*stride) of `a` (+ add) for every output pixel, padded to padTo rows
    int tRowsAct(const Act& a, const std::vector<int>& ids, int oh, int ow, int stride, int padTo, int64_t add);
    // pixels (y, x) of a rectangle [ylo, yhi) x [xlo, xhi) of every listed frame, row-major, padded to padTo rows
    int tRowsActRect(const Act& a, const std::vector<int>& ids, int ylo, int yhi, int xlo, int xhi, int padTo);
    // 32-channel chunk offsets of a kh x kw window (dilation dil) over channels [c0, c0+cin) of `a`; K order mirrors pack
    int tColsConvHW(const Act& a, int kh, int kw, int dil, int c0 = 0, int cin = -1);
    int tColsConv(const Act& a, int ksz, int dil) { return tColsConvHW(a, ksz, ksz, dil); }
    int tRowsLinear(int count, int ld, int padTo);
    int tColsLinear(int nchunks, int padTo);
    void need(int buf, int64_t elems);
};

class Plan : public PlanBuilder {
public:
    // decLo / decHi: rows [decLo, decHi) of the model-resolution output are all the caller will read (sttn-auto blends the strip back
    // only where the mask is set: vsr_sttn_auto_chunk) -- the decoder computes those rows and what they depend on, nothing else
    // (buildWindow); decHi <= decLo = the whole image
    // decXLo / decXHi: the same for columns (the GEMMs take rectangles; the two elementwise kernels of the decoder keep whole rows)
    Plan(const Model& model, int L, int precision = 0, int lanes = 1, int decLo = 0, int decHi = 0, int decXLo = 0, int decXHi = 0);
    // the bounds a plan with these arguments decodes: clipped to the image and widened to whole blocks of the output conv (all 0 = the
    // whole image: no promise, or the per-pixel form of the output conv).  Needs no plan: vsr_sttn_decode_rows asks once per area.
    static void decoder_bounds(const Geometry& g, int precision, int decLo, int decHi, int decXLo, int decXHi, int* lo, int* hi, int* xlo, int* xhi);
    int decLo = 0, decHi = 0;            // as given, clipped to the image and widened to whole 2-row blocks of the output conv
    int decXLo = 0, decXHi = 0;          // ... to whole 4-column blocks
    int L;
    int precision;
    int lanes;                           // 1 .. kMaxLanes: window w runs on lane w % lanes
    int firstWindowOp = -1;              // index of the first op that is not the encoder's: lane 1 may start once everything before it is done
    Geometry g;
    int nwindows = 0;
private:
    const Model& m_;
    const Tuning& tu_;
    int lane_ = 0;                       // lane of the window being built
    bool qkv0_ = false;                  // the first block's q/k/v live in BUF_QKV0 (Tuning::shareQkv0)
    int lb(int buf) const { return laneBuf(buf, lane_); }
    int64_t rowmaxElems_ = 0;            // BUF_ROWMAX handed out so far: every fused attention instance of the plan has its own array
    double trimmedFlops_ = 0;            // what the reference spends on last-block rows nobody reads (buildWindow)
    int pickTile(int N) const;
    // oy0 / oy1: patch rows [oy0, oy1) of every frame only (-1 = all): the query tokens of a last block that feeds a ranged decoder
    // fids: the frame of the q/k/v buffer that token frame t lives in (BUF_QKV0: the window's chunk frame ids); nullptr: t itself
    int tRowsTokens(int T, int s, int choff, int count, int padTo, int oy0 = 0, int oy1 = -1, int ox0 = 0, int ox1 = -1,
                    const std::vector<int>* fids = nullptr);
    int tColsPatch(int s, int padTo);
    int tRowsTokensAct(const Act& a, int T, int s, int padTo, int oy0 = 0, int oy1 = -1, int ox0 = 0, int ox1 = -1);
    int tColsPatchAct(const Act& a, int s, int padTo);
    void addConv(const char* tag, const Act& in, const std::vector<int>& inIds, const Act& out, int nOut,
                 int ksz, int stride, int dil, const ConvW& w, int act, const Act* res,
                 const std::vector<int>* resIds, int ylo = 0, int yhi = -1,      // [ylo, yhi): output rows computed (stride 1; default all)
                 int xlo = 0, int xhi = -1);                                       // [xlo, xhi): output columns computed (default all)
    // [attLo, attHi) x [attXLo, attXHi): feature rows / columns of the output that are read
    void addAttention(int Tq, int T, const BlockW& bw, int attLo = 0, int attHi = -1, int attXLo = 0, int attXHi = -1, int qkvBuf = -1,
                      const std::vector<int>* fids = nullptr);
    void buildWindow(const std::vector<int>& neighbors, const std::vector<int>& refs,
                     std::vector<int32_t>& visits);
};

// tile of the N <= 64 convs of every plan (VSR_N64_TILE, A/B knob): 128x64 -- the 256x64 instance of the rebuilt v3 kernel needs 385
// VGPRs, one wave per SIMD (profiles/r03_n64_tile_ab.log)
int n64Tile();

// cv2.resize INTER_LINEAR coefficient tables (OpenCV 4.11 imgproc/resize.cpp, resize()):
// ofs[d], fixed-point (x2048, round-half-even) and float taps {1-f, f}.  clampX selects the
// horizontal rule (index clamped, f reset to 0 at the borders); vertical keeps f and lets
// the row index be clipped at use.
void cv2_linear_tables(int ssize, int dsize, bool clampX, std::vector<int32_t>& ofs,
                       std::vector<int16_t>& icoef, std::vector<float>& fcoef);

} // namespace vsr
