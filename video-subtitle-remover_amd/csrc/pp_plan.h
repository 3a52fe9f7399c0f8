// Host-side description of the ProPainter generator (reference backend/inpaint/video/model/propainter.py:
// InpaintGenerator.img_propagation :316-319 and InpaintGenerator.forward :321-378) as flat op lists over symbolic
// buffers and offset tables -- SURVEY.md section 8(a) row a16.  Same IR as the STTN / RAFT / flow-completion plans.
#pragma once
#include "sttn_plan.h"

namespace vsr {

// OP_EW sub-kinds (numbering continues raft_plan.h / rfc_plan.h)
enum PpEwKind {
    EW_PP_MASK_F32 = 30,      // u8 masks -> fp32 {0,1}
    EW_PP_IMGPROP = 31,       // one step of the non-learnable bidirectional image propagation
    EW_PP_COPY = 32,          // (reserved)
    EW_PP_IM2COL3 = 33,       // cat[frames, masks_in, masks_updated] -> im2col of the encoder's first conv
    EW_PP_DS_FLOW = 34,       // flows at 1/4 resolution
    EW_PP_DS_MASK = 35,       // masks at 1/4 resolution into the per-frame mask slots
    EW_PP_FEATPROP_PREP = 36, // consistency mask + feature warp + small condition channels of a propagation step
    EW_PP_DEFORM_COLS = 37,   // flow-guided modulated deformable-conv columns
    EW_PP_LAYERNORM = 38,
    EW_PP_POOL = 39,          // depthwise 4x4 stride-4 pooling tokens
    EW_PP_FOLD = 40,          // overlapping-patch fold (optionally normalised)
    EW_PP_UNFOLD_GELU = 41,
    EW_PP_TANH_OUT = 42
};

enum PpBuf {
    PB_WEIGHTS = 0, PB_IN_FRAMES, PB_IN_MASK_U8, PB_IN_MASK_UPD_U8, PB_IN_FLOW_F, PB_IN_FLOW_B,
    PB_MASK_F, PB_BK, PB_BKM, PB_FW, PB_FWM, PB_OUT_MASK_U8,
    // generator
    PG_IM2COL, PG_E0, PG_E1, PG_E2, PG_ENC, PG_FEAT, PG_DSF_F, PG_DSF_B, PG_PROP, PG_T1, PG_T2, PG_T3, PG_OFF, PG_COLS, PG_BB, PG_FU,
    PG_X, PG_YQ, PG_QKV, PG_S, PG_P, PG_ATT, PG_Y2, PG_F1, PG_FMAP, PG_F2, PG_SC, PG_SCF, PG_DIN, PG_UP0, PG_D0, PG_D1, PG_UP1, PG_D2, PG_D3,
    PG_OUT,
    PG_TOKOUT,                // PP_PLAN_ENCODE: soft-split tokens [n][fh * fw][512] of every frame (the features are PG_FEAT's interiors)
    PB_COUNT
};

// What a PpGenPlan covers.  The encoder and the soft split of a REFERENCE frame are per-frame functions of that frame's inputs
// (propainter.py:333-335 -- the frames are a batch dimension -- and sparse_transformer.py:7-31), and PropainterInpaint's windows
// overlap: a frame is a local frame of two or three windows and a reference frame of up to n / 10 more (propainter_inpaint.py:
// 317-341), so the reference encodes every frame of a 70-frame batch 3.2 times.  PP_PLAN_ENCODE runs the encoder + soft split once
// per frame, PP_PLAN_CACHED is the generator without them: the engine puts the cached features of the local frames into the
// propagation buffer's input slots and the cached tokens of the reference frames into the token buffer (flow_engine.hip).
enum PpPlanMode { PP_PLAN_FULL = 0, PP_PLAN_ENCODE = 1, PP_PLAN_CACHED = 2 };

// InpaintGenerator.img_propagation(masked_frames, (flows_f, flows_b), masks, 'nearest'): t frames of H x W.
// Outputs: PB_FW (propagated frames, planar fp32 [t][3][H][W]) and PB_FWM (updated masks, fp32 [t][H][W]).
class PpImgPropPlan : public PlanBuilder {
public:
    PpImgPropPlan(int t, int H, int W);
    int t, H, W;
};

struct PpBlockW {
    ConvW qkv, proj, fc1, fc2;
    int64_t ln1g, ln1b, ln2g, ln2b, poolW, poolB;    // element offsets into the packed buffer
};

class PpModel {
public:
    PpModel() {}
    bool set_param(const std::string& name, const float* data, const int64_t* shape, int ndim, std::string& err);
    bool pack(std::string& err);
    bool packed_ready() const { return ready_; }
    static std::vector<std::string> expected_keys();
    ConvW enc0, enc2, enc4, enc6, enc8, enc10[2], enc12[4], enc14[8], enc16;
    ConvW off[2][4], deform[2], bb1[2], bb2[2], fuse1, fuse2;      // [0] = backward_1, [1] = forward_1
    ConvW ss, sc, scConv, dec0, dec2, dec4, dec6;
    PpBlockW blk[8];
    std::vector<float> packed;
    int wideN = 512;   // plans put problems of N >= wideN on 128 x 128 tiles (PpGenPlan::wideTile; the engine sets 128 for the fp16-operand arithmetic)
private:
    struct Raw { std::vector<float> v; std::vector<int64_t> shape; };
    std::map<std::string, Raw> raw_;
    bool ready_ = false;
    // general packer: rows [r0, r0+nrows) of weight `key` ([cout][cin][taps]); ciPos[ci] = position of input channel ci in the
    // padded channel axis of length cinPad (a multiple of 32); outPos[n] = row of output n in the padded row axis of length nPad
    bool pack_conv(const std::string& key, ConvW& cw, int cout, int cin, int taps, int r0, int nrows, const std::vector<int>& ciPos, int cinPad,
                   const std::vector<int>& outPos, int nPad, std::string& err);
    bool pack_plain(const std::string& key, ConvW& cw, int cout, int cin, int taps, std::string& err, const std::vector<int>* outPos = nullptr,
                    const std::vector<int>* ciPos = nullptr);   // outPos / ciPos: row / K-position permutations (patch_perm)
    int64_t push_vec(const std::string& key, int n, std::string& err);
};

// InpaintGenerator.forward(masked_frames, completed_flows, masks_in, masks_updated, num_local_frames) in eval mode:
// t frames (the first lt are the local ones) of H x W (multiples of 4 with (H/4, W/4) giving at least one token).
// windowMasked: one flag per attention window (row-major over ceil(fh/5) x ceil(fw/9)): SparseWindowAttention's
// "mask.sum > 0" test (:229-236), computed by the caller from the masks (host side, see flow_engine.hip).
class PpGenPlan : public PlanBuilder {
public:
    // decLo .. decXHi: a promise of the caller that it reads rows [decLo, decHi) and columns [decXLo, decXHi) of the output only
    // (PropainterInpaint blends a window's prediction into its frames where the dilated mask is set, propainter_inpaint.py:350-357):
    // the soft composition's embedding and the decoder's convs then run on the tokens / pixels those depend on.  0, 0 = everything.
    PpGenPlan(const PpModel& model, int t, int lt, int H, int W, const std::vector<uint8_t>& windowMasked, int decLo = 0, int decHi = 0,
              int decXLo = 0, int decXHi = 0, int mode = PP_PLAN_FULL);
    int t, lt, H, W;
    int mode = PP_PLAN_FULL;
    // PP_PLAN_CACHED: local frame k's cached features go to slot k of PG_PROP ([h + 2 propHalo][w + 2 propHalo][128] each, interior),
    // reference frame j's cached tokens to rows [j fh fw, (j + 1) fh fw) of PG_X; PP_PLAN_ENCODE leaves the features in PG_FEAT
    // ([h + 2 featHalo][w + 2 featHalo][128] per frame) and the tokens in PG_TOKOUT
    int propHalo = 1, featHalo = 3;
    int decLo = 0, decHi = 0, decXLo = 0, decXHi = 0;        // as taken (whole image: 0, H / 0, W)
    double refFlops = 0;                                      // the reference's count (flops = what this plan executes)
    int h, w;            // H/4, W/4
    int fh, fw;          // token grid (soft split 7/3/3)
    int gh, gw;          // token grid padded to whole 5x9 windows
    int ph, pw;          // pooled tokens per frame
    static void token_grid(int H, int W, int& fh, int& fw, int& gh, int& gw);
private:
    const PpModel& m_;
    double trimmedFlops_ = 0;
    int pickTile(int N) const;
    int wideTile(int N) const;
    int tColsChunks(const Act& a, int kh, int kw, int dil, const std::vector<int>& chunkCh);
    void gemm(const char* tag, int bufA, int64_t offA, int tRowA, int tColA, int K, int M, int bufC, int64_t offC, int tRowC, int tColC,
              const ConvW& w, int act, int bufR, int64_t offR, int tRowR, int tile, Op* appendTo = nullptr);
    void conv(const char* tag, const Act& in, const std::vector<int>& inIds, const std::vector<int>& chunkCh, int kh, int kw, int stride, int dil,
              const Act& out, const std::vector<int>& outIds, int c0out, const ConvW& w, int act, const Act* res, const std::vector<int>* resIds,
              Op* appendTo = nullptr);
    // the same for a stride-1 conv on the output pixels of rows [ylo, yhi) x columns [xlo, xhi) only (row tables over the rectangle)
    void convRect(const char* tag, const Act& in, const std::vector<int>& ids, const std::vector<int>& chunkCh, const Act& out, const ConvW& w,
                  int act, const Act* res, int ylo, int yhi, int xlo, int xhi);
    void upsample(const Act& in, const Act& out);
    Op& ew(int kind, const char* tag);
    // tq, ty0 .. tx1: queries of the first tq frames and of the windows that hold a token of rows [ty0, ty1) x columns [tx0, tx1) only
    // (the last block under a box promise); the keys are never restricted
    void attention(int blk, const std::vector<uint8_t>& windowMasked, int tq, int ty0, int ty1, int tx0, int tx1);
};

} // namespace vsr
