// Host-side description of RAFT_bi.forward (reference backend/inpaint/video/model/modules/flow_comp_raft.py:39-55,
// network backend/inpaint/video/raft/{raft,extractor,corr,update}.py) as a flat op list over symbolic buffers and
// offset tables -- SURVEY.md section 8(a) row a14.  Same IR as the STTN plan (sttn_plan.h): every conv and the
// all-pairs correlation are gather-GEMM problems; normalisation, the correlation pyramid / lookup, the GRU gate
// arithmetic, the flow bookkeeping and the convex upsampling are OP_EW ops (flow_kernels.hip).  Pure C++: the
// engine (flow_engine.hip) materialises it, tests replay it on the CPU against oracle/raft.py.
#pragma once
#include "sttn_plan.h"

namespace vsr {

// ---- OP_EW sub-kinds (Op::ew); operands in Op::ibuf / ioff / ipar / fpar, documented at the kernels ----
enum EwKind {
    EW_IM2COL7_U8 = 1,   // u8 frames -> normalised im2col of the 7x7 stride-2 stem conv
    EW_INORM_STATS = 2,  // InstanceNorm2d statistics per (frame, channel)
    EW_INORM_APPLY = 3,  // normalise (+ ReLU) (+ residual + ReLU) in place
    EW_CTX_SPLIT = 4,    // context features -> tanh(net) | relu(inp) of every pair-direction
    EW_FLOW_UPDATE = 5,  // coords1 (+)= delta ; flow = coords1 - coords0
    EW_IM2COL7_FLOW = 6, // im2col of the 7x7 conv over the 2-channel flow
    EW_AVGPOOL2 = 7,     // next level of the correlation pyramid
    EW_CORR_LOOKUP = 8,  // 4 levels x 9x9 bilinear samples around coords1
    EW_GRU_RH = 9,       // r * h
    EW_GRU_UPDATE = 10,  // h = (1-z) h + z q
    EW_CONVEX_UP = 11,   // convex 8x upsampling of the final flow
    EW_CORR_TRANSPOSE = 12 // correlation volumes of the backward pair-directions = transposes of the forward ones
};

enum RaftBuf {
    RB_WEIGHTS = 0, RB_IN_U8, RB_IM2COL, RB_S1A, RB_S1B, RB_S1C, RB_S2A, RB_S2B, RB_S2C, RB_S3A, RB_S3B, RB_S3C,
    RB_STATS, RB_FMAP, RB_CMAP, RB_PYR, RB_COORDS, RB_FLOW, RB_CORRF, RB_C1, RB_CORFLO, RB_FLOWCOL, RB_F1, RB_HXR,
    RB_ZR, RB_Q, RB_FH1, RB_DELTA, RB_MASKH, RB_MASK, RB_OUT,
    RB_ZRCTX0, RB_ZRCTX1, RB_QCTX0, RB_QCTX1,   // the context's share of the GRU convs, computed once per call (RaftPlan: ctxHoist)
    RB_COUNT
};

struct RaftEncW {                 // BasicEncoder (extractor.py:118-160)
    ConvW conv1;                  // 7x7 s2, 3 -> 64
    ConvW b1[3][2], b2[3][2];     // layer{1,2,3}.{0,1}.conv{1,2}
    ConvW ds[3];                  // layer{2,3}.0.downsample.0 (index 1, 2)
    ConvW conv2;                  // 1x1, 128 -> 256
};

class RaftModel {
public:
    RaftModel();
    bool set_param(const std::string& name, const float* data, const int64_t* shape, int ndim, std::string& err);
    bool pack(std::string& err);
    bool packed_ready() const { return ready_; }
    static std::vector<std::string> expected_keys();
    RaftEncW fnet, cnet;
    ConvW convc1, convc2, convf1, convf2, conv, zr[2], q[2], fh1, fh2, mask1, mask2;
    // the GRU convs cut along their input channels (cat[h | inp | motion], update.py:47-58): the `inp` third -- the context features, the
    // same in every iteration (raft.py:114-116,127) -- with the bias, and the rest (h or r*h, motion) without one.  conv(cat[a, b]) =
    // conv_a(a) + conv_b(b): the plan computes the context's share once per call and adds it in the epilogue of the per-iteration conv
    ConvW zrCtx[2], qCtx[2], zrVar[2], qVar[2];
    std::vector<float> packed;
private:
    struct Raw { std::vector<float> v; std::vector<int64_t> shape; };
    std::map<std::string, Raw> raw_;
    bool ready_ = false;
    // [cout][cin][kh][kw] -> [cout][K]; bn = prefix of a BatchNorm2d folded into the conv (eval mode), scale multiplies
    // weights and bias (the mask head's 0.25)
    bool pack_conv(const std::string& key, ConvW& cw, int cout, int cin, int kh, int kw, const std::string& bn, float scale,
                   std::string& err);
    bool pack_encoder(const std::string& prefix, RaftEncW& e, bool batchNorm, std::string& err);
    bool fuse_rows(const ConvW& a, const ConvW& b, ConvW& out);
    // input channels [c0, c0 + n) (and [c1, c1 + n1) behind them) of a packed conv with `cin` input channels as a conv of their own
    bool slice_cin(const ConvW& src, int cin, int taps, int c0, int n, int c1, int n1, bool keepBias, ConvW& out);
};

class RaftPlan : public PlanBuilder {
public:
    // t frames of H x W (multiples of 8, H/8 and W/8 >= 16 so that the coarsest pyramid level is at least 2x2),
    // `iters` GRU iterations: flows of the t-1 consecutive pairs in both directions
    RaftPlan(const RaftModel& model, int t, int H, int W, int iters);
    int t, H, W, iters;
    int pairs;                            // 2 (t-1) pair-directions: p < t-1 forward (p -> p+1), else backward
    int h8, w8;
    int lvlH[4], lvlW[4];
    int64_t lvlOff[4];                    // element offsets of the pyramid levels inside RB_PYR
private:
    const RaftModel& m_;
    int pickTile(int N) const;
    // conv as gather-GEMM: channels [c0in, c0in+cin) of `in` (or the chunk list `chunks`) -> channels [c0out, ..) of `out`
    void conv(const char* tag, const Act& in, const std::vector<int>& inIds, int c0in, int cin, const Act& out, int c0out, int nOut,
              int kh, int kw, int stride, const ConvW& w, int act, const Act* res, const std::vector<int>* chunks = nullptr);
    void linear(const char* tag, int bufA, int M, int K, int bufC, int ldC, const ConvW& w, int act);   // plain [M][K] x W -> [M][ldC]
    void inorm(const Act& x, bool relu, const Act* res);
    void encoder(const RaftEncW& e, bool instanceNorm, int outBuf);
    Op& ew(int kind, const char* tag);
};

} // namespace vsr
