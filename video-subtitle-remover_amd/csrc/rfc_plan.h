// Host-side description of the recurrent flow completion network (reference
// backend/inpaint/video/model/recurrent_flow_completion.py: RecurrentFlowCompleteNet.forward_bidirect_flow :313-339 +
// combine_flow :341-348) as a flat op list -- SURVEY.md section 8(a) row a15.  Same IR as the STTN and RAFT plans.
//
// The forward and the (time-reversed) backward completion are two independent sequences of T = t-1 flow fields; they
// run as a batch of S = 2 through every op.  Convs (2-D, the (1,k,k) and (3,1,1) 3-D ones, dilated, strided) and the
// deformable conv (after a gather kernel has produced its modulated, bilinearly sampled columns) are gather-GEMMs;
// channel concatenations and the frame selection of the recurrent propagation are offset tables over one buffer.
#pragma once
#include "sttn_plan.h"

namespace vsr {

// OP_EW sub-kinds continue raft_plan.h's numbering
enum RfcEwKind {
    EW_RFC_IM2COL5 = 20,  // masked flows + masks -> im2col of the 5x5 stride-2 replicate-padded stem conv
    EW_DEFORM_COLS = 21,  // offsets / masks -> modulated deformable-conv columns
    EW_RFC_COMBINE = 22   // predicted flows -> pred * mask + flow * (1 - mask), planar output
};

enum RfcBuf {
    FB_WEIGHTS = 0, FB_IN_FLOW_F, FB_IN_FLOW_B, FB_IN_MASK, FB_IM2COL, FB_X0, FB_A, FB_B, FB_C, FB_E1, FB_D, FB_E, FB_F, FB_E2, FB_M1, FB_M2,
    FB_PROP, FB_T1, FB_T2, FB_T3, FB_OFF, FB_COLS, FB_BB, FB_FUSED, FB_D2A, FB_UP2, FB_D2, FB_D1A, FB_UP1, FB_D1, FB_U0, FB_UP0, FB_PRED,
    FB_OUT_F, FB_OUT_B, FB_COUNT
};

class RfcModel {
public:
    RfcModel() {}
    bool set_param(const std::string& name, const float* data, const int64_t* shape, int ndim, std::string& err);
    bool pack(std::string& err);
    bool packed_ready() const { return ready_; }
    static std::vector<std::string> expected_keys();
    ConvW down, p1[4], p2[4], mid[3];
    ConvW off[2][4], deform[2], bb1[2], bb2[2], fusion;    // [0] = backward_, [1] = forward_
    ConvW dec2a, dec2b, dec1a, dec1b, up0, up1;
    std::vector<float> packed;
private:
    struct Raw { std::vector<float> v; std::vector<int64_t> shape; };
    std::map<std::string, Raw> raw_;
    bool ready_ = false;
    // weight [cout][cin][taps...] (2-D or 3-D kernel, taps = product of the kernel dims) -> [cout][K]
    bool pack_conv(const std::string& key, ConvW& cw, int cout, int cin, int taps, std::string& err);
};

class RfcPlan : public PlanBuilder {
public:
    // t frames (t-1 flow fields per direction) of H x W (multiples of 8)
    RfcPlan(const RfcModel& model, int t, int H, int W);
    int t, T, H, W;
    static const int S = 2;
private:
    const RfcModel& m_;
    int pickTile(int N) const;
    std::vector<int> seqIds(bool temporalHalo) const;   // frame ids (i-major, sequence-minor) inside a [S][T(+4)] buffer
    // generic gather-GEMM: A rows/cols and C rows given as tables built by the caller
    void gemm(const char* tag, int bufA, int64_t offA, int tRowA, int tColA, int K, int M, int bufC, int64_t offC, int tRowC, const ConvW& w,
              int act, int bufR, int64_t offR, int tRowR, int tile, bool append = false);
    void conv(const char* tag, const Act& in, const std::vector<int>& inIds, const Act& out, const std::vector<int>& outIds, int kh, int kw,
              int stride, int dil, const ConvW& w, int act, const Act* res, const std::vector<int>* resIds);
    void tconv(const char* tag, const Act& in, const std::vector<int>& ids, const Act& out, const ConvW& w, int act);   // (3,1,1) dilation 2
    void upsample(const Act& in, const Act& out);
};

} // namespace vsr
