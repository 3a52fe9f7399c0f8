// Host-side description of the ProPainter generator (reference backend/inpaint/video/model/propainter.py:
// InpaintGenerator.img_propagation :316-319 and InpaintGenerator.forward :321-378) as flat op lists over symbolic
// buffers and offset tables -- SURVEY.md section 8(a) row a16.  Same IR as the STTN / RAFT / flow-completion plans.
#pragma once
#include "sttn_plan.h"

namespace vsr {

// OP_EW sub-kinds (numbering continues raft_plan.h / rfc_plan.h)
enum PpEwKind {
    EW_PP_MASK_F32 = 30,   // u8 masks -> fp32 {0,1}
    EW_PP_IMGPROP = 31,    // one step of the non-learnable bidirectional image propagation
    EW_PP_COPY = 32        // plain copy (buffer, offset, count)
};

enum PpBuf {
    PB_WEIGHTS = 0, PB_IN_FRAMES, PB_IN_MASK_U8, PB_IN_MASK_UPD_U8, PB_IN_FLOW_F, PB_IN_FLOW_B,
    PB_MASK_F, PB_BK, PB_BKM, PB_FW, PB_FWM, PB_OUT_MASK_U8,
    PB_COUNT
};

// InpaintGenerator.img_propagation(masked_frames, (flows_f, flows_b), masks, 'nearest'): t frames of H x W.
// Outputs: PB_FW (propagated frames, planar fp32 [t][3][H][W]) and PB_FWM (updated masks, fp32 [t][H][W]).
class PpImgPropPlan : public PlanBuilder {
public:
    PpImgPropPlan(int t, int H, int W);
    int t, H, W;
};

} // namespace vsr
