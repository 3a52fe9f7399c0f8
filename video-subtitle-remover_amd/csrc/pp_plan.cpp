// Plan builders for the ProPainter generator (see pp_plan.h).
#include "pp_plan.h"
#include "gather_gemm.h"
#include <stdexcept>

namespace vsr {

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

} // namespace vsr
