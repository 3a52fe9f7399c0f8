// Host-side description of one big-LaMa generator call -- SURVEY.md 8(a) row a12.
//
// The reference only holds the network as a TorchScript blob (backend/inpaint/lama_inpaint.py:13 torch.jit.load('big-lama.pt');
// a missing blob, no source in the tree).  What is built here is the published generator the blob was exported from
// (advimman/lama, saicinpainting/training/modules/ffc.py FFCResNetGenerator + configs/training/generator/ffc_resnet_075.yaml:
// input 4 ch, ngf 64, 3 stride-2 FFC convs, n FFC residual blocks at 512 ch with 75 % "global" channels, no LFU, 3 transposed
// convs, 7x7 conv, sigmoid) wrapped as the exported module's forward(image, mask) does
//     masked = image * (1 - mask);  pred = generator(cat[masked, mask]);  out = mask * pred + (1 - mask) * image
// (PARITY UNPINNED: key names / shapes follow the published module tree; checked against oracle/lama.py BigLamaNet.)
//
// Same IR as the other engines (sttn_plan.h).  Every contraction is a gather-GEMM problem:
//   * 3x3 / 7x7 / 1x1 convolutions with eval-mode BatchNorm folded in; nn.Conv2d(padding_mode='reflect') reads a PHYSICAL
//     reflect halo that a border kernel fills after the producer (EW_LAMA_HALO);
//   * an FFC block keeps local (128) and global (384) channels in ONE NHWC tensor of 512 channels, so
//     convl2l(x_l) + convg2l(x_g) is a single 3x3 conv over 512 channels and ConcatTupleLayer is free;
//   * FourierUnit (rfftn -> 1x1 conv on stacked re/im -> irfftn, norm='ortho') as four DFT-matrix GEMMs (KN mode: the
//     activation rows are the contraction index): real DFT along W, complex DFT along H, [1x1 conv + BN + ReLU], inverse
//     complex DFT along H, complex-to-real inverse DFT along W with the `x + fu(x)` residual in its epilogue.  The feature
//     maps are (H/8) x (W/8) = 45 x 240 at the 1080p strip: neither a power of two, 6 % of the block's FLOPs as matrices;
//     the matrices are plan constants (PlanIR::consts), generated in double precision;
//   * ConvTranspose2d(3, stride 2, padding 1, output_padding 1) as four phase problems (1 / 2 / 2 / 4 taps) over a zero halo.
#pragma once
#include "sttn_plan.h"

namespace vsr {

enum LamaEw {
    EW_LAMA_IM2COL7 = 50,   // u8 image + mask -> pad to x8 (symmetric), /255, *(1-mask), cat mask, reflect pad 3, 7x7 im2col
    EW_LAMA_HALO = 51,      // fill the reflect halo of an NHWC activation from its interior (border pixels only)
    EW_LAMA_ADD_HALO = 52,  // dst = a + b on the interior, then dst's reflect halo (or none: zero halo kept)
    EW_LAMA_OUT = 53        // sigmoid (logits in 4x4-block layout), mask blend with the image, clip(.*255) -> u8 truncation, crop
};

enum LamaBuf {
    LB_WEIGHTS = 0, LB_IN_U8, LB_MASK_U8, LB_COLS, LB_D0, LB_D1, LB_D2, LB_XA, LB_XB, LB_Y1, LB_Y2, LB_XT, LB_S1, LB_S2, LB_FA, LB_FB,
    LB_U1, LB_U2, LB_U3, LB_PRED, LB_OUT_U8, LB_COUNT
};

constexpr int LAMA_CL = 128, LAMA_CG = 384, LAMA_C = 512, LAMA_CS = 192;   // local / global / total / spectral channels
constexpr int LAMA_OUT_BLOCK = 4, LAMA_PRED_LD = 64;   // the last conv writes logits as [4x4 pixel blocks][(dy, dx, c) padded to 64]

struct LamaFfcW {            // one FFC_BN_ACT of a residual block (ratio_gin = ratio_gout = 0.75)
    ConvW outL;              // [convl2l | convg2l]: 3x3, 512 -> 128, bn_l folded
    ConvW l2g;               // convl2g: 3x3, 128 -> 384, bn_g folded (scale + shift)
    ConvW st1;               // convg2g.conv1: 1x1, 384 -> 192, its BatchNorm folded
    ConvW fu;                // convg2g.fu.conv_layer: 1x1 over (part, c) <- (2c + part) channels, 384 -> 384, fu.bn folded
    ConvW st2;               // convg2g.conv2: 1x1, 192 -> 384, bn_g scale only (the shift rides on l2g)
};

class LamaModel {
public:
    bool set_param(const std::string& name, const float* data, const int64_t* shape, int ndim, std::string& err);
    bool pack(std::string& err);          // infers the number of residual blocks from the keys
    bool packed_ready() const { return ready_; }
    int nBlocks = 0;
    ConvW stem;                           // model.1: 7x7, 4 -> 64 (k = tap*4 + c, K 196 -> 224)
    ConvW down[2];                        // model.2 / model.3: 3x3 stride 2
    ConvW down3;                          // model.4: [convl2l ; convl2g] 3x3 stride 2, 256 -> 512
    std::vector<LamaFfcW> ffc;            // 2 per residual block
    ConvW up[3][4];                       // transposed convs, one packed matrix per output phase (a, b) = (y & 1, x & 1)
    ConvW last;                           // 7x7, 64 -> 3 (+ bias) packed for 4 x 4 output blocks: [48][10 * 10 * 64]
    std::vector<float> packed;
private:
    struct Raw { std::vector<float> v; std::vector<int64_t> shape; };
    std::map<std::string, Raw> raw_;
    bool ready_ = false;
    const Raw* get(const std::string& key, std::string& err) const;
    bool bn_affine(const std::string& bn, int c, std::vector<float>& scale, std::vector<float>& shift, std::string& err) const;
    bool pack_ffc(const std::string& p, LamaFfcW& f, std::string& err);
    bool pack_up(const std::string& key, const std::string& bn, int cin, int cout, ConvW out[4], std::string& err);
};

class LamaPlan : public PlanBuilder {
public:
    // B images of H x W (any size >= 16): padded to multiples of 8 exactly like lama_util.pad_img_to_modulo
    LamaPlan(const LamaModel& model, int B, int H, int W);
    int B, H, W, Hp, Wp, h, w, wf;
private:
    const LamaModel& m_;
    int pickTile(int N) const;
    Op& ew(int kind, const char* tag);
    void conv(const char* tag, const Act& in, int c0in, int cin, const Act& out, int c0out, int k, int stride, const ConvW& w, int act,
              const Act* res, int c0res);
    void halo(const Act& a);
    void addHalo(const Act& a, const Act& b, const Act& dst, bool reflect);
    void ffc(const LamaFfcW& f, const Act& x, const Act& y);
    void fourier(const LamaFfcW& f);
    void upconv(const char* tag, const Act& in, const Act& out, const ConvW w[4]);
    int64_t dft(const std::string& key, int rows, int cols, int kind, int n);       // offset of a DFT matrix inside consts
    std::map<std::string, int64_t> dftOff_;
};

} // namespace vsr
