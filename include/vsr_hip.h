/* libvsr_hip.so -- C-ABI of the MI355X (gfx950) STTN inpainting hot path.
 *
 * The reference (YaoFANGUK/video-subtitle-remover) has no FFI: its drop-in boundary is the
 * duck-typed Python plugin selected by --inpaint-mode in backend/main.py:375-386.  The Python
 * plugin classes of this repo (video-subtitle-remover_amd/backend/inpaint/ (sttn_auto_inpaint.py ...)) keep those
 * signatures and bind the entry points below through ctypes.  Each
 * entry point cites the reference code it replaces (paths relative to the reference root).
 *
 * Conventions: plain pointers and sizes, no torch types; every function returns 0 on success
 * or a negative VSR_ERR_* code (message via vsr_last_error(), thread-local); no exceptions
 * cross the ABI; "dev" pointers are device memory of the model's GPU, everything else is
 * host memory; `stream` is a hipStream_t passed as void* (NULL = default stream); a model
 * handle is not thread-safe (one per stream).  There is NO CPU fallback: without a HIP
 * device every compute entry point fails with VSR_ERR_NOGPU.
 */
#ifndef VSR_HIP_H
#define VSR_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSR_OK 0
#define VSR_ERR_ARG (-1)
#define VSR_ERR_STATE (-2)
#define VSR_ERR_HIP (-3)
#define VSR_ERR_NOGPU (-4)

#define VSR_VARIANT_STTN_AUTO 0 /* backend/inpaint/sttn/auto_sttn.py InpaintGenerator, 640x120 */
#define VSR_VARIANT_STTN_DET 1  /* backend/inpaint/sttn/network_sttn.py InpaintGenerator, 432x240 */

typedef struct vsr_sttn vsr_sttn_t;
typedef struct vsr_plan vsr_plan_t;

int vsr_version(void);
const char* vsr_last_error(void);
int vsr_device_count(void); /* 0 when no HIP device is visible; never fails */

/* ---------------------------------------------------------------------------------------
 * Model lifecycle.  Replaces STTNInpaint.__init__ (backend/inpaint/sttn_auto_inpaint.py:29-41:
 * InpaintGenerator().to(device); load_state_dict(torch.load(path)['netG']); eval()).
 * set_param takes one state_dict entry (reference key, fp32, checkpoint shape); finalize
 * verifies that exactly the 112 reference tensors are present with the reference shapes
 * (strict load), repacks conv weights [Cout][Cin][kh][kw] -> [Cout][(ky*kw+kx)*Cin+ci],
 * fuses Q/K/V 1x1 weights, and uploads to `device` (device < 0: pack only, host side).
 * ------------------------------------------------------------------------------------- */
int vsr_sttn_create(int variant, vsr_sttn_t** out);
int vsr_sttn_set_param(vsr_sttn_t* h, const char* key, const float* data, const int64_t* shape, int ndim);
int vsr_sttn_finalize(vsr_sttn_t* h, int device);
void vsr_sttn_destroy(vsr_sttn_t* h);
int vsr_sttn_geometry(const vsr_sttn_t* h, int32_t* model_w, int32_t* model_h, int32_t* neighbor_stride,
                      int32_t* ref_length);
int vsr_sttn_set_window(vsr_sttn_t* h, int neighbor_stride, int ref_length); /* config.sttnNeighborStride / sttnReferenceLength */
/* copy of the packed weight buffer (host) -- used by the CPU replay tests */
int64_t vsr_sttn_packed_weights(const vsr_sttn_t* h, float* out, int64_t capacity);

/* ---------------------------------------------------------------------------------------
 * STTNInpaint.inpaint(frames) (backend/inpaint/sttn_auto_inpaint.py:122-164):
 *   frames_dev : [L][model_h][model_w][3] uint8 BGR (already at model resolution)
 *   comp_dev   : [L][model_h][model_w][3] float32 RGB; holds integral values (a uint8 image)
 *                where counts[i] == 1, the pairwise 0.5/0.5 running average otherwise
 *   counts     : host [L], number of windows that decoded frame i
 * ------------------------------------------------------------------------------------- */
int vsr_sttn_inpaint(vsr_sttn_t* h, const uint8_t* frames_dev, int L, float* comp_dev, int32_t* counts,
                     void* stream);

/* ---------------------------------------------------------------------------------------
 * One iteration of the chunk loop of STTNAutoInpaint.__call__
 * (backend/inpaint/sttn_auto_inpaint.py:242-317; same math as STTNInpaint.__call__ :43-97):
 * for every inpaint area (ymin,ymax,xmin,xmax as returned by get_inpaint_area_by_mask,
 * tools/inpaint_tools.py:49-242; full width) crop the strip of the selected frames,
 * cv2.resize to model size, inpaint, cv2.resize back, astype(uint8), RGB->BGR and write
 * mask*comp + (1-mask)*frame IN PLACE into frames_dev.
 *   frames_dev : [L][H][W][3] uint8 BGR, modified in place
 *   mask_dev   : [H][W] uint8, non-zero = inpaint (the thresholded mask, cv2.threshold(.,127,1))
 *   areas      : host [n_areas][4]
 *   sel/nsel   : host indices (ascending) of the frames of this chunk that are inside the A/B
 *                sections (is_frame_number_in_ab_sections); NULL/0 = all L frames
 * ------------------------------------------------------------------------------------- */
int vsr_sttn_auto_chunk(vsr_sttn_t* h, uint8_t* frames_dev, int L, int H, int W, const uint8_t* mask_dev,
                        int n_areas, const int32_t* areas, const int32_t* sel, int nsel, void* stream);
/* The same with a promise about the mask: mask_rows host [n_areas][2] = for every area the rows [lo, hi) of its strip
 * (0 = the strip's first row) outside which mask_dev is zero.  The strip is written back only where the mask is set
 * (:312-315), so of the model-resolution output only the rows those strip rows are resized from are ever read: the decoder
 * computes these rows and what they depend on and nothing else -- the frames come out bit for bit as from vsr_sttn_auto_chunk
 * (a subtitle line in a 360-row strip: ~40 % of the decoder's rows, 4 % of the chunk).  lo >= hi for an area = no promise.
 * A mask that is set outside the promised rows gets undefined pixels there.  VSR_DECODE_ROWS=0 ignores the promise. */
int vsr_sttn_auto_chunk_rows(vsr_sttn_t* h, uint8_t* frames_dev, int L, int H, int W, const uint8_t* mask_dev,
                             int n_areas, const int32_t* areas, const int32_t* mask_rows, const int32_t* sel, int nsel, void* stream);
/* ... and about its columns: mask_cols host [n_areas][2] = the frame columns [lo, hi) outside which the mask is zero.  The GEMMs of
 * the decoder and of the last block then take rectangles.  Built and replayed on the CPU in round 4, run on the GPU in round 5 (bit-equal frames; the default of the Python side since): calling
 * this entry point is the opt-in (the Python side does so with its VSR_DECODE_COLS switch); VSR_DECODE_COLS=0 in the environment makes
 * it ignore mask_cols (= vsr_sttn_auto_chunk_rows). */
int vsr_sttn_auto_chunk_box(vsr_sttn_t* h, uint8_t* frames_dev, int L, int H, int W, const uint8_t* mask_dev,
                            int n_areas, const int32_t* areas, const int32_t* mask_rows, const int32_t* mask_cols,
                            const int32_t* sel, int nsel, void* stream);
/* the column half of the same two questions (what VSR_DECODE_COLS=1 makes vsr_sttn_auto_chunk_box / vsr_sttn_det_batch_box do): the model columns
 * [*col_lo, *col_hi) decoded for a mask in frame columns [mask_col_lo, mask_col_hi) of a frame_w-wide frame (0, 0: all), and the
 * FLOPs of a plan restricted to a box of model rows and columns */
int vsr_sttn_decode_cols(vsr_sttn_t* h, int frame_w, int mask_col_lo, int mask_col_hi, int32_t* col_lo, int32_t* col_hi);
double vsr_sttn_flops_box(vsr_sttn_t* h, int L, int row_lo, int row_hi, int col_lo, int col_hi);
int vsr_sttn_det_batch_box(vsr_sttn_t* h, uint8_t* frames_dev, int L, int H, int W, const uint8_t* mask_dev, int n_areas,
                           const int32_t* areas, const int32_t* mask_rows, const int32_t* mask_cols, void* stream);   /* the same for sttn-det */
/* model-resolution rows [*row_lo, *row_hi) decoded for a strip of strip_h rows whose mask lives in rows [mask_row_lo, mask_row_hi),
 * and the FLOPs of one L-frame call decoded that way (vsr_sttn_flops = the whole image) */
int vsr_sttn_decode_rows(vsr_sttn_t* h, int strip_h, int mask_row_lo, int mask_row_hi, int32_t* row_lo, int32_t* row_hi);
double vsr_sttn_flops_rows(vsr_sttn_t* h, int L, int row_lo, int row_hi);

/* ---------------------------------------------------------------------------------------
 * sttn-det (backend/inpaint/sttn_det_inpaint.py; model created with VSR_VARIANT_STTN_DET).
 * vsr_sttn_det_inpaint = STTNDetInpaint.inpaint(frames, masks) (:124-174): frames [L][240][432][3] uint8 BGR,
 * masks [L][240][432] uint8 as produced by cv2.resize of the 0/255 mask strip; the encoder sees
 * frames*(1-(mask/255>0.5)), the output is pred*(mask>0)+frame*(1-(mask>0)) averaged like sttn-auto.
 * vsr_sttn_det_batch = STTNDetInpaint.__call__(input_frames, input_mask) (:38-99): mask_dev is the raw
 * [H][W] 0/255 mask; every area strip (height int(W*5/18) landscape, :48-51) is resized to 432x240 together
 * with its mask strip, inpainted and the WHOLE strip is overwritten with the up-scaled composite (:93).
 * ------------------------------------------------------------------------------------- */
int vsr_sttn_det_inpaint(vsr_sttn_t* h, const uint8_t* frames_dev, const uint8_t* masks_dev, int L, float* comp_dev,
                         int32_t* counts, void* stream);
int vsr_sttn_det_batch(vsr_sttn_t* h, uint8_t* frames_dev, int L, int H, int W, const uint8_t* mask_dev, int n_areas,
                       const int32_t* areas, void* stream);
/* with the promise of vsr_sttn_auto_chunk_rows (mask_rows host [n_areas][2]: strip rows outside which mask_dev is zero): the
 * prediction is taken only where the resized mask is non-zero (:132,168), every other pixel of the composite is the input
 * frame -- the decoder runs on the model rows the mask rows are resized to; same frames */
int vsr_sttn_det_batch_rows(vsr_sttn_t* h, uint8_t* frames_dev, int L, int H, int W, const uint8_t* mask_dev, int n_areas,
                            const int32_t* areas, const int32_t* mask_rows, void* stream);

/* Arithmetic of the contractions.  0 (default): exact fp32 -- v_mfma_f32_32x32x2_f32, bitwise an fmaf chain.
 * 1: split-half -- fp32 data and fp32 accumulation, each fp32 operand fed to the f16 matrix cores as
 * hi = fp16(x), lo = fp16(x - hi) and a*b taken as a_lo*b_hi + a_hi*b_lo + a_hi*b_hi (22 significand bits per
 * operand; 5.3x the fp32-MFMA rate).  Operands beyond the fp16 range make a result non-finite; that is detected
 * on the device and the chunk is then recomputed with the exact kernels (vsr_sttn_fallbacks counts them).
 * 2: the same arithmetic with the split done once by the PRODUCER of each tensor: every GEMM operand (weights,
 * activations, softmax probabilities) is kept in HBM in "split format" -- each aligned group of 32 fp32 slots
 * (128 B) holds the 32 fp16 hi halves followed by the 32 fp16 lo halves -- so tiles stream into LDS by DMA and
 * the kernel does no conversion.  Scores, split-K partial sums and the decoder output stay plain fp32.  Same
 * range guard and fp32 fallback as mode 1.
 * 3: fp16 operands, fp32 accumulation (BASELINE.json's "fp16 MFMA path"): mode 2's tensors and kernels with the
 * lo halves left out of the contractions -- one v_mfma_f32_32x32x16_f16 per product, 11-bit operands; bias,
 * activation, residual adds, softmax and the decoder output stay fp32-accurate.  Same guard and fallback.
 * Environment default: VSR_PRECISION=split (mode 1) / VSR_PRECISION=2 / VSR_PRECISION=3.
 * WHAT THE RANGE GUARD DOES NOT SEE (round 6): modes 1-3 can stay inside the fp16 range and still miss the 50 dB bar --
 * on weights with sharp attention rows or heavy tails mode 3 measured 47 / 28 dB (STTN) where the benign draw gives 60.  Accuracy is
 * a property of the checkpoint, so the HOST side checks it: vsr_amd.engine.AccuracyGuard runs the first unit of work of a
 * reduced-precision mode (and every 256th) in mode 0 as well and demotes the handle when they differ by more than 50 dB.  A caller
 * that binds this ABI directly and uses modes 1-3 should do the same with its first chunk. */
/* how THIS library instance reads one of its process-wide switches (read once, at first use): "VSR_DECODE_ROWS", "VSR_DECODE_COLS",
 * "VSR_QKV0_SHARED", "VSR_TRIM_LAST_BLOCK" -> 0 / 1, anything else -> -1.  The Python side (vsr_amd/switches.py) asserts that both agree. */
int vsr_switch_state(const char* name);
int vsr_sttn_set_precision(vsr_sttn_t* h, int mode);
int64_t vsr_sttn_fallbacks(const vsr_sttn_t* h);
/* Streams a chunk's sliding windows are issued on (1 .. 4; default 2, environment VSR_STTN_LANES).  The windows of
 * STTNInpaint.inpaint (sttn_auto_inpaint.py:142-162) share nothing until their decoded frames are averaged into `comps`, so window w
 * runs on stream w % lanes (engine-owned streams beyond the caller's) in that lane's own window buffers while the averaging stays in window order (events):
 * the partial last round of tiles of one lane's launch is filled by another lane's kernel.  Results are identical for every
 * lane count; the caller's stream is joined behind the others before any entry point returns to it. */
int vsr_sttn_set_lanes(vsr_sttn_t* h, int lanes);

/* algorithmic model FLOPs of one inpaint(L) call (2*M*N*K over every conv / GEMM, unpadded): what this library contracts.  The
 * last transformer block of a window is computed for the neighbour frames only -- the decoder reads nothing else
 * (sttn_auto_inpaint.py:150) --; vsr_sttn_flops_reference counts those rows too, i.e. what the reference's modules compute
 * (642.8 GFLOP per frame of a 50-frame chunk) */
double vsr_sttn_flops(vsr_sttn_t* h, int L);
double vsr_sttn_flops_reference(vsr_sttn_t* h, int L);

/* GPU timing of the next calls with hipEvents on the launch stream, per op tag and per kernel symbol: enable = 1 brackets every op
 * (about 2.5 % of a chunk's wall time: 1 300 launches), 2 only the launches of the 128x64 NK gather-GEMM -- the dominant kernel
 * symbol, what bench.py's roofline object needs inside its timed region --, 0 = off */
int vsr_sttn_timing(vsr_sttn_t* h, int enable);
int vsr_sttn_timing_get(vsr_sttn_t* h, const char* tag_prefix, double* total_ms, int32_t* launches, double* flops);
int vsr_sttn_timing_reset(vsr_sttn_t* h);

/* ---------------------------------------------------------------------------------------
 * Kernel-level entry points (parity tests call the kernels through these).
 * ------------------------------------------------------------------------------------- */
#define VSR_GG_KC 32 /* K / N chunk granularity of the offset tables */
enum { VSR_BMODE_NK = 0, VSR_BMODE_KN = 1 };
enum { VSR_ACT_NONE = 0, VSR_ACT_LRELU02 = 1, VSR_ACT_RELU = 2, VSR_ACT_LRELU01 = 3,
       VSR_ACT_OUT_SPLIT = 0x100, /* variants 5, 6: OR-ed into act, C is written in split format */
       VSR_ACT_POST_RELU = 0x200, /* OR-ed into act: C = relu(act(..) + R)  (residual blocks, raft/extractor.py:48-58) */
       /* the two halves of softmax(QK^T / sqrt(D)) . V (auto_sttn.py:141-145) without a probability matrix in memory:
        * ROW_MAX on the score GEMM (NK, variant 3, splitK 1, no residual): besides C, the maximum of every output row is kept in
        * ((uint32_t*)R)[m] -- monotone unsigned encoding of the float, atomic max over the N tiles, zeroed by the caller;
        * A_EXP on the P.V GEMM (KN, variant 1 | VSR_VARIANT_A_EXP): the A operand is 2^(A[m][k] - rowmax[m]) (the caller folds
        * log2(e) into the scores' scale) with rowmax = ((const uint32_t*)bias)[m] in that encoding, and the row sums l[m] =
        * sum_k 2^(..) are taken while the tiles are staged:
        * splitK 1 writes C = (expA . B) / l; splitK > 1 writes the partial planes unnormalised and the partial sums to
        * ((float*)R)[split * tilesM * BM + m], for vsr_launch_reduce_scatter's caller to divide by their total. */
       VSR_ACT_ROW_MAX = 0x400, VSR_ACT_A_EXP = 0x800 };
#define VSR_VARIANT_A_EXP 0x100 /* OR-ed into the kernel variant 1 of a KN launch whose problems may carry VSR_ACT_A_EXP */
#define VSR_VARIANT_NARROW 8    /* problems of at most FOUR output columns (the last conv of a head: RAFT update.py:6-12 flow head conv2,
                                 * recurrent_flow_completion.py:271-276 upsample.2, propainter.py:268-276 decoder's last conv): a dot-product
                                 * kernel, exact fp32, no matrix cores (csrc/gather_gemm_narrow.h).  NK problems laid out for VSR_TILE_256x32,
                                 * splitK = 1, no residual, act in VSR_ACT_NONE..VSR_ACT_LRELU01, N * K floats of weights within 40 KB of LDS.
                                 * The flow engines pick it by themselves (VSR_GG_NARROW=0: never) */
enum { VSR_TILE_128x128 = 0, VSR_TILE_256x32 = 1, VSR_TILE_256x64 = 2, VSR_TILE_128x64 = 3,
       VSR_TILE_256x128 = 4, /* 8 waves; the fp16-operand kernel (variant 6, NK) only */
       VSR_TILE_256x256 = 5  /* 8 waves, one workgroup per CU; variants 5 / 6, NK only.  tilesM may exceed ceil(M / 256): a tile then covers
                              * roundup32(ceil(M / tilesM)) rows (gather_gemm_v7.h), which is how a launch is cut into whole rounds.
                              * (6 was the exact-fp32 288 x 256 tile of round 4, opt-in, removed in round 6: DESIGN 4.1) */ };

/* C[rowC[m]+colC[n/32]+n%32] = act(alpha*sum_k A[rowA[m]+colA[k/32]+k%32]*B(k,n) + bias[n]) + R[rowR[m]+colC[n/32]+n%32]
 *   NK: B(k,n) = B[rowB[n]+colB[k/32]+k%32]   KN: B(k,n) = B[rowB[k]+colB[n/32]+n%32]
 * covers torch conv2d / matmul on this path (auto_sttn.py:75-95,140-145,162-164,172-174,214-218) */
typedef struct GGProblem {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const float* R;
    const int32_t* rowA;
    const int32_t* colA;
    const int32_t* rowB;
    const int32_t* colB;
    const int32_t* rowC;
    const int32_t* colC;
    const int32_t* rowR;
    int32_t M, N, K;
    int32_t tilesM, tilesN;
    int32_t splitK;
    int32_t chunksPerSplit;
    int32_t tileStart;
    int32_t act;
    float alpha;
    int64_t splitStride;
} GGProblem;

/* resident launch list of gather-GEMM problems (fp32 tensors, kernel variants 1..4): descriptors and tile queues are uploaded
 * once; vsr_gemm_plan_run is asynchronous on `stream` (one memset + one launch).  The pointers inside the problems must stay
 * valid for the life of the plan.  Used by the text detector's convolutions (backend/tools/ocr_det.py). */
typedef struct vsr_gemm_plan vsr_gemm_plan_t;
int vsr_gemm_plan_create(const GGProblem* probs, int nprobs, int tile_cfg, int bmode, int variant, vsr_gemm_plan_t** out);
int vsr_gemm_plan_run(vsr_gemm_plan_t* p, void* stream);
void vsr_gemm_plan_destroy(vsr_gemm_plan_t* p);

/* P = softmax(scale * sum_splits S) row-wise (auto_sttn.py:141-143) */
typedef struct SMProblem {
    const float* S;
    float* P;
    int32_t M, N, ldS, ldP, nsplit;
    int32_t rowStart;
    float scale;
    int32_t flags;      /* bit 0: P is written in split format (precision mode 2) */
    int64_t splitStride;
} SMProblem;

/* probs: HOST array whose pointers are device pointers; tileStart / rowStart are filled in */
int vsr_run_gather_gemm(const GGProblem* probs, int nprobs, int tile_cfg, int bmode, void* stream);
/* same with an explicit kernel variant: 1 one workgroup per tile, 2 persistent (register staged), 3 persistent
 * LDS-DMA (fp32 MFMA), 4 persistent split-half operands on the f16 matrix cores (see vsr_sttn_set_precision),
 * 5 the same on split-format tensors: A, B and R are read in split format; C is written in split format when
 * act carries VSR_ACT_OUT_SPLIT, as plain fp32 otherwise; 6 = variant 5 with the fp16 hi halves alone as
 * operands (fp16 x fp16 -> fp32 accumulate, one MFMA per product); VSR_VARIANT_NARROW: see above */
int vsr_run_gather_gemm_variant(const GGProblem* probs, int nprobs, int tile_cfg, int bmode, int variant, void* stream);
int vsr_run_softmax(const SMProblem* probs, int nprobs, void* stream);
/* fp32 -> split format (variant 5 operands): dst[32c .. 32c+31] as bytes = fp16 hi[0..31] | fp16 lo[0..31] of
 * src[32c .. 32c+31]; n a multiple of 32 */
int vsr_launch_to_split(const float* src_dev, float* dst_dev, int64_t n, void* stream);
/* a KN operand in split format, B(k, n) = B[rowB[k] + colB[n / 32] + n % 32], as the dense NK operand dst[n * ld + k] in split format
 * (the P.V product of the split-format modes runs on the NK kernels: V of auto_sttn.py:195-203 is turned once per product);
 * K, N multiples of 32, ld >= K */
int vsr_launch_kn_to_nk_split(const float* B_dev, const int32_t* rowB_dev, const int32_t* colB_dev, int K, int N, int64_t ld,
                              float* dst_dev, void* stream);

/* cv2.resize(..., INTER_LINEAR) on uint8 (fixed-point path), tables from vsr_cv2_linear_tables;
 * frame_idx (device, nullable) gathers source frames (sttn_auto_inpaint.py:269-271) */
int vsr_launch_resize_u8(const uint8_t* src_dev, int64_t src_frame_stride, int src_row_stride, int sw, int sh,
                         uint8_t* dst_dev, int dw, int dh, int nframes, int channels /*1 or 3*/,
                         const int32_t* frame_idx_dev,
                         const int32_t* xofs_dev, const int16_t* ialpha_dev, const int32_t* yofs_dev,
                         const int16_t* ibeta_dev, void* stream);
/* Stack(BGR->RGB) + /255 + *2-1 (utils/sttn_utils.py:73,111; sttn_auto_inpaint.py:128) fused with
 * the im2col of encoder conv1 (auto_sttn.py:76): out [n*(ih/2)*(iw/2)][32].  premask (sttn-det,
 * sttn_det_inpaint.py:134,143): pixels whose resized mask [n][ih][iw] is >= 128 enter as 0 */
int vsr_launch_norm_im2col(const uint8_t* img_dev, int ih, int iw, int nframes, float* out_dev, int premask,
                           const uint8_t* mask_dev, void* stream);
/* F.interpolate(scale_factor=2, bilinear, align_corners=True) on NHWC with halos (auto_sttn.py:124-126) */
int vsr_launch_upsample2x(const float* src_dev, int H, int W, int C, int halo_src, float* dst_dev, int halo_dst,
                          int nframes, void* stream);
/* tanh, (x+1)/2, *255, astype(uint8), pairwise overlap average (sttn_auto_inpaint.py:150-162); with
 * mask_dev != NULL the sttn-det model-resolution blend pred*(mask>0) + frame*(1-(mask>0))
 * (sttn_det_inpaint.py:132,168) against the model-res BGR input frames in_bgr_dev */
int vsr_launch_decode_out(const float* y_dev, int ldy, int pix, int nframes, const int32_t* frame_idx_dev,
                          const int32_t* first_dev, float* comp_dev, const uint8_t* in_bgr_dev,
                          const uint8_t* mask_dev, void* stream);
/* cv2.resize(comp,(W,split_h)) + astype(uint8) + BGR2RGB + mask blend (sttn_auto_inpaint.py:312-315);
 * mask_dev == NULL overwrites the whole strip (sttn_det_inpaint.py:93) */
int vsr_launch_upscale_blend(const float* comp_dev, int mw, int mh, const int32_t* is_float_dev,
                             uint8_t* frames_dev, int64_t frame_stride, int row_stride,
                             const int32_t* frame_idx_dev, const uint8_t* mask_dev, int mask_row_stride, int W,
                             int sh, int nframes, const int32_t* xofs_dev, const int16_t* ialpha_dev,
                             const float* falpha_dev, const int32_t* yofs_dev, const int16_t* ibeta_dev,
                             const float* fbeta_dev, void* stream);
/* out[rowC[m]+colC[n/32]+n%32] = sum_s part[s*split_stride + m*N + n]: combines the split-K partial
 * planes of a PV product and scatters them into the NHWC attention buffer */
int vsr_launch_reduce_scatter(const float* part_dev, int nsplit, int64_t split_stride, int M, int N,
                              const int32_t* rowC_dev, const int32_t* colC_dev, float* out_dev, void* stream);
/* host: OpenCV 4.11 resize() INTER_LINEAR tables: ofs[dsize], icoef[2*dsize] (x2048), fcoef[2*dsize] */
int vsr_cv2_linear_tables(int ssize, int dsize, int clamp_x, int32_t* ofs, int16_t* icoef, float* fcoef);

/* ---------------------------------------------------------------------------------------
 * RAFT optical flow -- the first stage of --inpaint-mode propainter.
 * Replaces RAFT_bi (backend/inpaint/video/model/modules/flow_comp_raft.py:27-55) and the network behind it
 * (backend/inpaint/video/raft/raft.py:24-146, extractor.py, corr.py, update.py), "things" configuration
 * (small = False, corr_levels 4, corr_radius 4), exact fp32 (propainter_inpaint.py:230 keeps RAFT in fp32).
 * ------------------------------------------------------------------------------------- */
typedef struct vsr_raft vsr_raft_t;
int vsr_raft_create(vsr_raft_t** out);
/* one entry of torch.load('raft-things.pth') with DataParallel's "module." prefix removed (flow_comp_raft.py:17-19);
 * fp32 contiguous (num_batches_tracked as a 0-d value); unknown keys and wrong shapes are errors */
int vsr_raft_set_param(vsr_raft_t* h, const char* key, const float* data, const int64_t* shape, int ndim);
/* all keys present -> BatchNorm folded, weights packed and uploaded; device < 0: host only (plan introspection) */
int vsr_raft_finalize(vsr_raft_t* h, int device);
void vsr_raft_destroy(vsr_raft_t* h);
int64_t vsr_raft_packed_weights(const vsr_raft_t* h, float* out, int64_t capacity);
/* RAFT_bi.forward(frames, iters): frames_dev uint8 [t][H][W][3] on the device (RGB, or BGR with bgr = 1), normalised
 * as to_tensors()(frames) * 2 - 1 (propainter_inpaint.py:214); H, W multiples of 8 and >= 128.  Outputs fp32 on the
 * device, both [t-1][2][H][W] (x then y displacement): fwd[i] = flow frame i -> i+1, bwd[i] = flow frame i+1 -> i. */
int vsr_raft_flows(vsr_raft_t* h, const uint8_t* frames_dev, int t, int H, int W, int iters, int bgr, float* fwd_dev,
                   float* bwd_dev, void* stream);
/* test hook: copy `count` floats at `offset` of workspace buffer `buf` (ids = the plan's buffer ids) to the host,
 * after a device synchronisation -- stage-by-stage parity against the CPU replay of the plan */
/* arithmetic of the contractions: 0 (default) exact fp32 MFMA; 1 split-half fp16 operands (22 significand bits) with fp32 accumulation,
 * range-guarded: a call whose operands leave the fp16 range is redone in fp32 and counted by *_fallbacks.  The reference runs RAFT in
 * fp32 and the other two networks in fp16 on a GPU (propainter_inpaint.py:140-146,230,249-251); mode 1 is closer to fp32 than either.
 * 2: fp16 operands (fp32 tensors rounded on their way into the matrix cores), fp32 accumulation, bias / activation / residual in fp32,
 * same range guard -- the arithmetic class of the reference's `.half()` flow-completion and generator modules. */
int vsr_raft_set_precision(vsr_raft_t* h, int mode);
int64_t vsr_raft_fallbacks(const vsr_raft_t* h);
int vsr_raft_read_buffer(vsr_raft_t* h, int buf, int64_t offset, int64_t count, float* out_host);
/* algorithmic FLOPs of one vsr_raft_flows call (2*M*N*K over every conv and the all-pairs correlation) */
double vsr_raft_flops(vsr_raft_t* h, int t, int H, int W, int iters);

/* ---------------------------------------------------------------------------------------
 * Recurrent flow completion -- the second stage of --inpaint-mode propainter.
 * Replaces RecurrentFlowCompleteNet.forward_bidirect_flow + combine_flow
 * (backend/inpaint/video/model/recurrent_flow_completion.py:313-348; network :206-311, BidirectionalPropagation :49-126,
 * SecondOrderDeformableAlignment :10-46 with torchvision.ops.deform_conv2d).  Exact fp32.
 * ------------------------------------------------------------------------------------- */
typedef struct vsr_rfc vsr_rfc_t;
int vsr_rfc_create(vsr_rfc_t** out);
/* one entry of torch.load('recurrent_flow_completion.pth') (:269-272): fp32 contiguous, unknown keys / wrong shapes are errors */
int vsr_rfc_set_param(vsr_rfc_t* h, const char* key, const float* data, const int64_t* shape, int ndim);
int vsr_rfc_finalize(vsr_rfc_t* h, int device);
void vsr_rfc_destroy(vsr_rfc_t* h);
int64_t vsr_rfc_packed_weights(const vsr_rfc_t* h, float* out, int64_t capacity);
/* flows_f / flows_b: fp32 [t-1][2][H][W] on the device (RAFT's outputs, unmasked); masks: uint8 [t][H][W], non-zero = hole
 * (flow_masks of propainter_inpaint.py:215).  Outputs fp32 [t-1][2][H][W]: pred * mask + flow * (1 - mask) per direction
 * (combine_flow).  H, W multiples of 8. */
int vsr_rfc_complete(vsr_rfc_t* h, const float* flows_f_dev, const float* flows_b_dev, const uint8_t* masks_dev, int t, int H,
                     int W, float* out_f_dev, float* out_b_dev, void* stream);
int vsr_rfc_set_precision(vsr_rfc_t* h, int mode);        /* see vsr_raft_set_precision */
int64_t vsr_rfc_fallbacks(const vsr_rfc_t* h);
int vsr_rfc_read_buffer(vsr_rfc_t* h, int buf, int64_t offset, int64_t count, float* out_host);   /* test hook, see vsr_raft_read_buffer */
double vsr_rfc_flops(vsr_rfc_t* h, int t, int H, int W);

/* ---------------------------------------------------------------------------------------
 * ProPainter generator, built stage by stage.
 * vsr_pp_img_propagation = InpaintGenerator.img_propagation(masked_frames, completed_flows, masks, 'nearest')
 * (backend/inpaint/video/model/propainter.py:316-319; BidirectionalPropagation(3, learnable=False) :104-193).
 * ------------------------------------------------------------------------------------- */
typedef struct vsr_pp vsr_pp_t;
int vsr_pp_create(int device, vsr_pp_t** out);
void vsr_pp_destroy(vsr_pp_t* h);
/* masked_frames fp32 [t][3][H][W] in [-1,1] (frames * (1 - mask)), flows fp32 [t-1][2][H][W] (completed), masks uint8 [t][H][W]
 * (non-zero = hole), all on the device.  Outputs: propagated frames fp32 [t][3][H][W] and updated masks uint8 [t][H][W] {0,1}. */
int vsr_pp_img_propagation(vsr_pp_t* h, const float* masked_frames_dev, const float* flows_f_dev, const float* flows_b_dev,
                           const uint8_t* masks_dev, int t, int H, int W, float* out_frames_dev, uint8_t* out_masks_dev, void* stream);
/* InpaintGenerator weights: one entry of torch.load('ProPainter.pth') (propainter.py:308-311) per call, then finalize
 * (packs the grouped encoder convs, pads the 261 / 258-channel propagation convs, fuses q/k/v; uploads when the handle
 * has a device) */
int vsr_pp_set_param(vsr_pp_t* h, const char* key, const float* data, const int64_t* shape, int ndim);
int vsr_pp_finalize(vsr_pp_t* h);
int64_t vsr_pp_packed_weights(const vsr_pp_t* h, float* out, int64_t capacity);
/* host helper: one flag per 5x9 attention window (row-major), SparseWindowAttention's "window touches the hole" test
 * (sparse_transformer.py:229-236) from the HOST copy of the local frames' masks_in, uint8 [lt][H][W]; returns the number
 * of windows (or < 0).  The flags select the key set of every window and are part of the plan. */
int vsr_pp_window_flags(const uint8_t* masks_host, int lt, int H, int W, uint8_t* flags, int capacity);
/* InpaintGenerator.forward(masked_frames, completed_flows, masks_in, masks_updated, num_local_frames) in eval mode
 * (propainter.py:321-378): frames fp32 [t][3][H][W] (the first lt are the local ones), flows fp32 [lt-1][2][H][W], masks uint8
 * [t][H][W], all on the device; window_flags from vsr_pp_window_flags.  Output: tanh image fp32 [lt][3][H][W]. */
int vsr_pp_forward(vsr_pp_t* h, const float* frames_dev, const float* flows_f_dev, const float* flows_b_dev,
                   const uint8_t* masks_in_dev, const uint8_t* masks_updated_dev, int t, int lt, int H, int W,
                   const uint8_t* window_flags, int nflags, float* out_dev, void* stream);
/* The same with a promise of the caller: of the output it reads rows [row_lo, row_hi) and columns [col_lo, col_hi) only (the
 * plugin blends a window's prediction into its frames under the dilated mask, propainter_inpaint.py:350-357); lo = hi = 0: the
 * whole axis.  The soft composition's embedding and the decoder's convs then run on what that box depends on; inside it the
 * output is the one of vsr_pp_forward, outside it is undefined.  (Built and replayed on the CPU in round 4, run on the GPU in round 5: the plugin's
 * default since; VSR_PP_DECODE_BOX=0 makes it promise nothing.) */
int vsr_pp_forward_box(vsr_pp_t* h, const float* frames_dev, const float* flows_f_dev, const float* flows_b_dev,
                       const uint8_t* masks_in_dev, const uint8_t* masks_updated_dev, int t, int lt, int H, int W,
                       const uint8_t* window_flags, int nflags, int row_lo, int row_hi, int col_lo, int col_hi, float* out_dev, void* stream);
/* FLOPs of that call; *reference (may be NULL) = what the reference spends on the same window */
double vsr_pp_flops_box(vsr_pp_t* h, int t, int lt, int H, int W, const uint8_t* window_flags, int nflags, int row_lo, int row_hi,
                        int col_lo, int col_hi, double* reference);
/* The encoder and the soft split ONCE PER FRAME.  Both are per-frame functions of a frame's own inputs (propainter.py:333-335: the
 * frames are a batch dimension of the encoder; sparse_transformer.py:7-31), and the plugin's windows overlap -- a frame is a local
 * frame of two or three windows and, if it is one of the every-tenth reference candidates, a reference frame of most others
 * (propainter_inpaint.py:317-341) -- so InpaintGenerator.forward encodes every frame of a 70-frame batch 3.2 times over.
 *   vsr_pp_encode         n frames (fp32 [n][3][H][W], masks uint8 [n][H][W]) -> features fp32 [n][H/4][W/4][128]; for the first
 *                         ntok_frames of them also the soft-split tokens fp32 [ntok_frames][tokens][512] (what a REFERENCE frame
 *                         contributes; a local frame's tokens are taken after the feature propagation and are not cacheable)
 *   vsr_pp_forward_cached vsr_pp_forward_box without the encoder: cache_idx (host, t entries) names for each of the lt local frames
 *                         its entry of the feature cache and for each reference frame its entry of the token cache.
 * Same GEMM rows in the same K order as in vsr_pp_forward: the output is the same.  Built and replayed on the CPU in round 4
 * (tests/test_pp_replay.py::test_generator_replay_encoder_cache) and on the GPU in round 5 (tests/test_gpu_pp.py::test_generator_encoder_cache:
 * bit equality with vsr_pp_forward); the plugin's default since (VSR_PP_ENC_CACHE=0 restores the per-window encoder). */
int vsr_pp_encode(vsr_pp_t* h, const float* frames_dev, const uint8_t* masks_in_dev, const uint8_t* masks_updated_dev, int n, int ntok_frames,
                  int H, int W, float* feat_out_dev, float* tok_out_dev, void* stream);
int vsr_pp_forward_cached(vsr_pp_t* h, const float* feat_cache_dev, const float* tok_cache_dev, const int32_t* cache_idx,
                          const float* flows_f_dev, const float* flows_b_dev, const uint8_t* masks_in_dev, const uint8_t* masks_updated_dev,
                          int t, int lt, int H, int W, const uint8_t* window_flags, int nflags, int row_lo, int row_hi, int col_lo, int col_hi,
                          float* out_dev, void* stream);
int vsr_pp_token_count(int H, int W);                    /* soft-split tokens per frame */
/* The plugin's own array work (PropainterInpaint.inpaint, propainter_inpaint.py:190-361) on frames that stay in HBM as the uint8
 * BGR crops [n][h][w][3]; mask: the dilated mask uint8 [h][w] (non-zero = hole), the same for every frame (:195-197).
 *   prepare : masked_frames fp32 [n][3][h][w] = (to_tensors(RGB) * 2 - 1) * (1 - mask)                                  (:193-213,298)
 *   compose : updated_frames = frames * (1 - mask) + prop * mask                                                        (:314)
 *   blend   : one generator window (:345-357): pred fp32 [lt][3][h][w] (tanh output) of the frames frame_idx[0..lt), u8
 *             truncation, hole select, pairwise 0.5 / 0.5 average with truncation after every average unless first[k];
 *             comp uint8 [n][h][w][3] is kept in BGR order.  frame_idx / first: int32 device arrays. */
int vsr_pp_prepare_frames(const uint8_t* bgr_dev, const uint8_t* mask_dev, int n, int h, int w, float* masked_dev, void* stream);
int vsr_pp_compose_frames(const uint8_t* bgr_dev, const uint8_t* mask_dev, const float* prop_dev, int n, int h, int w, float* out_dev, void* stream);
int vsr_pp_blend_window(const float* pred_dev, const uint8_t* bgr_dev, const uint8_t* mask_dev, const int32_t* frame_idx_dev,
                        const int32_t* first_dev, int lt, int h, int w, uint8_t* comp_dev, void* stream);
int vsr_pp_set_precision(vsr_pp_t* h, int mode);          /* see vsr_raft_set_precision; applies to vsr_pp_forward */
int64_t vsr_pp_fallbacks(const vsr_pp_t* h);
int vsr_pp_read_buffer(vsr_pp_t* h, int buf, int64_t offset, int64_t count, float* out_host);   /* test hook */
double vsr_pp_flops(vsr_pp_t* h, int t, int lt, int H, int W, const uint8_t* window_flags, int nflags);

/* Launch timing of the flow engines (RAFT, flow completion, ProPainter generator, LaMa), bench / profile use: with enable != 0 every op
 * a replayed plan launches is bracketed by HIP events on the launch stream; vsr_flow_timing_get sums time (ms), launches and algorithmic
 * FLOPs of the records whose key starts with `prefix`.  Keys: "<raft|rfc|pp|lama>:gg:<tile cfg>:<bmode>:v<kernel variant>:<op tag>" for
 * gather-GEMM launches, "<engine>:op:<op tag>" for everything else.  Process-wide state, not thread-safe. */
int vsr_flow_timing(int enable);
int vsr_flow_timing_reset(void);
int vsr_flow_timing_get(const char* prefix, double* total_ms, int64_t* launches, double* flops);
int64_t vsr_flow_timing_keys(char* buf, int64_t capacity);   /* '\n'-separated keys; returns the bytes needed */

/* ---------------------------------------------------------------------------------------
 * LaMa.  Replaces `self.model = torch.jit.load(model_path)` and `self.model(image, mask)` of
 * backend/inpaint/lama_inpaint.py:13,24,55 together with the array work around them (lama_util.get_image /
 * pad_img_to_modulo / the (mask > 0) * 1 tensor :48-52, clip(x * 255).astype(uint8) and the crop :56-60).  The network is the
 * published big-LaMa generator (csrc/lama_plan.h); its weights arrive as the state_dict entries of the exported module's
 * generator (`model.N...`, an optional `generator.` prefix is dropped, `num_batches_tracked` ignored).
 * ------------------------------------------------------------------------------------- */
typedef struct vsr_lama vsr_lama_t;
int vsr_lama_create(vsr_lama_t** out);
int vsr_lama_set_param(vsr_lama_t* h, const char* key, const float* data, const int64_t* shape, int ndim);
int vsr_lama_finalize(vsr_lama_t* h, int device);       /* device < 0: pack only (host-side plan tests) */
void vsr_lama_destroy(vsr_lama_t* h);
int vsr_lama_blocks(const vsr_lama_t* h);               /* FFC residual blocks found in the state_dict (18 for big-lama) */
int64_t vsr_lama_packed_weights(const vsr_lama_t* h, float* out, int64_t capacity);
/* B images uint8 [H][W][3] (channel order as given: the reference feeds BGR frames) with masks uint8 [H][W] (non-zero = hole),
 * rows contiguous: image b at img_dev + b * img_frame_stride bytes, mask b at mask_dev + b * mask_frame_stride (0: one mask for
 * all), result b at out_dev + b * out_frame_stride (may alias the input).  Any H, W >= 16: padded to multiples of 8 exactly as
 * lama_util.pad_img_to_modulo does (bottom / right, symmetric) and cropped back.  The whole image is rewritten (lama_inpaint.py:106). */
int vsr_lama_inpaint(vsr_lama_t* h, const uint8_t* img_dev, int64_t img_frame_stride, const uint8_t* mask_dev, int64_t mask_frame_stride,
                     int B, int H, int W, uint8_t* out_dev, int64_t out_frame_stride, void* stream);
int vsr_lama_set_precision(vsr_lama_t* h, int mode);    /* see vsr_raft_set_precision */
int64_t vsr_lama_fallbacks(const vsr_lama_t* h);
int vsr_lama_read_buffer(vsr_lama_t* h, int buf, int64_t offset, int64_t count, float* out_host);   /* test hook */
double vsr_lama_flops(vsr_lama_t* h, int B, int H, int W);

/* ---------------------------------------------------------------------------------------
 * Text detector: the operators of the PP-OCRv5 detection inference programs the reference
 * loads through paddleocr (backend/tools/subtitle_detect.py:41-58, backend/models/V5/{ch_det,ch_det_fast}/inference.json).
 * NCHW fp32 device tensors; the host runner (backend/tools/ocr_det.py) walks the program and calls one launcher per op.
 * act: 0 none, 1 relu, 2 hardswish.  Paddle operator definitions: conv2d / depthwise_conv2d (zero padding (pt, pl) before,
 * output Ho x Wo), conv2d_transpose 2x2 stride 2 (weight [Cin][Cout][2][2], depthwise: [C][1][2][2]), elementwise add (op 0) /
 * multiply (op 1) with b of the same shape (bmode 0), [C] (1), [N*C] (2) or one scalar (3), unary relu 0 / hardswish 1 / hardsigmoid(p0, p1) 2 /
 * sigmoid 3 / x*p0+p1 4, inference batch_norm as x*scale[c]+shift[c], adaptive average pool to 1x1, max pool, nearest_interp
 * (integer scale), and the inference.yml pre-processing NormalizeImage + ToCHWImage on a BGR uint8 image.
 * ------------------------------------------------------------------------------------- */
int vsr_det_launch_conv2d(const float* x, const float* w, const float* bias, int N, int Cin, int H, int W, int Cout, int kh, int kw,
                          int sh, int sw, int pt, int pl, int Ho, int Wo, int depthwise, int act, float* out, void* stream);
int vsr_det_launch_deconv2x2(const float* x, const float* w, int N, int Cin, int H, int W, int Cout, int depthwise, float* out, void* stream);
int vsr_det_launch_binary(const float* a, const float* b, int op, int64_t total, int C, int64_t HW, int bmode, float* out, void* stream);
int vsr_det_launch_unary(const float* x, int64_t total, int kind, float p0, float p1, float* out, void* stream);
int vsr_det_launch_affine(const float* x, const float* scale, const float* shift, int64_t total, int C, int64_t HW, float* out, void* stream);
int vsr_det_launch_gap(const float* x, int64_t planes, int64_t HW, float* out, void* stream);
int vsr_det_launch_maxpool(const float* x, int64_t planes, int H, int W, int kh, int kw, int sh, int sw, int pt, int pl, int Ho, int Wo,
                           float* out, void* stream);
int vsr_det_launch_nearest(const float* x, int64_t planes, int H, int W, int s, float* out, void* stream);
int vsr_det_launch_normalize(const uint8_t* img_bgr, int H, int W, float* out_chw, void* stream);
/* DBPostProcess, device part (paddleocr DBPostProcess.boxes_from_bitmap on `prob > thresh`): 8-connected components by union-find;
 * labels int32 [H*W] (label = raster index of the component's first pixel, -1 = background), stats int32 [5*H*W] scratch, comps
 * int32 [cap][6] = (label, area, xmin, xmax, ymin, ymax) unordered, *count = components found (may exceed cap) */
int vsr_det_launch_ccl(const float* prob_dev, int H, int W, float thresh, int32_t* labels_dev, int32_t* stats_dev, int32_t* comps_dev, int cap,
                       int32_t* count_dev, void* stream);
/* The whole DBPostProcess of one probability map on the device (reference: backend/tools/subtitle_detect.py:41-82 calls paddleocr's
 * TextDetection, whose post-process is PaddleX's DBPostProcess with inference.yml:49-53's thresh / box_thresh / unclip_ratio): the
 * labelling above, the hole count (Euler number), then per component get_mini_boxes (hull of the per-row extreme pixels,
 * minimum-area rectangle), box_score_fast (cv2.fillPoly's raster of the integer-truncated rectangle), unclip (ClipperLib's rounded
 * offset), the second get_mini_boxes, rescale to the source image.  count int32 [4]; ext int32 [cap][H][2] scratch; out int32
 * [4 + 16*cap]: out[0] = components found, out[1] = holes (their borders are contours too: such a map is the host's), then per slot
 * (flag, x0,y0, x1,y1, x2,y2, x3,y3, score bits, ...): flag 1 = box (top-left, top-right, bottom-right, bottom-left), 0 = rejected,
 * -1 = not processed (component taller than 256 rows / offset polygon beyond the point buffer); out[0] > cap: nothing else is valid. */
int vsr_det_launch_db_boxes(const float* prob_dev, int H, int W, float thresh, int src_h, int src_w, float box_thresh, float unclip_ratio,
                            int min_size, int32_t* labels_dev, int32_t* stats_dev, int32_t* comps_dev, int32_t* count_dev, int32_t* ext_dev,
                            int32_t* out_dev, int cap, void* stream);
/* HOST helper of the post-process (no device work): border following over a thresholded map -- cv2.findContours(RETR_LIST) as
 * PaddleX's DBPostProcess calls it (Suzuki & Abe, 8-connected foreground; outer and hole borders in raster order of discovery).
 * bitmap uint8 [H*W] (non-zero = foreground); points_xy int32 [cap_points][2] receives every border's pixels (x, y) back to
 * back, border_start int64 [cap_borders] the index of each border's first point.  Returns 0, or -100 when a buffer was too small
 * (*n_borders / *n_points then hold the sizes needed).  Used for the maps the device path hands back (holes, overflow). */
int vsr_host_trace_borders(const uint8_t* bitmap, int H, int W, int32_t* points_xy, int64_t cap_points, int64_t* border_start,
                           int32_t cap_borders, int32_t* n_borders, int64_t* n_points);
int vsr_det_launch_copy(const void* src_dev, int64_t src_pitch, void* dst_dev, int64_t dst_pitch, int64_t width_bytes, int64_t rows, void* stream);   /* channel concat: one strided block copy per part */
/* layout changes around the dense convolutions that run as gather-GEMMs (vsr_gemm_plan_*): one NCHW image -> zero-padded NHWC
 * [Hp][Wp][Cp] (image origin at (pt, pl), Cp a multiple of 32), and GEMM output [P pixels][Np] -> NCHW [C][P] with an optional
 * per-channel affine (the batch_norm or bias add that follows the conv) and activation (act 1 relu, 2 hardswish) */
int vsr_det_launch_nchw_to_nhwc(const float* x, int n, int C, int H, int W, int pt, int pl, int Hp, int Wp, int Cp, float* out, void* stream);
int vsr_det_launch_nhwc_to_nchw(const float* in, int n, int C, int64_t P, int Np, const float* scale, const float* shift, int act, float* out, void* stream);
/* NHWC-resident detector plan (backend/tools/ocr_det_nhwc.py, round 6): activations stay in zero-haloed NHWC buffers between the convs of
 * the program.  A VIEW is (pointer to channel c0 of interior pixel (0, 0) of image 0, floats per image, floats per row, floats per pixel):
 * halo pixels and the channels a slice is padded with are zero and never written, so a tap outside the image reads a zero and a
 * channel concat (paddle concat, axis 1) is its producers writing their slices of one buffer.  Pointers and strides keep 16-byte alignment.
 *   to_view / from_view: NCHW planes <-> the interior of a view (to_view writes Cw >= C channels, a multiple of 32, zeros beyond C)
 *   dwconv_view: depthwise_conv2d (weights tap-major [kh * kw][C]) + the inference batch_norm_ as scale / shift (nullable) + act (0, 1 relu, 2 hardswish)
 *   nearest_view: nearest_interp by an integer scale
 *   dots_view: n_out = 1: conv2d 1x1 to ONE channel (w [C]); n_out = 4: conv2d_transpose 2x2 / stride 2 to one channel (w [(dy, dx)][C]);
 *              + bias (bias[0]) + act (0, 1, 2, 3 sigmoid); out is the plain [n][H][W] / [n][2H][2W] map */
int vsr_det_launch_to_view(const float* x, int n, int C, int H, int W, int Cw, float* out, int64_t img_stride, int64_t row_stride, int Cs, void* stream);
int vsr_det_launch_from_view(const float* in, int64_t img_stride, int64_t row_stride, int Cs, int n, int C, int H, int W, float* out,
                             int64_t out_img_stride, void* stream);
int vsr_det_launch_dwconv_view(const float* in, int64_t in_img, int64_t in_row, int in_cs, const float* w, const float* scale, const float* shift, int N,
                               int C, int kh, int kw, int sh, int sw, int pt, int pl, int Ho, int Wo, int act, float* out, int64_t out_img,
                               int64_t out_row, int out_cs, void* stream);
int vsr_det_launch_nearest_view(const float* in, int64_t in_img, int64_t in_row, int in_cs, int N, int C, int Ho, int Wo, int s, float* out,
                                int64_t out_img, int64_t out_row, int out_cs, void* stream);
/* im2col_view: the kh x kw neighbourhoods (zero padded by pt, pl) of an NCHW map of C channels, C * kh * kw <= 32, as one 32-float chunk per
 * pixel of a view: channel (c * kh + ky) * kw + kx, zeros beyond */
int vsr_det_launch_im2col_view(const float* x, int N, int C, int H, int W, int kh, int kw, int pt, int pl, float* out, int64_t out_img, int64_t out_row,
                               int out_cs, void* stream);
int vsr_det_launch_dots_view(const float* in, int64_t in_img, int64_t in_row, int in_cs, int N, int C, int H, int W, const float* w, const float* bias,
                             int n_out, int act, float* out, void* stream);

/* ---------------------------------------------------------------------------------------
 * Scene cuts: the per-frame arithmetic of the ContentDetector pass that
 * --inpaint-mode propainter runs over the whole video (backend/tools/subtitle_detect.py:158-170 ->
 * backend/scenedetect/detectors/content_detector.py:138-172; frames down-scaled first by
 * scene_manager.py:499-504 = vsr_launch_resize_u8).  Integer, bit-exact.
 * ------------------------------------------------------------------------------------- */
/* cv2.cvtColor(img, COLOR_BGR2HSV) on uint8 (H in [0,180)), interleaved in, interleaved out (content_detector.py:147) */
int vsr_launch_bgr2hsv_u8(const uint8_t* bgr_dev, uint8_t* hsv_dev, int64_t npix, void* stream);
/* sums[i][c] = sum over pixels |hsv[i+1][p][c] - hsv[i][p][c]|, i < npairs, over npairs+1 consecutive interleaved frames
 * (_mean_pixel_distance before its division, content_detector.py:28-35); sums_dev is overwritten */
int vsr_launch_absdiff_sums_u8x3(const uint8_t* hsv_dev, int npairs, int64_t pix_per_frame, uint64_t* sums_dev, void* stream);

/* ---------------------------------------------------------------------------------------
 * Frame transport: colour conversion of raw planar video.  The reference gets BGR frames from
 * cv2.VideoCapture.read() (backend/inpaint/sttn_auto_inpaint.py:254-262, backend/main.py:171-176) and hands bgr24 frames to an
 * ffmpeg pipe that converts to yuv420p (backend/tools/video_io.py:54-81) -- libswscale on the CPU on both sides.  The *.y4m
 * reader / writer of backend/tools/video_io.py keep the planes as stored and convert on the GPU: 8-bit BT.601, 16.16 fixed point,
 * bit-exact against the numpy statement of the same matrices.  Integer, HBM-bound.
 * ------------------------------------------------------------------------------------- */
/* planes_dev: nframes records `frame_bytes` apart, each [Y: H*W][U: chroma_h*chroma_w][V: same] (chroma_w = W or (W+1)/2, chroma_h = H
 * or (H+1)/2, nearest replication; chroma_w = 0: luma only, grey output) -> bgr_dev uint8 [nframes][H][W][3] */
int vsr_io_yuv_to_bgr(const uint8_t* planes_dev, int64_t frame_bytes, int H, int W, int chroma_w, int chroma_h, int full_range,
                      uint8_t* bgr_dev, int nframes, void* stream);
/* bgr_dev uint8 [nframes][H][W][3] -> records [Y][U][V]; subsample_420: chroma = rounded mean of each 2x2 block (edges repeated) */
int vsr_io_bgr_to_yuv(const uint8_t* bgr_dev, int H, int W, int subsample_420, int full_range, uint8_t* planes_dev,
                      int64_t frame_bytes, int nframes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Plan introspection (host only, no GPU needed): the op list the engine runs for inpaint(L),
 * with symbolic buffers and the offset tables -- replayed on the CPU by tests/.
 * ------------------------------------------------------------------------------------- */
/* sub-kinds of op kind 6 (RAFT, csrc/flow_kernels.hip) */
enum { VSR_EW_IM2COL7_U8 = 1, VSR_EW_INORM_STATS = 2, VSR_EW_INORM_APPLY = 3, VSR_EW_CTX_SPLIT = 4, VSR_EW_FLOW_UPDATE = 5,
       VSR_EW_IM2COL7_FLOW = 6, VSR_EW_AVGPOOL2 = 7, VSR_EW_CORR_LOOKUP = 8, VSR_EW_GRU_RH = 9, VSR_EW_GRU_UPDATE = 10,
       VSR_EW_CONVEX_UP = 11,
       /* flow completion (csrc/rfc_plan.h) */
       VSR_EW_RFC_IM2COL5 = 20, VSR_EW_DEFORM_COLS = 21, VSR_EW_RFC_COMBINE = 22,
       /* ProPainter generator (csrc/pp_plan.h) */
       VSR_EW_PP_MASK_F32 = 30, VSR_EW_PP_IMGPROP = 31, VSR_EW_PP_COPY = 32, VSR_EW_PP_IM2COL3 = 33, VSR_EW_PP_DS_FLOW = 34,
       VSR_EW_PP_DS_MASK = 35, VSR_EW_PP_FEATPROP_PREP = 36, VSR_EW_PP_DEFORM_COLS = 37, VSR_EW_PP_LAYERNORM = 38, VSR_EW_PP_POOL = 39,
       VSR_EW_PP_FOLD = 40, VSR_EW_PP_UNFOLD_GELU = 41, VSR_EW_PP_TANH_OUT = 42 };
typedef struct VsrOpInfo {
    int32_t kind; /* 0 norm_im2col, 1 gemm, 2 softmax, 3 upsample2x, 4 decode_out, 5 reduce_scatter, 6 RAFT elementwise */
    int32_t nitems, tile_cfg, bmode;
    int32_t buf_src, buf_dst, H, W, C, halo_src, halo_dst, n, ldy, pix, t_frame_idx, t_first, premask;
    /* reduce_scatter: part = buf_src + off_src, out = buf_dst + off_dst, M, N, nsplit, tables */
    int32_t M, N, nsplit, t_rowC, t_colC;
    int32_t buf_mask; /* sttn-det: resized-mask byte buffer used by norm_im2col (premask) and decode_out, else -1 */
    int64_t off_src, off_dst, split_stride;
    double flops;
    char tag[32];
    /* kind 6: sub-kind (VSR_EW_*) and its generic operands: buffers, element offsets, int / float parameters */
    int32_t ew;
    int32_t ibuf[4];
    int64_t ioff[4];
    int32_t ipar[16];
    float fpar[4];
} VsrOpInfo;
typedef struct VsrGemmInfo {
    int32_t bufA, bufB, bufC, bufR;
    int64_t offA, offB, offC, offR, offBias;
    int32_t tRowA, tColA, tRowB, tColB, tRowC, tColC, tRowR;
    int32_t M, N, K, tilesM, tilesN, splitK, chunksPerSplit;
    int64_t splitStride;
    float alpha;
    int32_t act;
} VsrGemmInfo;
typedef struct VsrSoftmaxInfo {
    int32_t bufS, bufP;
    int64_t offS, offP, splitStride;
    int32_t M, N, ldS, ldP, nsplit;
    float scale;
} VsrSoftmaxInfo;

int vsr_plan_create(const vsr_sttn_t* h, int L, vsr_plan_t** out);
int vsr_plan_create_rows(const vsr_sttn_t* h, int L, int row_lo, int row_hi, vsr_plan_t** out);   /* the decoder on model rows [row_lo, row_hi) only */
int vsr_plan_create_box(const vsr_sttn_t* h, int L, int row_lo, int row_hi, int col_lo, int col_hi, vsr_plan_t** out);   /* ... and columns */
int vsr_raft_plan_create(const vsr_raft_t* h, int t, int H, int W, int iters, vsr_plan_t** out);
int vsr_rfc_plan_create(const vsr_rfc_t* h, int t, int H, int W, vsr_plan_t** out);
int vsr_pp_imgprop_plan_create(int t, int H, int W, vsr_plan_t** out);
int vsr_pp_gen_plan_create(const vsr_pp_t* h, int t, int lt, int H, int W, const uint8_t* window_flags, int nflags, vsr_plan_t** out);
/* mode: 0 the whole generator, 1 encoder + soft split of t frames only (vsr_pp_encode), 2 the generator on cached encoder features / tokens
 * (vsr_pp_forward_cached); see csrc/pp_plan.h PpPlanMode */
int vsr_pp_gen_plan_create_mode(const vsr_pp_t* h, int t, int lt, int H, int W, const uint8_t* window_flags, int nflags, int row_lo, int row_hi,
                                int col_lo, int col_hi, int mode, vsr_plan_t** out);
int vsr_pp_gen_plan_create_box(const vsr_pp_t* h, int t, int lt, int H, int W, const uint8_t* window_flags, int nflags, int row_lo, int row_hi,
                               int col_lo, int col_hi, vsr_plan_t** out);
int vsr_lama_plan_create(const vsr_lama_t* h, int B, int H, int W, vsr_plan_t** out);
int64_t vsr_plan_consts(const vsr_plan_t* p, float* out, int64_t capacity);   /* fp32 plan constants (buffer id -2), returns the count */
void vsr_plan_destroy(vsr_plan_t* p);
int vsr_plan_num_buffers(const vsr_plan_t* p);
int64_t vsr_plan_buffer_elems(const vsr_plan_t* p, int buf);
int vsr_plan_num_tables(const vsr_plan_t* p);
int64_t vsr_plan_table_len(const vsr_plan_t* p, int t);
int vsr_plan_table_copy(const vsr_plan_t* p, int t, int32_t* out);
int vsr_plan_num_ops(const vsr_plan_t* p);
int vsr_plan_op(const vsr_plan_t* p, int i, VsrOpInfo* out);
int vsr_plan_op_lane(const vsr_plan_t* p, int i);   /* stream the op is issued on (vsr_sttn_set_lanes; 0 for every other plan), -1 on a bad index */
int vsr_plan_op_gemm(const vsr_plan_t* p, int i, int j, VsrGemmInfo* out);
int vsr_plan_op_softmax(const vsr_plan_t* p, int i, int j, VsrSoftmaxInfo* out);
int vsr_plan_counts(const vsr_plan_t* p, int32_t* counts);
double vsr_plan_flops(const vsr_plan_t* p);

#ifdef __cplusplus
}
#endif
#endif /* VSR_HIP_H */
