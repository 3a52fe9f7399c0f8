// Device engines + C-ABI (include/vsr_hip.h, "RAFT" and "flow completion" sections) of the optical-flow stages of
// --inpaint-mode propainter: SURVEY.md section 8(a) rows a14 (RAFT_bi.forward, flow_comp_raft.py:39-55) and a15
// (RecurrentFlowCompleteNet.forward_bidirect_flow + combine_flow, recurrent_flow_completion.py:313-348).
//
// Same structure as sttn_engine.hip: a workspace allocated and zeroed once (NHWC fp32 activations whose physical zero
// halos are the conv padding), a plan (raft_plan.h / rfc_plan.h) materialised into device descriptors once per shape
// and replayed on the caller's stream without host synchronisation.  Exact fp32 arithmetic (v_mfma_f32_32x32x2_f32):
// the reference runs RAFT in fp32 even in its fp16 mode (propainter_inpaint.py:230).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>
#include "../../include/vsr_hip.h"
#include "elementwise.h"
#include "gather_gemm.h"
#include "plan_c.h"
#include "flow_kernels.h"
#include "raft_plan.h"
#include "rfc_plan.h"
#include "pp_kernels.h"
#include "pp_attn.h"
#include "pp_plan.h"
#include "lama_kernels.h"
#include "lama_plan.h"

using namespace vsr;

static int rfail(int code, const std::string& msg) { return vsr_internal_fail(code, msg.c_str()); }
#define HIPCHK(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return rfail(VSR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define RCCHK(expr)                                                                                        \
    do {                                                                                                   \
        int rc_ = (expr);                                                                                  \
        if (rc_ != 0) return rc_;                                                                          \
    } while (0)

namespace {

// ---- launch timing (bench / profile only): HIP events on the launch stream around every op of a replayed plan, keyed
// "<engine>:gg:<tileCfg>:<bmode>:v<variant>:<op tag>" (gather-GEMM launches) or "<engine>:op:<op tag>" (everything else).  Process-wide,
// off by default; vsr_flow_timing_get sums the records whose key starts with a prefix (resolving the pending events first).
struct FlowTimingRec { hipEvent_t a, b; std::string key; double flops; };
static int g_flowTiming = 0;
static std::mutex g_flowTimingMu;      // run_plan() runs on several host threads at once (generator lanes in the guarded arithmetics, batch lanes)
static std::vector<FlowTimingRec> g_flowPending;
static std::map<std::string, std::pair<double, std::pair<int64_t, double>>> g_flowTimed;

// VSR_PP_FLASH (default 1): the generator's window attention as ONE fused launch per transformer block (pp_attn_kernels.hip) instead of
// the plan's three ops QK^T / row softmax / P.V; 0 keeps the three launches (A/B runs, and the form tests/_replay_pp.py replays)
static bool pp_flash_enabled()
{
    static const bool on = [] { const char* e = getenv("VSR_PP_FLASH"); return !(e && atoi(e) == 0); }();
    return on;
}

struct FlowOpDev {
    const Op* op = nullptr;
    const void* dDesc = nullptr;   // GGProblem* (device)
    int nitems = 0, total = 0, nQueues = 1;
    // fused window attention: 1 = this op (the QK^T of an attention triple) launches the fused kernel instead, 2 = covered by it (skipped);
    // 3 / 4: the fold / unfold of an FFN pair whose GELU moved into the fold; 5: a GEMM of <= 4 output columns on the dot-product kernel
    int fused = 0;
    int narrowN = 0, narrowK = 0;  // fused == 5: the largest N / K of the op's problems
    bool thin = false;             // exact fp32: a short / few-tile GEMM on the one-workgroup-per-tile kernel (thin_variant below), same bits
    const void* dAttn = nullptr;   // PpAttnProblem* (device), fused == 1
    int attnItems = 0, attnTiles = 0;
    double attnFlops = 0;
};

// a plan on the device: tables, gather-GEMM descriptors with baked buffer pointers, tile queues
struct FlowPlanDev {
    std::unique_ptr<PlanIR> plan;
    int32_t* dTables = nullptr;
    std::vector<int64_t> toff;
    void* dDescs = nullptr;
    unsigned int* dQueues = nullptr;
    float* dConsts = nullptr;      // PlanIR::consts (BUF_PLAN_CONST)
    void* dAttn = nullptr;         // PpAttnProblem descriptors of the fused window-attention launches
    std::vector<FlowOpDev> ops;
    ~FlowPlanDev()
    {
        if (dAttn) (void)hipFree(dAttn);
        if (dConsts) (void)hipFree(dConsts);
        if (dTables) (void)hipFree(dTables);
        if (dDescs) (void)hipFree(dDescs);
        if (dQueues) (void)hipFree(dQueues);
    }
};

// workspace of one engine: buffer ids are the plan's; `bytes` marks the ids whose size is in bytes (u8 inputs)
struct Workspace {
    std::vector<void*> bufs;
    std::vector<int64_t> cap;
    std::vector<bool> bytes;
    int weights = 0;
    double* statAcc = nullptr;
    int64_t statAccCap = 0;
    const char* engine = "flow";   // name in the timing keys
    // arithmetic of the contractions: 0 = exact fp32 MFMA (gather_gemm_v3), 1 = split-half fp16 operands with fp32 accumulation
    // (gather_gemm_v4), 2 = fp16 operands (rounded in the kernel), fp32 accumulation (gather_gemm_v4<HI_ONLY>: the reference's GPU
    // arithmetic for flow completion and the generator).  1 and 2 are guarded by dRangeFlag: a call whose operands leave the fp16
    // range is redone in fp32 (`fallbacks` counts them)
    int precision = 0;
    unsigned int* dRangeFlag = nullptr;
    int64_t fallbacks = 0;
    // ensure_clean(): the per-frame layout the zero halos are currently valid for, and how far into every buffer (elements)
    std::string layout;
    std::vector<int64_t> clean;
    void init(int n, int weightsBuf, std::initializer_list<int> byteBufs)
    {
        bufs.assign(n, nullptr);
        cap.assign(n, 0);
        bytes.assign(n, false);
        for (int b : byteBufs) bytes[b] = true;
        weights = weightsBuf;
    }
    int64_t nbytes(int b, int64_t elems) const { return bytes[b] ? elems : elems * 4; }
    float* f(int b, int64_t off = 0) const { return (float*)bufs[b] + off; }
    bool needs_growth(const PlanIR& P) const
    {
        for (size_t b = 0; b < bufs.size(); ++b)
            if ((int)b != weights && P.bufElems[b] > cap[b]) return true;
        return false;
    }
    void release()
    {
        for (void*& p : bufs)
            if (p) { (void)hipFree(p); p = nullptr; }
        if (statAcc) { (void)hipFree(statAcc); statAcc = nullptr; }
        if (dRangeFlag) { (void)hipFree(dRangeFlag); dRangeFlag = nullptr; }
    }
};

} // namespace

static bool thin_variant(const Op& op, int totalTiles);

// grows the workspace to the plan's needs and bakes the plan (callers drop their cached plans first when it grows)
static int materialize(Workspace& ws, std::unique_ptr<PlanIR> plan, std::unique_ptr<FlowPlanDev>* out)
{
    std::unique_ptr<FlowPlanDev> pd(new FlowPlanDev);
    pd->plan = std::move(plan);
    const PlanIR& P = *pd->plan;
    const int nb = (int)ws.bufs.size();
    if (ws.needs_growth(P)) {
        for (int b = 0; b < nb; ++b) {
            if (b == ws.weights || P.bufElems[b] <= ws.cap[b]) continue;
            if (ws.bufs[b]) { HIPCHK(hipFree(ws.bufs[b])); ws.bufs[b] = nullptr; ws.cap[b] = 0; }
            const int64_t bytes = ws.nbytes(b, P.bufElems[b]);
            HIPCHK(hipMalloc(&ws.bufs[b], (size_t)bytes));
            HIPCHK(hipMemset(ws.bufs[b], 0, (size_t)bytes));   // zero halos, once
            ws.cap[b] = P.bufElems[b];
        }
        HIPCHK(hipDeviceSynchronize());
    }
    pd->toff.resize(P.tables.size());
    int64_t tot = 0;
    for (size_t i = 0; i < P.tables.size(); ++i) { pd->toff[i] = tot; tot += (int64_t)((P.tables[i].size() + 3) / 4 * 4); }
    std::vector<int32_t> flat((size_t)tot, 0);
    for (size_t i = 0; i < P.tables.size(); ++i)
        memcpy(flat.data() + pd->toff[i], P.tables[i].data(), P.tables[i].size() * sizeof(int32_t));
    HIPCHK(hipMalloc((void**)&pd->dTables, (size_t)(tot > 0 ? tot : 4) * sizeof(int32_t)));
    HIPCHK(hipMemcpy(pd->dTables, flat.data(), (size_t)tot * sizeof(int32_t), hipMemcpyHostToDevice));
    auto T = [&](int id) -> const int32_t* { return id < 0 ? nullptr : pd->dTables + pd->toff[id]; };
    if (!P.consts.empty()) {
        HIPCHK(hipMalloc((void**)&pd->dConsts, P.consts.size() * sizeof(float)));
        HIPCHK(hipMemcpy(pd->dConsts, P.consts.data(), P.consts.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    auto F = [&](int buf, int64_t off) -> float* { return buf == BUF_PLAN_CONST ? pd->dConsts + off : (buf < 0 ? nullptr : ws.f(buf, off)); };
    size_t descBytes = 0;
    for (const Op& op : P.ops)
        descBytes += (op.gemm.size() * sizeof(GGProblem) + 63) / 64 * 64 + (op.softmax.size() * sizeof(SMProblem) + 63) / 64 * 64;
    std::vector<char> hostDesc(descBytes + 64, 0);
    HIPCHK(hipMalloc(&pd->dDescs, descBytes + 64));
    size_t cursor = 0;
    for (const Op& op : P.ops) {
        FlowOpDev od;
        od.op = &op;
        if (op.kind == OP_GEMM) {
            GGProblem* hp = (GGProblem*)(hostDesc.data() + cursor);
            int tileStart = 0;
            for (size_t j = 0; j < op.gemm.size(); ++j) {
                const GemmItem& g = op.gemm[j];
                GGProblem& q = hp[j];
                q.A = F(g.bufA, g.offA); q.B = F(g.bufB, g.offB); q.C = F(g.bufC, g.offC);
                q.bias = g.offBias >= 0 ? F(ws.weights, g.offBias) : nullptr;
                q.R = g.bufR >= 0 ? F(g.bufR, g.offR) : nullptr;
                q.rowA = T(g.tRowA); q.colA = T(g.tColA); q.rowB = T(g.tRowB); q.colB = T(g.tColB);
                q.rowC = T(g.tRowC); q.colC = T(g.tColC); q.rowR = T(g.tRowR);
                q.M = g.M; q.N = g.N; q.K = g.K; q.tilesM = g.tilesM; q.tilesN = g.tilesN;
                q.splitK = g.splitK; q.chunksPerSplit = g.chunksPerSplit; q.tileStart = tileStart;
                q.act = g.act; q.alpha = g.alpha; q.splitStride = g.splitStride;
                tileStart += g.tilesM * g.tilesN * g.splitK;
            }
            od.dDesc = (char*)pd->dDescs + cursor;
            od.nitems = (int)op.gemm.size();
            od.total = tileStart;
            od.thin = thin_variant(op, tileStart);
            od.nQueues = 8;
            for (const GemmItem& g : op.gemm)
                if (g.tilesN > 4) od.nQueues = 1;
            cursor += (op.gemm.size() * sizeof(GGProblem) + 63) / 64 * 64;
        } else if (op.kind == OP_SOFTMAX) {
            SMProblem* hp = (SMProblem*)(hostDesc.data() + cursor);
            int rowStart = 0;
            for (size_t j = 0; j < op.softmax.size(); ++j) {
                const SoftmaxItem& sm = op.softmax[j];
                SMProblem& q = hp[j];
                q.S = F(sm.bufS, sm.offS); q.P = F(sm.bufP, sm.offP);
                q.M = sm.M; q.N = sm.N; q.ldS = sm.ldS; q.ldP = sm.ldP; q.nsplit = sm.nsplit; q.rowStart = rowStart;
                q.scale = sm.scale; q.flags = 0; q.splitStride = sm.splitStride;
                rowStart += (sm.M + 3) / 4 * 4;
            }
            od.dDesc = (char*)pd->dDescs + cursor;
            od.nitems = (int)op.softmax.size();
            od.total = rowStart;
            cursor += (op.softmax.size() * sizeof(SMProblem) + 63) / 64 * 64;
        }
        pd->ops.push_back(od);
    }
    HIPCHK(hipMemcpy(pd->dDescs, hostDesc.data(), descBytes, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&pd->dQueues, (pd->ops.size() + 1) * 8 * sizeof(unsigned int)));
    // ---- fused window attention: every [attn.qk, attn.softmax, attn.pv] triple of the generator's plan (pp_plan.cpp attention(): item j
    // of the three ops is the same (window, head) problem) becomes one launch of k_pp_flash_attn_f32.  The plan itself keeps the three
    // ops -- they are what the CPU replay executes and what the FLOP counts are taken from.
    if (pp_flash_enabled()) {
        std::vector<PpAttnProblem> all;
        struct Span { size_t op, first, count; int tiles; double flops; };
        std::vector<Span> spans;
        for (size_t i = 0; i + 2 < P.ops.size(); ++i) {
            const Op &qk = P.ops[i], &sm = P.ops[i + 1], &pv = P.ops[i + 2];
            if (qk.kind != OP_GEMM || sm.kind != OP_SOFTMAX || pv.kind != OP_GEMM || qk.tag != "attn.qk" || pv.tag != "attn.pv") continue;
            if (qk.gemm.size() != sm.softmax.size() || qk.gemm.size() != pv.gemm.size() || qk.gemm.empty()) continue;
            bool ok = qk.bmode == VSR_BMODE_NK && pv.bmode == VSR_BMODE_KN;
            for (size_t j = 0; ok && j < qk.gemm.size(); ++j) {
                const GemmItem &a = qk.gemm[j], &b = pv.gemm[j];
                ok = a.K == 128 && b.N == 128 && a.M == b.M && a.splitK == 1 && b.splitK == 1 && a.bufB == b.bufB && a.offBias < 0 && b.offBias < 0 &&
                     a.bufR < 0 && b.bufR < 0 && sm.softmax[j].nsplit == 1 && sm.softmax[j].N == a.N && a.act == VSR_ACT_NONE && b.act == VSR_ACT_NONE;
            }
            if (!ok) continue;
            // large problems first: their workgroups walk thousands of keys, the per-frame problems of the unmasked windows 45
            std::vector<size_t> order(qk.gemm.size());
            for (size_t j = 0; j < order.size(); ++j) order[j] = j;
            std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return qk.gemm[x].N > qk.gemm[y].N; });
            Span sp{i, all.size(), order.size(), 0, qk.flops + pv.flops};
            for (size_t j : order) {
                const GemmItem &a = qk.gemm[j], &b = pv.gemm[j];
                PpAttnProblem q;
                q.Q = F(a.bufA, a.offA); q.K = F(a.bufB, a.offB); q.V = F(b.bufB, b.offB); q.O = F(b.bufC, b.offC);
                q.qrow = T(a.tRowA); q.krow = T(a.tRowB); q.orow = T(b.tRowC);
                q.M = a.M; q.nk = a.N; q.tileStart = sp.tiles;
                q.scale = sm.softmax[j].scale * 1.4426950408889634f;
                sp.tiles += (a.M + 127) / 128;
                all.push_back(q);
            }
            spans.push_back(sp);
            i += 2;
        }
        if (!all.empty()) {
            HIPCHK(hipMalloc(&pd->dAttn, all.size() * sizeof(PpAttnProblem)));
            HIPCHK(hipMemcpy(pd->dAttn, all.data(), all.size() * sizeof(PpAttnProblem), hipMemcpyHostToDevice));
            for (const Span& sp : spans) {
                FlowOpDev& od = pd->ops[sp.op];
                od.fused = 1;
                od.dAttn = (const PpAttnProblem*)pd->dAttn + sp.first;
                od.attnItems = (int)sp.count; od.attnTiles = sp.tiles; od.attnFlops = sp.flops;
                pd->ops[sp.op + 1].fused = 2;
                pd->ops[sp.op + 2].fused = 2;
            }
        }
    }
    // ---- the FFN's fold -> unfold + GELU pair (pp_plan.cpp "tr.fold" / "tr.unfold"): the GELU once per map element inside the fold, the
    // unfold as a float4 copy (pp_gen_kernels.hip).  VSR_PP_FOLD_GELU=0 keeps the plan's two kernels.
    static const bool foldGelu = [] { const char* e = getenv("VSR_PP_FOLD_GELU"); return !(e && atoi(e) == 0); }();
    if (foldGelu) {
        for (size_t i = 0; i + 1 < P.ops.size(); ++i) {
            const Op &a = P.ops[i], &b = P.ops[i + 1];
            if (a.kind != OP_EW || b.kind != OP_EW || a.ew != EW_PP_FOLD || b.ew != EW_PP_UNFOLD_GELU) continue;
            // same map: the fold's output buffer, frames, size and channels are the unfold's input; no halo; C and ld multiples of 4
            if (a.ibuf[1] != b.ibuf[0] || a.ipar[7] != 0 || a.ipar[1] != b.ipar[0] || a.ipar[4] != b.ipar[3] || a.ipar[5] != b.ipar[4] ||
                a.ipar[6] != b.ipar[5] || b.ipar[5] % 4 || b.ipar[6] % 4)
                continue;
            pd->ops[i].fused = 3;
            pd->ops[i + 1].fused = 4;
        }
    }
    // ---- GEMMs of at most four output columns (the last conv of a head: RAFT "upd.fh2", flow completion "dec.u1", generator "dec.6"):
    // the dot-product kernel of gather_gemm_narrow.h instead of the 256 x 32 MFMA tile, in every arithmetic mode (it is exact fp32).
    // VSR_GG_NARROW=0 keeps the plan's tile.
    static const bool narrow = [] { const char* e = getenv("VSR_GG_NARROW"); return !(e && atoi(e) == 0); }();
    if (narrow) {
        for (size_t i = 0; i < P.ops.size(); ++i) {
            const Op& op = P.ops[i];
            if (op.kind != OP_GEMM || op.bmode != VSR_BMODE_NK || op.tileCfg != VSR_TILE_256x32 || op.gemm.empty() || pd->ops[i].fused) continue;
            bool ok = true;
            int maxN = 0, maxK = 0;
            for (const GemmItem& g : op.gemm) {
                ok = ok && g.N <= 4 && g.tilesN == 1 && g.splitK == 1 && g.bufR < 0 && g.K % VSR_GG_KC == 0 &&
                     (int64_t)(g.N <= 2 ? 2 : 4) * g.K <= vsr_gg_narrow_cap() &&
                     (g.act == VSR_ACT_NONE || g.act == VSR_ACT_LRELU02 || g.act == VSR_ACT_RELU || g.act == VSR_ACT_LRELU01);
                maxN = std::max(maxN, g.N);
                maxK = std::max(maxK, g.K);
            }
            if (!ok || (int64_t)(maxN <= 2 ? 2 : 4) * maxK > vsr_gg_narrow_cap()) continue;
            pd->ops[i].fused = 5;
            pd->ops[i].narrowN = maxN;
            pd->ops[i].narrowK = maxK;
        }
    }
    *out = std::move(pd);
    return 0;
}

// Exact fp32: the persistent LDS-DMA kernel (variant 3) pays a fixed cost per tile (queue fetch, pipeline fill, the prologue of the next
// tile) that long-K problems amortise and short ones do not; the one-workgroup-per-tile kernel (variant 1) runs the same MFMA sequence per
// output element -- the same bits, tests/test_gpu_kernels.py::test_one_workgroup_per_tile_kernel_equals_the_persistent_one -- without it.
// The rule follows the detector plan's (ocr_det_nhwc.thin_variant): K <= 256, or at most VSR_FLOW_THIN_TILES (768 = one round of three
// workgroups per CU) tiles with K <= VSR_FLOW_THIN_K (2600).  Measured per op tag on a 68-frame propainter batch
// (profiles/r06c_flow_thin_ab.log): the feature-propagation convs of the generator and of flow completion -- eight DEPENDENT single-round
// launches per frame and direction, M = 43 200, N = 128, K = 1152 / 2592 -- run at 92-106 TF instead of 65-88 (-120 ms of kernel time per
// batch), everything with more tiles or longer K (RAFT's GRU, the token GEMMs, the encoder) loses 5-12 % on variant 1 and stays.
// VSR_FLOW_THIN=0: every GEMM on variant 3 (round 5's behaviour); =2: every GEMM on variant 1 (A/B).
static int flow_thin_mode() { static const int m = [] { const char* e = getenv("VSR_FLOW_THIN"); return e ? atoi(e) : 1; }(); return m; }
static bool thin_variant(const Op& op, int totalTiles)
{
    static const int maxTiles = [] { const char* e = getenv("VSR_FLOW_THIN_TILES"); return e ? atoi(e) : 768; }();
    static const int maxK = [] { const char* e = getenv("VSR_FLOW_THIN_K"); return e ? atoi(e) : 2600; }();
    static const int shortK = [] { const char* e = getenv("VSR_FLOW_THIN_SHORT_K"); return e ? atoi(e) : 256; }();
    if (flow_thin_mode() == 0 || op.kind != OP_GEMM || op.gemm.empty()) return false;
    if (flow_thin_mode() == 2) return true;
    int K = 0;
    for (const GemmItem& g : op.gemm) K = std::max(K, g.splitK > 1 ? g.chunksPerSplit * VSR_GG_KC : g.K);
    return K <= shortK || (totalTiles <= maxTiles && K <= maxK);
}

// replays a materialised plan; `bgr`: channel order of RAFT's u8 input frames
static int run_plan(const Workspace& ws, FlowPlanDev* pd, int bgr, hipStream_t stream)
{
    const int variant = ws.precision == 1 ? 4 : (ws.precision == 2 ? 7 : 3);
    unsigned int* rangeFlag = ws.precision != 0 ? ws.dRangeFlag : nullptr;
    HIPCHK(hipMemsetAsync(pd->dQueues, 0, (pd->ops.size() + 1) * 8 * sizeof(unsigned int), stream));
    auto B = [&](int buf, int64_t off) -> float* { return ws.f(buf, off); };
    size_t idx = 0;
    for (const FlowOpDev& od : pd->ops) {
        const Op& op = *od.op;
        unsigned int* queue = pd->dQueues + 8 * idx++;
        int rc = 0;
        if (od.fused == 2) continue;                     // part of a fused attention launch
        const int opVariant = (variant == 3 && od.thin) ? 1 : variant;
        FlowTimingRec tr;
        if (g_flowTiming) {
            if (od.fused == 1) tr.key = std::string(ws.engine) + ":gg:flash:0:v" + (ws.precision == 2 ? "7" : "3") + ":attn.flash";
            else if (od.fused == 5) tr.key = std::string(ws.engine) + ":gg:narrow:0:v3:" + op.tag;
            else
            tr.key = std::string(ws.engine) + (op.kind == OP_GEMM ? ":gg:" + std::to_string(op.tileCfg) + ":" + std::to_string(op.bmode) + ":v" +
                                                                        std::to_string(opVariant) + ":" : ":op:") + op.tag;
            tr.flops = od.fused == 1 ? od.attnFlops : op.flops;
            HIPCHK(hipEventCreate(&tr.a));
            HIPCHK(hipEventCreate(&tr.b));
            HIPCHK(hipEventRecord(tr.a, stream));
        }
        if (od.fused == 1) {
            // fp16 operands in the fp16-operand mode; the split-half mode (operands as hi / lo pairs) keeps the exact kernel
            rc = vsr_pp_launch_flash_attn((const PpAttnProblem*)od.dAttn, od.attnItems, od.attnTiles, ws.precision == 2 ? 1 : 0, rangeFlag, stream);
        } else if (od.fused == 5) {
            rc = vsr_launch_gather_gemm_narrow_dev((const GGProblem*)od.dDesc, od.nitems, od.total, od.narrowN, od.narrowK, stream);
        } else if (op.kind == OP_GEMM) {
            rc = vsr_launch_gather_gemm_dev((const GGProblem*)od.dDesc, od.nitems, od.total, op.tileCfg, op.bmode, queue, opVariant, od.nQueues,
                                            rangeFlag, stream);
        } else if (op.kind == OP_SOFTMAX) {
            rc = vsr_launch_softmax_dev((const SMProblem*)od.dDesc, od.nitems, od.total, stream);
        } else if (op.kind == OP_UPSAMPLE2X) {
            rc = vsr_launch_upsample2x(B(op.bufSrc, 0), op.H, op.W, op.C, op.haloS, B(op.bufDst, 0), op.haloD, op.n, stream);
        } else if (op.kind == OP_EW) {
            const int* ip = op.ipar;
            switch (op.ew) {
            case EW_IM2COL7_U8:
                rc = vsr_raft_launch_im2col7_u8((const uint8_t*)ws.bufs[op.ibuf[0]], ip[0], ip[1], ip[2], bgr, B(op.ibuf[1], 0), stream);
                break;
            case EW_INORM_STATS:
                rc = vsr_raft_launch_inorm_stats(B(op.ibuf[0], 0), ip[0], ip[1], ip[2], ip[3], ip[4], ws.statAcc, B(op.ibuf[1], 0), stream);
                break;
            case EW_INORM_APPLY:
                rc = vsr_raft_launch_inorm_apply(B(op.ibuf[0], 0), ip[0], ip[1], ip[2], ip[3], ip[4], B(op.ibuf[1], 0), ip[5],
                                                 op.ibuf[2] >= 0 ? B(op.ibuf[2], 0) : nullptr, ip[6], stream);
                break;
            case EW_CTX_SPLIT:
                rc = vsr_raft_launch_ctx_split(B(op.ibuf[0], 0), pd->dTables + pd->toff[ip[5]], ip[0], ip[1], ip[2], ip[3], ip[4],
                                               B(op.ibuf[1], 0), stream);
                break;
            case EW_FLOW_UPDATE:
                rc = vsr_raft_launch_flow_update(B(op.ibuf[0], 0), ip[7], B(op.ibuf[1], 0), B(op.ibuf[2], 0), B(op.ibuf[3], 0), ip[0], ip[1],
                                                 ip[2], ip[3], ip[4], ip[5], ip[6], stream);
                break;
            case EW_IM2COL7_FLOW:
                rc = vsr_raft_launch_im2col7_flow(B(op.ibuf[0], 0), ip[0], ip[1], ip[2], B(op.ibuf[1], 0), stream);
                break;
            case EW_AVGPOOL2:
                rc = vsr_raft_launch_avgpool2(B(op.ibuf[0], op.ioff[0]), ip[0], ip[1], ip[2], B(op.ibuf[0], op.ioff[1]), stream);
                break;
            case EW_CORR_TRANSPOSE:
                rc = vsr_raft_launch_corr_transpose(B(op.ibuf[0], op.ioff[0]), B(op.ibuf[0], op.ioff[1]), ip[0], ip[1], stream);
                break;
            case EW_CORR_LOOKUP: {
                const float* lv[4];
                for (int l = 0; l < 4; ++l) lv[l] = B(op.ibuf[0], op.ioff[l]);
                rc = vsr_raft_launch_corr_lookup(lv, ip + 1, ip + 5, B(op.ibuf[1], 0), ip[0], ip[9], B(op.ibuf[2], 0), stream);
                break;
            }
            case EW_GRU_RH:
                rc = vsr_raft_launch_gru_rh(B(op.ibuf[0], 0), B(op.ibuf[1], 0), ip[0], ip[1], ip[2], ip[3], ip[4], ip[5], ip[6], stream);
                break;
            case EW_GRU_UPDATE:
                rc = vsr_raft_launch_gru_update(B(op.ibuf[0], 0), B(op.ibuf[1], 0), B(op.ibuf[2], 0), ip[0], ip[1], ip[2], ip[3], ip[4], ip[5],
                                                stream);
                break;
            case EW_CONVEX_UP:
                rc = vsr_raft_launch_convex_up(B(op.ibuf[0], 0), B(op.ibuf[1], 0), ip[0], ip[1], ip[2], B(op.ibuf[2], 0), stream);
                break;
            case EW_RFC_IM2COL5:
                rc = vsr_rfc_launch_im2col5(B(op.ibuf[0], 0), B(op.ibuf[1], 0), (const uint8_t*)ws.bufs[op.ibuf[2]], ip[0], ip[1], ip[2],
                                            B(op.ibuf[3], 0), stream);
                break;
            case EW_DEFORM_COLS:
                rc = vsr_rfc_launch_deform_cols(B(op.ibuf[0], op.ioff[0]), B(op.ibuf[0], op.ioff[1]), B(op.ibuf[1], 0), ip[5], op.fpar[0], ip[0],
                                                ip[1], ip[2], ip[3], ip[4], B(op.ibuf[2], 0), stream);
                break;
            case EW_RFC_COMBINE:
                rc = vsr_rfc_launch_combine(B(op.ibuf[0], 0), ip[3], B(FB_IN_FLOW_F, 0), B(FB_IN_FLOW_B, 0), (const uint8_t*)ws.bufs[FB_IN_MASK],
                                            ip[0], ip[1], ip[2], B(op.ibuf[1], 0), B(op.ibuf[2], 0), stream);
                break;
            case EW_PP_MASK_F32:
                rc = vsr_pp_launch_mask_f32((const uint8_t*)ws.bufs[op.ibuf[0]], ip[0], B(op.ibuf[1], 0), stream);
                break;
            case EW_PP_IMGPROP: {
                // ipar: C, h, w, first, frame, previous frame, flow index, direction (pp_plan.cpp)
                const int64_t hw = (int64_t)ip[1] * ip[2], fe = (int64_t)ip[0] * hw;
                const float* fprop = B(ip[7] == 0 ? PB_IN_FLOW_F : PB_IN_FLOW_B, (int64_t)ip[6] * 2 * hw);
                const float* fcheck = B(ip[7] == 0 ? PB_IN_FLOW_B : PB_IN_FLOW_F, (int64_t)ip[6] * 2 * hw);
                rc = vsr_pp_launch_imgprop(B(op.ibuf[2], ip[5] * fe), B(op.ibuf[3], ip[5] * hw), B(op.ibuf[0], ip[4] * fe), B(op.ibuf[1], ip[4] * hw),
                                           fprop, fcheck, ip[0], ip[1], ip[2], ip[3], B(op.ibuf[2], ip[4] * fe), B(op.ibuf[3], ip[4] * hw), stream);
                break;
            }
            case EW_PP_IM2COL3:
                rc = vsr_pp_launch_im2col3(B(op.ibuf[0], 0), (const uint8_t*)ws.bufs[op.ibuf[1]], (const uint8_t*)ws.bufs[op.ibuf[2]], ip[0], ip[1],
                                           ip[2], B(op.ibuf[3], 0), stream);
                break;
            case EW_PP_DS_FLOW:
                rc = vsr_pp_launch_ds_flow(B(op.ibuf[0], 0), ip[0], ip[1], ip[2], B(op.ibuf[1], 0), stream);
                break;
            case EW_PP_DS_MASK:
                rc = vsr_pp_launch_ds_mask((const uint8_t*)ws.bufs[op.ibuf[0]], (const uint8_t*)ws.bufs[op.ibuf[1]], ip[0], ip[1], ip[2],
                                           B(op.ibuf[2], op.ioff[2]), ip[3], ip[4], stream);
                break;
            case EW_PP_FEATPROP_PREP: {
                // ipar: h, w, halo, C, warped slot, misc slot ; ioff: propagated feature, flow (prop), flow (check), mask slot
                const int64_t fe = (int64_t)(ip[0] + 2 * ip[2]) * (ip[1] + 2 * ip[2]) * ip[3];
                rc = vsr_pp_launch_featprop_prep(B(op.ibuf[0], op.ioff[0]), B(op.ibuf[1], op.ioff[1]), B(op.ibuf[2], op.ioff[2]),
                                                 B(op.ibuf[0], op.ioff[3]), ip[0], ip[1], ip[2], ip[3], B(op.ibuf[0], ip[4] * fe),
                                                 B(op.ibuf[0], ip[5] * fe), stream);
                break;
            }
            case EW_PP_DEFORM_COLS:
                rc = vsr_pp_launch_deform_cols(B(op.ibuf[0], op.ioff[0]), B(op.ibuf[1], 0), ip[4], B(op.ibuf[2], op.ioff[2]), op.fpar[0], ip[0], ip[1],
                                               ip[2], ip[3], B(op.ibuf[3], 0), stream);
                break;
            case EW_PP_LAYERNORM:
                rc = vsr_pp_launch_layernorm(B(op.ibuf[0], 0), B(ws.weights, op.ioff[0]), B(ws.weights, op.ioff[1]), ip[0], ip[1], ip[2], ip[3], ip[4],
                                             ip[5], B(op.ibuf[1], 0), stream);
                break;
            case EW_PP_POOL:
                rc = vsr_pp_launch_pool(B(op.ibuf[0], 0), B(ws.weights, op.ioff[2]), B(ws.weights, op.ioff[3]), ip[0], ip[1], ip[2], ip[3], ip[4],
                                        ip[5], B(op.ibuf[0], op.ioff[1]), stream);
                break;
            case EW_PP_FOLD:
                rc = (od.fused == 3 ? vsr_pp_launch_fold_gelu : vsr_pp_launch_fold)(B(op.ibuf[0], 0), ip[0], ip[1], ip[2], ip[3], ip[4], ip[5], ip[6],
                                                                                    ip[7], ip[8], B(op.ibuf[1], 0), stream);
                break;
            case EW_PP_UNFOLD_GELU:
                rc = (od.fused == 4 ? vsr_pp_launch_unfold_plain : vsr_pp_launch_unfold_gelu)(B(op.ibuf[0], 0), ip[0], ip[1], ip[2], ip[3], ip[4], ip[5],
                                                                                              ip[6], B(op.ibuf[1], 0), stream);
                break;
            case EW_PP_TANH_OUT:
                rc = vsr_pp_launch_tanh_out(B(op.ibuf[0], 0), ip[0], ip[1], ip[2], ip[3], B(op.ibuf[1], 0), stream);
                break;
            case EW_LAMA_IM2COL7:
                rc = vsr_lama_launch_im2col7((const uint8_t*)ws.bufs[op.ibuf[0]], (const uint8_t*)ws.bufs[op.ibuf[1]], ip[0], ip[1], ip[2], ip[3], ip[4],
                                             B(op.ibuf[2], 0), stream);
                break;
            case EW_LAMA_HALO:
                rc = vsr_lama_launch_halo(B(op.ibuf[0], 0), ip[0], ip[1], ip[2], ip[3], ip[4], stream);
                break;
            case EW_LAMA_ADD_HALO:
                rc = vsr_lama_launch_add_halo(B(op.ibuf[0], 0), B(op.ibuf[1], 0), B(op.ibuf[2], 0), ip[0], ip[1], ip[2], ip[3], ip[4], ip[5], stream);
                break;
            case EW_LAMA_OUT:
                rc = vsr_lama_launch_out(B(op.ibuf[0], 0), (const uint8_t*)ws.bufs[op.ibuf[1]], (const uint8_t*)ws.bufs[op.ibuf[2]], ip[0], ip[1], ip[2],
                                         ip[3], ip[4], (uint8_t*)ws.bufs[op.ibuf[3]], stream);
                break;
            default:
                return rfail(VSR_ERR_STATE, "unknown elementwise op");
            }
        } else {
            return rfail(VSR_ERR_STATE, "unexpected op kind in a flow plan");
        }
        if (rc != 0) return rfail(VSR_ERR_HIP, "kernel launch failed: " + op.tag + ": " + hipGetErrorString(hipGetLastError()));
        if (g_flowTiming) {
            HIPCHK(hipEventRecord(tr.b, stream));
            std::lock_guard<std::mutex> lk(g_flowTimingMu);
            g_flowPending.push_back(tr);
        }
    }
    return 0;
}

// kernels write interiors only and rely on zero halos: another frame size or count moves the halos, so the workspace
// is cleared when the geometry changes (never in steady state)
// `plan`: clear what THAT plan addresses (its bufElems) instead of every buffer's capacity.  The generator's workspace is sized by its
// largest plan (the encoder over all 68 frames of a batch: tens of GB) while the plans that alternate -- the sliding windows, whose
// frame counts differ by a reference frame or two -- touch a fraction of it: clearing capacities cost 300 ms per 68-frame batch
// (2688 fills of up to 13.9 ms, profiles/r05_propainter_f32_kernel_stats.csv).  A plan never reads beyond its own bufElems.
static int clear_workspace(Workspace& ws, hipStream_t stream, const PlanIR* plan = nullptr)
{
    for (size_t b = 0; b < ws.bufs.size(); ++b) {
        if ((int)b == ws.weights || !ws.bufs[b]) continue;
        int64_t n = ws.cap[b];
        if (plan && b < plan->bufElems.size() && plan->bufElems[b] < n) n = plan->bufElems[b];
        if (n > 0) HIPCHK(hipMemsetAsync(ws.bufs[b], 0, (size_t)ws.nbytes((int)b, n), stream));
    }
    return 0;
}

// The generator's variant of the rule above.  Its plans for one (kind of call, lt, H, W) differ in the NUMBER of frames only (the
// sliding windows of a batch carry 5 or 6 reference frames: t alternates), and every frame-indexed buffer is laid out frame after
// frame with the same per-frame geometry: a plan with more frames reaches further into the buffers, it does not move a halo.  So the
// workspace keeps, per buffer, the extent that has been cleared since the layout last changed, and a call clears only what its plan
// addresses beyond that -- nothing at all in steady state.  (Clearing on every change of t was 1 890 fills and 66 ms per 68-frame
// batch, profiles/r05_propainter_f32_kernel_stats_after.csv.)  Buffers that grew were zeroed by materialize().
static int ensure_clean(Workspace& ws, const std::string& layout, const PlanIR& plan, hipStream_t stream)
{
    if (ws.clean.size() != ws.bufs.size()) ws.clean.assign(ws.bufs.size(), 0);
    if (ws.layout != layout) {
        ws.layout = layout;
        std::fill(ws.clean.begin(), ws.clean.end(), 0);
    }
    for (size_t b = 0; b < ws.bufs.size(); ++b) {
        if ((int)b == ws.weights || !ws.bufs[b] || b >= plan.bufElems.size()) continue;
        const int64_t need = plan.bufElems[b] < ws.cap[b] ? plan.bufElems[b] : ws.cap[b];
        if (need <= ws.clean[b]) continue;
        const int64_t from = ws.clean[b];
        HIPCHK(hipMemsetAsync((char*)ws.bufs[b] + ws.nbytes((int)b, from), 0, (size_t)ws.nbytes((int)b, need - from), stream));
        ws.clean[b] = need;
    }
    return 0;
}

static int range_guard_arm(Workspace& ws, hipStream_t stream)
{
    if (ws.precision == 0) return 0;
    if (!ws.dRangeFlag) HIPCHK(hipMalloc(&ws.dRangeFlag, sizeof(unsigned int)));
    HIPCHK(hipMemsetAsync(ws.dRangeFlag, 0, sizeof(unsigned int), stream));
    return 0;
}

// split-half mode only: waits for the plan and reports whether a contraction saw a non-finite accumulator
static int range_guard_fired(Workspace& ws, hipStream_t stream, bool* fired)
{
    *fired = false;
    if (ws.precision == 0) return 0;
    unsigned int v = 0;
    HIPCHK(hipMemcpyAsync(&v, ws.dRangeFlag, sizeof(v), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    if (v) { *fired = true; ws.fallbacks++; }
    return 0;
}

static int set_precision(Workspace& ws, int mode)
{
    if (mode < 0 || mode > 2)
        return rfail(VSR_ERR_ARG, "precision must be 0 (exact fp32), 1 (split-half fp16 operands, fp32 accumulate) or 2 (fp16 operands, fp32 accumulate)");
    ws.precision = mode;
    return 0;
}

static int upload_weights(Workspace& ws, const std::vector<float>& packed, int device)
{
    if (device >= vsr_device_count()) return rfail(VSR_ERR_NOGPU, "no such HIP device; there is no CPU fallback");
    HIPCHK(hipSetDevice(device));
    const size_t bytes = packed.size() * sizeof(float);
    HIPCHK(hipMalloc(&ws.bufs[ws.weights], bytes));
    HIPCHK(hipMemcpy(ws.bufs[ws.weights], packed.data(), bytes, hipMemcpyHostToDevice));
    ws.cap[ws.weights] = (int64_t)packed.size();
    return 0;
}

static int read_buffer(const Workspace& ws, int device, int buf, int64_t offset, int64_t count, float* out_host)
{
    if (!out_host || buf < 0 || buf >= (int)ws.bufs.size() || ws.bytes[buf] || offset < 0 || count < 0) return rfail(VSR_ERR_ARG, "bad argument");
    if (device < 0 || !ws.bufs[buf] || offset + count > ws.cap[buf]) return rfail(VSR_ERR_STATE, "buffer not allocated / range outside it");
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out_host, ws.f(buf, offset), (size_t)count * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

// ---------------------------------------------------------------------------------------
// RAFT
// ---------------------------------------------------------------------------------------
struct vsr_raft {
    RaftModel model;
    int device = -1;
    bool finalized = false;
    Workspace ws;
    std::tuple<int, int, int> geom{0, 0, 0};   // (t, H, W) the halos of the workspace are currently laid out for
    std::map<std::tuple<int, int, int, int>, std::unique_ptr<FlowPlanDev>> plans;
    vsr_raft() { ws.init(RB_COUNT, RB_WEIGHTS, {RB_IN_U8}); ws.engine = "raft"; }
};

static int raft_plan_dev(vsr_raft* h, int t, int H, int W, int iters, FlowPlanDev** out)
{
    const auto key = std::make_tuple(t, H, W, iters);
    auto it = h->plans.find(key);
    if (it != h->plans.end()) { *out = it->second.get(); return 0; }
    std::unique_ptr<PlanIR> plan;
    try {
        plan.reset(new RaftPlan(h->model, t, H, W, iters));
    } catch (const std::exception& e) {
        return rfail(VSR_ERR_ARG, std::string("raft plan: ") + e.what());
    }
    const int64_t accNeed = (int64_t)t * 256 * 2;
    if (h->ws.statAccCap < accNeed) {
        if (h->ws.statAcc) HIPCHK(hipFree(h->ws.statAcc));
        HIPCHK(hipMalloc((void**)&h->ws.statAcc, (size_t)accNeed * sizeof(double)));
        h->ws.statAccCap = accNeed;
    }
    if (h->ws.needs_growth(*plan)) h->plans.clear();   // pointers baked into the cached plans die with the old buffers
    std::unique_ptr<FlowPlanDev> pd;
    RCCHK(materialize(h->ws, std::move(plan), &pd));
    *out = pd.get();
    h->plans[key] = std::move(pd);
    return 0;
}

// ---------------------------------------------------------------------------------------
// recurrent flow completion
// ---------------------------------------------------------------------------------------
struct vsr_rfc {
    RfcModel model;
    int device = -1;
    bool finalized = false;
    Workspace ws;
    std::tuple<int, int, int> geom{0, 0, 0};
    std::map<std::tuple<int, int, int>, std::unique_ptr<FlowPlanDev>> plans;
    vsr_rfc() { ws.init(FB_COUNT, FB_WEIGHTS, {FB_IN_MASK}); ws.engine = "rfc"; }
};

static int rfc_plan_dev(vsr_rfc* h, int t, int H, int W, FlowPlanDev** out)
{
    const auto key = std::make_tuple(t, H, W);
    auto it = h->plans.find(key);
    if (it != h->plans.end()) { *out = it->second.get(); return 0; }
    std::unique_ptr<PlanIR> plan;
    try {
        plan.reset(new RfcPlan(h->model, t, H, W));
    } catch (const std::exception& e) {
        return rfail(VSR_ERR_ARG, std::string("flow-completion plan: ") + e.what());
    }
    if (h->ws.needs_growth(*plan)) h->plans.clear();
    std::unique_ptr<FlowPlanDev> pd;
    RCCHK(materialize(h->ws, std::move(plan), &pd));
    *out = pd.get();
    h->plans[key] = std::move(pd);
    return 0;
}

// ---------------------------------------------------------------------------------------
// ProPainter generator
// ---------------------------------------------------------------------------------------
struct vsr_pp {
    PpModel model;
    bool finalized = false;
    int device = -1;
    Workspace ws;
    std::string geom;                          // shape key the halos of the workspace are currently laid out for
    std::map<std::tuple<int, int, int>, std::unique_ptr<FlowPlanDev>> imgPlans;
    std::map<std::string, std::unique_ptr<FlowPlanDev>> genPlans;
    vsr_pp() { ws.init(PB_COUNT, PB_WEIGHTS, {PB_IN_MASK_U8, PB_IN_MASK_UPD_U8, PB_OUT_MASK_U8}); ws.engine = "pp"; }
};

// ---------------------------------------------------------------------------------------
// LaMa
// ---------------------------------------------------------------------------------------
struct vsr_lama {
    LamaModel model;
    int device = -1;
    bool finalized = false;
    Workspace ws;
    std::tuple<int, int, int> geom{0, 0, 0};
    std::map<std::tuple<int, int, int>, std::unique_ptr<FlowPlanDev>> plans;
    vsr_lama() { ws.init(LB_COUNT, LB_WEIGHTS, {LB_IN_U8, LB_MASK_U8, LB_OUT_U8}); ws.engine = "lama"; }
};

static int lama_plan_dev(vsr_lama* h, int B, int H, int W, FlowPlanDev** out)
{
    const auto key = std::make_tuple(B, H, W);
    auto it = h->plans.find(key);
    if (it != h->plans.end()) { *out = it->second.get(); return 0; }
    std::unique_ptr<PlanIR> plan;
    try {
        plan.reset(new LamaPlan(h->model, B, H, W));
    } catch (const std::exception& e) {
        return rfail(VSR_ERR_ARG, std::string("lama plan: ") + e.what());
    }
    if (h->ws.needs_growth(*plan)) h->plans.clear();
    std::unique_ptr<FlowPlanDev> pd;
    RCCHK(materialize(h->ws, std::move(plan), &pd));
    *out = pd.get();
    h->plans[key] = std::move(pd);
    return 0;
}

extern "C" {

int vsr_flow_timing(int enable)
{
    g_flowTiming = enable ? 1 : 0;
    return 0;
}

static int flow_timing_collect()
{
    std::lock_guard<std::mutex> lk(g_flowTimingMu);
    if (g_flowPending.empty()) return 0;
    HIPCHK(hipDeviceSynchronize());
    for (FlowTimingRec& tr : g_flowPending) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, tr.a, tr.b));
        auto& acc = g_flowTimed[tr.key];
        acc.first += ms;
        acc.second.first += 1;
        acc.second.second += tr.flops;
        (void)hipEventDestroy(tr.a);
        (void)hipEventDestroy(tr.b);
    }
    g_flowPending.clear();
    return 0;
}

int vsr_flow_timing_reset(void)
{
    RCCHK(flow_timing_collect());
    std::lock_guard<std::mutex> lk(g_flowTimingMu);
    g_flowTimed.clear();
    return 0;
}

int vsr_flow_timing_get(const char* prefix, double* total_ms, int64_t* launches, double* flops)
{
    if (!prefix) return rfail(VSR_ERR_ARG, "bad argument");
    RCCHK(flow_timing_collect());
    std::lock_guard<std::mutex> lk(g_flowTimingMu);
    double ms = 0, fl = 0;
    int64_t n = 0;
    const size_t pl = strlen(prefix);
    for (const auto& kv : g_flowTimed)
        if (kv.first.compare(0, pl, prefix) == 0) { ms += kv.second.first; n += kv.second.second.first; fl += kv.second.second.second; }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    if (flops) *flops = fl;
    return 0;
}

/* keys recorded so far, '\n'-separated, into buf (returns the length needed incl. the terminator) */
int64_t vsr_flow_timing_keys(char* buf, int64_t capacity)
{
    if (flow_timing_collect() != 0) return -1;
    std::lock_guard<std::mutex> lk(g_flowTimingMu);
    std::string all;
    for (const auto& kv : g_flowTimed) { all += kv.first; all += '\n'; }
    if (buf && capacity > 0) {
        const size_t n = all.size() < (size_t)capacity - 1 ? all.size() : (size_t)capacity - 1;
        memcpy(buf, all.data(), n);
        buf[n] = 0;
    }
    return (int64_t)all.size() + 1;
}

int vsr_lama_create(vsr_lama_t** out)
{
    if (!out) return rfail(VSR_ERR_ARG, "null out pointer");
    *out = new vsr_lama;
    return 0;
}

int vsr_lama_set_param(vsr_lama_t* h, const char* key, const float* data, const int64_t* shape, int ndim)
{
    if (!h || !key || !data || !shape || ndim < 0 || ndim > 8) return rfail(VSR_ERR_ARG, "bad argument");
    if (h->finalized) return rfail(VSR_ERR_STATE, "model already finalized");
    std::string err;
    if (!h->model.set_param(key, data, shape, ndim, err)) return rfail(VSR_ERR_ARG, err);
    return 0;
}

int vsr_lama_finalize(vsr_lama_t* h, int device)
{
    if (!h) return rfail(VSR_ERR_ARG, "null handle");
    if (h->finalized) return rfail(VSR_ERR_STATE, "model already finalized");
    std::string err;
    if (!h->model.pack(err)) return rfail(VSR_ERR_ARG, err);
    if (device >= 0) RCCHK(upload_weights(h->ws, h->model.packed, device));
    h->device = device;
    h->finalized = true;
    return 0;
}

void vsr_lama_destroy(vsr_lama_t* h)
{
    if (!h) return;
    if (h->device >= 0) {
        (void)hipSetDevice(h->device);
        (void)hipDeviceSynchronize();
        h->plans.clear();
        h->ws.release();
    }
    delete h;
}

int vsr_lama_blocks(const vsr_lama_t* h) { return (h && h->model.packed_ready()) ? h->model.nBlocks : -1; }

int64_t vsr_lama_packed_weights(const vsr_lama_t* h, float* out, int64_t capacity)
{
    if (!h || !h->model.packed_ready()) { rfail(VSR_ERR_STATE, "model not finalized"); return -1; }
    const int64_t n = (int64_t)h->model.packed.size();
    if (out && capacity >= n) memcpy(out, h->model.packed.data(), (size_t)n * sizeof(float));
    return n;
}

// LamaInpaint._inpaint_batch's network call + post-processing for B images whose rows are contiguous (full-width strips of
// frames resident in HBM): img[b] = img_dev + b * img_frame_stride (H*W*3 bytes each), mask[b] = mask_dev + b * mask_frame_stride
// (0: one mask for all), out likewise; out may alias img.
int vsr_lama_inpaint(vsr_lama_t* h, const uint8_t* img_dev, int64_t img_frame_stride, const uint8_t* mask_dev, int64_t mask_frame_stride,
                     int B, int H, int W, uint8_t* out_dev, int64_t out_frame_stride, void* stream_)
{
    if (!h || !img_dev || !mask_dev || !out_dev || B < 1) return rfail(VSR_ERR_ARG, "bad argument");
    if (!h->finalized || h->device < 0)
        return rfail(VSR_ERR_NOGPU, "model is not finalized on a HIP device (no GPU / finalize(device<0)); there is no CPU fallback");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t stream = (hipStream_t)stream_;
    FlowPlanDev* pd = nullptr;
    RCCHK(lama_plan_dev(h, B, H, W, &pd));
    if (h->geom != std::make_tuple(B, H, W)) {
        RCCHK(clear_workspace(h->ws, stream, pd->plan.get()));
        h->geom = std::make_tuple(B, H, W);
    }
    const size_t ibytes = (size_t)H * W * 3, mbytes = (size_t)H * W;
    HIPCHK(hipMemcpy2DAsync(h->ws.bufs[LB_IN_U8], ibytes, img_dev, (size_t)img_frame_stride, ibytes, (size_t)B, hipMemcpyDeviceToDevice, stream));
    if (mask_frame_stride == 0) {
        for (int b = 0; b < B; ++b)
            HIPCHK(hipMemcpyAsync((uint8_t*)h->ws.bufs[LB_MASK_U8] + (size_t)b * mbytes, mask_dev, mbytes, hipMemcpyDeviceToDevice, stream));
    } else {
        HIPCHK(hipMemcpy2DAsync(h->ws.bufs[LB_MASK_U8], mbytes, mask_dev, (size_t)mask_frame_stride, mbytes, (size_t)B, hipMemcpyDeviceToDevice, stream));
    }
    RCCHK(range_guard_arm(h->ws, stream));
    RCCHK(run_plan(h->ws, pd, 0, stream));
    bool fired = false;
    RCCHK(range_guard_fired(h->ws, stream, &fired));
    if (fired) {
        const int mode_ = h->ws.precision;
        h->ws.precision = 0;
        const int rc = vsr_lama_inpaint(h, img_dev, img_frame_stride, mask_dev, mask_frame_stride, B, H, W, out_dev, out_frame_stride, stream_);
        h->ws.precision = mode_;
        return rc;
    }
    HIPCHK(hipMemcpy2DAsync(out_dev, (size_t)out_frame_stride, h->ws.bufs[LB_OUT_U8], ibytes, ibytes, (size_t)B, hipMemcpyDeviceToDevice, stream));
    return 0;
}

int vsr_lama_set_precision(vsr_lama_t* h, int mode) { return h ? set_precision(h->ws, mode) : rfail(VSR_ERR_ARG, "null handle"); }
int64_t vsr_lama_fallbacks(const vsr_lama_t* h) { return h ? h->ws.fallbacks : -1; }

int vsr_lama_read_buffer(vsr_lama_t* h, int buf, int64_t offset, int64_t count, float* out_host)
{
    if (!h) return rfail(VSR_ERR_ARG, "null handle");
    return read_buffer(h->ws, h->device, buf, offset, count, out_host);
}

double vsr_lama_flops(vsr_lama_t* h, int B, int H, int W)
{
    if (!h || !h->model.packed_ready()) return -1.0;
    try {
        return LamaPlan(h->model, B, H, W).flops;
    } catch (const std::exception&) {
        return -1.0;
    }
}

int vsr_lama_plan_create(const vsr_lama_t* h, int B, int H, int W, vsr_plan_t** out)
{
    if (!h || !out) return rfail(VSR_ERR_ARG, "bad argument");
    if (!h->model.packed_ready()) return rfail(VSR_ERR_STATE, "model not finalized");
    try {
        std::unique_ptr<vsr_plan> p(new vsr_plan);
        p->plan.reset(new LamaPlan(h->model, B, H, W));
        *out = p.release();
    } catch (const std::exception& e) {
        return rfail(VSR_ERR_ARG, std::string("lama plan: ") + e.what());
    }
    return 0;
}

int64_t vsr_plan_consts(const vsr_plan_t* p, float* out, int64_t capacity)
{
    if (!p) return -1;
    const int64_t n = (int64_t)p->plan->consts.size();
    if (out && capacity >= n && n) memcpy(out, p->plan->consts.data(), (size_t)n * sizeof(float));
    return n;
}

int vsr_raft_create(vsr_raft_t** out)
{
    if (!out) return rfail(VSR_ERR_ARG, "null out pointer");
    *out = new vsr_raft();
    return 0;
}

int vsr_raft_set_param(vsr_raft_t* h, const char* key, const float* data, const int64_t* shape, int ndim)
{
    if (!h || !key || !data || (ndim > 0 && !shape)) return rfail(VSR_ERR_ARG, "bad argument");
    if (h->finalized) return rfail(VSR_ERR_STATE, "model already finalized");
    std::string err;
    if (!h->model.set_param(key, data, shape, ndim, err)) return rfail(VSR_ERR_ARG, err);
    return 0;
}

int vsr_raft_finalize(vsr_raft_t* h, int device)
{
    if (!h) return rfail(VSR_ERR_ARG, "null handle");
    if (h->finalized) return rfail(VSR_ERR_STATE, "model already finalized");
    std::string err;
    if (!h->model.pack(err)) return rfail(VSR_ERR_ARG, err);
    if (device >= 0) RCCHK(upload_weights(h->ws, h->model.packed, device));
    h->device = device;
    h->finalized = true;
    return 0;
}

void vsr_raft_destroy(vsr_raft_t* h)
{
    if (!h) return;
    if (h->device >= 0) {
        (void)hipSetDevice(h->device);
        (void)hipDeviceSynchronize();
        h->plans.clear();
        h->ws.release();
    }
    delete h;
}

int64_t vsr_raft_packed_weights(const vsr_raft_t* h, float* out, int64_t capacity)
{
    if (!h || !h->model.packed_ready()) { rfail(VSR_ERR_STATE, "model not finalized"); return -1; }
    const int64_t n = (int64_t)h->model.packed.size();
    if (out && capacity >= n) memcpy(out, h->model.packed.data(), (size_t)n * sizeof(float));
    return n;
}

int vsr_raft_flows(vsr_raft_t* h, const uint8_t* frames_dev, int t, int H, int W, int iters, int bgr, float* fwd_dev, float* bwd_dev,
                   void* stream_)
{
    if (!h || !frames_dev || !fwd_dev || !bwd_dev) return rfail(VSR_ERR_ARG, "bad argument");
    if (!h->finalized || h->device < 0)
        return rfail(VSR_ERR_NOGPU, "model is not finalized on a HIP device (no GPU / finalize(device<0)); there is no CPU fallback");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t stream = (hipStream_t)stream_;
    FlowPlanDev* pd = nullptr;
    RCCHK(raft_plan_dev(h, t, H, W, iters, &pd));
    // every halo'd buffer of a RAFT plan (encoder stages, the recurrent state, c1 / corflo / f1 / fh: raft_plan.cpp) is frame after frame
    // (or pair after pair) with a per-frame geometry fixed by (H, W): another frame count reaches further, it does not move a halo.  The
    // runs of a batch (propainter_inpaint.py raft_runs: 18, 18, 18, 17 frames of a 68-frame batch) therefore share one clean workspace;
    // clearing on every change of t was 124 fills and 10 ms per batch (profiles/r06_propainter_f32_raft_kernel_stats.csv)
    RCCHK(ensure_clean(h->ws, "raft:" + std::to_string(H) + ":" + std::to_string(W), *pd->plan, stream));
    h->geom = std::make_tuple(t, H, W);
    RCCHK(range_guard_arm(h->ws, stream));
    HIPCHK(hipMemcpyAsync(h->ws.bufs[RB_IN_U8], frames_dev, (size_t)t * H * W * 3, hipMemcpyDeviceToDevice, stream));
    RCCHK(run_plan(h->ws, pd, bgr ? 1 : 0, stream));
    bool fired = false;
    RCCHK(range_guard_fired(h->ws, stream, &fired));
    if (fired) {                                  // redo the call with exact fp32 contractions
        const int mode_ = h->ws.precision;
        h->ws.precision = 0;
        const int rc = vsr_raft_flows(h, frames_dev, t, H, W, iters, bgr, fwd_dev, bwd_dev, stream_);
        h->ws.precision = mode_;
        return rc;
    }
    const size_t half = (size_t)(t - 1) * 2 * H * W * sizeof(float);
    HIPCHK(hipMemcpyAsync(fwd_dev, h->ws.bufs[RB_OUT], half, hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemcpyAsync(bwd_dev, (char*)h->ws.bufs[RB_OUT] + half, half, hipMemcpyDeviceToDevice, stream));
    return 0;
}

int vsr_raft_set_precision(vsr_raft_t* h, int mode) { return h ? set_precision(h->ws, mode) : rfail(VSR_ERR_ARG, "null handle"); }
int64_t vsr_raft_fallbacks(const vsr_raft_t* h) { return h ? h->ws.fallbacks : -1; }

int vsr_raft_read_buffer(vsr_raft_t* h, int buf, int64_t offset, int64_t count, float* out_host)
{
    if (!h) return rfail(VSR_ERR_ARG, "null handle");
    return read_buffer(h->ws, h->device, buf, offset, count, out_host);
}

double vsr_raft_flops(vsr_raft_t* h, int t, int H, int W, int iters)
{
    if (!h || !h->model.packed_ready()) { rfail(VSR_ERR_ARG, "bad argument"); return -1.0; }
    try {
        RaftPlan p(h->model, t, H, W, iters);
        return p.flops;
    } catch (const std::exception& e) {
        rfail(VSR_ERR_ARG, std::string("raft plan: ") + e.what());
        return -1.0;
    }
}

int vsr_raft_plan_create(const vsr_raft_t* h, int t, int H, int W, int iters, vsr_plan_t** out)
{
    if (!h || !out) return rfail(VSR_ERR_ARG, "bad argument");
    if (!h->model.packed_ready()) return rfail(VSR_ERR_STATE, "model not finalized");
    try {
        std::unique_ptr<vsr_plan> p(new vsr_plan);
        p->plan.reset(new RaftPlan(h->model, t, H, W, iters));
        *out = p.release();
    } catch (const std::exception& e) {
        return rfail(VSR_ERR_ARG, std::string("raft plan: ") + e.what());
    }
    return 0;
}

// ---- flow completion ----------------------------------------------------------------------------------------------

int vsr_rfc_create(vsr_rfc_t** out)
{
    if (!out) return rfail(VSR_ERR_ARG, "null out pointer");
    *out = new vsr_rfc();
    return 0;
}

int vsr_rfc_set_param(vsr_rfc_t* h, const char* key, const float* data, const int64_t* shape, int ndim)
{
    if (!h || !key || !data || (ndim > 0 && !shape)) return rfail(VSR_ERR_ARG, "bad argument");
    if (h->finalized) return rfail(VSR_ERR_STATE, "model already finalized");
    std::string err;
    if (!h->model.set_param(key, data, shape, ndim, err)) return rfail(VSR_ERR_ARG, err);
    return 0;
}

int vsr_rfc_finalize(vsr_rfc_t* h, int device)
{
    if (!h) return rfail(VSR_ERR_ARG, "null handle");
    if (h->finalized) return rfail(VSR_ERR_STATE, "model already finalized");
    std::string err;
    if (!h->model.pack(err)) return rfail(VSR_ERR_ARG, err);
    if (device >= 0) RCCHK(upload_weights(h->ws, h->model.packed, device));
    h->device = device;
    h->finalized = true;
    return 0;
}

void vsr_rfc_destroy(vsr_rfc_t* h)
{
    if (!h) return;
    if (h->device >= 0) {
        (void)hipSetDevice(h->device);
        (void)hipDeviceSynchronize();
        h->plans.clear();
        h->ws.release();
    }
    delete h;
}

int64_t vsr_rfc_packed_weights(const vsr_rfc_t* h, float* out, int64_t capacity)
{
    if (!h || !h->model.packed_ready()) { rfail(VSR_ERR_STATE, "model not finalized"); return -1; }
    const int64_t n = (int64_t)h->model.packed.size();
    if (out && capacity >= n) memcpy(out, h->model.packed.data(), (size_t)n * sizeof(float));
    return n;
}

int vsr_rfc_complete(vsr_rfc_t* h, const float* flows_f_dev, const float* flows_b_dev, const uint8_t* masks_dev, int t, int H, int W,
                     float* out_f_dev, float* out_b_dev, void* stream_)
{
    if (!h || !flows_f_dev || !flows_b_dev || !masks_dev || !out_f_dev || !out_b_dev) return rfail(VSR_ERR_ARG, "bad argument");
    if (!h->finalized || h->device < 0)
        return rfail(VSR_ERR_NOGPU, "model is not finalized on a HIP device (no GPU / finalize(device<0)); there is no CPU fallback");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t stream = (hipStream_t)stream_;
    FlowPlanDev* pd = nullptr;
    RCCHK(rfc_plan_dev(h, t, H, W, &pd));
    if (h->geom != std::make_tuple(t, H, W)) {
        RCCHK(clear_workspace(h->ws, stream, pd->plan.get()));
        h->geom = std::make_tuple(t, H, W);
    }
    const size_t fbytes = (size_t)(t - 1) * 2 * H * W * sizeof(float);
    HIPCHK(hipMemcpyAsync(h->ws.bufs[FB_IN_FLOW_F], flows_f_dev, fbytes, hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemcpyAsync(h->ws.bufs[FB_IN_FLOW_B], flows_b_dev, fbytes, hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemcpyAsync(h->ws.bufs[FB_IN_MASK], masks_dev, (size_t)t * H * W, hipMemcpyDeviceToDevice, stream));
    RCCHK(range_guard_arm(h->ws, stream));
    RCCHK(run_plan(h->ws, pd, 0, stream));
    bool fired = false;
    RCCHK(range_guard_fired(h->ws, stream, &fired));
    if (fired) {
        const int mode_ = h->ws.precision;
        h->ws.precision = 0;
        const int rc = vsr_rfc_complete(h, flows_f_dev, flows_b_dev, masks_dev, t, H, W, out_f_dev, out_b_dev, stream_);
        h->ws.precision = mode_;
        return rc;
    }
    HIPCHK(hipMemcpyAsync(out_f_dev, h->ws.bufs[FB_OUT_F], fbytes, hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemcpyAsync(out_b_dev, h->ws.bufs[FB_OUT_B], fbytes, hipMemcpyDeviceToDevice, stream));
    return 0;
}

int vsr_rfc_set_precision(vsr_rfc_t* h, int mode) { return h ? set_precision(h->ws, mode) : rfail(VSR_ERR_ARG, "null handle"); }
int64_t vsr_rfc_fallbacks(const vsr_rfc_t* h) { return h ? h->ws.fallbacks : -1; }

int vsr_rfc_read_buffer(vsr_rfc_t* h, int buf, int64_t offset, int64_t count, float* out_host)
{
    if (!h) return rfail(VSR_ERR_ARG, "null handle");
    return read_buffer(h->ws, h->device, buf, offset, count, out_host);
}

double vsr_rfc_flops(vsr_rfc_t* h, int t, int H, int W)
{
    if (!h || !h->model.packed_ready()) { rfail(VSR_ERR_ARG, "bad argument"); return -1.0; }
    try {
        RfcPlan p(h->model, t, H, W);
        return p.flops;
    } catch (const std::exception& e) {
        rfail(VSR_ERR_ARG, std::string("flow-completion plan: ") + e.what());
        return -1.0;
    }
}

int vsr_rfc_plan_create(const vsr_rfc_t* h, int t, int H, int W, vsr_plan_t** out)
{
    if (!h || !out) return rfail(VSR_ERR_ARG, "bad argument");
    if (!h->model.packed_ready()) return rfail(VSR_ERR_STATE, "model not finalized");
    try {
        std::unique_ptr<vsr_plan> p(new vsr_plan);
        p->plan.reset(new RfcPlan(h->model, t, H, W));
        *out = p.release();
    } catch (const std::exception& e) {
        return rfail(VSR_ERR_ARG, std::string("flow-completion plan: ") + e.what());
    }
    return 0;
}

// ---- ProPainter generator --------------------------------------------------------------------------------------------

int vsr_pp_create(int device, vsr_pp_t** out)
{
    if (!out) return rfail(VSR_ERR_ARG, "null out pointer");
    if (device >= 0 && device >= vsr_device_count()) return rfail(VSR_ERR_NOGPU, "no such HIP device; there is no CPU fallback");
    *out = new vsr_pp();
    (*out)->device = device;
    return 0;
}

void vsr_pp_destroy(vsr_pp_t* h)
{
    if (!h) return;
    if (h->device >= 0) {
        (void)hipSetDevice(h->device);
        (void)hipDeviceSynchronize();
        h->imgPlans.clear();
        h->genPlans.clear();
        h->ws.release();
    }
    delete h;
}

int vsr_pp_set_param(vsr_pp_t* h, const char* key, const float* data, const int64_t* shape, int ndim)
{
    if (!h || !key || !data || (ndim > 0 && !shape)) return rfail(VSR_ERR_ARG, "bad argument");
    if (h->finalized) return rfail(VSR_ERR_STATE, "model already finalized");
    std::string err;
    if (!h->model.set_param(key, data, shape, ndim, err)) return rfail(VSR_ERR_ARG, err);
    return 0;
}

int vsr_pp_finalize(vsr_pp_t* h)
{
    if (!h) return rfail(VSR_ERR_ARG, "null handle");
    if (h->finalized) return rfail(VSR_ERR_STATE, "model already finalized");
    std::string err;
    if (!h->model.pack(err)) return rfail(VSR_ERR_ARG, err);
    if (h->device >= 0) RCCHK(upload_weights(h->ws, h->model.packed, h->device));
    h->finalized = true;
    return 0;
}

int64_t vsr_pp_packed_weights(const vsr_pp_t* h, float* out, int64_t capacity)
{
    if (!h || !h->model.packed_ready()) { rfail(VSR_ERR_STATE, "model not finalized"); return -1; }
    const int64_t n = (int64_t)h->model.packed.size();
    if (out && capacity >= n) memcpy(out, h->model.packed.data(), (size_t)n * sizeof(float));
    return n;
}

// SparseWindowAttention's masked-window test (sparse_transformer.py:229-236) from host masks: nearest 1/4 downsampling
// (propainter.py:345), MaxPool2d(7, 3, 3) (:351-356), zero padding to whole windows, window max, any local frame
int vsr_pp_window_flags(const uint8_t* masks_host, int lt, int H, int W, uint8_t* flags, int capacity)
{
    if (!masks_host || !flags || lt < 1 || H % 4 || W % 4) return rfail(VSR_ERR_ARG, "bad argument");
    int fh, fw, gh, gw;
    PpGenPlan::token_grid(H, W, fh, fw, gh, gw);
    const int nwh = gh / 5, nww = gw / 9, h = H / 4, w = W / 4;
    if (capacity < nwh * nww) return rfail(VSR_ERR_ARG, "flag buffer too small");
    memset(flags, 0, (size_t)nwh * nww);
    for (int f = 0; f < lt; ++f)
        for (int ty = 0; ty < fh; ++ty)
            for (int tx = 0; tx < fw; ++tx) {
                bool any = false;
                for (int ky = 0; ky < 7 && !any; ++ky)
                    for (int kx = 0; kx < 7 && !any; ++kx) {
                        const int y = 3 * ty - 3 + ky, x = 3 * tx - 3 + kx;
                        if (y >= 0 && y < h && x >= 0 && x < w && masks_host[((int64_t)f * H + 4 * y) * W + 4 * x]) any = true;
                    }
                if (any) flags[(ty / 5) * nww + tx / 9] = 1;
            }
    return nwh * nww;
}

int vsr_pp_forward(vsr_pp_t* h, const float* frames_dev, const float* flows_f_dev, const float* flows_b_dev, const uint8_t* masks_in_dev,
                   const uint8_t* masks_updated_dev, int t, int lt, int H, int W, const uint8_t* window_flags, int nflags, float* out_dev,
                   void* stream_)
{
    return vsr_pp_forward_box(h, frames_dev, flows_f_dev, flows_b_dev, masks_in_dev, masks_updated_dev, t, lt, H, W, window_flags, nflags, 0, 0, 0, 0,
                              out_dev, stream_);
}

int vsr_pp_forward_box(vsr_pp_t* h, const float* frames_dev, const float* flows_f_dev, const float* flows_b_dev, const uint8_t* masks_in_dev,
                       const uint8_t* masks_updated_dev, int t, int lt, int H, int W, const uint8_t* window_flags, int nflags, int row_lo,
                       int row_hi, int col_lo, int col_hi, float* out_dev, void* stream_)
{
    if (!h || !frames_dev || !masks_in_dev || !masks_updated_dev || !window_flags || !out_dev || (lt > 1 && (!flows_f_dev || !flows_b_dev)))
        return rfail(VSR_ERR_ARG, "bad argument");
    if (!h->finalized || h->device < 0)
        return rfail(VSR_ERR_NOGPU, "model is not finalized on a HIP device; there is no CPU fallback");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t stream = (hipStream_t)stream_;
    std::string key = std::to_string(t) + ":" + std::to_string(lt) + ":" + std::to_string(H) + ":" + std::to_string(W) + ":" +
                      std::to_string(row_lo) + "-" + std::to_string(row_hi) + ":" + std::to_string(col_lo) + "-" + std::to_string(col_hi) + ":";
    key.append((const char*)window_flags, (size_t)nflags);
    FlowPlanDev* pd = nullptr;
    auto it = h->genPlans.find(key);
    if (it != h->genPlans.end()) {
        pd = it->second.get();
    } else {
        std::unique_ptr<PlanIR> plan;
        try {
            plan.reset(new PpGenPlan(h->model, t, lt, H, W, std::vector<uint8_t>(window_flags, window_flags + nflags), row_lo, row_hi, col_lo, col_hi));
        } catch (const std::exception& e) {
            return rfail(VSR_ERR_ARG, std::string("generator plan: ") + e.what());
        }
        if (h->ws.needs_growth(*plan)) { h->genPlans.clear(); h->imgPlans.clear(); }
        std::unique_ptr<FlowPlanDev> npd;
        RCCHK(materialize(h->ws, std::move(plan), &npd));
        pd = npd.get();
        h->genPlans[key] = std::move(npd);
    }
    const std::string geom = "gen:" + std::to_string(lt) + ":" + std::to_string(H) + ":" + std::to_string(W);      // (not t: ensure_clean)
    RCCHK(ensure_clean(h->ws, geom, *pd->plan, stream));
    h->geom = geom;
    const size_t hw = (size_t)H * W;
    HIPCHK(hipMemcpyAsync(h->ws.bufs[PB_IN_FRAMES], frames_dev, (size_t)t * 3 * hw * sizeof(float), hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemcpyAsync(h->ws.bufs[PB_IN_MASK_U8], masks_in_dev, (size_t)t * hw, hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemcpyAsync(h->ws.bufs[PB_IN_MASK_UPD_U8], masks_updated_dev, (size_t)t * hw, hipMemcpyDeviceToDevice, stream));
    if (lt > 1) {
        HIPCHK(hipMemcpyAsync(h->ws.bufs[PB_IN_FLOW_F], flows_f_dev, (size_t)(lt - 1) * 2 * hw * sizeof(float), hipMemcpyDeviceToDevice, stream));
        HIPCHK(hipMemcpyAsync(h->ws.bufs[PB_IN_FLOW_B], flows_b_dev, (size_t)(lt - 1) * 2 * hw * sizeof(float), hipMemcpyDeviceToDevice, stream));
    }
    RCCHK(range_guard_arm(h->ws, stream));
    RCCHK(run_plan(h->ws, pd, 0, stream));
    bool fired = false;
    RCCHK(range_guard_fired(h->ws, stream, &fired));
    if (fired) {
        const int mode_ = h->ws.precision;
        h->ws.precision = 0;
        const int rc = vsr_pp_forward_box(h, frames_dev, flows_f_dev, flows_b_dev, masks_in_dev, masks_updated_dev, t, lt, H, W, window_flags, nflags,
                                          row_lo, row_hi, col_lo, col_hi, out_dev, stream_);
        h->ws.precision = mode_;
        return rc;
    }
    HIPCHK(hipMemcpyAsync(out_dev, h->ws.bufs[PG_OUT], (size_t)lt * 3 * hw * sizeof(float), hipMemcpyDeviceToDevice, stream));
    return 0;
}

// ---- the encoder and the soft split once per frame (csrc/pp_plan.h PpPlanMode) ----------------------------------------------
static int pp_plan_for(vsr_pp_t* h, const std::string& key, int t, int lt, int H, int W, const uint8_t* window_flags, int nflags, int row_lo,
                       int row_hi, int col_lo, int col_hi, int mode, FlowPlanDev** out)
{
    auto it = h->genPlans.find(key);
    if (it != h->genPlans.end()) { *out = it->second.get(); return 0; }
    std::unique_ptr<PlanIR> plan;
    try {
        plan.reset(new PpGenPlan(h->model, t, lt, H, W,
                                 window_flags ? std::vector<uint8_t>(window_flags, window_flags + nflags) : std::vector<uint8_t>(), row_lo, row_hi,
                                 col_lo, col_hi, mode));
    } catch (const std::exception& e) {
        return rfail(VSR_ERR_ARG, std::string("generator plan: ") + e.what());
    }
    if (h->ws.needs_growth(*plan)) { h->genPlans.clear(); h->imgPlans.clear(); }
    std::unique_ptr<FlowPlanDev> npd;
    RCCHK(materialize(h->ws, std::move(plan), &npd));
    *out = npd.get();
    h->genPlans[key] = std::move(npd);
    return 0;
}

int vsr_pp_encode(vsr_pp_t* h, const float* frames_dev, const uint8_t* masks_in_dev, const uint8_t* masks_updated_dev, int n, int ntok_frames,
                  int H, int W, float* feat_out_dev, float* tok_out_dev, void* stream_)
{
    if (!h || !frames_dev || !masks_in_dev || !masks_updated_dev || !feat_out_dev || n < 1 || ntok_frames < 0 || ntok_frames > n ||
        (ntok_frames > 0 && !tok_out_dev))
        return rfail(VSR_ERR_ARG, "bad argument");
    if (!h->finalized || h->device < 0)
        return rfail(VSR_ERR_NOGPU, "model is not finalized on a HIP device; there is no CPU fallback");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t stream = (hipStream_t)stream_;
    const std::string geom = "enc:" + std::to_string(n) + ":" + std::to_string(ntok_frames) + ":" + std::to_string(H) + ":" + std::to_string(W);
    FlowPlanDev* pd = nullptr;
    RCCHK(pp_plan_for(h, geom, n, ntok_frames, H, W, nullptr, 0, 0, 0, 0, 0, PP_PLAN_ENCODE, &pd));
    RCCHK(ensure_clean(h->ws, "enc:" + std::to_string(H) + ":" + std::to_string(W), *pd->plan, stream));      // (frame counts only reach further)
    h->geom = geom;
    const size_t hw = (size_t)H * W;
    HIPCHK(hipMemcpyAsync(h->ws.bufs[PB_IN_FRAMES], frames_dev, (size_t)n * 3 * hw * sizeof(float), hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemcpyAsync(h->ws.bufs[PB_IN_MASK_U8], masks_in_dev, (size_t)n * hw, hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemcpyAsync(h->ws.bufs[PB_IN_MASK_UPD_U8], masks_updated_dev, (size_t)n * hw, hipMemcpyDeviceToDevice, stream));
    RCCHK(range_guard_arm(h->ws, stream));
    RCCHK(run_plan(h->ws, pd, 0, stream));
    bool fired = false;
    RCCHK(range_guard_fired(h->ws, stream, &fired));
    if (fired) {
        const int mode_ = h->ws.precision;
        h->ws.precision = 0;
        const int rc = vsr_pp_encode(h, frames_dev, masks_in_dev, masks_updated_dev, n, ntok_frames, H, W, feat_out_dev, tok_out_dev, stream_);
        h->ws.precision = mode_;
        return rc;
    }
    // the features leave PG_FEAT's halo (3) behind: [n][h][w][128] dense
    const PpGenPlan& gp = static_cast<const PpGenPlan&>(*pd->plan);
    const int fh_ = gp.h, fw_ = gp.w, halo = gp.featHalo;
    const size_t rowB = (size_t)fw_ * 128 * sizeof(float), pitchB = (size_t)(fw_ + 2 * halo) * 128 * sizeof(float);
    const size_t frameElems = (size_t)(fh_ + 2 * halo) * (fw_ + 2 * halo) * 128;
    for (int f = 0; f < n; ++f) {
        const float* src = (const float*)h->ws.bufs[PG_FEAT] + f * frameElems + ((size_t)halo * (fw_ + 2 * halo) + halo) * 128;
        HIPCHK(hipMemcpy2DAsync(feat_out_dev + (size_t)f * fh_ * fw_ * 128, rowB, src, pitchB, rowB, (size_t)fh_, hipMemcpyDeviceToDevice, stream));
    }
    if (ntok_frames > 0)
        HIPCHK(hipMemcpyAsync(tok_out_dev, h->ws.bufs[PG_TOKOUT], (size_t)ntok_frames * gp.fh * gp.fw * 512 * sizeof(float), hipMemcpyDeviceToDevice,
                              stream));
    return 0;
}

int vsr_pp_forward_cached(vsr_pp_t* h, const float* feat_cache_dev, const float* tok_cache_dev, const int32_t* cache_idx,
                          const float* flows_f_dev, const float* flows_b_dev, const uint8_t* masks_in_dev, const uint8_t* masks_updated_dev, int t,
                          int lt, int H, int W, const uint8_t* window_flags, int nflags, int row_lo, int row_hi, int col_lo, int col_hi,
                          float* out_dev, void* stream_)
{
    if (!h || !feat_cache_dev || !cache_idx || !masks_in_dev || !masks_updated_dev || !window_flags || !out_dev || t < 1 || lt < 1 || lt > t ||
        (t > lt && !tok_cache_dev) || (lt > 1 && (!flows_f_dev || !flows_b_dev)))
        return rfail(VSR_ERR_ARG, "bad argument");
    for (int k = 0; k < t; ++k)
        if (cache_idx[k] < 0) return rfail(VSR_ERR_ARG, "negative cache index");
    if (!h->finalized || h->device < 0)
        return rfail(VSR_ERR_NOGPU, "model is not finalized on a HIP device; there is no CPU fallback");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t stream = (hipStream_t)stream_;
    const std::string geom = "genc:" + std::to_string(t) + ":" + std::to_string(lt) + ":" + std::to_string(H) + ":" + std::to_string(W);
    std::string key = geom + ":" + std::to_string(row_lo) + "-" + std::to_string(row_hi) + ":" + std::to_string(col_lo) + "-" + std::to_string(col_hi) + ":";
    key.append((const char*)window_flags, (size_t)nflags);
    FlowPlanDev* pd = nullptr;
    RCCHK(pp_plan_for(h, key, t, lt, H, W, window_flags, nflags, row_lo, row_hi, col_lo, col_hi, PP_PLAN_CACHED, &pd));
    RCCHK(ensure_clean(h->ws, "genc:" + std::to_string(lt) + ":" + std::to_string(H) + ":" + std::to_string(W), *pd->plan, stream));
    h->geom = geom;
    const PpGenPlan& gp = static_cast<const PpGenPlan&>(*pd->plan);
    const size_t hw = (size_t)H * W;
    {   // the local frames' features into the propagation buffer's input slots (interiors; the halos stay zero), the reference frames' tokens
        const int fh_ = gp.h, fw_ = gp.w, halo = gp.propHalo;
        const size_t rowB = (size_t)fw_ * 128 * sizeof(float), pitchB = (size_t)(fw_ + 2 * halo) * 128 * sizeof(float);
        const size_t slotElems = (size_t)(fh_ + 2 * halo) * (fw_ + 2 * halo) * 128;
        for (int k = 0; k < lt; ++k) {
            float* dst = (float*)h->ws.bufs[PG_PROP] + k * slotElems + ((size_t)halo * (fw_ + 2 * halo) + halo) * 128;
            HIPCHK(hipMemcpy2DAsync(dst, pitchB, feat_cache_dev + (size_t)cache_idx[k] * fh_ * fw_ * 128, rowB, rowB, (size_t)fh_,
                                    hipMemcpyDeviceToDevice, stream));
        }
        const size_t tokElems = (size_t)gp.fh * gp.fw * 512;
        for (int k = lt; k < t; ++k)
            HIPCHK(hipMemcpyAsync((float*)h->ws.bufs[PG_X] + k * tokElems, tok_cache_dev + (size_t)cache_idx[k] * tokElems, tokElems * sizeof(float),
                                  hipMemcpyDeviceToDevice, stream));
    }
    HIPCHK(hipMemcpyAsync(h->ws.bufs[PB_IN_MASK_U8], masks_in_dev, (size_t)t * hw, hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemcpyAsync(h->ws.bufs[PB_IN_MASK_UPD_U8], masks_updated_dev, (size_t)t * hw, hipMemcpyDeviceToDevice, stream));
    if (lt > 1) {
        HIPCHK(hipMemcpyAsync(h->ws.bufs[PB_IN_FLOW_F], flows_f_dev, (size_t)(lt - 1) * 2 * hw * sizeof(float), hipMemcpyDeviceToDevice, stream));
        HIPCHK(hipMemcpyAsync(h->ws.bufs[PB_IN_FLOW_B], flows_b_dev, (size_t)(lt - 1) * 2 * hw * sizeof(float), hipMemcpyDeviceToDevice, stream));
    }
    RCCHK(range_guard_arm(h->ws, stream));
    RCCHK(run_plan(h->ws, pd, 0, stream));
    bool fired = false;
    RCCHK(range_guard_fired(h->ws, stream, &fired));
    if (fired) {
        const int mode_ = h->ws.precision;
        h->ws.precision = 0;
        const int rc = vsr_pp_forward_cached(h, feat_cache_dev, tok_cache_dev, cache_idx, flows_f_dev, flows_b_dev, masks_in_dev, masks_updated_dev, t,
                                             lt, H, W, window_flags, nflags, row_lo, row_hi, col_lo, col_hi, out_dev, stream_);
        h->ws.precision = mode_;
        return rc;
    }
    HIPCHK(hipMemcpyAsync(out_dev, h->ws.bufs[PG_OUT], (size_t)lt * 3 * hw * sizeof(float), hipMemcpyDeviceToDevice, stream));
    return 0;
}

int vsr_pp_token_count(int H, int W)
{
    if (H < 28 || W < 28 || H % 4 || W % 4) return rfail(VSR_ERR_ARG, "bad argument");
    int fh, fw, gh, gw;
    PpGenPlan::token_grid(H, W, fh, fw, gh, gw);
    return fh * fw;
}

int vsr_pp_set_precision(vsr_pp_t* h, int mode)
{
    if (!h) return rfail(VSR_ERR_ARG, "null handle");
    const int rc = set_precision(h->ws, mode);
    if (rc != 0) return rc;
    // the tile shape of the wide GEMMs follows the arithmetic (PpGenPlan::wideTile): plans baked for the other shape go
    const int wideN = mode == 2 ? 128 : 512;
    if (wideN != h->model.wideN) {
        h->model.wideN = wideN;
        if (!h->genPlans.empty()) {
            if (h->device >= 0) { HIPCHK(hipSetDevice(h->device)); HIPCHK(hipDeviceSynchronize()); }      // nothing in flight may still read their tables
            h->genPlans.clear();
        }
    }
    return 0;
}
int64_t vsr_pp_fallbacks(const vsr_pp_t* h) { return h ? h->ws.fallbacks : -1; }

int vsr_pp_read_buffer(vsr_pp_t* h, int buf, int64_t offset, int64_t count, float* out_host)
{
    if (!h) return rfail(VSR_ERR_ARG, "null handle");
    return read_buffer(h->ws, h->device, buf, offset, count, out_host);
}

double vsr_pp_flops(vsr_pp_t* h, int t, int lt, int H, int W, const uint8_t* window_flags, int nflags)
{
    if (!h || !h->model.packed_ready() || !window_flags) { rfail(VSR_ERR_ARG, "bad argument"); return -1.0; }
    try {
        PpGenPlan p(h->model, t, lt, H, W, std::vector<uint8_t>(window_flags, window_flags + nflags));
        return p.flops;
    } catch (const std::exception& e) {
        rfail(VSR_ERR_ARG, std::string("generator plan: ") + e.what());
        return -1.0;
    }
}

double vsr_pp_flops_box(vsr_pp_t* h, int t, int lt, int H, int W, const uint8_t* window_flags, int nflags, int row_lo, int row_hi, int col_lo,
                        int col_hi, double* reference)
{
    if (!h || !h->model.packed_ready() || !window_flags) { rfail(VSR_ERR_ARG, "bad argument"); return -1.0; }
    try {
        PpGenPlan p(h->model, t, lt, H, W, std::vector<uint8_t>(window_flags, window_flags + nflags), row_lo, row_hi, col_lo, col_hi);
        if (reference) *reference = p.refFlops;
        return p.flops;
    } catch (const std::exception& e) {
        rfail(VSR_ERR_ARG, std::string("generator plan: ") + e.what());
        return -1.0;
    }
}

int vsr_pp_gen_plan_create_box(const vsr_pp_t* h, int t, int lt, int H, int W, const uint8_t* window_flags, int nflags, int row_lo, int row_hi,
                               int col_lo, int col_hi, vsr_plan_t** out)
{
    if (!h || !out || !window_flags) return rfail(VSR_ERR_ARG, "bad argument");
    if (!h->model.packed_ready()) return rfail(VSR_ERR_STATE, "model not finalized");
    try {
        std::unique_ptr<vsr_plan> p(new vsr_plan);
        p->plan.reset(new PpGenPlan(h->model, t, lt, H, W, std::vector<uint8_t>(window_flags, window_flags + nflags), row_lo, row_hi, col_lo, col_hi));
        *out = p.release();
    } catch (const std::exception& e) {
        return rfail(VSR_ERR_ARG, std::string("generator plan: ") + e.what());
    }
    return 0;
}

int vsr_pp_gen_plan_create_mode(const vsr_pp_t* h, int t, int lt, int H, int W, const uint8_t* window_flags, int nflags, int row_lo,
                                int row_hi, int col_lo, int col_hi, int mode, vsr_plan_t** out)
{
    if (!h || !out || (!window_flags && mode != PP_PLAN_ENCODE)) return rfail(VSR_ERR_ARG, "bad argument");
    if (!h->model.packed_ready()) return rfail(VSR_ERR_STATE, "model not finalized");
    try {
        std::unique_ptr<vsr_plan> p(new vsr_plan);
        p->plan.reset(new PpGenPlan(h->model, t, lt, H, W,
                                    window_flags ? std::vector<uint8_t>(window_flags, window_flags + nflags) : std::vector<uint8_t>(), row_lo,
                                    row_hi, col_lo, col_hi, mode));
        *out = p.release();
    } catch (const std::exception& e) {
        return rfail(VSR_ERR_ARG, std::string("generator plan: ") + e.what());
    }
    return 0;
}

int vsr_pp_gen_plan_create(const vsr_pp_t* h, int t, int lt, int H, int W, const uint8_t* window_flags, int nflags, vsr_plan_t** out)
{
    if (!h || !out || !window_flags) return rfail(VSR_ERR_ARG, "bad argument");
    if (!h->model.packed_ready()) return rfail(VSR_ERR_STATE, "model not finalized");
    try {
        std::unique_ptr<vsr_plan> p(new vsr_plan);
        p->plan.reset(new PpGenPlan(h->model, t, lt, H, W, std::vector<uint8_t>(window_flags, window_flags + nflags)));
        *out = p.release();
    } catch (const std::exception& e) {
        return rfail(VSR_ERR_ARG, std::string("generator plan: ") + e.what());
    }
    return 0;
}

int vsr_pp_img_propagation(vsr_pp_t* h, const float* masked_frames_dev, const float* flows_f_dev, const float* flows_b_dev,
                           const uint8_t* masks_dev, int t, int H, int W, float* out_frames_dev, uint8_t* out_masks_dev, void* stream_)
{
    if (!h || !masked_frames_dev || !masks_dev || !out_frames_dev || !out_masks_dev || (t > 1 && (!flows_f_dev || !flows_b_dev)))
        return rfail(VSR_ERR_ARG, "bad argument");
    if (h->device < 0) return rfail(VSR_ERR_NOGPU, "handle was created without a HIP device; there is no CPU fallback");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t stream = (hipStream_t)stream_;
    const auto key = std::make_tuple(t, H, W);
    FlowPlanDev* pd = nullptr;
    auto it = h->imgPlans.find(key);
    if (it != h->imgPlans.end()) {
        pd = it->second.get();
    } else {
        std::unique_ptr<PlanIR> plan;
        try {
            plan.reset(new PpImgPropPlan(t, H, W));
        } catch (const std::exception& e) {
            return rfail(VSR_ERR_ARG, std::string("image-propagation plan: ") + e.what());
        }
        if (h->ws.needs_growth(*plan)) { h->imgPlans.clear(); h->genPlans.clear(); }
        std::unique_ptr<FlowPlanDev> npd;
        RCCHK(materialize(h->ws, std::move(plan), &npd));
        pd = npd.get();
        h->imgPlans[key] = std::move(npd);
    }
    h->geom = "img";                                        // planar buffers only, no halos: nothing to clear, but the generator must re-clear
    const size_t hw = (size_t)H * W;
    HIPCHK(hipMemcpyAsync(h->ws.bufs[PB_IN_FRAMES], masked_frames_dev, (size_t)t * 3 * hw * sizeof(float), hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemcpyAsync(h->ws.bufs[PB_IN_MASK_U8], masks_dev, (size_t)t * hw, hipMemcpyDeviceToDevice, stream));
    if (t > 1) {
        HIPCHK(hipMemcpyAsync(h->ws.bufs[PB_IN_FLOW_F], flows_f_dev, (size_t)(t - 1) * 2 * hw * sizeof(float), hipMemcpyDeviceToDevice, stream));
        HIPCHK(hipMemcpyAsync(h->ws.bufs[PB_IN_FLOW_B], flows_b_dev, (size_t)(t - 1) * 2 * hw * sizeof(float), hipMemcpyDeviceToDevice, stream));
    }
    RCCHK(run_plan(h->ws, pd, 0, stream));
    HIPCHK(hipMemcpyAsync(out_frames_dev, h->ws.bufs[PB_FW], (size_t)t * 3 * hw * sizeof(float), hipMemcpyDeviceToDevice, stream));
    if (vsr_pp_launch_mask_u8(h->ws.f(PB_FWM), (int64_t)t * hw, out_masks_dev, stream) != 0) return rfail(VSR_ERR_HIP, "mask conversion launch failed");
    return 0;
}

int vsr_pp_imgprop_plan_create(int t, int H, int W, vsr_plan_t** out)
{
    if (!out) return rfail(VSR_ERR_ARG, "bad argument");
    try {
        std::unique_ptr<vsr_plan> p(new vsr_plan);
        p->plan.reset(new PpImgPropPlan(t, H, W));
        *out = p.release();
    } catch (const std::exception& e) {
        return rfail(VSR_ERR_ARG, std::string("image-propagation plan: ") + e.what());
    }
    return 0;
}

} // extern "C"
