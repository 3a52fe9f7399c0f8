// Device engine + C-ABI (include/vsr_hip.h) of the STTN hot path.
//
// Host side of libvsr_hip.so: owns the HBM workspace (every activation is NHWC fp32 with a
// physical zero halo, allocated once and zeroed once -- kernels only ever write interiors, so
// the halos ARE the conv zero padding), materialises a vsr::Plan (sttn_plan.h) into device
// descriptor arrays once per chunk length L, and replays it on the caller's stream: ~16
// launches per transformer block, no host synchronisation, no allocation in steady state.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>
#include "../../include/vsr_hip.h"
#include "elementwise.h"
#include "gather_gemm.h"
#include "sttn_plan.h"

using namespace vsr;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}
// error sink shared with flow_engine.hip (plan_c.h)
int vsr_internal_fail(int code, const char* msg) { return fail(code, msg); }

#define HIPCHK(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return fail(VSR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                   \
    } while (0)
#define RCCHK(expr)                                                                                        \
    do {                                                                                                   \
        int rc_ = (expr);                                                                                  \
        if (rc_ != 0) return rc_;                                                                          \
    } while (0)

namespace {

struct OpDev {
    int kind = 0, tileCfg = 0, bmode = 0;
    const void* dDesc = nullptr; // GGProblem* / SMProblem* (device)
    int nitems = 0, total = 0;   // total = workgroups (GEMM) or padded rows (softmax)
    int nQueues = 1;             // persistent gather-GEMM: 8 = per-XCD tile ranges, 1 = global queue
    const void* dDesc7 = nullptr; // fp16-operand mode: the large NK problems of the op, cut for the 256 x 256 kernel (gather_gemm_v7.h)
    int nitems7 = 0, total7 = 0;
    struct Vt { const float* B; const int32_t* rowB; const int32_t* colB; int K, N; int64_t ld; float* dst; };
    std::vector<Vt> vts;          // KN problems among them (P.V): their B operand is turned to NK form first (vsr_launch_kn_to_nk_split)
    // elementwise
    const void* src = nullptr;
    void* dst = nullptr;
    int H = 0, W = 0, C = 0, haloS = 0, haloD = 0, n = 0, ldy = 0, pix = 0, premask = 0;
    const int32_t* tFrameIdx = nullptr;
    const int32_t* tFirst = nullptr;
    const uint8_t* maskU8 = nullptr;   // sttn-det: resized mask bytes
    const uint8_t* inU8 = nullptr;     // sttn-det: model-res input frames (decode blend)
    // reduce_scatter
    int M = 0, N = 0, nsplit = 0;
    int64_t splitStride = 0;
    const int32_t* tRowC = nullptr;
    const int32_t* tColC = nullptr;
    int rowLo = 0, rowHi = 0;    // UPSAMPLE2X: output rows written; DECODE_OUT: image rows decoded (Plan::decLo)
    int split = 0;               // precision mode 2: this elementwise op reads / writes split-format tensors
    bool aexp = false;           // GEMM: the launch carries VSR_ACT_A_EXP problems (P.V of a fused attention)
    bool thin = false;           // exact fp32, NK: a short-K / single-round launch on the one-workgroup-per-tile kernel (sttn_thin below): same bits
    const float* lsum = nullptr; // reduce_scatter of a fused attention: partial row sums [nsplit][ldL]
    int ldL = 0;
    double flops = 0;
    int lane = 0;                // stream the op is issued on (Plan::lanes)
    std::string tag;
};

struct PlanDev {
    std::unique_ptr<Plan> plan;
    uint64_t lastUse = 0;              // vsr_sttn::useClock at the last lookup (least-recently-used eviction of the plan cache)
    int32_t* dTables = nullptr;
    void* dDescs = nullptr;
    int32_t* dIsFloat = nullptr;
    unsigned int* dQueues = nullptr;   // 8 tile-queue counters (one per XCD) per op (persistent gather-GEMM), zeroed per run
    int32_t* dTables7 = nullptr;       // row tables of the transposed P.V operands (n -> n * ld)
    float* dVt = nullptr;              // the transposed operands themselves, one region per lane
    std::vector<OpDev> ops;
    ~PlanDev()
    {
        if (dTables7) (void)hipFree(dTables7);
        if (dVt) (void)hipFree(dVt);
        if (dQueues) (void)hipFree(dQueues);
        if (dTables) (void)hipFree(dTables);
        if (dDescs) (void)hipFree(dDescs);
        if (dIsFloat) (void)hipFree(dIsFloat);
    }
};

struct StripTables { // cv2.resize tables for one (W, split_h): down to model size and back up
    void* dev = nullptr;
    const int32_t *dxofs, *dyofs, *uxofs, *uyofs;
    const int16_t *dialpha, *dibeta, *uialpha, *uibeta;
    const float *ufalpha, *ufbeta;
};

struct TimingRec {
    std::string tag, kernel;
    double flops;
    hipEvent_t a, b;
};

} // namespace

static bool sttn_thin(int precision, int bmode, const std::vector<GGProblem>& probs, int totalTiles);

struct vsr_sttn {
    Model model;
    int device = -1;
    bool finalized = false;
    void* bufs[BUF_COUNT] = {};
    int64_t cap[BUF_COUNT] = {};
    std::map<int64_t, std::unique_ptr<PlanDev>> plans;
    uint64_t useClock = 0;
    std::map<std::pair<int, int>, StripTables> strips;
    float* compAreas = nullptr;
    int64_t compAreasCap = 0;
    int32_t* dSel = nullptr;
    int dSelCap = 0;
    int precision = 0;                 // 0 = exact fp32 MFMA, 1 = split-half f16 MFMA with range guard + fp32 fallback,
                                       // 2 = split-half on split-format tensors, 3 = fp16 operands on the same tensors
    float* weightsSplit = nullptr;     // mode 2: the packed weights in split format (biases are read from BUF_WEIGHTS)
    unsigned int* dRangeFlag = nullptr;
    int64_t fallbacks = 0;             // chunks recomputed in fp32 because the range guard fired
    // lanes: the sliding windows of a chunk are independent until their decoded frames are averaged into BUF_COMP, so odd
    // windows are issued on a second stream with their own window buffers -- the tail of one lane's launch (a partial last round
    // of tiles) is filled by the other lane's kernel.  Same arithmetic in the same order: results do not change.
    int lanes = 2;               // VSR_STTN_LANES / vsr_sttn_set_lanes: 1 = everything on the caller's stream
    hipStream_t laneStream[kMaxLanes] = {};     // [0] unused (the caller's stream)
    hipEvent_t evFork = nullptr, evJoin[kMaxLanes] = {};
    std::vector<hipEvent_t> evDecode;
    int timing = 0;              // 0 off; 1 every op; 2 only the launches of the 128x64 NK gather-GEMM (the dominant kernel symbol)
    std::vector<TimingRec> pending;
    std::map<std::string, std::pair<double, std::pair<int, double>>> timed; // tag -> (ms, (launches, flops))
    explicit vsr_sttn(int variant) : model(variant)
    {
        const char* e = getenv("VSR_PRECISION");
        precision = !e ? 0 : (e[0] == '3' ? 3 : (e[0] == '2' ? 2 : ((e[0] == '1' || e[0] == 's') ? 1 : 0)));   // "1" / "split" / "2" / "3"
        const char* l = getenv("VSR_STTN_LANES");
        lanes = (l && l[0] >= '1' && l[0] <= '0' + kMaxLanes) ? l[0] - '0' : 2;
    }
};

#include "plan_c.h"

static int64_t bufBytes(int buf, int64_t elems) { return (buf == BUF_IN_U8 || buf == BUF_MASK_U8) ? elems : elems * 4; }

// tile queues of a launch whose problems have many N tiles per row block (QK^T, QKV, wide 1x1 convs): one global queue in
// row-major order (round 2: 101 vs 86 TF against per-XCD ranges on the QKV GEMM).  VSR_GG_SWIZZLE=1 asks the exact-fp32 kernel
// v3 for per-XCD ranges walked in groups of 8 row blocks instead: it halves the fabric reads of the 4800-token QK^T (788 -> 391 MB
// per launch, L2 hits 64 -> 80 %, profiles/r03_pmc_l2.log) at the same speed -- the operand latency was hidden either way -- so it
// stays an option.
static int gg_wide_queues()
{
    static const int v = [] { const char* e = getenv("VSR_GG_SWIZZLE"); return (e && atoi(e) == 1) ? (8 | 0x100) : 1; }();
    return v;
}


// decLo / decHi: the rows of the model-resolution output the caller will read (Plan::decLo; 0, 0 = all)
static int build_plan_dev(vsr_sttn* h, int L, int precision, PlanDev** out, int decLo = 0, int decHi = 0, int decXLo = 0, int decXHi = 0)
{
    // the key packs the four decoder bounds into 10 bits each
    if (h->model.g.modelH >= 1024 || h->model.g.modelW >= 1024 || decLo < 0 || decHi >= 1024 || decXLo < 0 || decXHi >= 1024)
        return fail(VSR_ERR_ARG, "plan key: model resolution / decoder bounds beyond 1023");
    const int64_t key = ((((((int64_t)L * 4 + precision) * 8 + h->lanes) * 1024 + decLo) * 1024 + decHi) * 1024 + decXLo) * 1024 + decXHi;
    const bool fmt = precision >= 2;       // split-format tensors: everything a GEMM reads (see gather_gemm_v5.h)
    auto plainF32 = [](int buf) { buf = baseBuf(buf); return buf == BUF_S || buf == BUF_PVPART || buf == BUF_D4 || buf == BUF_COMP; };
    if (fmt && !h->weightsSplit) {
        const int64_t n = (int64_t)h->model.packed.size();
        HIPCHK(hipMalloc((void**)&h->weightsSplit, (size_t)n * sizeof(float)));
        if (vsr_launch_to_split((const float*)h->bufs[BUF_WEIGHTS], h->weightsSplit, n, nullptr) != 0)
            return fail(VSR_ERR_HIP, "weight split launch failed");
        HIPCHK(hipDeviceSynchronize());
    }
    auto it = h->plans.find(key);
    if (it != h->plans.end()) { it->second->lastUse = ++h->useClock; *out = it->second.get(); return 0; }
    std::unique_ptr<PlanDev> pd(new PlanDev);
    pd->lastUse = ++h->useClock;
    try {
        pd->plan.reset(new Plan(h->model, L, precision, h->lanes, decLo, decHi, decXLo, decXHi));
    } catch (const std::exception& e) {
        return fail(VSR_ERR_ARG, std::string("plan: ") + e.what());
    }
    const Plan& P = *pd->plan;
    // ---- workspace: grow buffers if this L needs more; baked pointers of cached plans die with it
    bool grow = false;
    for (int b = 0; b < BUF_COUNT; ++b)
        if (b != BUF_WEIGHTS && P.bufElems[b] > h->cap[b]) grow = true;
    if (grow) {
        h->plans.clear();
        for (int b = 0; b < BUF_COUNT; ++b) {
            if (b == BUF_WEIGHTS || P.bufElems[b] <= h->cap[b]) continue;
            if (h->bufs[b]) { HIPCHK(hipFree(h->bufs[b])); h->bufs[b] = nullptr; h->cap[b] = 0; }
            const int64_t bytes = bufBytes(b, P.bufElems[b]);
            HIPCHK(hipMalloc(&h->bufs[b], (size_t)bytes));
            HIPCHK(hipMemset(h->bufs[b], 0, (size_t)bytes)); // zero halos, once
            h->cap[b] = P.bufElems[b];
        }
        HIPCHK(hipDeviceSynchronize());
    }
    // ---- tables: one allocation
    std::vector<int64_t> toff(P.tables.size());
    int64_t tot = 0;
    for (size_t i = 0; i < P.tables.size(); ++i) { toff[i] = tot; tot += (int64_t)((P.tables[i].size() + 3) / 4 * 4); }
    std::vector<int32_t> flat((size_t)tot, 0);
    for (size_t i = 0; i < P.tables.size(); ++i)
        memcpy(flat.data() + toff[i], P.tables[i].data(), P.tables[i].size() * sizeof(int32_t));
    HIPCHK(hipMalloc((void**)&pd->dTables, (size_t)(tot > 0 ? tot : 4) * sizeof(int32_t)));
    HIPCHK(hipMemcpy(pd->dTables, flat.data(), (size_t)tot * sizeof(int32_t), hipMemcpyHostToDevice));
    auto T = [&](int id) -> const int32_t* { return id < 0 ? nullptr : pd->dTables + toff[id]; };
    auto F = [&](int buf, int64_t off) -> float* { return buf < 0 ? nullptr : (float*)h->bufs[buf] + off; };

    // ---- descriptors: one allocation, 64-byte aligned records
    // split-format modes (precision 2, 3): the large NK problems of an op (convs, QKV, the fine attention scales) go to the 256 x 256 kernel, each cut
    // into whole rounds + a remainder (vsr_v7_split: up to two descriptors); the rest of the op stays on the 128 x 64 kernel
    static const bool v7on = [] { const char* e = getenv("VSR_F16_V7"); return !(e && atoi(e) == 0); }();
    // (exact fp32 stays on the 128 x 64 kernel: the 288 x 256 one-workgroup-per-CU tile of round 4 -- gather_gemm_v8.h, opt-in, deleted in
    // round 6 -- won 0.7 % on a single-lane step and lost 2 % with the default two window lanes)
    const int cus = vsr_gg_cus();
    // NK problems as they are; KN problems (P.V: B = V, n-contiguous) after their B operand has been turned to NK form into a
    // scratch tensor (one pass over V per product, ~2 % of the product's own time).  Split-K problems keep their partial planes.
    // COMPANIONS (round 6, OPT-IN: VSR_F16_V7_COMPANIONS=1): an op whose launch already has such a problem (the fine attention scales) takes its
    // smaller problems along -- the coarse scales of the same product (375 / 60 tokens, K = 12 288 / 76 800 split 12 / 75 ways; P.V with
    // N = 3 200 / 12 288) are a second launch on the 128 x 64 kernel, which cannot share a CU with this one (one workgroup per CU: 128 KB
    // of LDS).  Measured and NOT a win (profiles/r06c_f16_companions_ab.log): the score launches 40.4 -> 37.9 ms per two chunks, P.V
    // unchanged, config 5 681 / 669 -> 668 / 651 fps -- the short tiles cost the 256 x 256 launch what the second launch cost, and the two
    // window lanes were hiding that one.  Same bits either way (test_gather_gemm_fp16_256x256_equals_128x64).
    static const bool companions = [] { const char* e = getenv("VSR_F16_V7_COMPANIONS"); return e && atoi(e) == 1; }();
    auto anchorV7 = [&](const Op& op, const GemmItem& g) {
        if (!v7on || precision < 2 || (g.act & (VSR_ACT_ROW_MAX | VSR_ACT_A_EXP))) return false;
        if (op.bmode == VSR_BMODE_KN && (g.K % 32 || g.N % 32)) return false;
        const int64_t units = (int64_t)((g.M + 255) / 256) * ((g.N + 255) / 256) * g.splitK;
        return g.N >= 192 && g.K >= 256 && g.M >= 1024 && units * 2 >= cus;
    };
    auto forV7 = [&](const Op& op, const GemmItem& g) {
        if (anchorV7(op, g)) return true;
        if (!companions || !v7on || precision < 2 || (g.act & (VSR_ACT_ROW_MAX | VSR_ACT_A_EXP))) return false;
        if (op.bmode == VSR_BMODE_KN && (g.K % 32 || g.N % 32)) return false;
        const int kSlice = g.splitK > 1 ? g.chunksPerSplit * VSR_GG_KC : g.K;
        if (kSlice < 256 || g.N < 32) return false;
        for (const GemmItem& o : op.gemm)
            if (anchorV7(op, o)) return true;
        return false;
    };
    // scratch for the transposed operands: per lane the largest sum over one op's KN problems; row tables n -> n * ld
    std::vector<int64_t> vtLane(kMaxLanes, 0);
    int64_t tab7 = 0;
    for (const Op& op : P.ops) {
        if (op.kind != OP_GEMM || op.bmode != VSR_BMODE_KN) continue;
        int64_t sum = 0;
        for (const GemmItem& g : op.gemm)
            if (forV7(op, g)) { sum += (int64_t)g.N * g.K; tab7 += (g.N + 3) / 4 * 4; }
        if (sum > vtLane[op.lane]) vtLane[op.lane] = sum;
    }
    std::vector<int64_t> vtBase(kMaxLanes, 0);
    int64_t vtTotal = 0;
    for (int l = 0; l < kMaxLanes; ++l) { vtBase[l] = vtTotal; vtTotal += (vtLane[l] + 31) / 32 * 32; }
    if (vtTotal > 0) HIPCHK(hipMalloc((void**)&pd->dVt, (size_t)vtTotal * sizeof(float)));
    std::vector<int32_t> hostTab7((size_t)tab7, 0);
    if (tab7 > 0) HIPCHK(hipMalloc((void**)&pd->dTables7, (size_t)tab7 * sizeof(int32_t)));
    int64_t tab7Cursor = 0;
    size_t descBytes = 0;
    for (const Op& op : P.ops) {
        size_t n7 = 0;
        if (op.kind == OP_GEMM) for (const GemmItem& g : op.gemm) if (forV7(op, g)) ++n7;
        descBytes += ((op.gemm.size() + n7) * sizeof(GGProblem) + 63) / 64 * 64 + 64;
        descBytes += (op.softmax.size() * sizeof(SMProblem) + 63) / 64 * 64;
    }
    std::vector<char> hostDesc(descBytes + 64, 0);
    HIPCHK(hipMalloc(&pd->dDescs, descBytes + 64));
    size_t cursor = 0;
    for (const Op& op : P.ops) {
        OpDev od;
        od.kind = op.kind; od.tileCfg = op.tileCfg; od.bmode = op.bmode; od.flops = op.flops; od.tag = op.tag; od.lane = op.lane;
        if (op.kind == OP_GEMM) {
            std::vector<GGProblem> small, big;
            int64_t vtCursor = 0;
            for (size_t j = 0; j < op.gemm.size(); ++j) {
                const GemmItem& g = op.gemm[j];
                GGProblem q{};
                q.A = F(g.bufA, g.offA); q.B = F(g.bufB, g.offB); q.C = F(g.bufC, g.offC);
                if (fmt && g.bufB == BUF_WEIGHTS) q.B = h->weightsSplit + g.offB;
                q.bias = g.offBias >= 0 ? F(g.bufBias, g.offBias) : nullptr;      // bufBias 0 = BUF_WEIGHTS; BUF_ROWMAX for VSR_ACT_A_EXP
                q.R = g.bufR >= 0 ? F(g.bufR, g.offR) : nullptr;
                q.rowA = T(g.tRowA); q.colA = T(g.tColA); q.rowB = T(g.tRowB); q.colB = T(g.tColB);
                q.rowC = T(g.tRowC); q.colC = T(g.tColC); q.rowR = T(g.tRowR);
                q.M = g.M; q.N = g.N; q.K = g.K; q.tilesM = g.tilesM; q.tilesN = g.tilesN;
                q.splitK = g.splitK; q.chunksPerSplit = g.chunksPerSplit;
                q.act = g.act | ((fmt && !plainF32(g.bufC)) ? VSR_ACT_OUT_SPLIT : 0);
                q.alpha = g.alpha; q.splitStride = g.splitStride;
                if (forV7(op, g)) {
                    if (op.bmode == VSR_BMODE_KN) {
                        // B(k, n) = V[rowB[k] + colB[n / 32] + n % 32]  ->  Vt[n * K + k], K = the padded token count (A's chunk table
                        // 32 kc serves as Vt's too)
                        OpDev::Vt vt;
                        vt.B = q.B; vt.rowB = q.rowB; vt.colB = q.colB; vt.K = g.K; vt.N = g.N; vt.ld = g.K;
                        vt.dst = pd->dVt + vtBase[op.lane] + vtCursor;
                        vtCursor += (int64_t)g.N * g.K;
                        for (int n = 0; n < g.N; ++n) hostTab7[(size_t)(tab7Cursor + n)] = (int32_t)((int64_t)n * g.K);
                        q.B = vt.dst; q.rowB = pd->dTables7 + tab7Cursor; q.colB = q.colA;
                        tab7Cursor += (g.N + 3) / 4 * 4;
                        od.vts.push_back(vt);
                    }
                    GGProblem two[2];
                    const int n = vsr_v7_split(&q, cus, two);
                    for (int k = 0; k < n; ++k) big.push_back(two[k]);
                } else {
                    small.push_back(q);
                }
            }
            static const bool dumpRouting = [] { const char* e = getenv("VSR_STTN_DUMP_ROUTING"); return e && atoi(e) == 1; }();
            if (dumpRouting) {                  // which kernel every problem of the plan goes to (stderr, once per plan build)
                for (const GemmItem& g : op.gemm)
                    fprintf(stderr, "routing %-16s lane %d bmode %d M %6d N %6d K %6d splitK %2d tiles %3dx%-3d act %#x -> %s\n", op.tag.c_str(), op.lane, op.bmode, g.M,
                            g.N, g.K, g.splitK, g.tilesM, g.tilesN, g.act, forV7(op, g) ? "256x256" : "tile of the op");
            }
            int tileStart = 0;
            for (GGProblem& q : small) { q.tileStart = tileStart; tileStart += q.tilesM * q.tilesN * q.splitK; }
            int tileStart7 = 0;
            for (GGProblem& q : big) { q.tileStart = tileStart7; tileStart7 += q.tilesM * q.tilesN * q.splitK; }
            GGProblem* hp = (GGProblem*)(hostDesc.data() + cursor);
            for (size_t j = 0; j < small.size(); ++j) hp[j] = small[j];
            const size_t smallBytes = (small.size() * sizeof(GGProblem) + 63) / 64 * 64;
            GGProblem* hp7 = (GGProblem*)(hostDesc.data() + cursor + smallBytes);
            for (size_t j = 0; j < big.size(); ++j) hp7[j] = big[j];
            od.dDesc7 = (char*)pd->dDescs + cursor + smallBytes;
            od.nitems7 = (int)big.size();
            od.total7 = tileStart7;
            od.dDesc = (char*)pd->dDescs + cursor;
            od.nitems = (int)small.size();
            od.total = tileStart;
            od.nQueues = 8;
            od.aexp = op.ipar[0] != 0;
            od.thin = sttn_thin(precision, op.bmode, small, tileStart);
            for (const GGProblem& q : small)
                if (q.tilesN > 4) od.nQueues = gg_wide_queues();
            cursor += smallBytes + (big.size() * sizeof(GGProblem) + 63) / 64 * 64;
        } else if (op.kind == OP_SOFTMAX) {
            SMProblem* hp = (SMProblem*)(hostDesc.data() + cursor);
            int rowStart = 0;
            for (size_t j = 0; j < op.softmax.size(); ++j) {
                const SoftmaxItem& s = op.softmax[j];
                SMProblem& q = hp[j];
                q.S = F(s.bufS, s.offS); q.P = F(s.bufP, s.offP);
                q.M = s.M; q.N = s.N; q.ldS = s.ldS; q.ldP = s.ldP; q.nsplit = s.nsplit; q.rowStart = rowStart;
                q.scale = s.scale; q.flags = fmt ? 1 : 0; q.splitStride = s.splitStride;
                rowStart += (s.M + 3) / 4 * 4;
            }
            od.dDesc = (char*)pd->dDescs + cursor;
            od.nitems = (int)op.softmax.size();
            od.total = rowStart;
            cursor += (op.softmax.size() * sizeof(SMProblem) + 63) / 64 * 64;
        } else if (op.kind == OP_REDUCE_SCATTER) {
            od.src = F(op.bufSrc, op.offSrc);
            od.dst = F(op.bufDst, op.offDst);
            od.M = op.M; od.N = op.N; od.nsplit = op.nsplit; od.splitStride = op.splitStride;
            od.tRowC = T(op.tRowC); od.tColC = T(op.tColC);
            od.split = fmt ? 1 : 0;
            if (op.ibuf[0] >= 0) { od.lsum = F(op.ibuf[0], op.ioff[0]); od.ldL = op.ipar[0]; }
        } else {
            od.split = (fmt && op.kind != OP_DECODE_OUT) ? 1 : 0;
            od.src = op.bufSrc >= 0 ? h->bufs[op.bufSrc] : nullptr;
            od.dst = op.bufDst >= 0 ? h->bufs[op.bufDst] : nullptr;
            od.H = op.H; od.W = op.W; od.C = op.C; od.haloS = op.haloS; od.haloD = op.haloD; od.n = op.n;
            od.ldy = op.ldy; od.pix = op.pix; od.premask = op.premask;
            od.rowLo = op.ipar[1]; od.rowHi = op.ipar[2];
            od.tFrameIdx = T(op.tFrameIdx); od.tFirst = T(op.tFirst);
            od.maskU8 = op.bufMask >= 0 ? (const uint8_t*)h->bufs[op.bufMask] : nullptr;
            od.inU8 = (const uint8_t*)h->bufs[BUF_IN_U8];
        }
        pd->ops.push_back(std::move(od));
    }
    HIPCHK(hipMemcpy(pd->dDescs, hostDesc.data(), descBytes, hipMemcpyHostToDevice));
    if (tab7 > 0) HIPCHK(hipMemcpy(pd->dTables7, hostTab7.data(), (size_t)tab7 * sizeof(int32_t), hipMemcpyHostToDevice));
    std::vector<int32_t> isf(L);
    for (int i = 0; i < L; ++i) isf[i] = P.compCount[i] > 1 ? 1 : 0;
    HIPCHK(hipMalloc((void**)&pd->dIsFloat, (size_t)L * sizeof(int32_t)));
    HIPCHK(hipMemcpy(pd->dIsFloat, isf.data(), (size_t)L * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&pd->dQueues, (pd->ops.size() + 1) * 16 * sizeof(unsigned int)));
    *out = pd.get();
    if (h->plans.size() >= 48) {
        // a long video with ever new mask rows: drop the 16 least recently used plans -- never a promise-free one (decoder bounds all
        // 0: the hot default of its L) and never the one a multi-area call is in the middle of (they are the most recent ones)
        HIPCHK(hipDeviceSynchronize());            // (nothing in flight may still read the tables that go)
        std::vector<std::pair<uint64_t, int64_t>> byUse;
        for (const auto& kv : h->plans)
            if (kv.first % ((int64_t)1 << 40) != 0) byUse.push_back({kv.second->lastUse, kv.first});
        std::sort(byUse.begin(), byUse.end());
        for (size_t i = 0; i < byUse.size() && i < 16; ++i) h->plans.erase(byUse[i].second);
        if (h->plans.size() >= 48) h->plans.clear();      // only promise-free plans of 48 different lengths: start over
    }
    h->plans[key] = std::move(pd);
    return 0;
}

// gather-GEMM kernel variant (gather_gemm.hip): 1 = one workgroup per tile, 2 = persistent with
// register-staged double buffer, 3 = persistent with LDS-DMA double buffer
// Measured on the 1080p bench (profiles/): NK problems (convs, QKV, QK^T) are fastest on v3, the KN
// problem (P.V, n-contiguous B) on v1.  VSR_GG_VARIANT / VSR_PV_VARIANT override for A/B runs.
static int gg_variant(int bmode, int precision = 0)
{
    if (precision >= 2) return precision == 3 ? 6 : 5;   // split-format tensors: every GEMM must speak the format
    if (precision) {   // split-half mode; VSR_SPLIT_PV_VARIANT lets the P.V product stay on an fp32 kernel for A/B runs
        static const int pv = [] { const char* e = getenv("VSR_SPLIT_PV_VARIANT"); int x = e ? atoi(e) : 4; return (x < 1 || x > 4) ? 4 : x; }();
        return bmode == VSR_BMODE_KN ? pv : 4;
    }
    static const int nk = [] { const char* e = getenv("VSR_GG_VARIANT"); int x = e ? atoi(e) : 3; return (x < 1 || x > 4) ? 3 : x; }();
    static const int kn = [] {
        const char* e = getenv("VSR_PV_VARIANT");
        const char* g = getenv("VSR_GG_VARIANT");
        int x = e ? atoi(e) : ((g && atoi(g) <= 3) ? atoi(g) : 1);
        return (x < 1 || x > 3) ? 1 : x;
    }();
    return bmode == VSR_BMODE_KN ? kn : nk;
}
static bool use_persistent() { return gg_variant(VSR_BMODE_NK) >= 2 || gg_variant(VSR_BMODE_KN) >= 2; }

// Exact fp32, NK problems on the default persistent kernel (variant 3): a launch whose problems are short (K <= VSR_STTN_THIN_SHORT_K) or
// that is a single round of tiles (<= VSR_STTN_THIN_TILES with K <= VSR_STTN_THIN_K) goes to the one-workgroup-per-tile kernel (variant 1) --
// the same MFMA sequence per output element, the same bits (tests/test_gpu_kernels.py::test_one_workgroup_per_tile_kernel_equals_the_
// persistent_one), without the persistent kernel's per-tile fixed cost.  The flow engines' rule (flow_engine.hip thin_variant); the score GEMM
// of a fused attention leaves its row maxima in variant 3's epilogue and stays.  OPT-IN (VSR_STTN_THIN=1), measured on the headline
// (profiles/r06c_sttn_thin_ab.log): the decoder's launches 72.7 -> 76.9 TF, the single-lane step +1.3 %, but with the default two window
// lanes -- whose launches already fill each other's tails -- 220.2 -> 219.7 fps; the fused QKV 1x1 conv (K = 256, 6 750 tiles) is 100 TF on
// either kernel.  Off; the flow engines, whose feature-propagation launches are dependent single rounds, gain (flow_engine.hip).
static bool sttn_thin(int precision, int bmode, const std::vector<GGProblem>& probs, int totalTiles)
{
    static const int mode = [] { const char* e = getenv("VSR_STTN_THIN"); return e ? atoi(e) : 0; }();
    static const int maxTiles = [] { const char* e = getenv("VSR_STTN_THIN_TILES"); return e ? atoi(e) : 768; }();
    static const int maxK = [] { const char* e = getenv("VSR_STTN_THIN_K"); return e ? atoi(e) : 2600; }();
    static const int shortK = [] { const char* e = getenv("VSR_STTN_THIN_SHORT_K"); return e ? atoi(e) : 256; }();
    if (mode == 0 || precision != 0 || bmode != VSR_BMODE_NK || probs.empty() || gg_variant(VSR_BMODE_NK) != 3) return false;
    int K = 0;
    for (const GGProblem& q : probs) {
        if (q.act & VSR_ACT_ROW_MAX) return false;
        K = std::max(K, q.splitK > 1 ? q.chunksPerSplit * VSR_GG_KC : q.K);
    }
    return K <= shortK || (totalTiles <= maxTiles && K <= maxK);
}

static int run_plan(vsr_sttn* h, PlanDev* pd, hipStream_t stream)
{
    const int prec = pd->plan->precision;
    const bool persistent = prec ? true : use_persistent();
    if (persistent) HIPCHK(hipMemsetAsync(pd->dQueues, 0, (pd->ops.size() + 1) * 16 * sizeof(unsigned int), stream));
    if (pd->plan->bufElems[BUF_ROWMAX] > 0)     // fused attention: every instance's row maxima start below every float
        HIPCHK(hipMemsetAsync(h->bufs[BUF_ROWMAX], 0, (size_t)pd->plan->bufElems[BUF_ROWMAX] * sizeof(unsigned int), stream));
    // lanes: everything up to the first window op (the memsets above, the encoder) is on the caller's stream; lane 1 starts
    // behind it (evFork); OP_DECODE_OUT -- the one op of a window that touches shared state, the running average in BUF_COMP --
    // is chained from window to window across the lanes (evDecode); the caller's stream ends behind lane 1 (evJoin)
    const bool laned = pd->plan->lanes > 1;
    hipStream_t laneOf[kMaxLanes] = {stream, stream, stream, stream};
    if (laned) {
        if (!h->evFork) HIPCHK(hipEventCreateWithFlags(&h->evFork, hipEventDisableTiming));
        for (int l = 1; l < pd->plan->lanes; ++l) {
            if (!h->laneStream[l]) HIPCHK(hipStreamCreateWithFlags(&h->laneStream[l], hipStreamNonBlocking));
            if (!h->evJoin[l]) HIPCHK(hipEventCreateWithFlags(&h->evJoin[l], hipEventDisableTiming));
            laneOf[l] = h->laneStream[l];
        }
    }
    bool forked[kMaxLanes] = {};
    int lastDecodeLane = -1;
    size_t nDecode = 0;
    size_t opIndex = 0;
    for (const OpDev& od : pd->ops) {
        unsigned int* queue = persistent ? pd->dQueues + 16 * opIndex : nullptr;   // [0..7] tile ranges, [8] the 256 x 256 kernel's counter
        if (laned && (int)opIndex == pd->plan->firstWindowOp) HIPCHK(hipEventRecord(h->evFork, laneOf[0]));
        ++opIndex;
        hipStream_t const stream = laneOf[od.lane];            // shadows the caller's stream for this op
        if (laned && od.lane > 0 && !forked[od.lane]) {
            HIPCHK(hipStreamWaitEvent(stream, h->evFork, 0));
            forked[od.lane] = true;
        }
        if (laned && od.kind == OP_DECODE_OUT && lastDecodeLane >= 0 && lastDecodeLane != od.lane)
            HIPCHK(hipStreamWaitEvent(stream, h->evDecode[nDecode - 1], 0));
        TimingRec tr;
        // (mode 2 brackets the launches of the NK kernel the convolutions are on, the 128 x 64 tile: the symbol `roofline` is taken on)
        const bool timed = h->timing == 1 || (h->timing == 2 && od.kind == OP_GEMM && od.tileCfg == VSR_TILE_128x64 && od.bmode == VSR_BMODE_NK && !od.thin);
        if (timed) {
            tr.tag = od.tag; tr.flops = od.flops;
            // an op whose problems all went to the 8-wave kernel is that kernel's launch; one that was cut in two keeps the tile's name + "m"
            const bool big = od.kind == OP_GEMM && od.total7 > 0;
            tr.kernel = od.kind != OP_GEMM ? ("kernel:op:" + std::to_string(od.kind))
                      : (big && od.total == 0) ? ("kernel:gg:" + std::to_string(VSR_TILE_256x256) + ":0:v7")
                      : ("kernel:gg:" + std::to_string(od.tileCfg) + ":" + std::to_string(od.bmode) + ":v" +
                         std::to_string(od.thin ? 1 : gg_variant(od.bmode, prec)) + (od.aexp ? "x" : "") + (big ? "m" : ""));
            HIPCHK(hipEventCreate(&tr.a));
            HIPCHK(hipEventCreate(&tr.b));
            HIPCHK(hipEventRecord(tr.a, stream));
        }
        int rc = 0;
        // fp16-operand mode, OPT-IN (VSR_F16_QK_SPLIT=1): the attention SCORES from both halves of their operands (three MFMAs per
        // product, the split-format mode's arithmetic; the tensors carry the lo halves anyway).  Tried in round 6 as a fix for the
        // weight-statistics sweep -- on weights with sharp attention rows (synth.py "peaked": logits x16) the hi-only mode gave 46.9 dB
        // against the oracle, on heavy-tailed ones 27.8 dB -- and measured: 47.4 / 28.0 dB at -7.5 % on config 5 (747.9 -> 691.9 fps,
        // profiles/r06_fifth_call.log).  The error is not made in the score GEMM: every GEMM in front of it hands 11-bit q and k to
        // logits that are 16x larger.  What protects the mode is the accuracy guard of the host side (engine.py AccuracyGuard).
        static const bool qkSplit = [] { const char* e = getenv("VSR_F16_QK_SPLIT"); return e && atoi(e) == 1; }();
        const bool scoresSplit = prec == 3 && qkSplit && od.kind == OP_GEMM && od.tag == "attn.qk";
        switch (od.kind) {
        case OP_GEMM:
            for (const OpDev::Vt& vt : od.vts)
                if (rc == 0) rc = vsr_launch_kn_to_nk_split(vt.B, vt.rowB, vt.colB, vt.K, vt.N, vt.ld, vt.dst, stream);
            if (rc == 0 && od.total7 > 0)
                rc = vsr_launch_gather_gemm_dev((const GGProblem*)od.dDesc7, od.nitems7, od.total7, VSR_TILE_256x256, VSR_BMODE_NK, queue + 8,
                                                (prec == 3 && !scoresSplit) ? 6 : 5, 1, h->dRangeFlag, stream);
            if (rc == 0 && od.total > 0)
                rc = vsr_launch_gather_gemm_dev((const GGProblem*)od.dDesc, od.nitems, od.total, od.tileCfg, od.bmode, queue,
                                                (scoresSplit ? 5 : od.thin ? 1 : gg_variant(od.bmode, prec)) | (od.aexp ? VSR_VARIANT_A_EXP : 0), od.nQueues,
                                                prec ? h->dRangeFlag : nullptr, stream);
            break;
        case OP_SOFTMAX:
            rc = vsr_launch_softmax_dev((const SMProblem*)od.dDesc, od.nitems, od.total, stream);
            break;
        case OP_NORM_IM2COL:
            rc = vsr_launch_norm_im2col_fmt((const uint8_t*)od.src, od.H, od.W, od.n, (float*)od.dst, od.premask, od.maskU8, od.split, stream);
            break;
        case OP_UPSAMPLE2X:
            rc = vsr_launch_upsample2x_rows((const float*)od.src, od.H, od.W, od.C, od.haloS, (float*)od.dst, od.haloD, od.n, od.split,
                                            od.rowLo, od.rowHi > od.rowLo ? od.rowHi : 2 * od.H, stream);
            break;
        case OP_DECODE_OUT:
            if (od.rowHi > od.rowLo && od.W > 0)
                rc = vsr_launch_decode_out_rows((const float*)od.src, od.ldy, od.pix, od.n, od.tFrameIdx, od.tFirst, (float*)od.dst,
                                                od.maskU8 ? od.inU8 : nullptr, od.maskU8, od.W, od.rowLo * od.W, (od.rowHi - od.rowLo) * od.W, stream);
            else
                rc = vsr_launch_decode_out_blk((const float*)od.src, od.ldy, od.pix, od.n, od.tFrameIdx, od.tFirst, (float*)od.dst,
                                               od.maskU8 ? od.inU8 : nullptr, od.maskU8, od.W, stream);
            break;
        case OP_REDUCE_SCATTER:
            rc = vsr_launch_reduce_scatter_fmt((const float*)od.src, od.nsplit, od.splitStride, od.M, od.N, od.tRowC, od.tColC,
                                               (float*)od.dst, od.split, od.lsum, od.ldL, stream);
            break;
        default:
            return fail(VSR_ERR_STATE, "unknown op kind");
        }
        if (rc != 0) return fail(VSR_ERR_HIP, "kernel launch failed: " + od.tag + ": " + hipGetErrorString(hipGetLastError()));
        if (timed) {
            HIPCHK(hipEventRecord(tr.b, stream));
            h->pending.push_back(tr);
        }
        if (laned && od.kind == OP_DECODE_OUT) {
            if (h->evDecode.size() <= nDecode) {
                hipEvent_t e;
                HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                h->evDecode.push_back(e);
            }
            HIPCHK(hipEventRecord(h->evDecode[nDecode], stream));
            lastDecodeLane = od.lane;
            ++nDecode;
        }
    }
    for (int l = 1; l < kMaxLanes; ++l)
        if (forked[l]) {
            HIPCHK(hipEventRecord(h->evJoin[l], laneOf[l]));
            HIPCHK(hipStreamWaitEvent(laneOf[0], h->evJoin[l], 0));
        }
    return 0;
}

// process-wide switches of this file, read once (vsr_switch_state reports them; vsr_amd/switches.py holds the same defaults)
static bool switch_rows_on() { static const bool v = [] { const char* e = getenv("VSR_DECODE_ROWS"); return !(e && atoi(e) == 0); }(); return v; }
static bool switch_cols_on() { static const bool v = [] { const char* e = getenv("VSR_DECODE_COLS"); return e && atoi(e) == 1; }(); return v; }

static int collect_timing(vsr_sttn* h, hipStream_t stream)
{
    if (h->pending.empty()) return 0;
    HIPCHK(hipStreamSynchronize(stream));
    for (TimingRec& tr : h->pending) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, tr.a, tr.b));
        for (const std::string* key : {&tr.tag, &tr.kernel}) {
            auto& acc = h->timed[*key];
            acc.first += ms;
            acc.second.first += 1;
            acc.second.second += tr.flops;
        }
        (void)hipEventDestroy(tr.a);
        (void)hipEventDestroy(tr.b);
    }
    h->pending.clear();
    return 0;
}

static int need_gpu(vsr_sttn* h)
{
    if (!h) return fail(VSR_ERR_ARG, "null handle");
    if (!h->finalized || h->device < 0)
        return fail(VSR_ERR_NOGPU, "model is not finalized on a HIP device (no GPU / finalize(device<0)); there is no CPU fallback");
    HIPCHK(hipSetDevice(h->device));
    return 0;
}

static int get_strip_tables(vsr_sttn* h, int W, int sh, StripTables** out)
{
    auto key = std::make_pair(W, sh);
    auto it = h->strips.find(key);
    if (it != h->strips.end()) { *out = &it->second; return 0; }
    const int mw = h->model.g.modelW, mh = h->model.g.modelH;
    std::vector<int32_t> dxo, dyo, uxo, uyo;
    std::vector<int16_t> dxa, dya, uxa, uya;
    std::vector<float> dxf, dyf, uxf, uyf;
    cv2_linear_tables(W, mw, true, dxo, dxa, dxf);
    cv2_linear_tables(sh, mh, false, dyo, dya, dyf);
    cv2_linear_tables(mw, W, true, uxo, uxa, uxf);
    cv2_linear_tables(mh, sh, false, uyo, uya, uyf);
    // layout (bytes, each 16-aligned)
    auto al = [](size_t v) { return (v + 15) / 16 * 16; };
    size_t o = 0;
    const size_t o_dxo = o; o += al(dxo.size() * 4);
    const size_t o_dyo = o; o += al(dyo.size() * 4);
    const size_t o_uxo = o; o += al(uxo.size() * 4);
    const size_t o_uyo = o; o += al(uyo.size() * 4);
    const size_t o_dxa = o; o += al(dxa.size() * 2);
    const size_t o_dya = o; o += al(dya.size() * 2);
    const size_t o_uxa = o; o += al(uxa.size() * 2);
    const size_t o_uya = o; o += al(uya.size() * 2);
    const size_t o_uxf = o; o += al(uxf.size() * 4);
    const size_t o_uyf = o; o += al(uyf.size() * 4);
    std::vector<char> hb(o, 0);
    memcpy(hb.data() + o_dxo, dxo.data(), dxo.size() * 4);
    memcpy(hb.data() + o_dyo, dyo.data(), dyo.size() * 4);
    memcpy(hb.data() + o_uxo, uxo.data(), uxo.size() * 4);
    memcpy(hb.data() + o_uyo, uyo.data(), uyo.size() * 4);
    memcpy(hb.data() + o_dxa, dxa.data(), dxa.size() * 2);
    memcpy(hb.data() + o_dya, dya.data(), dya.size() * 2);
    memcpy(hb.data() + o_uxa, uxa.data(), uxa.size() * 2);
    memcpy(hb.data() + o_uya, uya.data(), uya.size() * 2);
    memcpy(hb.data() + o_uxf, uxf.data(), uxf.size() * 4);
    memcpy(hb.data() + o_uyf, uyf.data(), uyf.size() * 4);
    StripTables st;
    HIPCHK(hipMalloc(&st.dev, o));
    HIPCHK(hipMemcpy(st.dev, hb.data(), o, hipMemcpyHostToDevice));
    char* d = (char*)st.dev;
    st.dxofs = (const int32_t*)(d + o_dxo); st.dyofs = (const int32_t*)(d + o_dyo);
    st.uxofs = (const int32_t*)(d + o_uxo); st.uyofs = (const int32_t*)(d + o_uyo);
    st.dialpha = (const int16_t*)(d + o_dxa); st.dibeta = (const int16_t*)(d + o_dya);
    st.uialpha = (const int16_t*)(d + o_uxa); st.uibeta = (const int16_t*)(d + o_uya);
    st.ufalpha = (const float*)(d + o_uxf); st.ufbeta = (const float*)(d + o_uyf);
    h->strips[key] = st;
    *out = &h->strips[key];
    return 0;
}

// =========================================================================================
extern "C" {

int vsr_version(void) { return 100; }
const char* vsr_last_error(void) { return g_err.c_str(); }

int vsr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int vsr_sttn_create(int variant, vsr_sttn_t** out)
{
    if (!out || (variant != VSR_VARIANT_STTN_AUTO && variant != VSR_VARIANT_STTN_DET)) return fail(VSR_ERR_ARG, "bad variant");
    *out = new vsr_sttn(variant);
    return 0;
}

int vsr_sttn_set_param(vsr_sttn_t* h, const char* key, const float* data, const int64_t* shape, int ndim)
{
    if (!h || !key || !data || !shape || ndim <= 0 || ndim > 4) return fail(VSR_ERR_ARG, "bad argument");
    if (h->finalized) return fail(VSR_ERR_STATE, "model already finalized");
    std::string err;
    if (!h->model.set_param(key, data, shape, ndim, err)) return fail(VSR_ERR_ARG, err);
    return 0;
}

int vsr_sttn_finalize(vsr_sttn_t* h, int device)
{
    if (!h) return fail(VSR_ERR_ARG, "null handle");
    if (h->finalized) return fail(VSR_ERR_STATE, "model already finalized");
    std::string err;
    if (!h->model.pack(err)) return fail(VSR_ERR_ARG, err);
    if (device >= 0) {
        if (device >= vsr_device_count()) return fail(VSR_ERR_NOGPU, "no such HIP device; there is no CPU fallback");
        HIPCHK(hipSetDevice(device));
        const size_t bytes = h->model.packed.size() * sizeof(float);
        HIPCHK(hipMalloc(&h->bufs[BUF_WEIGHTS], bytes));
        HIPCHK(hipMemcpy(h->bufs[BUF_WEIGHTS], h->model.packed.data(), bytes, hipMemcpyHostToDevice));
        h->cap[BUF_WEIGHTS] = (int64_t)h->model.packed.size();
    }
    h->device = device;
    h->finalized = true;
    return 0;
}

void vsr_sttn_destroy(vsr_sttn_t* h)
{
    if (!h) return;
    if (h->device >= 0) {
        (void)hipSetDevice(h->device);
        (void)hipDeviceSynchronize();
        h->plans.clear();
        for (int b = 0; b < BUF_COUNT; ++b)
            if (h->bufs[b]) (void)hipFree(h->bufs[b]);
        for (auto& kv : h->strips)
            if (kv.second.dev) (void)hipFree(kv.second.dev);
        if (h->compAreas) (void)hipFree(h->compAreas);
        if (h->dRangeFlag) (void)hipFree(h->dRangeFlag);
        if (h->weightsSplit) (void)hipFree(h->weightsSplit);
        if (h->dSel) (void)hipFree(h->dSel);
        for (hipEvent_t e : h->evDecode) (void)hipEventDestroy(e);
        if (h->evFork) (void)hipEventDestroy(h->evFork);
        for (int l = 1; l < kMaxLanes; ++l) {
            if (h->evJoin[l]) (void)hipEventDestroy(h->evJoin[l]);
            if (h->laneStream[l]) (void)hipStreamDestroy(h->laneStream[l]);
        }
    }
    delete h;
}

int vsr_sttn_geometry(const vsr_sttn_t* h, int32_t* mw, int32_t* mh, int32_t* ns, int32_t* rl)
{
    if (!h) return fail(VSR_ERR_ARG, "null handle");
    if (mw) *mw = h->model.g.modelW;
    if (mh) *mh = h->model.g.modelH;
    if (ns) *ns = h->model.g.neighborStride;
    if (rl) *rl = h->model.g.refLength;
    return 0;
}

int vsr_sttn_set_window(vsr_sttn_t* h, int neighbor_stride, int ref_length)
{
    if (!h || neighbor_stride <= 0 || ref_length <= 0) return fail(VSR_ERR_ARG, "bad window");
    if (h->device >= 0) { (void)hipSetDevice(h->device); (void)hipDeviceSynchronize(); }
    h->plans.clear();
    h->model.g.neighborStride = neighbor_stride;
    h->model.g.refLength = ref_length;
    return 0;
}

int64_t vsr_sttn_packed_weights(const vsr_sttn_t* h, float* out, int64_t capacity)
{
    if (!h || !h->model.packed_ready()) return fail(VSR_ERR_STATE, "model not packed");
    const int64_t n = (int64_t)h->model.packed.size();
    if (out && capacity >= n) memcpy(out, h->model.packed.data(), (size_t)n * sizeof(float));
    return n;
}

// split-half mode: clear the range flag before the GEMMs run, read it back afterwards (one 4-byte copy and
// a stream sync per chunk); `fired` tells the caller to recompute with the exact fp32 plan
static int guard_begin(vsr_sttn* h, hipStream_t stream)
{
    if (!h->precision) return 0;
    if (!h->dRangeFlag) HIPCHK(hipMalloc((void**)&h->dRangeFlag, sizeof(unsigned int)));
    HIPCHK(hipMemsetAsync(h->dRangeFlag, 0, sizeof(unsigned int), stream));
    return 0;
}
static int guard_end(vsr_sttn* h, hipStream_t stream, bool* fired)
{
    *fired = false;
    if (!h->precision) return 0;
    unsigned int v = 0;
    HIPCHK(hipMemcpyAsync(&v, h->dRangeFlag, sizeof(v), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    if (v) { *fired = true; h->fallbacks++; }
    return 0;
}

static int inpaint_common(vsr_sttn* h, const uint8_t* frames_dev, const uint8_t* masks_dev, int L, float* comp_dev,
                          int32_t* counts, hipStream_t stream)
{
    PlanDev* pd = nullptr;
    RCCHK(build_plan_dev(h, L, h->precision, &pd));
    const Geometry& g = h->model.g;
    const size_t n = (size_t)L * g.modelH * g.modelW * 3;
    HIPCHK(hipMemcpyAsync(h->bufs[BUF_IN_U8], frames_dev, n, hipMemcpyDeviceToDevice, stream));
    if (masks_dev)
        HIPCHK(hipMemcpyAsync(h->bufs[BUF_MASK_U8], masks_dev, n / 3, hipMemcpyDeviceToDevice, stream));
    RCCHK(guard_begin(h, stream));
    RCCHK(run_plan(h, pd, stream));
    bool fired = false;
    RCCHK(guard_end(h, stream, &fired));
    if (fired) {                                   // out of fp16 range somewhere: exact fp32 kernels
        RCCHK(build_plan_dev(h, L, 0, &pd));
        RCCHK(run_plan(h, pd, stream));
    }
    HIPCHK(hipMemcpyAsync(comp_dev, h->bufs[BUF_COMP], n * sizeof(float), hipMemcpyDeviceToDevice, stream));
    if (counts) memcpy(counts, pd->plan->compCount.data(), (size_t)L * sizeof(int32_t));
    if (h->timing) RCCHK(collect_timing(h, stream));
    return 0;
}

int vsr_sttn_inpaint(vsr_sttn_t* h, const uint8_t* frames_dev, int L, float* comp_dev, int32_t* counts, void* stream_)
{
    RCCHK(need_gpu(h));
    if (!frames_dev || !comp_dev || L <= 0) return fail(VSR_ERR_ARG, "bad argument");
    if (h->model.g.variant != VSR_VARIANT_STTN_AUTO) return fail(VSR_ERR_STATE, "sttn-det model: use vsr_sttn_det_inpaint (it needs the masks)");
    return inpaint_common(h, frames_dev, nullptr, L, comp_dev, counts, (hipStream_t)stream_);
}

int vsr_sttn_det_inpaint(vsr_sttn_t* h, const uint8_t* frames_dev, const uint8_t* masks_dev, int L, float* comp_dev,
                         int32_t* counts, void* stream_)
{
    RCCHK(need_gpu(h));
    if (!frames_dev || !masks_dev || !comp_dev || L <= 0) return fail(VSR_ERR_ARG, "bad argument");
    if (h->model.g.variant != VSR_VARIANT_STTN_DET) return fail(VSR_ERR_STATE, "not an sttn-det model");
    return inpaint_common(h, frames_dev, masks_dev, L, comp_dev, counts, (hipStream_t)stream_);
}

// shared body of the strip-level entries: crop + resize down, inpaint, resize up + write back.
//   det == false: sttn-auto (mask01 thresholded, blend only where the mask is set)
//   det == true : sttn-det  (raw 0..255 mask resized with the frames, whole strip overwritten)
//   maskRows (sttn-auto, nullable): per area, the strip rows [lo, hi) outside which the caller's mask is zero.  The strip is blended
//   back only where the mask is set (vsr_launch_upscale_blend), so only the model-resolution rows those strip rows are resized
//   from are ever read: the decoder computes them and what they depend on (Plan::decLo), the same values as before.
// The model columns [*lo, *hi) (whole groups of eight) that the horizontal taps of the resize back to W columns read for the frame
// columns [c0, c1); 0, 0 = no restriction (an empty or out-of-range promise).
static void model_cols_of_mask(bool det, int mw, int W, int c0, int c1, int* lo, int* hi)
{
    *lo = *hi = 0;
    if (c0 < 0 || c1 > W || c0 >= c1) return;
    std::vector<int32_t> ofs;
    std::vector<int16_t> ic;
    std::vector<float> fc;
    int xl = mw, xh = 0;
    if (!det) {
        cv2_linear_tables(mw, W, true, ofs, ic, fc);
        for (int dx = c0; dx < c1; ++dx) {
            const int a = ofs[dx] < 0 ? 0 : (ofs[dx] < mw ? ofs[dx] : mw - 1), b = a + 1 < mw ? a + 1 : mw - 1;
            xl = a < xl ? a : xl;
            xh = b + 1 > xh ? b + 1 : xh;
        }
    } else {
        cv2_linear_tables(W, mw, true, ofs, ic, fc);
        for (int x = 0; x < mw; ++x) {
            const int a = ofs[x] < 0 ? 0 : (ofs[x] < W ? ofs[x] : W - 1), b = a + 1 < W ? a + 1 : W - 1;
            if (b >= c0 && a < c1) { xl = x < xl ? x : xl; xh = x + 1 > xh ? x + 1 : xh; }
        }
    }
    if (xh > xl) { *lo = xl / 8 * 8; *hi = (xh + 7) / 8 * 8 < mw ? (xh + 7) / 8 * 8 : mw; }
}

static int strips_common(vsr_sttn* h, bool det, uint8_t* frames_dev, int L, int H, int W, const uint8_t* mask_dev, int n_areas,
                         const int32_t* areas, const int32_t* sel, int nsel, hipStream_t stream, const int32_t* maskRows = nullptr,
                         const int32_t* maskCols = nullptr)
{
    if (!frames_dev || !mask_dev || L <= 0 || H <= 0 || W <= 0 || n_areas < 0 || (n_areas > 0 && !areas))
        return fail(VSR_ERR_ARG, "bad argument");
    if (n_areas == 0) return 0;
    const Geometry& g = h->model.g;
    const int mw = g.modelW, mh = g.modelH;
    const int Ls = (sel && nsel > 0) ? nsel : L;
    const int32_t* dSel = nullptr;
    if (sel && nsel > 0) {
        for (int i = 0; i < nsel; ++i)
            if (sel[i] < 0 || sel[i] >= L) return fail(VSR_ERR_ARG, "frame selection out of range");
        if (h->dSelCap < nsel) {
            if (h->dSel) HIPCHK(hipFree(h->dSel));
            HIPCHK(hipMalloc((void**)&h->dSel, (size_t)nsel * sizeof(int32_t)));
            h->dSelCap = nsel;
        }
        HIPCHK(hipMemcpyAsync(h->dSel, sel, (size_t)nsel * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream)); // `sel` is caller memory
        dSel = h->dSel;
    }
    // model-resolution rows every area needs (0, 0 = all)
    const bool rowsOn = switch_rows_on();
    // (columns: a caller opts in by handing over mask_cols (the _box entry points) AND this side must have been started with
    // VSR_DECODE_COLS=1 -- the same rule as vsr_amd/switches.py, so that a direct C caller gets the same default as the Python one)
    const bool colsOn = switch_cols_on();
    std::vector<int> decLo((size_t)n_areas, 0), decHi((size_t)n_areas, 0), decXLo((size_t)n_areas, 0), decXHi((size_t)n_areas, 0);
    if (maskRows && rowsOn) {
        for (int k = 0; k < n_areas; ++k) {
            const int sh = areas[4 * k + 1] - areas[4 * k];
            const int r0 = maskRows[2 * k], r1 = maskRows[2 * k + 1];
            if (sh <= 0 || r0 < 0 || r1 > sh || r0 >= r1) continue;          // no promise for this strip: the whole image
            std::vector<int32_t> ofs;
            std::vector<int16_t> ic;
            std::vector<float> fc;
            int lo = mh, hi = 0;
            if (!det) {
                // sttn-auto: the strip is blended back where the mask is set -- the model rows the vertical taps of the resize back
                // (k_upscale_blend) read for those strip rows
                cv2_linear_tables(mh, sh, false, ofs, ic, fc);
                auto clampRow = [&](int y) { return y < 0 ? 0 : (y < mh ? y : mh - 1); };
                for (int dy = r0; dy < r1; ++dy) {
                    const int a = clampRow(ofs[dy]), b = clampRow(ofs[dy] + 1);
                    lo = a < lo ? a : lo;
                    hi = b + 1 > hi ? b + 1 : hi;
                }
            } else {
                // sttn-det: the prediction is taken where the RESIZED mask is non-zero (sttn_det_inpaint.py:132,168), at model
                // resolution -- the model rows one of whose two source rows (the vertical taps of the resize down) is a mask row
                cv2_linear_tables(sh, mh, false, ofs, ic, fc);
                auto clampRow = [&](int y) { return y < 0 ? 0 : (y < sh ? y : sh - 1); };
                for (int y = 0; y < mh; ++y) {
                    const int a = clampRow(ofs[y]), b = clampRow(ofs[y] + 1);
                    if (b >= r0 && a < r1) { lo = y < lo ? y : lo; hi = y + 1 > hi ? y + 1 : hi; }
                }
            }
            // whole groups of four rows: every distinct (L, range) is a plan of its own (40-60 MB of offset tables, 30 ms to build),
            // and the detected boxes of a video differ by a few pixels from interval to interval
            if (hi > lo) { decLo[k] = lo / 4 * 4; decHi[k] = (hi + 3) / 4 * 4 < mh ? (hi + 3) / 4 * 4 : mh; }
            if (hi > lo && maskCols && colsOn) {
                // the same along x: sttn-auto, the model columns the horizontal taps of the resize back read for the mask's columns;
                // sttn-det, the model columns one of whose two source columns (the taps of the resize down) is a mask column
                model_cols_of_mask(det, mw, W, maskCols[2 * k], maskCols[2 * k + 1], &decXLo[k], &decXHi[k]);
            }
        }
    }
    PlanDev* pd = nullptr;
    RCCHK(build_plan_dev(h, Ls, h->precision, &pd, decLo[0], decHi[0], decXLo[0], decXHi[0]));
    const int64_t compElems = (int64_t)Ls * mh * mw * 3;
    if (n_areas > 1 && h->compAreasCap < compElems * n_areas) {
        if (h->compAreas) HIPCHK(hipFree(h->compAreas));
        HIPCHK(hipMalloc((void**)&h->compAreas, (size_t)(compElems * n_areas) * sizeof(float)));
        h->compAreasCap = compElems * n_areas;
    }
    const int64_t frameStride = (int64_t)H * W * 3;
    // pass 1: every strip is cropped from the ORIGINAL frames and inpainted (reference
    // sttn_auto_inpaint.py:257-283 / sttn_det_inpaint.py:66-82), pass 2 writes the strips back in order
    for (int attempt = 0; attempt < 2; ++attempt) {
    RCCHK(guard_begin(h, stream));
    for (int k = 0; k < n_areas; ++k) {
        const int ymin = areas[4 * k], ymax = areas[4 * k + 1];
        const int sh = ymax - ymin;
        if (ymin < 0 || ymax > H || sh <= 0) return fail(VSR_ERR_ARG, "inpaint area outside the frame");
        StripTables* st = nullptr;
        RCCHK(get_strip_tables(h, W, sh, &st));
        if (vsr_launch_resize_u8(frames_dev + (int64_t)ymin * W * 3, frameStride, W * 3, W, sh, (uint8_t*)h->bufs[BUF_IN_U8], mw,
                                 mh, Ls, 3, dSel, st->dxofs, st->dialpha, st->dyofs, st->dibeta, stream) != 0)
            return fail(VSR_ERR_HIP, "resize launch failed");
        if (det) // cv2.resize(mask_crop, (432, 240)) -- the same strip mask for every frame (frame stride 0)
            if (vsr_launch_resize_u8(mask_dev + (int64_t)ymin * W, 0, W, W, sh, (uint8_t*)h->bufs[BUF_MASK_U8], mw, mh, Ls, 1,
                                     nullptr, st->dxofs, st->dialpha, st->dyofs, st->dibeta, stream) != 0)
                return fail(VSR_ERR_HIP, "mask resize launch failed");
        if (k > 0 || attempt > 0) RCCHK(build_plan_dev(h, Ls, attempt ? 0 : h->precision, &pd, decLo[k], decHi[k], decXLo[k], decXHi[k]));
        RCCHK(run_plan(h, pd, stream));
        if (n_areas > 1)
            HIPCHK(hipMemcpyAsync(h->compAreas + compElems * k, h->bufs[BUF_COMP], (size_t)compElems * sizeof(float),
                                  hipMemcpyDeviceToDevice, stream));
    }
    bool fired = false;
    if (attempt == 0) RCCHK(guard_end(h, stream, &fired));
    if (!fired) break;                             // (fired: the frames are still untouched, pass 1 is redone in exact fp32)
    }
    for (int k = 0; k < n_areas; ++k) {
        const int ymin = areas[4 * k], ymax = areas[4 * k + 1];
        const int sh = ymax - ymin;
        StripTables* st = nullptr;
        RCCHK(get_strip_tables(h, W, sh, &st));
        const float* comp = n_areas > 1 ? h->compAreas + compElems * k : (const float*)h->bufs[BUF_COMP];
        if (vsr_launch_upscale_blend(comp, mw, mh, pd->dIsFloat, frames_dev + (int64_t)ymin * W * 3, frameStride, W * 3, dSel,
                                     det ? nullptr : mask_dev + (int64_t)ymin * W, W, W, sh, Ls, st->uxofs, st->uialpha,
                                     st->ufalpha, st->uyofs, st->uibeta, st->ufbeta, stream) != 0)
            return fail(VSR_ERR_HIP, "blend launch failed");
    }
    if (h->timing) RCCHK(collect_timing(h, stream));
    return 0;
}

int vsr_sttn_auto_chunk(vsr_sttn_t* h, uint8_t* frames_dev, int L, int H, int W, const uint8_t* mask_dev, int n_areas,
                        const int32_t* areas, const int32_t* sel, int nsel, void* stream_)
{
    RCCHK(need_gpu(h));
    if (h->model.g.variant != VSR_VARIANT_STTN_AUTO) return fail(VSR_ERR_STATE, "not an sttn-auto model");
    return strips_common(h, false, frames_dev, L, H, W, mask_dev, n_areas, areas, sel, nsel, (hipStream_t)stream_);
}

int vsr_sttn_auto_chunk_rows(vsr_sttn_t* h, uint8_t* frames_dev, int L, int H, int W, const uint8_t* mask_dev, int n_areas,
                             const int32_t* areas, const int32_t* mask_rows, const int32_t* sel, int nsel, void* stream_)
{
    RCCHK(need_gpu(h));
    if (h->model.g.variant != VSR_VARIANT_STTN_AUTO) return fail(VSR_ERR_STATE, "not an sttn-auto model");
    return strips_common(h, false, frames_dev, L, H, W, mask_dev, n_areas, areas, sel, nsel, (hipStream_t)stream_, mask_rows);
}

int vsr_sttn_auto_chunk_box(vsr_sttn_t* h, uint8_t* frames_dev, int L, int H, int W, const uint8_t* mask_dev, int n_areas,
                            const int32_t* areas, const int32_t* mask_rows, const int32_t* mask_cols, const int32_t* sel, int nsel, void* stream_)
{
    RCCHK(need_gpu(h));
    if (h->model.g.variant != VSR_VARIANT_STTN_AUTO) return fail(VSR_ERR_STATE, "not an sttn-auto model");
    return strips_common(h, false, frames_dev, L, H, W, mask_dev, n_areas, areas, sel, nsel, (hipStream_t)stream_, mask_rows, mask_cols);
}

double vsr_sttn_flops_rows(vsr_sttn_t* h, int L, int row_lo, int row_hi)
{
    if (!h || !h->model.packed_ready() || L <= 0) { fail(VSR_ERR_ARG, "bad argument"); return -1.0; }
    try {
        Plan p(h->model, L, 0, 1, row_lo, row_hi);
        return p.flops;
    } catch (const std::exception& e) {
        fail(VSR_ERR_ARG, e.what());
        return -1.0;
    }
}

double vsr_sttn_flops_box(vsr_sttn_t* h, int L, int row_lo, int row_hi, int col_lo, int col_hi)
{
    if (!h || !h->model.packed_ready() || L <= 0) { fail(VSR_ERR_ARG, "bad argument"); return -1.0; }
    try {
        Plan p(h->model, L, 0, 1, row_lo, row_hi, col_lo, col_hi);
        return p.flops;
    } catch (const std::exception& e) {
        fail(VSR_ERR_ARG, e.what());
        return -1.0;
    }
}

int vsr_sttn_decode_cols(vsr_sttn_t* h, int frame_w, int mask_col_lo, int mask_col_hi, int32_t* col_lo, int32_t* col_hi)
{
    if (!h || !col_lo || !col_hi || frame_w <= 0 || mask_col_lo < 0 || mask_col_hi > frame_w || mask_col_lo >= mask_col_hi)
        return fail(VSR_ERR_ARG, "bad argument");
    int lo = 0, hi = 0;
    model_cols_of_mask(h->model.g.variant == VSR_VARIANT_STTN_DET, h->model.g.modelW, frame_w, mask_col_lo, mask_col_hi, &lo, &hi);
    *col_lo = lo; *col_hi = hi;
    return 0;
}

int vsr_sttn_decode_rows(vsr_sttn_t* h, int strip_h, int mask_row_lo, int mask_row_hi, int32_t* row_lo, int32_t* row_hi)
{
    if (!h || !row_lo || !row_hi || strip_h <= 0 || mask_row_lo < 0 || mask_row_hi > strip_h || mask_row_lo >= mask_row_hi)
        return fail(VSR_ERR_ARG, "bad argument");
    const int mh = h->model.g.modelH;
    std::vector<int32_t> ofs;
    std::vector<int16_t> ic;
    std::vector<float> fc;
    int lo = mh, hi = 0;
    if (h->model.g.variant != VSR_VARIANT_STTN_DET) {    // the same two rules as strips_common
        cv2_linear_tables(mh, strip_h, false, ofs, ic, fc);
        auto clampRow = [&](int y) { return y < 0 ? 0 : (y < mh ? y : mh - 1); };
        for (int dy = mask_row_lo; dy < mask_row_hi; ++dy) {
            const int a = clampRow(ofs[dy]), b = clampRow(ofs[dy] + 1);
            lo = a < lo ? a : lo;
            hi = b + 1 > hi ? b + 1 : hi;
        }
    } else {
        cv2_linear_tables(strip_h, mh, false, ofs, ic, fc);
        auto clampRow = [&](int y) { return y < 0 ? 0 : (y < strip_h ? y : strip_h - 1); };
        for (int y = 0; y < mh; ++y) {
            const int a = clampRow(ofs[y]), b = clampRow(ofs[y] + 1);
            if (b >= mask_row_lo && a < mask_row_hi) { lo = y < lo ? y : lo; hi = y + 1 > hi ? y + 1 : hi; }
        }
    }
    if (hi <= lo) { lo = 0; hi = mh; }
    else { lo = lo / 4 * 4; hi = (hi + 3) / 4 * 4 < mh ? (hi + 3) / 4 * 4 : mh; }      // as strips_common
    int wlo = 0, whi = 0, wxlo = 0, wxhi = 0;               // (the widening to whole output-conv blocks, as Plan::Plan does it)
    Plan::decoder_bounds(h->model.g, 0, lo, hi, 0, 0, &wlo, &whi, &wxlo, &wxhi);
    *row_lo = wlo; *row_hi = whi;
    return 0;
}

int vsr_sttn_det_batch(vsr_sttn_t* h, uint8_t* frames_dev, int L, int H, int W, const uint8_t* mask_dev, int n_areas,
                       const int32_t* areas, void* stream_)
{
    RCCHK(need_gpu(h));
    if (h->model.g.variant != VSR_VARIANT_STTN_DET) return fail(VSR_ERR_STATE, "not an sttn-det model");
    return strips_common(h, true, frames_dev, L, H, W, mask_dev, n_areas, areas, nullptr, 0, (hipStream_t)stream_);
}

int vsr_sttn_det_batch_rows(vsr_sttn_t* h, uint8_t* frames_dev, int L, int H, int W, const uint8_t* mask_dev, int n_areas,
                            const int32_t* areas, const int32_t* mask_rows, void* stream_)
{
    RCCHK(need_gpu(h));
    if (h->model.g.variant != VSR_VARIANT_STTN_DET) return fail(VSR_ERR_STATE, "not an sttn-det model");
    return strips_common(h, true, frames_dev, L, H, W, mask_dev, n_areas, areas, nullptr, 0, (hipStream_t)stream_, mask_rows);
}

int vsr_sttn_det_batch_box(vsr_sttn_t* h, uint8_t* frames_dev, int L, int H, int W, const uint8_t* mask_dev, int n_areas,
                           const int32_t* areas, const int32_t* mask_rows, const int32_t* mask_cols, void* stream_)
{
    RCCHK(need_gpu(h));
    if (h->model.g.variant != VSR_VARIANT_STTN_DET) return fail(VSR_ERR_STATE, "not an sttn-det model");
    return strips_common(h, true, frames_dev, L, H, W, mask_dev, n_areas, areas, nullptr, 0, (hipStream_t)stream_, mask_rows, mask_cols);
}

int vsr_sttn_set_precision(vsr_sttn_t* h, int mode)
{
    if (!h || mode < 0 || mode > 3)
        return fail(VSR_ERR_ARG, "precision mode must be 0 (f32), 1 (split-half f16 MFMA), 2 (split-half on split-format tensors) or 3 (fp16 operands)");
    h->precision = mode;
    return 0;
}

int vsr_sttn_set_lanes(vsr_sttn_t* h, int lanes)
{
    if (!h || lanes < 1 || lanes > kMaxLanes) return fail(VSR_ERR_ARG, "lanes must be 1 .. 4");
    h->lanes = lanes;                  // plans are cached per (L, precision, lanes)
    return 0;
}

int64_t vsr_sttn_fallbacks(const vsr_sttn_t* h) { return h ? h->fallbacks : -1; }

double vsr_sttn_flops(vsr_sttn_t* h, int L)
{
    if (!h || !h->model.packed_ready() || L <= 0) { fail(VSR_ERR_ARG, "bad argument"); return -1.0; }
    try {
        Plan p(h->model, L);
        return p.flops;
    } catch (const std::exception& e) {
        fail(VSR_ERR_ARG, e.what());
        return -1.0;
    }
}

double vsr_sttn_flops_reference(vsr_sttn_t* h, int L)
{
    if (!h || !h->model.packed_ready() || L <= 0) { fail(VSR_ERR_ARG, "bad argument"); return -1.0; }
    try {
        Plan p(h->model, L);
        return p.refFlops;
    } catch (const std::exception& e) {
        fail(VSR_ERR_ARG, e.what());
        return -1.0;
    }
}

int vsr_switch_state(const char* name)
{
    if (!name) return -1;
    const std::string n(name);
    if (n == "VSR_DECODE_ROWS") return switch_rows_on() ? 1 : 0;
    if (n == "VSR_DECODE_COLS") return switch_cols_on() ? 1 : 0;
    if (n == "VSR_QKV0_SHARED") return Tuning::get(0).shareQkv0 ? 1 : 0;
    if (n == "VSR_TRIM_LAST_BLOCK") return Tuning::get(0).trimLastBlock ? 1 : 0;
    return -1;
}

int vsr_sttn_timing(vsr_sttn_t* h, int enable)
{
    if (!h) return fail(VSR_ERR_ARG, "null handle");
    h->timing = enable < 0 ? 0 : (enable > 2 ? 1 : enable);
    return 0;
}

int vsr_sttn_timing_reset(vsr_sttn_t* h)
{
    if (!h) return fail(VSR_ERR_ARG, "null handle");
    h->timed.clear();
    return 0;
}

int vsr_sttn_timing_get(vsr_sttn_t* h, const char* prefix, double* total_ms, int32_t* launches, double* flops)
{
    if (!h || !prefix) return fail(VSR_ERR_ARG, "bad argument");
    double ms = 0, fl = 0;
    int n = 0;
    const size_t pl = strlen(prefix);
    for (const auto& kv : h->timed)
        if (kv.first.compare(0, pl, prefix) == 0) { ms += kv.second.first; n += kv.second.second.first; fl += kv.second.second.second; }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    if (flops) *flops = fl;
    return 0;
}

// ---- kernel-level entry points -----------------------------------------------------------
static void tile_dims(int cfg, int& BM, int& BN)
{
    BM = (cfg == VSR_TILE_128x128 || cfg == VSR_TILE_128x64) ? 128 : 256;
    BN = cfg == VSR_TILE_256x256 ? 256 : (cfg == VSR_TILE_128x128 || cfg == VSR_TILE_256x128) ? 128 : ((cfg == VSR_TILE_256x64 || cfg == VSR_TILE_128x64) ? 64 : 32);
}

static int run_gather_gemm_variant(const GGProblem* probs, int nprobs, int tile_cfg, int bmode, int variant, void* stream_);

int vsr_run_gather_gemm(const GGProblem* probs, int nprobs, int tile_cfg, int bmode, void* stream_)
{
    return run_gather_gemm_variant(probs, nprobs, tile_cfg, bmode, gg_variant(bmode), stream_);
}

int vsr_run_gather_gemm_variant(const GGProblem* probs, int nprobs, int tile_cfg, int bmode, int variant, void* stream_)
{
    if (variant != (1 | VSR_VARIANT_A_EXP) && variant != VSR_VARIANT_NARROW && (variant < 1 || variant > 6))
        return fail(VSR_ERR_ARG, "kernel variant must be 1..6, VSR_VARIANT_NARROW or 1 | VSR_VARIANT_A_EXP");
    return run_gather_gemm_variant(probs, nprobs, tile_cfg, bmode, variant, stream_);
}

// ---- resident gather-GEMM launch lists (vsr_gemm_plan_*): descriptors and tile queues stay on the device, a run is one
// memset + one launch on the caller's stream, nothing synchronises ----
struct vsr_gemm_plan {
    GGProblem* d = nullptr;
    unsigned int* queue = nullptr;
    int nprobs = 0, total = 0, tileCfg = 0, bmode = 0, variant = 3, nQueues = 8, device = 0;
};

int vsr_gemm_plan_create(const GGProblem* probs, int nprobs, int tile_cfg, int bmode, int variant, vsr_gemm_plan_t** out)
{
    if (!probs || nprobs <= 0 || !out) return fail(VSR_ERR_ARG, "bad argument");
    if (variant < 1 || variant > 4) return fail(VSR_ERR_ARG, "kernel variant must be 1..4 (fp32 tensors)");
    if (vsr_device_count() <= 0) return fail(VSR_ERR_NOGPU, "no HIP device; there is no CPU fallback");
    std::vector<GGProblem> hp(probs, probs + nprobs);
    int BM, BN;
    tile_dims(tile_cfg, BM, BN);
    std::unique_ptr<vsr_gemm_plan> p(new vsr_gemm_plan);
    for (auto& q : hp) {
        if (q.K % VSR_GG_KC) return fail(VSR_ERR_ARG, "K must be a multiple of 32");
        if (q.tilesM != (q.M + BM - 1) / BM || q.tilesN != (q.N + BN - 1) / BN) return fail(VSR_ERR_ARG, "tile counts do not match the tile config");
        if (q.splitK < 1 || (int64_t)q.splitK * q.chunksPerSplit < q.K / VSR_GG_KC) return fail(VSR_ERR_ARG, "bad split-K");
        q.tileStart = p->total;
        p->total += q.tilesM * q.tilesN * q.splitK;
        if (q.tilesN > 4) p->nQueues = gg_wide_queues();
    }
    HIPCHK(hipGetDevice(&p->device));
    const size_t descBytes = (hp.size() * sizeof(GGProblem) + 15) / 16 * 16;
    HIPCHK(hipMalloc((void**)&p->d, descBytes + 8 * sizeof(unsigned int)));
    if (hipMemcpy(p->d, hp.data(), hp.size() * sizeof(GGProblem), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(p->d);
        return fail(VSR_ERR_HIP, "gemm plan: descriptor upload failed");
    }
    p->queue = (unsigned int*)((char*)p->d + descBytes);
    p->nprobs = nprobs;
    p->tileCfg = tile_cfg;
    p->bmode = bmode;
    p->variant = variant;
    *out = p.release();
    return 0;
}

__global__ void k_zero_queue(unsigned int* q) { if (threadIdx.x < 8) q[threadIdx.x] = 0u; }

int vsr_gemm_plan_run(vsr_gemm_plan_t* p, void* stream_)
{
    if (!p || !p->d) return fail(VSR_ERR_ARG, "null plan");
    hipStream_t stream = (hipStream_t)stream_;
    // The tile counters are zeroed by a KERNEL, not by hipMemsetAsync: a resident plan is what a caller may capture into a HIP graph
    // (ocr_det.py run_graphed), and a graph holding captured memset nodes faulted on its second replay after eager work on the same
    // stream -- every time with this kernel family, never with kernel nodes only, and a queue that is not zeroed at all is harmless at
    // those sizes, so the faulting write is the memset node's own (profiles/r06_det_graph_triage.log).
    // VSR_PLAN_ZERO_KERNEL=0: hipMemsetAsync as before (reproduces the fault); 2: not zeroed at all (triage only).
    static const int zeroMode = [] { const char* e = getenv("VSR_PLAN_ZERO_KERNEL"); return e ? atoi(e) : 1; }();
    if (p->variant >= 2 && zeroMode == 0) HIPCHK(hipMemsetAsync(p->queue, 0, 8 * sizeof(unsigned int), stream));
    else if (p->variant >= 2 && zeroMode != 2) hipLaunchKernelGGL(k_zero_queue, dim3(1), dim3(64), 0, stream, p->queue);
    const int rc = vsr_launch_gather_gemm_dev(p->d, p->nprobs, p->total, p->tileCfg, p->bmode, p->variant >= 2 ? p->queue : nullptr, p->variant,
                                              p->nQueues, nullptr, stream);
    if (rc != 0) return fail(VSR_ERR_HIP, "gather-gemm launch failed (unsupported tile/bmode?)");
    return 0;
}

void vsr_gemm_plan_destroy(vsr_gemm_plan_t* p)
{
    if (!p) return;
    if (p->d) (void)hipFree(p->d);
    delete p;
}

static int run_gather_gemm_variant(const GGProblem* probs, int nprobs, int tile_cfg, int bmode, int variant, void* stream_)
{
    if (!probs || nprobs <= 0) return fail(VSR_ERR_ARG, "bad argument");
    if (vsr_device_count() <= 0) return fail(VSR_ERR_NOGPU, "no HIP device; there is no CPU fallback");
    hipStream_t stream = (hipStream_t)stream_;
    std::vector<GGProblem> hp(probs, probs + nprobs);
    int BM, BN;
    tile_dims(tile_cfg, BM, BN);
    int total = 0;
    for (auto& p : hp) {
        if (p.K % VSR_GG_KC) return fail(VSR_ERR_ARG, "K must be a multiple of 32");
        if (tile_cfg == VSR_TILE_256x256) {
            // dynamic tile height: any tilesM whose tiles of roundup32(ceil(M / tilesM)) <= BM rows cover M (gather_gemm_v7.h)
            const int rows = p.tilesM > 0 ? (((p.M + p.tilesM - 1) / p.tilesM) + 31) / 32 * 32 : 0;
            if (p.tilesM <= 0 || rows > BM || (int64_t)rows * p.tilesM < p.M || p.tilesN != (p.N + BN - 1) / BN)
                return fail(VSR_ERR_ARG, "256x256 tile: tilesM must give tiles of at most 256 rows that cover M");
        } else
        if (p.tilesM != (p.M + BM - 1) / BM || p.tilesN != (p.N + BN - 1) / BN) return fail(VSR_ERR_ARG, "tile counts do not match the tile config");
        if (p.splitK < 1 || (int64_t)p.splitK * p.chunksPerSplit < p.K / VSR_GG_KC) return fail(VSR_ERR_ARG, "bad split-K");
        p.tileStart = total;
        total += p.tilesM * p.tilesN * p.splitK;
    }
    GGProblem* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, hp.size() * sizeof(GGProblem) + 64));   // + 8 queue counters
    HIPCHK(hipMemcpy(d, hp.data(), hp.size() * sizeof(GGProblem), hipMemcpyHostToDevice));
    unsigned int* queue = nullptr;
    if (variant >= 2) {
        queue = (unsigned int*)((char*)d + (hp.size() * sizeof(GGProblem) + 15) / 16 * 16);
        HIPCHK(hipMemset(queue, 0, 8 * sizeof(unsigned int)));
    }
    int nQueues = 8;
    for (const auto& p : hp)
        if (p.tilesN > 4) nQueues = gg_wide_queues();
    int rc;
    if (variant == VSR_VARIANT_NARROW) {       // the dot-product kernel: what the flow engines check before they pick it (flow_engine.hip)
        int maxN = 0, maxK = 0;
        bool ok = tile_cfg == VSR_TILE_256x32 && bmode == VSR_BMODE_NK;
        for (const auto& p : hp) {
            ok = ok && p.N >= 1 && p.N <= 4 && p.splitK == 1 && p.R == nullptr && (p.act & ~3) == 0;
            maxN = std::max(maxN, (int)p.N);
            maxK = std::max(maxK, (int)p.K);
        }
        ok = ok && (int64_t)(maxN <= 2 ? 2 : 4) * maxK <= vsr_gg_narrow_cap();
        if (!ok) {
            (void)hipFree(d);
            return fail(VSR_ERR_ARG, "VSR_VARIANT_NARROW: NK problems on the 256x32 tile with N <= 4, no split-K, no residual, a plain activation and N * K within vsr_gg_narrow_cap()");
        }
        rc = vsr_launch_gather_gemm_narrow_dev(d, nprobs, total, maxN, maxK, stream);
    } else
    rc = vsr_launch_gather_gemm_dev(d, nprobs, total, tile_cfg, bmode, queue, variant, nQueues, nullptr, stream);
    hipError_t e = hipStreamSynchronize(stream);
    (void)hipFree(d);
    if (rc != 0) return fail(VSR_ERR_HIP, "gather-gemm launch failed (unsupported tile/bmode?)");
    if (e != hipSuccess) return fail(VSR_ERR_HIP, std::string("gather-gemm: ") + hipGetErrorString(e));
    return 0;
}

int vsr_run_softmax(const SMProblem* probs, int nprobs, void* stream_)
{
    if (!probs || nprobs <= 0) return fail(VSR_ERR_ARG, "bad argument");
    if (vsr_device_count() <= 0) return fail(VSR_ERR_NOGPU, "no HIP device; there is no CPU fallback");
    hipStream_t stream = (hipStream_t)stream_;
    std::vector<SMProblem> hp(probs, probs + nprobs);
    int rows = 0;
    for (auto& p : hp) { p.rowStart = rows; rows += (p.M + 3) / 4 * 4; }
    SMProblem* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, hp.size() * sizeof(SMProblem)));
    HIPCHK(hipMemcpy(d, hp.data(), hp.size() * sizeof(SMProblem), hipMemcpyHostToDevice));
    const int rc = vsr_launch_softmax_dev(d, nprobs, rows, stream);
    hipError_t e = hipStreamSynchronize(stream);
    (void)hipFree(d);
    if (rc != 0) return fail(VSR_ERR_HIP, "softmax launch failed");
    if (e != hipSuccess) return fail(VSR_ERR_HIP, std::string("softmax: ") + hipGetErrorString(e));
    return 0;
}

int vsr_cv2_linear_tables(int ssize, int dsize, int clamp_x, int32_t* ofs, int16_t* icoef, float* fcoef)
{
    if (ssize <= 0 || dsize <= 0 || !ofs || !icoef || !fcoef) return fail(VSR_ERR_ARG, "bad argument");
    std::vector<int32_t> o;
    std::vector<int16_t> ic;
    std::vector<float> fc;
    cv2_linear_tables(ssize, dsize, clamp_x != 0, o, ic, fc);
    memcpy(ofs, o.data(), o.size() * 4);
    memcpy(icoef, ic.data(), ic.size() * 2);
    memcpy(fcoef, fc.data(), fc.size() * 4);
    return 0;
}

// ---- plan introspection (host only) ------------------------------------------------------
int vsr_plan_create(const vsr_sttn_t* h, int L, vsr_plan_t** out)
{
    if (!h || !out || L <= 0) return fail(VSR_ERR_ARG, "bad argument");
    if (!h->model.packed_ready()) return fail(VSR_ERR_STATE, "model not finalized");
    try {
        std::unique_ptr<vsr_plan> p(new vsr_plan);
        p->plan.reset(new Plan(h->model, L, 0, h->lanes));
        *out = p.release();
    } catch (const std::exception& e) {
        return fail(VSR_ERR_ARG, std::string("plan: ") + e.what());
    }
    return 0;
}
int vsr_plan_create_rows(const vsr_sttn_t* h, int L, int row_lo, int row_hi, vsr_plan_t** out)
{
    if (!h || !out || L <= 0) return fail(VSR_ERR_ARG, "bad argument");
    if (!h->model.packed_ready()) return fail(VSR_ERR_STATE, "model not finalized");
    try {
        std::unique_ptr<vsr_plan> p(new vsr_plan);
        p->plan.reset(new Plan(h->model, L, 0, h->lanes, row_lo, row_hi));
        *out = p.release();
    } catch (const std::exception& e) {
        return fail(VSR_ERR_ARG, std::string("plan: ") + e.what());
    }
    return 0;
}
int vsr_plan_create_box(const vsr_sttn_t* h, int L, int row_lo, int row_hi, int col_lo, int col_hi, vsr_plan_t** out)
{
    if (!h || !out || L <= 0) return fail(VSR_ERR_ARG, "bad argument");
    if (!h->model.packed_ready()) return fail(VSR_ERR_STATE, "model not finalized");
    try {
        std::unique_ptr<vsr_plan> p(new vsr_plan);
        p->plan.reset(new Plan(h->model, L, 0, h->lanes, row_lo, row_hi, col_lo, col_hi));
        *out = p.release();
    } catch (const std::exception& e) {
        return fail(VSR_ERR_ARG, std::string("plan: ") + e.what());
    }
    return 0;
}
void vsr_plan_destroy(vsr_plan_t* p) { delete p; }
int vsr_plan_num_buffers(const vsr_plan_t* p) { return p ? (int)p->plan->bufElems.size() : 0; }
int64_t vsr_plan_buffer_elems(const vsr_plan_t* p, int buf) { return (p && buf >= 0 && buf < (int)p->plan->bufElems.size()) ? p->plan->bufElems[buf] : -1; }
int vsr_plan_num_tables(const vsr_plan_t* p) { return p ? (int)p->plan->tables.size() : 0; }
int64_t vsr_plan_table_len(const vsr_plan_t* p, int t) { return (p && t >= 0 && t < (int)p->plan->tables.size()) ? (int64_t)p->plan->tables[t].size() : -1; }
int vsr_plan_table_copy(const vsr_plan_t* p, int t, int32_t* out)
{
    if (!p || !out || t < 0 || t >= (int)p->plan->tables.size()) return fail(VSR_ERR_ARG, "bad table");
    memcpy(out, p->plan->tables[t].data(), p->plan->tables[t].size() * sizeof(int32_t));
    return 0;
}
int vsr_plan_num_ops(const vsr_plan_t* p) { return p ? (int)p->plan->ops.size() : 0; }
int vsr_plan_op(const vsr_plan_t* p, int i, VsrOpInfo* o)
{
    if (!p || !o || i < 0 || i >= (int)p->plan->ops.size()) return fail(VSR_ERR_ARG, "bad op");
    const Op& op = p->plan->ops[i];
    memset(o, 0, sizeof(*o));
    o->kind = op.kind;
    o->nitems = (int)(op.kind == OP_GEMM ? op.gemm.size() : op.softmax.size());
    o->tile_cfg = op.tileCfg; o->bmode = op.bmode;
    o->buf_src = op.bufSrc; o->buf_dst = op.bufDst; o->H = op.H; o->W = op.W; o->C = op.C;
    o->halo_src = op.haloS; o->halo_dst = op.haloD; o->n = op.n; o->ldy = op.ldy; o->pix = op.pix;
    o->t_frame_idx = op.tFrameIdx; o->t_first = op.tFirst; o->premask = op.premask;
    o->M = op.M; o->N = op.N; o->nsplit = op.nsplit; o->t_rowC = op.tRowC; o->t_colC = op.tColC;
    o->buf_mask = op.bufMask;
    o->off_src = op.offSrc; o->off_dst = op.offDst; o->split_stride = op.splitStride;
    o->flops = op.flops;
    o->ew = op.ew;
    for (int k = 0; k < 4; ++k) { o->ibuf[k] = op.ibuf[k]; o->ioff[k] = op.ioff[k]; o->fpar[k] = op.fpar[k]; }
    for (int k = 0; k < 16; ++k) o->ipar[k] = op.ipar[k];
    strncpy(o->tag, op.tag.c_str(), sizeof(o->tag) - 1);
    return 0;
}
int vsr_plan_op_lane(const vsr_plan_t* p, int i) { return (p && i >= 0 && i < (int)p->plan->ops.size()) ? p->plan->ops[i].lane : -1; }

int vsr_plan_op_gemm(const vsr_plan_t* p, int i, int j, VsrGemmInfo* o)
{
    if (!p || !o || i < 0 || i >= (int)p->plan->ops.size()) return fail(VSR_ERR_ARG, "bad op");
    const Op& op = p->plan->ops[i];
    if (op.kind != OP_GEMM || j < 0 || j >= (int)op.gemm.size()) return fail(VSR_ERR_ARG, "bad gemm item");
    const GemmItem& g = op.gemm[j];
    o->bufA = g.bufA; o->bufB = g.bufB; o->bufC = g.bufC; o->bufR = g.bufR;
    o->offA = g.offA; o->offB = g.offB; o->offC = g.offC; o->offR = g.offR; o->offBias = g.offBias;
    o->tRowA = g.tRowA; o->tColA = g.tColA; o->tRowB = g.tRowB; o->tColB = g.tColB; o->tRowC = g.tRowC;
    o->tColC = g.tColC; o->tRowR = g.tRowR;
    o->M = g.M; o->N = g.N; o->K = g.K; o->tilesM = g.tilesM; o->tilesN = g.tilesN; o->splitK = g.splitK;
    o->chunksPerSplit = g.chunksPerSplit; o->splitStride = g.splitStride; o->alpha = g.alpha; o->act = g.act;
    return 0;
}
int vsr_plan_op_softmax(const vsr_plan_t* p, int i, int j, VsrSoftmaxInfo* o)
{
    if (!p || !o || i < 0 || i >= (int)p->plan->ops.size()) return fail(VSR_ERR_ARG, "bad op");
    const Op& op = p->plan->ops[i];
    if (op.kind != OP_SOFTMAX || j < 0 || j >= (int)op.softmax.size()) return fail(VSR_ERR_ARG, "bad softmax item");
    const SoftmaxItem& s = op.softmax[j];
    o->bufS = s.bufS; o->bufP = s.bufP; o->offS = s.offS; o->offP = s.offP; o->splitStride = s.splitStride;
    o->M = s.M; o->N = s.N; o->ldS = s.ldS; o->ldP = s.ldP; o->nsplit = s.nsplit; o->scale = s.scale;
    return 0;
}
int vsr_plan_counts(const vsr_plan_t* p, int32_t* counts)
{
    if (!p || !counts) return fail(VSR_ERR_ARG, "bad argument");
    memcpy(counts, p->plan->compCount.data(), p->plan->compCount.size() * sizeof(int32_t));
    return 0;
}
double vsr_plan_flops(const vsr_plan_t* p) { return p ? p->plan->flops : -1.0; }

} // extern "C"
