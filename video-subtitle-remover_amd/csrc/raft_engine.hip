// Device engine + C-ABI (include/vsr_hip.h, "RAFT" section) of the optical-flow path: SURVEY.md section 8(a) row a14,
// reference RAFT_bi.forward (backend/inpaint/video/model/modules/flow_comp_raft.py:39-55).
//
// Same structure as sttn_engine.hip: the workspace is allocated and zeroed once (NHWC fp32 activations whose physical
// zero halos are the conv padding), a vsr::RaftPlan (raft_plan.h) is materialised into device descriptors once per
// (t, H, W, iters) and replayed on the caller's stream without host synchronisation.  Exact fp32 arithmetic
// (v_mfma_f32_32x32x2_f32): the reference runs RAFT in fp32 even in its fp16 mode (propainter_inpaint.py:230).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>
#include "../../include/vsr_hip.h"
#include "gather_gemm.h"
#include "plan_c.h"
#include "raft_kernels.h"
#include "raft_plan.h"

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

struct RaftOpDev {
    const Op* op = nullptr;
    const void* dDesc = nullptr;   // GGProblem* (device)
    int nitems = 0, total = 0, nQueues = 1;
};

struct RaftPlanDev {
    std::unique_ptr<RaftPlan> plan;
    int32_t* dTables = nullptr;
    std::vector<int64_t> toff;
    void* dDescs = nullptr;
    unsigned int* dQueues = nullptr;
    std::vector<RaftOpDev> ops;
    ~RaftPlanDev()
    {
        if (dTables) (void)hipFree(dTables);
        if (dDescs) (void)hipFree(dDescs);
        if (dQueues) (void)hipFree(dQueues);
    }
};

} // namespace

struct vsr_raft {
    RaftModel model;
    int device = -1;
    bool finalized = false;
    void* bufs[RB_COUNT] = {};
    int64_t cap[RB_COUNT] = {};
    double* statAcc = nullptr;
    int64_t statAccCap = 0;
    std::tuple<int, int, int> geom{0, 0, 0};   // (t, H, W) the halos of the workspace are currently laid out for
    std::map<std::tuple<int, int, int, int>, std::unique_ptr<RaftPlanDev>> plans;
};

static int64_t rbBytes(int buf, int64_t elems) { return buf == RB_IN_U8 ? elems : elems * 4; }

static int raft_plan_dev(vsr_raft* h, int t, int H, int W, int iters, RaftPlanDev** out)
{
    const auto key = std::make_tuple(t, H, W, iters);
    auto it = h->plans.find(key);
    if (it != h->plans.end()) { *out = it->second.get(); return 0; }
    std::unique_ptr<RaftPlanDev> pd(new RaftPlanDev);
    try {
        pd->plan.reset(new RaftPlan(h->model, t, H, W, iters));
    } catch (const std::exception& e) {
        return rfail(VSR_ERR_ARG, std::string("raft plan: ") + e.what());
    }
    const RaftPlan& P = *pd->plan;
    bool grow = false;
    for (int b = 0; b < RB_COUNT; ++b)
        if (b != RB_WEIGHTS && P.bufElems[b] > h->cap[b]) grow = true;
    if (grow) {
        h->plans.clear();              // baked pointers of cached plans die with the old buffers
        for (int b = 0; b < RB_COUNT; ++b) {
            if (b == RB_WEIGHTS || P.bufElems[b] <= h->cap[b]) continue;
            if (h->bufs[b]) { HIPCHK(hipFree(h->bufs[b])); h->bufs[b] = nullptr; h->cap[b] = 0; }
            const int64_t bytes = rbBytes(b, P.bufElems[b]);
            HIPCHK(hipMalloc(&h->bufs[b], (size_t)bytes));
            HIPCHK(hipMemset(h->bufs[b], 0, (size_t)bytes));   // zero halos, once
            h->cap[b] = P.bufElems[b];
        }
        HIPCHK(hipDeviceSynchronize());
    }
    const int64_t accNeed = (int64_t)t * 256 * 2;
    if (h->statAccCap < accNeed) {
        if (h->statAcc) HIPCHK(hipFree(h->statAcc));
        HIPCHK(hipMalloc((void**)&h->statAcc, (size_t)accNeed * sizeof(double)));
        h->statAccCap = accNeed;
    }
    // tables: one allocation
    pd->toff.resize(P.tables.size());
    int64_t tot = 0;
    for (size_t i = 0; i < P.tables.size(); ++i) { pd->toff[i] = tot; tot += (int64_t)((P.tables[i].size() + 3) / 4 * 4); }
    std::vector<int32_t> flat((size_t)tot, 0);
    for (size_t i = 0; i < P.tables.size(); ++i)
        memcpy(flat.data() + pd->toff[i], P.tables[i].data(), P.tables[i].size() * sizeof(int32_t));
    HIPCHK(hipMalloc((void**)&pd->dTables, (size_t)(tot > 0 ? tot : 4) * sizeof(int32_t)));
    HIPCHK(hipMemcpy(pd->dTables, flat.data(), (size_t)tot * sizeof(int32_t), hipMemcpyHostToDevice));
    auto T = [&](int id) -> const int32_t* { return id < 0 ? nullptr : pd->dTables + pd->toff[id]; };
    auto F = [&](int buf, int64_t off) -> float* { return buf < 0 ? nullptr : (float*)h->bufs[buf] + off; };
    // gather-GEMM descriptors: one allocation, 64-byte aligned records
    size_t descBytes = 0;
    for (const Op& op : P.ops) descBytes += (op.gemm.size() * sizeof(GGProblem) + 63) / 64 * 64;
    std::vector<char> hostDesc(descBytes + 64, 0);
    HIPCHK(hipMalloc(&pd->dDescs, descBytes + 64));
    size_t cursor = 0;
    for (const Op& op : P.ops) {
        RaftOpDev od;
        od.op = &op;
        if (op.kind == OP_GEMM) {
            GGProblem* hp = (GGProblem*)(hostDesc.data() + cursor);
            int tileStart = 0;
            for (size_t j = 0; j < op.gemm.size(); ++j) {
                const GemmItem& g = op.gemm[j];
                GGProblem& q = hp[j];
                q.A = F(g.bufA, g.offA); q.B = F(g.bufB, g.offB); q.C = F(g.bufC, g.offC);
                q.bias = g.offBias >= 0 ? F(RB_WEIGHTS, g.offBias) : nullptr;
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
            od.nQueues = 8;
            for (const GemmItem& g : op.gemm)
                if (g.tilesN > 4) od.nQueues = 1;
            cursor += (op.gemm.size() * sizeof(GGProblem) + 63) / 64 * 64;
        }
        pd->ops.push_back(od);
    }
    HIPCHK(hipMemcpy(pd->dDescs, hostDesc.data(), descBytes, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&pd->dQueues, (pd->ops.size() + 1) * 8 * sizeof(unsigned int)));
    *out = pd.get();
    h->plans[key] = std::move(pd);
    return 0;
}

static int raft_run(vsr_raft* h, RaftPlanDev* pd, int bgr, hipStream_t stream)
{
    const RaftPlan& P = *pd->plan;
    HIPCHK(hipMemsetAsync(pd->dQueues, 0, (pd->ops.size() + 1) * 8 * sizeof(unsigned int), stream));
    auto B = [&](int buf, int64_t off) -> float* { return (float*)h->bufs[buf] + off; };
    size_t idx = 0;
    for (const RaftOpDev& od : pd->ops) {
        const Op& op = *od.op;
        unsigned int* queue = pd->dQueues + 8 * idx++;
        int rc = 0;
        if (op.kind == OP_GEMM) {
            rc = vsr_launch_gather_gemm_dev((const GGProblem*)od.dDesc, od.nitems, od.total, op.tileCfg, op.bmode, queue, 3, od.nQueues,
                                            nullptr, stream);
        } else if (op.kind == OP_EW) {
            const int* ip = op.ipar;
            switch (op.ew) {
            case EW_IM2COL7_U8:
                rc = vsr_raft_launch_im2col7_u8((const uint8_t*)h->bufs[op.ibuf[0]], ip[0], ip[1], ip[2], bgr, B(op.ibuf[1], 0), stream);
                break;
            case EW_INORM_STATS:
                rc = vsr_raft_launch_inorm_stats(B(op.ibuf[0], 0), ip[0], ip[1], ip[2], ip[3], ip[4], h->statAcc, B(op.ibuf[1], 0), stream);
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
            default:
                return rfail(VSR_ERR_STATE, "unknown RAFT op");
            }
        } else {
            return rfail(VSR_ERR_STATE, "unexpected op kind in a RAFT plan");
        }
        if (rc != 0) return rfail(VSR_ERR_HIP, "RAFT kernel launch failed: " + op.tag + ": " + hipGetErrorString(hipGetLastError()));
    }
    (void)P;
    return 0;
}

extern "C" {

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
    if (device >= 0) {
        if (device >= vsr_device_count()) return rfail(VSR_ERR_NOGPU, "no such HIP device; there is no CPU fallback");
        HIPCHK(hipSetDevice(device));
        const size_t bytes = h->model.packed.size() * sizeof(float);
        HIPCHK(hipMalloc(&h->bufs[RB_WEIGHTS], bytes));
        HIPCHK(hipMemcpy(h->bufs[RB_WEIGHTS], h->model.packed.data(), bytes, hipMemcpyHostToDevice));
        h->cap[RB_WEIGHTS] = (int64_t)h->model.packed.size();
    }
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
        for (int b = 0; b < RB_COUNT; ++b)
            if (h->bufs[b]) (void)hipFree(h->bufs[b]);
        if (h->statAcc) (void)hipFree(h->statAcc);
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
    RaftPlanDev* pd = nullptr;
    RCCHK(raft_plan_dev(h, t, H, W, iters, &pd));
    if (h->geom != std::make_tuple(t, H, W)) {
        // kernels write interiors only and rely on zero halos: another frame size or count moves the halos, so the
        // workspace is cleared when the geometry changes (never in steady state)
        for (int b = 0; b < RB_COUNT; ++b)
            if (b != RB_WEIGHTS && h->bufs[b]) HIPCHK(hipMemsetAsync(h->bufs[b], 0, (size_t)rbBytes(b, h->cap[b]), stream));
        h->geom = std::make_tuple(t, H, W);
    }
    HIPCHK(hipMemcpyAsync(h->bufs[RB_IN_U8], frames_dev, (size_t)t * H * W * 3, hipMemcpyDeviceToDevice, stream));
    RCCHK(raft_run(h, pd, bgr ? 1 : 0, stream));
    const size_t half = (size_t)(t - 1) * 2 * H * W * sizeof(float);
    HIPCHK(hipMemcpyAsync(fwd_dev, h->bufs[RB_OUT], half, hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemcpyAsync(bwd_dev, (char*)h->bufs[RB_OUT] + half, half, hipMemcpyDeviceToDevice, stream));
    return 0;
}

int vsr_raft_read_buffer(vsr_raft_t* h, int buf, int64_t offset, int64_t count, float* out_host)
{
    if (!h || !out_host || buf <= RB_IN_U8 || buf >= RB_COUNT || offset < 0 || count < 0) return rfail(VSR_ERR_ARG, "bad argument");
    if (h->device < 0 || !h->bufs[buf] || offset + count > h->cap[buf]) return rfail(VSR_ERR_STATE, "buffer not allocated / range outside it");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out_host, (const float*)h->bufs[buf] + offset, (size_t)count * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
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

} // extern "C"
