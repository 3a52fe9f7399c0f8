// The C handle behind vsr_plan_t: shared by the STTN engine (sttn_engine.hip) and the RAFT engine (flow_engine.hip).
#pragma once
#include <memory>
#include "sttn_plan.h"

struct vsr_plan {
    std::unique_ptr<vsr::PlanIR> plan;
};

// records the message returned by vsr_last_error() and returns `code`
int vsr_internal_fail(int code, const char* msg);
