// Internal view of the HBM-bound kernels (elementwise.hip); SMProblem lives in the public header.
#pragma once
#include <stdint.h>
#include "../../include/vsr_hip.h"

// d_probs: DEVICE array (rowStart filled, multiples of 4); totalRows = padded row count
extern "C" int vsr_launch_softmax_dev(const SMProblem* d_probs, int nprobs, int totalRows, void* stream);
