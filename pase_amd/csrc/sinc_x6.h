// sinc_x6.h -- internal interface between the C-ABI entry points (conv_gemm.hip / wgrad_gemm.hip) and sinc_x6.hip (the
// one-input-channel SincNet layer on the bf16 matrix pipe).  Not part of the ABI.
#pragma once
#include "hip_compat.h"
#include "pase_amd.h"

struct PaseSincPlan {
    int n_kg;            // forward: k-groups of 16 taps; weight gradient: k-groups of 16 positions per stage (4)
    int nwin;            // windows of one LDS image
    int tiles_per_seq;   // forward: 256-column tiles per sequence; weight gradient: 64-position stages per sequence
    long pack_bytes;     // bytes of PaseConvGemm::wx6 / PaseWgrad::gx6 the launch needs
};

bool pase_sinc_x6_plan(const PaseConvGemm& p, PaseSincPlan& pl);
int pase_sinc_x6_pack(const PaseConvGemm& p, const PaseSincPlan& pl, hipStream_t st);
int pase_sinc_x6_launch(const PaseConvGemm& p, const PaseSincPlan& pl, hipStream_t st);
bool pase_sinc_x6_wgrad_plan(const PaseWgrad& w, PaseSincPlan& pl);
// ab != nullptr: the gradient operand is the apply pass of *ab evaluated on load (w.g is not read)
bool pase_sinc_x6_wgrad_act_bwd_ok(const PaseWgrad& w, const PaseActBwd& ab);
int pase_sinc_x6_wgrad_launch(const PaseWgrad& w, const PaseSincPlan& pl, hipStream_t st, const PaseActBwd* ab = nullptr);
