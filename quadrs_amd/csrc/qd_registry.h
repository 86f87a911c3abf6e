// qd_registry.h — table entries of the shape-specialised (FixedGeo) chain kernels.
// (the table itself is kFixed in quadrs_hip.hip)
#pragma once
#include "qd_chain.h"

namespace qd {

typedef void (*chain_fn)(const ChainParams);

struct FixedEntry {
    int fmt, nco;
    uint32_t W, S, D, T, G;
    int lb;              // register budget the build targets: waves per SIMD (4 -> 128 VGPRs, 2 -> 256)
    int nt;              // workgroup size
    int pad;             // LDS pad elements per row (FixedGeo::kPad request: 1 or 2)
    int batch;           // tiles per FFT batch (FixedGeo::kBatch)
    int flags;           // FixedGeo FLAGS_ (kGeoPlanar | kGeoBakedTaps | kGeoNoSplit | ...)
    int rch, whole;      // prefetch shape (k_chain RCH / WHOLE) and FIR knobs, so a plan-time re-specialisation builds the same kernel
    int firb, firr;
    chain_fn fn;
    const char *name;
};

#define QD_FIXED(F, NCO, W, S, D, T, G, RCH, WHOLE, LB, NAME) \
    { F, NCO, W, S, D, T, G, LB, qd::kThreads, 1, 1, 0, RCH, WHOLE, 8, 1, qd::k_chain<F, NCO, qd::FixedGeo<W, S, D, T, G>, true, RCH, WHOLE, true, LB>, NAME }
#define QD_FIXED_FB(F, NCO, W, S, D, T, G, RCH, WHOLE, LB, PAD, BATCH, FLAGS, NAME) \
    { F, NCO, W, S, D, T, G, LB, qd::kThreads, PAD, BATCH, FLAGS, RCH, WHOLE, 8, 1, qd::k_chain<F, NCO, qd::FixedGeo<W, S, D, T, G, 8, 1, PAD, BATCH, FLAGS>, true, RCH, WHOLE, true, LB>, NAME }

#define QD_FIXED_NTF(F, NCO, W, S, D, T, G, RCH, WHOLE, LB, NT, FIRB, FIRR, PAD, FLAGS, NAME) \
    { F, NCO, W, S, D, T, G, LB, NT, PAD, 1, FLAGS, RCH, WHOLE, FIRB, FIRR, qd::k_chain<F, NCO, qd::FixedGeo<W, S, D, T, G, FIRB, FIRR, PAD, 1, FLAGS>, true, RCH, WHOLE, true, LB, NT>, NAME }

}  // namespace qd
