// Every environment knob of libcapdec_hip.so in ONE place.  capdec_create parses the environment once into the context's
// Tuning (capi_context.hip: tuning_from_env); the launchers of the other translation units receive a pointer to it through
// GemmEpilogue::tune / KvCache::tune / their arguments -- no translation unit reads the environment on its own, nothing
// is latched in function-local statics, and two contexts created under different environments behave differently, as
// the tests that monkeypatch the environment expect.
//
// PRODUCT knobs select among correct code paths (A/B switches kept for regression hunting; every default is the
// measured-fastest setting, none is needed for normal use).  MEASUREMENT knobs exist only in builds with -DCAPDEC_MEASURE
// (capdec_amd/lib/libcapdec_hip_measure.so, used by tools/ and by bench.py's untimed tail): ablations that produce wrong
// results on purpose, ring-depth / occupancy overrides, per-block phase stamps, the diverged-beam hook.  The shipped
// library contains none of that code.
#pragma once
#include <string>

namespace capdec {

struct Tuning {
    // ---- product
    int gemm_mode = 3;            // CAPDEC_GEMM_MODE: f32 0 | bf16x3 1 | bf16 2 | f16x2 3 (default) | f16 4
    bool batch_invariant = false; // CAPDEC_BATCH_INVARIANT
    bool compact = true;          // CAPDEC_COMPACT=0: finished captions stay in the batch
    bool x3_pack_a = true;        // CAPDEC_X3_PACKA=0: bf16x3 mode, LayerNorm writes fp32 (the GEMM splits A itself)
    bool pack_chain = true;       // CAPDEC_X3_CHAIN=0: attention / fc epilogue write fp32 instead of packed operands
    bool splitk = true;           // CAPDEC_SPLITK=0: no split-K at all
    bool splitk_mid = true;       // CAPDEC_SPLITK_MID=0: no split-K for mid-size batches
    bool x1_splitk = true;        // CAPDEC_X1_SPLITK=0: none for the one-plane (bf16 / f16) kernels
    bool fuse_ln = true;          // CAPDEC_FUSE_LN=0: separate split-K reduce and LayerNorm
    int h2_persist = 512;         // CAPDEC_H2_PERSIST: blocks of the persistent form (0 = one block per tile)
    int h2w = 1;                  // CAPDEC_H2W: 0 = round-2 kernels only, 1 = planners, 2 / 8 = force a round-3 wide tile,
                                  //             10 / 14 = force a round-4 ping-pong tile (tests)
    int pp = 2;                   // CAPDEC_PP: ping-pong planner: 0 never, 2 mid-size launches (default); 1 / 3 (also / only large
                                  //             launches: the 256 x 256 tile) act in measurement builds only
    bool lmhead_wide = true;      // CAPDEC_LMHEAD_WIDE=0: 128-row lm_head tiles at every size
    bool lmhead_k3 = true;        // CAPDEC_LMHEAD_K3=0: the wide lm_head keeps k candidates per tile (no exact second pass)
    int lmhead_k3_max = 60;       // CAPDEC_LMHEAD_K3_MAX: per mille of the rows taking the second pass above which a decode
                                  //   call goes back to k per tile (checked at the poll points; break-even is ~80)
    bool kv_direct = true;        // CAPDEC_KV_DIRECT=0: the attention kernel appends K / V itself
    bool clip_trunc = true;       // CAPDEC_CLIP_TRUNC=0: the CLIP text tower computes all 77 positions of every caption (default: only
                                  //   the positions up to a chunk's last EOT; captions sorted by length)
    bool rn_packed = true;        // CAPDEC_RN_PACKED=0: fp32 im2col in the ResNet tower
    bool rn_implicit = true;      // CAPDEC_RN_IMPLICIT=0: fp32 activations in the ResNet tower
    bool train_f16x2 = true;      // CAPDEC_TRAIN_F16X2=0: the train step's backward GEMMs on the native fp32 MFMA GEMM instead of the
                                  //   two-fp16-plane kernels (A/B: 39.4 vs 23.6 ms per step; both parity-tested)
    bool train_attn_blk = true;   // CAPDEC_TRAIN_ATTN_BLK=0: the train step's attention on the per-query wavefront kernels at every sequence
                                  //   length (the fallback above 128 positions) instead of the block-per-(sample, head) kernels
    bool hook_packa = false;      // CAPDEC_HOOK_PACKA: capdec_gemm_f32 (test hook) packs A first (the LayerNorm -> GEMM path)
    bool hook_cache = false;      // CAPDEC_HOOK_CACHE: ... and treats both operands as resident (micro-benchmarks)
    std::string rccl_lib;         // CAPDEC_RCCL_LIB: path of librccl for the C-ABI communicator
    // ---- measurement (read only with -DCAPDEC_MEASURE; constants otherwise)
    int h2_ns = 4;                // CAPDEC_H2_NS: ring depth 3 | 4 | 5 of the 128 x 128 kernel
    int h2_abl = 0;               // CAPDEC_H2_ABL 1..6: ablations, WRONG results
    int x1_ns = 3;                // CAPDEC_X1_NS=4: two blocks per CU for the one-plane kernels
    int x3_abl_dma = 0;           // CAPDEC_ABL_DMA (bf16x3 kernels)
    int x3_tile_m = 0;            // CAPDEC_X3_TILE_M=64
    int f32_bk = 0;               // CAPDEC_GEMM_BK
    int f32_lmhead_bk = 16;       // CAPDEC_LMHEAD_BK
    int att_preload = 1;          // CAPDEC_ATT_PRELOAD=0
    int att_wsync = 1;            // CAPDEC_ATT_WSYNC=0
    int att_dma = 1;              // CAPDEC_ATT_DMA=0
    int att_occ = 4;              // CAPDEC_ATT_OCC=3
    int att_na = 0;               // CAPDEC_ATT_NA=2 | 4
    int pp_abl = 0;               // CAPDEC_PP_ABL 1..8 (gemm_pp.hip)
    std::string pp_stamps;        // CAPDEC_PP_STAMPS=<file>: per-block phase stamps of the ping-pong GEMM
    int lmhead_k1 = 0;            // CAPDEC_LMHEAD_K1=1: the k = 1 lm_head epilogue whatever k is (WRONG results)
};

// parses the environment; on a malformed value returns false and sets *err (capdec_create then fails: a typo must not
// silently select another precision)
bool tuning_from_env(Tuning *t, std::string *err);
// what launchers use when their caller passes no Tuning (default-constructed: every default above)
const Tuning &default_tuning();

}  // namespace capdec
