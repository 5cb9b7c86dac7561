// epw_throughput.hip -- the forms of the windowed chain kernel (epw_body.h) for more sequences per launch than CUs: several workgroups share a
// CU, the register allocation is capped accordingly (WPE), the fixed configurations (SPEC) fold their flags away, and a candidate's drafter row is
// requested only once its rejection is known (LATE_Q, TPO bit 0).  BASELINE assesses evaluate_posterior's roofline target at this batch.
//   256 threads x 8 float4, three workgroups per CU (53 KB of LDS each): probability rows and -- round 5 -- raw cond / uncond bf16 rows
//   512 threads x 4 float4 at 128 VGPRs, two per CU: everything else on the 8192-id window
//   512 threads x 8 float4 at 128 VGPRs, two per CU (79 KB of LDS each): LlamaGen's 16384-id window, standard verify on dynamic trees (round 6)
#undef EPW_TRACE
#include "epw_body.h"

namespace lantern {

bool epw_launch_throughput(int kind, const EpwLaunch &l, const EpwArgs &args) {
    // lantern_tuning_set("epw_tp4", ..): 1 (default, round 6) = compact, four per CU, the serial wave rotates with the sequence and runs at raised priority, a
    // rejection's residual is normalised by a second pass over LDS (no spills): 135.9 - 137.8 us per 4096 sequences against round 5's 147.8 - 157.5 on the same
    // boxes; 2 = round 5's compact form; 3 = the neighbour scan on all waves instead of the rotation (139.5 - 140.1); 4 = 1 without the raised priority
    // (137.5 - 137.8); 0 = round 4's three-per-CU form (162 - 163)
    const int tp4 = tuning(TUNE_EPW_TP4);
    const bool compact = tp4 != 0;
    const bool raw512 = tuning(TUNE_EPW_TP_RAW) == 512;
#define TP(...) LANTERN_LAUNCH((epw_kernel<__VA_ARGS__>), l.grid, dim3(NTX), l.lds, l.st, args)
    switch (kind) {
    case EPW_TP_LUMINA_DEFAULT_TREE:
        if (compact) {          // the default tree on the smallest staged tables: 40 KB of LDS, 128 VGPRs -- FOUR workgroups per CU
            constexpr int NTX = 256;
            const size_t lds4 = epw_shared_offset(8192, false) + sizeof(EwSharedCompact) + (size_t)6 * epw_pd_cap(15, 6) * 4;
            if (tp4 == 2) LANTERN_LAUNCH((epw_kernel<256, 8, 2, 4, true, false, 2, 5>), l.grid, dim3(NTX), lds4, l.st, args);
            else if (tp4 == 3) LANTERN_LAUNCH((epw_kernel<256, 8, 2, 4, true, false, 2, 5 + 16 + 64>), l.grid, dim3(NTX), lds4, l.st, args);
            else if (tp4 == 4) LANTERN_LAUNCH((epw_kernel<256, 8, 2, 4, true, false, 2, 5 + 8 + 64>), l.grid, dim3(NTX), lds4, l.st, args);
            else LANTERN_LAUNCH((epw_kernel<256, 8, 2, 4, true, false, 2, 5 + 8 + 64 + 128>), l.grid, dim3(NTX), lds4, l.st, args);
        } else { constexpr int NTX = 256; TP(256, 8, 2, 3, true, false, 2, 1 + 8); }
        return true;
    case EPW_TP_LUMINA_STATIC:
        { constexpr int NTX = 256; TP(256, 8, 2, 3, true, false, 1, 1 + 8); return true; }
    case EPW_TP_LUMINA_DYNAMIC: { constexpr int NTX = 256; TP(256, 8, 2, 3, true, false, 3, 1 + 8); return true; }
    case EPW_TP_ANOLE_STATIC: { constexpr int NTX = 256; TP(256, 8, 2, 3, true, false, 4, 1 + 8); return true; }
    case EPW_TP_512_DEFAULT_TREE: { constexpr int NTX = 512; TP(512, 4, 2, 4, true, false, 2); return true; }
    case EPW_TP_512_PACKED: { constexpr int NTX = 512; TP(512, 4, 2, 4, true); return true; }
    case EPW_TP_512_ID0: { constexpr int NTX = 512; TP(512, 4, 0, 4); return true; }
    case EPW_TP_512_ID1: { constexpr int NTX = 512; TP(512, 4, 1, 4); return true; }
    case EPW_TP_512_ID2: { constexpr int NTX = 512; TP(512, 4, 2, 4); return true; }
    case EPW_TP_RAW_GENERIC: { constexpr int NTX = 512; TP(512, 4, 2, 4, true, true); return true; }
    // raw rows carry the row post-process's 18 KB of histograms: 71 KB of LDS per workgroup = two per CU whatever the thread count.  256 threads x 8
    // float4 at two waves per SIMD (no register cap to spill against, half the waves -- half the repeated scalar work -- per sequence), or
    // (lantern_tuning_set("epw_tp_raw", 512)) 512 threads at 128 VGPRs
    case EPW_TP_RAW_LUMINA_DEFAULT_TREE: if (raw512) { constexpr int NTX = 512; TP(512, 4, 2, 4, true, true, 2, 1); } else { constexpr int NTX = 256; TP(256, 8, 2, 2, true, true, 2, 1 + 8); } return true;
    case EPW_TP_RAW_LUMINA_STATIC: if (raw512) { constexpr int NTX = 512; TP(512, 4, 2, 4, true, true, 1, 1); } else { constexpr int NTX = 256; TP(256, 8, 2, 2, true, true, 1, 1 + 8); } return true;
    case EPW_TP_RAW_LUMINA_DYNAMIC: if (raw512) { constexpr int NTX = 512; TP(512, 4, 2, 4, true, true, 3, 1); } else { constexpr int NTX = 256; TP(256, 8, 2, 2, true, true, 3, 1 + 8); } return true;
    case EPW_TP_RAW_ANOLE_STATIC: if (raw512) { constexpr int NTX = 512; TP(512, 4, 2, 4, true, true, 4, 1); } else { constexpr int NTX = 256; TP(256, 8, 2, 2, true, true, 4, 1 + 8); } return true;
    // LlamaGen's standard verify (BASELINE config 2) on probability rows: the 64 KB window + EwSharedLite + the per-path tables = <= 80 KB, TWO workgroups of
    // 512 threads x 8 float4 per CU at 128 VGPRs (the generic instance: 1024 threads, 97 KB, one per CU)
    case EPW_TP_LLAMAGEN_DYNAMIC: {
        constexpr int NTX = 512;
        const int lg = tuning(TUNE_EPW_TP_LG);
        if (lg == 2) TP(512, 8, 1, 4, true, false, 5, 1 + 8 + 64 + 256);                 // + the residual normalised by a second LDS pass
        else if (lg == 3) TP(512, 8, 1, 4, true, false, 5, 1 + 8 + 256);                 // rows through registers
        else if (lg == 4) TP(512, 8, 1, 4, true, false, 5, 1 + 8 + 128 + 256 + 512);     // + raised priority of the serial section
        else TP(512, 8, 1, 4, true, false, 5, 1 + 8 + 256 + 512);                        // rows by LDS-DMA
        return true;
    }
    default: return false;
    }
#undef TP
}

}  // namespace lantern
