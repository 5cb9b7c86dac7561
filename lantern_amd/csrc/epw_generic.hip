// epw_generic.hip -- the argument-driven instances of the windowed chain kernel (epw_body.h, SPEC 0: mode, flags and sizes from the argument
// block; one workgroup per CU), by window width and neighbour-id mode; and the same with TopPLogitsWarper compiled in for rows that arrive as
// logits (LANTERN_ROWS_LOGITS with prm.top_p in (0, 1): drafters/utils.py:36-52 -- the reference applies the processor list per visited row
// inside evaluate_posterior, ea_model_llamagen.py:709-787).
#undef EPW_TRACE
#include "epw_body.h"

namespace lantern {

template <int NT, int E4, int TPO>
static void launch_ids(int idmode, const EpwLaunch &l, const EpwArgs &args) {
    if (idmode == 2) LANTERN_LAUNCH((epw_kernel<NT, E4, 2, 1, false, false, 0, TPO>), l.grid, dim3(NT), l.lds, l.st, args);
    else if (idmode == 1) LANTERN_LAUNCH((epw_kernel<NT, E4, 1, 1, false, false, 0, TPO>), l.grid, dim3(NT), l.lds, l.st, args);
    else LANTERN_LAUNCH((epw_kernel<NT, E4, 0, 1, false, false, 0, TPO>), l.grid, dim3(NT), l.lds, l.st, args);
}

template <int TPO>
static bool launch_width(int W, int idmode, const EpwLaunch &l, const EpwArgs &args) {
    if (W <= 1024) launch_ids<256, 1, TPO>(idmode, l, args);
    else if (W <= 2048) launch_ids<256, 2, TPO>(idmode, l, args);
    else if (W <= 4096) launch_ids<512, 2, TPO>(idmode, l, args);
    else if (W <= 8192) launch_ids<512, 4, TPO>(idmode, l, args);
    else if (W <= 16384) launch_ids<1024, 4, TPO>(idmode, l, args);
    else return false;
    return true;
}

bool epw_launch_generic(int W, int idmode, bool nucleus, const EpwLaunch &l, const EpwArgs &args) {
    return nucleus ? launch_width<2>(W, idmode, l, args) : launch_width<0>(W, idmode, l, args);
}

}  // namespace lantern
