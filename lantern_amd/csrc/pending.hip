// pending.hip -- entry points declared in include/lantern_hip.h whose kernels are not built yet.
// They fail loudly (LANTERN_E_UNSUPPORTED); nothing falls back to a CPU path.
#include "common.h"
using namespace lantern;

extern "C" int lantern_evaluate_posterior_greedy(const float *, const int32_t *, const int64_t *, int, int, int, int, int, int, int,
                                                 int, double, int, const uint16_t *, int, int, int32_t *, int32_t *, float *, void *) {
    set_error("evaluate_posterior_greedy: kernel not built yet");
    return LANTERN_E_UNSUPPORTED;
}
extern "C" int lantern_drafter_fc(const int64_t *, const void *, const void *, const void *, const void *, int, int, int, float, void *,
                                  void *) {
    set_error("drafter_fc: kernel not built yet");
    return LANTERN_E_UNSUPPORTED;
}
extern "C" int lantern_build_vq_table(const float *, int, int, uint16_t *, void *, void *) {
    set_error("build_vq_table: kernel not built yet");
    return LANTERN_E_UNSUPPORTED;
}
