// Shared between the conv and transposed-conv kernels (internal).
#pragma once
#include "san_common.h"

// Output channels each lane keeps in registers: the candidate that wastes the
// fewest padded channels (18 -> 18/36/72/144/288 exactly; 16 -> 32/64/128; ...).
static inline int san_pick_co_t(int cout) {
    const int cand[5] = {18, 16, 8, 4, 2};
    int best = 2, best_waste = 1 << 30;
    for (int i = 0; i < 5; ++i) {
        int ct = cand[i];
        int waste = san_cdiv(cout, ct) * ct - cout;
        if (waste < best_waste) {
            best_waste = waste;
            best = ct;
        }
    }
    return best;
}
