// Native replay of a recorded step (round 4).  CSModel.update() records one training step as a flat list of C-ABI calls and
// stream / event operations (model.py: RecordedStep); replaying that list from Python costs the host ~10 us per entry (ctypes
// marshalling of up to 25 arguments), ~40 ms per 47 ms step -- about as long as the GPU needs, so every stretch of short kernels
// ran host-bound.  Here the list is a "tape" of 64-bit words that one call walks in C:
//     word 0 of an entry: code | flags << 16 | nargs << 24      then nargs argument words
//     code < 0x8000: index of an int-returning prototype of include/san_hip.h (header order; san_replay_table.inc, generated)
//     code 0x8000:   hipEventRecord(event a[0], stream a[1])         code 0x8001: hipStreamWaitEvent(stream a[0], event a[1])
// Nothing here decides anything: the calls, their order and their arguments are exactly the recorded ones.
#include "san_common.h"

#include <cstring>

namespace {

inline float bits_f(uint64_t w) {
    const uint32_t u = (uint32_t)w;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline double bits_d(uint64_t w) {
    double d;
    memcpy(&d, &w, 8);
    return d;
}

constexpr uint64_t kEventRecord = 0x8000, kStreamWait = 0x8001;
constexpr uint64_t kFlagIgnoreRc = 1, kFlagPack = 2;

}  // namespace

extern "C" int san_replay_run(const void* tape_, size_t n_words, int skip_packs, long long* failed_word) {
    const uint64_t* tape = static_cast<const uint64_t*>(tape_);
    SAN_CHECK_ARG(tape != nullptr || n_words == 0, "null tape");
    size_t i = 0;
    while (i < n_words) {
        const uint64_t head = tape[i];
        const uint64_t code = head & 0xffffu, flags = (head >> 16) & 0xffu, nargs = (head >> 24) & 0xffu;
        const uint64_t* a = tape + i + 1;
        if (i + 1 + nargs > n_words) {
            if (failed_word) *failed_word = (long long)i;
            san_set_error("san_replay_run: entry at word %lld runs past the end of the tape", (long long)i);
            return SAN_E_ARG;
        }
        int rc = 0;
        if (!((flags & kFlagPack) && skip_packs)) {
            switch (code) {
#include "san_replay_table.inc"
                case kEventRecord: rc = (int)hipEventRecord((hipEvent_t)(uintptr_t)a[0], (hipStream_t)(uintptr_t)a[1]); break;
                case kStreamWait: rc = (int)hipStreamWaitEvent((hipStream_t)(uintptr_t)a[0], (hipEvent_t)(uintptr_t)a[1], 0); break;
                default:
                    if (failed_word) *failed_word = (long long)i;
                    san_set_error("san_replay_run: unknown code %llu at word %lld", (unsigned long long)code, (long long)i);
                    return SAN_E_ARG;
            }
        }
        if (rc != 0 && !(flags & kFlagIgnoreRc)) {
            if (failed_word) *failed_word = (long long)i;
            if (code >= 0x8000u) san_set_error("san_replay_run: %s failed: %s", code == kEventRecord ? "hipEventRecord" : "hipStreamWaitEvent", hipGetErrorString((hipError_t)rc));
            return rc;
        }
        i += 1 + nargs;
    }
    return SAN_OK;
}
