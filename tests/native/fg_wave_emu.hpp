// fg_wave_emu.hpp -- CPU stand-in for one 64-lane wavefront (test infrastructure, host only).
//
// The wave-cooperative tokenisers (flowgger_amd/csrc/fg_*2.hpp) are written against fg_wave.hpp.  Here the 64 lanes of a
// wave are 64 fibers (ucontext) of ONE host thread: a lane runs until it reaches a cross-lane primitive (ballot, shuffle,
// barrier), parks there, and the scheduler resumes the next lane; when all 64 have arrived the primitive's result is
// computed and every lane continues.  Deterministic, debuggable with gdb, and strict: a lane that returns -- or reaches a
// DIFFERENT primitive -- while others wait is reported as divergence, which on the GPU would be a hang or garbage.
#pragma once
#include <stdint.h>
#include <ucontext.h>

#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

namespace fg {
namespace emu {

enum { OP_SYNC = 0, OP_BALLOT = 1, OP_SHFL = 2, OP_SHFL_UP = 3 };

struct Wave {
    static constexpr int kLanes = 64;
    ucontext_t main_ctx;
    ucontext_t ctx[kLanes];
    std::vector<char> stacks;
    int cur = -1;
    uint64_t val[kLanes];
    int op[kLanes];
    uint32_t arg[kLanes];
    uint64_t res[kLanes];
    bool waiting[kLanes];
    bool done[kLanes];
    std::function<void()> body;
    unsigned long collectives = 0;
};

inline Wave*& current() {
    static thread_local Wave* w = nullptr;
    return w;
}
inline uint32_t lane() { return (uint32_t)current()->cur; }

inline uint64_t collective(uint64_t v, int op, uint32_t arg) {
    Wave* w = current();
    const int l = w->cur;
    w->val[l] = v;
    w->op[l] = op;
    w->arg[l] = arg;
    w->waiting[l] = true;
    swapcontext(&w->ctx[l], &w->main_ctx);
    return w->res[l];
}

inline void trampoline() {
    Wave* w = current();
    w->body();
    w->done[w->cur] = true;
    swapcontext(&w->ctx[w->cur], &w->main_ctx);
}

// Run `body` once per lane, in lockstep at the cross-lane primitives.  Throws std::runtime_error on divergence.
inline void run_wave(const std::function<void()>& body, size_t stack_bytes = 512u << 10) {
    static thread_local Wave* w = nullptr;
    if (!w) {
        w = new Wave();
        w->stacks.resize((size_t)Wave::kLanes * stack_bytes);
    }
    Wave* prev = current();
    current() = w;
    w->body = body;
    for (int l = 0; l < Wave::kLanes; ++l) {
        w->waiting[l] = w->done[l] = false;
        getcontext(&w->ctx[l]);
        w->ctx[l].uc_stack.ss_sp = w->stacks.data() + (size_t)l * stack_bytes;
        w->ctx[l].uc_stack.ss_size = stack_bytes;
        w->ctx[l].uc_link = nullptr;
        makecontext(&w->ctx[l], (void (*)())trampoline, 0);
    }
    std::string err;
    for (;;) {
        for (int l = 0; l < Wave::kLanes; ++l) {
            if (w->done[l] || w->waiting[l]) continue;
            w->cur = l;
            swapcontext(&w->main_ctx, &w->ctx[l]);
        }
        int n_done = 0, n_wait = 0, first = -1;
        for (int l = 0; l < Wave::kLanes; ++l) {
            n_done += w->done[l];
            if (w->waiting[l]) {
                ++n_wait;
                if (first < 0) first = l;
            }
        }
        if (n_done == Wave::kLanes) break;
        if (n_done != 0) {
            err = "wave divergence: " + std::to_string(n_done) + " lanes returned while " + std::to_string(n_wait) +
                  " wait at a cross-lane primitive (op " + std::to_string(w->op[first]) + ")";
            break;
        }
        const int op = w->op[first];
        for (int l = 0; l < Wave::kLanes; ++l)
            if (w->op[l] != op) {
                err = "wave divergence: lanes wait at different cross-lane primitives (" + std::to_string(op) + " vs " +
                      std::to_string(w->op[l]) + " on lane " + std::to_string(l) + ")";
                break;
            }
        if (!err.empty()) break;
        ++w->collectives;
        uint64_t b = 0;
        switch (op) {
            case OP_SYNC:
                for (int l = 0; l < Wave::kLanes; ++l) w->res[l] = 0;
                break;
            case OP_BALLOT:
                for (int l = 0; l < Wave::kLanes; ++l) b |= (uint64_t)(w->val[l] & 1u) << l;
                for (int l = 0; l < Wave::kLanes; ++l) w->res[l] = b;
                break;
            case OP_SHFL:
                for (int l = 0; l < Wave::kLanes; ++l) w->res[l] = w->val[w->arg[l] & 63u];
                break;
            case OP_SHFL_UP:
                for (int l = 0; l < Wave::kLanes; ++l) w->res[l] = (uint32_t)l >= w->arg[l] ? w->val[l - (int)w->arg[l]] : w->val[l];
                break;
            default:
                err = "unknown cross-lane primitive";
        }
        if (!err.empty()) break;
        for (int l = 0; l < Wave::kLanes; ++l) w->waiting[l] = false;
    }
    current() = prev;
    if (!err.empty()) throw std::runtime_error(err);
}

}  // namespace emu
}  // namespace fg
