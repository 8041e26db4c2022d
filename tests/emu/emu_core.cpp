// TEST INFRASTRUCTURE: lock-step wave64 emulator.  Each lane is a fiber; a
// cross-lane primitive parks the lane until every live lane of the wave has
// reached the same primitive (anything else is reported as a divergence bug).
#include "kx_wave.h"
#include "emu_core.h"
#include <stdio.h>
#include <stdlib.h>

extern "C" void kx_switch(void** save_sp, void* new_sp);
asm(R"(
.text
.globl kx_switch
.type kx_switch,@function
kx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
)");

namespace kxemu {
int cur_lane; u32 cur_block, num_blocks;
static const size_t STACK = 256 * 1024;
static void* sp[64]; static void* sched_sp; static bool alive[64];
static int op[64]; static u64 argA[64], argB[64], res[64];
static u8* stacks = nullptr;
static const std::function<void()>* cur_fn;
int failed = 0;

u64 arrive(int o, u64 a, u64 b)
{
    int const l = cur_lane;
    op[l] = o; argA[l] = a; argB[l] = b;
    kx_switch(&sp[l], sched_sp);
    return res[l];
}
static void trampoline()
{
    (*cur_fn)();
    alive[cur_lane] = false; op[cur_lane] = 0;
    kx_switch(&sp[cur_lane], sched_sp);
    abort();
}
static void run_wave()
{
    if (!stacks) stacks = (u8*)aligned_alloc(64, STACK * 64);
    for (int l = 0; l < 64; l++) {
        u64* top = (u64*)(stacks + STACK * (l + 1));
        top -= 2; top[0] = (u64)(void*)&trampoline; top[1] = 0;   // ret addr, then fake caller slot
        top -= 6; for (int i = 0; i < 6; i++) top[i] = 0;
        sp[l] = top; alive[l] = true; op[l] = 0;
    }
    for (;;) {
        int nalive = 0;
        for (int l = 0; l < 64; l++) if (alive[l]) { cur_lane = l; kx_switch(&sched_sp, sp[l]); }
        int o = 0; u64 mask = 0;
        for (int l = 0; l < 64; l++) if (alive[l]) {
            nalive++;
            if (!o) o = op[l];
            else if (op[l] != o) { if (!failed) fprintf(stderr, "kxemu: lanes diverged at a cross-lane primitive (lane %d op %d vs %d)\n", l, op[l], o); failed = 1; }
            if (op[l] == OP_BALLOT && argA[l]) mask |= 1ull << l;
        }
        if (!nalive) break;
        if (failed) { break; }
        for (int l = 0; l < 64; l++) if (alive[l]) {
            if (o == OP_BALLOT) res[l] = mask;
            else if (o == OP_SHFL) { int s = (int)argB[l]; res[l] = alive[s] ? argA[s] : 0; }
            else res[l] = 0;
        }
    }
}
void launch(u32 nblocks, const std::function<void()>& fn)
{
    cur_fn = &fn; num_blocks = nblocks;
    for (u32 b = 0; b < nblocks && !failed; b++) { cur_block = b; run_wave(); }
}
}
