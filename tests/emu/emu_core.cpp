// TEST INFRASTRUCTURE: lock-step wave64 emulator.  Each lane is a fiber; a
// cross-lane primitive parks the lane until every live lane of its wave has
// reached the same primitive (anything else is reported as a divergence bug);
// a workgroup barrier parks it until every live lane of the workgroup arrived.
#include "kx_wave.h"
#include "emu_core.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

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
int cur_lane; u32 cur_block, num_blocks; int cur_wave, waves_per_block = 1;
static const size_t STACK = 128 * 1024;
enum { F_DEAD = 0, F_READY = 1, F_WAIT = 2 };
struct Fiber { void* sp; int state; int op; u64 a, b, res; };
static std::vector<Fiber> fib;
static void* sched_sp; static int cur_fiber;
static u8* stacks = nullptr; static size_t stacks_n = 0;
static const std::function<void()>* cur_fn;
int failed = 0;
u64 stat[64];

u64 arrive(int o, u64 a, u64 b)
{
    Fiber& f = fib[cur_fiber];
    f.op = o; f.a = a; f.b = b; f.state = F_WAIT;
    kx_switch(&f.sp, sched_sp);
    return fib[cur_fiber].res;
}
static void trampoline()
{
    (*cur_fn)();
    fib[cur_fiber].state = F_DEAD;
    kx_switch(&fib[cur_fiber].sp, sched_sp);
    abort();
}
static void run_block(int W)
{
    int const N = W * 64;
    if (stacks_n < (size_t)N) { free(stacks); stacks = (u8*)aligned_alloc(64, STACK * N); stacks_n = N; }
    fib.assign(N, Fiber());
    for (int i = 0; i < N; i++) {
        u64* top = (u64*)(stacks + STACK * (i + 1));
        top -= 2; top[0] = (u64)(void*)&trampoline; top[1] = 0;
        top -= 6; for (int k = 0; k < 6; k++) top[k] = 0;
        fib[i].sp = top; fib[i].state = F_READY;
    }
    for (;;) {
        bool ran = false;
        for (int i = 0; i < N; i++) if (fib[i].state == F_READY) {
            cur_fiber = i; cur_wave = i / 64; cur_lane = i % 64;
            kx_switch(&sched_sp, fib[i].sp);
            ran = true;
        }
        int alive = 0, at_block_sync = 0; bool resolved = false;
        for (int i = 0; i < N; i++) if (fib[i].state != F_DEAD) { alive++; if (fib[i].state == F_WAIT && fib[i].op == OP_BLOCK_SYNC) at_block_sync++; }
        if (!alive) break;
        // quad-scoped exchanges (DPP on the GPU: no rendezvous beyond the four lanes) resolve as soon as the quad's live lanes are there
        for (int q = 0; q < N / 4 && !resolved; q++) {
            int n = 0, waiting = 0;
            for (int l = 0; l < 4; l++) { Fiber& f = fib[q * 4 + l]; if (f.state == F_DEAD) continue; n++; if (f.state == F_WAIT && f.op == OP_QUAD) waiting++; }
            if (!n || waiting != n) continue;
            for (int l = 0; l < 4; l++) { Fiber& f = fib[q * 4 + l]; if (f.state != F_WAIT) continue; Fiber& s2 = fib[q * 4 + (int)(f.b & 3)]; f.res = (s2.state == F_WAIT) ? s2.a : 0; }
            for (int l = 0; l < 4; l++) if (fib[q * 4 + l].state == F_WAIT) fib[q * 4 + l].state = F_READY;
            resolved = true;
        }
        if (resolved) continue;
        if (at_block_sync == alive) { for (int i = 0; i < N; i++) if (fib[i].state == F_WAIT) { fib[i].res = 0; fib[i].state = F_READY; } resolved = true; }
        else for (int w = 0; w < W; w++) {
            int o = 0, n = 0, waiting = 0; u64 mask = 0; bool mixed = false;
            for (int l = 0; l < 64; l++) { Fiber& f = fib[w * 64 + l]; if (f.state == F_DEAD) continue; n++;
                if (f.state == F_WAIT && f.op != OP_BLOCK_SYNC) { waiting++; if (!o) o = f.op; else if (f.op != o) mixed = true; if (f.op == OP_BALLOT && f.a) mask |= 1ull << l; } }
            if (!n || waiting != n) continue;
            if (mixed) { if (!failed) fprintf(stderr, "kxemu: lanes of wave %d diverged at a cross-lane primitive\n", w); failed = 1; break; }
            for (int l = 0; l < 64; l++) { Fiber& f = fib[w * 64 + l]; if (f.state != F_WAIT) continue;
                if (o == OP_BALLOT) f.res = mask;
                else if (o == OP_SHFL) { Fiber& s = fib[w * 64 + (int)f.b]; f.res = (s.state == F_WAIT) ? s.a : 0; }
                else f.res = 0; }
            for (int l = 0; l < 64; l++) if (fib[w * 64 + l].state == F_WAIT && fib[w * 64 + l].op != OP_BLOCK_SYNC) fib[w * 64 + l].state = F_READY;
            resolved = true;
        }
        if (failed) break;
        if (!ran && !resolved) { fprintf(stderr, "kxemu: deadlock (some lanes wait at a barrier others never reach)\n"); failed = 1; break; }
    }
}
void launch(u32 nblocks, const std::function<void()>& fn) { launch_block(nblocks, 1, fn); }
void launch_block(u32 nblocks, int waves, const std::function<void()>& fn)
{
    cur_fn = &fn; num_blocks = nblocks; waves_per_block = waves;
    for (u32 b = 0; b < nblocks && !failed; b++) { cur_block = b; run_block(waves); }
}
}

extern "C" __attribute__((visibility("default"))) void emu_stats(unsigned long long* out, int clear)
{
    for (int i = 0; i < 64; i++) { out[i] = kxemu::stat[i]; if (clear) kxemu::stat[i] = 0; }
}
