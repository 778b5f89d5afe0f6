// TEST INFRASTRUCTURE ONLY -- fiber scheduler behind tests/emu/shim/hip/hip_runtime.h.
#include <hip/hip_runtime.h>

#include <memory>

namespace iplan_emu {

Block* g_block = nullptr;
Fiber* g_cur = nullptr;

static constexpr size_t kStack = 192 * 1024;
static std::vector<std::unique_ptr<char[]>> g_stacks;

float* dyn_lds() {
    static std::vector<float> buf(40960 + 64);
    return reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(buf.data()) + 63) & ~uintptr_t(63));
}

#ifdef IPLAN_EMU_FAST_SWITCH
// void iplan_emu_switch(Ctx* from, Ctx* to): park the callee-saved registers on the current stack, swap stack pointers
asm(R"(
    .text
    .globl iplan_emu_switch
    .type iplan_emu_switch, @function
iplan_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq (%rsi), %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size iplan_emu_switch, .-iplan_emu_switch
)");
#endif

static void trampoline() {
    Block* b = g_block;
    b->body();
    g_cur->done = true;
    b->alive--;
    Wave& w = b->waves[g_cur->wave];
    w.nlanes--;
    if (w.arrived > 0 && w.arrived >= w.nlanes) { w.arrived = 0; w.gen++; }
    if (b->arrived > 0 && b->arrived >= b->alive) { b->arrived = 0; b->gen++; }
    switch_ctx(&g_cur->ctx, &b->sched);
    __builtin_unreachable();
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const int nthreads = (int)(block.x * block.y * block.z);
    while ((int)g_stacks.size() < nthreads) g_stacks.emplace_back(new char[kStack]);
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                Block b;
                b.bid = dim3(bx, by, bz);
                b.bdim = block;
                b.gdim = grid;
                b.body = body;
                b.alive = nthreads;
                b.fibers.resize(nthreads);
                b.waves.resize((nthreads + 63) / 64);
                for (int t = 0; t < nthreads; ++t) {
                    Fiber& f = b.fibers[t];
                    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                    f.lane = t & 63;
                    f.wave = t >> 6;
                    b.waves[f.wave].nlanes++;
#ifdef IPLAN_EMU_FAST_SWITCH
                    // fresh stack as iplan_emu_switch expects it: six zeroed callee-saved registers, then `ret` into the
                    // trampoline with rsp = 8 mod 16 (as after a call); the slot above is a null return address
                    uintptr_t top = (reinterpret_cast<uintptr_t>(g_stacks[t].get()) + kStack) & ~uintptr_t(15);
                    void** sp = reinterpret_cast<void**>(top);
                    *--sp = nullptr;
                    *--sp = reinterpret_cast<void*>(&trampoline);
                    for (int r = 0; r < 6; ++r) *--sp = nullptr;
                    f.ctx.sp = sp;
#else
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = g_stacks[t].get();
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, trampoline, 0);
#endif
                }
                g_block = &b;
                int remaining = nthreads;
                while (remaining > 0) {
                    remaining = 0;
                    for (int t = 0; t < nthreads; ++t) {
                        Fiber& f = b.fibers[t];
                        if (f.done) continue;
                        g_cur = &f;
                        switch_ctx(&b.sched, &f.ctx);
                        if (!f.done) remaining++;
                    }
                }
                g_block = nullptr;
                g_cur = nullptr;
            }
}

}  // namespace iplan_emu
