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

static void trampoline() {
    Block* b = g_block;
    b->body();
    g_cur->done = true;
    b->alive--;
    Wave& w = b->waves[g_cur->wave];
    w.nlanes--;
    if (w.arrived > 0 && w.arrived >= w.nlanes) { w.arrived = 0; w.gen++; }
    if (b->arrived > 0 && b->arrived >= b->alive) { b->arrived = 0; b->gen++; }
    swapcontext(&g_cur->ctx, &b->sched);
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
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = g_stacks[t].get();
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, trampoline, 0);
                }
                g_block = &b;
                int remaining = nthreads;
                while (remaining > 0) {
                    remaining = 0;
                    for (int t = 0; t < nthreads; ++t) {
                        Fiber& f = b.fibers[t];
                        if (f.done) continue;
                        g_cur = &f;
                        swapcontext(&b.sched, &f.ctx);
                        if (!f.done) remaining++;
                    }
                }
                g_block = nullptr;
                g_cur = nullptr;
            }
}

}  // namespace iplan_emu
