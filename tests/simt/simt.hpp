// simt.hpp -- TEST INFRASTRUCTURE ONLY.
//
// A small lock-step SIMT emulator so that the *actual kernel sources* under lz4net_amd/csrc/*.hpp
// can be compiled with g++ and executed in a GPU-less container.  Every lane of a workgroup is a
// ucontext fiber; cross-lane operations (readlane, shuffle, ballot, wave/mem sync, __syncthreads)
// are rendezvous points at which a fiber yields to the scheduler, which runs the live lanes of a
// wave in lane order between two rendezvous.  That is STRICTER than hardware lock-step (a lane may
// only observe another lane's memory writes across a wv::mem_sync()/collective), so code that
// passes here cannot depend on accidental instruction-level lock-step.
//
// It cannot see GPU memory-model, alignment or performance problems; those are what `-m gpu` tests
// and rocprof are for.  Nothing here is ever linked into liblz4hip.so.
#pragma once
#include <ucontext.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---- fiber switch ---------------------------------------------------------------------------------------------------------
// glibc's swapcontext() saves and restores the signal mask: two system calls per switch, and the emulator switches fibers
// millions of times per test.  On x86-64 a switch only has to exchange the callee-saved registers and the stack pointer
// (System V ABI); everywhere else ucontext is used as it is.
#if defined(__x86_64__)
extern "C" void simt_switch_stack(void** save_sp, void* const* load_sp);
asm(R"(
    .text
    .globl simt_switch_stack
    .type simt_switch_stack,@function
simt_switch_stack:
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
    .size simt_switch_stack, .-simt_switch_stack
)");
struct FiberCtx { void* sp = nullptr; };
inline void fiber_switch(FiberCtx* from, FiberCtx* to) { simt_switch_stack(&from->sp, &to->sp); }
inline void fiber_make(FiberCtx* c, unsigned char* stack, size_t bytes, void (*entry)())
{
    uintptr_t top = ((uintptr_t)stack + bytes) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                 // the entry function's "return address": it never returns
    *--sp = (void*)entry;            // popped by the `ret` of the first switch into this fiber
    for (int k = 0; k < 6; k++) *--sp = nullptr;   // rbp, rbx, r12 .. r15
    c->sp = (void*)sp;
}
#else
struct FiberCtx { ucontext_t uc; };
inline void fiber_switch(FiberCtx* from, FiberCtx* to) { swapcontext(&from->uc, &to->uc); }
inline void fiber_make(FiberCtx* c, unsigned char* stack, size_t bytes, void (*entry)())
{
    getcontext(&c->uc);
    c->uc.uc_stack.ss_sp = stack; c->uc.uc_stack.ss_size = bytes; c->uc.uc_link = nullptr;
    makecontext(&c->uc, entry, 0);
}
#endif

namespace simt {

enum WaitKind { WAIT_NONE = 0, WAIT_WAVE = 1, WAIT_BLOCK = 2 };

// a vector-memory load issued through wv::vm_load16_pred and not yet waited for (wv::vm_wait<N>): address + destination
struct PendingVm { uint64_t addr; uint32_t* dst; bool is_load; };

struct Lane {
    std::vector<PendingVm> vmq;   // in issue order
    FiberCtx ctx;
    std::vector<unsigned char> stack;
    int tid = 0;            // thread index in block
    bool done = false;
    int wait_kind = WAIT_NONE;
    int wait_site = 0;
    unsigned seq = 0;       // number of collectives executed (parity selects the exchange buffer)
    uint64_t slot[2] = {0, 0};
    unsigned slot_tag[2] = {~0u, ~0u};   // value of `seq` when slot[p] was published
};

struct Runtime {
    dim3 grid, block, block_idx;
    std::vector<Lane> lanes;
    Lane* cur = nullptr;
    FiberCtx sched;
    std::vector<unsigned char> lds;
    std::function<void()> body;
    uint64_t steps = 0;
};

inline Runtime& rt()
{
    static Runtime r;
    return r;
}

[[noreturn]] inline void die(const char* msg, int a = 0, int b = 0)
{
    std::fprintf(stderr, "SIMT-EMU FATAL: %s (%d, %d) block=%u tid=%d\n", msg, a, b, rt().block_idx.x,
                 rt().cur ? rt().cur->tid : -1);
    std::abort();
}

inline void yield(int kind, int site)
{
    Lane* me = rt().cur;
    me->wait_kind = kind;
    me->wait_site = site;
    fiber_switch(&me->ctx, &rt().sched);
}

inline void trampoline()
{
    Runtime& r = rt();
    r.body();
    r.cur->done = true;
    r.cur->wait_kind = WAIT_NONE;
    fiber_switch(&r.cur->ctx, &r.sched);
    die("resumed a finished lane");
}

// lanes [w*64, w*64+64) of the current block
inline Lane* wave_base(Lane* l) { return &rt().lanes[(size_t)(l->tid & ~63)]; }
inline int wave_width(Lane* l)
{
    int base = l->tid & ~63, n = (int)rt().lanes.size() - base;
    return n < 64 ? n : 64;
}

inline void run_block(size_t lds_bytes)
{
    Runtime& r = rt();
    const int nthreads = (int)(r.block.x * r.block.y * r.block.z);
    r.lanes.resize((size_t)nthreads);
    r.lds.assign(lds_bytes + 64, 0xCD);   // poison: real LDS is uninitialised too
    for (int t = 0; t < nthreads; t++) {
        Lane& l = r.lanes[(size_t)t];
        if (l.stack.empty()) l.stack.resize(256 * 1024);
        l.tid = t; l.done = false; l.wait_kind = WAIT_NONE; l.seq = 0; l.slot_tag[0] = l.slot_tag[1] = ~0u;
        l.vmq.clear();
        fiber_make(&l.ctx, l.stack.data(), l.stack.size(), (void (*)())trampoline);
    }
    const int nwaves = (nthreads + 63) / 64;
    std::vector<char> at_barrier((size_t)nwaves, 0);
    for (;;) {
        bool any_live = false, all_blocked = true;
        for (int w = 0; w < nwaves; w++) {
            int lo = w * 64, hi = lo + 64 < nthreads ? lo + 64 : nthreads;
            bool live = false;
            for (int t = lo; t < hi; t++) live |= !r.lanes[(size_t)t].done;
            if (!live) continue;
            any_live = true;
            if (at_barrier[(size_t)w]) continue;
            all_blocked = false;
            for (int t = lo; t < hi; t++) {
                Lane& l = r.lanes[(size_t)t];
                if (l.done) continue;
                r.cur = &l;
                fiber_switch(&r.sched, &l.ctx);
                r.steps++;
            }
            // every lane that is still alive must be waiting at the same rendezvous
            int kind = -1, site = -1;
            for (int t = lo; t < hi; t++) {
                Lane& l = r.lanes[(size_t)t];
                if (l.done) continue;
                if (kind < 0) { kind = l.wait_kind; site = l.wait_site; }
                else if (kind != l.wait_kind || site != l.wait_site) {
                    r.cur = &l;
                    die("divergent collective: lanes of one wave wait at different sites", site, l.wait_site);
                }
            }
            if (kind == WAIT_BLOCK) at_barrier[(size_t)w] = 1;
        }
        if (!any_live) break;
        if (all_blocked) std::fill(at_barrier.begin(), at_barrier.end(), 0);   // release __syncthreads
    }
    r.cur = nullptr;
}

// Launch `body` (a closure that calls the kernel function with its arguments) over a 1-D/2-D grid.
inline void launch(dim3 grid, dim3 block, size_t lds_bytes, std::function<void()> body)
{
    Runtime& r = rt();
    r.grid = grid; r.block = block; r.body = std::move(body);
    for (unsigned by = 0; by < grid.y; by++)
        for (unsigned bx = 0; bx < grid.x; bx++) {
            r.block_idx = dim3(bx, by, 0);
            run_block(lds_bytes);
        }
}

struct ThreadIdxProxy { unsigned x, y, z; };
inline ThreadIdxProxy tidx()
{
    Runtime& r = rt();
    unsigned t = (unsigned)r.cur->tid;
    return ThreadIdxProxy{ t % r.block.x, (t / r.block.x) % r.block.y, t / (r.block.x * r.block.y) };
}

// ---- collectives ---------------------------------------------------------------------------
// Exchange: publish v, rendezvous, return the wave's slots for this collective.  A lane took part in
// THIS collective iff took_part(lane, p, tag) -- lanes that already left the kernel did not, and a
// lane that leaves right after the rendezvous still did (its `done` flag says nothing about that).
inline const Lane* exchange(uint64_t v, int site, unsigned* parity_out, unsigned* tag_out)
{
    Lane* me = rt().cur;
    unsigned p = me->seq & 1u;
    me->slot[p] = v;
    me->slot_tag[p] = me->seq;
    *tag_out = me->seq;
    me->seq++;
    yield(WAIT_WAVE, site);
    *parity_out = p;
    return wave_base(me);
}
inline bool took_part(const Lane& l, unsigned p, unsigned tag) { return l.slot_tag[p] == tag; }

}  // namespace simt

// device atomics: lanes never run concurrently in the emulator
template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = (T)(o + v); return o; }
template <class T> inline T atomicCAS(T* p, T expected, T desired) { T o = *p; if (o == expected) *p = desired; return o; }
template <class T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }

#define threadIdx (simt::tidx())
#define blockIdx (simt::rt().block_idx)
#define blockDim (simt::rt().block)
#define gridDim (simt::rt().grid)
