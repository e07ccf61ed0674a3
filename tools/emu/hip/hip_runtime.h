// tools/emu/hip/hip_runtime.h -- a host stand-in for <hip/hip_runtime.h>, TEST INFRASTRUCTURE ONLY.
//
// Compiling a kernel source with `g++ -DSPNG_EMU -Itools/emu` makes this header shadow the HIP one:
// the kernels of csrc/*.hip then run on the CPU, one fiber per GPU thread, so that their LOGIC (token
// chains, prefix sums, pointer jumping, page bookkeeping) can be checked against zlib in the build
// container, which has no GPU.  Nothing here is part of the product and nothing in the product
// includes it.  What is modelled:
//   * a workgroup = blockDim fibers scheduled round-robin on one host thread; __shared__ = static;
//   * wave-wide builtins (ballot, shuffles, readlane, DPP row/wave shifts, ds_bpermute, the LDS fence
//     used as wave barrier) are rendezvous points of the 64 fibers of a wave: every lane that has not
//     returned must arrive at the same builtin (the kernels only use them in wave-uniform control flow;
//     a mismatch aborts with a message);
//   * __syncthreads is a rendezvous of the whole workgroup.
// What is not modelled: timing, memory ordering weaker than sequential, bank conflicts, divergence
// inside wave builtins.
#pragma once
#ifndef SPNG_EMU
#error "tools/emu is a host emulation shim: compile with -DSPNG_EMU"
#endif
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#if defined(__x86_64__) && !defined(EMU_UCONTEXT)
#define EMU_FAST_SWITCH 1      // a context switch of our own: glibc's swapcontext saves and restores the signal mask with a system
                               // call per switch -- half of the emulator's run time was spent in the kernel
#else
#include <ucontext.h>
#endif
#include <functional>
#include <vector>

typedef int hipError_t;
typedef void *hipStream_t;
static const hipError_t hipSuccess = 0, hipErrorInvalidValue = 1;
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
inline hipError_t hipGetLastError() { return hipSuccess; }

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__

namespace emu {

struct dim3e { unsigned x, y, z; };
enum { RUN = 0, WAIT_WAVE = 1, WAIT_BLOCK = 2, DONE = 3 };
enum { OP_BALLOT = 1, OP_SHFL, OP_DPP, OP_FENCE, OP_READFIRST, OP_SYNCOR };

#ifdef EMU_FAST_SWITCH
// Saves the callee-saved registers of the System V x86-64 ABI on the current stack, stores the stack pointer in *from, takes *to
// as the stack pointer and restores from there; a fresh fiber's stack is laid out as if it had been switched away from in front
// of its entry function (launch()).  Fibers are plain integer code: the x87 / SSE control words are the same everywhere.
struct Ctx { void *sp = nullptr; };
extern "C" void emu_switch(Ctx *from, Ctx *to);
__asm__(".text\n"
        ".globl emu_switch\n"
        ".type emu_switch,@function\n"
        "emu_switch:\n"
        "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
        "  movq %rsp, (%rdi)\n"
        "  movq (%rsi), %rsp\n"
        "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
        "  ret\n"
        ".size emu_switch,.-emu_switch\n");
#else
struct Ctx { ucontext_t uc; };
inline void emu_switch(Ctx *from, Ctx *to) { swapcontext(&from->uc, &to->uc); }
#endif

struct Fiber {
    Ctx ctx;
    char *stack = nullptr;                              // (from launch()'s pool: allocated once, never zeroed again)
    size_t stack_size = 0;
    dim3e tid, bid, bdim, gdim;
    int state = RUN;
    int op = 0;
    uint64_t in0 = 0, in1 = 0, out = 0;
    int src_lane = 0;
    int line = 0;                                       // source line of the wave builtin the fiber waits at (EMU_WHERE)
    const std::function<void()> *body = nullptr;
};

inline Fiber *&cur() { static Fiber *c = nullptr; return c; }
inline Ctx &sched_ctx() { static Ctx c; return c; }

inline void yield_to_sched() { emu_switch(&cur()->ctx, &sched_ctx()); }

inline uint64_t wave_op(int op, uint64_t a, uint64_t b = 0, int src = 0)
{
    Fiber *f = cur();
    f->op = op; f->in0 = a; f->in1 = b; f->src_lane = src; f->state = WAIT_WAVE;
    yield_to_sched();
    return f->out;
}

#ifdef EMU_FAST_SWITCH
inline void fiber_entry()                                       // (entered by emu_switch's `ret`: the fiber is cur())
{
    Fiber *f = cur();
    (*f->body)();
    f->state = DONE;
    emu_switch(&f->ctx, &sched_ctx());
    abort();                                                    // (a finished fiber is not switched to again)
}
#else
inline void trampoline(unsigned lo, unsigned hi)
{
    Fiber *f = (Fiber *)(((uintptr_t)hi << 32) | lo);
    (*f->body)();
    f->state = DONE;
    emu_switch(&f->ctx, &sched_ctx());
}
#endif

// resolves a wave whose live lanes all wait at a wave op
inline void resolve_wave(std::vector<Fiber> &fb, size_t base, size_t n)
{
    int op = 0;
    for (size_t l = 0; l < n; ++l) {
        Fiber &f = fb[base + l];
        if (f.state == DONE) continue;
        if (!op) op = f.op;
        if (f.op != op) {
            fprintf(stderr, "emu: wave diverged at a wave builtin (ops %d vs %d, block %u wave %zu)\n", op, f.op, f.bid.x, base / 64);
            for (size_t k = 0; k < n; ++k) if (fb[base + k].state != DONE) fprintf(stderr, " l%zu:op%d@%d", k, fb[base + k].op, fb[base + k].line);
            fprintf(stderr, "\n");
            abort();
        }
    }
    uint64_t ballot = 0;
    for (size_t l = 0; l < n; ++l) if (fb[base + l].state != DONE && fb[base + l].in0) ballot |= 1ull << l;
    int first = -1;
    for (size_t l = 0; l < n; ++l) if (fb[base + l].state != DONE) { first = (int)l; break; }
    for (size_t l = 0; l < n; ++l) {
        Fiber &f = fb[base + l];
        if (f.state == DONE) continue;
        switch (op) {
        case OP_BALLOT: f.out = ballot; break;
        case OP_FENCE: f.out = 0; break;
        case OP_READFIRST: f.out = fb[base + first].in0; break;
        case OP_SHFL: {
            const int s = f.src_lane & 63;
            f.out = ((size_t)s < n && fb[base + s].state != DONE) ? fb[base + s].in0 : f.in0;
            break;
        }
        case OP_DPP: {
            // in0 = src, in1 = old; src_lane = source lane or -1 (keep old / bound_ctrl zero handled by caller)
            const int s = f.src_lane;
            f.out = (s >= 0 && (size_t)s < n && fb[base + s].state != DONE) ? fb[base + s].in0 : f.in1;
            break;
        }
        default: fprintf(stderr, "emu: unknown wave op %d\n", op); abort();
        }
        f.state = RUN;
    }
}

// Runs `body` for every thread of every block, blocks one after another.
inline void launch(unsigned grid, unsigned block, const std::function<void()> &body, unsigned gridy = 1)
{
    static std::vector<Fiber> fb;
    for (unsigned b = 0; b < grid * gridy; ++b) {
        fb.clear(); fb.resize(block);
        for (unsigned t = 0; t < block; ++t) {
            Fiber &f = fb[t];
            // (stacks are kept between launches: a fresh 256 KB vector per fiber and launch was a mmap, a page-fault storm and a
            // munmap each)
            static std::vector<char *> pool;
            while (pool.size() <= t) pool.push_back((char *)malloc(256 << 10));
            f.stack = pool[t]; f.stack_size = 256 << 10;
            f.tid = {t, 0, 0}; f.bid = {b % grid, b / grid, 0}; f.bdim = {block, 1, 1}; f.gdim = {grid, gridy, 1};
            f.body = &body; f.state = RUN;
#ifdef EMU_FAST_SWITCH
            // the stack as emu_switch leaves one: six saved registers, the address it returns to (the entry function), and a
            // slot above it so that the entry function finds the stack aligned as after a call
            uintptr_t top = ((uintptr_t)f.stack + f.stack_size) & ~(uintptr_t)15;
            void **sp = (void **)top;
            *--sp = nullptr;                                    // (where a return address would be: fiber_entry never returns)
            *--sp = (void *)fiber_entry;
            for (int r = 0; r < 6; ++r) *--sp = nullptr;        // rbp, rbx, r12 - r15
            f.ctx.sp = sp;
#else
            getcontext(&f.ctx.uc);
            f.ctx.uc.uc_stack.ss_sp = f.stack; f.ctx.uc.uc_stack.ss_size = f.stack_size; f.ctx.uc.uc_link = nullptr;
            const uintptr_t p = (uintptr_t)&f;
            makecontext(&f.ctx.uc, (void (*)())trampoline, 2, (unsigned)(p & 0xffffffffu), (unsigned)(p >> 32));
#endif
        }
        for (;;) {
            bool progressed = false, all_done = true;
            for (unsigned t = 0; t < block; ++t) {
                Fiber &f = fb[t];
                if (f.state == RUN) { cur() = &f; emu_switch(&sched_ctx(), &f.ctx); progressed = true; }
                if (f.state != DONE) all_done = false;
            }
            if (all_done) break;
            // waves
            for (unsigned w = 0; w * 64 < block; ++w) {
                const size_t base = (size_t)w * 64, n = block - base < 64 ? block - base : 64;
                bool any = false, all = true;
                for (size_t l = 0; l < n; ++l) {
                    const int st = fb[base + l].state;
                    if (st == DONE) continue;
                    any = true;
                    if (st != WAIT_WAVE) all = false;
                }
                if (any && all) { resolve_wave(fb, base, n); progressed = true; }
            }
            // block barrier
            {
                bool any = false, all = true; uint64_t orv = 0;
                for (unsigned t = 0; t < block; ++t) {
                    const int st = fb[t].state;
                    if (st == DONE) continue;
                    any = true;
                    if (st != WAIT_BLOCK) all = false; else orv |= fb[t].in0;
                }
                if (any && all) { for (unsigned t = 0; t < block; ++t) if (fb[t].state == WAIT_BLOCK) { fb[t].out = orv; fb[t].state = RUN; } progressed = true; }
            }
            if (!progressed) {
                fprintf(stderr, "emu: deadlock in block %u:", b);
                for (unsigned t = 0; t < block && t < 16; ++t) fprintf(stderr, " t%u:%d/op%d", t, fb[t].state, fb[t].op);
                fprintf(stderr, "\n");
                abort();
            }
        }
    }
    cur() = nullptr;
}

inline uint64_t block_sync(uint64_t v)
{
    Fiber *f = cur();
    f->in0 = v; f->state = WAIT_BLOCK;
    yield_to_sched();
    return f->out;
}

}  // namespace emu

#define threadIdx (emu::cur()->tid)
#define blockIdx (emu::cur()->bid)
#define blockDim (emu::cur()->bdim)
#define gridDim (emu::cur()->gdim)

// ---- builtins ------------------------------------------------------------------------------------------
#define __ATOMIC_SEQ_CST_EMU 5
inline void __builtin_amdgcn_fence(int, const char *, ...) { emu::wave_op(emu::OP_FENCE, 0); }
inline void __syncthreads() { emu::block_sync(0); }
inline int __syncthreads_or(int p) { return emu::block_sync(p ? 1 : 0) != 0; }
inline unsigned long long __ballot(int p) { return emu::wave_op(emu::OP_BALLOT, p ? 1 : 0); }
inline int __shfl(int v, int lane, int = 64) { return (int)(uint32_t)emu::wave_op(emu::OP_SHFL, (uint32_t)v, 0, lane); }
inline int __shfl_xor(int v, int m, int = 64) { return (int)(uint32_t)emu::wave_op(emu::OP_SHFL, (uint32_t)v, 0, (int)((threadIdx.x & 63) ^ (unsigned)m)); }
inline unsigned __shfl_xor(unsigned v, int m, int = 64) { return (unsigned)__shfl_xor((int)v, m); }
inline int __shfl_up(int v, int d, int = 64)
{
    const int lane = (int)(threadIdx.x & 63);
    return (int)(uint32_t)emu::wave_op(emu::OP_SHFL, (uint32_t)v, 0, lane >= d ? lane - d : lane);
}
inline int __shfl_down(int v, int d, int = 64)
{
    const int lane = (int)(threadIdx.x & 63);
    return (int)(uint32_t)emu::wave_op(emu::OP_SHFL, (uint32_t)v, 0, lane + d < 64 ? lane + d : lane);
}
inline int __builtin_amdgcn_readlane(int v, int lane) { return (int)(uint32_t)emu::wave_op(emu::OP_SHFL, (uint32_t)v, 0, lane); }
inline int __builtin_amdgcn_readfirstlane(int v) { return (int)(uint32_t)emu::wave_op(emu::OP_READFIRST, (uint32_t)v); }
inline int __builtin_amdgcn_ds_bpermute(int addr, int v) { return (int)(uint32_t)emu::wave_op(emu::OP_SHFL, (uint32_t)v, 0, (addr >> 2) & 63); }
// DPP: row_shr:n (0x111..0x11f), row_shl:n (0x101..0x10f), wave_shr:1 (0x138), wave_shl:1 (0x130); full row/bank masks
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
    const int lane = (int)(threadIdx.x & 63);
    int s = -1;
    if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = ctrl & 15; if ((lane & 15) >= n) s = lane - n; }
    else if (ctrl >= 0x101 && ctrl <= 0x10f) { const int n = ctrl & 15; if ((lane & 15) + n < 16) s = lane + n; }
    else if (ctrl == 0x138) { if (lane >= 1) s = lane - 1; }
    else if (ctrl == 0x130) { if (lane < 63) s = lane + 1; }
    else { fprintf(stderr, "emu: dpp ctrl 0x%x not modelled\n", ctrl); abort(); }
    if (row_mask != 0xf || bank_mask != 0xf) { fprintf(stderr, "emu: dpp masks not modelled\n"); abort(); }
    const uint32_t fallback = bound_ctrl ? 0u : (uint32_t)old;
    return (int)(uint32_t)emu::wave_op(emu::OP_DPP, (uint32_t)src, fallback, s);
}
inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned base)
{
    const unsigned lane = threadIdx.x & 63;
    const unsigned m = lane >= 32 ? mask : (mask & ((1u << lane) - 1));
    return base + (unsigned)__builtin_popcount(m);
}
inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned base)
{
    const unsigned lane = threadIdx.x & 63;
    const unsigned m = lane <= 32 ? 0u : (mask & ((1u << (lane - 32)) - 1));
    return base + (unsigned)__builtin_popcount(m);
}
inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)((((uint64_t)hi << 32) | lo) >> (sh & 31)); }
inline unsigned __builtin_amdgcn_sad_u8(unsigned a, unsigned b, unsigned acc)
{
    for (int k = 0; k < 4; ++k) { const int x = (a >> (8 * k)) & 255, y = (b >> (8 * k)) & 255; acc += (unsigned)(x > y ? x - y : y - x); }
    return acc;
}
inline unsigned __builtin_amdgcn_udot4(unsigned a, unsigned b, unsigned acc, bool)
{
    for (int k = 0; k < 4; ++k) acc += ((a >> (8 * k)) & 255) * ((b >> (8 * k)) & 255);
    return acc;
}
inline unsigned __brev(unsigned v) { unsigned r = 0; for (int k = 0; k < 32; ++k) r |= ((v >> k) & 1u) << (31 - k); return r; }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline float __fdividef(float a, float b) { return a / b; }
inline float __builtin_amdgcn_rcpf(float a) { return 1.0f / a; }
inline unsigned long long __builtin_readcyclecounter_emu() { return 0; }
#ifdef __clang__
// (clang -- the host compiler for kernels written with its vector extensions -- has the __hip_atomic builtins itself)
#ifndef __HIP_MEMORY_SCOPE_WORKGROUP
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#endif
#else
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
#define __HIP_MEMORY_SCOPE_AGENT 1
template <class T> inline T __hip_atomic_fetch_add(T *p, T v, int, int) { T o = *p; *p = o + v; return o; }
template <class T> inline T __hip_atomic_fetch_or(T *p, T v, int, int) { T o = *p; *p = o | v; return o; }
template <class T> inline T __hip_atomic_load(const T *p, int, int) { return *p; }
template <class T> inline void __hip_atomic_store(T *p, T v, int, int) { *p = v; }
#endif
inline unsigned atomicOr(unsigned *p, unsigned v) { unsigned o = *p; *p = o | v; return o; }
inline void __threadfence() {}
inline unsigned atomicAdd(unsigned *p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
inline unsigned atomicMax(unsigned *p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }
inline unsigned atomicMin(unsigned *p, unsigned v) { unsigned o = *p; if (v < o) *p = v; return o; }
#ifndef __clang__
template <class T> inline T __hip_atomic_fetch_min(T *p, T v, int, int) { T o = *p; if (v < o) *p = v; return o; }
#endif
inline void __builtin_amdgcn_s_setprio(int) {}
// v_alignbyte_b32: ({hi, lo} >> 8 * (sh & 3)) & 0xffffffff
inline unsigned __builtin_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)((((uint64_t)hi << 32) | lo) >> (8 * (sh & 3))); }
inline int __any(int p) { return emu::wave_op(emu::OP_BALLOT, p ? 1 : 0) != 0; }
// v_perm_b32: byte k of the result = byte sel[k] of {hi, lo} (0-3: lo, 4-7: hi); 0x0c: 0x00 (the other special selectors are not used)
inline unsigned __builtin_amdgcn_perm(unsigned hi, unsigned lo, unsigned sel)
{
    const uint64_t both = (uint64_t)hi << 32 | lo;
    unsigned r = 0;
    for (int k = 0; k < 4; ++k) {
        const unsigned c = (sel >> (8 * k)) & 0xff;
        unsigned b;
        if (c <= 7) b = (unsigned)(both >> (8 * c)) & 0xff;
        else if (c == 0x0c) b = 0;
        else if (c >= 0x0d) b = 0xff;
        else { fprintf(stderr, "emu: v_perm selector 0x%02x not modelled\n", c); abort(); }
        r |= b << (8 * k);
    }
    return r;
}
// v_lerp_u8: per byte (a + b + (c & 1)) >> 1
inline unsigned __builtin_amdgcn_lerp(unsigned a, unsigned b, unsigned c)
{
    unsigned r = 0;
    for (int k = 0; k < 4; ++k) r |= ((((a >> (8 * k)) & 255) + ((b >> (8 * k)) & 255) + ((c >> (8 * k)) & 1)) >> 1) << (8 * k);
    return r;
}
// s_sleep inside a wave-uniform spin loop: the wave meets and the scheduler turns to the other waves of the workgroup
inline void __builtin_amdgcn_s_sleep(int) { emu::wave_op(emu::OP_FENCE, 0); }
inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }

// ---- diagnostics: remember the source line of every wave builtin a fiber waits at (printed when a wave diverges) -----------
#define __builtin_amdgcn_readfirstlane(v) (emu::cur()->line = __LINE__, __builtin_amdgcn_readfirstlane(v))
#define __builtin_amdgcn_readlane(v, l) (emu::cur()->line = __LINE__, __builtin_amdgcn_readlane(v, l))
#define __ballot(p) (emu::cur()->line = __LINE__, __ballot(p))
#define __shfl(...) (emu::cur()->line = __LINE__, __shfl(__VA_ARGS__))
#define __shfl_xor(...) (emu::cur()->line = __LINE__, __shfl_xor(__VA_ARGS__))
#define __builtin_amdgcn_fence(...) (emu::cur()->line = __LINE__, __builtin_amdgcn_fence(__VA_ARGS__))
#define __builtin_amdgcn_update_dpp(...) (emu::cur()->line = __LINE__, __builtin_amdgcn_update_dpp(__VA_ARGS__))
#define __builtin_amdgcn_s_sleep(n) (emu::cur()->line = __LINE__, __builtin_amdgcn_s_sleep(n))
