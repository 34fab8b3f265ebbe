// Cooperative fibers: K protocol loops (verifier <-> prover) in ONE host thread.
//
// The reference drives one prover with one verifier loop per process (reference src/verifier.cpp:118-373, called from
// src/main_demo_vgg.cpp:40). A lock-step batch (include/zkcnn_hip.h: zk_batch_*) proves K pictures on one resident circuit with one kernel
// launch per round for all of them, which needs the K loops to advance together: each loop runs on its own small stack, and wherever its
// prover would wait for the GPU the HIP library's yield callback hands the thread to the next loop. Nothing spins, nothing locks, and the
// verifier code stays the one template (verifier.hpp) that every other mode uses.
//
// x86-64 System V: the callee-saved registers are saved by hand (the GPU box's host). Any other host: POSIX ucontext (swapcontext), so that the CPU
// oracle -- which includes this header through session.hpp and never yields -- builds everywhere (round-4 advisor finding).
// Stacks: ZKCNN_FIBER_STACK_MB megabytes each (default 4; the mapping is only reserved until touched), a guard page below.
// Define ZKFIBER_IMPLEMENTATION in exactly one translation unit per shared library.
#pragma once
#include <sys/mman.h>
#include <cstddef>
#include <cstdint>
#include <exception>
#include <functional>
#include <stdexcept>

#include <cstdlib>
#if !defined(__x86_64__) && !defined(ZKFIBER_UCONTEXT)
#define ZKFIBER_UCONTEXT 1          // (also selectable by hand: the fallback is tested on x86-64 that way)
#endif
#ifdef ZKFIBER_UCONTEXT
#include <ucontext.h>
#endif

#ifndef ZKFIBER_UCONTEXT
// saves the callee-saved registers and the stack pointer of the running context in *save_sp, continues the context whose stack pointer is load_sp
extern "C" void zkfiber_switch(void **save_sp, void *load_sp);

#ifdef ZKFIBER_IMPLEMENTATION
__asm__(".text\n"
        ".globl zkfiber_switch\n"
        ".hidden zkfiber_switch\n"
        ".type zkfiber_switch,@function\n"
        "zkfiber_switch:\n"
        "    pushq %rbp\n"
        "    pushq %rbx\n"
        "    pushq %r12\n"
        "    pushq %r13\n"
        "    pushq %r14\n"
        "    pushq %r15\n"
        "    movq %rsp, (%rdi)\n"
        "    movq %rsi, %rsp\n"
        "    popq %r15\n"
        "    popq %r14\n"
        "    popq %r13\n"
        "    popq %r12\n"
        "    popq %rbx\n"
        "    popq %rbp\n"
        "    ret\n"
        ".size zkfiber_switch,.-zkfiber_switch\n");
#endif
#endif      // !ZKFIBER_UCONTEXT

namespace zkfiber {

class fiber;
inline fiber *&current() {
    static thread_local fiber *f = nullptr;
    return f;
}

inline size_t default_stack_bytes() {
    const char *v = getenv("ZKCNN_FIBER_STACK_MB");
    const long mb = v ? atol(v) : 4;
    return (size_t) (mb >= 1 && mb <= 1024 ? mb : 4) << 20;
}

class fiber {
public:
    explicit fiber(std::function<void()> fn, size_t stack_bytes = 0) : entry(std::move(fn)), bytes((stack_bytes ? stack_bytes : default_stack_bytes()) + 4096) {
        stack = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
        if (stack == MAP_FAILED) { stack = nullptr; throw std::runtime_error("fiber: no memory for a stack"); }
        if (mprotect(stack, 4096, PROT_NONE) != 0) {                           // guard page below the stack: without it an overflow would be silent
            munmap(stack, bytes);
            stack = nullptr;
            throw std::runtime_error("fiber: cannot protect the guard page");
        }
#ifdef ZKFIBER_UCONTEXT
        if (getcontext(&uc) != 0) { munmap(stack, bytes); stack = nullptr; throw std::runtime_error("fiber: getcontext failed"); }
        uc.uc_stack.ss_sp = (char *) stack + 4096;
        uc.uc_stack.ss_size = bytes - 4096;
        uc.uc_link = nullptr;
        makecontext(&uc, (void (*)()) &fiber::fiber_main, 0);
#else
        // first switch: six register slots, then the `ret` into fiber_main with the stack aligned as after a call
        uintptr_t top = ((uintptr_t) stack + bytes) & ~(uintptr_t) 15;
        void **ret_slot = (void **) (top - 16);
        *ret_slot = (void *) &fiber::fiber_main;
        void **regs = ret_slot - 6;
        for (int i = 0; i < 6; ++i) regs[i] = nullptr;
        sp = regs;
#endif
    }
    ~fiber() { if (stack) munmap(stack, bytes); }
    fiber(const fiber &) = delete;
    fiber &operator=(const fiber &) = delete;

    bool done() const { return finished; }
    // runs the fiber until it yields or ends; an exception that escaped its function is rethrown here, in the driver
    void resume() {
        if (finished) return;
        fiber *outer = current();
        current() = this;
#ifdef ZKFIBER_UCONTEXT
        swapcontext(&caller_uc, &uc);
#else
        zkfiber_switch(&caller_sp, sp);
#endif
        current() = outer;
        if (error) { std::exception_ptr e = error; error = nullptr; std::rethrow_exception(e); }
    }
    // called on a fiber: back to whoever resumed it
    static void yield() {
        fiber *f = current();
        if (!f) return;                        // not on a fiber: nothing to hand over to
#ifdef ZKFIBER_UCONTEXT
        swapcontext(&f->uc, &f->caller_uc);
#else
        zkfiber_switch(&f->sp, f->caller_sp);
#endif
    }

private:
    static void fiber_main() {
        fiber *f = current();
        try { f->entry(); } catch (...) { f->error = std::current_exception(); }
        f->finished = true;
#ifdef ZKFIBER_UCONTEXT
        swapcontext(&f->uc, &f->caller_uc);
#else
        zkfiber_switch(&f->sp, f->caller_sp);
#endif
        __builtin_trap();                      // a finished fiber is never resumed
    }
    std::function<void()> entry;
    void *stack = nullptr;
    size_t bytes;
#ifdef ZKFIBER_UCONTEXT
    ucontext_t uc, caller_uc;
#else
    void *sp = nullptr, *caller_sp = nullptr;
#endif
    bool finished = false;
    std::exception_ptr error;
};

}  // namespace zkfiber
