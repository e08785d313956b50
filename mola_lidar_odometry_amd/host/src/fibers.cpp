// fibers.cpp -- see molahip_host/fibers.h
#include "molahip_host/fibers.h"

#include <sys/mman.h>

#include <stdexcept>

#include "molahip.h"

namespace molahip_host {

struct FiberScheduler::Fiber {
  ucontext_t ctx{};
  void* stack = nullptr;
  size_t stack_bytes = 0;
  std::function<void()> fn;
  std::exception_ptr error;
  bool started = false, done = false;
  FiberScheduler* owner = nullptr;
  ~Fiber() {
    if (stack) munmap(stack, stack_bytes);
  }
};

static thread_local FiberScheduler* tl_sched = nullptr;

FiberScheduler::FiberScheduler(size_t stack_bytes) : stack_bytes_((stack_bytes + 4095) & ~size_t(4095)) {}
FiberScheduler::~FiberScheduler() = default;

FiberScheduler* FiberScheduler::current() { return tl_sched; }
bool FiberScheduler::in_fiber() { return tl_sched && tl_sched->running_; }

void FiberScheduler::trampoline(unsigned lo, unsigned hi) {
  Fiber* f = reinterpret_cast<Fiber*>(((unsigned long long)hi << 32) | lo);
  try {
    f->fn();
  } catch (...) {
    f->error = std::current_exception();
  }
  f->done = true;
  FiberScheduler* s = f->owner;
  s->running_ = nullptr;
  swapcontext(&f->ctx, &s->main_);  // never resumed
}

FiberScheduler::Handle FiberScheduler::spawn(std::function<void()> fn) {
  auto f = std::make_shared<Fiber>();
  f->fn = std::move(fn);
  f->owner = this;
  f->stack_bytes = stack_bytes_ + 4096;
  f->stack = mmap(nullptr, f->stack_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
  if (f->stack == MAP_FAILED) {
    f->stack = nullptr;
    throw std::runtime_error("FiberScheduler: cannot allocate a fiber stack");
  }
  mprotect(f->stack, 4096, PROT_NONE);  // guard page below the stack
  getcontext(&f->ctx);
  f->ctx.uc_stack.ss_sp = static_cast<char*>(f->stack) + 4096;
  f->ctx.uc_stack.ss_size = stack_bytes_;
  f->ctx.uc_link = nullptr;
  const unsigned long long p = reinterpret_cast<unsigned long long>(f.get());
  makecontext(&f->ctx, reinterpret_cast<void (*)()>(&FiberScheduler::trampoline), 2, (unsigned)(p & 0xFFFFFFFFu), (unsigned)(p >> 32));
  fibers_.push_back(f);
  Handle h;
  h.f_ = f;
  return h;
}

void FiberScheduler::yield() {
  FiberScheduler* s = tl_sched;
  if (!s || !s->running_) return;
  Fiber* f = s->running_;
  s->running_ = nullptr;
  s->n_switches_++;
  swapcontext(&f->ctx, &s->main_);
}

void FiberScheduler::hook(void*) { yield(); }

void FiberScheduler::run() {
  if (tl_sched) throw std::runtime_error("FiberScheduler::run: a scheduler is already running on this thread");
  tl_sched = this;
  mh_set_wait_hook(&FiberScheduler::hook, nullptr);
  for (;;) {
    bool any = false;
    for (size_t i = 0; i < fibers_.size(); i++) {  // (fibers_ may grow while we iterate: index, not iterator)
      std::shared_ptr<Fiber> f = fibers_[i];
      if (f->done) continue;
      any = true;
      f->started = true;
      running_ = f.get();
      swapcontext(&main_, &f->ctx);
      running_ = nullptr;
    }
    if (!any) break;
    // finished fibers leave the list (their handles keep them alive as long as somebody asks)
    size_t w = 0;
    for (size_t i = 0; i < fibers_.size(); i++)
      if (!fibers_[i]->done) fibers_[w++] = fibers_[i];
    fibers_.resize(w);
  }
  mh_set_wait_hook(nullptr, nullptr);
  tl_sched = nullptr;
}

bool FiberScheduler::Handle::done() const { return !f_ || f_->done; }
void FiberScheduler::Handle::wait() {
  if (!f_) return;
  while (!f_->done) {
    if (!FiberScheduler::in_fiber()) throw std::runtime_error("FiberScheduler::Handle::wait outside a fiber of a running scheduler");
    FiberScheduler::yield();
  }
  if (f_->error) {
    std::exception_ptr e = f_->error;
    f_->error = nullptr;
    std::rethrow_exception(e);
  }
}

}  // namespace molahip_host
