// fibers.h -- several sequences on ONE host thread.
//
// The reference runs independent sequences as independent PROCESSES (eval/cli_kitti.sh:23-36: GNU parallel -j3).  On one GPU
// that wastes it (a small-layer alignment is a chain of short dependent kernels), and N host THREADS in one process
// contend for the HIP runtime's locks: with eight of them every upload / filter / map-update call took 3-4x its solo
// time (two batches in flight at once were slower still).  So the multi-sequence runner keeps ONE thread in
// the runtime: every sequence is a fiber (ucontext) of that thread, and libmolahip's blocking waits hand control to the
// scheduler through mh_set_wait_hook() -- while one sequence waits for its filter counts or for the batch alignment, the
// others issue their work.  Plain cooperative round robin; no fiber ever runs on another thread.
#pragma once
#include <ucontext.h>

#include <exception>
#include <functional>
#include <memory>
#include <vector>

namespace molahip_host {

class FiberScheduler {
 public:
  struct Fiber;
  // what spawn() returns: done() / wait() (yields until the fiber has finished; rethrows what it threw)
  class Handle {
   public:
    Handle() = default;
    bool valid() const { return (bool)f_; }
    bool done() const;
    void wait();
   private:
    friend class FiberScheduler;
    std::shared_ptr<Fiber> f_;
  };

  explicit FiberScheduler(size_t stack_bytes = 1u << 20);
  ~FiberScheduler();
  FiberScheduler(const FiberScheduler&) = delete;
  FiberScheduler& operator=(const FiberScheduler&) = delete;

  Handle spawn(std::function<void()> fn);  // from the thread that owns the scheduler (outside run(), or from a fiber)
  void run();                              // until every fiber has finished; installs libmolahip's wait hook meanwhile
  size_t switches() const { return n_switches_; }

  static FiberScheduler* current();  // the scheduler whose run() is active on this thread, or nullptr
  static bool in_fiber();
  static void yield();               // inside a fiber: let the others run; elsewhere: nothing

 private:
  static void trampoline(unsigned lo, unsigned hi);
  static void hook(void*);
  std::vector<std::shared_ptr<Fiber>> fibers_;
  ucontext_t main_{};
  Fiber* running_ = nullptr;
  size_t stack_bytes_;
  size_t n_switches_ = 0;
};

}  // namespace molahip_host
