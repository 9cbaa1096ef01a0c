// simt.h — TEST INFRASTRUCTURE ONLY.  Not part of the product, never loaded by it.
//
// A 64-lane lock-step shim for the WAVEFRONT kernels of fgumi_amd/csrc/fastpath.hip ("compile the wavefront kernels'
// bodies for the host behind a 64-lane lock-step shim — __ballot, __shfl*, DPP, mbcnt, LDS as an array, __syncthreads as a phase
// boundary").  tests/devemu runs lane-per-item kernels as a serial loop over threads; that cannot run a kernel whose lanes talk to one
// another.  Here every thread of a workgroup is a FIBER (its own stack, a ~15 ns hand-written context switch, one OS thread), and a
// cross-lane operation is a rendezvous:
//
//   * a lane that reaches __ballot / __shfl* / readlane / readfirstlane / DPP / ds_bpermute / uicmp / __any / __all / a wave barrier deposits
//     its operands and yields; when EVERY live lane of its wavefront waits — at the same kind of operation — the operation is evaluated for
//     all of them at once with the hardware's semantics (lanes that have left the kernel are inactive: they provide nothing, and
//     readfirstlane takes the lowest live lane) and the lanes run on;
//   * __syncthreads / __syncthreads_and are the same across the workgroup's wavefronts;
//   * __shared__ variables are per-OS-thread statics (one workgroup runs at a time), dynamic LDS a buffer per launch, global and LDS atomics
//     plain operations (fibers are cooperative: nothing interleaves inside one);
//   * between two rendezvous a lane runs ALONE.  That is the difference from hardware: real lanes execute every instruction together, so a
//     kernel that lets lane B read what lane A stored "one instruction earlier" without a wave barrier in between works on the GPU by
//     accident of lock-step and FAILS here (lane B may run first) — the kernels mark those hand-overs with wave_sync(), which is a
//     rendezvous.  Lanes of one wavefront that wait at DIFFERENT operations (a cross-lane operation under lane-divergent control flow)
//     stop the run with a report: the kernels are written in wave-uniform phases, and this checks it.
//
// What it cannot show: timing, occupancy, register pressure, bank conflicts, the memory model across wavefronts on real caches.
#pragma once
#include <dlfcn.h>
#include <signal.h>
#include <ucontext.h>
#include <unistd.h>
#include <sys/mman.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

namespace wavemu {

constexpr size_t STACK_BYTES = 512 * 1024;
constexpr int MAX_THREADS = 1024;

enum Op : uint32_t { OP_NONE = 0, OP_BALLOT, OP_SHFL, OP_READLANE, OP_READFIRST, OP_DPP, OP_SYNC, OP_BLOCK_SYNC, OP_BLOCK_AND, OP_BLOCK_COUNT };

struct Idx { uint32_t x = 0, y = 0, z = 0; };
struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = true;
  uint32_t op = OP_NONE;
  uint64_t a = 0, b = 0, c = 0, result = 0;
  const char* site = nullptr;    // source location of the operation's macro (for reports)
  const void* pc = nullptr;      // where the operation was called from: the return address of wave_op in the INLINED kernel code — two arms of a divergent
                                 // conditional that both call the same small helper (uni(), rlane()) are two places, as they are two branches on the device
  Idx tid;
};
struct Block {
  Fiber f[MAX_THREADS];
  int n = 0;
  Idx bidx, bdim, gdim;
  uint8_t* dyn = nullptr;
  size_t dyn_cap = 0;
  void* main_sp = nullptr;
  int cur = -1;
  std::function<void()> body;
  std::string error;
};
inline Block& blk() { static thread_local Block* b = new Block(); return *b; }
inline Fiber& cur() { Block& B = blk(); return B.f[B.cur]; }
inline uint8_t* dyn_lds() { return blk().dyn; }

// ---- context switch (x86-64 System V: callee-saved registers on the stack, the stack pointer swapped) -------------------------------------
extern "C" void wavemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl wavemu_switch
.type wavemu_switch,@function
wavemu_switch:
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

inline void yield_to_scheduler() { Block& B = blk(); Fiber& f = B.f[B.cur]; wavemu_switch(&f.sp, B.main_sp); }

extern "C" inline void wavemu_entry() {
  Block& B = blk();
  try { B.body(); } catch (const std::exception& e) { if (B.error.empty()) B.error = e.what(); }
  B.f[B.cur].done = true;
  yield_to_scheduler();
  abort();      // (a finished fiber is never resumed)
}

inline void prepare(Fiber& f) {
  if (!f.stack) {
    void* m = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) throw std::runtime_error("wavemu: mmap of a fiber stack failed");
    f.stack = (char*)m;
  }
  // the frame wavemu_switch pops: six callee-saved registers, then the return address = wavemu_entry; the stack is 16-byte aligned at the
  // entry of a function AFTER its call pushed the return address, i.e. (rsp + 8) % 16 == 0 when wavemu_entry starts
  uint64_t* top = (uint64_t*)(f.stack + STACK_BYTES - 64);
  top = (uint64_t*)((uintptr_t)top & ~(uintptr_t)15);
  *--top = 0;                                   // (alignment pad: wavemu_entry starts with rsp % 16 == 8)
  *--top = (uint64_t)(void*)&wavemu_entry;      // return address of the first switch
  for (int i = 0; i < 6; i++) *--top = 0;       // rbp rbx r12 r13 r14 r15
  f.sp = top;
  f.done = false; f.op = OP_NONE; f.site = nullptr;
}

// ---- the rendezvous -------------------------------------------------------------------------------------------------------------------------
__attribute__((noinline)) inline uint64_t wave_op(uint32_t op, uint64_t a, uint64_t b, uint64_t c, const char* site) {
  Fiber& f = cur();
  f.op = op; f.a = a; f.b = b; f.c = c; f.site = site; f.pc = __builtin_return_address(0);
  yield_to_scheduler();
  return f.result;
}

inline uint64_t dpp_source(uint32_t ctrl, uint32_t lane, bool* valid) {
  // the source lane of a DPP control word for `lane` (only the controls the kernels use)
  *valid = true;
  const uint32_t row = lane & ~15u, l = lane & 15u;
  if (ctrl <= 0xFF) { const uint32_t q = lane & ~3u, sel = (ctrl >> (2 * (lane & 3u))) & 3u; return q + sel; }     // quad_perm
  if (ctrl >= 0x101 && ctrl <= 0x10F) { const uint32_t n = ctrl - 0x100; if (l + n > 15) { *valid = false; return lane; } return lane + n; }   // row_shl
  if (ctrl >= 0x111 && ctrl <= 0x11F) { const uint32_t n = ctrl - 0x110; if (l < n) { *valid = false; return lane; } return lane - n; }        // row_shr
  if (ctrl >= 0x121 && ctrl <= 0x12F) { const uint32_t n = ctrl - 0x120; return row + ((l + 16 - n) & 15u); }                                  // row_ror
  if (ctrl == 0x140) return row + (15u - l);                                              // row_mirror
  if (ctrl == 0x141) return row + ((l & 8u) | (7u - (l & 7u)));                           // row_half_mirror
  if (ctrl == 0x142) { if (lane < 16) { *valid = false; return lane; } return row - 1; }  // row_bcast:15: lane 15 of the row before
  if (ctrl == 0x143) { if (lane < 32) { *valid = false; return lane; } return (lane & ~31u) - 1; }   // row_bcast:31: lane 31 of the half before
  throw std::runtime_error("wavemu: DPP control 0x" + std::to_string(ctrl) + " is not emulated");
}

inline void resolve_wave(Block& B, int w0, int w1) {
  // every live lane of the wavefront [w0, w1) waits at a wave-level operation
  uint32_t op = OP_NONE;
  const char* site = nullptr;
  const void* pc = nullptr;
  unsigned long long live = 0;
  for (int i = w0; i < w1; i++) {
    Fiber& f = B.f[i];
    if (f.done) continue;
    if (op == OP_NONE) { op = f.op; site = f.site; pc = f.pc; }
    else if (f.op != op || f.pc != pc)
    {
      // (the library is built with line tables: `addr2line -i -e libwavemu.so <offset>` names the place in the inlined kernel code)
      Dl_info di;
      char where[160];
      const uintptr_t base = dladdr(pc, &di) ? (uintptr_t)di.dli_fbase : 0;
      snprintf(where, sizeof(where), "; library offsets 0x%zx / 0x%zx, lanes %d / %d", (size_t)((uintptr_t)pc - base), (size_t)((uintptr_t)f.pc - base), __builtin_ctzll(live ? live : 1), i - w0);
      throw std::runtime_error(std::string("wavemu: lanes of one wavefront wait at different cross-lane operations (") + (site ? site : "?") + " / " + (f.site ? f.site : "?") +
                               "): a cross-lane operation under lane-divergent control flow" + where);
    }
    live |= 1ull << (i - w0);
  }
  auto lane_of = [&](int i) -> Fiber& { return B.f[w0 + i]; };
  switch (op) {
    case OP_BALLOT: {
      unsigned long long m = 0;
      for (int i = 0; i < w1 - w0; i++) if ((live >> i) & 1 && lane_of(i).a) m |= 1ull << i;
      for (int i = 0; i < w1 - w0; i++) if ((live >> i) & 1) lane_of(i).result = m;
      break;
    }
    case OP_SHFL: {      // a = value, b = source lane (already reduced to 0 .. 63); an inactive source provides the lane's own value
      uint64_t v[64];
      for (int i = 0; i < w1 - w0; i++) v[i] = lane_of(i).a;
      for (int i = 0; i < w1 - w0; i++) if ((live >> i) & 1) { const uint32_t s = (uint32_t)lane_of(i).b & 63u; lane_of(i).result = ((int)s < w1 - w0 && ((live >> s) & 1)) ? v[s] : v[i]; }
      break;
    }
    case OP_READLANE: {  // a = value, b = lane (wave-uniform)
      uint64_t v[64];
      for (int i = 0; i < w1 - w0; i++) v[i] = lane_of(i).a;
      for (int i = 0; i < w1 - w0; i++) if ((live >> i) & 1) { const uint32_t s = (uint32_t)lane_of(i).b & 63u; lane_of(i).result = (int)s < w1 - w0 ? v[s] : 0; }   // (v_readlane reads the register of an inactive lane as it stands)
      break;
    }
    case OP_READFIRST: {
      const int first = live ? __builtin_ctzll(live) : 0;
      const uint64_t v = lane_of(first).a;
      for (int i = 0; i < w1 - w0; i++) if ((live >> i) & 1) lane_of(i).result = v;
      break;
    }
    case OP_DPP: {       // a = source value, b = old value, c = ctrl | row_mask << 16 | bank_mask << 20 | bound_ctrl << 24
      uint64_t v[64];
      for (int i = 0; i < w1 - w0; i++) v[i] = lane_of(i).a;
      for (int i = 0; i < w1 - w0; i++) {
        if (!((live >> i) & 1)) continue;
        Fiber& f = lane_of(i);
        const uint32_t ctrl = (uint32_t)f.c & 0xFFFFu, row_mask = ((uint32_t)f.c >> 16) & 15u, bank_mask = ((uint32_t)f.c >> 20) & 15u;
        const bool bound = (((uint32_t)f.c >> 24) & 1u) != 0;
        bool valid;
        const uint32_t s = (uint32_t)dpp_source(ctrl, (uint32_t)i, &valid);
        const bool enabled = ((row_mask >> (i >> 4)) & 1u) && ((bank_mask >> ((i >> 2) & 3)) & 1u);
        if (!enabled) f.result = f.b;
        else if (!valid || (int)s >= w1 - w0 || !((live >> s) & 1)) f.result = bound ? 0 : f.b;
        else f.result = v[s];
      }
      break;
    }
    case OP_SYNC:
      break;
    default:
      throw std::runtime_error("wavemu: unknown wave operation");
  }
  for (int i = w0; i < w1; i++) if (!B.f[i].done) B.f[i].op = OP_NONE;
}

inline void run_block(Block& B) {
  for (int i = 0; i < B.n; i++) prepare(B.f[i]);
  for (;;) {
    bool progress = false;
    int alive = 0;
    for (int i = 0; i < B.n; i++) {
      Fiber& f = B.f[i];
      if (f.done) continue;
      alive++;
      if (f.op != OP_NONE) continue;
      B.cur = i;
      wavemu_switch(&B.main_sp, f.sp);
      B.cur = -1;
      progress = true;
      if (!B.error.empty()) throw std::runtime_error(B.error);
    }
    if (!alive) break;
    // wave-level rendezvous
    for (int w0 = 0; w0 < B.n; w0 += 64) {
      const int w1 = w0 + 64 < B.n ? w0 + 64 : B.n;
      int live = 0, waiting = 0;
      for (int i = w0; i < w1; i++) if (!B.f[i].done) { live++; if (B.f[i].op != OP_NONE && B.f[i].op < OP_BLOCK_SYNC) waiting++; }
      if (live && waiting == live) { resolve_wave(B, w0, w1); progress = true; }
    }
    // workgroup barrier
    {
      int live = 0, waiting = 0;
      bool all = true;
      uint64_t count = 0;
      for (int i = 0; i < B.n; i++) if (!B.f[i].done) { live++; if (B.f[i].op >= OP_BLOCK_SYNC) { waiting++; if (B.f[i].op == OP_BLOCK_AND && !B.f[i].a) all = false; if (B.f[i].op == OP_BLOCK_COUNT && B.f[i].a) count++; } }
      if (live && waiting == live) {
        for (int i = 0; i < B.n; i++) if (!B.f[i].done) { B.f[i].result = B.f[i].op == OP_BLOCK_COUNT ? count : all ? 1 : 0; B.f[i].op = OP_NONE; }
        progress = true;
      }
    }
    if (!progress) {
      std::string why = "wavemu: deadlock in workgroup " + std::to_string(B.bidx.x) + ":";
      for (int i = 0; i < B.n && why.size() < 600; i++) if (!B.f[i].done) why += " [" + std::to_string(i) + ": op " + std::to_string(B.f[i].op) + " " + (B.f[i].site ? B.f[i].site : "") + "]";
      throw std::runtime_error(why);
    }
  }
}

// a memory fault inside a fiber: say where the lane last met its wavefront (the kernel reads or writes outside its buffers — on the device such a read
// usually lands in the same page and goes unnoticed), then end the process
inline void on_segv(int, siginfo_t* si, void* uc_) {
  const ucontext_t* uc = (const ucontext_t*)uc_;
  const uintptr_t rip = uc ? (uintptr_t)uc->uc_mcontext.gregs[REG_RIP] : 0;
  Block& B = blk();
  char msg[512];
  const Fiber* f = (B.cur >= 0 && B.cur < B.n) ? &B.f[B.cur] : nullptr;
  Dl_info di;
  const uintptr_t base = (f && f->pc && dladdr(f->pc, &di)) ? (uintptr_t)di.dli_fbase : 0;
  const int n = snprintf(msg, sizeof(msg), "wavemu: memory fault at address %p in workgroup %u, thread %d; the lane's last cross-lane operation: %s (library offset 0x%zx); faulting instruction at library offset 0x%zx\n", si->si_addr, B.bidx.x, B.cur,
                         (f && f->site) ? f->site : "none yet", f && f->pc ? (size_t)((uintptr_t)f->pc - base) : (size_t)0, (size_t)(rip - base));
  if (n > 0) { ssize_t w = write(2, msg, (size_t)n); (void)w; }
  _exit(139);
}
inline void install_fault_handler() {
  static bool done = false;
  if (done) return;
  done = true;
  static char alt[64 * 1024];
  stack_t ss; ss.ss_sp = alt; ss.ss_size = sizeof(alt); ss.ss_flags = 0;
  sigaltstack(&ss, nullptr);
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = on_segv; sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
  sigaction(SIGSEGV, &sa, nullptr);
}
struct dim3_ { uint32_t x, y, z; };
template <class G, class Bk, class F> void launch(G grid, Bk block, size_t shmem, F&& body) {
  install_fault_handler();
  Block& B = blk();
  const uint32_t nt = block.x * (block.y ? block.y : 1) * (block.z ? block.z : 1);
  if (nt == 0 || nt > (uint32_t)MAX_THREADS) throw std::runtime_error("wavemu: workgroup of " + std::to_string(nt) + " threads");
  // Dynamic LDS: the start of an 8 GiB reservation of address space.  The kernels compute some LDS addresses unconditionally for lanes whose result
  // is discarded ("every lane computes position 0", an index out of a descriptor the lane does not own): on the device such a READ returns
  // whatever — here it must not fault, so everything behind the launch's LDS bytes is readable (untouched pages read as zero, nothing is
  // committed) and NOT writable: a stray LDS WRITE still stops the run.
  constexpr size_t LDS_RESERVE = (size_t)8 << 30;
  if (!B.dyn) {
    void* m = mmap(nullptr, LDS_RESERVE, PROT_READ, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) throw std::runtime_error("wavemu: could not reserve address space for the dynamic LDS");
    B.dyn = (uint8_t*)m;
  }
  {
    const size_t page = 4096, want = ((shmem + 256 + page - 1) / page) * page;
    if (want != B.dyn_cap) {
      if (B.dyn_cap > want) mprotect(B.dyn + want, B.dyn_cap - want, PROT_READ);
      mprotect(B.dyn, want, PROT_READ | PROT_WRITE);
      B.dyn_cap = want;
    }
  }
  B.n = (int)nt;
  B.bdim.x = block.x; B.bdim.y = block.y ? block.y : 1; B.bdim.z = 1;
  B.gdim.x = grid.x; B.gdim.y = grid.y ? grid.y : 1; B.gdim.z = 1;
  B.body = body;
  B.error.clear();
  for (uint32_t by = 0; by < B.gdim.y; by++)
    for (uint32_t bx = 0; bx < B.gdim.x; bx++) {
      B.bidx.x = bx; B.bidx.y = by; B.bidx.z = 0;
      for (uint32_t t = 0; t < nt; t++) { B.f[t].tid.x = t % B.bdim.x; B.f[t].tid.y = t / B.bdim.x; B.f[t].tid.z = 0; }
      memset(B.dyn, 0xCD, shmem);           // (LDS is not zero on the device either)
      run_block(B);
    }
}

// ---- the intrinsics ----------------------------------------------------------------------------------------------------------------------------
#define WAVEMU_STR2(x) #x
#define WAVEMU_STR(x) WAVEMU_STR2(x)
#define WAVEMU_SITE __FILE__ ":" WAVEMU_STR(__LINE__)
inline uint32_t lane_id() { const Fiber& f = cur(); const Block& B = blk(); return (f.tid.x + f.tid.y * B.bdim.x) & 63u; }
__attribute__((always_inline)) inline unsigned long long ballot(bool p, const char* site) { return wave_op(OP_BALLOT, p ? 1 : 0, 0, 0, site); }
__attribute__((always_inline)) inline uint64_t shfl64(uint64_t v, uint32_t src, const char* site) { return wave_op(OP_SHFL, v, src & 63u, 0, site); }
__attribute__((always_inline)) inline uint32_t readlane(uint32_t v, uint32_t l, const char* site) { return (uint32_t)wave_op(OP_READLANE, v, l, 0, site); }
__attribute__((always_inline)) inline uint32_t readfirstlane(uint32_t v, const char* site) { return (uint32_t)wave_op(OP_READFIRST, v, 0, 0, site); }
__attribute__((always_inline)) inline uint32_t update_dpp(uint32_t old, uint32_t src, uint32_t ctrl, uint32_t row_mask, uint32_t bank_mask, bool bound, const char* site) {
  return (uint32_t)wave_op(OP_DPP, src, old, ctrl | (row_mask << 16) | (bank_mask << 20) | ((bound ? 1u : 0u) << 24), site);
}
__attribute__((always_inline)) inline void wave_barrier(const char* site) { (void)wave_op(OP_SYNC, 0, 0, 0, site); }
__attribute__((always_inline)) inline void syncthreads(const char* site) { (void)wave_op(OP_BLOCK_SYNC, 0, 0, 0, site); }
__attribute__((always_inline)) inline int syncthreads_and(int p, const char* site) { return (int)wave_op(OP_BLOCK_AND, p ? 1 : 0, 0, 0, site); }
__attribute__((always_inline)) inline int syncthreads_count(int p, const char* site) { return (int)wave_op(OP_BLOCK_COUNT, p ? 1 : 0, 0, 0, site); }
__attribute__((always_inline)) inline unsigned long long uicmp(uint32_t a, uint32_t b, int cond, const char* site) {
  bool r;
  switch (cond) {   // LLVM ICmp predicates
    case 32: r = a == b; break; case 33: r = a != b; break; case 34: r = a > b; break; case 35: r = a >= b; break; case 36: r = a < b; break; case 37: r = a <= b; break;
    case 38: r = (int32_t)a > (int32_t)b; break; case 39: r = (int32_t)a >= (int32_t)b; break; case 40: r = (int32_t)a < (int32_t)b; break; case 41: r = (int32_t)a <= (int32_t)b; break;
    default: throw std::runtime_error("wavemu: icmp predicate");
  }
  return ballot(r, site);
}
inline uint32_t ubfe(uint32_t v, uint32_t off, uint32_t width) { off &= 31u; width &= 31u; return width ? (v >> off) & ((1u << width) - 1u) : 0u; }
inline uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8u * (sh & 3u))); }
inline uint32_t perm(uint32_t a, uint32_t b, uint32_t sel) {
  const uint64_t src = ((uint64_t)a << 32) | b;
  uint32_t r = 0;
  for (int i = 0; i < 4; i++) {
    const uint32_t s = (sel >> (8 * i)) & 0xFFu;
    uint32_t byte;
    if (s < 8u) byte = (uint32_t)(src >> (8 * s)) & 0xFFu;
    else if (s == 0x0C) byte = 0u;
    else if (s >= 0x0D) byte = 0xFFu;
    else { const uint32_t w = (s == 8u) ? b & 0xFFFFu : (s == 9u) ? b >> 16 : (s == 10u) ? a & 0xFFFFu : a >> 16; byte = (w & 0x8000u) ? 0xFFu : 0u; }   // (8 - 11: sign bytes)
    r |= byte << (8 * i);
  }
  return r;
}
inline uint32_t mbcnt_lo(uint32_t mask, uint32_t acc) { const uint32_t l = lane_id(); const uint32_t m = l >= 32 ? 0xFFFFFFFFu : ((1u << l) - 1u); return acc + (uint32_t)__builtin_popcount(mask & m); }
inline uint32_t mbcnt_hi(uint32_t mask, uint32_t acc) { const uint32_t l = lane_id(); const uint32_t m = l <= 32 ? 0u : ((1u << (l - 32)) - 1u); return acc + (uint32_t)__builtin_popcount(mask & m); }

}  // namespace wavemu
