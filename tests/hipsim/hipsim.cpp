// TEST INFRASTRUCTURE - scheduler of the hipsim CPU interpreter (see hip/hip_runtime.h).
#include <setjmp.h>
#include <ucontext.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hip/hip_runtime.h"

namespace hipsim {

Dim3 threadIdx_, blockIdx_, blockDim_, gridDim_;
alignas(16) static unsigned char g_lds[160 * 1024];
unsigned char *dyn_lds() { return g_lds; }

namespace {
enum State { READY, WAIT_BLOCK, WAIT_WAVE, DONE };
enum WaveOp { OP_NONE, OP_SHFL_XOR, OP_SHFL_REL, OP_MFMA, OP_MFMA_H, OP_GLDS, OP_GLDS4, OP_GLDS_MASKED };

struct Fiber {
  ucontext_t ctx;   // first entry only (makecontext: a fresh stack)
  jmp_buf jb;       // every later switch: _setjmp / _longjmp (swapcontext costs a sigprocmask system call per switch -
  bool started = false;  // a third of the interpreter's run time)
  std::vector<unsigned char> stack;
  State state = READY;
  Dim3 tid;
  // wave-op operands / results
  WaveOp op = OP_NONE;
  float a = 0, b = 0;
  float ha[8], hb[8];  // f16 MFMA operands (8 consecutive k per lane)
  int imm = 0;
  const float *gsrc = nullptr;
  float *ldst = nullptr;
  f32x16 c, d;
  float fres = 0;
};

std::vector<Fiber> g_f;
ucontext_t g_sched;
jmp_buf g_sched_jb;
int g_cur = -1;
const std::function<void()> *g_body = nullptr;

void trampoline() {
  (*g_body)();
  g_f[g_cur].state = DONE;
  _longjmp(g_sched_jb, 1);
}

void yield_to_sched() {
  if (_setjmp(g_f[g_cur].jb) == 0) _longjmp(g_sched_jb, 1);
}

// run fiber i until it yields or finishes (in a function of its own: the frame the fibers jump back into holds no
// scheduler loop state)
__attribute__((noinline)) void switch_to(int i) {
  if (_setjmp(g_sched_jb) == 0) {
    if (g_f[i].started) _longjmp(g_f[i].jb, 1);
    g_f[i].started = true;
    setcontext(&g_f[i].ctx);  // onto the fiber's fresh stack; control returns through g_sched_jb
  }
}

void resolve_wave(int w0, int w1) {
  // all live lanes of the wave [w0, w1) are waiting: perform the op
  WaveOp op = OP_NONE;
  for (int i = w0; i < w1; ++i)
    if (g_f[i].state == WAIT_WAVE) {
      if (op == OP_NONE) op = g_f[i].op;
      else if (op != g_f[i].op) { fprintf(stderr, "hipsim: divergent wave ops\n"); abort(); }
    }
  if (op == OP_SHFL_XOR) {
    for (int i = w0; i < w1; ++i) {
      if (g_f[i].state != WAIT_WAVE) continue;
      int src = w0 + (((i - w0) ^ g_f[i].imm) & 63);
      g_f[i].fres = (src < w1 && g_f[src].state == WAIT_WAVE) ? g_f[src].a : 0.0f;
    }
  } else if (op == OP_SHFL_REL) {
    for (int i = w0; i < w1; ++i) {
      if (g_f[i].state != WAIT_WAVE) continue;
      const int src = i + g_f[i].imm;
      const bool ok = src >= w0 && src < w0 + 64 && src < w1 && g_f[src].state == WAIT_WAVE;
      g_f[i].fres = ok ? g_f[src].a : g_f[i].a;
    }
  } else if (op == OP_GLDS) {
    // LDS destination = base of the first active lane + lane*16 bytes (M0 semantics)
    float *base = nullptr;
    for (int i = w0; i < w1 && !base; ++i)
      if (g_f[i].state == WAIT_WAVE) base = g_f[i].ldst;
    for (int i = w0; i < w1; ++i)
      if (g_f[i].state == WAIT_WAVE) memcpy(base + 4 * (i - w0), g_f[i].gsrc, 16);
  } else if (op == OP_GLDS_MASKED) {
    // every lane arrives; `b` != 0 marks the lanes enabled in EXEC, imm = bytes per lane.
    // The LDS base (M0) is taken from the first enabled lane.
    float *base = nullptr;
    for (int i = w0; i < w1 && !base; ++i)
      if (g_f[i].state == WAIT_WAVE && g_f[i].b != 0.0f) base = g_f[i].ldst;
    for (int i = w0; i < w1; ++i)
      if (g_f[i].state == WAIT_WAVE && g_f[i].b != 0.0f)
        memcpy(reinterpret_cast<char *>(base) + (size_t)g_f[i].imm * (i - w0), g_f[i].gsrc, g_f[i].imm);
  } else if (op == OP_GLDS4) {
    float *base = nullptr;
    for (int i = w0; i < w1 && !base; ++i)
      if (g_f[i].state == WAIT_WAVE) base = g_f[i].ldst;
    for (int i = w0; i < w1; ++i)
      if (g_f[i].state == WAIT_WAVE) base[i - w0] = *g_f[i].gsrc;
  } else if (op == OP_MFMA) {
    if (w1 - w0 != 64) { fprintf(stderr, "hipsim: MFMA needs a full wave\n"); abort(); }
    for (int i = w0; i < w1; ++i)
      if (g_f[i].state != WAIT_WAVE) { fprintf(stderr, "hipsim: MFMA with exited lanes\n"); abort(); }
    for (int l = 0; l < 64; ++l) {
      Fiber &f = g_f[w0 + l];
      const int col = l & 31;
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = f.c[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(g_f[w0 + row + 32 * k].a, g_f[w0 + col + 32 * k].b, acc);
        f.d[r] = acc;
      }
    }
  } else if (op == OP_MFMA_H) {
    // v_mfma_f32_32x32x16_f16: lane l holds A[i = l&31][k = 8*(l>>5) + e], B[k][j = l&31]
    // (layout checked on MI355X with tools/probes/mfma_f16_layout.hip); D as the fp32 MFMA
    if (w1 - w0 != 64) { fprintf(stderr, "hipsim: MFMA needs a full wave\n"); abort(); }
    for (int i = w0; i < w1; ++i)
      if (g_f[i].state != WAIT_WAVE) { fprintf(stderr, "hipsim: MFMA with exited lanes\n"); abort(); }
    for (int l = 0; l < 64; ++l) {
      Fiber &f = g_f[w0 + l];
      const int col = l & 31;
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = f.c[r];
        for (int k = 0; k < 16; ++k)
          acc = fmaf(g_f[w0 + row + 32 * (k >> 3)].ha[k & 7], g_f[w0 + col + 32 * (k >> 3)].hb[k & 7], acc);
        f.d[r] = acc;
      }
    }
  }
  for (int i = w0; i < w1; ++i)
    if (g_f[i].state == WAIT_WAVE) g_f[i].state = READY;
}

void run_block(int nthreads) {
  for (int i = 0; i < nthreads; ++i) {
    Fiber &f = g_f[i];
    f.state = READY;
    f.op = OP_NONE;
    f.started = false;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack.data();
    f.ctx.uc_stack.ss_size = f.stack.size();
    f.ctx.uc_link = &g_sched;
    makecontext(&f.ctx, trampoline, 0);
  }
  const int nwaves = (nthreads + 63) / 64;
  for (;;) {
    bool progressed = false;
    int done = 0;
    for (int i = 0; i < nthreads; ++i) {
      if (g_f[i].state == DONE) { ++done; continue; }
      if (g_f[i].state != READY) continue;
      g_cur = i;
      threadIdx_ = g_f[i].tid;
      switch_to(i);
      progressed = true;
    }
    if (done == nthreads) return;
    // wave rendezvous
    for (int w = 0; w < nwaves; ++w) {
      const int w0 = w * 64, w1 = std::min(nthreads, w0 + 64);
      int waiting = 0, live = 0;
      for (int i = w0; i < w1; ++i) {
        if (g_f[i].state != DONE) ++live;
        if (g_f[i].state == WAIT_WAVE) ++waiting;
      }
      if (live > 0 && waiting == live) { resolve_wave(w0, w1); progressed = true; }
    }
    // block barrier
    int waiting = 0, live = 0;
    for (int i = 0; i < nthreads; ++i) {
      if (g_f[i].state != DONE) ++live;
      if (g_f[i].state == WAIT_BLOCK) ++waiting;
    }
    if (live > 0 && waiting == live) {
      for (int i = 0; i < nthreads; ++i)
        if (g_f[i].state == WAIT_BLOCK) g_f[i].state = READY;
      progressed = true;
    }
    if (!progressed) { fprintf(stderr, "hipsim: deadlock (mixed barrier / wave waits)\n"); abort(); }
  }
}
}  // namespace

void syncthreads() {
  g_f[g_cur].state = WAIT_BLOCK;
  yield_to_sched();
}

float shfl_xor(float v, int mask) {
  Fiber &f = g_f[g_cur];
  f.op = OP_SHFL_XOR; f.a = v; f.imm = mask; f.state = WAIT_WAVE;
  yield_to_sched();
  return g_f[g_cur].fres;
}

float shfl_rel(float v, int delta) {
  Fiber &f = g_f[g_cur];
  f.op = OP_SHFL_REL; f.a = v; f.imm = delta; f.state = WAIT_WAVE;
  yield_to_sched();
  return g_f[g_cur].fres;
}

float shfl_rel0(float v, int delta) {  // DPP wavefront shift: lanes without a source lane read 0
  const int lane = g_cur % 64;
  const float r = shfl_rel(v, delta);
  return (lane + delta < 0 || lane + delta > 63) ? 0.0f : r;
}

f32x16 mfma32x32x2(float a, float b, f32x16 c) {
  Fiber &f = g_f[g_cur];
  f.op = OP_MFMA; f.a = a; f.b = b; f.c = c; f.state = WAIT_WAVE;
  yield_to_sched();
  return g_f[g_cur].d;
}

f32x16 mfma32x32x16h(const float *a8, const float *b8, f32x16 c) {
  Fiber &f = g_f[g_cur];
  f.op = OP_MFMA_H; f.c = c; f.state = WAIT_WAVE;
  for (int k = 0; k < 8; ++k) { f.ha[k] = a8[k]; f.hb[k] = b8[k]; }
  yield_to_sched();
  return g_f[g_cur].d;
}

void glds16(const float *gsrc_lane, float *lds_wave_base) {
  Fiber &f = g_f[g_cur];
  f.op = OP_GLDS; f.gsrc = gsrc_lane; f.ldst = lds_wave_base; f.state = WAIT_WAVE;
  yield_to_sched();
}

void glds4(const float *gsrc_lane, float *lds_wave_base) {
  Fiber &f = g_f[g_cur];
  f.op = OP_GLDS4; f.gsrc = gsrc_lane; f.ldst = lds_wave_base; f.state = WAIT_WAVE;
  yield_to_sched();
}

void glds_masked(bool active, int bytes, const float *gsrc_lane, float *lds_wave_base) {
  Fiber &f = g_f[g_cur];
  f.op = OP_GLDS_MASKED; f.gsrc = gsrc_lane; f.ldst = lds_wave_base; f.b = active ? 1.0f : 0.0f; f.imm = bytes;
  f.state = WAIT_WAVE;
  yield_to_sched();
}

void launch(const std::function<void()> &body, Dim3 grid, Dim3 block, size_t shmem) {
  if (shmem > sizeof(g_lds)) { fprintf(stderr, "hipsim: LDS request %zu too large\n", shmem); abort(); }
  const int nthreads = block.x * block.y * block.z;
  if ((int)g_f.size() < nthreads) {
    g_f.resize(nthreads);
    for (auto &f : g_f)
      if (f.stack.empty()) f.stack.resize(256 * 1024);
  }
  for (int i = 0; i < nthreads; ++i)
    g_f[i].tid = Dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
  g_body = &body;
  blockDim_ = block;
  gridDim_ = grid;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx_ = Dim3(bx, by, bz);
        memset(g_lds, 0xA5, shmem);  // poison: uninitialised LDS reads show up as garbage
        run_block(nthreads);
      }
}

}  // namespace hipsim
