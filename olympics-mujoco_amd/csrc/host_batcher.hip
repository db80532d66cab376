// Host physics batcher: the step BEFORE the hot path.
//
// The reference steps ONE MuJoCo environment per process (mushroom MuJoCo.step ->
// mujoco.mj_step, rl/algos/ppo.py:200-207 fans processes out with ray).  Here one process
// owns N environment slots; a persistent thread pool runs the per-env physics callback over
// contiguous env ranges, reading controls from and writing qpos/qvel rows into PINNED host
// staging laid out exactly as the kernels consume them ([N,nq] / [N,nv] f64 rows), so the
// only data movement per step is  D2H ctrl [N,nu] f64  and  H2D qpos+qvel.
// In a MuJoCo build the callback is `mj_step(model, data[env])` plus two memcpy's; this image
// has no MuJoCo, so the built-in callback is the kinematic stand-in qpos += dt * qvel.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "oly_common.h"

// Persistent worker pool shared by both batchers.  A vec step is short (tens of microseconds of GPU work
// between two physics phases), so a condition-variable wake of every worker per step costs as much as the
// physics stand-in itself: workers therefore poll the generation counter for a bounded time after each
// job (they are still hot when the next step arrives) and only then block; the caller polls the pending
// counter the same way.  Long gaps (the policy update) put the workers to sleep.
struct WorkerPool {
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable cv_go;
  std::atomic<long> generation{0};
  std::atomic<int> pending{0};
  std::atomic<int> sleepers{0};
  std::atomic<bool> stop{false};
  static constexpr int kSpinUs = 200;

  static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
  static void relax() { __builtin_ia32_pause(); }

  template <class Job>
  void start(int n, Job job) {
    for (int id = 0; id < n; ++id)
      threads.emplace_back([this, job, id] {
        long seen = 0;
        for (;;) {
          const double t0 = now();
          int polls = 0;
          while (generation.load(std::memory_order_acquire) == seen && !stop.load(std::memory_order_acquire)) {
            relax();
            if ((++polls & 63) == 0 && (now() - t0) * 1e6 > kSpinUs) {
              std::unique_lock<std::mutex> lk(mu);
              sleepers.fetch_add(1);
              cv_go.wait(lk, [&] { return stop.load() || generation.load() != seen; });
              sleepers.fetch_sub(1);
            }
          }
          if (stop.load(std::memory_order_acquire)) return;
          seen = generation.load(std::memory_order_acquire);
          job(id);
          pending.fetch_sub(1, std::memory_order_release);
        }
      });
  }
  // run one job on every worker and wait for all of them
  void run() {
    pending.store((int)threads.size(), std::memory_order_relaxed);
    generation.fetch_add(1);                       // seq_cst: pairs with the sleeper's increment-then-check
    if (sleepers.load() > 0) {
      std::lock_guard<std::mutex> lk(mu);
      cv_go.notify_all();
    }
    int polls = 0;
    while (pending.load(std::memory_order_acquire) != 0) {
      relax();
      if ((++polls & 1023) == 0) std::this_thread::yield();
    }
  }
  void shutdown() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop.store(true, std::memory_order_release);
    }
    cv_go.notify_all();
    for (auto& t : threads) t.join();
    threads.clear();
  }
};

struct oly_batcher {
  oly_ctx* ctx;
  int N, nq, nv, nu, n_act, n_threads;
  double dt;
  oly_physics_fn fn;
  void* user;
  double *h_qpos, *h_qvel, *h_ctrl;        // pinned host
  double *d_qpos, *d_qvel, *d_ctrl, *d_prev;  // device
  // mapped mode: the kernels address the pinned staging directly (device aliases of h_*), so a step
  // is two launches and one synchronise with no copy commands in between
  int mapped;
  double *m_qpos, *m_qvel, *m_ctrl;
  // use_foot_forces: W substep contact snapshots per env
  int W, C;
  oly_physics_contacts_fn cfn;
  void* cuser;
  unsigned char *h_con, *d_con;            // one slab: ncon [W,N] | geom1 [W,N,C] | geom2 | force6 [W,N,C,6]
  size_t con_bytes, off_g1, off_g2, off_f6;
  double* d_grf;                           // [N, n_grf] window mean
  uint8_t* d_over;                         // [N] overflow bytes of oly_il_ground_forces (the host check below decides)
  std::atomic<int> overflow_env;           // first environment whose contacts overflowed the C staged slots, or -1
  // packed mode: the worker reduces each env's slots to the first-contact force of every sensor pair
  // right after its physics callback (per-thread scratch slots), only [W,N,n_grf] doubles cross PCIe
  int packed;
  double *h_grfw, *d_grfw;                 // pinned / device [W,N,n_grf]
  std::vector<std::vector<unsigned char>> scratch;   // one env's slot slab per worker thread
  // thread pool
  WorkerPool pool;
  double timing[3];
};

namespace {

void kinematic_step(int, const double*, double* qpos, double* qvel, void* user) {
  const oly_batcher* b = static_cast<const oly_batcher*>(user);
  const int n = b->nq < b->nv ? b->nq : b->nv;
  for (int i = 0; i < n; ++i) qpos[i] += b->dt * qvel[i];
}

// One env's W contact snapshots -> per-substep ground-force vector: for every sensor pair the FIRST
// contact between its two collision groups (either geom order), its force[:3]; zeros without one.
// The host twin of il_grf_kernel's filter (UnitreeH1._get_ground_forces, UnitreeH1.py:113-123), including its
// overflow rule: the callback reports the RAW data.ncon; a substep with more contacts than the C slots it
// could fill is still exact when every sensor pair has its first contact among them, otherwise (or with a
// negative count) the environment is recorded in overflow_env and the step fails with OLY_ERANGE.
// out == nullptr: detection only (slot mode, where the device kernel does the reduction).
void pack_env(oly_batcher* b, int e, const oly_il_contacts& oc, double* out_base) {
  const oly_ctx* ctx = b->ctx;
  const int P = ctx->grf.n_pairs, K = 3 * P, C = b->C;
  for (int w = 0; w < b->W; ++w) {
    const int nc_raw = oc.ncon[(size_t)w * oc.ncon_stride];
    if (!out_base && nc_raw >= 0 && nc_raw <= C) continue;
    double* out = out_base ? out_base + ((size_t)w * b->N + e) * K : nullptr;
    if (out) for (int k = 0; k < K; ++k) out[k] = 0.0;
    const int nc = nc_raw < 0 ? 0 : (nc_raw > C ? C : nc_raw);
    const int32_t *g1 = oc.geom1 + (size_t)w * oc.geom_stride, *g2 = oc.geom2 + (size_t)w * oc.geom_stride;
    const double* f6 = oc.force6 + (size_t)w * oc.force_stride;
    bool over = nc_raw < 0;
    for (int p = 0; p < P; ++p) {
      bool found = false;
      for (int i = 0; i < nc; ++i) {
        const int a = g1[i], c2 = g2[i];
        const int ga = (a >= 0 && a < ctx->grf.ngeom) ? ctx->grf_group_host[a] : -1;
        const int gb = (c2 >= 0 && c2 < ctx->grf.ngeom) ? ctx->grf_group_host[c2] : -1;
        if (ga < 0 || gb < 0) continue;
        if ((ga == ctx->grf.pair_a[p] && gb == ctx->grf.pair_b[p]) || (ga == ctx->grf.pair_b[p] && gb == ctx->grf.pair_a[p])) {
          if (out) { const double* f = f6 + (size_t)i * 6; out[3 * p] = f[0]; out[3 * p + 1] = f[1]; out[3 * p + 2] = f[2]; }
          found = true;
          break;
        }
      }
      if (!found && nc_raw > C) over = true;
    }
    if (over) {
      int none = -1;
      b->overflow_env.compare_exchange_strong(none, e);
    }
  }
}

void run_range(oly_batcher* b, int lo, int hi, int id) {
  if (b->cfn && b->packed) {
    const size_t W = (size_t)b->W, C = (size_t)b->C;
    unsigned char* sc = b->scratch[id].data();
    int32_t* ncon = reinterpret_cast<int32_t*>(sc);
    const size_t off_g1 = (sizeof(int32_t) * W + 15) & ~(size_t)15;
    const size_t off_g2 = off_g1 + sizeof(int32_t) * W * C;
    const size_t off_f6 = (off_g2 + sizeof(int32_t) * W * C + 15) & ~(size_t)15;
    for (int e = lo; e < hi; ++e) {
      oly_il_contacts oc;
      oc.W = b->W; oc.C = b->C;
      oc.ncon = ncon; oc.ncon_stride = 1;
      oc.geom1 = reinterpret_cast<int32_t*>(sc + off_g1);
      oc.geom2 = reinterpret_cast<int32_t*>(sc + off_g2);
      oc.geom_stride = (long)C;
      oc.force6 = reinterpret_cast<double*>(sc + off_f6);
      oc.force_stride = (long)(C * 6);
      b->cfn(e, b->h_ctrl + (size_t)e * b->nu, b->h_qpos + (size_t)e * b->nq, b->h_qvel + (size_t)e * b->nv, &oc,
             b->cuser);
      pack_env(b, e, oc, b->h_grfw);
    }
    return;
  }
  if (b->cfn) {
    const size_t N = (size_t)b->N, C = (size_t)b->C;
    for (int e = lo; e < hi; ++e) {
      oly_il_contacts oc;
      oc.W = b->W; oc.C = b->C;
      oc.ncon = reinterpret_cast<int32_t*>(b->h_con) + e; oc.ncon_stride = (long)N;
      oc.geom1 = reinterpret_cast<int32_t*>(b->h_con + b->off_g1) + (size_t)e * C;
      oc.geom2 = reinterpret_cast<int32_t*>(b->h_con + b->off_g2) + (size_t)e * C;
      oc.geom_stride = (long)(N * C);
      oc.force6 = reinterpret_cast<double*>(b->h_con + b->off_f6) + (size_t)e * C * 6;
      oc.force_stride = (long)(N * C * 6);
      b->cfn(e, b->h_ctrl + (size_t)e * b->nu, b->h_qpos + (size_t)e * b->nq, b->h_qvel + (size_t)e * b->nv, &oc,
             b->cuser);
      pack_env(b, e, oc, nullptr);          // overflow detection only: a count compare unless a substep overflowed
    }
    return;
  }
  void* user = b->fn == kinematic_step ? static_cast<void*>(b) : b->user;
  for (int e = lo; e < hi; ++e)
    b->fn(e, b->h_ctrl + (size_t)e * b->nu, b->h_qpos + (size_t)e * b->nq, b->h_qvel + (size_t)e * b->nv, user);
}

void worker_main(oly_batcher* b, int id) {
  const int per = (b->N + b->n_threads - 1) / b->n_threads;
  const int lo = id * per, hi = lo + per < b->N ? lo + per : b->N;
  if (lo < hi) run_range(b, lo, hi, id);
}

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" int oly_batcher_create(oly_batcher** out, oly_ctx* ctx, int N, int n_threads, double dt,
                                  oly_physics_fn physics, void* user) {
  if (!out || !ctx) return OLY_EINVAL;
  *out = nullptr;
  if (!ctx->il_ok) OLY_FAIL(ctx, OLY_ENOTCONF, "oly_batcher_create before oly_il_configure");
  if (N <= 0 || n_threads < 0) OLY_FAIL(ctx, OLY_EINVAL, "oly_batcher_create: bad N or n_threads");
  OLY_HIP(ctx, hipSetDevice(ctx->device));
  oly_batcher* b = new (std::nothrow) oly_batcher();
  if (!b) return OLY_ENOMEM;
  const IlDev& h = ctx->il_host;
  b->ctx = ctx; b->N = N; b->nq = h.nq; b->nv = h.nv; b->nu = h.nu; b->n_act = h.n_act; b->dt = dt;
  b->fn = physics ? physics : kinematic_step;
  b->user = user;
  unsigned hw = std::thread::hardware_concurrency();
  if (n_threads == 0) n_threads = hw ? (int)hw : 1;
  if (n_threads > N) n_threads = N;
  b->n_threads = n_threads;
  b->h_qpos = b->h_qvel = b->h_ctrl = nullptr;
  b->d_qpos = b->d_qvel = b->d_ctrl = b->d_prev = nullptr;
  b->W = b->C = 0; b->cfn = nullptr; b->cuser = nullptr; b->h_con = b->d_con = nullptr; b->d_grf = nullptr;
  b->d_over = nullptr; b->overflow_env.store(-1);
  b->packed = 0; b->h_grfw = b->d_grfw = nullptr;
  b->mapped = 0; b->m_qpos = b->m_qvel = b->m_ctrl = nullptr;
  const size_t sq = sizeof(double) * N * b->nq, sv = sizeof(double) * N * b->nv, sc = sizeof(double) * N * b->nu;
  bool ok = hipHostMalloc(reinterpret_cast<void**>(&b->h_qpos), sq, hipHostMallocDefault) == hipSuccess &&
            hipHostMalloc(reinterpret_cast<void**>(&b->h_qvel), sv, hipHostMallocDefault) == hipSuccess &&
            hipHostMalloc(reinterpret_cast<void**>(&b->h_ctrl), sc, hipHostMallocDefault) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&b->d_qpos), sq) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&b->d_qvel), sv) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&b->d_ctrl), sc) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&b->d_prev), sizeof(double) * N) == hipSuccess;
  if (!ok) {
    oly_batcher_destroy(b);
    OLY_FAIL(ctx, OLY_ENOMEM, "oly_batcher_create: allocation failed (N=%d)", N);
  }
  memset(b->h_qpos, 0, sq); memset(b->h_qvel, 0, sv); memset(b->h_ctrl, 0, sc);
  (void)hipMemset(b->d_prev, 0, sizeof(double) * N);
  b->pool.start(n_threads, [b](int id) { worker_main(b, id); });
  *out = b;
  return OLY_OK;
}

extern "C" void oly_batcher_destroy(oly_batcher* b) {
  if (!b) return;
  b->pool.shutdown();
  if (b->h_qpos) (void)hipHostFree(b->h_qpos);
  if (b->h_qvel) (void)hipHostFree(b->h_qvel);
  if (b->h_ctrl) (void)hipHostFree(b->h_ctrl);
  if (b->d_qpos) (void)hipFree(b->d_qpos);
  if (b->d_qvel) (void)hipFree(b->d_qvel);
  if (b->d_ctrl) (void)hipFree(b->d_ctrl);
  if (b->d_prev) (void)hipFree(b->d_prev);
  if (b->h_con) (void)hipHostFree(b->h_con);
  if (b->d_con) (void)hipFree(b->d_con);
  if (b->d_grf) (void)hipFree(b->d_grf);
  if (b->d_over) (void)hipFree(b->d_over);
  if (b->h_grfw) (void)hipHostFree(b->h_grfw);
  if (b->d_grfw) (void)hipFree(b->d_grfw);
  delete b;
}

extern "C" int oly_batcher_enable_contacts(oly_batcher* b, int W, int C, oly_physics_contacts_fn physics, void* user) {
  if (!b) return OLY_EINVAL;
  oly_ctx* ctx = b->ctx;
  if (W <= 0 || C <= 0 || !physics) OLY_FAIL(ctx, OLY_EINVAL, "oly_batcher_enable_contacts: bad W, C or NULL physics");
  if (!ctx->grf_ok) OLY_FAIL(ctx, OLY_ENOTCONF, "oly_batcher_enable_contacts before oly_grf_configure");
  if (ctx->il_host.n_grf != 3 * ctx->grf.n_pairs)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_batcher_enable_contacts: model n_grf=%d but %d sensor pairs", ctx->il_host.n_grf,
             ctx->grf.n_pairs);
  if (b->h_con) OLY_FAIL(ctx, OLY_EINVAL, "oly_batcher_enable_contacts: already enabled");
  OLY_HIP(ctx, hipSetDevice(ctx->device));
  const size_t N = (size_t)b->N;
  auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
  b->off_g1 = al(sizeof(int32_t) * W * N);
  b->off_g2 = al(b->off_g1 + sizeof(int32_t) * W * N * C);
  b->off_f6 = al(b->off_g2 + sizeof(int32_t) * W * N * C);
  b->con_bytes = al(b->off_f6 + sizeof(double) * W * N * C * 6);
  if (hipHostMalloc(reinterpret_cast<void**>(&b->h_con), b->con_bytes, hipHostMallocDefault) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&b->d_con), b->con_bytes) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&b->d_grf), sizeof(double) * N * ctx->il_host.n_grf) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&b->d_over), N) != hipSuccess)
    OLY_FAIL(ctx, OLY_ENOMEM, "oly_batcher_enable_contacts: allocation failed");
  memset(b->h_con, 0, b->con_bytes);
  b->W = W; b->C = C; b->cuser = user;
  b->cfn = physics;   // workers read cfn under the step's generation hand-shake only
  return OLY_OK;
}

extern "C" int oly_batcher_enable_contacts_packed(oly_batcher* b, int W, int C, oly_physics_contacts_fn physics,
                                                  void* user) {
  if (!b) return OLY_EINVAL;
  oly_ctx* ctx = b->ctx;
  if (W <= 0 || C <= 0 || !physics) OLY_FAIL(ctx, OLY_EINVAL, "oly_batcher_enable_contacts_packed: bad W, C or NULL physics");
  if (!ctx->grf_ok || !ctx->grf_group_host) OLY_FAIL(ctx, OLY_ENOTCONF, "oly_batcher_enable_contacts_packed before oly_grf_configure");
  if (ctx->il_host.n_grf != 3 * ctx->grf.n_pairs)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_batcher_enable_contacts_packed: model n_grf=%d but %d sensor pairs", ctx->il_host.n_grf,
             ctx->grf.n_pairs);
  if (b->cfn) OLY_FAIL(ctx, OLY_EINVAL, "oly_batcher_enable_contacts_packed: contacts already enabled");
  OLY_HIP(ctx, hipSetDevice(ctx->device));
  const size_t N = (size_t)b->N, K = (size_t)ctx->il_host.n_grf;
  if (hipHostMalloc(reinterpret_cast<void**>(&b->h_grfw), sizeof(double) * W * N * K, hipHostMallocDefault) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&b->d_grfw), sizeof(double) * W * N * K) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&b->d_grf), sizeof(double) * N * K) != hipSuccess)
    OLY_FAIL(ctx, OLY_ENOMEM, "oly_batcher_enable_contacts_packed: allocation failed");
  const size_t slab = ((sizeof(int32_t) * W + 15) & ~(size_t)15) + 2 * sizeof(int32_t) * W * C + 16 + sizeof(double) * W * C * 6;
  b->scratch.assign((size_t)(b->n_threads > 0 ? b->n_threads : 1), std::vector<unsigned char>(slab, 0));
  b->W = W; b->C = C; b->cuser = user; b->packed = 1;
  b->cfn = physics;
  return OLY_OK;
}

extern "C" int oly_batcher_set_mapped(oly_batcher* b, int on) {
  if (!b) return OLY_EINVAL;
  oly_ctx* ctx = b->ctx;
  if (on && !b->m_qpos) {
    OLY_HIP(ctx, hipHostGetDevicePointer(reinterpret_cast<void**>(&b->m_qpos), b->h_qpos, 0));
    OLY_HIP(ctx, hipHostGetDevicePointer(reinterpret_cast<void**>(&b->m_qvel), b->h_qvel, 0));
    OLY_HIP(ctx, hipHostGetDevicePointer(reinterpret_cast<void**>(&b->m_ctrl), b->h_ctrl, 0));
  }
  b->mapped = on & 3;   // bit 0: controls down, bit 1: state rows up
  return OLY_OK;
}

extern "C" double* oly_batcher_qpos(oly_batcher* b) { return b ? b->h_qpos : nullptr; }
extern "C" double* oly_batcher_qvel(oly_batcher* b) { return b ? b->h_qvel : nullptr; }
extern "C" double* oly_batcher_prev(oly_batcher* b) { return b ? b->d_prev : nullptr; }

extern "C" int oly_batcher_set_prev(oly_batcher* b, const double* prev_dev, oly_stream stream) {
  if (!b || !prev_dev) return OLY_EINVAL;
  OLY_HIP(b->ctx, hipMemcpyAsync(b->d_prev, prev_dev, sizeof(double) * b->N, hipMemcpyDeviceToDevice, oly_s(stream)));
  return OLY_OK;
}

extern "C" int oly_batcher_last_timing(const oly_batcher* b, double out3[3]) {
  if (!b || !out3) return OLY_EINVAL;
  for (int i = 0; i < 3; ++i) out3[i] = b->timing[i];
  return OLY_OK;
}

extern "C" int oly_batcher_step(oly_batcher* b, const float* action, void* obs, float* reward,
                                uint8_t* absorbing, uint8_t* fall_code, int out_flags, oly_stream stream) {
  if (!b) return OLY_EINVAL;
  oly_ctx* ctx = b->ctx;
  if (!action || !obs || !reward || !absorbing) OLY_FAIL(ctx, OLY_EINVAL, "oly_batcher_step: NULL pointer");
  hipStream_t s = oly_s(stream);
  const double t0 = now_s();
  // (1) controls: device K5 -> pinned host, fp64 (what data.ctrl holds)
  int rc = oly_il_ctrl(ctx, b->N, action, (b->mapped & 1) ? b->m_ctrl : b->d_ctrl, OLY_OUT_CTRL_F64, stream);
  if (rc) return rc;
  if (!(b->mapped & 1))
    OLY_HIP(ctx, hipMemcpyAsync(b->h_ctrl, b->d_ctrl, sizeof(double) * b->N * b->nu, hipMemcpyDeviceToHost, s));
  OLY_HIP(ctx, hipStreamSynchronize(s));
  const double t1 = now_s();
  // (2) physics on the host threads
  b->overflow_env.store(-1);
  b->pool.run();
  const double t2 = now_s();
  // A contact overflow is reported AFTER the step has been completed on the device: the host physics has already
  // advanced, so returning here would leave the device-side state (prev, observations) one step behind it.
  const int oe = b->overflow_env.load();
  // (3) state rows up, post-physics path on the device
  if (!(b->mapped & 2)) {
    OLY_HIP(ctx, hipMemcpyAsync(b->d_qpos, b->h_qpos, sizeof(double) * b->N * b->nq, hipMemcpyHostToDevice, s));
    OLY_HIP(ctx, hipMemcpyAsync(b->d_qvel, b->h_qvel, sizeof(double) * b->N * b->nv, hipMemcpyHostToDevice, s));
  }
  const double* grf = nullptr;
  if (b->cfn && b->packed) {
    const size_t K = (size_t)ctx->il_host.n_grf;
    OLY_HIP(ctx, hipMemcpyAsync(b->d_grfw, b->h_grfw, sizeof(double) * b->W * b->N * K, hipMemcpyHostToDevice, s));
    rc = oly_il_grf_window(ctx, b->W, b->N, (int)K, b->d_grfw, b->d_grf, stream);
    if (rc) return rc;
    grf = b->d_grf;
  } else if (b->cfn) {
    OLY_HIP(ctx, hipMemcpyAsync(b->d_con, b->h_con, b->con_bytes, hipMemcpyHostToDevice, s));
    rc = oly_il_ground_forces(ctx, b->W, b->N, b->C, reinterpret_cast<const int32_t*>(b->d_con),
                              reinterpret_cast<const int32_t*>(b->d_con + b->off_g1),
                              reinterpret_cast<const int32_t*>(b->d_con + b->off_g2),
                              reinterpret_cast<const double*>(b->d_con + b->off_f6), nullptr, b->d_grf, b->d_over, stream);
    if (rc) return rc;
    grf = b->d_grf;
  } else if (ctx->il_host.n_grf > 0) {
    OLY_FAIL(ctx, OLY_ENOTCONF, "oly_batcher_step: the model has foot-force columns; call oly_batcher_enable_contacts");
  }
  rc = oly_il_step(ctx, 1, b->N, (b->mapped & 2) ? b->m_qpos : b->d_qpos, (b->mapped & 2) ? b->m_qvel : b->d_qvel, nullptr, grf,
                   b->d_prev, b->d_prev, obs, reward,
                   absorbing, fall_code, nullptr, out_flags & ~OLY_OUT_CTRL_F64, stream);
  const double t3 = now_s();
  b->timing[0] = t1 - t0; b->timing[1] = t2 - t1; b->timing[2] = t3 - t2;
  if (rc == OLY_OK && oe >= 0)
    OLY_FAIL(ctx, OLY_ERANGE, "oly_batcher_step: environment %d has more contacts than the %d staged slots and a sensor pair "
             "without a contact among them (or a negative count): its foot-force columns of this step are not the "
             "reference's; the step itself was completed (host and device state agree); enable contacts with more slots",
             oe, b->C);
  return rc;
}

// ---------------------------------------------------------------------------------------------
// RL robot (StickFigureA3): same pool, one pinned slab holding every readback array so that the
// whole post-physics state of N envs goes up in ONE copy.
// ---------------------------------------------------------------------------------------------
struct oly_a3_batcher {
  oly_ctx* ctx;
  int N, C, nq, nv, nu, n_threads;
  oly_a3_physics_fn fn;
  void* user;
  unsigned char *h_slab, *d_slab;
  size_t slab_bytes;
  size_t off[16];              // byte offsets of the 16 readback arrays inside the slab
  double *h_target, *d_target;  // [N,nu]
  double* m_target;             // device alias of h_target (mapped mode: no D2H copy command)
  // K3 outputs (device)
  int32_t *d_nr, *d_nl;
  double *d_grf_r, *d_grf_l, *d_minz;
  uint8_t* d_bad;
  // compact mode (oly_a3_batcher_set_compact): after its physics callback a worker repacks the env's row
  // into a second pinned slab that holds only what the kernels read: the base quaternion and angular
  // velocity (7 of the 49 qpos / qvel numbers), the actuator and site rows, and the USED contact slots as
  // 64-byte records in per-thread arenas; a second pool round copies the arenas end to end and writes the
  // per-env record offsets; one H2D copy of (fixed part + used records); oly_contact_reduce_csr + the
  // strided oly_a3_step.  Pure data movement on the host, ~690 instead of 1884 bytes per env at 4 contacts.
  int compact;
  unsigned char *hc_slab, *dc_slab;
  size_t coffs[16];            // byte offsets inside the compact slab (C_* below)
  size_t c_fixed_bytes, c_slab_bytes;
  std::vector<std::vector<oly_contact_record>> arena;   // per worker
  std::vector<int> arena_count, arena_base;
  int* loc;                    // [N] record offset of env e inside its worker's arena
  int job_pack_only, job_phase;
  WorkerPool pool;
  double timing[3];
};

namespace {

enum { A_QPOS, A_QVEL, A_ALEN, A_AVEL, A_LFP, A_RFP, A_LFV, A_RFV, A_ROOTP, A_ROOTQ, A_HEAD, A_NCON, A_G1, A_G2, A_F6, A_CZ };

void a3_slots(const oly_a3_batcher* b, unsigned char* base, int e, oly_a3_readback* rb) {
  auto d = [&](int k, int w) { return reinterpret_cast<double*>(base + b->off[k]) + (size_t)e * w; };
  auto i = [&](int k, int w) { return reinterpret_cast<int32_t*>(base + b->off[k]) + (size_t)e * w; };
  rb->qpos = d(A_QPOS, b->nq); rb->qvel = d(A_QVEL, b->nv);
  rb->act_len = d(A_ALEN, b->nu); rb->act_vel = d(A_AVEL, b->nu);
  rb->lf_pos = d(A_LFP, 3); rb->rf_pos = d(A_RFP, 3); rb->lf_vel = d(A_LFV, 3); rb->rf_vel = d(A_RFV, 3);
  rb->root_pos = d(A_ROOTP, 3); rb->root_quat = d(A_ROOTQ, 4); rb->head_pos = d(A_HEAD, 3);
  rb->ncon = i(A_NCON, 1); rb->geom1 = i(A_G1, b->C); rb->geom2 = i(A_G2, b->C);
  rb->force6 = d(A_F6, 6 * b->C); rb->cpos_z = d(A_CZ, b->C);
}

void a3_hold_state(int, const double*, const oly_a3_readback*, void*) {}

enum { C_BQ, C_AV, C_ALEN, C_AVEL, C_LFP, C_RFP, C_LFV, C_RFV, C_ROOTP, C_ROOTQ, C_HEAD, C_NCON, C_COFF, C_REC };

template <class T>
T* c_ptr(const oly_a3_batcher* b, unsigned char* base, int k) { return reinterpret_cast<T*>(base + b->coffs[k]); }

// env e: full pinned row -> compact row + its used contact slots appended to the worker's arena
void a3_pack_env(oly_a3_batcher* b, int id, int e, const oly_a3_readback& rb, int& cnt) {
  unsigned char* h = b->hc_slab;
  const int nu = b->nu;
  auto cp = [&](int k, const double* src, int w) { memcpy(c_ptr<double>(b, h, k) + (size_t)e * w, src, sizeof(double) * w); };
  cp(C_BQ, rb.qpos + 3, 4);
  cp(C_AV, rb.qvel + 3, 3);
  cp(C_ALEN, rb.act_len, nu); cp(C_AVEL, rb.act_vel, nu);
  cp(C_LFP, rb.lf_pos, 3); cp(C_RFP, rb.rf_pos, 3); cp(C_LFV, rb.lf_vel, 3); cp(C_RFV, rb.rf_vel, 3);
  cp(C_ROOTP, rb.root_pos, 3); cp(C_ROOTQ, rb.root_quat, 4); cp(C_HEAD, rb.head_pos, 3);
  const int nc_raw = rb.ncon[0];
  const int nc = nc_raw < 0 ? 0 : (nc_raw > b->C ? b->C : nc_raw);
  c_ptr<int32_t>(b, h, C_NCON)[e] = nc_raw;          // the raw count: more than C marks the env bad on the device
  b->loc[e] = cnt;            // cnt: the worker's own running count (a shared counter array would false-share)
  oly_contact_record* out = b->arena[id].data() + cnt;
  for (int i = 0; i < nc; ++i) {
    out[i].geom1 = rb.geom1[i];
    out[i].geom2 = rb.geom2[i];
    memcpy(out[i].force6, rb.force6 + 6 * i, sizeof(double) * 6);
    out[i].pos_z = rb.cpos_z[i];
  }
  cnt += nc;
}

void a3_worker(oly_a3_batcher* b, int id) {
  const int per = (b->N + b->n_threads - 1) / b->n_threads;
  const int lo = id * per, hi = lo + per < b->N ? lo + per : b->N;
  if (b->compact && b->job_phase == 1) {             // second round: arenas end to end, per-env offsets
    const int base = b->arena_base[id];
    memcpy(c_ptr<oly_contact_record>(b, b->hc_slab, C_REC) + base, b->arena[id].data(),
           sizeof(oly_contact_record) * (size_t)b->arena_count[id]);
    int32_t* coff = c_ptr<int32_t>(b, b->hc_slab, C_COFF);
    for (int e = lo; e < hi; ++e) coff[e] = base + b->loc[e];
    return;
  }
  int cnt = 0;
  for (int e = lo; e < hi; ++e) {
    oly_a3_readback rb;
    a3_slots(b, b->h_slab, e, &rb);
    if (!b->job_pack_only) b->fn(e, b->h_target + (size_t)e * b->nu, &rb, b->user);
    if (b->compact) a3_pack_env(b, id, e, rb, cnt);
  }
  if (b->compact) b->arena_count[id] = cnt;
}

}  // namespace

extern "C" int oly_a3_batcher_create(oly_a3_batcher** out, oly_ctx* ctx, int N, int C, int n_threads,
                                     oly_a3_physics_fn physics, void* user) {
  if (!out || !ctx) return OLY_EINVAL;
  *out = nullptr;
  if (!ctx->a3_ok) OLY_FAIL(ctx, OLY_ENOTCONF, "oly_a3_batcher_create before oly_a3_configure");
  if (!ctx->contact_ok) OLY_FAIL(ctx, OLY_ENOTCONF, "oly_a3_batcher_create before oly_contact_configure");
  if (N <= 0 || C <= 0 || n_threads < 0)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_a3_batcher_create: bad N, C or n_threads");
  OLY_HIP(ctx, hipSetDevice(ctx->device));
  oly_a3_batcher* b = new (std::nothrow) oly_a3_batcher();
  if (!b) return OLY_ENOMEM;
  const A3Dev& h = ctx->a3_host;
  b->ctx = ctx; b->N = N; b->C = C; b->nq = h.nq; b->nv = h.nv; b->nu = h.nu; b->user = user;
  b->fn = physics ? physics : a3_hold_state;   // NULL: the staging is left as written (framework-overhead runs)
  unsigned hw = std::thread::hardware_concurrency();
  if (n_threads == 0) n_threads = hw ? (int)hw : 1;
  if (n_threads > N) n_threads = N;
  b->n_threads = n_threads;
  const size_t w8[11] = {(size_t)h.nq, (size_t)h.nv, (size_t)h.nu, (size_t)h.nu, 3, 3, 3, 3, 3, 4, 3};
  size_t o = 0;
  auto put = [&](int k, size_t bytes) { b->off[k] = o; o = (o + bytes + 15) & ~(size_t)15; };
  for (int k = 0; k < 11; ++k) put(k, sizeof(double) * N * w8[k]);
  put(A_NCON, sizeof(int32_t) * N);
  put(A_G1, sizeof(int32_t) * N * C);
  put(A_G2, sizeof(int32_t) * N * C);
  put(A_F6, sizeof(double) * N * C * 6);
  put(A_CZ, sizeof(double) * N * C);
  b->slab_bytes = o;
  b->h_slab = b->d_slab = nullptr; b->h_target = b->d_target = nullptr; b->m_target = nullptr;
  b->compact = 0; b->hc_slab = b->dc_slab = nullptr; b->loc = nullptr; b->job_pack_only = 0; b->job_phase = 0;
  b->d_nr = b->d_nl = nullptr; b->d_grf_r = b->d_grf_l = b->d_minz = nullptr; b->d_bad = nullptr;
  bool ok = hipHostMalloc(reinterpret_cast<void**>(&b->h_slab), o, hipHostMallocDefault) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&b->d_slab), o) == hipSuccess &&
            hipHostMalloc(reinterpret_cast<void**>(&b->h_target), sizeof(double) * N * h.nu, hipHostMallocDefault) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&b->d_target), sizeof(double) * N * h.nu) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&b->d_nr), sizeof(int32_t) * N) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&b->d_nl), sizeof(int32_t) * N) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&b->d_grf_r), sizeof(double) * N) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&b->d_grf_l), sizeof(double) * N) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&b->d_minz), sizeof(double) * N) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&b->d_bad), N) == hipSuccess;
  if (!ok) {
    oly_a3_batcher_destroy(b);
    OLY_FAIL(ctx, OLY_ENOMEM, "oly_a3_batcher_create: allocation failed (N=%d C=%d)", N, C);
  }
  memset(b->h_slab, 0, o);
  memset(b->h_target, 0, sizeof(double) * N * h.nu);
  b->pool.start(n_threads, [b](int id) { a3_worker(b, id); });
  *out = b;
  return OLY_OK;
}

extern "C" void oly_a3_batcher_destroy(oly_a3_batcher* b) {
  if (!b) return;
  b->pool.shutdown();
  if (b->h_slab) (void)hipHostFree(b->h_slab);
  if (b->d_slab) (void)hipFree(b->d_slab);
  if (b->h_target) (void)hipHostFree(b->h_target);
  if (b->d_target) (void)hipFree(b->d_target);
  if (b->hc_slab) (void)hipHostFree(b->hc_slab);
  if (b->dc_slab) (void)hipFree(b->dc_slab);
  free(b->loc);
  for (void* p : {(void*)b->d_nr, (void*)b->d_nl, (void*)b->d_grf_r, (void*)b->d_grf_l, (void*)b->d_minz, (void*)b->d_bad})
    if (p) (void)hipFree(p);
  delete b;
}

extern "C" int oly_a3_batcher_slots(oly_a3_batcher* b, int env, oly_a3_readback* rb) {
  if (!b || !rb || env < 0 || env >= b->N) return OLY_EINVAL;
  a3_slots(b, b->h_slab, env, rb);
  return OLY_OK;
}

extern "C" int oly_a3_batcher_upload(oly_a3_batcher* b, oly_stream stream) {
  if (!b) return OLY_EINVAL;
  OLY_HIP(b->ctx, hipMemcpyAsync(b->d_slab, b->h_slab, b->slab_bytes, hipMemcpyHostToDevice, oly_s(stream)));
  return OLY_OK;
}

extern "C" int oly_a3_batcher_set_compact(oly_a3_batcher* b, int on) {
  if (!b) return OLY_EINVAL;
  oly_ctx* ctx = b->ctx;
  if (on && !b->hc_slab) {
    const int N = b->N, nu = b->nu;
    const size_t w8[11] = {4, 3, (size_t)nu, (size_t)nu, 3, 3, 3, 3, 3, 4, 3};
    size_t o = 0;
    auto put = [&](int k, size_t bytes) { b->coffs[k] = o; o = (o + bytes + 15) & ~(size_t)15; };
    for (int k = 0; k < 11; ++k) put(k, sizeof(double) * N * w8[k]);
    put(C_NCON, sizeof(int32_t) * N);
    put(C_COFF, sizeof(int32_t) * N);
    b->c_fixed_bytes = o;
    put(C_REC, sizeof(oly_contact_record) * (size_t)N * b->C);
    b->c_slab_bytes = o;
    OLY_HIP(ctx, hipSetDevice(ctx->device));
    if (hipHostMalloc(reinterpret_cast<void**>(&b->hc_slab), o, hipHostMallocDefault) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&b->dc_slab), o) != hipSuccess ||
        !(b->loc = static_cast<int*>(malloc(sizeof(int) * N))))
      OLY_FAIL(ctx, OLY_ENOMEM, "oly_a3_batcher_set_compact: allocation failed");
    memset(b->hc_slab, 0, o);
    const int per = (N + b->n_threads - 1) / b->n_threads;
    b->arena.assign((size_t)b->n_threads, std::vector<oly_contact_record>((size_t)per * b->C));
    b->arena_count.assign((size_t)b->n_threads, 0);
    b->arena_base.assign((size_t)b->n_threads, 0);
  }
  b->compact = on ? 1 : 0;
  return OLY_OK;
}

extern "C" int oly_a3_batcher_set_mapped(oly_a3_batcher* b, int on) {
  if (!b) return OLY_EINVAL;
  b->m_target = nullptr;
  if (on & 1) OLY_HIP(b->ctx, hipHostGetDevicePointer(reinterpret_cast<void**>(&b->m_target), b->h_target, 0));
  return OLY_OK;
}

extern "C" int oly_a3_batcher_last_timing(const oly_a3_batcher* b, double out3[3]) {
  if (!b || !out3) return OLY_EINVAL;
  for (int i = 0; i < 3; ++i) out3[i] = b->timing[i];
  return OLY_OK;
}

extern "C" int oly_a3_batcher_step(oly_a3_batcher* b, const float* action, const oly_a3_state* st, void* obs,
                                   float* rew6, float* reward, uint8_t* done, int out_flags, int with_physics,
                                   oly_stream stream) {
  if (!b) return OLY_EINVAL;
  oly_ctx* ctx = b->ctx;
  if (!st || !obs || !rew6 || !reward || !done || (with_physics && !action))
    OLY_FAIL(ctx, OLY_EINVAL, "oly_a3_batcher_step: NULL pointer");
  hipStream_t s = oly_s(stream);
  const double t0 = now_s();
  double t1 = t0, t2 = t0;
  if (with_physics) {
    int rc = oly_a3_pd_target(ctx, b->N, action, b->m_target ? b->m_target : b->d_target, stream);
    if (rc) return rc;
    if (!b->m_target)
      OLY_HIP(ctx, hipMemcpyAsync(b->h_target, b->d_target, sizeof(double) * b->N * b->nu, hipMemcpyDeviceToHost, s));
    OLY_HIP(ctx, hipStreamSynchronize(s));
    t1 = now_s();
    b->job_pack_only = 0; b->job_phase = 0;
    b->pool.run();
    t2 = now_s();
  } else if (b->compact) {                       // rows written through oly_a3_batcher_slots: pack them
    b->job_pack_only = 1; b->job_phase = 0;
    b->pool.run();
  }
  if (b->compact) {
    int total = 0;
    for (int i = 0; i < b->n_threads; ++i) { b->arena_base[i] = total; total += b->arena_count[i]; }
    b->job_phase = 1;
    b->pool.run();
    b->job_phase = 0;
    t2 = with_physics ? now_s() : t2;
    unsigned char* d = b->dc_slab;
    OLY_HIP(ctx, hipMemcpyAsync(d, b->hc_slab, b->c_fixed_bytes + sizeof(oly_contact_record) * (size_t)total,
                                hipMemcpyHostToDevice, s));
    int rc = oly_contact_reduce_csr(ctx, b->N, b->C, c_ptr<int32_t>(b, d, C_NCON), c_ptr<int32_t>(b, d, C_COFF),
                                    c_ptr<oly_contact_record>(b, d, C_REC), (int64_t)total, b->d_nr, b->d_nl, b->d_grf_r,
                                    b->d_grf_l,
                                    b->d_minz, b->d_bad, stream);
    if (rc) return rc;
    oly_a3_inputs in;
    in.qpos = c_ptr<double>(b, d, C_BQ); in.qvel = c_ptr<double>(b, d, C_AV);
    in.act_len = c_ptr<double>(b, d, C_ALEN); in.act_vel = c_ptr<double>(b, d, C_AVEL);
    in.lf_pos = c_ptr<double>(b, d, C_LFP); in.rf_pos = c_ptr<double>(b, d, C_RFP);
    in.lf_vel = c_ptr<double>(b, d, C_LFV); in.rf_vel = c_ptr<double>(b, d, C_RFV);
    in.root_pos = c_ptr<double>(b, d, C_ROOTP); in.root_quat = c_ptr<double>(b, d, C_ROOTQ);
    in.head_pos = c_ptr<double>(b, d, C_HEAD);
    in.grf_l = b->d_grf_l; in.grf_r = b->d_grf_r; in.min_z = b->d_minz; in.n_r = b->d_nr; in.n_l = b->d_nl;
    in.bad = b->d_bad;
    rc = oly_a3_step_strided(ctx, b->N, &in, st, obs, rew6, reward, done, out_flags, 1, stream);
    const double t3 = now_s();
    b->timing[0] = t1 - t0; b->timing[1] = t2 - t1; b->timing[2] = t3 - t2;
    return rc;
  }
  OLY_HIP(ctx, hipMemcpyAsync(b->d_slab, b->h_slab, b->slab_bytes, hipMemcpyHostToDevice, s));
  oly_a3_readback rb;
  a3_slots(b, b->d_slab, 0, &rb);  // device addresses of the arrays (env 0 = base)
  int rc = oly_contact_reduce(ctx, b->N, b->C, rb.ncon, rb.geom1, rb.geom2, rb.force6, rb.cpos_z, b->d_nr, b->d_nl,
                              nullptr, nullptr, b->d_grf_r, b->d_grf_l, b->d_minz, b->d_bad, stream);
  if (rc) return rc;
  oly_a3_inputs in;
  in.qpos = rb.qpos; in.qvel = rb.qvel; in.act_len = rb.act_len; in.act_vel = rb.act_vel;
  in.lf_pos = rb.lf_pos; in.rf_pos = rb.rf_pos; in.lf_vel = rb.lf_vel; in.rf_vel = rb.rf_vel;
  in.root_pos = rb.root_pos; in.root_quat = rb.root_quat; in.head_pos = rb.head_pos;
  in.grf_l = b->d_grf_l; in.grf_r = b->d_grf_r; in.min_z = b->d_minz; in.n_r = b->d_nr; in.n_l = b->d_nl;
  in.bad = b->d_bad;
  rc = oly_a3_step(ctx, b->N, &in, st, obs, rew6, reward, done, out_flags, stream);
  const double t3 = now_s();
  b->timing[0] = t1 - t0; b->timing[1] = t2 - t1; b->timing[2] = t3 - t2;
  return rc;
}
