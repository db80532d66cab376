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

struct oly_batcher {
  oly_ctx* ctx;
  int N, nq, nv, nu, n_act, n_threads;
  double dt;
  oly_physics_fn fn;
  void* user;
  double *h_qpos, *h_qvel, *h_ctrl;        // pinned host
  double *d_qpos, *d_qvel, *d_ctrl, *d_prev;  // device
  // thread pool
  std::vector<std::thread> workers;
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  long generation;
  int pending;
  bool stop;
  double timing[3];
};

namespace {

void kinematic_step(int, const double*, double* qpos, double* qvel, void* user) {
  const oly_batcher* b = static_cast<const oly_batcher*>(user);
  const int n = b->nq < b->nv ? b->nq : b->nv;
  for (int i = 0; i < n; ++i) qpos[i] += b->dt * qvel[i];
}

void run_range(oly_batcher* b, int lo, int hi) {
  void* user = b->fn == kinematic_step ? static_cast<void*>(b) : b->user;
  for (int e = lo; e < hi; ++e)
    b->fn(e, b->h_ctrl + (size_t)e * b->nu, b->h_qpos + (size_t)e * b->nq, b->h_qvel + (size_t)e * b->nv, user);
}

void worker_main(oly_batcher* b, int id) {
  long seen = 0;
  const int per = (b->N + b->n_threads - 1) / b->n_threads;
  const int lo = id * per, hi = lo + per < b->N ? lo + per : b->N;
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(b->mu);
      b->cv_go.wait(lk, [&] { return b->stop || b->generation != seen; });
      if (b->stop) return;
      seen = b->generation;
    }
    if (lo < hi) run_range(b, lo, hi);
    {
      std::lock_guard<std::mutex> lk(b->mu);
      if (--b->pending == 0) b->cv_done.notify_one();
    }
  }
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
  b->generation = 0; b->pending = 0; b->stop = false;
  b->h_qpos = b->h_qvel = b->h_ctrl = nullptr;
  b->d_qpos = b->d_qvel = b->d_ctrl = b->d_prev = nullptr;
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
  for (int i = 0; i < n_threads; ++i) b->workers.emplace_back(worker_main, b, i);
  *out = b;
  return OLY_OK;
}

extern "C" void oly_batcher_destroy(oly_batcher* b) {
  if (!b) return;
  {
    std::lock_guard<std::mutex> lk(b->mu);
    b->stop = true;
  }
  b->cv_go.notify_all();
  for (auto& t : b->workers) t.join();
  if (b->h_qpos) (void)hipHostFree(b->h_qpos);
  if (b->h_qvel) (void)hipHostFree(b->h_qvel);
  if (b->h_ctrl) (void)hipHostFree(b->h_ctrl);
  if (b->d_qpos) (void)hipFree(b->d_qpos);
  if (b->d_qvel) (void)hipFree(b->d_qvel);
  if (b->d_ctrl) (void)hipFree(b->d_ctrl);
  if (b->d_prev) (void)hipFree(b->d_prev);
  delete b;
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
  int rc = oly_il_ctrl(ctx, b->N, action, b->d_ctrl, OLY_OUT_CTRL_F64, stream);
  if (rc) return rc;
  OLY_HIP(ctx, hipMemcpyAsync(b->h_ctrl, b->d_ctrl, sizeof(double) * b->N * b->nu, hipMemcpyDeviceToHost, s));
  OLY_HIP(ctx, hipStreamSynchronize(s));
  const double t1 = now_s();
  // (2) physics on the host threads
  {
    std::unique_lock<std::mutex> lk(b->mu);
    b->pending = b->n_threads;
    ++b->generation;
    b->cv_go.notify_all();
    b->cv_done.wait(lk, [&] { return b->pending == 0; });
  }
  const double t2 = now_s();
  // (3) state rows up, post-physics path on the device
  OLY_HIP(ctx, hipMemcpyAsync(b->d_qpos, b->h_qpos, sizeof(double) * b->N * b->nq, hipMemcpyHostToDevice, s));
  OLY_HIP(ctx, hipMemcpyAsync(b->d_qvel, b->h_qvel, sizeof(double) * b->N * b->nv, hipMemcpyHostToDevice, s));
  rc = oly_il_step(ctx, 1, b->N, b->d_qpos, b->d_qvel, nullptr, nullptr, b->d_prev, b->d_prev, obs, reward,
                   absorbing, fall_code, nullptr, out_flags & ~OLY_OUT_CTRL_F64, stream);
  const double t3 = now_s();
  b->timing[0] = t1 - t0; b->timing[1] = t2 - t1; b->timing[2] = t3 - t2;
  return rc;
}
