// K4: trajectory reference lookup on a device-resident table.
//   oly_traj_reset  Trajectory.reset_trajectory     utils/trajectory.py:289-323
//   oly_traj_next   Trajectory.get_next_sample      utils/trajectory.py:389-401
//   oly_traj_euler  play_trajectory_from_velocity   loco_env_base.py:515-519
// The reference keeps [key][traj][step]; the device copy is re-laid out at upload time as
// [traj][step][key] so that one sample is ONE contiguous row (n_keys * 8 B): a lookup is a
// row copy, n_keys consecutive lanes per env, coalesced on both sides.  Bound: HBM / L2,
// 2 * n_keys * 8 B + cursor per env.
#include <vector>

#include "oly_common.h"

namespace {
constexpr int THREADS = 256;

__global__ __launch_bounds__(THREADS) void traj_reset_kernel(TrajDev tj, int N,
                                                             const int* __restrict__ traj_no,
                                                             const int* __restrict__ step,
                                                             int* __restrict__ cur_traj,
                                                             int* __restrict__ cur_step,
                                                             double* __restrict__ origin,
                                                             double* __restrict__ sample) {
  const long total = (long)N * tj.n_keys;
  const long stride = (long)gridDim.x * THREADS;
  for (long e = (long)blockIdx.x * THREADS + threadIdx.x; e < total; e += stride) {
    const int n = (int)(e / tj.n_keys), k = (int)(e - (long)n * tj.n_keys);
    int j = traj_no[n], s = step[n];
    // out-of-range indices are clamped (the host entry point cannot validate device data)
    j = min(max(j, 0), tj.n_traj - 1);
    s = min(max(s, 0), tj.len - 1);
    const double* row = tj.rows + ((size_t)j * tj.len + s) * tj.n_keys;
    double v = row[k];
    if (k == 0) v -= row[0];
    if (k == 1) v -= row[1];
    sample[e] = v;
    if (k == 0) { cur_traj[n] = j; cur_step[n] = s; origin[2 * n] = row[0]; origin[2 * n + 1] = row[1]; }
  }
}

// One launch must see a consistent cursor: every lane of env n reads cur_step[n] (old value),
// lane k == 0 publishes the increment to step_out (a separate buffer when called through the
// C entry point would need a second pass; instead the increment is written by a follow-up
// kernel on the same stream).
__global__ __launch_bounds__(THREADS) void traj_next_kernel(TrajDev tj, int N,
                                                            const uint8_t* __restrict__ active,
                                                            const int* __restrict__ cur_traj,
                                                            const int* __restrict__ cur_step,
                                                            const double* __restrict__ origin,
                                                            double* __restrict__ sample) {
  const long total = (long)N * tj.n_keys;
  const long stride = (long)gridDim.x * THREADS;
  for (long e = (long)blockIdx.x * THREADS + threadIdx.x; e < total; e += stride) {
    const int n = (int)(e / tj.n_keys), k = (int)(e - (long)n * tj.n_keys);
    if (active && !active[n]) continue;
    const int s = cur_step[n] + 1;
    if (s >= tj.len || s < 0) continue;  // sample = None: row left untouched
    const int j = min(max(cur_traj[n], 0), tj.n_traj - 1);
    double v = tj.rows[((size_t)j * tj.len + s) * tj.n_keys + k];
    if (k == 0) v -= origin[2 * n];
    if (k == 1) v -= origin[2 * n + 1];
    sample[e] = v;
  }
}

__global__ __launch_bounds__(THREADS) void traj_advance_kernel(int len, int N,
                                                               const uint8_t* __restrict__ active,
                                                               int* __restrict__ cur_step,
                                                               uint8_t* __restrict__ at_end) {
  const int n = blockIdx.x * THREADS + threadIdx.x;
  if (n >= N) return;
  if (active && !active[n]) { at_end[n] = 0; return; }
  int s = cur_step[n];
  if (s < len) s += 1;  // saturates at len (the reference's callers reset right after a None)
  cur_step[n] = s;
  at_end[n] = (s >= len) ? 1 : 0;
}

__global__ __launch_bounds__(THREADS) void traj_euler_kernel(int n_keys, int N, int n_qpos, double dt,
                                                             const double* __restrict__ curr_qpos,
                                                             double* __restrict__ sample) {
  const long total = (long)N * n_qpos;
  const long stride = (long)gridDim.x * THREADS;
  for (long e = (long)blockIdx.x * THREADS + threadIdx.x; e < total; e += stride) {
    const int n = (int)(e / n_qpos), j = (int)(e - (long)n * n_qpos);
    double* s = sample + (size_t)n * n_keys;
    s[j] = curr_qpos[e] + dt * s[n_qpos + j];
  }
}

inline int blocks_for(long n) {
  long b = (n + THREADS - 1) / THREADS;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// Expert data for the discriminator (GAIL._fit_discriminator, gail_TRPO.py:176-202): row m of
// create_dataset's `states` is flattened sample m of the table restricted to the kept keys and row m
// of `next_states` is sample m + 1 (utils/trajectory.py:158-175), so the device-resident table IS the
// dataset: a minibatch is a row gather by caller-supplied indices, narrowed to float32 like the
// reference's .astype(np.float32).  One lane per output element, n_cols consecutive lanes per row.
__global__ __launch_bounds__(THREADS) void expert_gather_kernel(TrajDev tj, long B, const long* __restrict__ idx,
                                                                int n_cols, const int* __restrict__ cols,
                                                                float* __restrict__ out_states,
                                                                float* __restrict__ out_next) {
  const long total = B * n_cols;
  const long stride = (long)gridDim.x * THREADS;
  const long M = (long)tj.n_traj * tj.len;
  for (long e = (long)blockIdx.x * THREADS + threadIdx.x; e < total; e += stride) {
    const long b = e / n_cols;
    const int c = (int)(e - b * n_cols);
    long m = idx[b];
    m = m < 0 ? 0 : (m > M - 2 ? M - 2 : m);      // dataset rows are 0 .. M-2 (clamped: device data cannot raise)
    const int k = cols[c];
    const double* row = tj.rows + (size_t)m * tj.n_keys;
    out_states[e] = (float)row[k];
    if (out_next) out_next[e] = (float)row[tj.n_keys + k];
  }
}

// The whole dataset as arrays (create_dataset's return value), float64 like the reference's.
__global__ __launch_bounds__(THREADS) void expert_dataset_kernel(TrajDev tj, int n_cols, const int* __restrict__ cols,
                                                                 double* __restrict__ states,
                                                                 double* __restrict__ next_states,
                                                                 double* __restrict__ absorbing,
                                                                 double* __restrict__ last) {
  const long M = (long)tj.n_traj * tj.len;
  const long total = (M - 1) * n_cols;
  const long stride = (long)gridDim.x * THREADS;
  for (long e = (long)blockIdx.x * THREADS + threadIdx.x; e < total; e += stride) {
    const long m = e / n_cols;
    const int k = cols[(int)(e - m * n_cols)];
    const double* row = tj.rows + (size_t)m * tj.n_keys;
    states[e] = row[k];
    next_states[e] = row[tj.n_keys + k];
  }
  for (long m = (long)blockIdx.x * THREADS + threadIdx.x; m < M; m += stride) {
    if (m < M - 1 && absorbing) absorbing[m] = 0.0;
    if (last) last[m] = ((m + 1) % tj.len == 0) ? 1.0 : 0.0;      // split_points[1:] - 1 (equal-length trajectories)
  }
}
}  // namespace

extern "C" int oly_traj_upload(oly_ctx* ctx, int n_keys, int n_traj, int len, const double* table_host) {
  if (!ctx) return OLY_EINVAL;
  if (n_keys < 2 || n_traj <= 0 || len <= 0 || !table_host)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_traj_upload: bad shape [%d,%d,%d]", n_keys, n_traj, len);
  OLY_HIP(ctx, hipSetDevice(ctx->device));
  const size_t count = (size_t)n_keys * n_traj * len;
  std::vector<double> rows(count);
  for (int k = 0; k < n_keys; ++k)
    for (int j = 0; j < n_traj; ++j)
      for (int s = 0; s < len; ++s)
        rows[((size_t)j * len + s) * n_keys + k] = table_host[((size_t)k * n_traj + j) * len + s];
  if (ctx->traj.rows) { (void)hipFree(ctx->traj.rows); ctx->traj.rows = nullptr; ctx->traj_ok = false; }
  if (hipMalloc(&ctx->traj.rows, count * sizeof(double)) != hipSuccess)
    OLY_FAIL(ctx, OLY_ENOMEM, "oly_traj_upload: hipMalloc of %zu bytes failed", count * sizeof(double));
  OLY_HIP(ctx, hipMemcpy(ctx->traj.rows, rows.data(), count * sizeof(double), hipMemcpyHostToDevice));
  ctx->traj.n_keys = n_keys; ctx->traj.n_traj = n_traj; ctx->traj.len = len;
  ctx->traj_ok = true;
  return OLY_OK;
}

extern "C" int oly_traj_reset(oly_ctx* ctx, int N, const int32_t* traj_no, const int32_t* step,
                              int32_t* cur_traj, int32_t* cur_step, double* origin, double* sample,
                              oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (!ctx->traj_ok) OLY_FAIL(ctx, OLY_ENOTCONF, "oly_traj_reset before oly_traj_upload");
  if (N < 0 || (N > 0 && (!traj_no || !step || !cur_traj || !cur_step || !origin || !sample)))
    OLY_FAIL(ctx, OLY_EINVAL, "oly_traj_reset: bad argument");
  if (N == 0) return OLY_OK;
  hipLaunchKernelGGL(traj_reset_kernel, dim3(blocks_for((long)N * ctx->traj.n_keys)), dim3(THREADS), 0,
                     oly_s(stream), ctx->traj, N, traj_no, step, cur_traj, cur_step, origin, sample);
  OLY_LAUNCH_CHECK(ctx, "traj_reset_kernel");
  return OLY_OK;
}

extern "C" int oly_traj_next(oly_ctx* ctx, int N, const uint8_t* active, const int32_t* cur_traj,
                             int32_t* cur_step, const double* origin, double* sample, uint8_t* at_end,
                             oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (!ctx->traj_ok) OLY_FAIL(ctx, OLY_ENOTCONF, "oly_traj_next before oly_traj_upload");
  if (N < 0 || (N > 0 && (!cur_traj || !cur_step || !origin || !sample || !at_end)))
    OLY_FAIL(ctx, OLY_EINVAL, "oly_traj_next: bad argument");
  if (N == 0) return OLY_OK;
  hipLaunchKernelGGL(traj_next_kernel, dim3(blocks_for((long)N * ctx->traj.n_keys)), dim3(THREADS), 0,
                     oly_s(stream), ctx->traj, N, active, cur_traj, cur_step, origin, sample);
  hipLaunchKernelGGL(traj_advance_kernel, dim3((N + THREADS - 1) / THREADS), dim3(THREADS), 0,
                     oly_s(stream), ctx->traj.len, N, active, cur_step, at_end);
  OLY_LAUNCH_CHECK(ctx, "traj_next kernels");
  return OLY_OK;
}

extern "C" int oly_traj_euler(oly_ctx* ctx, int N, int n_qpos, double dt, const double* curr_qpos,
                              double* sample, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (!ctx->traj_ok) OLY_FAIL(ctx, OLY_ENOTCONF, "oly_traj_euler before oly_traj_upload");
  if (N < 0 || n_qpos <= 0 || 2 * n_qpos > ctx->traj.n_keys || (N > 0 && (!curr_qpos || !sample)))
    OLY_FAIL(ctx, OLY_EINVAL, "oly_traj_euler: bad argument");
  if (N == 0) return OLY_OK;
  hipLaunchKernelGGL(traj_euler_kernel, dim3(blocks_for((long)N * n_qpos)), dim3(THREADS), 0, oly_s(stream),
                     ctx->traj.n_keys, N, n_qpos, dt, curr_qpos, sample);
  OLY_LAUNCH_CHECK(ctx, "traj_euler_kernel");
  return OLY_OK;
}

extern "C" int64_t oly_expert_rows(oly_ctx* ctx) {
  if (!ctx || !ctx->traj_ok) return -1;
  return (int64_t)ctx->traj.n_traj * ctx->traj.len - 1;
}

extern "C" int oly_expert_gather(oly_ctx* ctx, int64_t B, const int64_t* idx, int n_cols, const int32_t* cols,
                                 float* out_states, float* out_next, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (!ctx->traj_ok) OLY_FAIL(ctx, OLY_ENOTCONF, "oly_expert_gather before oly_traj_upload");
  if (B < 0 || n_cols <= 0 || n_cols > ctx->traj.n_keys || !cols || (B > 0 && (!idx || !out_states)))
    OLY_FAIL(ctx, OLY_EINVAL, "oly_expert_gather: bad argument (B=%ld n_cols=%d)", (long)B, n_cols);
  if ((long)ctx->traj.n_traj * ctx->traj.len < 2) OLY_FAIL(ctx, OLY_ERANGE, "oly_expert_gather: the table has no transition");
  if (B == 0) return OLY_OK;
  long blocks = (B * n_cols + THREADS - 1) / THREADS;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(expert_gather_kernel, dim3((unsigned)blocks), dim3(THREADS), 0, oly_s(stream), ctx->traj, (long)B,
                     reinterpret_cast<const long*>(idx), n_cols, cols, out_states, out_next);
  OLY_LAUNCH_CHECK(ctx, "expert_gather_kernel");
  return OLY_OK;
}

extern "C" int oly_expert_dataset(oly_ctx* ctx, int n_cols, const int32_t* cols, double* states, double* next_states,
                                  double* absorbing, double* last, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (!ctx->traj_ok) OLY_FAIL(ctx, OLY_ENOTCONF, "oly_expert_dataset before oly_traj_upload");
  if (n_cols <= 0 || n_cols > ctx->traj.n_keys || !cols || !states || !next_states)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_expert_dataset: bad argument");
  const long M = (long)ctx->traj.n_traj * ctx->traj.len;
  if (M < 2) OLY_FAIL(ctx, OLY_ERANGE, "oly_expert_dataset: the table has no transition");
  long blocks = ((M - 1) * n_cols + THREADS - 1) / THREADS;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(expert_dataset_kernel, dim3((unsigned)blocks), dim3(THREADS), 0, oly_s(stream), ctx->traj, n_cols,
                     cols, states, next_states, absorbing, last);
  OLY_LAUNCH_CHECK(ctx, "expert_dataset_kernel");
  return OLY_OK;
}
