// K3: foot-contact reduction for the RL robots.
//   get_{r,l}foot_floor_contacts   mujoco_robot_interface.py:245-273  (floor must be geom1)
//   get_{r,l}foot_grf              :275-297  (sum of ||force6||_2 in contact order)
//   check_bad_collisions           :393-399
//   contact point of _calc_height_reward   tasks/rewards.py:29-33
//
// Layout: SLOTS = 16 lanes (one DPP row of the 64-wide wave) per environment, one lane per
// contact slot, 4 environments per wave.  Each lane loads its own geom pair and force6 vector:
// a wave reads 4 x (64 + 64 + 768 + 128) contiguous bytes, fully coalesced.  The per-env
// reductions run inside the 16-lane group with wave shuffles: counts and index lists by ballot
// + popcount (bit-exact), the force sum as an IN-ORDER serial chain over the group (same
// rounding as the reference's python loop), the minimum by a shuffle tree.
// Bound: HBM, 4 + C*(4+4+48+8) B read per env (1028 B at C = 16).
#include <cstdlib>

#include "oly_common.h"

namespace {
constexpr int THREADS = 256;
constexpr int SLOTS = 16;

// CSR = false: padded slots, contact i of env n at [n, i] of geom1 / geom2 / force6 / pos_z.
// CSR = true:  compact records (oly_contact_record, 64 B), contact i of env n at rec[coff[n] + i]: what the
//              host batcher ships when only the used slots cross PCIe; geom1 = coff, force6 = the records.
// nrec (CSR only): number of records behind `force6`; an offset + slot at or beyond it is never dereferenced and
// marks the environment bad.
template <bool CSR>
__global__ __launch_bounds__(THREADS) void contact_kernel(ContactDev cd, int N, int C, long nrec,
                                                          const int* __restrict__ ncon,
                                                          const int* __restrict__ geom1,
                                                          const int* __restrict__ geom2,
                                                          const double* __restrict__ force6,
                                                          const double* __restrict__ pos_z,
                                                          int* __restrict__ n_r, int* __restrict__ n_l,
                                                          int* __restrict__ idx_r, int* __restrict__ idx_l,
                                                          double* __restrict__ grf_r,
                                                          double* __restrict__ grf_l,
                                                          double* __restrict__ min_z,
                                                          uint8_t* __restrict__ bad) {
  const int lane = threadIdx.x & 63;
  const int slot = lane & (SLOTS - 1);
  const int grp = lane / SLOTS;  // env within the wave
  const long wave = ((long)blockIdx.x * THREADS + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * THREADS) >> 6;
  const int passes = (C + SLOTS - 1) / SLOTS;
  for (long base = wave * 4; base < N; base += nwaves * 4) {
    const long n = base + grp;
    const bool env_ok = n < N;
    const int nc_raw = env_ok ? ncon[n] : 0;
    const int nc = min(max(nc_raw, 0), C);
    int cnt_r = 0, cnt_l = 0;
    double sum_r = 0.0, sum_l = 0.0, mz = 0.0;
    bool have = false, oob = false;
    for (int ps = 0; ps < passes; ++ps) {
      const int i = ps * SLOTS + slot;
      bool is_r = false, is_l = false;
      double nrm = 0.0, pz = 0.0;
      bool in_range = env_ok && i < nc;
      if (CSR && in_range) {
        const long e = (long)geom1[n] + i;
        if (e < 0 || e >= nrec) { in_range = false; oob = true; }
      }
      if (in_range) {
        const size_t e = CSR ? (size_t)geom1[n] + i : (size_t)n * C + i;
        const oly_contact_record* rec = reinterpret_cast<const oly_contact_record*>(force6) + e;
        const int g1 = CSR ? rec->geom1 : geom1[e], g2 = CSR ? rec->geom2 : geom2[e];
        if (g1 >= 0 && g1 < cd.ngeom && g2 >= 0 && g2 < cd.ngeom) {
          const int b1 = cd.geom_bodyid[g1], b2 = cd.geom_bodyid[g2];
          is_r = (b1 == cd.floor_body) && (b2 == cd.rfoot_body);
          is_l = (b1 == cd.floor_body) && (b2 == cd.lfoot_body);
        }
        if (is_r || is_l) {
          const double* f = CSR ? rec->force6 : force6 + e * 6;
          double s = 0.0;
#pragma unroll
          for (int k = 0; k < 6; ++k) s += f[k] * f[k];
          nrm = sqrt(s);
          pz = CSR ? rec->pos_z : pos_z[e];
        }
      }
      // ballots over the whole wave, then this env's 16-bit field
      const unsigned long long br = __ballot(is_r), bl = __ballot(is_l);
      const unsigned mr = (unsigned)((br >> (grp * SLOTS)) & 0xffffu);
      const unsigned ml = (unsigned)((bl >> (grp * SLOTS)) & 0xffffu);
      const unsigned below = (1u << slot) - 1u;
      if (is_r && idx_r) idx_r[(size_t)n * C + cnt_r + __popc(mr & below)] = i;
      if (is_l && idx_l) idx_l[(size_t)n * C + cnt_l + __popc(ml & below)] = i;
      cnt_r += __popc(mr);
      cnt_l += __popc(ml);
      // in-order sums over the MATCHING slots only (ascending slot = contact order): every lane
      // of the group ends up with the same chain ((0 + n_a) + n_b) + ... as the reference's loop;
      // the trip count is the largest number of foot contacts among the wave's four environments
      unsigned rem = mr | ml;
      while (__any(rem != 0u)) {
        const int k = rem ? (__ffs((int)rem) - 1) : 0;
        const double vk = __shfl(nrm, grp * SLOTS + k, 64);
        const double zk = __shfl(pz, grp * SLOTS + k, 64);
        if (rem) {
          if ((mr >> k) & 1u) sum_r += vk;
          if ((ml >> k) & 1u) sum_l += vk;
          if (!have || zk < mz) mz = zk;
          have = true;
          rem &= rem - 1u;
        }
      }
    }
    // a record index outside the record array (CSR): the environment cannot be reduced, report it as bad
    const bool any_oob = ((__ballot(oob) >> (grp * SLOTS)) & 0xffffull) != 0ull;
    if (env_ok) {
      // -1 padding of the index lists
      for (int i = slot; i < C; i += SLOTS) {
        if (idx_r && i >= cnt_r) idx_r[(size_t)n * C + i] = -1;
        if (idx_l && i >= cnt_l) idx_l[(size_t)n * C + i] = -1;
      }
      if (slot == 0) {
        n_r[n] = cnt_r;
        n_l[n] = cnt_l;
        grf_r[n] = sum_r;
        grf_l[n] = sum_l;
        min_z[n] = have ? mz : 0.0;
        // against the RAW count: an environment with more contacts than the C staged slots (or a
        // negative count) cannot be reduced faithfully, so it is reported as a bad collision
        // instead of silently dropping the surplus (check_bad_collisions iterates all data.ncon)
        bad[n] = (uint8_t)(((cnt_r + cnt_l) != nc_raw) || any_oob);
      }
    }
  }
}
// IL ground forces: 16 lanes per environment walk the W substeps in order; per substep the
// first matching contact of every sensor pair is found by ballot + find-first-set over the
// 16-lane group (slot order == contact order), lanes 0..2 of the group fetch its force[:3].
// The reference scans ALL data.ncon contacts (UnitreeH1.py:113-123 -> _get_collision_force); only C slots
// are staged, so a substep whose raw count exceeds C is exact as long as every sensor pair found its first
// contact among the staged slots.  Otherwise (a pair without a hit, surplus contacts unseen; or a negative
// count) the environment's overflow byte is set: the caller must not use its row (the facade raises).
__global__ __launch_bounds__(THREADS) void il_grf_kernel(GrfDev gd, int W, int N, int C,
                                                         const int* __restrict__ ncon,
                                                         const int* __restrict__ geom1,
                                                         const int* __restrict__ geom2,
                                                         const double* __restrict__ force6,
                                                         double* __restrict__ grf_step,
                                                         double* __restrict__ grf_mean,
                                                         uint8_t* __restrict__ overflow) {
  const int lane = threadIdx.x & 63;
  const int slot = lane & (SLOTS - 1);
  const int grp = lane / SLOTS;
  const long wave = ((long)blockIdx.x * THREADS + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * THREADS) >> 6;
  const int passes = (C + SLOTS - 1) / SLOTS;
  const int ncomp = 3 * gd.n_pairs;
  for (long base = wave * 4; base < N; base += nwaves * 4) {
    const long n = base + grp;
    const bool env_ok = n < N;
    double acc = 0.0;  // lane `slot` < ncomp accumulates component `slot` of the window sum
    bool over = false;
    for (int w = 0; w < W; ++w) {
      const int nc_raw = env_ok ? ncon[(size_t)w * N + n] : 0;
      const int nc = min(max(nc_raw, 0), C);
      int first[OLY_MAX_GRF_PAIRS];
#pragma unroll
      for (int k = 0; k < OLY_MAX_GRF_PAIRS; ++k) first[k] = -1;
      for (int ps = 0; ps < passes; ++ps) {
        const int i = ps * SLOTS + slot;
        int ga = -1, gb = -1;
        if (env_ok && i < nc) {
          const size_t e = ((size_t)w * N + n) * C + i;
          const int g1 = geom1[e], g2 = geom2[e];
          if (g1 >= 0 && g1 < gd.ngeom) ga = gd.geom_group[g1];
          if (g2 >= 0 && g2 < gd.ngeom) gb = gd.geom_group[g2];
        }
#pragma unroll
        for (int k = 0; k < OLY_MAX_GRF_PAIRS; ++k) {
          if (k >= gd.n_pairs) break;
          const bool hit = ga >= 0 && gb >= 0 &&
                           ((ga == gd.pair_a[k] && gb == gd.pair_b[k]) || (ga == gd.pair_b[k] && gb == gd.pair_a[k]));
          const unsigned long long m = (__ballot(hit) >> (grp * SLOTS)) & 0xFFFFull;
          if (first[k] < 0 && m) first[k] = ps * SLOTS + (__ffsll((long long)m) - 1);
        }
      }
      if (nc_raw < 0) over = true;
      if (nc_raw > C) {
#pragma unroll
        for (int k = 0; k < OLY_MAX_GRF_PAIRS; ++k)
          if (k < gd.n_pairs && first[k] < 0) over = true;
      }
      if (env_ok && slot < ncomp) {
        const int k = slot / 3, c = slot - 3 * k;
        double v = 0.0;
        if (first[k] >= 0) v = force6[(((size_t)w * N + n) * C + first[k]) * 6 + c];
        if (grf_step) grf_step[((size_t)w * N + n) * ncomp + slot] = v;
        acc += v;
      }
    }
    if (env_ok && slot < ncomp) grf_mean[(size_t)n * ncomp + slot] = acc / (double)W;
    if (env_ok && slot == 0) overflow[n] = (uint8_t)over;
  }
}

// Window mean of per-substep ground-force vectors that are already dense rows (the host batcher packs
// the first-contact force of every sensor pair while it copies the contacts out of mjData): lane per
// output element, the W samples added in substep order, divided by W (RunningAveragedWindow after one
// control step, loco_env_base.py:1072-1084,1163-1174).  HBM: (W + 1) * K * 8 B per environment.
__global__ __launch_bounds__(THREADS) void il_grf_window_kernel(int W, long NK, const double* __restrict__ step,
                                                                double* __restrict__ mean) {
  const long stride = (long)gridDim.x * THREADS;
  for (long e = (long)blockIdx.x * THREADS + threadIdx.x; e < NK; e += stride) {
    double acc = 0.0;
    for (int w = 0; w < W; ++w) acc += step[(size_t)w * NK + e];
    mean[e] = acc / (double)W;
  }
}

}  // namespace

extern "C" int oly_il_grf_window(oly_ctx* ctx, int W, int N, int K, const double* grf_step, double* grf_mean,
                                 oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (W <= 0 || N < 0 || K <= 0) OLY_FAIL(ctx, OLY_EINVAL, "oly_il_grf_window: bad W, N or K");
  if (N == 0) return OLY_OK;
  if (!grf_step || !grf_mean) OLY_FAIL(ctx, OLY_EINVAL, "oly_il_grf_window: NULL pointer");
  const long NK = (long)N * K;
  long blocks = (NK + THREADS - 1) / THREADS;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(il_grf_window_kernel, dim3((unsigned)blocks), dim3(THREADS), 0, oly_s(stream), W, NK, grf_step,
                     grf_mean);
  OLY_LAUNCH_CHECK(ctx, "il_grf_window_kernel");
  return OLY_OK;
}

extern "C" int oly_contact_configure(oly_ctx* ctx, int ngeom, const int32_t* geom_bodyid_host,
                                     int floor_body, int rfoot_body, int lfoot_body) {
  if (!ctx) return OLY_EINVAL;
  if (ngeom <= 0 || !geom_bodyid_host) OLY_FAIL(ctx, OLY_EINVAL, "oly_contact_configure: bad argument");
  OLY_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->contact.geom_bodyid) { (void)hipFree(ctx->contact.geom_bodyid); ctx->contact.geom_bodyid = nullptr; }
  ctx->contact_ok = false;
  if (hipMalloc(&ctx->contact.geom_bodyid, sizeof(int) * ngeom) != hipSuccess)
    OLY_FAIL(ctx, OLY_ENOMEM, "oly_contact_configure: hipMalloc failed");
  OLY_HIP(ctx, hipMemcpy(ctx->contact.geom_bodyid, geom_bodyid_host, sizeof(int) * ngeom,
                         hipMemcpyHostToDevice));
  ctx->contact.ngeom = ngeom; ctx->contact.floor_body = floor_body;
  ctx->contact.rfoot_body = rfoot_body; ctx->contact.lfoot_body = lfoot_body;
  ctx->contact_ok = true;
  return OLY_OK;
}

extern "C" int oly_contact_reduce(oly_ctx* ctx, int N, int C, const int32_t* ncon, const int32_t* geom1,
                                  const int32_t* geom2, const double* force6, const double* pos_z,
                                  int32_t* n_r, int32_t* n_l, int32_t* idx_r, int32_t* idx_l,
                                  double* grf_r, double* grf_l, double* min_z, uint8_t* bad,
                                  oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (!ctx->contact_ok) OLY_FAIL(ctx, OLY_ENOTCONF, "oly_contact_reduce before oly_contact_configure");
  if (N < 0 || C <= 0) OLY_FAIL(ctx, OLY_EINVAL, "oly_contact_reduce: bad N or C");
  if (N == 0) return OLY_OK;
  if (!ncon || !geom1 || !geom2 || !force6 || !pos_z || !n_r || !n_l || !grf_r || !grf_l || !min_z || !bad)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_contact_reduce: NULL pointer");
  long waves = ((long)N + 3) / 4;
  long blocks = (waves * 64 + THREADS - 1) / THREADS;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(contact_kernel<false>, dim3((unsigned)blocks), dim3(THREADS), 0, oly_s(stream), ctx->contact,
                     N, C, 0L, ncon, geom1, geom2, force6, pos_z, n_r, n_l, idx_r, idx_l, grf_r, grf_l, min_z,
                     bad);
  OLY_LAUNCH_CHECK(ctx, "contact_kernel");
  return OLY_OK;
}

extern "C" int oly_contact_reduce_csr(oly_ctx* ctx, int N, int C, const int32_t* ncon, const int32_t* coff,
                                      const oly_contact_record* records, int64_t n_records, int32_t* n_r, int32_t* n_l,
                                      double* grf_r, double* grf_l, double* min_z, uint8_t* bad,
                                      oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (!ctx->contact_ok) OLY_FAIL(ctx, OLY_ENOTCONF, "oly_contact_reduce_csr before oly_contact_configure");
  if (N < 0 || C <= 0 || n_records < 0) OLY_FAIL(ctx, OLY_EINVAL, "oly_contact_reduce_csr: bad N, C or n_records");
  if (N == 0) return OLY_OK;
  if (!ncon || !coff || !records || !n_r || !n_l || !grf_r || !grf_l || !min_z || !bad)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_contact_reduce_csr: NULL pointer");
  long waves = ((long)N + 3) / 4;
  long blocks = (waves * 64 + THREADS - 1) / THREADS;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(contact_kernel<true>, dim3((unsigned)blocks), dim3(THREADS), 0, oly_s(stream), ctx->contact,
                     N, C, (long)n_records, ncon, coff, nullptr, reinterpret_cast<const double*>(records), nullptr, n_r, n_l,
                     nullptr, nullptr, grf_r, grf_l, min_z, bad);
  OLY_LAUNCH_CHECK(ctx, "contact_kernel<csr>");
  return OLY_OK;
}

extern "C" int oly_grf_configure(oly_ctx* ctx, int ngeom, const int32_t* geom_group_host, int n_pairs,
                                 const int32_t* pair_a, const int32_t* pair_b) {
  if (!ctx) return OLY_EINVAL;
  ctx->grf_ok = false;
  if (ngeom <= 0 || !geom_group_host || n_pairs <= 0 || n_pairs > OLY_MAX_GRF_PAIRS || !pair_a || !pair_b)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_grf_configure: bad argument (ngeom=%d n_pairs=%d)", ngeom, n_pairs);
  OLY_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->grf.geom_group) { (void)hipFree(ctx->grf.geom_group); ctx->grf.geom_group = nullptr; }
  if (hipMalloc(&ctx->grf.geom_group, sizeof(int) * ngeom) != hipSuccess)
    OLY_FAIL(ctx, OLY_ENOMEM, "oly_grf_configure: hipMalloc failed");
  OLY_HIP(ctx, hipMemcpy(ctx->grf.geom_group, geom_group_host, sizeof(int) * ngeom, hipMemcpyHostToDevice));
  free(ctx->grf_group_host);
  ctx->grf_group_host = static_cast<int*>(malloc(sizeof(int) * ngeom));
  if (!ctx->grf_group_host) OLY_FAIL(ctx, OLY_ENOMEM, "oly_grf_configure: out of host memory");
  memcpy(ctx->grf_group_host, geom_group_host, sizeof(int) * ngeom);
  ctx->grf.ngeom = ngeom;
  ctx->grf.n_pairs = n_pairs;
  for (int k = 0; k < n_pairs; ++k) { ctx->grf.pair_a[k] = pair_a[k]; ctx->grf.pair_b[k] = pair_b[k]; }
  ctx->grf_ok = true;
  return OLY_OK;
}

extern "C" int oly_il_ground_forces(oly_ctx* ctx, int W, int N, int C, const int32_t* ncon,
                                    const int32_t* geom1, const int32_t* geom2, const double* force6,
                                    double* grf_step, double* grf_mean, uint8_t* overflow, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (!ctx->grf_ok) OLY_FAIL(ctx, OLY_ENOTCONF, "oly_il_ground_forces before oly_grf_configure");
  if (W <= 0 || N < 0 || C <= 0) OLY_FAIL(ctx, OLY_EINVAL, "oly_il_ground_forces: bad W, N or C");
  if (N == 0) return OLY_OK;
  if (!ncon || !geom1 || !geom2 || !force6 || !grf_mean || !overflow)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_il_ground_forces: NULL pointer");
  long waves = ((long)N + 3) / 4;
  long blocks = (waves * 64 + THREADS - 1) / THREADS;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(il_grf_kernel, dim3((unsigned)blocks), dim3(THREADS), 0, oly_s(stream), ctx->grf, W, N, C,
                     ncon, geom1, geom2, force6, grf_step, grf_mean, overflow);
  OLY_LAUNCH_CHECK(ctx, "il_grf_kernel");
  return OLY_OK;
}
