// Device helpers shared by the one-launch-per-step vec kernel (K10, csrc/k10_vec_step.hip) and the persistent
// rollout kernel (K13, csrc/k13_rollout.hip): both evaluate WalkingTask / get_obs with these very functions on the
// same arguments (same libm entry points, -ffp-contract=off), which is what makes their results bit-identical.
#pragma once
#include "oly_common.h"

namespace oly_a3v {
constexpr double PI = 3.141592653589793;
constexpr double EPS = 2.220446049250313e-16;
enum { F_NONE = 0, F_SINCOS = 1, F_TAN = 2, F_EXP = 3, F_ATAN2 = 4 };

__device__ __forceinline__ void quat2mat(double w, double x, double y, double z, double R[3][3]) {
  const double nq = w * w + x * x + y * y + z * z;
  if (nq < EPS) {
    R[0][0] = 1; R[0][1] = 0; R[0][2] = 0;
    R[1][0] = 0; R[1][1] = 1; R[1][2] = 0;
    R[2][0] = 0; R[2][1] = 0; R[2][2] = 1;
    return;
  }
  const double s = 2.0 / nq;
  const double X = x * s, Y = y * s, Z = z * s;
  const double wX = w * X, wY = w * Y, wZ = w * Z;
  const double xX = x * X, xY = x * Y, xZ = x * Z;
  const double yY = y * Y, yZ = y * Z, zZ = z * Z;
  R[0][0] = 1.0 - (yY + zZ); R[0][1] = xY - wZ;         R[0][2] = xZ + wY;
  R[1][0] = xY + wZ;         R[1][1] = 1.0 - (xX + zZ); R[1][2] = yZ - wX;
  R[2][0] = xZ - wY;         R[2][1] = yZ + wX;         R[2][2] = 1.0 - (xX + yY);
}

__device__ __forceinline__ double vnorm3(double a0, double a1, double a2) {
  return sqrt(a0 * a0 + a1 * a1 + a2 * a2);
}

// one function on one argument per lane; lanes of different classes diverge, so a call costs one
// evaluation per class present in the wave
__device__ __forceinline__ void eval_task(int cls, double a, double b, double& r0, double& r1) {
  r0 = 0.0;
  r1 = 0.0;
  if (cls == F_SINCOS) {
    sincos(a, &r0, &r1);   // bit-identical to sin() / cos() on gfx950 (tools/hip/check_sincos.hip: 2^24 arguments)
  } else if (cls == F_TAN) {
    r0 = tan(a);
  } else if (cls == F_EXP) {
    r0 = exp(a);
  } else if (cls == F_ATAN2) {
    r0 = atan2(a, b);
  }
}

}  // namespace oly_a3v
