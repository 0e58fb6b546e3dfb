// cs_robust.h -- g2o's robust kernels on the device (and in host code that must agree with it).
//
// RobustKernel::robustify(e, rho) of object_slam/Thirdparty/g2o/g2o/core/robust_kernel_impl.cpp:78-165 maps an edge's squared error
// e = err^T Omega err to rho(e) and rho'(e); the solver uses rho for chi2 (sparse_optimizer.cpp:100-114) and rho' as the weight of
// Omega and of -Omega err in the quadratic form (base_binary_edge.hpp:88-111, base_edge.h:96-102 -- the second-derivative term is
// commented out in the reference, so rho'' is never needed).  The kinds are the classes the reference registers
// (robust_kernel_impl.cpp:169-174); `delta` is RobustKernel::delta().
//
// Two members of the vendored g2o are single precision and are reproduced as such: RobustKernelHuber::dsqr (robust_kernel_impl.h:86,
// assigned delta * delta by setDelta, robust_kernel_impl.cpp:65-69) and RobustKernelTukey::_deltaSqr / _invDeltaSqr (:107-108; here
// derived from delta the way a caller of setDeltaSqr(d * d, 1 / (d * d)) would set them).
#pragma once
#include "cs_se3.h"

namespace cs {

enum { RK_NONE = 0, RK_HUBER = 1, RK_PSEUDO_HUBER = 2, RK_CAUCHY = 3, RK_SATURATED = 4, RK_DCS = 5, RK_TUKEY = 6, RK_KINDS = 7 };

// Huber alone (the projection edges' fast path: delta <= 0 means no kernel)
CS_HD void huber_rho(double e, double delta, double& rho0, double& rho1) {
  if (delta > 0) {
    const double dsqr = (double)(float)(delta * delta);      // `float dsqr`
    if (e <= dsqr) { rho0 = e; rho1 = 1.0; }
    else { const double sq = sqrt(e); rho0 = 2 * sq * delta - dsqr; rho1 = delta / sq; }
  } else { rho0 = e; rho1 = 1.0; }
}

CS_HD void robust_rho(int kind, double delta, double e, double& rho0, double& rho1) {
  switch (kind) {
    case RK_HUBER: huber_rho(e, delta, rho0, rho1); break;
    case RK_PSEUDO_HUBER: {
      const double dsqr = delta * delta, dsqrReci = 1. / dsqr, aux1 = dsqrReci * e + 1.0, aux2 = sqrt(aux1);
      rho0 = 2 * dsqr * (aux2 - 1); rho1 = 1. / aux2;
      break;
    }
    case RK_CAUCHY: {
      const double dsqr = delta * delta, dsqrReci = 1. / dsqr, aux = dsqrReci * e + 1.0;
      rho0 = dsqr * log(aux); rho1 = 1. / aux;
      break;
    }
    case RK_SATURATED: {
      const double dsqr = delta * delta;
      if (e <= dsqr) { rho0 = e; rho1 = 1.; } else { rho0 = dsqr; rho1 = 0.; }
      break;
    }
    case RK_DCS: {
      double scale = (2.0 * delta) / (delta + e);
      if (scale >= 1.0) scale = 1.0;
      rho0 = scale * e * scale; rho1 = scale * scale;
      break;
    }
    case RK_TUKEY: {
      const double deltaSqr = (double)(float)(delta * delta), invDeltaSqr = (double)(float)(1.0 / (delta * delta));
      if (e <= deltaSqr) { const double factor = e * invDeltaSqr, d = 1 - factor, dd = d * d; rho0 = deltaSqr * (1 - dd * d); rho1 = 3 * dd; }
      else { rho0 = deltaSqr; rho1 = 0.; }
      break;
    }
    default: rho0 = e; rho1 = 1.0;
  }
}

}  // namespace cs
