// ba_math.h - per-observation / per-block arithmetic of the bundle-adjustment
// inner loop, written once and inlined into the gfx950 kernels (ba_obs_kernels.h, ba_schur_kernels.h, ba_schur_window_kernels.h).
// Every function is plain fp64 register arithmetic on tiny fixed-size blocks
// (<= 6x6): no MFMA, no memory traffic.  Reference citations are to
// alexflint/pysfm.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#define BA_HD __host__ __device__ __forceinline__
#else
#define BA_HD inline
#endif

namespace ba {

enum { SENSOR_GAUSS = 0, SENSOR_CAUCHY = 1, SENSOR_HUBER = 2, SENSOR_TABLE = 3 };

struct Sensor {
  int kind;
  double L[4];    // Gaussian: r = L e, L = chol(cov^-1) (sensor_model.py:16-17, 23-29)
  double sigma;   // Cauchy (sensor_model.py:41-43)
  double k;       // Huber threshold (not in the reference)
  int fast;       // set by the host, uniform over a launch: bit 0 = Gaussian with L = I (r = e, Jr = I),
                  // bit 1 = K = I.  The kernels are bound by fp64 issue, and these two cases (the reference's
                  // own defaults: bundle.py:139, synthetic_data.py K = eye) skip a fifth of the per-observation work.
  // SENSOR_TABLE: any isotropic robustifier r = h(rho) e, rho = |e| (the reference's plug-in point, sensor_model.py:19-32:
  // an object with four methods) - h sampled by the host on a grid uniform in log2(rho): node i at rho_i = 2^(u0 + i / inv_du)
  // holds (h(rho_i), dh/du(rho_i)); cubic Hermite interpolation in u (h to ~1e-13, its derivative to ~3e-10 at 256 nodes per
  // octave).  J = h I + (h'(rho) / rho) e e^T.  Below the first node h is the constant tab[0]; beyond the last, the last cell's
  // cubic.  The table's address and shape travel in the fields the other kinds use (the struct is a kernel argument of every
  // launch): L[0] = the device address (bit pattern), L[1] = u0, L[2] = inv_du, L[3] = number of nodes.
};
enum { FAST_UNIT_GAUSS = 1, FAST_K_IDENTITY = 2 };

// residual r and 2x2 Jacobian J (row-major) of the sensor model at error e.
// Gaussian: sensor_model.py:23-29.  Cauchy: sensor_model.py:48-69, including
// its linear window |e| < 1e-5 and its log(1 + rho^2/sigma^2) (not log1p).
// Huber: rho_H(s) = s^2 (s <= k), 2ks - k^2 otherwise, same vector-residual form.
// Table: any isotropic robustifier the caller defines in Python, interpolated (see Sensor).
// TABLE: whether this instance can evaluate SENSOR_TABLE.  The kernels on the trial's fast path are compiled without it (their
// registers and code size are what they were); a handle whose sensor model is a table takes the general kernels (k_linearize,
// k_camera_blocks, k_schur_pairs / the dense reduction, k_backsub, k_cost), which are.
template <bool TABLE = false>
BA_HD void sensor_eval(const Sensor& s, double e0, double e1, double r[2], double J[4]) {
  if (s.kind == SENSOR_GAUSS) {
    r[0] = s.L[0] * e0 + s.L[1] * e1;
    r[1] = s.L[2] * e0 + s.L[3] * e1;
    J[0] = s.L[0]; J[1] = s.L[1]; J[2] = s.L[2]; J[3] = s.L[3];
    return;
  }
  const double rho2 = e0 * e0 + e1 * e1;
  const double rho = sqrt(rho2);
  if (s.kind == SENSOR_CAUCHY) {
    if (rho < 1e-5) {
      const double is = 1.0 / s.sigma;
      r[0] = e0 * is; r[1] = e1 * is;
      J[0] = is; J[1] = 0.0; J[2] = 0.0; J[3] = is;
      return;
    }
    const double s2 = s.sigma * s.sigma;
    const double g = sqrt(log(1.0 + rho2 / s2));
    const double gr = g / rho;
    r[0] = e0 * gr; r[1] = e1 * gr;
    // J = ee^T / (rho g (rho^2+sigma^2)) + (rho I - ee^T/rho) g / rho^2
    const double a = 1.0 / (rho * g * (rho2 + s2));
    const double c = g / rho2;
    const double ir = 1.0 / rho;
    J[0] = e0 * e0 * a + (rho - e0 * e0 * ir) * c;
    J[1] = e0 * e1 * a + (-e0 * e1 * ir) * c;
    J[2] = J[1];
    J[3] = e1 * e1 * a + (rho - e1 * e1 * ir) * c;
    return;
  }
  if (TABLE && s.kind == SENSOR_TABLE) {
    unsigned long long bits;
    __builtin_memcpy(&bits, &s.L[0], 8);
    const double* tab = reinterpret_cast<const double*>(bits);
    const double u0 = s.L[1], inv_du = s.L[2];
    const int n = (int)s.L[3];
    double h = tab[0], q = 0.0;                           // q = h'(rho) / rho
    if (rho2 > 0.0) {
      const double t = (0.5 * log2(rho2) - u0) * inv_du;
      if (t > 0.0) {
        int i = (int)t;
        if (i > n - 2) i = n - 2;
        const double f = t - (double)i, du = 1.0 / inv_du;
        const double h0 = tab[2 * i], m0 = tab[2 * i + 1] * du, h1 = tab[2 * i + 2], m1 = tab[2 * i + 3] * du;
        // Hermite cubic on [0, 1]: c0 + c1 f + c2 f^2 + c3 f^3
        const double c2 = 3.0 * (h1 - h0) - 2.0 * m0 - m1, c3 = 2.0 * (h0 - h1) + m0 + m1;
        h = h0 + f * (m0 + f * (c2 + f * c3));
        const double dhdu = (m0 + f * (2.0 * c2 + 3.0 * f * c3)) * inv_du;
        q = dhdu * 1.4426950408889634 / rho2;           // dh/drho = dh/du / (rho ln 2); then / rho
      }
    }
    r[0] = e0 * h; r[1] = e1 * h;
    J[0] = h + e0 * e0 * q;
    J[1] = e0 * e1 * q;
    J[2] = J[1];
    J[3] = h + e1 * e1 * q;
    return;
  }
  // Huber
  if (rho <= s.k) {
    r[0] = e0; r[1] = e1;
    J[0] = 1.0; J[1] = 0.0; J[2] = 0.0; J[3] = 1.0;
    return;
  }
  const double g = sqrt(2.0 * s.k * rho - s.k * s.k);
  const double gr = g / rho;
  r[0] = e0 * gr; r[1] = e1 * gr;
  const double q = ((s.k / g) * rho - g) / (rho2 * rho);
  J[0] = gr + e0 * e0 * q;
  J[1] = e0 * e1 * q;
  J[2] = J[1];
  J[3] = gr + e1 * e1 * q;
}

// e = pr(K (R x + t)) - z       (algebra.py:5-12, bundle.py:14-19, 243-248)
// cam = [R row-major (9) | t (3)].  Returns the homogeneous prediction in p and 1 / p[2] in iz: ONE fp64 division
// (~11 instructions on this hardware) serves the projection and its Jacobian; p0 * iz differs from p0 / p2 by one
// rounding (1e-16 relative).
BA_HD void reproj_error(const double* K, const double* cam, const double* x,
                        double z0, double z1, double p[3], double e[2], double& iz, int fast = 0) {
  const double y0 = cam[0] * x[0] + cam[1] * x[1] + cam[2] * x[2] + cam[9];
  const double y1 = cam[3] * x[0] + cam[4] * x[1] + cam[5] * x[2] + cam[10];
  const double y2 = cam[6] * x[0] + cam[7] * x[1] + cam[8] * x[2] + cam[11];
  if (fast & FAST_K_IDENTITY) {
    p[0] = y0; p[1] = y1; p[2] = y2;
  } else {
    p[0] = K[0] * y0 + K[1] * y1 + K[2] * y2;
    p[1] = K[3] * y0 + K[4] * y1 + K[5] * y2;
    p[2] = K[6] * y0 + K[7] * y1 + K[8] * y2;
  }
  iz = 1.0 / p[2];
  e[0] = p[0] * iz - z0;
  e[1] = p[1] * iz - z1;
}

// residual only (Bundle.residual, bundle.py:251-252)
template <bool TABLE = false>
BA_HD void obs_residual(const double* K, const double* cam, const double* x, double z0, double z1,
                        const Sensor& s, double e[2], double r[2]) {
  double p[3], J[4], iz;
  reproj_error(K, cam, x, z0, z1, p, e, iz, s.fast);
  if (s.fast & FAST_UNIT_GAUSS) { r[0] = e[0]; r[1] = e[1]; return; }
  sensor_eval<TABLE>(s, e[0], e[1], r, J);
}

// Bundle.Jresidual (bundle.py:255-277): Jc = Jr [J_R | J_t] (2x6), Jp = Jr J_x (2x3)
//   Jpr (bundle.py:8-11), J_t = Jpr K, J_x = J_t R, J_R = J_x skew(-x) (lie.py:38-40)
template <bool TABLE = false>
BA_HD void obs_linearize(const double* K, const double* cam, const double* x, double z0, double z1,
                         const Sensor& s, double e[2], double r[2], double Jc[12], double Jp[6]) {
  double p[3], Jr[4], iz;
  reproj_error(K, cam, x, z0, z1, p, e, iz, s.fast);
  const bool unit = (s.fast & FAST_UNIT_GAUSS) != 0;
  if (unit) { r[0] = e[0]; r[1] = e[1]; }
  else sensor_eval<TABLE>(s, e[0], e[1], r, Jr);
  const double jp02 = -(p[0] * iz) * iz;
  const double jp12 = -(p[1] * iz) * iz;
  double Jt[6], Jx[6], JR[6];
  if (s.fast & FAST_K_IDENTITY) {
    Jt[0] = iz; Jt[1] = 0.0; Jt[2] = jp02;
    Jt[3] = 0.0; Jt[4] = iz; Jt[5] = jp12;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      Jx[c] = iz * cam[c] + jp02 * cam[6 + c];
      Jx[3 + c] = iz * cam[3 + c] + jp12 * cam[6 + c];
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      Jt[c] = iz * K[c] + jp02 * K[6 + c];
      Jt[3 + c] = iz * K[3 + c] + jp12 * K[6 + c];
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        Jx[rr * 3 + c] = Jt[rr * 3 + 0] * cam[c] + Jt[rr * 3 + 1] * cam[3 + c] + Jt[rr * 3 + 2] * cam[6 + c];
  }
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    // J_x * skew(-x),  skew(-x) = [[0, x2, -x1], [-x2, 0, x0], [x1, -x0, 0]]
    JR[rr * 3 + 0] = -Jx[rr * 3 + 1] * x[2] + Jx[rr * 3 + 2] * x[1];
    JR[rr * 3 + 1] = Jx[rr * 3 + 0] * x[2] - Jx[rr * 3 + 2] * x[0];
    JR[rr * 3 + 2] = -Jx[rr * 3 + 0] * x[1] + Jx[rr * 3 + 1] * x[0];
  }
  if (unit) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      Jc[c] = JR[c]; Jc[6 + c] = JR[3 + c];
      Jc[3 + c] = Jt[c]; Jc[9 + c] = Jt[3 + c];
      Jp[c] = Jx[c]; Jp[3 + c] = Jx[3 + c];
    }
    return;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    Jc[c] = Jr[0] * JR[c] + Jr[1] * JR[3 + c];
    Jc[6 + c] = Jr[2] * JR[c] + Jr[3] * JR[3 + c];
    Jc[3 + c] = Jr[0] * Jt[c] + Jr[1] * Jt[3 + c];
    Jc[9 + c] = Jr[2] * Jt[c] + Jr[3] * Jt[3 + c];
    Jp[c] = Jr[0] * Jx[c] + Jr[1] * Jx[3 + c];
    Jp[3 + c] = Jr[2] * Jx[c] + Jr[3] * Jx[3 + c];
  }
}

// W = Jc^T Jp  (6x3 row-major): the HCP block of one observation
// (bundle_adjuster.py:232)
BA_HD void block_W(const double Jc[12], const double Jp[6], double W[18]) {
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) W[a * 3 + c] = Jc[a] * Jp[c] + Jc[6 + a] * Jp[3 + c];
}

// symmetric 3x3 stored as [xx, xy, xz, yy, yz, zz]
BA_HD void sym3_apply(const double A[6], const double v[3], double out[3]) {
  out[0] = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
  out[1] = A[1] * v[0] + A[3] * v[1] + A[4] * v[2];
  out[2] = A[2] * v[0] + A[4] * v[1] + A[5] * v[2];
}

// T = W * A  (6x3 times symmetric 3x3)
BA_HD void block_T(const double W[18], const double A[6], double T[18]) {
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    const double w0 = W[a * 3], w1 = W[a * 3 + 1], w2 = W[a * 3 + 2];
    T[a * 3 + 0] = w0 * A[0] + w1 * A[1] + w2 * A[2];
    T[a * 3 + 1] = w0 * A[1] + w1 * A[3] + w2 * A[4];
    T[a * 3 + 2] = w0 * A[2] + w1 * A[4] + w2 * A[5];
  }
}

// Eigen-decomposition of a symmetric 3x3 by cyclic Jacobi rotations.
// A (sym6) -> eigenvalues w[3], eigenvectors in the columns of V (row-major 3x3).
BA_HD void sym3_eig(const double A[6], double w[3], double V[9]) {
  double a00 = A[0], a01 = A[1], a02 = A[2], a11 = A[3], a12 = A[4], a22 = A[5];
  V[0] = 1; V[1] = 0; V[2] = 0; V[3] = 0; V[4] = 1; V[5] = 0; V[6] = 0; V[7] = 0; V[8] = 1;
  for (int sweep = 0; sweep < 24; ++sweep) {
    const double off = fabs(a01) + fabs(a02) + fabs(a12);
    const double diag = fabs(a00) + fabs(a11) + fabs(a22);
    if (off <= 1e-300 || off <= 1e-22 * diag) break;
    // rotate (0,1)
    if (a01 != 0.0) {
      const double th = (a11 - a00) / (2.0 * a01);
      const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
      const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
      a00 -= t * a01; a11 += t * a01; a01 = 0.0;
      const double b02 = c * a02 - s * a12, b12 = s * a02 + c * a12;
      a02 = b02; a12 = b12;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double v0 = V[r * 3 + 0], v1 = V[r * 3 + 1];
        V[r * 3 + 0] = c * v0 - s * v1; V[r * 3 + 1] = s * v0 + c * v1;
      }
    }
    // rotate (0,2)
    if (a02 != 0.0) {
      const double th = (a22 - a00) / (2.0 * a02);
      const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
      const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
      a00 -= t * a02; a22 += t * a02; a02 = 0.0;
      const double b01 = c * a01 - s * a12, b12 = s * a01 + c * a12;
      a01 = b01; a12 = b12;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double v0 = V[r * 3 + 0], v2 = V[r * 3 + 2];
        V[r * 3 + 0] = c * v0 - s * v2; V[r * 3 + 2] = s * v0 + c * v2;
      }
    }
    // rotate (1,2)
    if (a12 != 0.0) {
      const double th = (a22 - a11) / (2.0 * a12);
      const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
      const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
      a11 -= t * a12; a22 += t * a12; a12 = 0.0;
      const double b01 = c * a01 - s * a02, b02 = s * a01 + c * a02;
      a01 = b01; a02 = b02;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double v1 = V[r * 3 + 1], v2 = V[r * 3 + 2];
        V[r * 3 + 1] = c * v1 - s * v2; V[r * 3 + 2] = s * v1 + c * v2;
      }
    }
  }
  w[0] = a00; w[1] = a11; w[2] = a22;
}

// numpy.linalg.pinv(A, rcond) for a symmetric 3x3 (bundle_adjuster.py:256):
// singular values s_i = |lambda_i|; keep s_i > rcond * s_max; out = sum v v^T / lambda.
BA_HD void sym3_pinv(const double A[6], double rcond, double out[6]) {
  double w[3], V[9];
  sym3_eig(A, w, V);
  const double smax = fmax(fabs(w[0]), fmax(fabs(w[1]), fabs(w[2])));
  const double cut = rcond * smax;
#pragma unroll
  for (int i = 0; i < 6; ++i) out[i] = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (fabs(w[k]) > cut) {
      const double iw = 1.0 / w[k];
      const double v0 = V[k], v1 = V[3 + k], v2 = V[6 + k];
      out[0] += iw * v0 * v0; out[1] += iw * v0 * v1; out[2] += iw * v0 * v2;
      out[3] += iw * v1 * v1; out[4] += iw * v1 * v2; out[5] += iw * v2 * v2;
    }
  }
}

// numpy.linalg.pinv(A, rcond) again, with a short cut for the common case: for a positive definite
// block, lambda_min / lambda_max >= det / trace^3 (lambda_min >= det / lambda_max^2, lambda_max <= trace),
// so when det > rcond * trace^3 nothing is cut off and the pseudo-inverse IS the inverse - 30 flops by
// cofactors instead of the Jacobi eigen-solve (a few thousand cycles of dependent fp64 work per wavefront).
BA_HD void sym3_pinv_fast(const double A[6], double rcond, double out[6]) {
  const double tr = A[0] + A[3] + A[5];
  const double c00 = A[3] * A[5] - A[4] * A[4];
  const double c01 = A[2] * A[4] - A[1] * A[5];
  const double c02 = A[1] * A[4] - A[2] * A[3];
  const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  const bool easy = A[0] > 0.0 && A[3] > 0.0 && A[5] > 0.0 && c00 > 0.0 && det > rcond * tr * tr * tr && det < INFINITY;
  if (easy) {
    const double id = 1.0 / det;
    out[0] = c00 * id; out[1] = c01 * id; out[2] = c02 * id;
    out[3] = (A[0] * A[5] - A[2] * A[2]) * id;
    out[4] = (A[1] * A[2] - A[0] * A[4]) * id;
    out[5] = (A[0] * A[3] - A[1] * A[1]) * id;
  } else {
    sym3_pinv(A, rcond, out);
  }
}

// numpy.linalg.inv for a symmetric 3x3 (bundle_adjuster.py:254).  Returns false
// when the block is singular (numpy raises LinAlgError there).
// A = L D L^T for a symmetric 3 x 3 matrix [a00 a01 a02 a11 a12 a22], L unit lower: D[3], Lo = {L10, L20, L21}.
// Made for the per-point pseudo-inverse (positive semi-definite, possibly rank deficient: its eigenvalues
// are either >= rcond * the largest, or exactly cut to zero, which the factorisation sees as a pivot of
// ~1e-16 * trace): a pivot below tol_rel * trace is a cut direction - its column of L and its D are zero
// (for a semi-definite matrix the rest of that column is zero too).  sym3_ldl_tolerance puts tol_rel into
// the gap between the two; 0 for the plain inverse (no cut: only an exactly zero pivot is dropped).
BA_HD double sym3_ldl_tolerance(double rcond) {
  return rcond < 0.0 ? 0.0 : fmin(1e-10, fmax(1e-15, 1e-3 * rcond));
}

BA_HD void sym3_ldl(const double A[6], double tol_rel, double D[3], double Lo[3]) {
  const double tol = tol_rel * (fabs(A[0]) + fabs(A[3]) + fabs(A[5]));
  const bool z0 = !(fabs(A[0]) > tol);
  const double i0 = z0 ? 0.0 : 1.0 / A[0];
  D[0] = z0 ? 0.0 : A[0];
  Lo[0] = A[1] * i0;
  Lo[1] = A[2] * i0;
  const double d1 = A[3] - Lo[0] * A[1];
  const bool z1 = !(fabs(d1) > tol);
  const double i1 = z1 ? 0.0 : 1.0 / d1;
  D[1] = z1 ? 0.0 : d1;
  const double t = A[4] - Lo[1] * A[1];
  Lo[2] = t * i1;
  const double d2 = A[5] - Lo[1] * A[2] - Lo[2] * t;
  D[2] = fabs(d2) > tol ? d2 : 0.0;
}

BA_HD bool sym3_inv(const double A[6], double out[6]) {
  const double c00 = A[3] * A[5] - A[4] * A[4];
  const double c01 = A[2] * A[4] - A[1] * A[5];
  const double c02 = A[1] * A[4] - A[2] * A[3];
  const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  if (det == 0.0 || !(fabs(det) < INFINITY)) {
#pragma unroll
    for (int i = 0; i < 6; ++i) out[i] = 0.0;
    return false;
  }
  const double id = 1.0 / det;
  out[0] = c00 * id; out[1] = c01 * id; out[2] = c02 * id;
  out[3] = (A[0] * A[5] - A[2] * A[2]) * id;
  out[4] = (A[1] * A[2] - A[0] * A[4]) * id;
  out[5] = (A[0] * A[3] - A[1] * A[1]) * id;
  return true;
}

// SO3.exp (lie.py:21-34): Rodrigues, identity when |m| < 1e-8.  E row-major 3x3.
BA_HD void so3_exp(const double m[3], double E[9]) {
  const double th2 = m[0] * m[0] + m[1] * m[1] + m[2] * m[2];
  const double th = sqrt(th2);
  if (th < 1e-8) {
    E[0] = 1; E[1] = 0; E[2] = 0; E[3] = 0; E[4] = 1; E[5] = 0; E[6] = 0; E[7] = 0; E[8] = 1;
    return;
  }
  const double A = sin(th) / th;
  const double B = (1.0 - cos(th)) / th2;
  // skew(m)^2 = m m^T - |m|^2 I
  E[0] = 1.0 + B * (m[0] * m[0] - th2);
  E[1] = -A * m[2] + B * m[0] * m[1];
  E[2] = A * m[1] + B * m[0] * m[2];
  E[3] = A * m[2] + B * m[0] * m[1];
  E[4] = 1.0 + B * (m[1] * m[1] - th2);
  E[5] = -A * m[0] + B * m[1] * m[2];
  E[6] = -A * m[1] + B * m[0] * m[2];
  E[7] = A * m[0] + B * m[1] * m[2];
  E[8] = 1.0 + B * (m[2] * m[2] - th2);
}

// Camera.perturb (bundle.py:76-80): R <- R exp(d[0:3]), t <- t + d[3:6]
BA_HD void camera_perturb(const double cam[12], const double d[6], double out[12]) {
  double E[9];
  so3_exp(d, E);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      out[r * 3 + c] = cam[r * 3] * E[c] + cam[r * 3 + 1] * E[3 + c] + cam[r * 3 + 2] * E[6 + c];
  out[9] = cam[9] + d[3]; out[10] = cam[10] + d[4]; out[11] = cam[11] + d[5];
}

}  // namespace ba
