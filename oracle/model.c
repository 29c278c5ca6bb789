/* model.c -- CPU ORACLE (test infrastructure, never shipped): restatement of
 * the constraint models and the ODE-style step pieces around the solve.
 * Reference: eggshell/joints.cc, contact.cc, utils.cc, ensembles.cc. */
#include "egs_oracle.h"
#include "linalg.h"

static const double kGravity[3] = {0.0, 0.0, -9.8}; /* constants.h:8 */

/* utils.cc:233-237: Quaterniond::FromTwoVectors(a,b).toRotationMatrix().
 * Follows Eigen 3.3.8 Geometry/Quaternion.h (setFromTwoVectors): v0,v1
 * normalised, c = v1.v0; if c >= -1+1e-12: axis = v0 x v1, s = sqrt((1+c)*2),
 * vec = axis/s (as axis*(1/s)), w = s/2.
 * DEVIATION (documented in DESIGN.md): for c < -1+1e-12 Eigen takes the axis
 * from a JacobiSVD null vector, which is implementation-defined; here the axis
 * is normalize(v0 x e) with e the coordinate axis least aligned with v0. */
void orc_align_vectors(const double a[3], const double b[3], double Rout[9]) {
  double v0[3], v1[3];
  double na = dot3(a, a), nb = dot3(b, b);
  if (na > 0) { double s = sqrt(na); v0[0] = a[0] / s; v0[1] = a[1] / s; v0[2] = a[2] / s; }
  else { v0[0] = a[0]; v0[1] = a[1]; v0[2] = a[2]; }
  if (nb > 0) { double s = sqrt(nb); v1[0] = b[0] / s; v1[1] = b[1] / s; v1[2] = b[2] / s; }
  else { v1[0] = b[0]; v1[1] = b[1]; v1[2] = b[2]; }
  double c = dot3(v1, v0);
  double qw, q[3];
  if (c < -1.0 + 1e-12) {
    if (c < -1.0) c = -1.0;
    double ax = fabs(v0[0]), ay = fabs(v0[1]), az = fabs(v0[2]);
    double e[3] = {0, 0, 0};
    if (ax <= ay && ax <= az) e[0] = 1; else if (ay <= az) e[1] = 1; else e[2] = 1;
    double axis[3];
    cross3(v0, e, axis);
    double n = sqrt(dot3(axis, axis));
    axis[0] /= n; axis[1] /= n; axis[2] /= n;
    double w2 = (1.0 + c) * 0.5;
    qw = sqrt(w2);
    double sv = sqrt(1.0 - w2);
    q[0] = axis[0] * sv; q[1] = axis[1] * sv; q[2] = axis[2] * sv;
  } else {
    double axis[3];
    cross3(v0, v1, axis);
    double s = sqrt((1.0 + c) * 2.0);
    double invs = 1.0 / s;
    q[0] = axis[0] * invs; q[1] = axis[1] * invs; q[2] = axis[2] * invs;
    qw = s * 0.5;
  }
  quat_to_R(qw, q[0], q[1], q[2], Rout);
}

/* utils.cc:82-89: AngleAxisd(|w| dt, w.normalized()) -> quaternion -> matrix.
 * normalized() leaves a zero vector unchanged, giving the identity. */
void orc_w_to_R(const double w[3], double dt, double Rout[9]) {
  double n2 = dot3(w, w);
  double nrm = sqrt(n2);
  double ax[3] = {w[0], w[1], w[2]};
  if (n2 > 0) { ax[0] = w[0] / nrm; ax[1] = w[1] / nrm; ax[2] = w[2] / nrm; }
  double half = 0.5 * (nrm * dt);
  double s = sin(half), c = cos(half);
  quat_to_R(c, s * ax[0], s * ax[1], s * ax[2], Rout);
}

/* joints.cc:3-35 and contact.cc:14-117 (FrictionModel::BOX, contact.h:42). */
void orc_assemble(int n, const double *p, const double *R, int m,
                  const int32_t *kind, const int32_t *body0,
                  const int32_t *body1, const double *data, double *J0,
                  double *J1, uint8_t *is_eq, double *lo, double *hi,
                  double *err) {
  (void)n;
  for (int i = 0; i < m; ++i) {
    const double *d = data + 7 * i;
    double *j0 = J0 + 18 * i, *j1 = J1 + 18 * i;
    const int b0 = body0[i], b1 = body1[i];
    for (int k = 0; k < 18; ++k) { j0[k] = 0.0; j1[k] = 0.0; }
    if (kind[i] == ORC_JOINT_BALL) {
      /* joints.cc:13-35: j0 = [I, -[R0 c0]x], j1 = [-I, [R1 c1]x] or 0. */
      double rc0[3], cm[9];
      mat3_vec(R + 9 * b0, d, rc0);
      cross_mat(rc0, cm);
      for (int r = 0; r < 3; ++r) {
        j0[6 * r + r] = 1.0;
        for (int c = 0; c < 3; ++c) j0[6 * r + 3 + c] = -1.0 * cm[3 * r + c];
      }
      double e[3];
      for (int k = 0; k < 3; ++k) e[k] = p[3 * b0 + k] + rc0[k];
      if (b1 >= 0) {
        double rc1[3];
        mat3_vec(R + 9 * b1, d + 3, rc1);
        cross_mat(rc1, cm);
        for (int r = 0; r < 3; ++r) {
          j1[6 * r + r] = -1.0;
          for (int c = 0; c < 3; ++c) j1[6 * r + 3 + c] = cm[3 * r + c];
        }
        /* joints.cc:8: p0 + R0 c0 - p1 - R1 c1, left to right */
        for (int k = 0; k < 3; ++k) e[k] = (e[k] - p[3 * b1 + k]) - rc1[k];
      } else {
        for (int k = 0; k < 3; ++k) e[k] = e[k] - d[3 + k]; /* joints.cc:6 */
      }
      for (int k = 0; k < 3; ++k) {
        err[3 * i + k] = e[k];
        is_eq[3 * i + k] = 1;
        lo[3 * i + k] = 0.0;
        hi[3 * i + k] = 0.0;
      }
    } else {
      /* contact.cc:38-117 */
      const double zaxis[3] = {0, 0, 1};
      double Rn[9];
      orc_align_vectors(d + 3, zaxis, Rn);
      if (b0 >= 0) {
        double rel[3], cm[9], rw[9];
        for (int k = 0; k < 3; ++k) rel[k] = d[k] - p[3 * b0 + k];
        cross_mat(rel, cm);
        mat3_mul(Rn, cm, rw); /* R * J_w0, J_w0 = [pos-p0]x */
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) {
            j0[6 * r + c] = -Rn[3 * r + c]; /* R * (-I) */
            j0[6 * r + 3 + c] = rw[3 * r + c];
          }
      }
      if (b1 >= 0) {
        double rel[3], cm[9], rw[9];
        for (int k = 0; k < 3; ++k) rel[k] = d[k] - p[3 * b1 + k];
        cross_mat(rel, cm);
        for (int k = 0; k < 9; ++k) cm[k] = -1.0 * cm[k]; /* J_w1 = -[pos-p1]x */
        mat3_mul(Rn, cm, rw);
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) {
            j1[6 * r + c] = Rn[3 * r + c];
            j1[6 * r + 3 + c] = rw[3 * r + c];
          }
      }
      err[3 * i + 0] = 0.0;
      err[3 * i + 1] = 0.0;
      err[3 * i + 2] = -d[6]; /* contact.cc:18-20 */
      for (int k = 0; k < 3; ++k) is_eq[3 * i + k] = 0;
      lo[3 * i + 0] = -1.0; lo[3 * i + 1] = -1.0; lo[3 * i + 2] = 0.0; /* :109 */
      hi[3 * i + 0] = 1.0;  hi[3 * i + 1] = 1.0;  hi[3 * i + 2] = INFINITY;
    }
  }
}

/* ensembles.cc:202-212 */
void orc_minv_blocks(int n, const double *R, const double *mass,
                     const double *I_body, double *Minv) {
  for (int b = 0; b < n; ++b) {
    double *M = Minv + 36 * b;
    for (int k = 0; k < 36; ++k) M[k] = 0.0;
    double im = 1.0 / mass[b];
    M[0] = im; M[7] = im; M[14] = im;
    double RI[9], Rt[9], Ig[9], Iinv[9];
    mat3_mul(R + 9 * b, I_body + 9 * b, RI); /* body.h:58: R I R^T */
    mat3_T(R + 9 * b, Rt);
    mat3_mul(RI, Rt, Ig);
    mat3_inv(Ig, Iinv);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) M[6 * (3 + r) + 3 + c] = Iinv[3 * r + c];
  }
}

/* ensembles.cc:214-222 */
void orc_external_force(int n, const double *R, const double *w,
                        const double *mass, const double *I_body,
                        double *f_ext) {
  for (int b = 0; b < n; ++b) {
    double RI[9], Rt[9], Ig[9], cm[9], t[9], tq[3];
    mat3_mul(R + 9 * b, I_body + 9 * b, RI);
    mat3_T(R + 9 * b, Rt);
    mat3_mul(RI, Rt, Ig);
    cross_mat(w + 3 * b, cm);
    for (int k = 0; k < 9; ++k) cm[k] = -1.0 * cm[k];
    mat3_mul(cm, Ig, t);
    mat3_vec(t, w + 3 * b, tq);
    for (int k = 0; k < 3; ++k) {
      f_ext[6 * b + k] = mass[b] * kGravity[k];
      f_ext[6 * b + 3 + k] = tq[k];
    }
  }
}

static inline double dot6(const double *a, const double *b) {
  return ((((a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]) + a[3] * b[3]) +
          a[4] * b[4]) + a[5] * b[5];
}

/* ensembles.cc:569-570. u_b = v_b/dt + Minv_b f_b. */
void orc_ode_rhs(int n, const double *v, const double *w, const double *Minv,
                 const double *f_ext, int m, const int32_t *body0,
                 const int32_t *body1, const double *J0, const double *J1,
                 const double *err, double dt, double erp, double *rhs) {
  (void)n;
  const double k = -erp / dt / dt;
  for (int i = 0; i < m; ++i) {
    double u0[6] = {0}, u1[6] = {0};
    const int b[2] = {body0[i], body1[i]};
    double *u[2] = {u0, u1};
    for (int s = 0; s < 2; ++s) {
      if (b[s] < 0) continue;
      const double *M = Minv + 36 * b[s];
      const double *f = f_ext + 6 * b[s];
      for (int r = 0; r < 6; ++r) {
        double vel = r < 3 ? v[3 * b[s] + r] : w[3 * b[s] + r - 3];
        u[s][r] = vel / dt + dot6(M + 6 * r, f);
      }
    }
    for (int r = 0; r < 3; ++r) {
      double ju = dot6(J0 + 18 * i + 6 * r, u0) + dot6(J1 + 18 * i + 6 * r, u1);
      rhs[3 * i + r] = k * err[3 * i + r] - ju;
    }
  }
}

/* ensembles.cc:535 and 572: v_new = v + dt * Minv (f_ext + J^T lambda). */
void orc_velocity_update(int n, const double *v, const double *w,
                         const double *Minv, const double *f_ext, int m,
                         const int32_t *body0, const int32_t *body1,
                         const double *J0, const double *J1,
                         const double *lambda, double dt, double *vnew) {
  /* g = f_ext + J^T lambda accumulated in constraint order */
  for (int b = 0; b < n; ++b)
    for (int k = 0; k < 6; ++k) vnew[6 * b + k] = f_ext[6 * b + k];
  for (int i = 0; i < m; ++i) {
    const int b[2] = {body0[i], body1[i]};
    const double *J[2] = {J0 + 18 * i, J1 + 18 * i};
    for (int s = 0; s < 2; ++s) {
      if (b[s] < 0) continue;
      for (int c = 0; c < 6; ++c) {
        double t = (J[s][c] * lambda[3 * i] + J[s][6 + c] * lambda[3 * i + 1]) +
                   J[s][12 + c] * lambda[3 * i + 2];
        vnew[6 * b[s] + c] += t;
      }
    }
  }
  for (int b = 0; b < n; ++b) {
    double g[6], out[6];
    for (int k = 0; k < 6; ++k) g[k] = vnew[6 * b + k];
    for (int r = 0; r < 6; ++r) {
      double vel = r < 3 ? v[3 * b + r] : w[3 * b + r - 3];
      out[r] = vel + dt * dot6(Minv + 36 * b + 6 * r, g);
    }
    for (int k = 0; k < 6; ++k) vnew[6 * b + k] = out[k];
  }
}

/* ensembles.cc:577-591 */
void orc_position_update(int n, double *p, double *R, const double *v6_old,
                         const double *v6_new, double dt) {
  for (int b = 0; b < n; ++b) {
    double wm[3], Q[9];
    for (int k = 0; k < 3; ++k) {
      double vm = (v6_old[6 * b + k] + v6_new[6 * b + k]) / 2.0;
      p[3 * b + k] = p[3 * b + k] + dt * vm;
      wm[k] = (v6_old[6 * b + 3 + k] + v6_new[6 * b + 3 + k]) / 2.0;
    }
    orc_w_to_R(wm, dt, Q);
    mat3_mul(Q, R + 9 * b, R + 9 * b);
  }
}

/* ensembles.cc:668-707 (Chain::Chain, InitLinks, InitJoints, SetAnchor) and
 * body.h:25-34, body.cc:19-36 (box inertia, side 0.3, m = 1). */
void orc_chain(int num_links, const double anchor[3], double *p, double *R,
               double *v, double *w, double *mass, double *I_body,
               int32_t *kind, int32_t *body0, int32_t *body1, double *data) {
  /* q = AngleAxis(0.9553.., Z) * AngleAxis(pi/4, X) */
  const double az = 0.95531661812451, ax = M_PI / 4;
  double qz_w = cos(az / 2), qz_z = sin(az / 2);
  double qx_w = cos(ax / 2), qx_x = sin(ax / 2);
  /* (w1, 0,0,z1) * (w2, x2,0,0) */
  double qw = qz_w * qx_w;
  double qx = qz_w * qx_x;
  double qy = qz_z * qx_x;
  double qzz = qz_z * qx_w;
  double Rm[9];
  quat_to_R(qw, qx, qy, qzz, Rm);
  const double side = 0.3;
  const double I = 1.0 / 12 * (side * side + side * side);
  for (int i = 0; i < num_links; ++i) {
    p[3 * i] = sqrt(3.0) * 0.3 * i + anchor[0];
    p[3 * i + 1] = 0 + anchor[1];
    p[3 * i + 2] = 0 + anchor[2];
    memcpy(R + 9 * i, Rm, sizeof Rm);
    for (int k = 0; k < 3; ++k) { v[3 * i + k] = 0; w[3 * i + k] = 0; }
    mass[i] = 1.0;
    for (int k = 0; k < 9; ++k) I_body[9 * i + k] = 0;
    I_body[9 * i] = I; I_body[9 * i + 4] = I; I_body[9 * i + 8] = I;
  }
  const double c1[3] = {0.15, -0.15, 0.15}, c2[3] = {-0.15, 0.15, -0.15};
  int j = 0;
  for (int i = 0; i < num_links - 1; ++i, ++j) {
    kind[j] = ORC_JOINT_BALL; body0[j] = i; body1[j] = i + 1;
    for (int k = 0; k < 3; ++k) { data[7 * j + k] = c1[k]; data[7 * j + 3 + k] = c2[k]; }
    data[7 * j + 6] = 0;
  }
  /* SetAnchor: joint (component 0, c0 = 0) to the world point p0 */
  kind[j] = ORC_JOINT_BALL; body0[j] = 0; body1[j] = -1;
  for (int k = 0; k < 3; ++k) { data[7 * j + k] = 0; data[7 * j + 3 + k] = p[k]; }
  data[7 * j + 6] = 0;
}
