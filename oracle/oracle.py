"""ctypes binding of the CPU ORACLE (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, bench.py's cpu_baseline leg
and __graft_entry__.smoke().  The product package (eggshell_amd/) must never
import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

JOINT_BALL, CONTACT_BOX = 0, 1
JACOBI, GAUSS_SEIDEL, SOR = 0, 1, 2


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_lit_residual.restype = C.c_double
    return _LIB


_LIB_PARITY = None
BUILD_FLAGS = {"parity": "gcc -O2 -DNDEBUG -march=x86-64-v3 -ffp-contract=off (the checker's build: roundings = the HIP kernels')",
               "refflags": "gcc -O2 -DNDEBUG -march=x86-64-v3, GCC's default contraction (-ffp-contract=fast): the reference's "
                           "flags (common.mk:159-160, 187-188) with x86-64-v3 standing in for -march=native"}


class timing_build:
    """`with orc.timing_build() as flags:` -- inside the block every oracle call runs in liboracle_refflags.so, the
    same sources built with the reference's flags (oracle/Makefile).  For TIMING the CPU baseline only: results of
    that build are not used to check anything."""

    def __enter__(self):
        global _LIB, _LIB_PARITY
        so = os.path.join(_HERE, "liboracle_refflags.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle_refflags.so"])
        lib()
        _LIB_PARITY = _LIB
        _LIB = C.CDLL(so)
        _LIB.orc_lit_residual.restype = C.c_double
        return BUILD_FLAGS["refflags"]

    def __exit__(self, *exc):
        global _LIB
        _LIB = _LIB_PARITY
        return False


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class System(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("Minv", C.c_void_p),
                ("body0", C.c_void_p), ("body1", C.c_void_p),
                ("J0", C.c_void_p), ("J1", C.c_void_p), ("is_eq", C.c_void_p),
                ("lo", C.c_void_p), ("hi", C.c_void_p)]


class Sys:
    """Flat constraint system (the arrays sparse::*Iteration would consume)."""

    def __init__(self, Minv, body0, body1, J0, J1, is_eq, lo, hi):
        self.Minv = _f64(Minv).reshape(-1, 36)
        self.body0 = _i32(body0)
        self.body1 = _i32(body1)
        self.J0 = _f64(J0).reshape(-1, 18)
        self.J1 = _f64(J1).reshape(-1, 18)
        self.is_eq = np.ascontiguousarray(is_eq, dtype=np.uint8)
        self.lo = _f64(lo)
        self.hi = _f64(hi)
        self.n = self.Minv.shape[0]
        self.m = self.body0.shape[0]
        self.c = System(self.n, self.m, _p(self.Minv), _p(self.body0),
                        _p(self.body1), _p(self.J0), _p(self.J1),
                        _p(self.is_eq), _p(self.lo), _p(self.hi))

    @property
    def ref(self):
        return C.byref(self.c)


def assemble(p, R, kind, body0, body1, data):
    p, R, data = _f64(p), _f64(R), _f64(data)
    kind, body0, body1 = _i32(kind), _i32(body0), _i32(body1)
    m = kind.shape[0]
    J0 = np.zeros((m, 18)); J1 = np.zeros((m, 18))
    is_eq = np.zeros(3 * m, np.uint8)
    lo = np.zeros(3 * m); hi = np.zeros(3 * m); err = np.zeros(3 * m)
    lib().orc_assemble(C.c_int(p.shape[0]), _p(p), _p(R), C.c_int(m), _p(kind),
                       _p(body0), _p(body1), _p(data), _p(J0), _p(J1),
                       _p(is_eq), _p(lo), _p(hi), _p(err))
    return J0, J1, is_eq, lo, hi, err


def minv_blocks(R, mass, I_body):
    R, mass, I_body = _f64(R), _f64(mass), _f64(I_body)
    n = mass.shape[0]
    out = np.zeros((n, 36))
    lib().orc_minv_blocks(C.c_int(n), _p(R), _p(mass), _p(I_body), _p(out))
    return out


def external_force(R, w, mass, I_body):
    R, w, mass, I_body = _f64(R), _f64(w), _f64(mass), _f64(I_body)
    n = mass.shape[0]
    out = np.zeros((n, 6))
    lib().orc_external_force(C.c_int(n), _p(R), _p(w), _p(mass), _p(I_body), _p(out))
    return out


def ode_rhs(v, w, Minv, f_ext, body0, body1, J0, J1, err, dt, erp):
    v, w, Minv, f_ext = _f64(v), _f64(w), _f64(Minv), _f64(f_ext)
    body0, body1, J0, J1, err = _i32(body0), _i32(body1), _f64(J0), _f64(J1), _f64(err)
    m = body0.shape[0]
    rhs = np.zeros(3 * m)
    lib().orc_ode_rhs(C.c_int(v.shape[0]), _p(v), _p(w), _p(Minv), _p(f_ext),
                      C.c_int(m), _p(body0), _p(body1), _p(J0), _p(J1), _p(err),
                      C.c_double(dt), C.c_double(erp), _p(rhs))
    return rhs


def velocity_update(v, w, Minv, f_ext, body0, body1, J0, J1, lam, dt):
    v, w, Minv, f_ext = _f64(v), _f64(w), _f64(Minv), _f64(f_ext)
    body0, body1, J0, J1, lam = _i32(body0), _i32(body1), _f64(J0), _f64(J1), _f64(lam)
    n = v.shape[0]
    out = np.zeros((n, 6))
    lib().orc_velocity_update(C.c_int(n), _p(v), _p(w), _p(Minv), _p(f_ext),
                              C.c_int(body0.shape[0]), _p(body0), _p(body1),
                              _p(J0), _p(J1), _p(lam), C.c_double(dt), _p(out))
    return out


def position_update(p, R, v6_old, v6_new, dt):
    p, R = _f64(p).copy(), _f64(R).copy()
    v6_old, v6_new = _f64(v6_old), _f64(v6_new)
    lib().orc_position_update(C.c_int(p.shape[0]), _p(p), _p(R), _p(v6_old),
                              _p(v6_new), C.c_double(dt))
    return p, R


def align_vectors(a, b):
    a, b = _f64(a), _f64(b)
    out = np.zeros(9)
    lib().orc_align_vectors(_p(a), _p(b), _p(out))
    return out.reshape(3, 3)


def w_to_R(w, dt):
    w = _f64(w)
    out = np.zeros(9)
    lib().orc_w_to_R(_p(w), C.c_double(dt), _p(out))
    return out.reshape(3, 3)


def chain(num_links, anchor=(0.0, 0.0, 2.0)):
    n = num_links
    anchor = _f64(anchor)
    p = np.zeros((n, 3)); R = np.zeros((n, 9)); v = np.zeros((n, 3)); w = np.zeros((n, 3))
    mass = np.zeros(n); I_body = np.zeros((n, 9))
    kind = np.zeros(n, np.int32); b0 = np.zeros(n, np.int32); b1 = np.zeros(n, np.int32)
    data = np.zeros((n, 7))
    lib().orc_chain(C.c_int(n), _p(anchor), _p(p), _p(R), _p(v), _p(w), _p(mass),
                    _p(I_body), _p(kind), _p(b0), _p(b1), _p(data))
    return dict(p=p, R=R, v=v, w=w, mass=mass, I_body=I_body, kind=kind,
                body0=b0, body1=b1, data=data)


# ---- literal O(m^2) -------------------------------------------------------
def _vec_out(fn, s, x, *scalars):
    x = _f64(x)
    out = np.zeros(3 * s.m)
    fn(s.ref, _p(x), *[C.c_double(v) for v in scalars], _p(out))
    return out


def lit_Lx(s, x): return _vec_out(lib().orc_lit_Lx, s, x)
def lit_Ux(s, x): return _vec_out(lib().orc_lit_Ux, s, x)
def lit_Dx(s, x, eps=0.0, scale=1.0): return _vec_out(lib().orc_lit_Dx, s, x, eps, scale)
def lit_JMJtX(s, x, eps=0.0): return _vec_out(lib().orc_lit_JMJtX, s, x, eps)
def lit_solve_diag(s, rhs, eps=0.0, scale=1.0): return _vec_out(lib().orc_lit_solve_diag, s, rhs, eps, scale)


def lit_solve_lower(s, rhs, eps=0.0, scale=1.0, quirks=0):
    rhs = _f64(rhs); out = np.zeros(3 * s.m)
    lib().orc_lit_solve_lower(s.ref, _p(rhs), C.c_double(eps), C.c_double(scale), C.c_int(quirks), _p(out))
    return out


def lit_solve_upper(s, rhs, eps=0.0, scale=1.0, quirks=0):
    rhs = _f64(rhs); out = np.zeros(3 * s.m)
    lib().orc_lit_solve_upper(s.ref, _p(rhs), C.c_double(eps), C.c_double(scale), C.c_int(quirks), _p(out))
    return out


def lit_residual(s, rhs, x, cfm):
    rhs, x = _f64(rhs), _f64(x)
    return lib().orc_lit_residual(s.ref, _p(rhs), _p(x), C.c_double(cfm))


def lit_iterate(s, rhs, cfm, method, omega=1.5, max_iters=500, tol=1e-9, quirks=0):
    rhs = _f64(rhs)
    x = np.zeros(3 * s.m); res = C.c_double(0)
    it = lib().orc_lit_iterate(s.ref, _p(rhs), C.c_double(cfm), C.c_int(method),
                               C.c_double(omega), C.c_int(max_iters), C.c_double(tol),
                               C.c_int(quirks), _p(x), C.byref(res))
    return x, it, res.value


def dense_JMJt(s, eps=0.0):
    A = np.zeros((3 * s.m, 3 * s.m))
    lib().orc_dense_JMJt(s.ref, C.c_double(eps), _p(A))
    return A


# ---- fast O(nnz) ----------------------------------------------------------
def fast_iterate(s, rhs, cfm, method, omega=1.5, max_iters=500, tol=1e-9, check_every=1):
    rhs = _f64(rhs)
    x = np.zeros(3 * s.m); a = np.zeros((s.n, 6)); res = C.c_double(0)
    it = lib().orc_fast_iterate_f64(s.ref, _p(rhs), C.c_double(cfm), C.c_int(method),
                                    C.c_double(omega), C.c_int(max_iters), C.c_double(tol),
                                    C.c_int(check_every), _p(x), _p(a), C.byref(res))
    return x, a, it, res.value


def fast_wres(s, rhs, cfm, x, a):
    """w = A x - rhs element by element from x and the accumulators fast_iterate returned."""
    rhs, x, a = _f64(rhs), _f64(x), _f64(a)
    w = np.zeros(3 * s.m)
    lib().orc_fast_wres_f64(s.ref, _p(rhs), C.c_double(cfm), _p(x), _p(a), _p(w))
    return w


def fast_iterate_f32(s, rhs, cfm, method, omega=1.5, max_iters=500, tol=0.0, check_every=1):
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    Minv, J0, J1, lo, hi, rhs = f(s.Minv), f(s.J0), f(s.J1), f(s.lo), f(s.hi), f(rhs)
    x = np.zeros(3 * s.m, np.float32); a = np.zeros((s.n, 6), np.float32); res = C.c_float(0)
    it = lib().orc_fast_iterate_f32(C.c_int(s.n), C.c_int(s.m), _p(Minv), _p(s.body0), _p(s.body1),
                                    _p(J0), _p(J1), _p(s.is_eq), _p(lo), _p(hi), _p(rhs),
                                    C.c_float(cfm), C.c_int(method), C.c_float(omega),
                                    C.c_int(max_iters), C.c_float(tol), C.c_int(check_every),
                                    _p(x), _p(a), C.byref(res))
    return x, a, it, res.value


MV_LOWER, MV_UPPER, MV_DIAG, MV_FULL = 1, 2, 4, 8


def fast_matvec(s, x, parts=MV_FULL, eps=0.0, scale=1.0):
    """O(nnz) twin of CalculateSparse{Lx,Ux,Dx,...,JMJtX} in the HIP kernels' operation order."""
    x = _f64(x)
    y = np.zeros(3 * s.m)
    lib().orc_fast_matvec_f64(s.ref, _p(x), C.c_int(parts), C.c_double(eps), C.c_double(scale), _p(y))
    return y


def fast_matvec_f32(s, x, parts=MV_FULL, eps=0.0, scale=1.0):
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    Minv, J0, J1, x = f(s.Minv), f(s.J0), f(s.J1), f(x)
    y = np.zeros(3 * s.m, np.float32)
    lib().orc_fast_matvec_f32(C.c_int(s.n), C.c_int(s.m), _p(Minv), _p(s.body0), _p(s.body1), _p(J0), _p(J1),
                              _p(x), C.c_int(parts), C.c_float(eps), C.c_float(scale), _p(y))
    return y


# ---- dense LCP ------------------------------------------------------------
def murty(A, b, lo=None, hi=None):
    A, b = _f64(A), _f64(b)
    dim = b.shape[0]
    lo = np.zeros(dim) if lo is None else _f64(np.broadcast_to(lo, (dim,)))
    hi = np.full(dim, np.inf) if hi is None else _f64(np.broadcast_to(hi, (dim,)))
    x = np.zeros(dim); w = np.zeros(dim); piv = C.c_int(0)
    ok = lib().orc_murty(C.c_int(dim), _p(A), _p(b), _p(lo), _p(hi), _p(x), _p(w), C.byref(piv))
    return bool(ok), x, w, piv.value


def check_murty(A, b, x, w, S, err=0.0, lo=None, hi=None):
    A, b, x, w = _f64(A), _f64(b), _f64(x), _f64(w)
    dim = b.shape[0]
    lo = np.zeros(dim) if lo is None else _f64(lo)
    hi = np.full(dim, np.inf) if hi is None else _f64(hi)
    S = np.ascontiguousarray(S, dtype=np.uint8).copy()
    Cc = lo.copy()
    ok = lib().orc_check_murty(C.c_int(dim), _p(A), _p(b), _p(x), _p(w), _p(S), _p(Cc),
                               _p(lo), _p(hi), C.c_double(err))
    return bool(ok), S


def mixed_constraints(A, b, Ceq, lo, hi, use_bounds=0):
    A, b, lo, hi = _f64(A), _f64(b), _f64(lo), _f64(hi)
    Ceq = np.ascontiguousarray(Ceq, dtype=np.uint8)
    dim = b.shape[0]
    x = np.zeros(dim); w = np.zeros(dim); piv = C.c_int(0)
    ok = lib().orc_mixed_constraints(C.c_int(dim), _p(A), _p(b), _p(Ceq), _p(lo), _p(hi),
                                     C.c_int(use_bounds), _p(x), _p(w), C.byref(piv))
    return bool(ok), x, w, piv.value


# ---- iterations on an explicit dense matrix (dense_iter.c) --------------------
def dense_iterate(A, b, method, Ceq=None, lo=None, hi=None, omega=1.5, max_iters=500, tol=1e-9):
    """sparse::{Jacobi,GaussSeidel,SOR}Iteration(A, b[, C, x_lo, x_hi]) (sparse_iterations.cc:72-144): x, sweeps, residual."""
    A, b = _f64(A), _f64(b)
    n = b.shape[0]
    Ceq = np.ones(n, np.uint8) if Ceq is None else np.ascontiguousarray(Ceq, dtype=np.uint8)
    lo = np.zeros(n) if lo is None else _f64(lo)
    hi = np.zeros(n) if hi is None else _f64(hi)
    x = np.zeros(n); res = C.c_double(0)
    fn = lib().orc_dense_iterate
    it = fn(C.c_int(n), _p(A), _p(b), _p(Ceq), _p(lo), _p(hi), C.c_int(method), C.c_double(omega), C.c_int(max_iters),
            C.c_double(tol), _p(x), C.byref(res))
    return x, int(it), res.value


# ---- toolkit/lcp.cc: incremental-factor box LCP (lcp_toolkit.c) --------------
def tk_cholesky(A):
    L = _f64(A).copy()
    n = L.shape[0]
    rc = lib().otk_cholesky(_p(L), C.c_int(n))
    return rc == 0, L


def tk_lsolve(L, m, b, transpose=False):
    L, x = _f64(L), _f64(b).copy()
    (lib().otk_ltsolve if transpose else lib().otk_lsolve)(_p(L), C.c_int(L.shape[0]), C.c_int(m), _p(x))
    return x


def tk_add_cholesky_row(A, m, L):
    A, L = _f64(A), _f64(L).copy()
    rc = lib().otk_add_cholesky_row(_p(A), C.c_int(A.shape[0]), C.c_int(m), _p(L))
    return rc == 0, L


def tk_swap_cholesky_rows(A, i, m, L):
    A, L = _f64(A), _f64(L).copy()
    n = A.shape[0]
    work = np.zeros(2 * n)
    rc = lib().otk_swap_cholesky_rows(_p(A), C.c_int(n), C.c_int(i), C.c_int(m), _p(L), _p(work))
    return rc == 0, L


def tk_swap_rows_and_columns(A, i, j, perm):
    A = _f64(A).copy()
    perm = np.ascontiguousarray(perm, dtype=np.int32).copy()
    lib().otk_swap_rows_and_columns(_p(A), C.c_int(A.shape[0]), C.c_int(i), C.c_int(j), _p(perm))
    return A, perm


def tk_box_dantzig(A, b, lo, hi):
    """SolveLCP_BoxDantzig: returns ok, x, w, A permuted in place (lower triangle), perm, pivots."""
    A = _f64(A).copy()
    b, lo, hi = _f64(b), _f64(lo), _f64(hi)
    n = b.shape[0]
    x = np.zeros(n); w = np.zeros(n); perm = np.zeros(n, np.int32); piv = C.c_int(0)
    ok = lib().otk_box_dantzig(C.c_int(n), _p(A), _p(b), _p(lo), _p(hi), _p(x), _p(w), _p(perm), C.byref(piv))
    return ok == 1, x, w, A, perm, piv.value


def tk_box_murty(A, b, lo, hi, max_iterations=2**31 - 1):
    """SolveLCP_BoxMurty on a LinearReducer: returns ok, x, w, A permuted in place, perm, iterations."""
    A = _f64(A).copy()
    b, lo, hi = _f64(b), _f64(lo), _f64(hi)
    n = b.shape[0]
    x = np.zeros(n); w = np.zeros(n); perm = np.zeros(n, np.int32); it = C.c_int(0)
    ok = lib().otk_box_murty(C.c_int(n), _p(A), _p(b), _p(lo), _p(hi), C.c_int(max_iterations), _p(x), _p(w), _p(perm), C.byref(it))
    return ok == 1, x, w, A, perm, it.value


def tk_box_schur(A, b, lo, hi, algorithm=0, max_iterations=2**31 - 1, nub=-1, q6=True):
    """SolveLCP_BoxSchur (toolkit/lcp.cc:627-747): returns ok, x, w, A permuted in place, perm of the partition,
    nub (test_nub_from_SolveLCP_BoxSchur), inner iterations."""
    A = _f64(A).copy()
    b, lo, hi = _f64(b), _f64(lo), _f64(hi)
    n = b.shape[0]
    x = np.zeros(n); w = np.zeros(n); perm = np.zeros(n, np.int32); it = C.c_int(0); nub_out = C.c_int(0)
    ok = lib().otk_box_schur(C.c_int(n), _p(A), _p(b), _p(lo), _p(hi), C.c_int(algorithm), C.c_int(max_iterations),
                             C.c_int(nub), C.c_int(1 if q6 else 0), _p(x), _p(w), _p(perm), C.byref(nub_out), C.byref(it))
    return ok == 1, x, w, A, perm, nub_out.value, it.value


# ---- collision ------------------------------------------------------------
def collide_box_ground(c, R, side=(0.3, 0.3, 0.3)):
    c, R, side = _f64(c), _f64(R), _f64(side)
    out = np.zeros((8, 7))
    n = lib().orc_collide_box_ground(_p(c), _p(R), _p(side), _p(out))
    return out[:n].copy()


def line_closest_approach(pa, ua, pb, ub):
    a, b = C.c_double(0), C.c_double(0)
    lib().orc_line_closest_approach(_p(_f64(pa)), _p(_f64(ua)), _p(_f64(pb)), _p(_f64(ub)), C.byref(a), C.byref(b))
    return a.value, b.value


def clip_polygon(poly, normal, d):
    poly = _f64(poly)
    out = np.zeros((64, 2))
    k = lib().orc_clip_polygon(_p(poly), C.c_int(poly.shape[0]), _p(_f64(normal)), C.c_double(d), _p(out))
    return out[:k].copy()


def box_rectangle(bc, bR, bhalf, rc, rR, rhalf):
    out = np.zeros((32, 2))
    k = lib().orc_box_rectangle(_p(_f64(bc)), _p(_f64(bR)), _p(_f64(bhalf)), _p(_f64(rc)), _p(_f64(rR)),
                                _p(_f64(rhalf)), _p(out))
    return out[:k].copy()


def collide_boxes_info(c1, R1, c2, R2, s1=(0.3, 0.3, 0.3), s2=(0.3, 0.3, 0.3)):
    """CollideBoxes with its CollisionInfo: (contacts[n][7], code, separating_axis[3], depth)."""
    c1, R1, c2, R2, s1, s2 = _f64(c1), _f64(R1), _f64(c2), _f64(R2), _f64(s1), _f64(s2)
    out = np.zeros((16, 7)); code = C.c_int(0); info = np.zeros(4)
    n = lib().orc_collide_boxes_info(_p(c1), _p(R1), _p(s1), _p(c2), _p(R2), _p(s2), _p(out),
                                     C.c_int(16), C.byref(code), _p(info))
    return out[:n].copy(), code.value, info[:3].copy(), float(info[3])


def collide_boxes(c1, R1, c2, R2, s1=(0.3, 0.3, 0.3), s2=(0.3, 0.3, 0.3)):
    c1, R1, c2, R2, s1, s2 = _f64(c1), _f64(R1), _f64(c2), _f64(R2), _f64(s1), _f64(s2)
    out = np.zeros((16, 7)); code = C.c_int(0)
    n = lib().orc_collide_boxes(_p(c1), _p(R1), _p(s1), _p(c2), _p(R2), _p(s2), _p(out),
                                C.c_int(16), C.byref(code))
    return out[:n].copy(), code.value
