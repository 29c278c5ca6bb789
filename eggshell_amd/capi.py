"""ctypes binding of libeggshell_amd.so (the C ABI in include/eggshell_amd.h).

Plumbing only: numpy arrays in, numpy arrays out.  There is NO CPU fallback;
if the library is missing or no gfx950 device is usable the calls raise.
"""
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libeggshell_amd.so")

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_STALL, ERR_UNSUPPORTED, ERR_LCP_FAILED, ERR_INTERNAL = range(8)
JACOBI, GAUSS_SEIDEL, SOR = 0, 1, 2
F64, F32 = 0, 1
JOINT_BALL, CONTACT_BOX = 0, 1
MV_LOWER, MV_UPPER, MV_DIAG, MV_FULL = 1, 2, 4, 8
SCHED_QUAD, SCHED_ISO, SCHED_QUAD_PATCHES, SCHED_LANE_PATCHES, SCHED_ALL_GLOBAL, SCHED_STATIC, SCHED_LEAN = 1, 2, 4, 8, 16, 32, 64

# every symbol include/eggshell_amd.h declares
EXPORTS = [
    "egs_default_params", "egs_context_create", "egs_context_destroy", "egs_last_error",
    "egs_context_synchronize", "egs_timer_start", "egs_timer_stop", "egs_kernel_time",
    "egs_solve_blocks", "egs_problem_create", "egs_problem_create_batch", "egs_problem_destroy", "egs_problem_set_blocks",
    "egs_problem_solve", "egs_problem_get_lambda", "egs_problem_get_accumulators",
    "egs_problem_set_state", "egs_problem_set_mass", "egs_problem_set_constraints", "egs_problem_assemble",
    "egs_problem_step", "egs_problem_get_blocks", "egs_problem_get_velocity",
    "egs_problem_advance", "egs_problem_get_state",
    "egs_problem_get_stats", "egs_mixed_constraints_solve", "egs_debug_plan", "egs_debug_plan_slots",
    "egs_update_contacts", "egs_update_contacts_joints", "egs_world_create", "egs_world_destroy", "egs_world_set_bodies",
    "egs_world_set_joints", "egs_world_step", "egs_world_get_bodies", "egs_world_get_contacts",
    "egs_world_get_lambda", "egs_world_info",
    "egs_problem_matvec", "egs_problem_get_matvec", "egs_problem_get_wres", "egs_matvec_blocks",
    "egs_debug_matvec_plan", "egs_debug_choose_oversize_schedule", "egs_debug_plan_timetable", "egs_box_lcp_dantzig", "egs_box_lcp_murty",
    "egs_box_lcp_batch", "egs_box_lcp_schur", "egs_dense_condition", "egs_dense_iterate", "egs_debug_plan_patches", "egs_problem_debug_trace",
    "egs_mixed_constraints_solve_limits", "egs_problem_dense_system", "egs_problem_dense_condition", "egs_problem_step_dense",
]


class SolveParams(C.Structure):
    _fields_ = [("method", C.c_int32), ("max_iters", C.c_int32), ("check_every", C.c_int32),
                ("reserved", C.c_int32), ("omega", C.c_double), ("cfm", C.c_double),
                ("tol", C.c_double)]


class SolveStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("status", C.c_int32), ("residual", C.c_double),
                ("n_islands", C.c_int32), ("n_tiles", C.c_int32), ("n_global", C.c_int32),
                ("reserved", C.c_int32), ("schedule", C.c_int32), ("tile_constraints", C.c_int32)]


class EgsError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("eggshell_amd status %d: %s" % (status, msg))
        self.status = status


_lib = None


def load():
    """Load the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                LIB_PATH + " is missing: run __graft_entry__.build() (there is no CPU fallback)")
        _lib = C.CDLL(LIB_PATH)
        _lib.egs_last_error.restype = C.c_char_p
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def _u8(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.uint8)


def debug_plan_patches(n_bodies, body0, body1):
    """Host only: the body patches of an oversize island -- n_patches, patch / lane per constraint, remote flags per side."""
    b0, b1 = _i32(body0), _i32(body1)
    m = b0.shape[0]
    npat = C.c_int32(0)
    cp, cl, r0, r1 = (np.zeros(m, np.int32) for _ in range(4))
    st = load().egs_debug_plan_patches(C.c_int32(n_bodies), C.c_int32(m), _p(b0), _p(b1), C.byref(npat), _p(cp), _p(cl), _p(r0), _p(r1))
    if st != OK:
        raise RuntimeError("egs_debug_plan_patches failed: %d" % st)
    return npat.value, cp, cl, r0, r1


def params(method=GAUSS_SEIDEL, max_iters=500, tol=1e-9, cfm=0.0, omega=1.5, check_every=1):
    return SolveParams(method, max_iters, check_every, 0, omega, cfm, tol)


class Context:
    def __init__(self, device=0):
        self.h = C.c_void_p()
        self._children = weakref.WeakSet()   # problems / worlds must be destroyed before their context
        st = load().egs_context_create(C.c_int(device), C.byref(self.h))
        if st != OK:
            self.h = C.c_void_p()
            raise EgsError(st, "egs_context_create failed (no usable gfx950 device?)")

    def check(self, st):
        if st != OK:
            raise EgsError(st, load().egs_last_error(self.h).decode())

    def close(self):
        if self.h:
            for child in list(self._children):
                child.close()
            load().egs_context_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self.check(load().egs_context_synchronize(self.h))

    def timer_start(self):
        self.check(load().egs_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float(0)
        self.check(load().egs_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def kernel_time(self, reset=True):
        s = C.c_double(0); n = C.c_int64(0)
        self.check(load().egs_kernel_time(self.h, C.byref(s), C.byref(n), C.c_int(1 if reset else 0)))
        return s.value, n.value

    def solve_blocks(self, Minv, body0, body1, J0, J1, is_eq, lo, hi, rhs, prm, precision=F64):
        """sparse::{Jacobi,GaussSeidel,SOR}Iteration on flat arrays (entry 1)."""
        Minv, J0, J1, lo, hi, rhs = map(_f64, (Minv, J0, J1, lo, hi, rhs))
        body0, body1, is_eq = _i32(body0), _i32(body1), _u8(is_eq)
        n = Minv.reshape(-1, 36).shape[0]
        m = body0.shape[0]
        x = np.zeros(3 * m)
        st = SolveStats()
        self.check(load().egs_solve_blocks(self.h, C.c_int32(n), _p(Minv), C.c_int32(m), _p(body0),
                                           _p(body1), _p(J0), _p(J1), _p(is_eq), _p(lo), _p(hi), _p(rhs),
                                           C.byref(prm), C.c_int32(precision), _p(x), C.byref(st)))
        return x, st

    def matvec_blocks(self, Minv, body0, body1, J0, J1, x, parts=MV_FULL, eps=0.0, scale=1.0, precision=F64):
        """sparse::CalculateSparse{JMJtX,Lx,Ux,Dx,...} on flat arrays (one-shot form)."""
        Minv, J0, J1, x = map(_f64, (Minv, J0, J1, x))
        body0, body1 = _i32(body0), _i32(body1)
        n, m = Minv.reshape(-1, 36).shape[0], body0.shape[0]
        y = np.zeros(3 * m)
        self.check(load().egs_matvec_blocks(self.h, C.c_int32(n), _p(Minv), C.c_int32(m), _p(body0), _p(body1),
                                            _p(J0), _p(J1), C.c_int32(parts), C.c_double(eps), C.c_double(scale),
                                            C.c_int32(precision), _p(x), _p(y)))
        return y

    def update_contacts(self, pos, R, side=None, max_contacts=None, joints=None):
        """Ensemble::UpdateContacts + contact pruning on the GPU (reference order).
        joints = (body0, body1, data[m][7]) adds the joint-vs-contact pruning."""
        pos, R = _f64(pos), _f64(R)
        if joints is not None:
            jb0, jb1, jd = _i32(joints[0]), _i32(joints[1]), _f64(joints[2])
            n = pos.reshape(-1, 3).shape[0]
            side = _f64(np.tile([0.3, 0.3, 0.3], (n, 1)) if side is None else side)
            cap = int(max_contacts if max_contacts is not None else 64 * n + 64)
            b0 = np.zeros(cap, np.int32); b1 = np.zeros(cap, np.int32); data = np.zeros((cap, 7))
            m = C.c_int32(0)
            self.check(load().egs_update_contacts_joints(self.h, C.c_int32(n), _p(pos), _p(R), _p(side),
                                                         C.c_int32(jb0.shape[0]), _p(jb0), _p(jb1), _p(jd), C.c_int32(cap),
                                                         C.byref(m), _p(b0), _p(b1), _p(data)))
            return b0[:m.value].copy(), b1[:m.value].copy(), data[:m.value].copy()
        n = pos.reshape(-1, 3).shape[0]
        side = _f64(np.tile([0.3, 0.3, 0.3], (n, 1)) if side is None else side)
        cap = int(max_contacts if max_contacts is not None else 64 * n + 64)
        b0 = np.zeros(cap, np.int32); b1 = np.zeros(cap, np.int32); data = np.zeros((cap, 7))
        m = C.c_int32(0)
        self.check(load().egs_update_contacts(self.h, C.c_int32(n), _p(pos), _p(R), _p(side), C.c_int32(cap),
                                              C.byref(m), _p(b0), _p(b1), _p(data)))
        return b0[:m.value].copy(), b1[:m.value].copy(), data[:m.value].copy()

    def mixed_constraints_solve(self, A, b, Ceq, lo, hi, use_bounds=0, max_pivots=0, max_seconds=0.0):
        A, b, lo, hi = map(_f64, (A, b, lo, hi))
        Ceq = _u8(Ceq)
        N = b.shape[0]
        x = np.zeros(N); w = np.zeros(N); ok = C.c_int32(0); piv = C.c_int32(0)
        st = load().egs_mixed_constraints_solve_limits(self.h, C.c_int32(N), _p(A), _p(b), _p(Ceq), _p(lo), _p(hi),
                                                       C.c_int32(use_bounds), C.c_int32(max_pivots), C.c_double(max_seconds),
                                                       _p(x), _p(w), C.byref(ok), C.byref(piv))
        if st not in (OK, ERR_LCP_FAILED):
            self.check(st)
        return bool(ok.value), x, w, piv.value


    def dense_iterate(self, A, b, prm, Ceq=None, lo=None, hi=None):
        """sparse::{Jacobi,GaussSeidel,SOR}Iteration(A, b[, C, x_lo, x_hi]) on an explicit matrix: x, stats."""
        A, b = _f64(A), _f64(b)
        n = b.shape[0]
        x = np.zeros(n)
        st = SolveStats()
        args = (None, None, None) if Ceq is None else (_p(_u8(Ceq)), _p(_f64(lo)), _p(_f64(hi)))
        keep = (Ceq, lo, hi)
        if Ceq is not None:
            c8, l8, h8 = _u8(Ceq), _f64(lo), _f64(hi)
            args = (_p(c8), _p(l8), _p(h8))
        self.check(load().egs_dense_iterate(self.h, C.c_int32(n), _p(A), _p(b), args[0], args[1], args[2], C.byref(prm), _p(x), C.byref(st)))
        return x, st

    def dense_condition(self, A):
        """GetConditionNumber of a symmetric positive definite matrix (utils.cc:256-261) on the device: (estimate, pivot bound)."""
        A = _f64(A)
        est, pb = C.c_double(0), C.c_double(0)
        self.check(load().egs_dense_condition(self.h, C.c_int32(A.shape[0]), _p(A), C.byref(est), C.byref(pb)))
        return est.value, pb.value

    def box_lcp_murty(self, A, b, lo, hi, max_iterations=0):
        """lcp::SolveLCP_BoxMurty on a LinearReducer (toolkit/lcp.cc:213-328, 380-442), n <= 1024."""
        return self.box_lcp_dantzig(A, b, lo, hi, max_iterations, _entry="egs_box_lcp_murty")

    def box_lcp_dantzig(self, A, b, lo, hi, max_steps=0, _entry="egs_box_lcp_dantzig"):
        """lcp::SolveLCP_BoxDantzig with the incremental factor (toolkit/lcp.cc:444-619), n <= 1024.
        Returns ok, x, w, A permuted in place (lower triangle), perm, pivot steps."""
        A = _f64(A).copy()
        b, lo, hi = map(_f64, (b, lo, hi))
        n = b.shape[0]
        x = np.zeros(n); w = np.zeros(n); perm = np.zeros(n, np.int32); ok = C.c_int32(0); piv = C.c_int32(0)
        st = getattr(load(), _entry)(self.h, C.c_int32(n), _p(A), _p(b), _p(lo), _p(hi), C.c_int32(max_steps),
                                        _p(x), _p(w), _p(perm), C.byref(ok), C.byref(piv))
        if st not in (OK, ERR_LCP_FAILED):
            self.check(st)
        return bool(ok.value), x, w, A, perm, piv.value

    def box_lcp_batch_packed(self, algorithm, ns, A, b, lo, hi, max_steps=0, max_seconds=0.0):
        """egs_box_lcp_batch on packed arrays (problem k's matrix at sum_{j<k} n_j^2, its vectors at sum_{j<k} n_j):
        returns ok, x, w, A (permuted in place, a copy), perm, pivots -- packed the same way."""
        ns = _i32(ns)
        A = _f64(A).copy()
        b, lo, hi = map(_f64, (b, lo, hi))
        tot, cnt = int(ns.sum()), len(ns)
        x = np.zeros(tot); w = np.zeros(tot); perm = np.zeros(tot, np.int32)
        ok = np.zeros(cnt, np.int32); piv = np.zeros(cnt, np.int32)
        self.check(load().egs_box_lcp_batch(self.h, C.c_int32(algorithm), C.c_int32(cnt), _p(ns), _p(A), _p(b), _p(lo), _p(hi),
                                            C.c_int32(max_steps), C.c_double(max_seconds), _p(x), _p(w), _p(perm), _p(ok), _p(piv)))
        return ok, x, w, A, perm, piv

    def box_lcp_batch(self, algorithm, As, bs, los, his, max_steps=0, max_seconds=0.0):
        """`len(As)` independent box LCPs in one launch (egs_box_lcp_batch); algorithm 0 = BoxMurty, 1 = BoxDantzig.
        Returns lists ok, x, w, A (permuted in place), perm, pivots -- problem k's entries equal its single call's."""
        ns = np.array([len(b) for b in bs], np.int32)
        A = np.concatenate([_f64(a).reshape(-1) for a in As])
        b, lo, hi = (np.concatenate([_f64(v) for v in vs]) for vs in (bs, los, his))
        cnt = len(ns)
        ok, x, w, A, perm, piv = self.box_lcp_batch_packed(algorithm, ns, A, b, lo, hi, max_steps, max_seconds)
        vo = np.concatenate([[0], np.cumsum(ns)]); ao = np.concatenate([[0], np.cumsum(ns.astype(np.int64) ** 2)])
        return ([bool(v) for v in ok], [x[vo[k]:vo[k + 1]] for k in range(cnt)], [w[vo[k]:vo[k + 1]] for k in range(cnt)],
                [A[ao[k]:ao[k + 1]].reshape(ns[k], ns[k]) for k in range(cnt)], [perm[vo[k]:vo[k + 1]] for k in range(cnt)],
                [int(v) for v in piv])

    def box_lcp_schur(self, A, b, lo, hi, algorithm=0, nub=-1, reference_quirks=True, max_iterations=0, max_seconds=0.0):
        """lcp::SolveLCP_BoxSchur (toolkit/lcp.cc:627-747).  Returns ok, x, w, A permuted in place (lower triangle),
        perm of the partition, nub, inner pivots."""
        A = _f64(A).copy()
        b, lo, hi = map(_f64, (b, lo, hi))
        n = b.shape[0]
        x = np.zeros(n); w = np.zeros(n); perm = np.zeros(n, np.int32)
        ok = C.c_int32(0); piv = C.c_int32(0); nub_out = C.c_int32(0)
        st = load().egs_box_lcp_schur(self.h, C.c_int32(n), _p(A), _p(b), _p(lo), _p(hi), C.c_int32(algorithm), C.c_int32(nub),
                                      C.c_int32(1 if reference_quirks else 0), C.c_int32(max_iterations), C.c_double(max_seconds),
                                      _p(x), _p(w), _p(perm), C.byref(ok), C.byref(nub_out), C.byref(piv))
        if st not in (OK, ERR_LCP_FAILED):
            self.check(st)
        return bool(ok.value), x, w, A, perm, nub_out.value, piv.value


class Problem:
    """Device-resident ensemble (or batch of ensembles)."""

    def __init__(self, ctx, n_bodies, body0, body1, precision=F64):
        self.ctx = ctx
        self.body0, self.body1 = _i32(body0), _i32(body1)
        self.n, self.m = int(n_bodies), int(self.body0.shape[0])
        self.h = C.c_void_p()
        ctx.check(load().egs_problem_create(ctx.h, C.c_int32(self.n), C.c_int32(self.m), _p(self.body0),
                                            _p(self.body1), C.c_int32(precision), C.byref(self.h)))
        ctx._children.add(self)

    def set_mass(self, inv_mass, inv_inertia):
        """Compact M^-1: 1/m [n] and the inverse global-frame inertia [n][9]."""
        self.ctx.check(load().egs_problem_set_mass(self.h, _p(_f64(inv_mass)), _p(_f64(inv_inertia))))

    @classmethod
    def batch(cls, ctx, n_bodies, n_constraints, body0, body1, precision=F64):
        """E independent ensembles in one problem (egs_problem_create_batch): per-ensemble
        sizes and ensemble-LOCAL body indices; returns (problem, body_offset, constraint_offset)."""
        nb, nc = _i32(n_bodies), _i32(n_constraints)
        self = cls.__new__(cls)
        self.ctx = ctx
        self.body0, self.body1 = _i32(body0), _i32(body1)     # local indices, as passed
        self.n, self.m = int(nb.sum()), int(nc.sum())
        boff = np.zeros(nb.shape[0] + 1, np.int32); coff = np.zeros(nb.shape[0] + 1, np.int32)
        self.h = C.c_void_p()
        ctx.check(load().egs_problem_create_batch(ctx.h, C.c_int32(nb.shape[0]), _p(nb), _p(nc), _p(self.body0),
                                                  _p(self.body1), C.c_int32(precision), C.byref(self.h),
                                                  _p(boff), _p(coff)))
        ctx._children.add(self)
        return self, boff, coff

    def close(self):
        if self.h:
            load().egs_problem_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_blocks(self, Minv=None, J0=None, J1=None, is_eq=None, lo=None, hi=None, rhs=None):
        a = [_f64(Minv), _f64(J0), _f64(J1), _u8(is_eq), _f64(lo), _f64(hi), _f64(rhs)]
        self.ctx.check(load().egs_problem_set_blocks(self.h, *[_p(v) for v in a]))

    def set_state(self, pos=None, R=None, v=None, w=None, Minv=None, f_ext=None):
        a = [_f64(pos), _f64(R), _f64(v), _f64(w), _f64(Minv), _f64(f_ext)]
        self.ctx.check(load().egs_problem_set_state(self.h, *[_p(v_) for v_ in a]))

    def set_constraints(self, kind, data):
        kind, data = _i32(kind), _f64(data)
        self.ctx.check(load().egs_problem_set_constraints(self.h, _p(kind), _p(data)))

    def assemble(self, dt, erp=0.2):
        self.ctx.check(load().egs_problem_assemble(self.h, C.c_double(dt), C.c_double(erp)))

    def solve(self, prm, want_stats=True):
        st = SolveStats()
        self.ctx.check(load().egs_problem_solve(self.h, C.byref(prm), C.byref(st) if want_stats else None))
        return st

    def step(self, dt, erp, prm, want_stats=False):
        st = SolveStats()
        self.ctx.check(load().egs_problem_step(self.h, C.c_double(dt), C.c_double(erp), C.byref(prm),
                                               C.byref(st) if want_stats else None))
        return st

    def stats(self):
        st = SolveStats()
        self.ctx.check(load().egs_problem_get_stats(self.h, C.byref(st)))
        return st

    def lambda_(self):
        x = np.zeros(3 * self.m)
        self.ctx.check(load().egs_problem_get_lambda(self.h, _p(x)))
        return x

    def debug_trace(self, max_sweeps=1000):
        """EGS_TRACE_UPDATES=1: completion stamps [sweeps][m] (100 MHz ticks) of the last 4-lane patch launch."""
        buf = np.zeros(max_sweeps * self.m, np.uint64)
        sw = C.c_int32(0)
        self.ctx.check(load().egs_problem_debug_trace(self.h, _p(buf), C.c_int64(buf.shape[0]), C.byref(sw)))
        return buf[:sw.value * self.m].reshape(sw.value, self.m)

    def wres(self):
        """w = A lambda - rhs of the last solve (the solve kernels' epilogue)."""
        w = np.zeros(3 * self.m)
        self.ctx.check(load().egs_problem_get_wres(self.h, _p(w)))
        return w

    def matvec(self, x=None, parts=MV_FULL, eps=0.0, scale=1.0, fetch=True):
        """y = part(J M^-1 J^T) x (egs_problem_matvec).  x=None: the device-resident lambda;
        fetch=False leaves y on the device (asynchronous; matvec_result() reads it)."""
        xx = _f64(x) if x is not None else None
        y = np.zeros(3 * self.m) if fetch else None
        self.ctx.check(load().egs_problem_matvec(self.h, C.c_int32(parts), C.c_double(eps), C.c_double(scale),
                                                 _p(xx), _p(y)))
        return y

    def matvec_result(self):
        y = np.zeros(3 * self.m)
        self.ctx.check(load().egs_problem_get_matvec(self.h, _p(y)))
        return y

    def dense_system(self, cfm=0.0):
        """A = J M^-1 J^T + cfm I (ensembles.cc:510, 513-521), built on the device."""
        A = np.zeros((3 * self.m, 3 * self.m))
        self.ctx.check(load().egs_problem_dense_system(self.h, C.c_double(cfm), _p(A)))
        return A

    def dense_condition(self, cfm=0.0):
        est = C.c_double(0)
        self.ctx.check(load().egs_problem_dense_condition(self.h, C.c_double(cfm), C.byref(est)))
        return est.value

    def step_dense(self, dt, erp=0.2, cfm=0.0, use_bounds=0):
        """StepVelocities_ODE through the dense path: returns (ok, pivots)."""
        ok = C.c_int32(0); piv = C.c_int32(0)
        st = load().egs_problem_step_dense(self.h, C.c_double(dt), C.c_double(erp), C.c_double(cfm), C.c_int32(use_bounds),
                                           C.byref(ok), C.byref(piv))
        if st not in (OK, ERR_LCP_FAILED):
            self.ctx.check(st)
        return bool(ok.value), piv.value

    def accumulators(self):
        a = np.zeros((self.n, 6))
        self.ctx.check(load().egs_problem_get_accumulators(self.h, _p(a)))
        return a

    def velocity(self):
        v = np.zeros((self.n, 6))
        self.ctx.check(load().egs_problem_get_velocity(self.h, _p(v)))
        return v

    def advance(self, dt):
        self.ctx.check(load().egs_problem_advance(self.h, C.c_double(dt)))

    def state(self):
        pos = np.zeros((self.n, 3)); R = np.zeros((self.n, 9)); v = np.zeros((self.n, 3)); w = np.zeros((self.n, 3))
        self.ctx.check(load().egs_problem_get_state(self.h, _p(pos), _p(R), _p(v), _p(w)))
        return pos, R, v, w

    def blocks(self):
        m = self.m
        J0 = np.zeros((m, 18)); J1 = np.zeros((m, 18)); is_eq = np.zeros(3 * m, np.uint8)
        lo = np.zeros(3 * m); hi = np.zeros(3 * m); rhs = np.zeros(3 * m); err = np.zeros(3 * m)
        self.ctx.check(load().egs_problem_get_blocks(self.h, _p(J0), _p(J1), _p(is_eq), _p(lo), _p(hi),
                                                     _p(rhs), _p(err)))
        return J0, J1, is_eq, lo, hi, rhs, err


def debug_plan(n_bodies, body0, body1, tile_size=256):
    """Host-only view of the schedule (islands, tiles, tickets); needs no GPU."""
    body0, body1 = _i32(body0), _i32(body1)
    m = body0.shape[0]
    ni, nt, ng = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    out = [np.zeros(m, np.int32) for _ in range(5)]
    st = load().egs_debug_plan(C.c_int32(n_bodies), C.c_int32(m), _p(body0), _p(body1), C.c_int32(tile_size),
                               C.byref(ni), C.byref(nt), C.byref(ng), *[_p(o) for o in out])
    if st != OK:
        raise EgsError(st, "egs_debug_plan failed")
    return dict(n_islands=ni.value, n_tiles=nt.value, n_global=ng.value, cons_tile=out[0],
                pos0=out[1], cnt0=out[2], pos1=out[3], cnt1=out[4])


def debug_choose_oversize_schedule(n_patch_tiles, quad_per_cu, patch_per_cu, cu_count=256, patches=True, quad_patches=True):
    """0 = 4-lane patches, 1 = 1-lane patches, 2 = all-global kernel (host only)."""
    return int(load().egs_debug_choose_oversize_schedule(C.c_int32(n_patch_tiles), C.c_int32(quad_per_cu),
                                                         C.c_int32(patch_per_cu), C.c_int32(cu_count),
                                                         C.c_int32(1 if patches else 0), C.c_int32(1 if quad_patches else 0)))


def debug_matvec_plan(n_bodies, body0, body1, tile_size=256):
    """Host-only view of the mat-vec schedule (tiles, shared bodies, boundary list); needs no GPU."""
    body0, body1 = _i32(body0), _i32(body1)
    m = body0.shape[0]
    vals = [C.c_int32(0) for _ in range(4)]
    ct = np.full(m, -1, np.int32); cl = np.full(m, -1, np.int32)
    st = load().egs_debug_matvec_plan(C.c_int32(n_bodies), C.c_int32(m), _p(body0), _p(body1), C.c_int32(tile_size),
                                      *[C.byref(v) for v in vals], _p(ct), _p(cl))
    if st != OK:
        raise EgsError(st, "egs_debug_matvec_plan failed")
    return dict(n_tiles=vals[0].value, n_islands=vals[1].value, n_shared_bodies=vals[2].value,
                n_boundary=vals[3].value, cons_tile=ct, cons_lane=cl)


def debug_plan_slots(n_bodies, body0, body1, tile_size=256):
    """Host-only: per constraint its lane in the tile, the LDS slots of its two sides and
    the slot count of its tile (see egs_debug_plan_slots)."""
    body0, body1 = _i32(body0), _i32(body1)
    m = body0.shape[0]
    out = [np.zeros(m, np.int32) for _ in range(4)]
    st = load().egs_debug_plan_slots(C.c_int32(n_bodies), C.c_int32(m), _p(body0), _p(body1), C.c_int32(tile_size),
                                     *[_p(o) for o in out])
    if st != OK:
        raise EgsError(st, "egs_debug_plan_slots failed")
    return dict(lane=out[0], slot0=out[1], slot1=out[2], tile_nslots=out[3])


def debug_plan_timetable(n_bodies, body0, body1, tile_size=256):
    """Host-only: per constraint its level in the list-order dependency DAG and the period / depth of
    its tile's static timetable (see egs_debug_plan_timetable)."""
    body0, body1 = _i32(body0), _i32(body1)
    m = body0.shape[0]
    out = [np.zeros(m, np.int32) for _ in range(3)]
    runs = C.c_int32(0)
    st = load().egs_debug_plan_timetable(C.c_int32(n_bodies), C.c_int32(m), _p(body0), _p(body1), C.c_int32(tile_size),
                                         *[_p(o) for o in out], C.byref(runs))
    if st != OK:
        raise EgsError(st, "egs_debug_plan_timetable failed")
    return dict(level=out[0], period=out[1], depth=out[2], runs=bool(runs.value))


class World:
    """Ensemble::Step resident on the device (collide -> solve -> integrate)."""

    def __init__(self, ctx, n_bodies, precision=F64):
        self.ctx, self.n = ctx, int(n_bodies)
        self.h = C.c_void_p()
        ctx.check(load().egs_world_create(ctx.h, C.c_int32(self.n), C.c_int32(precision), C.byref(self.h)))
        ctx._children.add(self)

    def close(self):
        if self.h:
            load().egs_world_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_bodies(self, pos, R, v, w, Minv, f_ext, side=None):
        side = _f64(np.tile([0.3, 0.3, 0.3], (self.n, 1)) if side is None else side)
        a = [_f64(pos), _f64(R), _f64(v), _f64(w), _f64(Minv), _f64(f_ext), side]
        self.ctx.check(load().egs_world_set_bodies(self.h, *[_p(x) for x in a]))

    def set_joints(self, body0, body1, data):
        body0, body1, data = _i32(body0), _i32(body1), _f64(data)
        self.ctx.check(load().egs_world_set_joints(self.h, C.c_int32(body0.shape[0]), _p(body0), _p(body1), _p(data)))

    def step(self, dt, erp, prm, detect_contacts=True, want_stats=False):
        st = SolveStats()
        self.ctx.check(load().egs_world_step(self.h, C.c_double(dt), C.c_double(erp), C.byref(prm),
                                             C.c_int32(1 if detect_contacts else 0),
                                             C.byref(st) if want_stats else None))
        return st

    def bodies(self):
        pos = np.zeros((self.n, 3)); R = np.zeros((self.n, 9)); v = np.zeros((self.n, 3)); w = np.zeros((self.n, 3))
        self.ctx.check(load().egs_world_get_bodies(self.h, _p(pos), _p(R), _p(v), _p(w)))
        return pos, R, v, w

    def info(self):
        a, b, c = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self.ctx.check(load().egs_world_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(n_constraints=a.value, n_contacts=b.value, replans=c.value)

    def contacts(self):
        m = self.info()["n_contacts"]
        b0 = np.zeros(m, np.int32); b1 = np.zeros(m, np.int32); data = np.zeros((m, 7)); mo = C.c_int32(0)
        self.ctx.check(load().egs_world_get_contacts(self.h, C.c_int32(m), C.byref(mo), _p(b0), _p(b1), _p(data)))
        return b0, b1, data

    def lambda_(self):
        rows = 3 * self.info()["n_constraints"]
        x = np.zeros(rows); ro = C.c_int32(0)
        self.ctx.check(load().egs_world_get_lambda(self.h, C.c_int32(rows), C.byref(ro), _p(x)))
        return x
