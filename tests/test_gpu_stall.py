"""GPU (-m gpu): the device-side ordering waits are bounded and a timed-out wait is never
silent (EGS_ERR_STALL).  EGS_DEBUG_SPIN_LIMIT=1 forces the time-out: the flag is sticky, so
an asynchronous step reports it at the next synchronising call, and a world refuses to
integrate the bodies with the lambda of a stalled solve.  The static-timetable kernel
(step_solve.hip) has no waits at all, so these tests pin the ticket kernels (EGS_STEP=0);
the last test checks that the timetable is indifferent to the spin limit."""
import numpy as np
import pytest

from eggshell_amd import capi, scenes
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def pile_problem(ctx, shape=(3, 3, 6)):
    sc = scenes.box_stack(*shape, jitter=1e-3, seed=2)
    Minv = orc.minv_blocks(sc["R"], sc["mass"], sc["I_body"])
    f_ext = orc.external_force(sc["R"], sc["w"], sc["mass"], sc["I_body"])
    pr = capi.Problem(ctx, sc["p"].shape[0], sc["body0"], sc["body1"])
    pr.set_state(sc["p"], sc["R"], sc["v"], sc["w"], Minv, f_ext)
    pr.set_constraints(sc["kind"], sc["data"])
    return sc, Minv, f_ext, pr


@pytest.fixture(autouse=True)
def ticket_kernels(request, monkeypatch):
    if "timetable" not in request.node.name:
        monkeypatch.setenv("EGS_STEP", "0")


@pytest.mark.parametrize("quad", ["0", "1"])
def test_async_step_stall_is_reported_and_sticky(ctx, quad, monkeypatch):
    monkeypatch.setenv("EGS_QUAD", quad)
    sc, Minv, f_ext, pr = pile_problem(ctx)
    prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=20, tol=0.0, cfm=0.01)
    pr.step(5e-3, 0.2, prm)                      # healthy, asynchronous
    good = pr.lambda_()
    monkeypatch.setenv("EGS_DEBUG_SPIN_LIMIT", "1")
    pr.step(5e-3, 0.2, prm)                      # stalls on the device; nobody is looking yet
    monkeypatch.delenv("EGS_DEBUG_SPIN_LIMIT")
    with pytest.raises(capi.EgsError) as e:
        pr.step(5e-3, 0.2, prm)                  # a healthy step afterwards must not wipe the flag: either it
        pr.lambda_()                             # sees the landed flag itself, or the synchronising getter does
    assert e.value.status == capi.ERR_STALL
    # reported once, then cleared: the problem is usable again and gives the healthy answer
    pr.step(5e-3, 0.2, prm)
    assert np.array_equal(pr.lambda_(), good)
    pr.close()


def test_stall_with_stats_fails_the_call(ctx, monkeypatch):
    sc, Minv, f_ext, pr = pile_problem(ctx)
    monkeypatch.setenv("EGS_DEBUG_SPIN_LIMIT", "1")
    with pytest.raises(capi.EgsError) as e:
        pr.step(5e-3, 0.2, capi.params(method=capi.SOR, max_iters=10, tol=0.0, cfm=0.01), want_stats=True)
    assert e.value.status == capi.ERR_STALL
    with pytest.raises(capi.EgsError) as e:       # tolerance-terminated loop
        pr.step(5e-3, 0.2, capi.params(method=capi.GAUSS_SEIDEL, max_iters=50, tol=1e-9, cfm=0.01))
    assert e.value.status == capi.ERR_STALL
    monkeypatch.delenv("EGS_DEBUG_SPIN_LIMIT")
    st = pr.step(5e-3, 0.2, capi.params(method=capi.SOR, max_iters=10, tol=0.0, cfm=0.01), want_stats=True)
    assert st.status == capi.OK
    pr.close()


def test_world_does_not_integrate_a_stalled_solve(ctx, monkeypatch):
    sc = scenes.box_stack(3, 3, 5, jitter=1e-3, seed=5)
    Minv = orc.minv_blocks(sc["R"], sc["mass"], sc["I_body"])
    f_ext = orc.external_force(sc["R"], sc["w"], sc["mass"], sc["I_body"])
    w = capi.World(ctx, sc["p"].shape[0])
    w.set_bodies(sc["p"], sc["R"], sc["v"], sc["w"], Minv, f_ext)
    prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=20, tol=0.0, cfm=0.01)
    w.step(5e-3, 0.2, prm)
    before = w.bodies()
    monkeypatch.setenv("EGS_DEBUG_SPIN_LIMIT", "1")
    with pytest.raises(capi.EgsError) as e:
        w.step(5e-3, 0.2, prm)
    assert e.value.status == capi.ERR_STALL
    monkeypatch.delenv("EGS_DEBUG_SPIN_LIMIT")
    after = w.bodies()
    for a, b in zip(before, after):
        assert np.array_equal(a, b)               # the state was not advanced
    w.step(5e-3, 0.2, prm)                        # and the world goes on
    assert not np.array_equal(w.bodies()[0], before[0])
    w.close()


def test_timetable_kernel_has_no_waits_to_time_out(ctx, monkeypatch):
    monkeypatch.setenv("EGS_QUAD", "0")
    monkeypatch.setenv("EGS_STEP", "1")
    sc, Minv, f_ext, pr = pile_problem(ctx)
    prm = capi.params(method=capi.GAUSS_SEIDEL, max_iters=20, tol=0.0, cfm=0.01)
    st = pr.step(5e-3, 0.2, prm, want_stats=True)
    assert st.schedule & capi.SCHED_STATIC
    good = pr.lambda_()
    monkeypatch.setenv("EGS_DEBUG_SPIN_LIMIT", "1")
    st = pr.step(5e-3, 0.2, prm, want_stats=True)
    assert st.status == capi.OK and st.schedule & capi.SCHED_STATIC
    assert np.array_equal(pr.lambda_(), good)
    pr.close()
