"""GPU: the Kalman-based estimators (SURVEY a9 ClockRecovery + KalmanFilter, a10 FreqDevEstimator) under EVERY evaluation order
the switch offers (include/m17hip.h m17hip_set_kalman_order; oracle/m17_oracle_dsp.hpp kalman_order): the HIP arithmetic equals
the oracle's bit for bit at operator level, and the whole chain stays bit-exact under each order.  What the choice of order
can move in the decoded output is measured by tools/kalman_sensitivity.py."""
import ctypes as C

import numpy as np
import pytest

import m17hip
import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = m17hip.Context(64, 96000)
    yield c
    c.set_kalman_order(3)
    c.close()


@pytest.fixture
def oracle_order():
    lib = ol.oracle()
    yield lambda order: lib.m17o_set_kalman_order(C.c_int(order))
    lib.m17o_set_kalman_order(C.c_int(3))


def _oracle_trace(z, dt, wrap, z0):
    out = np.zeros(z.shape + (6,), dtype=np.float32)
    for r in range(z.shape[0]):
        zz = np.ascontiguousarray(z[r]); dd = np.ascontiguousarray(dt[r])
        ol.oracle().m17o_kalman_trace(ol._p(zz), ol._p(dd), C.c_size_t(zz.size), C.c_int(wrap), C.c_float(z0), ol._p(out[r]))
    return out


@pytest.mark.parametrize("order", range(8))
def test_kalman_update_bit_exact_each_order(ctx, oracle_order, order):
    """kal_update (the device function K5 calls) == the oracle's Kalman2::update: index filter with wrap-around
    (KalmanFilter.h:41-65) and symbol filter (:91-107), noisy measurements, irregular dt, NaN / inf measurements."""
    oracle_order(order)
    rng = np.random.default_rng(100 + order)
    rows, n = 96, 300
    z = ((5.0 + rng.normal(0, 2.5, (rows, n))) % 10.0).astype(np.float32)
    dt = rng.choice([1920, 1920, 960, 3840, 5760, 1, 77, 19200], (rows, n)).astype(np.uint32)
    got = ctx.kalman_trace(z, dt, 10, z0=4.0, order=order)
    assert np.array_equal(got, _oracle_trace(z, dt, 10, 4.0))
    assert (got[..., 0] >= 0).all() and (got[..., 0] < 10).all()
    lv = (rng.choice([-5.2, 5.2, 0.6], (rows, 1)) * (1 + rng.normal(0, 0.05, (rows, n)))).astype(np.float32)
    lv[3, 50] = np.nan; lv[4, 10] = np.inf; lv[5, 200] = -np.inf; lv[6, :] = 0.0
    d192 = np.full((rows, n), 192, dtype=np.uint32)
    got = ctx.kalman_trace(lv, d192, 0, z0=float(lv[0, 0]), order=order)
    exp = _oracle_trace(lv, d192, 0, float(lv[0, 0]))
    assert np.array_equal(got, exp, equal_nan=True)
    assert np.isnan(got[3, 50:, :2]).all() and np.isfinite(got[7]).all()   # a NaN measurement poisons x, never P
    # the level filters as K5 runs them — the covariance from the gain schedule (detail/core.h level_schedule), state arithmetic only —
    # over more updates than the schedule is long (its last entry is the covariance's fixed point)
    n2 = 900
    lv = (rng.choice([-5.2, 5.2, 0.6], (rows, 1)) * (1 + rng.normal(0, 0.05, (rows, n2)))).astype(np.float32)
    lv[3, 700] = np.nan; lv[4, 10] = np.inf; lv[6, :] = 0.0
    d192 = np.full((rows, n2), 192, dtype=np.uint32)
    got = ctx.kalman_trace(lv, d192, -1, z0=float(lv[0, 0]), order=order)
    exp = _oracle_trace(lv, d192, 0, float(lv[0, 0]))
    assert np.array_equal(got[..., :2], exp[..., :2], equal_nan=True)


@pytest.mark.parametrize("order", [0, 1, 2, 3, 7])
def test_full_chain_bit_exact_each_order(ctx, oracle_order, order):
    """Records, diagnostics and live counters of a randomised multi-burst scenario, HIP == oracle, under each order."""
    oracle_order(order)
    ctx.set_kalman_order(order)
    rng = np.random.default_rng(4242)    # the same streams for every order
    Cn, T = 64, 96000
    x = np.zeros((Cn, T), dtype=np.int16)
    for c in range(Cn):
        pos = 0
        while pos < T - 8000:
            n = min(int(rng.integers(8000, 40000)), T - pos)
            p = ol.gen_params(seed=int(rng.integers(1, 1 << 30)), kind=int(rng.choice([0, 1, 2, 4])), n_frames=int(rng.integers(2, 16)),
                              lead_in=int(rng.integers(0, 4000)), lead_sigma=float(rng.choice([0.0, 300.0, 40000.0])),
                              noise_sigma=float(rng.choice([0.0, 300.0, 1000.0, 2500.0])), tail_sigma=float(rng.choice([0.0, 300.0, 3000.0])),
                              dc_offset=float(rng.choice([0.0, 500.0, -2000.0])), gain=float(rng.choice([1.0, 0.4, 1.5])),
                              phase=int(rng.integers(-1, 10)), total=n)
            x[c, pos:pos + n] = ol.generate(p)[:n]
            pos += n
    recs, counts, diags = ol.demod_batch(x, cap=2 * (T // 1920 + 2) + 4, threads=8)
    exp = np.concatenate([recs[c, :counts[c]] for c in range(Cn)])
    ctx.upload(x); ctx.reset(); ctx.run()
    got = ctx.frames(); d = ctx.diag()
    assert exp.size > 4 * Cn and got.tobytes() == exp.tobytes()
    for f in d.dtype.names:
        assert np.array_equal(d[f], diags[f], equal_nan=True), f


def test_orders_are_distinguishable_in_the_estimates(ctx):
    """The switch is live in K5: on a noisy multi-burst stream the deviation / offset / clock floats of some channel differ
    in the last place between the eager (0) and the blaze-restructured (3) order — and the frame payloads do not."""
    p = ol.gen_params(seed=9, kind=-1, n_frames=44, lead_in=3072, noise_sigma=1500.0, tail_sigma=1500.0, lead_sigma=40000.0, total=96000)
    x = ol.generate_batch(p, 64, 96000, threads=8)
    res = {}
    for order in (0, 3):
        ctx.set_kalman_order(order)
        ctx.upload(x); ctx.reset(); ctx.run()
        res[order] = (ctx.frames().copy(), ctx.diag().copy())
    ctx.set_kalman_order(3)
    a, b = res[0][1], res[3][1]
    assert any((a[f].view(np.uint32) != b[f].view(np.uint32)).any() for f in ("deviation", "offset", "clock"))
    assert res[0][0].size == res[3][0].size and np.array_equal(res[0][0]["payload"], res[3][0]["payload"])
