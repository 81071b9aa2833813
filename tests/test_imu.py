"""IMU front end (the step before the measurement path): esekf::predict and ImuProcess on the host.
CPU tests -- no device work: lio_imu_process is called with ctx = NULL (propagate only).  The GPU half
(point compensation with the propagated poses) is in test_gpu_parity.py."""
import numpy as np
import pytest

from better_fastlio2_amd import capi
from oracle import imu_model as im


def _rand_state(rng, rate=0.0):
    def rq():
        q = rng.normal(size=4)
        return q / np.linalg.norm(q)
    g = rng.normal(size=3)
    g = g / np.linalg.norm(g) * 9.809
    return capi.make_state26(rng.normal(size=3) * 10, rq(), rq(), rng.normal(size=3) * 0.1, rng.normal(size=3),
                             rng.normal(size=3) * 0.01, rng.normal(size=3) * 0.05, g)


def _rand_cov(rng):
    A = rng.normal(size=(23, 23))
    return A @ A.T * 1e-3 + np.eye(23) * 1e-4


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_predict_matches_restatement(seed):
    """lio_eskf_predict vs the numpy restatement of esekfom.hpp:280-383 over a random IMU stream."""
    rng = np.random.default_rng(seed)
    x, P = _rand_state(rng), _rand_cov(rng)
    kf = capi.Esekf(3)
    kf.change_x(x); kf.change_P(P)
    Q = capi.process_noise_cov()
    assert np.array_equal(Q, im.process_noise_cov())
    Q[np.arange(12), np.arange(12)] *= rng.uniform(0.5, 20, 12)
    xo, Po = x.copy(), P.copy()
    for _ in range(60):
        acc = rng.normal(size=3) * 2 + [0, 0, 9.8]
        gyr = rng.normal(size=3) * 0.5
        dt = rng.choice([0.005, 0.01, 0.0, -0.002, 0.0137])
        kf.predict(dt, Q, acc, gyr)
        xo, Po = im.predict(xo, Po, dt, Q, acc, gyr)
    assert np.abs(kf.get_x() - xo).max() < 1e-12
    assert np.abs(kf.get_P() - Po).max() < 1e-12 * np.abs(Po).max()
    assert np.allclose(kf.get_P(), kf.get_P().T, rtol=0, atol=1e-12 * np.abs(Po).max())


def test_predict_constant_input_closed_form():
    """Physics: a constant body rate integrates to the exact rotation exp(w T) (every step is the exact exponential
    about the same axis); with no rotation the velocity is exactly (R a + g) t and the position follows the
    reference's explicit Euler rule p += dt * v_old, i.e. n (n - 1) / 2 * dt^2 * (R a + g)."""
    kf = capi.Esekf(3)
    q0 = np.array([0.1, -0.2, 0.3, 0.9]); q0 /= np.linalg.norm(q0)
    x = capi.make_state26([1, 2, 3], q0, [0, 0, 0, 1], [0, 0, 0], [0, 0, 0], [0.01, -0.02, 0.005], [0, 0, 0], [0, 0, -9.809])
    kf.change_x(x); kf.change_P(np.eye(23) * 1e-3)
    Q = capi.process_noise_cov()
    w = np.array([0.3, -0.1, 0.7])
    n, dt = 200, 0.005
    for _ in range(n):
        kf.predict(dt, Q, [0.0, 0.0, 0.0], w + x[17:20])          # gyro - bg = w
    xr = kf.get_x()
    th = w * n * dt
    ang = np.linalg.norm(th)
    dq = np.concatenate([np.sin(ang / 2) * th / ang, [np.cos(ang / 2)]])
    qe = im.qmul(q0, dq)
    assert np.abs(xr[3:7] - qe).max() < 1e-13
    # translation, no rotation
    a = np.array([0.5, -0.25, 9.9])
    x2 = capi.make_state26([0, 0, 0], q0, [0, 0, 0, 1], [0, 0, 0], [0, 0, 0], [0, 0, 0], [0.1, 0.0, -0.1], [0, 0, -9.809])
    kf.change_x(x2)
    for _ in range(n):
        kf.predict(dt, Q, a + x2[20:23], [0.0, 0.0, 0.0])
    xr = kf.get_x()
    acc_w = im.qmat(q0) @ a + x2[23:26]
    assert np.abs(xr[14:17] - acc_w * n * dt).max() < 1e-12
    assert np.abs(xr[0:3] - acc_w * dt * dt * n * (n - 1) / 2).max() < 1e-12
    assert np.array_equal(xr[3:7], q0) and np.array_equal(xr[23:26], x2[23:26]) and np.array_equal(xr[7:14], x2[7:14])


def test_transition_matrix_against_finite_differences():
    """The transition matrix predict() uses for P equals the derivative of the propagation on the manifold,
    d( propagate(x [+] e) [-] propagate(x) ) / de, up to the reference's own first-order treatment of the
    rotation block (its exp() there has scale 1/2 = 0, esekfom.hpp:313): compared with a step of 1 ms, where that
    term is |w| dt ~ 5e-4."""
    rng = np.random.default_rng(7)
    x = _rand_state(rng)
    acc = np.array([0.3, -0.2, 9.7]); gyr = np.array([0.4, -0.3, 0.2])
    dt = 1e-3
    Q = im.process_noise_cov()
    x1, _, F = im.predict(x, np.eye(23), dt, Q, acc, gyr, return_F=True)
    L = capi.load()
    eps = 1e-6
    Fn = np.zeros((23, 23))
    for j in range(23):
        e = np.zeros(23); e[j] = eps
        xp = x.copy(); L.lio_state_boxplus(capi._ptr(xp), capi._ptr(e))
        xm = x.copy(); em = -e; L.lio_state_boxplus(capi._ptr(xm), capi._ptr(em))
        xp1, _ = im.predict(xp, np.eye(23), dt, Q, acc, gyr)
        xm1, _ = im.predict(xm, np.eye(23), dt, Q, acc, gyr)
        d = np.zeros(23)
        L.lio_state_boxminus(capi._ptr(xp1), capi._ptr(xm1), capi._ptr(d))
        Fn[:, j] = d / (2 * eps)
    err = np.abs(F - Fn)
    assert err.max() < 2e-3, err.max()
    mask = np.ones((23, 23), bool); mask[3:6, 3:6] = False          # the block the reference leaves at identity
    assert err[mask].max() < 2e-5, err[mask].max()


def _imu_stream(rng, t0, t1, rate, bias_g, grav_dir, moving):
    ts = np.arange(t0, t1, 1.0 / rate)
    rows = []
    for t in ts:
        acc = -grav_dir * 9.79 + rng.normal(0, 0.02, 3)
        gyr = bias_g + rng.normal(0, 0.002, 3)
        if moving:
            acc = acc + np.array([0.8 * np.sin(3 * t), 0.5 * np.cos(2 * t), 0.2 * np.sin(5 * t)])
            gyr = gyr + np.array([0.2 * np.sin(2 * t), -0.1 * np.cos(3 * t), 0.3 * np.sin(t)])
        rows.append(np.concatenate([[t], acc, gyr]))
    return np.array(rows).reshape(-1, 7)


@pytest.mark.parametrize("seed,rate,scan_hz", [(11, 200.0, 10.0), (12, 100.0, 20.0), (13, 400.0, 10.0)])
def test_imu_process_matches_restatement(seed, rate, scan_hz):
    """ImuProcess::Process over a stream of scans: static initialisation (gravity, gyroscope bias, extrinsics,
    P), then forward propagation scan by scan -- state, covariance, the IMUpose list and the running statistics
    equal the numpy restatement of src/IMU_Processing.hpp:174-332 at every scan."""
    rng = np.random.default_rng(seed)
    T = np.array([0.05, -0.02, 0.1])
    ang = 0.3
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]])
    gdir = np.array([0.02, -0.03, -1.0]); gdir /= np.linalg.norm(gdir)
    bias = np.array([0.003, -0.002, 0.001])
    p = capi.ImuProcess(); p.set_extrinsic(T, R); p.set_cov(0.1, 0.1, 0.0001, 0.0001)
    o = im.ImuProcess(); o.Lidar_T, o.Lidar_R = T.copy(), R.copy()
    kf = capi.Esekf(3)
    xo, Po = kf.get_x(), kf.get_P()
    period = 1.0 / scan_hz
    t = 100.0
    n_und = 0
    for scan in range(14):
        beg, end = t, t + period * rng.uniform(0.9, 1.0)
        # the IMU samples of a scan end a little before or after the last LiDAR point (both branches of :324-326)
        imu_end = end + rng.uniform(-0.6, 0.6) / rate
        imu = _imu_stream(rng, beg - rng.uniform(0, 1) / rate, imu_end, rate, bias, gdir, moving=scan >= 3)
        if scan == 6:
            imu = imu[:0]                                            # no IMU in this scan: Process returns at once
        ready = p.process(kf, None, beg, end, imu)
        xo, Po, ready_o = o.process(imu, beg, end, xo, Po)
        assert ready is False                                        # ctx = NULL: nothing was undistorted
        info = p.info()
        assert info["need_init"] == o.need_init and info["init_iter_num"] == o.init_iter_num
        assert np.abs(info["mean_acc"] - o.mean_acc).max() < 1e-13 and np.abs(info["mean_gyr"] - o.mean_gyr).max() < 1e-15
        assert np.abs(info["cov_acc"] - o.cov_acc).max() < 1e-15 and np.abs(info["cov_gyr"] - o.cov_gyr).max() < 1e-15
        assert np.abs(kf.get_x() - xo).max() < 1e-11, (scan, np.abs(kf.get_x() - xo).max())
        assert np.abs(kf.get_P() - Po).max() < 1e-11 * max(1.0, np.abs(Po).max())
        if ready_o:
            n_und += 1
            assert info["last_lidar_end_time"] == end
            gp, op = p.poses(), np.array(o.poses)
            assert gp.shape == op.shape and gp.shape[0] >= 2
            assert np.abs(gp - op).max() < 1e-11
            assert gp[0, 0] == 0.0 and np.all(np.diff(gp[1:, 0]) > 0)      # the first IMU sample may precede the scan
        t += period
    assert n_und >= 8
    xs = kf.get_x()
    assert abs(np.linalg.norm(xs[23:26]) - 9.809) < 1e-9             # gravity stays on its sphere
    assert abs(np.linalg.norm(xs[3:7]) - 1.0) < 1e-9
    # extrinsics as configured, bias as measured at rest
    assert np.abs(im.qmat(xs[7:11]) - R).max() < 1e-12 and np.array_equal(xs[11:14], T)


def test_imu_init_sets_gravity_bias_and_covariance():
    """IMU_init (src/IMU_Processing.hpp:174-231): after the first scan with more than MAX_INI_COUNT samples the
    state holds gravity opposite to the mean specific force with length 9.809, bg = mean rate, and the documented
    diagonal of P."""
    rng = np.random.default_rng(5)
    gdir = np.array([0.1, 0.05, -1.0]); gdir /= np.linalg.norm(gdir)
    bias = np.array([0.01, -0.02, 0.03])
    imu = _imu_stream(rng, 0.0, 0.1, 200.0, bias, gdir, moving=False)
    p = capi.ImuProcess()
    kf = capi.Esekf(3)
    assert p.process(kf, None, 0.0, 0.1, imu) is False
    info = p.info()
    assert not info["need_init"] and info["init_iter_num"] == len(imu) + 1
    x = kf.get_x()
    assert np.abs(x[23:26] / 9.809 - gdir).max() < 5e-3
    assert np.abs(x[17:20] - bias).max() < 2e-3 and np.array_equal(x[17:20], info["mean_gyr"])
    d = np.diag(kf.get_P())
    assert np.array_equal(d, [1, 1, 1, 1, 1, 1] + [1e-5] * 6 + [1, 1, 1] + [1e-4] * 3 + [1e-3] * 3 + [1e-5] * 2)
    assert np.count_nonzero(kf.get_P()) == 23
    # a second, shorter first scan keeps initialising (fewer than MAX_INI_COUNT samples seen)
    p2 = capi.ImuProcess()
    p2.process(kf, None, 0.0, 0.02, imu[:4])
    assert p2.info()["need_init"] and p2.info()["init_iter_num"] == 5
    p2.process(kf, None, 0.02, 0.06, imu[4:12])
    assert not p2.info()["need_init"]


def test_predict_keeps_the_covariance_symmetric_and_matches_the_dense_product():
    """esekf::predict forms F P F^T + W Q W^T from the non-zero entries of F and W as combinations of padded rows (an AVX2 clone
    exists); against the dense numpy product of the oracle's F and W the result agrees to rounding, whatever P holds -- dense,
    diagonal (exact zeros everywhere), badly scaled -- and a symmetric P stays symmetric to rounding (esekfom.hpp:380-381)."""
    import numpy as np
    from better_fastlio2_amd import capi, synth
    from oracle import imu_model as im
    rng = np.random.default_rng(7)
    for case in range(12):
        x = np.zeros(26)
        x[:3] = rng.normal(0, 1, 3)
        for a in (3, 7):
            q = rng.normal(size=4); x[a:a + 4] = q / np.linalg.norm(q)
        x[11:14] = rng.normal(0, 0.1, 3); x[14:17] = rng.normal(0, 1, 3); x[17:23] = rng.normal(0, 0.01, 6)
        g = rng.normal(size=3); x[23:26] = g / np.linalg.norm(g) * 9.81
        A = rng.normal(size=(23, 23))
        P = A @ A.T * 1e-3 if case % 3 else np.diag(rng.uniform(1e-8, 1e-1, 23))
        if case % 4 == 1:
            s = np.diag(10.0 ** rng.uniform(-4, 4, 23)); P = s @ P @ s
        B = rng.normal(size=(12, 12)); Q = B @ B.T * 1e-4
        kf = capi.Esekf(3); kf.change_x(x); kf.change_P(P)
        acc = rng.normal(0, 3, 3) + [0, 0, 9.8]; gyr = rng.normal(0, 0.5, 3); dt = float(rng.uniform(0.001, 0.02))
        kf.predict(dt, Q, acc, gyr)
        xo, Po = im.predict(x, P, dt, Q, acc, gyr)
        Pg = kf.get_P()
        scale = np.sqrt(np.outer(np.abs(np.diag(Po)), np.abs(np.diag(Po)))) + 1e-300
        assert np.abs(Pg - Po).max() <= 5e-13 * np.abs(Po).max() and (np.abs(Pg - Po) / scale).max() < 1e-9
        assert (np.abs(Pg - Pg.T) / scale).max() < 1e-12
