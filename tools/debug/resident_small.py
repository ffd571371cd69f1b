import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_loop_parity as T
import gpu_se_amd
for driver, B, n, dtc, dtp, end in [("resident", 1, 256, 0.1, 0.1, 3), ("host", 1, 256, 0.1, 0.1, 3), ("resident", 1, 256, 1.0, 1.0, 6), ("host", 1, 256, 1.0, 1.0, 6)]:
    gpu_se_amd.seed(100)
    ticks, n_p, n_u = T.schedule(end, dtc, dtp)
    noise = T.draw_noise(7, ticks, n_p, n_u, B, n, pf=True)
    sim, est0, anc = T.run_driver(driver, B, n, dtc, dtp, end, True, noise)
    x_plant = np.asarray(sim.xs)[0] if driver == "host" else np.asarray(sim.xs)[0, 0]
    o = T.oracle_rollout(noise, 0, est0[0], True, dtc, dtp, end, x_plant, ancestors=[a[0] for a in anc])
    got = T.records(sim, 0, B, driver)
    print(driver, n, dtc, dtp, "flips", o.ancestor_flips)
    for k in ("us", "xs", "xs_f", "ys_f", "covariance_point_size"):
        d = np.abs(np.asarray(got[k]) - np.asarray(o.__dict__[k]))
        print("   ", k, d.max(), "first tick > 1e-9:", int(np.argmax(d.reshape(len(d), -1).max(axis=1) > 1e-9)) if d.max() > 1e-9 else None)
