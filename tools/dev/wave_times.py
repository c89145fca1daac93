"""Spread of the run times of k_substeps' wavefronts inside one launch (MQE_WAVE_TIMES=1 -> mqe_debug_wave_times):
    python tools/dev/wave_times.py [task] [num_envs] [steps]
prints, for a few launches of a walking rollout, the percentiles of (exit - launch start) and of the wavefronts' own durations."""
import os, sys, ctypes as C
os.environ["MQE_WAVE_TIMES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "multiagent-quadruped-environment_amd")]
import numpy as np, torch
from helpers import make_desc, hip_engine
from mqe.engine import abi
task = sys.argv[1] if len(sys.argv) > 1 else "go1gate"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 120
d, k, _ = make_desc(task, n)
e = hip_engine(d, k)
e.reset_all()
Aw = e.tensor(abi.T_WRAPPER_OBS).shape[1]
g = torch.Generator(device="cuda"); g.manual_seed(1234)
buf = np.zeros((n, 4), np.int64)
prev_dur = None
for t in range(steps):
    e.step(torch.rand(n, Aw, 3, device="cuda", generator=g) * 2 - 1)
    if t == steps - 2:      # how well does a wavefront's duration in one step predict the next step's (load balancing by last step's load)?
        e._call("debug_wave_times", C.c_void_p(buf.ctypes.data))
        prev_dur = (buf[:, 1] - buf[:, 0]).astype(np.float64) * 0.01
    if t in (5, 30, 60, 90, steps - 1):
        e._call("debug_wave_times", C.c_void_p(buf.ctypes.data))
        nw = int((buf[:, 1] > 0).sum())
        b = buf[:nw].astype(np.float64) * 0.01            # 100 MHz -> us
        t0 = b[:, 0].min()
        end, dur, start = b[:, 1] - t0, b[:, 1] - b[:, 0], b[:, 0] - t0
        pc = lambda x: " ".join(f"{np.percentile(x, q):7.1f}" for q in (0, 10, 50, 90, 99, 100))
        hw, xcc = buf[:nw, 2], buf[:nw, 3] & 0xF
        simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
        key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)
        if t == steps - 1:
            uniq, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
            print("SIMDs used", len(uniq), "waves per SIMD: min", cnt.min(), "max", cnt.max())
            # do the waves of one SIMD leave together?  spread of exits inside a SIMD vs between SIMD means
            mean_exit = np.bincount(inv, weights=end) / cnt
            within = np.sqrt(np.bincount(inv, weights=(end - mean_exit[inv]) ** 2).sum() / nw)
            print(f"exit: std within a SIMD {within:.2f} us, std of SIMD means {mean_exit.std():.2f} us, std of all {end.std():.2f} us")
            lastexit = np.zeros(len(uniq)); np.maximum.at(lastexit, inv, end)
            print("last exit per SIMD percentiles [" + pc(lastexit) + "]")
            print(f"launch waits {lastexit.max() - lastexit.mean():.1f} us for its slowest SIMD beyond the mean SIMD (what a perfect balance of the envs over the SIMDs could win)")
            if prev_dur is not None:
                print(f"correlation of a wavefront's duration with its duration one step earlier: {np.corrcoef(prev_dur[:nw], dur)[0, 1]:.2f}")
                sumdur = np.bincount(inv, weights=dur)
                print(f"per SIMD: sum of its waves' durations vs its last exit: corr {np.corrcoef(sumdur, lastexit)[0, 1]:.2f}; max duration vs last exit: corr {np.corrcoef(np.maximum.reduceat(dur[np.argsort(inv, kind='stable')], np.r_[0, np.cumsum(cnt)[:-1]]), lastexit)[0, 1]:.2f}")
            for b in (0, 1, 2, 3, 8, 9, 1024, 2048, 3072):
                print("  block", b, "xcc", xcc[b], "se", se[b], "sh", sh[b], "cu", cu[b], "simd", simd[b], "wave slot", hw[b] & 15)
            # which block ids share a SIMD with block 0?
            print("  blocks on block 0's SIMD:", np.nonzero(key == key[0])[0].tolist(), " on block 8's:", np.nonzero(key == key[8])[0].tolist())
        print(f"step {t:3d} waves {nw}: start [{pc(start)}]  duration [{pc(dur)}]  exit [{pc(end)}] us  (min p10 p50 p90 p99 max)")
