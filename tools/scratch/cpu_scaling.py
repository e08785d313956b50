"""Development probe: OpenMP scaling of the CPU oracle on the box it runs on."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mola_lidar_odometry_amd import synth
from oracle import oracle_c as oc
w = synth.workload_c2()
m = oc.Map(1.0, 20).insert(w.map_xyz)
print("nproc", os.cpu_count(), "omp max", oc.max_threads())
for nt in (1, 8, 16, 32, 64, 128):
    t = time.time(); r = oc.match_points(m, w.scan_xyz, w.T_guess, w.threshold[0], 0.0, nt); dm = time.time() - t
    p = oc.ICPParams(max_iterations=20, disable_stall_test=True, threshold=w.threshold, kernel_param=w.kernel_param)
    t = time.time(); res = oc.icp_align(m, w.scan_xyz, w.T_guess, p, n_threads=nt); da = time.time() - t
    print(f"threads {nt:4d}: match {dm*1e3:8.1f} ms   align {da*1e3:8.1f} ms  -> {1/da:6.2f} scans/s")
