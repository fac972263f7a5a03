import sys; sys.path.insert(0,'/root/repo')
import numpy as np, ctypes as C
import bench, ct_icp_amd as cia
from ct_icp_amd import se3, synthetic as syn, _lib as L
inp = bench.make_inputs(0, 20)
gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(0.8,0.1,30)], default_radius=0.75))
gm.InsertPointCloud(inp["map_points"])
pose0 = syn.perturb_pose(inp["pose_gt"], 0.003, 0.03, seed=4)
world0 = se3.ct_transform(pose0, inp["tbe"], inp["t"], inp["raw"])
s = cia.GnSolver(gm); s.set_keypoints(inp["raw"], world0, inp["t"])
o = cia.CTICPOptions(solver=cia.GN, num_iters_icp=3, threshold_orientation_norm=0.0, debug_print=False)
from ct_icp_amd.registration import _c_options
co = _c_options(o); co.debug_print = 2
pose = pose0.copy(); tbe = np.ascontiguousarray(inp["tbe"]); summ = L.Summary(); dp = C.POINTER(C.c_double)
for _ in range(2):
    L.check(gm.handle, L.lib().ctgn_solve(gm.handle, pose.ctypes.data_as(dp), tbe.ctypes.data_as(dp), C.byref(co), None, C.byref(summ)))
