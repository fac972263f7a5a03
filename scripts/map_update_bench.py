"""Map-update throughput: host mirror + delta upload vs device-resident maintenance, same frames."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import bench, ct_icp_amd as cia
inp = bench.make_inputs(0, 20)
frames = np.split(inp["map_points"], np.cumsum(inp["map_counts"])[:-1])
for dev in (False, True):
    gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(0.8, 0.1, 30)], default_radius=0.75, device_updates=dev))
    gm.InsertPointCloud(frames[0]); gm.Sync()
    t0 = time.perf_counter()
    n = 0
    for f in frames[1:]:
        gm.InsertPointCloud(f)
        gm.RemoveElementsFarFromLocation(f.mean(axis=0), 100.0)
        gm.Sync()
        n += len(f)
    dt = time.perf_counter() - t0
    print(f"device_updates={dev}: {len(frames) - 1} frames, {n} points, {dt * 1e3 / (len(frames) - 1):.3f} ms/frame, {n / dt / 1e6:.2f} Mpts/s, map {gm.NumPoints()} pts")
