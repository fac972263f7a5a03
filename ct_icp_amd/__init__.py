"""ct_icp_amd — MI355X (gfx950) Gauss–Newton CT-ICP registration path, a drop-in for
ct_icp::CT_ICP_Registration with `solver: GN` (reference src/ct_icp/ct_icp.cpp:709-996) behind a C ABI
(include/ctgn.h, ct_icp_amd/libctgn.so). See DESIGN.md / INTEGRATION.md."""
from .types import (GN, CERES, ROBUST, CTICPOptions, ICPSummary, Pose, PreviousFrameMotionModel, TrajectoryFrame,
                    WPOINT3D_DTYPE, AdaptiveGridSamplingOptions)
from .map import GpuVoxelMap, GpuVoxelMapOptions, ResolutionParam
from .registration import (CT_ICP_Registration, FramePipeline, GnSolver, grid_sampling, transform_points, pinned_array,
                           AdaptiveSamplePointsInGrid)
from ._lib import CtgnError

__all__ = ["GN", "CERES", "ROBUST", "CTICPOptions", "ICPSummary", "Pose", "PreviousFrameMotionModel",
           "TrajectoryFrame", "WPOINT3D_DTYPE", "GpuVoxelMap", "GpuVoxelMapOptions", "ResolutionParam",
           "CT_ICP_Registration", "FramePipeline", "GnSolver", "grid_sampling", "transform_points", "pinned_array", "AdaptiveSamplePointsInGrid",
           "AdaptiveGridSamplingOptions", "CtgnError"]
