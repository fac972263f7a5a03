"""The drop-in boundary, compiled and run inside the reference's own code.

integration/gpu_map.h (GpuVoxelMap : ct_icp::ISlamMap) and integration/gn_gpu_arm.h (the GPU arms of DoRegisterGaussNewton /
DoRegisterCeres) are the files INTEGRATION.md tells a maintainer to add. `make -C oracle glue` compiles them VERBATIM against the
reference's headers (include/ct_icp/{ct_icp,map,motion_model}.h, SlamCore/...; third-party headers from oracle/shims/), compiles the
reference's ct_icp.cpp with the two documented one-line insertions, and links everything with libctgn.so into oracle/_ref/glue_check.

CPU (here, where /root/reference exists): the build succeeds, the documented insertions are the only difference to the reference's file,
INTEGRATION.md quotes the glue files verbatim, and the program reports "no-device" (libctgn has no CPU fallback).
GPU (`-m gpu`, the prebuilt binary travels with the snapshot): the reference's CT_ICP_Registration::Register gives the same poses on
GpuVoxelMap (GPU arm) as on its own MultipleResolutionVoxelMap (its CPU loop), for solver GN and CERES, and handles FLOAT32 views."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
BIN = os.path.join(ROOT, "oracle", "_ref", "glue_check")
have_ref = os.path.isdir(os.path.join(REF, "src", "ct_icp"))


@pytest.mark.skipif(not have_ref, reason="/root/reference absent: the glue is compiled where the reference's headers are")
def test_glue_compiles_against_the_reference_headers_and_links():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "ct_icp_amd", "csrc"), "all"])
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "oracle"), "_ref"])
    subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle"), "_ref/obj/glue_check.o", "_ref/obj/ct_icp_with_gpu_arms.o"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "glue"])
    assert os.path.exists(BIN)
    # what the compiler is fed (make glue-print: never written to disk) differs from the reference's file by exactly the three documented lines
    ref_lines = open(os.path.join(REF, "src", "ct_icp", "ct_icp.cpp")).read().splitlines()
    new_lines = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "glue-print"], capture_output=True, text=True,
                               check=True).stdout.splitlines()
    assert not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "obj", "ct_icp_with_gpu_arms.cpp"))
    added = [l.strip() for l in new_lines if l not in ref_lines]
    assert len(new_lines) == len(ref_lines) + 3 and len(added) == 3
    assert added[0] == "#include <ct_icp/gn_gpu_arm.h>"
    assert added[1].startswith("if (auto gpu = GpuCeres(") and added[2].startswith("if (auto gpu = GpuGaussNewton(")
    arm_doc = open(os.path.join(ROOT, "integration", "gn_gpu_arm.h")).read()
    for a in added[1:]:
        assert a in arm_doc                                   # the statements are the ones the header documents
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and ("glue no-device" in out.stdout or "glue ALL OK" in out.stdout), out.stdout + out.stderr


def test_integration_md_quotes_the_glue_files_verbatim():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```cpp\n(.*?)```", md, flags=re.S)
    for name in ("gpu_map.h", "gn_gpu_arm.h", "odometry_gpu_arm.h"):
        src = open(os.path.join(ROOT, "integration", name)).read()
        assert any(b.strip() == src.strip() for b in blocks), f"INTEGRATION.md must embed integration/{name} verbatim"
    # the real ProxyView members (SlamCore/data/view.h:113-116,186-189), not invented accessors
    assert "item_buffer.view_data_ptr + v.offset_in_item" in md and "src_property_type" in md
    for invented in ("view_data_ptr()", "item_size()", "src_type()"):
        assert invented not in md


@pytest.mark.gpu
def test_reference_register_dispatches_to_the_gpu_and_matches_its_cpu_loop():
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/glue_check was not built (needs /root/reference at build time)")
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    for case in ("glue insert ok", "glue neighbourhoods ok", "glue GN ok", "glue CERES ok", "glue PointCloud(float32 views) ok", "glue ALL OK"):
        assert case in out.stdout, out.stdout
