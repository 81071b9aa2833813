"""INTEGRATION.md shows the C++ adaptor a maintainer adds to the reference node.  Eigen/PCL/ROS are not in
this image, so the adaptor cannot be built against the reference -- but it can be type-checked against the
real include/lio_hip.h with stand-in types (tests/csrc/adaptor_stubs): a renamed entry point or a changed
argument list in the header breaks this test, not a user."""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cpp_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.findall(r"```cpp\n(.*?)```", text, re.S)


def test_adaptor_of_integration_md_type_checks_against_the_header():
    blocks = _cpp_blocks()
    adaptor = [b for b in blocks if "struct LioHip" in b]
    model = [b for b in blocks if "void h_share_model(" in b]
    assert len(adaptor) == 1 and len(model) == 1
    src = "\n".join([
        adaptor[0],
        "// file-scope globals of src/laserMapping.cpp the callback reaches (laserMapping.cpp:44-116)",
        "static bool scan_dirty = true, extrinsic_est_en = false;",
        "static StubCloud* feats_down_body = nullptr;",
        "static int feats_down_size = 0, effct_feat_num = 0;",
        "static double res_mean_last = 0.0;",
        "static LioHip ikdtree(0.2f);",
        model[0],
        "int main() { return 0; }",
    ])
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "adaptor_check.cpp")
        open(path, "w").write(src)
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I", os.path.join(ROOT, "include"),
                            "-I", os.path.join(ROOT, "tests", "csrc", "adaptor_stubs"), path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_every_abi_call_in_integration_md_exists_in_the_header():
    header = open(os.path.join(ROOT, "include", "lio_hip.h")).read()
    declared = set(re.findall(r"\b(lio_[A-Za-z0-9_]+)\s*\(", header))
    used = set(re.findall(r"\b(lio_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "INTEGRATION.md")).read()))
    assert used and used <= declared, sorted(used - declared)


def test_standalone_loop_of_integration_md_type_checks_as_c():
    """Section 3c's per-scan loop (IMU front end -> voxel filter -> update -> map_incremental) is plain C against
    include/lio_hip.h: wrapped into a function with the variables it names, it must compile."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```c\n(.*?)```", text, re.S)
    loop = [b for b in blocks if "lio_imu_process(" in b]
    assert len(loop) == 1
    src = "\n".join([
        "#include <stddef.h>",
        "#include <lio_hip.h>",
        "void one_scan(lio_imu* imu, lio_eskf* kf, lio_ctx* ctx, double lidar_beg_time, double lidar_end_time,",
        "              const float* raw_pts, int n_raw, int stride, int time_offset, const lio_imu_sample* imu_samples, int n_imu,",
        "              float filter_size_surf_min, double filter_size_map_min, double LASER_POINT_COV, int extrinsic_est_en) {",
        "    int ready = 0, feats_down_size = 0, n_add = 0, n_nodown = 0, add_counter = 0;",
        "    double x26[26];",
        "    lio_update_log log;",
        loop[0],
        "    (void)feats_down_size; (void)n_add; (void)n_nodown; (void)add_counter; (void)log;",
        "}",
    ])
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "loop_check.c")
        open(path, "w").write(src)
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), path],
                           capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
