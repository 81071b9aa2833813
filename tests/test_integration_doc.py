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


def _gxx_syntax(args, src_text=None, path=None):
    with tempfile.TemporaryDirectory() as d:
        if src_text is not None:
            path = os.path.join(d, "check.cpp")
            open(path, "w").write(src_text)
        return subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include")] + args + [path],
                              capture_output=True, text=True)


def test_adaptor_header_and_demo_compile_against_the_stand_in_types():
    """include/lio_hip_adaptor.hpp + examples/adaptor_demo.cpp (the reference's globals and call sites, statement by
    statement) against tests/csrc/adaptor_stubs: a renamed entry point or a changed argument list in lio_hip.h, or a
    facade member that no longer matches a call site of src/laserMapping.cpp, breaks this test, not a user."""
    r = _gxx_syntax(["-I", os.path.join(ROOT, "tests", "csrc", "adaptor_stubs")], path=os.path.join(ROOT, "examples", "adaptor_demo.cpp"))
    assert r.returncode == 0, r.stderr


def test_adaptor_compiles_against_the_real_headers_when_they_exist():
    """Where Eigen, PCL and the reference checkout are installed the same demo is built against the REAL state_ikfom,
    dyn_share_datastruct<double> and pcl::PointXYZINormal (-DLIO_ADAPTOR_REAL_HEADERS).  Not in this image."""
    import glob
    import pytest
    eigen = [p for p in ("/usr/include/eigen3", "/usr/local/include/eigen3") if os.path.exists(os.path.join(p, "Eigen", "Core"))]
    pcl = sorted(glob.glob("/usr/include/pcl-*")) + sorted(glob.glob("/usr/local/include/pcl-*"))
    ref = "/root/reference/include"
    if not (eigen and pcl and os.path.exists(ref)):
        pytest.skip("Eigen / PCL / the reference headers are not installed: the adaptor is checked against stand-ins only")
    r = _gxx_syntax(["-DLIO_ADAPTOR_REAL_HEADERS", "-I", eigen[0], "-I", pcl[-1], "-I", ref, "-I", os.path.join(ref, "IKFoM_toolkit")],
                    path=os.path.join(ROOT, "examples", "adaptor_demo.cpp"))
    assert r.returncode == 0, r.stderr


def test_the_edits_quoted_in_integration_md_compile_with_the_adaptor():
    """Section 3's diff of src/laserMapping.cpp, pasted behind the globals it names."""
    blocks = _cpp_blocks()
    edits = [b for b in blocks if "lio::KdTreeFacade<PointType> ikdtree;" in b]
    assert len(edits) == 1
    body = edits[0]
    # statements (c) and (d) live inside functions of the node
    head, tail = body.split("// (c)")
    src = "\n".join([
        "#include <memory>",
        '#include "use-ikfom.hpp"',
        "#include <ikd-Tree/ikd_Tree.h>",
        "using std::vector;",
        "struct PointCloudXYZI { typedef std::shared_ptr<StubCloud> Ptr; };",
        "static PointCloudXYZI::Ptr feats_down_body, feats_down_world, normvec, laserCloudOri, corr_normvect;",
        "static vector<PointVector> Nearest_Points;",
        "static bool point_selected_surf[100000], extrinsic_est_en;",
        "static float res_last[100000];",
        "static int feats_down_size, effct_feat_num;",
        "static double res_mean_last;",
        head,
        "void node_body() {",
        "// (c)" + tail,
        "}",
        "int main() { return 0; }",
    ])
    r = _gxx_syntax(["-I", os.path.join(ROOT, "tests", "csrc", "adaptor_stubs")], src_text=src)
    assert r.returncode == 0, r.stderr


def test_every_abi_call_in_integration_md_exists_in_the_header():
    header = open(os.path.join(ROOT, "include", "lio_hip.h")).read() + open(os.path.join(ROOT, "include", "lio_hip_adaptor.hpp")).read()
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
