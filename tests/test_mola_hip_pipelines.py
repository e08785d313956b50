"""Static proof that the reference-side binding can LOAD: the real stack is absent from this image, so
nothing here runs mola-lidar-odometry-cli -- the checks are the ones that failed statically in round 2:

  * the pipeline files handed to the real CLI (pipelines/make_mola_hip.py -> lidar3d-{default,ndt}-mola-hip.yaml) are the
    reference's files with ONLY the ICP class name changed: every key mola::LidarOdometry::initialize_frontend requires
    (module/src/LidarOdometry.cpp:246-483) is there, every class the files name is either one the reference file names
    itself (MRPT's factory has it once the stack is loaded) or one the adapter registers (MRPT_INITIALIZER /
    registerClass(CLASS_ID(...)), the pattern of module/src/register.cpp:40-46);
  * the adapter reads mola::HashedVoxelPointCloud / mola::NDT through their visitors (they are not CPointsMap), finds
    mola_metric_maps in its CMake file, and takes the two-matcher NDT shape (lidar3d-ndt.yaml:195-210);
  * INTEGRATION.md and tools/parity_pin.py point at the generated files;
  * the MOLA_HIP_* switches the adapter reads (molahip_host/plugin_switches.h, compiled into this repository's host
    layer) parse as documented.

/root/reference exists in the authoring container only; on a box without it the pre-generated files under
pipelines/generated/ (built by __graft_entry__.build(), git-ignored, shipped with the snapshot) are checked instead.
"""
import difflib
import importlib.util
import os
import re
import sys

import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADAPTERS = os.path.join(ROOT, "mola_lidar_odometry_amd", "host", "adapters")
REF_SRC = "/root/reference/module/src/LidarOdometry.cpp"

spec = importlib.util.spec_from_file_location("make_mola_hip", os.path.join(ROOT, "pipelines", "make_mola_hip.py"))
mk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mk)

# What initialize_frontend demands, as (block, key); block "" = top level of the file, "params" = c["params"].
# Written from LidarOdometry.cpp:261 (ENSURE_YAML_ENTRY_EXISTS lidar_sensor_labels), :278 (ASSERT_ isSequence),
# :297 (ASSERT_ has local_map_updates), :304 (YAML_LOAD_REQ min_icp_goodness), :336, :340 (ENSURE_YAML_ENTRY_EXISTS),
# :125-131 (AdaptiveThreshold::initialize, all REQ when the block exists), :197 (MultipleLidarOptions lidar_count REQ).
REQUIRED_STATIC = [("params", "lidar_sensor_labels"), ("", "insert_observation_into_local_map"), ("params", "local_map_updates"),
                   ("params", "min_icp_goodness"), ("", "navstate_fuse_params"), ("", "icp_settings_with_vel"),
                   ("params.adaptive_threshold", "enabled"), ("params.adaptive_threshold", "initial_sigma"),
                   ("params.adaptive_threshold", "min_motion"), ("params.adaptive_threshold", "kp"),
                   ("params.adaptive_threshold", "alpha"), ("params.multiple_lidars", "lidar_count")]


def required_from_reference_source():
    """The same list read off the reference's source (so that a newer snapshot's additions are caught): the macros inside
    initialize_frontend, where `cfg` is c["params"] and `c` the file's top level."""
    src = open(REF_SRC).read()
    body = src[src.index("void LidarOdometry::initialize_frontend"):src.index("void LidarOdometry::spinOnce")]
    req = []
    for var, key in re.findall(r'ENSURE_YAML_ENTRY_EXISTS\((\w+),\s*"(\w+)"\)', body):
        req.append(("params" if var == "cfg" else "", key))
    for var, key in re.findall(r'ASSERT_\((\w+)\.has\("(\w+)"\)\)', body):
        req.append(("params" if var == "cfg" else "", key))
    for key in re.findall(r'ASSERT_\(c\["(\w+)"\]\.isSequence\(\)\)', body):
        req.append(("", key))
    for key in re.findall(r"YAML_LOAD_REQ\(params_,\s*(\w+),", body):
        req.append(("params", key))
    return req


def _subst(text):
    """${VAR|default} -> default and $f{expr} -> expr, enough for PyYAML to read the structure (values are not evaluated)."""
    prev = None
    while prev != text:
        prev = text
        text = re.sub(r"\$\{[^{}|]*\|([^{}]*)\}", lambda m: m.group(1), text)
        text = re.sub(r"\$f\{([^{}]*)\}", lambda m: m.group(1), text)
    return text


def _class_names(node, out):
    if isinstance(node, dict):
        for k, v in node.items():
            if k in ("class", "class_name") and isinstance(v, str):
                out.add(v)
            _class_names(v, out)
    elif isinstance(node, list):
        for v in node:
            _class_names(v, out)
    return out


def _registered_by_adapter():
    names = set()
    for f in os.listdir(ADAPTERS):
        if f.endswith(".cpp"):
            names |= set(re.findall(r"registerClass\(CLASS_ID\(([\w:]+)\)\)", open(os.path.join(ADAPTERS, f)).read()))
    return names


@pytest.fixture(scope="module")
def pipelines(tmp_path_factory):
    """{name: (reference text or None, generated text)} for both files, plain and --device-map."""
    ref_dir = mk.find_reference_dir()
    out = {}
    if ref_dir:
        for dm in (False, True):
            d = tmp_path_factory.mktemp("gen_dm" if dm else "gen")
            mk.generate(ref_dir, str(d), device_map=dm)
            for name in mk.PIPELINES:
                g = name.replace(".yaml", "-mola-hip.yaml")
                out[(g, dm)] = (open(os.path.join(ref_dir, name)).read(), open(os.path.join(str(d), g)).read())
    else:
        gen = os.path.join(ROOT, "pipelines", "generated")
        for name in mk.PIPELINES:
            g = os.path.join(gen, name.replace(".yaml", "-mola-hip.yaml"))
            if os.path.exists(g):
                out[(os.path.basename(g), False)] = (None, open(g).read())
    if not out:
        pytest.skip("neither the reference pipelines nor pipelines/generated/ are present")
    return out


def test_generated_pipelines_differ_from_the_reference_only_in_class_lines(pipelines):
    checked = 0
    for (name, dm), (ref, gen) in pipelines.items():
        if ref is None:
            continue
        a, b = ref.splitlines(), gen.splitlines()
        assert len(a) == len(b), name
        changed = [(i + 1, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y]
        assert changed, name
        for no, old, new in changed:
            key = old.split(":", 1)[0].strip()
            assert key in (("class_name", "class", "plugin") if dm else ("class_name",)), (name, no, old, new)
            assert old.split(":", 1)[0] == new.split(":", 1)[0]  # same key, same indentation
        icp = [c for c in changed if c[1].strip().startswith("class_name")]
        assert len(icp) == 1 and icp[0][1].strip() == "class_name: mp2p_icp::ICP" and icp[0][2].strip() == "class_name: mp2p_icp::ICP_HIP"
        # what `diff` prints is exactly those lines
        d = [l for l in difflib.unified_diff(a, b, lineterm="", n=0) if l[:1] in "+-" and l[:3] not in ("+++", "---")]
        assert len(d) == 2 * len(changed)
        if not dm:
            assert len(changed) == 1, (name, changed)
        elif name.startswith("lidar3d-default"):
            assert len(changed) == 3 and any("mola::HashedVoxelPointCloudHIP" in c[2] for c in changed) \
                and any("libmolahip_mp2p_icp.so" in c[2] for c in changed)
        checked += 1
    if not checked:
        pytest.skip("reference pipelines not present: nothing to diff against")


def test_every_key_the_reference_frontend_requires_is_present(pipelines):
    required = list(REQUIRED_STATIC)
    if os.path.exists(REF_SRC):
        from_src = required_from_reference_source()
        assert {("params", "lidar_sensor_labels"), ("", "navstate_fuse_params"), ("", "icp_settings_with_vel"),
                ("", "insert_observation_into_local_map"), ("params", "min_icp_goodness"), ("params", "local_map_updates")} <= set(from_src)
        required += from_src
    for (name, dm), (_, gen) in pipelines.items():
        cfg = yaml.safe_load(_subst(gen))
        for block, key in required:
            node = cfg
            for part in [p for p in block.split(".") if p]:
                assert part in node, "%s: block %r missing" % (name, block)
                node = node[part]
            assert key in node, "%s: %s%s missing (LidarOdometry.cpp:246-483 requires it)" % (name, block + "." if block else "", key)
        assert isinstance(cfg["insert_observation_into_local_map"], list) and cfg["insert_observation_into_local_map"]
        labels = cfg["params"]["lidar_sensor_labels"]
        assert (isinstance(labels, list) and labels) or isinstance(labels, str)
        # the sections the front end warns about when absent (:374-456) are all there too
        for sec in ("observations_generator", "observations_filter_adjust_timestamps", "observations_filter_1st_pass",
                    "observations_filter_2nd_pass", "observations_filter_final_pass", "localmap_generator"):
            assert sec in cfg, (name, sec)
        # the local map's plugin key survives (round 2's -hip file had dropped it)
        md = cfg["localmap_generator"][0]["params"]["metric_map_definition"]
        assert md.get("plugin"), name


def test_every_class_name_resolves_to_upstream_or_to_what_the_plugin_registers(pipelines):
    registered = _registered_by_adapter()
    assert "mp2p_icp::ICP_HIP" in registered and "mola::HashedVoxelPointCloudHIP" in registered
    for (name, dm), (ref, gen) in pipelines.items():
        names = _class_names(yaml.safe_load(_subst(gen)), set())
        upstream = _class_names(yaml.safe_load(_subst(ref)), set()) if ref is not None else \
            {n for n in names if n.startswith(("mp2p_icp::", "mp2p_icp_filters::", "mola::")) and not n.endswith("HIP")}
        for n in names:
            assert n in upstream or n in registered, "%s names %r: neither an upstream class of the reference file nor registered by the plugin" % (name, n)
        assert not any(n.startswith("mp2p_icp_hip::") for n in names), "mirror-namespace names can never be in MRPT's factory"
        assert "mp2p_icp::ICP_HIP" in names and "mp2p_icp::ICP" not in names
        if dm and name.startswith("lidar3d-default"):
            assert "mola::HashedVoxelPointCloudHIP" in names
        # solver / matchers / quality stay upstream classes: the plugin reads THEIR parsed parameters
        icp = yaml.safe_load(_subst(gen))["icp_settings_with_vel"]
        assert [s["class"] for s in icp["solvers"]] == ["mp2p_icp::Solver_GaussNewton"]
        assert [m["class"] for m in icp["matchers"]][-1] == "mp2p_icp::Matcher_Points_DistanceThreshold"
        assert [q["class"] for q in icp["quality"]] == ["mp2p_icp::QualityEvaluator_PairedRatio"]


def test_adapter_sources_match_what_the_pipelines_need():
    # the ICP_HIP adapter = mp2p_icp_plugin.cpp + the helpers it shares with the granular classes (molahip_mrpt_common.h)
    plugin = open(os.path.join(ADAPTERS, "mp2p_icp_plugin.cpp")).read() + "\n" + open(os.path.join(ADAPTERS, "molahip_mrpt_common.h")).read()
    code = "\n".join(l.split("//", 1)[0] for l in plugin.splitlines())  # comments stripped
    # registration under the one name the generated files use, parent = the upstream ICP
    assert re.search(r"IMPLEMENTS_MRPT_OBJECT\(ICP_HIP,\s*mp2p_icp::ICP,\s*mp2p_icp\)", code)
    assert "registerClass(CLASS_ID(mp2p_icp::ICP_HIP))" in code
    assert "mp2p_icp_hip::" not in code
    # the default map is read through its visitors, with its own header, never asserted to be a CPointsMap
    assert "#include <mola_metric_maps/HashedVoxelPointCloud.h>" in plugin and "#include <mola_metric_maps/NDT.h>" in plugin
    assert "visitAllVoxels" in code and "visitAllPoints" in code
    assert not re.search(r"dynamic_cast<const mrpt::maps::CPointsMap\*>\(&g\);\s*ASSERT_", code)
    assert "dynamic_cast<const mola::HashedVoxelPointCloud*>" in code and "dynamic_cast<const mola::NDT*>" in code
    # an unknown map class or pipeline shape goes to the upstream loop instead of throwing
    assert code.count("return upstream_align(pcLocal, pcGlobal, initialGuessLocalWrtGlobal, p, result, prior, outputDebugInfo);") >= 2
    assert "ICP::align(pcLocal, pcGlobal, guess, p, result, prior, outputDebugInfo);" in code
    # the two-matcher NDT shape reaches the device: Matcher_Point2Plane first, its threshold schedule, its pairings back
    assert "#include <mp2p_icp/Matcher_Point2Plane.h>" in plugin
    for needle in ("dynamic_cast<const Matcher_Point2Plane*>(ms[0].get())", "q.pt2pl_threshold", "ndt_max_eigen_ratio",
                   "mh_icp_get_pt2pl_pairs", "paired_pt2pl"):
        assert needle in code, needle
    # the App. B switches come from the shared header
    assert '#include "molahip_host/plugin_switches.h"' in plugin
    for needle in ("sw.index_mode", "sw.cov_step_xyz", "sw.min_delta", "sw.pt2pl_mode", "kernel_from_upstream_name", "sw.force_cpu"):
        assert needle in code, needle
    cm = open(os.path.join(ADAPTERS, "CMakeLists.txt")).read()
    assert "find_package(mola_metric_maps REQUIRED)" in cm and "find_package(mp2p_icp REQUIRED)" in cm
    assert re.search(r"target_link_libraries\([^)]*mola::mola_metric_maps", cm)
    # every C-ABI function the adapters call exists in the header with that name
    header = open(os.path.join(ROOT, "include", "molahip.h")).read()
    declared = set(re.findall(r"MH_API\s+[\w\s\*]+?\b(mh_\w+)\s*\(", header))
    for f in os.listdir(ADAPTERS):
        if f.endswith((".cpp", ".h")):
            src = "\n".join(l.split("//", 1)[0] for l in open(os.path.join(ADAPTERS, f)).read().splitlines())
            for fn in set(re.findall(r"\b(mh_[a-z0-9_]+)\s*\(", src)):
                if fn in ("mh_check",):
                    continue
                assert fn in declared, "%s calls %s, which include/molahip.h does not declare" % (f, fn)


def test_granular_pipelines_name_the_device_matcher_and_solver_classes(tmp_path):
    """--granular (BASELINE north_star: "keeping the mp2p_icp::ICP / Matcher / Solver plugin API"): upstream ICP loop, the
    solver / matcher `class:` lines name what host/adapters/mp2p_icp_granular.cpp registers; nothing else changes."""
    ref_dir = mk.find_reference_dir()
    if not ref_dir:
        pytest.skip("no reference pipelines on this box")
    registered = _registered_by_adapter()
    for n in ("mp2p_icp::Matcher_Points_DistanceThreshold_HIP", "mp2p_icp::Matcher_Point2Plane_HIP", "mp2p_icp::Solver_GaussNewton_HIP"):
        assert n in registered, n
    extra = ["extras/lidar3d-dual-map.yaml", "extras/lidar3d-near-far.yaml"]  # shapes the fused loop does not take
    mk.generate(ref_dir, str(tmp_path), granular=True, pipelines=list(mk.PIPELINES) + extra)
    for name in list(mk.PIPELINES) + extra:
        ref = open(os.path.join(ref_dir, name)).read()
        gen = open(os.path.join(str(tmp_path), os.path.basename(name).replace(".yaml", "-mola-hip-granular.yaml"))).read()
        rl, gl = ref.splitlines(), gen.splitlines()
        assert len(rl) == len(gl)
        changed = [(a, b) for a, b in zip(rl, gl) if a != b]
        assert changed and all(re.fullmatch(r"\s+-?\s*class:\s*\S+", a.split("#")[0].rstrip()) for a, _ in changed), changed
        y = yaml.safe_load(_subst(gen))
        names = _class_names(y, set())
        upstream = _class_names(yaml.safe_load(_subst(ref)), set())
        assert all(n in upstream or n in registered for n in names), names - upstream - registered
        icp = y["icp_settings_with_vel"]
        assert icp["class_name"] == "mp2p_icp::ICP"  # upstream's loop: gates, hooks, quality evaluators are its own
        assert all(s["class"].endswith("_HIP") for s in icp["solvers"])
        assert all(m["class"].endswith("_HIP") for m in icp["matchers"] if "Matcher_Points_DistanceThreshold" in m["class"] or
                   "Matcher_Point2Plane" in m["class"])
    # and the classes derive from the upstream ones (YAML parameters parsed by upstream's initialize())
    g = open(os.path.join(ADAPTERS, "mp2p_icp_granular.cpp")).read()
    for cls, base in (("Matcher_Points_DistanceThreshold_HIP", "Matcher_Points_DistanceThreshold"), ("Matcher_Point2Plane_HIP", "Matcher_Point2Plane"),
                      ("Solver_GaussNewton_HIP", "Solver_GaussNewton")):
        assert re.search(r"class %s : public %s\b" % (cls, base), g)
        assert re.search(r"IMPLEMENTS_MRPT_OBJECT\(%s,\s*mp2p_icp::%s,\s*mp2p_icp\)" % (cls, base), g)
    for call in ("mh_nn_search_k(", "mh_nn_search_pt2pl(", "mh_gn_solve("):
        assert call in g
    cm = open(os.path.join(ADAPTERS, "CMakeLists.txt")).read()
    assert "mp2p_icp_granular.cpp" in cm


def test_docs_and_tools_point_at_the_generated_files():
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "pipelines/make_mola_hip.py" in integ and "lidar3d-default-mola-hip.yaml" in integ and "lidar3d-ndt-mola-hip.yaml" in integ
    joined = integ.replace("\\\n", " ")  # shell line continuations
    cmds = [l for l in joined.splitlines() if l.startswith("mola-lidar-odometry-cli -l")]
    assert len(cmds) >= 2 and all(re.search(r"-c \S*-mola-hip(-granular)?\.yaml", l) for l in cmds), cmds
    assert not re.search(r"mola-lidar-odometry-cli[^\n]*-c pipelines/lidar3d-(default|ndt)-hip\.yaml", joined)
    pin = open(os.path.join(ROOT, "tools", "parity_pin.py")).read()
    assert "lidar3d-default-mola-hip.yaml" in pin and "make_mola_hip" in pin


def test_build_generates_the_pipelines_when_the_reference_is_present():
    if not mk.find_reference_dir():
        pytest.skip("no reference pipelines on this box")
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "make_mola_hip" in entry
    assert "pipelines/generated/" in open(os.path.join(ROOT, ".gitignore")).read()


def test_plugin_switches_parse_as_documented(monkeypatch):
    sys.path.insert(0, ROOT)
    from mola_lidar_odometry_amd import _mp2p_icp_hip as h
    for v in ("MOLA_HIP_ROBUST_KERNEL", "MOLA_HIP_INDEX_MODE", "MOLA_HIP_COV_STEP_XYZ", "MOLA_HIP_COV_STEP_ANG", "MOLA_HIP_MIN_DELTA",
              "MOLA_HIP_MAX_COST", "MOLA_HIP_PT2PL_MODE", "MOLA_HIP_FAR_VOXEL_METRIC", "MOLA_HIP_FORCE_CPU", "MOLA_HIP_MATCHED_POINTS"):
        monkeypatch.delenv(v, raising=False)
    try:
        h.reload_plugin_switches()
        d = h.plugin_switches()
        assert d == dict(gm_form=1, index_mode=0, cov_step_xyz=1e-7, cov_step_ang=1e-7, min_delta=1e-7, max_cost=0.0, pt2pl_mode=0,
                         matched_points=0, far_voxel_metric=0, force_cpu=False)
        # upstream enumerator NAMES -> MH_KERNEL_*; "GemanMcClure" follows the switch, the others do not
        assert h.kernel_from_upstream_name("GemanMcClure") == 1 and h.kernel_from_upstream_name("RobustKernel::Cauchy") == 4
        assert h.kernel_from_upstream_name("None") == 0
        monkeypatch.setenv("MOLA_HIP_ROBUST_KERNEL", "GemanMcClure_KISS")
        monkeypatch.setenv("MOLA_HIP_INDEX_MODE", "trunc")
        monkeypatch.setenv("MOLA_HIP_COV_STEP_XYZ", "1e-6")
        monkeypatch.setenv("MOLA_HIP_MIN_DELTA", "1e-9")
        monkeypatch.setenv("MOLA_HIP_PT2PL_MODE", "centroid")
        monkeypatch.setenv("MOLA_HIP_FAR_VOXEL_METRIC", "l1")
        monkeypatch.setenv("MOLA_HIP_FORCE_CPU", "1")
        h.reload_plugin_switches()
        d = h.plugin_switches()
        assert (d["gm_form"], d["index_mode"], d["cov_step_xyz"], d["min_delta"], d["pt2pl_mode"], d["far_voxel_metric"], d["force_cpu"]) == \
            (2, 1, 1e-6, 1e-9, 1, 1, True)
        assert h.kernel_from_upstream_name("RobustKernel::GemanMcClure") == 2 and h.kernel_from_upstream_name("Cauchy") == 4
        # termination reasons travel by NAME
        assert [h.term_reason_name(t) for t in range(7)] == ["Undefined", "NoPairings", "SolverError", "MaxIterations", "Stalled",
                                                              "QualityCheckpointFailed", "HookRequest"]
    finally:
        for v in ("MOLA_HIP_ROBUST_KERNEL", "MOLA_HIP_INDEX_MODE", "MOLA_HIP_COV_STEP_XYZ", "MOLA_HIP_MIN_DELTA", "MOLA_HIP_PT2PL_MODE",
                  "MOLA_HIP_FAR_VOXEL_METRIC", "MOLA_HIP_FORCE_CPU"):
            monkeypatch.delenv(v, raising=False)
        h.reload_plugin_switches()
