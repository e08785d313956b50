#!/usr/bin/env python3
"""make_mola_hip.py -- derive the pipeline files the REAL mola-lidar-odometry-cli takes with the libmolahip plugin.

    python pipelines/make_mola_hip.py [--ref-dir <mola_lidar_odometry>/pipelines] [--out-dir pipelines/generated]
                                      [--device-map] [--granular]

For each of the reference's lidar3d-default.yaml / lidar3d-ndt.yaml it writes <name>-mola-hip.yaml: the reference file
AS IT IS -- every section mola::LidarOdometry::initialize_frontend reads (module/src/LidarOdometry.cpp:246-483:
params.lidar_sensor_labels :261, insert_observation_into_local_map :278, params.local_map_updates :297,
params.min_icp_goodness :304, navstate_fuse_params :336, icp_settings_with_vel :340, the generators and filter chains
:374-456), every class name MRPT's factory already knows -- with ONE kind of line changed:

    icp_settings_with_vel / icp_settings_without_vel:   class_name: mp2p_icp::ICP  ->  class_name: mp2p_icp::ICP_HIP

mp2p_icp::ICP_HIP is what host/adapters/mp2p_icp_plugin.cpp registers (the pattern of module/src/register.cpp:40-46);
solvers, matchers and quality evaluators stay the upstream classes (the plugin derives from mp2p_icp::ICP and reads their
parsed parameters), so does every filter and generator.  With --device-map the local map of lidar3d-default.yaml:230-231
becomes the device-owned class of host/adapters/hashed_voxel_pointcloud_hip.h (`class:` + `plugin:` lines).

With --granular it ALSO writes <name>-mola-hip-granular.yaml: the ICP class stays upstream's mp2p_icp::ICP (its host
loop, its iteration gates, its quality evaluators), and the `class:` lines of the solvers / matchers blocks name the
device classes of host/adapters/mp2p_icp_granular.cpp instead:

    mp2p_icp::Solver_GaussNewton                -> mp2p_icp::Solver_GaussNewton_HIP
    mp2p_icp::Matcher_Points_DistanceThreshold  -> mp2p_icp::Matcher_Points_DistanceThreshold_HIP
    mp2p_icp::Matcher_Point2Plane               -> mp2p_icp::Matcher_Point2Plane_HIP

(BASELINE.json north_star: "keeping the mp2p_icp::ICP / Matcher / Solver plugin API"; class names at
lidar3d-default.yaml:185,196 and lidar3d-ndt.yaml:185,195,202).  The same substitution works on ANY pipeline built from
these classes (--pipelines extras/lidar3d-dual-map.yaml ...): that is the path for shapes the fused loop does not take.

Why a generator instead of committed copies: the reference's files are not copied into this repository (they are its
sources), and a maintainer's installed mola_lidar_odometry may be newer than the snapshot this was written against --
the rules below follow whatever that version's files say.  The unverified-upstream switches are NOT
pipeline edits in plugin mode: the upstream solver parses `robustKernel` with its own enum, so the plugin reads them from
the environment (MOLA_HIP_*: mp2p_icp_plugin.cpp, tools/parity_pin.py).

The stand-alone driver (molahip-lo-cli, run_odometry.py) does not use these files: it reads pipelines/lidar3d-*-hip.yaml
or the reference's own files.  tests/test_mola_hip_pipelines.py checks the output statically.
"""
import argparse
import hashlib
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PIPELINES = ("lidar3d-default.yaml", "lidar3d-ndt.yaml")
ICP_BLOCKS = ("icp_settings_with_vel", "icp_settings_without_vel")
ICP_CLASS, ICP_CLASS_HIP = "mp2p_icp::ICP", "mp2p_icp::ICP_HIP"
MAP_CLASS, MAP_CLASS_HIP = "mola::HashedVoxelPointCloud", "mola::HashedVoxelPointCloudHIP"
PLUGIN_SO = "libmolahip_mp2p_icp.so"
GRANULAR = {"mp2p_icp::Solver_GaussNewton": "mp2p_icp::Solver_GaussNewton_HIP",
            "mp2p_icp::Matcher_Points_DistanceThreshold": "mp2p_icp::Matcher_Points_DistanceThreshold_HIP",
            "mp2p_icp::Matcher_Point2Plane": "mp2p_icp::Matcher_Point2Plane_HIP"}

_top_key = re.compile(r"^([A-Za-z_][\w]*)\s*:")


def find_reference_dir(explicit=None):
    """The directory holding the reference's pipelines: --ref-dir, $MOLA_LO_PIPELINES_DIR, an installed
    mola_lidar_odometry (ament / catkin share directory), or this project's read-only reference checkout."""
    cands = [explicit, os.environ.get("MOLA_LO_PIPELINES_DIR")]
    for prefix in os.environ.get("AMENT_PREFIX_PATH", "").split(os.pathsep) + os.environ.get("CMAKE_PREFIX_PATH", "").split(os.pathsep):
        if prefix:
            cands.append(os.path.join(prefix, "share", "mola_lidar_odometry", "pipelines"))
    cands.append("/root/reference/pipelines")
    for c in cands:
        if c and all(os.path.isfile(os.path.join(c, p)) for p in PIPELINES):
            return c
    return None


def transform(text, device_map=False, granular=False):
    """-> (new text, [(line number, old line, new line)]).  Line-level, so comments and layout stay untouched.
    granular: the ICP class stays upstream's, the solver / matcher `class:` lines inside the ICP blocks get the _HIP names."""
    out, changes, block = [], [], None
    for no, line in enumerate(text.splitlines(keepends=True), 1):
        m = _top_key.match(line)
        if m:
            block = m.group(1)
        new = line
        body = line.split("#", 1)[0].rstrip()
        if granular and block in ICP_BLOCKS:
            m2 = re.fullmatch(r"\s+-?\s*class:\s*['\"]?([\w:]+)['\"]?", body)
            if m2 and m2.group(1) in GRANULAR:
                new = line.replace(m2.group(1), GRANULAR[m2.group(1)], 1)
        elif block in ICP_BLOCKS and re.fullmatch(r"\s+class_name:\s*['\"]?%s['\"]?" % re.escape(ICP_CLASS), body):
            new = line.replace(ICP_CLASS, ICP_CLASS_HIP, 1)
        elif device_map and block == "localmap_generator":
            if re.fullmatch(r"\s+class:\s*['\"]?%s['\"]?" % re.escape(MAP_CLASS), body):
                new = line.replace(MAP_CLASS, MAP_CLASS_HIP, 1)
            elif re.match(r"\s+plugin:\s*", body) and any(MAP_CLASS_HIP in c[2] for c in changes):
                indent = line[: len(line) - len(line.lstrip())]
                new = "%splugin: '%s'\n" % (indent, PLUGIN_SO)
        if new != line:
            changes.append((no, line, new))
        out.append(new)
    return "".join(out), changes


def generate(ref_dir, out_dir, device_map=False, granular=False, pipelines=PIPELINES):
    os.makedirs(out_dir, exist_ok=True)
    report = {}
    for name in pipelines:
        src = os.path.join(ref_dir, name)
        text = open(src, encoding="utf-8").read()
        new, changes = transform(text, device_map)
        if not any(ICP_CLASS_HIP in c[2] for c in changes):
            raise RuntimeError("%s: no 'class_name: %s' line inside %s -- the reference layout changed, update the rules"
                               % (src, ICP_CLASS, " / ".join(ICP_BLOCKS)))
        dst = os.path.join(out_dir, os.path.basename(name).replace(".yaml", "-mola-hip.yaml"))
        with open(dst, "w", encoding="utf-8") as f:
            f.write(new)
        report[dst] = {"source": src, "source_sha256": hashlib.sha256(text.encode()).hexdigest(),
                       "changed_lines": [c[0] for c in changes]}
        if granular:
            new_g, changes_g = transform(text, device_map, granular=True)
            if not any(v in c[2] for c in changes_g for v in GRANULAR.values()):
                raise RuntimeError("%s: no solver / matcher class line to substitute inside %s" % (src, " / ".join(ICP_BLOCKS)))
            dst_g = os.path.join(out_dir, os.path.basename(name).replace(".yaml", "-mola-hip-granular.yaml"))
            with open(dst_g, "w", encoding="utf-8") as f:
                f.write(new_g)
            report[dst_g] = {"source": src, "source_sha256": hashlib.sha256(text.encode()).hexdigest(),
                             "changed_lines": [c[0] for c in changes_g]}
    return report


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ref-dir")
    ap.add_argument("--out-dir", default=os.path.join(HERE, "generated"))
    ap.add_argument("--device-map", action="store_true",
                    help="also select mola::HashedVoxelPointCloudHIP (device-owned local map) in lidar3d-default")
    ap.add_argument("--granular", action="store_true",
                    help="also write <name>-mola-hip-granular.yaml: upstream ICP loop, device Matcher / Solver classes")
    ap.add_argument("--pipelines", nargs="*", default=list(PIPELINES), help="files below the reference's pipelines directory")
    args = ap.parse_args(argv)
    ref = find_reference_dir(args.ref_dir)
    if not ref:
        print("make_mola_hip: no reference pipelines found (--ref-dir / MOLA_LO_PIPELINES_DIR)", file=sys.stderr)
        return 2
    for dst, r in generate(ref, args.out_dir, args.device_map, args.granular, args.pipelines).items():
        print("%s  <- %s  (changed lines: %s)" % (dst, r["source"], r["changed_lines"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
