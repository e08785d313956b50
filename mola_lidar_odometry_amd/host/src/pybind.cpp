// pybind.cpp -- Python view of the C++ host layer, used by tests/test_host_layer.py so that the parity tests
// drive ICP::align() exactly the way mola::LidarOdometry does (LidarOdometry.cpp:961-962).
#include <pybind11/functional.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "mola_lidar_odometry_hip/LidarOdometry.h"
#include "mp2p_icp_hip/mp2p_icp_hip.h"
#include "molahip_host/plugin_switches.h"

namespace py = pybind11;
using namespace mp2p_icp_hip;

static std::shared_ptr<PointCloud> make_cloud(py::array_t<float, py::array::c_style | py::array::forcecast> xyz) {
  auto c = std::make_shared<PointCloud>();
  auto r = xyz.unchecked<2>();
  for (py::ssize_t i = 0; i < r.shape(0); i++) c->insertPoint(r(i, 0), r(i, 1), r(i, 2));
  return c;
}

PYBIND11_MODULE(_mp2p_icp_hip, m) {
  m.doc() = "mp2p_icp_hip: C++ host layer of libmolahip (mirrors the mp2p_icp plugin API)";
  py::class_<TPose3D>(m, "TPose3D")
      .def(py::init<>())
      .def(py::init([](double x, double y, double z, double yaw, double pitch, double roll) { return TPose3D{x, y, z, yaw, pitch, roll}; }))
      .def_readwrite("x", &TPose3D::x).def_readwrite("y", &TPose3D::y).def_readwrite("z", &TPose3D::z)
      .def_readwrite("yaw", &TPose3D::yaw).def_readwrite("pitch", &TPose3D::pitch).def_readwrite("roll", &TPose3D::roll);
  py::class_<CPose3D>(m, "CPose3D")
      .def(py::init<>())
      .def(py::init<const TPose3D&>())
      .def_static("from_matrix", [](std::vector<double> T) { CPose3D p; if (T.size() != 12) throw std::runtime_error("need 12 values"); std::copy(T.begin(), T.end(), p.T); return p; })
      .def("asTPose", &CPose3D::asTPose)
      .def("matrix", [](const CPose3D& p) { return std::vector<double>(p.T, p.T + 12); })
      .def("__add__", &CPose3D::operator+)
      .def("__sub__", &CPose3D::operator-);
  py::class_<CPose3DPDFGaussianInf>(m, "CPose3DPDFGaussianInf")
      .def(py::init([](const CPose3D& mean, std::vector<double> info) {
        CPose3DPDFGaussianInf p; p.mean = mean; if (info.size() != 36) throw std::runtime_error("need 36 values");
        std::copy(info.begin(), info.end(), p.cov_inv); return p; }));
  py::class_<Config>(m, "Config")
      .def_static("FromYamlText", &Config::FromYamlText)
      .def_static("FromYamlFile", &Config::FromYamlFile)
      .def("has", &Config::has)
      .def("__getitem__", [](const Config& c, const std::string& k) { return c[k]; })
      .def("at", [](const Config& c, size_t i) { return c.at(i); })
      .def("size", &Config::size)
      .def("asString", &Config::asString);
  m.def("evaluate_expression", &evaluate_expression);
  // MOLA_HIP_* switches (molahip_host/plugin_switches.h, shared with the mp2p_icp adapter): re-read the environment, and
  // show what was read -- what the adapter would pass to the C ABI for an upstream `RobustKernel::<name>`
  m.def("reload_plugin_switches", [] {
    molahip_host::reload_plugin_switches();   // this module's copy of the cache (what plugin_switches() below shows) ...
    mp2p_icp_hip::reload_plugin_switches();   // ... and the host library's, which is the one the alignments read
  });
  m.def("library_matched_points", [] { return mp2p_icp_hip::plugin_switch_matched_points(); });
  m.def("plugin_switches", [] {
    const auto& s = molahip_host::plugin_switches();
    py::dict d;
    d["gm_form"] = s.gm_form; d["index_mode"] = s.index_mode; d["cov_step_xyz"] = s.cov_step_xyz; d["cov_step_ang"] = s.cov_step_ang;
    d["min_delta"] = s.min_delta; d["max_cost"] = s.max_cost; d["pt2pl_mode"] = s.pt2pl_mode; d["matched_points"] = s.matched_points;
    d["far_voxel_metric"] = s.far_voxel_metric; d["force_cpu"] = s.force_cpu;
    return d;
  });
  m.def("kernel_from_upstream_name", [](const std::string& n) { return molahip_host::kernel_from_upstream_name(n.c_str(), molahip_host::plugin_switches()); });
  m.def("term_reason_name", [](uint32_t t) { return std::string(enum2str(molahip_host::term_reason_to<IterTermReason>(t))); });
  m.def("evaluate_compiled", [](const std::string& e, const std::map<std::string, double>& v) { return CompiledExpression(e).evaluate(v); });
  py::class_<ParameterSource>(m, "ParameterSource")
      .def(py::init<>())
      .def("updateVariable", &ParameterSource::updateVariable)
      .def("realize", &ParameterSource::realize);
  py::class_<Layer, std::shared_ptr<Layer>>(m, "Layer");
  py::class_<PointCloud, Layer, std::shared_ptr<PointCloud>>(m, "PointCloud")
      .def(py::init(&make_cloud))
      .def("size", &PointCloud::size);
  py::class_<HashedVoxelPointCloud, Layer, std::shared_ptr<HashedVoxelPointCloud>>(m, "HashedVoxelPointCloud")
      .def(py::init([](float vs, uint32_t cap) { return std::make_shared<HashedVoxelPointCloud>(vs, cap); }))
      .def("setPoints", [](HashedVoxelPointCloud& h, py::array_t<float, py::array::c_style | py::array::forcecast> xyz) {
        auto c = make_cloud(xyz); h.setPoints(c->x.data(), c->y.data(), c->z.data(), c->size()); })
      .def("insertPoints", [](HashedVoxelPointCloud& h, py::array_t<float, py::array::c_style | py::array::forcecast> xyz) {
        auto c = make_cloud(xyz); h.insertPoints(c->x.data(), c->y.data(), c->z.data(), c->size()); })
      .def("size", &HashedVoxelPointCloud::size)
      .def("voxelCount", &HashedVoxelPointCloud::voxelCount);
  py::class_<NDT, HashedVoxelPointCloud, std::shared_ptr<NDT>>(m, "NDT")
      .def(py::init([](float vs, uint32_t cap, float min_dist, float ratio) { return std::make_shared<NDT>(vs, cap, min_dist, ratio); }))
      .def("planeCount", &NDT::planeCount);
  py::class_<metric_map_t>(m, "metric_map_t")
      .def(py::init<>())
      .def("set_layer", [](metric_map_t& mm, const std::string& n, std::shared_ptr<Layer> l) { mm.layers[n] = std::move(l); });
  py::class_<Parameters>(m, "Parameters")
      .def(py::init<>())
      .def_readwrite("maxIterations", &Parameters::maxIterations)
      .def_readwrite("minAbsStep_trans", &Parameters::minAbsStep_trans)
      .def_readwrite("minAbsStep_rot", &Parameters::minAbsStep_rot)
      .def_readwrite("generateDebugFiles", &Parameters::generateDebugFiles)
      .def_readwrite("debugFileNameFormat", &Parameters::debugFileNameFormat);
  py::enum_<IterTermReason>(m, "IterTermReason")
      .value("Undefined", IterTermReason::Undefined).value("NoPairings", IterTermReason::NoPairings)
      .value("SolverError", IterTermReason::SolverError).value("MaxIterations", IterTermReason::MaxIterations)
      .value("Stalled", IterTermReason::Stalled).value("QualityCheckpointFailed", IterTermReason::QualityCheckpointFailed)
      .value("HookRequest", IterTermReason::HookRequest);
  py::class_<Results>(m, "Results")
      .def(py::init<>())
      .def_readonly("quality", &Results::quality)
      .def_readonly("nIterations", &Results::nIterations)
      .def_readonly("terminationReason", &Results::terminationReason)
      .def("pose", [](const Results& r) { return std::vector<double>(r.optimal_tf.mean.T, r.optimal_tf.mean.T + 12); })
      .def("cov", [](const Results& r) { return std::vector<double>(r.optimal_tf.cov, r.optimal_tf.cov + 36); })
      .def("n_pairs", [](const Results& r) { return r.finalPairings.size(); })
      .def("n_pairs_pt2pl", [](const Results& r) { return r.finalPairings.pl_lx.size(); })
      .def("pair_global_idx", [](const Results& r) { return r.finalPairings.globalIdx; })
      .def("pair_local_idx", [](const Results& r) { return r.finalPairings.localIdx; });
  py::class_<ICP, std::shared_ptr<ICP>>(m, "ICP")
      .def("align", [](ICP& icp, const metric_map_t& l, const metric_map_t& g, const TPose3D& guess, const Parameters& p,
                       std::optional<CPose3DPDFGaussianInf> prior) { Results r; icp.align(l, g, guess, p, r, prior); return r; },
           py::arg("pcLocal"), py::arg("pcGlobal"), py::arg("initialGuess"), py::arg("params"), py::arg("prior") = std::nullopt)
      .def("attachToParameterSource", &ICP::attachToParameterSource)
      .def("setDeviceHook", &ICP::setDeviceHook)
      .def("clearHooks", &ICP::clearHooks)
      .def("forceGenericPath", &ICP::forceGenericPath)
      .def("setHookReplay", &ICP::setHookReplay)
      .def("lastAlignUsedFusedPath", &ICP::lastAlignUsedFusedPath)
      .def("setIterationHook", [](ICP& icp, std::function<bool(uint32_t, std::vector<double>)> f) {
        icp.setIterationHook([f](const ICP::IterationHook_Input& in) {
          ICP::IterationHook_Output o;
          o.request_stop = f(in.currentIteration, std::vector<double>(in.currentSolution->optimalPose.T, in.currentSolution->optimalPose.T + 12));
          return o; }); });
  // ---- stand-alone odometry driver
  using mola_hip::LidarOdometry;
  auto rec2dict = [](const LidarOdometry::ScanRecord& r) {
    py::dict d;
    d["timestamp"] = r.timestamp; d["dropped"] = r.dropped; d["first_scan"] = r.first_scan; d["icp_run"] = r.icp_run;
    d["icp_good"] = r.icp_good; d["had_motion_model"] = r.had_motion_model; d["map_updated"] = r.map_updated;
    d["restarted"] = r.restarted;
    d["pose"] = std::vector<double>(r.pose.T, r.pose.T + 12);
    d["init_guess"] = std::vector<double>(r.init_guess.T, r.init_guess.T + 12);
    d["goodness"] = r.goodness; d["sigma"] = r.sigma; d["estimated_sensor_max_range"] = r.estimated_sensor_max_range;
    d["instantaneous_sensor_max_range"] = r.instantaneous_sensor_max_range;
    d["icp_iterations"] = r.icp_iterations; d["twist_corrections"] = r.twist_corrections; d["align_calls"] = r.align_calls;
    d["termination"] = r.termination; d["n_raw"] = r.n_raw; d["n_for_map"] = r.n_for_map; d["n_for_icp"] = r.n_for_icp;
    d["n_map_points"] = r.n_map_points; d["n_map_voxels"] = r.n_map_voxels;
    d["twist"] = std::vector<double>{r.twist.vx, r.twist.vy, r.twist.vz, r.twist.wx, r.twist.wy, r.twist.wz};
    d["decim_map_resolution"] = r.decim_map_resolution; d["decim_icp_resolution"] = r.decim_icp_resolution;
    d["map_voxel_size"] = r.map_voxel_size;
    return d;
  };
  py::class_<LidarOdometry>(m, "LidarOdometry", py::dynamic_attr())
      .def(py::init([](int device, bool own_context) {
             // default: the process-wide context on device 0; device >= 0 / own_context: a context (stream + scratch) of
             // its own on that device -- one per GPU rank, and required when several drivers run in threads of one process
             if (device < 0 && !own_context) return std::make_unique<LidarOdometry>();
             return std::make_unique<LidarOdometry>(std::make_shared<DeviceContext>(device < 0 ? 0 : device)); }),
           py::arg("device") = -1, py::arg("own_context") = false)
      .def("initialize", &LidarOdometry::initialize)
      .def("reset", &LidarOdometry::reset)
      .def("onLidar", [rec2dict](LidarOdometry& lo, double stamp, py::array_t<float, py::array::c_style | py::array::forcecast> xyz,
                                 std::optional<py::array_t<float, py::array::c_style | py::array::forcecast>> t,
                                 std::array<int, 3> xyz_fields, int t_field) {
        // [n,3] points or [n,k] float32 records (a KITTI .bin is [n,4]; a PointCloud2 payload of float fields likewise):
        // the rows go to the device as they are and are split into channels there.  xyz_fields / t_field = column
        // indices of the coordinates / of a per-point time stamp (-1: none, or the separate array `t`)
        if (xyz.ndim() != 2 || xyz.shape(1) < 3) throw std::runtime_error("xyz must be [n,3] (or [n,k>=3] records)");
        const size_t n = (size_t)xyz.shape(0), k = (size_t)xyz.shape(1);
        for (int f : xyz_fields)
          if (f < 0 || (size_t)f >= k) throw std::runtime_error("xyz_fields out of range");
        if (t_field >= (int)k) throw std::runtime_error("t_field out of range");
        const float* tp = nullptr;
        if (t) {
          if ((size_t)t->size() != n) throw std::runtime_error("t must have n entries");
          tp = t->data();
        }
        {
          py::gil_scoped_release nogil;  // the numpy buffers are only read through their pointers: other drivers' threads may run
          (void)lo.onLidarInterleaved(stamp, xyz.data(), n, k * sizeof(float), 4u * (size_t)xyz_fields[0],
                                       4u * (size_t)xyz_fields[1], 4u * (size_t)xyz_fields[2],
                                       t_field >= 0 ? 4ll * t_field : -1ll, tp);
        }
        return rec2dict(lo.records().back()); },  // (records(): the map counters of this record, read back now)
           py::arg("timestamp"), py::arg("xyz"), py::arg("t") = std::nullopt,
           py::arg("xyz_fields") = std::array<int, 3>{0, 1, 2}, py::arg("t_field") = -1)
      .def("setAlignBatcher", &LidarOdometry::setAlignBatcher)
      .def("prefetch", [](py::object self, py::array xyz_any, std::optional<py::array> t_any,
                          std::array<int, 3> xyz_fields, int t_field) {
        // announce the NEXT scan (same arguments as the onLidar call that will follow): upload + first filter pass run
        // on a second stream while the current scan is registered.  The worker thread reads the caller's buffers, so
        // no hidden temporary may stand in for them: float32, C-contiguous arrays only (anything else is rejected
        // instead of converted), kept alive on the object -- the announced scan AND the one whose worker may still be
        // running (it is joined by the onLidar call that picks it up, which comes before the next prefetch but one).
        auto strict = [](const py::array& a, const char* what) {
          if (!py::dtype::of<float>().is(a.dtype()) || !(a.flags() & py::array::c_style))
            throw std::runtime_error(std::string(what) + " must be a C-contiguous float32 array (prefetch reads it from a worker thread)");
        };
        strict(xyz_any, "xyz");
        if (t_any) strict(*t_any, "t");
        auto xyz = py::array_t<float, py::array::c_style>::ensure(xyz_any);
        std::optional<py::array_t<float, py::array::c_style>> t;
        if (t_any) t = py::array_t<float, py::array::c_style>::ensure(*t_any);
        if (!xyz || (t_any && !*t)) throw std::runtime_error("prefetch: unusable array");
        if (xyz.data() != xyz_any.data() || (t && t->data() != t_any->data())) throw std::runtime_error("prefetch: array was copied");
        LidarOdometry& lo = self.cast<LidarOdometry&>();
        if (xyz.ndim() != 2 || xyz.shape(1) < 3) throw std::runtime_error("xyz must be [n,3] (or [n,k>=3] records)");
        const size_t n = (size_t)xyz.shape(0), k = (size_t)xyz.shape(1);
        for (int f : xyz_fields)
          if (f < 0 || (size_t)f >= k) throw std::runtime_error("xyz_fields out of range");
        if (t_field >= (int)k) throw std::runtime_error("t_field out of range");
        const float* tp = nullptr;
        if (t) {
          if ((size_t)t->size() != n) throw std::runtime_error("t must have n entries");
          tp = t->data();
        }
        lo.prefetchInterleaved(xyz.data(), n, k * sizeof(float), 4u * (size_t)xyz_fields[0], 4u * (size_t)xyz_fields[1],
                               4u * (size_t)xyz_fields[2], t_field >= 0 ? 4ll * t_field : -1ll, tp);
        if (py::hasattr(self, "_prefetch_keepalive")) self.attr("_prefetch_inflight") = self.attr("_prefetch_keepalive");
        self.attr("_prefetch_keepalive") = py::make_tuple(xyz, t ? py::object(*t) : py::none()); },
           py::arg("xyz"), py::arg("t") = std::nullopt, py::arg("xyz_fields") = std::array<int, 3>{0, 1, 2},
           py::arg("t_field") = -1)
      .def("records", [rec2dict](const LidarOdometry& lo) { py::list l; for (auto& r : lo.records()) l.append(rec2dict(r)); return l; })
      .def("trajectory", [](const LidarOdometry& lo) {
        py::list l;
        for (auto& [t, p] : lo.estimatedTrajectory()) l.append(py::make_tuple(t, std::vector<double>(p.T, p.T + 12)));
        return l; })
      .def("saveTrajectoryTUM", &LidarOdometry::saveTrajectoryTUM)
      .def("dynamicVariables", &LidarOdometry::dynamicVariables)
      .def("describePipeline", &LidarOdometry::describePipeline)
      .def("profile", [](const LidarOdometry& lo) { return lo.profile(); })
      .def("localMapSize", [](const LidarOdometry& lo) { return lo.localMap() ? lo.localMap()->size() : 0; });
  py::class_<AlignBatcher, std::shared_ptr<AlignBatcher>>(m, "AlignBatcher")
      .def(py::init<size_t>(), py::arg("participants"))
      .def("leave", [](AlignBatcher& b) { b.leave(); }, py::call_guard<py::gil_scoped_release>())
      .def("batches", &AlignBatcher::batches)
      .def("jobs", &AlignBatcher::jobs);
  m.def("icp_pipeline_from_yaml", [](const Config& c) { auto t = icp_pipeline_from_yaml(c); return py::make_tuple(std::get<0>(t), std::get<1>(t)); });
}
