// molahip-lo-cli: stand-alone LiDAR odometry over a KITTI odometry sequence folder, no Python in the loop.
//
// The role mola-lidar-odometry-cli plays in the reference's evaluation scripts (eval/cli_kitti.sh:23-50, relative to
// /root/reference): one sequence in, one TUM trajectory out.  Everything numeric happens in mola_hip::LidarOdometry
// (device-resident filters, ICP, local map); this file only reads velodyne/*.bin (float32 x,y,z,intensity rows),
// times.txt, and announces the next scan so that its upload and first filter pass overlap with the current ICP loop.
//
//   molahip-lo-cli --pipeline pipelines/lidar3d-default-hip.yaml --seq-dir /data/kitti/sequences/00 --out 00.tum
//                  [--device 0 | --devices 0,1,..|all] [--no-prefetch] [--max-scans N] [--time-field BYTES] [--scan-log FILE]
// --devices: config 4 of BASELINE.json without Python -- the sequences are assigned to the listed GPUs by LPT (longest
// sequence first onto the least loaded device, what eval/cli_kitti.sh:9,23-36 leaves to GNU parallel's job slots), every
// device runs its share like a --seq-dir list on one GPU (a host thread per sequence, one AlignBatcher per device); the
// "gather" is the host-side join of the per-sequence TUM files and ONE summary line.  A device may be listed twice
// (two independent batchers on it: the multi-device code path on a single-GPU box).  --plan-only prints the assignment.
// --seq-dir also takes a MulRan sequence folder (eval/cli_mulran.sh:23-36, `--input-mulran-seq KAIST01` with
// MULRAN_BASE_DIR, apps/mola-lidar-odometry-cli.cpp:186-208): <dir>/sensor_data/Ouster/<stamp in ns>.bin (or <dir>/Ouster/),
// float32 x,y,z,intensity rows like KITTI's, the scan's time stamp is its file name.
// Several --seq-dir run together on the one GPU (what eval/cli_kitti.sh:23 does with GNU parallel -j3, as processes):
// a host thread per sequence, their alignments merged into lock-step batches (mp2p_icp_hip::AlignBatcher).
#include <algorithm>
#include <chrono>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <memory>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include <dirent.h>

#include <sys/stat.h>

#include <atomic>
#include <condition_variable>
#include <mutex>

#include "mola_lidar_odometry_hip/LidarOdometry.h"
#include "molahip.h"

namespace {

std::vector<std::string> list_bins(const std::string& dir) {
  std::vector<std::string> out;
  DIR* d = opendir(dir.c_str());
  if (!d) throw std::runtime_error("cannot open " + dir);
  while (dirent* e = readdir(d)) {
    const std::string n = e->d_name;
    if (n.size() > 4 && n.compare(n.size() - 4, 4, ".bin") == 0) out.push_back(dir + "/" + n);
  }
  closedir(d);
  std::sort(out.begin(), out.end());
  return out;
}

bool dir_exists(const std::string& d) {
  DIR* h = opendir(d.c_str());
  if (h) closedir(h);
  return h != nullptr;
}
// ".../1561000444390857630.bin" -> 1561000444.390857630 [s]
double stamp_of_name(const std::string& path) {
  const size_t slash = path.rfind('/');
  const std::string base = path.substr(slash == std::string::npos ? 0 : slash + 1);
  return 1e-9 * strtod(base.c_str(), nullptr);
}

std::vector<float> read_bin(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot read " + path);
  fseek(f, 0, SEEK_END);
  const long bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (bytes < 0 || bytes % 16 != 0) {
    fclose(f);
    throw std::runtime_error(path + ": not a sequence of float32 x,y,z,intensity rows");
  }
  std::vector<float> v((size_t)bytes / 4);
  const size_t got = v.empty() ? 0 : fread(v.data(), 4, v.size(), f);
  fclose(f);
  if (got != v.size()) throw std::runtime_error("short read on " + path);
  return v;
}

constexpr size_t kWarmScans = 5;

struct SequenceReport {
  std::string seq_dir, out;
  size_t scans = 0, good = 0, keyframes = 0, iterations = 0;
  std::shared_ptr<void> keep_alive;        // the sequence's driver object (device buffers and all), destroyed after the reports
  double seconds = 0, steady_seconds = 0;  // steady: without the first kWarmScans scans (context, code objects, first map)
  size_t steady_scans = 0;
  std::string error;
  std::map<std::string, double> profile;  // LidarOdometry::profile(): host seconds per stage, whole run
  int device = 0;
  double mean_icp_points = 0, mean_map_layer_points = 0, mean_raw_points = 0;
  uint64_t final_map_points = 0, max_map_points = 0, final_map_voxels = 0;
  std::vector<double> scan_seconds;  // --scan-log: registration time of every scan
};

struct RunOptions {
  long max_scans = -1;
  bool prefetch = true;
  long long time_field = -1;  // byte offset of a float32 per-point time stamp inside the 16-byte record, or -1
  std::string scan_log;       // CSV of per-scan registration times and layer / map sizes
};

// what the replay leaves behind besides the trajectory: layer and map sizes (records()), optionally a per-scan CSV
void finish_report(const mola_hip::LidarOdometry& lo, const RunOptions& opt, SequenceReport& rep) {
  const auto& recs = lo.records();  // (resolves the map counters)
  double s_icp = 0, s_map = 0, s_raw = 0;
  size_t n_icp = 0;
  for (const auto& r : recs) {
    s_raw += (double)r.n_raw;
    if (r.icp_run) {
      s_icp += (double)r.n_for_icp;
      s_map += (double)r.n_for_map;
      n_icp++;
    }
    rep.max_map_points = std::max<uint64_t>(rep.max_map_points, r.n_map_points);
  }
  if (n_icp) rep.mean_icp_points = s_icp / (double)n_icp, rep.mean_map_layer_points = s_map / (double)n_icp;
  if (!recs.empty()) {
    rep.mean_raw_points = s_raw / (double)recs.size();
    rep.final_map_points = recs.back().n_map_points;
    rep.final_map_voxels = recs.back().n_map_voxels;
  }
  if (!opt.scan_log.empty()) {
    const std::string path = rep.out.size() > 4 && opt.scan_log == "auto" ? rep.out.substr(0, rep.out.size() - 4) + "_scans.csv" : opt.scan_log;
    FILE* f = fopen(path.c_str(), "w");
    if (f) {
      fprintf(f, "scan,seconds,n_raw,n_for_map,n_for_icp,n_map_points,icp_iterations,align_calls,goodness,sigma,map_updated,icp_good\n");
      for (size_t k = 0; k < recs.size(); k++)
        fprintf(f, "%zu,%.7f,%llu,%llu,%llu,%llu,%u,%u,%.4f,%.4f,%d,%d\n", k, k < rep.scan_seconds.size() ? rep.scan_seconds[k] : 0.0,
                (unsigned long long)recs[k].n_raw, (unsigned long long)recs[k].n_for_map, (unsigned long long)recs[k].n_for_icp,
                (unsigned long long)recs[k].n_map_points, recs[k].icp_iterations, recs[k].align_calls, recs[k].goodness, recs[k].sigma,
                (int)recs[k].map_updated, (int)recs[k].icp_good);
      fclose(f);
    }
  }
}

// files + stamps of a KITTI or MulRan sequence folder
void list_sequence(const std::string& seq_dir, long max_scans, std::vector<std::string>& files, std::vector<double>& stamps) {
  const std::string ouster = dir_exists(seq_dir + "/sensor_data/Ouster") ? seq_dir + "/sensor_data/Ouster"
                             : (dir_exists(seq_dir + "/Ouster") ? seq_dir + "/Ouster" : std::string());
  if (!dir_exists(seq_dir + "/velodyne") && !ouster.empty()) {
    // MulRan: one <time stamp in nanoseconds>.bin per sweep (numeric order = lexical order for equal-length names)
    files = list_bins(ouster);
    std::sort(files.begin(), files.end(), [](const std::string& a, const std::string& b) { return stamp_of_name(a) < stamp_of_name(b); });
    const double t0 = files.empty() ? 0.0 : stamp_of_name(files[0]);
    for (const auto& f : files) stamps.push_back(stamp_of_name(f) - t0);  // relative seconds (TUM output keeps sub-ms digits)
  } else {
    files = list_bins(seq_dir + "/velodyne");
    std::ifstream ts(seq_dir + "/times.txt");
    double t;
    while (ts >> t) stamps.push_back(t);
  }
  if (max_scans >= 0 && (size_t)max_scans < files.size()) files.resize((size_t)max_scans);
  while (stamps.size() < files.size()) stamps.push_back(0.1 * (double)stamps.size());  // 10 Hz when times.txt is absent
}

// one sequence, start to end; with a batcher its alignments join those of the other sequences of the process
void run_sequence(const std::string& pipeline, const std::string& seq_dir, const std::string& out, int device, const RunOptions& opt,
                  std::shared_ptr<mp2p_icp_hip::AlignBatcher> batcher, SequenceReport& rep) {
  rep.seq_dir = seq_dir;
  rep.out = out;
  rep.device = device;
  const long max_scans = opt.max_scans;
  const bool prefetch = opt.prefetch;
  mp2p_icp_hip::AlignBatcher::Membership member(batcher);  // leave() however this sequence ends
  try {
    // (a page-locked read-ahead ring with asynchronous uploads was tried here: the eight copies and the filter batch then land
    // on the alignment's first iterations -- 8 sequences 4000 scans/s against 4700, one sequence 0.89 ms per scan against
    // 0.86 -- so the threads keep uploading from pageable memory.  Round 3 also had a mode with every sequence a ucontext
    // fiber of ONE host thread, --fibers: 2410-2650 scans/s for eight sequences against 4580-4920 with threads; removed.)
    std::vector<std::string> files;
    std::vector<double> stamps;
    list_sequence(seq_dir, max_scans, files, stamps);

    // MOLAHIP_STARTUP_LOG=1: where a sequence's first second goes (stderr; seconds since the process's first sequence started)
    static const bool startup_log = getenv("MOLAHIP_STARTUP_LOG") != nullptr;
    static const auto t_proc = std::chrono::steady_clock::now();
    auto stamp = [&](const char* what, size_t k = 0) {
      if (startup_log)
        fprintf(stderr, "[startup %s] %.4f s  %s %zu\n", out.c_str(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t_proc).count(), what, k);
    };
    stamp("files listed");
    // (on the heap, handed to the report at the end: its tear-down -- a hipFree per buffer, each a device-wide wait -- is not part
    //  of the run; main() prints the reports first and then leaves without it)
    auto lo_owner = std::make_shared<mola_hip::LidarOdometry>(std::make_shared<mp2p_icp_hip::DeviceContext>(device));
    mola_hip::LidarOdometry& lo = *lo_owner;
    rep.keep_alive = lo_owner;
    stamp("device context");
    lo.initialize(mp2p_icp_hip::Config::FromYamlFile(pipeline));
    stamp("pipeline initialised");
    if (batcher) lo.setAlignBatcher(batcher);

    std::vector<float> cur, nxt;  // both stay alive while the driver may still read them
    if (!files.empty()) cur = read_bin(files[0]);
    for (size_t k = 0; k < files.size(); k++) {
      const bool has_next = k + 1 < files.size();
      if (has_next) nxt = read_bin(files[k + 1]);  // (file reading is not part of the registration time)
      const auto t0 = std::chrono::steady_clock::now();
      if (has_next && prefetch) lo.prefetchInterleaved(nxt.data(), nxt.size() / 4, 16, 0, 4, 8, opt.time_field);
      const auto& rec = lo.onLidarInterleaved(stamps[k], cur.data(), cur.size() / 4, 16, 0, 4, 8, opt.time_field);
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      rep.seconds += dt;
      rep.scan_seconds.push_back(dt);
      if (k >= kWarmScans) {
        rep.steady_seconds += dt;
        rep.steady_scans++;
      }
      if (k < 8 || k + 1 == files.size()) stamp("scan done", k);
      if (k + 1 == kWarmScans) lo.resetProfile();  // the stage table is the steady state's (context, code objects, first map left out)
      rep.good += rec.icp_good ? 1 : 0;
      rep.keyframes += rec.map_updated ? 1 : 0;
      rep.iterations += rec.icp_iterations;
      rep.scans++;
      if (has_next) {
        // the announced buffer must keep its address until it has been registered: swap contents, not storage roles
        cur.swap(nxt);
      }
    }
    lo.saveTrajectoryTUM(out);
    stamp("trajectory saved");
    rep.profile = lo.profile();
    finish_report(lo, opt, rep);
    stamp("report finished");
  } catch (const std::exception& e) {
    rep.error = e.what();
  }
}

// Longest-processing-time assignment of sequences (cost = number of scans) to device slots: the schedule DESIGN.md section 4
// prices config 4 with (11 KITTI sequences on 8 GPUs: makespan = sequence 02).  Returns slot index per sequence.
std::vector<size_t> lpt_assign(const std::vector<size_t>& scans, size_t n_slots, std::vector<size_t>* load_out = nullptr) {
  std::vector<size_t> order(scans.size()), slot(scans.size(), 0), load(n_slots, 0);
  for (size_t k = 0; k < order.size(); k++) order[k] = k;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return scans[a] > scans[b]; });
  for (size_t k : order) {
    const size_t best = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());  // first of the least loaded
    slot[k] = best;
    load[best] += scans[k];
  }
  if (load_out) *load_out = load;
  return slot;
}

std::vector<int> parse_devices(const std::string& arg) {
  std::vector<int> out;
  if (arg == "all") {
    int n = 0;
    if (mh_device_count(&n) != MH_OK || n <= 0) throw std::runtime_error("--devices all: no HIP device");
    for (int d = 0; d < n; d++) out.push_back(d);
    return out;
  }
  size_t pos = 0;
  while (pos <= arg.size()) {
    const size_t c = arg.find(',', pos);
    const std::string tok = arg.substr(pos, c == std::string::npos ? std::string::npos : c - pos);
    if (tok.empty() || tok.find_first_not_of("0123456789") != std::string::npos) throw std::runtime_error("--devices: expected a list like 0,1,2 or 'all'");
    out.push_back(atoi(tok.c_str()));
    if (c == std::string::npos) break;
    pos = c + 1;
  }
  return out;
}

}  // namespace

int main(int argc, char** argv) {
  std::string pipeline, out = "trajectory.tum";
  std::vector<std::string> seq_dirs;
  std::vector<int> devices;
  RunOptions opt;
  bool print_profile = false, plan_only = false;
  const char* usage = "usage: molahip-lo-cli --pipeline FILE.yaml --seq-dir DIR [--seq-dir DIR ...] --out FILE.tum [--device N | --devices 0,1,..|all] "
                      "[--no-prefetch] [--max-scans N] [--profile] [--time-field BYTES] [--scan-log FILE|auto] [--plan-only]\n"
                      "  several --seq-dir: the sequences run together, one host thread each, the alignments of the sequences that share\n"
                      "  a GPU merged into lock-step batches; trajectories go to FILE_<k>.tum\n"
                      "  --devices: the sequences are spread over the listed GPUs (longest first onto the least loaded device)\n";
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto val = [&](const char* name) -> std::string {
      if (i + 1 >= argc) throw std::runtime_error(std::string("missing value for ") + name);
      return argv[++i];
    };
    try {
      if (a == "--pipeline") pipeline = val("--pipeline");
      else if (a == "--seq-dir") seq_dirs.push_back(val("--seq-dir"));
      else if (a == "--out") out = val("--out");
      else if (a == "--device") devices = {atoi(val("--device").c_str())};
      else if (a == "--devices") devices = parse_devices(val("--devices"));
      else if (a == "--max-scans") opt.max_scans = atol(val("--max-scans").c_str());
      else if (a == "--time-field") opt.time_field = atoll(val("--time-field").c_str());
      else if (a == "--scan-log") opt.scan_log = val("--scan-log");
      else if (a == "--no-prefetch") opt.prefetch = false;
      else if (a == "--profile") print_profile = true;
      else if (a == "--plan-only") plan_only = true;
      else throw std::runtime_error("unknown argument " + a);
    } catch (const std::exception& e) {
      fprintf(stderr, "%s\n%s", e.what(), usage);
      return 2;
    }
  }
  if (pipeline.empty() || seq_dirs.empty()) {
    fprintf(stderr, "%s", usage);
    return 2;
  }
  if (devices.empty()) devices = {0};
  const size_t N = seq_dirs.size(), D = devices.size();
  // which sequence runs where: scan counts are known from the folders (no GPU needed for the plan)
  std::vector<size_t> n_scans(N, 0), slot(N, 0), load(D, 0);
  try {
    for (size_t k = 0; k < N; k++) {
      std::vector<std::string> files;
      std::vector<double> stamps;
      list_sequence(seq_dirs[k], opt.max_scans, files, stamps);
      n_scans[k] = files.size();
    }
  } catch (const std::exception& e) {
    fprintf(stderr, "molahip-lo-cli: %s\n", e.what());
    return 1;
  }
  slot = lpt_assign(n_scans, D, &load);
  const std::string stem = out.size() > 4 && out.compare(out.size() - 4, 4, ".tum") == 0 ? out.substr(0, out.size() - 4) : out;
  if (plan_only) {
    size_t total = 0, makespan = 0;
    for (size_t k = 0; k < N; k++) total += n_scans[k];
    for (size_t d = 0; d < D; d++) makespan = std::max(makespan, load[d]);
    printf("{\"plan\": [");
    for (size_t k = 0; k < N; k++)
      printf("%s{\"sequence_dir\": \"%s\", \"scans\": %zu, \"slot\": %zu, \"device\": %d}", k ? ", " : "", seq_dirs[k].c_str(), n_scans[k], slot[k], devices[slot[k]]);
    printf("], \"device_load_scans\": [");
    for (size_t d = 0; d < D; d++) printf("%s%zu", d ? ", " : "", load[d]);
    printf("], \"total_scans\": %zu, \"makespan_scans\": %zu, \"speedup_bound\": %.4f}\n", total, makespan, makespan ? (double)total / (double)makespan : 0.0);
    return 0;
  }
  std::vector<SequenceReport> reps(N);
  std::vector<std::shared_ptr<mp2p_icp_hip::AlignBatcher>> batchers(D);
  std::vector<size_t> per_slot(D, 0);
  for (size_t k = 0; k < N; k++) per_slot[slot[k]]++;
  const auto t0 = std::chrono::steady_clock::now();
  {  // the HIP runtime is initialised ONCE, here, inside the measured wall clock -- eight threads doing it at the same moment took
     // 0.20 s to their first context against 0.12 s (MOLAHIP_STARTUP_LOG=1)
    int32_t n_dev = 0;
    (void)mh_device_count(&n_dev);
  }
  if (N == 1) {
    run_sequence(pipeline, seq_dirs[0], out, devices[0], opt, nullptr, reps[0]);
  } else {
    // a batcher per device slot: the sequences of a slot advance together, the slots independently of each other
    for (size_t d = 0; d < D; d++)
      if (per_slot[d] > 1) batchers[d] = std::make_shared<mp2p_icp_hip::AlignBatcher>(per_slot[d]);
    std::vector<std::thread> th;
    for (size_t k = 0; k < N; k++)
      th.emplace_back(run_sequence, pipeline, seq_dirs[k], stem + "_" + std::to_string(k) + ".tum", devices[slot[k]], std::cref(opt),
                      batchers[slot[k]], std::ref(reps[k]));
    for (auto& t : th) t.join();
  }
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  int rc = 0;
  size_t total = 0;
  for (const auto& r : reps) {
    if (!r.error.empty()) {
      fprintf(stderr, "molahip-lo-cli: %s: %s\n", r.seq_dir.c_str(), r.error.c_str());
      rc = 1;
    }
    total += r.scans;
    printf("{\"sequence_dir\": \"%s\", \"device\": %d, \"scans\": %zu, \"good\": %zu, \"keyframes\": %zu, \"icp_iterations\": %zu, "
           "\"seconds\": %.6f, \"scans_per_s\": %.3f, \"steady_scans_per_s\": %.3f, \"mean_raw_points\": %.1f, \"mean_icp_points\": %.1f, "
           "\"mean_map_layer_points\": %.1f, \"final_map_points\": %llu, \"max_map_points\": %llu, \"final_map_voxels\": %llu, \"tum\": \"%s\"}\n",
           r.seq_dir.c_str(), r.device, r.scans, r.good, r.keyframes, r.iterations, r.seconds, r.seconds > 0 ? r.scans / r.seconds : 0.0,
           r.steady_seconds > 0 ? r.steady_scans / r.steady_seconds : 0.0, r.mean_raw_points, r.mean_icp_points, r.mean_map_layer_points,
           (unsigned long long)r.final_map_points, (unsigned long long)r.max_map_points, (unsigned long long)r.final_map_voxels, r.out.c_str());
    if (print_profile && r.scans) {  // host milliseconds per scan and stage (LidarOdometry::profile()), steady state
      const double ns = (double)(r.steady_scans ? r.steady_scans : r.scans);
      printf("{\"profile_ms_per_scan\": {");
      bool first = true;
      for (const auto& kv : r.profile) {
        // "icp.*" and "prefetch_*" entries are COUNTS per scan (align calls, host polls, iterations), the rest milliseconds
        const bool count = kv.first.compare(0, 4, "icp.") == 0 || kv.first.compare(0, 9, "prefetch_") == 0;
        printf("%s\"%s\": %.4f", first ? "" : ", ", kv.first.c_str(), (count ? 1.0 : 1e3) * kv.second / ns);
        first = false;
      }
      printf("}}\n");
    }
  }
  {
    // the sequences of a device advance together (one batch per round), so a device's slowest thread is its registration
    // time; the job's is the slowest device's (the makespan): steady rate = all steady scans / that
    size_t steady = 0, n_batches = 0, n_jobs = 0, f_batches = 0, f_jobs = 0, f_timeouts = 0;
    double slowest = 0, assembling = 0, running = 0;
    std::vector<double> dev_seconds(D, 0), dev_steady_seconds(D, 0);
    std::vector<size_t> dev_scans(D, 0), dev_steady(D, 0);
    for (size_t k = 0; k < N; k++) {
      const auto& r = reps[k];
      steady += r.steady_scans;
      slowest = std::max(slowest, r.steady_seconds);
      dev_scans[slot[k]] += r.scans;
      dev_steady[slot[k]] += r.steady_scans;
      dev_seconds[slot[k]] = std::max(dev_seconds[slot[k]], r.seconds);
      dev_steady_seconds[slot[k]] = std::max(dev_steady_seconds[slot[k]], r.steady_seconds);
    }
    for (const auto& b : batchers)
      if (b) {
        n_batches += b->batches(), n_jobs += b->jobs(), f_batches += b->filterBatches(), f_jobs += b->filterJobs(), f_timeouts += b->filterTimeouts();
        assembling += b->secondsAssembling(), running += b->secondsRunning();
      }
    const double nb = n_batches ? (double)n_batches : 1.0;
    uint64_t loops_started = 0, loops_abandoned = 0;
    mh_debug_loop_stats(&loops_started, &loops_abandoned);
    printf("{\"sequences\": %zu, \"devices\": %zu, \"scans\": %zu, \"wall_seconds\": %.6f, \"scans_per_s\": %.3f, \"steady_scans_per_s\": %.3f, "
           "\"batches\": %zu, \"jobs_per_batch\": %.2f, \"ms_per_batch_assembling\": %.4f, \"ms_per_batch_running\": %.4f, "
           "\"filter_batches\": %zu, \"filter_jobs\": %zu, \"filter_timeouts\": %zu, \"one_launch_loops\": %llu, "
           "\"one_launch_loops_abandoned\": %llu, \"per_device\": [",
           N, D, total, wall, wall > 0 ? total / wall : 0.0, slowest > 0 ? steady / slowest : 0.0, n_batches, n_jobs / nb,
           1e3 * assembling / nb, 1e3 * running / nb, f_batches, f_jobs, f_timeouts, (unsigned long long)loops_started,
           (unsigned long long)loops_abandoned);
    for (size_t d = 0; d < D; d++)
      printf("%s{\"slot\": %zu, \"device\": %d, \"sequences\": %zu, \"scans\": %zu, \"registration_seconds\": %.6f, \"scans_per_s\": %.3f, "
             "\"steady_scans_per_s\": %.3f}",
             d ? ", " : "", d, devices[d], per_slot[d], dev_scans[d], dev_seconds[d], dev_seconds[d] > 0 ? dev_scans[d] / dev_seconds[d] : 0.0,
             dev_steady_seconds[d] > 0 ? dev_steady[d] / dev_steady_seconds[d] : 0.0);
    printf("]}\n");
  }
  // Everything asked for is on disk and on stdout.  The drivers' destructors would now hipFree a few hundred buffers one by one,
  // each call a wait for the whole device: 0.2 s for eight sequences -- the operating system takes the memory back faster
  // (MOLAHIP_FULL_TEARDOWN=1 runs them, e.g. under a leak checker).
  fflush(stdout);
  fflush(stderr);
  // (exit(), not _exit(): handlers registered with atexit -- a profiler's finalisation, the runtime's own -- still run; the
  //  drivers, held by `reps` on this frame, are not destroyed by it)
  if (getenv("MOLAHIP_FULL_TEARDOWN") == nullptr) exit(rc);
  return rc;
}
