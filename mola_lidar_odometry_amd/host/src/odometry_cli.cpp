// molahip-lo-cli: stand-alone LiDAR odometry over a KITTI odometry sequence folder, no Python in the loop.
//
// The role mola-lidar-odometry-cli plays in the reference's evaluation scripts (eval/cli_kitti.sh:23-50, relative to
// /root/reference): one sequence in, one TUM trajectory out.  Everything numeric happens in mola_hip::LidarOdometry
// (device-resident filters, ICP, local map); this file only reads velodyne/*.bin (float32 x,y,z,intensity rows),
// times.txt, and announces the next scan so that its upload and first filter pass overlap with the current ICP loop.
//
//   molahip-lo-cli --pipeline pipelines/lidar3d-default-hip.yaml --seq-dir /data/kitti/sequences/00 --out 00.tum
//                  [--device 0] [--no-prefetch] [--max-scans N]
// --seq-dir also takes a MulRan sequence folder (eval/cli_mulran.sh:23-36, `--input-mulran-seq KAIST01` with
// MULRAN_BASE_DIR, apps/mola-lidar-odometry-cli.cpp:186-208): <dir>/sensor_data/Ouster/<stamp in ns>.bin (or <dir>/Ouster/),
// float32 x,y,z,intensity rows like KITTI's, the scan's time stamp is its file name.
// Several --seq-dir run together on the one GPU (what eval/cli_kitti.sh:23 does with GNU parallel -j3, as processes):
// a host thread per sequence, their alignments merged into lock-step batches (mp2p_icp_hip::AlignBatcher).
#include <algorithm>
#include <chrono>
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
#include "molahip_host/fibers.h"

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
  double seconds = 0, steady_seconds = 0;  // steady: without the first kWarmScans scans (context, code objects, first map)
  size_t steady_scans = 0;
  std::string error;
  std::map<std::string, double> profile;  // LidarOdometry::profile(): host seconds per stage, whole run
};

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

// --fibers: every sequence is a fiber of the ONE thread that talks to the HIP runtime (molahip_host/fibers.h).  Its scans
// are read ahead by a plain reader thread (no HIP calls) into a ring of page-locked buffers, so that the uploads are
// asynchronous copies; a buffer is handed back once the scan AFTER its own has been registered.
struct SequenceFeed {
  static constexpr size_t kDepth = 4;
  std::vector<std::string> files;
  void* buf[kDepth] = {nullptr, nullptr, nullptr, nullptr};
  size_t n_floats[kDepth] = {0, 0, 0, 0};
  size_t cap_bytes = 0;
  std::atomic<size_t> read_upto{0}, consumed{0};
  std::atomic<bool> failed{false}, stop{false};
  std::string error;
  std::thread th;
  void start() {
    for (const auto& f : files) {
      struct stat st;
      if (stat(f.c_str(), &st) == 0 && (size_t)st.st_size > cap_bytes) cap_bytes = (size_t)st.st_size;
    }
    for (size_t i = 0; i < kDepth; i++)
      if (mh_host_alloc_pinned(cap_bytes ? cap_bytes : 16, &buf[i]) != MH_OK) throw std::runtime_error(std::string("pinned buffer: ") + mh_last_error_string());
    th = std::thread([this] {
      for (size_t k = 0; k < files.size(); k++) {
        while (!stop && k >= consumed.load(std::memory_order_acquire) + kDepth) std::this_thread::sleep_for(std::chrono::microseconds(100));  // (a full ring: nothing to do for a scan's time)
        if (stop) return;
        FILE* f = fopen(files[k].c_str(), "rb");
        size_t got = 0;
        if (f) {
          got = fread(buf[k % kDepth], 1, cap_bytes, f);
          fclose(f);
        }
        if (!f || got % 16 != 0) {
          error = files[k] + ": not a sequence of float32 x,y,z,intensity rows";
          failed = true;
          read_upto.store(files.size(), std::memory_order_release);
          return;
        }
        n_floats[k % kDepth] = got / 4;
        read_upto.store(k + 1, std::memory_order_release);
      }
    });
  }
  ~SequenceFeed() {
    stop = true;
    if (th.joinable()) th.join();
    for (void* b : buf) (void)mh_host_free_pinned(b);
  }
};

void run_sequence_fiber(const std::string& pipeline, const std::string& seq_dir, const std::string& out, int device, long max_scans,
                        std::shared_ptr<mp2p_icp_hip::AlignBatcher> batcher, SequenceReport& rep) {
  rep.seq_dir = seq_dir;
  rep.out = out;
  mp2p_icp_hip::AlignBatcher::Membership member(batcher);  // leave() however this sequence ends
  try {
    SequenceFeed feed;
    std::vector<double> stamps;
    list_sequence(seq_dir, max_scans, feed.files, stamps);
    feed.start();
    mola_hip::LidarOdometry lo(std::make_shared<mp2p_icp_hip::DeviceContext>(device));
    lo.initialize(mp2p_icp_hip::Config::FromYamlFile(pipeline));
    lo.setInputPinned(true);
    if (batcher) lo.setAlignBatcher(batcher);
    const size_t n = feed.files.size();
    for (size_t k = 0; k < n; k++) {
      const size_t need = std::min(k + 2, n);  // this scan and the next one (announced to the prefetch)
      while (feed.read_upto.load(std::memory_order_acquire) < need) molahip_host::FiberScheduler::yield();
      if (feed.failed) throw std::runtime_error(feed.error);
      const bool has_next = k + 1 < n;
      const auto t0 = std::chrono::steady_clock::now();
      if (has_next) lo.prefetchInterleaved(feed.buf[(k + 1) % SequenceFeed::kDepth], feed.n_floats[(k + 1) % SequenceFeed::kDepth] / 4, 16, 0, 4, 8);
      const auto& rec = lo.onLidarInterleaved(stamps[k], feed.buf[k % SequenceFeed::kDepth], feed.n_floats[k % SequenceFeed::kDepth] / 4, 16, 0, 4, 8);
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      rep.seconds += dt;
      if (k >= kWarmScans) {
        rep.steady_seconds += dt;
        rep.steady_scans++;
      }
      if (k + 1 == kWarmScans) lo.resetProfile();
      rep.good += rec.icp_good ? 1 : 0;
      rep.keyframes += rec.map_updated ? 1 : 0;
      rep.iterations += rec.icp_iterations;
      rep.scans++;
      feed.consumed.store(k, std::memory_order_release);  // buffers of scans < k may be refilled (k + 1 is in flight)
    }
    lo.saveTrajectoryTUM(out);
    rep.profile = lo.profile();
  } catch (const std::exception& e) {
    rep.error = e.what();
  }
}

// one sequence, start to end; with a batcher its alignments join those of the other sequences of the process
void run_sequence(const std::string& pipeline, const std::string& seq_dir, const std::string& out, int device, long max_scans,
                  bool prefetch, std::shared_ptr<mp2p_icp_hip::AlignBatcher> batcher, SequenceReport& rep) {
  rep.seq_dir = seq_dir;
  rep.out = out;
  mp2p_icp_hip::AlignBatcher::Membership member(batcher);  // leave() however this sequence ends
  try {
    // (the page-locked read-ahead ring of the fiber mode was tried here as well: with asynchronous uploads the eight
    // copies and the filter batch land on the alignment's first iterations -- 8 sequences 4000 scans/s against 4700, one
    // sequence 0.89 ms per scan against 0.86 -- so the threads keep uploading from pageable memory)
    std::vector<std::string> files;
    std::vector<double> stamps;
    list_sequence(seq_dir, max_scans, files, stamps);

    mola_hip::LidarOdometry lo(std::make_shared<mp2p_icp_hip::DeviceContext>(device));
    lo.initialize(mp2p_icp_hip::Config::FromYamlFile(pipeline));
    if (batcher) lo.setAlignBatcher(batcher);

    std::vector<float> cur, nxt;  // both stay alive while the driver may still read them
    if (!files.empty()) cur = read_bin(files[0]);
    for (size_t k = 0; k < files.size(); k++) {
      const bool has_next = k + 1 < files.size();
      if (has_next) nxt = read_bin(files[k + 1]);  // (file reading is not part of the registration time)
      const auto t0 = std::chrono::steady_clock::now();
      if (has_next && prefetch) lo.prefetchInterleaved(nxt.data(), nxt.size() / 4, 16, 0, 4, 8);
      const auto& rec = lo.onLidarInterleaved(stamps[k], cur.data(), cur.size() / 4, 16, 0, 4, 8);
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      rep.seconds += dt;
      if (k >= kWarmScans) {
        rep.steady_seconds += dt;
        rep.steady_scans++;
      }
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
    rep.profile = lo.profile();
  } catch (const std::exception& e) {
    rep.error = e.what();
  }
}

}  // namespace

int main(int argc, char** argv) {
  std::string pipeline, out = "trajectory.tum";
  std::vector<std::string> seq_dirs;
  int device = 0;
  long max_scans = -1;
  bool prefetch = true, print_profile = false, fibers = false;
  const char* usage = "usage: molahip-lo-cli --pipeline FILE.yaml --seq-dir DIR [--seq-dir DIR ...] --out FILE.tum [--device N] "
                      "[--no-prefetch] [--max-scans N] [--profile] [--fibers]\n"
                      "  several --seq-dir: the sequences run together on the one GPU, one host thread each, their alignments\n"
                      "  merged into lock-step batches; trajectories go to FILE_<k>.tum\n";
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
      else if (a == "--device") device = atoi(val("--device").c_str());
      else if (a == "--max-scans") max_scans = atol(val("--max-scans").c_str());
      else if (a == "--no-prefetch") prefetch = false;
      else if (a == "--profile") print_profile = true;
      else if (a == "--fibers") fibers = true;
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
  const size_t N = seq_dirs.size();
  std::vector<SequenceReport> reps(N);
  std::shared_ptr<mp2p_icp_hip::AlignBatcher> batcher_keep;
  const auto t0 = std::chrono::steady_clock::now();
  if (fibers) {
    // ONE thread in the HIP runtime: the sequences (and their prefetch workers) are fibers of this thread
    molahip_host::FiberScheduler sched;
    auto batcher = N > 1 ? std::make_shared<mp2p_icp_hip::AlignBatcher>(N) : nullptr;
    batcher_keep = batcher;
    const std::string stem = out.size() > 4 && out.compare(out.size() - 4, 4, ".tum") == 0 ? out.substr(0, out.size() - 4) : out;
    for (size_t k = 0; k < N; k++) {
      const std::string o = N == 1 ? out : stem + "_" + std::to_string(k) + ".tum";
      sched.spawn([&, k, o] { run_sequence_fiber(pipeline, seq_dirs[k], o, device, max_scans, batcher, reps[k]); });
    }
    sched.run();
  } else if (N == 1) {
    run_sequence(pipeline, seq_dirs[0], out, device, max_scans, prefetch, nullptr, reps[0]);
  } else {
    auto batcher = std::make_shared<mp2p_icp_hip::AlignBatcher>(N);
    batcher_keep = batcher;
    std::vector<std::thread> th;
    const std::string stem = out.size() > 4 && out.compare(out.size() - 4, 4, ".tum") == 0 ? out.substr(0, out.size() - 4) : out;
    for (size_t k = 0; k < N; k++)
      th.emplace_back(run_sequence, pipeline, seq_dirs[k], stem + "_" + std::to_string(k) + ".tum", device, max_scans, prefetch,
                      batcher, std::ref(reps[k]));
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
    printf("{\"sequence_dir\": \"%s\", \"scans\": %zu, \"good\": %zu, \"keyframes\": %zu, \"icp_iterations\": %zu, "
           "\"seconds\": %.6f, \"scans_per_s\": %.3f, \"steady_scans_per_s\": %.3f, \"tum\": \"%s\"}\n",
           r.seq_dir.c_str(), r.scans, r.good, r.keyframes, r.iterations, r.seconds, r.seconds > 0 ? r.scans / r.seconds : 0.0,
           r.steady_seconds > 0 ? r.steady_scans / r.steady_seconds : 0.0, r.out.c_str());
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
  if (N > 1) {
    // the sequences advance together (one batch per round), so the slowest thread's registration time is the job's
    size_t steady = 0;
    double slowest = 0;
    for (const auto& r : reps) {
      steady += r.steady_scans;
      slowest = r.steady_seconds > slowest ? r.steady_seconds : slowest;
    }
    // the batcher's view of a round (all scans, warm-up included): waiting for the last sequence to arrive / the batch call
    const double nb = batcher_keep && batcher_keep->batches() ? (double)batcher_keep->batches() : 1.0;
    printf("{\"sequences\": %zu, \"scans\": %zu, \"wall_seconds\": %.6f, \"scans_per_s\": %.3f, \"steady_scans_per_s\": %.3f, "
           "\"batches\": %zu, \"jobs_per_batch\": %.2f, \"ms_per_batch_assembling\": %.4f, \"ms_per_batch_running\": %.4f, "
           "\"filter_batches\": %zu, \"filter_jobs\": %zu, \"filter_timeouts\": %zu}\n",
           N, total, wall, wall > 0 ? total / wall : 0.0, slowest > 0 ? steady / slowest : 0.0,
           batcher_keep ? batcher_keep->batches() : (size_t)0, batcher_keep ? batcher_keep->jobs() / nb : 0.0,
           batcher_keep ? 1e3 * batcher_keep->secondsAssembling() / nb : 0.0, batcher_keep ? 1e3 * batcher_keep->secondsRunning() / nb : 0.0,
           batcher_keep ? batcher_keep->filterBatches() : (size_t)0, batcher_keep ? batcher_keep->filterJobs() : (size_t)0,
           batcher_keep ? batcher_keep->filterTimeouts() : (size_t)0);
  }
  return rc;
}
