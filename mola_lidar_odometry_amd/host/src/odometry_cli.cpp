// molahip-lo-cli: stand-alone LiDAR odometry over a KITTI odometry sequence folder, no Python in the loop.
//
// The role mola-lidar-odometry-cli plays in the reference's evaluation scripts (eval/cli_kitti.sh:23-50, relative to
// /root/reference): one sequence in, one TUM trajectory out.  Everything numeric happens in mola_hip::LidarOdometry
// (device-resident filters, ICP, local map); this file only reads velodyne/*.bin (float32 x,y,z,intensity rows),
// times.txt, and announces the next scan so that its upload and first filter pass overlap with the current ICP loop.
//
//   molahip-lo-cli --pipeline pipelines/lidar3d-default-hip.yaml --seq-dir /data/kitti/sequences/00 --out 00.tum
//                  [--device 0] [--no-prefetch] [--max-scans N]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <dirent.h>

#include "mola_lidar_odometry_hip/LidarOdometry.h"

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

}  // namespace

int main(int argc, char** argv) {
  std::string pipeline, seq_dir, out = "trajectory.tum";
  int device = 0;
  long max_scans = -1;
  bool prefetch = true;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto val = [&](const char* name) -> std::string {
      if (i + 1 >= argc) throw std::runtime_error(std::string("missing value for ") + name);
      return argv[++i];
    };
    try {
      if (a == "--pipeline") pipeline = val("--pipeline");
      else if (a == "--seq-dir") seq_dir = val("--seq-dir");
      else if (a == "--out") out = val("--out");
      else if (a == "--device") device = atoi(val("--device").c_str());
      else if (a == "--max-scans") max_scans = atol(val("--max-scans").c_str());
      else if (a == "--no-prefetch") prefetch = false;
      else throw std::runtime_error("unknown argument " + a);
    } catch (const std::exception& e) {
      fprintf(stderr, "%s\nusage: molahip-lo-cli --pipeline FILE.yaml --seq-dir DIR --out FILE.tum [--device N] "
                      "[--no-prefetch] [--max-scans N]\n", e.what());
      return 2;
    }
  }
  if (pipeline.empty() || seq_dir.empty()) {
    fprintf(stderr, "usage: molahip-lo-cli --pipeline FILE.yaml --seq-dir DIR --out FILE.tum [--device N] [--no-prefetch] "
                    "[--max-scans N]\n");
    return 2;
  }
  try {
    std::vector<std::string> files = list_bins(seq_dir + "/velodyne");
    if (max_scans >= 0 && (size_t)max_scans < files.size()) files.resize((size_t)max_scans);
    std::vector<double> stamps;
    {
      std::ifstream ts(seq_dir + "/times.txt");
      double t;
      while (ts >> t) stamps.push_back(t);
    }
    while (stamps.size() < files.size()) stamps.push_back(0.1 * (double)stamps.size());  // 10 Hz when times.txt is absent

    mola_hip::LidarOdometry lo(std::make_shared<mp2p_icp_hip::DeviceContext>(device));
    lo.initialize(mp2p_icp_hip::Config::FromYamlFile(pipeline));

    std::vector<float> cur, nxt;  // both stay alive while the driver may still read them
    if (!files.empty()) cur = read_bin(files[0]);
    size_t good = 0, keyframes = 0, iterations = 0;
    double seconds = 0;
    for (size_t k = 0; k < files.size(); k++) {
      const bool has_next = k + 1 < files.size();
      if (has_next) nxt = read_bin(files[k + 1]);  // (file reading is not part of the registration time)
      const auto t0 = std::chrono::steady_clock::now();
      if (has_next && prefetch) lo.prefetchInterleaved(nxt.data(), nxt.size() / 4, 16, 0, 4, 8);
      const auto& rec = lo.onLidarInterleaved(stamps[k], cur.data(), cur.size() / 4, 16, 0, 4, 8);
      seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      good += rec.icp_good ? 1 : 0;
      keyframes += rec.map_updated ? 1 : 0;
      iterations += rec.icp_iterations;
      if (has_next) {
        // the announced buffer must keep its address until it has been registered: swap contents, not storage roles
        cur.swap(nxt);
      }
    }
    lo.saveTrajectoryTUM(out);
    printf("{\"sequence_dir\": \"%s\", \"scans\": %zu, \"good\": %zu, \"keyframes\": %zu, \"icp_iterations\": %zu, "
           "\"seconds\": %.6f, \"scans_per_s\": %.3f, \"tum\": \"%s\"}\n",
           seq_dir.c_str(), files.size(), good, keyframes, iterations, seconds, seconds > 0 ? files.size() / seconds : 0.0,
           out.c_str());
  } catch (const std::exception& e) {
    fprintf(stderr, "molahip-lo-cli: %s\n", e.what());
    return 1;
  }
  return 0;
}
