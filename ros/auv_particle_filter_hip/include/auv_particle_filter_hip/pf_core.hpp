// pf_core.hpp -- the `auv_pf` node's logic over the C ABI (include/mcl.h), ROS-free and header-only: what the roscpp
// node (src/auv_pf_node.cpp) wraps with subscribers / publishers / tf, and what examples/pf_core_example.cpp drives
// without ROS (tests/test_cpp_core.py compiles it on CPU and compares it with the Python mirror on the GPU).
//
// Mirrors auv_particle_filter/scripts/auv_pf.py method for method: parameters and their defaults (:27-56), the
// covariance-string parser (:40-44), `diving` starts true (:103), odom_callback / predict (:201-216), gps_odom_cb /
// update / resample (:125-192) with the gate specified away as in SURVEY A.10 ("every fix after the first predict
// while not diving"), update_loc_pose (:218-260: covariance in the first nine of 36 slots, tf translation z = 0).
// Beyond the reference: the bathymetric map and the MBES ping (LaserScan geometry of toy_mbes_manipulator.cpp:69-73;
// a point cloud in base_frame as mbes_receptor.cpp:126-165 leaves it); the landmark map and the receptors' detections
// (BASELINE config 5: map_provider_node.py:35-56, toy_mbes_receptor.cpp:68-110).  One owner at a time: the node serialises its
// callbacks with a mutex (the reference's three rospy threads are unlocked).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>

#include "mcl.h"

namespace auv_pf_hip {

struct Params {
  int particle_count = 10;                                                 // auv_pf.py:27
  std::string map_frame = "map", base_frame = "base_link", utm_frame = "utm", odom_frame = "sam/odom";  // :28-31
  double measurement_std = 0.01;                                           // :39
  std::string motion_covariance = "[0.0000, 0.0000, 0.0, 0.0, 0.0, 0.000000000001]";   // launch defaults (no code default)
  std::string init_covariance = "[0.1, 0.1, 0.0, 0.0, 0.0, 0.0]";
  std::string resampling_noise_covariance = "[1., 1., 0.0, 0.0, 0.0, 0.0001]";
  std::string particle_poses_topic = "/particle_poses", odom_corrected_topic = "/average_pose";   // :64,71
  std::string aux_dive = "/dive", gps_odom_topic = "/gps", odom_topic = "odom";                    // :101,106,110
  // beyond the reference
  std::string resample_scheme = "systematic";
  int seed = 0, device = 0;
  std::string mbes_topic = "/mbes_scan", mbes_pointcloud_topic = "", mbes_points_frame = "base";
  double mbes_std = 0.2, mbes_range_max = 100.0;
  std::string mbes_sensor_offset = "[0.0, 0.0, 0.0, 0.0, 0.0, 0.0]";
  std::string map_grid_file = "", map_mesh_file = "";
  // BASELINE config 5: landmark map (the .yaml of the reference's map provider, map_provider_node.py:35-56, or x y z
  // rows; map frame; models with z < rocks_depth kept) and the detections of the MBES receptors
  // (geometry_msgs/PoseArray in base_frame, toy_mbes_receptor.cpp:68-110; ekf_slam.cpp:41 names the topic parameter)
  std::string landmark_map_file = "", lm_detect_topic = "/landmarks_detected";
  double rocks_depth = 1e300, landmark_std = 0.3, landmark_gate = 11.345, landmark_sync_tol = 1e-3, landmark_max_age = 0.5;
  bool landmark_late_drop = false;   // ~landmark_late = 'drop': detections that arrive AFTER their ping are ignored (default 'update': an update + resampling of their own -- two resamplings, two doses of resampling noise, per ping)
  int landmark_k = 1;
};

// the reference's ad-hoc parser (auv_pf.py:40-44): strip the brackets, split on ", "
inline bool parse_cov_string(const std::string& in, double out[6]) {
  std::string s;
  for (char c : in)
    if (c != '[' && c != ']') s.push_back(c);
  size_t pos = 0;
  for (int k = 0; k < 6; ++k) {
    const size_t next = s.find(", ", pos);
    const std::string tok = s.substr(pos, next == std::string::npos ? std::string::npos : next - pos);
    char* end = nullptr;
    out[k] = std::strtod(tok.c_str(), &end);
    if (end == tok.c_str()) return false;
    if (next == std::string::npos) return k == 5;
    pos = next + 2;
  }
  return true;
}

// tf.transformations.quaternion_from_euler, axes 'sxyz' (auv_pf.py:233)
inline void quaternion_from_euler(double roll, double pitch, double yaw, double q[4]) {
  const double cr = std::cos(roll / 2), sr = std::sin(roll / 2), cp = std::cos(pitch / 2), sp = std::sin(pitch / 2);
  const double cy = std::cos(yaw / 2), sy = std::sin(yaw / 2);
  q[0] = cp * (sr * cy) - sp * (cr * sy);
  q[1] = cp * (sr * sy) + sp * (cr * cy);
  q[2] = cp * (cr * sy) - sp * (sr * cy);
  q[3] = cp * (cr * cy) + sp * (sr * sy);
}

// Map files: an ASCII .ply triangle mesh, or a height grid as text header + raw floats (".mclgrid":
// "mclgrid nx ny origin_x origin_y res\n" followed by nx*ny little-endian float32, z[ix*ny + iy]); both are also read
// by the Python node (smarc_navigation_amd/auv_pf.py:load_map_file).
struct MapFile {
  bool is_grid = false;
  int nx = 0, ny = 0;
  double ox = 0, oy = 0, res = 1;
  std::vector<float> z, verts;
  std::vector<uint32_t> tris;
};
inline bool load_map_file(const std::string& path, MapFile& m, std::string& err) {
  std::ifstream f(path, std::ios::binary);
  if (!f) {
    err = "cannot open " + path;
    return false;
  }
  std::string magic;
  f >> magic;
  if (magic == "mclgrid") {
    f >> m.nx >> m.ny >> m.ox >> m.oy >> m.res;
    f.get();  // the newline that ends the header
    if (!f || m.nx < 2 || m.ny < 2 || !(m.res > 0)) {
      err = path + ": bad mclgrid header";
      return false;
    }
    m.z.resize((size_t)m.nx * m.ny);
    f.read((char*)m.z.data(), (std::streamsize)(sizeof(float) * m.z.size()));
    if (f.gcount() != (std::streamsize)(sizeof(float) * m.z.size())) {
      err = path + ": truncated height array";
      return false;
    }
    m.is_grid = true;
    return true;
  }
  if (magic == "ply") {
    std::string line;
    long nv = 0, nf = 0;
    bool ascii = false;
    std::getline(f, line);
    while (std::getline(f, line)) {
      std::istringstream t(line);
      std::string a, b;
      t >> a >> b;
      if (a == "format") ascii = b == "ascii";
      if (a == "element" && b == "vertex") t >> nv;
      if (a == "element" && b == "face") t >> nf;
      if (a == "end_header") break;
    }
    if (!ascii || nv < 3 || nf < 1) {
      err = path + ": only ASCII PLY triangle meshes are read";
      return false;
    }
    m.verts.resize((size_t)nv * 3);
    for (long i = 0; i < nv; ++i) {
      std::getline(f, line);
      std::istringstream t(line);
      t >> m.verts[3 * i] >> m.verts[3 * i + 1] >> m.verts[3 * i + 2];
    }
    for (long i = 0; i < nf; ++i) {
      std::getline(f, line);
      std::istringstream t(line);
      int cnt = 0;
      t >> cnt;
      std::vector<uint32_t> idx(cnt > 0 ? cnt : 0);
      for (auto& v : idx) t >> v;
      for (int k = 1; k + 1 < cnt; ++k) {  // fan-triangulate polygons
        m.tris.push_back(idx[0]);
        m.tris.push_back(idx[k]);
        m.tris.push_back(idx[k + 1]);
      }
    }
    if (!f && !f.eof()) {
      err = path + ": read error";
      return false;
    }
    return true;
  }
  err = path + ": expected an ASCII .ply mesh or a .mclgrid height grid";
  return false;
}

// Landmark map: n x (x, y, z) in the map frame.  A .yaml / .yml file is read the way the reference's map provider reads
// its Gazebo model list (map_provider_node.py:43-52) without a YAML library: every `position:` block contributes its
// x / y / z scalars, in whichever order they appear; anything else is whitespace-separated x y z rows.  Landmarks with
// z >= rocks_depth are dropped (the provider's filter).
inline bool load_landmark_file(const std::string& path, double rocks_depth, std::vector<double>& xyz, std::string& err) {
  std::ifstream f(path);
  if (!f) {
    err = "cannot open " + path;
    return false;
  }
  xyz.clear();
  const bool yaml = path.size() > 4 && (path.rfind(".yaml") == path.size() - 5 || path.rfind(".yml") == path.size() - 4);
  std::string line;
  if (yaml) {
    double p[3] = {0, 0, 0};
    int have = 0;
    bool in_pos = false;
    auto flush = [&]() {
      if (have == 7 && p[2] < rocks_depth) xyz.insert(xyz.end(), p, p + 3);
      have = 0;
    };
    while (std::getline(f, line)) {
      const size_t c = line.find(':');
      if (c == std::string::npos) continue;
      std::string key = line.substr(0, c), val = line.substr(c + 1);
      key.erase(0, key.find_first_not_of(" \t-{"));
      key.erase(key.find_last_not_of(" \t") + 1);
      if (key == "position") {
        flush();
        in_pos = true;
        // flow style on one line: position: {x: 1, y: 2, z: 3}
        for (const char* k : {"x", "y", "z"}) {
          const size_t q = val.find(std::string(k) + ":");
          if (q != std::string::npos) {
            p[k[0] - 'x'] = std::strtod(val.c_str() + q + 2, nullptr);
            have |= 1 << (k[0] - 'x');
          }
        }
        if (have == 7) {
          flush();
          in_pos = false;
        }
        continue;
      }
      if (in_pos && key.size() == 1 && key[0] >= 'x' && key[0] <= 'z') {
        p[key[0] - 'x'] = std::strtod(val.c_str(), nullptr);
        have |= 1 << (key[0] - 'x');
        if (have == 7) {
          flush();
          in_pos = false;
        }
      } else if (in_pos) {
        in_pos = false;
        have = 0;
      }
    }
    flush();
  } else {
    while (std::getline(f, line)) {
      std::istringstream t(line);
      double p[3];
      if ((t >> p[0] >> p[1] >> p[2]) && p[2] < rocks_depth) xyz.insert(xyz.end(), p, p + 3);
    }
  }
  if (xyz.empty()) {
    err = path + ": no landmarks (all filtered by rocks_depth?)";
    return false;
  }
  return true;
}

class Core {
 public:
  Core() = default;
  Core(const Core&) = delete;
  Core& operator=(const Core&) = delete;
  ~Core() {
    if (h_) mcl_destroy(h_);
  }

  // m2o: row-major 4x4 map <- odom (mcl_matrix_from_tf of the tf lookup, auv_pf.py:79-81)
  bool init(const Params& p, const double m2o[16]) {
    p_ = p;
    mcl_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.n_particles = p.particle_count;
    cfg.device = p.device;
    cfg.seed = (uint64_t)p.seed;
    cfg.rng_mode = MCL_RNG_NATIVE;
    cfg.resample_scheme = p.resample_scheme == "residual"      ? MCL_RESAMPLE_RESIDUAL
                          : p.resample_scheme == "stratified"  ? MCL_RESAMPLE_STRATIFIED
                          : p.resample_scheme == "multinomial" ? MCL_RESAMPLE_MULTINOMIAL
                                                               : MCL_RESAMPLE_SYSTEMATIC;
    cfg.meas_std = p.measurement_std;
    if (!parse_cov_string(p.init_covariance, cfg.init_cov) || !parse_cov_string(p.motion_covariance, cfg.process_cov) ||
        !parse_cov_string(p.resampling_noise_covariance, cfg.resample_cov) ||
        !parse_cov_string(p.mbes_sensor_offset, offset_)) {
      err_ = "a covariance / offset string does not hold six numbers separated by \", \"";
      return false;
    }
    std::memcpy(cfg.m2o, m2o, sizeof cfg.m2o);
    int rc = mcl_create(&cfg, &h_);
    if (rc != MCL_OK) {
      err_ = std::string("mcl_create: ") + mcl_last_error(nullptr);
      h_ = nullptr;
      return false;
    }
    if (!check(mcl_init_particles(h_, nullptr))) return false;
    for (const std::string* path : {&p.map_grid_file, &p.map_mesh_file})
      if (!path->empty() && !load_map(*path)) return false;
    if (!p.landmark_map_file.empty()) {
      std::vector<double> xyz;
      if (!load_landmark_file(p.landmark_map_file, p.rocks_depth, xyz, err_)) return false;
      if (!set_landmarks(xyz.data(), (int64_t)(xyz.size() / 3))) return false;
    }
    return true;
  }

  bool set_landmarks(const double* xyz, int64_t n) {
    const bool ok = check(mcl_set_landmarks(h_, xyz, n));
    has_landmarks_ = has_landmarks_ || ok;
    return ok;
  }
  bool has_landmarks() const { return has_landmarks_; }

  // Landmark detections of one ping (base_frame positions, toy_mbes_receptor.cpp:75-105: stamped with the ping's stamp,
  // published AFTER the receptor has processed the ping).  The usual order -- the ping with this stamp, or a later one,
  // has already been through ping_scan / ping_points --: a measurement update of their own followed by the resampling,
  // like a GPS fix (detections and ranges are independent measurements of the same pose).  Ahead of their ping (a bag
  // replayed by topic): held until the ping with the same stamp (within landmark_sync_tol: the stamps are copies of one
  // another), whose likelihood they join before the resampling (mcl_update_landmarks, accumulate = 1).  Without a
  // bathymetric map always an update of their own.  Detections older than landmark_max_age seconds of the filter's clock
  // (the latest odometry stamp) are dropped: their base_frame positions describe a pose the cloud has long left.
  bool detections(double stamp, const double* xyz, int n_det) {
    if (old_time_ == 0.0 || !has_landmarks_ || n_det < 1) return true;
    if (time_ - stamp > p_.landmark_max_age) return true;
    if (has_map_ && (!have_ping_ || stamp > last_ping_stamp_ + p_.landmark_sync_tol)) {
      pending_stamp_ = stamp;
      pending_det_.assign(xyz, xyz + (size_t)n_det * 3);
      return true;
    }
    if (p_.landmark_late_drop && has_map_) return true;   // (after its ping; one resampling per ping asked for)
    return check(mcl_update_landmarks(h_, xyz, n_det, p_.landmark_std, p_.landmark_k, p_.landmark_gate, nullptr, 0)) &&
           check(mcl_resample(h_, nullptr, 0, nullptr));
  }

  bool load_map(const std::string& path) {
    MapFile m;
    if (!load_map_file(path, m, err_)) return false;
    const bool ok = m.is_grid ? check(mcl_set_map_grid(h_, m.z.data(), m.nx, m.ny, m.ox, m.oy, m.res))
                              : check(mcl_set_map_mesh(h_, m.verts.data(), (int64_t)(m.verts.size() / 3), m.tris.data(),
                                                       (int64_t)(m.tris.size() / 3)));
    has_map_ = has_map_ || ok;
    return ok;
  }

  void start_timing(double stamp) { time_ = old_time_ = stamp; }   // auv_pf.py:96-98
  void dive(bool diving) { diving_ = diving; }                      // :100-103 (starts true)
  bool has_map() const { return has_map_; }
  const std::string& error() const { return err_; }
  const Params& params() const { return p_; }

  // odom_callback + predict (auv_pf.py:201-216): twist.linear (body frame), twist.angular.z, orientation, position.z
  bool odom(double stamp, const double v[3], double w_z, const double q[4], double z) {
    time_ = stamp;
    bool ok = true;
    if (old_time_ != 0.0 && time_ > old_time_) {
      mcl_odom od;
      od.stamp = stamp;
      for (int k = 0; k < 3; ++k) od.v[k] = v[k];
      od.w_z = w_z;
      for (int k = 0; k < 4; ++k) od.q[k] = q[k];
      od.z = z;
      ok = check(mcl_predict(h_, &od, time_ - old_time_, nullptr));
    }
    old_time_ = time_;
    return ok;
  }

  // gps_odom_cb -> update -> resample (auv_pf.py:125-192); (x, y) already transformed utm -> map (once, not per particle)
  bool gps(double x_map, double y_map) {
    if (old_time_ == 0.0 || diving_) return true;
    return check(mcl_update_gps(h_, x_map, y_map)) && check(mcl_resample(h_, nullptr, 0, nullptr));
  }

  // one ping as a LaserScan: angle_min + k * angle_increment, ranges[k]
  bool ping_scan(const float* ranges, int n, double angle_min, double angle_increment, double range_max, double stamp = 0.0) {
    if (old_time_ == 0.0 || !has_map_ || n < 1) return true;
    angles_.resize(n);
    for (int k = 0; k < n; ++k) angles_[k] = (float)(angle_min + angle_increment * k);
    return check(mcl_update_mbes(h_, ranges, angles_.data(), n, p_.mbes_std, range_max, offset_)) &&
           accumulate_pending(stamp) && check(mcl_resample(h_, nullptr, 0, nullptr));
  }

  // one ping as points (x, y, z triples): every point is a beam's hit.  In the sensor frame beam b looks along
  // (0, sin a, -cos a), so a_b = atan2(y, -z) and the range is |p|; points in base_frame are taken back through the
  // sensor offset first; the beams are handed over in ascending angle.  NaN points are dropped.
  bool ping_points(const float* xyz, int n_points, bool in_sensor_frame, double stamp = 0.0) {
    if (old_time_ == 0.0 || !has_map_) return true;
    double R[9];
    rot(offset_[3], offset_[4], offset_[5], R);
    std::vector<std::pair<float, float>> beams;  // (angle, range)
    beams.reserve(n_points);
    for (int i = 0; i < n_points; ++i) {
      double p[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
      if (!(p[0] == p[0] && p[1] == p[1] && p[2] == p[2])) continue;
      if (!in_sensor_frame) {
        const double d[3] = {p[0] - offset_[0], p[1] - offset_[1], p[2] - offset_[2]};
        for (int c = 0; c < 3; ++c) p[c] = R[0 + c] * d[0] + R[3 + c] * d[1] + R[6 + c] * d[2];  // R^T d
      }
      beams.emplace_back((float)std::atan2(p[1], -p[2]), (float)std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]));
    }
    if (beams.empty()) return true;
    std::stable_sort(beams.begin(), beams.end(), [](const std::pair<float, float>& a, const std::pair<float, float>& b) { return a.first < b.first; });
    angles_.resize(beams.size());
    ranges_.resize(beams.size());
    for (size_t k = 0; k < beams.size(); ++k) {
      angles_[k] = beams[k].first;
      ranges_[k] = beams[k].second;
    }
    return check(mcl_update_mbes(h_, ranges_.data(), angles_.data(), (int)beams.size(), p_.mbes_std, p_.mbes_range_max, offset_)) &&
           accumulate_pending(stamp) && check(mcl_resample(h_, nullptr, 0, nullptr));
  }

  // update_loc_pose (auv_pf.py:218-260): mean pose, arithmetic mean of the wrapped yaws, 3 x 3 position covariance
  // written row-major into the first nine of the 36 covariance slots, quaternion of (roll, pitch, yaw means)
  bool loc_pose(double mean6[6], double* yaw, double cov36[36], double quat[4]) {
    double cov9[9];
    if (!check(mcl_mean_cov(h_, mean6, yaw, cov9))) return false;
    std::fill(cov36, cov36 + 36, 0.0);
    for (int k = 0; k < 9; ++k) cov36[k] = cov9[k];
    quaternion_from_euler(mean6[3], mean6[4], *yaw, quat);
    return true;
  }

  // PoseArray payload (auv_pf.py:264-277): n x (x, y, z, qx, qy, qz, qw)
  bool poses(std::vector<double>& pose7) {
    pose7.resize((size_t)p_.particle_count * 7);
    return check(mcl_get_poses(h_, pose7.data()));
  }

  mcl_handle* handle() { return h_; }

 private:
  // right after an MBES update: detections that arrived ahead of THIS ping onto its likelihood; held detections of an
  // earlier ping (which never came) are dropped -- never applied to another ping --, of a later one kept
  bool accumulate_pending(double ping_stamp) {
    last_ping_stamp_ = ping_stamp;
    have_ping_ = true;
    if (pending_det_.empty()) return true;
    if (pending_stamp_ > ping_stamp + p_.landmark_sync_tol) return true;
    std::vector<double> det;
    det.swap(pending_det_);
    if (pending_stamp_ < ping_stamp - p_.landmark_sync_tol) return true;
    return check(mcl_update_landmarks(h_, det.data(), (int)(det.size() / 3), p_.landmark_std, p_.landmark_k, p_.landmark_gate,
                                      nullptr, 1));
  }
  static void rot(double roll, double pitch, double yaw, double R[9]) {
    const double cr = std::cos(roll), sr = std::sin(roll), cp = std::cos(pitch), sp = std::sin(pitch);
    const double cy = std::cos(yaw), sy = std::sin(yaw);
    const double M[9] = {cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr, sy * cp, sy * sp * sr + cy * cr,
                         sy * sp * cr - cy * sr, -sp, cp * sr, cp * cr};
    std::memcpy(R, M, sizeof M);
  }
  bool check(int rc) {
    if (rc == MCL_OK) return true;
    err_ = std::string(mcl_status_string(rc)) + ": " + mcl_last_error(h_);
    return false;
  }

  Params p_;
  mcl_handle* h_ = nullptr;
  double offset_[6] = {0, 0, 0, 0, 0, 0};
  double time_ = 0.0, old_time_ = 0.0;
  bool diving_ = true, has_map_ = false;   // auv_pf.py:103
  bool has_landmarks_ = false;
  double pending_stamp_ = 0.0, last_ping_stamp_ = 0.0;
  bool have_ping_ = false;
  std::vector<double> pending_det_;
  std::vector<float> angles_, ranges_;
  std::string err_;
};

}  // namespace auv_pf_hip
