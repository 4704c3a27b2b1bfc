// The roscpp node's ROS-free core (ros/auv_particle_filter_hip/include/auv_particle_filter_hip/pf_core.hpp) driven
// without ROS: a map file, odometry, a LaserScan ping, the same ping as points in base_frame, a GPS fix, a tick.
//   g++ -std=c++14 -Iinclude -Iros/auv_particle_filter_hip/include examples/pf_core_example.cpp \
//       -Lsmarc_navigation_amd -lmcl_hip -Wl,-rpath,$PWD/smarc_navigation_amd -o /tmp/pf_core_example
//   /tmp/pf_core_example map.mclgrid ranges.txt [landmarks.yaml]   -> one line of numbers (tests/test_cpp_core.py compares it with
//   the Python mirror fed the same inputs)
#include <cstdio>
#include <cstdlib>

#include "auv_particle_filter_hip/pf_core.hpp"

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s <map file> <ranges file: one range per line, angles linspace(-1, 1)>\n", argv[0]);
    return 2;
  }
  auv_pf_hip::Params p;
  p.particle_count = 4096;
  p.seed = 11;
  p.init_covariance = "[0.5, 0.5, 0.0, 0.0, 0.0, 0.01]";
  p.motion_covariance = "[0.001, 0.001, 0.0, 0.0, 0.0, 0.00001]";
  p.resampling_noise_covariance = "[0.01, 0.01, 0.0, 0.0, 0.0, 0.0001]";
  p.measurement_std = 1.0;
  p.mbes_sensor_offset = "[0.3, 0.0, -0.1, 0.0, 0.05, 0.0]";
  p.map_grid_file = argv[1];
  if (argc > 3) {   // BASELINE config 5: a landmark map; two detections arrive with the first ping
    p.landmark_map_file = argv[3];
    p.landmark_k = 2;
  }
  std::vector<float> ranges;
  {
    std::ifstream f(argv[2]);
    float r;
    while (f >> r) ranges.push_back(r);
  }
  const int B = (int)ranges.size();
  if (B < 2) return 2;
  const double tr[3] = {0.5, -0.5, 0.0}, q0[4] = {0, 0, 0, 1};
  double m2o[16];
  mcl_matrix_from_tf(tr, q0, m2o);
  auv_pf_hip::Core core;
  if (!core.init(p, m2o)) {
    std::fprintf(stderr, "init: %s\n", core.error().c_str());
    return 1;
  }
  core.start_timing(100.0);
  const double v[3] = {1.0, 0.0, 0.0};
  const double amin = -1.0, ainc = (double)(float)(2.0 / (B - 1));   // (float32, as a sensor_msgs/LaserScan carries it)
  // the ping as points in base_frame: sensor-frame hit (0, r sin a, -r cos a) through the sensor offset
  double off[6];
  auv_pf_hip::parse_cov_string(p.mbes_sensor_offset, off);
  const double cp = std::cos(off[4]), sp = std::sin(off[4]);   // (roll = yaw = 0 in this example)
  std::vector<float> pts;
  for (int k = B - 1; k >= 0; --k) {   // reversed: the core sorts the beams
    const double a = amin + ainc * k, r = ranges[k];
    const double s[3] = {0.0, r * std::sin(a), -r * std::cos(a)};
    pts.push_back((float)(cp * s[0] + sp * s[2] + off[0]));
    pts.push_back((float)(s[1] + off[1]));
    pts.push_back((float)(-sp * s[0] + cp * s[2] + off[2]));
  }
  for (int k = 0; k < 3; ++k) {
    if (!core.odom(100.02 + 0.02 * k, v, 0.05, q0, -2.0)) return 1;
    if (k == 1 && argc > 3) {
      const double det[6] = {3.0, 4.0, -16.0, -2.0, -6.0, -15.5};   // base_frame, stamped like the ping
      if (!core.detections(100.02 + 0.02 * k, det, 2)) return 1;
    }
    if (k == 1 && !core.ping_scan(ranges.data(), B, amin, ainc, 80.0, 100.02 + 0.02 * k)) return 1;
    if (k == 2 && !core.ping_points(pts.data(), B, false)) return 1;
  }
  core.dive(false);
  if (!core.gps(0.6, -0.4)) return 1;
  double mean[6], yaw, cov36[36], quat[4];
  if (!core.loc_pose(mean, &yaw, cov36, quat)) {
    std::fprintf(stderr, "loc_pose: %s\n", core.error().c_str());
    return 1;
  }
  std::vector<double> poses;
  if (!core.poses(poses)) return 1;
  std::printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %zu\n", mean[0], mean[1], mean[2], yaw, cov36[0], cov36[1],
              cov36[4], quat[2], poses.size());
  return 0;
}
