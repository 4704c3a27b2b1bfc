// mcl_dr_impl.h -- dead-reckoning integrator (include/mcl_dr.h): the host-side producer of the
// particle filter's Odometry input.  Sequential, a few dozen flops per message: plain C++ on the
// host by design (nothing here is data-parallel).  Restates the behaviour of
// sam_dead_reckoning/scripts/dr_node.py (class VehicleDR) and sam_mm.py (class SAM); the reference's
// arithmetic order is kept so that a replay agrees with the node to rounding.
#pragma once
#include <cmath>
#include <cstring>
#include "../../include/mcl_dr.h"

struct mcl_dr {
  double dvl_period, dr_period;
  bool init_heading = false, init_m2o = false, init_stim = false, dvl_on = false, depth_meas = false;
  bool gps_registered = true;
  double init_quat[4] = {0, 0, 0, 1};
  double pos[3] = {0, 0, 0};      // pos_t
  double rot[3] = {0, 0, 0};      // rot_t (roll, pitch measured; yaw integrated, never wrapped)
  double vel_rot[3] = {0, 0, 0};
  double t_stim_prev = 0.0, t_dvl_prev = 0.0, t_now = 0.0;
  double dvl_vel[3] = {0, 0, 0};  // dvl_latest.velocity
  double b2p[3] = {0, 0, 0};
  double base_depth = 0.0;
  double u_rpm = 0.0, u_dr = 0.0; // control input of the thrust model (dr_node.py:85-86)
  double thrust_cmd = 0.0;        // thruster_horizontal_radians
};

namespace dr_detail {

constexpr double kPi = 3.14159265358979323846;

inline void quat_from_euler(double roll, double pitch, double yaw, double q[4]) {
  const double cr = std::cos(roll / 2.0), sr = std::sin(roll / 2.0);
  const double cp = std::cos(pitch / 2.0), sp = std::sin(pitch / 2.0);
  const double cy = std::cos(yaw / 2.0), sy = std::sin(yaw / 2.0);
  q[0] = cp * (sr * cy) - sp * (cr * sy);
  q[1] = cp * (sr * sy) + sp * (cr * cy);
  q[2] = cp * (cr * sy) - sp * (sr * cy);
  q[3] = cp * (cr * cy) + sp * (sr * sy);
}

// SAM.eom (sam_mm.py:30-120): nudot = M^-1 tau with the constant mass matrix
//   M = [[m, 0, -m y_g], [0, m, m x_g], [-m y_g, m x_g, Izz]],  y_g = 0
// tau = (F cos d, -F sin d, 0), F = KT rpm, d = -dr.  The inverse is written out (block form).
inline void sam_motion(double rpm, double dr, double nudot[3]) {
  const double m = 15.4, Izz = 1.6202, x_g = 0.4, KT = 0.3;
  const double d = dr * -1.0;
  const double F = KT * (rpm * 1);
  const double t0 = F * std::cos(d), t1 = -F * std::sin(d);
  const double mx = m * x_g;
  const double det = m * Izz - mx * mx;
  nudot[0] = t0 / m;
  nudot[1] = (Izz / det) * t1;
  nudot[2] = (-mx / det) * t1;
}

inline double clip(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

}  // namespace dr_detail

extern "C" {

int mcl_dr_create(const mcl_dr_config* cfg, mcl_dr** out) {
  if (!cfg || !out) return MCL_ERR_INVALID;
  if (!(cfg->dvl_period > 0.0) || !(cfg->dr_period > 0.0)) return MCL_ERR_INVALID;
  mcl_dr* h = new mcl_dr();
  h->dvl_period = cfg->dvl_period;
  h->dr_period = cfg->dr_period;
  h->u_dr = dr_detail::clip(0.0, -7 * dr_detail::kPi / 180, 7 * dr_detail::kPi / 180);
  *out = h;
  return MCL_OK;
}

void mcl_dr_destroy(mcl_dr* h) { delete h; }

int mcl_dr_heading(mcl_dr* h, const double q[4]) {
  if (!h || !q) return MCL_ERR_INVALID;
  for (int k = 0; k < 4; ++k) h->init_quat[k] = q[k];
  h->init_heading = true;
  return MCL_OK;
}

int mcl_dr_gps(mcl_dr* h, double gx_map, double gy_map, int have_pressure_tf, const double b2p_trans[3],
               int* initialised, double m2o_t[3], double m2o_q[4]) {
  if (!h) return MCL_ERR_INVALID;
  if (have_pressure_tf && !b2p_trans) return MCL_ERR_INVALID;
  if (initialised) *initialised = 0;
  if (!h->gps_registered) return MCL_OK;  // the node unregisters the subscriber once initialised (:144)
  if (h->init_heading) {
    // map -> odom: translation = the fix, rotation = yaw of the heading quaternion only (:131-142)
    double e[3], q[4];
    euler_from_quat(h->init_quat, e);
    dr_detail::quat_from_euler(0.0, 0.0, e[2], q);
    if (m2o_t) {
      m2o_t[0] = gx_map;
      m2o_t[1] = gy_map;
      m2o_t[2] = 0.0;
    }
    if (m2o_q)
      for (int k = 0; k < 4; ++k) m2o_q[k] = q[k];
    if (initialised) *initialised = 1;
    h->init_m2o = true;
    h->gps_registered = false;
  }
  // pressure sensor present -> AUV, else surface vehicle (:150-161)
  if (have_pressure_tf) {
    for (int k = 0; k < 3; ++k) h->b2p[k] = b2p_trans[k];
    h->depth_meas = true;
  }
  return MCL_OK;
}

int mcl_dr_imu(mcl_dr* h, double stamp, const double q[4], const double ang_vel[3]) {
  if (!h || !q || !ang_vel) return MCL_ERR_INVALID;
  if (h->init_stim && h->init_m2o) {
    double e[3];
    euler_from_quat(q, e);
    for (int k = 0; k < 3; ++k) h->vel_rot[k] = ang_vel[k];
    const double dt = stamp - h->t_stim_prev;
    for (int k = 0; k < 3; ++k) h->rot[k] = h->rot[k] + h->vel_rot[k] * dt;
    h->t_stim_prev = stamp;
    // (:290-291 wraps a loop variable, not rot_t: yaw stays unwrapped)
    h->rot[0] = e[0];
    h->rot[1] = e[1];
  } else {
    h->t_stim_prev = stamp;
    h->init_stim = true;
  }
  return MCL_OK;
}

int mcl_dr_dvl(mcl_dr* h, double stamp, const double vel[3]) {
  if (!h || !vel) return MCL_ERR_INVALID;
  for (int k = 0; k < 3; ++k) h->dvl_vel[k] = vel[k];
  if (h->dvl_on) {
    h->t_dvl_prev = stamp;
  } else {
    h->t_dvl_prev = stamp;
    h->t_now = stamp;
    h->dvl_on = true;
  }
  return MCL_OK;
}

int mcl_dr_depth(mcl_dr* h, double z) {
  if (!h) return MCL_ERR_INVALID;
  if (h->depth_meas) h->base_depth = z + h->b2p[0] * std::sin(h->rot[1]);
  return MCL_OK;
}

int mcl_dr_thrust_cmd(mcl_dr* h, double thruster_horizontal_radians) {
  if (!h) return MCL_ERR_INVALID;
  h->thrust_cmd = thruster_horizontal_radians;
  return MCL_OK;
}

int mcl_dr_thrust(mcl_dr* h, double rpm1, double rpm2) {
  if (!h) return MCL_ERR_INVALID;
  h->u_dr = dr_detail::clip(-h->thrust_cmd, -7 * dr_detail::kPi / 180, 7 * dr_detail::kPi / 180);
  h->u_rpm = rpm1 + rpm2;
  return MCL_OK;
}

int mcl_dr_tick(mcl_dr* h, mcl_dr_odom* out) {
  if (!h || !out) return MCL_ERR_INVALID;
  std::memset(out, 0, sizeof(*out));
  out->q[3] = 1.0;
  out->t_now = h->t_now;
  if (!(h->init_m2o && h->init_stim)) return MCL_OK;
  double pose[6] = {h->pos[0], h->pos[1], h->pos[2], h->rot[0], h->rot[1], h->rot[2]};
  double lin[3] = {0.0, 0.0, 0.0};
  if (h->dvl_on) {
    // rows 0-1 of fullRotation (:257-270; its rot_y has a malformed third row, which only reaches row 2)
    const double cr = std::cos(pose[3]), sr = std::sin(pose[3]), cp = std::cos(pose[4]), sp = std::sin(pose[4]);
    const double cy = std::cos(pose[5]), sy = std::sin(pose[5]);
    // M1 = rot_y' * rot_x, rot_y' = [[cp,0,sp],[0,1,0],[-sp,cp,0]]
    const double M1[9] = {cp, sp * sr, sp * cr, 0.0, cr, -sr, -sp, cp * cr, -(cp * sr)};
    const double R0[3] = {cy * M1[0] + -sy * M1[3], cy * M1[1] + -sy * M1[4], cy * M1[2] + -sy * M1[5]};
    const double R1[3] = {sy * M1[0] + cy * M1[3], sy * M1[1] + cy * M1[4], sy * M1[2] + cy * M1[5]};
    const bool fresh = h->t_now - h->t_dvl_prev < h->dvl_period;
    if (fresh && std::fabs(h->dvl_vel[1]) < 0.2 && std::fabs(h->dvl_vel[0]) < 1.5 && h->dvl_vel[0] > -0.1) {
      for (int k = 0; k < 3; ++k) lin[k] = h->dvl_vel[k];
      out->used_dvl = 1;
    } else {
      // thrust model: the node integrates acceleration * dr_period as if it were a velocity (:193-198)
      double acc[3];
      dr_detail::sam_motion(h->u_rpm, h->u_dr, acc);
      lin[0] = acc[0] * h->dr_period;
      lin[1] = -acc[1] * h->dr_period;
      lin[2] = 0.0 * h->dr_period;
    }
    const double s[3] = {lin[0] * h->dr_period, lin[1] * h->dr_period, lin[2] * h->dr_period};
    pose[0] += (R0[0] * s[0] + R0[1] * s[1]) + R0[2] * s[2];
    pose[1] += (R1[0] * s[0] + R1[1] * s[1]) + R1[2] * s[2];
  }
  pose[2] = h->base_depth;
  out->published = 1;
  for (int k = 0; k < 3; ++k) {
    out->pos[k] = pose[k];
    out->rpy[k] = pose[3 + k];
    out->lin_vel[k] = lin[k];
    out->ang_vel[k] = h->vel_rot[k];
  }
  dr_detail::quat_from_euler(pose[3], pose[4], pose[5], out->q);
  h->t_now += h->dr_period;
  out->t_now = h->t_now;
  for (int k = 0; k < 3; ++k) h->pos[k] = pose[k];
  return MCL_OK;
}

int mcl_dr_to_odom(const mcl_dr_odom* in, double stamp, mcl_odom* out) {
  if (!in || !out) return MCL_ERR_INVALID;
  if (!in->published) return MCL_ERR_INVALID;
  out->stamp = stamp;
  for (int k = 0; k < 3; ++k) out->v[k] = in->lin_vel[k];
  out->w_z = in->ang_vel[2];
  for (int k = 0; k < 4; ++k) out->q[k] = in->q[k];
  out->z = in->pos[2];
  return MCL_OK;
}

}  // extern "C"
