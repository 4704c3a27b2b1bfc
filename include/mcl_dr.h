/* mcl_dr.h -- C ABI of the dead-reckoning integrator that produces the particle filter's
 * Odometry input from raw IMU / DVL / depth / thruster streams.
 *
 * Replaces the callbacks of the reference node sam_dead_reckoning/scripts/dr_node.py (class
 * VehicleDR) one for one, so a recorded stream can be replayed without ROS:
 *
 *   mcl_dr_heading      sbg_cb          dr_node.py:251-254
 *   mcl_dr_gps          gps_cb          dr_node.py:108-161  (utm->map of the fix stays with tf)
 *   mcl_dr_imu          stim_cb         dr_node.py:273-302
 *   mcl_dr_dvl          dvl_cb          dr_node.py:305-336
 *   mcl_dr_depth        depth_cb        dr_node.py:244-248
 *   mcl_dr_thrust_cmd   thrust_cmd_cb   dr_node.py:104-105
 *   mcl_dr_thrust       thrust_cb       dr_node.py:238-241
 *   mcl_dr_tick         dr_timer        dr_node.py:165-236  (DVL plausibility gates :179-182,
 *                                        thrust motion model sam_mm.py:30-120, fullRotation :257-270)
 *
 * Host-only, sequential by nature (one message at a time): no GPU work here.  Same conventions as
 * mcl.h: every call returns an int status (0 = OK, negative = mcl_status), one handle is owned by
 * one thread at a time, quaternions are (x, y, z, w), Euler angles static-xyz.
 * The reference's quirks are kept and documented in DESIGN.md 5d (yaw is never wrapped -- the wrap
 * loop at :290-291 assigns its loop variable; the motion-model branch integrates acceleration * dt
 * as a velocity). */
#ifndef MCL_DR_H
#define MCL_DR_H
#include "mcl.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mcl_dr mcl_dr;

typedef struct mcl_dr_config {
  double dvl_period; /* ~dvl_period, default 0.2  (dr_node.py:34): a DVL sample older than this is not used */
  double dr_period;  /* ~dr_period,  default 0.02 (dr_node.py:35): timer period = integration step */
} mcl_dr_config;

/* What one timer tick publishes (nav_msgs/Odometry fields, dr_node.py:207-222). */
typedef struct mcl_dr_odom {
  int32_t published; /* 0 until both the map->odom transform and the first IMU sample exist (:167) */
  int32_t used_dvl;  /* 1: DVL velocity integrated, 0: thrust motion model or no DVL yet */
  double t_now;      /* the node's internal clock after the tick (:233) */
  double pos[3];     /* pose.pose.position */
  double q[4];       /* pose.pose.orientation = quaternion_from_euler(roll, pitch, yaw) */
  double rpy[3];
  double lin_vel[3]; /* twist.twist.linear (body frame) */
  double ang_vel[3]; /* twist.twist.angular */
} mcl_dr_odom;

int mcl_dr_create(const mcl_dr_config* cfg, mcl_dr** out);
void mcl_dr_destroy(mcl_dr* h);

int mcl_dr_heading(mcl_dr* h, const double q[4]);
/* gx_map, gy_map: the fix already transformed utm -> map.  have_pressure_tf / b2p_trans: whether the
 * base_link -> pressure_link transform exists and its translation (:150-161).  *initialised = 1 when
 * this call fixed the map -> odom transform (then m2o_t / m2o_q receive it, :134-142); later calls
 * are ignored like the unregistered subscriber (:144). */
int mcl_dr_gps(mcl_dr* h, double gx_map, double gy_map, int have_pressure_tf, const double b2p_trans[3],
               int* initialised, double m2o_t[3], double m2o_q[4]);
int mcl_dr_imu(mcl_dr* h, double stamp, const double q[4], const double ang_vel[3]);
int mcl_dr_dvl(mcl_dr* h, double stamp, const double vel[3]);
int mcl_dr_depth(mcl_dr* h, double z);
int mcl_dr_thrust_cmd(mcl_dr* h, double thruster_horizontal_radians);
int mcl_dr_thrust(mcl_dr* h, double rpm1, double rpm2);
int mcl_dr_tick(mcl_dr* h, mcl_dr_odom* out);

/* The fields auv_pf's odom_callback consumes (auv_particle.py:45-70) from a published tick. */
int mcl_dr_to_odom(const mcl_dr_odom* in, double stamp, mcl_odom* out);

#ifdef __cplusplus
}
#endif
#endif
